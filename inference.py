"""`python inference.py ...` - the reference's entry point (its inference.py:8-31) with the diffusion leg on MI355X.

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 inference.py ...` starts one process
per GPU (RCCL over xGMI); independent trajectories (`--renderings a.pt,b.pt,...`, the clips of `--mode sparse_view_interp`)
are sharded over the ranks with no collective inside the DDIM loop (viewcrafter_amd/parallel.py).
"""
import os
from datetime import datetime

from configs.infer_config import get_parser


def main(argv=None):
    opts = get_parser().parse_args(argv)
    from viewcrafter_amd import parallel
    rank, world = parallel.init_distributed()
    if world > 1 and str(opts.device).startswith("cuda"):
        opts.device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    from viewcrafter import ViewCrafter
    if opts.exp_name is None:
        stamp = datetime.now().strftime("%Y%m%d_%H%M")
        opts.exp_name = f"{stamp}_{os.path.splitext(os.path.basename(opts.image_dir))[0]}"
    # rank 0 owns the result directory; the other ranks' side products of the reference's geometry stage go next to it
    opts.save_dir = os.path.join(opts.out_dir, opts.exp_name if rank == 0 else f"{opts.exp_name}.rank{rank}")
    os.makedirs(opts.save_dir, exist_ok=True)
    ok = False
    try:
        pvd = ViewCrafter(opts)
        if opts.renderings is not None:
            out = pvd.nvs_from_renderings(opts.renderings)
        elif opts.mode in ("single_view_target", "single_view_txt"):
            out = pvd.nvs_single_view()
        elif opts.mode == "single_view_eval":
            out = pvd.nvs_single_view_eval()
        elif opts.mode == "sparse_view_interp":
            out = pvd.nvs_sparse_view_interp()
        else:
            raise KeyError(f"Invalid Mode: {opts.mode}")
        ok = True
        return out
    finally:
        parallel.shutdown(barrier=ok)       # a rank that failed must not make the others wait in a barrier


if __name__ == "__main__":
    main()
