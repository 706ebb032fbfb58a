"""`python inference.py ...` — the reference's entry point (its inference.py:8-31) with the diffusion leg on MI355X."""
import os
from datetime import datetime

from configs.infer_config import get_parser
from viewcrafter import ViewCrafter

if __name__ == "__main__":
    opts = get_parser().parse_args()
    if opts.exp_name is None:
        stamp = datetime.now().strftime("%Y%m%d_%H%M")
        opts.exp_name = f"{stamp}_{os.path.splitext(os.path.basename(opts.image_dir))[0]}"
    opts.save_dir = os.path.join(opts.out_dir, opts.exp_name)
    os.makedirs(opts.save_dir, exist_ok=True)
    pvd = ViewCrafter(opts)
    if opts.renderings is not None:
        pvd.nvs_from_renderings(opts.renderings)
    elif opts.mode in ("single_view_target", "single_view_txt"):
        pvd.nvs_single_view()
    elif opts.mode == "single_view_eval":
        pvd.nvs_single_view_eval()
    elif opts.mode == "sparse_view_interp":
        pvd.nvs_sparse_view_interp()
    else:
        raise KeyError(f"Invalid Mode: {opts.mode}")
