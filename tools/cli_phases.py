"""Where the wall-clock of `python inference.py --renderings ...` goes at full size (evidence run; see tools/cli_fullsize.py)."""
import json, os, sys, tempfile, time
t_start = time.perf_counter()
import torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); os.chdir(ROOT)
from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
tmp = tempfile.mkdtemp(dir="/tmp")
cfg = os.path.join(ROOT, "configs", "inference_pvd_1024.yaml")
m = build_diffusion_model(cfg, device="cuda", conditioners="config"); randomize_parameters(m, seed=7)
torch.save({"state_dict": {k: v.detach().float().cpu() for k, v in m.state_dict().items()}}, os.path.join(tmp, "model.ckpt"))
del m; torch.cuda.empty_cache()
torch.save(torch.rand(25, 576, 1024, 3), os.path.join(tmp, "render.pt"))
from configs.infer_config import get_parser
import viewcrafter
from viewcrafter_amd.utils.video_io import save_video
opts = get_parser().parse_args(["--renderings", os.path.join(tmp, "render.pt"), "--ckpt_path", os.path.join(tmp, "model.ckpt"), "--config", cfg,
                                "--out_dir", os.path.join(tmp, "out"), "--exp_name", "e", "--prompt", "", "--seed", "123"])
opts.save_dir = os.path.join(tmp, "out", "e"); os.makedirs(opts.save_dir)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"{label:40s} {time.perf_counter()-t0:7.2f} s", flush=True); return r
print(f"{'import torch + package':40s} {time.perf_counter()-t_start:7.2f} s (includes writing the synthetic checkpoint)")
vc = T("ViewCrafter(opts) = build + ckpt load", lambda: viewcrafter.ViewCrafter(opts))
clip = T("torch.load(render.pt)", lambda: torch.load(os.path.join(tmp, "render.pt")))
o1 = T("run_diffusion #1 (first call: packing)", lambda: vc.run_diffusion(clip))
o2 = T("run_diffusion #2", lambda: vc.run_diffusion(clip))
T("torch.save(out) + save_video", lambda: (torch.save(o2.cpu(), os.path.join(opts.save_dir, "d.pt")), save_video((o2 + 1) / 2, os.path.join(opts.save_dir, "d.mp4"), fps=10, value_range=(0.0, 1.0))))
import subprocess; subprocess.run(["rm", "-rf", tmp])
