"""`python inference.py` end to end at the real size (evidence run, ~12 GB scratch): a synthetic 10.4 GB checkpoint with the key
set of a ViewCrafter_25 checkpoint and a saved point-cloud render clip [25, 576, 1024, 3] in [0, 1] go through the command line -
configs/infer_config.py flags -> ViewCrafter.setup_diffusion (YAML -> instantiate_from_config -> strict checkpoint load) ->
run_diffusion -> image_guided_synthesis (CLIP towers, Resampler, VAE encode, 50 DDIM steps, CFG 7.5 / rescale 0.7, VAE decode)
-> diffusion0.pt + video file.  The empty prompt is used because CLIP's BPE vocabulary is not available offline."""
import json, os, subprocess, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters   # noqa: E402

tmp = tempfile.mkdtemp(dir=os.environ.get("VCX_SCRATCH", "/tmp"))
cfg = os.path.join(ROOT, "configs", "inference_pvd_1024.yaml")
m = build_diffusion_model(cfg, device="cuda", conditioners="config")
randomize_parameters(m, seed=7)
torch.save({"state_dict": {k: v.detach().float().cpu() for k, v in m.state_dict().items()}}, os.path.join(tmp, "model.ckpt"))
del m
torch.cuda.empty_cache()
g = torch.Generator().manual_seed(1)
torch.save(torch.rand(25, 576, 1024, 3, generator=g), os.path.join(tmp, "render.pt"))
cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--renderings", os.path.join(tmp, "render.pt"), "--ckpt_path", os.path.join(tmp, "model.ckpt"),
       "--config", cfg, "--out_dir", os.path.join(tmp, "out"), "--exp_name", "e", "--prompt", "", "--seed", "123"]
t0 = time.perf_counter()
r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
wall = time.perf_counter() - t0
assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
out = torch.load(os.path.join(tmp, "out", "e", "diffusion0.pt"))
files = sorted(os.listdir(os.path.join(tmp, "out", "e")))
assert list(out.shape) == [25, 576, 1024, 3] and torch.isfinite(out).all() and float(out.min()) >= -1.0 and float(out.max()) <= 1.0
print(json.dumps({"command": "python inference.py --renderings render.pt --ckpt_path model.ckpt --config configs/inference_pvd_1024.yaml --prompt ''",
                  "process_wall_s": round(wall, 1), "output": list(out.shape), "range": [round(float(out.min()), 3), round(float(out.max()), 3)],
                  "std": round(float(out.std()), 4), "files": files}))
subprocess.run(["rm", "-rf", tmp])
