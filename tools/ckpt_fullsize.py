"""Full-size checkpoint ingestion through the product's own path (measurement / evidence, needs ~25 GB of scratch disk):
a synthetic Lightning-layout checkpoint with every tensor of the 576x1024 model - UNet 1.44 B, VAE, both OpenCLIP towers,
Resampler, schedule buffers: the key set a real ViewCrafter_25 checkpoint has - is written with torch.save, then loaded by
`build_diffusion_model(config, ckpt_path=...)` = instantiate_from_config + load_model_checkpoint (strict=True,
utils/diffusion_utils.py:83-108), and one image_guided_synthesis call (2 DDIM steps) runs on the loaded weights.
Also repeats the load from the DeepSpeed layout ({'module': {'_forward_module.' + key}})."""
import json, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters   # noqa: E402
from viewcrafter_amd.utils.diffusion_utils import image_guided_synthesis   # noqa: E402

cfg = os.path.join(ROOT, "configs", "inference_pvd_1024.yaml")
dev = "cuda"
T, h, w = 25, 72, 128
res = {}
src = build_diffusion_model(cfg, device=dev, conditioners="config")
randomize_parameters(src, seed=7)
sd = {k: v.detach().float().cpu() for k, v in src.state_dict().items()}
res["entries"] = len(sd)
res["params_M"] = round(sum(v.numel() for v in sd.values()) / 1e6, 1)
tmp = tempfile.mkdtemp(dir=os.environ.get("VCX_SCRATCH", "/tmp"))
path = os.path.join(tmp, "model.ckpt")
t0 = time.perf_counter()
torch.save({"state_dict": sd, "epoch": 1, "global_step": 1000}, path)
res["ckpt_GB"] = round(os.path.getsize(path) / 1e9, 2)
res["torch_save_s"] = round(time.perf_counter() - t0, 1)
probe = ["model.diffusion_model.input_blocks.1.0.temopral_conv.conv2.3.weight", "first_stage_model.decoder.mid.attn_1.q.weight",
         "image_proj_model.latents", "model.diffusion_model.init_attn.0.proj_in.weight"]
want = {k: sd[k].clone() for k in probe}
g = torch.Generator().manual_seed(123)
videos = (torch.rand(1, 3, T, h * 8, w * 8, generator=g) * 2 - 1).to(dev)
kw = dict(n_samples=1, ddim_steps=2, ddim_eta=0.0, unconditional_guidance_scale=7.5, cfg_img=None, fs=10, text_input=False,
          multiple_cond_cfg=False, timestep_spacing="uniform_trailing", guidance_rescale=0.7, condition_index=[0])
with torch.no_grad():
    torch.manual_seed(5)
    ref_out = image_guided_synthesis(src, [""], videos, [1, 4, T, h, w], **kw).cpu()
del src
torch.cuda.empty_cache()
for layout in ("lightning", "deepspeed"):
    if layout == "deepspeed":
        torch.save({"module": {"_forward_module." + k: v for k, v in sd.items()}}, path)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = build_diffusion_model(cfg, device=dev, ckpt_path=path, conditioners="config")
    torch.cuda.synchronize()
    res[f"{layout}_build_and_strict_load_s"] = round(time.perf_counter() - t0, 1)
    got = m.state_dict()
    assert all(torch.equal(got[k].float().cpu(), want[k]) for k in probe), layout
    with torch.no_grad():
        torch.manual_seed(5)
        out = image_guided_synthesis(m, [""], videos, [1, 4, T, h, w], **kw).cpu()
    assert torch.equal(out, ref_out), f"{layout}: output of the loaded model differs from the model the checkpoint was written from"
    res[f"{layout}_output_equal"] = True
    del m, got
    torch.cuda.empty_cache()
os.remove(path); os.rmdir(tmp)
print(json.dumps(res))
