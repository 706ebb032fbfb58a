"""Sanity + timing of the three-conditioning sampler (ddim_multiplecond) at 25x72x128; first pass is warm-up."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import synth_conditioning
from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
from viewcrafter_amd.lvdm.models.samplers.ddim_multiplecond import DDIMSampler
model = build_diffusion_model("/root/repo/configs/inference_pvd_1024.yaml", device="cuda", conditioners="identity")
randomize_parameters(model, seed=0)
T, h, w = 25, 72, 128
x, cond, uc = synth_conditioning(T, h, w, "cuda")
uc2 = {"c_crossattn": [torch.cat([uc["c_crossattn"][0][:, :77], cond["c_crossattn"][0][:, 77:]], 1)], "c_concat": cond["c_concat"]}
fs = torch.tensor([10], device="cuda")
s = DDIMSampler(model)
for share in (True, True, False, True, False):
    s.share_cfg_prefix = share
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        out, _ = s.sample(S=50, conditioning=cond, batch_size=1, shape=[4, T, h, w], verbose=False, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uc, eta=1.0, cfg_img=3.0, mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing",
                          guidance_rescale=0.7, x_T=x, unconditional_conditioning_img_nonetext=uc2, timesteps=4)
    torch.cuda.synchronize()
    print("multi-cond (3 conditionings), shared prefix", share, "finite", bool(torch.isfinite(out).all()), f"{(time.perf_counter()-t0):.2f} s for the steps run")
