"""Wall-clock of the whole GPU part of one video at full size, all of it on libvcx, through the reference's own entry point
`image_guided_synthesis` (utils/diffusion_utils.py:117-201): OpenCLIP image tower + Resampler on the condition frame, OpenCLIP
text tower on the empty prompt, VAE encode of the condition clip, the 50-step DDIM loop (CFG 7.5, rescale 0.7, eta 1.0) and the
per-frame VAE decode.  Synthetic weights and inputs (no checkpoints offline); DUSt3R and the point-cloud render are not part of it.

    python tools/video_e2e.py [--workload ViewCrafter_25_576x1024x25] [--steps 50]
"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS          # noqa: E402
from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters   # noqa: E402
from viewcrafter_amd.utils.diffusion_utils import get_latent_z, image_guided_synthesis   # noqa: E402


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25", choices=list(WORKLOADS))
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    cfg, T, h, w = WORKLOADS[args.workload]
    dev = "cuda"
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg), device=dev, conditioners="config")   # everything native
    randomize_parameters(model)
    g = torch.Generator().manual_seed(123)
    videos = (torch.rand(1, 3, T, h * 8, w * 8, generator=g) * 2 - 1).to(dev)          # point-cloud renders in [-1, 1]
    noise_shape = [1, 4, T, h, w]
    kw = dict(n_samples=1, ddim_eta=1.0, unconditional_guidance_scale=7.5, cfg_img=None, fs=10, text_input=False,
              multiple_cond_cfg=False, timestep_spacing="uniform_trailing", guidance_rescale=0.7, condition_index=[0])
    with torch.no_grad():
        image_guided_synthesis(model, [""], videos, noise_shape, ddim_steps=2, **kw)     # warms packing / lazy init
        img = videos[:, :, 0]
        tokens, t_clip = timed(lambda: model.embedder(img))
        _, t_proj = timed(lambda: model.image_proj_model(tokens))
        _, t_text = timed(lambda: model.get_learned_conditioning([""]))
        _, t_enc = timed(lambda: get_latent_z(model, videos))
        z = torch.randn(1, 4, T, h, w, device=dev)
        _, t_dec = timed(lambda: model.decode_first_stage(z))
        out, t_all = timed(lambda: image_guided_synthesis(model, [""], videos, noise_shape, ddim_steps=args.steps, **kw))
    assert torch.isfinite(out).all() and list(out.shape) == [1, 1, 3, T, h * 8, w * 8]
    res = {"workload": args.workload, "ddim_steps": args.steps, "image_guided_synthesis_s": round(t_all, 3),
           "parts": {"clip_image_tower_s": round(t_clip, 4), "resampler_s": round(t_proj, 4), "clip_text_tower_s": round(t_text, 4),
                     "vae_encode_s": round(t_enc, 3), "vae_decode_s": round(t_dec, 3),
                     "ddim_loop_s_by_difference": round(t_all - 2 * (t_clip + t_proj + t_text) - t_enc - t_dec, 3)},
           "frames": list(out.shape)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
