"""BASELINE.json configs[0] / configs[1] shapes on the MI355X: the 320x512 model (`configs/inference_pvd_512.yaml`) at latent
16x40x64 - 16 frames take the PER-FRAME image-token branch of the denoiser (reference openaimodel3d.py:556-560: l_context ==
77 + t * 16), at the 320x512 graph's own level sizes (40x64, 20x32, 10x16, 5x8: 5x8 = 40 tokens at the deepest level).

  * `UNetModel.forward` (openaimodel3d.py:548-603) against the REFERENCE'S OWN CODE (oracle/_ref bytecode, fp32, on this GPU)
    and against the oracle restatement;
  * `python inference.py --renderings ...`'s own entry function `inference.main` at full width (1.44 B-parameter UNet, ViT-H
    towers, Resampler, VAE): YAML -> instantiate_from_config -> strict load of a 10 GB Lightning checkpoint -> setup_diffusion ->
    run_diffusion -> image_guided_synthesis (5 DDIM steps, CFG 7.5, rescale 0.7, eta 1, 320x512x16) -> diffusion0.pt, against
    the REFERENCE'S OWN `image_guided_synthesis` (utils/diffusion_utils.py:117-201) + `DDIMSampler` + `VIPLatentDiffusion` +
    `UNetModel` + `AutoencoderKL` in fp32 on the same weights, inputs and Gaussian draws (oracle.weights.NamedRandn replaces
    torch.randn inside run_diffusion on both sides; the condition encoders - open_clip / kornia are not in the image - are the
    product's on both sides, so the cross-attention context is identical and what is compared is the denoising path);
  * the same through the multi-condition sampler (`--multiple_cond_cfg --cfg_img 3`, ddim_multiplecond.py:220-236) at 25x40x64,
    BASELINE configs[1]'s latent (the shared image-token branch, 3 evaluations per step).

Stated tolerance: forward rel-L2 <= 5e-3; decoded clip rel-L2 <= 3e-2 and PSNR >= 30 dB after 5 eta = 1 steps.  The ASSERTIONS sit at
~2x the values measured on the MI355X (4.4e-3 / 58.7 dB, 3.6e-3 / 60.7 dB), not at the stated tolerance: a regression must fail them.
"""
import os

import pytest
import torch

from oracle import lvdm_oracle as O
from oracle.weights import NamedRandn
from tests.util import psnr, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML_512 = os.path.join(ROOT, "configs", "inference_pvd_512.yaml")


def test_unet_forward_at_config1_latent_16x40x64_vs_the_reference_code():
    from oracle import ref_runner as R
    from tests.test_fullconfig_gpu import FWD_TOL, _inputs, _model
    model, params = _model("inference_pvd_512.yaml")
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    T, h, w = 16, 40, 64
    x, ctx = _inputs(T, h, w, seed=77)
    assert ctx.shape[1] == 77 + T * 16                    # the per-frame image-token branch
    ts, fs = torch.tensor([799], device=DEV), torch.tensor([10], device=DEV)
    sd = {k: v.detach() for k, v in unet.state_dict().items()}
    with torch.no_grad():
        y = unet(x, ts, context=ctx, fs=fs)
        got = O.unet_forward(sd, hp, x, ts, ctx, fs)
    assert y.shape == (1, 4, T, h, w) and torch.isfinite(y).all()
    e_oracle_hip = rel_l2(y, got)
    msg = f"\n[config 1 latent 16x40x64, inference_pvd_512.yaml] HIP forward vs fp32 oracle: {e_oracle_hip:.3e}"
    if R.available():
        with torch.device("meta"):
            ref_unet = R.reference_unet(hp)
        ref_unet.load_state_dict(sd, strict=True, assign=True)
        with torch.no_grad():
            want = ref_unet(x, ts, context=ctx, fs=fs)
        e_ref_oracle, e_ref_hip = rel_l2(got, want), rel_l2(y, want)
        msg += f";  oracle vs the reference's UNetModel (oracle/_ref, fp32): {e_ref_oracle:.2e};  HIP vs the reference: {e_ref_hip:.3e}"
        del ref_unet
        assert e_ref_oracle <= 2e-5 and e_ref_hip <= FWD_TOL, msg
    print(msg)
    torch.cuda.empty_cache()
    assert e_oracle_hip <= FWD_TOL, msg


class _PatchedRandn:
    """torch.randn -> NamedRandn(prefix) for the duration of a with-block."""

    def __init__(self, prefix):
        self.fake = NamedRandn(prefix)

    def __enter__(self):
        self.real, torch.randn = torch.randn, self.fake
        return self.fake

    def __exit__(self, *exc):
        torch.randn = self.real


@pytest.fixture(scope="module")
def full_512(tmp_path_factory):
    """The full 320x512 model with its conditioners (2.6 B parameters), synthetic weights, written out as the Lightning checkpoint
    `inference.py --ckpt_path` loads; plus the reference's VIPLatentDiffusion around the same UNet / VAE tensors with the
    product's condition encoders attached."""
    from oracle import ref_runner as R
    if not R.available():
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    from viewcrafter_amd.config import load_yaml
    tmp = tmp_path_factory.mktemp("cfg1")
    m = build_diffusion_model(YAML_512, device=DEV, conditioners="config")
    randomize_parameters(m, seed=7)
    ckpt = str(tmp / "model.ckpt")
    torch.save({"state_dict": {k: v.detach().float().cpu() for k, v in m.state_dict().items()}, "global_step": 0}, ckpt)
    params = load_yaml(YAML_512)["model"]["params"]
    unet_sd = {k: v.detach() for k, v in m.model.diffusion_model.state_dict().items()}
    vae_sd = {k: v.detach() for k, v in m.first_stage_model.state_dict().items()}
    ref = R.reference_diffusion(params, unet_sd, vae_sd, DEV)
    ref.cond_stage_model, ref.embedder, ref.image_proj_model = m.cond_stage_model, m.embedder, m.image_proj_model
    ref.perframe_ae = False          # ONE posterior draw [(b t), 4, h, w] like the product's batched encode (tests/golden/gen_golden.py::gen_igs does the same)
    yield dict(model=m, ref=ref, ckpt=ckpt, tmp=tmp, params=params)
    os.remove(ckpt)


def _reference_clip(ref_model, renderings, opts, noise_shape, prefix, **kw):
    """What the reference's ViewCrafter.run_diffusion (viewcrafter.py:93-106) does, with the reference's own image_guided_synthesis."""
    from utils.diffusion_utils import image_guided_synthesis as ref_igs          # oracle/_ref (sys.path set by ref_runner)
    videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0).to(DEV)
    with _PatchedRandn(prefix) as fake, torch.no_grad():
        out = ref_igs(ref_model, [opts.prompt], videos, noise_shape, opts.n_samples, opts.ddim_steps, opts.ddim_eta,
                      opts.unconditional_guidance_scale, kw.get("cfg_img", opts.cfg_img), opts.frame_stride, opts.text_input,
                      kw.get("multiple_cond_cfg", opts.multiple_cond_cfg), opts.timestep_spacing, opts.guidance_rescale, [0])
    return torch.clamp(out[0][0].permute(1, 2, 3, 0), -1., 1.), fake.calls


def test_inference_main_five_steps_at_config1_vs_the_reference_driver(full_512, monkeypatch):
    import inference
    import viewcrafter
    from configs.infer_config import get_parser
    T, H, W = 16, 320, 512
    tmp = full_512["tmp"]
    g = torch.Generator().manual_seed(11)
    renderings = torch.rand(T, H, W, 3, generator=g)
    rpath = str(tmp / "render.pt")
    torch.save(renderings, rpath)
    argv = ["--renderings", rpath, "--config", YAML_512, "--ckpt_path", full_512["ckpt"], "--out_dir", str(tmp / "out"), "--exp_name", "e",
            "--device", "cuda:0", "--ddim_steps", "5", "--video_length", str(T), "--height", str(H), "--width", str(W), "--prompt", "",
            "--seed", "123"]
    calls = {}
    real_run = viewcrafter.ViewCrafter.run_diffusion

    def run_with_named_draws(self, r):
        with _PatchedRandn("cfg1_cli") as fake:
            out = real_run(self, r)
        calls["ours"] = fake.calls
        return out
    monkeypatch.setattr(viewcrafter.ViewCrafter, "run_diffusion", run_with_named_draws)
    out = inference.main(argv)                           # the command line's own entry function, in this process
    monkeypatch.undo()
    saved = torch.load(os.path.join(str(tmp / "out"), "e", "diffusion0.pt"))
    assert tuple(saved.shape) == (T, H, W, 3) and torch.equal(saved, out.cpu()) and torch.isfinite(saved).all()
    opts = get_parser().parse_args(argv)
    want, ref_calls = _reference_clip(full_512["ref"], renderings, opts, [1, 4, T, H // 8, W // 8], "cfg1_cli")
    assert calls["ours"] == ref_calls, f"Gaussian draws differ from the reference's driver: {calls['ours']} vs {ref_calls}"
    e, p = rel_l2(saved, want), psnr(saved, want)
    print(f"\n[config 1: inference.main, 320x512x16, 5 DDIM steps, CFG 7.5, eta 1, full width] decoded clip vs the reference's own "
          f"image_guided_synthesis / DDIMSampler / UNetModel / AutoencoderKL (fp32, same draws): rel-L2 {e:.3e}, PSNR {p:.1f} dB; "
          f"{ref_calls} Gaussian draws on both sides")
    torch.cuda.empty_cache()
    assert e <= 9e-3 and p >= 52.0        # measured 4.4e-3 / 58.7 dB (profiles/r04p_gpu_pytest.log): ~2x, so that a 2x regression fails


def test_multicond_sampler_at_25x40x64_vs_the_reference_driver(full_512):
    """--multiple_cond_cfg --cfg_img 3 (ddim_multiplecond.py:220-236: three evaluations per step, the un-fixed scale-array
    indexing of :33) through image_guided_synthesis at BASELINE configs[1]'s latent, 3 steps."""
    from configs.infer_config import get_parser
    from viewcrafter_amd.utils.diffusion_utils import image_guided_synthesis
    T, H, W = 25, 320, 512
    g = torch.Generator().manual_seed(12)
    renderings = torch.rand(T, H, W, 3, generator=g)
    opts = get_parser().parse_args(["--ddim_steps", "3", "--prompt", ""])
    noise_shape = [1, 4, T, H // 8, W // 8]
    videos = (renderings * 2. - 1.).permute(3, 0, 1, 2).unsqueeze(0).to(DEV)
    with _PatchedRandn("cfg1_multi") as fake, torch.no_grad():
        out = image_guided_synthesis(full_512["model"], [""], videos, noise_shape, 1, 3, opts.ddim_eta, opts.unconditional_guidance_scale,
                                     3.0, opts.frame_stride, False, True, opts.timestep_spacing, opts.guidance_rescale, [0])
    ours = torch.clamp(out[0][0].permute(1, 2, 3, 0), -1., 1.)
    want, ref_calls = _reference_clip(full_512["ref"], renderings, opts, noise_shape, "cfg1_multi", multiple_cond_cfg=True, cfg_img=3.0)
    assert fake.calls == ref_calls
    e, p = rel_l2(ours, want), psnr(ours, want)
    print(f"\n[multi-condition CFG, 320x512x25, 3 steps, cfg 7.5 / cfg_img 3, full width] decoded clip vs the reference's own driver + "
          f"ddim_multiplecond sampler (fp32, same draws): rel-L2 {e:.3e}, PSNR {p:.1f} dB")
    torch.cuda.empty_cache()
    assert e <= 7.5e-3 and p >= 54.0      # measured 3.6e-3 / 60.7 dB: ~2x
