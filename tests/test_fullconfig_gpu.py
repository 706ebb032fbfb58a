"""Whole-forward / whole-trajectory parity at BASELINE.json's REAL configurations (SURVEY.md §7 tier T3).

The fp32 oracle (oracle/lvdm_oracle.py, pinned to reference-generated goldens on tiny shapes) is plain device-agnostic
PyTorch: on the 288 GB MI355X it runs the full 1.44 B-parameter graph at 25x72x128 - vanilla attention included, walked
in batch-head chunks - in seconds.  So the HIP path is compared with it directly at the shapes the benchmark is quoted
on, instead of inferring full-size numerics from the tiny graph:

  * `UNetModel.forward`  (reference openaimodel3d.py:548-603) at (T, h, w) = (25, 72, 128), (16, 72, 128), (25, 40, 64)
    - configs 4, 3, 2 of BASELINE.json; T = 16 takes the per-frame image-token branch (L = 77 + 16 T), T = 25 the shared one;
  * `decode_first_stage` -> `AutoencoderKL.decode` (ae_modules.py:539-578) of two frames at 72x128 -> 576x1024: the
    d = 512 AttnBlock at N = 9216 and the 128/256-channel 576x1024 activations;
  * a 50-step eta = 0 DDIM trajectory (ddim.py:137-281; CFG 7.5, guidance rescale 0.7, uniform_trailing, dynamic
    rescale, v-prediction) at (25, 40, 64), ours vs the oracle sampler driving the oracle UNet.

  * (round 3) 10-step eta = 0 trajectories at the two 576x1024 configurations themselves, (25, 72, 128) and (16, 72, 128);
    `encode_first_stage` (ddpm3d.py:621-644 -> Encoder.forward ae_modules.py:430-463) of two 576x1024 frames; and an
    fp16-RANGE stress test: the residual branches' last layers are scaled by a power of two until the ORACLE's residual
    stream reaches several thousand (real checkpoints are known to run the deep UNet levels and the VAE's 128/256-channel
    576x1024 blocks that hot), and the fp16 residual stream of the HIP path must stay finite and inside 2x the bound.

Stated fp16 tolerance (rel-L2 against fp32): forward <= 5e-3, decode / encode <= 8e-3, final latent <= 1e-2.  Weights are the
seeded synthetic ones of builder.randomize_parameters (no checkpoints offline); the oracle reads the very same fp32
tensors.  A per-block error table is printed for every forward (and is what to read first when a bound fails).
"""
import os

import pytest
import torch

from oracle import lvdm_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FWD_TOL, DEC_TOL, TRAJ_TOL = 5e-3, 8e-3, 7e-3      # trajectories: measured 2.3e-3 ... 3.5e-3 (stated tolerance 1e-2); ~2x the largest
PSNR_MIN = 53.0                                     # decoded frames: measured 59.4 ... 62.1 dB (stated 30 dB); 6 dB = 2x the error
_MODELS = {}


def _model(yaml_name):
    """One 1.44 B-parameter model per YAML, built once per session (both stay resident: 2 x 9 GB of 288)."""
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    from viewcrafter_amd.config import load_yaml
    if yaml_name not in _MODELS:
        path = os.path.join(ROOT, "configs", yaml_name)
        m = build_diffusion_model(path, device=DEV, conditioners="identity")
        randomize_parameters(m, seed=0)
        params = load_yaml(path)["model"]["params"]
        _MODELS[yaml_name] = (m, params)
    return _MODELS[yaml_name]


def _inputs(T, h, w, seed, B=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 8, T, h, w, generator=g)
    ctx = torch.randn(B, 77 + 256, 1024, generator=g)      # text tokens + 16 queries x video_length 16 Resampler tokens
    return x.to(DEV), ctx.to(DEV)


def _unet_taps(unet):
    """Forward hooks on the product UNet's blocks: outputs are channels-last fp16 [n, H, W, C] -> fp32 [n, C, H, W]."""
    taps, hooks = {}, []

    def add(name, mod):
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o)))
    for i, blk in enumerate(unet.input_blocks):
        if i > 0:
            add(f"input_blocks.{i}", blk)
    add("middle_block", unet.middle_block)
    for i, blk in enumerate(unet.output_blocks):
        add(f"output_blocks.{i}", blk)
    return taps, hooks


def _block_table(ours, ref):
    rows = []
    for name, r in ref.items():
        if name not in ours:
            continue
        o = ours[name].float().permute(0, 3, 1, 2)
        rows.append((name, float((o - r).norm() / (r.norm() + 1e-30)), float(r.abs().max())))
    return rows


@pytest.mark.parametrize("tag,yaml_name,T,h,w", [("ViewCrafter_25 576x1024x25", "inference_pvd_1024.yaml", 25, 72, 128),
                                                 ("ViewCrafter_16 576x1024x16", "inference_pvd_1024.yaml", 16, 72, 128),
                                                 ("ViewCrafter_25_512 320x512x25", "inference_pvd_512.yaml", 25, 40, 64)])
def test_unet_forward_at_real_config_vs_fp32_oracle_on_gpu(tag, yaml_name, T, h, w):
    model, params = _model(yaml_name)
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    x, ctx = _inputs(T, h, w, seed=1234 + T + h)
    ts = torch.tensor([599], device=DEV)
    fs = torch.tensor([10], device=DEV)
    taps, hooks = _unet_taps(unet)
    try:
        with torch.no_grad():
            y = unet(x, ts, context=ctx, fs=fs)
    finally:
        for hk in hooks:
            hk.remove()
    assert y.shape == (1, 4, T, h, w) and y.dtype == torch.float32 and torch.isfinite(y).all()
    sd = {k: v.detach() for k, v in unet.state_dict().items()}      # the same fp32 tensors, no copy
    ref_taps = {}
    with torch.no_grad():
        ref = O.unet_forward(sd, hp, x, ts, ctx, fs, taps=ref_taps)
    e = rel_l2(y, ref)
    rows = _block_table(taps, ref_taps)
    print(f"\n[{tag}] UNet forward rel-L2 vs fp32 oracle on the MI355X = {e:.3e}   (|ref| max {float(ref.abs().max()):.3f})")
    print("  per block (rel-L2, max |ref|): " + "  ".join(f"{n.replace('input_blocks', 'in').replace('output_blocks', 'out').replace('middle_block', 'mid')}:{r:.1e}/{m:.0f}"
                                                             for n, r, m in rows))
    fp16_headroom = max(m for _, _, m in rows)
    assert fp16_headroom < 3e4, f"residual stream reaches {fp16_headroom:.0f}: fp16 range at risk"
    del taps, ref_taps
    torch.cuda.empty_cache()
    assert e <= FWD_TOL, f"{tag}: {e:.3e} > {FWD_TOL:.0e}; worst blocks: {sorted(rows, key=lambda r: -r[1])[:4]}"


def test_vae_decode_576x1024_vs_fp32_oracle_on_gpu():
    """Two frames through decode_first_stage at the real size (d = 512 AttnBlock over 9216 tokens, 576x1024 activations);
    latents ~ N(0, 1) so that the decoder sees z / 0.18215 (std 5.5), the range real samples have."""
    model, params = _model("inference_pvd_1024.yaml")
    dd = dict(params["first_stage_config"]["params"]["ddconfig"])
    g = torch.Generator().manual_seed(77)
    z = torch.randn(1, 4, 2, 72, 128, generator=g).to(DEV)
    with torch.no_grad():
        out = model.decode_first_stage(z)
    assert out.shape == (1, 3, 2, 576, 1024) and torch.isfinite(out).all()
    sd = {k: v.detach() for k, v in model.first_stage_model.state_dict().items()}
    with torch.no_grad():
        ref = O.decode_first_stage(sd, dd, z, scale_factor=params["scale_factor"])
    e = rel_l2(out, ref)
    mse = float(((out - ref) ** 2).mean())
    peak = float(ref.abs().max())
    print(f"\n[VAE decode 2 x 576x1024] rel-L2 vs fp32 oracle on the MI355X = {e:.3e}; max |ref| {peak:.2f}, rmse {mse ** 0.5:.3e}")
    torch.cuda.empty_cache()
    assert e <= DEC_TOL


@pytest.mark.parametrize("tag,yaml_name,T,h,w,steps", [
    ("ViewCrafter_25_512 320x512x25", "inference_pvd_512.yaml", 25, 40, 64, 50),
    ("ViewCrafter_25 576x1024x25", "inference_pvd_1024.yaml", 25, 72, 128, 10),
    ("ViewCrafter_16 576x1024x16", "inference_pvd_1024.yaml", 16, 72, 128, 10)])
def test_ddim_trajectory_vs_fp32_oracle_on_gpu(tag, yaml_name, T, h, w, steps):
    """BASELINE configs 2 / 4 / 3 end to end: DDIM steps (eta = 0, injected x_T, CFG 7.5, guidance rescale 0.7, dynamic rescale,
    v-prediction; reference ddim.py:137-281) of the product sampler on the HIP UNet against the oracle sampler on the oracle
    UNet, same weights / conditioning; then both latents through the product VAE for a PSNR.  50 steps at 320x512, 10 at the
    two 576x1024 configurations (20 fp32 oracle forwards of 83 / 52 TFLOP each)."""
    from viewcrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    from tests.util import psnr
    model, params = _model(yaml_name)
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    g = torch.Generator().manual_seed(123)
    x_T = torch.randn(1, 4, T, h, w, generator=g).to(DEV)
    cat = (torch.randn(1, 4, T, h, w, generator=g) * 0.8).to(DEV)
    ctx = torch.randn(1, 77 + 256, 1024, generator=g).to(DEV)
    uctx = torch.randn(1, 77 + 256, 1024, generator=g).to(DEV)
    cond = {"c_crossattn": [ctx], "c_concat": [cat]}
    uc = {"c_crossattn": [uctx], "c_concat": [cat]}
    fs = torch.tensor([10], device=DEV)
    sampler = DDIMSampler(model)
    with torch.no_grad():
        ours, inter = sampler.sample(S=steps, conditioning=cond, batch_size=1, shape=[4, T, h, w], verbose=False,
                                     unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, cfg_img=None,
                                     mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                     x_T=x_T, log_every_t=max(1, steps // 5), unconditional_conditioning_img_nonetext=None)
    assert torch.isfinite(ours).all()
    sd = {k: v.detach() for k, v in unet.state_dict().items()}
    tables = O.diffusion_tables(params["timesteps"], params["linear_start"], params["linear_end"], params["rescale_betas_zero_snr"])
    scale_arr = O.dynamic_rescale_table(params["timesteps"], params["base_scale"])
    assert torch.allclose(scale_arr, model.scale_arr.cpu()) and torch.allclose(tables["alphas_cumprod"], model.alphas_cumprod.cpu())

    def apply_oracle(x, t, c):
        return O.unet_forward(sd, hp, torch.cat([x, c["c_concat"][0]], dim=1), t.to(DEV), c["c_crossattn"][0], fs)
    with torch.no_grad():
        ref, preds = O.ddim_sample(apply_oracle, tables, scale_arr, x_T, cond, uc, steps=steps, eta=0.0, cfg_scale=7.5,
                                   guidance_rescale=0.7, spacing="uniform_trailing", parameterization="v")
    e = rel_l2(ours, ref)
    e_first = rel_l2(inter["pred_x0"][1], preds[0])
    with torch.no_grad():
        p = psnr(model.decode_first_stage(ours[:, :, :2].contiguous()), model.decode_first_stage(ref[:, :, :2].contiguous()))
    print(f"\n[{tag}, {steps} DDIM steps, eta 0] final latent rel-L2 vs fp32 oracle trajectory = {e:.3e} "
          f"(first pred_x0 {e_first:.3e}); decoded frames PSNR {p:.1f} dB")
    torch.cuda.empty_cache()
    assert e <= TRAJ_TOL
    assert p >= PSNR_MIN


def test_vae_encode_576x1024_vs_fp32_oracle_on_gpu():
    """`encode_first_stage` (reference ddpm3d.py:621-644 -> AutoencoderKL.encode autoencoder.py:97-102 -> Encoder.forward
    ae_modules.py:430-463) of two 576x1024 frames: the posterior moments against the fp32 oracle, and the sampled, scaled
    latent against `scale_factor * (mean + std * noise)` of the oracle's moments with the same CPU noise stream
    (distributions.py:35-40 draws on the CPU)."""
    model, params = _model("inference_pvd_1024.yaml")
    dd = dict(params["first_stage_config"]["params"]["ddconfig"])
    g = torch.Generator().manual_seed(91)
    x = (torch.rand(1, 3, 2, 576, 1024, generator=g) * 2 - 1).to(DEV)         # frames in [-1, 1], like run_diffusion's input
    frames = x.permute(0, 2, 1, 3, 4).reshape(2, 3, 576, 1024)
    with torch.no_grad():
        post = model.first_stage_model.encode(frames)
        torch.manual_seed(4242)
        z = model.encode_first_stage(x)
    assert post.parameters.shape == (2, 8, 72, 128) and z.shape == (1, 4, 2, 72, 128) and torch.isfinite(z).all()
    sd = {k: v.detach() for k, v in model.first_stage_model.state_dict().items()}
    with torch.no_grad():
        ref = torch.cat([O.vae_encode_moments(sd, dd, frames[i:i + 1]) for i in range(2)], dim=0)
    e = rel_l2(post.parameters, ref)
    torch.manual_seed(4242)
    noise = torch.randn(2, 4, 72, 128).to(DEV)
    mean, logvar = ref.chunk(2, dim=1)
    z_ref = params["scale_factor"] * (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise)
    z_ref = z_ref.view(1, 2, 4, 72, 128).permute(0, 2, 1, 3, 4)
    ez = rel_l2(z, z_ref)
    print(f"\n[VAE encode 2 x 576x1024] moments rel-L2 vs fp32 oracle on the MI355X = {e:.3e}; sampled latent {ez:.3e}; "
          f"max |moments| {float(ref.abs().max()):.2f}")
    torch.cuda.empty_cache()
    assert e <= DEC_TOL and ez <= DEC_TOL


# ---------------------------------------------------------------------------------------------------------------------------
# fp16-range stress: the HIP path keeps its residual stream in fp16 (the reference's alternates fp16 / fp32 under autocast,
# its conv / linear OUTPUTS are fp16 as well).  Synthetic N(0, 1/fan_in) weights leave the stream at |x| < 40; real checkpoints
# do not.  Every residual branch ends in one layer (ResBlock out_layers.3, TemporalConvBlock conv4, the transformer blocks'
# to_out / ff.net.2; VAE ResnetBlock conv2, AttnBlock proj_out): scaling those by 2^k scales what each branch adds to the
# stream while every branch INPUT stays normalised, so the stream itself grows by ~2^k.  Powers of two keep the scaling
# exactly reversible in fp32 (the session's cached model is restored bit for bit).
# ---------------------------------------------------------------------------------------------------------------------------
_UNET_BRANCH_ENDS = ("out_layers.3.", "temopral_conv.conv4.3.", "attn1.to_out.0.", "attn2.to_out.0.", "ff.net.2.")
_VAE_BRANCH_ENDS = ("conv2.", "attn_1.proj_out.")


def _scale_branch_ends(module, ends, factor):
    with torch.no_grad():
        for name, p in module.named_parameters():
            if any(e in name for e in ends):
                p.mul_(factor)
    for m in module.modules():
        if hasattr(m, "_drop_packed"):
            m._drop_packed()


def _pick_scale(run_oracle, apply_scale, k0, lo=2.0e3, hi=1.5e4, target=5.0e3):
    """Scale by 2^k0, measure the oracle's max |activation|; if outside [lo, hi] move k once (the stream is ~linear in 2^k)."""
    import math
    apply_scale(2.0 ** k0)
    k = k0
    peak, out = run_oracle()
    if not (lo <= peak <= hi):
        dk = int(round(math.log2(target / peak)))
        apply_scale(2.0 ** dk)
        k += dk
        peak, out = run_oracle()
    return k, peak, out


def test_fp16_range_stress_unet_residual_stream():
    T, h, w = 25, 40, 64
    model, params = _model("inference_pvd_512.yaml")
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    x, ctx = _inputs(T, h, w, seed=4321)
    ts = torch.tensor([399], device=DEV)
    fs = torch.tensor([10], device=DEV)
    applied = [1.0]

    def apply_scale(f):
        _scale_branch_ends(unet, _UNET_BRANCH_ENDS, f)
        applied[0] *= f

    def run_oracle():
        sd = {k: v.detach() for k, v in unet.state_dict().items()}
        taps = {}
        with torch.no_grad():
            ref = O.unet_forward(sd, hp, x, ts, ctx, fs, taps=taps)
        peaks = {k: float(v.abs().max()) for k, v in taps.items()}
        return max(peaks.values()), (ref, peaks)
    try:
        k, peak, (ref, peaks) = _pick_scale(run_oracle, apply_scale, k0=7)
        assert 2.0e3 <= peak <= 3.0e4, f"stress precondition: oracle stream peak {peak:.0f} at 2^{k}"
        deep = max(v for n, v in peaks.items() if n in ("input_blocks.10", "input_blocks.11", "middle_block", "output_blocks.0",
                                                        "output_blocks.1", "output_blocks.2"))
        with torch.no_grad():
            y = unet(x, ts, context=ctx, fs=fs)
    finally:
        apply_scale(1.0 / applied[0])
    assert applied[0] == 1.0
    e = rel_l2(y, ref)
    print(f"\n[fp16 range stress, UNet 25x40x64] branch ends x 2^{k}: oracle residual stream peaks at {peak:.0f} "
          f"(deep 9x16 / 5x8 levels {deep:.0f}); HIP forward finite = {bool(torch.isfinite(y).all())}, rel-L2 vs fp32 oracle {e:.3e}")
    torch.cuda.empty_cache()
    assert torch.isfinite(y).all()
    assert e <= 2 * FWD_TOL


def test_fp16_range_stress_vae_decoder_576x1024():
    model, params = _model("inference_pvd_1024.yaml")
    vae = model.first_stage_model
    dd = dict(params["first_stage_config"]["params"]["ddconfig"])
    g = torch.Generator().manual_seed(78)
    z = torch.randn(1, 4, 1, 72, 128, generator=g).to(DEV)
    frame = z[:, :, 0] / params["scale_factor"]
    applied = [1.0]

    def apply_scale(f):
        _scale_branch_ends(vae.decoder, _VAE_BRANCH_ENDS, f)
        applied[0] *= f

    def run_oracle():
        sd = {k: v.detach() for k, v in vae.state_dict().items()}
        taps = {}
        with torch.no_grad():
            ref = O.vae_decode(sd, dd, frame, taps=taps)
        return max(taps.values()), (ref, taps)
    try:
        k, peak, (ref, taps) = _pick_scale(run_oracle, apply_scale, k0=8)
        assert 2.0e3 <= peak <= 3.0e4, f"stress precondition: oracle stream peak {peak:.0f} at 2^{k}"
        wide = max(v for n, v in taps.items() if n.startswith("decoder.up.0.") or n.startswith("decoder.up.1."))
        with torch.no_grad():
            out = model.decode_first_stage(z)[:, :, 0]
    finally:
        apply_scale(1.0 / applied[0])
    assert applied[0] == 1.0
    e = rel_l2(out, ref)
    print(f"\n[fp16 range stress, VAE decode 576x1024] conv2 / proj_out x 2^{k}: oracle residual stream peaks at {peak:.0f} "
          f"(128/256-channel 576x1024 / 288x512 blocks {wide:.0f}); HIP decode finite = {bool(torch.isfinite(out).all())}, "
          f"rel-L2 vs fp32 oracle {e:.3e}")
    torch.cuda.empty_cache()
    assert torch.isfinite(out).all()
    assert e <= 2 * DEC_TOL


@pytest.mark.parametrize("T", [25, 16])
def test_reference_code_itself_at_576x1024_on_gpu_pins_oracle_and_hip_path(T):
    """The reference's OWN UNetModel.forward (openaimodel3d.py:548-603; oracle/_ref = bytecode of /root/reference's modules built by
    oracle/build_ref.py, which travels to the GPU box) in fp32 on the MI355X at the headline latent (T, 72, 128) - vanilla attention
    with its [125, 9216, 9216] fp32 score tensor and all (288 GB makes that possible) - next to (a) the oracle restatement: they must
    agree to fp32 rounding, which pins the oracle to the reference AT THE BENCHMARK'S SIZE, not only on the tiny golden graphs; and
    (b) the HIP path, inside the stated fp16 tolerance, against the reference code directly.  T = 25 takes the shared image-token
    branch, T = 16 the per-frame one.  Same for AutoencoderKL.decode of one 576x1024 frame."""
    from oracle import ref_runner as R
    if not R.available():
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    model, params = _model("inference_pvd_1024.yaml")
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    sd = {k: v.detach() for k, v in unet.state_dict().items()}
    x, ctx = _inputs(T, 72, 128, seed=31 + T)
    ts, fs = torch.tensor([399], device=DEV), torch.tensor([10], device=DEV)
    with torch.device("meta"):
        ref_unet = R.reference_unet(hp)
    ref_unet.load_state_dict(sd, strict=True, assign=True)          # the product's fp32 tensors themselves, no second copy
    with torch.no_grad():
        want = ref_unet(x, ts, context=ctx, fs=fs)
        torch.cuda.empty_cache()
        got = O.unet_forward(sd, hp, x, ts, ctx, fs)
        y = unet(x, ts, context=ctx, fs=fs)
    e_oracle, e_hip = rel_l2(got, want), rel_l2(y, want)
    print(f"\n[reference code, fp32, MI355X, latent {T}x72x128] oracle restatement vs reference: rel-L2 {e_oracle:.2e} "
          f"(max |diff| {float((got - want).abs().max()):.2e}, max |ref| {float(want.abs().max()):.2f});  HIP path vs reference: {e_hip:.3e}")
    del ref_unet
    torch.cuda.empty_cache()
    assert e_oracle <= 2e-5
    assert e_hip <= FWD_TOL
    if T == 25:
        dd = dict(params["first_stage_config"]["params"]["ddconfig"])
        vsd = {k: v.detach() for k, v in model.first_stage_model.state_dict().items()}
        with torch.device("meta"):
            ref_vae = R.reference_vae(dd)
        ref_vae.load_state_dict(vsd, strict=True, assign=True)
        z = torch.randn(1, 4, 72, 128, generator=torch.Generator().manual_seed(5)).to(DEV) / params["scale_factor"]
        with torch.no_grad():
            want = ref_vae.decode(z)
            got = O.vae_decode(vsd, dd, z)
            ours = model.first_stage_model.decode(z)
        e_oracle, e_hip = rel_l2(got, want), rel_l2(ours, want)
        print(f"[reference code, fp32, MI355X, VAE decode 576x1024] oracle vs reference {e_oracle:.2e};  HIP path vs reference {e_hip:.3e}")
        del ref_vae
        torch.cuda.empty_cache()
        assert e_oracle <= 2e-5
        assert e_hip <= DEC_TOL


def test_ddim_trajectory_vs_the_reference_sampler_itself_at_576x1024x25():
    """The whole loop through the reference's own classes on the MI355X, fp32: VIPLatentDiffusion.apply_model -> DiffusionWrapper ->
    UNetModel.forward driven by DDIMSampler.sample (ddim.py:42-281: make_schedule, CFG 7.5 as two B = 1 forwards, guidance rescale 0.7,
    v-prediction, dynamic rescale, eta = 0) for 10 steps at latent 25x72x128 from an injected x_T, then decode_first_stage of two
    frames - all of it oracle/_ref bytecode.  Against it: (a) the oracle sampler on the oracle UNet (pins the restated LOOP at the
    headline size; fp32 on both sides), (b) the product sampler on the HIP UNet, final latent and decoded frames."""
    from oracle import ref_runner as R
    if not R.available():
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    from viewcrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    from tests.util import psnr
    T, h, w, steps = 25, 72, 128, 10
    model, params = _model("inference_pvd_1024.yaml")
    unet = model.model.diffusion_model
    hp = dict(params["unet_config"]["params"])
    sd = {k: v.detach() for k, v in unet.state_dict().items()}
    vsd = {k: v.detach() for k, v in model.first_stage_model.state_dict().items()}
    g = torch.Generator().manual_seed(321)
    x_T = torch.randn(1, 4, T, h, w, generator=g).to(DEV)
    cat = (torch.randn(1, 4, T, h, w, generator=g) * 0.8).to(DEV)
    ctx = torch.randn(1, 77 + 256, 1024, generator=g).to(DEV)
    uctx = torch.randn(1, 77 + 256, 1024, generator=g).to(DEV)
    cond = {"c_crossattn": [ctx], "c_concat": [cat]}
    uc = {"c_crossattn": [uctx], "c_concat": [cat]}
    fs = torch.tensor([10], device=DEV)
    kw = dict(S=steps, conditioning=cond, batch_size=1, shape=[4, T, h, w], verbose=False, unconditional_guidance_scale=7.5,
              unconditional_conditioning=uc, eta=0.0, cfg_img=None, mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing",
              guidance_rescale=0.7, x_T=x_T, log_every_t=1, unconditional_conditioning_img_nonetext=None)
    ref_model = R.reference_diffusion(params, sd, vsd, DEV)
    from lvdm.models.samplers.ddim import DDIMSampler as RefSampler            # the reference's (sys.path set by ref_runner)
    assert RefSampler is not DDIMSampler
    with torch.no_grad():
        want, want_inter = RefSampler(ref_model).sample(**kw)
        want_frames = ref_model.decode_first_stage(want[:, :, :2].contiguous())
        torch.cuda.empty_cache()
        ours, inter = DDIMSampler(model).sample(**kw)
        ours_frames = model.decode_first_stage(ours[:, :, :2].contiguous())
    tables = O.diffusion_tables(params["timesteps"], params["linear_start"], params["linear_end"], params["rescale_betas_zero_snr"])
    scale_arr = O.dynamic_rescale_table(params["timesteps"], params["base_scale"])
    assert torch.allclose(scale_arr, ref_model.scale_arr.cpu()) and torch.allclose(tables["alphas_cumprod"], ref_model.alphas_cumprod.cpu())

    def apply_oracle(x, t, c):
        return O.unet_forward(sd, hp, torch.cat([x, c["c_concat"][0]], dim=1), t.to(DEV), c["c_crossattn"][0], fs)
    with torch.no_grad():
        got, _ = O.ddim_sample(apply_oracle, tables, scale_arr, x_T, cond, uc, steps=steps, eta=0.0, cfg_scale=7.5,
                               guidance_rescale=0.7, spacing="uniform_trailing", parameterization="v")
    e_oracle, e_hip = rel_l2(got, want), rel_l2(ours, want)
    e_first = rel_l2(inter["pred_x0"][1], want_inter["pred_x0"][1])
    e_frames, p = rel_l2(ours_frames, want_frames), psnr(ours_frames, want_frames)
    print(f"\n[reference DDIMSampler + VIPLatentDiffusion + UNetModel, fp32, MI355X, 25x72x128, {steps} steps] oracle loop vs reference: "
          f"{e_oracle:.2e};  HIP path vs reference: final latent {e_hip:.3e} (first pred_x0 {e_first:.3e}), decoded 576x1024 frames "
          f"rel-L2 {e_frames:.3e} / PSNR {p:.1f} dB")
    del ref_model
    torch.cuda.empty_cache()
    assert e_oracle <= 1e-4
    assert e_hip <= TRAJ_TOL and p >= PSNR_MIN
