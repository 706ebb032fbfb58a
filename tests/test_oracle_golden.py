"""Pin the oracle (oracle/lvdm_oracle.py) against golden vectors produced by the reference code itself
(tests/golden/gen_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import lvdm_oracle as O
from oracle.weights import synth_input, synth_state_dict
from tests.tiny_config import TINY_DDCONFIG, TINY_RESAMPLER, TINY_UNET

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def sd_from_fixture(keys, shapes, prefix=""):
    shp = {str(k): eval(str(s)) for k, s in zip(keys, shapes)}
    return synth_state_dict({prefix + k: v for k, v in shp.items()}), shp


SCHEDULE_BUFFERS = ("betas", "alphas", "sqrt_", "log_one", "posterior", "scale_arr", "lvlb", "logvar")


def split_model_state_dict(keys, shapes):
    """Full-model fixture keys -> (UNet-relative, VAE-relative) synthetic state dicts.  Weights are synthesised from
    the FULL key ('model.diffusion_model.…'), exactly as gen_golden.py loaded them into the reference model."""
    shp = {str(k): eval(str(s)) for k, s in zip(keys, shapes)}
    sd = synth_state_dict(shp, skip=SCHEDULE_BUFFERS)
    up, vp = "model.diffusion_model.", "first_stage_model."
    return ({k[len(up):]: v for k, v in sd.items() if k.startswith(up)},
            {k[len(vp):]: v for k, v in sd.items() if k.startswith(vp)})


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_schedules_match_reference():
    g = load("schedules")
    betas = O.make_beta_schedule_linear(1000, 0.00085, 0.012)
    assert np.array_equal(betas, g["betas_linear"])
    assert np.allclose(O.rescale_zero_terminal_snr(betas), g["betas_zero_snr"], rtol=0, atol=1e-15)
    for method, n in (("uniform_trailing", 50), ("uniform_trailing", 5), ("uniform_trailing", 10), ("uniform", 50), ("quad", 20)):
        assert np.array_equal(O.make_ddim_timesteps(method, n, 1000), g[f"ddim_timesteps_{method}_{n}"]), (method, n)
    assert list(O.make_ddim_timesteps("uniform_trailing", 5, 1000)) == [199, 399, 599, 799, 999]
    acp = O.diffusion_tables()["alphas_cumprod"]
    ts = O.make_ddim_timesteps("uniform_trailing", 50, 1000)
    for eta in (0.0, 1.0):
        s, a, ap = O.make_ddim_sampling_parameters(acp, ts, eta)
        assert np.allclose(np.asarray(s, dtype=np.float64), g[f"ddim_sigmas_eta{eta}"], rtol=1e-6, atol=1e-9)
        assert np.allclose(np.asarray(a, dtype=np.float64), g[f"ddim_alphas_eta{eta}"], rtol=1e-7)
        assert np.allclose(np.asarray(ap, dtype=np.float64), g[f"ddim_alphas_prev_eta{eta}"], rtol=1e-7)
    t = torch.tensor([0, 19, 500, 999])
    assert np.allclose(O.timestep_embedding(t, 320).numpy(), g["timestep_embedding_320"], atol=1e-6)
    assert np.allclose(O.timestep_embedding(t, 65).numpy(), g["timestep_embedding_65"], atol=1e-6)
    a, b = synth_input("cfg_a", (2, 4, 3, 8, 8)), synth_input("cfg_b", (2, 4, 3, 8, 8))
    assert np.allclose(O.rescale_noise_cfg(a, b, 0.7).numpy(), g["rescale_noise_cfg"], atol=1e-6)


def test_diffusion_tables_match_reference_model_buffers():
    g = load("ddim_tiny")
    tb = O.diffusion_tables()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        assert np.array_equal(tb[k].numpy(), g["model_" + k]), k
    assert float(tb["alphas_cumprod"][-1]) == 0.0   # zero terminal SNR
    assert np.array_equal(O.dynamic_rescale_table(1000, 0.3, 400).numpy(), g["model_scale_arr"])


@pytest.mark.parametrize("tag,shape,L", [("perframe", (1, 4, 32, 16), 77 + 64), ("shared", (2, 3, 16, 32), 77 + 40)])
def test_unet_forward_matches_reference(tag, shape, L):
    g = load("unet_tiny")
    sd, _ = sd_from_fixture(g["unet_keys"], g["unet_shapes"])
    b, t, h, w = shape
    x = synth_input(f"unet_x_{tag}", (b, 8, t, h, w))
    ctx = synth_input(f"unet_ctx_{tag}", (b, L, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, TINY_UNET, x, torch.tensor([999, 399][:b]), ctx, torch.tensor([10, 3][:b]))
    assert y.shape == g[f"unet_out_{tag}"].shape
    assert max_rel(y.numpy(), g[f"unet_out_{tag}"]) < 2e-5


def test_vae_matches_reference():
    g = load("vae_tiny")
    sd, _ = sd_from_fixture(g["vae_keys"], g["vae_shapes"])
    with torch.no_grad():
        dec = O.vae_decode(sd, TINY_DDCONFIG, synth_input("vae_z", (2, 4, 8, 16)))
        mom = O.vae_encode_moments(sd, TINY_DDCONFIG, synth_input("vae_img", (1, 3, 64, 32), scale=0.5))
    assert max_rel(dec.numpy(), g["vae_decode"]) < 2e-5
    assert max_rel(mom.numpy(), g["vae_encode_moments"]) < 2e-5
    assert max_rel(mom[:, :4].numpy(), g["vae_encode_mode"]) < 2e-5


@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_ddim_trajectory_matches_reference(eta):
    """5-step DDIM (CFG 7.5, rescale 0.7, uniform_trailing, dynamic rescale, v-pred) + decode_first_stage."""
    g = load("ddim_tiny")
    unet_sd, vae_sd = split_model_state_dict(g["model_keys"], g["model_shapes"])
    b, t, h, w = 1, 4, 32, 16
    cd = TINY_UNET["context_dim"]
    ctx, uctx = synth_input("ddim_ctx", (b, 77 + 16 * t, cd)), synth_input("ddim_uctx", (b, 77 + 16 * t, cd))
    cat = synth_input("ddim_cat", (b, 4, t, h, w), scale=0.8)
    x_T = synth_input("ddim_xT", (b, 4, t, h, w))
    fs = torch.tensor([10] * b)

    def apply_model(x, ts, c):   # DiffusionWrapper 'hybrid', ddpm3d.py:1437-1443
        return O.unet_forward(unet_sd, TINY_UNET, torch.cat([x, cat], 1), ts, c, fs)

    counter = [0]

    def noise_fn(shape):
        counter[0] += 1
        return synth_input(f"ddim_noise_{counter[0]}", shape)

    with torch.no_grad():
        v = apply_model(x_T, torch.tensor([999]), ctx)
        assert max_rel(v.numpy(), g["apply_model"]) < 2e-5
        x0, preds = O.ddim_sample(apply_model, O.diffusion_tables(), O.dynamic_rescale_table(1000, 0.3, 400), x_T, ctx, uctx,
                                  steps=5, eta=eta, cfg_scale=7.5, guidance_rescale=0.7, noise_fn=noise_fn)
        assert max_rel(torch.stack(preds).numpy(), g[f"ddim_pred_x0_eta{eta}"]) < 2e-4
        assert max_rel(x0.numpy(), g[f"ddim_samples_eta{eta}"]) < 2e-4
        if eta == 0.0:
            dec = O.decode_first_stage(vae_sd, TINY_DDCONFIG, torch.from_numpy(g["ddim_samples_eta0.0"]))
            assert max_rel(dec.numpy()[..., ::4, ::4], g["decode_first_stage_sub4"]) < 2e-5


def test_multicond_ddim_trajectory_matches_reference():
    """`--multiple_cond_cfg`: 3-way guidance (cfg 7.5, cfg_img 3.0) and the un-fixed ddim_scale_arr_prev of ddim_multiplecond.py."""
    g = load("ddim_tiny")
    unet_sd, _ = split_model_state_dict(g["model_keys"], g["model_shapes"])
    b, t, h, w = 1, 4, 32, 16
    cd = TINY_UNET["context_dim"]
    ctx, uctx = synth_input("ddim_ctx", (b, 77 + 16 * t, cd)), synth_input("ddim_uctx", (b, 77 + 16 * t, cd))
    uctx2 = torch.cat([uctx[:, :77], ctx[:, 77:]], 1)
    cat = synth_input("ddim_cat", (b, 4, t, h, w), scale=0.8)
    x_T = synth_input("ddim_xT", (b, 4, t, h, w))
    fs = torch.tensor([10] * b)

    def apply_model(x, ts, c):
        return O.unet_forward(unet_sd, TINY_UNET, torch.cat([x, cat], 1), ts, c, fs)
    with torch.no_grad():
        x0, preds = O.ddim_sample(apply_model, O.diffusion_tables(), O.dynamic_rescale_table(1000, 0.3, 400), x_T, ctx, uctx,
                                  steps=5, eta=0.0, cfg_scale=7.5, guidance_rescale=0.7, uncond_img=uctx2, cfg_img=3.0)
    assert max_rel(torch.stack(preds).numpy(), g["multicond_pred_x0"]) < 2e-4
    assert max_rel(x0.numpy(), g["multicond_samples"]) < 2e-4


def test_resampler_oracle_matches_reference():
    """image_proj_model (reference lvdm/modules/encoders/resampler.py, imported unmodified by gen_golden.py)."""
    g = load("resampler_tiny")
    sd, _ = sd_from_fixture(g["resampler_keys"], g["resampler_shapes"])
    for tag, (b, n1) in {"a": (2, 17), "b": (1, 40)}.items():
        x = synth_input(f"resampler_x_{tag}", (b, n1, TINY_RESAMPLER["embedding_dim"]))
        y = O.resampler_forward(sd, TINY_RESAMPLER, x)
        assert y.shape == g[f"resampler_out_{tag}"].shape
        assert max_rel(y.numpy(), g[f"resampler_out_{tag}"]) <= 2e-5, tag


def test_clip_encoders_oracle_matches_reference():
    """The reference's FrozenOpenCLIPEmbedder / FrozenOpenCLIPImageEmbedderV2 code (condition.py, run by gen_golden.py on the
    open_clip / kornia stand-ins) against the state-dict restatement the GPU tests use as their fp32 oracle."""
    from oracle import clip_oracle as C
    from tests.tiny_config import CLIP_TINY_CFG
    assert C.CLIP_CONFIGS["vcx-tiny-test"] == CLIP_TINY_CFG
    g = load("clip_tiny")
    t, v = CLIP_TINY_CFG["text"], CLIP_TINY_CFG["vision"]
    sd, _ = sd_from_fixture(g["clip_text_keys"], g["clip_text_shapes"])
    assert not any(k.startswith("model.visual.") for k in sd) and "model.transformer.resblocks.0.attn.in_proj_weight" in sd
    y = C.clip_text_forward(sd, C.tokenize_empty(2), t["heads"], t["layers"], layer_idx=1)
    assert max_rel(y.numpy(), g["clip_text_empty"]) <= 2e-5
    y = C.clip_text_forward(sd, torch.from_numpy(g["clip_text_tokens"]), t["heads"], t["layers"], layer_idx=1)
    assert max_rel(y.numpy(), g["clip_text_random"]) <= 2e-5
    sd, _ = sd_from_fixture(g["clip_image_keys"], g["clip_image_shapes"])
    assert not any(k.startswith("model.transformer.") for k in sd) and "model.visual.conv1.weight" in sd
    for tag, shp in {"down": (2, 3, 320, 448), "up": (1, 3, 96, 64)}.items():
        x = torch.tanh(synth_input(f"clip_image_{tag}", shp))
        y = C.clip_image_forward(sd, x, v["width"] // v["head_width"], v["layers"], v["patch_size"])
        assert y.shape == g[f"clip_image_{tag}"].shape
        assert max_rel(y.numpy(), g[f"clip_image_{tag}"]) <= 2e-5, tag


def test_clip_towers_match_transformers():
    """Third-party pin of the OpenCLIP tower internals: Hugging Face transformers ships an independent
    implementation of the CLIP architecture; its random-initialised towers, with their parameters renamed to open_clip's
    state-dict names (oracle/clip_hf.py), must agree with the oracle's restatement at the points the reference taps: the text
    tower at layer 'penultimate' through ln_final (condition.py:218-237) and the vision tower's token stream after the last block
    (condition.py:347-378)."""
    pytest.importorskip("transformers")
    from oracle import clip_hf as H
    from oracle import clip_oracle as C
    t, v = H.HF_TINY_CFG["text"], H.HF_TINY_CFG["vision"]
    tm, tsd = H.build_text()
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(2, t["vocab_size"], (3, 77), generator=g)
    tokens[:, 0] = 0
    ref = H.hf_text_penultimate(tm, tokens)
    y = C.clip_text_forward(tsd, tokens, t["heads"], t["layers"], layer_idx=1)
    assert max_rel(y.numpy(), ref.numpy()) <= 2e-5
    y_last = C.clip_text_forward(tsd, tokens, t["heads"], t["layers"], layer_idx=0)       # layer='last' differs: the tap is real
    assert max_rel(y_last.numpy(), ref.numpy()) > 1e-2
    vm, vsd = H.build_vision()
    x = torch.randn(2, 3, 224, 224, generator=g)
    ref = H.hf_vision_tokens(vm, x)
    y = C.clip_image_tower(vsd, x, v["width"] // v["head_width"], v["layers"], v["patch_size"])
    assert y.shape == ref.shape == (2, 17, v["width"])
    assert max_rel(y.numpy(), ref.numpy()) <= 2e-5


def test_oracle_matches_the_reference_code_itself_when_oracle_ref_is_built():
    """oracle/_ref (oracle/build_ref.py: bytecode of the reference's own modules) run next to the oracle restatement on fresh
    inputs and the deterministic synthetic weights: UNet forward (both image-token branches) and VAE decode / encode.  This is
    the same check the committed goldens make, but on the code itself, wherever the _ref tree travelled to."""
    from oracle import ref_runner as R
    if not R.available():
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    import sys
    from oracle.weights import synth_state_dict
    from tests.tiny_config import TINY_DDCONFIG, TINY_UNET
    ref = R.reference_unet(TINY_UNET)
    # the stand-ins for cv2 / pytorch_lightning / torchvision exist only while the reference is being imported: a fake package left
    # in sys.modules breaks `import transformers...CLIP*` later in the same process (it probes find_spec("torchvision"))
    assert all(n not in sys.modules or hasattr(sys.modules[n], "__file__") for n in ("cv2", "pytorch_lightning", "torchvision"))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, seed=0)
    ref.load_state_dict(sd, strict=True)
    for T, L in ((4, 77 + 64), (3, 77 + 256)):            # per-frame image tokens (77 + 16 T) / shared ones
        x = synth_input(f"ref_x_{T}", (1, 8, T, 16, 16))
        ctx = synth_input(f"ref_ctx_{T}", (1, L, TINY_UNET["context_dim"]))
        ts, fs = torch.tensor([640]), torch.tensor([12])
        with torch.no_grad():
            want = ref(x, ts, context=ctx, fs=fs)
            got = O.unet_forward(sd, TINY_UNET, x, ts, ctx, fs)
        assert max_rel(got.numpy(), want.numpy()) <= 2e-5, T
    vae = R.reference_vae(TINY_DDCONFIG)
    vsd = synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=0)
    vae.load_state_dict(vsd, strict=True)
    z = synth_input("ref_z", (2, 4, 8, 16))
    img = synth_input("ref_img", (1, 3, 64, 32), scale=0.5)
    with torch.no_grad():
        assert max_rel(O.vae_decode(vsd, TINY_DDCONFIG, z).numpy(), vae.decode(z).numpy()) <= 2e-5
        assert max_rel(O.vae_encode_moments(vsd, TINY_DDCONFIG, img).numpy(), vae.encode(img).parameters.numpy()) <= 2e-5


def test_kornia_resize_two_independent_restatements_agree():
    """kornia is absent from /root/reference and from the image, so the pre-processing of the vision tower (reference
    condition.py:322-329) cannot be pinned to kornia's code.  It is pinned to kornia's published ALGORITHM twice, independently:
    oracle/clip_oracle.py (torch conv2d + F.interpolate - what wrote tests/golden/clip_tiny.npz under the reference's embedder
    code) and oracle/kornia_numpy.py (dense fp64 matrices built element by element from the formulas; shares no code with torch).
    Down-scaling with both axes / one axis shrinking, up-scaling (no blur), same size (identity), odd sizes."""
    from oracle import clip_oracle as C
    from oracle import kornia_numpy as K
    g = torch.Generator().manual_seed(0)
    for shp in [(1, 3, 576, 1024), (2, 3, 320, 512), (1, 3, 96, 64), (1, 3, 224, 224), (1, 3, 301, 500), (1, 3, 224, 600), (1, 2, 200, 250)]:
        x = torch.tanh(torch.randn(*shp, generator=g, dtype=torch.float64))
        mean, std = C.CLIP_MEAN[:shp[1]], C.CLIP_STD[:shp[1]]
        a = C.kornia_normalize((C.kornia_resize(x, (224, 224), True) + 1) / 2, mean, std).numpy()
        b = K.clip_preprocess(x.numpy(), mean=mean, std=std)
        assert np.abs(a - b).max() <= 1e-11, shp
        a = C.kornia_resize(x, (224, 224), False).numpy()                        # antialias off
        My, Mx = K.resize_matrices(shp[2], shp[3], 224, antialias=False)
        assert np.abs(a - np.einsum("oh,bchw,pw->bcop", My, x.numpy(), Mx, optimize=True)).max() <= 1e-11, shp


def test_clip_preprocess_against_pillow_documents_the_delta():
    """A third implementation that IS in the image, with DIFFERENT published rules: Pillow's BICUBIC (Keys a = -0.5, support scaled
    by the shrink factor as its anti-aliasing, pixel-centre alignment) against kornia's rule (Gaussian pre-blur + a = -0.75,
    corner alignment).  They are not expected to agree bit for bit; on a smooth 576x1024 image they agree to 41.6 dB - which
    catches gross mistakes (transposed axes, wrong scale, missing blur) and is the measured delta DESIGN.md quotes."""
    pytest.importorskip("PIL")
    from PIL import Image
    from oracle import kornia_numpy as K
    yy, xx = np.mgrid[0:576, 0:1024]
    img = (0.5 + 0.5 * np.sin(xx / 37.0) * np.cos(yy / 23.0)).astype(np.float32)
    pil = np.asarray(Image.fromarray(img, mode="F").resize((224, 224), Image.BICUBIC))
    ours = K.clip_preprocess(img[None, None] * 2 - 1, mean=(0.0,), std=(1.0,))[0, 0]
    psnr_db = 10 * np.log10(1.0 / float(((pil - ours) ** 2).mean()))
    assert 38.0 <= psnr_db <= 46.0, psnr_db
    assert 10 * np.log10(1.0 / float(((pil - ours.T[:224, :224]) ** 2).mean())) < 25.0      # ... and a transposed result would not pass


def test_oracle_unet_with_scale_shift_norm_matches_reference():
    """use_scale_shift_norm=True (reference openaimodel3d.py:221-225) - not used by the shipped YAMLs, accepted by the reference's
    constructor: golden written by the reference's own UNetModel with that flag (gen_golden.py::gen_unet_ssn)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_ssn.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(g["unet_keys"], g["unet_shapes"])}
    sd = synth_state_dict(shapes)
    x = synth_input("unet_ssn_x", (2, 8, 3, 16, 32))
    ctx = synth_input("unet_ssn_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, dict(TINY_UNET, use_scale_shift_norm=True), x, torch.tensor([999, 399]), ctx, torch.tensor([10, 3]))
    assert np.abs(y.numpy() - g["unet_out"]).max() <= 2e-5 * np.abs(g["unet_out"]).max() + 1e-6


def test_oracle_unet_with_conv1x1_projections_matches_reference():
    """use_linear=False (reference attention.py:266-267, 287-288, 331-336: proj_in / proj_out as 1x1 Conv2d / Conv1d) - not used by the shipped
    YAMLs, accepted by the reference's constructor: golden written by the reference's own UNetModel with that flag (gen_golden.py::gen_unet_conv1x1)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_conv1x1.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(g["unet_keys"], g["unet_shapes"])}
    assert shapes["input_blocks.1.1.proj_in.weight"][2:] == (1, 1) and len(shapes["input_blocks.1.2.proj_in.weight"]) == 3
    sd = synth_state_dict(shapes)
    x = synth_input("unet_c11_x", (2, 8, 3, 16, 32))
    ctx = synth_input("unet_c11_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, dict(TINY_UNET, use_linear=False), x, torch.tensor([999, 399]), ctx, torch.tensor([10, 3]))
    assert np.abs(y.numpy() - g["unet_out"]).max() <= 2e-5 * np.abs(g["unet_out"]).max() + 1e-6


@pytest.mark.parametrize("name,tag,flags", [("unet_tiny_updown", "ud", dict(resblock_updown=True)), ("unet_tiny_noconv", "nc", dict(conv_resample=False)),
                                            ("unet_tiny_causal", "ca", dict(use_causal_attention=True))])
def test_oracle_unet_sampling_variants_match_reference(name, tag, flags):
    """resblock_updown=True (reference openaimodel3d.py:441-451, 529-538, 210-215) and conv_resample=False (:70-72, 98-103) - not used by the
    shipped YAMLs, accepted by the reference's constructor: goldens written by the reference's own UNetModel (gen_golden.py::_gen_unet_variant)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(g["unet_keys"], g["unet_shapes"])}
    sd = synth_state_dict(shapes)
    x = synth_input(f"unet_{tag}_x", (2, 8, 3, 16, 32))
    ctx = synth_input(f"unet_{tag}_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, dict(TINY_UNET, **flags), x, torch.tensor([999, 399]), ctx, torch.tensor([10, 3]))
    assert np.abs(y.numpy() - g["unet_out"]).max() <= 2e-5 * np.abs(g["unet_out"]).max() + 1e-6


def test_oracle_unet_with_relative_position_matches_reference():
    """use_relative_position=True (reference attention.py:20-40, 59-62, 104-108, 120-123), temporal_length 2 with 5 frames (distances clipped):
    golden written by the reference's own UNetModel (gen_golden.py::gen_unet_relpos)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_relpos.npz"))
    shapes = {str(k): eval(str(s)) for k, s in zip(g["unet_keys"], g["unet_shapes"])}
    assert shapes["input_blocks.1.2.transformer_blocks.0.attn1.relative_position_k.embeddings_table"] == (5, 64)
    sd = synth_state_dict(shapes)
    x = synth_input("unet_rp_x", (1, 8, 5, 16, 16))
    ctx = synth_input("unet_rp_ctx", (1, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, dict(TINY_UNET, use_relative_position=True, temporal_length=2), x, torch.tensor([599]), ctx, torch.tensor([10]))
    assert np.abs(y.numpy() - g["unet_out"]).max() <= 2e-5 * np.abs(g["unet_out"]).max() + 1e-6


def _adapter_features(b, t, h, w):
    return [synth_input(f"adapter_{i}", (b * t, TINY_UNET["model_channels"] * m, h >> i, w >> i), scale=0.5)
            for i, m in enumerate(TINY_UNET["channel_mult"])]


def test_oracle_unet_with_features_adapter_matches_reference():
    """features_adapter (reference openaimodel3d.py:582-588: adapter maps added behind input blocks 2, 5, 8, 11) - golden written by
    the reference's own UNetModel.forward (gen_golden.py::gen_unet_adapter)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_adapter.npz"))
    u = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny.npz"))
    sd = synth_state_dict({str(k): eval(str(s)) for k, s in zip(u["unet_keys"], u["unet_shapes"])})
    x = synth_input("unet_ad_x", (1, 8, 3, 16, 32))
    ctx = synth_input("unet_ad_ctx", (1, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = O.unet_forward(sd, TINY_UNET, x, torch.tensor([599]), ctx, torch.tensor([10]), features_adapter=_adapter_features(1, 3, 16, 32))
        y0 = O.unet_forward(sd, TINY_UNET, x, torch.tensor([599]), ctx, torch.tensor([10]))
    assert np.abs(y.numpy() - g["unet_out"]).max() <= 2e-5 * np.abs(g["unet_out"]).max() + 1e-6
    assert np.abs(y0.numpy() - g["unet_out"]).max() > 1e-2                       # the adapter maps matter
    with pytest.raises(AssertionError, match="Wrong features_adapter"):
        O.unet_forward(sd, TINY_UNET, x, torch.tensor([599]), ctx, torch.tensor([10]), features_adapter=_adapter_features(1, 3, 16, 32) + [x])
