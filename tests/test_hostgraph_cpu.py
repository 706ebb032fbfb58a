"""The HOST side of the product UNet / VAE - the Python graph that decides which libvcx kernel gets which pointers, strides, flags,
moment buffers and concat targets - executed on a GPU-less machine against the reference goldens and the fp32 oracle.

tests/cpu_kernels.py restates the `viewcrafter_amd.ops` launchers in plain PyTorch from the contracts of include/vcx.h and is swapped
in for ONE test at a time; the product has no CPU path (test_forward_fails_loudly_without_gpu still holds).  This is where a plumbing
mistake - a moment buffer with the wrong leading dimension, a skip copied behind the wrong columns, a missing replicate under the
shared CFG prefix - shows up before a GPU minute is spent.
"""
import importlib
import math

import pytest
import torch

from oracle import lvdm_oracle as O
from oracle.weights import synth_input
from tests import cpu_kernels
from tests.tiny_config import TINY_DDCONFIG, TINY_UNET
from tests.util import golden, load_synth, rel_l2

UNET_TOL = 5e-3


def _unet(monkeypatch, level):
    """A tiny product UNet whose launches go to tests/cpu_kernels.py, with VCX_GN_EPILOGUE_STATS = level."""
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules import attention, flow
    from viewcrafter_amd.lvdm.modules.networks import openaimodel3d as om
    for mod in (flow, attention, om):
        monkeypatch.setattr(mod, "GN_STATS_LEVEL", level, raising=False)
        monkeypatch.setattr(mod, "GN_EPILOGUE_STATS", level >= 1, raising=False)
    m = om.UNetModel(**TINY_UNET).eval()
    sd = load_synth(m)
    return m, sd


@pytest.mark.parametrize("level", [2, 1, 0])
@pytest.mark.parametrize("tag,shape,L", [("perframe", (1, 4, 32, 16), 77 + 64), ("shared", (2, 3, 16, 32), 77 + 40)])
def test_host_graph_of_the_unet_vs_reference_golden(monkeypatch, level, tag, shape, L):
    m, _ = _unet(monkeypatch, level)
    g = golden("unet_tiny")
    b, t, h, w = shape
    x = synth_input(f"unet_x_{tag}", (b, 8, t, h, w))
    ctx = synth_input(f"unet_ctx_{tag}", (b, L, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([999, 399][:b]), context=ctx, fs=torch.tensor([10, 3][:b]))
    e = rel_l2(y, g[f"unet_out_{tag}"])
    print(f"host graph on CPU kernels, GN statistics level {level}, {tag}: rel-L2 vs the reference golden = {e:.3e}")
    assert y.shape == g[f"unet_out_{tag}"].shape and e <= UNET_TOL


@pytest.mark.parametrize("level", [2, 0])
@pytest.mark.parametrize("tag,shape,L", [("perframe", (1, 4, 32, 16), 77 + 64), ("shared", (2, 3, 16, 32), 77 + 40)])
def test_host_graph_with_the_transformer_groupnorms_folded_into_proj_in(monkeypatch, level, tag, shape, L):
    """TemporalTransformer.norm as per-video weights / bias of proj_in (vcx_groupnorm_fold_linear_f16 + one GEMM per video) at EVERY
    level of the tiny graph (the product folds from 16 MB per video up): statistics from the producer's moments (level 2) or from
    a statistics pass (level 0), two videos with different statistics ("shared": b = 2), against the reference golden."""
    from viewcrafter_amd import ops
    from viewcrafter_amd.lvdm.modules import attention
    m, _ = _unet(monkeypatch, level)
    monkeypatch.setattr(attention, "GN_FOLD", True)
    monkeypatch.setattr(attention, "GN_FOLD_MIN_BYTES", 0)
    # ... and SpatialTransformer.norm as per-FRAME weights (the product: only where vcx_gemm_units_f16 takes all frames in one launch)
    monkeypatch.setattr(attention, "spatial_fold_ok", lambda n, pixels, C, D: pixels % 8 == 0)
    calls = dict(fold=0, videos=0)
    fold = ops.group_norm_fold_linear

    def counted(w32, bias, gamma, beta, stats, eps, **k):
        calls["fold"] += 1
        calls["videos"] += stats.shape[0]
        return fold(w32, bias, gamma, beta, stats, eps, **k)
    monkeypatch.setattr(ops, "group_norm_fold_linear", counted)
    g = golden("unet_tiny")
    b, t, h, w = shape
    x = synth_input(f"unet_x_{tag}", (b, 8, t, h, w))
    ctx = synth_input(f"unet_ctx_{tag}", (b, L, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([999, 399][:b]), context=ctx, fs=torch.tensor([10, 3][:b]))
    e = rel_l2(y, g[f"unet_out_{tag}"])
    n_tt = sum(isinstance(mod, attention.TemporalTransformer) for mod in m.modules())
    n_st = sum(isinstance(mod, attention.SpatialTransformer) for mod in m.modules())
    print(f"GroupNorm folded into proj_in ({calls['fold']} of {n_tt} temporal + {n_st} spatial transformers, {calls['videos']} weight sets), GN statistics "
          f"level {level}, {tag}: rel-L2 vs the reference golden = {e:.3e}")
    assert calls["fold"] == n_tt + n_st and calls["videos"] == n_tt * b + n_st * b * t
    assert y.shape == g[f"unet_out_{tag}"].shape and e <= UNET_TOL


def test_moments_travel_with_the_activations_and_the_concat_is_written_in_place(monkeypatch):
    """Level 2 (round 4): count what the graph asks of the kernels at a latent where every level but the deepest has whole 64-row
    strips - statistics passes only where no producer could supply moments, one copy per concat instead of two - and check the
    result against levels 1 / 0 (same graph, statistics passes) and the oracle, incl. the shared CFG prefix (replicated moments)."""
    from viewcrafter_amd import ops
    b, t, h, w, L = 1, 2, 32, 64, 77 + 32            # levels: 2048, 512, 128 and 32 pixels per frame
    x = synth_input("hg_x", (b, 8, t, h, w))
    ctx = synth_input("hg_ctx", (2, L, TINY_UNET["context_dim"]))
    ts, fs = torch.tensor([459]), torch.tensor([10])
    outs, counts, noise = {}, {}, {}
    for level in (2, 1, 0):
        with monkeypatch.context() as mp:
            m, sd = _unet(mp, level)
            calls = dict(stats_pass=0, from_moments=0, copy2d=0)
            gn, from_cs, cp = ops.group_norm, ops.group_norm_stats_from_colstats, ops.copy2d

            def counted_gn(x_, *a, stats=None, **k):
                calls["stats_pass"] += stats is None
                return gn(x_, *a, stats=stats, **k)

            def counted_from(*a, **k):
                calls["from_moments"] += 1
                return from_cs(*a, **k)

            def counted_copy(*a, **k):
                calls["copy2d"] += 1
                return cp(*a, **k)
            mp.setattr(ops, "group_norm", counted_gn)
            mp.setattr(ops, "group_norm_stats_from_colstats", counted_from)
            mp.setattr(ops, "copy2d", counted_copy)
            with torch.no_grad():
                y = m(x, ts, context=ctx[:1].contiguous(), fs=fs)
                counts[level] = dict(calls)
                y2 = m(x, ts, context=ctx, fs=fs, cfg_repeat=2)
                full = m(torch.cat([x, x]), torch.cat([ts, ts]), context=ctx, fs=torch.cat([fs, fs]))
            outs[level] = y
            # shared prefix == replicated batch (replicated moments included).  On the GPU kernels this holds bit for bit
            # (tests/test_model_gpu.py::test_cfg_shared_prefix_is_bit_identical); the host BLAS behind these stand-ins is not
            # batch-invariant, so here: to rounding noise
            noise[level] = (rel_l2(y2, full), rel_l2(y2[:1], y))
            assert max(noise[level]) <= 5e-3, (level, noise[level])       # (wrong moments for the replicated half would be >> this)
    ref = O.unet_forward(sd, TINY_UNET, x, ts, ctx[:1], fs)
    errs = {lv: rel_l2(o, ref) for lv, o in outs.items()}
    print(f"statistics passes / norms fed by moments / copies per forward: {counts};  rel-L2 vs oracle: {errs};  prefix-vs-batch noise: {noise}")
    assert all(e <= UNET_TOL for e in errs.values())
    assert rel_l2(outs[2], outs[0]) <= 5e-3                           # (fp16 rounding chaos between two summation orders: ~2e-3)
    total = {lv: c["stats_pass"] + c["from_moments"] for lv, c in counts.items()}
    assert total[2] == total[1] == total[0]                           # same norms, different sources of their statistics
    assert counts[0]["from_moments"] == 0
    assert counts[2]["stats_pass"] < counts[1]["stats_pass"] < counts[0]["stats_pass"]
    # what is left at level 2: the very first norms (behind conv_in: 8 input channels), and the levels with 128 / 32 pixels per
    # frame where a frame is not a whole number of strips for a per-frame norm
    assert counts[2]["stats_pass"] <= counts[0]["stats_pass"] // 2
    assert counts[2]["copy2d"] < counts[1]["copy2d"]                   # one copy per concat (+ the moments) instead of two


def test_host_graph_of_the_vae_vs_reference_golden(monkeypatch):
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.models.autoencoder import AutoencoderKL
    m = AutoencoderKL(ddconfig=TINY_DDCONFIG, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
    load_synth(m)
    g = golden("vae_tiny")
    with torch.no_grad():
        dec = m.decode(synth_input("vae_z", (2, 4, 8, 16)))
        post = m.encode(synth_input("vae_img", (1, 3, 64, 32), scale=0.5))
        odd = m.decode(synth_input("vae_z_9x15", (1, 4, 9, 15)))     # 135 tokens in the attention block: padded rows
    assert rel_l2(dec, g["vae_decode"]) <= 8e-3 and rel_l2(post.parameters, g["vae_encode_moments"]) <= 8e-3
    sd = {k: v for k, v in m.state_dict().items()}
    assert rel_l2(odd, O.vae_decode(sd, TINY_DDCONFIG, synth_input("vae_z_9x15", (1, 4, 9, 15)))) <= 8e-3


def _clip_preprocess_cpu(x, size, antialias, mean, std):
    from oracle import clip_oracle as C
    return C.kornia_normalize((C.kornia_resize(x.float(), (size, size), antialias) + 1.0) / 2.0, mean, std)


@pytest.mark.parametrize("tag,kw", [("cfg", dict(n_samples=2, multiple_cond_cfg=False, cfg_img=None)),
                                    ("multicond", dict(n_samples=1, multiple_cond_cfg=True, cfg_img=3.0))])
def test_host_graph_of_image_guided_synthesis_vs_the_references_own_run(monkeypatch, tag, kw):
    """The whole driver - both OpenCLIP towers, Resampler, VAE encode, DDIM sampler (CFG / multi-condition, 5 steps, eta 1), UNet,
    VAE decode - as the product's host code issues it, on the CPU stand-ins, against the fixture the reference's OWN
    `image_guided_synthesis` wrote (tests/golden/gen_golden.py::gen_igs).  The GPU twin is tests/test_entry_gpu.py."""
    from oracle.weights import NamedRandn
    from tests.tiny_config import CLIP_TINY, CLIP_TINY_CFG, IGS_H, IGS_T, IGS_W, igs_model_params
    from tests.util import SCHEDULE_BUFFERS
    from viewcrafter_amd import ops
    from viewcrafter_amd.config import Config
    from viewcrafter_amd.lvdm.modules.encoders import condition as cond
    from viewcrafter_amd.utils.diffusion_utils import image_guided_synthesis, instantiate_from_config
    cpu_kernels.install(monkeypatch)
    monkeypatch.setattr(ops, "clip_preprocess", _clip_preprocess_cpu)
    monkeypatch.setattr(ops, "ddim_step", _ddim_step_cpu)
    cond.CLIP_CONFIGS[CLIP_TINY] = CLIP_TINY_CFG
    R = "lvdm.modules.encoders."
    params = Config.wrap(igs_model_params("lvdm.modules.networks.openaimodel3d.UNetModel", "lvdm.models.autoencoder.AutoencoderKL",
                                          R + "condition.FrozenOpenCLIPEmbedder", R + "condition.FrozenOpenCLIPImageEmbedderV2",
                                          R + "resampler.Resampler", device="cpu"))
    m = instantiate_from_config(Config(target="lvdm.models.ddpm3d.VIPLatentDiffusion", params=params)).eval()
    load_synth(m, skip=SCHEDULE_BUFFERS)
    g = golden("igs_tiny")
    videos = torch.tanh(synth_input("igs_videos", (1, 3, IGS_T, IGS_H, IGS_W)))
    fake = NamedRandn(f"igs_{tag}_randn")
    monkeypatch.setattr(torch, "randn", fake)
    with torch.no_grad():
        vid = image_guided_synthesis(m, [""], videos, [1, 4, IGS_T, IGS_H // 8, IGS_W // 8], ddim_steps=5, ddim_eta=1.0,
                                     unconditional_guidance_scale=7.5, fs=10, text_input=False, timestep_spacing="uniform_trailing",
                                     guidance_rescale=0.7, condition_index=[0], **kw)
    monkeypatch.undo()
    assert fake.calls == int(g[f"igs_{tag}_randn_calls"])
    e = rel_l2(vid[..., ::4, ::4], g[f"igs_{tag}_sub4"])
    print(f"host graph of image_guided_synthesis[{tag}] on CPU stand-ins vs the reference's own run: rel-L2 {e:.3e}")
    assert e <= 3e-2


def _ddim_step_cpu(x, v_cond, v_uncond, noise, coef, ws=None, v_img=None, cfg_img=0.0):
    """include/vcx.h vcx_ddim_step3_f32 in plain PyTorch (fp64 sums): coef = {sqrt_acp_t, sqrt_1m_acp_t, a_prev, sigma_t, scale_ratio,
    cfg_scale, guidance_rescale, parameterization_is_v}."""
    sa, s1, a_prev, sigma, ratio, cfg, resc, is_v = [float(c) for c in coef[:8]]
    if v_uncond is None:
        v = v_cond
    elif v_img is None:
        v = v_uncond + cfg * (v_cond - v_uncond)
    else:
        v = v_uncond + cfg_img * (v_img - v_uncond) + cfg * (v_cond - v_img)
    if v_uncond is not None and resc > 0:
        dims = tuple(range(1, v.dim()))
        std_pos, std_cfg = v_cond.double().std(dim=dims, keepdim=True), v.double().std(dim=dims, keepdim=True)
        v = (resc * (v.double() * (std_pos / std_cfg)) + (1 - resc) * v.double()).float()
    if is_v:
        e_t, pred_x0 = sa * v + s1 * x, sa * x - s1 * v
    else:
        e_t, pred_x0 = v, (x - s1 * v) / sa
    pred_x0 = pred_x0 * ratio
    x_prev = math.sqrt(a_prev) * pred_x0 + math.sqrt(max(1.0 - a_prev - sigma ** 2, 0.0)) * e_t
    if noise is not None and sigma != 0:
        x_prev = x_prev + sigma * noise
    return x_prev, pred_x0


def test_host_graph_with_scale_shift_norm_vs_reference_golden(monkeypatch):
    """ResBlock(use_scale_shift_norm=True): (1 + scale, shift) folded into per-video GroupNorm affine parameters."""
    import numpy as np
    import os
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    m = UNetModel(**dict(TINY_UNET, use_scale_shift_norm=True)).eval()
    load_synth(m)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_ssn.npz"))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["unet_keys"]]
    x = synth_input("unet_ssn_x", (2, 8, 3, 16, 32))
    ctx = synth_input("unet_ssn_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
    e = rel_l2(y, g["unet_out"])
    print(f"host graph with use_scale_shift_norm vs the reference golden: {e:.3e}")
    assert e <= UNET_TOL


def test_host_graph_with_conv1x1_projections_vs_reference_golden(monkeypatch):
    """use_linear=False: SpatialTransformer / TemporalTransformer projections as 1x1 Conv2d / Conv1d parameters (strict state-dict names and
    shapes of the reference), packed as the same [out, in] matrices."""
    import numpy as np
    import os
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    m = UNetModel(**dict(TINY_UNET, use_linear=False)).eval()
    load_synth(m)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_conv1x1.npz"))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["unet_keys"]]
    assert [str(tuple(m.state_dict()[str(k)].shape)) for k in g["unet_keys"]] == [str(s) for s in g["unet_shapes"]]
    x = synth_input("unet_c11_x", (2, 8, 3, 16, 32))
    ctx = synth_input("unet_c11_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
    e = rel_l2(y, g["unet_out"])
    print(f"host graph with use_linear=False vs the reference golden: {e:.3e}")
    assert e <= UNET_TOL


@pytest.mark.parametrize("level", [2, 0])
@pytest.mark.parametrize("name,tag,flags", [("unet_tiny_updown", "ud", dict(resblock_updown=True)), ("unet_tiny_noconv", "nc", dict(conv_resample=False)),
                                            ("unet_tiny_causal", "ca", dict(use_causal_attention=True))])
def test_host_graph_with_sampling_variants_vs_reference_golden(monkeypatch, name, tag, flags, level):
    """resblock_updown=True: ResBlock(down / up) - AvgPool2d / nearest 2x between SiLU and the first convolution (up: the convolution's fused
    gather) and on the skip path - in the place of the resampling convolutions; conv_resample=False: the resampling layers without a
    convolution (their results copied into the up path's concat targets).  Same state-dict keys as the reference's modules."""
    import numpy as np
    import os
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules import attention, flow
    from viewcrafter_amd.lvdm.modules.networks import openaimodel3d as om
    for mod in (flow, attention, om):
        monkeypatch.setattr(mod, "GN_STATS_LEVEL", level, raising=False)
        monkeypatch.setattr(mod, "GN_EPILOGUE_STATS", level >= 1, raising=False)
    m = om.UNetModel(**dict(TINY_UNET, **flags)).eval()
    load_synth(m)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["unet_keys"]]
    x = synth_input(f"unet_{tag}_x", (2, 8, 3, 16, 32))
    ctx = synth_input(f"unet_{tag}_ctx", (2, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([999, 399]), context=ctx, fs=torch.tensor([10, 3]))
    e = rel_l2(y, g["unet_out"])
    print(f"host graph with {flags} (GN statistics level {level}) vs the reference golden: {e:.3e}")
    assert y.shape == g["unet_out"].shape and e <= UNET_TOL


def test_host_graph_with_relative_position_vs_reference_golden(monkeypatch):
    """use_relative_position=True: the embedding tables as parameters of every temporal CrossAttention (the reference's names and shapes), q Ek^T and
    (probabilities by clipped distance) Ev as GEMMs around the temporal attention launcher; temporal_length 2 with 5 frames (clipping)."""
    import numpy as np
    import os
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    m = UNetModel(**dict(TINY_UNET, use_relative_position=True, temporal_length=2)).eval()
    load_synth(m)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_relpos.npz"))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["unet_keys"]]
    assert [str(tuple(m.state_dict()[str(k)].shape)) for k in g["unet_keys"]] == [str(s) for s in g["unet_shapes"]]
    x = synth_input("unet_rp_x", (1, 8, 5, 16, 16))
    ctx = synth_input("unet_rp_ctx", (1, 77 + 40, TINY_UNET["context_dim"]))
    with torch.no_grad():
        y = m(x, torch.tensor([599]), context=ctx, fs=torch.tensor([10]))
    e = rel_l2(y, g["unet_out"])
    print(f"host graph with use_relative_position vs the reference golden: {e:.3e}")
    assert e <= UNET_TOL


def test_host_graph_with_features_adapter_vs_reference_golden(monkeypatch):
    """UNetModel.forward(features_adapter=[...]) (reference openaimodel3d.py:582-588), also under the shared CFG prefix (the maps are
    given once per video and shared by the r evaluations); a list of the wrong length is refused like the reference's assert."""
    import numpy as np
    import os
    from tests.test_oracle_golden import _adapter_features
    m, _ = _unet(monkeypatch, 2)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_tiny_adapter.npz"))
    x = synth_input("unet_ad_x", (1, 8, 3, 16, 32))
    ctx = synth_input("unet_ad_ctx", (1, 77 + 40, TINY_UNET["context_dim"]))
    feats = _adapter_features(1, 3, 16, 32)
    ts, fs = torch.tensor([599]), torch.tensor([10])
    with torch.no_grad():
        y = m(x, ts, context=ctx, fs=fs, features_adapter=feats)
        y2 = m(x, ts, context=torch.cat([ctx, ctx]), fs=fs, features_adapter=feats, cfg_repeat=2)
    e = rel_l2(y, g["unet_out"])
    print(f"host graph with features_adapter vs the reference golden: {e:.3e}")
    assert e <= UNET_TOL and rel_l2(y2[1:], g["unet_out"]) <= UNET_TOL and rel_l2(y2[:1], g["unet_out"]) <= UNET_TOL
    with pytest.raises(ValueError, match="Wrong features_adapter"):        # one map too many: the reference's assert (:588)
        m(x, ts, context=ctx, fs=fs, features_adapter=feats + [feats[0]])
    with pytest.raises(IndexError):                                         # one too few: the reference's list index fails as well
        m(x, ts, context=ctx, fs=fs, features_adapter=feats[:3])


@pytest.mark.parametrize("variant", [dict(temporal_attention=False), dict(temporal_conv=False), dict(addition_attention=False, fs_condition=False)])
def test_host_graph_of_graph_variants_the_shipped_yamls_do_not_use(monkeypatch, variant):
    """Blocks that end in a SpatialTransformer (no temporal attention: its result is COPIED into the next block's concat target),
    ResBlocks without the temporal convolution (the spatial conv 2 writes the target itself), no init_attn / fps embedding: same
    constructor arguments as the reference accepts, checked against the oracle on the same weights."""
    cpu_kernels.install(monkeypatch)
    from viewcrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    hp = dict(TINY_UNET, **variant)
    m = UNetModel(**hp).eval()
    sd = load_synth(m)
    x = synth_input("var_x", (1, 8, 2, 16, 32))
    ctx = synth_input("var_ctx", (1, 77 + 32, TINY_UNET["context_dim"]))
    ts, fs = torch.tensor([459]), torch.tensor([10])
    with torch.no_grad():
        y = m(x, ts, context=ctx, fs=fs)
        ref = O.unet_forward(sd, hp, x, ts, ctx, fs)
    e = rel_l2(y, ref)
    print(f"host graph variant {variant}: rel-L2 vs oracle {e:.3e}")
    assert e <= UNET_TOL


@pytest.mark.parametrize("t,h,w,r", [(3, 24, 40, 2), (2, 8, 24, 3), (5, 16, 16, 1), (1, 32, 32, 2)])
def test_host_graph_at_odd_sizes_and_three_conditionings(monkeypatch, t, h, w, r):
    """Latents whose token counts are not multiples of 8 / 64 at some level (padded attention rows, no moments there), T = 1,
    and r = 3 conditionings (multi-condition CFG) through the shared prefix - against the oracle and the replicated batch."""
    m, sd = _unet(monkeypatch, 2)
    x = synth_input(f"odd_x_{h}", (1, 8, t, h, w))
    ctx = synth_input(f"odd_ctx_{h}", (r, 77 + 48, TINY_UNET["context_dim"]))
    ts, fs = torch.tensor([459]), torch.tensor([10])
    with torch.no_grad():
        y = m(x, ts, context=ctx, fs=fs, cfg_repeat=r) if r > 1 else m(x, ts, context=ctx, fs=fs)
        for i in range(r):
            ref = O.unet_forward(sd, TINY_UNET, x, ts, ctx[i:i + 1], fs)
            assert rel_l2(y[i:i + 1], ref) <= UNET_TOL, (i, rel_l2(y[i:i + 1], ref))
