"""Bit-reproducibility soak (round 4).  Round 3 found two run-dependent results in ONE template instantiation (the 128x128 LNFOLD_T
epilogue, profiles/r03_experiments.md section 11 / profiles/r04_pkfma_rootcause.md) and a ten-call loop over small shapes was the only
run-time guard.  Here:

  * every `gemm_dma_kernel` instantiation the dispatcher can reach - 4 tile configurations x {linear, conv} x {plain (+ bias, residual,
    per-image addend), fp32 output, GEGLU, LNFOLD, LNFOLD + GEGLU, LNFOLD_T, COLSTATS} = 34 kernels - 200 calls each on a problem
    that fills the persistent grid several times over, every call compared bit for bit with the first (also the column moments);
  * the attention / normalisation kernels at the benchmark's own shapes, 50 calls each;
  * the whole B = 2 (cond + uncond) UNet forward at the headline latent 25x72x128, 20 runs, identical bits - with and without the
    shared CFG prefix.

All reductions in libvcx have a fixed order and there are no floating-point atomics, so ANY differing bit is a defect."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = int(os.environ.get("VCX_SOAK_CALLS", "200"))


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def _soak(fn, calls, what):
    first = fn()
    first = tuple(t.clone() for t in (first if isinstance(first, tuple) else (first,)))
    torch.cuda.synchronize()
    for i in range(1, calls):
        out = fn()
        out = out if isinstance(out, tuple) else (out,)
        for a, b in zip(out, first):
            if not torch.equal(a, b):
                nd = int((a != b).sum())
                idx = (a != b).nonzero()[:8].tolist()
                pytest.fail(f"{what}: call {i} differs from call 0 in {nd} elements, first at {idx}")
    assert all(torch.isfinite(t.float()).all() for t in first), what


# variant -> (conv?, needs NF % 4 == 0 i.e. tile configs 0 / 2 only)
VARIANTS = {"plain": (False, False), "f32": (False, False), "geglu": (False, True), "lnfold": (False, False), "lnfold_geglu": (False, True),
            "lnfold_t": (False, False), "conv": (True, False), "conv_f32": (True, False), "conv_geglu": (True, True),
            "conv_colstats": (True, False)}


_DATA = {}


def _gemm_data():
    """Operands shared by the 34 cases (built once: 33 k x 640 activations, a 3x3 convolution input with 65-pixel rows)."""
    if not _DATA:
        from viewcrafter_amd.packing import pack_conv
        # the persistent grid is filled at least twice in every configuration (128-row tiles: 2 blocks / CU, 256-row tiles: 1 block / CU)
        M, K, N = 32768 + 128, 640, 1280           # 2 (256x320) ... 5 (128x128) rounds; ragged last tile rows in the 256-row configurations
        n_img, Hh, Ww, cin = 16, 32, 64 + 1, 128   # M = 33280 image rows, 65 columns: border taps in every tile
        w32 = (rnd(N, K, seed=2) / math.sqrt(K)).to(DEV)
        _DATA.update(M=M, K=K, N=N, n_img=n_img, Hh=Hh, Ww=Ww, x=(rnd(M, K, seed=1) * 2 + 0.5).to(DEV).half(), bias=rnd(N, seed=3).to(DEV),
                     res=rnd(M, N, seed=4).to(DEV).half(), gamma=(1 + 0.3 * rnd(K, seed=5)).to(DEV), beta=(0.2 * rnd(K, seed=6)).to(DEV),
                     w32=w32, w=w32.half(), xi=rnd(n_img, Hh, Ww, cin, seed=7).to(DEV).half(),
                     wc=pack_conv(rnd(N, cin, 3, 3, seed=8) / math.sqrt(9 * cin)).to(DEV).half(), rowadd=rnd(n_img, N, seed=9).to(DEV),
                     resc=rnd(n_img * Hh * Ww, N, seed=10).to(DEV).half())
    return _DATA


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_every_dma_gemm_instantiation_is_bit_reproducible(cfg, variant):
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import fold_layernorm, pack_geglu
    conv, needs_nf4 = VARIANTS[variant]
    if needs_nf4 and cfg in (1, 3):
        pytest.skip("GEGLU needs whole 64-column blocks per wave: tile configurations 128x128 and 256x256 only")
    d = _gemm_data()
    M, N, K, n_img, Hh, Ww = d["M"], d["N"], d["K"], d["n_img"], d["Hh"], d["Ww"]
    x, bias, res, gamma, beta, w32, w = d["x"], d["bias"], d["res"], d["gamma"], d["beta"], d["w32"], d["w"]
    xi, wc, rowadd, resc = d["xi"], d["wc"], d["rowadd"], d["resc"]
    prev = ops.tune_set("GEMM_CFG", cfg)
    try:
        if variant == "plain":
            fn = lambda: ops.linear(x, w, bias, residual=res)
        elif variant == "f32":
            fn = lambda: ops.linear(x, w, bias, out_f32=True)
        elif variant == "geglu":
            wg, bg = pack_geglu(w, bias)
            fn = lambda: ops.linear(x, wg, bg, geglu=True)
        elif variant in ("lnfold", "lnfold_geglu", "lnfold_t"):
            wf, colsum, bias_f = fold_layernorm(w32, gamma, beta, bias if variant != "lnfold_t" else None)
            st = ops.row_stats(x, 1e-5)
            if variant == "lnfold":
                fn = lambda: ops.linear(x, wf, bias_f, residual=res, ln_stats=st, ln_colsum=colsum)
            elif variant == "lnfold_geglu":
                wg, bg = pack_geglu(wf, bias_f)
                cg = pack_geglu(wf, colsum)[1]
                fn = lambda: ops.linear(x, wg, bg, geglu=True, ln_stats=st, ln_colsum=cg)
            else:
                fn = lambda: ops.gemm(wf, x, M=N, N=M, K=K, lda=K, bias=bias_f, bias_m=True, ln_stats=st, ln_colsum=colsum, ln_t=True)
        elif variant == "conv":
            fn = lambda: ops.conv2d(xi, wc, bias, kh=3, kw=3, residual=resc, rowadd=rowadd, rowadd_div=Hh * Ww)
        elif variant == "conv_f32":
            fn = lambda: ops.conv2d(xi, wc, bias, kh=3, kw=3, out_f32=True)
        elif variant == "conv_geglu":
            wg, bg = pack_geglu(wc, bias)
            fn = lambda: ops.conv2d(xi, wg, bg, kh=3, kw=3, geglu=True)
        else:
            xs = xi[:, :, :64].contiguous()            # COLSTATS: whole 64-row strips per frame (32 x 64 pixels)
            Ms = n_img * Hh * 64
            def fn():
                cs = ops.colstats_buffer(Ms, N, DEV)
                y = ops.conv2d(xs, wc, bias, kh=3, kw=3, rowadd=rowadd, rowadd_div=Hh * 64, colstats=cs)
                return y, cs
        _soak(fn, CALLS, f"gemm_dma cfg {cfg} {variant}")
    finally:
        ops.tune_set("GEMM_CFG", prev)


def test_attention_and_norm_kernels_are_bit_reproducible_at_the_benchmark_shapes():
    from viewcrafter_amd import ops
    calls = max(CALLS // 4, 10)
    # spatial self-attention of level 0 (flash2, 9216 keys) and level 1 (phased kernel, 2304 keys), 4 frames x 5 / 10 heads
    for N_img, D in ((9216, 320), (2304, 640)):
        n, heads = 4, D // 64
        tokens = n * N_img
        qk = rnd(tokens, 2 * D, seed=21).to(DEV).half()
        vt = rnd(D, tokens, seed=22).to(DEV).half()

        def attn():
            o = torch.empty((tokens, D), dtype=torch.float16, device=DEV)
            return ops.flash_attn(qk, qk[:, D:], vt, o, n_groups=n, heads=heads, nq=N_img, nk=N_img, kv_rows=N_img, kv_div=1, ldq=2 * D,
                                  ldk=2 * D, ldvt=tokens, ldo=D, scale=0.125, log2_logits=True)
        _soak(attn, calls, f"flash attention {N_img} keys")
    # temporal attention, level 0: 25 frames, 2304 pixels
    B, T, P, D = 1, 25, 2304, 320
    qkv = rnd(B * T * P, 3 * D, seed=23).to(DEV).half()

    def tattn():
        o = torch.empty((B * T * P, D), dtype=torch.float16, device=DEV)
        return ops.temporal_attn(qkv, o, B=B, T=T, P=P, heads=D // 64, ld=3 * D, k_off=D, v_off=2 * D, ldo=D, scale=0.125)
    _soak(tattn, calls, "temporal attention")
    x = (rnd(2, 25 * 2304, 320, seed=24) * 2 + 0.7).to(DEV).half()
    g, b = (1 + 0.2 * rnd(320, seed=25)).to(DEV), (0.1 * rnd(320, seed=26)).to(DEV)
    _soak(lambda: ops.group_norm(x, g, b, 1e-5, True), calls, "groupnorm per video")
    xr = x.view(-1, 320)
    _soak(lambda: ops.layer_norm(xr, g, b, 1e-5), calls, "layernorm")
    _soak(lambda: ops.row_stats(xr, 1e-5), calls, "row statistics")


def test_round5_kernels_are_bit_reproducible_at_the_benchmark_shapes():
    """The kernels of round 5 at the shapes the benchmark gives them, under a full chip: the weight-stationary K = 320 kernel in its plain,
    residual, GEGLU and one-weight-set-per-frame / per-video forms, the folded-GroupNorm weight builder, both forms of the resident
    cross-attention kernel (level 0: 9216 queries x 5 heads per frame, 77 + 256 keys per video) and the d = 512 flash kernel of the
    VAE."""
    from viewcrafter_amd import ops
    calls = max(CALLS // 4, 10)
    M, C = 25 * 9216, 320
    x = rnd(M, C, seed=31).to(DEV).half()
    w = (rnd(C, C, seed=32) / math.sqrt(C)).to(DEV).half()
    bias = rnd(C, seed=33).to(DEV)
    res = rnd(M, C, seed=34).to(DEV).half()
    _soak(lambda: ops.linear(x, w, bias), calls, "weight-stationary linear")
    _soak(lambda: ops.linear(x, w, bias, residual=res), calls, "weight-stationary linear + residual")
    # the weight-stationary GEGLU projection of level 0 (460800 x 2560 x 320 in the benchmark: here one video, ten column blocks per row stream)
    from viewcrafter_amd.packing import pack_geglu
    wg, bg = pack_geglu((rnd(2560, C, seed=38) / math.sqrt(C)).to(DEV), rnd(2560, seed=39).to(DEV))
    wg, bg = wg.half(), bg.float().contiguous()
    assert ops.tune_get("GEMM_WS") == 1
    _soak(lambda: ops.linear(x, wg, bg, geglu=True), max(calls // 2, 10), "weight-stationary GEGLU projection")
    # ... and the LayerNorm-folded q | k | v projection (N = 960: four column blocks per row stream, the last three quarters full)
    wq = (rnd(960, C, seed=61) / math.sqrt(C)).to(DEV).half()
    bq, csq = rnd(960, seed=62).to(DEV), (0.01 * rnd(960, seed=63)).to(DEV)
    st = ops.row_stats(x, 1e-5)
    _soak(lambda: ops.linear(x, wq, bq, ln_stats=st, ln_colsum=csq), max(calls // 2, 10), "weight-stationary LayerNorm-folded projection")
    stats = ops.group_norm_stats(x.view(25, 9216, C))
    g, b = (1 + 0.2 * rnd(C, seed=35)).to(DEV), (0.1 * rnd(C, seed=36)).to(DEV)
    w32 = (rnd(C, C, seed=37) / math.sqrt(C)).to(DEV)
    _soak(lambda: ops.group_norm_fold_linear(w32, bias, g, b, stats, 1e-6), calls, "GroupNorm fold: weight / bias sets")
    wn, bn = ops.group_norm_fold_linear(w32, bias, g, b, stats, 1e-6)
    _soak(lambda: ops.gemm_units(x, wn, bn, unit_rows=9216), calls, "one weight set per frame (25 x 9216 rows)")
    _soak(lambda: ops.gemm_units(x, wn[:5].contiguous(), bn[:5].contiguous(), unit_rows=5 * 9216), calls, "one weight set per 5 frames")
    # resident cross-attention, level 0 of one video
    T, nq, heads = 25, 9216, 5
    q = (rnd(T * nq, C, seed=41) * 0.18).to(DEV).half()
    kt = torch.zeros(80, C); kt[:77] = rnd(77, C, seed=42); kt = kt.to(DEV).half()
    vt = torch.zeros(80, C); vt[:77] = rnd(77, C, seed=43); vt_t = vt.to(DEV).half().t().contiguous()
    ki = rnd(256, C, seed=44).to(DEV).half(); vi_t = rnd(256, C, seed=45).to(DEV).half().t().contiguous()
    for form in (1, 2):
        prev = ops.tune_set("XATTN_RESIDENT", form)
        try:
            def xattn():
                o = torch.empty((T * nq, C), dtype=torch.float16, device=DEV)
                return ops.flash_attn_dual(q, kt, vt_t, ki, vi_t, o, n_groups=T, heads=heads, nq=nq, nk1=77, kv_rows1=80, kv_div1=T, ldk1=C, ldvt1=80,
                                           nk2=256, kv_rows2=256, kv_div2=T, ldk2=C, ldvt2=256, ldq=C, ldo=C, scale=0.125, log2_logits=True)
            _soak(xattn, calls, f"resident cross-attention, form {'second' if form == 1 else 'first'}")
        finally:
            ops.tune_set("XATTN_RESIDENT", prev)
    # VAE AttnBlock: 4 frames x 9216 tokens x 512
    n, N, Cv = 4, 9216, 512
    qv, kv = rnd(n * N, Cv, seed=51).to(DEV).half(), rnd(n * N, Cv, seed=52).to(DEV).half()
    vvt = rnd(Cv, n * N, seed=53).to(DEV).half()

    def vae_attn():
        o = torch.empty((n * N, Cv), dtype=torch.float16, device=DEV)
        return ops.flash_attn_d512(qv, kv, vvt, o, n_groups=n, nq=N, nk=N, kv_rows=N, ldq=Cv, ldk=Cv, ldvt=n * N, ldo=Cv, scale=Cv ** -0.5)
    _soak(vae_attn, max(calls // 5, 5), "flash attention, one head of 512")


def test_round6_kernels_are_bit_reproducible_at_the_benchmark_shapes():
    """The kernels of round 6 under a full chip: the weight-stationary layer WITH row statistics (output and statistics; its merge of the
    four strips of a row passes through LDS between waves), the one-launch GroupNorm statistics (from column moments and the direct
    statistics pass), the GroupNorm apply over a split concat, and the 3x3 convolution with a two-source K tail and column moments."""
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import pack_conv
    calls = max(CALLS // 4, 10)
    M, C = 25 * 9216, 320
    x = rnd(M, C, seed=71).to(DEV).half()
    w = (rnd(C, C, seed=72) / math.sqrt(C)).to(DEV).half()
    bias = rnd(C, seed=73).to(DEV)
    res = rnd(M, C, seed=74).to(DEV).half()

    def with_rowstats(residual):
        st = torch.empty(M, 2, device=DEV)
        return ops.linear(x, w, bias, residual=residual, rowstats=st), st
    _soak(lambda: with_rowstats(res), calls, "weight-stationary linear + residual + row statistics")
    _soak(lambda: with_rowstats(None), calls, "weight-stationary linear + row statistics")
    wn = (rnd(25, C, C, seed=75) / math.sqrt(C)).to(DEV).half()
    bn = rnd(25, C, seed=76).to(DEV)

    def units_rowstats():
        st = torch.empty(M, 2, device=DEV)
        return ops.gemm_units(x, wn, bn, unit_rows=9216, rowstats=st), st
    _soak(units_rowstats, calls, "one weight set per frame + row statistics")
    # GroupNorm statistics: column moments of 25 frames x 9216 pixels x 320 (144 strips per frame), of 2 videos x 900 strips x 640, and the
    # direct statistics pass at the 9 x 16-pixel level
    cs = ops.colstats_buffer(M, C, DEV)
    ops.linear(x, w, bias, colstats=cs)
    _soak(lambda: ops.group_norm_stats_from_colstats(cs, 25, 9216, C), calls, "GroupNorm statistics from column moments, per frame")
    cs2 = torch.rand(2 * 900, 640, 2, device=DEV)
    _soak(lambda: ops.group_norm_stats_from_colstats(cs2, 2, 57600, 640), calls, "GroupNorm statistics from column moments, per video")
    x3 = rnd(50, 144, 1280, seed=77).to(DEV).half()
    _soak(lambda: ops.group_norm_stats(x3), calls, "GroupNorm statistics pass, direct form")
    x3v = x3.view(2, 3600, 1280)
    _soak(lambda: ops.group_norm_stats(x3v), calls, "GroupNorm statistics pass, 16-pixel chunks")
    # the in_layers norm of an up-path ResBlock over [h | skip] in place, level 0: 640 + 320 channels
    h, skip = rnd(25, 9216, 640, seed=78).to(DEV).half(), rnd(25, 9216, 320, seed=79).to(DEV).half()
    g, b = (1 + 0.2 * rnd(960, seed=80)).to(DEV), (0.1 * rnd(960, seed=81)).to(DEV)
    st960 = ops.group_norm_stats(torch.cat([h, skip], dim=2).contiguous())
    _soak(lambda: ops.group_norm(h, g, b, 1e-5, True, stats=st960, x2=skip), max(calls // 2, 10), "GroupNorm apply over a split concat")
    # ... and that block's second convolution with the skip convolution as a K tail: 25 x 72 x 128 x 320, K = 2880 + 640 + 320, column moments
    a = rnd(25, 72, 128, C, seed=82).to(DEV).half()
    wcat = torch.cat([pack_conv(rnd(C, C, 3, 3, seed=83) / math.sqrt(9 * C)), rnd(C, 960, seed=84) / math.sqrt(960)], dim=1).to(DEV).half().contiguous()
    cs3 = ops.colstats_buffer(M, C, DEV)
    hv, sv = h.view(M, 640), skip.view(M, 320)
    _soak(lambda: (ops.conv2d(a, wcat, bias, kh=3, kw=3, tail=[hv, sv], colstats=cs3), cs3), max(calls // 5, 5), "3x3 convolution with a two-source K tail + column moments")


@pytest.mark.parametrize("share_prefix", [True, False])
def test_b2_unet_forward_at_25x72x128_is_bit_reproducible_over_20_runs(share_prefix):
    """The cond + uncond evaluation of one DDIM step at the headline latent, as the sampler launches it (B = 2, with the shared
    CFG prefix and as a plain batched forward): 20 runs, identical bits."""
    from tests.test_fullconfig_gpu import _inputs, _model
    model, _ = _model("inference_pvd_1024.yaml")
    unet = model.model.diffusion_model
    T, h, w = 25, 72, 128
    x, ctx = _inputs(T, h, w, seed=99, B=2)
    ts, fs = torch.tensor([599, 599], device=DEV), torch.tensor([10, 10], device=DEV)
    if share_prefix:
        x = x[:1].contiguous()
        ts, fs = ts[:1], fs[:1]

    def fwd():
        with torch.no_grad():
            if share_prefix:
                return unet._forward(x, ts, context=ctx, fs=fs, cfg_repeat=2)
            return unet(x, ts, context=ctx, fs=fs)
    runs = int(os.environ.get("VCX_SOAK_FORWARDS", "20"))
    _soak(fwd, runs, f"B=2 UNet forward 25x72x128 (shared prefix: {share_prefix})")
    torch.cuda.empty_cache()
