"""Full-size checks (BASELINE.json configs[3]: 576x1024x25 -> latent 25x72x128, B=2 for CFG) of the individual kernels
through size-independent properties (the whole-forward comparison with the fp32 oracle at these shapes lives in
tests/test_fullconfig_gpu.py; here every tile configuration / tail split / walk order is hit with exact or near-exact
identities):
linearity / delta-kernel identities for the GEMM-conv engine (all tile configurations, tail split, fused upsample),
partition-of-unity for attention (V = 1 => O = 1), zero-mean/unit-variance for GroupNorm/LayerNorm, and algebraic identities
of the DDIM update."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, T, H, W = 2, 25, 72, 128


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("M,N,K", [(B * T * H * W, 320, 320), (B * T * H * W // 4, 640, 640), (110000, 320, 128), (100000, 512, 64),
                                   (B * T * H * W, 960, 320),
                                   # >= 16 column tiles: the 8-row panel walk, with a ragged last panel (113 row tiles)
                                   (B * T * H * W // 16, 5120, 128)])
def test_gemm_linearity_and_subsample_exactness_at_full_size(M, N, K):
    """(x1 + x2) W^T == x1 W^T + x2 W^T up to fp16 rounding, and random rows agree with an fp32 matmul: exercises the
    256x{256,320} tiles, the tail split onto small tiles and the persistent tile walk (row-major and panel) at the real M."""
    from viewcrafter_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x1 = torch.randn(M, K, device=DEV, generator=g).half()
    x2 = torch.randn(M, K, device=DEV, generator=g).half()
    w = (torch.randn(N, K, device=DEV, generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, device=DEV, generator=g)
    y1, y2 = ops.linear(x1, w, b, out_f32=True), ops.linear(x2, w, None, out_f32=True)
    xs = (x1.float() + x2.float()).half()
    ys = ops.linear(xs, w, b, out_f32=True)
    idx = torch.randint(0, M, (4096,), device=DEV, generator=g)
    idx[:3] = torch.tensor([0, M - 1, M - 129], device=DEV)          # first row, last row, a row of the tail split
    ref = xs[idx].float() @ w.float().t() + b
    assert rel(ys[idx], ref) < 2e-6
    assert rel(ys, y1 + y2) < 2e-3                                   # xs is rounded to fp16: linear up to that rounding
    assert torch.isfinite(ys).all()


@pytest.mark.parametrize("C,h,w,ups", [(320, 72, 128, 0), (640, 36, 64, 0), (640, 36, 64, 1), (1280, 18, 32, 1)])
def test_conv_delta_kernel_is_a_shift_at_full_size(C, h, w, ups):
    """A 3x3 kernel that is 1 at tap (ky,kx) on the channel diagonal reproduces the (zero-padded, optionally
    nearest-upsampled) input shifted by that tap — bit-exact, at the real level shapes, for every tap."""
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import pack_conv
    n = B * T
    x = torch.randn(n, h, w, C, device=DEV).half()
    xu = x if not ups else x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    for ky, kx in ((0, 0), (1, 1), (2, 1), (0, 2)):
        wt = torch.zeros(C, C, 3, 3, device=DEV)
        wt[torch.arange(C), torch.arange(C), ky, kx] = 1.0
        y = ops.conv2d(x, pack_conv(wt.half()), None, kh=3, kw=3, ups=ups)
        ref = F.pad(xu, (0, 0, 1, 1, 1, 1))[:, ky:ky + xu.shape[1], kx:kx + xu.shape[2], :]
        assert torch.equal(y, ref), (ky, kx)


def test_temporal_conv_delta_kernel_shifts_frames():
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import pack_conv
    C, P = 320, H * W
    x = torch.randn(B, T, P, C, device=DEV).half()
    for kt in range(3):
        wt = torch.zeros(C, C, 3, 1, 1, device=DEV)
        wt[torch.arange(C), torch.arange(C), kt] = 1.0
        y = ops.temporal_conv3(x, pack_conv(wt.half()), None)
        ref = F.pad(x, (0, 0, 0, 0, 1, 1))[:, kt:kt + T]
        assert torch.equal(y, ref), kt


@pytest.mark.parametrize("C,h,w", [(320, 72, 128), (640, 36, 64), (1280, 18, 32)])
def test_attention_partition_of_unity_at_full_size(C, h, w):
    """softmax rows sum to one: with V = const the output is that constant for every query, head and frame (flash self-
    attention with both query-block variants, and temporal attention)."""
    from viewcrafter_amd import ops
    heads, N, G = C // 64, h * w, B * T
    qk = torch.randn(G * N, 2 * C, device=DEV).half()
    vt = torch.full((C, G * N), 0.75, device=DEV, dtype=torch.float16)
    out = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    ops.flash_attn(qk, qk[:, C:], vt, out, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C,
                   ldvt=G * N, ldo=C, scale=0.125)
    assert float((out.float() - 0.75).abs().max()) <= 1e-3
    qkv = torch.randn(G * N, 3 * C, device=DEV).half()
    qkv[:, 2 * C:] = -1.25
    o2 = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    ops.temporal_attn(qkv, o2, B=B, T=T, P=N, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, ldo=C, scale=0.125)
    assert float((o2.float() + 1.25).abs().max()) <= 2e-3


@pytest.mark.parametrize("C,h,w", [(320, 72, 128), (1920, 36, 64)])
def test_groupnorm_and_layernorm_moments_at_full_size(C, h, w):
    from viewcrafter_amd import ops
    n = B * T
    x = (torch.randn(n, h * w, C, device=DEV) * 3 + 1.5).half()
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    for view in (x, x.view(B, T * h * w, C)):                           # per-frame and per-video statistics
        y = ops.group_norm(view, one, zero, 1e-5, False).float().view(view.shape[0], view.shape[1], 32, C // 32)
        m, v = y.mean(dim=(1, 3)), y.var(dim=(1, 3), unbiased=False)
        assert float(m.abs().max()) < 2e-3 and float((v - 1).abs().max()) < 4e-3
    y = ops.layer_norm(x.view(-1, C), one, zero, 1e-5).float()
    assert float(y.mean(-1).abs().max()) < 2e-3 and float((y.var(-1, unbiased=False) - 1).abs().max()) < 5e-3


def test_ddim_update_identities_at_full_size():
    """(i) equal cond/uncond predictions make guidance and its rescale the identity; (ii) sigma = 0, a_prev = 1, ratio 1 returns
    pred_x0; (iii) x0 and eps recombine to x: sqrt(a) x0 + sqrt(1-a) eps == x for the v-parameterisation."""
    from viewcrafter_amd import ops
    x = torch.randn(1, 4, T, H, W, device=DEV)
    v = torch.randn(1, 4, T, H, W, device=DEV)
    a = 0.37
    sa, s1 = math.sqrt(a), math.sqrt(1 - a)
    xp1, x01 = ops.ddim_step(x, v, v.clone(), None, [sa, s1, 0.6, 0.0, 1.0, 7.5, 0.7, 1.0])
    xp2, x02 = ops.ddim_step(x, v, None, None, [sa, s1, 0.6, 0.0, 1.0, 1.0, 0.0, 1.0])
    assert float((x01 - x02).abs().max()) < 1e-5 and float((xp1 - xp2).abs().max()) < 1e-5
    xp3, x03 = ops.ddim_step(x, v, None, None, [sa, s1, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0])
    assert float((xp3 - x03).abs().max()) < 1e-6
    eps = sa * v + s1 * x
    assert float((sa * x03 + s1 * eps - x).abs().max()) < 1e-5
