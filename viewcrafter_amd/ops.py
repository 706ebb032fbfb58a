"""Thin torch-tensor front end of the libvcx C ABI.

Tensors are only used for device memory and streams; every function below launches one (or two)
hand-written gfx950 kernels through ctypes.  Activations are fp16 channels-last, i.e. a video
latent is [B, T, H, W, C] and all token matrices are [rows, C] views of it.

Which reference call sites each entry point stands in for (nn.Linear / Conv2d / Conv3d attention.py:53-57,
openaimodel3d.py:69-186, GroupNorm basics.py:76-81, LayerNorm attention.py:226-228, memory_efficient_attention
attention.py:175,187, the DDIM update ddim.py:228-279, ...) is tabulated in INTEGRATION.md section 2 and include/vcx.h.
"""
import ctypes
import os

import torch

from ._lib import (GEMM_BIAS_M, GEMM_BIAS_N, GEMM_COLSTATS, GEMM_CONV_SLABK, GEMM_GEGLU, GEMM_LNFOLD, GEMM_LNFOLD_T, GEMM_OUT_F32, GEMM_RESIDUAL, GEMM_ROWADD, GEMM_ROWSTATS, PROF_FAMILIES,
                   TUNE, GemmDesc, VcxError, check, lib)
from .packing import conv_slab_major

_f16 = torch.float16
_f32 = torch.float32
# LayerNorm statistics from the producing layer's epilogue (VCX_GEMM_ROWSTATS, round 6); 0 = a statistics pass everywhere (A/B runs)
LN_ROWSTATS = os.environ.get("VCX_LN_ROWSTATS", "1") != "0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _dev16(*tensors):
    """The C ABI takes raw pointers: a host tensor or a wrong element type would be read as garbage, not rejected."""
    for t in tensors:
        if t is not None and (t.dtype is not _f16 or not t.is_cuda):
            raise VcxError(f"expected a GPU fp16 tensor, got {t.dtype} on {t.device}")


def _dev32(*tensors):
    for t in tensors:
        if t is not None and (t.dtype is not _f32 or not t.is_cuda):
            raise VcxError(f"expected a GPU fp32 tensor, got {t.dtype} on {t.device}")


def require_gpu():
    """Raise unless a gfx950 device and the built library are available (no fallback path)."""
    if not torch.cuda.is_available():
        raise VcxError("viewcrafter_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                       "and there is no CPU fallback")
    lib()


# ------------------------------------------------------------------------------------------
# GEMM / convolution
# ------------------------------------------------------------------------------------------
def gemm(a, w, *, M, N, K, lda, out=None, ldc=None, bias=None, bias_m=False, residual=None, ldr=None, rowadd=None,
         rowadd_div=0, geglu=False, out_f32=False, alpha=1.0, conv=None, ldw=None, ln_stats=None, ln_colsum=None, ln_t=False, colstats=None,
         colstats_ld=None, colstats_col=0, rowstats=None, rowstats_eps=1e-5, tail=None):
    """out[M, N] = epilogue(alpha * X W^T); see include/vcx.h.  `conv` = dict(in_h, in_w, out_h, out_w, cin, kh, kw,
    stride, pad_h, pad_w, ups) switches X to the im2col gather of a channels-last image.  `ln_stats` (from row_stats) +
    `ln_colsum` select the folded-LayerNorm epilogue (VCX_GEMM_LNFOLD; `ln_t`: the normalised rows are the W operand).
    `colstats` (fp32 [M / 64, N, 2], see colstats_buffer) makes the layer write the column moments of its output for the
    GroupNorm behind it (VCX_GEMM_COLSTATS); with `colstats_ld` / `colstats_col` the buffer is [M / 64, colstats_ld, 2] and this
    call fills the columns [colstats_col, colstats_col + N) - the moments of a tensor that is one part of a channel concat.
    `rowstats` (fp32 [M, 2], see rowstats_ok) makes the layer write LayerNorm's (mean, rstd) of its output rows (VCX_GEMM_ROWSTATS).
    `tail` (convolutions): one or two fp16 tensors [M, k_j] whose rows are the last sum(k_j) K columns of the problem - K then counts
    them (include/vcx.h tail_a0 / tail_a1; see conv_tail_ok)."""
    n_out = N // 2 if geglu else N
    _dev16(a, w, residual)
    _dev32(bias, rowadd)
    if out is None:
        out = torch.empty((M, n_out), dtype=_f32 if out_f32 else _f16, device=a.device)
        ldc = n_out
    elif ldc is None:
        ldc = out.stride(0)
    d = GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    flags = 0
    if bias is not None:
        d.bias = bias.data_ptr()
        flags |= GEMM_BIAS_M if bias_m else GEMM_BIAS_N
    if rowadd is not None:
        if rowadd.dim() != 2 or rowadd.stride(1) != 1 or rowadd.shape[1] != n_out:
            raise VcxError(f"rowadd must be [rows, N = {n_out}] fp32 with unit column stride (a column slice of a wider matrix is fine), got {tuple(rowadd.shape)}")
        d.rowadd = rowadd.data_ptr()
        d.rowadd_div = rowadd_div
        d.rowadd_ld = rowadd.stride(0) if rowadd.shape[0] > 1 else n_out
        flags |= GEMM_ROWADD
    if residual is not None:
        d.residual = residual.data_ptr()
        d.ldr = ldr if ldr is not None else residual.stride(0)
        flags |= GEMM_RESIDUAL
    if geglu:
        flags |= GEMM_GEGLU
    if out_f32:
        flags |= GEMM_OUT_F32
    if ln_stats is not None:
        _dev32(ln_stats, ln_colsum)
        d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), ln_colsum.data_ptr()
        flags |= GEMM_LNFOLD_T if ln_t else GEMM_LNFOLD
    if colstats is not None:
        _dev32(colstats)
        cld = n_out if colstats_ld is None else int(colstats_ld)
        if colstats.numel() != (M // 64) * cld * 2 or colstats_col < 0 or colstats_col + n_out > cld:
            raise VcxError(f"colstats must hold [M / 64, {cld}, 2] floats (M={M}, N={n_out}, first column {colstats_col}), got {tuple(colstats.shape)}")
        d.colstats = colstats.data_ptr() + 8 * int(colstats_col)
        d.ldcs = cld
        flags |= GEMM_COLSTATS
    if rowstats is not None:
        _dev32(rowstats)
        if rowstats.numel() != 2 * M or not rowstats.is_contiguous():
            raise VcxError(f"rowstats must hold [M = {M}, 2] floats, got {tuple(rowstats.shape)}")
        d.rowstats, d.rowstats_eps = rowstats.data_ptr(), float(rowstats_eps)
        flags |= GEMM_ROWSTATS
    if tail:
        if conv is None or len(tail) > 2:
            raise VcxError("gemm: a K tail belongs to a convolution and has one or two sources")
        _dev16(*tail)
        for j, t in enumerate(tail):
            if t.dim() != 2 or t.shape[0] != M or t.stride(1) != 1:
                raise VcxError(f"gemm: tail source {j} must be [M = {M}, k] with unit column stride, got {tuple(t.shape)}")
            if j == 0:
                d.tail_a0, d.tail_lda0, d.tail_k0 = t.data_ptr(), t.stride(0), t.shape[1]
            else:
                d.tail_a1, d.tail_lda1, d.tail_k1 = t.data_ptr(), t.stride(0), t.shape[1]
    d.lda, d.M, d.N, d.K = lda, M, N, K
    d.ldw = ldw if ldw is not None else K
    d.ldc = ldc
    if conv is not None:
        d.mode = 1
        d.in_h, d.in_w, d.out_h, d.out_w = conv["in_h"], conv["in_w"], conv["out_h"], conv["out_w"]
        d.cin, d.kh, d.kw = conv["cin"], conv["kh"], conv["kw"]
        d.stride, d.pad_h, d.pad_w, d.ups = conv["stride"], conv["pad_h"], conv["pad_w"], conv.get("ups", 0)
        if conv.get("slabk", conv_slab_major(conv["cin"], conv["kh"] * conv["kw"])):     # the order pack_conv() produced
            flags |= GEMM_CONV_SLABK
    d.flags = flags
    d.alpha = alpha
    check(lib().vcx_gemm_f16(ctypes.byref(d), _stream()), "vcx_gemm_f16")
    return out


def rowstats_ok(M, N, K, *, lda=None, ldc=None, ldr=0, unit_rows=None):
    """Will the layer producing an [M, N] output - vcx_gemm_f16 (bias / residual at most), or vcx_gemm_units_f16 with `unit_rows` rows per
    weight set - write LayerNorm's row statistics of its output (VCX_GEMM_ROWSTATS)?  Only the pipelined weight-stationary kernel has
    that epilogue: a mirror of its dispatch conditions in csrc/gemm.hip (N = K = 320, M >= 8192, 32-bit extents, knobs GEMM_DMA /
    GEMM_WS on and no forced tile configuration); elsewhere the consumer makes its statistics pass (row_stats)."""
    lim = 0xFFFF0000
    lda, ldc = K if lda is None else lda, N if ldc is None else ldc
    ok = (LN_ROWSTATS and N == 320 and K == 320 and M >= 8192 and tune_get("GEMM_DMA") != 0 and tune_get("GEMM_WS") != 0 and tune_get("GEMM_CFG") < 0
          and 2 * ((M - 1) * lda + K) < lim and 2 * (M + 256) * ldc < lim and 2 * (M + 256) * ldr < lim and 8 * M < lim)
    if unit_rows is not None:
        ok = ok and M // unit_rows > 1 and M // unit_rows <= 65535 and unit_rows % 32 == 0 and unit_rows >= 1024
    return ok


def rowstats_buffer(M, device):
    return torch.empty((M, 2), dtype=_f32, device=device)


def gemm_units(a, wn, bn, *, unit_rows, out=None, rowstats=None, rowstats_eps=1e-5):
    """out[M, N] = a[M, K] Wn[u]^T + bn[u] for the rows of unit u = m // unit_rows: wn [units, N, K] fp16, bn [units, N] fp32 (the
    sets of group_norm_fold_linear).  One launch of the weight-stationary kernel where it applies, else unit by unit (include/vcx.h).
    `rowstats`: as in gemm (one-launch form only, see rowstats_ok)."""
    M, K = a.shape
    units, N, K2 = wn.shape
    _dev16(a, wn, out)
    _dev32(bn)
    if K2 != K or M != units * unit_rows or tuple(bn.shape) != (units, N) or not wn.is_contiguous() or not bn.is_contiguous():
        raise VcxError(f"gemm_units: a {tuple(a.shape)} / wn {tuple(wn.shape)} / bn {tuple(bn.shape)} do not describe {units} units of {unit_rows} rows")
    if out is None:
        out = torch.empty((M, N), dtype=_f16, device=a.device)
    d = GemmDesc()
    d.A, d.W, d.C, d.bias = a.data_ptr(), wn.data_ptr(), out.data_ptr(), bn.data_ptr()
    d.lda, d.M, d.N, d.K, d.ldw, d.ldc = a.stride(0), M, N, K, K, out.stride(0)
    d.flags, d.alpha = GEMM_BIAS_N, 1.0
    if rowstats is not None:
        _dev32(rowstats)
        if rowstats.numel() != 2 * M or not rowstats.is_contiguous():
            raise VcxError(f"rowstats must hold [M = {M}, 2] floats, got {tuple(rowstats.shape)}")
        d.rowstats, d.rowstats_eps = rowstats.data_ptr(), float(rowstats_eps)
        d.flags |= GEMM_ROWSTATS
    check(lib().vcx_gemm_units_f16(ctypes.byref(d), int(unit_rows), N * K, N, _stream()), "vcx_gemm_units_f16")
    return out


def lnfold_ok(rows, n_out, K, *, lda=None, ldc=None, transposed=False):
    """Will vcx_gemm_f16 take a folded-LayerNorm projection (VCX_GEMM_LNFOLD / _T) of `rows` token rows x K channels onto n_out
    outputs?  The epilogue exists in the DMA kernel only; this mirrors its preconditions in csrc/gemm.hip (`dma_ok`: knob
    GEMM_DMA, K % 64 == 0, GEMM N % 8 == 0, every extent and the output offsets up to 256 rows past the end below 4 GiB).  Where it
    says no - a longer clip or a larger frame than any shipped config - the caller runs layer_norm + the plain projection, as
    before the fold existed, instead of getting VCX_EINVAL from every transformer block."""
    lim = 0xFFFF0000
    lda = K if lda is None else lda
    M, N = (n_out, rows) if transposed else (rows, n_out)          # GEMM axes: LNFOLD_T puts the weight rows on M
    ldc = N if ldc is None else ldc
    a_rows, a_ld, w_rows, w_ld = (n_out, K, rows, lda) if transposed else (rows, lda, n_out, K)
    return (tune_get("GEMM_DMA") != 0 and K % 64 == 0 and N % 8 == 0 and (n_out % 4 == 0 and n_out >= 4)
            and 2 * ((a_rows - 1) * a_ld + K) < lim and 2 * ((w_rows - 1) * w_ld + K) < lim and 2 * (M + 256) * ldc < lim)


def linear(x, w, bias=None, **kw):
    """x [rows, K] (row stride may exceed K) times w [N, K]^T."""
    rows, K = x.shape
    return gemm(x, w, M=rows, N=w.shape[0], K=K, lda=x.stride(0), bias=bias, **kw)


def conv2d(x, w, bias, *, kh, kw, stride=1, pad_h=None, pad_w=None, ups=0, out_hw=None, **kwargs):
    """x [n, H, W, Cin] channels-last fp16, w [Cout, kh*kw*Cin] as packed by packing.pack_conv.  Returns [n, Ho, Wo, Cout]."""
    n, H, W, cin = x.shape
    if pad_h is None:
        pad_h = kh // 2
    if pad_w is None:
        pad_w = kw // 2
    if out_hw is None:
        He, We = H << ups, W << ups
        out_hw = ((He + 2 * pad_h - kh) // stride + 1, (We + 2 * pad_w - kw) // stride + 1)
    Ho, Wo = out_hw
    cout = w.shape[0]
    geom = dict(in_h=H, in_w=W, out_h=Ho, out_w=Wo, cin=cin, kh=kh, kw=kw, stride=stride, pad_h=pad_h, pad_w=pad_w,
                ups=ups)
    tail_k = sum(t.shape[1] for t in kwargs.get("tail") or ())
    out = gemm(x, w, M=n * Ho * Wo, N=cout, K=kh * kw * cin + tail_k, lda=x.stride(2), bias=bias, conv=geom, **kwargs)
    return out.view(n, Ho, Wo, -1)


def conv_tail_ok(M, cin, cout, taps, tail_ks, in_rows=None):
    """Will vcx_gemm_f16 take a convolution over a cin-channel image with a K tail of widths tail_ks (the 1x1 skip convolution of a
    ResBlock folded into its second convolution)?  The DMA kernel only: a mirror of `dma_ok` in csrc/gemm.hip."""
    lim = 0xFFFF0000
    K = taps * cin + sum(tail_ks)
    return (tune_get("GEMM_DMA") != 0 and cin % 64 == 0 and cout % 8 == 0 and all(k % 64 == 0 and k > 0 for k in tail_ks) and 0 < len(tail_ks) <= 2
            and 2 * (in_rows if in_rows is not None else M) * cin < lim and 2 * (cout - 1) * K + 2 * K < lim and 2 * (M + 256) * cout < lim
            and all(2 * M * k < lim for k in tail_ks))


def temporal_conv3(x, w, bias, **kwargs):
    """x [B, T, P, C]; (3,1,1) convolution along T with zero padding; w [Cout, 3*Cin] as packed by packing.pack_conv."""
    B, T, P, C = x.shape
    geom = dict(in_h=T, in_w=P, out_h=T, out_w=P, cin=C, kh=3, kw=1, stride=1, pad_h=1, pad_w=0, ups=0)
    out = gemm(x, w, M=B * T * P, N=w.shape[0], K=3 * C, lda=x.stride(2), bias=bias, conv=geom, **kwargs)
    return out.view(B, T, P, -1)


# ------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------
def colstats_ok(M, pixels, cin, cout, in_rows=None):
    """Can the layer producing an [M, cout] output - a convolution over a cin-channel image, or a linear layer with K = cin - write
    column moments for a GroupNorm whose statistics span `pixels` consecutive output rows?  (VCX_GEMM_COLSTATS: DMA kernel -
    cin % 64 == 0, cout % 8 == 0, 32-bit byte offsets - whole 64-row strips per statistics unit.)"""
    lim = 0xFFFF0000
    # (a mirror of the kernel-side test in vcx_gemm_f16, like lnfold_ok: with the DMA kernel switched off by the A/B knob the producers
    # fall back to a statistics pass instead of asking for an epilogue that kernel does not have - ADVICE r4)
    return (tune_get("GEMM_DMA") != 0 and pixels % 64 == 0 and M % 64 == 0 and cin % 64 == 0 and cout % 8 == 0 and 2 * (M + 256) * cout < lim
            and 2 * (in_rows if in_rows is not None else M) * cin < lim)


def colstats_buffer(M, cout, device):
    return torch.empty((M // 64, cout, 2), dtype=_f32, device=device)


def group_norm_stats_from_colstats(colstats, n_outer, pixels, C, groups=32):
    """(mean, variance) per (n, group) from the column moments a colstats= convolution wrote: what group_norm(stats=) takes."""
    _dev32(colstats)
    # the moments travel across module boundaries (flow.py, concat targets): a buffer that does not hold exactly the strips of this
    # tensor would be read out of bounds or give silently wrong statistics
    if pixels % 64 != 0 or colstats.numel() != (n_outer * pixels // 64) * C * 2:
        raise VcxError(f"colstats must hold [{n_outer} x {pixels} / 64, {C}, 2] floats with pixels % 64 == 0, got {tuple(colstats.shape)}")
    stats = torch.empty((n_outer, groups, 2), dtype=_f32, device=colstats.device)
    L = lib()
    ws = torch.empty((L.vcx_groupnorm_ws_bytes(n_outer, pixels, groups),), dtype=torch.uint8, device=colstats.device)
    check(L.vcx_groupnorm_stats_from_colstats_f32(colstats.data_ptr(), stats.data_ptr(), ws.data_ptr(), n_outer, pixels, C, groups, _stream()),
          "groupnorm_stats_from_colstats")
    return stats


def group_norm_stats(x, groups=32):
    """(mean, variance) per (n, group) of x [n_outer, pixels, C] fp16: the statistics pass of group_norm on its own."""
    n_outer, pixels, C = x.shape
    _dev16(x)
    L = lib()
    stats = torch.empty((n_outer, groups, 2), dtype=_f32, device=x.device)
    ws = torch.empty((L.vcx_groupnorm_ws_bytes(n_outer, pixels, groups),), dtype=torch.uint8, device=x.device)
    check(L.vcx_groupnorm_stats_f16(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), n_outer, pixels, C, groups, _stream()), "groupnorm_stats")
    return stats


def group_norm_fold_linear(w32, bias, gamma, beta, stats, eps, groups=32):
    """GroupNorm (no SiLU) folded into the linear layer behind it: w32 [N, C] fp32 master weights, stats [n, groups, 2] ->
    (Wn [n, N, C] fp16, bn [n, N] fp32) with Linear(GroupNorm(x_n)) = x_n Wn[n]^T + bn[n] (include/vcx.h)."""
    _dev32(w32, bias, gamma, beta, stats)
    N, C = w32.shape
    n = stats.shape[0]
    if not w32.is_contiguous() or tuple(stats.shape) != (n, groups, 2):
        raise VcxError(f"group_norm_fold_linear: contiguous [N, C] weights and [n, {groups}, 2] statistics expected, got {tuple(w32.shape)} / {tuple(stats.shape)}")
    wn = torch.empty((n, N, C), dtype=_f16, device=w32.device)
    bn = torch.empty((n, N), dtype=_f32, device=w32.device)
    check(lib().vcx_groupnorm_fold_linear_f16(w32.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), wn.data_ptr(),
                                              bn.data_ptr(), n, N, C, groups, eps, _stream()), "groupnorm_fold_linear")
    return wn, bn


def group_norm(x, gamma, beta, eps, silu, groups=32, out=None, stats=None, x2=None):
    """x [n_outer, pixels, C] fp16 (contiguous).  Statistics over (pixels, C/groups) - computed here, or handed in (`stats`
    [n_outer, groups, 2] = (mean, variance), from group_norm_stats_from_colstats).  x2 [n_outer, pixels, C2]: the norm runs over the
    channel concat [x | x2] without materialising it (vcx_groupnorm_apply2_f16; `stats` required)."""
    if x2 is not None:
        return _group_norm2(x, x2, gamma, beta, eps, silu, groups, out, stats)
    n_outer, pixels, C = x.shape
    _dev16(x, out)
    _dev32(gamma, beta)
    L = lib()
    s = _stream()
    if stats is None:
        stats = torch.empty((n_outer, groups, 2), dtype=_f32, device=x.device)
        ws = torch.empty((L.vcx_groupnorm_ws_bytes(n_outer, pixels, groups),), dtype=torch.uint8, device=x.device)
        check(L.vcx_groupnorm_stats_f16(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), n_outer, pixels, C, groups, s), "groupnorm_stats")
    else:
        _dev32(stats)
    if out is None:
        out = torch.empty_like(x)
    check(L.vcx_groupnorm_apply_f16(x.data_ptr(), out.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    n_outer, pixels, C, groups, eps, 1 if silu else 0, s), "groupnorm_apply")
    return out


def _group_norm2(x1, x2, gamma, beta, eps, silu, groups, out, stats):
    n_outer, pixels, c1 = x1.shape
    C = c1 + x2.shape[2]
    _dev16(x1, x2, out)
    _dev32(gamma, beta, stats)
    if stats is None or tuple(x2.shape[:2]) != (n_outer, pixels) or c1 % 8 != 0 or not x1.is_contiguous() or not x2.is_contiguous():
        raise VcxError(f"group_norm over a split concat needs statistics, contiguous halves and c1 % 8 == 0 (x1 {tuple(x1.shape)}, x2 {tuple(x2.shape)})")
    if out is None:
        out = torch.empty((n_outer, pixels, C), dtype=_f16, device=x1.device)
    check(lib().vcx_groupnorm_apply2_f16(x1.data_ptr(), c1, x2.data_ptr(), out.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                         n_outer, pixels, C, groups, eps, 1 if silu else 0, _stream()), "groupnorm_apply2")
    return out


def row_stats(x, eps=1e-5):
    """(mean, rstd) of every row of x [rows, C] fp16 -> [rows, 2] fp32: LayerNorm's statistics for a VCX_GEMM_LNFOLD projection."""
    rows, C = x.shape
    _dev16(x)
    stats = torch.empty((rows, 2), dtype=_f32, device=x.device)
    check(lib().vcx_rowstats_f16(x.data_ptr(), stats.data_ptr(), rows, C, eps, _stream()), "rowstats")
    return stats


def layer_norm(x, gamma, beta, eps=1e-5):
    rows, C = x.shape
    _dev16(x)
    _dev32(gamma, beta)
    out = torch.empty_like(x)
    check(lib().vcx_layernorm_f16(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, C, eps,
                                  _stream()), "layernorm")
    return out


# ------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------
ATTN_ACCUMULATE, ATTN_LOG2_LOGITS, ATTN_CAUSAL = 1, 2, 4
LOG2E = 1.4426950408889634


def flash_attn(q, k, vt, out, *, n_groups, heads, nq, nk, kv_rows, kv_div, ldq, ldk, ldvt, ldo, scale, accumulate=False,
               log2_logits=False):
    """softmax(scale Q K^T) V per (group, head); `log2_logits`: Q K^T already is the base-2 logit (scale * log2 e was folded
    into the projections, e.g. as the GEMM alpha) and `scale` is ignored."""
    flags = (ATTN_ACCUMULATE if accumulate else 0) | (ATTN_LOG2_LOGITS if log2_logits else 0)
    _dev16(q, k, vt, out)
    check(lib().vcx_attn_flash_d64_f16(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), n_groups, heads, nq,
                                       nk, kv_rows, kv_div, ldq, ldk, ldvt, ldo, scale, flags,
                                       _stream()), "attn_flash_d64")
    return out


def flash_attn_d512(q, k, vt, out, *, n_groups, nq, nk, kv_rows, ldq, ldk, ldvt, ldo, scale):
    """softmax(scale Q K^T) V per group with ONE head of dim 512 (VAE AttnBlock); K rows / V^T columns of group g start at g * kv_rows."""
    _dev16(q, k, vt, out)
    check(lib().vcx_attn_flash_d512_f16(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), n_groups, nq, nk, kv_rows, ldq, ldk, ldvt, ldo,
                                        scale, _stream()), "attn_flash_d512")
    return out


def flash_attn_dual(q, k1, vt1, k2, vt2, out, *, n_groups, heads, nq, nk1, kv_rows1, kv_div1, ldk1, ldvt1, nk2, kv_rows2, kv_div2,
                    ldk2, ldvt2, ldq, ldo, scale, log2_logits=False):
    """softmax(scale Q K1^T) V1 + softmax(scale Q K2^T) V2 in one pass over Q and O (text (+) image cross-attention)."""
    _dev16(q, k1, vt1, k2, vt2, out)
    check(lib().vcx_attn_flash_dual_d64_f16(q.data_ptr(), k1.data_ptr(), vt1.data_ptr(), k2.data_ptr(), vt2.data_ptr(),
                                            out.data_ptr(), n_groups, heads, nq, nk1, kv_rows1, kv_div1, ldk1, ldvt1, nk2,
                                            kv_rows2, kv_div2, ldk2, ldvt2, ldq, ldo, scale,
                                            ATTN_LOG2_LOGITS if log2_logits else 0, _stream()), "attn_flash_dual_d64")
    return out


def temporal_attn(qkv, out, *, B, T, P, heads, ld, k_off, v_off, ldo, scale, causal=False):
    """causal: frame t attends to frames <= t (TemporalTransformer(causal_attention=True), reference attention.py:343-345, 377-384)."""
    _dev16(qkv, out)
    if causal:
        check(lib().vcx_attn_temporal_d64_masked_f16(qkv.data_ptr(), out.data_ptr(), B, T, P, heads, ld, k_off, v_off, ldo, scale, ATTN_CAUSAL,
                                                     _stream()), "attn_temporal_d64(causal)")
    else:
        check(lib().vcx_attn_temporal_d64_f16(qkv.data_ptr(), out.data_ptr(), B, T, P, heads, ld, k_off, v_off, ldo, scale,
                                              _stream()), "attn_temporal_d64")
    return out


def temporal_attn_rel(qkv, out, relg, relp, *, R, B, T, P, heads, ld, k_off, v_off, ldo, scale, causal=False):
    """Temporal attention with relative position (reference attention.py:104-108, 120-123): relg [tokens, heads, 64] = q Ek^T is added to the
    logits, relp [tokens, heads, 64] (zeroed by the caller) receives the probabilities by clipped distance; see include/vcx.h."""
    _dev16(qkv, out, relg, relp)
    check(lib().vcx_attn_temporal_d64_rel_f16(qkv.data_ptr(), out.data_ptr(), relg.data_ptr(), relp.data_ptr(), int(R), B, T, P, heads, ld, k_off, v_off,
                                              ldo, scale, ATTN_CAUSAL if causal else 0, _stream()), "attn_temporal_d64(rel)")
    return out


def softmax_rows_(x, n=None):
    rows = x.shape[0]
    check(lib().vcx_softmax_rows_f16(x.data_ptr(), rows, n if n is not None else x.shape[1], x.stride(0), _stream()),
          "softmax_rows")
    return x


# ------------------------------------------------------------------------------------------
# element-wise / layout
# ------------------------------------------------------------------------------------------
def silu_f32(x):
    out = torch.empty_like(x)
    check(lib().vcx_silu_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "silu")
    return out


def gelu_(x):
    """exact-erf GELU in place on a contiguous fp16 tensor (nn.GELU)."""
    check(lib().vcx_gelu_f16(x.data_ptr(), x.data_ptr(), x.numel(), _stream()), "gelu")
    return x


def clip_preprocess(x, size, antialias, mean, std):
    """x fp32 [B, C, H, W] in [-1, 1] -> fp32 [B, C, size, size]: kornia-style anti-aliased bicubic resize, (x + 1) / 2, CLIP
    mean / std (reference condition.py:322-329) in one kernel."""
    x = x.contiguous()
    _dev32(x)
    B, C, H, W = x.shape
    out = torch.empty((B, C, size, size), dtype=_f32, device=x.device)
    m = (ctypes.c_float * C)(*[float(v) for v in mean])
    s = (ctypes.c_float * C)(*[float(v) for v in std])
    check(lib().vcx_clip_preprocess_f32(x.data_ptr(), out.data_ptr(), B, C, H, W, int(size), 1 if antialias else 0, m, s, _stream()),
          "clip_preprocess")
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=_f32, device=t.device)
    check(lib().vcx_timestep_embedding_f32(t.data_ptr(), out.data_ptr(), t.shape[0], dim, max_period, _stream()),
          "timestep_embedding")
    return out


def to_f16(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_f16, device=x.device)
    check(lib().vcx_cast_f32_to_f16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "cast_f32_to_f16")
    return out


def to_f32(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_f32, device=x.device)
    check(lib().vcx_cast_f16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "cast_f16_to_f32")
    return out


def copy2d(src, dst, rows, cols, lds, ldd):
    check(lib().vcx_copy2d_f16(src.data_ptr(), dst.data_ptr(), rows, cols, lds, ldd, _stream()), "copy2d")


def avgpool2x2(x):
    """[n, H, W, C] fp16 channels-last -> [n, H // 2, W // 2, C]: AvgPool2d(2, 2) (reference Downsample(use_conv=False), openaimodel3d.py:70-72)."""
    n, H, W, C = x.shape
    x = x.contiguous()
    out = torch.empty((n, H // 2, W // 2, C), dtype=_f16, device=x.device)
    check(lib().vcx_avgpool2x2_f16(x.data_ptr(), out.data_ptr(), n, H, W, C, _stream()), "avgpool2x2")
    return out


def upsample2x(x):
    """[n, H, W, C] fp16 channels-last -> [n, 2H, 2W, C]: F.interpolate(scale_factor=2, mode='nearest') (reference Upsample, openaimodel3d.py:98-103)."""
    n, H, W, C = x.shape
    x = x.contiguous()
    out = torch.empty((n, 2 * H, 2 * W, C), dtype=_f16, device=x.device)
    check(lib().vcx_upsample2x_f16(x.data_ptr(), out.data_ptr(), n, H, W, C, _stream()), "upsample2x")
    return out


def repeat_rows(x, r):
    """[rows, C] (row stride may exceed C) -> [r * rows, C]: r copies stacked (the batch axis of a channels-last activation)."""
    rows, C = x.shape
    out = torch.empty((r * rows, C), dtype=x.dtype, device=x.device)
    assert x.dtype == _f16
    for i in range(r):
        copy2d(x, out[i * rows:], rows, C, x.stride(0), C)
    return out


def concat_channels(a, b):
    """[rows, Ca] ++ [rows, Cb] -> [rows, Ca+Cb] (torch.cat(dim=1) of the reference's NCHW tensors)."""
    rows, ca = a.shape
    cb = b.shape[1]
    out = torch.empty((rows, ca + cb), dtype=_f16, device=a.device)
    copy2d(a, out, rows, ca, a.stride(0), ca + cb)
    copy2d(b, out[:, ca:], rows, cb, b.stride(0), ca + cb)
    return out


def add_nchw_(h, feat):
    """h [n, H, W, C] fp16 channels-last += feat [n, C, H, W] (any float dtype): adapter features (openaimodel3d.py:582-585)."""
    n, H, W, C = h.shape
    feat = feat.float().contiguous()
    if tuple(feat.shape) != (n, C, H, W):
        raise VcxError(f"adapter feature {tuple(feat.shape)} does not match the activation [{n}, {C}, {H}, {W}]")
    _dev16(h)
    _dev32(feat)
    check(lib().vcx_add_nchw_f32_to_nhwc_f16(feat.data_ptr(), h.data_ptr(), n, C, H * W, _stream()), "add_nchw")
    return h


def ncthw_to_nthwc(src, dst, c_off=0, scale=1.0):
    """src fp32 [B, C, T, H, W] -> dst fp16 [B, T, H, W, ldc][..., c_off:c_off+C]."""
    B, C, T, H, W = src.shape
    src = src.contiguous()
    check(lib().vcx_ncthw_f32_to_nthwc_f16(src.data_ptr(), dst.data_ptr(), B, C, T, H * W, dst.shape[-1], c_off, scale,
                                           _stream()), "ncthw_to_nthwc")
    return dst


def nthwc_to_ncthw(src, C=None):
    """src [B, T, H, W, ldc] (fp16 or fp32) -> fp32 [B, C, T, H, W]."""
    B, T, H, W, ldc = src.shape
    C = ldc if C is None else C
    out = torch.empty((B, C, T, H, W), dtype=_f32, device=src.device)
    check(lib().vcx_nthwc_to_ncthw_f32(src.data_ptr(), out.data_ptr(), B, C, T, H * W, ldc,
                                       1 if src.dtype == _f32 else 0, _stream()), "nthwc_to_ncthw")
    return out


def ddim_step(x, v_cond, v_uncond, noise, coef, ws=None, v_img=None, cfg_img=0.0):
    """One DDIM update on fp32 [B, ...] tensors; coef = 8 host floats (see include/vcx.h); v_img/cfg_img select the
    multi-condition guidance of vcx_ddim_step3_f32."""
    B = x.shape[0]
    n = x.numel() // B
    x_prev = torch.empty_like(x)
    pred_x0 = torch.empty_like(x)
    if ws is None:
        ws = torch.empty((lib().vcx_ddim_ws_bytes(B, n) // 8,), dtype=torch.float64, device=x.device)
    c = (ctypes.c_float * 9)(*([float(v) for v in coef[:8]] + [float(cfg_img)]))
    check(lib().vcx_ddim_step3_f32(x.data_ptr(), v_cond.data_ptr(), _ptr(v_uncond), _ptr(v_img), _ptr(noise),
                                   x_prev.data_ptr(), pred_x0.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(), B, n, c,
                                   _stream()), "ddim_step")
    return x_prev, pred_x0


# ------------------------------------------------------------------------------------------
# experiment knobs (include/vcx.h VCX_TUNE_*): A/B tooling only, the defaults are the product
# ------------------------------------------------------------------------------------------
def tune_set(name, value):
    """Set knob `name` (a key of _lib.TUNE), return its previous value."""
    return lib().vcx_tune_set(TUNE[name][0], int(value))


def tune_get(name):
    return lib().vcx_tune_get(TUNE[name][0])


def tune_report():
    """Knobs that are NOT at their default (bench.py prints this next to its numbers): {} in production."""
    return {k: tune_get(k) for k, (_, d) in TUNE.items() if tune_get(k) != d}


# ------------------------------------------------------------------------------------------
# profiling
# ------------------------------------------------------------------------------------------
def profile_begin(max_records=1 << 16):
    check(lib().vcx_profile_begin(max_records), "profile_begin")


def profile_end():
    buf = (ctypes.c_double * (4 * len(PROF_FAMILIES)))()
    check(lib().vcx_profile_end(buf), "profile_end")
    out = {}
    for i, name in enumerate(PROF_FAMILIES):
        out[name] = dict(launches=int(buf[4 * i]), ms=buf[4 * i + 1], flops=buf[4 * i + 2], bytes=buf[4 * i + 3])
    return out
