"""TEST INFRASTRUCTURE ONLY - a plain-PyTorch (CPU, fp32 arithmetic, fp16 storage) restatement of the libvcx entry points at the level
of `viewcrafter_amd.ops`, written from the contracts in include/vcx.h.

Purpose: the HOST side of the product - which kernel is asked to do what, with which strides, flags, moment buffers and concat
targets (viewcrafter_amd/lvdm/modules/*.py, several hundred launches per UNet forward) - can be executed and checked against the
reference goldens on a GPU-less machine, before GPU minutes are spent on it.  `install(monkeypatch)` swaps these functions in for
the ctypes launchers of `viewcrafter_amd.ops` for the duration of ONE test; the product never imports this file, has no CPU path
and still raises VcxError without a GPU (tests/test_host_logic.py::test_forward_fails_loudly_without_gpu).  What this cannot check
is the kernels themselves - that is what the `-m gpu` suite is for.
"""
import math

import torch
import torch.nn.functional as F

_f16, _f32 = torch.float16, torch.float32


def _view2d(t, rows, cols, ld, extra=0):
    return t.as_strided((rows, cols), (ld, 1), t.storage_offset() + extra)


def gemm(a, w, *, M, N, K, lda, out=None, ldc=None, bias=None, bias_m=False, residual=None, ldr=None, rowadd=None, rowadd_div=0,
         geglu=False, out_f32=False, alpha=1.0, conv=None, ldw=None, ln_stats=None, ln_colsum=None, ln_t=False, colstats=None,
         colstats_ld=None, colstats_col=0, rowstats=None, rowstats_eps=1e-5, tail=None):
    n_out = N // 2 if geglu else N
    tail_k = sum(t.shape[1] for t in tail or ())
    ldw = K if ldw is None else ldw
    W = _view2d(w, N, K, ldw).float()
    K -= tail_k          # the gather's part of K; the tail columns are appended below
    if conv is None:
        X = _view2d(a, M, K, lda).float()
    else:
        g = conv
        cin, kh, kw, stride, ups = g["cin"], g["kh"], g["kw"], g["stride"], g.get("ups", 0)
        n_img = M // (g["out_h"] * g["out_w"])
        img = a.as_strided((n_img, g["in_h"], g["in_w"], cin), (g["in_h"] * g["in_w"] * lda, g["in_w"] * lda, lda, 1), a.storage_offset()).float()
        oy = torch.arange(g["out_h"]).view(-1, 1)
        ox = torch.arange(g["out_w"]).view(1, -1)
        taps = []
        for ky in range(kh):
            for kx in range(kw):
                iy, ix = oy * stride + ky - g["pad_h"], ox * stride + kx - g["pad_w"]
                ok = (iy >= 0) & (iy < (g["in_h"] << ups)) & (ix >= 0) & (ix < (g["in_w"] << ups))
                sy, sx = (iy.clamp(min=0) >> ups).clamp(max=g["in_h"] - 1), (ix.clamp(min=0) >> ups).clamp(max=g["in_w"] - 1)
                v = img[:, sy.expand(g["out_h"], g["out_w"]), sx.expand(g["out_h"], g["out_w"])]           # [n, oh, ow, cin]
                taps.append(v * ok.expand(g["out_h"], g["out_w"]).unsqueeze(-1).float())
        X = torch.stack(taps, dim=3)                                                                       # [n, oh, ow, taps, cin]
        from viewcrafter_amd.packing import conv_slab_major
        if conv.get("slabk", conv_slab_major(cin, kh * kw)):
            X = X.view(n_img, g["out_h"], g["out_w"], kh * kw, cin // 64, 64).permute(0, 1, 2, 4, 3, 5)
        X = X.reshape(M, K)
    if tail:
        X = torch.cat([X] + [t.float() for t in tail], dim=1)
    acc = X @ W.t()
    if ln_stats is not None:
        st = ln_stats.float().view(-1, 2)
        cs = ln_colsum.float()
        if ln_t:      # stats per output column n, colsum per output row m
            acc = alpha * st[:N, 1].view(1, N) * (acc - st[:N, 0].view(1, N) * cs[:M].view(M, 1))
        else:
            acc = alpha * st[:M, 1].view(M, 1) * (acc - st[:M, 0].view(M, 1) * cs[:N].view(1, N))
    else:
        acc = acc * alpha
    if bias is not None:
        acc = acc + (bias.float()[:M].view(M, 1) if bias_m else bias.float()[:N].view(1, N))
    if rowadd is not None:
        rows = torch.arange(M) // rowadd_div
        acc = acc + rowadd.float()[rows]           # [rows, N]; may be a column slice of a wider matrix
    if geglu:
        a3 = acc.view(M, N // 64, 2, 32)
        acc = (a3[:, :, 0] * F.gelu(a3[:, :, 1])).reshape(M, n_out)
    if residual is not None:
        acc = acc + _view2d(residual, M, n_out, ldr if ldr is not None else residual.stride(0)).float()
    if out is None:
        out = torch.empty((M, n_out), dtype=_f32 if out_f32 else _f16)
        ldc = n_out
    elif ldc is None:
        ldc = out.stride(0)
    stored = acc.to(out.dtype)
    _view2d(out, M, n_out, ldc).copy_(stored)
    if colstats is not None:
        assert M % 64 == 0 and not out_f32 and not geglu
        cld = n_out if colstats_ld is None else colstats_ld
        v = stored.float().view(M // 64, 64, n_out)
        mean = v.mean(1)
        m2 = ((v - mean.unsqueeze(1)) ** 2).sum(1)
        colstats.view(M // 64, cld, 2)[:, colstats_col:colstats_col + n_out] = torch.stack([mean, m2], dim=-1)
    if rowstats is not None:       # VCX_GEMM_ROWSTATS: LayerNorm's statistics of the ROUNDED output rows
        assert not out_f32 and not geglu and conv is None
        rowstats.view(M, 2).copy_(row_stats(stored, rowstats_eps))
    return out


def gemm_units(a, wn, bn, *, unit_rows, out=None, rowstats=None, rowstats_eps=1e-5):
    M, K = a.shape
    units, N, _ = wn.shape
    if out is None:
        out = torch.empty((M, N), dtype=_f16)
    for u in range(units):
        gemm(a[u * unit_rows:], wn[u], M=unit_rows, N=N, K=K, lda=a.stride(0), out=out[u * unit_rows:], ldc=out.stride(0), bias=bn[u],
             rowstats=None if rowstats is None else rowstats[u * unit_rows:(u + 1) * unit_rows], rowstats_eps=rowstats_eps)
    return out


def group_norm_stats_from_colstats(colstats, n_outer, pixels, C, groups=32):
    strips = pixels // 64
    cs = colstats.view(n_outer, strips, -1, 2)[:, :, :C].double()
    mean_c = cs[..., 0].mean(1)                                                     # [n, C] (equal counts)
    m2_c = cs[..., 1].sum(1) + 64.0 * ((cs[..., 0] - mean_c.unsqueeze(1)) ** 2).sum(1)
    cpg = C // groups
    mg = mean_c.view(n_outer, groups, cpg)
    mean_g = mg.mean(2)
    m2_g = m2_c.view(n_outer, groups, cpg).sum(2) + pixels * ((mg - mean_g.unsqueeze(2)) ** 2).sum(2)
    return torch.stack([mean_g, m2_g / (pixels * cpg)], dim=-1).float()


def group_norm_stats(x, groups=32):
    n, pixels, C = x.shape
    xf = x.float().view(n, pixels, groups, C // groups)
    return torch.stack([xf.mean(dim=(1, 3)), xf.var(dim=(1, 3), unbiased=False)], dim=-1)


def group_norm_fold_linear(w32, bias, gamma, beta, stats, eps, groups=32):
    """include/vcx.h vcx_groupnorm_fold_linear_f16: the mean term on the fp16-ROUNDED scaled weight."""
    N, C = w32.shape
    cpg = C // groups
    mean = stats[..., 0].float().repeat_interleave(cpg, dim=1)                          # [n, C]
    a = torch.rsqrt(stats[..., 1].float() + eps).repeat_interleave(cpg, dim=1) * gamma.float()
    wn = (w32.float()[None] * a[:, None, :]).to(_f16)                                   # [n, N, C]
    bn = (w32.float() @ beta.float())[None] - (wn.float() * mean[:, None, :]).sum(-1)
    if bias is not None:
        bn = bn + bias.float()[None]
    return wn, bn.float().contiguous()


def group_norm(x, gamma, beta, eps, silu, groups=32, out=None, stats=None, x2=None):
    if x2 is not None:
        x = torch.cat([x, x2], dim=2)
    n, pixels, C = x.shape
    xf = x.float().view(n, pixels, groups, C // groups)
    if stats is None:
        mean, var = xf.mean(dim=(1, 3)), xf.var(dim=(1, 3), unbiased=False)
    else:
        mean, var = stats[..., 0], stats[..., 1]
    y = ((xf - mean.view(n, 1, groups, 1)) * torch.rsqrt(var.view(n, 1, groups, 1) + eps)).view(n, pixels, C) * gamma.float() + beta.float()
    if silu:
        y = F.silu(y)
    if out is None:
        out = torch.empty_like(x)
    out.copy_(y.to(_f16))
    return out


def conv_tail_ok(M, cin, cout, taps, tail_ks, in_rows=None):
    return cin % 64 == 0 and cout % 8 == 0 and all(k % 64 == 0 and k > 0 for k in tail_ks) and 0 < len(tail_ks) <= 2


def row_stats(x, eps=1e-5):
    xf = x.float()
    mean = xf.mean(1)
    return torch.stack([mean, torch.rsqrt(xf.var(1, unbiased=False) + eps)], dim=1)


def layer_norm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps).to(_f16)


def _attend(qm, km, vtm, scale, log2):
    s = qm.float() @ km.float().t()
    p = torch.softmax(s * (math.log(2.0) if log2 else scale), dim=-1)
    return p.to(_f16).float() @ vtm.float().t()


def _heads(q, k, vt, g, h, nq, nk, kv_rows, kv_div, ldq, ldk, ldvt):
    qm = q.as_strided((nq, 64), (ldq, 1), q.storage_offset() + g * nq * ldq + h * 64)
    km = k.as_strided((nk, 64), (ldk, 1), k.storage_offset() + (g // kv_div) * kv_rows * ldk + h * 64)
    vm = vt.as_strided((64, nk), (ldvt, 1), vt.storage_offset() + h * 64 * ldvt + (g // kv_div) * kv_rows)
    return qm, km, vm


def flash_attn(q, k, vt, out, *, n_groups, heads, nq, nk, kv_rows, kv_div, ldq, ldk, ldvt, ldo, scale, accumulate=False, log2_logits=False):
    for g in range(n_groups):
        for h in range(heads):
            qm, km, vm = _heads(q, k, vt, g, h, nq, nk, kv_rows, kv_div, ldq, ldk, ldvt)
            o = out.as_strided((nq, 64), (ldo, 1), out.storage_offset() + g * nq * ldo + h * 64)
            r = _attend(qm, km, vm, scale, log2_logits)
            o.copy_(((o.float() + r) if accumulate else r).to(_f16))
    return out


def flash_attn_d512(q, k, vt, out, *, n_groups, nq, nk, kv_rows, ldq, ldk, ldvt, ldo, scale):
    for g in range(n_groups):
        qm = _view2d(q, nq, 512, ldq, g * nq * ldq).float()
        km = _view2d(k, nk, 512, ldk, g * kv_rows * ldk).float()
        vtm = _view2d(vt, 512, nk, ldvt, g * kv_rows).float()
        _view2d(out, nq, 512, ldo, g * nq * ldo).copy_(((qm @ km.t() * scale).softmax(-1) @ vtm.t()).to(_f16))
    return out


def flash_attn_dual(q, k1, vt1, k2, vt2, out, *, n_groups, heads, nq, nk1, kv_rows1, kv_div1, ldk1, ldvt1, nk2, kv_rows2, kv_div2, ldk2, ldvt2,
                    ldq, ldo, scale, log2_logits=False):
    for g in range(n_groups):
        for h in range(heads):
            qm, ka, va = _heads(q, k1, vt1, g, h, nq, nk1, kv_rows1, kv_div1, ldq, ldk1, ldvt1)
            _, kb, vb = _heads(q, k2, vt2, g, h, nq, nk2, kv_rows2, kv_div2, ldq, ldk2, ldvt2)
            o = out.as_strided((nq, 64), (ldo, 1), out.storage_offset() + g * nq * ldo + h * 64)
            o.copy_((_attend(qm, ka, va, scale, log2_logits) + _attend(qm, kb, vb, scale, log2_logits)).to(_f16))
    return out


def temporal_attn(qkv, out, *, B, T, P, heads, ld, k_off, v_off, ldo, scale, causal=False):
    x = qkv.as_strided((B, T, P, ld), (T * P * ld, P * ld, ld, 1), qkv.storage_offset()).float()
    o = out.as_strided((B, T, P, ldo), (T * P * ldo, P * ldo, ldo, 1), out.storage_offset())
    for h in range(heads):
        qh = x[..., h * 64:h * 64 + 64].permute(0, 2, 1, 3)                           # [B, P, T, 64]
        kh = x[..., k_off + h * 64:k_off + h * 64 + 64].permute(0, 2, 1, 3)
        vh = x[..., v_off + h * 64:v_off + h * 64 + 64].permute(0, 2, 1, 3)
        logits = qh @ kh.transpose(-1, -2) * scale
        if causal:
            logits = logits.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf"))
        p = torch.softmax(logits, dim=-1).to(_f16).float()
        o[..., h * 64:h * 64 + 64] = (p @ vh).permute(0, 2, 1, 3).to(_f16)
    return out


def temporal_attn_rel(qkv, out, relg, relp, *, R, B, T, P, heads, ld, k_off, v_off, ldo, scale, causal=False):
    x = qkv.as_strided((B, T, P, ld), (T * P * ld, P * ld, ld, 1), qkv.storage_offset()).float()
    o = out.as_strided((B, T, P, ldo), (T * P * ldo, P * ldo, ldo, 1), out.storage_offset())
    g = relg.view(B, T, P, heads, 64).float()
    pr = relp.view(B, T, P, heads, 64)
    dist = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).clamp(-R, R) + R                    # [t, s]
    for h in range(heads):
        qh = x[..., h * 64:h * 64 + 64].permute(0, 2, 1, 3)                           # [B, P, T, 64]
        kh = x[..., k_off + h * 64:k_off + h * 64 + 64].permute(0, 2, 1, 3)
        vh = x[..., v_off + h * 64:v_off + h * 64 + 64].permute(0, 2, 1, 3)
        gh = g[:, :, :, h].permute(0, 2, 1, 3)                                          # [B, P, T(query), 64(slot)]
        logits = (qh @ kh.transpose(-1, -2) + torch.gather(gh, -1, dist.expand(B, P, T, T))) * scale
        if causal:
            logits = logits.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf"))
        p = torch.softmax(logits, dim=-1).to(_f16).float()
        o[..., h * 64:h * 64 + 64] = (p @ vh).permute(0, 2, 1, 3).to(_f16)
        slots = torch.zeros(B, P, T, 64).scatter_add_(-1, dist.expand(B, P, T, T), p)
        pr[:, :, :, h] = slots.permute(0, 2, 1, 3).to(_f16)
    return out


def softmax_rows_(x, n=None):
    n = x.shape[1] if n is None else n
    n8 = (n + 7) // 8 * 8
    x[:, :n] = torch.softmax(x[:, :n].float(), -1).to(_f16)
    x[:, n:n8] = 0
    return x


def copy2d(src, dst, rows, cols, lds, ldd):
    _view2d(dst, rows, cols, ldd).copy_(_view2d(src, rows, cols, lds))


def avgpool2x2(x):
    return F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).to(_f16).contiguous()


def upsample2x(x):
    return F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).contiguous()


def add_nchw_(h, feat):
    h.copy_((h.float() + feat.float().permute(0, 2, 3, 1)).to(_f16))
    return h


def ncthw_to_nthwc(src, dst, c_off=0, scale=1.0):
    C = src.shape[1]
    dst[..., c_off:c_off + C] = (src.float() * scale).permute(0, 2, 3, 4, 1).to(_f16)
    return dst


def nthwc_to_ncthw(src, C=None):
    C = src.shape[-1] if C is None else C
    return src[..., :C].float().permute(0, 4, 1, 2, 3).contiguous()


def timestep_embedding(t, dim, max_period=10000.0):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=_f32) / half)
    args = t.float()[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1) if dim % 2 else emb


_TUNE = {}


def install(monkeypatch):
    """Swap the launchers of viewcrafter_amd.ops for the functions above (one test's lifetime)."""
    from viewcrafter_amd import _lib, ops
    _TUNE.clear()
    table = dict(require_gpu=lambda: None, gemm=gemm, group_norm_stats_from_colstats=group_norm_stats_from_colstats, group_norm=group_norm,
                 group_norm_stats=group_norm_stats, group_norm_fold_linear=group_norm_fold_linear, gemm_units=gemm_units,
                 row_stats=row_stats, layer_norm=layer_norm, conv_tail_ok=conv_tail_ok, flash_attn=flash_attn, flash_attn_d512=flash_attn_d512, flash_attn_dual=flash_attn_dual, temporal_attn=temporal_attn, temporal_attn_rel=temporal_attn_rel,
                 softmax_rows_=softmax_rows_, copy2d=copy2d, avgpool2x2=avgpool2x2, upsample2x=upsample2x, add_nchw_=add_nchw_, ncthw_to_nthwc=ncthw_to_nthwc, nthwc_to_ncthw=nthwc_to_ncthw,
                 timestep_embedding=timestep_embedding, silu_f32=lambda x: F.silu(x.float()), gelu_=lambda x: x.copy_(F.gelu(x.float()).to(_f16)),
                 to_f16=lambda x: x.to(_f16).contiguous(), to_f32=lambda x: x.float().contiguous(),
                 tune_get=lambda name: _TUNE.get(name, _lib.TUNE[name][1]),
                 tune_set=lambda name, v: _TUNE.update({name: int(v)}))
    for name, fn in table.items():
        monkeypatch.setattr(ops, name, fn)
