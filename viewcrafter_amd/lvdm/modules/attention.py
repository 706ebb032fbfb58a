"""Transformer blocks of the lvdm UNet on the gfx950 kernels.

Mirrors the classes of the reference's lvdm/modules/attention.py that the ViewCrafter graph instantiates
(CrossAttention, FeedForward/GEGLU, BasicTransformerBlock, SpatialTransformer, TemporalTransformer) with identical
constructor arguments and parameter names/shapes, so reference checkpoints load with strict=True.  The parameters are
held in stock nn.Linear / nn.LayerNorm / nn.GroupNorm containers; `forward` never calls them — it runs the packed
fp16 copies through libvcx (no PyTorch fallback).

Activations are channels-last fp16.  A spatial transformer sees tokens [(b t h w), C]; a temporal transformer sees the
SAME row order (Linear/LayerNorm are row-wise, so no '(b t) c h w <-> (b h w) t c' copies are needed — only the
temporal attention kernel itself walks frames with a stride of h*w rows).
"""
import math
import os

import torch
from torch import nn

from ... import ops
from ...packing import fold_layernorm, pack_geglu
from ..common import default
from .flow import GN_STATS_LEVEL, out_kwargs


def _f16(t):
    return t.detach().to(torch.float16).contiguous()


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


# nn.LayerNorm -> nn.Linear pairs of BasicTransformerBlock run as row statistics + ONE projection of the un-normalised token rows
# (VCX_GEMM_LNFOLD, include/vcx.h): the normalised copy of the token stream is neither written nor re-read.  Needs the DMA GEMM
# kernel (channel count % 64 == 0); VCX_LN_FOLD=0 keeps the separate LayerNorm kernel (A/B runs, tools/lnfold_ab.py).
FOLD_LAYERNORM = os.environ.get("VCX_LN_FOLD", "1") != "0"
# ... in front of the GEGLU projection too, from C = 640 up (round 6): there the projection runs on the tiled engine either way and the fold
# replaces a read + write pass over the token stream by a statistics pass (same box, tools/step_ab.py lnff = 0 / 2 / 1, profiles/r06s_lnff_ab.txt:
# LayerNorm family 3.14 -> 2.67 ms at no GEMM cost, step -0.45 ms).  NOT at C = 320: the level-0 projection would leave the weight-stationary
# GEGLU kernel for the tiled engine's folded epilogue, whose two extra multiply-adds per output pair cost more than the pass they save (fold
# everywhere: GEMM +1.0 ms, step level; round 3: profiles/r03_experiments.md section 8).  VCX_LN_FOLD_FF=0: never; VCX_LN_FOLD_FF_MIN_DIM: the width.
FOLD_LAYERNORM_FF = os.environ.get("VCX_LN_FOLD_FF", "1") != "0"
FOLD_LAYERNORM_FF_MIN_DIM = int(os.environ.get("VCX_LN_FOLD_FF_MIN_DIM", "640"))


# TemporalTransformer.norm -> proj_in (reference attention.py:331-336,369-372) and SpatialTransformer.norm -> proj_in (:265-269,299):
# a GroupNorm without SiLU in front of a linear layer is an affine map per (statistics unit, channel), so it runs as per-unit scaled
# weights and bias of the projection (vcx_groupnorm_fold_linear_f16 + vcx_gemm_units_f16 on the UN-normalised rows) and the normalised
# copy of the tensor is neither written nor re-read.  Temporal norms (unit = video): wherever a video's tensor is at least
# GN_FOLD_MIN_BYTES (below that the apply pass is cheaper than a launch per video).  Spatial norms (unit = frame, 50 of them): only
# where all frames go through ONE launch - the weight-stationary kernel at C = 320, level 0 of the UNet, whose blocks keep a weight
# set in registers anyway.  VCX_GN_FOLD=0: the separate apply pass everywhere (A/B runs, tools/step_ab.py).
GN_FOLD = os.environ.get("VCX_GN_FOLD", "1") != "0"
GN_FOLD_MIN_BYTES = int(os.environ.get("VCX_GN_FOLD_MIN_BYTES", str(16 << 20)))
GN_FOLD_SPATIAL = os.environ.get("VCX_GN_FOLD_SPATIAL", "1") != "0"


def spatial_fold_ok(n, pixels, C, D):
    """Does vcx_gemm_units_f16 take n frames of `pixels` rows in ONE weight-stationary launch (csrc/gemm.hip)?"""
    return (GN_FOLD and GN_FOLD_SPATIAL and n > 1 and C == 320 and D == 320 and pixels % 32 == 0 and pixels >= 1024 and n * pixels >= 8192
            and 2 * n * pixels * C >= GN_FOLD_MIN_BYTES and 2 * (n * pixels + 256) * C < 0xFFFF0000 and ops.tune_get("GEMM_DMA") != 0
            and ops.tune_get("GEMM_WS") != 0 and ops.tune_get("GEMM_CFG") < 0)      # (a forced tile configuration means the tiled engine: ADVICE r5)


def _folded(pj, rows, K, lda, transposed=False):
    """Will ln_linear / ln_linear_t run this pack as a folded projection (and so want row statistics)?"""
    return pj["colsum"] is not None and ops.lnfold_ok(rows, pj["w"].shape[0], K, lda=lda, transposed=transposed)


def _ln_projection(w, ln, alpha=1.0, bias=None):
    """Packed form of `Linear(w, bias)(LayerNorm(.)) * alpha` (alpha scales the product, not the layer's own bias): folded
    (w', colsum, bias') when `ln` is given, plain fp16 weights otherwise (colsum None: the caller normalises first).  A folded
    pack keeps what it needs to build the plain form on demand (`_plain`: shapes the folded kernel cannot take, ops.lnfold_ok)."""
    if ln is not None:
        wf, colsum, bias_f = fold_layernorm(w, ln.weight, ln.bias, None)
        bias_f = bias_f * alpha
        if bias is not None:
            bias_f = bias_f + bias.detach().float()
        return dict(w=wf, colsum=colsum, bias=bias_f.contiguous(), eps=ln.eps, _src=(w, bias), _plain=None)
    return dict(w=_f16(w), colsum=None, bias=None if bias is None else _f32(bias), eps=None)


def _plain_of(pj):
    """The un-folded fp16 weights (+ fp32 bias) of a folded pack, built once when a call cannot use the fold."""
    if pj["_plain"] is None:
        w, bias = pj["_src"]
        with torch.no_grad():
            pj["_plain"] = (_f16(w), None if bias is None else _f32(bias))
    return pj["_plain"]


def ln_linear(t, pj, lnp, alpha=1.0, stats=None, **kw):
    """Linear(LayerNorm(t)) * alpha for a pack of _ln_projection: the folded projection of the un-normalised rows where the pack
    is folded AND the kernel takes the shape, layer_norm + plain projection otherwise.  lnp = (gamma, beta, eps); `stats` =
    row statistics of t when the caller already has them."""
    rows, K = t.shape
    if pj["colsum"] is not None and ops.lnfold_ok(rows, pj["w"].shape[0], K, lda=t.stride(0)):
        st = stats if stats is not None else ops.row_stats(t, lnp[2])
        return ops.linear(t, pj["w"], pj["bias"], alpha=alpha, ln_stats=st, ln_colsum=pj["colsum"], **kw)
    w, bias = (pj["w"], pj["bias"]) if pj["colsum"] is None else _plain_of(pj)
    return ops.linear(ops.layer_norm(t, *lnp), w, bias, alpha=alpha, **kw)


def ln_linear_t(t, pj, lnp, stats=None):
    """The transposed projection out[D, tokens] = W LayerNorm(t)^T (V^T of the spatial self-attention), same rule as ln_linear."""
    tokens, K = t.shape
    D = pj["w"].shape[0]
    if pj["colsum"] is not None and ops.lnfold_ok(tokens, D, K, lda=t.stride(0), transposed=True):
        st = stats if stats is not None else ops.row_stats(t, lnp[2])
        return ops.gemm(pj["w"], t, M=D, N=tokens, K=K, lda=K, ldw=t.stride(0), bias=pj["bias"], bias_m=True, ln_stats=st,
                        ln_colsum=pj["colsum"], ln_t=True)
    w, bias = (pj["w"], pj["bias"]) if pj["colsum"] is None else _plain_of(pj)
    return ops.gemm(w, ops.layer_norm(t, *lnp), M=D, N=tokens, K=K, lda=K, bias=bias, bias_m=bias is not None)


def _sig(tensors):
    return tuple((p.data_ptr(), p._version) for p in tensors)


class PackedModule(nn.Module):
    """Base for modules that keep kernel-layout copies of their parameters in `self._pk` (built lazily on the first
    forward, dropped whenever parameters are moved/cast or a state dict is loaded).  A pack that embeds parameters of ANOTHER
    module (the folded LayerNorm of the owning BasicTransformerBlock, `_pre_norm`) also remembers their identity and version
    counter and is rebuilt when either changes - `blk.norm1.load_state_dict(...)`, `.data` assignment or in-place edits of the
    norm no longer leave stale folded weights behind."""

    def __init__(self):
        super().__init__()
        self._pk = None
        self._pk_sig = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_packed())

    def _drop_packed(self):
        self._pk = None

    def _apply(self, fn, recurse=True):
        self._pk = None
        return super()._apply(fn, recurse)

    def _foreign_params(self):
        pre = getattr(self, "_pre_norm", None)
        return [pre[0].weight, pre[0].bias] if pre else []

    def packed(self):
        foreign = self._foreign_params()
        if self._pk is not None and foreign and _sig(foreign) != self._pk_sig:
            self._pk = None
        if self._pk is None:
            with torch.no_grad():
                self._pk = self._pack()
            self._pk_sig = _sig(foreign)
        return self._pk

    def _pack(self):
        raise NotImplementedError


class RelativePosition(nn.Module):
    """Reference attention.py:20-40: a table of 2 R + 1 embeddings indexed by clamp(s - t, -R, R) + R.  Parameter container: the two contractions
    with it are 64-wide GEMMs around the temporal attention kernel (TemporalTransformer.forward)."""

    def __init__(self, num_units, max_relative_position):
        super().__init__()
        self.num_units, self.max_relative_position = num_units, max_relative_position
        self.embeddings_table = nn.Parameter(torch.empty(max_relative_position * 2 + 1, num_units))
        nn.init.xavier_uniform_(self.embeddings_table)


class CrossAttention(PackedModule):
    """Reference attention.py:42-209.  Parameter container + packing; the attention itself is launched by the owning
    transformer (it needs the token geometry).  relative_position (`use_relative_position`; false in the ViewCrafter YAMLs) and the causal
    mask exist in the temporal transformer only: two table GEMMs around / a flag of the temporal attention kernel."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0., relative_position=False,
                 temporal_length=None, video_length=None, image_cross_attention=False, image_cross_attention_scale=1.0,
                 image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        if relative_position and (temporal_length is None or 2 * temporal_length + 1 > 64 or context_dim is not None):
            raise NotImplementedError("relative_position: temporal self-attention with temporal_length <= 31 (2R + 1 table rows in 64 slots)")
        if image_cross_attention_scale_learnable:
            raise NotImplementedError("image_cross_attention_scale_learnable is not used by the ViewCrafter configs")
        if dim_head != 64:
            raise NotImplementedError(f"libvcx attention kernels are built for head dim 64 (got {dim_head})")
        if image_cross_attention and image_cross_attention_scale != 1.0:
            raise NotImplementedError("image_cross_attention_scale != 1.0 (the dual-softmax kernel sums the two outputs unweighted; "
                                      "the ViewCrafter configs keep the constant 1.0, attention.py:62)")
        inner_dim = dim_head * heads
        self.is_self = context_dim is None
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head, self.inner_dim = heads, dim_head, inner_dim
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self.relative_position = bool(relative_position)
        if self.relative_position:      # (reference attention.py:59-62)
            self.relative_position_k = RelativePosition(num_units=dim_head, max_relative_position=temporal_length)
            self.relative_position_v = RelativePosition(num_units=dim_head, max_relative_position=temporal_length)
        self.video_length = video_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale = image_cross_attention_scale
        self.text_context_len = text_context_len
        if image_cross_attention:
            self.to_k_ip = nn.Linear(context_dim, inner_dim, bias=False)
            self.to_v_ip = nn.Linear(context_dim, inner_dim, bias=False)
        self.query_dim = query_dim
        self._pre_norm = None      # [LayerNorm] in front of the query-side projections (set by BasicTransformerBlock; a list so
        self.kind = None           # that it is not registered as a sub-module);  kind: "spatial" | "temporal" (owner's layout)

    def _pack(self):
        """wo / bo always; the query-side projections in the form their owner launches them - spatial self-attention: Q|K as one
        [tokens, 2D] projection carrying sqrt(scale log2 e) each + V^T; temporal: Q|K|V as one; cross-attention: Q carrying
        scale log2 e, and K / V (+ image K / V) weights for the context side - with the preceding LayerNorm folded in where possible."""
        pk = dict(wo=_f16(self.to_out[0].weight), bo=_f32(self.to_out[0].bias))
        ln = self._pre_norm[0] if (self._pre_norm and FOLD_LAYERNORM and self.query_dim % 64 == 0) else None
        wq, wk, wv = self.to_q.weight.detach(), self.to_k.weight.detach(), self.to_v.weight.detach()
        if not self.is_self:
            pk["q"] = _ln_projection(wq, ln, alpha=self.scale * ops.LOG2E)
            pk["wk"], pk["wv"] = _f16(wk), _f16(wv)
        elif self.kind == "temporal":
            pk["qkv"] = _ln_projection(torch.cat([wq, wk, wv], 0), ln)
        else:
            pk["qk"] = _ln_projection(torch.cat([wq, wk], 0), ln, alpha=math.sqrt(self.scale * ops.LOG2E))
            pk["v"] = _ln_projection(wv, ln)
        if self.image_cross_attention:
            pk["wk_ip"], pk["wv_ip"] = _f16(self.to_k_ip.weight), _f16(self.to_v_ip.weight)
        if self.relative_position:
            # Ek as the weight of  relg = q Ek^T  ([64 slots, 64 dims], rows 2R + 1 .. 63 zero) and Ev^T as the weight of  out += relp Ev  ([64 dims, 64 slots])
            ek, ev = self.relative_position_k.embeddings_table.detach(), self.relative_position_v.embeddings_table.detach()
            wk_rel = torch.zeros((64, 64), dtype=torch.float32, device=ek.device)
            wv_rel = torch.zeros((64, 64), dtype=torch.float32, device=ek.device)
            wk_rel[:ek.shape[0]] = ek.float()
            wv_rel[:, :ev.shape[0]] = ev.float().t()
            pk["rel_k"], pk["rel_v"], pk["rel_R"] = _f16(wk_rel), _f16(wv_rel), self.relative_position_k.max_relative_position
        return pk

    def forward(self, *args, **kwargs):
        raise RuntimeError("CrossAttention is driven by SpatialTransformer/TemporalTransformer in this package")


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(PackedModule):
    """Reference attention.py:425-442 with glu=True: Linear(C, 8C) -> x * gelu(gate) -> Linear(4C, C).  The GEGLU gate
    is applied inside the first GEMM's epilogue (weights row-interleaved by pack_geglu)."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward is on the ViewCrafter path")
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))
        self.dim = dim
        self._pre_norm = None      # [LayerNorm] in front of the GEGLU projection (set by BasicTransformerBlock)

    def _pack(self):
        proj = self.net[0].proj
        ln = self._pre_norm[0] if (self._pre_norm and FOLD_LAYERNORM and FOLD_LAYERNORM_FF and self.dim % 64 == 0 and self.dim >= FOLD_LAYERNORM_FF_MIN_DIM) else None
        p1 = _ln_projection(proj.weight.detach(), ln, bias=proj.bias)
        w1, b1 = pack_geglu(p1["w"], p1["bias"])
        colsum = None if p1["colsum"] is None else pack_geglu(p1["w"], p1["colsum"])[1]
        return dict(w1=w1, b1=b1, colsum=colsum, w2=_f16(self.net[2].weight), b2=_f32(self.net[2].bias))

    def run(self, t, ln_params):
        """t + FF(LayerNorm(t)); ln_params = (gamma, beta, eps) of the LayerNorm in front (used when it is not folded into w1)."""
        pk = self.packed()
        if pk["colsum"] is not None and ops.lnfold_ok(t.shape[0], pk["w1"].shape[0], t.shape[1], lda=t.stride(0)):
            g = ops.linear(t, pk["w1"], pk["b1"], geglu=True, ln_stats=ops.row_stats(t, ln_params[2]), ln_colsum=pk["colsum"])
        else:
            if pk["colsum"] is not None and "w1_plain" not in pk:      # folded pack, shape the folded kernel cannot take
                proj = self.net[0].proj
                pk["w1_plain"], pk["b1_plain"] = pack_geglu(_f16(proj.weight), _f32(proj.bias))
            w1, b1 = (pk["w1"], pk["b1"]) if pk["colsum"] is None else (pk["w1_plain"], pk["b1_plain"])
            g = ops.linear(ops.layer_norm(t, *ln_params), w1, b1, geglu=True)
        return ops.linear(g, pk["w2"], pk["b2"], residual=t)


class BasicTransformerBlock(PackedModule):
    """Reference attention.py:212-246 (attn1 self, attn2 cross or - in temporal blocks - self again, GEGLU FF)."""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attention_cls=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is not used by the ViewCrafter configs")
        attn_cls = CrossAttention if attention_cls is None else attention_cls
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              video_length=video_length, image_cross_attention=image_cross_attention,
                              image_cross_attention_scale=image_cross_attention_scale,
                              image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                              text_context_len=text_context_len)
        self.image_cross_attention = image_cross_attention
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint
        self.attn1._pre_norm, self.attn2._pre_norm, self.ff._pre_norm = [self.norm1], [self.norm2], [self.norm3]

    def set_kind(self, kind):
        self.attn1.kind = self.attn2.kind = kind

    def _drop_packed(self):
        super()._drop_packed()
        for m in (self.attn1, self.attn2, self.ff):      # their packs hold this block's LayerNorm parameters
            m._drop_packed()

    def _pack(self):
        return [(_f32(n.weight), _f32(n.bias), n.eps) for n in (self.norm1, self.norm2, self.norm3)]

    def ln_params(self):
        return self.packed()


class ContextKV:
    """Projected cross-attention keys/values of one SpatialTransformer block for one conditioning tensor.  They depend
    only on the context, i.e. they are constant over all DDIM steps and are cached by UNetModel."""
    __slots__ = ("k_txt", "vt_txt", "k_img", "vt_img", "n_txt_rows", "n_txt", "n_img", "img_per_frame")


def project_context(attn2, ctx):
    """ctx: dict(txt=[B*80, D] fp16 (rows 77..79 zero), img=[G*Li, D] fp16 or None, n_img=Li, per_frame=bool)."""
    pk = attn2.packed()
    kv = ContextKV()
    txt = ctx["txt"]
    D = txt.shape[1]
    C = attn2.inner_dim
    kv.k_txt = ops.linear(txt, pk["wk"])                                        # [B*80, C]
    kv.vt_txt = ops.gemm(pk["wv"], txt, M=C, N=txt.shape[0], K=D, lda=D)        # [C, B*80]  (V^T for the flash kernel)
    kv.n_txt_rows = txt.shape[0]
    kv.n_txt = int(ctx.get("n_txt", attn2.text_context_len))      # real text tokens (a context shorter than 77 is zero-padded to 80 rows)
    kv.k_img = kv.vt_img = None
    kv.n_img, kv.img_per_frame = ctx.get("n_img", 0), ctx.get("per_frame", False)
    if attn2.image_cross_attention and ctx.get("img") is not None:
        img = ctx["img"]
        kv.k_img = ops.linear(img, pk["wk_ip"])
        kv.vt_img = ops.gemm(pk["wv_ip"], img, M=C, N=img.shape[0], K=D, lda=D)
    return kv


class SpatialTransformer(PackedModule):
    """Reference attention.py:249-310 (use_linear=True, or False = 1x1 Conv2d projections): GN(eps 1e-6) -> proj_in -> [LN, self-attn, LN, text (+) image
    cross-attn, LN, GEGLU-FF] -> proj_out -> + x."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, use_checkpoint=True,
                 disable_self_attn=False, use_linear=False, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False):
        super().__init__()
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.n_heads = n_heads
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        # use_linear=False (reference attention.py:266-267, 287-288; not used by the ViewCrafter YAMLs): 1x1 Conv2d projections - the same
        # matmul on a channels-last row; the parameters keep the reference's [out, in, 1, 1] shape so a checkpoint loads strictly
        self.proj_in = nn.Linear(in_channels, inner_dim) if use_linear else nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                  disable_self_attn=disable_self_attn, checkpoint=use_checkpoint, attention_cls=None,
                                  video_length=video_length, image_cross_attention=image_cross_attention,
                                  image_cross_attention_scale_learnable=image_cross_attention_scale_learnable)
            for _ in range(depth)])
        for blk in self.transformer_blocks:
            blk.set_kind("spatial")
        self.proj_out = nn.Linear(inner_dim, in_channels) if use_linear else nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0)
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)
        self.use_linear = use_linear

    def _pack(self):
        win = self.proj_in.weight.detach().reshape(self.proj_in.weight.shape[0], -1)       # ([out, in, 1, 1] of the 1x1-conv form)
        wout = self.proj_out.weight.detach().reshape(self.proj_out.weight.shape[0], -1)
        return dict(gn_w=_f32(self.norm.weight), gn_b=_f32(self.norm.bias), win=_f16(win), win32=_f32(win),
                    bin=_f32(self.proj_in.bias), wout=_f16(wout), bout=_f32(self.proj_out.bias))

    def project_context(self, ctx):
        return [project_context(blk.attn2, ctx) for blk in self.transformer_blocks]

    def forward(self, x, context_kv=None, frames_per_video=1, cfg_repeat=1, colstats=None, want_colstats=False, **kwargs):
        """x [n, H, W, C] fp16 channels-last, n = b*t frames; context_kv from project_context().
        cfg_repeat = r > 1: x holds ONE copy of a batch whose r conditionings (classifier-free guidance: cond / uncond / ...)
        share everything up to here; everything that does not depend on the context - GroupNorm, proj_in, the whole
        self-attention of the first block, LayerNorm and the Q projection of its cross-attention - is computed once and the
        token stream is replicated r times right before the first cross-attention (context_kv holds r * b videos).  The
        result [r * n, H, W, C] equals the forward of the r-fold replicated input bit for bit.
        colstats: column moments of x (no statistics pass for the norm); want_colstats: let proj_out write the moments of the
        result for the GroupNorm of the layer behind this one.  -> (result, its moments or None)"""
        n, H, W, C = x.shape
        N_img = H * W
        pk = self.packed()
        # colstats: column moments of x from the convolution that produced it (ResBlock): the norm then needs no statistics pass
        stats = None if colstats is None else ops.group_norm_stats_from_colstats(colstats, n, N_img, C)
        D_in = pk["win"].shape[0]
        fold = N_img % 8 == 0 and spatial_fold_ok(n, N_img, C, D_in)      # the norm as per-frame weights / bias of proj_in
        a = None if fold else ops.group_norm(x.view(n, N_img, C), pk["gn_w"], pk["gn_b"], self.norm.eps, False, stats=stats)
        # The attention kernels address a frame's keys / values at 16-byte granularity: h*w must be a multiple of 8.  It is at every
        # level of 576x1024 and 320x512; for other --height / --width (e.g. 384x640: 6x10 = 60 tokens at the deepest level) each
        # frame's token rows are padded with zero rows up to the next multiple of 8 for the length of this block - all its layers
        # are row-wise except the attentions, which take nq = padded rows (results of pad rows are dropped) and nk = real keys.
        N = (N_img + 7) // 8 * 8
        xin = x.reshape(n * N_img, C)
        if N != N_img:
            def pad_frames(src):
                dst = torch.zeros((n * N, C), dtype=torch.float16, device=x.device)
                ops.copy2d(src, dst, n, N_img * C, N_img * C, N * C)           # one "row" per frame
                return dst
            xin, a = pad_frames(xin), pad_frames(a.view(n * N_img, C))
        tokens = n * N
        if fold:
            if stats is None:
                stats = ops.group_norm_stats(x.view(n, N_img, C))
            wn, bn = ops.group_norm_fold_linear(pk["win32"], pk["bin"], pk["gn_w"], pk["gn_b"], stats, self.norm.eps)
        # LayerNorm statistics of the token stream from the layer that writes it (VCX_GEMM_ROWSTATS, round 6): proj_in for norm1 of the
        # first block, attn1's output projection for norm2 - where the producer is the weight-stationary kernel (C = 320) and the
        # consumer a folded projection; row_stats (one read of the tensor) otherwise
        blk0 = self.transformer_blocks[0]
        rs_eps = blk0.norm1.eps
        rs = None
        if _folded(blk0.attn1.packed()["qk"], tokens, D_in, D_in) and ops.rowstats_ok(tokens, D_in, C, lda=C, unit_rows=N_img if fold else None):
            rs = ops.rowstats_buffer(tokens, x.device)
        if fold:
            t = ops.gemm_units(xin, wn, bn, unit_rows=N_img, rowstats=rs, rowstats_eps=rs_eps)
        else:
            t = ops.linear(a.view(tokens, C), pk["win"], pk["bin"], rowstats=rs, rowstats_eps=rs_eps)
        D, heads = t.shape[1], self.n_heads
        for bi, (blk, kv) in enumerate(zip(self.transformer_blocks, context_kv)):
            ln = blk.ln_params()
            a1, a2 = blk.attn1.packed(), blk.attn2.packed()
            # ---- self-attention over the h*w tokens of each frame
            # scale * log2(e) rides in the projections (sqrt of it on Q and on K: one fp16 rounding each, as without it), so the
            # attention kernel gets base-2 logits and its running max can live in the MFMA accumulator (VCX_ATTN_LOG2_LOGITS)
            pqk, pv, qk_alpha = a1["qk"], a1["v"], math.sqrt(blk.attn1.scale * ops.LOG2E)
            # norm1 folded into both projections (they read the token stream itself) wherever the pack is folded and the kernel
            # takes the shape; layer_norm + plain projections otherwise (ln_linear / ln_linear_t)
            st = rs if (bi == 0 and rs is not None) else (ops.row_stats(t, ln[0][2]) if pqk["colsum"] is not None else None)
            qk = ln_linear(t, pqk, ln[0], alpha=qk_alpha, stats=st)                 # [tokens, 2D]
            vt = ln_linear_t(t, pv, ln[0], stats=st)                                # [D, tokens]
            o = torch.empty((tokens, D), dtype=torch.float16, device=x.device)
            ops.flash_attn(qk, qk[:, D:], vt, o, n_groups=n, heads=heads, nq=N, nk=N_img, kv_rows=N, kv_div=1, ldq=2 * D,
                           ldk=2 * D, ldvt=tokens, ldo=D, scale=blk.attn1.scale, log2_logits=True)
            st2 = ops.rowstats_buffer(tokens, x.device) if (_folded(a2["q"], tokens, D, D) and ops.rowstats_ok(tokens, D, D, ldr=D)) else None
            t = ops.linear(o, a1["wo"], a1["bo"], residual=t, rowstats=st2, rowstats_eps=ln[1][2])
            # ---- cross-attention: softmax(Q K_txt) V_txt + softmax(Q K_img) V_img
            q2 = ln_linear(t, a2["q"], ln[1], alpha=blk.attn2.scale * ops.LOG2E, stats=st2)
            if bi == 0 and cfg_repeat > 1:      # from here on the r conditionings differ
                t, q2, xin = ops.repeat_rows(t, cfg_repeat), ops.repeat_rows(q2, cfg_repeat), ops.repeat_rows(xin, cfg_repeat)
                n, tokens = n * cfg_repeat, tokens * cfg_repeat
            o2 = torch.empty((tokens, D), dtype=torch.float16, device=x.device)
            nb = kv.n_txt_rows // 80
            if kv.k_img is not None:     # text (+) image in one pass over q2 / o2
                ops.flash_attn_dual(q2, kv.k_txt, kv.vt_txt, kv.k_img, kv.vt_img, o2, n_groups=n, heads=heads, nq=N,
                                    nk1=kv.n_txt, kv_rows1=80, kv_div1=frames_per_video, ldk1=D, ldvt1=nb * 80,
                                    nk2=kv.n_img, kv_rows2=kv.n_img, kv_div2=1 if kv.img_per_frame else frames_per_video,
                                    ldk2=D, ldvt2=kv.vt_img.shape[1], ldq=D, ldo=D, scale=blk.attn2.scale, log2_logits=True)
            else:
                ops.flash_attn(q2, kv.k_txt, kv.vt_txt, o2, n_groups=n, heads=heads, nq=N, nk=kv.n_txt,
                               kv_rows=80, kv_div=frames_per_video, ldq=D, ldk=D, ldvt=nb * 80, ldo=D, scale=blk.attn2.scale,
                               log2_logits=True)
            t = ops.linear(o2, a2["wo"], a2["bo"], residual=t)
            # ---- feed-forward
            t = blk.ff.run(t, ln[2])
        kw, cs_out = out_kwargs(None, want_colstats and N == N_img, tokens, N_img, D, C, x.device)
        out = ops.linear(t, pk["wout"], pk["bout"], residual=xin, **kw)
        if N != N_img:
            unpadded = torch.empty((n * N_img, C), dtype=torch.float16, device=x.device)
            ops.copy2d(out, unpadded, n, N_img * C, N * C, N_img * C)
            out = unpadded
        return out.view(n, H, W, C), cs_out


class TemporalTransformer(PackedModule):
    """Reference attention.py:313-412 with only_self_att=True: GN over (C/32, T, H, W) -> proj_in -> [LN, self-attn
    over T, LN, self-attn over T again (context None, :389-390), LN, GEGLU-FF] -> proj_out -> + x."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, use_checkpoint=True,
                 use_linear=False, only_self_att=True, causal_attention=False, causal_block_size=1,
                 relative_position=False, temporal_length=None):
        super().__init__()
        if not only_self_att:
            raise NotImplementedError("ViewCrafter uses temporal self-attention only")
        if relative_position and temporal_length is None:
            raise AssertionError("relative_position needs temporal_length (reference attention.py:339)")
        # causal_attention (reference attention.py:343-345, 377-384: a lower-triangular mask over the frames, handed to attn1 AND attn2 of every block,
        # :241-243; `use_causal_attention`, not used by the ViewCrafter YAMLs): VCX_ATTN_CAUSAL of the temporal attention kernel
        if causal_attention and temporal_length is None:
            raise AssertionError("causal_attention needs temporal_length (reference attention.py:344)")
        self.causal_attention = bool(causal_attention)
        self.temporal_length = temporal_length
        self.only_self_att = only_self_att
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.n_heads = n_heads
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        if not use_linear:   # init_attn: Conv1d k=1 (openaimodel3d.py:389-399); same matmul, weight [inner, C, 1]
            self.proj_in = nn.Conv1d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
            self.proj_out = nn.Conv1d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0)
        else:
            self.proj_in = nn.Linear(in_channels, inner_dim)
            self.proj_out = nn.Linear(inner_dim, in_channels)
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)

        def attention_cls(**kw):
            return CrossAttention(temporal_length=temporal_length, relative_position=bool(relative_position), **kw)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=None,
                                  attention_cls=attention_cls, checkpoint=use_checkpoint) for _ in range(depth)])
        for blk in self.transformer_blocks:
            blk.set_kind("temporal")
        self.use_linear = use_linear

    def _pack(self):
        win = self.proj_in.weight.detach().reshape(self.proj_in.weight.shape[0], -1)
        wout = self.proj_out.weight.detach().reshape(self.proj_out.weight.shape[0], -1)
        return dict(gn_w=_f32(self.norm.weight), gn_b=_f32(self.norm.bias), win=_f16(win), win32=_f32(win), bin=_f32(self.proj_in.bias),
                    wout=_f16(wout), bout=_f32(self.proj_out.bias))

    def forward(self, x, context=None, colstats=None, want_colstats=False, target=None):
        """x [B, T, P, C] fp16 channels-last (P = h*w).  colstats: column moments of x (the per-video norm then needs no statistics
        pass); want_colstats / target: proj_out writes the moments of the result / the result itself into a concat buffer
        (lvdm/modules/flow.py).  -> (result [B, T, P, C or ld], its moments or None)"""
        B, T, P, C = x.shape
        pk = self.packed()
        tokens = B * T * P
        xin = x.reshape(tokens, C)
        stats = None if (colstats is None or GN_STATS_LEVEL < 2) else ops.group_norm_stats_from_colstats(colstats, B, T * P, C)
        D, heads = pk["win"].shape[0], self.n_heads
        rows = T * P
        if GN_FOLD and C % 64 == 0 and 2 * rows * C >= GN_FOLD_MIN_BYTES and 2 * rows * max(C, D) < 0xFFFF0000 and ops.tune_get("GEMM_DMA") != 0:
            # the norm as per-video weights / bias of proj_in; the projection reads the un-normalised rows of each video
            if stats is None:
                stats = ops.group_norm_stats(x.view(B, rows, C))
            wn, bn = ops.group_norm_fold_linear(pk["win32"], pk["bin"], pk["gn_w"], pk["gn_b"], stats, self.norm.eps)
            fold = True
        else:
            a = ops.group_norm(x.view(B, rows, C), pk["gn_w"], pk["gn_b"], self.norm.eps, False, stats=stats)
            fold = False
        # LayerNorm statistics from the producing layer (VCX_GEMM_ROWSTATS): proj_in for norm1, attn1's output projection for norm2
        blk0 = self.transformer_blocks[0]
        st = None
        if _folded(blk0.attn1.packed()["qkv"], tokens, D, D) and ops.rowstats_ok(tokens, D, C, lda=C, unit_rows=rows if fold else None):
            st = ops.rowstats_buffer(tokens, x.device)
        if fold:
            t = ops.gemm_units(xin, wn, bn, unit_rows=rows, rowstats=st, rowstats_eps=blk0.norm1.eps)
        else:
            t = ops.linear(a.view(tokens, C), pk["win"], pk["bin"], rowstats=st, rowstats_eps=blk0.norm1.eps)
        for blk in self.transformer_blocks:
            ln = blk.ln_params()
            for ai, (attn, lnp) in enumerate(((blk.attn1, ln[0]), (blk.attn2, ln[1]))):
                ap = attn.packed()
                qkv = ln_linear(t, ap["qkv"], lnp, stats=st)                         # [tokens, 3D]
                o = torch.empty((tokens, D), dtype=torch.float16, device=x.device)
                if attn.relative_position:
                    # logits += q Ek^T (per head: a [tokens, 64] x [64, 64] GEMM on the q columns of qkv), out += (probabilities by clipped distance) Ev
                    relg = torch.empty((tokens, heads, 64), dtype=torch.float16, device=x.device)
                    relp = torch.zeros((tokens, heads, 64), dtype=torch.float16, device=x.device)
                    for h_ in range(heads):
                        ops.gemm(qkv[:, h_ * 64:], ap["rel_k"], M=tokens, N=64, K=64, lda=3 * D, out=relg[:, h_], ldc=heads * 64)
                    ops.temporal_attn_rel(qkv, o, relg, relp, R=ap["rel_R"], B=B, T=T, P=P, heads=heads, ld=3 * D, k_off=D, v_off=2 * D, ldo=D,
                                          scale=attn.scale, causal=self.causal_attention)
                    for h_ in range(heads):
                        ops.gemm(relp[:, h_], ap["rel_v"], M=tokens, N=64, K=64, lda=heads * 64, out=o[:, h_ * 64:], ldc=D, residual=o[:, h_ * 64:], ldr=D)
                else:
                    ops.temporal_attn(qkv, o, B=B, T=T, P=P, heads=heads, ld=3 * D, k_off=D, v_off=2 * D, ldo=D,
                                      scale=attn.scale, causal=self.causal_attention)
                st = None
                if ai == 0 and _folded(blk.attn2.packed()["qkv"], tokens, D, D) and ops.rowstats_ok(tokens, D, D, ldr=D):
                    st = ops.rowstats_buffer(tokens, x.device)
                t = ops.linear(o, ap["wo"], ap["bo"], residual=t, rowstats=st, rowstats_eps=ln[1][2])
            t = blk.ff.run(t, ln[2])
        kw, cs_out = out_kwargs(target, want_colstats, tokens, P, D, C, x.device)
        out = ops.linear(t, pk["wout"], pk["bout"], residual=xin, **kw)
        return out.view(B, T, P, -1), cs_out
