"""The lvdm 3-D UNet denoiser on the gfx950 kernels.

Drop-in for the reference's lvdm/modules/networks/openaimodel3d.py::UNetModel on the ViewCrafter inference path: same
constructor keywords (configs/inference_pvd_*.yaml `unet_config.params`), same state-dict names/shapes (strict
checkpoint load), same call signature `forward(x, timesteps, context, features_adapter=None, fs=None, **ignored)`.

MI355X-first differences from the reference implementation (none changes the result beyond fp16 rounding):
  * one channels-last fp16 master layout [B, T, H, W, C]: the reference's '(b t) c h w' <-> 'b c t h w' <->
    '(b h w) t c' copies are views here;
  * every conv is an implicit GEMM on MFMA (csrc/gemm.hip) with bias / timestep-embedding add / residual / GEGLU fused
    into the epilogue; nearest-2x upsampling is fused into the following conv's gather;
  * cross-attention K/V projections depend only on the conditioning and are cached across DDIM steps;
  * cond/uncond (CFG) run as one B=2 forward, halving weight traffic.
There is no PyTorch compute fallback: without libvcx.so and a gfx950 device forward() raises.
"""
import os

import torch
from torch import nn

from .... import ops
from ....packing import pack_conv
from ..attention import PackedModule, SpatialTransformer, TemporalTransformer, _f16, _f32
from ..flow import CAT_SPLIT, GN_EPILOGUE_STATS, GN_STATS_LEVEL, CatTarget, Flow, out_kwargs as _out_kwargs

# A ResBlock's 1x1 skip convolution as a K tail of its second 3x3 convolution (round 6, include/vcx.h tail_a0 / tail_a1).  VCX_SKIP_FOLD=0: the
# separate convolution + residual of rounds 1-5 (A/B runs; also switches the split concat off, which needs the fold).
SKIP_FOLD = os.environ.get("VCX_SKIP_FOLD", "1") != "0"
# emb_layers of all ResBlocks as ONE projection per forward (round 6); VCX_EMB_BATCHED=0: one launch per block
EMB_BATCHED = os.environ.get("VCX_EMB_BATCHED", "1") != "0"

class TimestepBlock(nn.Module):
    """Marker: modules whose forward takes the timestep embedding (reference openaimodel3d.py:19-28)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Reference openaimodel3d.py:30-48: routes (emb | context | 5-D view) to each child by type.
    x is channels-last [n = b*t, H, W, C]."""

    def forward(self, x, emb, context=None, batch_size=None, cfg_repeat=1, flow=None, x2=None):
        """x2: the skip half of a channel concat [x | x2] that the first layer (a ResBlock) reads in place (lvdm/modules/flow.py).
        cfg_repeat = r > 1 (only for a block that holds the first SpatialTransformer of the graph): the input is one copy
        of an r-fold replicated batch; the transformer replicates it where the conditionings start to differ and the
        layers after it see batch_size * r videos.  flow (Flow): moments of x in, moments of the result out, and the concat
        target of the last layer; with a target the returned tensor is a strided view of its left columns."""
        layers = list(self)
        if x2 is not None and not (layers and isinstance(layers[0], ResBlock)):
            raise ValueError("a split concat [x | x2] can only enter a block that starts with a ResBlock")
        colstats = flow.colstats if flow is not None else None      # column moments of x from the layer that produced it
        for i, layer in enumerate(layers):
            last = i + 1 == len(layers)
            nxt = None if last else layers[i + 1]
            if GN_STATS_LEVEL >= 2:     # whoever consumes the output starts with a GroupNorm
                want = isinstance(nxt, (ResBlock, SpatialTransformer, TemporalTransformer)) if not last else bool(flow is not None and flow.want)
            else:                       # round-3 behaviour: only the SpatialTransformer right behind a ResBlock
                want = isinstance(layer, ResBlock) and isinstance(nxt, SpatialTransformer)
            target = flow.target if (last and flow is not None) else None
            if isinstance(layer, ResBlock):
                x, colstats = layer(x, emb, batch_size=batch_size, want_colstats=want, colstats=colstats, target=target, x2=x2 if i == 0 else None)
            elif isinstance(layer, SpatialTransformer):
                x, colstats = layer(x, context_kv=context[id(layer)], frames_per_video=x.shape[0] // batch_size, cfg_repeat=cfg_repeat,
                                    colstats=colstats, want_colstats=want and GN_STATS_LEVEL >= 2)
                if cfg_repeat > 1:
                    # x now holds batch_size * r videos: a ResBlock FOLLOWING the transformer inside this block reads one emb row
                    # per video through a raw pointer (rowadd), so emb must grow with x here, not after the block returns
                    emb = ops.repeat_rows(emb, cfg_repeat)
                batch_size, cfg_repeat = batch_size * cfg_repeat, 1
            elif isinstance(layer, TemporalTransformer):
                n, H, W, C = x.shape
                x, colstats = layer(x.view(batch_size, n // batch_size, H * W, C), colstats=colstats, want_colstats=want, target=target)
                x = x.view(n, H, W, -1)
            elif isinstance(layer, (Downsample, Upsample)):
                # (without a convolution - conv_resample: false - there is no epilogue to write moments or a concat target from: the plain result
                # is copied into the target below, like any other layer that cannot write there)
                x, colstats = layer(x, want_colstats=want, target=target if layer.use_conv else None)
            else:
                x, colstats = layer(x), None
            if target is not None and (not isinstance(layer, (ResBlock, TemporalTransformer, Downsample, Upsample))
                                       or (isinstance(layer, (Downsample, Upsample)) and not layer.use_conv)):
                # a last layer that cannot write into a concat target (a SpatialTransformer in a graph without temporal attention, a
                # bare convolution): its result is copied into the target's left columns, and the next norm makes its statistics pass
                n_, H_, W_, C_ = x.shape
                ops.copy2d(x.reshape(n_ * H_ * W_, C_), target.data, n_ * H_ * W_, C_, C_, target.data_ld)
                target.moments, colstats = None, None
                x = target.data.view(n_, H_, W_, target.data_ld)
        if flow is not None:
            flow.colstats = colstats
        if flow is not None and flow.target is not None:      # x is the whole concat buffer: hand back the block's own columns
            return x[..., :flow.target.c_left]
        return x


def _adapter_rows(feat, frames):
    """Adapter maps for a tensor of `frames` frames under the shared guidance prefix: the caller may hand maps for the
    un-replicated batch (b t frames: repeated where the r evaluations already flow as one batch) or for the replicated batch the
    reference's forward sees (r b t frames, r identical copies: the first b t are taken while the prefix still runs un-replicated)."""
    if feat.shape[0] == frames:
        return feat
    if frames % feat.shape[0] == 0:
        return feat.repeat(frames // feat.shape[0], 1, 1, 1)
    if feat.shape[0] % frames == 0:
        return feat[:frames]
    raise ValueError(f"features_adapter map of {feat.shape[0]} frames for a tensor of {frames} frames")


def _out_channels_of(block):
    """channels of the tensor a TimestepEmbedSequential returns"""
    last = list(block)[-1]
    return last.out_channels if hasattr(last, "out_channels") else last.in_channels


class Downsample(PackedModule):
    """Reference openaimodel3d.py:51-77 (use_conv=True, dims=2): Conv2d 3x3 stride 2 pad 1 stored as `.op`."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if dims != 2 or padding != 1:
            raise NotImplementedError("ViewCrafter uses dims=2")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:       # conv_resample: false (reference openaimodel3d.py:70-72; not used by the ViewCrafter YAMLs): AvgPool2d(2, 2), no parameters
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def _pack(self):
        return dict(w=_f16(pack_conv(self.op.weight.detach())), b=_f32(self.op.bias)) if self.use_conv else {}

    def forward(self, x, want_colstats=False, target=None):
        """-> (y [n, H/2, W/2, C] - or [.., ld] when written into a concat target -, column moments of y or None)"""
        if not self.use_conv:
            return ops.avgpool2x2(x), None
        pk = self.packed()
        n, H, W, cin = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        kw, cs = _out_kwargs(target, want_colstats, n * Ho * Wo, Ho * Wo, cin, self.out_channels, x.device, in_rows=n * H * W)
        return ops.conv2d(x, pk["w"], pk["b"], kh=3, kw=3, stride=2, **kw), cs


class Upsample(PackedModule):
    """Reference openaimodel3d.py:80-106: nearest 2x then Conv2d 3x3 (`.conv`); the interpolation is folded into the
    convolution's gather (VCX mode-1 `ups`)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if dims != 2 or padding != 1:
            raise NotImplementedError("ViewCrafter uses dims=2")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def _pack(self):
        return dict(w=_f16(pack_conv(self.conv.weight.detach())), b=_f32(self.conv.bias)) if self.use_conv else {}

    def forward(self, x, want_colstats=False, target=None):
        if not self.use_conv:      # conv_resample: false (reference openaimodel3d.py:98-103): the nearest 2x alone
            return ops.upsample2x(x), None
        pk = self.packed()
        n, H, W, cin = x.shape
        kw, cs = _out_kwargs(target, want_colstats, n * 4 * H * W, 4 * H * W, cin, self.out_channels, x.device, in_rows=n * H * W)
        return ops.conv2d(x, pk["w"], pk["b"], kh=3, kw=3, ups=1, **kw), cs


class TemporalConvBlock(PackedModule):
    """Reference openaimodel3d.py:239-279: 4 x [GroupNorm(32) over (C/32, T, H, W) -> SiLU -> Conv3d (3,1,1)] + identity.
    Module indices follow the reference (conv1: conv at .2; conv2-4: Dropout at .2, conv at .3)."""

    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError("tempspatial_aware is False in the ViewCrafter configs")
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        k, p = (3, 1, 1), (1, 0, 0)
        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_channels), nn.SiLU(), nn.Conv3d(in_channels, out_channels, k, padding=p))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_channels, in_channels, k, padding=p))
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def _pack(self):
        out = []
        for seq in (self.conv1, self.conv2, self.conv3, self.conv4):
            gn, conv = seq[0], seq[-1]
            out.append((_f32(gn.weight), _f32(gn.bias), gn.eps, _f16(pack_conv(conv.weight.detach())), _f32(conv.bias)))
        return out

    def forward(self, x, colstats=None, want_colstats=False, target=None):
        """x [B, T, P, C] fp16.  `colstats`: column moments of x written by the layer that produced it (ops.gemm colstats=):
        the first norm then needs no statistics pass; the three inner norms get theirs from this block's own convolutions.
        want_colstats: also produce the moments of the OUTPUT (for the per-frame GroupNorm behind this block); target: write the
        output (and its moments) into a concat buffer.  -> (y, moments of y or None)"""
        B, T, P, C = x.shape
        y = x
        stages = self.packed()
        for i, (gw, gb, eps, w, b) in enumerate(stages):
            cy = y.shape[-1]
            stats = None if colstats is None else ops.group_norm_stats_from_colstats(colstats, B, T * P, cy)
            a = ops.group_norm(y.reshape(B, T * P, cy), gw, gb, eps, True, stats=stats)
            last = i == len(stages) - 1
            cout = w.shape[0]
            kw = {}
            if last and target is not None:
                kw = target.kwargs(cy, B * T * P)
                colstats = target.moments
            else:
                need = (not last and ops.colstats_ok(B * T * P, T * P, cy, cout)) or (last and want_colstats and ops.colstats_ok(B * T * P, P, cy, cout))
                colstats = ops.colstats_buffer(B * T * P, cout, x.device) if (GN_EPILOGUE_STATS and need) else None
                if colstats is not None:
                    kw = dict(colstats=colstats)
            y = ops.temporal_conv3(a.view(B, T, P, -1), w, b, residual=x.reshape(B * T * P, C) if last else None, **kw)
        return y, colstats


class ResBlock(PackedModule, TimestepBlock):
    """Reference openaimodel3d.py:109-236 (no up/down, 1x1 skip when channels change; use_scale_shift_norm - the FiLM form of
    :221-225, not used by the shipped YAMLs - folds (1 + scale, shift) into per-video GroupNorm affine parameters)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, use_conv=False, up=False, down=False, use_temporal_conv=False,
                 tempspatial_aware=False):
        super().__init__()
        if use_conv or dims != 2 or (up and down):
            raise NotImplementedError("ResBlock variant not used by the ViewCrafter configs")
        # up / down (reference openaimodel3d.py:160-165, 210-215; `resblock_updown: true`, not used by the ViewCrafter YAMLs): the sampling sits
        # between SiLU and the first convolution (h_upd) and on the skip path (x_upd), both without parameters
        self.up, self.down = bool(up), bool(down)
        self.updown = self.up or self.down
        self.channels, self.emb_channels = channels, emb_channels
        self.out_channels = out_channels or channels
        self.use_temporal_conv = use_temporal_conv
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        nn.init.zeros_(self.out_layers[-1].weight)
        nn.init.zeros_(self.out_layers[-1].bias)
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        if self.use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1,
                                                   spatial_aware=tempspatial_aware)
        self._emb_all = None       # set by UNetModel._forward for the length of one forward: this block's columns of the batched emb projection

    def _pack(self):
        gn1, c1 = self.in_layers[0], self.in_layers[2]
        gn2, c2 = self.out_layers[0], self.out_layers[3]
        pk = dict(g1=(_f32(gn1.weight), _f32(gn1.bias), gn1.eps), w1=_f16(pack_conv(c1.weight.detach())), b1=_f32(c1.bias),
                  g2=(_f32(gn2.weight), _f32(gn2.bias), gn2.eps), w2=_f16(pack_conv(c2.weight.detach())), b2=_f32(c2.bias),
                  we=_f16(self.emb_layers[1].weight), be=_f32(self.emb_layers[1].bias))
        if not isinstance(self.skip_connection, nn.Identity):
            pk["ws"] = _f16(pack_conv(self.skip_connection.weight.detach()))
            pk["bs"] = _f32(self.skip_connection.bias)
            # the 1x1 skip convolution as a K tail of the second 3x3 convolution (include/vcx.h tail_a0 / tail_a1): its weight columns behind
            # that convolution's, the two biases summed - `return self.skip_connection(x) + h` (reference openaimodel3d.py:228-235) in one
            # launch: neither the skip tensor nor its re-read as a residual exist, and the matrix work runs at the long-K rate
            pk["w2s"] = torch.cat([pk["w2"], pk["ws"]], dim=1).contiguous()
            pk["b2s"] = (pk["b2"] + pk["bs"]).contiguous()
        return pk

    def forward(self, x, emb, batch_size=None, want_colstats=False, colstats=None, target=None, x2=None):
        """x2 [n, H, W, C2]: the input is the channel concat [x | x2] (up path: h and the skip, reference openaimodel3d.py:596), read in place
        where the kernels can (statistics from `colstats`, the skip convolution folded: SKIP_FOLD) and materialised otherwise.
        x [n, H, W, Cin] fp16; emb = SiLU(time+fs embedding) as fp16 [B, emb_channels] (one row per video: the
        reference repeats it over the T frames, openaimodel3d.py:563).  colstats: column moments of x (the in_layers norm then
        needs no statistics pass); want_colstats: produce the moments of the output for a per-frame GroupNorm behind this block
        (SpatialTransformer.norm, the next block's in_layers); target: write output and moments into a concat buffer.
        -> (h, moments of h or None)"""
        n, H, W, cin = x.shape
        pk = self.packed()
        cout = self.out_channels
        B = emb.shape[0]
        M = n * H * W
        tail_ks = [cin] if x2 is None else [cin, x2.shape[-1]]
        fold_skip = SKIP_FOLD and "ws" in pk and ops.conv_tail_ok(M, cout, cout, 9, tail_ks)
        if x2 is not None and not (fold_skip and colstats is not None and cin % 8 == 0 and x.is_contiguous() and x2.is_contiguous()):
            x, x2 = ops.concat_channels(x.reshape(M, cin), x2.reshape(M, x2.shape[-1])).view(n, H, W, -1), None      # the materialised concat
            tail_ks = [x.shape[-1]]
            fold_skip = SKIP_FOLD and "ws" in pk and ops.conv_tail_ok(M, cout, cout, 9, tail_ks)
        c1, cin = cin, sum(tail_ks) if x2 is not None else x.shape[-1]
        stats_in = None if colstats is None else ops.group_norm_stats_from_colstats(colstats, n, H * W, cin)
        a = ops.group_norm(x.view(n, H * W, c1 if x2 is not None else cin), *pk["g1"], True, stats=stats_in, x2=None if x2 is None else x2.view(n, H * W, -1))
        conv1_kw = {}
        if self.updown:
            if x2 is not None:
                raise ValueError("an up / down ResBlock does not take a split concat")
            a = a.view(n, H, W, cin)
            if self.up:       # nearest 2x in front of the convolution = the convolution's fused gather; the skip path needs the tensor
                conv1_kw, x, H, W = dict(ups=1), ops.upsample2x(x), 2 * H, 2 * W
            else:
                a, x, H, W = ops.avgpool2x2(a), ops.avgpool2x2(x), H // 2, W // 2
            M = n * H * W
            fold_skip = SKIP_FOLD and "ws" in pk and ops.conv_tail_ok(M, cout, cout, 9, [cin])
        # [B, Cout] fp32: this block's columns of the one projection UNetModel made of the embedding, or its own launch (a block used alone)
        emb_out = self._emb_all[:B] if self._emb_all is not None else ops.linear(emb, pk["we"], pk["be"], out_f32=True)
        # The norms behind this block's own convolutions take their statistics from those convolutions' epilogues (column moments
        # per 64-row strip, VCX_GEMM_COLSTATS) instead of a pass over the tensor: out_layers' norm (per frame) from conv 1, the
        # first norm of the temporal block (per video) from conv 2 - where frames are whole strips (not at 9x16 = 144 pixels).
        hin, win = a.shape[1:3] if self.updown else (H, W)       # the first convolution's INPUT grid (up: half the output's)
        cs1 = ops.colstats_buffer(M, cout, x.device) if (GN_EPILOGUE_STATS and ops.colstats_ok(M, H * W, cin, cout, in_rows=n * hin * win)) else None
        if not self.use_scale_shift_norm:
            h = ops.conv2d(a.view(n, hin, win, cin), pk["w1"], pk["b1"], kh=3, kw=3, rowadd=emb_out, rowadd_div=(n // B) * H * W, colstats=cs1, **conv1_kw)
            stats = None if cs1 is None else ops.group_norm_stats_from_colstats(cs1, n, H * W, cout)
            a = ops.group_norm(h.view(n, H * W, cout), *pk["g2"], True, stats=stats)
        else:
            # norm(h) * (1 + scale) + shift with (scale, shift) = the two halves of emb_out, one pair per video: the GroupNorm's
            # affine parameters of that video become gamma (1 + scale) and beta (1 + scale) + shift ([C] vectors, a few hundred
            # floats of host-side plumbing per video), the statistics are untouched
            h = ops.conv2d(a.view(n, hin, win, cin), pk["w1"], pk["b1"], kh=3, kw=3, colstats=cs1, **conv1_kw)
            stats = None if cs1 is None else ops.group_norm_stats_from_colstats(cs1, n, H * W, cout)
            h3, a, fpv = h.view(n, H * W, cout), torch.empty((n, H * W, cout), dtype=torch.float16, device=x.device), n // B
            gam, bet, eps = pk["g2"]
            for v in range(B):
                sc1 = 1.0 + emb_out[v, :cout]
                ops.group_norm(h3[v * fpv:(v + 1) * fpv], (gam * sc1).contiguous(), (bet * sc1 + emb_out[v, cout:]).contiguous(), eps, True,
                               stats=None if stats is None else stats[v * fpv:(v + 1) * fpv], out=a[v * fpv:(v + 1) * fpv])
        if fold_skip:       # skip_connection(x) rides in the K loop of the second convolution: x (both halves of a split concat) is its K tail
            w2, b2 = pk["w2s"], pk["b2s"]
            skip_kw = dict(tail=[x.reshape(M, -1)] + ([] if x2 is None else [x2.reshape(M, -1)]))
        else:
            w2, b2 = pk["w2"], pk["b2"]
            skip_kw = dict(residual=ops.conv2d(x, pk["ws"], pk["bs"], kh=1, kw=1).view(M, cout) if "ws" in pk else x.reshape(M, cout))
        temporal = self.use_temporal_conv and batch_size
        if temporal:
            need2 = ops.colstats_ok(M, (n // batch_size) * H * W, cout, cout)
            cs2 = ops.colstats_buffer(M, cout, x.device) if (GN_EPILOGUE_STATS and need2) else None
            h = ops.conv2d(a.view(n, H, W, cout), w2, b2, kh=3, kw=3, colstats=cs2, **skip_kw)
            h, cs_out = self.temopral_conv(h.view(batch_size, n // batch_size, H * W, cout), colstats=cs2, want_colstats=want_colstats,
                                           target=target)
            return h.view(n, H, W, -1), cs_out
        if target is not None:
            kw = target.kwargs(cout, M)
            cs_out = target.moments
        else:
            cs_out = ops.colstats_buffer(M, cout, x.device) if (GN_EPILOGUE_STATS and want_colstats and ops.colstats_ok(M, H * W, cout, cout)) else None
            kw = {} if cs_out is None else dict(colstats=cs_out)
        h = ops.conv2d(a.view(n, H, W, cout), w2, b2, kh=3, kw=3, **skip_kw, **kw)
        return h, cs_out


class UNetModel(PackedModule):
    """Reference openaimodel3d.py:281-603."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False,
                 resblock_updown=False, num_heads=-1, num_head_channels=-1, transformer_depth=1, use_linear=False,
                 use_checkpoint=False, temporal_conv=False, tempspatial_aware=False, temporal_attention=True,
                 use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        if num_head_channels == -1:
            raise NotImplementedError("set num_head_channels (the ViewCrafter configs use 64)")
        if dims != 2:
            raise NotImplementedError("UNet variant not used by the ViewCrafter configs")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.dropout, self.channel_mult, self.conv_resample = dropout, channel_mult, conv_resample
        self.temporal_attention = temporal_attention
        self.use_checkpoint = use_checkpoint       # accepted and ignored (inference only)
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.addition_attention = addition_attention
        self.temporal_length = temporal_length
        self.image_cross_attention = image_cross_attention
        self.default_fs, self.fs_condition = default_fs, fs_condition
        self.context_dim = context_dim
        time_embed_dim = model_channels * 4

        def spatial(ch):
            return SpatialTransformer(ch, ch // num_head_channels, num_head_channels, depth=transformer_depth,
                                      context_dim=context_dim, use_linear=use_linear, use_checkpoint=use_checkpoint,
                                      disable_self_attn=False, video_length=temporal_length,
                                      image_cross_attention=image_cross_attention,
                                      image_cross_attention_scale_learnable=image_cross_attention_scale_learnable)

        def temporal(ch, heads=None, linear=use_linear, causal=use_causal_attention):
            return TemporalTransformer(ch, heads or ch // num_head_channels, num_head_channels, depth=transformer_depth,
                                       context_dim=context_dim, use_linear=linear, use_checkpoint=use_checkpoint,
                                       only_self_att=temporal_selfatt_only, causal_attention=causal,
                                       relative_position=use_relative_position, temporal_length=temporal_length)

        def res(cin, cout):
            return ResBlock(cin, time_embed_dim, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, tempspatial_aware=tempspatial_aware,
                            use_temporal_conv=temporal_conv)

        self.time_embed = nn.Sequential(nn.Linear(model_channels, time_embed_dim), nn.SiLU(),
                                        nn.Linear(time_embed_dim, time_embed_dim))
        if fs_condition:
            self.fps_embedding = nn.Sequential(nn.Linear(model_channels, time_embed_dim), nn.SiLU(),
                                               nn.Linear(time_embed_dim, time_embed_dim))
            nn.init.zeros_(self.fps_embedding[-1].weight)
            nn.init.zeros_(self.fps_embedding[-1].bias)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        if addition_attention:   # 8 heads, Conv1d projections (use_linear not forwarded), openaimodel3d.py:387-399
            self.init_attn = TimestepEmbedSequential(temporal(model_channels, heads=8, linear=False, causal=False))       # (causal_attention=False there, :398)
        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(spatial(ch))
                    if temporal_attention:
                        layers.append(temporal(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                # (resblock_updown: reference openaimodel3d.py:441-451 - a ResBlock(down=True) without temporal convolution in the place of the stride-2 convolution)
                self.input_blocks.append(TimestepEmbedSequential(
                    ResBlock(ch, time_embed_dim, dropout, out_channels=ch, dims=dims, use_checkpoint=use_checkpoint,
                             use_scale_shift_norm=use_scale_shift_norm, down=True)
                    if resblock_updown else Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2
        layers = [res(ch, ch), spatial(ch)]
        if temporal_attention:
            layers.append(temporal(ch))
        layers.append(res(ch, ch))
        self.middle_block = TimestepEmbedSequential(*layers)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [res(ch + ich, mult * model_channels)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(spatial(ch))
                    if temporal_attention:
                        layers.append(temporal(ch))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, time_embed_dim, dropout, out_channels=ch, dims=dims, use_checkpoint=use_checkpoint,
                                           use_scale_shift_norm=use_scale_shift_norm, up=True)
                                  if resblock_updown else Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        nn.init.zeros_(self.out[-1].weight)
        nn.init.zeros_(self.out[-1].bias)
        self._ctx_cache = {}
        # hipGraph replay of the whole forward (launch-bound at the small UNet levels): opt-in, see forward_graphed()
        self.use_hip_graph = False
        self._graphs = {}

    # ------------------------------------------------------------------ packing / caches
    def _drop_packed(self):
        super()._drop_packed()
        self._ctx_cache = {}
        self._graphs = {}

    def _apply(self, fn, recurse=True):
        self._ctx_cache = {}
        self._graphs = {}
        return super()._apply(fn, recurse)

    def _pack(self):
        c0 = self.input_blocks[0][0]
        gn, co = self.out[0], self.out[2]
        pk = dict(w_in=_f16(pack_conv(c0.weight.detach())), b_in=_f32(c0.bias),
                  g_out=(_f32(gn.weight), _f32(gn.bias), gn.eps), w_out=_f16(pack_conv(co.weight.detach())), b_out=_f32(co.bias),
                  te=[(_f16(l.weight), _f32(l.bias)) for l in (self.time_embed[0], self.time_embed[2])])
        if self.fs_condition:
            pk["fe"] = [(_f16(l.weight), _f32(l.bias)) for l in (self.fps_embedding[0], self.fps_embedding[2])]
        # emb_layers of ALL ResBlocks as one projection (round 6): `self.emb_layers(emb)` of every block (reference openaimodel3d.py:216-219)
        # reads the same embedding, and 22 launches of an M = 2 GEMM are 22 x 24 us of latency per forward - one launch writes
        # [B, sum of the blocks' widths] and each block's convolution reads its columns of that matrix (VCX_GEMM_ROWADD with rowadd_ld)
        rbs = self.resblocks()
        pk["emb_w"] = torch.cat([_f16(rb.emb_layers[1].weight) for rb in rbs], dim=0).contiguous()
        pk["emb_b"] = torch.cat([_f32(rb.emb_layers[1].bias) for rb in rbs], dim=0).contiguous()
        off, pk["emb_off"] = 0, {}
        for rb in rbs:
            n_e = rb.emb_layers[1].weight.shape[0]
            pk["emb_off"][id(rb)] = (off, n_e)
            off += n_e
        return pk

    def resblocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def _foreign_params(self):      # the pack embeds the ResBlocks' emb_layers: rebuilt when any of them changes (PackedModule.packed)
        return [p for rb in self.resblocks() for p in (rb.emb_layers[1].weight, rb.emb_layers[1].bias)]

    def spatial_transformers(self):
        return [m for m in self.modules() if isinstance(m, SpatialTransformer)]

    def _context_kv(self, context, t):
        """Split/pad the conditioning (openaimodel3d.py:553-562) and project it through every SpatialTransformer's
        to_k/to_v(/_ip) once; cached on the identity of the context tensor (constant over the DDIM loop)."""
        key = (context.data_ptr(), context._version, tuple(context.shape), t, str(context.device))
        hit = self._ctx_cache.get(key)
        if hit is not None:
            return hit
        b, L, D = context.shape
        ctx16 = ops.to_f16(context) if context.dtype != torch.float16 else context.contiguous()
        txt = torch.zeros((b, 80, D), dtype=torch.float16, device=context.device)
        ntxt = min(L, 77)
        txt[:, :ntxt] = ctx16[:, :ntxt]
        ctx = dict(txt=txt.view(b * 80, D), img=None, n_img=0, per_frame=False)
        if self.image_cross_attention and L > 77:
            n_img = L - 77
            if L == 77 + t * 16:            # per-frame image tokens (the hard-coded branch, openaimodel3d.py:556-560)
                ctx.update(img=ctx16[:, 77:].reshape(b * t * 16, D).contiguous(), n_img=16, per_frame=True)
            else:
                if n_img % 8 != 0:
                    raise ValueError(f"image context length {n_img} must be a multiple of 8")
                ctx.update(img=ctx16[:, 77:].reshape(b * n_img, D).contiguous(), n_img=n_img, per_frame=False)
        ctx["n_txt"] = ntxt
        kv = {id(st): st.project_context(ctx) for st in self.spatial_transformers()}
        if len(self._ctx_cache) >= 4:
            self._ctx_cache.clear()
        self._ctx_cache[key] = kv
        kv["_keepalive"] = context   # keep the key's data_ptr from being recycled while cached
        return kv

    # ------------------------------------------------------------------ forward
    def _embed(self, pk, timesteps, fs, b, device):
        mc = self.model_channels
        t_emb = ops.to_f16(ops.timestep_embedding(timesteps, mc))
        h = ops.linear(t_emb, *pk["te"][0], out_f32=True)
        emb = ops.linear(ops.to_f16(ops.silu_f32(h)), *pk["te"][1], out_f32=True)          # [B, 4*mc] fp32
        if self.fs_condition:
            if fs is None:
                fs = torch.full((b,), self.default_fs, dtype=torch.long, device=device)
            f_emb = ops.to_f16(ops.timestep_embedding(fs, mc))
            h = ops.linear(f_emb, *pk["fe"][0], out_f32=True)
            emb = ops.linear(ops.to_f16(ops.silu_f32(h)), *pk["fe"][1], out_f32=True, rowadd=emb, rowadd_div=1)
        # every ResBlock starts its emb branch with SiLU (emb_layers.0): do it once
        return ops.to_f16(ops.silu_f32(emb))

    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, **kwargs):
        if self.use_hip_graph and features_adapter is None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            return self.forward_graphed(x, timesteps, context, fs, cfg_repeat=int(kwargs.get("cfg_repeat", 1) or 1))
        return self._forward(x, timesteps, context=context, features_adapter=features_adapter, fs=fs, **kwargs)

    def forward_graphed(self, x, timesteps, context, fs, cfg_repeat=1):
        """Replay the forward as one hipGraph (captured through torch.cuda.CUDAGraph: every libvcx call is stream-ordered,
        allocation-free and sync-free, so the ctypes launches are captured like any other kernel).  One graph per
        (shapes, conditioning tensor); inputs are copied into static buffers, the output buffer is reused by the next
        replay (the DDIM update consumes it first)."""
        parts = list(x) if isinstance(x, (list, tuple)) else [x]
        key = (tuple(tuple(p.shape) for p in parts), context.data_ptr(), context._version, tuple(context.shape),
               None if fs is None else tuple(fs.shape), cfg_repeat)
        ent = self._graphs.get(key)
        if ent is None:
            static = dict(parts=[p.detach().clone().float() for p in parts], t=timesteps.detach().clone(),
                          fs=None if fs is None else fs.detach().clone())
            with torch.no_grad():
                self._forward(static["parts"], static["t"], context=context, fs=static["fs"], cfg_repeat=cfg_repeat)   # warm-up: packs, caches K/V
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._forward(static["parts"], static["t"], context=context, fs=static["fs"], cfg_repeat=cfg_repeat)
            if len(self._graphs) >= 4:
                self._graphs.clear()
            # the captured launches bake in the addresses of the projected context K/V (allocated by the warm-up in the
            # ordinary caching allocator): the entry owns them, so clearing _ctx_cache cannot free what a replay reads
            ent = self._graphs[key] = (graph, static, out, context, self._context_kv(context, static["parts"][0].shape[2]))
        graph, static, out = ent[:3]
        for dst, src in zip(static["parts"], parts):
            dst.copy_(src)
        static["t"].copy_(timesteps)
        if fs is not None:
            static["fs"].copy_(fs)
        graph.replay()
        return out

    def _forward(self, x, timesteps, context=None, features_adapter=None, fs=None, cfg_repeat=1, **kwargs):
        """x [B, in_channels, T, h, w] fp32 (or a list of tensors to be concatenated on channels, which is how
        DiffusionWrapper avoids materialising torch.cat([x] + c_concat)); timesteps [B] int64; context [B, L, D];
        fs [B] int64.  Extra keywords (cfg_img, unconditional_conditioning_img_nonetext, ...) leak in from the
        sampler exactly as in the reference and are ignored.  Returns [B, out_channels, T, h, w] fp32.

        cfg_repeat = r > 1 (set by the sampler for classifier-free guidance): context is [r * B, L, D] - r conditionings of
        the SAME x / timesteps / fs (reference ddim.py:223-224 evaluates them one after the other on identical inputs).
        Every layer ahead of the first cross-attention (conv_in, init_attn, the first ResBlock with its temporal
        convolutions, and in the first SpatialTransformer GroupNorm, proj_in and the whole 9216-token self-attention) sees
        identical data in all r evaluations, so it runs once on B videos and the activations are replicated where the
        conditionings enter; the output [r * B, ...] is bit-identical to the forward of the r-fold replicated batch
        (tests/test_model_gpu.py::test_cfg_shared_prefix_is_bit_identical)."""
        ops.require_gpu()
        parts = list(x) if isinstance(x, (list, tuple)) else [x]
        b, _, t, hh, ww = parts[0].shape
        device = parts[0].device
        cin = sum(p.shape[1] for p in parts)
        if cin != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {cin}")
        r = int(cfg_repeat or 1)
        if context.shape[0] != b * r:
            raise ValueError(f"context batch {context.shape[0]} != {b} videos x cfg_repeat {r}")
        pk = self.packed()
        emb = self._embed(pk, timesteps, fs, b, device)
        ckv = self._context_kv(context, t)
        rbs = self.resblocks()
        if EMB_BATCHED and all(o % 4 == 0 for o, _ in pk["emb_off"].values()):
            emb_all = ops.linear(emb, pk["emb_w"], pk["emb_b"], out_f32=True)                 # [b, sum of widths] fp32: every block's emb_layers at once
            if r > 1:     # rows of the replicated batch: video v reads row v % b - the first b rows serve the layers ahead of the split
                n_all = emb_all.shape[1]
                emb_all = ops.repeat_rows(emb_all.view(torch.float16).view(b, 2 * n_all), r).view(torch.float32).view(r * b, n_all)
            for rb in rbs:
                o, n_e = pk["emb_off"][id(rb)]
                rb._emb_all = emb_all[:, o:o + n_e]
        try:
            return self._forward_body(parts, pk, emb, ckv, b, t, hh, ww, cin, r, device, features_adapter)
        finally:
            for rb in rbs:
                rb._emb_all = None

    def _forward_body(self, parts, pk, emb, ckv, b, t, hh, ww, cin, r, device, features_adapter):
        # batch currently flowing through the graph: b until the first SpatialTransformer replicated it, b * r afterwards
        cur_b = b

        def replicate(a):      # [cur_b * t, H, W, C] -> [r * cur_b * t, H, W, C]
            n_, H_, W_, C_ = a.shape
            return ops.repeat_rows(a.view(n_ * H_ * W_, C_), r).view(r * n_, H_, W_, C_)

        h = torch.empty((b, t, hh, ww, cin), dtype=torch.float16, device=device)
        off = 0
        for p in parts:
            ops.ncthw_to_nthwc(p.float(), h, c_off=off)
            off += p.shape[1]
        h = h.view(b * t, hh, ww, cin)

        lvl2 = GN_STATS_LEVEL >= 2

        def replicate_cs(c):   # the moments of a replicated tensor: its strips, r times
            if c is None:
                return None
            strips, C_, _ = c.shape
            return ops.repeat_rows(c.view(torch.float16).view(strips, C_ * 4), r).view(torch.float32).view(r * strips, C_, 2)

        hs = []                # (skip tensor, its column moments or None)
        adapter_idx = 0
        cs = None              # column moments of h (None: not known - the consumer makes its statistics pass)
        for i, module in enumerate(self.input_blocks):
            flow = Flow(colstats=cs, want=lvl2)
            if i == 0:
                h = ops.conv2d(h, pk["w_in"], pk["b_in"], kh=3, kw=3)
                flow.colstats = None
                if self.addition_attention:
                    h = self.init_attn(h, emb, context=ckv, batch_size=cur_b, flow=flow)
            elif cur_b != b * r and any(isinstance(layer, SpatialTransformer) for layer in module):
                h = module(h, emb, context=ckv, batch_size=cur_b, cfg_repeat=r, flow=flow)     # the conditionings enter here
                cur_b = b * r
                emb = ops.repeat_rows(emb, r)
                hs = [(replicate(a), replicate_cs(c)) for a, c in hs]                           # skips recorded so far
            else:
                h = module(h, emb, context=ckv, batch_size=cur_b, flow=flow)
            cs = flow.colstats
            if features_adapter is not None and (i + 1) % 3 == 0:      # plug-in adapter features (openaimodel3d.py:582-585)
                h = ops.add_nchw_(h.contiguous(), _adapter_rows(features_adapter[adapter_idx], h.shape[0]))
                adapter_idx, cs = adapter_idx + 1, None                 # the moments of h no longer describe it
            hs.append((h, cs))
        if features_adapter is not None and len(features_adapter) != adapter_idx:
            raise ValueError("Wrong features_adapter")                  # the reference's assertion, openaimodel3d.py:588
        if cur_b != b * r:     # a graph without attention in the input path: replicate ahead of the middle block
            h, cs, hs, emb, cur_b = replicate(h), replicate_cs(cs), [(replicate(a), replicate_cs(c)) for a, c in hs], ops.repeat_rows(emb, r), b * r
        b = cur_b
        if not lvl2:           # round-3 data path: materialised concat (two copies), statistics pass for every cross-module norm
            h = self.middle_block(h, emb, context=ckv, batch_size=b)
            for module in self.output_blocks:
                skip, _ = hs.pop()
                n, H, W, c1 = h.shape
                h = ops.concat_channels(h.view(n * H * W, c1), skip.view(n * H * W, skip.shape[-1])).view(n, H, W, -1)
                h = module(h, emb, context=ckv, batch_size=b)
            cs = None
        else:
            # Up path (openaimodel3d.py:595-597: h = torch.cat([h, hs.pop()], dim=1); h = module(h, ...)).  The block that PRODUCES h
            # writes it straight into the left columns of the next block's concatenated input - and its column moments into the left
            # columns of that tensor's moment buffer - so only the skip half is copied, and the GroupNorm in front of the next
            # ResBlock takes its statistics from the two producers' epilogues.
            def target_for(n_, H_, W_, c_left, consumer):
                skip, skip_cs = hs[-1]
                M_ = n_ * H_ * W_
                ok = skip_cs is not None and ops.colstats_ok(M_, H_ * W_, 64, c_left + skip.shape[-1])
                # split (the concat is read in place) only into a block that starts with a ResBlock - the layer that can read two halves
                split = CAT_SPLIT and SKIP_FOLD and c_left % 64 == 0 and skip.shape[-1] % 64 == 0 and isinstance(list(consumer)[0], ResBlock)
                return CatTarget(M_, c_left, skip.shape[-1], device, with_moments=ok, split=split)
            n, H, W, _ = h.shape
            tgt = target_for(n, H, W, _out_channels_of(self.middle_block), self.output_blocks[0])
            flow = Flow(colstats=cs, target=tgt)
            h = self.middle_block(h, emb, context=ckv, batch_size=b, flow=flow)
            for j, module in enumerate(self.output_blocks):
                skip, skip_cs = hs.pop()
                n, H, W, c1 = h.shape
                M, c2 = n * H * W, skip.shape[-1]
                if not tgt.split:
                    ops.copy2d(skip.view(M, c2), tgt.data[:, c1:], M, c2, c2, tgt.ld)
                if tgt.moments is not None:      # the skip's moments behind the producer's: 8 bytes per (strip, column), as fp16 quads
                    ops.copy2d(skip_cs.view(torch.float16).view(M // 64, c2 * 4), tgt.moments.view(torch.float16).view(M // 64, tgt.ld * 4)[:, c1 * 4:],
                               M // 64, c2 * 4, c2 * 4, tgt.ld * 4)
                hcat, cat_cs = tgt.data.view(n, H, W, tgt.data_ld), tgt.moments
                x2 = skip if tgt.split else None          # split: the ResBlock reads [h | skip] in place
                tgt = None
                if j + 1 < len(self.output_blocks):
                    last = list(module)[-1]
                    up = 2 if isinstance(last, Upsample) or (isinstance(last, ResBlock) and last.up) else 1
                    tgt = target_for(n, H * up, W * up, _out_channels_of(module), self.output_blocks[j + 1])
                flow = Flow(colstats=cat_cs, want=tgt is None, target=tgt)
                h = module(hcat, emb, context=ckv, batch_size=b, flow=flow, x2=x2)
            cs = flow.colstats
        n, H, W, c = h.shape
        stats = None if cs is None else ops.group_norm_stats_from_colstats(cs, n, H * W, c)
        a = ops.group_norm(h.view(n, H * W, c), *pk["g_out"], True, stats=stats)
        y = ops.conv2d(a.view(n, H, W, c), pk["w_out"], pk["b_out"], kh=3, kw=3, out_f32=True)   # [n, H, W, out] fp32
        return ops.nthwc_to_ncthw(y.view(b, t, H, W, self.out_channels))
