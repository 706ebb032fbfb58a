"""Moments and concat targets that travel with the activations through the UNet graph (round 4).

GroupNorm statistics from the PRODUCING layer's epilogue (column moments per 64-row strip, VCX_GEMM_COLSTATS) instead of a pass over
the tensor.  VCX_GN_EPILOGUE_STATS: 0 = statistics pass everywhere; 1 = inside ResBlock / TemporalConvBlock only (round 3: 91 of the
166 norms of a step); 2 (default, round 4) = also across module boundaries - the moments ride along with the activation from block
to block (a transformer's proj_out + residual, the down / up-sampling convolutions), skips keep theirs until the up path
concatenates them, and the up path's producers write data AND moments straight into the concatenated buffer (no copy of the left
half, openaimodel3d.py:596).  A/B runs: tools/gnstats_ab.sh.
"""
import os

import torch

from ... import ops

GN_STATS_LEVEL = int(os.environ.get("VCX_GN_EPILOGUE_STATS", "2"))
GN_EPILOGUE_STATS = GN_STATS_LEVEL >= 1


# Round 6: the channel concat of the up path (reference openaimodel3d.py:596) is not materialised where the consuming ResBlock can read
# its two halves in place - the in_layers norm through vcx_groupnorm_apply2_f16, the 1x1 skip convolution as a K tail of the block's
# second 3x3 convolution (SKIP_FOLD in openaimodel3d.py) - so the copy of the skip tensor behind the producer's columns disappears; the
# MOMENTS of the two halves still meet in one buffer (8 bytes per strip and column).  VCX_CAT_SPLIT=0: the round-4 concat buffer (A/B runs).
CAT_SPLIT = os.environ.get("VCX_CAT_SPLIT", "1") != "0"


class CatTarget:
    """Where the last layer of a block writes when its output is the LEFT part of the next block's channel concat: `data`
    [M, data_ld] fp16 (columns [0, c_left) are this block's output; data_ld = ld = c_left + c_right with the skip tensor copied
    behind them, or - `split` - data_ld = c_left: a plain tensor, the skip stays where it is) and, when the shapes allow it,
    `moments` [M / 64, ld, 2] fp32 for the column moments of those columns."""
    __slots__ = ("data", "moments", "ld", "c_left", "data_ld", "split")

    def __init__(self, M, c_left, c_right, device, with_moments, split=False):
        self.c_left = c_left
        self.ld = c_left + c_right
        self.split = bool(split and with_moments)
        self.data_ld = c_left if self.split else self.ld
        self.data = torch.empty((M, self.data_ld), dtype=torch.float16, device=device)
        self.moments = torch.empty((M // 64, self.ld, 2), dtype=torch.float32, device=device) if with_moments else None

    def kwargs(self, k, in_rows):
        """keywords for the producing ops.gemm / conv2d / linear call.  k = its K (linear) or cin (convolution), in_rows = rows of
        its input: a producer the DMA GEMM kernel cannot take (K % 64 != 0, >= 4 GiB operand) writes the data only - the moment
        buffer is dropped and the consumer makes its statistics pass."""
        if self.moments is not None and (k % 64 != 0 or 2 * in_rows * k >= 0xFFFF0000 or ops.tune_get("GEMM_DMA") == 0):
            self.moments = None
        kw = dict(out=self.data, ldc=self.data_ld)
        if self.moments is not None:
            kw.update(colstats=self.moments, colstats_ld=self.ld, colstats_col=0)
        return kw


class Flow:
    """State handed from block to block by UNetModel._forward: `colstats` = column moments of the tensor going INTO the block
    (None: unknown) and, after the call, of the tensor coming out; `want` = the caller can use the output's moments; `target` =
    the CatTarget the block's last layer writes to (None: an ordinary tensor)."""
    __slots__ = ("colstats", "want", "target")

    def __init__(self, colstats=None, want=False, target=None):
        self.colstats, self.want, self.target = colstats, want, target


def out_kwargs(target, want, M, pixels, k, cout, device, in_rows=None):
    """(kwargs for the producing ops call, the moments buffer its output's GroupNorm can use or None)"""
    if target is not None:
        kw = target.kwargs(k, in_rows if in_rows is not None else M)
        return kw, target.moments
    if want and GN_STATS_LEVEL >= 2 and ops.colstats_ok(M, pixels, k, cout, in_rows):
        cs = ops.colstats_buffer(M, cout, device)
        return dict(colstats=cs), cs
    return {}, None


