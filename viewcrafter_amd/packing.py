"""One-time repacking of reference-layout parameters into the layouts the gfx950 kernels read.

Reference layouts (binding because checkpoints are loaded with strict=True):
conv weights [Cout, Cin, kh, kw] / [Cout, Cin, 3, 1, 1] / [Cout, Cin, 1], linear [out, in].
Kernel layouts: fp16 [N_out][K] with K ordered (tap, cin); GEGLU rows interleaved in blocks of 32.
"""
import torch


def conv_slab_major(cin, taps):
    """Convolutions with cin % 64 == 0 and more than one tap keep their K axis in slabs of 64 input channels with the taps
    inside (VCX_GEMM_CONV_SLABK, include/vcx.h): the 9 (or 3) tap re-reads of a tile's input are then consecutive K-steps
    and hit L2 instead of going back to the memory side.  pack_conv and ops.conv2d / temporal_conv3 share this predicate."""
    return cin % 64 == 0 and taps > 1


def pack_conv(w):
    """[Cout, Cin, *kernel] -> [Cout, taps*Cin]: K ordered (tap, c) - the im2col order of the channels-last gather in
    csrc/gemm.hip - or (c / 64, tap, c % 64) where conv_slab_major() says so."""
    cout, cin = w.shape[0], w.shape[1]
    wk = w.reshape(cout, cin, -1)            # [Cout, Cin, taps]
    taps = wk.shape[2]
    if conv_slab_major(cin, taps):
        return wk.view(cout, cin // 64, 64, taps).permute(0, 1, 3, 2).reshape(cout, -1).contiguous()
    return wk.permute(0, 2, 1).reshape(cout, -1).contiguous()


def pad_cin(w, cin_to):
    """Zero-pad the input-channel dim of a conv weight (e.g. the VAE's 4-channel conv_in to 8)."""
    if w.shape[1] == cin_to:
        return w
    out = w.new_zeros((w.shape[0], cin_to) + tuple(w.shape[2:]))
    out[:, :w.shape[1]] = w
    return out


def pack_geglu(w, b):
    """GEGLU.proj (lvdm/modules/attention.py:415-422) has rows [0, D) = x and [D, 2D) = gate.  Interleave them in
    blocks of 32 so that x_j and gate_j land in the same lane of one wave's accumulators (VCX_GEMM_GEGLU)."""
    two_d = w.shape[0]
    d = two_d // 2
    assert d % 32 == 0, "GEGLU inner dim must be a multiple of 32"
    idx = torch.arange(d, device=w.device).view(-1, 32)
    perm = torch.cat([idx, idx + d], dim=1).reshape(-1)     # [x0..x31, g0..g31, x32.., ...]
    return w[perm].contiguous(), (b[perm].contiguous() if b is not None else None)


def fold_layernorm(w, gamma, beta, bias=None):
    """nn.LayerNorm(gamma, beta) followed by nn.Linear(w [N, K], bias) as ONE projection of the un-normalised rows
    (VCX_GEMM_LNFOLD, include/vcx.h):  LN(x) w^T + bias = rstd (x w'^T - mean colsum) + bias'  with
    w' = gamma o w rounded to fp16, colsum = the fp32 row sums of that ROUNDED w' (so that x w'^T - mean colsum is
    sum_k (x_k - mean) w'_k exactly, whatever the common offset of the row), bias' = bias + w beta in fp32.
    Returns (w' fp16, colsum fp32, bias' fp32)."""
    w32, g32, b32 = w.detach().float(), gamma.detach().float(), beta.detach().float()
    wf = (w32 * g32[None, :]).to(torch.float16)
    colsum = wf.double().sum(dim=1).float()        # exact sum of the fp16 values, rounded once
    bias_f = w32 @ b32
    if bias is not None:
        bias_f = bias_f + bias.detach().float()
    return wf.contiguous(), colsum.contiguous(), bias_f.contiguous()
