"""AutoencoderKL encoder/decoder (reference lvdm/modules/networks/ae_modules.py:364-578) on the gfx950 kernels.

Same sub-module names/shapes as the reference (`conv_in`, `mid.block_1`, `mid.attn_1`, `up.{i}.block.{j}`,
`up.{i}.upsample.conv`, `norm_out`, `conv_out`, ...).  Frames are processed channels-last fp16; convolutions are the
implicit-GEMM kernel; the single-head d=C attention of the mid block materialises its [N, N] score matrix per frame
with two GEMMs and a row-softmax kernel (d = 512 does not fit the d = 64 flash tiling).
"""
import os

import torch
from torch import nn

from .... import ops
from ....packing import pack_conv, pad_cin
from ..attention import PackedModule, _f16, _f32


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _pack3(conv, cin_to=None):
    w = conv.weight.detach()
    if cin_to is not None:
        w = pad_cin(w, cin_to)
    return _f16(pack_conv(w)), _f32(conv.bias)


def _gn(x, gn, silu):
    n, H, W, C = x.shape
    return ops.group_norm(x.view(n, H * W, C), gn[0], gn[1], gn[2], silu).view(n, H, W, C)


class ResnetBlock(PackedModule):
    """Reference ae_modules.py:151-210 with temb_channels=0: GN -> swish -> conv3x3 -> GN -> swish -> conv3x3 (+ 1x1
    `nin_shortcut` when channels change)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("VAE ResnetBlock variant not used by AutoencoderKL")
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(self.out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(self.out_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, self.out_channels, kernel_size=1, stride=1, padding=0)

    def _pack(self):
        pk = dict(g1=(_f32(self.norm1.weight), _f32(self.norm1.bias), self.norm1.eps), c1=_pack3(self.conv1),
                  g2=(_f32(self.norm2.weight), _f32(self.norm2.bias), self.norm2.eps), c2=_pack3(self.conv2))
        if self.in_channels != self.out_channels:
            pk["cs"] = _pack3(self.nin_shortcut)
        return pk

    def forward(self, x, temb=None):
        pk = self.packed()
        n, H, W, _ = x.shape
        h = ops.conv2d(_gn(x, pk["g1"], True), *pk["c1"], kh=3, kw=3)
        skip = ops.conv2d(x, *pk["cs"], kh=1, kw=1) if "cs" in pk else x
        return ops.conv2d(_gn(h, pk["g2"], True), *pk["c2"], kh=3, kw=3,
                          residual=skip.reshape(n * H * W, self.out_channels))


# the d = 512 flash kernel for AttnBlock (every AutoencoderKL config has 512 channels there); VCX_VAE_FUSED_ATTN=0: the GEMM -> row softmax
# -> GEMM sequence of rounds 1-4, frame by frame (A/B runs)
FUSED_ATTN = os.environ.get("VCX_VAE_FUSED_ATTN", "1") != "0"


class AttnBlock(PackedModule):
    """Reference ae_modules.py:26-78: GN -> q,k,v 1x1 -> softmax(q k^T C^-1/2) v -> 1x1 -> + x (one head, d = C)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def _pack(self):
        return dict(g=(_f32(self.norm.weight), _f32(self.norm.bias), self.norm.eps), q=_pack3(self.q), k=_pack3(self.k),
                    v=_pack3(self.v), o=_pack3(self.proj_out))

    def forward(self, x):
        pk = self.packed()
        n, H, W, C = x.shape
        N = H * W
        # The GEMMs address a frame's tokens at 16-byte granularity.  72x128 / 40x64 latents give N % 8 == 0; for any other
        # --height / --width each frame's token rows are padded with zero rows to the next multiple of 8 for the length of this
        # block, exactly as SpatialTransformer does for the UNet (round 4: this used to raise, so a free size that passed the UNet
        # died in the decoder).  Pad KEYS never enter: the score GEMM writes the N real columns of a zero-initialised [Np, Np]
        # buffer and the softmax runs over those N; pad QUERY rows compute garbage that is dropped on the way out.
        Np = (N + 7) // 8 * 8
        hn = _gn(x, pk["g"], False).view(n * N, C)
        xin = x.reshape(n * N, C)
        if Np != N:
            def pad_frames(src):
                dst = torch.zeros((n * Np, C), dtype=torch.float16, device=x.device)
                ops.copy2d(src, dst, n, N * C, N * C, Np * C)           # one "row" per frame
                return dst
            hn, xin = pad_frames(hn), pad_frames(xin)
        q = ops.linear(hn, *pk["q"])
        k = ops.linear(hn, *pk["k"])
        vt = ops.gemm(pk["v"][0], hn, M=C, N=n * Np, K=C, lda=C, bias=pk["v"][1], bias_m=True)      # [C, n*Np]
        o = torch.empty((n * Np, C), dtype=torch.float16, device=x.device)
        scale = float(int(C) ** (-0.5))
        if C == 512 and FUSED_ATTN:
            # one launch over all frames, the N x N score matrix (170 MB per frame at 72x128) never written: flash_d512_kernel
            ops.flash_attn_d512(q, k, vt, o, n_groups=n, nq=Np, nk=N, kv_rows=Np, ldq=C, ldk=C, ldvt=n * Np, ldo=C, scale=scale)
            n_loop = 0
        else:
            n_loop = n
            s = torch.zeros((Np, Np), dtype=torch.float16, device=x.device)
        for i in range(n_loop):   # other widths: one frame at a time through a materialised S (N x N)
            qi, ki = q[i * Np:(i + 1) * Np], k[i * Np:(i + 1) * Np]
            ops.gemm(qi, ki, M=Np, N=N, K=C, lda=C, out=s, ldc=Np, alpha=scale)
            ops.softmax_rows_(s, N)
            ops.gemm(s, vt[:, i * Np:], M=Np, N=C, K=Np, lda=Np, ldw=n * Np, out=o[i * Np:(i + 1) * Np], ldc=C)
        out = ops.linear(o, *pk["o"], residual=xin)
        if Np != N:
            unpadded = torch.empty((n * N, C), dtype=torch.float16, device=x.device)
            ops.copy2d(out, unpadded, n, N * C, Np * C, N * C)
            out = unpadded
        return out.view(n, H, W, C)


def make_attn(in_channels, attn_type="vanilla"):
    if attn_type == "vanilla":
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError(f"attn_type {attn_type} is not used by AutoencoderKL")


class Downsample(PackedModule):
    """Reference ae_modules.py:90-109: pad (0,1,0,1) then Conv2d 3x3 stride 2 (no padding)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=True in AutoencoderKL")
        self.with_conv, self.in_channels = with_conv, in_channels
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def _pack(self):
        return _pack3(self.conv)

    def forward(self, x):
        n, H, W, _ = x.shape
        return ops.conv2d(x, *self.packed(), kh=3, kw=3, stride=2, pad_h=0, pad_w=0, out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))


class Upsample(PackedModule):
    """Reference ae_modules.py:111-127: nearest 2x (fused into the gather) then Conv2d 3x3."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=True in AutoencoderKL")
        self.with_conv, self.in_channels = with_conv, in_channels
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        return _pack3(self.conv)

    def forward(self, x):
        return ops.conv2d(x, *self.packed(), kh=3, kw=3, ups=1)


class Encoder(PackedModule):
    """Reference ae_modules.py:364-463."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        return dict(cin=_pack3(self.conv_in, cin_to=8), g=(_f32(self.norm_out.weight), _f32(self.norm_out.bias), self.norm_out.eps),
                    cout=_pack3(self.conv_out))

    def forward(self, x):
        """x [n, H, W, 8] fp16 channels-last (3 image channels zero-padded to 8).  Returns [n, h, w, 2z] fp16."""
        pk = self.packed()
        h = ops.conv2d(x, *pk["cin"], kh=3, kw=3)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return ops.conv2d(_gn(h, pk["g"], True), *pk["cout"], kh=3, kw=3)


class Decoder(PackedModule):
    """Reference ae_modules.py:466-578."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("Decoder variant not used by AutoencoderKL")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        return dict(cin=_pack3(self.conv_in, cin_to=8), g=(_f32(self.norm_out.weight), _f32(self.norm_out.bias), self.norm_out.eps),
                    cout=_pack3(self.conv_out))

    def forward(self, z):
        """z [n, h, w, 8] fp16 channels-last (z_channels zero-padded to 8).  Returns [n, H, W, out_ch] fp32."""
        pk = self.packed()
        h = ops.conv2d(z, *pk["cin"], kh=3, kw=3)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return ops.conv2d(_gn(h, pk["g"], True), *pk["cout"], kh=3, kw=3, out_f32=True)
