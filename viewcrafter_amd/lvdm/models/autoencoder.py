"""AutoencoderKL (reference lvdm/models/autoencoder.py:12-107), inference subset, on the gfx950 kernels."""
import torch
from torch import nn

from ... import ops
from ...packing import pack_conv, pad_cin
from ..distributions import DiagonalGaussianDistribution
from ..modules.attention import PackedModule, _f16, _f32
from ..modules.networks.ae_modules import Decoder, Encoder


class AutoencoderKL(PackedModule):
    """Constructor keywords follow configs/inference_pvd_*.yaml `first_stage_config.params`; `lossconfig`, `monitor`,
    `ckpt_path` etc. are accepted and ignored (training / Lightning only)."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        super().__init__()
        assert ddconfig["double_z"]
        if ddconfig["z_channels"] > 8 or embed_dim > 8 or ddconfig["in_channels"] > 8:
            raise NotImplementedError("latent / image channel counts above 8 are not supported")
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = nn.Identity()
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim, self.input_dim = embed_dim, input_dim
        self.z_channels = ddconfig["z_channels"]
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def _pack(self):
        pq, q = self.post_quant_conv, self.quant_conv
        return dict(pq=(_f16(pack_conv(pad_cin(pq.weight.detach(), 8))), _f32(pq.bias)),
                    q=(_f16(pack_conv(q.weight.detach())), _f32(q.bias)))

    def decode(self, z, **kwargs):
        """z [N, z_channels, h, w] fp32 -> [N, 3, 8h, 8w] fp32 (reference autoencoder.py:104-107)."""
        ops.require_gpu()
        pk = self.packed()
        n, c, h, w = z.shape
        z8 = torch.zeros((n, 1, h, w, 8), dtype=torch.float16, device=z.device)
        ops.ncthw_to_nthwc(z.float().reshape(n, c, 1, h, w), z8)
        # post_quant_conv (1x1, embed_dim -> z_channels) written into a zero 8-channel buffer for the 3x3 conv_in
        zq = torch.zeros((n * h * w, 8), dtype=torch.float16, device=z.device)
        ops.gemm(z8.view(n * h * w, 8), pk["pq"][0], M=n * h * w, N=self.z_channels, K=8, lda=8, bias=pk["pq"][1], out=zq, ldc=8)
        dec = self.decoder(zq.view(n, h, w, 8))                                   # [n, H, W, 3] fp32
        H, W = dec.shape[1], dec.shape[2]
        return ops.nthwc_to_ncthw(dec.view(n, 1, H, W, dec.shape[-1])).reshape(n, dec.shape[-1], H, W)

    def encode(self, x, **kwargs):
        """x [N, 3, H, W] fp32 in [-1, 1] -> DiagonalGaussianDistribution (reference autoencoder.py:97-102)."""
        ops.require_gpu()
        pk = self.packed()
        n, c, H, W = x.shape
        x8 = torch.zeros((n, 1, H, W, 8), dtype=torch.float16, device=x.device)
        ops.ncthw_to_nthwc(x.float().reshape(n, c, 1, H, W), x8)
        hq = self.encoder(x8.view(n, H, W, 8))                                    # [n, h, w, 2z] fp16
        _, h, w, c2 = hq.shape
        mom = ops.gemm(hq.view(n * h * w, c2), pk["q"][0], M=n * h * w, N=2 * self.embed_dim, K=c2, lda=c2, bias=pk["q"][1],
                       out_f32=True)
        moments = ops.nthwc_to_ncthw(mom.view(n, 1, h, w, 2 * self.embed_dim)).reshape(n, 2 * self.embed_dim, h, w)
        return DiagonalGaussianDistribution(moments)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
