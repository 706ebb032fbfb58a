"""`image_proj_model` of the image conditioning path on libvcx (reference lvdm/modules/encoders/resampler.py).

Resampler (resampler.py:96-145): learned latent queries (num_queries per frame x video_length) cross-attend, `depth` times,
to the CLIP image tokens concatenated with the latents themselves (PerceiverAttention :51-93), each followed by a
LayerNorm -> Linear -> GELU -> Linear feed-forward (:27-34), then proj_out + LayerNorm.  It runs once per video, ahead of
the DDIM loop (utils/diffusion_utils.py:133-135).  Parameter names and shapes match the reference state dict
(`image_proj_model.*`), so checkpoints load strictly.

Kernel mapping: every Linear is a vcx_gemm_f16 call (q, k as [tokens, heads*64]; V is produced already transposed by
running the projection with the operand roles swapped, as the spatial transformer does), attention is the d = 64 flash
kernel over n1 + n2 keys, GELU is vcx_gelu_f16, LayerNorms are vcx_layernorm_f16.  Token streams are fp16; the result
is returned in fp32 like the reference module.
"""
import torch
import torch.nn as nn

from .... import ops
from ..attention import PackedModule, _f16, _f32


class ImageProjModel(PackedModule):
    """Reference resampler.py:9-23 (single-embedding projector; not used by the ViewCrafter YAMLs, kept for the API)."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def _pack(self):
        return dict(w=_f16(self.proj.weight), b=_f32(self.proj.bias), g=_f32(self.norm.weight), be=_f32(self.norm.bias))

    def forward(self, image_embeds):
        ops.require_gpu()
        pk = self.packed()
        x = image_embeds.reshape(-1, image_embeds.shape[-1])
        x = x.half().contiguous() if x.dtype != torch.float16 else x.contiguous()
        t = ops.linear(x, pk["w"], pk["b"]).view(-1, self.cross_attention_dim)
        out = ops.layer_norm(t, pk["g"], pk["be"], self.norm.eps)
        return ops.to_f32(out).view(-1, self.clip_extra_context_tokens, self.cross_attention_dim)


def FeedForward(dim, mult=4):
    """Reference resampler.py:27-34 (kept as an nn.Sequential so that the state-dict keys `layers.i.1.{0,1,3}` match)."""
    inner_dim = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim, bias=False), nn.GELU(), nn.Linear(inner_dim, dim, bias=False))


class PerceiverAttention(nn.Module):
    """Reference resampler.py:51-93; parameter container, driven by Resampler.forward."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError(f"libvcx attention kernels are built for head dim 64 (got {dim_head})")
        self.scale = dim_head ** -0.5
        self.dim_head, self.heads = dim_head, heads
        inner_dim = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)


class Resampler(PackedModule):
    """Reference resampler.py:96-145."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        self.num_queries = num_queries
        self.video_length = video_length
        if video_length is not None:
            num_queries = num_queries * video_length
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([
            nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult)])
            for _ in range(depth)])
        self.dim, self.heads, self.inner = dim, heads, dim_head * heads

    def _pack(self):
        inner = self.inner
        layers = []
        for attn, ff in self.layers:
            wkv = attn.to_kv.weight                      # [2*inner, dim]; .chunk(2, -1) of the output: K rows first, then V
            layers.append(dict(
                n1=(_f32(attn.norm1.weight), _f32(attn.norm1.bias), attn.norm1.eps),
                n2=(_f32(attn.norm2.weight), _f32(attn.norm2.bias), attn.norm2.eps),
                wq=_f16(attn.to_q.weight), wk=_f16(wkv[:inner]), wv=_f16(wkv[inner:]), wo=_f16(attn.to_out.weight),
                scale=attn.scale,
                nf=(_f32(ff[0].weight), _f32(ff[0].bias), ff[0].eps), w1=_f16(ff[1].weight), w2=_f16(ff[3].weight)))
        return dict(layers=layers, lat=_f16(self.latents[0]), win=_f16(self.proj_in.weight), bin=_f32(self.proj_in.bias),
                    wout=_f16(self.proj_out.weight), bout=_f32(self.proj_out.bias),
                    no=(_f32(self.norm_out.weight), _f32(self.norm_out.bias), self.norm_out.eps))

    @torch.no_grad()
    def forward(self, x):
        """x [B, n1, embedding_dim] (CLIP image tokens) -> [B, num_queries(*video_length), output_dim] fp32."""
        ops.require_gpu()
        pk = self.packed()
        B, n1, E = x.shape
        D, inner, heads = self.dim, self.inner, self.heads
        n2 = pk["lat"].shape[0]
        nk = n1 + n2
        nkp = (nk + 7) // 8 * 8                                       # flash kernel: kv rows per group padded to 8
        xin = x.reshape(B * n1, E)
        xin = xin.contiguous() if xin.dtype == torch.float16 else ops.to_f16(xin.float().contiguous())
        xp = ops.linear(xin, pk["win"], pk["bin"])                    # [B*n1, D]
        lat = pk["lat"].unsqueeze(0).expand(B, n2, D).contiguous().view(B * n2, D)
        kvin = torch.zeros((B, nkp, D), dtype=torch.float16, device=x.device)
        for L in pk["layers"]:
            xn = ops.layer_norm(xp, *L["n1"])
            ln = ops.layer_norm(lat, *L["n2"])
            kvin[:, :n1].copy_(xn.view(B, n1, D))                     # kv_input = cat(x, latents) along the tokens (:77)
            kvin[:, n1:nk].copy_(ln.view(B, n2, D))
            kv2 = kvin.view(B * nkp, D)
            q = ops.linear(ln, L["wq"])                               # [B*n2, inner]
            k = ops.linear(kv2, L["wk"])                              # [B*nkp, inner]
            vt = ops.gemm(L["wv"], kv2, M=inner, N=B * nkp, K=D, lda=D)   # [inner, B*nkp] = V^T
            o = torch.empty((B * n2, inner), dtype=torch.float16, device=x.device)
            ops.flash_attn(q, k, vt, o, n_groups=B, heads=heads, nq=n2, nk=nk, kv_rows=nkp, kv_div=1, ldq=inner, ldk=inner,
                           ldvt=B * nkp, ldo=inner, scale=L["scale"])
            lat = ops.linear(o, L["wo"], None, residual=lat)          # attn(x, latents) + latents (:140)
            h = ops.linear(ops.layer_norm(lat, *L["nf"]), L["w1"])
            lat = ops.linear(ops.gelu_(h), L["w2"], None, residual=lat)   # ff(latents) + latents (:141)
        out = ops.layer_norm(ops.linear(lat, pk["wout"], pk["bout"]), *pk["no"])
        return ops.to_f32(out).view(B, n2, -1)
