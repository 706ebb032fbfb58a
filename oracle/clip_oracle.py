"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32) oracle of the two OpenCLIP condition encoders of the ViewCrafter path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

What it restates, and how it is pinned:

* `clip_text_forward` / `clip_image_forward` restate the *reference's own* glue around the CLIP towers:
  FrozenOpenCLIPEmbedder.encode_with_transformer / text_transformer_forward (lvdm/modules/encoders/condition.py:218-237:
  token + positional embedding, the first `layers - layer_idx` residual blocks under the causal mask, ln_final) and
  FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer (condition.py:347-378: preprocess, patch conv, class token,
  positional embedding, ln_pre, all blocks, NO ln_post / proj).  Pinned: tests/golden/gen_golden.py runs that reference
  code itself (imported from /root/reference) and the outputs are committed as tests/golden/clip_*_tiny.npz.
* The towers themselves live in two third-party packages that are absent from /root/reference and from this image:
  `open_clip` (requirements.txt: `open_clip_torch`, unpinned; arch "ViT-H-14", pretrained "laion2b_s32b_b79k",
  condition.py:183,290) and `kornia` (requirements.txt:8, unpinned; kornia.geometry.resize + kornia.enhance.normalize,
  condition.py:325-332).  `StandInCLIP` and `kornia_resize` / `kornia_normalize` below restate their published behaviour
  (open_clip's CLIP / VisionTransformer / ResidualAttentionBlock module tree and parameter names, built on
  torch.nn.MultiheadAttention exactly as open_clip does; kornia's "gaussian blur with sigma = (factor - 1) / 2, kernel
  2 * 2 * sigma made odd, reflect border, then F.interpolate(bicubic, align_corners=True)") and serve as the stand-ins the
  golden generator installs under those module names.  No golden vector of the real packages can be produced here; the tower
  internals (clip_text_forward, clip_image_tower) are pinned instead to an INDEPENDENT third-party implementation of the same
  architecture that is in the image: transformers' CLIPTextModel / CLIPVisionModel (oracle/clip_hf.py,
  tests/test_oracle_golden.py::test_clip_towers_match_transformers).  The kornia resize restatement stays PARITY UNPINNED.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

# open_clip model_configs/ViT-H-14.json (published architecture of the tower the YAMLs name) + a tiny one for tests
CLIP_CONFIGS = {
    "ViT-H-14": dict(embed_dim=1024,
                     vision=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14, mlp_ratio=4.0),
                     text=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0)),
    "vcx-tiny-test": dict(embed_dim=64,
                          vision=dict(image_size=224, layers=3, width=160, head_width=80, patch_size=56, mlp_ratio=2.0),
                          text=dict(context_length=77, vocab_size=49408, width=128, heads=2, layers=3, mlp_ratio=2.0)),
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------------
# stand-in for open_clip's module tree (names and shapes as in an open_clip CLIP state dict)
# ------------------------------------------------------------------------------------------------
class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ls_1 = nn.Identity()
        self.ln_2 = nn.LayerNorm(d_model)
        mlp_width = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, mlp_width)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(mlp_width, d_model))]))
        self.ls_2 = nn.Identity()

    def forward(self, q_x, attn_mask=None):
        y = self.ln_1(q_x)
        x = q_x + self.ls_1(self.attn(y, y, y, need_weights=False, attn_mask=attn_mask)[0])
        return x + self.ls_2(self.mlp(self.ln_2(x)))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio=4.0):
        super().__init__()
        self.width, self.layers = width, layers
        self.grad_checkpointing = False
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for r in self.resblocks:
            x = r(x, attn_mask=attn_mask)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim):
        super().__init__()
        self.input_patchnorm = False
        self.image_size, self.patch_size = (image_size, image_size), (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.patch_dropout = nn.Identity()
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class StandInCLIP(nn.Module):
    def __init__(self, arch="ViT-H-14"):
        super().__init__()
        cfg = CLIP_CONFIGS[arch]
        v, t = cfg["vision"], cfg["text"]
        self.visual = VisionTransformer(v["image_size"], v["patch_size"], v["width"], v["layers"], v["width"] // v["head_width"],
                                        v["mlp_ratio"], cfg["embed_dim"])
        self.transformer = Transformer(t["width"], t["layers"], t["heads"], t["mlp_ratio"])
        self.context_length, self.vocab_size = t["context_length"], t["vocab_size"]
        self.token_embedding = nn.Embedding(t["vocab_size"], t["width"])
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(t["context_length"], t["width"]))
        self.ln_final = nn.LayerNorm(t["width"])
        self.text_projection = nn.Parameter(t["width"] ** -0.5 * torch.randn(t["width"], cfg["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        mask = torch.full((t["context_length"], t["context_length"]), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)
        nn.init.normal_(self.token_embedding.weight, std=0.02)


def tokenize_empty(n, context_length=77):
    """open_clip.tokenize([""] * n): <start_of_text> <end_of_text> then zeros (the only prompt the stand-in can tokenize)."""
    t = torch.zeros(n, context_length, dtype=torch.long)
    t[:, 0], t[:, 1] = 49406, 49407
    return t


# ------------------------------------------------------------------------------------------------
# stand-in for kornia.geometry.resize(..., interpolation='bicubic', align_corners=True, antialias=True) / enhance.normalize
# ------------------------------------------------------------------------------------------------
def _gauss1d(ks, sigma, dtype, device):
    x = torch.arange(ks, dtype=dtype, device=device) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2) / (2.0 * sigma * sigma))
    return g / g.sum()


def kornia_resize(x, size, antialias=True):
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    if antialias and max(factors) > 1:
        sig = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        c = x.shape[1]
        ky = _gauss1d(ks[0], sig[0], x.dtype, x.device).view(1, 1, -1, 1).expand(c, 1, -1, 1)
        kx = _gauss1d(ks[1], sig[1], x.dtype, x.device).view(1, 1, 1, -1).expand(c, 1, 1, -1)
        x = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        x = F.conv2d(F.conv2d(x, kx, groups=c), ky, groups=c)
    return F.interpolate(x, size=size, mode="bicubic", align_corners=True)


def kornia_normalize(x, mean, std):
    mean = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    std = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    return (x - mean) / std


# ------------------------------------------------------------------------------------------------
# restatement of the reference glue on a plain state dict {name: tensor} (keys as `model.<...>` of the embedder modules)
# ------------------------------------------------------------------------------------------------
def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _block(x, sd, p, heads, mask=None):
    """open_clip ResidualAttentionBlock on x [B, L, W] (batch-first here; the math is layout independent)."""
    B, L, W = x.shape
    d = W // heads
    y = _ln(x, sd, p + ".ln_1")
    qkv = y @ sd[p + ".attn.in_proj_weight"].t() + sd[p + ".attn.in_proj_bias"]
    q, k, v = [t.view(B, L, heads, d).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * d ** -0.5
    if mask is not None:
        s = s + mask
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, W)
    x = x + o @ sd[p + ".attn.out_proj.weight"].t() + sd[p + ".attn.out_proj.bias"]
    y = _ln(x, sd, p + ".ln_2")
    h = F.gelu(y @ sd[p + ".mlp.c_fc.weight"].t() + sd[p + ".mlp.c_fc.bias"])
    return x + h @ sd[p + ".mlp.c_proj.weight"].t() + sd[p + ".mlp.c_proj.bias"]


def clip_text_forward(sd, tokens, heads, layers, layer_idx=1):
    """condition.py:218-237 with layer='penultimate' (layer_idx 1): [B, 77] int64 -> [B, 77, width]."""
    sd = {k: v.float() for k, v in sd.items()}
    x = sd["model.token_embedding.weight"][tokens] + sd["model.positional_embedding"]
    L = x.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(layers - layer_idx):
        x = _block(x, sd, f"model.transformer.resblocks.{i}", heads, mask)
    return _ln(x, sd, "model.ln_final")


def clip_image_forward(sd, img, heads, layers, patch, antialias=True):
    """condition.py:325-378: img [B, 3, H, W] in [-1, 1] -> [B, grid^2 + 1, width] (tokens after the last block, no ln_post)."""
    x = kornia_resize(img.float(), (224, 224), antialias)
    x = kornia_normalize((x + 1.0) / 2.0, CLIP_MEAN, CLIP_STD)
    return clip_image_tower(sd, x, heads, layers, patch)


def clip_image_tower(sd, x, heads, layers, patch):
    """The vision tower proper on pre-processed pixels [B, 3, S, S] (open_clip VisionTransformer.forward up to and including
    the last residual block; checked against transformers' independent CLIPVisionModel in oracle/clip_hf.py)."""
    sd = {k: v.float() for k, v in sd.items()}
    x = F.conv2d(x.float(), sd["model.visual.conv1.weight"], stride=patch)
    x = x.flatten(2).transpose(1, 2)                                   # [B, grid^2, width]
    cls = sd["model.visual.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], 1) + sd["model.visual.positional_embedding"]
    x = _ln(x, sd, "model.visual.ln_pre")
    for i in range(layers):
        x = _block(x, sd, f"model.visual.transformer.resblocks.{i}", heads)
    return x
