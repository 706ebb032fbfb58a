"""The two OpenCLIP condition encoders of the ViewCrafter YAMLs on libvcx (reference lvdm/modules/encoders/condition.py).

* `FrozenOpenCLIPEmbedder` (condition.py:174-240, `cond_stage_config`, layer "penultimate"): token + positional embedding,
  the first `layers - 1` residual attention blocks of the text tower under the causal mask, `ln_final` -> [B, 77, 1024].
* `FrozenOpenCLIPImageEmbedderV2` (condition.py:302-378, `img_cond_stage_config`): kornia-style bicubic resize to 224x224
  (+ anti-alias blur when shrinking), CLIP mean/std, 14x14 patch embedding, class token, positional embedding, `ln_pre`,
  all blocks of the vision tower, no `ln_post` / projection -> [B, 257, 1280] (consumed by the Resampler).

Both run once per video, ahead of the DDIM loop (utils/diffusion_utils.py:121-135).  In the reference the towers come from
the third-party `open_clip` package (arch "ViT-H-14"); here `self.model` is a parameter container with open_clip's module
tree and parameter names (`model.visual.transformer.resblocks.N.attn.in_proj_weight`, ...), so the `cond_stage_model.*` /
`embedder.*` entries of a ViewCrafter checkpoint load strictly, and the arithmetic is libvcx: LayerNorm, GEMM (+bias,
+residual), exact-erf GELU; attention per (image, head) as S = alpha Q K^T + mask (GEMM epilogue adds the fp16 mask that
carries the causal triangle and disables the padded key columns), row softmax, P V - the head dim of the vision tower is
80, so the d = 64 flash kernel does not apply, and the 77 / 257-token problems are tiny.  Token rows are padded to a
multiple of 8 per batch element to keep every operand 16-byte aligned.

Tokenisation needs CLIP's BPE vocabulary, which is data, not code: `tokenize` uses `open_clip` when it is importable, a
vocabulary file named by $VCX_CLIP_BPE (open_clip's bpe_simple_vocab_16e6.txt.gz) otherwise, and without either accepts
only the empty prompt (<start_of_text><end_of_text>, what image_guided_synthesis feeds for text_input=False) or ready-made
token tensors.
"""
import gzip
import html
import os
from functools import lru_cache

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ..attention import PackedModule, _f16, _f32

# open_clip model_configs/ViT-H-14.json; further entries may be registered (tests use a tiny one)
CLIP_CONFIGS = {
    "ViT-H-14": dict(embed_dim=1024,
                     vision=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14, mlp_ratio=4.0),
                     text=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0)),
}
MASKED = -60000.0          # additive fp16 mask value: exp(MASKED - max) underflows to exactly 0


# ------------------------------------------------------------------------------------------------
# parameter containers with open_clip's names
# ------------------------------------------------------------------------------------------------
class _Attn(nn.Module):
    """nn.MultiheadAttention's parameters (in_proj_weight / in_proj_bias / out_proj.{weight,bias})."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * width ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class _Block(nn.Module):
    def __init__(self, width, mlp_ratio):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _Attn(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.ModuleDict(dict(c_fc=nn.Linear(width, int(width * mlp_ratio)), c_proj=nn.Linear(int(width * mlp_ratio), width)))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([_Block(width, mlp_ratio) for _ in range(layers)])


class _Visual(nn.Module):
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim):
        super().__init__()
        self.image_size, self.patch_size = image_size, patch_size
        self.grid = image_size // patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class _CLIP(nn.Module):
    """What is left of an open_clip CLIP after the embedder's `del model.visual` / `del model.transformer`."""

    def __init__(self, arch, keep):
        super().__init__()
        if isinstance(arch, dict):          # an open_clip model_config given inline (YAML `params: {arch: {...}}`; the tests' tiny towers)
            cfg = arch
        elif arch in CLIP_CONFIGS:
            cfg = CLIP_CONFIGS[arch]
        else:
            raise KeyError(f"unknown CLIP arch {arch!r}; known: {sorted(CLIP_CONFIGS)}")
        v, t = cfg["vision"], cfg["text"]
        if keep == "visual":
            self.visual = _Visual(v["image_size"], v["patch_size"], v["width"], v["layers"], v["width"] // v["head_width"],
                                  v["mlp_ratio"], cfg["embed_dim"])
        else:
            self.transformer = _Transformer(t["width"], t["layers"], t["heads"], t["mlp_ratio"])
        self.context_length, self.vocab_size = t["context_length"], t["vocab_size"]
        self.token_embedding = nn.Embedding(t["vocab_size"], t["width"])
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(t["context_length"], t["width"]))
        self.ln_final = nn.LayerNorm(t["width"])
        self.text_projection = nn.Parameter(t["width"] ** -0.5 * torch.randn(t["width"], cfg["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)


def _pack_blocks(tr, n_layers):
    out = []
    for blk in tr.resblocks[:n_layers]:
        W = tr.width
        w_in, b_in = blk.attn.in_proj_weight, blk.attn.in_proj_bias
        out.append(dict(
            ln1=(_f32(blk.ln_1.weight), _f32(blk.ln_1.bias), blk.ln_1.eps), ln2=(_f32(blk.ln_2.weight), _f32(blk.ln_2.bias), blk.ln_2.eps),
            wqk=_f16(w_in[:2 * W]), bqk=_f32(b_in[:2 * W]), wv=_f16(w_in[2 * W:]), bv=_f32(b_in[2 * W:]),
            wo=_f16(blk.attn.out_proj.weight), bo=_f32(blk.attn.out_proj.bias),
            w1=_f16(blk.mlp["c_fc"].weight), b1=_f32(blk.mlp["c_fc"].bias),
            w2=_f16(blk.mlp["c_proj"].weight), b2=_f32(blk.mlp["c_proj"].bias)))
    return out


def _run_blocks(x, layers, B, L, Lp, W, heads, mask):
    """x [B*Lp, W] fp16 token rows (L valid + padding per batch element), open_clip ResidualAttentionBlock x len(layers):
    x += out_proj(MHA(ln_1 x)); x += c_proj(gelu(c_fc(ln_2 x)))."""
    d = W // heads
    T = B * Lp
    scale = d ** -0.5
    s = torch.empty((L, Lp), dtype=torch.float16, device=x.device)
    o = torch.zeros((T, W), dtype=torch.float16, device=x.device)          # padded rows stay zero
    for P in layers:
        y = ops.layer_norm(x, *P["ln1"])
        qk = ops.linear(y, P["wqk"], P["bqk"])                               # [T, 2W]: q | k
        vt = ops.gemm(P["wv"], y, M=W, N=T, K=W, lda=W, bias=P["bv"], bias_m=True)   # [W, T] = V^T
        for b in range(B):
            r0 = b * Lp
            for h in range(heads):
                q = qk[r0:, h * d:]
                k = qk[r0:, W + h * d:]
                ops.gemm(q, k, M=L, N=Lp, K=d, lda=2 * W, ldw=2 * W, out=s, ldc=Lp, alpha=scale, residual=mask, ldr=Lp)
                ops.softmax_rows_(s)
                ops.gemm(s, vt[h * d:, r0:], M=L, N=d, K=Lp, lda=Lp, ldw=T, out=o[r0:, h * d:], ldc=W)
        x = ops.linear(o, P["wo"], P["bo"], residual=x)
        h1 = ops.linear(ops.layer_norm(x, *P["ln2"]), P["w1"], P["b1"])
        x = ops.linear(ops.gelu_(h1), P["w2"], P["b2"], residual=x)
    return x


def _attn_mask(L, Lp, causal, device):
    m = torch.zeros((L, Lp), dtype=torch.float32, device=device)
    m[:, L:] = MASKED
    if causal:
        m[:, :L] += torch.full((L, L), MASKED, device=device).triu_(1)
    return m.clamp_(min=MASKED).half()


# ------------------------------------------------------------------------------------------------
# tokenizer
# ------------------------------------------------------------------------------------------------
SOT, EOT = 49406, 49407


@lru_cache()
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class BPETokenizer:
    """CLIP's byte-pair tokenizer from its published description: lower-cased, whitespace-collapsed text, split by the CLIP
    regex, bytes mapped to printable code points, merges applied in rank order with `</w>` marking the word end; the
    vocabulary is the 256 byte symbols, their `</w>` forms, the first 49152 - 512 - 2 merges and the two specials."""

    def __init__(self, bpe_path):
        import regex
        merges = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1]]
        vocab = list(_bytes_to_unicode().values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {v: i for i, v in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = _bytes_to_unicode()
        self.pat = regex.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)
        self.cache = {}

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self.cache[token] = word
        return word

    def encode(self, text):
        text = " ".join(html.unescape(html.unescape(text)).split()).strip().lower()
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(tok))
        return ids


@lru_cache()
def _bpe_from_env():
    path = os.environ.get("VCX_CLIP_BPE")
    return BPETokenizer(path) if path else None


def tokenize(texts, context_length=77):
    """open_clip.tokenize: [B, context_length] int64, <start_of_text> ids <end_of_text>, zero padded, truncated with the
    end token kept."""
    if isinstance(texts, str):
        texts = [texts]
    try:
        import open_clip                                   # the real thing, when the user has it
        return open_clip.tokenize(texts, context_length)
    except ImportError:
        pass
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        if t.strip() == "":
            ids = []
        else:
            bpe = _bpe_from_env()
            if bpe is None:
                raise RuntimeError(
                    "tokenising a non-empty prompt needs CLIP's BPE vocabulary: install open_clip_torch or point VCX_CLIP_BPE at "
                    "its bpe_simple_vocab_16e6.txt.gz (the empty prompt and ready-made token tensors work without it)")
            ids = bpe.encode(t)
        ids = [SOT] + ids + [EOT]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = EOT
        out[i, :len(ids)] = torch.tensor(ids)
    return out


# ------------------------------------------------------------------------------------------------
# the two embedders
# ------------------------------------------------------------------------------------------------
class AbstractEncoder(PackedModule):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenOpenCLIPEmbedder(AbstractEncoder):
    """Reference condition.py:174-240 (text tower)."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last"):
        super().__init__()
        assert layer in self.LAYERS
        self.model = _CLIP(arch, keep="text")
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _pack(self):
        m = self.model
        tr = m.transformer
        return dict(layers=_pack_blocks(tr, tr.layers - self.layer_idx), tok=_f16(m.token_embedding.weight),
                    pos=_f16(m.positional_embedding), lnf=(_f32(m.ln_final.weight), _f32(m.ln_final.bias), m.ln_final.eps))

    def forward(self, text):
        tokens = text if torch.is_tensor(text) else tokenize(text, self.model.context_length)
        return self.encode_with_transformer(tokens)

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        """[B, 77] int64 -> [B, 77, width] fp32 (condition.py:218-225)."""
        ops.require_gpu()
        pk = self.packed()
        dev = pk["tok"].device
        tokens = tokens.to(dev)
        B, L = tokens.shape
        tr = self.model.transformer
        W, Lp = tr.width, (L + 7) // 8 * 8
        x = torch.zeros((B, Lp, W), dtype=torch.float16, device=dev)
        x[:, :L] = pk["tok"][tokens] + pk["pos"][:L]
        x = _run_blocks(x.view(B * Lp, W), pk["layers"], B, L, Lp, W, tr.heads, _attn_mask(L, Lp, True, dev))
        x = ops.layer_norm(x, *pk["lnf"])
        return ops.to_f32(x).view(B, Lp, W)[:, :L].contiguous()

    def encode(self, text):
        return self(text)


def clip_preprocess(x, size=224, antialias=True, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """condition.py:322-329: kornia.geometry.resize(bicubic, align_corners=True, antialias) -> [0, 1] -> CLIP mean / std, as ONE
    libvcx kernel (vcx_clip_preprocess_f32; round 4 - it was plain torch conv2d / interpolate before).  kornia's anti-aliasing is
    a separable Gaussian (sigma = (factor - 1) / 2 per axis, kernel 4 sigma made odd, >= 3, mirror border) applied only when
    shrinking.  Pinned by two independent restatements of kornia's published algorithm (oracle/clip_oracle.py on torch ops,
    oracle/kornia_numpy.py as fp64 matrices) - kornia itself is not in the image."""
    return ops.clip_preprocess(x.float(), size, antialias, mean, std)


class FrozenOpenCLIPImageEmbedderV2(AbstractEncoder):
    """Reference condition.py:302-378 (vision tower, token output)."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled", antialias=True):
        super().__init__()
        self.model = _CLIP(arch, keep="visual")
        self.device, self.layer, self.antialias = device, layer, antialias
        if layer == "penultimate":
            raise NotImplementedError()
        if freeze:
            self.freeze()
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    def preprocess(self, x):
        return clip_preprocess(x, self.model.visual.image_size, self.antialias, self.mean.tolist(), self.std.tolist())

    def _pack(self):
        v = self.model.visual
        K = 3 * v.patch_size ** 2
        Kp = (K + 7) // 8 * 8
        w = torch.zeros((v.conv1.weight.shape[0], Kp), dtype=torch.float16, device=v.conv1.weight.device)
        w[:, :K] = v.conv1.weight.detach().reshape(-1, K).half()            # [width, (c, py, px)]
        return dict(layers=_pack_blocks(v.transformer, v.transformer.layers), wpatch=w, K=K, Kp=Kp,
                    cls=_f16(v.class_embedding), pos=_f16(v.positional_embedding),
                    lnpre=(_f32(v.ln_pre.weight), _f32(v.ln_pre.bias), v.ln_pre.eps))

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    @torch.no_grad()
    def encode_with_vision_transformer(self, x):
        """x [B, 3, H, W] in [-1, 1] -> [B, grid^2 + 1, width] fp32 (condition.py:347-378)."""
        ops.require_gpu()
        pk = self.packed()
        v = self.model.visual
        dev = pk["cls"].device
        x = self.preprocess(x.to(dev))
        B, p, g = x.shape[0], v.patch_size, v.grid
        W = v.transformer.width
        L = g * g + 1
        Lp = (L + 7) // 8 * 8
        # patches as rows of (c, py, px), the Conv2d weight's own order
        pat = torch.zeros((B * g * g, pk["Kp"]), dtype=torch.float16, device=dev)
        pat[:, :pk["K"]] = x.view(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, pk["K"]).half()
        tok = torch.zeros((B, Lp, W), dtype=torch.float16, device=dev)
        tok[:, 0] = (pk["cls"].float() + pk["pos"][0].float()).half()
        t2 = tok.view(B * Lp, W)
        for b in range(B):       # patch embedding + positional embedding, written behind the class token
            ops.gemm(pat[b * g * g:], pk["wpatch"], M=g * g, N=W, K=pk["Kp"], lda=pk["Kp"], out=t2[b * Lp + 1:], ldc=W,
                     residual=pk["pos"][1:], ldr=W)
        xt = ops.layer_norm(t2, *pk["lnpre"])
        xt = _run_blocks(xt, pk["layers"], B, L, Lp, W, v.transformer.heads, _attn_mask(L, Lp, False, dev))
        return ops.to_f32(xt).view(B, Lp, W)[:, :L].contiguous()

    def encode(self, image):
        return self(image)
