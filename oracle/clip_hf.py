"""TEST INFRASTRUCTURE ONLY - third-party pin for the OpenCLIP towers of the condition encoders.

The reference delegates the towers to `open_clip` (lvdm/modules/encoders/condition.py:174-240, 302-378), which is in neither
/root/reference nor this image; oracle/clip_oracle.py restates its module tree from the published architecture.  What IS in the
image is Hugging Face `transformers` (5.x): `CLIPTextModel` / `CLIPVisionModel` are an independent implementation of the same
CLIP architecture (pre-LN residual blocks, fused-QKV-equivalent attention with 1/sqrt(d) scaling, causal text mask, class token +
learned positions, `pre_layrnorm`, exact-erf GELU when `hidden_act="gelu"`).  This module builds small random-initialised HF
towers, renames their parameters to open_clip's state-dict names (q / k / v projections concatenated into `in_proj_*`), and
returns the HF outputs at the points the reference taps: text = `ln_final(hidden_states[-2])` (layer "penultimate",
condition.py:218-237), vision = `hidden_states[-1]` (all blocks, no ln_post, condition.py:347-378) on pre-processed pixels.
tests/test_oracle_golden.py checks oracle/clip_oracle.py against it, tests/test_model_gpu.py the HIP towers; nothing is stored:
the same wheel is on the GPU box.  Only tests/ may import this file.  (kornia's resize stays documented-unpinned.)
"""
import torch

# tiny towers: vision head dim 80 like ViT-H-14, text head dim 64; a small vocabulary (the embedding table is the only big tensor)
HF_TINY_CFG = dict(embed_dim=64,
                   vision=dict(image_size=224, layers=3, width=160, head_width=80, patch_size=56, mlp_ratio=2.0),
                   text=dict(context_length=77, vocab_size=1024, width=128, heads=2, layers=3, mlp_ratio=2.0))


def _strip(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def _blocks(src, dst_prefix, n_layers, out):
    for i in range(n_layers):
        s, d = f"encoder.layers.{i}.", f"{dst_prefix}.resblocks.{i}."
        out[d + "attn.in_proj_weight"] = torch.cat([src[s + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[d + "attn.in_proj_bias"] = torch.cat([src[s + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        for a, b in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"),
                     ("mlp.fc2", "mlp.c_proj")):
            out[d + b + ".weight"] = src[s + a + ".weight"]
            out[d + b + ".bias"] = src[s + a + ".bias"]


def build_text(cfg=HF_TINY_CFG, seed=0):
    """-> (HF model, {open_clip name: tensor} with the `model.` prefix of FrozenOpenCLIPEmbedder's state dict)."""
    from transformers import CLIPTextConfig, CLIPTextModel
    t = cfg["text"]
    torch.manual_seed(seed)
    m = CLIPTextModel(CLIPTextConfig(vocab_size=t["vocab_size"], hidden_size=t["width"], intermediate_size=int(t["width"] * t["mlp_ratio"]),
                                     num_hidden_layers=t["layers"], num_attention_heads=t["heads"], max_position_embeddings=t["context_length"],
                                     hidden_act="gelu", layer_norm_eps=1e-5, projection_dim=cfg["embed_dim"], bos_token_id=0, eos_token_id=1,
                                     pad_token_id=1, attention_dropout=0.0)).eval()
    with torch.no_grad():           # HF initialises LayerNorm to (1, 0) and biases to 0: make every parameter count
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape))
    src = _strip({k: v.detach().clone() for k, v in m.state_dict().items()}, "text_model.")
    out = {"model.token_embedding.weight": src["embeddings.token_embedding.weight"],
           "model.positional_embedding": src["embeddings.position_embedding.weight"],
           "model.ln_final.weight": src["final_layer_norm.weight"], "model.ln_final.bias": src["final_layer_norm.bias"]}
    _blocks(src, "model.transformer", t["layers"], out)
    return m, out


def hf_text_penultimate(m, tokens):
    """What the reference taps with layer='penultimate': the stream after all but the last block, through ln_final."""
    with torch.no_grad():
        hs = m(input_ids=tokens, output_hidden_states=True).hidden_states
        core = getattr(m, "text_model", m)
        return core.final_layer_norm(hs[-2])


def build_vision(cfg=HF_TINY_CFG, seed=1):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    v = cfg["vision"]
    torch.manual_seed(seed)
    m = CLIPVisionModel(CLIPVisionConfig(hidden_size=v["width"], intermediate_size=int(v["width"] * v["mlp_ratio"]), num_hidden_layers=v["layers"],
                                         num_attention_heads=v["width"] // v["head_width"], image_size=v["image_size"], patch_size=v["patch_size"],
                                         hidden_act="gelu", layer_norm_eps=1e-5, projection_dim=cfg["embed_dim"], attention_dropout=0.0)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1 and "class_embedding" not in n:
                p.add_(0.1 * torch.randn(p.shape))
    src = _strip({k: v_.detach().clone() for k, v_ in m.state_dict().items()}, "vision_model.")
    out = {"model.visual.class_embedding": src["embeddings.class_embedding"],
           "model.visual.conv1.weight": src["embeddings.patch_embedding.weight"],
           "model.visual.positional_embedding": src["embeddings.position_embedding.weight"],
           "model.visual.ln_pre.weight": src["pre_layrnorm.weight"], "model.visual.ln_pre.bias": src["pre_layrnorm.bias"],
           "model.visual.ln_post.weight": src["post_layernorm.weight"], "model.visual.ln_post.bias": src["post_layernorm.bias"]}
    _blocks(src, "model.visual.transformer", v["layers"], out)
    return m, out


def hf_vision_tokens(m, pixels):
    """Token stream after the last block (no post_layernorm): [B, grid^2 + 1, width]."""
    with torch.no_grad():
        return m(pixel_values=pixels, output_hidden_states=True).hidden_states[-1]
