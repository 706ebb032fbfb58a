"""Tiny hyper-parameters shared by the golden generator, the oracle tests and the GPU parity tests.

Same graph as configs/inference_pvd_1024.yaml (all block types, both context branches, init_attn, fs conditioning),
narrower: model_channels 64 (GroupNorm-32 divisibility holds), head dim 64, context dim 128, VAE ch 32.
"""
import copy

TINY_UNET = dict(
    in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=128, use_linear=True,
    use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
    image_cross_attention=True, default_fs=10, fs_condition=True,
)

TINY_DDCONFIG = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
                     num_res_blocks=2, attn_resolutions=[], dropout=0.0)


# image_proj_model (configs/inference_pvd_1024.yaml:100-111), narrowed: dim 128, 2 heads of 64, 2 layers, 4 queries x 3 frames
TINY_RESAMPLER = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=192, output_dim=128, ff_mult=4,
                      video_length=3)


def tiny_model_params(unet_target, vae_target, base_scale=0.3):
    """`params` of the model YAML (configs/inference_pvd_1024.yaml:6-110) with the tiny sub-configs and Identity
    conditioners (the CLIP encoders run once per video and are out of scope)."""
    ident = {"target": "torch.nn.Identity"}
    return copy.deepcopy(dict(
        rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012, num_timesteps_cond=1,
        log_every_t=200, timesteps=1000, first_stage_key="video", cond_stage_key="caption", cond_stage_trainable=False,
        image_proj_model_trainable=True, conditioning_key="hybrid", image_size=[32, 16], channels=4, scale_by_std=False,
        scale_factor=0.18215, use_ema=False, uncond_prob=0.05, uncond_type="empty_seq", rand_cond_frame=True,
        use_dynamic_rescale=True, base_scale=base_scale, fps_condition_type="fps", perframe_ae=True, loop_video="Flase",
        unet_config={"target": unet_target, "params": TINY_UNET},
        first_stage_config={"target": vae_target,
                            "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": TINY_DDCONFIG,
                                       "lossconfig": ident}},
        cond_stage_config=ident, img_cond_stage_config=ident, image_proj_stage_config=ident,
    ))


# tiny OpenCLIP-shaped towers (vision head dim 80 like ViT-H; 224 / 56 = 4x4 patches + class token = 17 tokens)
CLIP_TINY = "vcx-tiny-test"
CLIP_TINY_CFG = dict(embed_dim=64,
                     vision=dict(image_size=224, layers=3, width=160, head_width=80, patch_size=56, mlp_ratio=2.0),
                     text=dict(context_length=77, vocab_size=49408, width=128, heads=2, layers=3, mlp_ratio=2.0))


# image_guided_synthesis fixture (tests/golden/igs_tiny.npz): 4 frames of 256x128 (latent 32x16), the tiny UNet / VAE, the tiny
# OpenCLIP towers as cond_stage / embedder and a Resampler sized for them (vision width 160 in, 16 queries x 4 frames = 64 image
# tokens out = the per-frame branch 77 + 16 T of openaimodel3d.py:556)
IGS_T, IGS_H, IGS_W = 4, 256, 128
IGS_RESAMPLER = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=160, output_dim=128, ff_mult=4,
                     video_length=IGS_T)


def igs_model_params(unet_target, vae_target, clip_text_target, clip_image_target, resampler_target, arch=CLIP_TINY, **embedder_kw):
    p = tiny_model_params(unet_target, vae_target)
    p["cond_stage_config"] = {"target": clip_text_target, "params": dict(arch=arch, freeze=True, layer="penultimate", **embedder_kw)}
    p["img_cond_stage_config"] = {"target": clip_image_target, "params": dict(arch=arch, freeze=True, **embedder_kw)}
    p["image_proj_stage_config"] = {"target": resampler_target, "params": dict(IGS_RESAMPLER)}
    return copy.deepcopy(p)
