"""Diffusion driver (reference utils/diffusion_utils.py): config instantiation, checkpoint loading and
`image_guided_synthesis`, the function ViewCrafter.run_diffusion calls (viewcrafter.py:93-106)."""
import importlib
from collections import OrderedDict

import torch

# The reference YAMLs name reference classes; on this path they are served by the MI355X implementations.
TARGET_ALIASES = {
    "lvdm.models.ddpm3d.VIPLatentDiffusion": "viewcrafter_amd.lvdm.models.ddpm3d.VIPLatentDiffusion",
    "lvdm.models.ddpm3d.LatentVisualDiffusion": "viewcrafter_amd.lvdm.models.ddpm3d.LatentVisualDiffusion",
    "lvdm.models.ddpm3d.LatentDiffusion": "viewcrafter_amd.lvdm.models.ddpm3d.LatentDiffusion",
    "lvdm.modules.networks.openaimodel3d.UNetModel": "viewcrafter_amd.lvdm.modules.networks.openaimodel3d.UNetModel",
    "lvdm.models.autoencoder.AutoencoderKL": "viewcrafter_amd.lvdm.models.autoencoder.AutoencoderKL",
    "lvdm.modules.encoders.resampler.Resampler": "viewcrafter_amd.lvdm.modules.encoders.resampler.Resampler",
    "lvdm.modules.encoders.resampler.ImageProjModel": "viewcrafter_amd.lvdm.modules.encoders.resampler.ImageProjModel",
    "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder": "viewcrafter_amd.lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder",
    "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2":
        "viewcrafter_amd.lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2",
}


def count_params(model, verbose=False):
    """Reference utils/diffusion_utils.py:12-16."""
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{type(model).__name__} has {n * 1e-6:.2f} M params.")
    return n


def check_istarget(name, para_list):
    """Reference utils/diffusion_utils.py:19-28: does the full parameter name contain any of the partial names?"""
    return any(part in name for part in para_list)


def setup_dist(args):
    """Reference utils/diffusion_utils.py:74-81 (nccl = RCCL on ROCm, env:// rendezvous); `args.local_rank` selects the GPU."""
    import torch.distributed as dist
    if dist.is_initialized():
        return
    torch.cuda.set_device(args.local_rank)
    dist.init_process_group("nccl", init_method="env://")


def get_obj_from_str(string, reload=False):
    string = TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module, package=None)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    """Reference diffusion_utils.py:31-38: `target` dotted path + `params` kwargs; the two sentinel strings give None."""
    if "target" not in config:
        if config == "__is_first_stage__" or config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_model_checkpoint(model, ckpt):
    """Reference diffusion_utils.py:83-108: Lightning ('state_dict', strict, with the framestride_embed ->
    fps_embedding rename retry) or DeepSpeed ('module', 16-char prefix) layouts.  torch>=2.6 defaults to
    weights_only=True, which Lightning checkpoints do not satisfy."""
    state_dict = torch.load(ckpt, map_location="cpu", weights_only=False)
    if "state_dict" in list(state_dict.keys()):
        state_dict = state_dict["state_dict"]
        try:
            model.load_state_dict(state_dict, strict=True)
        except RuntimeError:
            renamed = OrderedDict((k.replace("framestride_embed", "fps_embedding"), v) for k, v in state_dict.items())
            model.load_state_dict(renamed, strict=True)
    else:
        new_sd = OrderedDict((key[16:], val) for key, val in state_dict["module"].items())
        model.load_state_dict(new_sd)
    print(">>> model checkpoint loaded.")
    return model


def get_latent_z(model, videos):
    """Reference diffusion_utils.py:110-115."""
    b, c, t, h, w = videos.shape
    x = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = model.encode_first_stage(x)
    return z.view(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()


def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, timestep_spacing="uniform", guidance_rescale=0.0,
                           condition_index=None, **kwargs):
    """Reference diffusion_utils.py:117-201.  videos [B, 3, T, H, W] in [-1, 1]; returns [B, n_samples, 3, T, H, W]."""
    from ..lvdm.models.samplers.ddim import DDIMSampler
    if multiple_cond_cfg:
        from ..lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSamplerMulti
        ddim_sampler = DDIMSamplerMulti(model)
    else:
        ddim_sampler = DDIMSampler(model)
    batch_size = noise_shape[0]
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if not text_input:
        prompts = [""] * batch_size
    assert condition_index is not None, "Error: condition index is None!"

    img = videos[:, :, condition_index[0]]
    img_emb = model.image_proj_model(model.embedder(img))
    cond_emb = model.get_learned_conditioning(prompts)
    cond = {"c_crossattn": [torch.cat([cond_emb, img_emb], dim=1)]}
    if model.model.conditioning_key == "hybrid":
        img_cat_cond = get_latent_z(model, videos)
        cond["c_concat"] = [img_cat_cond]

    if unconditional_guidance_scale != 1.0:
        if model.uncond_type == "empty_seq":
            uc_emb = model.get_learned_conditioning(batch_size * [""])
        elif model.uncond_type == "zero_embed":
            uc_emb = torch.zeros_like(cond_emb)
        uc_img_emb = model.image_proj_model(model.embedder(torch.zeros_like(img)))
        uc = {"c_crossattn": [torch.cat([uc_emb, uc_img_emb], dim=1)]}
        if model.model.conditioning_key == "hybrid":
            uc["c_concat"] = [img_cat_cond]
    else:
        uc = None

    if multiple_cond_cfg and cfg_img != 1.0:
        uc_2 = {"c_crossattn": [torch.cat([uc_emb, img_emb], dim=1)]}
        if model.model.conditioning_key == "hybrid":
            uc_2["c_concat"] = [img_cat_cond]
        kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})

    batch_variants = []
    for _ in range(n_samples):
        samples, _ = ddim_sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:],
                                         verbose=False, unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None,
                                         fs=fs, timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale,
                                         **kwargs)
        batch_variants.append(model.decode_first_stage(samples))
    return torch.stack(batch_variants).permute(1, 0, 2, 3, 4, 5)
