"""Model construction helpers shared by viewcrafter.py, bench.py and the tests.

`build_diffusion_model` is what ViewCrafter.setup_diffusion (reference viewcrafter.py:384-404) does: load the YAML,
force use_checkpoint=False, instantiate, move to the device, set perframe_ae, optionally load a checkpoint.
"""
import copy
import math

import torch

from .config import Config, load_yaml
from .utils.diffusion_utils import instantiate_from_config, load_model_checkpoint

IDENTITY = Config(target="torch.nn.Identity")


def model_config_from_yaml(path, conditioners="config"):
    """conditioners: 'config' keeps the YAML's targets (the two OpenCLIP encoders and the Resampler `image_proj_model` all
    resolve to this package's libvcx implementations through TARGET_ALIASES; neither open_clip nor kornia is needed - a
    non-empty text prompt needs CLIP's BPE vocabulary, see lvdm/modules/encoders/condition.py); 'clip_external' replaces only the two CLIP encoders with nn.Identity
    (their token embeddings are computed elsewhere and fed in, the projector runs natively); 'identity' replaces all
    three (benchmarks and tests feed the final context tensors)."""
    cfg = load_yaml(path)
    mc = copy.deepcopy(cfg["model"])
    mc["params"]["unet_config"]["params"]["use_checkpoint"] = False
    if conditioners == "identity":
        for k in ("cond_stage_config", "img_cond_stage_config", "image_proj_stage_config"):
            mc["params"][k] = IDENTITY
    elif conditioners == "clip_external":
        for k in ("cond_stage_config", "img_cond_stage_config"):
            mc["params"][k] = IDENTITY
    elif conditioners != "config":
        raise ValueError(f"conditioners must be 'config', 'clip_external' or 'identity' (got {conditioners!r})")
    return Config.wrap(mc)


def build_diffusion_model(config_path, device="cuda", ckpt_path=None, perframe_ae=True, conditioners="config",
                          init_on_device=True):
    mc = model_config_from_yaml(config_path, conditioners)
    if init_on_device and str(device) != "cpu":
        with torch.device(device):
            model = instantiate_from_config(mc)
    else:
        model = instantiate_from_config(mc).to(device)
    model.perframe_ae = perframe_ae
    if ckpt_path is not None:
        model = load_model_checkpoint(model, ckpt_path).to(device)
    return model.eval()


@torch.no_grad()
def randomize_parameters(model, seed=0, std_scale=1.0):
    """Synthetic weights for benchmarks without checkpoints: every conv/linear weight ~ N(0, 1/fan_in) (keeps
    activations O(1) through the 1.4 B-parameter net), norm scales 1 + 0.1 N, biases 0.05 N — in particular the 79
    layers the reference zero-initialises become non-trivial."""
    g = torch.Generator(device=next(model.parameters()).device)
    g.manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() >= 2:
            fan_in = p[0].numel()
            p.copy_(torch.randn(p.shape, generator=g, device=p.device) * (std_scale / math.sqrt(fan_in)))
        elif name.endswith("weight"):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device))
        else:
            p.copy_(0.05 * torch.randn(p.shape, generator=g, device=p.device))
    for m in model.modules():           # in-place copies do not invalidate the packed fp16 weights by themselves
        if hasattr(m, "_drop_packed"):
            m._drop_packed()
    return model
