"""TEST INFRASTRUCTURE ONLY - run the reference's own code from oracle/_ref/ (bytecode built by oracle/build_ref.py).

Only tests/, tests/golden/gen_golden.py and bench.py's `cpu_baseline` leg import this.  The reference imports three
third-party packages that are absent from this image and unused on the inference path (cv2, pytorch_lightning, torchvision):
they are stubbed exactly as SURVEY.md §8(c) lists (LightningModule = nn.Module with a .device property, rank_zero_only =
identity, torchvision.utils.make_grid = no-op).
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


_available = None


def available():
    """True when oracle/_ref is present AND importable by THIS interpreter.  The tree is bytecode: built by another CPython minor
    version it fails with ImportError (bad magic number) - callers then fall back to the oracle port (bench.py `cpu_baseline.kind`
    = "port") or skip, instead of dying inside the import."""
    global _available
    if _available is None:
        _available = False
        if os.path.exists(os.path.join(REF_DIR, "lvdm", "modules", "networks", "openaimodel3d.pyc")):
            try:
                import_reference()
                _available = True
            except Exception as e:      # ImportError / bad magic / a stale tree
                print(f"[ref_runner] oracle/_ref is present but not importable here ({type(e).__name__}: {e}); using the oracle port",
                      file=sys.stderr)
                for name in [n for n in sys.modules if n == "lvdm" or n.startswith("lvdm.") or n == "utils.diffusion_utils"]:
                    sys.modules.pop(name, None)
                if REF_DIR in sys.path:
                    sys.path.remove(REF_DIR)
    return _available


_REF_MODULES = ("lvdm.models.ddpm3d", "lvdm.models.autoencoder", "lvdm.models.samplers.ddim", "lvdm.models.samplers.ddim_multiplecond",
                "lvdm.modules.networks.openaimodel3d", "lvdm.modules.networks.ae_modules", "lvdm.modules.attention",
                "lvdm.modules.encoders.resampler", "utils.diffusion_utils")


def install_stubs():
    """cv2 / pytorch_lightning / torchvision stand-ins (only created when the real module is absent); returns the names created."""
    made = []
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
        made.append("cv2")
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(torch.nn.Module):
            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")
        pl.LightningModule = LightningModule
        plu = types.ModuleType("pytorch_lightning.utilities")
        plu.rank_zero_only = lambda f: f
        pl.utilities = plu
        sys.modules["pytorch_lightning"] = pl
        sys.modules["pytorch_lightning.utilities"] = plu
        made += ["pytorch_lightning", "pytorch_lightning.utilities"]
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tv.utils = tvu
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
        made += ["torchvision", "torchvision.utils"]
    return made


_imported = False


def import_reference(path=None):
    """Import the reference's modules (default: the bytecode tree oracle/_ref) behind the stubs, then take the stubs out of
    sys.modules again: the reference modules keep the names they bound at import, and nothing else in the process (transformers
    probes `find_spec("torchvision")` / `find_spec("cv2")` when IT is imported) ever sees a fake package."""
    global _imported
    path = path or REF_DIR
    if path not in sys.path:
        sys.path.insert(0, path)
    if not _imported:
        import importlib
        made = install_stubs()
        try:
            for name in _REF_MODULES:
                importlib.import_module(name)
        finally:
            for name in made:
                sys.modules.pop(name, None)
        _imported = True
    return path


def reference_unet(hp, state_dict=None):
    """The reference's UNetModel (openaimodel3d.py:281-603) built from the YAML's unet_config.params, fp32 on the CPU."""
    import_reference()
    from lvdm.modules.networks.openaimodel3d import UNetModel
    kw = dict(hp)
    kw["use_checkpoint"] = False
    m = UNetModel(**kw).eval()
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True)
    return m


def reference_vae(ddconfig, embed_dim=4, state_dict=None):
    import_reference()
    from lvdm.models.autoencoder import AutoencoderKL
    m = AutoencoderKL(ddconfig=dict(ddconfig), lossconfig={"target": "torch.nn.Identity"}, embed_dim=embed_dim).eval()
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True)
    return m


class AttrDict(dict):
    """Stand-in for the OmegaConf nodes the reference's constructors read (cfg['k'], cfg.k, 'k' in cfg, cfg.get)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in o.items()})
        if isinstance(o, (list, tuple)):
            return [AttrDict.wrap(v) for v in o]
        return o


def reference_diffusion(params, unet_sd, vae_sd, device):
    """The reference's VIPLatentDiffusion (ddpm3d.py:1029-1080 on LatentDiffusion :462-1027 on DDPM :39-460) for the YAML's
    `model.params`, with the reference's UNetModel / AutoencoderKL inside and Identity conditioners (the CLIP encoders are outside
    this path).  The two networks are built on the meta device and take the given tensors themselves (`assign=True`: no second
    copy of 1.44 B parameters, no random init); the schedule buffers are computed by the reference's own __init__."""
    import copy
    import_reference()
    from lvdm.models.ddpm3d import VIPLatentDiffusion
    p = copy.deepcopy(dict(params))
    ident = {"target": "torch.nn.Identity"}
    hp = dict(p["unet_config"]["params"])
    fp = dict(p["first_stage_config"]["params"])
    p["unet_config"] = {"target": "torch.nn.Identity", "params": hp}              # nn.Identity swallows the kwargs; swapped below
    p["first_stage_config"] = {"target": "torch.nn.Identity", "params": fp}
    p["cond_stage_config"] = p["img_cond_stage_config"] = p["image_proj_stage_config"] = ident
    model = VIPLatentDiffusion(**AttrDict.wrap(p)).eval().to(device)
    with torch.device("meta"):
        unet = reference_unet(hp)
        vae = reference_vae(fp["ddconfig"], embed_dim=fp.get("embed_dim", 4))
    unet.load_state_dict(unet_sd, strict=True, assign=True)
    vae.load_state_dict(vae_sd, strict=True, assign=True)
    model.model.diffusion_model = unet
    model.first_stage_model = vae
    return model
