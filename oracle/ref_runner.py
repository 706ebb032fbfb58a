"""TEST INFRASTRUCTURE ONLY - run the reference's own code from oracle/_ref/ (bytecode built by oracle/build_ref.py).

Only tests/, tests/golden/gen_golden.py and bench.py's `cpu_baseline` leg import this.  The reference imports three
third-party packages that are absent from this image and unused on the inference path (cv2, pytorch_lightning, torchvision):
they are stubbed exactly as SURVEY.md §8(c) lists (LightningModule = nn.Module with a .device property, rank_zero_only =
identity, torchvision.utils.make_grid = no-op).
"""
import importlib.machinery
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def available():
    return os.path.exists(os.path.join(REF_DIR, "lvdm", "modules", "networks", "openaimodel3d.pyc"))


def install_stubs():
    """cv2 / pytorch_lightning / torchvision stand-ins (only created when the real module is absent)."""
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
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
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tv.utils = tvu
        tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)   # transformers probes find_spec("torchvision")
        tvu.__spec__ = importlib.machinery.ModuleSpec("torchvision.utils", None)
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu


def import_reference(path=None):
    """Put the reference (default: the bytecode tree oracle/_ref) on sys.path behind the stubs."""
    install_stubs()
    path = path or REF_DIR
    if path not in sys.path:
        sys.path.insert(0, path)
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
