"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from oracle.weights import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SCHEDULE_BUFFERS = ("betas", "alphas", "sqrt_", "log_one", "posterior", "scale_arr", "lvlb", "logvar")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def load_synth(module, seed=0, skip=()):
    """Fill a module with the deterministic synthetic weights keyed by its own state-dict names."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed, skip=skip)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return sd


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten() if not torch.is_tensor(a) else a.detach().double().cpu().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten() if not torch.is_tensor(b) else b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def psnr(a, b, peak=2.0):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    mse = float(((a - b) ** 2).mean())
    return 10 * np.log10(peak * peak / max(mse, 1e-30))
