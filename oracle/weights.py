"""ORACLE — TEST INFRASTRUCTURE ONLY.

Deterministic synthetic parameters keyed by state-dict name, so the reference model (golden generation, build
container), the oracle and the product model (GPU box) all see bit-identical weights without shipping checkpoints:
value(key) = f(numpy.default_rng(seed ^ crc32(key))).  No checkpoints exist offline, and the reference's own
initialisation zeroes 79 output layers, which would make parity vacuous.
"""
import zlib

import numpy as np
import torch


def synth_tensor(key, shape, seed=0):
    rng = np.random.default_rng((seed * 1000003) ^ zlib.crc32(key.encode()))
    shape = tuple(shape)
    x = rng.standard_normal(shape, dtype=np.float32)
    if len(shape) >= 2:                       # conv / linear weight: keep activations O(1)
        fan_in = int(np.prod(shape[1:]))
        x *= 1.0 / np.sqrt(fan_in)
    elif key.endswith("weight"):              # norm scale
        x = 1.0 + 0.1 * x
    else:                                     # bias / norm shift
        x *= 0.05
    return torch.from_numpy(x)


def synth_state_dict(shapes, seed=0, skip=()):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    out = {}
    for k, shp in shapes.items():
        if any(k.startswith(s) for s in skip):
            continue
        out[k] = synth_tensor(k, shp, seed)
    return out


def synth_input(name, shape, seed=0, scale=1.0):
    rng = np.random.default_rng((seed * 1000003) ^ zlib.crc32(("input:" + name).encode()))
    return torch.from_numpy(rng.standard_normal(tuple(shape), dtype=np.float32) * np.float32(scale))


class NamedRandn:
    """Drop-in for `torch.randn` while a fixture is generated / replayed: the k-th call returns synth_input(f"{prefix}_{k}").
    The reference and the product draw their Gaussians in the same order (posterior sample of the condition clip, x_T, one
    noise tensor per DDIM step), so patching both with this makes `image_guided_synthesis` comparable end to end."""

    def __init__(self, prefix):
        self.prefix, self.calls = prefix, 0

    def __call__(self, *size, device=None, dtype=None, generator=None, **kwargs):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        self.calls += 1
        t = synth_input(f"{self.prefix}_{self.calls}", size)
        if dtype is not None:
            t = t.to(dtype)
        return t.to(device) if device is not None else t
