"""Small helpers of the reference's lvdm/common.py that the hot path uses."""
from inspect import isfunction

import torch


def exists(val):
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def extract_into_tensor(a, t, x_shape):
    """lvdm/common.py:25-28: a.gather(-1, t) reshaped to broadcast over x."""
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    """lvdm/common.py:31-34."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def gather_data(data, return_np=True):
    """lvdm/common.py:8-14 (dead code in the reference; here it backs the multi-GPU result gather over RCCL)."""
    import torch.distributed as dist
    data_list = [torch.zeros_like(data) for _ in range(dist.get_world_size())]
    dist.all_gather(data_list, data)
    if return_np:
        data_list = [d.cpu().numpy() for d in data_list]
    return data_list
