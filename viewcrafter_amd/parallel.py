"""Trajectory sharding across the GPUs of one node (one process per GPU, torch.distributed on RCCL over xGMI).

The unit of work on this path is one trajectory / clip = one `image_guided_synthesis` call: independent noise and
conditioning, no cross-sample operation anywhere in the UNet, the VAE or the sampler.  So the DDIM loop
needs NO in-step collective; the only communication is
  (i)  once at start-up: broadcast of the weights from rank 0 (2.9 GB fp16 / 5.8 GB fp32; instead of N disk reads) and
       of any conditioning the trajectories share,
  (ii) once at the end: gather of the decoded clips (or of timing scalars) to rank 0.
Rank r takes trajectories r, r + W, r + 2W, ...  The reference has no multi-GPU inference at all; its only collective
helper is the dead `gather_data` (lvdm/common.py:8-14), kept there for API parity.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment.  backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' on CPU."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def rank_world():
    """(rank, world size) of the current process group, (0, 1) without one."""
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def shard_indices(n_items, rank, world):
    """Round-robin ownership: rank r owns items r, r+W, ...  (balanced to within one item)."""
    return list(range(rank, n_items, world))


def owner_of(index, world):
    return index % world


@torch.no_grad()
def broadcast_module_(module, src=0, bucket_bytes=256 << 20, _force=False):
    """Make every rank's parameters and buffers equal to rank `src`'s with a few large broadcasts (xGMI is
    point-to-point: few big messages beat thousands of small ones).  Tensors are grouped by (dtype, device) into flat
    buckets; under RCCL a tensor that lives on the host is staged through the current GPU.
    `_force`: run the collectives even in a one-rank group (tests/test_entry_gpu.py exercises the RCCL path on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _force):
        return module
    rccl = dist.get_backend() == "nccl"
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), group in groups.items():
        staged = rccl and device.type != "cuda"
        bucket, size = [], 0
        for t in group + [None]:
            if t is not None and (size + t.numel() * t.element_size() <= bucket_bytes or not bucket):
                bucket.append(t)
                size += t.numel() * t.element_size()
                continue
            flat = torch.cat([b.reshape(-1) for b in bucket])
            if staged:
                flat = flat.cuda()
            dist.broadcast(flat, src=src)
            if staged:
                flat = flat.to(device)
            off = 0
            for b in bucket:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    drop_packed_copies(module)
    return module


def drop_packed_copies(module):
    """Parameters were overwritten in place: every PackedModule below `module` (the root usually is a plain nn.Module)
    must forget its kernel-layout fp16 copies, and the UNet its cached context K/V and captured graphs."""
    for m in module.modules():
        if hasattr(m, "_drop_packed"):
            m._drop_packed()
    return module


def broadcast_tensor(t, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def broadcast_tensor_list(tensors, src=0, error=None):
    """Rank `src` holds a list of tensors (any shapes / dtypes), the others pass None: afterwards every rank holds the list.
    The shapes travel first as one small object broadcast, the payload as one dist.broadcast per tensor (on the GPU under RCCL,
    on the host under gloo).  `error` (rank src only): a failure that happened while PRODUCING the list - it is broadcast in place
    of the metadata and raised on every rank together, so that nobody waits in a collective the source never joins."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if error is not None:
            raise RuntimeError(error)
        return list(tensors)
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta = [("error", str(error)) if error is not None else
                ("ok", [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors])]
    dist.broadcast_object_list(meta, src=src)
    kind, info = meta[0]
    if kind == "error":
        raise RuntimeError(f"rank {src} failed while producing the clips: {info}")
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    out = []
    for i, (shape, dtype) in enumerate(info):
        t = tensors[i].to(dev).contiguous() if rank == src else torch.empty(shape, dtype=getattr(torch, dtype), device=dev)
        dist.broadcast(t, src=src)
        out.append(tensors[i] if rank == src else t)
    return out


def gather_results(local, n_items, dst=0, _force=False):
    """local: {item index: tensor} owned by this rank (all tensors of one shape/dtype).  Returns on rank `dst` the list
    of all n_items results in item order (None elsewhere).  One all_gather of a padded stack per call.
    (`_force`: as in broadcast_module_.)"""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _force):
        return [local[i] for i in range(n_items)]
    rank, world = dist.get_rank(), dist.get_world_size()
    if n_items == 0:
        return [] if rank == dst else None
    per_rank = (n_items + world - 1) // world
    example = next(iter(local.values())) if local else None
    # every rank reports the (shape, dtype) of ALL its results first: a mismatch must fail on every rank together - if only the
    # rank holding the odd clip raised, the others would wait in the all_gather below until the launcher kills them
    mine = sorted({(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in local.values()})
    meta = [None] * world
    dist.all_gather_object(meta, mine)
    kinds = sorted({k for m in meta for k in m})
    if len(kinds) != 1:
        raise ValueError(f"gather_results needs results of one shape and dtype, the ranks hold {kinds} "
                         "(clips of different length / size must be generated in separate launches)")
    shape, dtype = kinds[0]
    dev = example.device if example is not None else (torch.device("cuda", torch.cuda.current_device())
                                                      if dist.get_backend() == "nccl" else torch.device("cpu"))
    stack = torch.zeros((per_rank,) + tuple(shape), dtype=getattr(torch, dtype), device=dev)
    for slot, idx in enumerate(shard_indices(n_items, rank, world)):
        stack[slot] = local[idx]
    out = [torch.empty_like(stack) for _ in range(world)]
    dist.all_gather(out, stack)
    if rank != dst:
        return None
    return [out[owner_of(i, world)][i // world] for i in range(n_items)]


def run_sharded(fn, items, gather=True, lanes=1, model=None):
    """Apply fn(item, index) to the items this rank owns; optionally gather the tensor results on rank 0.  lanes > 1: that many of the
    rank's items in flight at a time, interleaved step by step on their own HIP streams (interleave.run_interleaved); `model`: the
    module the lanes share - its lazily built weight packs are built here, on the caller's stream, before the lanes start."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    mine = list(shard_indices(len(items), rank, world))
    if lanes > 1 and len(mine) > 1:
        from .interleave import prepack, run_interleaved
        if model is not None and torch.cuda.is_available():
            prepack(model)
        local = dict(zip(mine, run_interleaved(fn, [(i, items[i]) for i in mine], n_lanes=lanes)))
    else:
        local = {i: fn(items[i], i) for i in mine}
    return gather_results(local, len(items)) if gather else local


def shutdown(barrier=True):
    """(Barrier +) destroy_process_group when a group exists (end of a torchrun launch)."""
    if dist.is_initialized():
        try:
            if barrier:
                dist.barrier()
        finally:
            dist.destroy_process_group()
