"""Multi-process (world_size 2, gloo, CPU) coverage of the trajectory-sharding path used with RCCL on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from viewcrafter_amd import parallel


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 9, 25):
        for w in (1, 2, 4, 8):
            owned = [parallel.shard_indices(n, r, w) for r in range(w)]
            flat = sorted(i for o in owned for i in o)
            assert flat == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
            assert all(parallel.owner_of(i, w) == r for r, o in enumerate(owned) for i in o)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different weights on every rank before the broadcast
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GroupNorm(4, 32), torch.nn.Linear(32, 8))
    net.register_buffer("table", torch.randn(10))
    parallel.broadcast_module_(net, src=0, bucket_bytes=1024)   # tiny buckets: exercises the multi-bucket path
    flat = torch.cat([p.reshape(-1) for p in net.parameters()] + [net.table])
    ref = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(ref, flat)
    same = all(torch.equal(ref[0], x) for x in ref)

    def fn(item, idx):                                  # stands in for one image_guided_synthesis call
        return torch.full((2, 3), float(item * 10 + idx))
    res = parallel.run_sharded(fn, list(range(n_items)))
    if rank == 0:
        q.put((same, [float(t[0, 0]) for t in res]))
    else:
        assert res is None
        q.put((same, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 8])
def test_broadcast_and_gather_two_ranks(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[0] for o in outs), "weights differ after broadcast"
    gathered = [o[1] for o in outs if o[1] is not None][0]
    assert gathered == [float(i * 10 + i) for i in range(n_items)]


def _worker_mismatch(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    empty = parallel.gather_results({}, 0)
    # clip 1 (owned by rank 1) has another length: EVERY rank must raise, none may be left waiting in the all_gather
    local = {rank: torch.zeros(3 + rank, 2)}
    try:
        parallel.gather_results(local, 2)
        q.put((rank, empty, "no error"))
    except ValueError as e:
        q.put((rank, empty, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_rejects_mixed_shapes_on_every_rank_and_handles_zero_items():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][1] == [] and outs[1][1] is None
    assert all("one shape and dtype" in o[2] for o in outs), outs


def _bcast_list_worker(rank, world, port, fail, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    clips = [torch.arange(24, dtype=torch.float32).view(2, 3, 4) + i for i in range(3)] + [torch.ones(5, dtype=torch.float64)] if rank == 0 else None
    try:
        got = parallel.broadcast_tensor_list(clips, src=0, error="DUSt3R went wrong" if (fail and rank == 0) else None)
        q.put((rank, [(tuple(t.shape), str(t.dtype), float(t.sum())) for t in got]))
    except RuntimeError as e:
        q.put((rank, "raised: " + str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_broadcast_tensor_list_two_ranks_and_a_failing_source(fail):
    """The clips of the sparse-view mode travel from rank 0 (the only rank that runs DUSt3R + the render) to every rank; a failure
    on rank 0 while producing them is raised on ALL ranks instead of leaving the others in a collective rank 0 never joins."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_list_worker, args=(r, 2, port, fail, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail:
        assert all(isinstance(v, str) and "DUSt3R went wrong" in v for v in outs.values()), outs
    else:
        assert outs[0] == outs[1] and len(outs[0]) == 4 and outs[0][3] == ((5,), "torch.float64", 5.0)


def test_interleaved_clips_draw_the_same_numbers_as_a_plain_loop():
    """viewcrafter_amd/interleave.py: two clips in flight at a time, taking turns at every sampler step, each with the global generator as
    part of its context - every clip must draw exactly what it draws running alone after manual_seed(seed + index), whatever the
    interleaving, and the generator must be left as a plain loop leaves it."""
    import torch
    from viewcrafter_amd.interleave import run_interleaved, step_yield

    def clip(item, index):
        if index > 0:
            torch.manual_seed(100 + index)
        acc = torch.zeros(3)
        for _ in range(item):                     # `item` sampler steps, one draw each, baton handed on after every step
            acc = acc * 0.5 + torch.randn(3)
            step_yield()
        return acc + torch.rand(3)                # (the decode-side draw behind the loop)

    items = [(0, 5), (1, 3), (2, 7), (3, 1), (4, 4)]
    torch.manual_seed(7)
    want = [clip(item, index) for index, item in items]
    end_state = torch.random.get_rng_state()
    for lanes in (2, 3):
        torch.manual_seed(7)
        got = run_interleaved(clip, items, n_lanes=lanes)
        assert all(torch.equal(a, b) for a, b in zip(got, want)), f"{lanes} lanes"
        assert torch.equal(torch.random.get_rng_state(), end_state)

    def boom(item, index):
        step_yield()
        if index == 1:
            raise RuntimeError("clip 1 failed")
        step_yield()
        return torch.zeros(1)
    import pytest
    with pytest.raises(RuntimeError, match="clip 1 failed"):
        run_interleaved(boom, [(0, 0), (1, 0)], n_lanes=2)
