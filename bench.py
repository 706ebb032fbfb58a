"""Headline benchmark: DDIM steps/s of the ViewCrafter denoising loop on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Started as a plain `python bench.py --gpus N` the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (the same line
the driver uses; under that launcher - WORLD_SIZE set - it just runs as one rank).  Either way the line it prints carries
`n_gpus` = the world size it really ran with, which must equal --gpus, and the backend must be nccl (= RCCL).

Workload (N=1): configs[3] of BASELINE.json — ViewCrafter_25, 576x1024x25 frames (latent 25x72x128), the 1.44 B-parameter
lvdm UNet built from configs/inference_pvd_1024.yaml, one trajectory per GPU.  One step = one DDIM step = two UNet
evaluations (cond + uncond, run as one B=2 forward) + the fused CFG / guidance-rescale / v-prediction / dynamic-rescale /
x_{t-1} update, eta = 1 (fresh noise every step), fp16 storage with fp32 accumulation.  Weights are synthetic (no
checkpoints offline), data is synthetic of the real shapes; inputs are resident in HBM before the timed region.

Printed JSON (rank 0): the driver contract + `roofline` (dominant kernel = the MFMA GEMM/conv engine, measured with HIP
events on the launch stream over the timed region) + `roofline_flash` (second family, same method) + `cpu_baseline` (the
reference's own code from oracle/_ref - or, if that tree is not built, the fp32 oracle port - timed on the host cores at
BASELINE configs[0]'s shapes 16x40x64 and FLOP-scaled) + `gpu_eager_baseline` (the reference's own UNetModel.forward, or
the oracle graph, as plain PyTorch-ROCm eager ops under fp16 autocast on this MI355X: what the hand-written kernels buy
over the libraries) + `parity` (HIP path vs the fp32 oracle run on the GPU at the
bench's own latent) + `sec_per_video` (a real 50-step sample() + 25-frame decode after the timed region, N = 1).
N > 1: rank 0 initialises the weights, the others receive them by RCCL broadcast (`broadcast_s`), all ranks assert equal
checksums, and the decoded clips are gathered on rank 0 (`gather_s`) - both outside the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MI355X_FP16_DENSE_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
WORKLOADS = {
    # name: (config yaml, T, h, w)
    "ViewCrafter_25_576x1024x25": ("inference_pvd_1024.yaml", 25, 72, 128),
    "ViewCrafter_16_576x1024x16": ("inference_pvd_1024.yaml", 16, 72, 128),
    "ViewCrafter_25_512_320x512x25": ("inference_pvd_512.yaml", 25, 40, 64),
}


def csrc_hash(names=("gemm_dma.hip", "gemm_ws.hip", "gemm.hip", "gemm_args.h", "gemm_epilogue.h")):
    """sha256 over kernel sources (default: the GEMM engine): profiles/pmc_traffic.json records the hashes it was measured on."""
    import hashlib
    hsh = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "viewcrafter_amd", "csrc", name), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def synth_conditioning(T, h, w, device, seed=123, B=1):
    """CPU generator with the reference's default seed, then moved to the device."""
    g = torch.Generator().manual_seed(seed)
    L = 77 + 256          # text tokens + Resampler image tokens (16 queries x video_length 16)
    x_T = torch.randn(B, 4, T, h, w, generator=g)
    cat = torch.randn(B, 4, T, h, w, generator=g) * 0.8
    ctx = torch.randn(B, L, 1024, generator=g)
    uctx = torch.randn(B, L, 1024, generator=g)
    cond = {"c_crossattn": [ctx.to(device)], "c_concat": [cat.to(device)]}
    uc = {"c_crossattn": [uctx.to(device)], "c_concat": [cond["c_concat"][0]]}
    return x_T.to(device), cond, uc


def cpu_baseline(model, hp, dd, flops_per_step_full, budget_s=75.0):
    """The reference's own code on the host cores: `UNetModel.forward` (openaimodel3d.py:548-603) fp32 at BASELINE
    configs[0]'s shapes (16 frames, 40x64 latent, the full 1.44 B-parameter width), after a warm-up call; one DDIM step = 2
    forwards; FLOP-scaled to the bench workload (labelled extrapolated) - and one `AutoencoderKL.decode` of a 40x64 latent
    (320x512 frame).  The code that runs is the reference's, imported from oracle/_ref/ (bytecode compiled from /root/reference by
    oracle/build_ref.py at build time: `"kind": "reference"`); if that tree is missing the oracle restatement is timed instead
    (`"kind": "port"`).  The same 16 frames at 24x40 are timed first; a host too slow for the 40x64 call within `budget_s`
    (predicted from that run) reports those.  Only this function, gpu_legs() below and nothing in the timed region touch oracle/."""
    from oracle import lvdm_oracle as O
    from oracle import ref_runner as R
    unet = model.model.diffusion_model
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in model.first_stage_model.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    ts, fs = torch.tensor([499]), torch.tensor([10])
    kind = "reference" if R.available() else "port"
    if kind == "reference":
        ref_unet = R.reference_unet(hp, sd)
        ref_vae = R.reference_vae(dd, state_dict=vsd)

        def forward(x, ctx):
            return ref_unet(x, ts, context=ctx, fs=fs)

        def decode(z):
            return ref_vae.decode(z)
    else:
        def forward(x, ctx):
            return O.unet_forward(sd, hp, x, ts, ctx, fs)

        def decode(z):
            return O.vae_decode(vsd, dd, z)

    def run(T, h, w):
        x = torch.randn(1, 8, T, h, w, generator=g)
        ctx = torch.randn(1, 77 + 256, 1024, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            y = forward(x, ctx)
        assert torch.isfinite(y).all()
        return time.perf_counter() - t0
    run(2, 8, 8)                                         # threads, allocator, oneDNN primitives
    T, h, w = 16, 24, 40
    dt = run(T, h, w)                                    # 4.6 TFLOP: predicts the 12.6 TFLOP call (FLOP ratio 2.74)
    if dt * UNET_TFLOP[(16, 40, 64)] / UNET_TFLOP[(16, 24, 40)] <= budget_s:
        T, h, w = 16, 40, 64
        dt = run(T, h, w)
    flops = UNET_TFLOP.get((T, h, w))
    step_s = 2.0 * dt
    scale = flops_per_step_full / (2.0 * flops * 1e12) if flops else None
    with torch.no_grad():
        decode(torch.randn(1, 4, 8, 8, generator=g))     # warm-up
        z = torch.randn(1, 4, 40, 64, generator=g) / 0.18215
        t0 = time.perf_counter()
        frame = decode(z)
        dec_s = time.perf_counter() - t0
    assert frame.shape == (1, 3, 320, 512) and torch.isfinite(frame).all()
    what = ("the reference's own UNetModel.forward / AutoencoderKL.decode (oracle/_ref: bytecode of /root/reference's lvdm, compiled at build time)"
            if kind == "reference" else "fp32 oracle restatement (oracle/_ref not built)")
    return dict(value=(1.0 / (step_s * scale)) if scale else None, unit="DDIM steps/s", cores=torch.get_num_threads(), kind=kind,
                sample=f"{what} on the host, full 1.44B-param width, latent {T}x{h}x{w} (BASELINE configs[0] shapes"
                       f"{'' if (h, w) == (40, 64) else ' reduced to fit the time budget'}): {dt:.1f} s/forward = {step_s:.1f} s per DDIM "
                       f"step there ({flops} TFLOP/forward) -> x{scale:.1f} FLOP-scaled to the bench workload (extrapolated); "
                       f"VAE decode of one 320x512 frame (1.564 TFLOP): {dec_s:.2f} s",
                sec_per_step_at_sample=step_s, sample_latent=[T, h, w], vae_decode_s_per_frame=dec_s, vae_decode_frame=[320, 512])


# UNet forward TFLOP of the reference graph by latent (torch.utils.flop_counter on meta tensors; 16x24x40
# counted the same way)
UNET_TFLOP = {(25, 72, 128): 82.761, (16, 72, 128): 52.336, (25, 40, 64): 20.187, (16, 40, 64): 12.603, (16, 24, 40): 4.595}


def gpu_legs(model, hp, dd, T, h, w, device, x, ts, ctx, fs, z_dec):
    """On the MI355X, outside the timed region: (a) parity of the HIP path against the fp32 oracle at the bench's own
    latent (UNet forward; VAE decode of 2 frames), (b) the reference's own UNetModel (oracle/_ref; the oracle graph if that is missing) as PyTorch eager
    ops under fp16 autocast = the un-accelerated same-GPU baseline (hipBLASLt / rocBLAS / MIOpen kernels, vanilla attention, NCHW)."""
    from oracle import lvdm_oracle as O
    unet = model.model.diffusion_model
    sd = {k: v.detach() for k, v in unet.state_dict().items()}
    out = {}
    with torch.no_grad():
        y = unet(x, ts, context=ctx, fs=fs)
        ref = O.unet_forward(sd, hp, x, ts, ctx, fs)
        rel = float((y.double() - ref.double()).norm() / ref.double().norm())
        vsd = {k: v.detach() for k, v in model.first_stage_model.state_dict().items()}
        dec = model.decode_first_stage(z_dec)
        dref = O.decode_first_stage(vsd, dd, z_dec, scale_factor=model.scale_factor)
        drel = float((dec.double() - dref.double()).norm() / dref.double().norm())
        del dec, dref
        out["parity"] = {"unet_forward_rel_l2": rel, "vae_decode_rel_l2": drel, "latent": [T, h, w],
                         "oracle": "fp32 oracle/lvdm_oracle.py run on the same MI355X, same weights and inputs",
                         "bounds": {"unet_forward": 5e-3, "vae_decode": 8e-3}}
        from oracle import ref_runner as R
        if R.available():       # the reference's own UNetModel (oracle/_ref bytecode), sharing the product's fp32 parameter tensors
            with torch.device("meta"):
                ref_unet = R.reference_unet(hp)
            ref_unet.load_state_dict(sd, strict=True, assign=True)

            def eager(xx):
                return ref_unet(xx, ts, context=ctx, fs=fs)
            kind = ("the reference's own UNetModel.forward (oracle/_ref: bytecode of /root/reference's lvdm; fp32 weights, vanilla attention - "
                    "xformers is not in this image) under torch.autocast(fp16) as viewcrafter.py:98 runs it, on this MI355X")
        else:
            def eager(xx):
                return O.unet_forward(sd, hp, xx, ts, ctx, fs)
            kind = "oracle graph as PyTorch-ROCm eager ops under torch.autocast(fp16) on this MI355X (oracle/_ref not built)"
        with torch.autocast("cuda", dtype=torch.float16):
            eager(x)                                                    # warm-up: MIOpen find, hipBLASLt heuristics
            torch.cuda.synchronize()
            dts = []
            for _ in range(3):                                          # best of three: MIOpen / hipBLASLt pick kernels per process and the
                t0 = time.perf_counter()                                # first timed step of a process has been 2-4x the others
                for _ in range(2):                                      # cond + uncond = one DDIM step of the reference
                    ye = eager(x)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt = min(dts)
        erel = float((ye.double() - ref.double()).norm() / ref.double().norm())
        out["gpu_eager_baseline"] = {"value": 1.0 / dt, "unit": "DDIM steps/s", "ms_per_step": 1e3 * dt,
                                     "ms_per_step_all": [round(1e3 * d, 1) for d in dts],
                                     "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE", "unset (library default)"),
                                     "kind": kind + " (2 sequential B=1 forwards, no DDIM update; min of 3 steps - context, not a yardstick: "
                                                    "the kernels MIOpen selects differ from process to process)", "rel_l2_vs_fp32": erel}
    torch.cuda.empty_cache()
    return out


def self_launch(argv, n):
    """`python bench.py --gpus N` (N > 1) without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`,
    one rank per GPU on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: re-executing as {' '.join(cmd[1:8])} bench.py ...", file=sys.stderr)
    sys.stderr.flush()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def per_rank_rates(elapsed_local, steps, device):
    """Every rank's own steps/s (all_gather of the local elapsed time) and the max-over-ranks elapsed time `value` is built on."""
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    allt = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allt, t)
    el = [float(x.item()) for x in allt]
    return max(el), [steps / e for e in el]


def stub_main(args, world, rank):
    """`--stub`: the launch / barrier / max-over-ranks timing / reporting harness on CPU ranks (gloo) with a stand-in step
    function - no model, no GPU, numbers meaningless.  tests/test_entry_cpu.py runs `python bench.py --gpus 2 --stub`
    to cover the self-launch path; the line says "data": "stub" so it can never be mistaken for a measurement."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus, f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks"
    g = torch.Generator().manual_seed(123 + rank)
    x = torch.randn(4, 8, 16, 16, generator=g)

    def one_step(x, i):
        return 0.99 * x + 0.01 * torch.tanh(x.roll(1, -1))

    def sync():
        if world > 1:
            dist.barrier()
    for i in range(args.warmup):
        x = one_step(x, i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        x = one_step(x, i)
    local = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    rates = [args.steps / local]
    if world > 1:
        elapsed, _ = per_rank_rates(elapsed, args.steps, "cpu")
        _, rates = per_rank_rates(local, args.steps, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "DDIM steps/sec (STUB: harness self-test, no model)", "value": world * args.steps / elapsed,
                          "unit": "DDIM steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "stub", "backend": dist.get_backend() if world > 1 else None,
                          "per_rank_steps_per_s": rates, "config": {"workload": "stub", "parallelism": f"trajectory-sharded x{world}"}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the UNet forward as a hipGraph in the timed region "
                    "(HIP-event profiling cannot run inside a graph: the roofline is then measured on extra eager steps)")
    ap.add_argument("--no-profile", action="store_true", help="no per-launch HIP events in the timed region")
    ap.add_argument("--no-gpu-legs", action="store_true", help="skip parity-vs-oracle and the eager fp16 baseline on the GPU")
    ap.add_argument("--no-extra", action="store_true", help="skip the step times of the other two published configurations")
    ap.add_argument("--no-video", action="store_true", help="skip the measured 50-step video after the timed region")
    ap.add_argument("--no-telemetry", action="store_true", help="no clock / power sampling thread beside the timed region")
    ap.add_argument("--no-shared-prefix", action="store_true", help="evaluate cond and uncond as a plain B=2 forward (A/B of the "
                    "shared CFG prefix: the layers ahead of the first cross-attention see identical inputs and run once by default)")
    ap.add_argument("--stub", action="store_true", help="CPU self-test of the launch / timing / reporting harness (gloo ranks, a "
                    "stand-in step function, no model): prints \"data\": \"stub\"; not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)               # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launch line and --gpus must agree "
                         "(a line with the wrong n_gpus is worse than no line)")
    if args.stub:
        return stub_main(args, world, rank)
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X GPU (torch.cuda.device_count() == 0)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ndev >= world:
            dist.init_process_group("nccl", device_id=device)       # RCCL over xGMI, one rank per GPU
        elif os.environ.get("VCX_BENCH_SHARE_GPU") == "1":
            # fewer GPUs than ranks (smoke-testing the launch line on a 1-GPU box): RCCL cannot put two ranks on one device, so
            # the control-plane collectives fall back to gloo; the line is marked and the numbers are meaningless
            print(f"[bench] {world} ranks on {ndev} GPU(s): sharing devices, gloo control plane (smoke test only)", file=sys.stderr)
            dist.init_process_group("gloo")
        else:
            raise SystemExit(f"bench.py --gpus {world} needs {world} GPUs, this node shows {ndev} (VCX_BENCH_SHARE_GPU=1 smoke-tests "
                             "the launch line on fewer)")
        assert dist.get_world_size() == args.gpus, f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks"
        assert dist.get_backend() == "nccl" or os.environ.get("VCX_BENCH_SHARE_GPU") == "1", dist.get_backend()

    from viewcrafter_amd import ops
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    from viewcrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    ops.require_gpu()

    cfg_name, T, h, w = WORKLOADS[args.workload]
    torch.manual_seed(123)
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg_name), device=device, conditioners="identity")
    bcast_s = None
    rccl = world > 1 and dist.get_backend() == "nccl"
    if not rccl or rank == 0:       # (smoke mode on a box with fewer GPUs than ranks: every rank seeds its own identical copy)
        randomize_parameters(model, seed=0)
    if rccl:
        # the product's start-up: rank 0 owns the weights, the others receive them in 256 MB RCCL broadcasts over xGMI
        from viewcrafter_amd import parallel
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        parallel.broadcast_module_(model, src=0)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_s = time.perf_counter() - t0
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        assert all(torch.equal(allc[0], c) for c in allc), f"rank weights differ after the broadcast: {[float(c) for c in allc]}"
    # independent trajectory per rank (batch-sharded: no data-path collective)
    x, cond, uc = synth_conditioning(T, h, w, device, seed=123 + rank)
    fs = torch.tensor([10], device=device)
    sampler = DDIMSampler(model)
    sampler.make_schedule(ddim_num_steps=50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    sampler._cfg_cache = None
    sampler.share_cfg_prefix = not args.no_shared_prefix
    n_sched = len(sampler.ddim_timesteps)
    x_start = x.clone()

    def one_step(x, i):
        # step i of back-to-back 50-step trajectories: a request for more than 50 steps starts the next trajectory from the same x_T
        if i and i % n_sched == 0:
            x = x_start.clone()
        index = n_sched - 1 - (i % n_sched)
        ts = torch.full((1,), int(sampler.ddim_timesteps[index]), device=device, dtype=torch.long)
        x, _ = sampler.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=7.5,
                                     unconditional_conditioning=uc, fs=fs, guidance_rescale=0.7,
                                     cfg_img=None, unconditional_conditioning_img_nonetext=None)
        return x

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    unet = model.model.diffusion_model
    unet.use_hip_graph = bool(args.graph)
    profile_in_region = not (args.graph or args.no_profile)
    # graphics clock / socket power sampled every 50 ms by a side thread from the warm-up on (tools/telemetry.py: amdsmi gpu_metrics or
    # sysfs hwmon); the statistics of the samples that fall inside the timed region go into the line as `telemetry`
    from tools.telemetry import Telemetry
    tm = Telemetry(device_index=dev_index, period_s=0.05) if rank == 0 and not args.no_telemetry else None
    if tm is not None:
        tm.__enter__()
    with torch.no_grad():
        for i in range(args.warmup):
            x = one_step(x, i)
        sync()
        if profile_in_region:
            ops.profile_begin(max(1 << 16, 1500 * args.steps))      # ~900 family runs per step: never exhaust the event pool
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            x = one_step(x, i)
        sync()
        elapsed = time.perf_counter() - t0
        if tm is not None:
            tm.__exit__(None, None, None)
        if profile_in_region:
            prof = ops.profile_end()
        else:   # per-family HIP-event timing on extra eager steps of the same loop (not part of `value`)
            unet.use_hip_graph = False
            n_extra = 2
            ops.profile_begin(1 << 16)
            xe = x
            for i in range(args.warmup + args.steps, args.warmup + args.steps + n_extra):
                xe = one_step(xe, i)
            torch.cuda.synchronize()
            prof = ops.profile_end()
            for v in prof.values():      # normalise to the timed region's step count
                for k in ("launches", "ms", "flops", "bytes"):
                    v[k] = v[k] * args.steps / max(n_extra, 1)
    assert torch.isfinite(x).all(), "non-finite latent after the timed steps"
    rank_rates = [args.steps / elapsed]
    if world > 1:
        elapsed, rank_rates = per_rank_rates(elapsed, args.steps, device if dist.get_backend() == "nccl" else "cpu")

    gather_s = None
    if world > 1 and not args.no_decode:
        # the product's end: every rank decodes its own trajectory, rank 0 receives all clips in one all_gather (uint8 frames)
        from viewcrafter_amd import parallel
        with torch.no_grad():
            clip = model.decode_first_stage(x)
        clip = ((clip[0].permute(1, 2, 3, 0).clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8)
        if dist.get_backend() != "nccl":
            clip = clip.cpu()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        clips = parallel.gather_results({rank: clip}, world)
        torch.cuda.synchronize()
        dist.barrier()
        gather_s = time.perf_counter() - t0
        if rank == 0:
            assert len(clips) == world and all(c.shape == clip.shape for c in clips)
        del clips, clip
    steps_per_s = world * args.steps / elapsed
    out = {
        "metric": "DDIM steps/sec (576x1024x25f latent 25x72x128, CFG 7.5, 50-step schedule)" if "576x1024x25" in args.workload
        else f"DDIM steps/sec ({args.workload})",
        "value": steps_per_s, "unit": "DDIM steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic", "backend": (dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else
                                                                             " (SHARED-GPU SMOKE TEST: not a measurement)")) if world > 1 else None,
        "per_rank_steps_per_s": rank_rates, "tune": ops.tune_report(), "launch_mode": "hipGraph replay" if args.graph else "eager",
        "roofline_measured_on": "the timed region" if profile_in_region else "extra eager steps after the timed region",
        "config": {"workload": args.workload, "trajectories_per_gpu": 1, "frames": T, "latent": [T, h, w],
                   "guidance": "CFG 7.5 + rescale 0.7, cond/uncond batched as B=2" + ("" if args.no_shared_prefix else
                               "; layers ahead of the first cross-attention (identical inputs in both evaluations) computed once"), "eta": 1.0,
                   "parallelism": f"trajectory-sharded x{world} (no in-step collective)"},
    }
    if rank == 0:
        if tm is not None:
            out["telemetry"] = tm.summary(t0, t0 + elapsed)
            out["telemetry"]["scope"] = "rank 0's GPU, samples inside the timed region"
        gemm = prof["gemm"]
        flops_per_step = sum(v["flops"] for v in prof.values()) / args.steps
        ach = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12 if gemm["ms"] > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MI355X_FP16_DENSE_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / MI355X_FP16_DENSE_TFLOPS, "traffic": None,
                           "kernel": "gemm_dma_kernel<TileCfg,CONV,GEGLU,OUT_F32> (csrc/gemm_dma.hip) + gemm_ws320_pipe / _geglu / _lnf kernels for the K = 320 layers of level 0 "
                                     "(csrc/gemm_ws.hip); <0.5% of FLOPs on the register-staged gemm_kernel fallback",
                           "launches_per_step": gemm["launches"] / args.steps,
                           "avg_launch_ms": gemm["ms"] / max(gemm["launches"], 1),
                           "algorithmic_tflop_per_launch_avg": gemm["flops"] / max(gemm["launches"], 1) / 1e12,
                           # context, not the peak `frac` is priced against: what the matrix pipes deliver on this part with nothing else running,
                           # measured (tools/mfma_shape_probe.py, profiles/r06z_mfma_shape_probe.txt: the engine's wave-tile K-step from registers
                           # only, 16x16x32 MFMAs, two waves per SIMD on all CUs; the SMU holds ~1.98 GHz at ~1.28 kW)
                           "peak_sustained_under_power_cap": {"tflops": 1981.0, "sclk_mhz": 1980.0, "power_w": 1283.0,
                                                              "frac_of_it": ach / 1981.0, "source": "profiles/r06z_mfma_shape_probe.txt"}}
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (bench.py cannot run rocprofv3 itself).  The
        # file records a hash of the GEMM sources it was measured on: a number measured on other kernels is not reported.
        tr = {}
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            now = csrc_hash()
            if args.workload != tr.get("workload", "ViewCrafter_25_576x1024x25"):
                out["roofline"]["traffic_note"] = f"PMC passes were run on {tr.get('workload', 'ViewCrafter_25_576x1024x25')}, not on this workload"
            elif tr.get("csrc_sha256") == now:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
                out["roofline"]["traffic_measured_at"] = {"csrc_sha256": now[:12], "commit": tr.get("commit")}
            else:
                print(f"[bench] profiles/pmc_traffic.json is STALE (measured on csrc {str(tr.get('csrc_sha256'))[:12]}, kernels are "
                      f"now {now[:12]}): roofline.traffic = null; re-run tools/pmc_passes.sh", file=sys.stderr)
                out["roofline"]["traffic_note"] = "stale PMC file: kernels changed since it was measured"
        except Exception as e:
            print(f"[bench] no usable profiles/pmc_traffic.json: {e}", file=sys.stderr)
        fl = prof["flash_attn"]
        fach = fl["flops"] / (fl["ms"] * 1e-3) / 1e12 if fl["ms"] > 0 else 0.0
        ftraffic = None       # same rule as above: only a number measured on the attention kernels as they are now, on this workload
        try:
            ff = tr["families"]["flash_attn"]
            if args.workload == tr.get("workload") and tr.get("attention_sha256") == csrc_hash(("attention.hip", "attention_v2.hip")):
                ftraffic = ff["hbm_bytes_per_launch"]
        except Exception:
            pass
        out["roofline_flash"] = {"bound": "mfma", "achieved": fach, "peak": MI355X_FP16_DENSE_TFLOPS, "unit": "TFLOP/s",
                                 "frac": fach / MI355X_FP16_DENSE_TFLOPS, "traffic": ftraffic,
                                 "kernel": "flash2_d64_kernel (csrc/attention_v2.hip, 9216-key self-attention) / flash_d64_kernel<2,...> / xattn_resident2_d64_kernel (csrc/attention.hip); FLOP = 4 N_q N_k d per head",
                                 "launches_per_step": fl["launches"] / args.steps, "avg_launch_ms": fl["ms"] / max(fl["launches"], 1),
                                 "algorithmic_tflop_per_launch_avg": fl["flops"] / max(fl["launches"], 1) / 1e12,
                                 "algorithmic_bytes_per_launch_avg": fl["bytes"] / max(fl["launches"], 1)}
        # the five GEMM problems furthest above their floor, from the committed per-shape table (tools/gemm_shapes.py --json; timed in
        # isolation on an MI355X) - reported only when it was measured on the GEMM sources as they are now
        try:
            gs = json.load(open(os.path.join(ROOT, "profiles", "gemm_shapes.json")))
            if gs.get("csrc_sha256") == csrc_hash() and gs.get("workload") == args.workload:
                out["gemm_top_excess"] = {"source": "profiles/gemm_shapes.json (tools/gemm_shapes.py)", "floor": gs["floor"], "sum_excess_ms": gs["excess_ms"],
                                          "isolated_total_ms": gs["total_ms"], "shapes": gs["top_excess"]}
        except Exception:
            pass
        out["kernel_families"] = {k: {"ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps,
                                      "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                                      "gbps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                                  for k, v in prof.items()}
        out["algorithmic_tflop_per_step"] = flops_per_step / 1e12
        out["whole_step_tflops"] = flops_per_step * steps_per_s / world / 1e12
        if not args.no_decode:
            with torch.no_grad():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                frames = model.decode_first_stage(x)
                torch.cuda.synchronize()
                dec_s = time.perf_counter() - t0          # includes first-call weight packing
                t0 = time.perf_counter()
                frames = model.decode_first_stage(x)
                torch.cuda.synchronize()
                dec_s = time.perf_counter() - t0
            assert frames.shape == (1, 3, T, 8 * h, 8 * w)
            out["vae_decode_s_per_video"] = dec_s
            out["sec_per_video_est"] = 50.0 / (steps_per_s / world) + dec_s
            out["reference_published"] = "120 s/video on A100-40G (README.md:117-119), scope of that timer unstated"
            del frames
            if world == 1 and not args.no_video:
                # BASELINE.json's metric is steps/s AND sec/video: one real 50-step sample() (eta = 1, CFG 7.5, rescale 0.7) + the
                # 25-frame decode, measured; the once-per-video conditioners (CLIP towers, Resampler, VAE encode: 0.15 s in
                # tools/video_e2e.py) are fed as synthetic tensors here and are not in this number
                with torch.no_grad():
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    smp, _ = sampler.sample(S=50, conditioning=cond, batch_size=1, shape=[4, T, h, w], verbose=False,
                                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, cfg_img=None,
                                            mask=None, x0=None, fs=fs, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                            unconditional_conditioning_img_nonetext=None)
                    vid = model.decode_first_stage(smp)
                    torch.cuda.synchronize()
                    out["sec_per_video"] = time.perf_counter() - t0
                assert torch.isfinite(vid).all() and vid.shape == (1, 3, T, 8 * h, 8 * w)
                out["sec_per_video_scope"] = "measured: DDIMSampler.sample(S=50) + decode_first_stage of all frames, one trajectory on one GPU"
                del vid, smp
        if world == 1 and not args.no_extra:
            # BASELINE.md publishes three s/video figures (120 / 75 / 50 s on A100-40G); the other two configurations run on the
            # SAME UNet (the two YAMLs differ in image_size / base_scale only), so their step times are measured here too:
            # 1 warm-up + 3 timed DDIM steps each, no profiling events, not part of `value`
            published = {"ViewCrafter_25_576x1024x25": 120.0, "ViewCrafter_16_576x1024x16": 75.0, "ViewCrafter_25_512_320x512x25": 50.0}
            out["extra"] = {}
            for name, (_, T2, h2, w2) in WORKLOADS.items():
                if name == args.workload:
                    continue
                x2, cond2, uc2 = synth_conditioning(T2, h2, w2, device, seed=123)
                sampler._cfg_cache = None

                def step2(xx, i):
                    index = n_sched - 1 - i
                    ts = torch.full((1,), int(sampler.ddim_timesteps[index]), device=device, dtype=torch.long)
                    xx, _ = sampler.p_sample_ddim(xx, cond2, ts, index=index, unconditional_guidance_scale=7.5, unconditional_conditioning=uc2,
                                                  fs=fs, guidance_rescale=0.7, cfg_img=None, unconditional_conditioning_img_nonetext=None)
                    return xx
                with torch.no_grad():
                    x2 = step2(x2, 0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(1, 4):
                        x2 = step2(x2, i)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 3
                assert torch.isfinite(x2).all()
                out["extra"][name] = {"ms_per_step": 1e3 * dt, "steps_per_s": 1.0 / dt, "latent": [T2, h2, w2], "steps_timed": 3,
                                      "sec_per_50_steps": 50 * dt, "reference_published_sec_per_video_a100": published[name]}
                del x2, cond2, uc2
            sampler._cfg_cache = None
            # Two clips per GPU on two HIP streams (viewcrafter_amd/interleave.py; what ViewCrafter.run_diffusion_many does with a rank's
            # clips under VCX_CLIPS_PER_GPU=2 - opt-in since round 6, the default is one clip after the other): 2 x 3 DDIM steps of the
            # headline workload one clip after the other, then interleaved step by step - aggregate rate of both clips, with the clock /
            # power telemetry of each leg.  `value` above is ONE trajectory per GPU.
            from viewcrafter_amd.interleave import run_interleaved, step_yield
            xa, conda, uca = synth_conditioning(T, h, w, device, seed=123)
            xb = torch.randn_like(xa)

            def clip(xx, index, nsteps=3):
                smp = DDIMSampler(model)
                smp.make_schedule(ddim_num_steps=50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
                smp.share_cfg_prefix = not args.no_shared_prefix
                for i in range(nsteps):
                    idx = n_sched - 1 - i
                    ts = torch.full((1,), int(smp.ddim_timesteps[idx]), device=device, dtype=torch.long)
                    xx, _ = smp.p_sample_ddim(xx, conda, ts, index=idx, unconditional_guidance_scale=7.5, unconditional_conditioning=uca, fs=fs,
                                              guidance_rescale=0.7, cfg_img=None, unconditional_conditioning_img_nonetext=None)
                    step_yield()
                return xx
            with torch.no_grad():
                res = {}
                tms = {}
                for lanes in (1, 2, 1, 2):
                    torch.cuda.synchronize()
                    tl = Telemetry(device_index=dev_index, period_s=0.05) if tm is not None else None
                    if tl is not None:
                        tl.__enter__()
                    t0l = time.perf_counter()
                    outs2 = run_interleaved(clip, [(0, xa), (1, xb)], n_lanes=lanes)
                    torch.cuda.synchronize()
                    res.setdefault(lanes, []).append(time.perf_counter() - t0l)
                    if tl is not None:
                        tl.__exit__(None, None, None)
                        sm = tl.summary()
                        tms.setdefault(lanes, []).append({k: (sm.get(k) or {}).get("mean") for k in ("sclk_mhz", "power_w")})
                    assert all(torch.isfinite(o).all() for o in outs2)
            seq, two = min(res[1]), min(res[2])
            out["extra"]["two_clips_per_gpu"] = {
                "workload": args.workload, "ddim_steps_per_clip": 3, "clips": 2,
                "one_after_the_other_steps_per_s": 6 / seq, "two_streams_steps_per_s": 6 / two, "gain": seq / two,
                "default": "one after the other (VCX_CLIPS_PER_GPU=1); the two-stream mode is opt-in: its sign depends on the box "
                           "(builder's boxes +3 ... +9 %, the driver's box of round 5 -8.7 %)",
                "telemetry_mean": {"one_after_the_other": tms.get(1), "two_streams": tms.get(2)},
                "note": "aggregate DDIM steps/s of two independent trajectories on one GPU; interleaved step by step on two HIP streams "
                        "(bit-identical outputs: tests/test_entry_gpu.py::test_two_clips_per_gpu_on_two_streams_equal_the_plain_loop); "
                        "the headline `value` stays one trajectory per GPU"}
            del xa, xb
        from viewcrafter_amd.config import load_yaml
        mp_ = load_yaml(os.path.join(ROOT, "configs", cfg_name))["model"]["params"]
        hp = dict(mp_["unet_config"]["params"])
        if world == 1 and not args.no_gpu_legs:      # N = 1 only: the other ranks would sit in the closing barrier meanwhile
            g = torch.Generator().manual_seed(99)
            xg = torch.randn(1, 8, T, h, w, generator=g).to(device)
            cg = torch.randn(1, 77 + 256, 1024, generator=g).to(device)
            zg = torch.randn(1, 4, 2, h, w, generator=g).to(device)
            out.update(gpu_legs(model, hp, dict(mp_["first_stage_config"]["params"]["ddconfig"]), T, h, w, device, xg,
                                torch.tensor([499], device=device), cg, fs, zg))
        if bcast_s is not None:
            out["broadcast_s"] = bcast_s
        if gather_s is not None:
            out["gather_s"] = gather_s
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, hp, dict(mp_["first_stage_config"]["params"]["ddconfig"]), flops_per_step)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
