"""Headline benchmark: DDIM steps/s of the ViewCrafter denoising loop on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W          (N > 1 under torch.distributed.run, one rank per GPU)

Workload (N=1): configs[3] of BASELINE.json — ViewCrafter_25, 576x1024x25 frames (latent 25x72x128), the 1.44 B-parameter
lvdm UNet built from configs/inference_pvd_1024.yaml, one trajectory per GPU.  One step = one DDIM step = two UNet
evaluations (cond + uncond, run as one B=2 forward) + the fused CFG / guidance-rescale / v-prediction / dynamic-rescale /
x_{t-1} update, eta = 1 (fresh noise every step), fp16 storage with fp32 accumulation.  Weights are synthetic (no
checkpoints offline), data is synthetic of the real shapes; inputs are resident in HBM before the timed region.

Printed JSON (rank 0): the driver contract + `roofline` (dominant kernel = the MFMA GEMM/conv engine, measured with HIP
events on the launch stream over the timed region) + `cpu_baseline` (the fp32 oracle = a port of the reference's
algorithm, timed on the host cores on a bounded sample of the same workload and FLOP-scaled) + extras
(`sec_per_video_est` = 50 steps + 25-frame VAE decode, per-family kernel time).
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


def cpu_baseline(model, hp, device, flops_per_step_full, threads=None):
    """Time the fp32 oracle (oracle/lvdm_oracle.py, a port of the reference algorithm) on the host cores for one DDIM
    step (= 2 UNet forwards) at a bounded latent size, check the GPU path against it, and FLOP-scale to the full
    workload.  Only this function touches oracle/ (as checker and as reported baseline)."""
    from oracle import lvdm_oracle as O
    from viewcrafter_amd import ops
    T, h, w = 2, 16, 32
    if threads:
        torch.set_num_threads(threads)
    unet = model.model.diffusion_model
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 8, T, h, w, generator=g)
    ctx = torch.randn(1, 77 + 256, 1024, generator=g)
    ts, fs = torch.tensor([499]), torch.tensor([10])
    with torch.no_grad():
        ops.profile_begin(1 << 14)
        y_gpu = unet(x.to(device), ts.to(device), context=ctx.to(device), fs=fs.to(device))
        torch.cuda.synchronize()
        prof = ops.profile_end()
        flops_small = sum(v["flops"] for v in prof.values())
        sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        O.unet_forward(sd, hp, x[:, :, :1], ts, ctx, fs)          # warm-up (threads, allocator)
        t0 = time.perf_counter()
        y_cpu = O.unet_forward(sd, hp, x, ts, ctx, fs)
        dt = time.perf_counter() - t0
    rel = float((y_gpu.cpu().double() - y_cpu.double()).norm() / y_cpu.double().norm())
    step_s_small = 2.0 * dt                                       # one DDIM step = cond + uncond forward
    scale = flops_per_step_full / (2.0 * flops_small)
    return dict(value=1.0 / (step_s_small * scale), unit="DDIM steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"fp32 oracle UNet forward, full 1.44B-param width, latent {T}x{h}x{w}: {dt:.2f} s/forward "
                       f"({flops_small/1e12:.3f} TFLOP) -> x{scale:.0f} FLOP-scaled to the full workload (extrapolated)",
                gpu_vs_oracle_rel_l2=rel)


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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        else:
            # fewer GPUs than ranks (only happens when the launch line is smoke-tested on a 1-GPU box): RCCL cannot
            # put two ranks on one device, so the control-plane collectives fall back to gloo; the numbers are meaningless
            print(f"[bench] {world} ranks on {ndev} GPU(s): sharing devices, gloo control plane (smoke test only)", file=sys.stderr)
            dist.init_process_group("gloo")
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} rank(s)", file=sys.stderr)

    from viewcrafter_amd import ops
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    from viewcrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    ops.require_gpu()

    cfg_name, T, h, w = WORKLOADS[args.workload]
    torch.manual_seed(123)
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg_name), device=device, conditioners="identity")
    randomize_parameters(model, seed=0)
    # independent trajectory per rank (batch-sharded: no data-path collective)
    x, cond, uc = synth_conditioning(T, h, w, device, seed=123 + rank)
    fs = torch.tensor([10], device=device)
    sampler = DDIMSampler(model)
    sampler.make_schedule(ddim_num_steps=50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    sampler._cfg_cache = None
    n_sched = len(sampler.ddim_timesteps)
    assert args.warmup + args.steps <= n_sched, "warmup + steps must fit the 50-step schedule"

    def one_step(x, i):
        index = n_sched - 1 - i
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
    with torch.no_grad():
        for i in range(args.warmup):
            x = one_step(x, i)
        sync()
        if profile_in_region:
            ops.profile_begin(1 << 16)
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            x = one_step(x, i)
        sync()
        elapsed = time.perf_counter() - t0
        if profile_in_region:
            prof = ops.profile_end()
        else:   # per-family HIP-event timing on extra eager steps of the same loop (not part of `value`)
            unet.use_hip_graph = False
            n_extra = min(2, n_sched - args.warmup - args.steps)
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
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    steps_per_s = world * args.steps / elapsed
    out = {
        "metric": "DDIM steps/sec (576x1024x25f latent 25x72x128, CFG 7.5, 50-step schedule)" if "576x1024x25" in args.workload
        else f"DDIM steps/sec ({args.workload})",
        "value": steps_per_s, "unit": "DDIM steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic", "launch_mode": "hipGraph replay" if args.graph else "eager",
        "roofline_measured_on": "the timed region" if profile_in_region else "extra eager steps after the timed region",
        "config": {"workload": args.workload, "trajectories_per_gpu": 1, "frames": T, "latent": [T, h, w],
                   "guidance": "CFG 7.5 + rescale 0.7, cond/uncond batched as B=2", "eta": 1.0,
                   "parallelism": f"trajectory-sharded x{world} (no in-step collective)"},
    }
    if rank == 0:
        gemm = prof["gemm"]
        flops_per_step = sum(v["flops"] for v in prof.values()) / args.steps
        ach = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12 if gemm["ms"] > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MI355X_FP16_DENSE_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / MI355X_FP16_DENSE_TFLOPS, "traffic": None,
                           "kernel": "gemm_dma_kernel<TileCfg,CONV,GEGLU,OUT_F32> (csrc/gemm_dma.hip; <0.5% of FLOPs on the register-staged gemm_kernel fallback)",
                           "launches_per_step": gemm["launches"] / args.steps,
                           "avg_launch_ms": gemm["ms"] / max(gemm["launches"], 1),
                           "algorithmic_tflop_per_launch_avg": gemm["flops"] / max(gemm["launches"], 1) / 1e12}
        try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (bench.py cannot run rocprofv3 itself)
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
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
        if not args.no_cpu_baseline:
            hp = dict(model.model.diffusion_model_hp) if hasattr(model.model, "diffusion_model_hp") else None
            from viewcrafter_amd.config import load_yaml
            hp = dict(load_yaml(os.path.join(ROOT, "configs", cfg_name))["model"]["params"]["unet_config"]["params"])
            out["cpu_baseline"] = cpu_baseline(model, hp, device, flops_per_step)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
