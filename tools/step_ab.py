"""Same-box, same-process A/B of one DDIM step (bench.py's own step: shared CFG prefix, DDIM update) under host-side switches and library
knobs, variants interleaved round by round.  Prints ms per step and the per-family HIP-event times of every round.

    python tools/step_ab.py [--rounds 3] [--steps 3] [--lib other/libvcx.so] [--workload ...] variant [variant ...]

A variant is  name:key=value,key=value  with keys
    gnfold   0 | 1   TemporalTransformer.norm folded into proj_in (viewcrafter_amd/lvdm/modules/attention.py GN_FOLD)
    gnspatial 0 | 1  ... and SpatialTransformer.norm at level 0 (GN_FOLD_SPATIAL)
    lnff     0 | 1 | 2   LayerNorm folded into the GEGLU projection (FOLD_LAYERNORM_FF); 2 = only at C >= 640 (levels 1-3)
    xattn    1 | 2   resident cross-attention kernel: second form | first form (knob XATTN_RESIDENT)
    ws       1 | 0   weight-stationary K = 320 kernel (knob GEMM_WS)
    lnrs     1 | 0   LayerNorm statistics from the producing layer's epilogue (VCX_GEMM_ROWSTATS; ops.LN_ROWSTATS)
e.g.   base:gnfold=0,xattn=2  gnfold:gnfold=1,xattn=2  xattn2:gnfold=0,xattn=1  all:gnfold=1,xattn=1
--lib runs everything on another build of the library of the SAME ABI (e.g. tools/_abl/libvcx_gelu_select.so, built by
tools/build_abl.sh): a library-level change is then compared across two invocations on the same box."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25")
    args = ap.parse_args()
    from viewcrafter_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from bench import WORKLOADS, synth_conditioning
    from viewcrafter_amd import ops
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    from viewcrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    from viewcrafter_amd.lvdm.modules import attention
    ops.require_gpu()
    print(f"library: {_lib.LIB_PATH}", flush=True)
    cfg_name, T, h, w = WORKLOADS[args.workload]
    torch.manual_seed(123)
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg_name), device="cuda", conditioners="identity")
    randomize_parameters(model, seed=0)
    x0, cond, uc = synth_conditioning(T, h, w, "cuda", seed=123)
    fs = torch.tensor([10], device="cuda")
    sampler = DDIMSampler(model)
    sampler.make_schedule(ddim_num_steps=50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    sampler._cfg_cache = None
    sampler.share_cfg_prefix = True
    n_sched = len(sampler.ddim_timesteps)

    def one_step(x, i):
        index = n_sched - 1 - (i % n_sched)
        ts = torch.full((1,), int(sampler.ddim_timesteps[index]), device="cuda", dtype=torch.long)
        x, _ = sampler.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, fs=fs,
                                     guidance_rescale=0.7, cfg_img=None, unconditional_conditioning_img_nonetext=None)
        return x

    variants = []
    for v in args.variants:
        name, _, kv = v.partition(":")
        variants.append((name, dict(item.split("=") for item in kv.split(",") if item)))

    def apply(settings):
        attention.GN_FOLD = settings.get("gnfold", "1") != "0"
        attention.GN_FOLD_SPATIAL = settings.get("gnspatial", "1") != "0"
        lnff, lnff_min = settings.get("lnff", "2") != "0", (640 if settings.get("lnff", "2") == "2" else 0)      # 2 (the product): levels 1-3 only (C >= 640)
        if lnff != attention.FOLD_LAYERNORM_FF or lnff_min != attention.FOLD_LAYERNORM_FF_MIN_DIM:
            attention.FOLD_LAYERNORM_FF, attention.FOLD_LAYERNORM_FF_MIN_DIM = lnff, lnff_min
            for m in model.modules():
                if isinstance(m, attention.FeedForward):
                    m._drop_packed()
        ops.tune_set("XATTN_RESIDENT", int(settings.get("xattn", "1")))
        ops.tune_set("GEMM_WS", int(settings.get("ws", "1")))
        ops.LN_ROWSTATS = settings.get("lnrs", "1") != "0"

    def run(settings, steps, profile):
        apply(settings)
        torch.manual_seed(7)                         # the eta = 1 noise of every run is the same draw
        x = x0.clone()
        with torch.no_grad():
            x = one_step(x, 0)                       # packs / caches of this variant
            torch.cuda.synchronize()
            if profile:
                ops.profile_begin(1 << 16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(1, 1 + steps):
                x = one_step(x, i)
            e1.record()
            torch.cuda.synchronize()
            prof = ops.profile_end() if profile else None
        assert torch.isfinite(x).all()
        return e0.elapsed_time(e1) / steps, prof, x

    finals = {}
    for name, st in variants:                        # warm-up of every variant
        run(st, 1, False)
    for r in range(args.rounds):
        for name, st in variants:
            ms, prof, x = run(st, args.steps, True)
            fam = {k: round(v["ms"] / args.steps, 2) for k, v in prof.items()}
            print(f"round {r} {name:10s} {ms:8.2f} ms/step  {fam}  launches {sum(v['launches'] for v in prof.values()) // args.steps}", flush=True)
            finals[name] = x
    names = list(finals)
    for n in names[1:]:
        a, b = finals[names[0]].float(), finals[n].float()
        print(f"latent after {1 + args.steps} steps: {n} vs {names[0]}: rel-L2 {float((a - b).norm() / b.norm()):.3e}")


if __name__ == "__main__":
    main()
