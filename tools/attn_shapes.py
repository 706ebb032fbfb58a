"""Per-launch picture of the attention kernels of one B=2 UNet forward (shared CFG prefix), timed IN SITU: every call of the three
attention entry points is intercepted, executed, and then repeated on the same (live) operands between two events.

    python tools/attn_shapes.py [--workload ViewCrafter_25_576x1024x25]

Columns: launches of the shape per forward, ms per launch, FLOP rate (4 nq nk d per (group, head), both key sets for the dual kernel),
algorithmic bytes (Q + O once, K / V^T once per kv group) and the rate they imply.
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25")
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    from bench import WORKLOADS, synth_conditioning
    from viewcrafter_amd.builder import build_diffusion_model, randomize_parameters
    cfg, T, h, w = WORKLOADS[args.workload]
    model = build_diffusion_model(os.path.join(ROOT, "configs", cfg), device="cuda", conditioners="identity")
    randomize_parameters(model)
    x, cond, uc = synth_conditioning(T, h, w, "cuda")
    both = {"c_crossattn": [torch.cat([cond["c_crossattn"][0], uc["c_crossattn"][0]], 0)], "c_concat": cond["c_concat"]}
    ts = torch.full((1,), 499, device="cuda", dtype=torch.long)
    fs = torch.tensor([10], device="cuda")
    L = _lib.lib()
    rows = collections.OrderedDict()

    def spy(name, real, describe):
        def call(*a):
            rc = real(*a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                real(*a)
            e1.record()
            torch.cuda.synchronize()
            key = (name,) + describe(a)[0]
            r = rows.setdefault(key, {"n": 0, "ms": 0.0, "flop": describe(a)[1], "bytes": describe(a)[2]})
            r["n"] += 1
            r["ms"] += e0.elapsed_time(e1) / args.reps
            return rc
        return call

    def d_flash(a):      # q k vt o n_groups heads nq nk kv_rows kv_div ldq ldk ldvt ldo scale flags stream
        g, hd, nq, nk, kvr, kvd, flags = a[4], a[5], a[6], a[7], a[8], a[9], a[15]
        fl = 4.0 * g * hd * nq * nk * 64
        by = 2.0 * 64 * hd * (2 * g * nq + 2 * ((g + kvd - 1) // kvd) * nk)
        return (g, hd, nq, nk, kvd, flags), fl, by

    def d_dual(a):       # q k1 vt1 k2 vt2 o n_groups heads nq nk1 kv_rows1 kv_div1 ldk1 ldvt1 nk2 kv_rows2 kv_div2 ...
        g, hd, nq, nk1, kvd1, nk2, kvd2 = a[6], a[7], a[8], a[9], a[11], a[14], a[16]
        fl = 4.0 * g * hd * nq * (nk1 + nk2) * 64
        by = 2.0 * 64 * hd * (2 * g * nq + 2 * ((g + kvd1 - 1) // kvd1) * nk1 + 2 * ((g + kvd2 - 1) // kvd2) * nk2)
        return (g, hd, nq, nk1, kvd1, nk2, kvd2), fl, by

    def d_temporal(a):   # qkv o B T P heads ld k_off v_off ldo scale stream
        B, T_, P, hd = a[2], a[3], a[4], a[5]
        fl = 4.0 * B * P * hd * T_ * T_ * 64
        by = 2.0 * B * T_ * P * hd * 64 * 4
        return (B, T_, P, hd), fl, by

    reals = {n: getattr(L, n) for n in ("vcx_attn_flash_d64_f16", "vcx_attn_flash_dual_d64_f16", "vcx_attn_temporal_d64_f16")}
    with torch.no_grad():
        model.apply_model(x, ts, both, fs=fs, cfg_repeat=2)
        L.vcx_attn_flash_d64_f16 = spy("flash", reals["vcx_attn_flash_d64_f16"], d_flash)
        L.vcx_attn_flash_dual_d64_f16 = spy("dual", reals["vcx_attn_flash_dual_d64_f16"], d_dual)
        L.vcx_attn_temporal_d64_f16 = spy("temporal", reals["vcx_attn_temporal_d64_f16"], d_temporal)
        try:
            model.apply_model(x, ts, both, fs=fs, cfg_repeat=2)
        finally:
            for n, f in reals.items():
                setattr(L, n, f)
    torch.cuda.synchronize()
    tot = collections.Counter()
    print(f"# attention launches of one forward, {args.workload}; in-situ re-execution, {args.reps} repetitions each")
    print(f"{'kernel':>9} {'shape (groups, heads, nq, nk, kv_div, ...)':>46} {'cnt':>4} {'ms':>8} {'total':>8} {'TF/s':>7} {'GB':>7} {'TB/s':>6}")
    for key, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        ms = r["ms"] / r["n"]
        tot[key[0]] += r["ms"]
        print(f"{key[0]:>9} {str(key[1:]):>46} {r['n']:4d} {ms:8.3f} {r['ms']:8.2f} {r['flop']/ms/1e9:7.0f} {r['bytes']/1e9:7.3f} {r['bytes']/ms/1e9:6.2f}")
    print("# totals per forward (ms): " + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
