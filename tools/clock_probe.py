"""Graphics clock and socket power per problem class (review r5 item 6): the largest GEMM / convolution problems of a forward (recorded by
tools/gemm_shapes.py) and the 9216-key flash kernel, each looped on its own for `--seconds` while tools/telemetry.py samples the SMU.
Says what "the clock under the matrix kernels" is on this box - and how far each class is from the matrix pipe's rate AT THAT CLOCK
(256 CUs x 4 SIMDs x 1024 FLOP per cycle).
    python tools/clock_probe.py [--top 14] [--seconds 1.2]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def probe(run, seconds):
    from telemetry import Telemetry
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 0
    with Telemetry(period_s=0.05) as tm:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                run()
            n += 16
            torch.cuda.synchronize()
        b.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    s = tm.summary(t0 + 0.3, t1)
    return a.elapsed_time(b) / n, (s.get("sclk_mhz") or {}).get("mean") or float("nan"), (s.get("power_w") or {}).get("mean") or float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=14)
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--workload", default="ViewCrafter_25_576x1024x25")
    args = ap.parse_args()
    import gemm_shapes
    from viewcrafter_amd import ops
    seen = gemm_shapes.record_forward(args.workload)
    rows = []
    for key, cnt in seen.items():
        d = dict(zip(gemm_shapes.FIELDS, key))
        rows.append((2.0 * d["M"] * d["N"] * d["K"] * cnt, key, cnt, d))
    rows.sort(key=lambda r: -r[0])
    print(f"{'cnt':>4} {'M':>8} {'N':>6} {'K':>6} {'kind':>9} {'flags':>5} {'ms':>8} {'TF/s':>6} {'sclk MHz':>9} {'W':>6} {'pipe @ clock':>12}")
    for fl_tot, key, cnt, d in rows[:args.top]:
        run = gemm_shapes.make_problem(key)
        ms, sclk, pw = probe(run, args.seconds)
        fl = 2.0 * d["M"] * d["N"] * d["K"]
        peak = 256 * 4 * 1024 * sclk * 1e6
        kind = f"conv{d['kh']}x{d['kw']}" if d["mode"] == 1 else "units" if d["mode"] == 2 else "linear"
        print(f"{cnt:4d} {d['M']:8d} {d['N']:6d} {d['K']:6d} {kind:>9} {d['flags']:5d} {ms:8.3f} {fl / ms / 1e9:6.0f} {sclk:9.0f} {pw:6.0f} {fl / (ms * 1e-3) / peak:12.2f}", flush=True)
        del run
        torch.cuda.empty_cache()
    # the 9216-key self-attention of level 0
    N, G, heads = 9216, 50, 5
    C = heads * 64
    qk = torch.randn(G * N, 2 * C, device="cuda")
    qk[:, :C] *= 0.125 * ops.LOG2E
    qk = qk.half()
    vt = torch.randn(C, G * N, device="cuda").half()
    o = torch.empty(G * N, C, device="cuda", dtype=torch.float16)
    ms, sclk, pw = probe(lambda: ops.flash_attn(qk, qk[:, C:], vt, o, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C,
                                                ldvt=G * N, ldo=C, scale=0.125, log2_logits=True), args.seconds)
    fl = 4.0 * G * heads * N * N * 64
    print(f"{'flash2 9216 keys, 50 x 5 heads':45s} {ms:8.3f} {fl / ms / 1e9:6.0f} {sclk:9.0f} {pw:6.0f} {fl / (ms * 1e-3) / (256 * 4 * 1024 * sclk * 1e6):12.2f}")
    x = torch.randn(50, 9216, 320, device="cuda").half()
    g, b = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    st = ops.group_norm_stats(x)
    y = torch.empty_like(x)
    ms, sclk, pw = probe(lambda: ops.group_norm(x, g, b, 1e-5, True, out=y, stats=st), args.seconds)
    print(f"{'GroupNorm apply 50 x 9216 x 320':45s} {ms:8.3f} {4.0 * x.numel() / ms / 1e9:6.0f} GB/s {sclk:6.0f} {pw:6.0f}")


if __name__ == "__main__":
    main()
