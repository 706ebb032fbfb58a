"""What do GroupNorm column moments in the convolution epilogue (VCX_GEMM_COLSTATS) cost the convolutions?  Every 3x3 / (3,1,1)
convolution shape of one DDIM step at 576x1024x25 (launch counts from profiles/r02_gemm_shapes.txt) with and without `colstats=`,
interleaved in one process.  (profiles/r03s_gn_epilogue_ablate.txt was taken with the timing-only precursor of the feature: plain
sums without the robust shift, built -DVCX_GN_EPI_ABLATION into a side library; the shipped epilogue replaced that code.)
  python tools/gn_epilogue_ablate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import ops  # noqa: E402
from viewcrafter_amd.packing import pack_conv  # noqa: E402

DEV = "cuda"
# (kind, frames, H, W, Cin, Cout, launches per step);  3x1 = temporal (3,1,1) convolution over [B, T, H*W, C]
SHAPES = [("3x3", 50, 72, 128, 320, 320, 4), ("3x3", 50, 72, 128, 640, 320, 2), ("3x3", 50, 72, 128, 960, 320, 1),
          ("3x3", 50, 36, 64, 640, 640, 5), ("3x3", 50, 36, 64, 1280, 640, 1), ("3x3", 50, 36, 64, 1920, 640, 1), ("3x3", 50, 36, 64, 960, 640, 1),
          ("3x3", 50, 18, 32, 1280, 1280, 5), ("3x3", 50, 18, 32, 2560, 1280, 2), ("3x3", 50, 18, 32, 1920, 1280, 1),
          ("3x3", 50, 9, 16, 1280, 1280, 11), ("3x3", 50, 9, 16, 2560, 1280, 3),
          ("3x1", 2, 72, 128, 320, 320, 16), ("3x1", 2, 36, 64, 640, 640, 20), ("3x1", 2, 18, 32, 1280, 1280, 20), ("3x1", 2, 9, 16, 1280, 1280, 28)]


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    tot = [0.0, 0.0]
    for kind, n, H, W, cin, cout, count in SHAPES:
        if kind == "3x3":
            x = torch.randn(n, H, W, cin, device=DEV).half()
            w = pack_conv(torch.randn(cout, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)).half()
            b = torch.randn(cout, device=DEV) * 0.1
            cs = ops.colstats_buffer(n * H * W, cout, DEV)
            run = lambda k=0: ops.conv2d(x, w, b, kh=3, kw=3, colstats=cs if k else None)
        else:
            x = torch.randn(n, 25, H * W, cin, device=DEV).half()
            w = pack_conv(torch.randn(cout, cin, 3, 1, 1, device=DEV) / (3 * cin) ** 0.5).half()
            b = torch.randn(cout, device=DEV) * 0.1
            cs = ops.colstats_buffer(n * 25 * H * W, cout, DEV)
            run = lambda k=0: ops.temporal_conv3(x, w, b, colstats=cs if k else None)
        res = {0: [], 1: []}
        for rep in range(3):
            for k in (0, 1):
                run(k)
                torch.cuda.synchronize()
                res[k].append(timed(lambda: run(k), 6))
        t0, t1 = sorted(res[0])[1], sorted(res[1])[1]
        tot[0] += t0 * count
        tot[1] += t1 * count
        print(f"conv{kind} {n}x{H}x{W} {cin:4d}->{cout:4d} x{count:2d}: shipped {t0:.3f} ms  with column moments {t1:.3f} ({(t1 / t0 - 1) * 100:+.1f} %)", flush=True)
    print(f"sum over the step's launches of these shapes: {tot[0]:.2f} ms -> {tot[1]:.2f} ms (+{tot[1] - tot[0]:.2f} ms)")


main()
