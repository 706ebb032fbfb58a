"""VCX_GEMM_ROWSTATS priced in isolation: the weight-stationary N = K = 320 layer with and without the row-statistics epilogue against the
statistics pass it replaces (vcx_rowstats_f16), at the benchmark's 460800 rows.   python tools/rowstats_ab.py [iters]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best, tot = min(best, ms), tot + ms
    return tot / 3, best


def main():
    from viewcrafter_amd import ops
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = "cuda"
    for M in (460800, 230400):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, 320, generator=g).to(dev).half()
        w = (torch.randn(320, 320, generator=g) / math.sqrt(320)).to(dev).half()
        b = torch.randn(320, generator=g).to(dev)
        res = torch.randn(M, 320, generator=g).to(dev).half()
        out = torch.empty(M, 320, device=dev, dtype=torch.float16)
        st = torch.empty(M, 2, device=dev)
        units = 50 if M == 460800 else 25
        wn = (torch.randn(units, 320, 320, generator=g) / math.sqrt(320)).to(dev).half()
        bn = torch.randn(units, 320, generator=g).to(dev)
        rows = [
            ("bias+res          ", lambda: ops.linear(x, w, b, residual=res, out=out)),
            ("bias+res +rowstats", lambda: ops.linear(x, w, b, residual=res, out=out, rowstats=st)),
            ("bias              ", lambda: ops.linear(x, w, b, out=out)),
            ("bias +rowstats    ", lambda: ops.linear(x, w, b, out=out, rowstats=st)),
            ("units             ", lambda: ops.gemm_units(x, wn, bn, unit_rows=M // units, out=out)),
            ("units +rowstats   ", lambda: ops.gemm_units(x, wn, bn, unit_rows=M // units, out=out, rowstats=st)),
            ("row_stats pass    ", lambda: ops.row_stats(out)),
        ]
        for name, fn in rows:
            avg, best = timed(fn, iters)
            print(f"M={M} {name} {avg * 1e3:8.1f} us (min {best * 1e3:8.1f})", flush=True)


if __name__ == "__main__":
    main()
