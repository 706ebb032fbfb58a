"""LayerNorm / row-statistics kernels: lanes per row (knob EXP1 in this experiment build; 0 = shipped table).  python tools/ln_variants.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops


def t(fn, it=12):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


tot = {}
print(f"{'rows x C (stats launches, LN launches per step)':52s}" + "".join(f"{'EXP1=' + str(v):>22s}" for v in (0, 1, 2)))
for rows, C, n_stats, n_ln in [(460800, 320, 22, 11), (115200, 640, 22, 11), (28800, 1280, 22, 11), (230400, 512, 2, 1)]:
    x = torch.randn(rows, C, device="cuda").half()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    res = {v: ([], []) for v in (0, 1, 2)}
    for _ in range(3):
        for v in (0, 1, 2):
            ops.tune_set("EXP1", v)
            res[v][0].append(t(lambda: ops.row_stats(x, 1e-5)))
            res[v][1].append(t(lambda: ops.layer_norm(x, g, b, 1e-5)))
    ops.tune_set("EXP1", 0)
    row = f"{rows:7d} x {C:5d}  ({n_stats:2d}, {n_ln:2d})".ljust(52)
    for v in (0, 1, 2):
        st, ln = sorted(res[v][0])[1], sorted(res[v][1])[1]
        tot[v] = tot.get(v, 0.0) + st * n_stats + ln * n_ln
        row += f"  stats {st * 1e3:6.1f} LN {ln * 1e3:6.1f} us"
    print(row, flush=True)
print("per step (ms):".ljust(52) + "".join(f"{tot[v]:22.2f}" for v in (0, 1, 2)))
