"""GroupNorm apply pass: pixels per block (knob EXP0; needs norm.hip built with -DVCX_EXPERIMENT_KNOBS - the product ignores the knob; 0 = the shipped rule max(128, pixels / 1024)).
The apply pass has no reduction, so its block size is free; the shipped value was inherited from the statistics kernel.
    python tools/gn_apply_ppb.py"""
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


shapes = [(50, 9216, 320, 22), (2, 230400, 320, 34), (50, 2304, 640, 20), (2, 57600, 640, 34), (50, 576, 1280, 20), (2, 14400, 1280, 34),
          (50, 9216, 640, 3), (50, 9216, 960, 1), (50, 2304, 1280, 2), (50, 2304, 1920, 1), (50, 576, 2560, 2)]
variants = [0, 16, 32, 48, 64, 96]
tot = {v: 0.0 for v in variants}
print(f"{'shape (n, pixels, C) x launches/step':44s}" + "".join(f"{('ppb ' + str(v)) if v else 'shipped':>11s}" for v in variants))
for n, pix, C, cnt in shapes:
    x = torch.randn(n, pix, C, device="cuda").half()
    y = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    st = torch.zeros(n, 32, 2, device="cuda"); st[..., 1] = 1.0
    res = {v: [] for v in variants}
    for _ in range(3):
        for v in variants:
            ops.tune_set("EXP0", v)
            res[v].append(t(lambda: ops.group_norm(x, g, b, 1e-5, True, out=y, stats=st)))
    ops.tune_set("EXP0", 0)
    row = f"({n:3d}, {pix:6d}, {C:4d}) x {cnt:2d}  {x.numel() * 4 / 1e6:7.1f} MB moved".ljust(44)
    for v in variants:
        m = sorted(res[v])[1]
        tot[v] += m * cnt
        row += f"{m * 1e3:8.1f} us"
    print(row, flush=True)
print("sum over one step's apply passes (ms):".ljust(44) + "".join(f"{tot[v]:11.2f}" for v in variants))
