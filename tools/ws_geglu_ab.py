"""GEGLU projection of level 0 (460800 x 2560 x 320): the weight-stationary kernel (knob GEMM_WS = 1) against the tiled engine (GEMM_WS = 2:
everything weight-stationary but GEGLU), interleaved in one process."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_geglu
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for M, C in [(460800, 320), (230400, 320)]:
    x = torch.randn(M, C, device="cuda").half()
    wp, bp = pack_geglu(torch.randn(8 * C, C, device="cuda") / math.sqrt(C), torch.randn(8 * C, device="cuda"))
    wp = wp.half(); bp = bp.float().contiguous()
    out = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    res = {1: [], 2: []}
    for r in range(5):
        for ws in (1, 2):
            ops.tune_set("GEMM_WS", ws)
            res[ws].append(t(lambda: ops.gemm(x, wp, M=M, N=8 * C, K=C, lda=C, out=out, ldc=4 * C, bias=bp, geglu=True)))
    ops.tune_set("GEMM_WS", 1)
    fl = 2.0 * M * 8 * C * C
    print(f"GEGLU {M} x {8*C} x {C}: weight-stationary {sorted(res[1])[2]:.3f} ms ({fl / sorted(res[1])[2] / 1e9:.0f} TFLOP/s, min {min(res[1]):.3f})   "
          f"tiled {sorted(res[2])[2]:.3f} ms ({fl / sorted(res[2])[2] / 1e9:.0f} TFLOP/s, min {min(res[2]):.3f})", flush=True)
