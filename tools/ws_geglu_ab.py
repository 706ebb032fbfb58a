"""GEGLU projection of level 0 (460800 x 2560 x 320): the weight-stationary kernel (knob GEMM_WS = 1; 3 = without the cross-XCD streams on the
CUs that 32 / tiles_n leaves over: 240 instead of 250 blocks) against the tiled engine (GEMM_WS = 2: everything weight-stationary but GEGLU),
interleaved in one process; also N = 1280 (five column blocks: 255 / 240 blocks)."""
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
for M, C, NM in [(460800, 320, 8), (230400, 320, 8), (460800, 320, 4)]:
    x = torch.randn(M, C, device="cuda").half()
    wp, bp = pack_geglu(torch.randn(NM * C, C, device="cuda") / math.sqrt(C), torch.randn(NM * C, device="cuda"))
    wp = wp.half(); bp = bp.float().contiguous()
    out = torch.empty(M, NM * C // 2, device="cuda", dtype=torch.float16)
    res = {1: [], 3: [], 2: []}
    for r in range(5):
        for ws in (1, 3, 2):
            ops.tune_set("GEMM_WS", ws)
            res[ws].append(t(lambda: ops.gemm(x, wp, M=M, N=NM * C, K=C, lda=C, out=out, ldc=NM * C // 2, bias=bp, geglu=True)))
    ops.tune_set("GEMM_WS", 1)
    fl = 2.0 * M * NM * C * C
    print(f"GEGLU {M} x {NM*C} x {C}: weight-stationary {sorted(res[1])[2]:.3f} ms ({fl / sorted(res[1])[2] / 1e9:.0f} TFLOP/s, min {min(res[1]):.3f})   "
          f"one-XCD streams only {sorted(res[3])[2]:.3f} ms (min {min(res[3]):.3f})   "
          f"tiled {sorted(res[2])[2]:.3f} ms ({fl / sorted(res[2])[2] / 1e9:.0f} TFLOP/s, min {min(res[2]):.3f})", flush=True)
