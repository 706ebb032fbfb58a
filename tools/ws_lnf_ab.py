"""LayerNorm-folded projections of level 0 (460800 x {960, 640} x 320, VCX_GEMM_LNFOLD): the weight-stationary kernel (knob GEMM_WS = 1)
against the tiled engine (GEMM_WS = 4: everything weight-stationary but these), interleaved in one process; other N for the dispatch rule."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
if len(sys.argv) > 1:          # a timing-only build (tools/build_abl.sh wlabl1 -DVCX_WL_ABL=1 ...): results are garbage, only N = 960
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from viewcrafter_amd import ops
print("library", _lib.LIB_PATH)
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
C = 320
for M, N in ([(460800, 960), (460800, 512)] if len(sys.argv) > 1 else [(460800, 960), (460800, 640), (230400, 960), (460800, 512), (460800, 1280), (460800, 1920)]):
    x = torch.randn(M, C, device="cuda").half()
    w = (torch.randn(N, C, device="cuda") / math.sqrt(C)).half()
    bias, colsum = torch.randn(N, device="cuda"), 0.01 * torch.randn(N, device="cuda")
    st = ops.row_stats(x, 1e-5)
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    res = {1: [], 4: []}
    for r in range(5):
        for ws in (1, 4):
            ops.tune_set("GEMM_WS", ws)
            res[ws].append(t(lambda: ops.gemm(x, w, M=M, N=N, K=C, lda=C, out=out, ldc=N, bias=bias, ln_stats=st, ln_colsum=colsum)))
    ops.tune_set("GEMM_WS", 1)
    fl, by = 2.0 * M * N * C, 2.0 * M * (N + C)
    print(f"LNFOLD {M} x {N} x {C}: weight-stationary {sorted(res[1])[2]:.3f} ms ({fl / sorted(res[1])[2] / 1e9:.0f} TFLOP/s, {by / sorted(res[1])[2] / 1e9:.2f} TB/s, min {min(res[1]):.3f})   "
          f"tiled {sorted(res[4])[2]:.3f} ms ({fl / sorted(res[4])[2] / 1e9:.0f} TFLOP/s, min {min(res[4]):.3f})", flush=True)
