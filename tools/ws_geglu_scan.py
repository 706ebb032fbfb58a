import math, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from viewcrafter_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_geglu
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
M, C = 460800, 320
x = torch.randn(M, C, device="cuda").half()
print('library', _lib.LIB_PATH)
for N in ((2560,) if len(sys.argv) > 1 else (256, 512, 1280, 2560, 5120)):
    wp, bp = pack_geglu(torch.randn(N, C, device="cuda") / math.sqrt(C), torch.randn(N, device="cuda"))
    wp = wp.half(); bp = bp.float().contiguous()
    out = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
    r = {}
    for ws in (1, 2):
        ops.tune_set("GEMM_WS", ws)
        r[ws] = min(t(lambda: ops.gemm(x, wp, M=M, N=N, K=C, lda=C, out=out, ldc=N // 2, bias=bp, geglu=True)) for _ in range(3))
    ops.tune_set("GEMM_WS", 1)
    print(f"N={N:5d}: weight-stationary {r[1]:.3f} ms  tiled {r[2]:.3f} ms   per 256 columns: {r[1] / (N / 256) * 1e3:.1f} / {r[2] / (N / 256) * 1e3:.1f} us", flush=True)
