"""Does reading a just-written tensor back to front hit the memory-side cache?  producer (LayerNorm, writes X front to back)
followed by a GEMM over X with its M tiles walked forward (VCX_GEMM_TUNE=0) or backward (=32)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
dev = "cuda"
def rh(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).half()
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for M, C in ((460800, 320), (115200, 640), (230400, 320)):
    x = rh(M, C); g = torch.ones(C, device=dev); b0 = torch.zeros(C, device=dev)
    w = rh(C, C, sc=1 / math.sqrt(C)); bias = torch.randn(C, device=dev)
    ln = lambda: ops.layer_norm(x, g, b0)
    y = ln()
    t_ln = timeit(ln)
    t_g = timeit(lambda: ops.linear(y, w, bias))
    t_pair = timeit(lambda: ops.linear(ops.layer_norm(x, g, b0), w, bias))
    print(f"M={M} C={C} ({M*C*2/1e6:.0f} MB): LN {t_ln:.3f} ms, GEMM alone {t_g:.3f} ms, LN->GEMM {t_pair:.3f} ms (sum {t_ln+t_g:.3f})")
