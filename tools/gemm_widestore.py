"""A/B helper: linear layers with N = 1280 on the 256x256 tile (VCX_GEMM_CFG=2), small and large K, with / without residual."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
dev = "cuda"
def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
def rh(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).half()
M, N = 115200, 1280
for K in (320, 1280):
    x = rh(M, K); w = rh(N, K, sc=1 / math.sqrt(K)); b = torch.randn(N, device=dev); r = rh(M, N)
    for res in (False, True):
        ms = timeit(lambda: ops.linear(x, w, b, residual=r if res else None))
        print(f"linear {M}x{N}x{K} res={int(res)}  {ms:7.3f} ms {2*M*N*K/ms/1e9:6.0f} TF/s")
