"""Current GEMM engine on square problems (reference point for tools/gemm_pp.hip)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for M, N, K in ((4096, 4096, 4096), (8192, 8192, 8192), (8192, 1280, 5120)):
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).half()
    for cfg in ("2", "3"):
        os.environ["VCX_GEMM_CFG"] = cfg
    ms = timeit(lambda: ops.linear(x, w))
    print(f"vcx_gemm {M}x{N}x{K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s")
