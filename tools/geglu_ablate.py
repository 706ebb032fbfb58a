import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_geglu
dev="cuda"
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for M, C in ((460800, 320), (115200, 640), (28800, 1280)):
    x = torch.randn(M, C, device=dev).half()
    wp, bp = pack_geglu(torch.randn(8 * C, C, device=dev) / math.sqrt(C), torch.randn(8 * C, device=dev)); wp = wp.half()
    row = f"geglu {M}x{8*C}x{C}: "
    for t in ("0", "4", "8", "12"):
        os.environ["VCX_GEMM_TUNE"] = t
        row += f"tune {t}: {timeit(lambda: ops.linear(x, wp, bp, geglu=True)):.3f} ms   "
    print(row)
