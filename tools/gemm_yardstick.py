"""Yardstick (measurement only, never on the product path): libvcx's GEMM engine against the vendor libraries on the SAME
plain-GEMM problems - torch.nn.functional.linear in fp16, routed to hipBLASLt and to rocBLAS.  Convolutions are listed by the
(M, N, K) of their implicit GEMM and timed here as a plain linear layer of that shape, on both sides."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
dev = "cuda"

def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

shapes = [  # (M, N, K, note)
    (460800, 2560, 320, "ff.0 level 0 (GEGLU in the product)"), (115200, 5120, 640, "ff.0 level 1"), (28800, 10240, 1280, "ff.0 level 2"),
    (460800, 320, 320, "to_out level 0"), (115200, 640, 640, "to_out level 1"), (28800, 1280, 1280, "to_out level 2"),
    (460800, 320, 1280, "ff.2 level 0"), (115200, 640, 2560, "ff.2 level 1"), (28800, 1280, 5120, "ff.2 level 2"),
    (460800, 960, 320, "qkv level 0"), (115200, 1920, 640, "qkv level 1"), (28800, 3840, 1280, "qkv level 2"),
    (460800, 320, 2880, "conv3x3 C=320 as GEMM"), (115200, 640, 5760, "conv3x3 C=640"), (28800, 1280, 11520, "conv3x3 C=1280"),
    (460800, 320, 960, "conv3x1 C=320"), (115200, 640, 1920, "conv3x1 C=640"), (28800, 1280, 3840, "conv3x1 C=1280"),
]
libs = []
for name in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(name); libs.append(name)
    except Exception as e:  # noqa: BLE001
        print("cannot select", name, e)
print(f"{'M':>7s} {'N':>6s} {'K':>6s}  {'libvcx ms':>10s} {'TF/s':>6s} " + " ".join(f"{('rocblas' if l == 'cublas' else l) + ' ms':>13s} {'TF/s':>6s}" for l in libs) + "  note")
tot = {"vcx": 0.0, **{l: 0.0 for l in libs}}
for M, N, K, note in shapes:
    x = (torch.randn(M, K, device=dev)).half(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).half(); b = torch.randn(N, device=dev)
    bh = b.half()
    fl = 2.0 * M * N * K
    t = timeit(lambda: ops.linear(x, w, b)); tot["vcx"] += t
    row = f"{M:7d} {N:6d} {K:6d}  {t:10.3f} {fl / t / 1e9:6.0f} "
    for l in libs:
        torch.backends.cuda.preferred_blas_library(l)
        tl = timeit(lambda: torch.nn.functional.linear(x, w, bh)); tot[l] += tl
        row += f"{tl:13.3f} {fl / tl / 1e9:6.0f} "
    print(row + "  " + note)
    del x, w
print(f"{'sum':>21s}  {tot['vcx']:10.3f} {'':6s} " + " ".join(f"{tot[l]:13.3f} {'':6s}" for l in libs))
