"""Timing-only bound for Winograd F(2x2, 3x3) on the UNet's 3x3 convolutions, from this engine's own kernels on this box:
direct implicit-GEMM convolution (shipped) against the cheapest conceivable un-fused Winograd pipeline
    input transform  (read X, write V = 16 matrices [tiles x Cin] = 4 x the bytes of X)      -> streamed at the box's copy bandwidth
    16 GEMMs [tiles x Cin] x [Cin x Cout]  = one linear GEMM with 4 x pixels rows, K = Cin  -> ops.linear on exactly that shape
    output transform (read M = 4 x the bytes of Y, write Y)                                  -> streamed at the copy bandwidth
No Winograd arithmetic is performed: the GEMM runs on random data of the right shape, the transforms are priced by bytes (their
VALU work - 32 adds per 16 values - is assumed free).  A fused kernel could save the V / M round trips but must then do the
transforms inside the operand path, which this engine feeds by LDS-DMA without touching a VALU.
    python tools/winograd_bound.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_conv
dev = "cuda"

def timeit(fn, iters=6):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

# copy bandwidth of this box (read + write bytes per second) on a 600 MB tensor
src = torch.randn(460800, 640, device=dev).half(); dst = torch.empty_like(src)
t = timeit(lambda: ops.copy2d(src, dst, src.shape[0], src.shape[1], src.shape[1], src.shape[1]))
bw = 2 * src.numel() * 2 / (t * 1e-3)
print(f"copy bandwidth (read + write): {bw / 1e12:.2f} TB/s")
del src, dst
print(f"{'conv 3x3 (50 frames)':28s} {'direct ms':>9s} {'TF/s':>6s} | {'wino GEMM ms':>12s} {'TF/s':>6s} {'transforms ms':>13s} {'total ms':>9s} {'vs direct':>9s}")
for C, h, w in [(320, 72, 128), (640, 36, 64), (1280, 18, 32), (1280, 9, 16)]:
    n = 50
    x = (torch.randn(n, h, w, C, device=dev)).half()
    wt = pack_conv((torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C)).half()); b = torch.randn(C, device=dev)
    td = timeit(lambda: ops.conv2d(x, wt, b, kh=3, kw=3))
    pix = n * h * w
    fl = 2 * pix * C * C * 9
    v = torch.randn(4 * pix, C, device=dev).half()             # the 16 transformed matrices, stacked
    wl = (torch.randn(C, C, device=dev) / math.sqrt(C)).half()
    tg = timeit(lambda: ops.linear(v, wl))
    xb = pix * C * 2
    tt = ((xb + 4 * xb) + (4 * xb + xb)) / bw * 1e3
    print(f"C={C:4d} {h}x{w} ({xb / 1e6:5.0f} MB)".ljust(28) + f" {td:9.3f} {fl / td / 1e9:6.0f} | {tg:12.3f} {fl / 2.25 / tg / 1e9:6.0f} {tt:13.3f} {tg + tt:9.3f} {(tg + tt) / td:8.2f}x", flush=True)
    del x, v
