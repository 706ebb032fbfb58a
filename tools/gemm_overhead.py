"""Fit tile time = a + b * ksteps for the large-tile GEMM: sweep K at fixed M, N (prints ms, TF/s and us per 256x320 tile)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_conv
dev = "cuda"

def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def rh(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).half()
M, N = 460800, 320
tiles = (M // 256) * (N // 320)
rounds = math.ceil(tiles / 256)
for K in (320, 640, 1280, 2560):
    x = rh(M, K); w = rh(N, K, sc=1 / math.sqrt(K)); b = torch.randn(N, device=dev); r = rh(M, N)
    for res in (False, True):
        ms = timeit(lambda: ops.linear(x, w, b, residual=r if res else None))
        print(f"linear K={K:5d} res={int(res)}  {ms:7.3f} ms {2*M*N*K/ms/1e9:6.0f} TF/s  {ms*1e3/rounds:7.1f} us/round ({K//64} ksteps)")
x = rh(50, 72, 128, 320)
for kh, kw in ((1, 1), (3, 1), (3, 3)):
    wt = pack_conv(rh(320, 320, kh, kw, sc=1 / math.sqrt(kh * kw * 320))); b = torch.randn(320, device=dev)
    ms = timeit(lambda: ops.conv2d(x, wt, b, kh=kh, kw=kw))
    K = 320 * kh * kw
    print(f"conv {kh}x{kw} K={K:5d}        {ms:7.3f} ms {2*M*N*K/ms/1e9:6.0f} TF/s  {ms*1e3/rounds:7.1f} us/round ({K//64} ksteps)")
