"""Run one GEMM/conv problem repeatedly (for rocprofv3 --pmc runs).  python tools/one_gemm.py conv|lin|geglu [iters]"""
import math
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_conv, pack_geglu

kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda"
if kind == "conv":
    x = (torch.randn(50, 72, 128, 320, device=dev)).half()
    w = pack_conv((torch.randn(320, 320, 3, 3, device=dev) / math.sqrt(2880)).half())
    b = torch.randn(320, device=dev)
    fn = lambda: ops.conv2d(x, w, b, kh=3, kw=3)
elif kind == "lin":
    x = torch.randn(460800, 320, device=dev).half()
    w = (torch.randn(320, 320, device=dev) / math.sqrt(320)).half()
    b = torch.randn(320, device=dev)
    fn = lambda: ops.linear(x, w, b, residual=x)
elif kind == "conv640":
    x = (torch.randn(50, 36, 64, 640, device=dev)).half()
    w = pack_conv((torch.randn(640, 640, 3, 3, device=dev) / math.sqrt(5760)).half())
    b = torch.randn(640, device=dev)
    fn = lambda: ops.conv2d(x, w, b, kh=3, kw=3)
elif kind == "big":
    x = torch.randn(28800, 5120, device=dev).half()
    w = (torch.randn(1280, 5120, device=dev) / math.sqrt(5120)).half()
    fn = lambda: ops.linear(x, w)
else:
    x = torch.randn(460800, 320, device=dev).half()
    wp, bp = pack_geglu(torch.randn(2560, 320, device=dev) / math.sqrt(320), torch.randn(2560, device=dev))
    wp = wp.half()
    fn = lambda: ops.linear(x, wp, bp, geglu=True)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
