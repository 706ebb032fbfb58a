"""Can a UNet forward (ctypes-launched kernels) be captured into a HIP graph via torch.cuda.CUDAGraph, and do the
VcxProfScope event pairs recorded inside the capture give valid timings on replay?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from viewcrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
from tests.tiny_config import TINY_UNET
from tests.util import load_synth

m = UNetModel(**TINY_UNET).eval()
load_synth(m)
m = m.cuda()
x = torch.randn(2, 8, 4, 32, 16, device="cuda")
ctx = torch.randn(2, 77 + 64, 128, device="cuda")
ts = torch.tensor([500, 500], device="cuda")
fs = torch.tensor([10, 10], device="cuda")
with torch.no_grad():
    y_ref = m(x, ts, context=ctx, fs=fs)          # warm: packs weights, caches context K/V
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = m(x, ts, context=ctx, fs=fs)
    torch.cuda.synchronize()
    print("eager ms/forward", (time.perf_counter() - t0) / 5 * 1e3)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = m(x, ts, context=ctx, fs=fs)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y_g = m(x, ts, context=ctx, fs=fs)
    torch.cuda.synchronize()
    x2 = torch.randn_like(x)
    y2_ref = m(x2, ts, context=ctx, fs=fs)
    x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    print("graph replay matches eager on new input:", float((y_g - y2_ref).abs().max()), float(y2_ref.abs().max()))
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graph ms/forward", (time.perf_counter() - t0) / 5 * 1e3)
    # events inside a capture
    try:
        ops.profile_begin(1 << 14)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            y_g2 = m(x, ts, context=ctx, fs=fs)
        g2.replay()
        torch.cuda.synchronize()
        prof = ops.profile_end()
        print("profile inside graph:", {k: (v["launches"], round(v["ms"], 3)) for k, v in prof.items()})
    except Exception as e:
        print("profiling inside capture failed:", repr(e)[:300])
