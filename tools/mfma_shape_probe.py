"""Clock, power and rate of the tiled engine's wave-tile K-step as 16x16x32 or as 32x32x16 MFMAs (tools/ubench_mfma_shape.hip): is one shape
cheaper per FLOP under the 1.4 kW cap?   python tools/mfma_shape_probe.py [seconds]"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from telemetry import Telemetry
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_abl", "libmfma_shape.so"))
lib.mfma_shape_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
out = torch.zeros(16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ITERS = 20000
flop_per_launch = 256 * 8 * ITERS * 2.0 * 64 * 160 * 64
for rnd in range(2):
    for shape in (16, 32, 16, 32):
        lib.mfma_shape_run(shape, 256, 1000, out.data_ptr(), s); torch.cuda.synchronize()
        n = 0
        with Telemetry(period_s=0.05) as tm:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); a.record()
            while time.perf_counter() - t0 < seconds:
                lib.mfma_shape_run(shape, 256, ITERS, out.data_ptr(), s); n += 1
                torch.cuda.synchronize()
            b.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
        sm = tm.summary(t0 + 0.3, t1)
        ms = a.elapsed_time(b) / n
        clk = (sm.get("sclk_mhz") or {}).get("mean") or float("nan"); pw = (sm.get("power_w") or {}).get("mean") or float("nan")
        tf = flop_per_launch / ms / 1e9
        cyc = ms * 1e-3 * clk * 1e6 / ITERS
        print(f"{'16x16x32' if shape == 16 else '32x32x16'}: {ms:8.3f} ms per launch  {tf:7.1f} TFLOP/s  clock {clk:6.0f} MHz  power {pw:6.0f} W  {cyc:7.1f} cycles per K-step (1280 = the pipe)  {tf / pw:5.3f} TFLOP/s per W", flush=True)
