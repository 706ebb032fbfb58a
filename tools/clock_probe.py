"""Sample the shader clock / power while a GEMM loop runs (is the 2.4 GHz peak clock sustained under MFMA load?)."""
import math, os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
x = torch.randn(28800, 5120, device="cuda").half(); w = (torch.randn(1280, 5120, device="cuda") / 70).half()
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            print(" | ".join(l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l or "mclk" in l), flush=True)
        except Exception as e:
            print("smi failed", e, flush=True)
        time.sleep(0.3)
print("idle:"); t = threading.Thread(target=sampler); t.start(); time.sleep(1.0)
print("load:", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): ops.linear(x, w)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
stop = True; t.join()
print(f"{n} gemms in {dt:.2f}s -> {2*28800*5120*1280*n/dt/1e12:.0f} TF/s")
