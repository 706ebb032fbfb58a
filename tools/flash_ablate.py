"""Timing-only ablations of the pipelined flash kernel (attention_v2.hip built with -DVCX_FLASH2_ABLATIONS: tools/build_abl.sh flash
-DVCX_FLASH2_ABLATIONS -> tools/_abl/libvcx_flash.so): which component of the key loop the time goes to - now with the graphics clock and
the socket power SAMPLED while each variant loops (tools/telemetry.py), so that a variant's time per tile is also stated in shader
cycles: time x clock.  Results of ablated variants are garbage by construction.
    python tools/flash_ablate.py [N] [frames] [heads] [seconds per variant]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from viewcrafter_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_abl", "libvcx_flash.so")
from viewcrafter_amd import ops
from telemetry import Telemetry

N = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
G = int(sys.argv[2]) if len(sys.argv) > 2 else 50
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 5
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
C = heads * 64
torch.manual_seed(0)
qk = torch.randn(G * N, 2 * C, device="cuda"); qk[:, :C] *= 0.125 * ops.LOG2E; qk = qk.half()
vt = torch.randn(C, G * N, device="cuda").half()
o = torch.empty(G * N, C, device="cuda", dtype=torch.float16)
names = {0: "full kernel", 1: "no vmcnt(0) at the barrier", 3: "no vmcnt, no barrier", 4: "no exp2", 8: "no softmax stream", 16: "no fragment reads",
         32: "no DMA", 35: "no DMA, no vmcnt, no barrier", 64: "no MFMA", 72: "no MFMA, no softmax (reads + DMA + barrier only)",
         59: "MFMA only (no DMA, reads, softmax, waits, barrier)"}
ops.tune_set("FLASH_IMPL", 2)
def run():
    ops.flash_attn(qk, qk[:, C:], vt, o, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C, ldvt=G * N, ldo=C,
                   scale=0.125, log2_logits=True)
tiles = N // 64
blocks_per_cu = G * heads * ((N + 255) // 256) / 256.0
print(f"N = {N}, {G} frames x {heads} heads: {blocks_per_cu:.2f} blocks per CU x {tiles} key tiles; one 64-key tile of a wave = 32 MFMAs 32x32x16 = 1024 matrix-pipe cycles")
print(f"{'ABL':>3} {'variant':52s} {'ms':>7} {'us/tile':>8} {'sclk MHz':>9} {'W':>6} {'cycles/tile':>11} {'pipe busy':>9}  'TF/s'")
for k in names:
    ops.tune_set("EXP0", k)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 0
    with Telemetry(period_s=0.05) as tm:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(8):
                run()
            n += 8
            torch.cuda.synchronize()
        b.record(); torch.cuda.synchronize()
        t1 = time.perf_counter()
    ms = a.elapsed_time(b) / n
    s = tm.summary(t0 + 0.3, t1)                 # the first 0.3 s: the clock is still settling
    sclk = (s.get("sclk_mhz") or {}).get("mean") or float("nan")
    pw = (s.get("power_w") or {}).get("mean") or float("nan")
    us_tile = ms * 1e3 / (blocks_per_cu * tiles)
    cyc = us_tile * sclk
    busy = (0.0 if (k & 64) else 1024.0) / cyc if cyc == cyc else float("nan")
    print(f"{k:3d} {names[k]:52s} {ms:7.3f} {us_tile:8.3f} {sclk:9.0f} {pw:6.0f} {cyc:11.0f} {busy:9.2f}  {4.0 * G * heads * N * N * 64 / ms / 1e9:6.0f}", flush=True)
ops.tune_set("EXP0", 0)
