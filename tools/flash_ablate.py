"""Timing-only ablations of the pipelined flash kernel (attention_v2.hip built with -DVCX_FLASH2_ABLATIONS into tools/_abl/libvcx_abl.so):
which component of the key loop the time goes to.  Results of ablated variants are garbage by construction.
    python tools/flash_ablate.py [N] [frames] [heads]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_abl", "libvcx_abl.so")
from viewcrafter_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
G = int(sys.argv[2]) if len(sys.argv) > 2 else 50
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 5
C = heads * 64
torch.manual_seed(0)
qk = torch.randn(G * N, 2 * C, device="cuda"); qk[:, :C] *= 0.125 * ops.LOG2E; qk = qk.half()
vt = torch.randn(C, G * N, device="cuda").half()
o = torch.empty(G * N, C, device="cuda", dtype=torch.float16)
names = {0: "full kernel", 1: "no vmcnt(0) at the barrier", 3: "no vmcnt, no barrier", 4: "no exp2", 8: "no softmax stream", 16: "no fragment reads",
         32: "no DMA", 35: "no DMA, no vmcnt, no barrier", 64: "no MFMA", 72: "no MFMA, no softmax (reads + DMA + barrier only)",
         59: "MFMA only (no DMA, reads, softmax, waits, barrier)"}
ops.tune_set("FLASH_IMPL", 2)
res = {k: [] for k in names}
def run():
    ops.flash_attn(qk, qk[:, C:], vt, o, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C, ldvt=G * N, ldo=C,
                   scale=0.125, log2_logits=True)
for r in range(3):
    for k in names:
        ops.tune_set("EXP0", k)
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): run()
        b.record(); torch.cuda.synchronize()
        res[k].append(a.elapsed_time(b) / 4)
ops.tune_set("EXP0", 0)
tiles = N // 64
for k, v in res.items():
    ms = sorted(v)[1]
    blocks_per_cu = G * heads * ((N + 255) // 256) / 256.0
    us_tile = ms * 1e3 / (blocks_per_cu * tiles)
    print(f"{k:3d} {names[k]:52s} {ms:7.3f} ms   {us_tile:6.3f} us per tile and CU   {4.0 * G * heads * N * N * 64 / ms / 1e9:6.0f} 'TF/s'", flush=True)
