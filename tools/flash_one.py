"""One flash-attention problem, for rocprofv3 counter passes: python tools/flash_one.py [N] [frames] [heads] [log2]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from viewcrafter_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pre = len(sys.argv) > 4 and sys.argv[4] == "log2"
C = heads * 64
torch.manual_seed(0)
qk = torch.randn(G * N, 2 * C, device="cuda").half()
if pre:
    qk = (qk.float() * (0.125 * ops.LOG2E) ** 0.5).half()
vt = torch.randn(C, G * N, device="cuda").half()
o = torch.empty(G * N, C, device="cuda", dtype=torch.float16)
def run():
    ops.flash_attn(qk, qk[:, C:], vt, o, n_groups=G, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * C, ldk=2 * C, ldvt=G * N,
                   ldo=C, scale=0.125, log2_logits=pre)
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f"flash N={N} G={G} heads={heads} log2={pre}: {ms:.3f} ms  {4.0 * G * heads * N * N * 64 / ms / 1e9:.0f} TF/s")
