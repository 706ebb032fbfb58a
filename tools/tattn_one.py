"""One temporal-attention problem (timing): python tools/tattn_one.py [P] [C]   (the VCX_TATTN_TUNE loop served the round-2 A/B of
the staged Q / K loads; the library no longer reads that variable)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
P = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
C = int(sys.argv[2]) if len(sys.argv) > 2 else 320
tunes = sys.argv[3:] or ["0"]
B, T, heads = 2, 25, C // 64
qkv = torch.randn(B * T * P, 3 * C, device="cuda").half()
o = torch.empty(B * T * P, C, device="cuda", dtype=torch.float16)
def run():
    ops.temporal_attn(qkv, o, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, ldo=C, scale=0.125)
res = {t: [] for t in tunes}
for r in range(3):
    for t in tunes:
        os.environ["VCX_TATTN_TUNE"] = t
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize()
        res[t].append(a.elapsed_time(b) / 10)
for t in tunes:
    ms = sorted(res[t])[1]
    print(f"tattn P={P} C={C} tune {t}: {ms:.3f} ms  {B*T*P*C*2*4/ms/1e9:.2f} TB/s")
