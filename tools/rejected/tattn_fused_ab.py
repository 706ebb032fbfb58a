"""Same-box A/B: temporal attention + output projection + residual as one launch (vcx_attn_temporal_proj_d64_f16) against the pair it
replaces (vcx_attn_temporal_d64_f16 -> vcx_gemm_f16 with residual), at the level-0 and init_attn shapes of the 576x1024x25 workload.
    python tools/tattn_fused_ab.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
dev = "cuda"


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for name, B, T, P, heads, cout in [("level 0 (C = 320, 5 heads), B = 2", 2, 25, 9216, 5, 320), ("init_attn (inner 512, 8 heads -> 320), B = 1", 1, 25, 9216, 8, 320),
                                   ("320x512x25 level 0, B = 2", 2, 25, 2560, 5, 320), ("576x1024x16 level 0, B = 2", 2, 16, 9216, 5, 320)]:
    C = heads * 64
    tokens = B * T * P
    qkv = torch.randn(tokens, 3 * C, device=dev).half()
    wo = (torch.randn(cout, C, device=dev) / math.sqrt(C)).half()
    bias = torch.randn(cout, device=dev)
    res = torch.randn(tokens, cout, device=dev).half()
    o = torch.empty((tokens, C), dtype=torch.float16, device=dev)
    out = torch.empty((tokens, cout), dtype=torch.float16, device=dev)
    kw = dict(B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, scale=0.125)

    def unfused():
        ops.temporal_attn(qkv, o, ldo=C, **kw)
        return ops.linear(o, wo, bias, residual=res, out=out, ldc=cout)

    def attn_only():
        return ops.temporal_attn(qkv, o, ldo=C, **kw)

    def fused():
        return ops.temporal_attn_proj(qkv, wo, bias, res, out=out, **kw)
    r = {k: [] for k in ("attention", "pair", "fused")}
    for _ in range(5):
        r["attention"].append(timeit(attn_only)); r["pair"].append(timeit(unfused)); r["fused"].append(timeit(fused))
    med = {k: sorted(v)[2] for k, v in r.items()}
    gb = 2.0 * tokens * (3 * C + 2 * cout) / 1e9
    print(f"{name}: attention alone {med['attention']:.3f} ms, attention + GEMM {med['pair']:.3f} ms, fused {med['fused']:.3f} ms "
          f"({gb / med['fused']:.0f} GB/s on its {gb:.2f} GB)", flush=True)
