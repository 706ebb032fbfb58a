"""Yardstick (measurement only): libvcx's d=64 flash attention against torch's scaled_dot_product_attention backends
(the vendor flash / memory-efficient kernels shipped with PyTorch-ROCm) on the UNet's spatial self-attention shapes.
The library gets its preferred contiguous [B, H, N, d] layout; libvcx reads the token-major projections the UNet produces."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda"

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

print(f"{'frames':>6s} {'heads':>5s} {'N':>6s}  {'libvcx ms':>10s} {'TF/s':>6s}  " + "  ".join(f"{n + ' ms':>18s} {'TF/s':>6s}" for n in ("sdpa flash", "sdpa mem-efficient")) + "   max|diff| vs sdpa")
for n, heads, N in ((50, 5, 9216), (50, 10, 2304), (50, 20, 576), (25, 5, 9216)):
    D = heads * 64; tokens = n * N; scale = 1.0 / 8.0
    qk = torch.randn(tokens, 2 * D, device=dev).half(); v = torch.randn(tokens, D, device=dev).half()
    vt = v.t().contiguous(); o = torch.empty(tokens, D, device=dev, dtype=torch.float16)
    fl = 4.0 * n * heads * N * N * 64
    t = timeit(lambda: ops.flash_attn(qk, qk[:, D:], vt, o, n_groups=n, heads=heads, nq=N, nk=N, kv_rows=N, kv_div=1, ldq=2 * D, ldk=2 * D,
                                      ldvt=tokens, ldo=D, scale=scale))
    row = f"{n:6d} {heads:5d} {N:6d}  {t:10.3f} {fl / t / 1e9:6.0f}  "
    q4 = qk[:, :D].reshape(n, N, heads, 64).permute(0, 2, 1, 3).contiguous(); k4 = qk[:, D:].reshape(n, N, heads, 64).permute(0, 2, 1, 3).contiguous()
    v4 = v.reshape(n, N, heads, 64).permute(0, 2, 1, 3).contiguous()
    diff = None
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
        try:
            with sdpa_kernel(be):
                tl = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, scale=scale))
                if diff is None:
                    ref = torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, scale=scale).permute(0, 2, 1, 3).reshape(tokens, D)
                    diff = (ref.float() - o.float()).abs().max().item()
            row += f"{tl:18.3f} {fl / tl / 1e9:6.0f}  "
        except Exception as e:  # noqa: BLE001
            row += f"{'n/a':>18s} {'':6s}  "
    print(row + f"   {diff if diff is not None else float('nan'):.2e}")
    del qk, v, vt, o, q4, k4, v4
