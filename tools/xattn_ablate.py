"""Timing-only ablations of the resident cross-attention kernel (second form) at the benchmark's level-0 / level-1 shapes: every
library given (tools/_abl/libvcx_xablN.so = csrc/attention.hip built with -DXABL=N, garbage results) against the product.
    python tools/xattn_ablate.py [lib ...]"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
libs = [_lib.LIB_PATH] + sys.argv[1:]
s = torch.cuda.current_stream().cuda_stream
def load(path):
    L = ctypes.CDLL(path)
    L.vcx_attn_flash_dual_d64_f16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_int64] * 2 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    L.vcx_tune_set.argtypes = [ctypes.c_int, ctypes.c_int]
    return L
Ls = {os.path.basename(p): load(p) for p in libs}
for (B, T, nq, heads) in [(2, 25, 9216, 5), (2, 25, 2304, 10)]:
    C = heads * 64; G = B * T
    q = (torch.randn(G * nq, C, device="cuda") * 0.18).half()
    kt = torch.zeros(B, 80, C, device="cuda"); kt[:, :77] = torch.randn(B, 77, C, device="cuda"); kt = kt.half()
    vt = torch.zeros(B, 80, C, device="cuda"); vt[:, :77] = torch.randn(B, 77, C, device="cuda")
    vt_t = vt.half().reshape(B * 80, C).t().contiguous()
    ki = torch.randn(B, 256, C, device="cuda").half(); vi_t = torch.randn(B, 256, C, device="cuda").half().reshape(B * 256, C).t().contiguous()
    o = torch.empty(G * nq, C, device="cuda", dtype=torch.float16)
    def call(L):
        rc = L.vcx_attn_flash_dual_d64_f16(q.data_ptr(), kt.data_ptr(), vt_t.data_ptr(), ki.data_ptr(), vi_t.data_ptr(), o.data_ptr(), G, heads, nq, 77, 80, T,
                                           C, B * 80, 256, 256, T, C, B * 256, C, C, 0.125, 2, s)
        assert rc == 0
    def t(L, it=10):
        call(L); call(L); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): call(L)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    res = {k: [] for k in Ls}
    for r in range(3):
        for k, L in Ls.items():
            res[k].append(t(L))
    Ls[os.path.basename(libs[0])].vcx_tune_set(3, 2); form1 = min(t(Ls[os.path.basename(libs[0])]) for _ in range(3)); Ls[os.path.basename(libs[0])].vcx_tune_set(3, 1)
    print(f"== nq={nq} heads={heads}: first form {form1:.3f} ms; " + "  ".join(f"{k.replace('libvcx_', '').replace('.so', '')}: {min(v):.3f}" for k, v in res.items()), flush=True)
