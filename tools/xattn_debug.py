"""Where does the second form of the resident cross-attention kernel differ from the first?  Runs one dual-attention problem on every
library given (tools/_abl/libvcx_xdbgN.so: -DXDBG=N builds of csrc/attention.hip) and prints the error of form 2 against form 1 by
query row, column, head and group.    python tools/xattn_debug.py T nq nk1 nk2 [lib ...]"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
T, nq, nk1, nk2 = [int(v) for v in sys.argv[1:5]]
libs = [_lib.LIB_PATH] + sys.argv[5:]
B, heads = 2, 2
C = heads * 64
G = B * T
r1, r2 = (nk1 + 7) // 8 * 8, (nk2 + 7) // 8 * 8
g = torch.Generator().manual_seed(5)
q = (torch.randn(G * nq, C, generator=g) * 0.125 * 1.4426950408889634).half().cuda()
kt = torch.zeros(B, r1, C); vt = torch.zeros(B, r1, C); ki = torch.zeros(B, r2, C); vi = torch.zeros(B, r2, C)
kt[:, :nk1] = torch.randn(B, nk1, C, generator=g); vt[:, :nk1] = torch.randn(B, nk1, C, generator=g)
ki[:, :nk2] = torch.randn(B, nk2, C, generator=g); vi[:, :nk2] = torch.randn(B, nk2, C, generator=g)
kt, vt, ki, vi = [t.half().cuda() for t in (kt, vt, ki, vi)]
vt_t = vt.reshape(B * r1, C).t().contiguous(); vi_t = vi.reshape(B * r2, C).t().contiguous()
s = torch.cuda.current_stream().cuda_stream
for path in libs:
    L = ctypes.CDLL(path)
    L.vcx_attn_flash_dual_d64_f16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_int64] * 2 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    L.vcx_tune_set.argtypes = [ctypes.c_int, ctypes.c_int]
    L.vcx_last_error.restype = ctypes.c_char_p
    outs = {}
    for form in (2, 1):           # knob value: 2 = first form, 1 = second form
        L.vcx_tune_set(3, form)
        o = torch.full((G * nq, C), 7.0, device="cuda", dtype=torch.float16)
        rc = L.vcx_attn_flash_dual_d64_f16(q.data_ptr(), kt.data_ptr(), vt_t.data_ptr(), ki.data_ptr(), vi_t.data_ptr(), o.data_ptr(), G, heads, nq, nk1, r1, T,
                                           C, B * r1, nk2, r2, T, C, B * r2, C, C, 0.125, 2, s)
        torch.cuda.synchronize()
        assert rc == 0, L.vcx_last_error()
        outs[form] = o.float().cpu()
    a, b = outs[1].view(B, T * nq, heads, 64), outs[2].view(B, T * nq, heads, 64)
    err = (a - b).abs()
    print(f"== {os.path.basename(path)}: max |form2 - form1| {float(err.max()):.3e}, elements off by > 0.02: {int((err > 0.02).sum())} of {err.numel()}")
    bad = err > 0.02
    if bad.any():
        print("   bad per video:", bad.sum(dim=(1, 2, 3)).tolist(), " per head:", bad.sum(dim=(0, 1, 3)).tolist())
        rows = bad.any(dim=3).any(dim=2)            # [B, T nq]
        for v in range(B):
            idx = rows[v].nonzero().flatten().tolist()
            runs, start = [], None
            for i in idx:
                if start is None: start = prev = i
                elif i == prev + 1: prev = i
                else: runs.append((start, prev)); start = prev = i
            if start is not None: runs.append((start, prev))
            print(f"   video {v}: bad query rows (of {T * nq}) {runs[:40]}")
        cols = bad.any(dim=1).any(dim=0)          # [heads, 64]
        for h_ in range(heads):
            print(f"   head {h_}: bad columns {cols[h_].nonzero().flatten().tolist()}")
        i = bad.nonzero()[0].tolist()
        print("   first bad element", i, "form2", float(a[tuple(i)]), "form1", float(b[tuple(i)]), " row values form2", a[i[0], i[1], i[2], :8].tolist(), "form1", b[i[0], i[1], i[2], :8].tolist())
