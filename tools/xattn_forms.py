"""form 2 vs form 1 of the resident cross-attention kernel over key counts, with one value set zeroed (which set is wrong?)"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
L = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else _lib.LIB_PATH)
L.vcx_attn_flash_dual_d64_f16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_int64] * 2 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
L.vcx_tune_set.argtypes = [ctypes.c_int, ctypes.c_int]
s = torch.cuda.current_stream().cuda_stream
B, heads, T, nq = 2, 2, 2, 128
C = heads * 64; G = B * T
def run(nk1, nk2, zero):
    r1, r2 = (nk1 + 7) // 8 * 8, (nk2 + 7) // 8 * 8
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(G * nq, C, generator=g) * 0.125 * 1.4426950408889634).half().cuda()
    kt = torch.zeros(B, r1, C); vt = torch.zeros(B, r1, C); ki = torch.zeros(B, r2, C); vi = torch.zeros(B, r2, C)
    kt[:, :nk1] = torch.randn(B, nk1, C, generator=g); vt[:, :nk1] = torch.randn(B, nk1, C, generator=g)
    ki[:, :nk2] = torch.randn(B, nk2, C, generator=g); vi[:, :nk2] = torch.randn(B, nk2, C, generator=g)
    if zero == 1: vt.zero_()
    if zero == 2: vi.zero_()
    kt, vt, ki, vi = [t.half().cuda() for t in (kt, vt, ki, vi)]
    vt_t = vt.reshape(B * r1, C).t().contiguous(); vi_t = vi.reshape(B * r2, C).t().contiguous()
    outs = {}
    for form in (2, 1):
        L.vcx_tune_set(3, form)
        o = torch.full((G * nq, C), 7.0, device="cuda", dtype=torch.float16)
        rc = L.vcx_attn_flash_dual_d64_f16(q.data_ptr(), kt.data_ptr(), vt_t.data_ptr(), ki.data_ptr(), vi_t.data_ptr(), o.data_ptr(), G, heads, nq, nk1, r1, T,
                                           C, B * r1, nk2, r2, T, C, B * r2, C, C, 0.125, 2, s)
        torch.cuda.synchronize(); assert rc == 0
        outs[form] = o.float().cpu()
    # fp32 reference
    def ref_set(k, v, nk):
        qh = q.float().cpu().view(B, T * nq, heads, 64)
        kh = k.float().cpu()[:, :nk].view(B, nk, heads, 64); vh = v.float().cpu()[:, :nk].view(B, nk, heads, 64)
        sc = torch.einsum("bqhd,bkhd->bhqk", qh, kh) * 0.6931471805599453
        return torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vh).reshape(G * nq, C)
    ref = ref_set(kt, vt, nk1) + ref_set(ki, vi, nk2)
    e1 = float((outs[2] - ref).abs().max()); e2 = float((outs[1] - ref).abs().max())
    return e1, e2
for nk1, nk2 in [(32, 32), (64, 64), (32, 64), (64, 32), (77, 256), (96, 256), (128, 256), (77, 32), (32, 256), (40, 40), (77, 64), (64, 256)]:
    row = f"nk1={nk1:3d} nk2={nk2:3d}:"
    for zero, name in ((0, "both"), (1, "V_txt=0"), (2, "V_img=0")):
        e1, e2 = run(nk1, nk2, zero)
        row += f"  {name}: form1 {e1:.1e} form2 {e2:.1e}{' <--' if e2 > 5e-3 else ''}"
    print(row, flush=True)
