"""A/B of the GroupNorm kernels of two builds of libvcx (timing only): python tools/gn_ab.py
The second build is any other revision of csrc/norm.hip linked into tools/libvcx_oldgn.so, e.g.
    git show <rev>:viewcrafter_amd/csrc/norm.hip > /tmp/norm_old.hip
    cd viewcrafter_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -I. -c /tmp/norm_old.hip -o /tmp/norm_old.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvcx_oldgn.so build/{api,gemm,gemm_dma,gemm_pp,attention,elementwise}.o /tmp/norm_old.o
(the .so is git-ignored; profiles/r02_experiments.md section 4 holds the numbers of the round-2 versions)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import _lib
libs = {"new": ctypes.CDLL(_lib.LIB_PATH), "old": ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libvcx_oldgn.so"))}
for L in libs.values():
    L.vcx_groupnorm_ws_bytes.restype = ctypes.c_size_t
    L.vcx_groupnorm_ws_bytes.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    L.vcx_groupnorm_stats_f16.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.vcx_groupnorm_apply_f16.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (n, pix, C) in [(50, 9216, 320), (2, 230400, 320), (50, 2304, 640), (2, 57600, 640), (50, 576, 1280), (50, 9216, 640), (2, 14400, 1280)]:
    x = torch.randn(n, pix, C, device="cuda").half(); y = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); st = torch.empty(n, 32, 2, device="cuda")
    row = f"n={n:3d} pix={pix:7d} C={C:5d} ({x.numel()*2/1e6:6.1f} MB): "
    for name, L in libs.items():
        ws = torch.empty(max(L.vcx_groupnorm_ws_bytes(n, pix, 32), 16), dtype=torch.uint8, device="cuda")
        ts = t(lambda: L.vcx_groupnorm_stats_f16(x.data_ptr(), st.data_ptr(), ws.data_ptr(), n, pix, C, 32, s))
        ta = t(lambda: L.vcx_groupnorm_apply_f16(x.data_ptr(), y.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), n, pix, C, 32, 1e-5, 1, s))
        row += f"{name}: stats {ts*1e3:7.1f} us ({x.numel()*2/ts/1e9:5.2f} TB/s) apply {ta*1e3:7.1f} us ({x.numel()*4/ta/1e9:5.2f} TB/s)   "
    print(row)
