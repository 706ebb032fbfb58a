"""Where does a tile's time go in the SERIAL form of the weight-stationary kernel (csrc/gemm_ws.hip, gemm_ws320_kernel - since the
pipelined form took over the plain modes this tool drives the COLSTATS mode; the recorded runs, profiles/r05g-r05j_ws_ablate.txt, are of the
plain modes of the serial form as of the commits before)?  Timing-only builds with parts removed
(-DVCX_WS_ABL=n into tools/_abl/, never libvcx.so):  1 = no MFMA loop,  2 = no epilogue,  3 = the epilogue's arithmetic without its stores.

    python tools/ws_ablate.py build      (CPU)          python tools/ws_ablate.py      (GPU box; interleaved, same process)
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tools", "_abl")
CSRC = os.path.join(ROOT, "viewcrafter_amd", "csrc")
VARIANTS = {1: "no MFMA loop", 2: "no epilogue"}


def build():
    os.makedirs(ABL, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]
    others = [os.path.join(CSRC, "build", f"{n}.o") for n in ("api", "gemm", "gemm_dma", "attention", "attention_v2", "norm", "elementwise")]
    for v in VARIANTS:
        o = f"/tmp/ws_abl{v}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DVCX_WS_ABL={v}", "-c", os.path.join(CSRC, "gemm_ws.hip"), "-o", o])
        dst = os.path.join(ABL, f"libvcx_ws{v}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", dst, o, *others])
        print("built", dst)


def run():
    import torch
    from viewcrafter_amd import _lib
    libs = {"shipped": ctypes.CDLL(_lib.LIB_PATH)}
    for v, name in VARIANTS.items():
        libs[name] = ctypes.CDLL(os.path.join(ABL, f"libvcx_ws{v}.so"))
    for L in libs.values():
        L.vcx_gemm_f16.argtypes = [ctypes.POINTER(_lib.GemmDesc), ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    M = N = 0
    M, N, K = 460800, 320, 320
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)

    def desc(res):
        d = _lib.GemmDesc()
        d.struct_size = ctypes.sizeof(_lib.GemmDesc)
        d.A, d.W, d.C, d.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr()
        d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.ldr = M, N, K, K, K, N, N
        d.flags, d.alpha, d.rowadd_div = 1 | (8 if res else 0), 1.0, 1
        if res:
            d.residual = r.data_ptr()
        return d

    def t(L, d, it=10):
        for _ in range(3):
            assert L.vcx_gemm_f16(ctypes.byref(d), s) == 0
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            L.vcx_gemm_f16(ctypes.byref(d), s)
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / it
    print(f"linear {M} x {N} x {K}, weight-stationary kernel; ms per call (three interleaved rounds)")
    for res in (False, True):
        d = desc(res)
        rows = {k: [] for k in libs}
        for _ in range(3):
            for k, L in libs.items():
                rows[k].append(t(L, d))
        print(("+ residual" if res else "bias only").ljust(12) + "   ".join(f"{k}: {sorted(v)[1]:.3f}" for k, v in rows.items()))


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
