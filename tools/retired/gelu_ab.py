"""Same-box A/B of the GEGLU layers with the two exact-erf GELU forms (csrc/gemm_args.h): the shipped exp2-of-a-polynomial normal
tail against the Abramowitz-Stegun 7.1.26 form of rounds 1-2.  The second library is built by
    cd viewcrafter_amd/csrc && for f in gemm gemm_dma elementwise; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math \\
        -fno-finite-math-only -DVCX_GELU_AS7126 -c $f.hip -o /tmp/abl/$f.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_abl/libvcx_gelu_as.so build/{api,attention,attention_v2,norm}.o /tmp/abl/{gemm,gemm_dma,elementwise}.o
(git-ignored).  python tools/gelu_ab.py [other-library [label]]   (default: the A&S build above; round 3 also used a copy of the previous
libvcx.so - the scalar Horner chain - against the packed-fp32 one)"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viewcrafter_amd import _lib
from viewcrafter_amd.packing import pack_geglu
OTHER = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "_abl", "libvcx_gelu_as.so")
OTHER_NAME = sys.argv[2] if len(sys.argv) > 2 else "A&S 7.1.26"
libs = {"exp2-poly (shipped)": ctypes.CDLL(_lib.LIB_PATH), OTHER_NAME: ctypes.CDLL(OTHER)}
for L in libs.values():
    L.vcx_gemm_f16.argtypes = [ctypes.POINTER(_lib.GemmDesc), ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
def t(fn, it=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for M, C in [(460800, 320), (115200, 640), (28800, 1280)]:
    x = torch.randn(M, C, device="cuda").half()
    wp, bp = pack_geglu(torch.randn(8 * C, C, device="cuda") / math.sqrt(C), torch.randn(8 * C, device="cuda"))
    wp = wp.half(); bp = bp.float().contiguous()
    outs = {}
    res = {k: [] for k in libs}
    for k in libs:
        outs[k] = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    def call(k):
        d = _lib.GemmDesc()
        d.A, d.W, d.C, d.bias = x.data_ptr(), wp.data_ptr(), outs[k].data_ptr(), bp.data_ptr()
        d.lda, d.M, d.N, d.K, d.ldw, d.ldc = C, M, 8 * C, C, C, 4 * C
        d.flags = _lib.GEMM_BIAS_N | _lib.GEMM_GEGLU
        d.alpha = 1.0
        assert libs[k].vcx_gemm_f16(ctypes.byref(d), s) == 0
    for r in range(5):
        for k in libs:
            res[k].append(t(lambda: call(k)))
    row = f"GEGLU {M}x{8*C}x{C}: "
    for k in libs:
        v = sorted(res[k]); row += f"{k}: {v[2]:.3f} ms (min {v[0]:.3f})   "
    d = (outs["exp2-poly (shipped)"].float() - outs[OTHER_NAME].float()).abs().max().item()
    print(row + f"max |difference| {d:.2e}", flush=True)
