"""Tile-walk panel height of the wide layers (csrc/gemm_args.h tile_coords): is there a better order than 8 rows x 4 columns per XCD?

VERDICT r3 item 7 asked for a weight-stationary order for the level-1 / 2 GEGLU layers (their weights, 6.5 / 26 MB, are re-streamed
once per 8-row panel).  tools/gemm_traffic_ablate.py bounds what ALL operand delivery costs there; this A/B tries the obvious
alternatives: panels of 4 / 16 / 32 row tiles (32 = weight-stationary: a weight tile is streamed once per 32 row tiles and the
activations are re-read 4x as often).  The variants are separate builds (-DVCX_TILE_PANEL=n, tools/_abl/, never libvcx.so):

    python tools/panel_ab.py build      (CPU)          python tools/panel_ab.py      (GPU box; interleaved, same process)
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tools", "_abl")
CSRC = os.path.join(ROOT, "viewcrafter_amd", "csrc")
PANELS = (4, 16, 32)


def build():
    os.makedirs(ABL, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]
    others = [os.path.join(CSRC, "build", f"{n}.o") for n in ("api", "attention", "attention_v2", "norm", "elementwise")]
    for p in PANELS:
        objs = []
        for f in ("gemm", "gemm_dma"):
            o = f"/tmp/panel{p}_{f}.o"
            subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DVCX_TILE_PANEL={p}", "-c", os.path.join(CSRC, f + ".hip"), "-o", o])
            objs.append(o)
        dst = os.path.join(ABL, f"libvcx_panel{p}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", dst, *objs, *others])
        print("built", dst)


def run():
    import torch
    from viewcrafter_amd import _lib
    from viewcrafter_amd.packing import pack_geglu
    libs = {"panel 8 (shipped)": ctypes.CDLL(_lib.LIB_PATH)}
    for p in PANELS:
        libs[f"panel {p}"] = ctypes.CDLL(os.path.join(ABL, f"libvcx_panel{p}.so"))
    for L in libs.values():
        L.vcx_gemm_f16.argtypes = [ctypes.POINTER(_lib.GemmDesc), ctypes.c_void_p]
        L.vcx_last_error.restype = ctypes.c_char_p
    s = torch.cuda.current_stream().cuda_stream

    def t(fn, it=8):
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    for M, C in [(115200, 640), (28800, 1280), (7200, 1280)]:
        x = torch.randn(M, C, device="cuda").half()
        wp, bp = pack_geglu(torch.randn(8 * C, C, device="cuda") / math.sqrt(C), torch.randn(8 * C, device="cuda"))
        wp, bp = wp.half(), bp.float().contiguous()
        outs = {k: torch.empty(M, 4 * C, device="cuda", dtype=torch.float16) for k in libs}

        def call(k):
            d = _lib.GemmDesc()
            d.A, d.W, d.C, d.bias = x.data_ptr(), wp.data_ptr(), outs[k].data_ptr(), bp.data_ptr()
            d.lda, d.M, d.N, d.K, d.ldw, d.ldc = C, M, 8 * C, C, C, 4 * C
            d.flags = _lib.GEMM_BIAS_N | _lib.GEMM_GEGLU
            d.alpha = 1.0
            assert libs[k].vcx_gemm_f16(ctypes.byref(d), s) == 0, libs[k].vcx_last_error()
        res = {k: [] for k in libs}
        for _ in range(5):
            for k in libs:
                res[k].append(t(lambda: call(k)))
        row = f"GEGLU {M}x{8 * C}x{C}: "
        for k in libs:
            v = sorted(res[k])
            row += f"{k}: {v[2]:.3f} ms (min {v[0]:.3f})   "
        same = all(torch.equal(outs[k], outs["panel 8 (shipped)"]) for k in libs)
        print(row + ("outputs bit-identical" if same else "OUTPUTS DIFFER"), flush=True)


if __name__ == "__main__":
    build() if (len(sys.argv) > 1 and sys.argv[1] == "build") else run()
