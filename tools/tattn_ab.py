"""Same-box A/B of the temporal attention kernel: the shipped build against another build of csrc/attention.hip (default: the previous
commit's, with the 13.8 KB-per-wave LDS patch = one block per CU).
    python tools/tattn_ab.py build [rev]   (CPU: git show rev:viewcrafter_amd/csrc/attention.hip -> tools/_abl/libvcx_tattn_old.so)
    python tools/tattn_ab.py               (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tools", "_abl")
CSRC = os.path.join(ROOT, "viewcrafter_amd", "csrc")


def build(rev):
    os.makedirs(ABL, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    src = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:viewcrafter_amd/csrc/attention.hip"], capture_output=True, text=True, check=True).stdout
    open(os.path.join(CSRC, "_attention_old.hip"), "w").write(src)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
                               "-c", os.path.join(CSRC, "_attention_old.hip"), "-o", "/tmp/attention_old.o"])
    finally:
        os.remove(os.path.join(CSRC, "_attention_old.hip"))
    others = [os.path.join(CSRC, "build", f"{n}.o") for n in ("api", "gemm", "gemm_dma", "gemm_ws", "attention_v2", "norm", "elementwise")]
    dst = os.path.join(ABL, "libvcx_tattn_old.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", dst, "/tmp/attention_old.o", *others])
    print("built", dst, "from", rev)


def run():
    import torch
    from viewcrafter_amd import _lib
    # python tools/tattn_ab.py run name=path [name=path ...]: the shipped library against the named builds (default: "previous" = libvcx_tattn_old.so)
    extra = dict(a.split("=", 1) for a in sys.argv[2:]) if len(sys.argv) > 2 else {"previous": os.path.join(ABL, "libvcx_tattn_old.so")}
    libs = {"shipped": ctypes.CDLL(_lib.LIB_PATH)}
    libs.update({k: ctypes.CDLL(os.path.join(ROOT, v) if not os.path.isabs(v) else v) for k, v in extra.items()})
    for L in libs.values():
        L.vcx_attn_temporal_d64_f16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream

    def t(fn, it=10):
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it
    tot = {k: 0.0 for k in libs}
    for name, B, T, P, heads, cnt in [("level 0", 2, 25, 9216, 5, 10), ("init_attn (B = 1, 8 heads)", 1, 25, 9216, 8, 2), ("level 1", 2, 25, 2304, 10, 10),
                                      ("level 2", 2, 25, 576, 20, 10), ("level 3", 2, 25, 144, 20, 2), ("16 frames, level 0", 2, 16, 9216, 5, 0)]:
        C = heads * 64
        tokens = B * T * P
        qkv = torch.randn(tokens, 3 * C, device="cuda").half()
        outs = {k: torch.empty((tokens, C), dtype=torch.float16, device="cuda") for k in libs}
        res = {k: [] for k in libs}
        for _ in range(5):
            for k, L in libs.items():
                res[k].append(t(lambda: L.vcx_attn_temporal_d64_f16(qkv.data_ptr(), outs[k].data_ptr(), B, T, P, heads, 3 * C, C, 2 * C, C, 0.125, s)))
        gb = 2.0 * tokens * 4 * C / 1e9
        row = f"{name:28s} x{cnt:2d}: "
        for k in libs:
            m = sorted(res[k])[2]
            tot[k] += m * cnt
            row += f"{k} {m * 1e3:7.1f} us ({gb / m:5.2f} TB/s)   "
        print(row + ("bit-identical" if all(torch.equal(outs["shipped"], o) for o in outs.values()) else "OUTPUTS DIFFER"), flush=True)
    print("per DDIM step (576x1024x25 launch counts): " + "   ".join(f"{k} {v:.2f} ms" for k, v in tot.items()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2] if len(sys.argv) > 2 else "HEAD")
    else:
        run()
