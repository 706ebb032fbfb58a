"""Is any TIME attached to the GEMM family's excess counter traffic (650 MB per launch against 439 MB algorithmic, 1.48x)?

The excess sits in the layers whose operands exceed an XCD's 4 MB L2 - the GEGLU / 4C->C projections of levels 1 and 2 and the deep
convolutions' tap re-reads (profiles/r02_experiments.md section 8) - and is served by the Infinity Cache.  This tool removes the
traffic WITHOUT touching the kernel: the same launches with a leading dimension of ZERO, so that every activation row (lda = 0)
and / or every weight row (ldw = 0) aliases one 2 KB line that never leaves L2 - identical instruction streams, identical tile
walk, identical LDS and MFMA work, operand traffic from the memory side ~0.  Interleaved, same box, same process:

    normal | lda = 0 (activations free) | ldw = 0 (weights free) | both

If "both" is not faster than "normal", no time is attached to the excess traffic of that layer; the difference is an UPPER bound
on what a weight-stationary tile order could recover.   python tools/gemm_traffic_ablate.py
(timing only: the aliased runs compute garbage by construction).
"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_conv, pack_geglu
dev = "cuda"


def timeit(fn, iters=6):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rh(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).half()


cases = []          # (name, launches per step, fn(lda_zero, ldw_zero), flop)


def lin(M, N, K, count, geglu=False, res=False, name=None):
    x, b = rh(M, K), torch.randn(N, device=dev)
    w = rh(N, K, sc=1 / math.sqrt(K))
    if geglu:
        w, b = pack_geglu(w, b)
    r = rh(M, N) if res else None
    out = torch.empty((M, N // 2 if geglu else N), dtype=torch.float16, device=dev)

    def fn(za, zw):
        return ops.gemm(x, w, M=M, N=N, K=K, lda=0 if za else K, ldw=0 if zw else K, bias=b, residual=r, geglu=geglu, out=out, ldc=out.shape[1])
    cases.append((name or f"{'geglu' if geglu else 'linear'} {M}x{N}x{K}{' +res' if res else ''}", count, fn, 2.0 * M * N * K))


def conv(C, h, w, count, n=50, cout=None, name=None):
    cout = cout or C
    x = rh(n, h, w, C)
    wt = pack_conv(rh(cout, C, 3, 3, sc=1 / math.sqrt(9 * C)))
    b = torch.randn(cout, device=dev)
    geom = dict(in_h=h, in_w=w, out_h=h, out_w=w, cin=C, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, ups=0)
    M, K = n * h * w, 9 * C

    def fn(za, zw):       # lda = 0: every pixel of the image is the same 128-byte-per-slab line
        return ops.gemm(x, wt, M=M, N=cout, K=K, lda=0 if za else C, ldw=0 if zw else K, bias=b, conv=geom)
    cases.append((name or f"conv3x3 {C}->{cout} {h}x{w}", count, fn, 2.0 * M * cout * K))


# the layers that carry the excess (launch counts per DDIM step from profiles/r02_gemm_shapes.txt), and two that do not, as controls
lin(115200, 5120, 640, 10, geglu=True)
lin(28800, 10240, 1280, 10, geglu=True)
lin(115200, 640, 2560, 10, res=True)
lin(28800, 1280, 5120, 10, res=True)
conv(1280, 18, 32, 5)
conv(2560, 18, 32, 2, cout=1280)
conv(640, 36, 64, 5)
lin(460800, 2560, 320, 10, geglu=True, name="control: geglu 460800x2560x320 (1.11x its algorithmic bytes)")
lin(460800, 320, 320, 29, res=True, name="control: linear 460800x320x320 +res (HBM-bound)")

variants = [("normal", False, False), ("lda=0", True, False), ("ldw=0", False, True), ("both", True, True)]
res = {v[0]: [[] for _ in cases] for v in variants}
for rnd in range(3):
    for name, za, zw in variants:
        for i, (_, _, fn, _) in enumerate(cases):
            res[name][i].append(timeit(lambda: fn(za, zw)))
med = lambda v: sorted(v)[len(v) // 2]
print(f"{'layer':62s} x/step " + " ".join(f"{v[0]:>9s}" for v in variants) + "   normal TF/s   (normal - both) x launches")
tot = 0.0
for i, (name, count, fn, fl) in enumerate(cases):
    t = {v[0]: med(res[v[0]][i]) for v in variants}
    gain = (t["normal"] - t["both"]) * count
    if not name.startswith("control"):
        tot += max(gain, 0.0)
    print(f"{name:62s} {count:5d}  " + " ".join(f"{t[v[0]]:9.3f}" for v in variants) + f"   {fl / t['normal'] / 1e9:9.0f}   {gain:+.3f} ms per step")
print(f"upper bound on the time attached to the excess operand traffic of the listed (non-control) layers: {tot:.2f} ms per step")
