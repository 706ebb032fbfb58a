"""Quick A/B of the GEMM engine on representative problems of one UNet forward (prints ms and TF/s)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("VCX_LIB"):       # another build of the library (tools/build_abl.sh): VCX_LIB=tools/_abl/libvcx_x.so python tools/gemm_quick.py
    from viewcrafter_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["VCX_LIB"])
from viewcrafter_amd import ops
from viewcrafter_amd.packing import pack_conv, pack_geglu
dev = "cuda"

def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def rh(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).half()

cases = []
def conv(C, h, w, n=50, ups=0, name=None):
    x = rh(n, h, w, C); wt = pack_conv(rh(C, C, 3, 3, sc=1 / math.sqrt(9 * C))); b = torch.randn(C, device=dev)
    fl = 2 * n * (h << ups) * (w << ups) * C * C * 9
    cases.append((name or f"conv3x3 C={C} {h}x{w}{' ups' if ups else ''}", lambda: ops.conv2d(x, wt, b, kh=3, kw=3, ups=ups), fl))
def lin(M, N, K, res=False, name=None):
    x = rh(M, K); wt = rh(N, K, sc=1 / math.sqrt(K)); b = torch.randn(N, device=dev); r = rh(M, N) if res else None
    cases.append((name or f"linear {M}x{N}x{K}{' +res' if res else ''}", lambda: ops.linear(x, wt, b, residual=r), 2 * M * N * K))
def lin_cs(M, N, K):
    x = rh(M, K); wt = rh(N, K, sc=1 / math.sqrt(K)); b = torch.randn(N, device=dev); r = rh(M, N); cs = ops.colstats_buffer(M, N, dev)
    cases.append((f"linear {M}x{N}x{K} +res +colstats", lambda: ops.linear(x, wt, b, residual=r, colstats=cs), 2 * M * N * K))
def lnf(M, N, K):
    from viewcrafter_amd.packing import fold_layernorm
    x = rh(M, K, sc=2.0); wf, cs, bf = fold_layernorm(torch.randn(N, K, device=dev) / math.sqrt(K), 1 + 0.3 * torch.randn(K, device=dev), 0.2 * torch.randn(K, device=dev), torch.randn(N, device=dev))
    st = ops.row_stats(x, 1e-5)
    cases.append((f"lnfold linear {M}x{N}x{K}", lambda: ops.linear(x, wf, bf, ln_stats=st, ln_colsum=cs), 2 * M * N * K))
def geglu(M, C):
    x = rh(M, C); wp, bp = pack_geglu(torch.randn(8 * C, C, device=dev) / math.sqrt(C), torch.randn(8 * C, device=dev)); wp = wp.half()
    cases.append((f"geglu {M}x{8*C}x{C}", lambda: ops.linear(x, wp, bp, geglu=True), 2 * M * 8 * C * C))
def tconv(C, P, T=25, B=2):
    x = rh(B, T, P, C); wt = pack_conv(rh(C, C, 3, 1, 1, sc=1 / math.sqrt(3 * C))); b = torch.randn(C, device=dev)
    cases.append((f"tconv C={C} P={P}", lambda: ops.temporal_conv3(x, wt, b), 2 * B * T * P * C * C * 3))

conv(320, 72, 128); conv(640, 36, 64); conv(1280, 18, 32); conv(640, 36, 64, ups=1)
tconv(320, 9216); tconv(1280, 576)
lin(460800, 320, 320, res=True); lin(460800, 320, 320); lin(460800, 960, 320); lin(460800, 320, 1280, res=True)
lin(115200, 640, 640, res=True); lin(28800, 1280, 5120, res=True); lin(28800, 3840, 1280)
geglu(460800, 320); geglu(115200, 640); geglu(28800, 1280)
lin(28800, 1280, 1280, res=True); lin(115200, 640, 2560, res=True); lin(460800, 640, 320)
lnf(460800, 960, 320); lnf(460800, 640, 320); lin_cs(460800, 320, 320)
conv(1280, 9, 16); tconv(640, 2304)
# A/B of dispatcher settings given on the command line, interleaved rounds, median and min.  Arguments: "auto" (the product),
# cfgN = force tile configuration N (knob GEMM_CFG), expN = knob EXP0 set to N (whatever experiment the library was built with)
tunes = [a for a in sys.argv[1:]] or ["auto"]
rounds = 3
res = {t: [[] for _ in cases] for t in tunes}
for r in range(rounds):
    for t in tunes:
        ops.tune_set("GEMM_CFG", int(t[3:]) if t.startswith("cfg") else -1)
        ops.tune_set("EXP0", int(t[3:]) if t.startswith("exp") else 0)
        ops.tune_set("GEMM_WS", int(t[2:]) if t.startswith("ws") else 1)        # ws0 = tiled engine for the N = K = 320 layers too
        for i, (name, fn, fl) in enumerate(cases):
            res[t][i].append(timeit(fn, iters=6))
med = lambda v: sorted(v)[len(v) // 2]
print(f"{'problem':34s} " + " ".join(f"{str(t) + ' ms (min)':>22s} {'TF/s':>6s}" for t in tunes))
tot = {t: 0.0 for t in tunes}
for i, (name, fn, fl) in enumerate(cases):
    row = f"{name:34s} "
    for t in tunes:
        m = med(res[t][i]); tot[t] += m
        row += f"{m:12.3f} ({min(res[t][i]):7.3f}) {fl/m/1e9:6.0f} "
    print(row)
print(f"{'sum':34s} " + " ".join(f"{tot[t]:12.3f} {'':17s}" for t in tunes))
