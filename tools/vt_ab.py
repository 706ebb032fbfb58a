"""A/B of the tile configurations on the V^T projection shapes (M = channels, N = tokens): all within 3 % - the shape is bound by
the per-tile fixed cost (2880 tiles of a K = 320 problem), not by the 75 % padding of its second row tile."""
import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from viewcrafter_amd import ops
dev="cuda"
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/iters
for D, tokens in ((320, 460800), (640, 115200), (1280, 28800), (320, 230400)):
    h1=(torch.randn(tokens, D, device=dev)).half(); wv=(torch.randn(D, D, device=dev)/math.sqrt(D)).half()
    row=f"V^T {D} x {tokens} x {D}: "
    for cfg in ("", "0", "1", "2", "3"):
        os.environ["VCX_GEMM_CFG"]=cfg
        try:
            t=timeit(lambda: ops.gemm(wv, h1, M=D, N=tokens, K=D, lda=D))
            row+=f" cfg{cfg or 'auto'} {t:.3f}"
        except Exception as e:
            row+=f" cfg{cfg} ERR"
    print(row)
