"""Per-problem A/B of the folded LayerNorm: [layer_norm kernel + projection] against [row_stats + projection with the
VCX_GEMM_LNFOLD epilogue], and the two projections alone.  Same process, interleaved.  python tools/lnfold_gemm_ab.py"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from viewcrafter_amd import ops  # noqa: E402
from viewcrafter_amd.packing import fold_layernorm, pack_geglu  # noqa: E402

DEV = "cuda"


def bench(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    rows = []
    for tokens, C in ((460800, 320), (115200, 640), (28800, 1280)):
        x = (torch.randn(tokens, C, device=DEV) * 2).half()
        gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
        for name, N, kind in (("qk", 2 * C, "lin"), ("q2", C, "lin"), ("qkv", 3 * C, "lin"), ("v^T", C, "t"), ("geglu", 8 * C, "geglu")):
            w = torch.randn(N, C, device=DEV) / math.sqrt(C)
            b = torch.randn(N, device=DEV) * 0.1
            wf, cs, bf = fold_layernorm(w, gamma, beta, b if kind == "geglu" else None)
            if kind == "geglu":
                wp, bp = pack_geglu(w.half(), b)
                wfp, bfp = pack_geglu(wf, bf)
                _, csp = pack_geglu(wf, cs)
                plain = lambda h: ops.linear(h, wp, bp, geglu=True)
                fold = lambda st: ops.linear(x, wfp, bfp, geglu=True, ln_stats=st, ln_colsum=csp)
            elif kind == "t":
                wh = w.half()
                plain = lambda h: ops.gemm(wh, h, M=N, N=tokens, K=C, lda=C)
                fold = lambda st: ops.gemm(wf, x, M=N, N=tokens, K=C, lda=C, bias=bf, bias_m=True, ln_stats=st, ln_colsum=cs, ln_t=True)
            else:
                wh = w.half()
                plain = lambda h: ops.linear(h, wh, alpha=0.5)
                fold = lambda st: ops.linear(x, wf, bf, alpha=0.5, ln_stats=st, ln_colsum=cs)
            h = ops.layer_norm(x, gamma, beta, 1e-5)
            st = ops.row_stats(x, 1e-5)
            t_ln = bench(lambda: ops.layer_norm(x, gamma, beta, 1e-5))
            t_st = bench(lambda: ops.row_stats(x, 1e-5))
            t_p = bench(lambda: plain(h))
            t_f = bench(lambda: fold(st))
            t_pair_p = bench(lambda: plain(ops.layer_norm(x, gamma, beta, 1e-5)))
            t_pair_f = bench(lambda: fold(ops.row_stats(x, 1e-5)))
            print(f"{tokens:7d} x {C:4d} {name:6s} N={N:5d}: LN {t_ln:.3f}  stats {t_st:.3f} | gemm plain {t_p:.3f}  folded {t_f:.3f} ({(t_f / t_p - 1) * 100:+.1f} %) | "
                  f"pair plain {t_pair_p:.3f}  folded {t_pair_f:.3f} ({(t_pair_f / t_pair_p - 1) * 100:+.1f} %)", flush=True)


main()
