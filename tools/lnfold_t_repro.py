"""Bit-reproducibility probe of the folded-LayerNorm projections (profiles/r03zz_lnfold_t_determinism.txt): the LNFOLD_T projection
out[1280, 6216] and the same operands as a plain LNFOLD projection, ten calls per forced tile configuration (knob GEMM_CFG), counting the
elements that differ from call 0.  Found hipcc's `v_pk_fma_f32 ... op_sel:[0,1,1]` form in the 128x128 LNFOLD_T kernel to be run-dependent
on the MI355X; tools/isa_audit.py --stores now rejects that form.   python tools/lnfold_t_repro.py"""
import math, sys, torch
sys.path.insert(0, ".")
from viewcrafter_amd import ops
from viewcrafter_amd.packing import fold_layernorm
DEV = "cuda"
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape)); return torch.randn(*shape, generator=g)
D, tokens = 1280, 6216
x = (rnd(tokens, D, seed=211) * 2 + 0.5).to(DEV).half()
gamma = (1 + 0.3 * rnd(D, seed=212)).to(DEV); beta = (0.2 * rnd(D, seed=213)).to(DEV)
wv = (rnd(D, D, seed=214) / math.sqrt(D)).to(DEV)
wf, colsum, bias_f = fold_layernorm(wv, gamma, beta, None)
st = ops.row_stats(x, 1e-5)
torch.cuda.synchronize()
def run_t():
    return ops.gemm(wf, x, M=D, N=tokens, K=D, lda=D, bias=bias_f, bias_m=True, ln_stats=st, ln_colsum=colsum, ln_t=True)
def run_n():
    return ops.linear(x, wf, bias_f, ln_stats=st, ln_colsum=colsum)
for name, fn in (("T", run_t), ("N", run_n)):
    for cfg in (-1, 0, 1, 2, 3):
        ops.tune_set("GEMM_CFG", cfg)
        outs = [fn().clone() for _ in range(10)]
        torch.cuda.synchronize()
        nd = [int((o != outs[0]).sum()) for o in outs[1:]]
        idx = [(o != outs[0]).nonzero() for o in outs[1:]]
        rows = sorted({int(r) // 16 % 4 for t in idx for r in t[:, 0].tolist()}); cols = sorted({int(c) % 16 for t in idx for c in t[:, 1].tolist()})
        print(name, "cfg", cfg, "differing elements vs run 0:", nd, "row-group b", rows, "col%16", cols, flush=True)
ops.tune_set("GEMM_CFG", -1)
