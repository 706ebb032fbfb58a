"""Parity test of tools/rejected/gemm_geglu_deferred.hip as it ran while the kernel was part of libvcx (commit "GEGLU with a
deferred epilogue ...", round 5): `git checkout <that commit>` to run it.  Result there: 14 passed on the MI355X at the first run, both
MFMA forms; the kernel is 4-14 % slower than the phased epilogue (profiles/r05c_geglu_deferred_ab.txt) and was taken out again."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import DEV, check, rel_l2, rnd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,D,K,bias", [(10000, 1280, 320, True),     # level-0 shape class: 400 tiles, one or two per block, ragged rows
                                         (70000, 640, 320, True),      # 1370 tiles: five or six tiles per block through the deferred path
                                         (4100, 608, 320, False),      # N = 1216: a partial last column tile (three 64-column blocks), no bias
                                         (9000, 2560, 640, True),      # 20 column tiles: the 8-row panel walk
                                         (785, 5120, 1280, True)])     # 20 K-steps; the last row tile has 17 valid rows
def test_gemm_geglu_deferred_matches_phased(M, D, K, bias):
    """csrc/gemm_geglu.hip (epilogue of tile i under the MFMAs of tile i + 1; value and gate rounded to fp16 first, as the reference's
    autocast Linear does) against the phased epilogue of gemm_dma.hip and against fp32: both within the kernel tolerance of the
    reference op (lvdm/modules/attention.py:415-422), and within two fp16 roundings of each other."""
    from viewcrafter_amd import ops
    from viewcrafter_amd.packing import pack_geglu
    x = rnd(M, K, seed=131).to(DEV).half()
    w = (rnd(2 * D, K, seed=132) / math.sqrt(K)).to(DEV)
    b = (rnd(2 * D, seed=133) * 0.3).to(DEV)
    wp, bp = pack_geglu(w, b)
    wp = wp.half()
    outs = {}
    for impl in (1, 2, 3):         # phased; deferred on 32x32x16 MFMAs (the product); deferred on 16x16x32 (kept for the A/B)
        prev = ops.tune_set("GEGLU_IMPL", impl)
        try:
            outs[impl] = ops.linear(x, wp, bp if bias else None, geglu=True)
            torch.cuda.synchronize()
        finally:
            ops.tune_set("GEGLU_IMPL", prev)
    h = x.float() @ w.half().float().t() + (b if bias else 0.0)
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    check(outs[1], ref, name="geglu phased")
    check(outs[2], ref, name="geglu deferred")
    check(outs[3], ref, name="geglu deferred (16x16x32)")
    assert rel_l2(outs[2], outs[1]) <= 1.5e-3
    assert not torch.equal(outs[1], outs[2]), "both settings of GEGLU_IMPL ran the same kernel"
    # the two deferred forms apply the same roundings to sums that differ only in the order of the K-slices inside a K-step
    assert rel_l2(outs[3], outs[2]) <= 3e-4
    # the deferred kernel writes nothing outside its rows / columns: a padded output buffer keeps its guard band
    prev = ops.tune_set("GEGLU_IMPL", 2)
    try:
        big = torch.full((M + 3, D + 8), 7.0, device=DEV, dtype=torch.float16)
        ops.gemm(x, wp, M=M, N=2 * D, K=K, lda=K, out=big, ldc=D + 8, bias=bp if bias else None, geglu=True)
        torch.cuda.synchronize()
    finally:
        ops.tune_set("GEGLU_IMPL", prev)
    assert torch.equal(big[:M, :D], outs[2]) and bool((big[M:] == 7.0).all()) and bool((big[:, D:] == 7.0).all())


