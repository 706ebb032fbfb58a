#!/bin/bash
# Round 6, call 14: L2 prefetch touch of a linear layer's activation rows one K-step ahead of their DMA (-DVCX_GEMM_L2_PREFETCH=1 build): timing, parity
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06ac}
{
echo "## product"; python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids
echo "## prefetch build"; VCX_LIB=tools/_abl/libvcx_pf.so python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids
echo "## product"; python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids | grep "linear\|geglu\|sum"
echo "## prefetch build"; VCX_LIB=tools/_abl/libvcx_pf.so python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids | grep "linear\|geglu\|sum"
} > gpurun_out/${tag}_pf_quick.txt
cat gpurun_out/${tag}_pf_quick.txt
timeout 900 python -c "
import os, sys
from viewcrafter_amd import _lib
_lib.LIB_PATH = os.path.abspath('tools/_abl/libvcx_pf.so')
import pytest
sys.exit(pytest.main(['tests/test_kernels_gpu.py', 'tests/test_soak_gpu.py', '-q', '-m', 'gpu', '-x', '-k', 'linear or gemm or geglu or lnfold or soak or units or rowstats']))
" 2>&1 | tail -4 > gpurun_out/${tag}_pf_tests.txt
cat gpurun_out/${tag}_pf_tests.txt
