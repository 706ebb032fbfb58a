#!/bin/bash
# Round 6, call 8: the wave-specialised tile configuration (knob GEMM_CFG = 4: 4 compute + 4 DMA waves, 256 x 160): parity, then timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06l}
VCX_TUNE_GEMM_CFG=4 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm_linear or conv3x3 or temporal_conv or colstats or lnfold or rowadd or k_tail or strided or conv1x1" 2>&1 | tail -12 > gpurun_out/${tag}_spec_tests.txt
cat gpurun_out/${tag}_spec_tests.txt
timeout 600 python tools/gemm_quick.py auto cfg3 cfg4 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_gemm_quick_spec.txt
cat gpurun_out/${tag}_gemm_quick_spec.txt
