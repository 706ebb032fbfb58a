#!/bin/bash
# Round 6, call 6: the K tail (folded skip convolution) and the split concat: kernel parity, model parity, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06i}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "k_tail or split_concat or conv" 2>&1 | tail -15 > gpurun_out/${tag}_kernel_tests.txt
cat gpurun_out/${tag}_kernel_tests.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullconfig_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/${tag}_model_tests.txt
cat gpurun_out/${tag}_model_tests.txt
{
VCX_SKIP_FOLD=0 timeout 300 python tools/step_ab.py --rounds 2 --steps 3 nofold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
timeout 300 python tools/step_ab.py --rounds 2 --steps 3 fold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
VCX_CAT_SPLIT=0 timeout 300 python tools/step_ab.py --rounds 2 --steps 3 foldnosplit:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
VCX_SKIP_FOLD=0 timeout 300 python tools/step_ab.py --rounds 2 --steps 3 nofold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-260
