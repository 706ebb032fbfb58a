#!/bin/bash
# Round 6, call 9: the batched emb projection: parity, model parity, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06n}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowadd or weight_stationary_320" 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -4 >> gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
{
VCX_EMB_BATCHED=0 timeout 300 python tools/step_ab.py --rounds 2 --steps 3 perblock:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
timeout 300 python tools/step_ab.py --rounds 2 --steps 3 batched:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
VCX_EMB_BATCHED=0 timeout 300 python tools/step_ab.py --rounds 2 --steps 3 perblock:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-260
