#!/bin/bash
# Round 6, call 2: parity of the one-launch GroupNorm statistics and of VCX_GEMM_ROWSTATS; their prices in isolation and in the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06b}
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rowstats or groupnorm or colstats or weight_stationary or units or lnfold or layernorm" 2>&1 | tail -15 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
timeout 300 python tools/rowstats_ab.py 50 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_rowstats_ab.txt
cat gpurun_out/${tag}_rowstats_ab.txt
{
timeout 400 python tools/step_ab.py --rounds 3 --steps 3 base:lnrs=0 rowstats:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round\|latent"
timeout 300 python tools/step_ab.py --lib tools/_abl/libvcx_gnold.so --rounds 2 --steps 3 gnold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
timeout 300 python tools/step_ab.py --rounds 2 --steps 3 gnnew:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
timeout 300 python tools/step_ab.py --lib tools/_abl/libvcx_gnold.so --rounds 2 --steps 3 gnold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-260
