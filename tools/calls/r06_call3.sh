#!/bin/bash
# Round 6, call 3: ROWSTATS after the bit_cast-of-a-vector-element fix: parity, price in isolation, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06c}
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowstats or groupnorm or colstats or weight_stationary or units or lnfold or layernorm" 2>&1 | tail -25 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
timeout 300 python tools/rowstats_ab.py 50 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_rowstats_ab.txt
cat gpurun_out/${tag}_rowstats_ab.txt
timeout 400 python tools/step_ab.py --rounds 3 --steps 3 base:lnrs=0 rowstats:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round\|latent" > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-260
