#!/bin/bash
# Round 6, call 11: one-launch GroupNorm statistics up to 4096 strips (per-video norms of level 0): parity, step A/B against the previous norm.hip
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06u}
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_soak_gpu.py -q -m gpu -k "groupnorm or colstats or round6" 2>&1 | tail -3 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
{
for lib in tools/_abl/libvcx_gnprev.so viewcrafter_amd/libvcx.so tools/_abl/libvcx_gnprev.so viewcrafter_amd/libvcx.so; do
timeout 300 python tools/step_ab.py --lib $lib --rounds 2 --steps 3 x:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
done
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-240
