#!/bin/bash
# Round 6, call 12: the GroupNorm statistics pass takes a thread's last one to three pixels as one batch of loads: parity (+ fuzz), step A/B against the previous library
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06x}
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_soak_gpu.py tests/test_fuzz_gpu.py -q -m gpu -k "groupnorm or colstats or round6 or group_norm" 2>&1 | tail -3 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
{
for lib in tools/_abl/libvcx_prev.so viewcrafter_amd/libvcx.so tools/_abl/libvcx_prev.so viewcrafter_amd/libvcx.so; do
timeout 300 python tools/step_ab.py --lib $lib --rounds 2 --steps 3 x:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
done
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-240
