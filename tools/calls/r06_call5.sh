#!/bin/bash
# Round 6, call 5: one-launch GroupNorm statistics with one batch of loads per thread: parity, step A/B against the two-kernel build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06h}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm or colstats" 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
cat gpurun_out/${tag}_tests.txt
{
timeout 300 python tools/step_ab.py --lib tools/_abl/libvcx_gnold.so --rounds 2 --steps 3 gnold:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
timeout 300 python tools/step_ab.py --rounds 2 --steps 3 gnnew:lnrs=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
} > gpurun_out/${tag}_step_ab.txt
cat gpurun_out/${tag}_step_ab.txt | cut -c1-260
cd /tmp; rm -rf /tmp/prof_${tag}
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag} -o ${tag} -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/step_ab.py --rounds 1 --steps 2 gnnew:lnrs=1 > /tmp/prof_${tag}.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/rocprof_summary.py $(find /tmp/prof_${tag} -name "*.db" | head -1) 2>&1 | grep "gn_\|calls" | cut -c1-150 > gpurun_out/${tag}_gn_kernels.txt
cat gpurun_out/${tag}_gn_kernels.txt
