#!/bin/bash
# Round 6, call 4: flash2 ablation table with sampled clock / power per variant; clock and power per GEMM problem class
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06f}
timeout 400 python tools/flash_ablate.py 9216 50 5 1.5 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_flash_ablate.txt
cat gpurun_out/${tag}_flash_ablate.txt
timeout 600 python tools/clock_probe.py --top 16 --seconds 1.2 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_clock_probe.txt
cat gpurun_out/${tag}_clock_probe.txt
