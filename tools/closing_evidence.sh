#!/bin/bash
# one gpurun call: per-shape tables, PMC traffic, default bench, bench under rocprofv3 (tools/closing_evidence.sh <tag> <commit>)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-rXX}; commit=${2:-unknown}
timeout 400 python tools/gemm_shapes.py --json gpurun_out/${tag}_gemm_shapes.json 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_gemm_shapes.txt
cp gpurun_out/${tag}_gemm_shapes.json profiles/gemm_shapes.json
timeout 300 python tools/attn_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_attn_shapes.txt
timeout 600 bash tools/pmc_passes.sh ${tag} ${commit} > gpurun_out/${tag}_pmc_passes.log 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
timeout 900 bash tools/final_profile.sh ${tag} 2>&1 | tail -5
head -3 gpurun_out/${tag}_gemm_shapes.txt; tail -3 gpurun_out/${tag}_attn_shapes.txt; tail -12 gpurun_out/${tag}_pmc_traffic.txt
