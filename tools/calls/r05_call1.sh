#!/bin/bash
# one short gpurun call: the weight-stationary kernel tests + the round-5 soak on the new block map, its A/B, the tile-configuration scan
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05ak}
timeout 420 python -m pytest tests/test_kernels_gpu.py -q -x -k "weight_stationary" > gpurun_out/${tag}_ws_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_ws_tests.log
timeout 300 python -m pytest tests/test_soak_gpu.py -q -x -k "round5" > gpurun_out/${tag}_soak.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_soak.log
timeout 180 python tools/ws_geglu_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ws_geglu_spare_ab.txt
timeout 420 python tools/gemm_cfg_scan.py 2> gpurun_out/${tag}_gemm_cfg_scan.err | grep -v amdgpu.ids > gpurun_out/${tag}_gemm_cfg_scan.txt
tail -4 gpurun_out/${tag}_ws_tests.log; tail -3 gpurun_out/${tag}_soak.log; cat gpurun_out/${tag}_ws_geglu_spare_ab.txt; grep -c . gpurun_out/${tag}_gemm_cfg_scan.txt; grep "<--\|^#" gpurun_out/${tag}_gemm_cfg_scan.txt | head -40; tail -5 gpurun_out/${tag}_gemm_cfg_scan.err
