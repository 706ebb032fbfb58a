#!/bin/bash
# one short gpurun call: weight-stationary kernel tests (incl. the LayerNorm-folded one), the two A/Bs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05am}
timeout 420 python -m pytest tests/test_kernels_gpu.py -q -x -k "weight_stationary" > gpurun_out/${tag}_ws_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_ws_tests.log
timeout 180 python tools/ws_lnf_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ws_lnf_ab.txt
timeout 180 python tools/ws_geglu_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ws_geglu_ab.txt
timeout 300 python -m pytest tests/test_soak_gpu.py -q -x -k "round5" > gpurun_out/${tag}_soak.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_soak.log; tail -3 gpurun_out/${tag}_soak.log
tail -25 gpurun_out/${tag}_ws_tests.log; cat gpurun_out/${tag}_ws_lnf_ab.txt gpurun_out/${tag}_ws_geglu_ab.txt
