#!/bin/bash
# the whole GPU suite in one gpurun call: tools/full_gpu_suite.sh <tag>  ->  gpurun_out/<tag>_gpu_pytest.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-run}
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_gpu_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_gpu_pytest.log
tail -n 8 gpurun_out/${tag}_gpu_pytest.log
