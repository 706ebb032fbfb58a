#!/bin/bash
# Round 6, call 7: the rest of the GPU suite behind the fuzz fix; the feed-forward pair in row chunks
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06k}
timeout 300 python tools/ff_chunk_ab.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ff_chunk_ab.txt
cat gpurun_out/${tag}_ff_chunk_ab.txt
timeout 2400 python -m pytest tests/test_fuzz_gpu.py tests/test_kernels_gpu.py tests/test_soak_gpu.py tests/test_entry_gpu.py tests/test_fullsize_gpu.py tests/test_config1_gpu.py -q -m gpu > gpurun_out/${tag}_gpu_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${tag}_gpu_pytest.log
tail -n 12 gpurun_out/${tag}_gpu_pytest.log
