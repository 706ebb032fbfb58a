#!/bin/bash
# Round 6, call 16: static s_setprio 1 for one half of an 8-wave GEMM tile (-DVCX_GEMM_PRIO_ABL=1: waves 4-7, =2: waves 0-3) against the product, tools/gemm_quick.py, two interleaved rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06ae}
{
for rep in 1 2; do
echo "## product"; python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids
echo "## waves 4-7 at priority 1"; VCX_LIB=tools/_abl/libvcx_prio1.so python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids | grep -v problem
echo "## waves 0-3 at priority 1"; VCX_LIB=tools/_abl/libvcx_prio2.so python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids | grep -v problem
done
} > gpurun_out/${tag}_prio_ab.txt
grep "##\|sum\|conv3x3 C=320\|geglu 115200\|linear 460800x320x1280" gpurun_out/${tag}_prio_ab.txt
