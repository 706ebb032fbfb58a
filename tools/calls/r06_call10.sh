#!/bin/bash
# Round 6, call 10: cache-policy hint on activation rows that are read once (linear layers with one column tile): nt / sc1 builds against the product
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06r}
for lib in "" tools/_abl/libvcx_ant2.so tools/_abl/libvcx_ant16.so ""; do
  echo "== library: ${lib:-product}"
  VCX_LIB=$lib timeout 300 python tools/gemm_quick.py auto 2>&1 | grep -v amdgpu.ids | grep "linear 460800x320\|lnfold\|tconv C=320\|conv3x3 C=320\|sum\|geglu 460800"
done > gpurun_out/${tag}_a_nt_ab.txt
cat gpurun_out/${tag}_a_nt_ab.txt
