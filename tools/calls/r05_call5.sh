#!/bin/bash
# GroupNorm apply pass with non-temporal stores / loads + stores (A/B builds) against the product, one box, one after the other
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05av}
for lib in viewcrafter_amd/libvcx.so tools/_abl/libvcx_gnnt2.so tools/_abl/libvcx_gnnt3.so viewcrafter_amd/libvcx.so; do
  timeout 150 python tools/step_ab.py --lib $lib --rounds 2 --steps 3 base:ws=1 2>&1 | grep -v amdgpu.ids | grep "library\|round"
done > gpurun_out/${tag}_gn_nt_ab.txt
cat gpurun_out/${tag}_gn_nt_ab.txt | cut -c1-230
