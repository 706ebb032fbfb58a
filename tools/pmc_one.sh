#!/bin/bash
# PMC traffic of single problems: bash tools/pmc_one.sh "<kinds>" [tune]   (FETCH_SIZE x2 = bytes read through the L2's fabric side)
kinds=${1:-"geglu"}; tune=${2:-0}
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for k in $kinds; do for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/p1_${k}_$(echo $c | tr ' ' '_'); rm -rf $d
  VCX_GEMM_TUNE=$tune rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python $R/tools/one_gemm.py $k 4 > /tmp/p1.log 2>&1 || tail -3 /tmp/p1.log
  echo "== $k $c (4 launches)"; python $R/tools/pmc_summary.py $(find $d -name "*.db" | head -1) | grep -A3 "gemm_" | grep -v "^--"
done; done
