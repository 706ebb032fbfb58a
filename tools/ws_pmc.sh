#!/bin/bash
# counters of the weight-stationary kernels against the tiled engine's: two separate --pmc passes with --kernel-trace only
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp && export TMPDIR=/tmp
tag=${1:-r05au}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
  rm -rf /tmp/wspmc
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/wspmc -o w -- python $root/tools/ws_pmc_once.py > /tmp/wspmc.log 2>&1
  echo "== $grp"
  python $root/tools/pmc_summary.py $(find /tmp/wspmc -name "*.db" | head -1) 2>&1 | grep -A4 "ws320_geglu\|ws320_lnf\|Li256ELi256ELi4ELi2EEELb0ELb1\|Li256ELi320ELi4ELi2EEELb0ELb0ELb0ELi1"
done > $root/gpurun_out/${tag}_ws_pmc.txt 2>&1
cat $root/gpurun_out/${tag}_ws_pmc.txt | head -60
