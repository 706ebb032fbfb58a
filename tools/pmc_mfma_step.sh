#!/bin/bash
# matrix-pipe occupancy per kernel over one DDIM step: one --pmc pass with --kernel-trace only (gpurun refuses --pmc with other trace domains)
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp && export TMPDIR=/tmp
tag=${1:-rXX}
rm -rf /tmp/pmc_mfma
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pmc_mfma -o m -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-profile --no-gpu-legs --no-video --no-extra > /tmp/pmc_mfma.log 2>&1
tail -1 /tmp/pmc_mfma.log | cut -c1-200
python $root/tools/pmc_mfma_step.py $(find /tmp/pmc_mfma -name "*.db" | head -1) > $root/gpurun_out/${tag}_pmc_mfma_step.txt 2>&1
cat $root/gpurun_out/${tag}_pmc_mfma_step.txt
