#!/bin/bash
# HBM traffic of the kernel families: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes gpurun_out/<tag>_pmc_traffic.{txt,json}; copy the json to
# profiles/pmc_traffic.json (bench.py reads it and checks the recorded kernel-source hash).
#   gpurun -- 'bash tools/pmc_passes.sh r02 <commit>'
set -u
tag=${1:-rXX}
commit=${2:-unknown}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python ${root}/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-profile --no-gpu-legs --no-video --no-extra > /tmp/pmc_$c.log 2>&1
  tail -2 /tmp/pmc_$c.log
done
cd ${root}
# algorithmic GEMM bytes per step from the bench line of the FETCH pass (kernel_families.gemm.gbps * ms)
algo=$(grep '^{"metric' /tmp/pmc_FETCH_SIZE.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['kernel_families']['gemm']; print(g['gbps']*1e9*g['ms_per_step']*1e-3)")
calls=$(grep '^{"metric' /tmp/pmc_FETCH_SIZE.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_families']['gemm']['launches_per_step'])")
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) \
  --json gpurun_out/${tag}_pmc_traffic.json --algo-bytes-per-step ${algo} --gemm-calls-per-step ${calls} --commit ${commit} | tee gpurun_out/${tag}_pmc_traffic.txt
