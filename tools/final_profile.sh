#!/bin/bash
# Regenerate the per-round evidence under gpurun_out/ (copy what should be judged into profiles/):
#   tools/final_profile.sh r01h      -> gpurun_out/r01h_{bench,bench_under_rocprof}.json, r01h_kernel_stats.txt
# Run on the GPU box from the repo root (gpurun -- 'bash tools/final_profile.sh r01h').
set -u
tag=${1:-rXX}
root=$(pwd)
mkdir -p gpurun_out
python bench.py 2>&1 | grep '^{"metric' > gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${tag}
rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag} -o ${tag} -- python ${root}/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-gpu-legs --no-video --no-extra > /tmp/bench_prof_${tag}.log 2>&1
cd ${root}
grep '^{"metric' /tmp/bench_prof_${tag}.log > gpurun_out/${tag}_bench_under_rocprof.json
python tools/rocprof_summary.py $(find /tmp/prof_${tag} -name "*.db" | head -1) > gpurun_out/${tag}_kernel_stats.txt 2>&1
python - <<PY
import json
b = json.load(open("gpurun_out/${tag}_bench.json")); r = json.load(open("gpurun_out/${tag}_bench_under_rocprof.json"))
print("bench        ", round(b["ms_per_step"], 2), "ms/step", round(b["value"], 3), b["unit"], {k: round(v["ms_per_step"], 2) for k, v in b["kernel_families"].items()})
print("under rocprof", round(r["ms_per_step"], 2), "ms/step", {k: round(v["ms_per_step"], 2) for k, v in r["kernel_families"].items()})
tot = 0.0
for l in open("gpurun_out/${tag}_kernel_stats.txt"):
    f = l.split()
    if len(f) > 10 and "gemm" in f[-1] and f[0].isdigit():
        tot += float(f[1])
print("GEMM kernels in the trace:", round(tot / 4, 2), "ms per step (4 steps traced)")
PY
