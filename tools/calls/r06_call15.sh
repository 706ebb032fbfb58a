#!/bin/bash
# Round 6, last call: the whole GPU suite, the closing evidence, and the driver's own bench command (20 steps behind 5 warm-up steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06ad}; commit=${2:-unknown}
bash tools/full_gpu_suite.sh ${tag}
bash tools/closing_evidence.sh ${tag} ${commit}
python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{"metric' > gpurun_out/${tag}_bench_driver_cmd.json
python - <<PY
import json
b = json.load(open("gpurun_out/${tag}_bench_driver_cmd.json"))
print("driver command:", round(b["ms_per_step"], 2), "ms/step", round(b["value"], 3), b["unit"], "frac", round(b["roofline"]["frac"], 4), {k: round(v["ms_per_step"], 2) for k, v in b["kernel_families"].items()}, b["telemetry"]["sclk_mhz"]["mean"], b["telemetry"]["power_w"]["mean"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
