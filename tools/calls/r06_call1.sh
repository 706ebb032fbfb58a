#!/bin/bash
# Round 6, call 1: what the box offers for telemetry; filler prices for the flash softmax stream; flash stream variants A/B; flash tests;
# the default bench (baseline of this box, with clock / power telemetry)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06a}
python tools/telemetry.py > gpurun_out/${tag}_telemetry_probe.txt 2>&1
timeout 120 tools/_abl/ubench_fill r6 > gpurun_out/${tag}_ubench_fill.txt 2>&1
timeout 300 python tools/flash_ab.py 5 tools/_abl/libvcx_flash.so 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_flash_variants_ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "flash or attn" 2>&1 | tail -5 > gpurun_out/${tag}_flash_tests.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -3 gpurun_out/${tag}_telemetry_probe.txt; cat gpurun_out/${tag}_ubench_fill.txt | head -70; cat gpurun_out/${tag}_flash_variants_ab.txt; cat gpurun_out/${tag}_flash_tests.txt
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","telemetry")})
print({k:round(v["ms_per_step"],2) for k,v in d["kernel_families"].items()})
print(d["extra"]["two_clips_per_gpu"])
print(d.get("sec_per_video"), d["roofline"]["frac"], d["roofline_flash"]["achieved"])
PY
