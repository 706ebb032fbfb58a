#!/bin/bash
# Same-box A/B of where the GroupNorm statistics come from (VCX_GN_EPILOGUE_STATS = 0 statistics pass | 1 conv epilogues inside a
# ResBlock, round 3 | 2 moments across module boundaries + concat written in place, round 4): bench.py twice each, interleaved.
#   gpurun -- 'bash tools/gnstats_ab.sh r04c "1 2"'
tag=${1:-rXX}
levels=${2:-"0 1 2"}
mkdir -p gpurun_out
for rep in 1 2; do
  for f in $levels; do
    VCX_GN_EPILOGUE_STATS=$f python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-legs --no-video --no-extra --no-decode 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_families']
print('VCX_GN_EPILOGUE_STATS=$f rep $rep: %.2f ms/step | gemm %.2f (%d) gn %.2f (%d launches) ln %.2f flash %.2f elementwise %.2f (%d)' % (d['ms_per_step'], k['gemm']['ms_per_step'], k['gemm']['launches_per_step'], k['groupnorm']['ms_per_step'], k['groupnorm']['launches_per_step'], k['layernorm']['ms_per_step'], k['flash_attn']['ms_per_step'], k['elementwise']['ms_per_step'], k['elementwise']['launches_per_step']))"
  done
done | tee gpurun_out/${tag}_gnstats_ab.txt
