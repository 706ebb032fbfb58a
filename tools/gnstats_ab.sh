#!/bin/bash
# Same-box A/B of GroupNorm statistics from the conv epilogue (VCX_GN_EPILOGUE_STATS=0 | 1): bench.py twice each, interleaved.
#   gpurun -- 'bash tools/gnstats_ab.sh r03w'
tag=${1:-rXX}
mkdir -p gpurun_out
for rep in 1 2; do
  for f in 0 1; do
    VCX_GN_EPILOGUE_STATS=$f python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-legs --no-video --no-extra --no-decode 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_families']
print('VCX_GN_EPILOGUE_STATS=$f rep $rep: %.2f ms/step | gemm %.2f (%d) gn %.2f (%d launches) ln %.2f flash %.2f' % (d['ms_per_step'], k['gemm']['ms_per_step'], k['gemm']['launches_per_step'], k['groupnorm']['ms_per_step'], k['groupnorm']['launches_per_step'], k['layernorm']['ms_per_step'], k['flash_attn']['ms_per_step']))"
  done
done | tee gpurun_out/${tag}_gnstats_ab.txt
