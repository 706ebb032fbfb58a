#!/bin/bash
# (needs csrc/norm.hip built with -DVCX_EXPERIMENT_KNOBS: the product library ignores the knob)
# In-situ A/B of the GroupNorm apply block size: VCX_TUNE_EXP0=-1 (rule of rounds 1-3: >= 128 pixels per block) vs default (32 / 16).
#   gpurun -- 'bash tools/gn_apply_ab.sh r04l'
tag=${1:-rXX}
mkdir -p gpurun_out
for rep in 1 2; do
  for f in -1 0; do
    VCX_TUNE_EXP0=$f python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-legs --no-video --no-extra --no-decode 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_families']
print('VCX_TUNE_EXP0=$f rep $rep: %.2f ms/step | gemm %.2f gn %.2f (%d launches, %.2f TB/s) ln %.2f flash %.2f elementwise %.2f  tune %s' % (d['ms_per_step'], k['gemm']['ms_per_step'], k['groupnorm']['ms_per_step'], k['groupnorm']['launches_per_step'], k['groupnorm']['gbps']/1e3, k['layernorm']['ms_per_step'], k['flash_attn']['ms_per_step'], k['elementwise']['ms_per_step'], d['tune']))"
  done
done | tee gpurun_out/${tag}_gn_apply_ab.txt
