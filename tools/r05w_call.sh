#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/xattn_debug2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05w_xdbg2.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "geglu or gelu or resident or dual or cross_attention or folded or groupnorm" > gpurun_out/r05w_kernels.log 2>&1; echo "kernels rc=$?" >> gpurun_out/r05w_kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -m gpu > gpurun_out/r05w_model.log 2>&1; echo "model rc=$?" >> gpurun_out/r05w_model.log
VCX_TUNE_XATTN_RESIDENT=2 timeout 300 python tools/attn_shapes.py 2>&1 | grep "dual\|totals" > gpurun_out/r05w_attn_shapes_form1.txt
timeout 300 python tools/attn_shapes.py 2>&1 | grep "dual\|totals" > gpurun_out/r05w_attn_shapes_form2.txt
timeout 600 python tools/step_ab.py --rounds 3 --steps 3 base:gnfold=0,xattn=2 gnfold:gnfold=1,xattn=2 xattn2:gnfold=0,xattn=1 all:gnfold=1,xattn=1 2>&1 | grep -v amdgpu.ids > gpurun_out/r05w_step_ab.txt
cat gpurun_out/r05w_xdbg2.txt; tail -n 4 gpurun_out/r05w_kernels.log; tail -n 4 gpurun_out/r05w_model.log; cat gpurun_out/r05w_attn_shapes_form1.txt gpurun_out/r05w_attn_shapes_form2.txt; cat gpurun_out/r05w_step_ab.txt
