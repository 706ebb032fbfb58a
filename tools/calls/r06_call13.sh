#!/bin/bash
# Round 6, call 13: where does a K-step of gemm_dma_kernel go today?  Timing-only builds (csrc/vcx_ablate.h VCX_DMA_ABL, tools/build_abl.sh dmaN -DVCX_DMA_ABL=N)
# against the product on the long-K convolutions and the large linear layers (tools/gemm_quick.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06ab}
pick="problem\|conv3x3 C=\|linear 460800x320x1280\|geglu 115200\|linear 115200x640x2560\|linear 28800x1280x5120"
{
echo "## product"; python tools/gemm_quick.py auto 2>&1 | grep "$pick"
for v in 1 2 3 8 11 16 27; do
echo "## VCX_DMA_ABL=$v (1 no activation DMA, 2 no weight DMA, 8 no fragment reads, 16 no wait / barrier)"; VCX_LIB=tools/_abl/libvcx_dma$v.so python tools/gemm_quick.py auto 2>&1 | grep "$pick" | grep -v problem
done
echo "## product again"; python tools/gemm_quick.py auto 2>&1 | grep "$pick" | grep -v problem
} > gpurun_out/${tag}_dma_ablate.txt
cat gpurun_out/${tag}_dma_ablate.txt
