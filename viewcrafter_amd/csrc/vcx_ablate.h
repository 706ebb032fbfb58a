// Every compile-time switch of libvcx that is NOT the product, in one place.  The product build (csrc/Makefile) defines none of them:
// all are 0 / undefined here, and a kernel source reads `#if VCX_WS_ABL == 1`, never `#if defined(...)`.  tools/build_abl.sh builds a
// second library with one of them set into tools/_abl/ (git-ignored) for a same-box A/B or a timing-only ablation; results of the
// timing-only builds are garbage by construction.  (Retired A/B forms - the round 1-2 GELU, the compare / select GELU tail, other panel
// heights of the tile walk, the GroupNorm pixels-per-block knob, non-temporal GroupNorm accesses - live in the history, not here.)
#pragma once

// gemm_ws.hip
#ifndef VCX_WS_ABL      // gemm_ws320_kernel, timing only (tools/ws_ablate.py): 1 no MFMA work, 2 no epilogue
#define VCX_WS_ABL 0
#endif
#ifndef VCX_WG_ABL      // gemm_ws320_geglu_kernel, timing only (tools/ws_geglu_scan.py), bits: 1 no epilogue chunks, 2 no MFMAs, 4 no per-tile barrier
#define VCX_WG_ABL 0
#endif
#ifndef VCX_WL_ABL      // gemm_ws320_lnf_kernel, timing only (tools/ws_lnf_ab.py), bits: 1 no arithmetic chunks, 2 no MFMAs, 8 no stores
#define VCX_WL_ABL 0
#endif
// gemm_dma.hip
#ifndef VCX_CONV_XSKIP_ABL   // timing only: a 3x3 convolution fetches the activation rows of its horizontal centre taps only (what one LDS image per (slab, ky)
#define VCX_CONV_XSKIP_ABL 0 // serving all three kx by row-shifted fragment reads would save in DMA: the upper bound of that change)
#endif
#ifndef VCX_DMA_ABL          // gemm_dma_kernel's main loop, timing only (tools/calls/r06_call13.sh), bits: 1 no activation DMA / 2 no weight DMA behind the
#define VCX_DMA_ABL 0        // kernel's first K-step, 8 no fragment reads (the first K-step's fragments stay in registers), 16 no DMA wait / barrier per K-step
#endif
// attention.hip
#ifndef XABL            // xattn_resident2_d64_kernel, timing only (tools/xattn_ablate.py), bits: 1 no exp2, 2 no softmax arithmetic, 4 no MFMAs,
#define XABL 0          // 8 no Q loads / O stores, 16 no fragment reads, 32 no deferred-max test
#endif
// attention_v2.hip: -DVCX_FLASH2_ABLATIONS makes vcx_flash2_launch read the scratch knobs EXP0 (timing-only ablation bits of the key loop,
//   tools/flash_ablate.py) and EXP1 (stream variants / row sums on the matrix pipe, tools/flash_ab.py); undefined = ONE kernel, no knob read
// norm.hip: -DVCX_GN_TWO_PHASE = the round-5 statistics plumbing (partials + finalize kernels everywhere), for the same-box A/B of round 6
