// Weight-stationary linear layer for N = K = 320 (level 0 of the UNet: attention output projections, SpatialTransformer /
// TemporalTransformer proj_in / proj_out; /root/reference lvdm/modules/attention.py:61,209,268,319,338 at 576x1024: M = 460800 token
// rows through a 320 x 320 weight, 36 launches per DDIM step).
//
// These layers are memory-bound: 0.6 KB in + 0.6 KB out (+ 0.6 KB residual) per row against 205 kFLOP, i.e. 0.16 ms of HBM time and
// 0.04 ms of matrix time per call.  The tiled engine (gemm_dma.hip) runs them at 4.0 TB/s (0.223 ms): per 256-row tile it streams the
// activation rows AND a 200 KB weight slice through LDS (55 % of its DMA bytes are weights it has fetched 1800 times before), its
// K-steps are paced by the LDS-DMA stream, and all 256 CUs alternate in lock step between a read phase and a write burst.
//
// MI355X-first alternative: a CU's register file is 512 KB - the whole 320 x 320 fp16 weight (200 KB) fits in it.  One block of four
// waves per CU (one per SIMD, 512 registers each); wave w keeps the MFMA A fragments of output columns 80 w .. 80 w + 79 for all ten
// 32-deep K slices in 200 registers for the lifetime of the block.  Only the activation rows move: 64-row tiles (40 KB) through a
// three-deep LDS ring by LDS-DMA, two tiles ahead, ONE barrier per tile; every wave reads the tile's B fragments from LDS (40 reads of
// 1 KB per 200 MFMAs) and owns a 64 x 80 output strip, which goes through the shared epilogue of gemm_epilogue.h (bias, per-image
// addend, residual, column moments for the GroupNorm behind - same accumulator layout as the tiled engine, so the results are the same
// bits).  No weight traffic after the first 200 KB per CU, 5.7 us of memory time per tile against 1.9 us of matrix time.
#include "gemm_epilogue.h"

using namespace vcxgemm;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFFFFu;

struct WsCfg {                      // the shape gemm_epilogue.h is instantiated for: one 64-row tile, four waves side by side
    static constexpr int TBM = 64, TBN = 320, NWM = 1, NWN = 4;
    static constexpr int THREADS = 256;
    static constexpr int MF = TBM / NWM / 16, NF = TBN / NWN / 16;        // 4 x 5 accumulator fragments per wave
};
[[maybe_unused]] constexpr int WS_K = 320, WS_KS = WS_K / 32;                               // ten 32-deep K slices
constexpr int WS_STAGE = WsCfg::TBM * WS_K * (int)sizeof(half_t);          // 40 KB
constexpr int WS_RING = 3;
[[maybe_unused]] constexpr int WS_PIECES = WS_STAGE / 1024 / 4;                             // LDS-DMA instructions per wave and tile (10)
constexpr size_t WS_STRIP = (size_t)WsCfg::TBN * sizeof(float);
constexpr size_t WS_SMEM = (size_t)WS_RING * WS_STAGE + 2 * WS_STRIP;

template <int LNF>      // 0 plain epilogue, 3 VCX_GEMM_COLSTATS
__global__ void __launch_bounds__(WsCfg::THREADS, 1) gemm_ws320_kernel(GemmArgs p, unsigned a_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MF = WsCfg::MF, NF = WsCfg::NF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);

    const int ntiles = p.tiles_m;
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- the wave's weight slice: A fragments of its 80 output columns (rows of W) for all K, straight from global memory, once
    h8 wf[NF][WS_KS];
    {
        const half_t* wrow = p.W + (size_t)(wave * (NF * 16) + lr) * p.ldw + lg * 8;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int kk = 0; kk < WS_KS; ++kk) wf[a][kk] = *reinterpret_cast<const h8*>(wrow + (size_t)a * 16 * p.ldw + kk * 32);
        // waited for HERE, with the builtin (which hipcc's wait-count bookkeeping sees): otherwise it waits for these fifty loads at
        // their first uses inside the tile loop - vmcnt(49) ... vmcnt(0) in every iteration, the last of which would drain the next
        // tile's LDS-DMA in the middle of this tile's MFMAs
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }

    // ---- LDS-DMA of one 64-row tile: 40 instructions of 8 rows x 128 bytes; wave w issues piece i = 0..9 = slab i / 2, rows
    // 8 (w + 4 (i & 1)) .. + 7.  Lane l fetches the 16-byte chunk that the XOR swizzle puts at position l & 7 of row l >> 3.
    const int drow = wave * 8 + (lane >> 3);                                     // tile row of the even pieces; odd pieces: + 32 (same swizzle term)
    const unsigned dsrc = (unsigned)((lane & 7) ^ ((drow >> 1) & 7)) * 16u;
    const unsigned rstep32 = 32u * (unsigned)p.lda * 2u;
    auto issue_tile = [&](int t, int buf) {
        const int m0 = p.m_begin + t * WsCfg::TBM + drow;
        const unsigned base = (unsigned)m0 * (unsigned)p.lda * 2u + dsrc;            // < 4 GiB for every row < M (checked by the caller)
        const unsigned va = m0 < p.M ? base : OOB, vb = m0 + 32 < p.M ? base + rstep32 : OOB;
        unsigned char* dst = smem_raw + buf * WS_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < WS_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dst + (i >> 1) * 8192 + (i & 1) * 4096), 16, (i & 1) ? vb : va,
                                                     (unsigned)(i >> 1) * (BK * 2), 0, 0);
    };

    float* sB = reinterpret_cast<float*>(smem_raw + WS_RING * WS_STAGE) + wave * (NF * 16);      // the wave's strip of column addends
    float* sS = sB + WsCfg::TBN;

    int t = blockIdx.x;
    if (t < ntiles) issue_tile(t, 0);
    if (t + G < ntiles) issue_tile(t + G, 1);
    for (int i = 0; t < ntiles; t += G, ++i) {
        const int buf = i % WS_RING;
        // The vector-memory counter retires in order.  Younger than this tile's ten pieces are: from the second iteration on the previous
        // epilogue's >= 12 output stores, and - if there is a next tile - its ten pieces.  Waiting for all but 10 (+ 10) operations
        // therefore covers this tile's pieces and leaves the next tile's, and most of the stores, in flight.
        __builtin_amdgcn_sched_barrier(0);
        {
            const bool next_in_flight = t + G < ntiles;        // (issued in the prologue or at the end of the previous iteration)
            if (i == 0) {
                if (next_in_flight) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (next_in_flight) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();            // every wave's pieces have landed; every wave is done with the tile before last
        __builtin_amdgcn_sched_barrier(0);
        const half_t* cx = reinterpret_cast<const half_t*>(smem_raw + buf * WS_STAGE);
        // the tile's residual pieces, requested ahead of the MFMAs: there when the epilogue starts (gemm_epilogue.h)
        epi_u4v rres[(NF / 2 + NF % 2) * MF];
        if (p.flags & VCX_GEMM_RESIDUAL) gemm_epilogue_fetch_residual<WsCfg>(p, t, 0, 0, wave, lane, rres);
        f4 acc[NF][MF];
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < MF; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        h8 xf[MF], xn[MF];
#pragma unroll
        for (int b = 0; b < MF; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(b * 16 + lr, lg));
#pragma unroll
        for (int kk = 0; kk < WS_KS; ++kk) {
            if (kk + 1 < WS_KS) {
#pragma unroll
                for (int b = 0; b < MF; ++b)
                    xn[b] = *reinterpret_cast<const h8*>(cx + ((kk + 1) >> 1) * (WsCfg::TBM * BK) + lds_off(b * 16 + lr, ((kk + 1) & 1) * 4 + lg));
            }
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < MF; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a][kk], xf[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < MF; ++b) xf[b] = xn[b];
        }
        // (the bias strip is the same for every tile of this kernel: written by the first epilogue, kept - unless a per-image addend rides in it)
        gemm_epilogue<WsCfg, false, false, LNF, true>(p, acc, t, 0, 0, wave, lane, sB, sS, nullptr, nullptr, rres, i > 0 && !(p.flags & VCX_GEMM_ROWADD));
        // the tile after next goes into the stage that the PREVIOUS tile used: every wave has passed this iteration's barrier, i.e. has
        // finished reading it.  Issued behind the epilogue's stores, so that the counts above hold.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (t + 2 * G < ntiles) issue_tile(t + 2 * G, (i + 2) % WS_RING);
    }
#endif
}

template <int LNF>
int launch_ws(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_kernel<LNF>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WS_SMEM, "vcx_gemm_f16(ws320)")) return VCX_ELAUNCH;
    const int nb = persistent_grid(a.tiles_m, 1);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(WsCfg::THREADS), WS_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320)");
}

}  // namespace

// Linear mode, N = K = 320, fp16 output, no GEGLU / LNFOLD / BIAS_M, 32-bit operand and output extents (the caller checks; it also fills
// a_bytes / c_bytes / r_bytes).  Sets the 64-row tiling itself.
int vcxgemm::launch_ws320(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M - a.m_begin + WsCfg::TBM - 1) / WsCfg::TBM;
    a.tiles_n = 1;
    return (a.flags & VCX_GEMM_COLSTATS) ? launch_ws<3>(a, s) : launch_ws<0>(a, s);
}
