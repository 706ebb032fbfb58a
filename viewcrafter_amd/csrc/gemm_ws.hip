// Weight-stationary linear layer for K = 320 and N = 320 j (level 0 of the UNet: attention output projections, SpatialTransformer /
// TemporalTransformer proj_in / proj_out; /root/reference lvdm/modules/attention.py:61-63,209,268,319,338 at 576x1024: M = 460800 token
// rows through a 320 x 320 weight block, 36 launches per DDIM step).
//
// These layers are memory-bound: 0.6 KB in + 0.6 KB out (+ 0.6 KB residual) per row against 205 kFLOP, i.e. 0.10-0.14 ms of HBM time and
// 0.04 ms of matrix time per call.  The tiled engine (gemm_dma.hip) runs them in 0.152 / 0.226 ms (without / with residual): per 256-row
// tile it streams the activation rows AND a 200 KB weight slice through LDS (55 % of its DMA bytes are weights it has fetched 1800 times
// before), its K-steps are paced by the LDS-DMA stream, and all 256 CUs alternate in lock step between a read phase and a write burst.
//
// MI355X-first alternative: a CU's register file is 512 KB - the whole 320 x 320 fp16 weight (200 KB) fits in it.  One block of four
// waves per CU (one per SIMD, 512 registers each); wave w keeps the MFMA A fragments of output columns 80 w .. 80 w + 79 for all ten
// 32-deep K slices in 200 registers for the lifetime of the block.  Only the activation rows move, as small tiles through an LDS ring
// filled by LDS-DMA several tiles ahead, ONE barrier per tile; every wave reads the tile's B fragments from LDS and owns an 80-column
// output strip.  Same MFMA shape, K order, epilogue arithmetic and access units as the tiled engine: the same bits
// (tests/test_kernels_gpu.py::test_gemm_weight_stationary_*).  Measured (profiles/r05o_ws_pipe_ab.txt, r05m_ws_bench_ab.txt): 0.131 / 0.170 ms
// in isolation, GEMM family -1.5 ms and step -1.35 ms in the benchmark.
//
//   gemm_ws320_pipe_kernel   bias / residual (32 launches per step): 32-row tiles, two accumulator sets - the finished tile's outputs are
//                            formed and stored inside the next tile's MFMA stream
//   gemm_ws320_kernel        per-image addend or column moments (VCX_GEMM_ROWADD / COLSTATS: the shared epilogue of gemm_epilogue.h),
//                            64-row tiles, tile after tile
//
// N = 320 j (j = 2, 3): a block owns ONE 320-column block of the weight for its lifetime; the j blocks that work on the same row tiles
// sit on the same XCD (block id = 8 slot + xcd, slot = j stream + column block), so the activation tile comes from HBM once and
// from that XCD's L2 for the others.
#include <type_traits>
#include <utility>
#include "gemm_epilogue.h"

using namespace vcxgemm;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFFFFu;

template <class F, int... I>
__device__ __forceinline__ void static_for_ws_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_ws(F&& f) { static_for_ws_impl(f, std::make_integer_sequence<int, N>{}); }

struct WsCfg {                      // the shape gemm_epilogue.h is instantiated for: one 64-row tile, four waves side by side
    static constexpr int TBM = 64, TBN = 320, NWM = 1, NWN = 4;
    static constexpr int THREADS = 256;
    [[maybe_unused]] static constexpr int MF = TBM / NWM / 16, NF = TBN / NWN / 16;        // 4 x 5 accumulator fragments per wave
};
[[maybe_unused]] constexpr int WS_K = 320, WS_KS = WS_K / 32;                               // ten 32-deep K slices
constexpr int WS_STAGE = WsCfg::TBM * WS_K * (int)sizeof(half_t);          // 40 KB
constexpr int WS_RING = 3;
[[maybe_unused]] constexpr int WS_PIECES = WS_STAGE / 1024 / 4;                             // LDS-DMA instructions per wave and tile (10)
constexpr size_t WS_STRIP = (size_t)WsCfg::TBN * sizeof(float);
constexpr size_t WS_SMEM = (size_t)WS_RING * WS_STAGE + 2 * WS_STRIP;

// Serial form, for the epilogues that only gemm_epilogue.h implements.  MODE 2: per-image addend (VCX_GEMM_ROWADD);  MODE 3: column
// moments (VCX_GEMM_COLSTATS).  The plain modes (bias, residual) run on gemm_ws320_pipe_kernel below.
template <int MODE>
__global__ void __launch_bounds__(WsCfg::THREADS, 1) gemm_ws320_kernel(GemmArgs p, unsigned a_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MF = WsCfg::MF, NF = WsCfg::NF;
    constexpr int LNF = MODE == 3 ? 3 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);

    const int ntiles = p.tiles_m;
    // block id = 8 slot + xcd (blocks are dealt round-robin over the 8 XCDs); slot = tiles_n * (row stream of the XCD) + column block
    const int cb = (blockIdx.x >> 3) % p.tiles_n;
    const int G = gridDim.x / p.tiles_n;                                          // row streams: stream s walks tiles s, s + G, ...
    const int t_first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) / p.tiles_n;
    const int ncol0 = cb * WsCfg::TBN;                                            // first output column of the block
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- the wave's weight slice: A fragments of its 80 output columns (rows of W) for all K, straight from global memory, once
    h8 wf[NF][WS_KS];
    {
        const half_t* wrow = p.W + (size_t)(ncol0 + wave * (NF * 16) + lr) * p.ldw + lg * 8;
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

    constexpr int P = 0;          // (no vector-memory request of this tile is older than the wait at its top: see the pipelined kernel for the general count)
    epi_u4v rres[(NF / 2 + NF % 2) * MF];

    int t = t_first;
    if (t < ntiles) issue_tile(t, 0);
    if (t + G < ntiles) issue_tile(t + G, 1);
    for (int i = 0; t < ntiles; t += G, ++i) {
        const int buf = i % WS_RING;
        // The vector-memory counter retires in order.  Younger than this tile's ten pieces are: from the second iteration on the previous
        // epilogue's output stores - EXACTLY 12 buffer stores
        // (two dwordx4 + one dwordx2 per 16-row group), plus 10 column-moment stores with COLSTATS (the ISA listing has them behind
        // an execz branch that is never taken: lanes with lr = 0 exist in every wave and every tile of a COLSTATS launch is a whole
        // 64-row strip) - and, if there is a next tile, its ten pieces.  The wait must leave ALL of those in flight: forcing even the
        // two oldest stores to be acknowledged here costs their full write latency in every iteration.
        __builtin_amdgcn_sched_barrier(0);
        {
#if defined(VCX_WS_ABL) && VCX_WS_ABL == 2
            constexpr int S = 0;
#else
            constexpr int S = LNF == 3 ? 22 : 12;
#endif
            const bool next_in_flight = t + G < ntiles;        // (issued in the prologue or at the end of the previous iteration)
            if (i == 0) {
                if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 10) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
            } else {
                if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + S + 10) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + S) : "memory");
            }
        }
        __builtin_amdgcn_s_barrier();            // every wave's pieces have landed; every wave is done with the tile before last
        __builtin_amdgcn_sched_barrier(0);
        const half_t* cx = reinterpret_cast<const half_t*>(smem_raw + buf * WS_STAGE);
        // this tile's residual pieces, requested ahead of its MFMAs: there when the epilogue starts (gemm_epilogue.h)
        if (p.flags & VCX_GEMM_RESIDUAL) gemm_epilogue_fetch_residual<WsCfg>(p, t, cb, 0, wave, lane, rres);
        __builtin_amdgcn_sched_barrier(0);       // requested HERE: left to itself hipcc sinks these loads to their first use, the tail of the MFMA stream
        f4 acc[NF][MF];
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < MF; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        h8 xf[MF], xn[MF];
#pragma unroll
        for (int b = 0; b < MF; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(b * 16 + lr, lg));
#if defined(VCX_WS_ABL) && VCX_WS_ABL == 1        // tools/ws_ablate.py: no MFMA work (timing only)
        for (int kk = 0; kk < 0; ++kk) {
#else
#pragma unroll
        for (int kk = 0; kk < WS_KS; ++kk) {
#endif
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
#if defined(VCX_WS_ABL) && VCX_WS_ABL == 2        // tools/ws_ablate.py: no epilogue (timing only; one store keeps the MFMAs alive)
        if (acc[0][0][0] == 12345.678f) *reinterpret_cast<float*>(p.C) = acc[1][1][1] + acc[4][3][2];
#else
        {
            // (the bias strip is the same for every tile of this kernel: written by the first epilogue, kept - unless a per-image addend rides in it)
            gemm_epilogue<WsCfg, false, false, LNF, true>(p, acc, t, cb, 0, wave, lane, sB, sS, nullptr, nullptr, rres, i > 0 && !(p.flags & VCX_GEMM_ROWADD));
        }
#endif
        // the tile after next goes into the stage that the PREVIOUS tile used: every wave has passed this iteration's barrier, i.e. has
        // finished reading it.  Issued behind the epilogue's stores, so that the counts above hold.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (t + 2 * G < ntiles) issue_tile(t + 2 * G, (i + 2) % WS_RING);
    }
#endif
}


// =============================================================================================================================
// The lean modes (bias at most, optional residual) once more, with the epilogue of tile i UNDER the MFMAs of tile i + 1.
//
// One wave per SIMD executes its tile serially: wait, 200 MFMAs, ~300 epilogue instructions, 12 stores - 3.9 us per 64-row tile, of
// which the matrix pipe works 1.5 (tools/ws_ablate.py).  Nothing else is resident on the SIMD to fill the rest, so the wave overlaps with
// itself: 32-row tiles (half the accumulators: 40 registers, so TWO sets fit beside the 200 weight registers), and while the MFMAs of
// tile i fill one set, the other set - tile i - 1, complete - is finished unit by unit (two fragments x 16 rows: multiply-add, residual,
// fp16 rounding, v_permlane16_swap, one store) in the same instruction stream, six units over the ten K slices.  This kernel has what
// the deferred GEGLU epilogue of section 1 of profiles/r05_experiments.md lacked: a matrix pipe that is two thirds idle and an issue
// port with nothing else to do.  Ring of five 20 KB stages, three tiles ahead; residual pieces requested when a tile's MFMAs start and
// used one tile later.  The vector-memory wait at the top of a tile is computed, not assumed: the wave counts the operations it
// issues, remembers the count behind each tile's DMA pieces, and waits for all but (issued - mark) - exact whatever the mode, the
// position in the stream and the tail.
// =============================================================================================================================
struct WpCfg {                      // 32-row tile, four waves side by side (the shape gemm_epilogue_fetch_residual is instantiated for)
    static constexpr int TBM = 32, TBN = 320, NWM = 1, NWN = 4;
    static constexpr int THREADS = 256;
    [[maybe_unused]] static constexpr int MF = TBM / NWM / 16, NF = TBN / NWN / 16;        // 2 x 5 accumulator fragments per wave
};
constexpr int WP_STAGE = WpCfg::TBM * WS_K * (int)sizeof(half_t);          // 20 KB: five [32 rows][64] slabs
[[maybe_unused]] constexpr int WP_RING = 5, WP_AHEAD = 3;
[[maybe_unused]] constexpr int WP_PIECES = WP_STAGE / 1024 / 4;            // LDS-DMA instructions per wave and tile (5)
constexpr size_t WP_SMEM = (size_t)WP_RING * WP_STAGE;

__device__ __forceinline__ void ws_wait_vmcnt(int n) {        // s_waitcnt vmcnt(n) for a wave-uniform run-time n (the count is an immediate)
    switch (n < 63 ? n : 63) {
#define VCX_WS_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define VCX_WS_W8(k) VCX_WS_W(k) VCX_WS_W(k + 1) VCX_WS_W(k + 2) VCX_WS_W(k + 3) VCX_WS_W(k + 4) VCX_WS_W(k + 5) VCX_WS_W(k + 6) VCX_WS_W(k + 7)
        VCX_WS_W8(0) VCX_WS_W8(8) VCX_WS_W8(16) VCX_WS_W8(24) VCX_WS_W8(32) VCX_WS_W8(40) VCX_WS_W8(48) VCX_WS_W8(56)
#undef VCX_WS_W8
#undef VCX_WS_W
    }
}

template <int I> using WInt = std::integral_constant<int, I>;

template <bool RES>
__global__ void __launch_bounds__(WpCfg::THREADS, 1) gemm_ws320_pipe_kernel(GemmArgs p, unsigned a_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MF = WpCfg::MF, NF = WpCfg::NF, UNITS = NF / 2 + NF % 2, NUNIT = UNITS * MF;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);

    int ntiles = p.tiles_m;
    const int cb = (blockIdx.x >> 3) % p.tiles_n;
    int G = gridDim.x / p.tiles_n;
    int t_first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) / p.tiles_n;
    if constexpr (!RES) {
        // vcx_gemm_units_f16: one weight / bias set per unit of unit_rows rows (a GroupNorm folded into this projection has one per
        // frame or per video).  gridDim.x / units consecutive blocks share a unit; a block keeps that unit's weights for its
        // lifetime and walks the unit's row tiles only (N = 320: one column block; unit_rows % 32 == 0: no tile straddles two units).
        if (p.unit_rows > 0) {
            const int bpu = gridDim.x / p.units, unit = blockIdx.x / bpu;
            p.m_begin += unit * p.unit_rows;
            p.M = p.m_begin + p.unit_rows;
            p.W += (int64_t)unit * p.w_unit_stride;
            if (p.flags & VCX_GEMM_BIAS_N) p.bias += (int64_t)unit * p.bias_unit_stride;
            ntiles = p.unit_rows / WpCfg::TBM;
            G = bpu;
            t_first = blockIdx.x % bpu;
        }
    }
    const int ncol0 = cb * WpCfg::TBN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;

    h8 wf[NF][WS_KS];
    f4 bv[NF];
    {
        const half_t* wrow = p.W + (size_t)(ncol0 + wave * (NF * 16) + lr) * p.ldw + lg * 8;
#pragma unroll
        for (int a = 0; a < NF; ++a) {
#pragma unroll
            for (int kk = 0; kk < WS_KS; ++kk) wf[a][kk] = *reinterpret_cast<const h8*>(wrow + (size_t)a * 16 * p.ldw + kk * 32);
            bv[a] = (p.flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + ncol0 + wave * (NF * 16) + a * 16 + lg * 4) : f4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);       // (with the builtin: hipcc's wait-count bookkeeping must see it - see gemm_ws320_kernel)
    }

    // LDS-DMA of one 32-row tile: 20 instructions of 8 rows x 128 bytes; wave w issues slab i = 0..4, rows 8 w .. 8 w + 7
    const int drow = wave * 8 + (lane >> 3);
    const unsigned dsrc = (unsigned)((lane & 7) ^ ((drow >> 1) & 7)) * 16u;
    int issued = 0;                           // vector-memory operations this wave has issued so far (program order = retirement order)
    int mark[WP_RING];                        // ... as of the last DMA piece of the tile in each stage
    auto issue_tile = [&](int t, int buf) {
        const int m0 = p.m_begin + t * WpCfg::TBM + drow;
        const unsigned v = m0 < p.M ? (unsigned)m0 * (unsigned)p.lda * 2u + dsrc : OOB;
        unsigned char* dst = smem_raw + buf * WP_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < WP_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dst + i * 4096), 16, v, (unsigned)i * (BK * 2), 0, 0);
        issued += WP_PIECES;
        mark[buf] = issued;
    };

    f4 acc[2][NF][MF];
    [[maybe_unused]] epi_u4v rr[2][RES ? NUNIT : 1];
    unsigned coff[2] = {0u, 0u};              // byte offset of (lane's row in group 0, wave's first column) of the tile in each accumulator set
    const unsigned odd = lg & 1, half = lg >> 1;
    const unsigned cstep = 32u * (unsigned)p.ldc;
    const float alpha = p.alpha;

    // unit J = b UNITS + u of the tile held in accumulator set PAR: the arithmetic, access units and store order of the plain path of
    // gemm_epilogue (and of gemm_ws320_kernel's lean epilogue) - the same bits
    auto unit = [&](auto PAR_, auto J_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, J = decltype(J_)::value, b = J / UNITS, u = J % UNITS;
        constexpr bool wide = u < NF / 2;
        constexpr int a = 2 * u;
        float v0[4], v1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[r] = __builtin_fmaf(acc[PAR][a][b][r], alpha, bv[a][r]);
            if (wide) v1[r] = __builtin_fmaf(acc[PAR][wide ? a + 1 : a][b][r], alpha, bv[wide ? a + 1 : a][r]);
        }
        if constexpr (RES) {
            const epi_u4v raw = rr[PAR][J];
            unsigned w0 = raw[0], w1 = raw[1], w2 = raw[2], w3 = raw[3];
            if (wide) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(w0, w2, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(w1, w3, false, false);
                w0 = s0[0]; w2 = s0[1]; w1 = s1[0]; w3 = s1[1];
            }
            const h4 r0 = __builtin_bit_cast(h4, u2v{w0, w1});
            const h4 r1 = __builtin_bit_cast(h4, u2v{w2, w3});
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] += (float)r0[r]; v1[r] += (float)r1[r]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(v0[r]), "+v"(v1[r]));      // fp32 first, then ONE fp16 rounding (no v_fma_mixlo_f16)
        const u2v p0 = __builtin_bit_cast(u2v, h4{(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3]});
        if constexpr (wide) {
            const u2v p1 = __builtin_bit_cast(u2v, h4{(half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]});
            const unsigned a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            const unsigned col = (2 * u + odd) * 16 + half * 8;
            __builtin_amdgcn_raw_buffer_store_b128(epi_u4v{s0[0], s1[0], s0[1], s1[1]}, srd_c, coff[PAR] + (unsigned)b * cstep + col * 2u, 0, 0);
        } else {
            const unsigned col = (unsigned)a * 16 + (unsigned)lg * 4;
            __builtin_amdgcn_raw_buffer_store_b64(p0, srd_c, coff[PAR] + (unsigned)b * cstep + col * 2u, 0, 0);
        }
        issued += 1;
    };

    // one tile: its MFMAs into accumulator set PAR; PEND: the other set holds a finished tile - its units ride along
    auto tile = [&](auto PAR_, auto PEND_, int t, int i) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool PEND = decltype(PEND_)::value != 0;
        const int buf = i % WP_RING;
        __builtin_amdgcn_sched_barrier(0);
        ws_wait_vmcnt(issued - mark[buf]);           // everything up to this tile's last DMA piece has retired
        __builtin_amdgcn_s_barrier();                // ... for every wave; and every wave is done with the tiles before
        __builtin_amdgcn_sched_barrier(0);
        if (t + WP_AHEAD * G < ntiles) issue_tile(t + WP_AHEAD * G, (i + WP_AHEAD) % WP_RING);      // into the stage of tile i - 2: long consumed
        if constexpr (RES) {
            gemm_epilogue_fetch_residual<WpCfg>(p, t, cb, 0, wave, lane, rr[PAR]);
            issued += NUNIT;
        }
        coff[PAR] = ((unsigned)(p.m_begin + t * WpCfg::TBM + lr) * (unsigned)p.ldc + (unsigned)(ncol0 + wave * (NF * 16))) * 2u;
        __builtin_amdgcn_sched_barrier(0);
        const half_t* cx = reinterpret_cast<const half_t*>(smem_raw + buf * WP_STAGE);
        h8 xf[MF], xn[MF];
#pragma unroll
        for (int b = 0; b < MF; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(b * 16 + lr, lg));
        static_for_ws<WS_KS>([&](auto KK_) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK_)::value;
            if constexpr (kk + 1 < WS_KS) {
#pragma unroll
                for (int b = 0; b < MF; ++b)
                    xn[b] = *reinterpret_cast<const h8*>(cx + ((kk + 1) >> 1) * (WpCfg::TBM * BK) + lds_off(b * 16 + lr, ((kk + 1) & 1) * 4 + lg));
            }
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < MF; ++b) {
                    if constexpr (kk == 0) acc[PAR][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a][kk], xf[b], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[PAR][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a][kk], xf[b], acc[PAR][a][b], 0, 0, 0);
                }
            if constexpr (kk + 1 < WS_KS) {
#pragma unroll
                for (int b = 0; b < MF; ++b) xf[b] = xn[b];
            }
            // six units over ten K slices: slices 1, 2, 4, 5, 7, 8 (the first leaves the residual request of the PREVIOUS tile one more
            // slice; the last keeps the stores clear of the barrier)
            if constexpr (PEND && kk % 3 != 0 && kk < 9) unit(WInt<PAR ^ 1>{}, WInt<(kk / 3) * 2 + (kk % 3) - 1>{});
        });
    };

    int t = t_first, i = 0;
#pragma unroll
    for (int k = 0; k < WP_AHEAD; ++k)
        if (t + k * G < ntiles) issue_tile(t + k * G, k);
    if (t < ntiles) {
        tile(WInt<0>{}, WInt<0>{}, t, i);
        t += G; ++i;
        for (;;) {
            if (t >= ntiles) {
                static_for_ws<NUNIT>([&](auto J_) __attribute__((always_inline)) { unit(WInt<0>{}, J_); });
                break;
            }
            tile(WInt<1>{}, WInt<1>{}, t, i);
            t += G; ++i;
            if (t >= ntiles) {
                static_for_ws<NUNIT>([&](auto J_) __attribute__((always_inline)) { unit(WInt<1>{}, J_); });
                break;
            }
            tile(WInt<0>{}, WInt<1>{}, t, i);
            t += G; ++i;
        }
    }
#endif
}

template <bool RES>
int launch_ws_pipe(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_pipe_kernel<RES>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WP_SMEM, "vcx_gemm_f16(ws320 pipe)")) return VCX_ELAUNCH;
    const int per_xcd = persistent_grid(1 << 30, 1) / 8;
    int streams_per_xcd = per_xcd / a.tiles_n;
    const int needed = (a.tiles_m + 7) / 8;
    if (streams_per_xcd > needed) streams_per_xcd = needed;
    hipLaunchKernelGGL(kern, dim3(8 * streams_per_xcd * a.tiles_n), dim3(WpCfg::THREADS), WP_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320 pipe)");
}

template <int MODE>
int launch_ws(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_kernel<MODE>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WS_SMEM, "vcx_gemm_f16(ws320)")) return VCX_ELAUNCH;
    // 8 XCDs x (slots per XCD) blocks; a slot group = the tiles_n column blocks of one row stream.  Fewer streams than the chip
    // has room for when the problem has fewer row tiles.
    const int per_xcd = persistent_grid(1 << 30, 1) / 8;
    int streams_per_xcd = per_xcd / a.tiles_n;
    const int needed = (a.tiles_m + 7) / 8;
    if (streams_per_xcd > needed) streams_per_xcd = needed;
    const int nb = 8 * streams_per_xcd * a.tiles_n;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(WsCfg::THREADS), WS_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320)");
}

}  // namespace

// Linear mode, K = 320, N = 320 j (j <= 4), fp16 output, no GEGLU / LNFOLD / LNFOLD_T / BIAS_M,
// 32-bit operand and output extents (the caller checks; it also fills a_bytes / c_bytes / r_bytes).  Sets the tiling itself.
// One launch, `units` weight / bias sets (vcx_gemm_units_f16): N = K = 320, bias at most, unit_rows % 32 == 0 (the caller checks).
int vcxgemm::launch_ws320_units(GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_pipe_kernel<false>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WP_SMEM, "vcx_gemm_units_f16(ws320 pipe)")) return VCX_ELAUNCH;
    a.tiles_n = 1;
    a.tiles_m = a.unit_rows / WpCfg::TBM;
    const int cus = persistent_grid(1 << 30, 1);
    int bpu = cus / a.units;                        // blocks per unit: the chip's CUs shared out, at least one, at most one per row tile
    if (bpu < 1) bpu = 1;
    if (bpu > a.tiles_m) bpu = a.tiles_m;
    hipLaunchKernelGGL(kern, dim3(bpu * a.units), dim3(WpCfg::THREADS), WP_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_units_f16(ws320 pipe)");
}

int vcxgemm::launch_ws320(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M - a.m_begin + WsCfg::TBM - 1) / WsCfg::TBM;
    a.tiles_n = a.N / WsCfg::TBN;
    if (a.flags & VCX_GEMM_COLSTATS) return launch_ws<3>(a, s);
    if (a.flags & VCX_GEMM_ROWADD) return launch_ws<2>(a, s);
    a.tiles_m = (a.M - a.m_begin + WpCfg::TBM - 1) / WpCfg::TBM;
    return (a.flags & VCX_GEMM_RESIDUAL) ? launch_ws_pipe<true>(a, s) : launch_ws_pipe<false>(a, s);
}
