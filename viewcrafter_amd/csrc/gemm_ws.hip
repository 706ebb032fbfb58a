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

// ---- shared by the four kernels of this file (review r5 item 8: one place for the block map, the stream walk and the grid)
// The block -> (column block cb, row stream) map.  G = gridDim.x / tiles_n row streams of tiles_n column blocks each; a stream walks the
// row tiles t_first, t_first + G, ...  Block id = 8 slot + xcd (ids are dealt round-robin over the 8 XCDs), slot = tiles_n x (row stream
// of the XCD) + column block: the column blocks of a stream share an XCD, i.e. its L2 holds the stream's activation tiles.  SPARE: G need
// not be a multiple of 8 - the last G & 7 streams take the CUs that division leaves over, their blocks spread across XCDs.
template <bool SPARE>
__device__ __forceinline__ void ws_block_map(int tiles_n, int& cb, int& G, int& t_first) {
    G = gridDim.x / tiles_n;
    const int nb_main = SPARE ? (G >> 3) * 8 * tiles_n : (int)gridDim.x;
    const int spare = (int)blockIdx.x - nb_main;
    cb = spare < 0 ? (int)(blockIdx.x >> 3) % tiles_n : spare % tiles_n;
    t_first = spare < 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) / tiles_n : (G & ~7) + spare / tiles_n;
}
// The walk of a row stream by the pipelined kernels: AHEAD + 1 tiles in flight before the first one is consumed, then tiles alternate
// between the two accumulator sets - tile(set, a finished tile is pending in the other set, t, i) - and last(set, i) finishes the set
// that holds the stream's final tile.
template <int AHEAD, class Issue, class Tile, class Last>
__device__ __forceinline__ void ws_walk(int t_first, int G, int ntiles, Issue&& issue_tile, Tile&& tile, Last&& last) {
    int t = t_first, i = 0;
#pragma unroll
    for (int k = 0; k <= AHEAD; ++k)
        if (t + k * G < ntiles) issue_tile(t + k * G, k);
    if (t >= ntiles) return;
    tile(WInt<0>{}, WInt<0>{}, t, i);
    t += G; ++i;
    for (;;) {
        if (t >= ntiles) { last(WInt<0>{}, i); break; }
        tile(WInt<1>{}, WInt<1>{}, t, i);
        t += G; ++i;
        if (t >= ntiles) { last(WInt<1>{}, i); break; }
        tile(WInt<0>{}, WInt<1>{}, t, i);
        t += G; ++i;
    }
}

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
    int cb, G, t_first;
    ws_block_map<false>(p.tiles_n, cb, G, t_first);
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
#if VCX_WS_ABL == 2
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
#if VCX_WS_ABL == 1        // tools/ws_ablate.py: no MFMA work (timing only)
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
#if VCX_WS_ABL == 2        // tools/ws_ablate.py: no epilogue (timing only; one store keeps the MFMAs alive)
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
// VCX_GEMM_ROWSTATS: two slots of [32 rows][4 waves] (mean, M2) partials behind the ring (the SAME __shared__ object: a second one makes
// hipcc drain the LDS-DMA queue in front of every fragment read)
constexpr size_t WP_RS_SLOT = (size_t)WpCfg::TBM * 4 * 2 * sizeof(float);
constexpr size_t WP_SMEM_RS = WP_SMEM + 2 * WP_RS_SLOT;


// RS (VCX_GEMM_ROWSTATS, round 6): the block owns whole output rows, so it also writes LayerNorm's (mean, rstd) of every ROUNDED output
// row - the statistics pass in front of the LayerNorm-folded projection behind this layer (vcx_rowstats_f16: one read of the tensor,
// 20 launches / 1.2 ms per DDIM step at level 0) disappears.  On the matrix pipe, which is two thirds idle here: a unit's eight packed
// fp16 outputs per lane ARE a 16x16x32 B operand (row = lane & 15, 32 of the wave's columns over the four lane groups), so
// ones x X is the row sum (exact products) and the diagonal of D^T D the row's sum of squares - two MFMAs per unit, no VALU reduction,
// no shuffles.  D = X minus a per-(row, wave) shift K: 0, or - where the strip's first three values say the row is offset - their
// median (value - K is then exact in fp16, and |mean| >> std does not cancel); per wave (mean, M2) of its 80 columns -> LDS -> one
// wave merges the four strips of the 32 rows of the tile before last (Chan, fixed order) and stores (mean, rstd).
template <bool RES, bool RS>
__global__ void __launch_bounds__(WpCfg::THREADS, 1) gemm_ws320_pipe_kernel(GemmArgs p, unsigned a_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MF = WpCfg::MF, NF = WpCfg::NF, UNITS = NF / 2 + NF % 2, NUNIT = UNITS * MF;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t srd_s = __builtin_amdgcn_make_buffer_rsrc(p.rowstats, 0, RS ? (int)(8u * (unsigned)p.M) : 0, 0x00020000);

    int ntiles = p.tiles_m;
    int cb, G, t_first;
    ws_block_map<false>(p.tiles_n, cb, G, t_first);
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
    // ---- ROWSTATS state: one pending tile at a time (its units all run inside the next tile's MFMA stream)
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    [[maybe_unused]] f4 rsS[MF], rsQ[MF];          // ones x D (every element = the row sum) / D^T D (diagonal = the row's sum of squares), per 16-row group
    [[maybe_unused]] h2v rsK[MF];                  // the shift of (row = lr of group b, this wave's strip), in both halves
    [[maybe_unused]] int rs_row[2] = {0, 0};       // first row of the tile in each accumulator set
    [[maybe_unused]] float* const rs_lds = reinterpret_cast<float*>(smem_raw + WP_SMEM);
    [[maybe_unused]] const h8 rs_ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

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
        // RS: the packed words are formed ONCE, as pairs (v_cvt_pk_f16_f32), and serve the statistics and the store; the plain kernels keep
        // the expression their listing was tuned with
        typedef float f2w __attribute__((ext_vector_type(2)));
        [[maybe_unused]] h2v x0, x1, x2 = {(half_t)0.f, (half_t)0.f}, x3 = x2;
        u2v p0;
        [[maybe_unused]] u2v p1s = {0u, 0u};
        if constexpr (RS) {
            x0 = __builtin_convertvector(f2w{v0[0], v0[1]}, h2v);
            x1 = __builtin_convertvector(f2w{v0[2], v0[3]}, h2v);
            p0 = u2v{__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1)};
            if constexpr (wide) {
                x2 = __builtin_convertvector(f2w{v1[0], v1[1]}, h2v);
                x3 = __builtin_convertvector(f2w{v1[2], v1[3]}, h2v);
                p1s = u2v{__builtin_bit_cast(unsigned, x2), __builtin_bit_cast(unsigned, x3)};
            }
        } else {
            p0 = __builtin_bit_cast(u2v, h4{(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3]});
            if constexpr (wide) p1s = __builtin_bit_cast(u2v, h4{(half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]});
        }
        if constexpr (RS) {
            if constexpr (u == 0) {
                // The shift of (row lr, this wave's strip), decided in the lane with lg = 0 from its eight values (fragments 0 and 1, columns
                // 0 .. 3 of each): the median of three of them where the row is OFFSET (|median| > 4 x the spread of the eight: every
                // value of such a row is within a factor 1.25 of the shift, so value - shift is exact in fp16, and the squares of the raw
                // values would cancel), else 0 (the deviations are the values themselves: exact, and |mean| <= ~11 std loses < 2e-5 of the
                // variance in fp32).  A shift that is neither - taken by accident in a row that is not offset - would round the
                // deviations to fp16 (1e-4 of that row's variance): eight samples make the accident a 1e-6 event.
                // (asm: __builtin_fmaxf puts a canonicalising v_max_f32 x, x in front of every operand)
                float hi8, lo8;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(hi8) : "v"(v0[0]), "v"(v0[1]), "v"(v0[2]));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(hi8) : "v"(v0[3]), "v"(v1[0]));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(hi8) : "v"(v1[1]), "v"(v1[2]));
                asm("v_max_f32 %0, %0, %1" : "+v"(hi8) : "v"(v1[3]));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo8) : "v"(v0[0]), "v"(v0[1]), "v"(v0[2]));
                asm("v_min3_f32 %0, %0, %1, %2" : "+v"(lo8) : "v"(v0[3]), "v"(v1[0]));
                asm("v_min3_f32 %0, %0, %1, %2" : "+v"(lo8) : "v"(v1[1]), "v"(v1[2]));
                asm("v_min_f32 %0, %0, %1" : "+v"(lo8) : "v"(v1[3]));
                const float med = __builtin_amdgcn_fmed3f(v0[0], v0[1], v0[2]);
                const half_t k3 = (half_t)(__builtin_fabsf(med) > 4.0f * (hi8 - lo8) ? med : 0.0f);      // (rounding is monotonic: the fp16 value of one of the three, or 0)
                const unsigned kk = (unsigned)__builtin_bit_cast(unsigned short, k3) * 0x10001u;
                rsK[b] = __builtin_bit_cast(h2v, (unsigned)__shfl((int)kk, lr));
            }
            const h2v k2 = rsK[b];
            // (never __builtin_bit_cast(h2v, p0[1]): a bit_cast of a vector ELEMENT reads element 0 with this hipcc - the finding of row_halves() in attention.hip)
            const h2v d0 = x0 - k2, d1 = x1 - k2;
            h2v d2 = {(half_t)0.f, (half_t)0.f}, d3 = d2;
            if constexpr (wide) { d2 = x2 - k2; d3 = x3 - k2; }
            const h8 xv = {x0[0], x0[1], x1[0], x1[1], x2[0], x2[1], x3[0], x3[1]};      // the stored values: their sum is exact (fp16 x 1.0 into fp32)
            const h8 dv = {d0[0], d0[1], d1[0], d1[1], d2[0], d2[1], d3[0], d3[1]};      // their deviations from the shift: the squares must not cancel
            if constexpr (u == 0) {
                rsS[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rs_ones, xv, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                rsQ[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dv, dv, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else {
                rsS[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rs_ones, xv, rsS[b], 0, 0, 0);
                rsQ[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dv, dv, rsQ[b], 0, 0, 0);
            }
        }
        if constexpr (wide) {
            const u2v p1 = p1s;
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

    // ROWSTATS, second step: group b of the pending tile (held in set PAR) is complete - (mean, M2) of this wave's 80 columns of row
    // b 16 + lr go to slot PAR; the diagonal element of row lr is element lr & 3 of the lane with lg = lr >> 2
    [[maybe_unused]] auto rs_part = [&](auto PAR_, auto B_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, b = decltype(B_)::value;
        const float S = rsS[b][0];
        const int r = lr & 3;
        float Q = rsQ[b][0];
        Q = r == 1 ? rsQ[b][1] : Q;
        Q = r == 2 ? rsQ[b][2] : Q;
        Q = r == 3 ? rsQ[b][3] : Q;
        constexpr float n1 = (float)(NF * 16), inv = 1.0f / n1;
        const float mean = S * inv;                                        // of the stored values, unshifted
        const float Sd = __builtin_fmaf(-n1, (float)rsK[b][0], S);         // sum of the deviations from the shift
        const float m2 = __builtin_fmaf(-Sd * inv, Sd, Q);
        if (lg == (lr >> 2))
            *reinterpret_cast<float2*>(rs_lds + PAR * (WP_RS_SLOT / sizeof(float)) + ((b * 16 + lr) * 4 + wave) * 2) = make_float2(mean, m2 > 0.f ? m2 : 0.f);
    };
    // ROWSTATS, last step: one wave merges the four strips of the 32 rows in `slot` (equal counts: ((0, 1), (2, 3)), a fixed order) and
    // stores (mean, rstd) of rows row0 .. row0 + 31 (rows >= M are dropped by the descriptor's range check)
    [[maybe_unused]] auto rs_fin = [&](int slot, int row0) __attribute__((always_inline)) {
        if (lane < WpCfg::TBM) {
            const f4* src = reinterpret_cast<const f4*>(rs_lds + slot * (WP_RS_SLOT / sizeof(float)) + lane * 8);
            const f4 x01 = src[0], x23 = src[1];           // (mean0, M2_0, mean1, M2_1), (mean2, M2_2, mean3, M2_3)
            constexpr float n1 = (float)(NF * 16);
            // (scalars behind empty asm statements: left to itself the SLP vectoriser pairs these into v_pk_add_f32 ... op_sel, the
            // encoding tools/isa_audit.py bans - profiles/r04_pkfma_rootcause.md)
            float m0 = x01[0], q0 = x01[1], m1 = x01[2], q1 = x01[3], m2 = x23[0], q2 = x23[1], m3 = x23[2], q3 = x23[3];
            asm volatile("" : "+v"(m0), "+v"(q0), "+v"(m1), "+v"(q1), "+v"(m2), "+v"(q2), "+v"(m3), "+v"(q3));
            // every statement takes an operand from the asm statement behind the previous one: no two of them can be paired
            float d01 = m1 - m0;
            asm volatile("" : "+v"(d01), "+v"(m3));
            float d23 = m3 - m2;
            asm volatile("" : "+v"(d23), "+v"(m0));
            float ma = __builtin_fmaf(0.5f, d01, m0);
            asm volatile("" : "+v"(ma), "+v"(m2));
            float mb = __builtin_fmaf(0.5f, d23, m2);
            asm volatile("" : "+v"(mb), "+v"(q0));
            float qa = q0 + q1;
            asm volatile("" : "+v"(qa), "+v"(q2));
            float qb = q2 + q3;
            asm volatile("" : "+v"(qb), "+v"(d01));
            float e01 = d01 * d01;
            asm volatile("" : "+v"(e01), "+v"(d23));
            float e23 = d23 * d23;
            asm volatile("" : "+v"(e23), "+v"(qa));
            qa = __builtin_fmaf(e01, 0.5f * n1, qa);
            asm volatile("" : "+v"(qa), "+v"(qb));
            qb = __builtin_fmaf(e23, 0.5f * n1, qb);
            asm volatile("" : "+v"(qb), "+v"(mb));
            float dd = mb - ma;
            asm volatile("" : "+v"(dd), "+v"(qa));
            const float mean = __builtin_fmaf(0.5f, dd, ma);
            float qs = qa + qb;
            asm volatile("" : "+v"(qs), "+v"(dd));
            const float q = __builtin_fmaf(dd * dd, n1, qs);
            const float rstd = rsqrtf(q * (1.0f / (4.0f * n1)) + p.rowstats_eps);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, make_float2(mean, rstd)), srd_s, (unsigned)(row0 + lane) * 8u, 0, 0);
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
        if constexpr (RS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and this wave's row-moment partials are in LDS
        __builtin_amdgcn_s_barrier();                // ... for every wave; and every wave is done with the tiles before
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const int fin_row = rs_row[PAR];       // the tile before last used this set: its partials (slot PAR) are complete behind this barrier
        if constexpr (RS) rs_row[PAR] = p.m_begin + t * WpCfg::TBM;
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
            // ROWSTATS: the moments of a 16-row group two K slices behind its last unit (no wait states); the tile before last is merged
            // and stored by one wave (which one rotates with the tile) behind the matrix work
            if constexpr (RS && PEND && kk == 6) rs_part(WInt<PAR ^ 1>{}, WInt<0>{});
            if constexpr (RS && PEND && kk == 9) rs_part(WInt<PAR ^ 1>{}, WInt<1>{});
        });
        if constexpr (RS) {
            if (i >= 2 && wave == (i & 3)) rs_fin(PAR, fin_row);
        }
    };

    // the last tile (set PAR, the i-th of this block): its units, its moments, and - behind one more barrier - the merges of the last two tiles
    auto finish = [&](auto PAR_, int i) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        if constexpr (RS) {       // (the wave that merges the tile before last may still be reading slot PAR: nobody overwrites it before this barrier)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        static_for_ws<NUNIT>([&](auto J_) __attribute__((always_inline)) { unit(PAR_, J_); });
        if constexpr (RS) {
            rs_part(PAR_, WInt<0>{});
            rs_part(PAR_, WInt<1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (i >= 2 && wave == 0) rs_fin(PAR ^ 1, rs_row[PAR ^ 1]);
            if (wave == 1) rs_fin(PAR, rs_row[PAR]);
        }
    };

    ws_walk<WP_AHEAD - 1>(t_first, G, ntiles, issue_tile, tile, finish);
#endif
}

// =============================================================================================================================
// Weight-stationary GEGLU projection for K = 320 (level 0: 460800 x 2560 x 320, ten launches per step, the largest excess over floor of
// the whole per-shape table: 0.93 ms against 0.56).  In the tiled engine the layer spends as long in its epilogue (bias, exact GELU of the
// gate, product, fp16 rounding: ~12 VALU instructions per output) as in its five K-steps, and the deferred-epilogue experiment of round 5
// showed why that cannot be hidden THERE: the K-step's own LDS / DMA traffic has spent the MFMA shadows, and a 16x16x32 MFMA has one free
// issue slot.  Here it can: a block keeps a 256-column slice of the packed weight (value | gate blocks of 32) in registers - 40 KB per
// wave, 160 registers - so a K-slice is two MFMAs and ONE LDS read, and the MFMAs are 32x32x16 (four to five free issue slots each,
// tools/ubench_fill.hip): the finished tile's 16 outputs per lane are formed two at a time in the shadows of the next tile's MFMAs
// (sched_group_barrier: one MFMA, six VALU), packed, exchanged between the lane halves and stored as two 16-byte pieces at its end.
// A wave's 64 packed columns are exactly one [32 value | 32 gate] block = the two 32-row A blocks of the MFMA, so value and gate of an
// output land in the same lane and register.  The ten column blocks that share a row stream sit on one XCD (block id = 8 slot + xcd):
// the activation tile comes from HBM once and from that L2 nine times.  Results agree with the tiled engine's GEGLU to fp16 rounding
// (32x32x16 sums K in another order than 16x16x32), are bit-reproducible and do not depend on M.
// =============================================================================================================================
struct WgCfg {
    static constexpr int TBM = 64, TBN = 256;
    static constexpr int THREADS = 256;
};
constexpr int WG_STAGE = WgCfg::TBM * WS_K * (int)sizeof(half_t);          // 40 KB: five [64 rows][64] slabs
[[maybe_unused]] constexpr int WG_RING = 3, WG_AHEAD = 1;                  // (a stage is re-filled one barrier after its tile was consumed)
[[maybe_unused]] constexpr int WG_PIECES = WG_STAGE / 1024 / 4;            // LDS-DMA instructions per wave and tile (10)
[[maybe_unused]] constexpr int WG_KS = WS_K / 16;                          // twenty 16-deep K slices
constexpr size_t WG_SMEM = (size_t)WG_RING * WG_STAGE;

// ---- gemm_ws320_geglu_kernel: the 160 weight registers are OWNED accumulator registers a[0:159] (fragment I = 20 ab + kk in a[4 I .. 4 I + 3]),
// read by the MFMAs as their A operand directly; the accumulators are ordinary VGPR tuples, so neither weights nor results pass through
// v_accvgpr_read (the compiler's own allocation shuttled 45 weight fragments and all 64 pending results per tile through it).
#define WG_CLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
template <int I>
__device__ __forceinline__ void wg_load_w(const h8& w) {
    typedef unsigned u4w __attribute__((ext_vector_type(4)));
    const u4w v = __builtin_bit_cast(u4w, w);
    if constexpr (I == 0) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a0", "a1", "a2", "a3");
    if constexpr (I == 1) asm volatile("v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %1\n\tv_accvgpr_write_b32 a6, %2\n\tv_accvgpr_write_b32 a7, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a4", "a5", "a6", "a7");
    if constexpr (I == 2) asm volatile("v_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %1\n\tv_accvgpr_write_b32 a10, %2\n\tv_accvgpr_write_b32 a11, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a8", "a9", "a10", "a11");
    if constexpr (I == 3) asm volatile("v_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %1\n\tv_accvgpr_write_b32 a14, %2\n\tv_accvgpr_write_b32 a15, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a12", "a13", "a14", "a15");
    if constexpr (I == 4) asm volatile("v_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %1\n\tv_accvgpr_write_b32 a18, %2\n\tv_accvgpr_write_b32 a19, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a16", "a17", "a18", "a19");
    if constexpr (I == 5) asm volatile("v_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %1\n\tv_accvgpr_write_b32 a22, %2\n\tv_accvgpr_write_b32 a23, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a20", "a21", "a22", "a23");
    if constexpr (I == 6) asm volatile("v_accvgpr_write_b32 a24, %0\n\tv_accvgpr_write_b32 a25, %1\n\tv_accvgpr_write_b32 a26, %2\n\tv_accvgpr_write_b32 a27, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a24", "a25", "a26", "a27");
    if constexpr (I == 7) asm volatile("v_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %1\n\tv_accvgpr_write_b32 a30, %2\n\tv_accvgpr_write_b32 a31, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a28", "a29", "a30", "a31");
    if constexpr (I == 8) asm volatile("v_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %1\n\tv_accvgpr_write_b32 a34, %2\n\tv_accvgpr_write_b32 a35, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a32", "a33", "a34", "a35");
    if constexpr (I == 9) asm volatile("v_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %1\n\tv_accvgpr_write_b32 a38, %2\n\tv_accvgpr_write_b32 a39, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a36", "a37", "a38", "a39");
    if constexpr (I == 10) asm volatile("v_accvgpr_write_b32 a40, %0\n\tv_accvgpr_write_b32 a41, %1\n\tv_accvgpr_write_b32 a42, %2\n\tv_accvgpr_write_b32 a43, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a40", "a41", "a42", "a43");
    if constexpr (I == 11) asm volatile("v_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %1\n\tv_accvgpr_write_b32 a46, %2\n\tv_accvgpr_write_b32 a47, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a44", "a45", "a46", "a47");
    if constexpr (I == 12) asm volatile("v_accvgpr_write_b32 a48, %0\n\tv_accvgpr_write_b32 a49, %1\n\tv_accvgpr_write_b32 a50, %2\n\tv_accvgpr_write_b32 a51, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a48", "a49", "a50", "a51");
    if constexpr (I == 13) asm volatile("v_accvgpr_write_b32 a52, %0\n\tv_accvgpr_write_b32 a53, %1\n\tv_accvgpr_write_b32 a54, %2\n\tv_accvgpr_write_b32 a55, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a52", "a53", "a54", "a55");
    if constexpr (I == 14) asm volatile("v_accvgpr_write_b32 a56, %0\n\tv_accvgpr_write_b32 a57, %1\n\tv_accvgpr_write_b32 a58, %2\n\tv_accvgpr_write_b32 a59, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a56", "a57", "a58", "a59");
    if constexpr (I == 15) asm volatile("v_accvgpr_write_b32 a60, %0\n\tv_accvgpr_write_b32 a61, %1\n\tv_accvgpr_write_b32 a62, %2\n\tv_accvgpr_write_b32 a63, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a60", "a61", "a62", "a63");
    if constexpr (I == 16) asm volatile("v_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %1\n\tv_accvgpr_write_b32 a66, %2\n\tv_accvgpr_write_b32 a67, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a64", "a65", "a66", "a67");
    if constexpr (I == 17) asm volatile("v_accvgpr_write_b32 a68, %0\n\tv_accvgpr_write_b32 a69, %1\n\tv_accvgpr_write_b32 a70, %2\n\tv_accvgpr_write_b32 a71, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a68", "a69", "a70", "a71");
    if constexpr (I == 18) asm volatile("v_accvgpr_write_b32 a72, %0\n\tv_accvgpr_write_b32 a73, %1\n\tv_accvgpr_write_b32 a74, %2\n\tv_accvgpr_write_b32 a75, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a72", "a73", "a74", "a75");
    if constexpr (I == 19) asm volatile("v_accvgpr_write_b32 a76, %0\n\tv_accvgpr_write_b32 a77, %1\n\tv_accvgpr_write_b32 a78, %2\n\tv_accvgpr_write_b32 a79, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a76", "a77", "a78", "a79");
    if constexpr (I == 20) asm volatile("v_accvgpr_write_b32 a80, %0\n\tv_accvgpr_write_b32 a81, %1\n\tv_accvgpr_write_b32 a82, %2\n\tv_accvgpr_write_b32 a83, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a80", "a81", "a82", "a83");
    if constexpr (I == 21) asm volatile("v_accvgpr_write_b32 a84, %0\n\tv_accvgpr_write_b32 a85, %1\n\tv_accvgpr_write_b32 a86, %2\n\tv_accvgpr_write_b32 a87, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a84", "a85", "a86", "a87");
    if constexpr (I == 22) asm volatile("v_accvgpr_write_b32 a88, %0\n\tv_accvgpr_write_b32 a89, %1\n\tv_accvgpr_write_b32 a90, %2\n\tv_accvgpr_write_b32 a91, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a88", "a89", "a90", "a91");
    if constexpr (I == 23) asm volatile("v_accvgpr_write_b32 a92, %0\n\tv_accvgpr_write_b32 a93, %1\n\tv_accvgpr_write_b32 a94, %2\n\tv_accvgpr_write_b32 a95, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a92", "a93", "a94", "a95");
    if constexpr (I == 24) asm volatile("v_accvgpr_write_b32 a96, %0\n\tv_accvgpr_write_b32 a97, %1\n\tv_accvgpr_write_b32 a98, %2\n\tv_accvgpr_write_b32 a99, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a96", "a97", "a98", "a99");
    if constexpr (I == 25) asm volatile("v_accvgpr_write_b32 a100, %0\n\tv_accvgpr_write_b32 a101, %1\n\tv_accvgpr_write_b32 a102, %2\n\tv_accvgpr_write_b32 a103, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a100", "a101", "a102", "a103");
    if constexpr (I == 26) asm volatile("v_accvgpr_write_b32 a104, %0\n\tv_accvgpr_write_b32 a105, %1\n\tv_accvgpr_write_b32 a106, %2\n\tv_accvgpr_write_b32 a107, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a104", "a105", "a106", "a107");
    if constexpr (I == 27) asm volatile("v_accvgpr_write_b32 a108, %0\n\tv_accvgpr_write_b32 a109, %1\n\tv_accvgpr_write_b32 a110, %2\n\tv_accvgpr_write_b32 a111, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a108", "a109", "a110", "a111");
    if constexpr (I == 28) asm volatile("v_accvgpr_write_b32 a112, %0\n\tv_accvgpr_write_b32 a113, %1\n\tv_accvgpr_write_b32 a114, %2\n\tv_accvgpr_write_b32 a115, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a112", "a113", "a114", "a115");
    if constexpr (I == 29) asm volatile("v_accvgpr_write_b32 a116, %0\n\tv_accvgpr_write_b32 a117, %1\n\tv_accvgpr_write_b32 a118, %2\n\tv_accvgpr_write_b32 a119, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a116", "a117", "a118", "a119");
    if constexpr (I == 30) asm volatile("v_accvgpr_write_b32 a120, %0\n\tv_accvgpr_write_b32 a121, %1\n\tv_accvgpr_write_b32 a122, %2\n\tv_accvgpr_write_b32 a123, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a120", "a121", "a122", "a123");
    if constexpr (I == 31) asm volatile("v_accvgpr_write_b32 a124, %0\n\tv_accvgpr_write_b32 a125, %1\n\tv_accvgpr_write_b32 a126, %2\n\tv_accvgpr_write_b32 a127, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a124", "a125", "a126", "a127");
    if constexpr (I == 32) asm volatile("v_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %1\n\tv_accvgpr_write_b32 a130, %2\n\tv_accvgpr_write_b32 a131, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a128", "a129", "a130", "a131");
    if constexpr (I == 33) asm volatile("v_accvgpr_write_b32 a132, %0\n\tv_accvgpr_write_b32 a133, %1\n\tv_accvgpr_write_b32 a134, %2\n\tv_accvgpr_write_b32 a135, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a132", "a133", "a134", "a135");
    if constexpr (I == 34) asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %1\n\tv_accvgpr_write_b32 a138, %2\n\tv_accvgpr_write_b32 a139, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a136", "a137", "a138", "a139");
    if constexpr (I == 35) asm volatile("v_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %1\n\tv_accvgpr_write_b32 a142, %2\n\tv_accvgpr_write_b32 a143, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a140", "a141", "a142", "a143");
    if constexpr (I == 36) asm volatile("v_accvgpr_write_b32 a144, %0\n\tv_accvgpr_write_b32 a145, %1\n\tv_accvgpr_write_b32 a146, %2\n\tv_accvgpr_write_b32 a147, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a144", "a145", "a146", "a147");
    if constexpr (I == 37) asm volatile("v_accvgpr_write_b32 a148, %0\n\tv_accvgpr_write_b32 a149, %1\n\tv_accvgpr_write_b32 a150, %2\n\tv_accvgpr_write_b32 a151, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a148", "a149", "a150", "a151");
    if constexpr (I == 38) asm volatile("v_accvgpr_write_b32 a152, %0\n\tv_accvgpr_write_b32 a153, %1\n\tv_accvgpr_write_b32 a154, %2\n\tv_accvgpr_write_b32 a155, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a152", "a153", "a154", "a155");
    if constexpr (I == 39) asm volatile("v_accvgpr_write_b32 a156, %0\n\tv_accvgpr_write_b32 a157, %1\n\tv_accvgpr_write_b32 a158, %2\n\tv_accvgpr_write_b32 a159, %3\n\ts_nop 1" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a156", "a157", "a158", "a159");
}
// D = W(a[4 I ..]) x X + C (first MFMA of a tile: C = the bias tuple) / D += W x X
template <int I>
__device__ __forceinline__ void wg_mfma_first(f16v& d, const h8& x, const f16v& c) {
    if constexpr (I == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %1, %2" : "=&v"(d) : "v"(x), "v"(c));
    if constexpr (I == 20) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[80:83], %1, %2" : "=&v"(d) : "v"(x), "v"(c));
}
template <int I>
__device__ __forceinline__ void wg_mfma_acc(f16v& d, const h8& x) {
    if constexpr (I == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[4:7], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[8:11], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[12:15], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[16:19], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 5) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[20:23], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 6) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[24:27], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 7) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[28:31], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 8) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[32:35], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 9) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[36:39], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 10) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[40:43], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 11) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[44:47], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 12) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[48:51], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 13) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[52:55], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 14) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[56:59], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 15) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[60:63], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 16) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[64:67], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 17) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[68:71], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 18) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[72:75], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 19) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[76:79], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 20) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[80:83], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 21) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[84:87], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 22) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[88:91], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 23) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[92:95], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 24) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[96:99], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 25) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[100:103], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 26) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[104:107], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 27) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[108:111], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 28) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[112:115], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 29) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[116:119], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 30) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[120:123], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 31) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[124:127], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[128:131], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 33) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[132:135], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 34) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[136:139], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 35) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[140:143], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 36) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[144:147], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 37) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[148:151], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 38) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[152:155], %1, %0" : "+v"(d) : "v"(x));
    if constexpr (I == 39) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[156:159], %1, %0" : "+v"(d) : "v"(x));
}

__global__ void __launch_bounds__(WgCfg::THREADS, 1) gemm_ws320_geglu_kernel(GemmArgs p, unsigned a_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);

    const int ntiles = p.tiles_m;
    // G row streams of tiles_n column blocks each: 8 (G >> 3) of them with all their blocks on one XCD (block id = 8 slot + xcd), and G & 7
    // more whose blocks take the CUs that division leaves over, across XCDs (ten column blocks: 3 streams per XCD = 240 blocks + 1
    // stream on the 16 spare CUs; its activation tiles come from the memory side ten times instead of once - 4 % of the rows)
    int cb, G, t_first;
    ws_block_map<true>(p.tiles_n, cb, G, t_first);
    const int ncol0 = cb * WgCfg::TBN;                     // first PACKED column (weight row) of the block; its outputs start at ncol0 / 2
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;

    // A fragments: lane (i = lq, hi) holds W[ncol0 + 64 wave + 32 ab + i][16 kk + 8 hi .. + 7]; ab = 0 the block's 32 value columns, 1 their gates
    // (into the owned accumulator registers: wg_load_w)
    // The bias rides in the accumulators: the first MFMA of a tile takes C = bias (accumulator register r = 4 q + c holds packed column
    // 8 q + 4 hi + c of its A block in every lane).  No multiply-add per output in the epilogue, and the 32 registers are MFMA
    // operands - they may live in the accumulator half of the file.  (alpha = 1 only: the dispatcher sends other calls to the tiled engine.)
    f16v binit[2];
    {
        const half_t* wrow = p.W + (size_t)(ncol0 + wave * 64 + lq) * p.ldw + hi * 8;
        // twenty loads in flight, then their twenty register writes (each wg_load_w is an asm volatile: written load by load, hipcc
        // waits with vmcnt(0) behind every single one - forty memory round trips in a row at the head of the kernel)
        static_for_ws<2>([&](auto AB_) __attribute__((always_inline)) {
            constexpr int ab = decltype(AB_)::value;
            h8 wt[WG_KS];
#pragma unroll
            for (int kk = 0; kk < WG_KS; ++kk) wt[kk] = *reinterpret_cast<const h8*>(wrow + (size_t)ab * 32 * p.ldw + kk * 16);
            static_for_ws<WG_KS>([&](auto KK_) __attribute__((always_inline)) { wg_load_w<WG_KS * ab + decltype(KK_)::value>(wt[decltype(KK_)::value]); });
        });
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 bq = (p.flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + ncol0 + wave * 64 + 32 * ab + 8 * q + 4 * hi) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) binit[ab][4 * q + c] = bq[c];
            }
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }

    // LDS-DMA of one 64-row tile: 40 instructions of 8 rows x 128 bytes; wave w issues piece i = 0..9 = slab i / 2, rows 8 (w + 4 (i & 1)) .. + 7
    const int drow = wave * 8 + (lane >> 3);
    const unsigned dsrc = (unsigned)((lane & 7) ^ ((drow >> 1) & 7)) * 16u;
    const unsigned rstep32 = 32u * (unsigned)p.lda * 2u;
    int issued = 0;
    int mark[WG_RING];
    auto issue_tile = [&](int t, int buf) {
        const int m0 = p.m_begin + t * WgCfg::TBM + drow;
        const unsigned base = (unsigned)m0 * (unsigned)p.lda * 2u + dsrc;
        const unsigned va = m0 < p.M ? base : OOB, vb = m0 + 32 < p.M ? base + rstep32 : OOB;
        unsigned char* dst = smem_raw + buf * WG_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < WG_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dst + (i >> 1) * 8192 + (i & 1) * 4096), 16, (i & 1) ? vb : va,
                                                     (unsigned)(i >> 1) * (BK * 2), 0, 0);
        issued += WG_PIECES;
        mark[buf] = issued;
    };
    // B fragment of K slice kk and row block mb: lane (m = 32 mb + lq, hi) holds X[m][16 kk + 8 hi .. + 7] = chunk 2 (kk & 3) + hi of slab kk >> 2
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xo[i] = lds_off(lq, 2 * i + hi);

    f16v acc[2][2][2];                        // [accumulator set][value | gate][row block]
    unsigned coff[2][2];                      // byte offset of (lane's row in row block mb, wave's first OUTPUT column + 8 hi) of the tile in each accumulator set

    // The finished tile's epilogue in 68 chunks, one behind each of the next tile's MFMAs 6 .. 73 (sched_barrier(0) on both sides of every
    // MFMA and chunk: the order below IS the instruction stream - left to the scheduler, sched_group_barrier or not, hipcc issues a
    // region's MFMAs back to back, the wave stalls on the busy pipe and the VALU work runs behind it, not beside it).  Per row block:
    //   chunks 0 .. 31: output pair c / 4 - accumulators; clamp, max, two Horner steps; four Horner steps; exp2, products, fp16 pair -
    //   two independent chains side by side (one wave per SIMD: nothing else hides the latency of a dependent chain)
    //   chunks 32, 33: half exchange + one 16-byte store each.     The arithmetic of gemm_epilogue's GEGLU path (bias in the accumulator).
    float hx[2], hg[2], ha[2], hm[2], hq[2];
    u2v pk[4];
    auto chunk = [&](auto PAR_, auto C_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, mb = decltype(C_)::value / 34, c = decltype(C_)::value % 34;
        if constexpr (c < 32) {
            constexpr int P = c >> 2, ph = c & 3;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                constexpr int dummy = 0; (void)dummy;
                const int r = 2 * P + e;
                if constexpr (ph == 0) {
                    hx[e] = acc[PAR][0][mb][r];
                    hg[e] = acc[PAR][1][mb][r];
                } else if constexpr (ph == 1) {
                    // min(|g|, 9) and max(g, 0) as the two instructions they are: on a value that comes out of an accumulator read hipcc
                    // puts a canonicalising v_max_f32 g, g, g in front of each (NaN rule of fminf / fmaxf; an MFMA result needs none)
                    asm("v_min_f32 %0, |%1|, %2" : "=v"(ha[e]) : "v"(hg[e]), "s"(9.0f));
                    asm("v_max_f32 %0, 0, %1" : "=v"(hm[e]) : "v"(hg[e]));
                    hq[e] = gelu_q_step<2>(gelu_q_step<1>(GELU_Q[0], ha[e]), ha[e]);
                } else if constexpr (ph == 2) {
                    hq[e] = gelu_q_step<6>(gelu_q_step<5>(gelu_q_step<4>(gelu_q_step<3>(hq[e], ha[e]), ha[e]), ha[e]), ha[e]);
                } else {
                    hx[e] = hx[e] * gelu_finish(hg[e], ha[e], hm[e], __builtin_amdgcn_exp2f(hq[e]));
                    asm volatile("" : "+v"(hx[e]));          // fp32 product first, then ONE fp16 rounding (no v_fma_mixlo_f16)
                }
            }
            if constexpr (ph == 3) {          // outputs 2 P, 2 P + 1 = word P & 1 of column group P / 2
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                pk[P >> 1][P & 1] = __builtin_bit_cast(unsigned, h2v{(half_t)hx[0], (half_t)hx[1]});
            }
        } else {
            // column group q holds columns 8 q + 4 hi .. + 3: after the half swap lanes 0-31 own columns 16 k .. + 7, lanes 32-63 columns 16 k + 8 .. + 15
            constexpr int k = c - 32;
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * k][0], pk[2 * k + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * k][1], pk[2 * k + 1][1], false, false);
            const epi_u4v w = {s0[0], s1[0], s0[1], s1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(w, srd_c, coff[PAR][mb], k * 32, 0);
            asm volatile("s_nop 1" : : "v"(w));          // (the wide-store rule of tools/isa_audit.py)
            issued += 1;
        }
    };

    // one tile: its 80 MFMAs into accumulator set PAR; PEND: the other set holds a finished tile whose chunks ride behind MFMAs 6 .. 73
    auto tile = [&](auto PAR_, auto PEND_, int t, int i) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool PEND = decltype(PEND_)::value != 0;
        const int buf = i % WG_RING;
        __builtin_amdgcn_sched_barrier(0);
#if !(VCX_WG_ABL & 4)        // timing-only ablations (tools/ws_geglu_scan.py): 1 no epilogue chunks, 2 no MFMAs, 4 no per-tile barrier / wait
        ws_wait_vmcnt(issued - mark[buf]);
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // (every wave has passed this barrier = has finished the tile before: its stage takes the tile after next)
        if (t + (WG_AHEAD + 1) * G < ntiles) issue_tile(t + (WG_AHEAD + 1) * G, (i + WG_AHEAD + 1) % WG_RING);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int m = p.m_begin + t * WgCfg::TBM + 32 * mb + lq;
            coff[PAR][mb] = m < p.M ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(ncol0 / 2 + wave * 32 + 8 * hi)) * 2u : OOB;
        }
        __builtin_amdgcn_sched_barrier(0);
        // four LDS byte addresses per tile (stage base + the lane's chunk of K slices 4 s + j), opaque to the compiler: every fragment
        // read is one of them plus an immediate (slab, row block) - left alone, hipcc keeps ~20 address registers per tile alive, and
        // with 256 VGPRs taken it parks values in a0 .. a3, i.e. in the owned weight registers
        unsigned xb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xb[j] = (unsigned)(buf * WG_STAGE) + 2u * (unsigned)xo[j];
            asm volatile("" : "+v"(xb[j]));
        }
        auto frag = [&](int kk, int mb) __attribute__((always_inline)) {
            return *reinterpret_cast<const h8*>(smem_raw + xb[kk & 3] + (kk >> 2) * (WgCfg::TBM * BK * 2) + mb * 4096);
        };
        // B fragments through a ring of registers, requested XR K slices (= 4 XR MFMAs) ahead: one slice of cover is less than the LDS latency
        constexpr int XR = 4;
        h8 xr[XR][2];
#pragma unroll
        for (int k = 0; k < XR; ++k)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) xr[k][mb] = frag(k, mb);
        static_for_ws<WG_KS>([&](auto KK_) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK_)::value;
            static_for_ws<4>([&](auto J_) __attribute__((always_inline)) {
                constexpr int j = decltype(J_)::value, ab = j & 1, mb = j >> 1, n = 4 * kk + j;
                __builtin_amdgcn_sched_barrier(0);
#if VCX_WG_ABL & 2
                if constexpr (kk == 0) acc[PAR][ab][mb] = binit[ab]; else acc[PAR][ab][mb][n & 15] += (float)xr[kk % XR][mb][0] * (float)xr[kk % XR][mb][1];
#else
                if constexpr (kk == 0) wg_mfma_first<WG_KS * ab>(acc[PAR][ab][mb], xr[kk % XR][mb], binit[ab]);
                else wg_mfma_acc<WG_KS * ab + kk>(acc[PAR][ab][mb], xr[kk % XR][mb]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ab == 1 && kk + XR < WG_KS) xr[kk % XR][mb] = frag(kk + XR, mb);
#if VCX_WG_ABL & 1
                if constexpr (PEND && n >= 6 && n < 74 && (n - 6) % 34 >= 32) chunk(WInt<PAR ^ 1>{}, WInt<n - 6>{});
#else
                if constexpr (PEND && n >= 6 && n < 74) chunk(WInt<PAR ^ 1>{}, WInt<n - 6>{});
#endif
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    auto drain = [&](auto PAR_) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs (asm: hipcc does not know their latency) have written their results
        __builtin_amdgcn_sched_barrier(0);
        static_for_ws<68>([&](auto C_) __attribute__((always_inline)) { chunk(PAR_, C_); });
    };

    ws_walk<WG_AHEAD>(t_first, G, ntiles, issue_tile, tile, [&](auto PAR_, int) __attribute__((always_inline)) { drain(PAR_); });
#endif
}

// =============================================================================================================================
// Weight-stationary LayerNorm-folded projection for K = 320 (level 0: the q | k | v projection of the spatial self-attention,
// 460800 x 960 x 320, ten launches per step, and the 640-column ones: VCX_GEMM_LNFOLD, out = alpha rstd_m (x_m W'^T - mean_m colsum_n) + bias'_n).
// The skeleton of gemm_ws320_geglu_kernel with a lighter epilogue: a block keeps a 256-column slice of the folded weight in the owned
// accumulator registers a[0:159] (wave w: columns 64 w .. 64 w + 63 = the two 32-row A blocks of a 32x32x16 MFMA), a K slice is one
// ds_read_b128 and two MFMAs, and the finished 64-row tile's outputs are formed behind the next tile's MFMAs 6 .. 45: 32 chunks of one
// accumulator register in both row blocks each (two v_accvgpr_read for colsum_n / bias'_n, which live in the owned registers a[160:223];
// four multiply-adds - the arithmetic of gemm_epilogue's LNF = 1 path: fma(acc, rb, fma(qb, colsum, bias)) - and a pair of fp16
// conversions on every second one) and 8 chunks of a half exchange + one 16-byte store.  The row terms (mean, rstd) of a tile are
// requested when its MFMAs start and used one tile later.  N % 64 == 0: the waves of the last column block that have no columns run
// on zero weights and store nothing.  Agrees with the tiled engine to fp16 rounding (32x32x16 sums K in another order), bit-reproducible,
// a row's bits independent of M and of the block map.
// =============================================================================================================================
constexpr int WL_STAGE = WG_STAGE + 1024;                                  // ... + the (mean, rstd) pairs of the tile's rows
constexpr size_t WL_SMEM = (size_t)WG_RING * WL_STAGE;
template <int R>
__device__ __forceinline__ void wl_const_write(float x) {        // a[160 + R] = x
    if constexpr (R == 0) asm volatile("v_accvgpr_write_b32 a160, %0" : : "v"(x) : "a160");
    if constexpr (R == 1) asm volatile("v_accvgpr_write_b32 a161, %0" : : "v"(x) : "a161");
    if constexpr (R == 2) asm volatile("v_accvgpr_write_b32 a162, %0" : : "v"(x) : "a162");
    if constexpr (R == 3) asm volatile("v_accvgpr_write_b32 a163, %0" : : "v"(x) : "a163");
    if constexpr (R == 4) asm volatile("v_accvgpr_write_b32 a164, %0" : : "v"(x) : "a164");
    if constexpr (R == 5) asm volatile("v_accvgpr_write_b32 a165, %0" : : "v"(x) : "a165");
    if constexpr (R == 6) asm volatile("v_accvgpr_write_b32 a166, %0" : : "v"(x) : "a166");
    if constexpr (R == 7) asm volatile("v_accvgpr_write_b32 a167, %0" : : "v"(x) : "a167");
    if constexpr (R == 8) asm volatile("v_accvgpr_write_b32 a168, %0" : : "v"(x) : "a168");
    if constexpr (R == 9) asm volatile("v_accvgpr_write_b32 a169, %0" : : "v"(x) : "a169");
    if constexpr (R == 10) asm volatile("v_accvgpr_write_b32 a170, %0" : : "v"(x) : "a170");
    if constexpr (R == 11) asm volatile("v_accvgpr_write_b32 a171, %0" : : "v"(x) : "a171");
    if constexpr (R == 12) asm volatile("v_accvgpr_write_b32 a172, %0" : : "v"(x) : "a172");
    if constexpr (R == 13) asm volatile("v_accvgpr_write_b32 a173, %0" : : "v"(x) : "a173");
    if constexpr (R == 14) asm volatile("v_accvgpr_write_b32 a174, %0" : : "v"(x) : "a174");
    if constexpr (R == 15) asm volatile("v_accvgpr_write_b32 a175, %0" : : "v"(x) : "a175");
    if constexpr (R == 16) asm volatile("v_accvgpr_write_b32 a176, %0" : : "v"(x) : "a176");
    if constexpr (R == 17) asm volatile("v_accvgpr_write_b32 a177, %0" : : "v"(x) : "a177");
    if constexpr (R == 18) asm volatile("v_accvgpr_write_b32 a178, %0" : : "v"(x) : "a178");
    if constexpr (R == 19) asm volatile("v_accvgpr_write_b32 a179, %0" : : "v"(x) : "a179");
    if constexpr (R == 20) asm volatile("v_accvgpr_write_b32 a180, %0" : : "v"(x) : "a180");
    if constexpr (R == 21) asm volatile("v_accvgpr_write_b32 a181, %0" : : "v"(x) : "a181");
    if constexpr (R == 22) asm volatile("v_accvgpr_write_b32 a182, %0" : : "v"(x) : "a182");
    if constexpr (R == 23) asm volatile("v_accvgpr_write_b32 a183, %0" : : "v"(x) : "a183");
    if constexpr (R == 24) asm volatile("v_accvgpr_write_b32 a184, %0" : : "v"(x) : "a184");
    if constexpr (R == 25) asm volatile("v_accvgpr_write_b32 a185, %0" : : "v"(x) : "a185");
    if constexpr (R == 26) asm volatile("v_accvgpr_write_b32 a186, %0" : : "v"(x) : "a186");
    if constexpr (R == 27) asm volatile("v_accvgpr_write_b32 a187, %0" : : "v"(x) : "a187");
    if constexpr (R == 28) asm volatile("v_accvgpr_write_b32 a188, %0" : : "v"(x) : "a188");
    if constexpr (R == 29) asm volatile("v_accvgpr_write_b32 a189, %0" : : "v"(x) : "a189");
    if constexpr (R == 30) asm volatile("v_accvgpr_write_b32 a190, %0" : : "v"(x) : "a190");
    if constexpr (R == 31) asm volatile("v_accvgpr_write_b32 a191, %0" : : "v"(x) : "a191");
    if constexpr (R == 32) asm volatile("v_accvgpr_write_b32 a192, %0" : : "v"(x) : "a192");
    if constexpr (R == 33) asm volatile("v_accvgpr_write_b32 a193, %0" : : "v"(x) : "a193");
    if constexpr (R == 34) asm volatile("v_accvgpr_write_b32 a194, %0" : : "v"(x) : "a194");
    if constexpr (R == 35) asm volatile("v_accvgpr_write_b32 a195, %0" : : "v"(x) : "a195");
    if constexpr (R == 36) asm volatile("v_accvgpr_write_b32 a196, %0" : : "v"(x) : "a196");
    if constexpr (R == 37) asm volatile("v_accvgpr_write_b32 a197, %0" : : "v"(x) : "a197");
    if constexpr (R == 38) asm volatile("v_accvgpr_write_b32 a198, %0" : : "v"(x) : "a198");
    if constexpr (R == 39) asm volatile("v_accvgpr_write_b32 a199, %0" : : "v"(x) : "a199");
    if constexpr (R == 40) asm volatile("v_accvgpr_write_b32 a200, %0" : : "v"(x) : "a200");
    if constexpr (R == 41) asm volatile("v_accvgpr_write_b32 a201, %0" : : "v"(x) : "a201");
    if constexpr (R == 42) asm volatile("v_accvgpr_write_b32 a202, %0" : : "v"(x) : "a202");
    if constexpr (R == 43) asm volatile("v_accvgpr_write_b32 a203, %0" : : "v"(x) : "a203");
    if constexpr (R == 44) asm volatile("v_accvgpr_write_b32 a204, %0" : : "v"(x) : "a204");
    if constexpr (R == 45) asm volatile("v_accvgpr_write_b32 a205, %0" : : "v"(x) : "a205");
    if constexpr (R == 46) asm volatile("v_accvgpr_write_b32 a206, %0" : : "v"(x) : "a206");
    if constexpr (R == 47) asm volatile("v_accvgpr_write_b32 a207, %0" : : "v"(x) : "a207");
    if constexpr (R == 48) asm volatile("v_accvgpr_write_b32 a208, %0" : : "v"(x) : "a208");
    if constexpr (R == 49) asm volatile("v_accvgpr_write_b32 a209, %0" : : "v"(x) : "a209");
    if constexpr (R == 50) asm volatile("v_accvgpr_write_b32 a210, %0" : : "v"(x) : "a210");
    if constexpr (R == 51) asm volatile("v_accvgpr_write_b32 a211, %0" : : "v"(x) : "a211");
    if constexpr (R == 52) asm volatile("v_accvgpr_write_b32 a212, %0" : : "v"(x) : "a212");
    if constexpr (R == 53) asm volatile("v_accvgpr_write_b32 a213, %0" : : "v"(x) : "a213");
    if constexpr (R == 54) asm volatile("v_accvgpr_write_b32 a214, %0" : : "v"(x) : "a214");
    if constexpr (R == 55) asm volatile("v_accvgpr_write_b32 a215, %0" : : "v"(x) : "a215");
    if constexpr (R == 56) asm volatile("v_accvgpr_write_b32 a216, %0" : : "v"(x) : "a216");
    if constexpr (R == 57) asm volatile("v_accvgpr_write_b32 a217, %0" : : "v"(x) : "a217");
    if constexpr (R == 58) asm volatile("v_accvgpr_write_b32 a218, %0" : : "v"(x) : "a218");
    if constexpr (R == 59) asm volatile("v_accvgpr_write_b32 a219, %0" : : "v"(x) : "a219");
    if constexpr (R == 60) asm volatile("v_accvgpr_write_b32 a220, %0" : : "v"(x) : "a220");
    if constexpr (R == 61) asm volatile("v_accvgpr_write_b32 a221, %0" : : "v"(x) : "a221");
    if constexpr (R == 62) asm volatile("v_accvgpr_write_b32 a222, %0" : : "v"(x) : "a222");
    if constexpr (R == 63) asm volatile("v_accvgpr_write_b32 a223, %0" : : "v"(x) : "a223");
}
template <int R>
__device__ __forceinline__ float wl_const_read() {               // a[160 + R]
    float x;
    if constexpr (R == 0) asm volatile("v_accvgpr_read_b32 %0, a160" : "=v"(x));
    if constexpr (R == 1) asm volatile("v_accvgpr_read_b32 %0, a161" : "=v"(x));
    if constexpr (R == 2) asm volatile("v_accvgpr_read_b32 %0, a162" : "=v"(x));
    if constexpr (R == 3) asm volatile("v_accvgpr_read_b32 %0, a163" : "=v"(x));
    if constexpr (R == 4) asm volatile("v_accvgpr_read_b32 %0, a164" : "=v"(x));
    if constexpr (R == 5) asm volatile("v_accvgpr_read_b32 %0, a165" : "=v"(x));
    if constexpr (R == 6) asm volatile("v_accvgpr_read_b32 %0, a166" : "=v"(x));
    if constexpr (R == 7) asm volatile("v_accvgpr_read_b32 %0, a167" : "=v"(x));
    if constexpr (R == 8) asm volatile("v_accvgpr_read_b32 %0, a168" : "=v"(x));
    if constexpr (R == 9) asm volatile("v_accvgpr_read_b32 %0, a169" : "=v"(x));
    if constexpr (R == 10) asm volatile("v_accvgpr_read_b32 %0, a170" : "=v"(x));
    if constexpr (R == 11) asm volatile("v_accvgpr_read_b32 %0, a171" : "=v"(x));
    if constexpr (R == 12) asm volatile("v_accvgpr_read_b32 %0, a172" : "=v"(x));
    if constexpr (R == 13) asm volatile("v_accvgpr_read_b32 %0, a173" : "=v"(x));
    if constexpr (R == 14) asm volatile("v_accvgpr_read_b32 %0, a174" : "=v"(x));
    if constexpr (R == 15) asm volatile("v_accvgpr_read_b32 %0, a175" : "=v"(x));
    if constexpr (R == 16) asm volatile("v_accvgpr_read_b32 %0, a176" : "=v"(x));
    if constexpr (R == 17) asm volatile("v_accvgpr_read_b32 %0, a177" : "=v"(x));
    if constexpr (R == 18) asm volatile("v_accvgpr_read_b32 %0, a178" : "=v"(x));
    if constexpr (R == 19) asm volatile("v_accvgpr_read_b32 %0, a179" : "=v"(x));
    if constexpr (R == 20) asm volatile("v_accvgpr_read_b32 %0, a180" : "=v"(x));
    if constexpr (R == 21) asm volatile("v_accvgpr_read_b32 %0, a181" : "=v"(x));
    if constexpr (R == 22) asm volatile("v_accvgpr_read_b32 %0, a182" : "=v"(x));
    if constexpr (R == 23) asm volatile("v_accvgpr_read_b32 %0, a183" : "=v"(x));
    if constexpr (R == 24) asm volatile("v_accvgpr_read_b32 %0, a184" : "=v"(x));
    if constexpr (R == 25) asm volatile("v_accvgpr_read_b32 %0, a185" : "=v"(x));
    if constexpr (R == 26) asm volatile("v_accvgpr_read_b32 %0, a186" : "=v"(x));
    if constexpr (R == 27) asm volatile("v_accvgpr_read_b32 %0, a187" : "=v"(x));
    if constexpr (R == 28) asm volatile("v_accvgpr_read_b32 %0, a188" : "=v"(x));
    if constexpr (R == 29) asm volatile("v_accvgpr_read_b32 %0, a189" : "=v"(x));
    if constexpr (R == 30) asm volatile("v_accvgpr_read_b32 %0, a190" : "=v"(x));
    if constexpr (R == 31) asm volatile("v_accvgpr_read_b32 %0, a191" : "=v"(x));
    if constexpr (R == 32) asm volatile("v_accvgpr_read_b32 %0, a192" : "=v"(x));
    if constexpr (R == 33) asm volatile("v_accvgpr_read_b32 %0, a193" : "=v"(x));
    if constexpr (R == 34) asm volatile("v_accvgpr_read_b32 %0, a194" : "=v"(x));
    if constexpr (R == 35) asm volatile("v_accvgpr_read_b32 %0, a195" : "=v"(x));
    if constexpr (R == 36) asm volatile("v_accvgpr_read_b32 %0, a196" : "=v"(x));
    if constexpr (R == 37) asm volatile("v_accvgpr_read_b32 %0, a197" : "=v"(x));
    if constexpr (R == 38) asm volatile("v_accvgpr_read_b32 %0, a198" : "=v"(x));
    if constexpr (R == 39) asm volatile("v_accvgpr_read_b32 %0, a199" : "=v"(x));
    if constexpr (R == 40) asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(x));
    if constexpr (R == 41) asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(x));
    if constexpr (R == 42) asm volatile("v_accvgpr_read_b32 %0, a202" : "=v"(x));
    if constexpr (R == 43) asm volatile("v_accvgpr_read_b32 %0, a203" : "=v"(x));
    if constexpr (R == 44) asm volatile("v_accvgpr_read_b32 %0, a204" : "=v"(x));
    if constexpr (R == 45) asm volatile("v_accvgpr_read_b32 %0, a205" : "=v"(x));
    if constexpr (R == 46) asm volatile("v_accvgpr_read_b32 %0, a206" : "=v"(x));
    if constexpr (R == 47) asm volatile("v_accvgpr_read_b32 %0, a207" : "=v"(x));
    if constexpr (R == 48) asm volatile("v_accvgpr_read_b32 %0, a208" : "=v"(x));
    if constexpr (R == 49) asm volatile("v_accvgpr_read_b32 %0, a209" : "=v"(x));
    if constexpr (R == 50) asm volatile("v_accvgpr_read_b32 %0, a210" : "=v"(x));
    if constexpr (R == 51) asm volatile("v_accvgpr_read_b32 %0, a211" : "=v"(x));
    if constexpr (R == 52) asm volatile("v_accvgpr_read_b32 %0, a212" : "=v"(x));
    if constexpr (R == 53) asm volatile("v_accvgpr_read_b32 %0, a213" : "=v"(x));
    if constexpr (R == 54) asm volatile("v_accvgpr_read_b32 %0, a214" : "=v"(x));
    if constexpr (R == 55) asm volatile("v_accvgpr_read_b32 %0, a215" : "=v"(x));
    if constexpr (R == 56) asm volatile("v_accvgpr_read_b32 %0, a216" : "=v"(x));
    if constexpr (R == 57) asm volatile("v_accvgpr_read_b32 %0, a217" : "=v"(x));
    if constexpr (R == 58) asm volatile("v_accvgpr_read_b32 %0, a218" : "=v"(x));
    if constexpr (R == 59) asm volatile("v_accvgpr_read_b32 %0, a219" : "=v"(x));
    if constexpr (R == 60) asm volatile("v_accvgpr_read_b32 %0, a220" : "=v"(x));
    if constexpr (R == 61) asm volatile("v_accvgpr_read_b32 %0, a221" : "=v"(x));
    if constexpr (R == 62) asm volatile("v_accvgpr_read_b32 %0, a222" : "=v"(x));
    if constexpr (R == 63) asm volatile("v_accvgpr_read_b32 %0, a223" : "=v"(x));
    return x;
}
template <int I>
__device__ __forceinline__ void wl_mfma_zero(f16v& d, const h8& x) {       // D = W(a[4 I ..]) x X (first MFMA of a tile)
    if constexpr (I == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %1, 0" : "=&v"(d) : "v"(x));
    if constexpr (I == 20) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[80:83], %1, 0" : "=&v"(d) : "v"(x));
}

__global__ void __launch_bounds__(WgCfg::THREADS, 1) gemm_ws320_lnf_kernel(GemmArgs p, unsigned a_bytes, unsigned s_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_stats), 0, (int)s_bytes, 0x00020000);

    const int ntiles = p.tiles_m;
    // the block -> (row stream, column block) map of gemm_ws320_geglu_kernel, spare-CU streams included
    int cb, G, t_first;
    ws_block_map<true>(p.tiles_n, cb, G, t_first);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int wcol = cb * WgCfg::TBN + wave * 64;          // first output column (weight row) of the wave
    const bool wave_on = wcol < p.N;                       // N % 64 == 0: a wave has all of its 64 columns or none

    // A fragments: lane (i = lq, hi) holds W'[wcol + 32 ab + i][16 kk + 8 hi .. + 7] (wg_load_w: the owned registers a[0:159]);
    // colsum / bias' of the lane's 2 x 16 columns (accumulator register r = 4 q + c <-> column 32 ab + 8 q + 4 hi + c) in a[160:223]
    {
        const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const half_t* wrow = p.W + (size_t)(wave_on ? wcol + lq : 0) * p.ldw + hi * 8;
        static_for_ws<2>([&](auto AB_) __attribute__((always_inline)) {         // twenty loads in flight, then their register writes
            constexpr int ab = decltype(AB_)::value;
            h8 wt[WG_KS];
#pragma unroll
            for (int kk = 0; kk < WG_KS; ++kk) wt[kk] = *reinterpret_cast<const h8*>(wrow + (size_t)(wave_on ? ab * 32 : 0) * p.ldw + kk * 16);
            static_for_ws<WG_KS>([&](auto KK_) __attribute__((always_inline)) {
                constexpr int kk = decltype(KK_)::value;
                wg_load_w<WG_KS * ab + kk>(wave_on ? wt[kk] : zero8);
            });
        });
        f4 cs[8], bq[8];                                  // (all sixteen loads in flight, then their register writes)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = (wave_on ? wcol : 0) + 32 * (i >> 2) + 8 * (i & 3) + 4 * hi;
            cs[i] = *reinterpret_cast<const f4*>(p.ln_colsum + n);
            bq[i] = (p.flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + n) : f4{0.f, 0.f, 0.f, 0.f};
        }
        static_for_ws<8>([&](auto Q_) __attribute__((always_inline)) {
            constexpr int i = decltype(Q_)::value, ab = i / 4, q = i % 4;
            wl_const_write<16 * ab + 4 * q + 0>(cs[i][0]); wl_const_write<16 * ab + 4 * q + 1>(cs[i][1]);
            wl_const_write<16 * ab + 4 * q + 2>(cs[i][2]); wl_const_write<16 * ab + 4 * q + 3>(cs[i][3]);
            wl_const_write<32 + 16 * ab + 4 * q + 0>(bq[i][0]); wl_const_write<32 + 16 * ab + 4 * q + 1>(bq[i][1]);
            wl_const_write<32 + 16 * ab + 4 * q + 2>(bq[i][2]); wl_const_write<32 + 16 * ab + 4 * q + 3>(bq[i][3]);
        });
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }

    // LDS-DMA of one 64-row tile (as gemm_ws320_geglu_kernel)
    const int drow = wave * 8 + (lane >> 3);
    const unsigned dsrc = (unsigned)((lane & 7) ^ ((drow >> 1) & 7)) * 16u;
    const unsigned rstep32 = 32u * (unsigned)p.lda * 2u;
    int issued = 0;
    int mark[WG_RING];
    auto issue_tile = [&](int t, int buf) {
        const int m0 = p.m_begin + t * WgCfg::TBM + drow;
        const unsigned base = (unsigned)m0 * (unsigned)p.lda * 2u + dsrc;
        const unsigned va = m0 < p.M ? base : OOB, vb = m0 + 32 < p.M ? base + rstep32 : OOB;
        unsigned char* dst = smem_raw + buf * WL_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < WG_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dst + (i >> 1) * 8192 + (i & 1) * 4096), 16, (i & 1) ? vb : va,
                                                     (unsigned)(i >> 1) * (BK * 2), 0, 0);
        issued += WG_PIECES;
        // the (mean, rstd) pairs of the tile's 64 rows ride along: 512 contiguous bytes of ln_stats behind the stage's activation rows,
        // one more piece of wave 0 (lanes 0-31 fetch two rows each; rows >= M read zeros through the descriptor's range check).  Through
        // LDS, not into registers: a register load that is used a tile later gets a compiler-placed s_waitcnt in front of that use which
        // must hold on every path into the tile - vmcnt(3), i.e. a drain of the DMA pieces just issued, in the first version of this kernel
        if (wave == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_s, (lds_ptr_t)(smem_raw + buf * WL_STAGE + WG_STAGE), 16,
                                                     lane < 32 ? (unsigned)(p.m_begin + t * WgCfg::TBM) * 8u + (unsigned)lane * 16u : OOB, 0, 0, 0);
            issued += 1;
        }
        mark[buf] = issued;
    };
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xo[i] = lds_off(lq, 2 * i + hi);

    f16v acc[2][2][2];                        // [accumulator set][column half ab][row block]
    unsigned coff[2][2];                      // byte offset of (lane's row in row block mb, wave's first column + 8 hi) of the tile in each accumulator set
    f2v st[2][2];                             // (mean, rstd) of the lane's row in row block mb of the tile in each accumulator set
    float rb[2], qb[2];                       // alpha rstd, -alpha rstd mean of the PENDING tile's rows
    float ho[2][2];
    u2v pk[2][4];

    // The finished tile's epilogue in 40 chunks (sched_barrier(0) on both sides of every MFMA and chunk: the order below IS the
    // instruction stream).  Per column half ab: chunks 0 .. 15 = accumulator register r in both row blocks, chunks 16 .. 19 =
    // (row block, column pair k): half exchange + one 16-byte store.
    auto chunk = [&](auto PAR_, auto C_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, ab = decltype(C_)::value / 20, c = decltype(C_)::value % 20;
        if constexpr (c < 16) {
            constexpr int r = c;
            const float cs = wl_const_read<16 * ab + r>(), bn = wl_const_read<32 + 16 * ab + r>();
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float t;
                // as the instructions they are, in this order: left to itself hipcc pairs them into v_pk_fma_f32 with half-swapped
                // operands (the form tools/isa_audit.py rejects library-wide)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(qb[mb]), "v"(cs), "v"(bn));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ho[mb][r & 1]) : "v"(acc[PAR][ab][mb][r]), "v"(rb[mb]), "v"(t));
            }
            if constexpr (r & 1) {            // outputs r - 1, r = word (r >> 1) & 1 of column group r >> 2
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    pk[mb][r >> 2][(r >> 1) & 1] = __builtin_bit_cast(unsigned, h2v{(half_t)ho[mb][0], (half_t)ho[mb][1]});
            }
        } else {
            // column group q holds columns 8 q + 4 hi .. + 3: after the half swap lanes 0-31 own columns 16 k .. + 7, lanes 32-63 columns 16 k + 8 .. + 15
            constexpr int mb = (c - 16) >> 1, k = (c - 16) & 1;
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[mb][2 * k][0], pk[mb][2 * k + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[mb][2 * k][1], pk[mb][2 * k + 1][1], false, false);
            const epi_u4v w = {s0[0], s1[0], s0[1], s1[1]};
#if !(VCX_WL_ABL & 8)
            __builtin_amdgcn_raw_buffer_store_b128(w, srd_c, coff[PAR][mb], ab * 64 + k * 32, 0);
#endif
            asm volatile("s_nop 1" : : "v"(w));          // (the wide-store rule of tools/isa_audit.py)
            issued += 1;
        }
    };
    // the pending tile's row terms out of the (mean, rstd) pairs requested one tile ago
    auto row_terms = [&](auto PAR_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            rb[mb] = p.alpha * st[PAR][mb][1];
            qb[mb] = -rb[mb] * st[PAR][mb][0];
        }
    };

    // one tile: its 80 MFMAs into accumulator set PAR; PEND: the other set holds a finished tile whose chunks ride behind MFMAs 6 .. 45
    auto tile = [&](auto PAR_, auto PEND_, int t, int i) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool PEND = decltype(PEND_)::value != 0;
        const int buf = i % WG_RING;
        __builtin_amdgcn_sched_barrier(0);
        ws_wait_vmcnt(issued - mark[buf]);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + (WG_AHEAD + 1) * G < ntiles) issue_tile(t + (WG_AHEAD + 1) * G, (i + WG_AHEAD + 1) % WG_RING);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int m = p.m_begin + t * WgCfg::TBM + 32 * mb + lq;
            st[PAR][mb] = *reinterpret_cast<const f2v*>(smem_raw + buf * WL_STAGE + WG_STAGE + (32 * mb + lq) * 8);      // (its stage is re-filled at the top of the next tile)
            coff[PAR][mb] = (m < p.M && wave_on) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(wcol + 8 * hi)) * 2u : OOB;
        }
        if constexpr (PEND) row_terms(WInt<PAR ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
        unsigned xb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xb[j] = (unsigned)(buf * WL_STAGE) + 2u * (unsigned)xo[j];
            asm volatile("" : "+v"(xb[j]));
        }
        auto frag = [&](int kk, int mb) __attribute__((always_inline)) {
            return *reinterpret_cast<const h8*>(smem_raw + xb[kk & 3] + (kk >> 2) * (WgCfg::TBM * BK * 2) + mb * 4096);
        };
        constexpr int XR = 4;
        h8 xr[XR][2];
#pragma unroll
        for (int k = 0; k < XR; ++k)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) xr[k][mb] = frag(k, mb);
        static_for_ws<WG_KS>([&](auto KK_) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK_)::value;
            static_for_ws<4>([&](auto J_) __attribute__((always_inline)) {
                constexpr int j = decltype(J_)::value, ab = j & 1, mb = j >> 1, n = 4 * kk + j;
                __builtin_amdgcn_sched_barrier(0);
#if VCX_WL_ABL & 2
                if constexpr (kk == 0) acc[PAR][ab][mb] = f16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; else acc[PAR][ab][mb][n & 15] += (float)xr[kk % XR][mb][0] * (float)xr[kk % XR][mb][1];
#else
                if constexpr (kk == 0) wl_mfma_zero<WG_KS * ab>(acc[PAR][ab][mb], xr[kk % XR][mb]);
                else wg_mfma_acc<WG_KS * ab + kk>(acc[PAR][ab][mb], xr[kk % XR][mb]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ab == 1 && kk + XR < WG_KS) xr[kk % XR][mb] = frag(kk + XR, mb);
#if VCX_WL_ABL & 1        // timing-only builds (tools/ws_lnf_ab.py <library>): 1 no arithmetic chunks, 2 no MFMAs, 8 no stores
                if constexpr (PEND && n >= 6 && n < 46 && (n - 6) % 20 >= 16) chunk(WInt<PAR ^ 1>{}, WInt<n - 6>{});
#else
                if constexpr (PEND && n >= 6 && n < 46) chunk(WInt<PAR ^ 1>{}, WInt<n - 6>{});
#endif
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    auto drain = [&](auto PAR_) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs (asm: hipcc does not know their latency) have written their results
        __builtin_amdgcn_sched_barrier(0);
        row_terms(PAR_);
        __builtin_amdgcn_sched_barrier(0);
        static_for_ws<40>([&](auto C_) __attribute__((always_inline)) { chunk(PAR_, C_); });
    };

    ws_walk<WG_AHEAD>(t_first, G, ntiles, issue_tile, tile, [&](auto PAR_, int) __attribute__((always_inline)) { drain(PAR_); });
#endif
}

// Blocks of a launch: 8 XCDs x (row streams per XCD) x tiles_n column blocks - fewer streams than the chip has room for when the problem
// has fewer row tiles - plus, with `spare`, whole streams across XCDs on the CUs that per_xcd / tiles_n leaves over (fewer than 8 of them:
// ws_block_map<true> reads their number as G & 7; ten column blocks: 250 blocks instead of 240).
int ws_grid(const GemmArgs& a, bool spare) {
    const int per_xcd = persistent_grid(1 << 30, 1) / 8;
    int streams_per_xcd = per_xcd / a.tiles_n;
    if (streams_per_xcd < 1) streams_per_xcd = 1;
    const int needed = (a.tiles_m + 7) / 8;
    int spare_streams = 0;
    if (spare && streams_per_xcd <= needed && 8 * streams_per_xcd * a.tiles_n < 8 * per_xcd)
        spare_streams = (8 * per_xcd - 8 * streams_per_xcd * a.tiles_n) / a.tiles_n;
    if (streams_per_xcd > needed) streams_per_xcd = needed;
    if (spare_streams > 7) spare_streams = 7;
    return (8 * streams_per_xcd + spare_streams) * a.tiles_n;
}

int launch_ws_lnf(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_lnf_kernel;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WL_SMEM, "vcx_gemm_f16(ws320 lnfold)")) return VCX_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(ws_grid(a, true)), dim3(WgCfg::THREADS), WL_SMEM, s, a, a.a_bytes, (unsigned)(8ull * (unsigned long long)a.M));
    return vcx_check_launch("vcx_gemm_f16(ws320 lnfold)");
}

int launch_ws_geglu(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_geglu_kernel;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WG_SMEM, "vcx_gemm_f16(ws320 geglu)")) return VCX_ELAUNCH;
    // (no spare-CU streams under knob GEMM_WS = 3: the A/B setting of tools/ws_geglu_ab.py, one-XCD streams only)
    hipLaunchKernelGGL(kern, dim3(ws_grid(a, vcx_tune(VCX_TUNE_GEMM_WS) != 3)), dim3(WgCfg::THREADS), WG_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320 geglu)");
}

template <bool RES, bool RS>
int launch_ws_pipe(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_pipe_kernel<RES, RS>;
    constexpr size_t smem = RS ? WP_SMEM_RS : WP_SMEM;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)smem, "vcx_gemm_f16(ws320 pipe)")) return VCX_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(ws_grid(a, false)), dim3(WpCfg::THREADS), smem, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320 pipe)");
}

template <int MODE>
int launch_ws(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_ws320_kernel<MODE>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)WS_SMEM, "vcx_gemm_f16(ws320)")) return VCX_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(ws_grid(a, false)), dim3(WsCfg::THREADS), WS_SMEM, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_f16(ws320)");
}

}  // namespace

// Linear mode, K = 320, N = 320 j (j <= 4), fp16 output, no GEGLU / LNFOLD / LNFOLD_T / BIAS_M,
// 32-bit operand and output extents (the caller checks; it also fills a_bytes / c_bytes / r_bytes).  Sets the tiling itself.
// One launch, `units` weight / bias sets (vcx_gemm_units_f16): N = K = 320, bias at most, unit_rows % 32 == 0 (the caller checks).
int vcxgemm::launch_ws320_units(GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds, lds_rs;
    const bool rs = (a.flags & VCX_GEMM_ROWSTATS) != 0;
    auto kern = rs ? gemm_ws320_pipe_kernel<false, true> : gemm_ws320_pipe_kernel<false, false>;
    const size_t smem = rs ? WP_SMEM_RS : WP_SMEM;
    if (!(rs ? lds_rs : lds).ensure(reinterpret_cast<const void*>(kern), (int)smem, "vcx_gemm_units_f16(ws320 pipe)")) return VCX_ELAUNCH;
    a.tiles_n = 1;
    a.tiles_m = a.unit_rows / WpCfg::TBM;
    const int cus = persistent_grid(1 << 30, 1);
    int bpu = cus / a.units;                        // blocks per unit: the chip's CUs shared out, at least one, at most one per row tile
    if (bpu < 1) bpu = 1;
    if (bpu > a.tiles_m) bpu = a.tiles_m;
    hipLaunchKernelGGL(kern, dim3(bpu * a.units), dim3(WpCfg::THREADS), smem, s, a, a.a_bytes);
    return vcx_check_launch("vcx_gemm_units_f16(ws320 pipe)");
}

// GEGLU, linear mode, K = 320, N % 256 == 0 (packed columns), bias at most, fp16 output, 32-bit extents (the caller checks).
int vcxgemm::launch_ws320_geglu(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M - a.m_begin + WgCfg::TBM - 1) / WgCfg::TBM;
    a.tiles_n = a.N / WgCfg::TBN;
    return launch_ws_geglu(a, s);
}

// LayerNorm-folded projection (VCX_GEMM_LNFOLD), linear mode, K = 320, N % 64 == 0, bias at most, fp16 output, 32-bit extents, 8 M < 4 GiB (the caller checks).
int vcxgemm::launch_ws320_lnfold(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M - a.m_begin + WgCfg::TBM - 1) / WgCfg::TBM;
    a.tiles_n = (a.N + WgCfg::TBN - 1) / WgCfg::TBN;
    return launch_ws_lnf(a, s);
}

int vcxgemm::launch_ws320(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M - a.m_begin + WsCfg::TBM - 1) / WsCfg::TBM;
    a.tiles_n = a.N / WsCfg::TBN;
    if (a.flags & VCX_GEMM_COLSTATS) return launch_ws<3>(a, s);
    if (a.flags & VCX_GEMM_ROWADD) return launch_ws<2>(a, s);
    a.tiles_m = (a.M - a.m_begin + WpCfg::TBM - 1) / WpCfg::TBM;
    if (a.flags & VCX_GEMM_ROWSTATS) return (a.flags & VCX_GEMM_RESIDUAL) ? launch_ws_pipe<true, true>(a, s) : launch_ws_pipe<false, true>(a, s);
    return (a.flags & VCX_GEMM_RESIDUAL) ? launch_ws_pipe<true, false>(a, s) : launch_ws_pipe<false, false>(a, s);
}
