// GEGLU projection (GEGLU.proj + gate, /root/reference lvdm/modules/attention.py:415-422) with a DEFERRED epilogue.
//
// gemm_dma.hip runs a tile's epilogue after a block-wide barrier: for the GEGLU layers that is 64 exact-erf GELUs, 128 bias
// multiply-adds and 8 wide stores per lane with the matrix pipe idle - as long as the whole main loop at K = 320
// (profiles/r02_experiments.md section 9: ~6 us of a 13.7 us tile).  Here the epilogue of tile i rides between the MFMAs of
// tile i+1:
//
//   * after the last K-step of a tile a lane's 64 (value, gate) pairs get their bias and are rounded to fp16 - 64 packed
//     registers (`sx`, `sg`) instead of 128 fp32 accumulators, which then start the next tile at zero.  fp16 is what the reference
//     holds at this point (the Linear's output under autocast); GELU is still evaluated in fp32 on the rounded gate and rounded once,
//     and value x gelu(gate) is ONE correctly rounded fp16 multiply (v_pk_mul_f16) - the reference's own sequence of roundings;
//   * during K-steps 0..3 of the next tile, 16-row group b = K-step of the saved tile is finished: one GELU per 4-MFMA group
//     (sixteen groups per K-step), the packed multiplies, the v_permlane16_swap widening and two dwordx4 stores - VALU work issued
//     into the shadow of the wave's own MFMAs;
//   * the stores are issued AFTER the K-step's LDS-DMA, and the K-step ends on `s_waitcnt vmcnt(2)` + a raw `s_barrier`: the vector
//     memory counter retires in order, so this waits for every DMA piece (older) and lets the two stores (youngest) fly
//     through the barrier.  (`__syncthreads()` would drain them: its fence is `vmcnt(0)`.)
//   * the last tile of a block is drained after the loop.
//
// Linear mode, fp16 in / out, BIAS_N optional, K >= 320 (five K-steps: four carry a slice, the last one prefetches the bias).
// Same tile (256 x 256, 4 x 2 waves), LDS image, DMA addressing and persistent XCD-aware walk as gemm_dma.hip; the two agree
// to fp16 rounding (tests/test_kernels_gpu.py::test_gemm_geglu_deferred_matches_phased).
#include <type_traits>
#include <utility>
#include "gemm_args.h"

using namespace vcxgemm;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFFFFu;
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int TBM = 256, TBN = 256, NWM = 4, NWN = 2, THREADS = 512;
constexpr int WM = TBM / NWM, WN = TBN / NWN;          // wave tile 64 x 128 (packed columns)
[[maybe_unused]] constexpr int MFRAG = WM / 16, NFRAG = WN / 16;        // 4 x 8 accumulator fragments
[[maybe_unused]] constexpr int NOUT = NFRAG / 2;                        // output fragments per 16-row group
[[maybe_unused]] constexpr int XROWS = TBM * 8 / THREADS, WROWS = TBN * 8 / THREADS, RSTEP = THREADS / 8;
constexpr size_t STAGES = (size_t)2 * (TBM + TBN) * BK * sizeof(half_t);
[[maybe_unused]] constexpr int RGROUPS = MFRAG - 1;                     // row groups of the saved tile kept in registers; the last one waits in LDS
constexpr size_t WAVE_LDS = 64 * NOUT * 16;            // per wave: the saved tile's last row group (16 bytes per lane and output fragment); the
                                                       // bias strip (WN floats) lives in the same bytes while a tile is packed
constexpr size_t SMEM = STAGES + 8 * WAVE_LDS;         // 128 KB + 32 KB: all of the CU's 160 KB, one block per CU

__device__ __forceinline__ constexpr int xfrag(int a) { return 4 * (a >> 1) + (a & 1); }

template <int I> using Int = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(Int<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }   // packed fragment of output fragment a's values; + 2: its gates

__global__ void __launch_bounds__(THREADS, 2) gemm_geglu_deferred_kernel(GemmArgs p, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sX = reinterpret_cast<half_t*>(smem_raw);
    half_t* sW = sX + 2 * TBM * BK;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);

    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int chunk = tid & 7;
    const int r0 = tid >> 3;
    const int wm = wave % NWM, wn = wave / NWM;
    unsigned char* sL = smem_raw + STAGES + wave * WAVE_LDS;      // the wave's private LDS patch
    float* sB = reinterpret_cast<float*>(sL);                       // ... as the strip of column addends

    // DMA source offsets of the tile being loaded.  A thread's rows are r0 + 64 i: their swizzled source chunk is the same for every
    // i and their byte offsets differ by a uniform step, so one VGPR per operand is kept and the rest is rebuilt at issue time
    // (rows >= M / >= N are sent out of range there: the descriptor's range check returns zeros).
    unsigned xoff0 = 0, woff0 = 0;
    int lm0 = 0, ln0 = 0;        // first row / weight row of this thread in the load tile
    const unsigned xstep = (unsigned)RSTEP * (unsigned)p.lda * 2u, wstep = (unsigned)RSTEP * (unsigned)p.ldw * 2u;
    const unsigned csrc = (unsigned)(chunk ^ ((r0 >> 1) & 7)) * 16u;     // (RSTEP i) >> 1 is a multiple of 8: the same for every i
    auto init_load = [&](int t) {
        int tile_m, tile_n;
        tile_coords(t, ntiles, p.tiles_n, tile_m, tile_n);
        lm0 = p.m_begin + tile_m * TBM + r0;
        ln0 = tile_n * TBN + r0;
        xoff0 = (unsigned)lm0 * (unsigned)p.lda * 2u + csrc;        // < 4 GiB for every row < M (checked by the caller)
        woff0 = (unsigned)ln0 * (unsigned)p.ldw * 2u + csrc;
    };
    auto load_tile = [&](int kt, int buf) {
        half_t* dx = sX + buf * TBM * BK + wave * 8 * BK;
        half_t* dw = sW + buf * TBN * BK + wave * 8 * BK;
        const unsigned soff = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const unsigned v = lm0 + RSTEP * i < p.M ? xoff0 + (unsigned)i * xstep : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + RSTEP * i * BK), 16, v, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const unsigned v = ln0 + RSTEP * i < p.N ? woff0 + (unsigned)i * wstep : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(dw + RSTEP * i * BK), 16, v, soff, 0, 0);
        }
    };

    f4 acc[NFRAG][MFRAG];

    // the previous tile's outputs-to-be: per output fragment a and 16-row group b the packed fp16 pairs (v0 v1)(v2 v3) of the
    // biased values and of the biased gates, and where they go
    unsigned sx[NOUT][RGROUPS][2], sg[NOUT][RGROUPS][2];
    u4v s3[NOUT];                // row group 3 (x0 x1 g0 g1 per output fragment): in LDS between the pack and its K-step
    unsigned s_coff0 = 0;        // byte offset of the lane's 8 output columns of fragment pair 0 in its row of group 0
    int s_ncol0 = 0;             // packed-space column of the wave's strip in the saved tile (column-range check of the stores)
    const unsigned cstep = 32u * (unsigned)p.ldc;

    // ---- one element of the saved tile: GELU of gate r of output fragment a in row group S; the packed multiplies and the
    // stores follow as soon as their inputs are complete.  J = 4 a + r runs 0..15 over a K-step's sixteen MFMA groups.
    float ge[4];
    unsigned ow[2][2];
    auto slice_part = [&](auto S_, auto J_) {
        constexpr int S = decltype(S_)::value, J = decltype(J_)::value;
        constexpr int a = J >> 2, r = J & 3;
        unsigned gw, xw;
        if constexpr (S < RGROUPS) { gw = sg[a][S][r >> 1]; xw = sx[a][S][r >> 1]; }
        else { gw = s3[a][2 + (r >> 1)]; xw = s3[a][r >> 1]; }
        const h2v gp = __builtin_bit_cast(h2v, gw);
        ge[r] = gelu_erf((float)gp[r & 1]);
        asm volatile("" : "+v"(ge[r]));      // evaluated HERE, among this group's MFMAs (the optimiser would sink it to its use, one group on)
        if constexpr (r == 1 || r == 3) {
            const h2v e = {(half_t)ge[r - 1], (half_t)ge[r]};
            const h2v xv = __builtin_bit_cast(h2v, xw);
            ow[a & 1][r >> 1] = __builtin_bit_cast(unsigned, xv * e);
        }
        if constexpr (r == 3 && (a & 1)) {
            // fragments a - 1 (vdst) and a (src): even-lg lanes end up with 8 contiguous columns of a - 1, odd-lg lanes of a
            const auto s0 = __builtin_amdgcn_permlane16_swap(ow[0][0], ow[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(ow[0][1], ow[1][1], false, false);
            // N is a multiple of 64 and the strip starts on a multiple of 128: a fragment pair (64 packed columns) is inside N or not
            // as a whole - a wave-uniform test.  The row-group step must ride in the VGPR offset (the range check that drops rows >= M
            // does not see soffset); the pair's column step may ride in soffset.
            const bool inside = s_ncol0 + 64 * (a >> 1) + 64 <= p.N;
            const unsigned voff = inside ? s_coff0 + (unsigned)S * cstep : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u4v{s0[0], s1[0], s0[1], s1[1]}, srd_c, voff, (a >> 1) * 64, 0);
        }
    };

    const int nk = p.K / BK;
    int ltile = blockIdx.x, lkt = 0;
    int cur = 0;
    f4 breg = {0.f, 0.f, 0.f, 0.f};

    // ---- one K-step of the current tile.  S >= 0: row group S of the saved tile is finished between the MFMAs.
    // FIRST: the tile's first K-step - its first MFMA per accumulator takes the constant 0 as C, so the accumulators need no zeroing
    // and are not live between a tile's pack and their first MFMA (the saved tile's registers and the accumulators overlap in time
    // only as far as the slices have not been consumed yet: 128 + 64 registers are never live together).
    auto kstep = [&](auto S_, auto FIRST_, bool last, int tile_n) {
        constexpr int S = decltype(S_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        // lane-derived addresses are rebuilt from an opaque copy of the lane id in every K-step: hoisted out of the tile loop they would
        // sit in ~20 registers for the whole kernel, and this kernel has none to spare
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        if (++lkt == nk) {
            lkt = 0;
            ltile += G;
            if (ltile < ntiles) init_load(ltile);
        }
        if (ltile < ntiles) load_tile(lkt, cur ^ 1);
        if constexpr (S == RGROUPS) {
#pragma unroll
            for (int a = 0; a < NOUT; ++a) s3[a] = *reinterpret_cast<const u4v*>(sL + (a * 64 + lane_k) * 16);
        }
        if (last && (p.flags & VCX_GEMM_BIAS_N) && lane_k < WN / 4)     // the tile's bias strip: requested a K-step ahead of its use
            breg = *reinterpret_cast<const f4*>(p.bias + min(tile_n * TBN + wn * WN + lane_k * 4, p.N - 4));
        __builtin_amdgcn_sched_barrier(0);
        const half_t* cx = sX + cur * TBM * BK;
        const half_t* cw = sW + cur * TBN * BK;
        const int lr = lane_k & 15, lg = lane_k >> 4;
        {
            h8 xf[MFRAG];
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, lg));
            h8 wcur = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, lg));
            static_for<2 * NFRAG>([&](auto J_) __attribute__((always_inline)) {
                constexpr int J = decltype(J_)::value, kk = J / NFRAG, a = J % NFRAG;
                h8 wnext = wcur;
                if constexpr (a + 1 < NFRAG) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + (a + 1) * 16 + lr, kk * 4 + lg));
                else if constexpr (kk == 0) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, 4 + lg));
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) {
                    if constexpr (FIRST && kk == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur, xf[b], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur, xf[b], acc[a][b], 0, 0, 0);
                    if constexpr (a == NFRAG - 1 && kk == 0)
                        xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, 4 + lg));
                }
                wcur = wnext;
                if constexpr (S >= 0) {
                    slice_part(S_, J_);
#ifndef VCX_GEGLU_NO_GROUPS
                    // one MFMA, then a few of the element's VALU operations, four times: the fillers sit in the MFMAs' shadows
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int q = 0; q < MFRAG; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // The DMA of the next K-step must have landed and every wave must be done reading `cur` before the roles swap.  The (up to)
        // two output stores of this K-step were issued after the DMA: vmcnt(2) covers the DMA and lets them fly.
        if constexpr (S >= 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    };

    init_load(ltile);
    load_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool pending = false;
    for (int ctile = blockIdx.x; ctile < ntiles; ctile += G) {
        int tile_m, tile_n;
        tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
        int k0 = 1;
        if (pending) {
            kstep(Int<0>{}, Int<1>{}, false, tile_n);
            kstep(Int<1>{}, Int<0>{}, false, tile_n);
            kstep(Int<2>{}, Int<0>{}, false, tile_n);
            kstep(Int<3>{}, Int<0>{}, false, tile_n);
            k0 = MFRAG;
        } else {
            kstep(Int<-1>{}, Int<1>{}, false, tile_n);
        }
        for (int k = k0; k < nk; ++k) kstep(Int<-1>{}, Int<0>{}, k == nk - 1, tile_n);
        // ---- pack: bias, fp16 rounding
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        const int lr_p = lane_p & 15, lg_p = lane_p >> 4;
        if (lane_p < WN / 4) *reinterpret_cast<f4*>(sB + lane_p * 4) = breg;
        const float alpha = p.alpha;
        u4v l3[NOUT];
#pragma unroll
        for (int a = 0; a < NOUT; ++a) {
            const f4 bx = *reinterpret_cast<const f4*>(sB + xfrag(a) * 16 + lg_p * 4);
            const f4 bg = *reinterpret_cast<const f4*>(sB + (xfrag(a) + 2) * 16 + lg_p * 4);
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) {
                const f4 vx = acc[xfrag(a)][b], vg = acc[xfrag(a) + 2][b];
                float x_[4], g_[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x_[r] = __builtin_fmaf(vx[r], alpha, bx[r]);
                    g_[r] = __builtin_fmaf(vg[r], alpha, bg[r]);
                }
                const unsigned x01 = __builtin_bit_cast(unsigned, h2v{(half_t)x_[0], (half_t)x_[1]});
                const unsigned x23 = __builtin_bit_cast(unsigned, h2v{(half_t)x_[2], (half_t)x_[3]});
                const unsigned g01 = __builtin_bit_cast(unsigned, h2v{(half_t)g_[0], (half_t)g_[1]});
                const unsigned g23 = __builtin_bit_cast(unsigned, h2v{(half_t)g_[2], (half_t)g_[3]});
                if (b < RGROUPS) { sx[a][b][0] = x01; sx[a][b][1] = x23; sg[a][b][0] = g01; sg[a][b][1] = g23; }
                else l3[a] = u4v{x01, x23, g01, g23};
            }
        }
        // the last row group waits in LDS (over the bias strip, which has been read by now: a wave's LDS operations execute in order)
#pragma unroll
        for (int a = 0; a < NOUT; ++a) *reinterpret_cast<u4v*>(sL + (a * 64 + lane_p) * 16) = l3[a];
        const int mbase = p.m_begin + tile_m * TBM + wm * WM + lr_p;
        s_ncol0 = tile_n * TBN + wn * WN;
        s_coff0 = ((unsigned)mbase * (unsigned)p.ldc + (unsigned)(tile_n * (TBN / 2) + wn * (WN / 2))) * 2u + (unsigned)(lg_p & 1) * 32u + (unsigned)(lg_p >> 1) * 16u;
        pending = true;
    }
    if (pending)         // the block's last tile
        static_for<MFRAG>([&](auto S_) __attribute__((always_inline)) {
            if constexpr (decltype(S_)::value == RGROUPS) {
                int lane_d = lane;
                asm volatile("" : "+v"(lane_d));
#pragma unroll
                for (int a = 0; a < NOUT; ++a) s3[a] = *reinterpret_cast<const u4v*>(sL + (a * 64 + lane_d) * 16);
            }
            static_for<16>([&](auto J_) __attribute__((always_inline)) { slice_part(S_, J_); });
        });
#endif
}


// =============================================================================================================================
// The same kernel on v_mfma_f32_32x32x16_f16.  Why: tools/ubench_fill.hip (profiles/r05a_ubench_fill.txt) - with two waves per SIMD
// a 16x16x32 MFMA leaves TWO free issue slots (2 fillers 17.7 cycles per MFMA, 3 fillers 22.1, 4 fillers 26.0), a 32x32x16 MFMA
// four to six (4 fillers 32.9 cycles, 6 fillers 34.6, 8 fillers 42.5) at twice the flops: the deferred epilogue is ~3.5 VALU
// operations per 16x16x32 MFMA, which the first form pays for in full (measured: slower than the phased epilogue,
// profiles/r05b_geglu_deferred_ab.txt) and this one mostly hides.
//
// Layout differences.  Wave tile 64 x 128 = 2 x 4 accumulator blocks of 32 x 32 (16 registers each).  Weight block = A operand
// (its 32 rows are output columns), activation block = B operand (its 32 rows are output rows): lane l holds, of block (nb, mb),
// the ONE output row m = 32 mb + l % 32 and the sixteen columns 32 nb + 8 q + 4 (l / 32) + r (q, r = 0..3; register 4 q + r).
// Operand fragments: lane l reads the 16-byte chunk 2 j + l / 32 of row l % 32 for the K-slice j (16 deep) - the same K positions
// for both operands, which is all the contraction needs; the XOR-swizzled [rows][64] LDS image of the DMA serves these reads
// conflict-free as it is.  The packed GEGLU weights alternate 32 value / 32 gate rows, so block nb = 2 ob holds the values and
// nb = 2 ob + 1 the gates of output block ob in the same lane and register.  A slice of the saved tile is (mb, ob): 16 outputs per
// lane, four per K-slice j; v_permlane32_swap pairs the 8-byte pieces of lanes l and l + 32 into dwordx4 stores.
// =============================================================================================================================
[[maybe_unused]] constexpr int MB = WM / 32, NB = WN / 32, OB = NB / 2;      // 2 x 4 accumulator blocks; 2 output blocks of 32 columns
[[maybe_unused]] constexpr int NSL = MB * OB;                                // slices of the saved tile (= 4)

__global__ void __launch_bounds__(THREADS, 2) gemm_geglu_deferred32_kernel(GemmArgs p, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NSL == 4 && NSL - 1 == RGROUPS, "slices 0..2 in registers, slice 3 in LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sX = reinterpret_cast<half_t*>(smem_raw);
    half_t* sW = sX + 2 * TBM * BK;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);

    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int chunk = tid & 7;
    const int r0 = tid >> 3;
    const int wm = wave % NWM, wn = wave / NWM;
    unsigned char* sL = smem_raw + STAGES + wave * WAVE_LDS;
    float* sB = reinterpret_cast<float*>(sL);

    unsigned xoff0 = 0, woff0 = 0;
    int lm0 = 0, ln0 = 0;
    const unsigned xstep = (unsigned)RSTEP * (unsigned)p.lda * 2u, wstep = (unsigned)RSTEP * (unsigned)p.ldw * 2u;
    const unsigned csrc = (unsigned)(chunk ^ ((r0 >> 1) & 7)) * 16u;
    auto init_load = [&](int t) {
        int tile_m, tile_n;
        tile_coords(t, ntiles, p.tiles_n, tile_m, tile_n);
        lm0 = p.m_begin + tile_m * TBM + r0;
        ln0 = tile_n * TBN + r0;
        xoff0 = (unsigned)lm0 * (unsigned)p.lda * 2u + csrc;
        woff0 = (unsigned)ln0 * (unsigned)p.ldw * 2u + csrc;
    };
    auto load_tile = [&](int kt, int buf) {
        half_t* dx = sX + buf * TBM * BK + wave * 8 * BK;
        half_t* dw = sW + buf * TBN * BK + wave * 8 * BK;
        const unsigned soff = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const unsigned v = lm0 + RSTEP * i < p.M ? xoff0 + (unsigned)i * xstep : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + RSTEP * i * BK), 16, v, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const unsigned v = ln0 + RSTEP * i < p.N ? woff0 + (unsigned)i * wstep : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(dw + RSTEP * i * BK), 16, v, soff, 0, 0);
        }
    };

    f16v acc[NB][MB];
    unsigned sx[RGROUPS][4][2], sg[RGROUPS][4][2];     // [slice][q][column pair]: packed fp16 values / gates of slices 0..2
    u4v s3[4];                                         // slice 3 per q (x01 x23 g01 g23): in LDS between the pack and its K-step
    unsigned s_coff0 = 0;                              // byte offset of (the lane's row in row block 0, its 8-byte column piece of output block 0)
    int s_ncol0 = 0;
    const unsigned cstep = 64u * (unsigned)p.ldc;      // 32 rows

    float ge[4];
    unsigned ow[2][2];
    // element (q, r) = J of slice S: GELU of its gate; packed multiplies and stores as soon as their inputs are complete
    auto slice_part = [&](auto S_, auto J_) {
        constexpr int S = decltype(S_)::value, J = decltype(J_)::value;
        constexpr int q = J >> 2, r = J & 3;
        unsigned gw, xw;
        if constexpr (S < RGROUPS) { gw = sg[S][q][r >> 1]; xw = sx[S][q][r >> 1]; }
        else { gw = s3[q][2 + (r >> 1)]; xw = s3[q][r >> 1]; }
        const h2v gp = __builtin_bit_cast(h2v, gw);
        ge[r] = gelu_erf((float)gp[r & 1]);
        asm volatile("" : "+v"(ge[r]));      // evaluated HERE, among this group's MFMAs (the optimiser would sink it to its use, one group on)
        if constexpr (r == 1 || r == 3) {
            const h2v e = {(half_t)ge[r - 1], (half_t)ge[r]};
            const h2v xv = __builtin_bit_cast(h2v, xw);
            ow[q & 1][r >> 1] = __builtin_bit_cast(unsigned, xv * e);
        }
        if constexpr (r == 3 && (q & 1)) {
            // column groups q - 1 (vdst) and q (src): lanes 0-31 end up with the 8 contiguous columns 16 (q / 2) .. + 7 of their row,
            // lanes 32-63 with .. + 8 .. + 15
            const auto s0 = __builtin_amdgcn_permlane32_swap(ow[0][0], ow[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(ow[0][1], ow[1][1], false, false);
            constexpr int mb = S / OB, ob = S % OB;
            // N is a multiple of 64 and the strip starts on a multiple of 128: an output block (64 packed columns) is inside N or not as a
            // whole.  The row-block step rides in the VGPR offset (the range check that drops rows >= M does not see soffset).
            const bool inside = s_ncol0 + 64 * ob + 64 <= p.N;
            const unsigned voff = inside ? s_coff0 + (unsigned)mb * cstep : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u4v{s0[0], s1[0], s0[1], s1[1]}, srd_c, voff, ob * 64 + (q >> 1) * 32, 0);
        }
    };

    const int nk = p.K / BK;
    int ltile = blockIdx.x, lkt = 0;
    int cur = 0;
    f4 breg = {0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](auto S_, auto FIRST_, bool last, int tile_n) {
        constexpr int S = decltype(S_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        if (++lkt == nk) {
            lkt = 0;
            ltile += G;
            if (ltile < ntiles) init_load(ltile);
        }
        if (ltile < ntiles) load_tile(lkt, cur ^ 1);
        if constexpr (S == RGROUPS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s3[q] = *reinterpret_cast<const u4v*>(sL + (q * 64 + lane_k) * 16);
        }
        if (last && (p.flags & VCX_GEMM_BIAS_N) && lane_k < WN / 4)
            breg = *reinterpret_cast<const f4*>(p.bias + min(tile_n * TBN + wn * WN + lane_k * 4, p.N - 4));
        __builtin_amdgcn_sched_barrier(0);
        const half_t* cx = sX + cur * TBM * BK;
        const half_t* cw = sW + cur * TBN * BK;
        const int lrow = lane_k & 31, hi = lane_k >> 5;
        const int sw = (lrow >> 1) & 7;             // the rows of a block start on a multiple of 32: the swizzle term is the lane's own
        auto frag = [&](const half_t* base, int row0, int j) {
            return *reinterpret_cast<const h8*>(base + (row0 + lrow) * BK + ((((2 * j) | hi) ^ sw) << 3));
        };
        {
            h8 xc[MB], xn[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) xc[mb] = frag(cx, wm * WM + mb * 32, 0);
            h8 wcur = frag(cw, wn * WN, 0);
            static_for<4 * NB>([&](auto J_) __attribute__((always_inline)) {
                constexpr int J = decltype(J_)::value, j = J / NB, nb = J % NB;
                h8 wnext = wcur;
                if constexpr (nb + 1 < NB) wnext = frag(cw, wn * WN + (nb + 1) * 32, j);
                else if constexpr (j < 3) wnext = frag(cw, wn * WN, j + 1);
                if constexpr (j < 3 && nb >= 1 && nb <= MB) xn[nb - 1] = frag(cx, wm * WM + (nb - 1) * 32, j + 1);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    if constexpr (FIRST && j == 0) {
                        const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur, xc[mb], zero, 0, 0, 0);
                    } else {
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur, xc[mb], acc[nb][mb], 0, 0, 0);
                    }
                }
                wcur = wnext;
                if constexpr (j < 3 && nb == NB - 1) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) xc[mb] = xn[mb];
                }
                if constexpr (S >= 0) {
                    slice_part(S_, J_);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int t = 0; t < MB; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (S >= 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    };

    init_load(ltile);
    load_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool pending = false;
    for (int ctile = blockIdx.x; ctile < ntiles; ctile += G) {
        int tile_m, tile_n;
        tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
        int k0 = 1;
        if (pending) {
            kstep(Int<0>{}, Int<1>{}, false, tile_n);
            kstep(Int<1>{}, Int<0>{}, false, tile_n);
            kstep(Int<2>{}, Int<0>{}, false, tile_n);
            kstep(Int<3>{}, Int<0>{}, false, tile_n);
            k0 = NSL;
        } else {
            kstep(Int<-1>{}, Int<1>{}, false, tile_n);
        }
        for (int k = k0; k < nk; ++k) kstep(Int<-1>{}, Int<0>{}, k == nk - 1, tile_n);
        // ---- pack: bias, fp16 rounding
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        const int lrow_p = lane_p & 31, hi_p = lane_p >> 5;
        if (lane_p < WN / 4) *reinterpret_cast<f4*>(sB + lane_p * 4) = breg;
        const float alpha = p.alpha;
        u4v l3[4];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 bx = *reinterpret_cast<const f4*>(sB + 64 * ob + 8 * q + 4 * hi_p);
                const f4 bg = *reinterpret_cast<const f4*>(sB + 64 * ob + 32 + 8 * q + 4 * hi_p);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    float x_[4], g_[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x_[r] = __builtin_fmaf(acc[2 * ob][mb][4 * q + r], alpha, bx[r]);
                        g_[r] = __builtin_fmaf(acc[2 * ob + 1][mb][4 * q + r], alpha, bg[r]);
                    }
                    const unsigned x01 = __builtin_bit_cast(unsigned, h2v{(half_t)x_[0], (half_t)x_[1]});
                    const unsigned x23 = __builtin_bit_cast(unsigned, h2v{(half_t)x_[2], (half_t)x_[3]});
                    const unsigned g01 = __builtin_bit_cast(unsigned, h2v{(half_t)g_[0], (half_t)g_[1]});
                    const unsigned g23 = __builtin_bit_cast(unsigned, h2v{(half_t)g_[2], (half_t)g_[3]});
                    const int sl = mb * OB + ob;
                    if (sl < RGROUPS) { sx[sl][q][0] = x01; sx[sl][q][1] = x23; sg[sl][q][0] = g01; sg[sl][q][1] = g23; }
                    else l3[q] = u4v{x01, x23, g01, g23};
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u4v*>(sL + (q * 64 + lane_p) * 16) = l3[q];
        const int mbase = p.m_begin + tile_m * TBM + wm * WM + lrow_p;
        s_ncol0 = tile_n * TBN + wn * WN;
        s_coff0 = ((unsigned)mbase * (unsigned)p.ldc + (unsigned)(tile_n * (TBN / 2) + wn * (WN / 2))) * 2u + (unsigned)hi_p * 16u;
        pending = true;
    }
    if (pending)
        static_for<NSL>([&](auto S_) __attribute__((always_inline)) {
            if constexpr (decltype(S_)::value == RGROUPS) {
                int lane_d = lane;
                asm volatile("" : "+v"(lane_d));
#pragma unroll
                for (int q = 0; q < 4; ++q) s3[q] = *reinterpret_cast<const u4v*>(sL + (q * 64 + lane_d) * 16);
            }
            static_for<16>([&](auto J_) __attribute__((always_inline)) { slice_part(S_, J_); });
        });
#endif
}

}  // namespace

// nk >= 5, linear, GEGLU, fp16 out, no LNFOLD / COLSTATS / ROWADD / RESIDUAL / BIAS_M: checked by the caller (gemm_dma.hip dispatch)
int vcxgemm::launch_geglu_deferred(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds16, lds32;
    const bool m16 = vcx_tune(VCX_TUNE_GEGLU_IMPL) == 3;      // the 16x16x32 form, kept for the A/B (tools/gemm_quick.py geglu3)
    auto kern = m16 ? gemm_geglu_deferred_kernel : gemm_geglu_deferred32_kernel;
    if (!(m16 ? lds16 : lds32).ensure(reinterpret_cast<const void*>(kern), (int)SMEM, "vcx_gemm_f16(geglu, deferred)")) return VCX_ELAUNCH;
    const int nb = persistent_grid(a.tiles_m * a.tiles_n, 1);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(THREADS), SMEM, s, a, a.a_bytes, a.w_bytes);
    return vcx_check_launch("vcx_gemm_f16(geglu, deferred)");
}
