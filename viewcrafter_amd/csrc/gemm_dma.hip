// GEMM / implicit-GEMM convolution, DMA variant: operand tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds`
// (no VGPR staging, no ds_write), addressed through buffer descriptors whose hardware range check supplies the zero
// fill: padded taps, rows beyond M and weight rows beyond N simply use an out-of-range offset.
//
// Why a second kernel: PMC on the register-staged kernel (gemm.hip; SQ_INSTS_VALU / SQ_INSTS_SALU / MFMA-busy, early round 1)
// showed ~5 VALU + 2 SALU instructions per MFMA — 64-bit address arithmetic, bounds predicates and exec-mask branches around
// every 16-byte load — and MFMA-busy of only ~28 %.  Here a K-step costs, per thread, 4 x (v_add + bit-extract + select) for the
// activation rows of a convolution (nothing at all for a linear layer: the K offset rides in the scalar soffset) and
// zero instructions for the weight rows.
//
// Same operand roles as gemm.hip (weight = MFMA A operand, activation = B operand, 64-deep K-steps, XOR-swizzled LDS,
// persistent XCD-aware tile walk, cross-tile software pipeline), tile shapes per TileCfg below; the LDS image is
// lane-linear per DMA instruction, so the swizzle is applied to the SOURCE chunk each lane fetches (both-sides rule).
//
// Eligibility (checked by vcx_gemm_f16, which falls back to gemm.hip otherwise): K % 64 == 0, N % 8 == 0 (N % 4 for the
// GEGLU and fp32 epilogues), operand / output extents < 4 GiB (32-bit buffer offsets), convolutions with cin % 64 == 0 (a
// K-step then lies inside one tap, so the tap is block-uniform); stride-2, (3,1,1) and fused nearest-2x taps included.
#include "gemm_epilogue.h"

using namespace vcxgemm;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFFFFu;   // voffset beyond any descriptor's num_records -> the load returns zeros

// Tile configuration: block tile TBM x TBN, NWM x NWN waves, each wave owns (TBM/NWM) x (TBN/NWN) outputs.
//   small : 128 x {128,160}, 2x2 waves (64 x {64,80} per wave), 2 blocks/CU      - few tiles / small M
//   large : 256 x {256,320}, 4x2 waves (64 x {128,160} per wave), 1 block/CU      - LDS bytes per MFMA drop from ~690 to
//           ~450 (DMA writes 230 -> 115-128, fragment reads 461 -> 333-384): the small tile is LDS-bandwidth bound
//           (profiles/r01_gemm_experiments.md: removing the DMA gives +25 %, removing barriers or DMA waits nothing).
//           4 (M) x 2 (N) rather than 2 x 4: a wave's output strip is 256-320 bytes of every row (whole 64-byte sectors,
//           all stores dwordx4) instead of 128-160; tools/ubench_store.hip: the store phase of a tile is 12-26 % shorter.
template <int TBM_, int TBN_, int NWM_, int NWN_>
struct TileCfg {
    static constexpr int TBM = TBM_, TBN = TBN_, NWM = NWM_, NWN = NWN_;
    static constexpr int THREADS = 64 * NWM * NWN;
    static constexpr int MF = TBM / NWM / 16;      // 16-row activation fragments per wave
    static constexpr int NF = TBN / NWN / 16;      // 16-col weight fragments per wave
    static constexpr int XROWS = TBM * 8 / THREADS;   // DMA instructions per thread for the activation tile
    static constexpr int WROWS = TBN * 8 / THREADS;
    static constexpr int RSTEP = THREADS / 8;         // tile rows covered by one DMA instruction of the block
    static constexpr size_t STAGES = (size_t)2 * (TBM + TBN) * BK * sizeof(half_t);
    static constexpr size_t STRIP = (size_t)TBN * NWM * sizeof(float);            // one strip of column addends per wave (epilogue)
    static constexpr size_t SMEM = STAGES + STRIP;
    static constexpr size_t SMEM_LNF = STAGES + 2 * STRIP;                        // + a second strip (folded-LayerNorm epilogue)
};

// LNF: 0 = plain epilogue, 1 = VCX_GEMM_LNFOLD, 2 = VCX_GEMM_LNFOLD_T (linear mode only), 3 = VCX_GEMM_COLSTATS (convolutions and, round 4, linear layers);
// separate instantiations, so the plain kernels keep their register allocation
// TAIL (round 6, convolutions only): the last (p.k2 + p.k3) / 64 K-steps of a tile read, for output row m, row m of p.A2 and then of p.A3
// - linear sources - instead of an input pixel: the 1x1 skip convolution of a ResBlock rides in the K loop of its second 3x3
// convolution (include/vcx.h tail_a0 / tail_a1).  Own instantiations: the kernels without a tail keep their listing.
template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32, int LNF = 0, bool TAIL = false>
__global__ void __launch_bounds__(Cfg::THREADS, 2) gemm_dma_kernel(GemmArgs p, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the body uses device-only types)
    constexpr int TBM = Cfg::TBM, BN = Cfg::TBN;
    constexpr int NFRAG = Cfg::NF, MFRAG = Cfg::MF;
    constexpr int WROWS = Cfg::WROWS, XROWS = Cfg::XROWS, RSTEP = Cfg::RSTEP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sX = reinterpret_cast<half_t*>(smem_raw);              // [2][TBM*BK]
    half_t* sW = sX + 2 * TBM * BK;                                 // [2][BN*BK]

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, (int)w_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t srd_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(TAIL ? p.A2 : p.A), 0, TAIL ? (int)p.a2_bytes : 0, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t srd_a3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(TAIL && p.A3 ? p.A3 : p.A), 0, TAIL ? (int)p.a3_bytes : 0, 0x00020000);
    [[maybe_unused]] const int nk_main = (p.K - (TAIL ? p.k2 + p.k3 : 0)) / BK, nk_a2 = TAIL ? p.k2 / BK : 0;      // K-steps of the gather / of the first linear source
    [[maybe_unused]] int lrow0 = 0;                            // first output row of the tile being loaded

    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int chunk = tid & 7;   // LDS chunk position inside the 128-byte row
    const int r0 = tid >> 3;     // tile row of this thread's first DMA instruction

    // ---- per-thread source offsets (bytes) of the tile being loaded
    unsigned xoff[XROWS];        // activation rows; OOB when the row is beyond M (linear mode)
    unsigned xmask[XROWS];       // conv: bit t set <=> tap t of this row is inside the image
    unsigned woff[WROWS];        // weight rows; OOB when beyond N
    int tap = 0, ci0 = 0;        // conv K walker (block-uniform): k = tap*cin + ci0
    unsigned tap_off = 0;        // conv: byte offset of (tap, ci0) relative to the row's (ky=0, kx=0, c=0) pixel
    int tky = 0, tkx = 0;
    auto init_load = [&](int t) {
        int tile_m, tile_n;
        tile_coords(t, ntiles, p.tiles_n, tile_m, tile_n);
        if (TAIL) lrow0 = p.m_begin + tile_m * TBM;
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int r = r0 + RSTEP * i;
            const int m = p.m_begin + tile_m * TBM + r;
            const unsigned csrc = (unsigned)(chunk ^ ((r >> 1) & 7)) * 16u;   // source chunk that lands at position `chunk`
            if (CONV) {
                const int hw = p.out_h * p.out_w;
                const int mm = m < p.M ? m : 0;
                const int img = mm / hw;
                const int rem = mm - img * hw;
                const int oy = rem / p.out_w;
                const int ox = rem - oy * p.out_w;
                // (iy0, ix0): tap (0,0) in the (possibly 2x nearest-upsampled) input grid; its source pixel is (iy0>>ups, ix0>>ups)
                const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
                const long long pix0 = ((long long)img * p.in_h + (iy0 >> p.ups)) * p.in_w + (ix0 >> p.ups);   // may be negative at the border
                xoff[i] = (unsigned)(pix0 * p.lda * 2) + csrc;                            // wraps; valid taps un-wrap it
                unsigned mask = 0;
                if (m < p.M) {
                    const int lim_h = p.in_h << p.ups, lim_w = p.in_w << p.ups;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = iy0 + ky, ix = ix0 + kx;
                            if (iy >= 0 && iy < lim_h && ix >= 0 && ix < lim_w) mask |= 1u << (ky * p.kw + kx);
                        }
                }
                // bits 30/31: parity of (iy0, ix0) - with fused upsampling the source step of a tap depends on it
                xmask[i] = mask | ((unsigned)(iy0 & 1) << 31) | ((unsigned)(ix0 & 1) << 30);
            } else {
                xoff[i] = m < p.M ? (unsigned)((long long)m * p.lda * 2) + csrc : OOB;
                xmask[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int r = r0 + RSTEP * i;
            const int n = tile_n * BN + r;
            const unsigned csrc = (unsigned)(chunk ^ ((r >> 1) & 7)) * 16u;
            woff[i] = n < p.N ? (unsigned)((long long)n * p.ldw * 2) + csrc : OOB;
        }
        tap = 0; ci0 = 0; tap_off = 0; tky = 0; tkx = 0;
    };

    [[maybe_unused]] bool primed = false;
    // issue the DMA of K-step kt of the load tile into LDS buffer `buf`
    auto load_tile = [&](int kt, int buf, int parts = 3) {       // parts: bit 0 = activation rows, bit 1 = weight rows
        if (VCX_DMA_ABL & 3) {                                    // (timing only, vcx_ablate.h: no activation / no weight DMA behind the kernel's first K-step)
            if (primed) parts &= ~(VCX_DMA_ABL & 3);
            primed = true;
        }
        half_t* dx = sX + buf * TBM * BK + wave * 8 * BK;
        half_t* dw = sW + buf * BN * BK + wave * 8 * BK;
        if (TAIL && CONV && (parts & 1) && kt >= nk_main) {
            // K tail: row m of a linear source (the descriptor's range check drops rows >= M and anything beyond the source); the K
            // walker of the gather rests - it is reset by the next tile's init_load
            const bool first = kt - nk_main < nk_a2;
            const __amdgpu_buffer_rsrc_t src = first ? srd_a2 : srd_a3;
            const long long ld2 = (first ? p.lda2 : p.lda3) * 2;
            const unsigned soff = (unsigned)(first ? kt - nk_main : kt - nk_main - nk_a2) * (BK * 2);
#pragma unroll
            for (int i = 0; i < XROWS; ++i) {
                const int r = r0 + RSTEP * i;
                const int m = lrow0 + r;
                const unsigned v = m < p.M ? (unsigned)((long long)m * ld2) + (unsigned)(chunk ^ ((r >> 1) & 7)) * 16u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_ptr_t)(dx + RSTEP * i * BK), 16, v, soff, 0, 0);
            }
        } else if (CONV && (parts & 1)) {
            if (p.ups) {
                // source offset of tap (ky,kx) relative to tap (0,0): ((by+ky)>>1, (bx+kx)>>1) pixels, by/bx = parity of iy0/ix0
                const unsigned cb = (unsigned)ci0 * 2u, rowb = (unsigned)(p.in_w * (int)p.lda * 2), pixb = (unsigned)((int)p.lda * 2);
                const unsigned y0 = (unsigned)(tky >> 1) * rowb, y1 = (unsigned)((tky + 1) >> 1) * rowb;
                const unsigned x0 = (unsigned)(tkx >> 1) * pixb, x1 = (unsigned)((tkx + 1) >> 1) * pixb;
#pragma unroll
                for (int i = 0; i < XROWS; ++i) {
                    const unsigned ok = (xmask[i] >> tap) & 1u;
                    const unsigned oy_ = (xmask[i] >> 31) ? y1 : y0, ox_ = ((xmask[i] >> 30) & 1u) ? x1 : x0;
                    const unsigned v = ok ? xoff[i] + oy_ + ox_ + cb : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + RSTEP * i * BK), 16, v, 0, 0, 0);
                }
            } else if (!(VCX_CONV_XSKIP_ABL && p.kw == 3 && tkx != 0)) {      // (timing-only ablation: the activation rows of 2 of 3 horizontal taps are not fetched)
#pragma unroll
                for (int i = 0; i < XROWS; ++i) {
                    const unsigned ok = (xmask[i] >> tap) & 1u;
                    const unsigned v = ok ? xoff[i] + tap_off : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + RSTEP * i * BK), 16, v, 0, 0, 0);
                }
            }
            if (p.flags & VCX_GEMM_CONV_SLABK) {         // advance the (block-uniform) K walker: taps inside a 64-channel slab
                ++tap;
                if (++tkx == p.kw) { tkx = 0; ++tky; }
                if (tap == p.kh * p.kw) { tap = 0; tkx = 0; tky = 0; ci0 += BK; }
                tap_off = (unsigned)((tky * p.in_w + tkx) * (int)p.lda * 2 + ci0 * 2);
            } else {                                     // channel slabs inside a tap
                ci0 += BK;
                tap_off += BK * 2;
                if (ci0 == p.cin) {
                    ci0 = 0;
                    ++tap;
                    if (++tkx == p.kw) { tkx = 0; ++tky; }
                    tap_off = (unsigned)((tky * p.in_w + tkx) * (int)p.lda * 2);
                }
            }
        } else if (!CONV && (parts & 1)) {
            const unsigned soff = (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < XROWS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_ptr_t)(dx + RSTEP * i * BK), 16, xoff[i], soff, 0, 0);
        }
        if (parts & 2) {
            const unsigned soffw = (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < WROWS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)(dw + RSTEP * i * BK), 16, woff[i], soffw, 0, 0);
        }
    };

    const int wm = wave % Cfg::NWM, wn = wave / Cfg::NWM;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int WM = TBM / Cfg::NWM, WN = BN / Cfg::NWN;   // wave tile

    f4 acc[NFRAG][MFRAG];
#pragma unroll
    for (int a = 0; a < NFRAG; ++a)
#pragma unroll
        for (int b = 0; b < MFRAG; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    int ltile = blockIdx.x, lkt = 0;
    int ctile = blockIdx.x, ckt = 0;
    int tile_m, tile_n;
    tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
    init_load(ltile);
    load_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70 | 0);   // vmcnt(0) (lgkmcnt/expcnt untouched): first tile landed in LDS
    __syncthreads();
    int cur = 0;
    for (;;) {
        if (++lkt == nk) {
            lkt = 0;
            ltile += G;
            if (ltile < ntiles) init_load(ltile);
        }
        const bool more = ltile < ntiles;
        // the two waves of a SIMD (wave w and w + 4) issue their DMA at different points of the K-step, so that one is in its
        // MFMA stream while the other sits in the (60-180 cycle per instruction) DMA issue
        // Convolutions only: their operands come from L2 / the Infinity Cache and land within half a K-step (-3..5 % time).  A
        // linear layer's activation rows come from HBM and need the whole K-step; issuing even just its weight slice late costs
        // 10-30 % (measured), so linear layers keep both waves early.
        const int late_parts = (CONV && Cfg::THREADS == 512 && wave >= 4) ? 3 : 0;
        if (more) load_tile(lkt, cur ^ 1, 3 & ~late_parts);        // async: lands in the other buffer while this one is consumed
        const half_t* cx = sX + cur * TBM * BK;
        const half_t* cw = sW + cur * BN * BK;
        // folded LayerNorm: the lane's per-row terms (LNF 1: mean, rstd of its 4 rows; LNF 2: colsum, bias' of them) are fetched
        // ahead of the tile's last K-step, so that the epilogue does not start with an exposed global-memory round trip
        [[maybe_unused]] float ln_r0[LNF ? MFRAG : 1], ln_r1[LNF ? MFRAG : 1];
        if ((LNF == 1 || LNF == 2) && ckt == nk - 1) {
            const int mrow = p.m_begin + tile_m * TBM + wm * WM + lr;
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) {
                const int mc = min(mrow + b * 16, p.M - 1);
                if (LNF == 1) {
                    const float2 st = reinterpret_cast<const float2*>(p.ln_stats)[mc];
                    ln_r0[b] = st.x;
                    ln_r1[b] = st.y;
                } else {
                    ln_r0[b] = p.ln_colsum[mc];
                    ln_r1[b] = (p.flags & VCX_GEMM_BIAS_M) ? p.bias[mc] : 0.f;
                }
            }
        }
        // Fragment reads are software-pipelined by hand: the next weight fragment is requested before the 8-16 MFMAs that
        // use the current one, and the activation fragments of the second K half are re-requested right after their last
        // use in the first half.  (Left to itself hipcc issues every ds_read immediately before the MFMA that needs it -
        // one exposed LDS round trip per MFMA group - and on gfx950 nothing else runs on the SIMD while it waits:
        // tools/ubench.hip shows MFMA and VALU/other issue of the two waves of a SIMD do not overlap.)
        {
            constexpr bool NOREAD = (VCX_DMA_ABL & 8) != 0;       // (timing only: the fragments of the kernel's first K-step stay in their registers)
            h8 xf[MFRAG];
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, lg));
            h8 wcur = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, lg));
            if (NOREAD) {
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) asm volatile("" : "+v"(xf[b]));
                asm volatile("" : "+v"(wcur));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 1 && more && late_parts) load_tile(lkt, cur ^ 1, late_parts);
#pragma unroll
                for (int a = 0; a < NFRAG; ++a) {
                    h8 wnext = wcur;
                    if (NOREAD) asm volatile("" : "+v"(wnext));
                    else if (a + 1 < NFRAG) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + (a + 1) * 16 + lr, kk * 4 + lg));
                    else if (kk == 0) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, 4 + lg));
#pragma unroll
                    for (int b = 0; b < MFRAG; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur, xf[b], acc[a][b], 0, 0, 0);
                        if (a == NFRAG - 1 && kk == 0 && !NOREAD)
                            xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, 4 + lg));
                    }
                    wcur = wnext;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (ckt == nk - 1) {
            float* sB = reinterpret_cast<float*>(smem_raw + Cfg::STAGES) + wave * WN;   // the wave's private strip of column addends
            gemm_epilogue<Cfg, GEGLU, OUT_F32, LNF>(p, acc, tile_m, tile_n, wm, wn, lane, sB,
                                                    reinterpret_cast<float*>(smem_raw + Cfg::STAGES + Cfg::STRIP) + wave * WN, ln_r0, ln_r1);
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        }
        // the DMA of the next K-step must have landed, and every wave must be done reading `cur`, before the roles swap
        if (!(VCX_DMA_ABL & 16)) {                                // (timing only: no wait, no barrier)
            __builtin_amdgcn_s_waitcnt(0x0f70 | 0);
            __syncthreads();
        }
        cur ^= 1;
        if (++ckt == nk) {
            ckt = 0;
            ctile += G;
            if (ctile >= ntiles) break;
            tile_coords(ctile, ntiles, p.tiles_n, tile_m, tile_n);
        }
    }
#endif
}

template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32, int LNF = 0, bool TAIL = false>
int launch(const GemmArgs& a, hipStream_t s) {
    static VcxLdsAttr lds;
    auto kern = gemm_dma_kernel<Cfg, CONV, GEGLU, OUT_F32, LNF, TAIL>;
    constexpr size_t smem = LNF ? Cfg::SMEM_LNF : Cfg::SMEM;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)smem, "vcx_gemm_f16(dma)")) return VCX_ELAUNCH;
    const int blocks_per_cu = Cfg::SMEM > 80 * 1024 ? 1 : 2;
    const int nb = persistent_grid(a.tiles_m * a.tiles_n, blocks_per_cu);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::THREADS), smem, s, a, a.a_bytes, a.w_bytes);
    return vcx_check_launch("vcx_gemm_f16(dma)");
}

template <class Cfg>
int dispatch(const GemmArgs& a, bool conv, bool geglu, bool f32, hipStream_t s) {
    if (a.k2 + a.k3 > 0)       // K tail: convolutions with fp16 output, plain or column-moment epilogue (checked by vcx_gemm_f16)
        return (a.flags & VCX_GEMM_COLSTATS) ? launch<Cfg, true, false, false, 3, true>(a, s) : launch<Cfg, true, false, false, 0, true>(a, s);
    if (a.flags & (VCX_GEMM_LNFOLD | VCX_GEMM_LNFOLD_T)) {      // linear, fp16 output (checked by vcx_gemm_f16)
        if (a.flags & VCX_GEMM_LNFOLD_T) return launch<Cfg, false, false, false, 2>(a, s);
        if (!geglu) return launch<Cfg, false, false, false, 1>(a, s);
        if constexpr (Cfg::NF % 4 == 0) return launch<Cfg, false, true, false, 1>(a, s);
        vcx_set_error("vcx_gemm_f16(dma): GEGLU needs whole 64-column packed blocks per wave");
        return VCX_EINVAL;
    }
    if (a.flags & VCX_GEMM_COLSTATS)      // fp16 output, no GEGLU / LNFOLD (checked by vcx_gemm_f16)
        return conv ? launch<Cfg, true, false, false, 3>(a, s) : launch<Cfg, false, false, false, 3>(a, s);
    if (geglu) {
        if constexpr (Cfg::NF % 4 == 0) return conv ? launch<Cfg, true, true, false>(a, s) : launch<Cfg, false, true, false>(a, s);
        vcx_set_error("vcx_gemm_f16(dma): GEGLU needs whole 64-column packed blocks per wave");
        return VCX_EINVAL;
    }
    if (f32) return conv ? launch<Cfg, true, false, true>(a, s) : launch<Cfg, false, false, true>(a, s);
    return conv ? launch<Cfg, true, false, false>(a, s) : launch<Cfg, false, false, false>(a, s);
}

}  // namespace

int vcxgemm::launch_dma(GemmArgs& a, int cfg, bool conv, bool geglu, bool f32, hipStream_t s) {
    switch (cfg) {
        case 0: return dispatch<TileCfg<128, 128, 2, 2>>(a, conv, geglu, f32, s);
        case 1: return dispatch<TileCfg<128, 160, 2, 2>>(a, conv, geglu, f32, s);
        case 2: return dispatch<TileCfg<256, 256, 4, 2>>(a, conv, geglu, f32, s);
        case 3: return dispatch<TileCfg<256, 320, 4, 2>>(a, conv, geglu, f32, s);
    }
    vcx_set_error("vcx_gemm_f16(dma): unknown tile configuration %d", cfg);
    return VCX_EINVAL;
}
