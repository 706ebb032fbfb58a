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
#include "gemm_args.h"

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
    static constexpr size_t SMEM = STAGES + (size_t)TBN * NWM * sizeof(float);   // + one bias strip per wave (epilogue)
};

template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32>
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

    // issue the DMA of K-step kt of the load tile into LDS buffer `buf`
    auto load_tile = [&](int kt, int buf, int parts = 3) {       // parts: bit 0 = activation rows, bit 1 = weight rows
        half_t* dx = sX + buf * TBM * BK + wave * 8 * BK;
        half_t* dw = sW + buf * BN * BK + wave * 8 * BK;
        if (CONV && (parts & 1)) {
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
            } else {
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
    const int flags = p.flags;
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
        // Fragment reads are software-pipelined by hand: the next weight fragment is requested before the 8-16 MFMAs that
        // use the current one, and the activation fragments of the second K half are re-requested right after their last
        // use in the first half.  (Left to itself hipcc issues every ds_read immediately before the MFMA that needs it -
        // one exposed LDS round trip per MFMA group - and on gfx950 nothing else runs on the SIMD while it waits:
        // tools/ubench.hip shows MFMA and VALU/other issue of the two waves of a SIMD do not overlap.)
        {
            h8 xf[MFRAG];
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, lg));
            h8 wcur = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, lg));
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 1 && more && late_parts) load_tile(lkt, cur ^ 1, late_parts);
#pragma unroll
                for (int a = 0; a < NFRAG; ++a) {
                    h8 wnext = wcur;
                    if (a + 1 < NFRAG) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + (a + 1) * 16 + lr, kk * 4 + lg));
                    else if (kk == 0) wnext = *reinterpret_cast<const h8*>(cw + lds_off(wn * WN + lr, 4 + lg));
#pragma unroll
                    for (int b = 0; b < MFRAG; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur, xf[b], acc[a][b], 0, 0, 0);
                        if (a == NFRAG - 1 && kk == 0)
                            xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * WM + b * 16 + lr, 4 + lg));
                    }
                    wcur = wnext;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (ckt == nk - 1) {
            // ---- epilogue: acc[a][b][r] = out[m][n], m = tile_m*BM + wm*64 + b*16 + lr, n = tile_n*BN + wn*(BN/2) + a*16 + lg*4 + r
            const int mbase = p.m_begin + tile_m * TBM + wm * WM + lr;
            const int nbase = tile_n * BN + wn * WN + lg * 4;
            if (GEGLU) {
                // Packed GEGLU weights come in 64-column blocks [32 value | 32 gate] (packing.py): of a wave's fragments, 4j and
                // 4j + 1 are values, 4j + 2 and 4j + 3 their gates; output fragment a = 2j + i pairs xfrag(a) with xfrag(a) + 2.
                auto xfrag = [](int a) { return 4 * (a >> 1) + (a & 1); };
                f4 bx[NFRAG / 2 + 1], bg[NFRAG / 2 + 1];
#pragma unroll
                for (int a = 0; a < NFRAG / 2; ++a) {
                    const int nx = min(nbase + xfrag(a) * 16, p.N - 36);
                    bx[a] = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nx) : f4{0.f, 0.f, 0.f, 0.f};
                    bg[a] = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nx + 32) : f4{0.f, 0.f, 0.f, 0.f};
                }
                // output through a buffer descriptor (rows >= M dropped by the range check), fragment pairs widened to dwordx4
                // with v_permlane16_swap exactly as in the plain epilogue below
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
                constexpr int NOUT = NFRAG / 2;                       // output fragments per wave (value x gate pairs)
                const int jstrip = tile_n * (BN / 2) + wn * (WN / 2);    // first output column of the wave
                const unsigned coff0 = ((unsigned)mbase * (unsigned)p.ldc + (unsigned)jstrip) * 2u;
                const unsigned cstep = 32u * (unsigned)p.ldc;
                const unsigned odd = lg & 1, half = lg >> 1;
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) {
                    u2v packed[NOUT];
#pragma unroll
                    for (int a = 0; a < NOUT; ++a) {
                        half_t o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xv = acc[xfrag(a)][b][r] * p.alpha + bx[a][r];
                            const float gv = acc[xfrag(a) + 2][b][r] * p.alpha + bg[a][r];
                            o[r] = (half_t)(xv * gelu_erf(gv));
                        }
                        packed[a] = __builtin_bit_cast(u2v, h4{o[0], o[1], o[2], o[3]});
                    }
                    const unsigned crow = coff0 + (unsigned)b * cstep;
#pragma unroll
                    for (int a = 0; a < NOUT; a += 2) {
                        if (a + 1 < NOUT) {
                            const unsigned a0 = packed[a][0], a1 = packed[a][1], b0 = packed[a + 1][0], b1 = packed[a + 1][1];
                            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                            const int col = (a + (int)odd) * 16 + (int)half * 8;             // 8 output columns of this lane
                            const int nx = tile_n * BN + wn * WN + xfrag(a + (int)odd) * 16;  // packed-space column of the fragment's values
                            const unsigned voff = nx + 48 <= p.N ? crow + (unsigned)col * 2u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b128(u4v{s0[0], s1[0], s0[1], s1[1]}, srd_c, voff, 0, 0);
                        } else {
                            const int nx = tile_n * BN + wn * WN + xfrag(a) * 16;
                            const unsigned voff = nx + 48 <= p.N ? crow + (unsigned)(a * 16 + lg * 4) * 2u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b64(packed[a], srd_c, voff, 0, 0);
                        }
                    }
                }
            } else {
                // Output and residual are addressed through buffer descriptors: rows >= M fall outside the extent (stores
                // dropped, loads return 0), columns >= N get an out-of-range offset - no exec-mask branches, one 32-bit
                // VALU add per access.  Column-fragment outer / 16-row group inner keeps one bias vector (4 registers) live,
                // and the residual fetch runs RD accesses ahead of its use (C may alias R, so the compiler cannot hoist
                // loads above earlier stores by itself; issuing them early here hides the memory round trip).
                constexpr int ES = OUT_F32 ? 4 : 2;
                const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.R), 0, (int)p.r_bytes, 0x00020000);
                const unsigned cstep = 16u * (unsigned)p.ldc * ES, rstep = 32u * (unsigned)p.ldr;
                const bool has_res = flags & VCX_GEMM_RESIDUAL;
                // a row-indexed addend (time embedding) is one row for the whole tile except where a tile straddles two frames
                const int m_first = p.m_begin + tile_m * TBM;
                const int radd_row = m_first / p.rowadd_div;
                const bool radd_tile = (flags & VCX_GEMM_ROWADD) && (min(m_first + TBM, p.M) - 1) / p.rowadd_div == radd_row;
                const bool per_row = (flags & VCX_GEMM_BIAS_M) || ((flags & VCX_GEMM_ROWADD) && !radd_tile);
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                // Wide accesses.  Lane (lr, lg) holds 4 columns (8 bytes of fp16) of row lr per fragment, the lanes lg = 0..3 of a
                // row four adjacent such pieces.  v_permlane16_swap exchanges the odd 16-lane rows of one register with the even
                // rows of another: applied to the packed words of fragments a (vdst) and a+1 (src) it leaves the even-lg lanes with
                // 8 contiguous columns of fragment a and the odd-lg lanes with 8 contiguous columns of fragment a+1 - one dwordx4
                // store (and, run backwards, one dwordx4 residual fetch) per lane and fragment PAIR instead of a dwordx2 per
                // fragment.  The store tail of a tile is issue-bound (MI355X_MICROARCH.md / T21 of the HIP guide): fewer,
                // wider instructions shorten it.  An odd last fragment keeps the dwordx2 form.
                constexpr int NPAIRF = OUT_F32 ? 0 : NFRAG / 2;                 // fragment pairs handled wide (fp16 output only)
                constexpr int UNITS = NPAIRF + (NFRAG - 2 * NPAIRF);            // accesses per 16-row group
                constexpr int NUNIT = UNITS * MFRAG;
                constexpr int RDU = UNITS;                                       // residual prefetch distance: one 16-row group
                const unsigned odd = lg & 1, half = lg >> 1;
                // byte offset (within the row, relative to the wave's strip) of this lane's access for unit u
                auto unit_col = [&](int u) { return u < NPAIRF ? (2 * u + (int)odd) * 16 + (int)half * 8 : (2 * NPAIRF + (u - NPAIRF)) * 16 + lg * 4; };
                const int nstrip = tile_n * BN + wn * WN;
                const unsigned coff0 = ((unsigned)mbase * (unsigned)p.ldc + (unsigned)nstrip) * ES;
                const unsigned roff0 = ((unsigned)mbase * (unsigned)p.ldr + (unsigned)nstrip) * 2u;
                u4v rr[RDU];                 // residual ring, one entry per unit (a narrow unit uses the first two words)
                // unit i = b * UNITS + u: all units of one 16-row group back to back, so that every 128-byte line of C is
                // completed within a few consecutive stores (half-written lines that linger get evicted from L2; measured 1.5x
                // slower with the loops the other way round).  The residual fetch runs one row group ahead of its use: C may
                // alias R, so the compiler cannot hoist loads above earlier stores by itself.
                auto fetch = [&](int i) {
                    const int u = i % UNITS;
                    const unsigned o = roff0 + (unsigned)(i / UNITS) * rstep + (unsigned)unit_col(u) * 2u;
                    if (u < NPAIRF) rr[i % RDU] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, o, 0, 0);
                    else {
                        const u2v t = __builtin_amdgcn_raw_buffer_load_b64(srd_r, o, 0, 0);
                        const unsigned t0 = t[0], t1 = t[1];
                        rr[i % RDU] = u4v{t0, t1, 0u, 0u};
                    }
                };
                if (has_res) {
#pragma unroll
                    for (int i = 0; i < RDU; ++i) fetch(i);
                }
                // Column addends (bias, plus the tile's time-embedding row when it is uniform over the tile) live in a private
                // LDS strip of the wave, not in registers: a 160-column strip would pin 40 VGPRs through the whole epilogue.
                float* sB = reinterpret_cast<float*>(smem_raw + Cfg::STAGES) + wave * WN;
                if (lane < WN / 4) {
                    const int nc = min(nstrip + lane * 4, p.N - 4);
                    f4 t = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nc) : f4{0.f, 0.f, 0.f, 0.f};
                    if (radd_tile) {
                        const f4 rv = *reinterpret_cast<const f4*>(p.rowadd + (int64_t)radd_row * p.N + nc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[r] += rv[r];
                    }
                    *reinterpret_cast<f4*>(sB + lane * 4) = t;
                }
                int bopaque = 0;     // re-read per 16-row group (an address the compiler cannot prove loop-invariant)
                // value of accumulator fragment (a, b) with bias / addend applied (everything but the residual)
                auto finish = [&](int a, int b, float (&v)[4]) {
                    if (per_row) {      // rare: V^T projections (per-row bias) and tiles that straddle two addend rows
                        // same arithmetic as the tile-uniform case, (bias + addend) first and one fma: a row's result must
                        // not depend on how the batch happens to align tiles with frames (bit-exact batch invariance)
                        const int mc = min(mbase + b * 16, p.M - 1);
                        f4 t = *reinterpret_cast<const f4*>(sB + bopaque + a * 16 + lg * 4);
                        if (flags & VCX_GEMM_ROWADD) {
                            const f4 rv = *reinterpret_cast<const f4*>(p.rowadd + (int64_t)(mc / p.rowadd_div) * p.N + min(nbase + a * 16, p.N - 4));
#pragma unroll
                            for (int r = 0; r < 4; ++r) t[r] += rv[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[a][b][r], p.alpha, t[r]);
                        if (flags & VCX_GEMM_BIAS_M) {
                            const float bm = p.bias[mc];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += bm;
                        }
                    } else {
                        const f4 t = *reinterpret_cast<const f4*>(sB + bopaque + a * 16 + lg * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[a][b][r], p.alpha, t[r]);
                    }
                };
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) {
                    const unsigned crow = coff0 + (unsigned)b * cstep;
                    asm volatile("" : "+v"(bopaque));
#pragma unroll
                    for (int u = 0; u < UNITS; ++u) {
                        const int i = b * UNITS + u;
                        const bool wide = u < NPAIRF;
                        const int a = wide ? 2 * u : 2 * NPAIRF + (u - NPAIRF);
                        const unsigned voff = nstrip + unit_col(u) < p.N ? crow + (unsigned)unit_col(u) * ES : OOB;
                        float v0[4], v1[4] = {0.f, 0.f, 0.f, 0.f};
                        finish(a, b, v0);
                        if (wide) finish(a + 1, b, v1);
                        if (has_res) {
                            const u4v raw = rr[i % RDU];
                            if (i + RDU < NUNIT) fetch(i + RDU);
                            unsigned w0 = raw[0], w1 = raw[1], w2 = raw[2], w3 = raw[3];
                            if (wide) {
                                // the fetched 8 columns are [piece of the even lane | piece of the odd lane] of ONE fragment: undo
                                // the exchange so that each lane gets its own pieces of fragments a (w0, w1) and a + 1 (w2, w3)
                                const auto s0 = __builtin_amdgcn_permlane16_swap(w0, w2, false, false);
                                const auto s1 = __builtin_amdgcn_permlane16_swap(w1, w3, false, false);
                                w0 = s0[0]; w2 = s0[1]; w1 = s1[0]; w3 = s1[1];
                            }
                            const h4 r0 = __builtin_bit_cast(h4, u2v{w0, w1});
                            const h4 r1 = __builtin_bit_cast(h4, u2v{w2, w3});
#pragma unroll
                            for (int r = 0; r < 4; ++r) { v0[r] += (float)r0[r]; v1[r] += (float)r1[r]; }
                        }
                        if (OUT_F32) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, f4{v0[0], v0[1], v0[2], v0[3]}), srd_c, voff, 0, 0);
                        } else {
                            const h4 o0 = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3]};
                            const u2v p0 = __builtin_bit_cast(u2v, o0);
                            if (wide) {
                                const h4 o1 = {(half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                                const u2v p1 = __builtin_bit_cast(u2v, o1);
                                const unsigned a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
                                // vdst = fragment a, src = fragment a + 1: even lanes end up with [own a | odd lane's a],
                                // odd lanes with [even lane's a+1 | own a+1]
                                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                                __builtin_amdgcn_raw_buffer_store_b128(u4v{s0[0], s1[0], s0[1], s1[1]}, srd_c, voff, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b64(p0, srd_c, voff, 0, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < MFRAG; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        }
        // the DMA of the next K-step must have landed, and every wave must be done reading `cur`, before the roles swap
        __builtin_amdgcn_s_waitcnt(0x0f70 | 0);
        __syncthreads();
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

template <class Cfg, bool CONV, bool GEGLU, bool OUT_F32>
int launch(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    auto kern = gemm_dma_kernel<Cfg, CONV, GEGLU, OUT_F32>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::SMEM) != hipSuccess) {
            vcx_set_error("vcx_gemm_f16(dma): cannot reserve %zu bytes of LDS", Cfg::SMEM);
            return VCX_ELAUNCH;
        }
        attr_set = true;
    }
    const int blocks_per_cu = Cfg::SMEM > 80 * 1024 ? 1 : 2;
    const int nb = persistent_grid(a.tiles_m * a.tiles_n, blocks_per_cu);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::THREADS), Cfg::SMEM, s, a, a.a_bytes, a.w_bytes);
    return vcx_check_launch("vcx_gemm_f16(dma)");
}

template <class Cfg>
int dispatch(const GemmArgs& a, bool conv, bool geglu, bool f32, hipStream_t s) {
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
