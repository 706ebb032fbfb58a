// Epilogue shared by the GEMM kernels that use the 16x16x32 accumulator layout of gemm_dma.hip / gemm_pp.hip:
// acc[a][b][r] = out[m][n] with m = tile_m*TBM + wm*WM + b*16 + lr, n = tile_n*TBN + wn*WN + a*16 + lg*4 + r.
// bias / per-image addend / residual / GEGLU gate / fp32 output; stores and residual fetches through buffer descriptors,
// widened to dwordx4 per fragment pair with v_permlane16_swap (see the comments inside).
#pragma once
#include "gemm_args.h"

namespace vcxgemm {

[[maybe_unused]] constexpr unsigned EPI_OOB = 0xFFFFFFFFu;
typedef unsigned epi_u4v __attribute__((ext_vector_type(4)));

// Residual pieces of a whole wave tile, fetched AHEAD of the tile's main loop (fp16 output, plain epilogue: the wide / narrow access
// units of gemm_epilogue below, in its order i = b UNITS + u).  For a kernel that runs one wave per SIMD (gemm_ws.hip): a residual
// load inside its epilogue is waited for with the whole vector-memory queue - output stores and operand DMA included - and nothing
// else runs on the SIMD meanwhile; requested before the MFMAs, the pieces are there when the epilogue starts and the epilogue
// issues stores only.  `gemm_epilogue<..., RPRE = true>` takes the array instead of fetching.
template <class Cfg>
__device__ __forceinline__ void gemm_epilogue_fetch_residual(const GemmArgs& p, int tile_m, int tile_n, int wm, int wn, int lane,
                                                             epi_u4v (&out)[(Cfg::NF / 2 + Cfg::NF % 2) * Cfg::MF]) {
    constexpr int WM = Cfg::TBM / Cfg::NWM, WN = Cfg::TBN / Cfg::NWN, NPAIRF = Cfg::NF / 2, UNITS = NPAIRF + Cfg::NF % 2;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned odd = lg & 1, half = lg >> 1;
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.R), 0, (int)p.r_bytes, 0x00020000);
    const int mbase = p.m_begin + tile_m * Cfg::TBM + wm * WM + lr;
    const unsigned roff0 = ((unsigned)mbase * (unsigned)p.ldr + (unsigned)(tile_n * Cfg::TBN + wn * WN)) * 2u;
    const unsigned rstep = 32u * (unsigned)p.ldr;
#pragma unroll
    for (int b = 0; b < Cfg::MF; ++b)
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int col = u < NPAIRF ? (2 * u + (int)odd) * 16 + (int)half * 8 : (2 * NPAIRF + (u - NPAIRF)) * 16 + lg * 4;
            const unsigned o = roff0 + (unsigned)b * rstep + (unsigned)col * 2u;
            if (u < NPAIRF) out[b * UNITS + u] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, o, 0, 0);
            else {
                const u2v t = __builtin_amdgcn_raw_buffer_load_b64(srd_r, o, 0, 0);
                const unsigned t0 = t[0], t1 = t[1];
                out[b * UNITS + u] = epi_u4v{t0, t1, 0u, 0u};
            }
        }
}

// sB: the wave's private LDS strip of WN floats (column addends); sS: a second one, only allocated for LNF != 0.
// LNF (folded LayerNorm, include/vcx.h VCX_GEMM_LNFOLD*): the accumulator holds x W'^T of the UN-normalised rows;
//   LNF = 1  out = alpha rstd_m (acc - mean_m colsum_n) + bias'_n  = fma(acc, rb, fma(qb, colsum_n, bias'_n)),  rb = alpha rstd_m, qb = -rb mean_m
//   LNF = 2  out = alpha rstd_n (acc - mean_n colsum_m) + bias'_m  = fma(acc, cs_n, fma(cq_n, colsum_m, bias'_m)), cs / cq in the two strips
template <class Cfg, bool GEGLU, bool OUT_F32, int LNF = 0, bool RPRE = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f4 (&acc)[Cfg::NF][Cfg::MF], int tile_m, int tile_n, int wm, int wn,
                                              int lane, float* sB, [[maybe_unused]] float* sS = nullptr,
                                              [[maybe_unused]] const float* ln_r0 = nullptr, [[maybe_unused]] const float* ln_r1 = nullptr,
                                              [[maybe_unused]] const epi_u4v* rpre = nullptr,      // RPRE: gemm_epilogue_fetch_residual's pieces
                                              bool strip_ready = false) {      // the strip(s) still hold THIS column tile's addends (a caller that walks rows only)
    // ln_r0 / ln_r1 [MFRAG]: per-lane row terms fetched by the caller ahead of the last K-step - LNF 1: (mean, rstd) of row
    // mbase + 16 b, LNF 2: (colsum, bias') of it
    constexpr int TBM = Cfg::TBM, BN = Cfg::TBN, NFRAG = Cfg::NF, MFRAG = Cfg::MF;
    constexpr int WM = TBM / Cfg::NWM, WN = BN / Cfg::NWN;
    [[maybe_unused]] constexpr unsigned OOB = EPI_OOB;
    const int lr = lane & 15, lg = lane >> 4;
    const int flags = p.flags;
    // ---- epilogue: acc[a][b][r] = out[m][n], m = tile_m*BM + wm*64 + b*16 + lr, n = tile_n*BN + wn*(BN/2) + a*16 + lg*4 + r
    const int mbase = p.m_begin + tile_m * TBM + wm * WM + lr;
    const int nbase = tile_n * BN + wn * WN + lg * 4;
    if (GEGLU) {
        // Packed GEGLU weights come in 64-column blocks [32 value | 32 gate] (packing.py): of a wave's fragments, 4j and
        // 4j + 1 are values, 4j + 2 and 4j + 3 their gates; output fragment a = 2j + i pairs xfrag(a) with xfrag(a) + 2.
        auto xfrag = [](int a) { return 4 * (a >> 1) + (a & 1); };
        f4 bx[NFRAG / 2 + 1], bg[NFRAG / 2 + 1];
        [[maybe_unused]] f4 sx[LNF ? NFRAG / 2 + 1 : 1], sg[LNF ? NFRAG / 2 + 1 : 1];
#pragma unroll
        for (int a = 0; a < NFRAG / 2; ++a) {
            const int nx = min(nbase + xfrag(a) * 16, p.N - 36);
            bx[a] = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nx) : f4{0.f, 0.f, 0.f, 0.f};
            bg[a] = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nx + 32) : f4{0.f, 0.f, 0.f, 0.f};
            if (LNF) {
                sx[a] = *reinterpret_cast<const f4*>(p.ln_colsum + nx);
                sg[a] = *reinterpret_cast<const f4*>(p.ln_colsum + nx + 32);
            }
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
            [[maybe_unused]] float rb = 0.f, qb = 0.f;
            if (LNF) {
                rb = p.alpha * ln_r1[b];
                qb = -rb * ln_r0[b];
            }
#pragma unroll
            for (int a = 0; a < NOUT; ++a) {
                half_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xv, gv;
                    if (LNF) {
                        xv = __builtin_fmaf(acc[xfrag(a)][b][r], rb, __builtin_fmaf(qb, sx[a][r], bx[a][r]));
                        gv = __builtin_fmaf(acc[xfrag(a) + 2][b][r], rb, __builtin_fmaf(qb, sg[a][r], bg[a][r]));
                    } else {
                        xv = acc[xfrag(a)][b][r] * p.alpha + bx[a][r];
                        gv = acc[xfrag(a) + 2][b][r] * p.alpha + bg[a][r];
                    }
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
        constexpr int RDU = RPRE ? NUNIT : UNITS;                        // residual prefetch distance: one 16-row group (RPRE: the caller fetched every piece)
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
        if (RPRE) {
#pragma unroll
            for (int i = 0; i < RDU; ++i) rr[i] = rpre[i];
        } else if (has_res) {
#pragma unroll
            for (int i = 0; i < RDU; ++i) fetch(i);
        }
        // Column addends (bias, plus the tile's time-embedding row when it is uniform over the tile) live in a private
        // LDS strip of the wave, not in registers: a 160-column strip would pin 40 VGPRs through the whole epilogue.
        if (lane < WN / 4 && !strip_ready) {
            const int nc = min(nstrip + lane * 4, p.N - 4);
            if (LNF == 2) {         // per-column (alpha rstd_n, -alpha rstd_n mean_n)
                const f4 s01 = *reinterpret_cast<const f4*>(p.ln_stats + 2 * nc), s23 = *reinterpret_cast<const f4*>(p.ln_stats + 2 * nc + 4);
                // Both vectors are complete BEFORE the first store and stay live behind the second one: hipcc otherwise re-used the data
                // registers of the first ds_write_b128 for the second product in the very next instructions (v_pk_mul_f32 into
                // v[n:n+1] one and two slots behind the store) - a wide LDS store still reads its data then, and single strip
                // elements went out stale, run-dependent (found by the full GPU suite; tools/isa_audit.py now scans every kernel of
                // the library for a VALU write to the data registers of a >= 96-bit LDS / memory store within two slots).
                f4 cs = {p.alpha * s01[1], p.alpha * s01[3], p.alpha * s23[1], p.alpha * s23[3]};
                f4 cq = {-cs[0] * s01[0], -cs[1] * s01[2], -cs[2] * s23[0], -cs[3] * s23[2]};
                asm volatile("" : "+v"(cs), "+v"(cq));
                *reinterpret_cast<f4*>(sB + lane * 4) = cs;
                *reinterpret_cast<f4*>(sS + lane * 4) = cq;
                asm volatile("" : : "v"(cs), "v"(cq));
            } else {
                f4 t = (flags & VCX_GEMM_BIAS_N) ? *reinterpret_cast<const f4*>(p.bias + nc) : f4{0.f, 0.f, 0.f, 0.f};
                if (radd_tile) {
                    const f4 rv = *reinterpret_cast<const f4*>(p.rowadd + (int64_t)radd_row * p.rowadd_ld + nc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] += rv[r];
                }
                *reinterpret_cast<f4*>(sB + lane * 4) = t;
                if (LNF == 1) *reinterpret_cast<f4*>(sS + lane * 4) = *reinterpret_cast<const f4*>(p.ln_colsum + nc);
            }
        }
        // per-lane row terms of the folded LayerNorm: LNF 1 (rb, qb) of the lane's row in each 16-row group, LNF 2 (colsum_m, bias'_m)
        constexpr bool FOLD = LNF == 1 || LNF == 2;      // (LNF 3 = VCX_GEMM_COLSTATS, below)
        [[maybe_unused]] float ln_a[FOLD ? MFRAG : 1], ln_b[FOLD ? MFRAG : 1];
        if (FOLD) {
#pragma unroll
            for (int b = 0; b < MFRAG; ++b) {
                if (LNF == 1) {
                    ln_a[b] = p.alpha * ln_r1[b];
                    ln_b[b] = -ln_a[b] * ln_r0[b];
                } else {
                    ln_a[b] = ln_r0[b];
                    ln_b[b] = ln_r1[b];
                }
            }
        }
        // LNF 3 (VCX_GEMM_COLSTATS): moments of the tile's OUTPUT columns for the GroupNorm that consumes them - per lane and column the
        // sum and sum of squares of (value - shift) over the lane's four rows, the shift being the first value of the lane's
        // 4-column piece in the first row of the wave's strip (uniform over the strip's rows: read with a row-share DPP), so
        // that rows whose common offset dwarfs their spread keep their digits (test_groupnorm_large_common_offset).  The values
        // are the fp16-ROUNDED outputs, the numbers the apply pass will read.
        [[maybe_unused]] float gs[LNF == 3 ? UNITS : 1][8], gq[LNF == 3 ? UNITS : 1][8], gk[LNF == 3 ? UNITS : 1][2];
        if (LNF == 3) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) gs[u][e] = gq[u][e] = 0.f;
        }
        int bopaque = 0;     // re-read per 16-row group (an address the compiler cannot prove loop-invariant)
        // value of accumulator fragment (a, b) with bias / addend applied (everything but the residual)
        auto finish = [&](int a, int b, float (&v)[4]) {
            if (FOLD) {
                const f4 t = *reinterpret_cast<const f4*>(sB + bopaque + a * 16 + lg * 4);
                const f4 u = *reinterpret_cast<const f4*>(sS + bopaque + a * 16 + lg * 4);
                // The row terms enter the packed multiply-adds as explicit (x, x) pairs behind an optimisation barrier.  Left alone,
                // hipcc used the four row terms straight out of their registers - for odd b the HIGH half of an aligned pair as
                // the LOW operand, `v_pk_fma_f32 ... op_sel:[0,1,1]` - in one instantiation (128x128 tile, LNFOLD_T), and exactly
                // that kernel returned run-dependent low halves in lanes 48-63 of the b = 1 row group (every other form,
                // op_sel_hi:[1,0,0] included, is bit-reproducible).  tools/isa_audit.py --stores rejects the form library-wide.
                typedef float f2v __attribute__((ext_vector_type(2)));
                f2v la = {ln_a[b], ln_a[b]}, lb = {ln_b[b], ln_b[b]};
                asm volatile("" : "+v"(la), "+v"(lb));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = LNF == 1 ? __builtin_fmaf(acc[a][b][r], la[r & 1], __builtin_fmaf(lb[r & 1], u[r], t[r]))
                                    : __builtin_fmaf(acc[a][b][r], t[r], __builtin_fmaf(u[r], la[r & 1], lb[r & 1]));
            } else if (per_row) {      // rare: V^T projections (per-row bias) and tiles that straddle two addend rows
                // same arithmetic as the tile-uniform case, (bias + addend) first and one fma: a row's result must
                // not depend on how the batch happens to align tiles with frames (bit-exact batch invariance)
                const int mc = min(mbase + b * 16, p.M - 1);
                f4 t = *reinterpret_cast<const f4*>(sB + bopaque + a * 16 + lg * 4);
                if (flags & VCX_GEMM_ROWADD) {
                    const f4 rv = *reinterpret_cast<const f4*>(p.rowadd + (int64_t)(mc / p.rowadd_div) * p.rowadd_ld + min(nbase + a * 16, p.N - 4));
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
                // the fp16 values that are stored; the column moments below are taken of exactly these numbers
                [[maybe_unused]] const h4 o0 = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3]};
                [[maybe_unused]] const h4 o1 = {(half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                [[maybe_unused]] u2v pk0 = __builtin_bit_cast(u2v, o0), pk1 = __builtin_bit_cast(u2v, o1);
                if (LNF == 3) {
                    // read the halves back out of the PACKED words (opaque to the optimiser: otherwise it keeps a second, scalar
                    // f32 -> f16 conversion per element beside the packed one that feeds the store)
                    unsigned q0 = pk0[0], q1 = pk0[1], q2 = pk1[0], q3 = pk1[1];
                    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                    pk0 = u2v{q0, q1};
                    pk1 = u2v{q2, q3};
                    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                    const h2v a01 = __builtin_bit_cast(h2v, q0), a23 = __builtin_bit_cast(h2v, q1);
                    const h2v b01 = __builtin_bit_cast(h2v, q2), b23 = __builtin_bit_cast(h2v, q3);
                    const float w0[4] = {(float)a01[0], (float)a01[1], (float)a23[0], (float)a23[1]};
                    const float w1[4] = {(float)b01[0], (float)b01[1], (float)b23[0], (float)b23[1]};
                    if (b == 0) {       // lane lr = 0 of the DPP row holds the strip's first row: its value is everybody's shift
                        gk[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w0[0]), 0x150, 0xf, 0xf, true));
                        gk[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w1[0]), 0x150, 0xf, 0xf, true));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d0 = w0[r] - gk[u][0], d1 = w1[r] - gk[u][1];
                        gs[u][r] += d0;
                        gq[u][r] = __builtin_fmaf(d0, d0, gq[u][r]);
                        gs[u][4 + r] += d1;
                        gq[u][4 + r] = __builtin_fmaf(d1, d1, gq[u][4 + r]);
                    }
                }
                if (OUT_F32) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, f4{v0[0], v0[1], v0[2], v0[3]}), srd_c, voff, 0, 0);
                } else {
                    const u2v p0 = pk0;
                    if (wide) {
                        const u2v p1 = pk1;
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
        if (LNF == 3) {
            auto row_sum = [](float x) {      // sum over the 16 lanes of a DPP row (the 16 tile rows a fragment spans), same value in all of them
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
                return x;
            };
            // one (mean, M2) pair per 64-row strip and column: colstats[strip][ldcs][2], strip = first row of the wave's rows / 64
            const int strip = (p.m_begin + tile_m * TBM + wm * WM) >> 6;
            float* dst = p.colstats + ((size_t)strip * (size_t)p.ldcs + nstrip) * 2;
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                float mo[8], qo[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float s1 = row_sum(gs[u][e]), s2 = row_sum(gq[u][e]);
                    mo[e] = gk[u][e >> 2] + s1 * (1.0f / 64.0f);
                    qo[e] = __builtin_fmaf(-s1 * (1.0f / 64.0f), s1, s2);
                }
                if (lr == 0 && strip * 64 < p.M) {     // (a tile's last strips may lie beyond M: M is a multiple of 64, not of the tile height)
                    const bool wide = u < NPAIRF;
                    const int c0 = wide ? (2 * u) * 16 + lg * 4 : (2 * NPAIRF + (u - NPAIRF)) * 16 + lg * 4;
                    if (nstrip + c0 < p.N) {
                        *reinterpret_cast<f4*>(dst + c0 * 2) = f4{mo[0], qo[0], mo[1], qo[1]};
                        *reinterpret_cast<f4*>(dst + c0 * 2 + 4) = f4{mo[2], qo[2], mo[3], qo[3]};
                    }
                    if (wide && nstrip + c0 + 16 < p.N) {
                        *reinterpret_cast<f4*>(dst + (c0 + 16) * 2) = f4{mo[4], qo[4], mo[5], qo[5]};
                        *reinterpret_cast<f4*>(dst + (c0 + 16) * 2 + 4) = f4{mo[6], qo[6], mo[7], qo[7]};
                    }
                }
            }
        }
    }
}

}  // namespace vcxgemm
