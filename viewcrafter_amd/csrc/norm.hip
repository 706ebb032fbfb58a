// GroupNorm (+SiLU) and LayerNorm on channels-last fp16 with fp32 statistics (HBM-bound).
#include "vcx_common.h"

namespace {

constexpr int MAXC = 4096;  // C/8 * pixel lanes must fit one block (<= 512 threads)

// ---------------------------------------------------------------------------------------
// GroupNorm.  x [n_outer][pixels][C].  grid = (chunks, n_outer); a block owns a contiguous pixel range.  Both kernels use
// the same thread map: thread t owns the 16-byte channel chunk (t % CW) and pixel lane (t / CW) with CW = C/8 and
// blockDim = CW * PL, so a wave reads whole pixel rows back to back, the per-channel constants (8 partial sums /
// 8 scale+shift pairs) live in registers, and the pixel loop is 4-way unrolled to keep four 16-byte loads in flight.
// ---------------------------------------------------------------------------------------
// Chan/Welford merge of two (count, mean, M2) triples in a fixed order: b is folded into a.
__device__ __forceinline__ void gn_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    // branch-free: nb = 0 gives f = 0 (a unchanged), na = 0 gives f = 1 (a := b); empty + empty stays (0, m, 0)
    const float n = na + nb, d = mb - ma;
    const float f = nb * __builtin_amdgcn_rcpf(fmaxf(n, 1.0f));    // 1-ulp reciprocal: a weight, not a sum
    ma = ma + d * f;
    qa = qa + qb + d * d * na * f;
    na = n;
}

// Cs < C (gridDim.z = C / Cs slices of whole groups, Cs % 8 == 0): a block sums the channels [blockIdx.z Cs, + Cs) only.  `stats` != nullptr
// (gridDim.x == 1: the block sees every pixel of its n): the block's moments ARE the statistics - (mean, biased variance) go straight
// to stats[n][group] and no finalize launch follows (small pixel counts: the 9 x 16-pixel level of the UNet).
__global__ void gn_stats_kernel(const half_t* __restrict__ x, float* __restrict__ partials, float* __restrict__ stats, int64_t pixels, int C,
                                int groups, int Cs, int64_t pix_per_block, int cw, int pl) {
    // Numerically robust and deterministic (no atomics).  A thread sums x - K_t and (x - K_t)^2 with K_t = the first value it
    // loads (a sample of the data: |mean| >> std does not cancel - torch's GroupNorm is Welford too - and no load has to be
    // waited for before the streaming loop starts).  After the loop the sums are re-based algebraically to a shift that is
    // uniform per (block, group) - the group's first channel at the block's first pixel, handed round through LDS - so that
    // the partial sums of the pixel lanes and of a group's channels simply add, in a fixed order: per-thread channel sums ->
    // LDS -> pixel lanes -> channels of the group -> (count, mean, M2) of the block in partials[n][chunk][group][3];
    // gn_finalize_kernel merges the chunks Chan-style in order.
    extern __shared__ float red[];                      // [pl][C][2] then reused as [C][2]; + [C] block shifts
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < pixels) ? p0 + pix_per_block : pixels;
    const int cpg = C / groups;
    const int c8 = tid % cw, plane = tid / cw;
    const half_t* xp = x + ((int64_t)n * pixels) * C + (int64_t)blockIdx.z * Cs + c8 * 8;
    const int gslice = Cs / cpg;                         // groups per block
    float* kshare = red + (size_t)pl * Cs * 2;          // x[p0][c] for every channel c of the slice (written by pixel lane 0)
    float s[8], ss[8], K[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; K[e] = 0.f; }
    int64_t pix = p0 + plane;
    const int64_t step = pl;
    const int64_t mine = pix < p1 ? (p1 - pix + step - 1) / step : 0;       // pixels this thread owns
    if (mine > 0) {                                      // the same line the loop fetches first: both loads fly together
        const h8 v0 = *reinterpret_cast<const h8*>(xp + pix * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) K[e] = (float)v0[e];
    }
    for (; pix + 3 * step < p1; pix += 4 * step) {
        h8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const h8*>(xp + (pix + u * step) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[u][e] - K[e];
                s[e] += f;
                ss[e] += f * f;
            }
    }
    for (; pix < p1; pix += step) {
        const h8 v = *reinterpret_cast<const h8*>(xp + pix * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e] - K[e];
            s[e] += f;
            ss[e] += f * f;
        }
    }
    if (plane == 0) {                                    // pixel lane 0 always owns pixel p0: K_t = x[p0][c]
#pragma unroll
        for (int e = 0; e < 8; ++e) kshare[c8 * 8 + e] = K[e];
    }
    __syncthreads();
    const float cnt_t = (float)mine;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // re-base from K_t to the group's block shift K_b: sum(x - K_b) = S + n d, sum((x - K_b)^2) = Q + 2 d S + n d^2, d = K_t - K_b
        const float kb = kshare[((c8 * 8 + e) / cpg) * cpg];
        const float d = K[e] - kb;
        red[((size_t)plane * Cs + c8 * 8 + e) * 2 + 0] = s[e] + cnt_t * d;
        red[((size_t)plane * Cs + c8 * 8 + e) * 2 + 1] = ss[e] + d * (2.f * s[e] + cnt_t * d);
    }
    __syncthreads();
    for (int c = tid; c < Cs; c += blockDim.x) {         // pixel lanes, in order
        float a = red[(size_t)c * 2], q = red[(size_t)c * 2 + 1];
        for (int p = 1; p < pl; ++p) {
            a += red[((size_t)p * Cs + c) * 2];
            q += red[((size_t)p * Cs + c) * 2 + 1];
        }
        red[(size_t)c * 2] = a;
        red[(size_t)c * 2 + 1] = q;
    }
    __syncthreads();
    if (tid < gslice) {                                   // channels of the group, in order
        float a = 0.f, q = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
            a += red[(size_t)c * 2];
            q += red[(size_t)c * 2 + 1];
        }
        const float cnt = (float)((p1 - p0) * cpg);
        const float m = a / cnt;
        const float m2 = q - a * m;
        const int g = blockIdx.z * gslice + tid;
        const float mean = kshare[tid * cpg] + m, m2c = m2 > 0.f ? m2 : 0.f;
        if (stats) {
            stats[((int64_t)n * groups + g) * 2 + 0] = mean;
            stats[((int64_t)n * groups + g) * 2 + 1] = m2c / cnt;
        } else {
            float* dst = partials + (((int64_t)n * gridDim.x + blockIdx.x) * groups + g) * 3;
            dst[0] = cnt;
            dst[1] = mean;
            dst[2] = m2c;
        }
    }
}

__global__ void __launch_bounds__(64) gn_finalize_kernel(const float* __restrict__ partials, float* __restrict__ stats, int chunks,
                                                         int groups) {
    // one wave per (n, group): lane l folds the chunks l, l + 64, l + 128, ... in order (loads issued up front, Chan merge,
    // branch-free), then the 64 lane results are folded in a fixed butterfly.  stats[n][g] = (mean, biased variance).
    const int n = blockIdx.x / groups, g = blockIdx.x % groups, lane = threadIdx.x;
    const float* src = partials + ((int64_t)n * chunks * groups + g) * 3;
    float na = 0.f, a = 0.f, m2 = 0.f;
    for (int c0 = lane; c0 < chunks; c0 += 64 * 4) {
        float t[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 64 * u;
            const float* q = src + (int64_t)(c < chunks ? c : 0) * groups * 3;
            t[u][0] = c < chunks ? q[0] : 0.f;
            t[u][1] = q[1];
            t[u][2] = c < chunks ? q[2] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) gn_merge(na, a, m2, t[u][0], t[u][1], t[u][2]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        // lane pairs (l, l ^ o): the lower lane is always the left operand, so both lanes compute the same bits
        const float nb = __shfl_xor(na, o), b = __shfl_xor(a, o), qb = __shfl_xor(m2, o);
        const bool low = (lane & o) == 0;
        float n0 = low ? na : nb, a0 = low ? a : b, q0 = low ? m2 : qb;
        gn_merge(n0, a0, q0, low ? nb : na, low ? b : a, low ? qb : m2);
        na = n0; a = a0; m2 = q0;
    }
    if (lane == 0) {
        stats[((int64_t)n * groups + g) * 2 + 0] = a;
        stats[((int64_t)n * groups + g) * 2 + 1] = na > 0.f ? m2 / na : 0.f;
    }
}

__global__ void __launch_bounds__(256) gn_finalize_small_kernel(const float* __restrict__ partials, float* __restrict__ stats, int chunks,
                                                                int groups) {
    // few chunks (<= 64): one block per n; thread (q, g), q < nq = 256 / groups, folds the chunks q, q + nq, ... of group g in
    // order, then the nq results of a group are folded in order.  Same arithmetic (gn_merge) as the wave-per-group kernel.
    __shared__ float red[256][3];
    const int n = blockIdx.x, nq = 256 / groups;
    const int g = threadIdx.x % groups, q = threadIdx.x / groups;
    if (q < nq) {
        const float* src = partials + ((int64_t)n * chunks * groups + g) * 3;
        float na = 0.f, a = 0.f, m2 = 0.f;
        for (int c = q; c < chunks; c += nq) {
            const float* t = src + (int64_t)c * groups * 3;
            gn_merge(na, a, m2, t[0], t[1], t[2]);
        }
        red[q * groups + g][0] = na; red[q * groups + g][1] = a; red[q * groups + g][2] = m2;
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        float na = red[g][0], a = red[g][1], m2 = red[g][2];
        for (int w = 1; w < nq; ++w) gn_merge(na, a, m2, red[w * groups + g][0], red[w * groups + g][1], red[w * groups + g][2]);
        stats[((int64_t)n * groups + g) * 2 + 0] = a;
        stats[((int64_t)n * groups + g) * 2 + 1] = na > 0.f ? m2 / na : 0.f;
    }
}

// x2 != nullptr (vcx_groupnorm_apply2_f16): the channels [0, c1) of a pixel come from x [n][pixels][c1], the channels [c1, C) from
// x2 [n][pixels][C - c1] - the two halves of a channel concat that is never materialised (c1 % 8 == 0: a thread's chunk lies in one half)
__global__ void gn_apply_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int c1, half_t* __restrict__ y, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int64_t pixels, int C,
                                int groups, float eps, int silu, int64_t pix_per_block, int cw, int pl) {
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int cpg = C / groups;
    const int c8 = tid % cw, plane = tid / cw;
    float sc[8], sh[8];             // x * sc + (beta - mean * sc): one fma per element; the rounding of mean * sc costs
#pragma unroll                      // 1e-7 |mean| / std of the output scale - below the fp16 output step up to |mean| ~ 1000 std
    for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        const int g = c / cpg;
        const float mean = stats[((int64_t)n * groups + g) * 2 + 0];
        const float var = stats[((int64_t)n * groups + g) * 2 + 1];
        const float a = rsqrtf(var + eps) * gamma[c];
        sc[e] = a;
        sh[e] = beta[c] - mean * a;
    }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < pixels) ? p0 + pix_per_block : pixels;
    int64_t Cx = C;                  // row stride of this thread's source
    const half_t* xp = x + ((int64_t)n * pixels) * C + c8 * 8;
    if (x2) {
        const bool right = c8 * 8 >= c1;
        Cx = right ? C - c1 : c1;
        xp = (right ? x2 - c1 : x) + ((int64_t)n * pixels) * Cx + c8 * 8;
    }
    half_t* yp = y + ((int64_t)n * pixels) * C + c8 * 8;
    const int64_t step = pl;
    int64_t pix = p0 + plane;
    for (; pix + 3 * step < p1; pix += 4 * step) {
        h8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const h8*>(xp + (pix + u * step) * Cx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[u][e] * sc[e] + sh[e];
                if (silu) f = vcx_silu(f);
                v[u][e] = (half_t)f;
            }
            *reinterpret_cast<h8*>(yp + (pix + u * step) * C) = v[u];
        }
    }
    for (; pix < p1; pix += step) {
        h8 v = *reinterpret_cast<const h8*>(xp + pix * Cx);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * sc[e] + sh[e];
            if (silu) f = vcx_silu(f);
            v[e] = (half_t)f;
        }
        *reinterpret_cast<h8*>(yp + pix * C) = v;
    }
}

// ---------------------------------------------------------------------------------------
// LayerNorm: LPR lanes per row (8..64, chosen so a lane holds <= 4 chunks of 8 channels), the row stays in registers:
// one HBM read, mean then centred variance from the registers (as torch computes them), one HBM write.
// ---------------------------------------------------------------------------------------
// STATS: write (mean, rstd) per row instead of the normalised row - the read-only pass in front of a projection that has the
// LayerNorm folded into its weights and epilogue (VCX_GEMM_LNFOLD_*): same summation order, hence the same statistics bit for bit.
template <int LPR, int CPL, bool STATS = false>
__global__ void __launch_bounds__(256) layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int C, float eps) {
    constexpr int RPB = 256 / LPR;                 // rows per block
    const int sub = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool rvalid = row < rows;
    const int nc8 = C >> 3;
    const half_t* xr = x + (rvalid ? row : 0) * C;
    h8 v[CPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = sub + j * LPR;
        if (c < nc8) {
            v[j] = *reinterpret_cast<const h8*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[j][e];
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (sub + j * LPR < nc8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)v[j][e] - mean;
                q += d * d;
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!rvalid) return;
    if (STATS) {
        if (sub == 0) reinterpret_cast<float2*>(y)[row] = make_float2(mean, rstd);
        return;
    }
    half_t* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = sub + j * LPR;
        if (c < nc8) {
            const f4 g0 = *reinterpret_cast<const f4*>(gamma + c * 8), g1 = *reinterpret_cast<const f4*>(gamma + c * 8 + 4);
            const f4 b0 = *reinterpret_cast<const f4*>(beta + c * 8), b1 = *reinterpret_cast<const f4*>(beta + c * 8 + 4);
            h8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)(((float)v[j][e] - mean) * rstd * g0[e] + b0[e]);
                o[e + 4] = (half_t)(((float)v[j][e + 4] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<h8*>(yr + c * 8) = o;
        }
    }
}

template <int LPR, int CPL>
void launch_ln(const half_t* x, half_t* y, const float* g, const float* b, int64_t rows, int C, float eps, hipStream_t s) {
    constexpr int RPB = 256 / LPR;
    const dim3 grid((unsigned)((rows + RPB - 1) / RPB));
    if (g) hipLaunchKernelGGL((layernorm_kernel<LPR, CPL, false>), grid, dim3(256), 0, s, x, y, g, b, rows, C, eps);
    else hipLaunchKernelGGL((layernorm_kernel<LPR, CPL, true>), grid, dim3(256), 0, s, x, y, g, b, rows, C, eps);   // y = float2 stats
}

void dispatch_ln(const half_t* xp, half_t* yp, const float* gamma, const float* beta, int64_t rows, int C, float eps, hipStream_t s) {
    const int nc8 = C >> 3;
    // C = 320 and C = 640 (the token widths of levels 0 / 1): half the lanes per row and five 16-byte chunks per lane - 5 loads in flight
    // per thread, 32 / 16 rows per block - measured 3-8 % faster than the 3-chunk forms on those widths (profiles/r04o_ln_variants.txt;
    // twice the lanes per row is 20-40 % slower)
    if (nc8 > 32 && nc8 <= 40) return launch_ln<8, 5>(xp, yp, gamma, beta, rows, C, eps, s);
    if (nc8 > 64 && nc8 <= 80) return launch_ln<16, 5>(xp, yp, gamma, beta, rows, C, eps, s);
    if (nc8 <= 8) launch_ln<8, 1>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 16) launch_ln<8, 2>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 32) launch_ln<16, 2>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 48) launch_ln<16, 3>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 64) launch_ln<16, 4>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 96) launch_ln<32, 3>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 128) launch_ln<32, 4>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 192) launch_ln<64, 3>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 256) launch_ln<64, 4>(xp, yp, gamma, beta, rows, C, eps, s);
    else if (nc8 <= 512) launch_ln<64, 8>(xp, yp, gamma, beta, rows, C, eps, s);
    else launch_ln<64, 16>(xp, yp, gamma, beta, rows, C, eps, s);
}

// thread geometry shared by the two GroupNorm kernels: cw channel chunks x pl pixel lanes, ~320 threads
void gn_geometry(int C, int& cw, int& pl) {
    cw = C >> 3;
    pl = 320 / cw;
    if (pl < 1) pl = 1;
    while (cw * pl > 512) --pl;
}

int64_t pick_pix_per_block(int n_outer, int64_t pixels) {
    // A function of `pixels` only: the summation order - hence the bits of the result - must not depend on how many
    // samples are batched (cond/uncond run as one B=2 forward and must equal two B=1 forwards exactly).
    // <= 1024 partials per n; >= 128 pixels per block on large images, 16 up to 4096 pixels: the 18 MB tensors of the 9 x 16-pixel level
    // (3600 pixels per video, n = 2) are 450 blocks instead of 58 - round 6: 24 -> ~8 us per pass.
    (void)n_outer;
#ifndef VCX_GN_TWO_PHASE          // (tools/build_abl.sh gnold -DVCX_GN_TWO_PHASE: the round-5 statistics plumbing, for the same-box A/B)
    if (pixels <= 4096) return 16;
#endif
    int64_t ppb = (pixels + 1023) / 1024;
    if (ppb < 128) ppb = 128;
    return ppb;
}

// Direct form of the statistics pass (gn_stats_kernel with `stats`): one block per (n, slice of whole groups) sees all pixels.  Taken
// up to 512 pixels when C splits into slices of Cs = the smallest multiple of a group AND of eight channels, doubled up to 64.
bool gn_direct_geometry(int64_t pixels, int C, int groups, int& Cs, int& cw, int& pl) {
#ifdef VCX_GN_TWO_PHASE
    return false;
#endif
    if (pixels > 512) return false;
    const int cpg = C / groups;
    Cs = cpg;
    while (Cs % 8 != 0) Cs += cpg;
    if (Cs > C || C % Cs != 0) return false;
    while (Cs < 64 && C % (2 * Cs) == 0) Cs *= 2;
    cw = Cs >> 3;
    pl = 512 / cw;
    if (pl > pixels) pl = (int)pixels;
    const int gslice = Cs / cpg;                        // one thread per group of the slice writes its statistics: the block must have them
    if (cw * pl < gslice) pl = (gslice + cw - 1) / cw;  // (pixel lanes beyond the image own no pixel: they contribute zeros - found by the fuzz at 1 pixel x 32 channels)
    if (pl < 1 || cw * pl > 1024) return false;
    return sizeof(float) * (2 * (size_t)pl * Cs + Cs) <= 64 * 1024;
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics from column moments in ONE launch (round 6; two kernels before: 287 launches of 5-9 us per DDIM step).  A block
// takes `gpb` whole groups (cols = gpb * C / groups consecutive columns) of one n and ALL strips: thread (column j, strip lane l) folds
// the strips l, l + L, ... of its column in order (eight 8-byte loads in flight), the L lanes of a column are folded in order, then the
// columns of each group in order -> stats[n][g] = (mean, biased variance).  No atomics, no second pass: bit-reproducible, and the
// geometry is a function of (strips, C, groups) only - the same bits for any batch size.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) gn_colstats_direct_kernel(const float2* __restrict__ cs, float* __restrict__ stats, int strips, int C,
                                                                  int groups, int gpb, int L) {
    extern __shared__ float red[];                      // [L][cols][3]
    const int cpg = C / groups, cols = gpb * cpg;
    const int n = blockIdx.y, g0 = blockIdx.x * gpb;
    const int tid = threadIdx.x, j = tid % cols, l = tid / cols;
    const int c = g0 * cpg + j;
    if (l < L) {
        float na = 0.f, a = 0.f, q = 0.f;
        if (c < C) {
            const float2* src = cs + ((int64_t)n * strips) * C + c;
            for (int sb = l; sb < strips; sb += 8 * L) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sb + u * L < strips ? sb + u * L : l) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) gn_merge(na, a, q, sb + u * L < strips ? 64.f : 0.f, v[u].x, sb + u * L < strips ? v[u].y : 0.f);
            }
        }
        float* r = red + ((size_t)l * cols + j) * 3;
        r[0] = na; r[1] = a; r[2] = q;
    }
    __syncthreads();
    if (tid < cols) {                                   // strip lanes of a column, in order
        float na = red[tid * 3], a = red[tid * 3 + 1], q = red[tid * 3 + 2];
        for (int k = 1; k < L; ++k) {
            const float* r = red + ((size_t)k * cols + tid) * 3;
            gn_merge(na, a, q, r[0], r[1], r[2]);
        }
        red[tid * 3] = na; red[tid * 3 + 1] = a; red[tid * 3 + 2] = q;
    }
    __syncthreads();
    if (tid < gpb && g0 + tid < groups) {               // columns of a group, in order
        const int c0 = tid * cpg;
        float na = red[c0 * 3], a = red[c0 * 3 + 1], q = red[c0 * 3 + 2];
        for (int k = c0 + 1; k < c0 + cpg; ++k) gn_merge(na, a, q, red[k * 3], red[k * 3 + 1], red[k * 3 + 2]);
        stats[((int64_t)n * groups + g0 + tid) * 2 + 0] = a;
        stats[((int64_t)n * groups + g0 + tid) * 2 + 1] = na > 0.f ? q / na : 0.f;
    }
}

// geometry of the one-launch form: up to 1024 strips per n, groups of at most 256 columns.  Strip lanes first - as many as give a
// thread ONE batch of eight loads (the kernel is a latency chain: r06e, 11.4 us per launch with two batches per thread on 200 blocks) -
// then as many whole groups per block as fit 1024 threads, 80 columns at most.  A function of (strips, C, groups) only.
bool gn_colstats_direct_geometry(int64_t strips, int C, int groups, int& gpb, int& L) {
#ifdef VCX_GN_TWO_PHASE
    return false;
#endif
    const int cpg = C / groups;
    // (up to 4096 strips for narrow groups - the per-video norms of level 0, 102 lanes x 36 strips - measured: GroupNorm family +0.35 ms against the
    // partials + finalize pair, profiles/r06u_step_ab.txt: the pair's 225 blocks stream the 18 MB of moments, 64 blocks with 80-byte rows do not)
    if (strips > 1024 || cpg > 256) return false;
    int lanes = (int)((strips + 7) / 8);
    if (lanes > 1024 / cpg) lanes = 1024 / cpg;         // (one group per block at least)
    gpb = 1024 / lanes / cpg;
    if (gpb > 80 / cpg) gpb = 80 / cpg;
    if (gpb < 1) gpb = 1;
    if (gpb > groups) gpb = groups;
    const int cols = gpb * cpg;
    L = 1024 / cols;
    if (L > strips) L = (int)strips;
    return L >= 1;
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics from column moments (VCX_GEMM_COLSTATS): colstats[n * strips + s][C] = (mean, M2) of 64 values each.  A block
// takes `spb` consecutive strips of one n and ALL columns (thread = column: coalesced 8-byte reads, 16 loads in flight), folds the
// strips of a column in order (gn_merge, equal counts), then the columns of each group in order, and writes (count, mean, M2) per
// (n, chunk, group) in gn_stats_kernel's layout - the same finalize kernels then merge the chunks.  No atomics: bit-reproducible and
// independent of the batch size.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_colstats_partials_kernel(const float2* __restrict__ cs, float* __restrict__ partials, int64_t strips,
                                                                   int C, int groups, int spb) {
    extern __shared__ float red[];                      // [C][3]
    const int n = blockIdx.y;
    const int64_t s0 = (int64_t)blockIdx.x * spb, s1 = (s0 + spb < strips) ? s0 + spb : strips;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float2* src = cs + ((int64_t)n * strips) * C + c;
        float na = 0.f, a = 0.f, q = 0.f;
        for (int64_t sb = s0; sb < s1; sb += 16) {
            float2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(sb + u < s1 ? sb + u : s1 - 1) * C];
#pragma unroll
            for (int u = 0; u < 16; ++u) gn_merge(na, a, q, sb + u < s1 ? 64.f : 0.f, v[u].x, sb + u < s1 ? v[u].y : 0.f);
        }
        red[c * 3] = na; red[c * 3 + 1] = a; red[c * 3 + 2] = q;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int cpg = C / groups, c0 = threadIdx.x * cpg;
        float na = red[c0 * 3], a = red[c0 * 3 + 1], q = red[c0 * 3 + 2];
        for (int c = c0 + 1; c < c0 + cpg; ++c) gn_merge(na, a, q, red[c * 3], red[c * 3 + 1], red[c * 3 + 2]);
        float* dst = partials + (((int64_t)n * gridDim.x + blockIdx.x) * groups + threadIdx.x) * 3;
        dst[0] = na; dst[1] = a; dst[2] = q;
    }
}

// GroupNorm folded into the linear layer behind it (vcx_groupnorm_fold_linear_f16): one wave per (output row o, statistics unit n).
//   Wn[n][o][c] = fp16(W[o][c] a[c]),  a[c] = gamma[c] rstd[n, g(c)]
//   bn[n][o]    = bias[o] + sum_c (W[o][c] beta[c] - float(Wn[n][o][c]) mean[n, g(c)])
// The mean term uses the ROUNDED weight - what the GEMM will multiply the un-normalised rows with - so a common offset of a group's
// channels cancels exactly, as in the folded LayerNorm (colsum of the rounded W').  Lane l owns the 4-channel pieces l, l + 64, ...;
// the row sum is a fixed butterfly: bit-reproducible and the same bits whatever the number of units.
__global__ void __launch_bounds__(64) gn_fold_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ stats, half_t* __restrict__ Wn,
                                                            float* __restrict__ bn, int N, int C, int groups, float eps) {
    const int o = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int cpg = C / groups;
    const float* wrow = W + (int64_t)o * C;
    half_t* orow = Wn + ((int64_t)n * N + o) * C;
    const float* st = stats + (int64_t)n * groups * 2;
    float acc = 0.f;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        const f4 w = *reinterpret_cast<const f4*>(wrow + c0);
        const f4 ga = *reinterpret_cast<const f4*>(gamma + c0);
        const f4 be = *reinterpret_cast<const f4*>(beta + c0);
        h4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c0 + e) / cpg;
            const float mean = st[2 * g], a = rsqrtf(st[2 * g + 1] + eps) * ga[e];
            const half_t wh = (half_t)(w[e] * a);
            out[e] = wh;
            acc += w[e] * be[e] - (float)wh * mean;
        }
        *reinterpret_cast<h4*>(orow + c0) = out;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) bn[(int64_t)n * N + o] = (bias ? bias[o] : 0.f) + acc;
}

}  // namespace

extern "C" int vcx_groupnorm_stats_from_colstats_f32(const float* colstats, float* stats, void* ws, int n_outer, int64_t pixels, int C,
                                                     int groups, void* stream) {
    VCX_REQUIRE(colstats && stats && ws, "vcx_groupnorm_stats_from_colstats_f32: null pointer");
    VCX_REQUIRE(n_outer > 0 && n_outer <= 65535 && pixels > 0 && pixels % 64 == 0 && C > 0 && C <= MAXC && groups > 0 && groups <= 64 && C % groups == 0,
                "vcx_groupnorm_stats_from_colstats_f32: need pixels %% 64 == 0, C %% groups == 0, groups <= 64 (pixels=%lld C=%d groups=%d)",
                (long long)pixels, C, groups);
    VCX_REQUIRE(((uintptr_t)colstats & 7) == 0 && ((uintptr_t)ws & 3) == 0, "vcx_groupnorm_stats_from_colstats_f32: alignment");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, 8.0 * n_outer * (double)(pixels / 64) * C);
    const int64_t strips = pixels / 64;
    int gpb, L;
    if (gn_colstats_direct_geometry(strips, C, groups, gpb, L)) {
        const int cols = gpb * (C / groups);
        hipLaunchKernelGGL(gn_colstats_direct_kernel, dim3((unsigned)((groups + gpb - 1) / gpb), (unsigned)n_outer), dim3(cols * L),
                           sizeof(float) * 3 * (size_t)cols * L, s, reinterpret_cast<const float2*>(colstats), stats, (int)strips, C, groups, gpb, L);
        return vcx_check_launch("vcx_groupnorm_stats_from_colstats_f32(direct)");
    }
    // <= 1024 chunks per n (the workspace of vcx_groupnorm_ws_bytes), a function of `pixels` only: same bits for any batch size
    int64_t spb = (strips + 1023) / 1024;
    if (spb < 16) spb = 16;
    const int chunks = (int)((strips + spb - 1) / spb);
    hipLaunchKernelGGL(gn_colstats_partials_kernel, dim3((unsigned)chunks, (unsigned)n_outer), dim3(256), sizeof(float) * 3 * (size_t)C, s,
                       reinterpret_cast<const float2*>(colstats), (float*)ws, strips, C, groups, (int)spb);
    int rc = vcx_check_launch("vcx_groupnorm_stats_from_colstats_f32");
    if (rc) return rc;
    if (chunks <= 64) hipLaunchKernelGGL(gn_finalize_small_kernel, dim3(n_outer), dim3(256), 0, s, (const float*)ws, stats, chunks, groups);
    else hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_outer * groups), dim3(64), 0, s, (const float*)ws, stats, chunks, groups);
    return vcx_check_launch("vcx_groupnorm_stats_from_colstats_f32(finalize)");
}

extern "C" size_t vcx_groupnorm_ws_bytes(int n_outer, int64_t pixels, int groups) {
    if (n_outer <= 0 || pixels <= 0 || groups <= 0) return 0;
    const int64_t ppb = pick_pix_per_block(n_outer, pixels);
    const int64_t chunks = (pixels + ppb - 1) / ppb;
    return sizeof(float) * 3 * (size_t)n_outer * (size_t)chunks * (size_t)groups;
}

extern "C" int vcx_groupnorm_stats_f16(const void* x, float* stats, void* ws, int n_outer, int64_t pixels, int C, int groups,
                                       void* stream) {
    VCX_REQUIRE(x && stats && ws, "vcx_groupnorm_stats_f16: null pointer");
    VCX_REQUIRE(n_outer > 0 && pixels > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0 && C % 8 == 0 && C <= MAXC,
                "vcx_groupnorm_stats_f16: need C %% 8 == 0, C %% groups == 0, groups <= 64, C <= %d (C=%d groups=%d)", MAXC, C, groups);
    VCX_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 3) == 0, "vcx_groupnorm_stats_f16: x must be 16-byte aligned");
    VCX_REQUIRE(n_outer <= 65535, "vcx_groupnorm_stats_f16: n_outer too large");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, 2.0 * n_outer * (double)pixels * C);
    int cw, pl, Cs;
    if (gn_direct_geometry(pixels, C, groups, Cs, cw, pl)) {       // small images: the block's moments are the statistics, one launch
        hipLaunchKernelGGL(gn_stats_kernel, dim3(1, (unsigned)n_outer, (unsigned)(C / Cs)), dim3(cw * pl), sizeof(float) * (2 * (size_t)pl * Cs + Cs), s,
                           (const half_t*)x, (float*)nullptr, stats, pixels, C, groups, Cs, pixels, cw, pl);
        return vcx_check_launch("vcx_groupnorm_stats_f16(direct)");
    }
    const int64_t ppb = pick_pix_per_block(n_outer, pixels);
    const int chunks = (int)((pixels + ppb - 1) / ppb);
    dim3 grid((unsigned)chunks, n_outer);
    gn_geometry(C, cw, pl);
    const size_t smem = sizeof(float) * (2 * (size_t)pl * C + C);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(cw * pl), smem, s, (const half_t*)x, (float*)ws, (float*)nullptr, pixels, C, groups, C, ppb, cw, pl);
    int rc = vcx_check_launch("vcx_groupnorm_stats_f16");
    if (rc) return rc;
    // the choice depends on the pixel count only (never on the batch): same bits for B = 1 and B = 2
    if (chunks <= 64) hipLaunchKernelGGL(gn_finalize_small_kernel, dim3(n_outer), dim3(256), 0, s, (const float*)ws, stats, chunks, groups);
    else hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_outer * groups), dim3(64), 0, s, (const float*)ws, stats, chunks, groups);
    return vcx_check_launch("vcx_groupnorm_stats_f16(finalize)");
}

static int gn_apply_launch(const void* x, const void* x2, int c1, void* y, const float* stats, const float* gamma, const float* beta,
                           int n_outer, int64_t pixels, int C, int groups, float eps, int silu, void* stream) {
    VCX_REQUIRE(x && y && stats && gamma && beta, "vcx_groupnorm_apply_f16: null pointer");
    VCX_REQUIRE(n_outer > 0 && pixels > 0 && C > 0 && C <= MAXC && groups > 0 && C % groups == 0 && C % 8 == 0,
                "vcx_groupnorm_apply_f16: need C %% 8 == 0, C <= %d (C=%d groups=%d)", MAXC, C, groups);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "vcx_groupnorm_apply_f16: x/y must be 16-byte aligned");
    VCX_REQUIRE(n_outer <= 65535, "vcx_groupnorm_apply_f16: n_outer too large");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, 4.0 * n_outer * (double)pixels * C);
    // The apply pass has no reduction, so its block size is free (the statistics kernels' is not - it fixes their summation order).
    // Many small blocks win: 32 pixels per block (16 for wide rows) = one or two 4-deep batches of 16-byte loads per thread, against
    // the statistics kernel's >= 128 that round 1-3 used here too: -12 % over the apply passes of a step, 5.8 TB/s on the level-0
    // tensors (profiles/r04k_gn_apply_ppb_*.txt; larger blocks lose badly: 512 pixels +37 %).
    const int64_t ppb = C <= 640 ? 32 : 16;
    dim3 grid((unsigned)((pixels + ppb - 1) / ppb), n_outer);
    int cw, pl;
    gn_geometry(C, cw, pl);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(cw * pl), 0, s, (const half_t*)x, (const half_t*)x2, c1, (half_t*)y, stats, gamma, beta, pixels, C,
                       groups, eps, silu, ppb, cw, pl);
    return vcx_check_launch("vcx_groupnorm_apply_f16");
}

extern "C" int vcx_groupnorm_apply_f16(const void* x, void* y, const float* stats, const float* gamma, const float* beta,
                                       int n_outer, int64_t pixels, int C, int groups, float eps, int silu, void* stream) {
    return gn_apply_launch(x, nullptr, 0, y, stats, gamma, beta, n_outer, pixels, C, groups, eps, silu, stream);
}

extern "C" int vcx_groupnorm_apply2_f16(const void* x1, int c1, const void* x2, void* y, const float* stats, const float* gamma, const float* beta,
                                        int n_outer, int64_t pixels, int C, int groups, float eps, int silu, void* stream) {
    VCX_REQUIRE(x2 && c1 > 0 && c1 < C && c1 % 8 == 0 && ((uintptr_t)x2 & 15) == 0, "vcx_groupnorm_apply2_f16: need 0 < c1 < C, c1 %% 8 == 0, x2 16-byte aligned (c1=%d C=%d)", c1, C);
    return gn_apply_launch(x1, x2, c1, y, stats, gamma, beta, n_outer, pixels, C, groups, eps, silu, stream);
}

extern "C" int vcx_groupnorm_fold_linear_f16(const float* W, const float* bias, const float* gamma, const float* beta, const float* stats,
                                             void* Wn, float* bn, int n_outer, int N, int C, int groups, float eps, void* stream) {
    VCX_REQUIRE(W && gamma && beta && stats && Wn && bn, "vcx_groupnorm_fold_linear_f16: null pointer");
    VCX_REQUIRE(n_outer > 0 && n_outer <= 65535 && N > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0,
                "vcx_groupnorm_fold_linear_f16: need C %% 4 == 0, C %% groups == 0, n_outer <= 65535 (C=%d groups=%d n_outer=%d)", C, groups, n_outer);
    VCX_REQUIRE((((uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)Wn & 7) == 0,
                "vcx_groupnorm_fold_linear_f16: W / gamma / beta must be 16-byte, Wn 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, (double)N * C * (4.0 + 2.0 * n_outer));
    hipLaunchKernelGGL(gn_fold_linear_kernel, dim3((unsigned)N, (unsigned)n_outer), dim3(64), 0, s, W, bias, gamma, beta, stats, (half_t*)Wn, bn, N, C,
                       groups, eps);
    return vcx_check_launch("vcx_groupnorm_fold_linear_f16");
}

extern "C" int vcx_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C,
                                 float eps, void* stream) {
    VCX_REQUIRE(x && y && gamma && beta, "vcx_layernorm_f16: null pointer");
    VCX_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "vcx_layernorm_f16: need C %% 8 == 0 (C=%d)", C);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                "vcx_layernorm_f16: pointers must be 16-byte aligned");
    VCX_REQUIRE(rows < (1ll << 31) && C <= 8192, "vcx_layernorm_f16: too many rows or C > 8192");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_LN, s, 0.0, 4.0 * rows * (double)C);
    dispatch_ln((const half_t*)x, (half_t*)y, gamma, beta, rows, C, eps, s);
    return vcx_check_launch("vcx_layernorm_f16");
}

extern "C" int vcx_rowstats_f16(const void* x, float* stats, int64_t rows, int C, float eps, void* stream) {
    VCX_REQUIRE(x && stats, "vcx_rowstats_f16: null pointer");
    VCX_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "vcx_rowstats_f16: need C %% 8 == 0 (C=%d)", C);
    VCX_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)stats & 7) == 0, "vcx_rowstats_f16: x must be 16-byte, stats 8-byte aligned");
    VCX_REQUIRE(rows < (1ll << 31) && C <= 8192, "vcx_rowstats_f16: too many rows or C > 8192");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_LN, s, 0.0, 2.0 * rows * (double)C + 8.0 * rows);
    dispatch_ln((const half_t*)x, reinterpret_cast<half_t*>(stats), nullptr, nullptr, rows, C, eps, s);
    return vcx_check_launch("vcx_rowstats_f16");
}
