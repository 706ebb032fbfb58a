// GroupNorm (+SiLU) and LayerNorm on channels-last fp16 with fp32 statistics (HBM-bound).
#include "vcx_common.h"

namespace {

constexpr int MAXC = 4096;  // LDS table limit for groupnorm_apply

// ---------------------------------------------------------------------------------------
// GroupNorm statistics: x [n_outer][pixels][C].  grid = (chunks, n_outer); a block reduces a
// contiguous pixel range.  Thread t owns channel chunk (t % CW) (8 channels = one 16-byte
// load) and pixel lane (t / CW), so a wave reads whole pixel rows back to back.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_stats_kernel(const half_t* __restrict__ x, float* __restrict__ stats,
                                                       int64_t pixels, int C, int groups, int64_t pix_per_block) {
    __shared__ float gsum[64], gsq[64];
    const int tid = threadIdx.x;
    if (tid < 64) { gsum[tid] = 0.f; gsq[tid] = 0.f; }
    __syncthreads();
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < pixels) ? p0 + pix_per_block : pixels;
    const int nc8 = C >> 3;
    const int cw = nc8 < 256 ? nc8 : 256;
    const int pl = 256 / cw;
    const int cpg = C / groups;
    const half_t* xn = x + (int64_t)n * pixels * C;
    if (tid < cw * pl) {
        const int plane = tid / cw;
        for (int c8 = tid % cw; c8 < nc8; c8 += cw) {
            float s[8], ss[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
            for (int64_t pix = p0 + plane; pix < p1; pix += pl) {
                const h8 v = *reinterpret_cast<const h8*>(xn + pix * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    s[e] += f;
                    ss[e] += f * f;
                }
            }
            // fold the 8 channels into their groups (consecutive channels mostly share one)
            int gcur = (c8 * 8) / cpg;
            float as = 0.f, aq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ge = (c8 * 8 + e) / cpg;
                if (ge != gcur) {
                    atomicAdd(&gsum[gcur], as);
                    atomicAdd(&gsq[gcur], aq);
                    as = aq = 0.f;
                    gcur = ge;
                }
                as += s[e];
                aq += ss[e];
            }
            atomicAdd(&gsum[gcur], as);
            atomicAdd(&gsq[gcur], aq);
        }
    }
    __syncthreads();
    if (tid < groups) {
        atomicAdd(&stats[((int64_t)n * groups + tid) * 2 + 0], gsum[tid]);
        atomicAdd(&stats[((int64_t)n * groups + tid) * 2 + 1], gsq[tid]);
    }
}

__global__ void __launch_bounds__(256) gn_apply_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int64_t pixels, int C, int groups,
                                                       float eps, int silu, int64_t pix_per_block) {
    __shared__ float sc[MAXC], sh[MAXC];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int cpg = C / groups;
    const float cnt = (float)((double)pixels * cpg);
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float su = stats[((int64_t)n * groups + g) * 2 + 0];
        const float sq = stats[((int64_t)n * groups + g) * 2 + 1];
        const float mean = su / cnt;
        float var = sq / cnt - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rstd = rsqrtf(var + eps);
        const float a = rstd * gamma[c];
        sc[c] = a;
        sh[c] = beta[c] - mean * a;
    }
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < pixels) ? p0 + pix_per_block : pixels;
    const int nc8 = C >> 3;
    const int64_t total = (p1 - p0) * nc8;
    const half_t* xn = x + ((int64_t)n * pixels + p0) * C;
    half_t* yn = y + ((int64_t)n * pixels + p0) * C;
    for (int64_t i = tid; i < total; i += 256) {
        const int c0 = (int)(i % nc8) * 8;
        h8 v = *reinterpret_cast<const h8*>(xn + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * sc[c0 + e] + sh[c0 + e];
            if (silu) f = vcx_silu(f);
            v[e] = (half_t)f;
        }
        *reinterpret_cast<h8*>(yn + i * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, two-pass statistics (mean, then centred variance) as torch
// computes them; the row is re-read from L1/L2 rather than held in registers.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const half_t* xr = x + row * C;
    half_t* yr = y + row * C;
    const int nc8 = C >> 3;
    float s = 0.f;
    for (int c = lane; c < nc8; c += 64) {
        const h8 v = *reinterpret_cast<const h8*>(xr + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e];
    }
    const float mean = vcx_wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < nc8; c += 64) {
        const h8 v = *reinterpret_cast<const h8*>(xr + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)v[e] - mean;
            q += d * d;
        }
    }
    const float rstd = rsqrtf(vcx_wave_sum(q) / (float)C + eps);
    for (int c = lane; c < nc8; c += 64) {
        h8 v = *reinterpret_cast<const h8*>(xr + c * 8);
        const f4 g0 = *reinterpret_cast<const f4*>(gamma + c * 8), g1 = *reinterpret_cast<const f4*>(gamma + c * 8 + 4);
        const f4 b0 = *reinterpret_cast<const f4*>(beta + c * 8), b1 = *reinterpret_cast<const f4*>(beta + c * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (half_t)(((float)v[e] - mean) * rstd * g0[e] + b0[e]);
            v[e + 4] = (half_t)(((float)v[e + 4] - mean) * rstd * g1[e] + b1[e]);
        }
        *reinterpret_cast<h8*>(yr + c * 8) = v;
    }
}

int64_t pick_pix_per_block(int n_outer, int64_t pixels) {
    // aim for >= ~2048 blocks chip-wide but keep at least 64 pixels per block
    int64_t chunks = (2048 + n_outer - 1) / n_outer;
    if (chunks < 1) chunks = 1;
    int64_t ppb = (pixels + chunks - 1) / chunks;
    if (ppb < 64) ppb = 64;
    return ppb;
}

}  // namespace

extern "C" int vcx_groupnorm_stats_f16(const void* x, float* stats, int n_outer, int64_t pixels, int C, int groups,
                                       void* stream) {
    VCX_REQUIRE(x && stats, "vcx_groupnorm_stats_f16: null pointer");
    VCX_REQUIRE(n_outer > 0 && pixels > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0 && C % 8 == 0,
                "vcx_groupnorm_stats_f16: need C %% 8 == 0, C %% groups == 0, groups <= 64 (C=%d groups=%d)", C, groups);
    VCX_REQUIRE(((uintptr_t)x & 15) == 0, "vcx_groupnorm_stats_f16: x must be 16-byte aligned");
    VCX_REQUIRE(n_outer <= 65535, "vcx_groupnorm_stats_f16: n_outer too large");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, 2.0 * n_outer * (double)pixels * C);
    if (hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)n_outer * groups, s) != hipSuccess) {
        vcx_set_error("vcx_groupnorm_stats_f16: memset failed");
        return VCX_ELAUNCH;
    }
    const int64_t ppb = pick_pix_per_block(n_outer, pixels);
    dim3 grid((unsigned)((pixels + ppb - 1) / ppb), n_outer);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, s, (const half_t*)x, stats, pixels, C, groups, ppb);
    return vcx_check_launch("vcx_groupnorm_stats_f16");
}

extern "C" int vcx_groupnorm_apply_f16(const void* x, void* y, const float* stats, const float* gamma, const float* beta,
                                       int n_outer, int64_t pixels, int C, int groups, float eps, int silu, void* stream) {
    VCX_REQUIRE(x && y && stats && gamma && beta, "vcx_groupnorm_apply_f16: null pointer");
    VCX_REQUIRE(n_outer > 0 && pixels > 0 && C > 0 && C <= MAXC && groups > 0 && C % groups == 0 && C % 8 == 0,
                "vcx_groupnorm_apply_f16: need C %% 8 == 0, C <= %d (C=%d groups=%d)", MAXC, C, groups);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "vcx_groupnorm_apply_f16: x/y must be 16-byte aligned");
    VCX_REQUIRE(n_outer <= 65535, "vcx_groupnorm_apply_f16: n_outer too large");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_GN, s, 0.0, 4.0 * n_outer * (double)pixels * C);
    const int64_t ppb = pick_pix_per_block(n_outer, pixels);
    dim3 grid((unsigned)((pixels + ppb - 1) / ppb), n_outer);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, s, (const half_t*)x, (half_t*)y, stats, gamma, beta, pixels, C,
                       groups, eps, silu, ppb);
    return vcx_check_launch("vcx_groupnorm_apply_f16");
}

extern "C" int vcx_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C,
                                 float eps, void* stream) {
    VCX_REQUIRE(x && y && gamma && beta, "vcx_layernorm_f16: null pointer");
    VCX_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "vcx_layernorm_f16: need C %% 8 == 0 (C=%d)", C);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                "vcx_layernorm_f16: pointers must be 16-byte aligned");
    const int64_t nblk = (rows + 3) / 4;
    VCX_REQUIRE(nblk < (1ll << 31), "vcx_layernorm_f16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_LN, s, 0.0, 4.0 * rows * (double)C);
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const half_t*)x, (half_t*)y, gamma, beta,
                       rows, C, eps);
    return vcx_check_launch("vcx_layernorm_f16");
}
