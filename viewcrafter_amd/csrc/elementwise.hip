// Element-wise, layout and DDIM-update kernels (all HBM/launch bound).
#include "vcx_common.h"
#include "gemm_args.h"   // gelu_erf
#include <math.h>

namespace {

__global__ void silu_f32_kernel(const float* x, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = vcx_silu(x[i]);
}

// out[b][0:half] = cos(t*f_j), out[b][half:2*half] = sin(t*f_j), f_j = exp(-ln(max_period) j / half)
__global__ void timestep_embedding_kernel(const int64_t* t, float* out, int B, int dim, float log_max_period) {
    const int half = dim / 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float freq = expf(-log_max_period * (float)j / (float)half);
    const float arg = (float)t[b] * freq;
    out[(int64_t)b * dim + j] = cosf(arg);
    out[(int64_t)b * dim + half + j] = sinf(arg);
    if ((dim & 1) && j == 0) out[(int64_t)b * dim + dim - 1] = 0.f;
}

__global__ void cast_f32_f16_kernel(const float* x, half_t* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (half_t)x[i];
}
__global__ void cast_f16_f32_kernel(const half_t* x, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (float)x[i];
}

__global__ void copy2d_kernel(const half_t* src, half_t* dst, int64_t rows, int c8, int64_t lds_, int64_t ldd) {
    const int64_t total = rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c8;
        const int c = (int)(i - r * c8);
        *reinterpret_cast<h8*>(dst + r * ldd + c * 8) = *reinterpret_cast<const h8*>(src + r * lds_ + c * 8);
    }
}

// src fp32 [B][C][T][HW] -> dst fp16 [B][T][HW][ldc] at channel offset c_off
__global__ void ncthw_to_nthwc_kernel(const float* src, half_t* dst, int B, int C, int T, int64_t HW, int ldc, int c_off,
                                      float scale) {
    const int64_t total = (int64_t)B * C * T * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW;
        int64_t r = i / HW;
        const int t = (int)(r % T);
        r /= T;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        dst[(((int64_t)b * T + t) * HW + p) * ldc + c_off + c] = (half_t)(src[i] * scale);
    }
}

template <bool SRC_F32>
__global__ void nthwc_to_ncthw_kernel(const void* src, float* dst, int B, int C, int T, int64_t HW, int ldc) {
    const int64_t total = (int64_t)B * C * T * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW;
        int64_t r = i / HW;
        const int t = (int)(r % T);
        r /= T;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int64_t si = (((int64_t)b * T + t) * HW + p) * ldc + c;
        dst[i] = SRC_F32 ? reinterpret_cast<const float*>(src)[si] : (float)reinterpret_cast<const half_t*>(src)[si];
    }
}

// ---------------------------------------------------------------------------------------
// DDIM step.  Pass 1: per-sample sums needed by the guidance rescale (unbiased std of v_cond
// and of the guided v).  Pass 2: the update.
// ---------------------------------------------------------------------------------------
struct DdimCoef {
    float sqrt_acp, sqrt_1m_acp, sqrt_a_prev, dir_coef, sigma, scale_ratio, cfg, rescale, cfg_img;
    int is_v, has_uncond, has_noise, has_img;
};

// guided prediction: u + s (c - u), or with an image-only branch i (multi-condition CFG,
// ddim_multiplecond.py:229-234): u + s_img (i - u) + s (c - i)
__device__ __forceinline__ float ddim_guided(float c, float u, float i, const DdimCoef& k) {
    return k.has_img ? u + k.cfg_img * (i - u) + k.cfg * (c - i) : u + k.cfg * (c - u);
}

__global__ void __launch_bounds__(256) ddim_reduce_kernel(const float* vc, const float* vu, const float* vi, double* ws, int64_t n,
                                                          DdimCoef k) {
    __shared__ double red[4][4];
    const int b = blockIdx.y;
    const float* c = vc + (int64_t)b * n;
    const float* u = vu + (int64_t)b * n;
    const float* im = k.has_img ? vi + (int64_t)b * n : u;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = c[i];
        const float g = ddim_guided(a, u[i], im[i], k);
        s0 += a; s1 += a * a; s2 += g; s3 += g * g;
    }
    s0 = vcx_wave_sum(s0); s1 = vcx_wave_sum(s1); s2 = vcx_wave_sum(s2); s3 = vcx_wave_sum(s3);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = s0; red[w][1] = s1; red[w][2] = s2; red[w][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {   // per-block partial, no atomics: ddim_update_kernel adds the blocks in order
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        ws[((int64_t)b * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = t;
    }
}

__global__ void __launch_bounds__(256) ddim_update_kernel(const float* x, const float* vc, const float* vu, const float* vi,
                                                          const float* noise, float* x_prev, float* pred_x0, const double* ws,
                                                          int64_t n, DdimCoef k, int nparts) {
    const int b = blockIdx.y;
    float mix = 1.0f;  // v = v_guided * mix
    if (k.has_uncond && k.rescale > 0.f) {
        const double nn = (double)n;
        double sc = 0, qc = 0, sg = 0, qg = 0;
        for (int j = 0; j < nparts; ++j) {               // fixed order -> bit-reproducible
            const double* pp = ws + ((int64_t)b * nparts + j) * 4;
            sc += pp[0]; qc += pp[1]; sg += pp[2]; qg += pp[3];
        }
        const double var_c = (qc - sc * sc / nn) / (nn - 1.0), var_g = (qg - sg * sg / nn) / (nn - 1.0);
        const float std_c = (float)sqrt(var_c > 0 ? var_c : 0), std_g = (float)sqrt(var_g > 0 ? var_g : 0);
        mix = k.rescale * (std_c / std_g) + (1.0f - k.rescale);
    }
    const int64_t off = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xi = x[off + i];
        float v = vc[off + i];
        if (k.has_uncond) {
            const float u = vu[off + i];
            v = ddim_guided(v, u, k.has_img ? vi[off + i] : u, k) * mix;
        }
        float e_t, x0;
        if (k.is_v) {
            e_t = k.sqrt_acp * v + k.sqrt_1m_acp * xi;
            x0 = k.sqrt_acp * xi - k.sqrt_1m_acp * v;
        } else {
            e_t = v;
            x0 = (xi - k.sqrt_1m_acp * v) / k.sqrt_acp;
        }
        x0 *= k.scale_ratio;
        float xp = k.sqrt_a_prev * x0 + k.dir_coef * e_t;
        if (k.has_noise) xp += k.sigma * noise[off + i];
        pred_x0[off + i] = x0;
        x_prev[off + i] = xp;
    }
}

inline unsigned grid_for(int64_t n, int cap = 4096) {
    int64_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

__global__ void gelu_f16_kernel(const half_t* x, half_t* y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const h8 v = reinterpret_cast<const h8*>(x)[i];
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)vcxgemm::gelu_erf((float)v[e]);
        reinterpret_cast<h8*>(y)[i] = o;
    }
}

// Image pre-processing in front of the OpenCLIP vision tower (reference condition.py:322-329: kornia.geometry.resize(bicubic,
// align_corners=True, antialias) -> (x + 1) / 2 -> normalize(mean, std)), one thread per output pixel.  kornia's anti-aliasing is a
// separable Gaussian over the whole image (mirror border without the edge sample) followed by torch's bicubic interpolation (Keys
// cubic convolution, A = -0.75, neighbour indices clamped): out = sum_i cy_i sum_k gy_k sum_j cx_j sum_l gx_l
// x[mirror(clamp(y0 - 1 + i) + k - ksy / 2)][mirror(clamp(x0 - 1 + j) + l - ksx / 2)] - the blurred image is never materialised
// (4 ksy x 4 ksx taps per output, 336 at 576x1024 -> 224; once per video).  ksy = ksx = 1 (no blur) when nothing shrinks.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__device__ __forceinline__ int mirror_index(int j, int n) {
    j = j < 0 ? -j : j;
    return j > n - 1 ? 2 * (n - 1) - j : j;
}
constexpr int CLIP_MAX_KS = 65;     // blur taps per axis: sigma <= 16, i.e. shrink factors up to 33 (a 7392-pixel side onto 224)
struct ClipPreArgs {
    int B, C, H, W, S, ksy, ksx;
    float gy[CLIP_MAX_KS], gx[CLIP_MAX_KS];     // normalised Gaussian taps (computed in fp64 on the host); {1} when ks = 1
    float mean[4], istd[4];
};
__global__ void clip_preprocess_kernel(const float* __restrict__ x, float* __restrict__ y, ClipPreArgs a) {
    const int64_t total = (int64_t)a.B * a.C * a.S * a.S;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % a.S), oy = (int)((i / a.S) % a.S);
    const int64_t bc = i / ((int64_t)a.S * a.S);
    const int c = (int)(bc % a.C);
    const float* src = x + bc * (int64_t)a.H * a.W;
    // align_corners source coordinate o (in - 1) / (out - 1) as an exact integer quotient + remainder: in fp32 the product would
    // carry ~1e-4 of a pixel at 4K, which on a noisy image is ~1e-3 of the output (torch's own fp32 path has that error; the fp64
    // restatement does not)
    const int qy = oy * (a.H - 1), qx = ox * (a.W - 1), den = a.S - 1;
    const int y0 = qy / den, x0 = qx / den;
    float cy[4], cx[4];
    cubic_coeffs((float)(qy - y0 * den) / (float)den, cy);
    cubic_coeffs((float)(qx - x0 * den) / (float)den, cx);
    const int hy = a.ksy >> 1, hx = a.ksx >> 1;
    float acc = 0.f;
    for (int iy = 0; iy < 4; ++iy) {
        const int yc = min(max(y0 - 1 + iy, 0), a.H - 1);
        for (int k = 0; k < a.ksy; ++k) {
            const float* row = src + (int64_t)mirror_index(yc + k - hy, a.H) * a.W;
            float racc = 0.f;
            for (int ix = 0; ix < 4; ++ix) {
                const int xc = min(max(x0 - 1 + ix, 0), a.W - 1);
                float g = 0.f;
                for (int l = 0; l < a.ksx; ++l) g = fmaf(a.gx[l], row[mirror_index(xc + l - hx, a.W)], g);
                racc = fmaf(cx[ix], g, racc);
            }
            acc = fmaf(cy[iy] * a.gy[k], racc, acc);
        }
    }
    y[i] = ((acc + 1.f) * 0.5f - a.mean[c]) * a.istd[c];
}

static void clip_gauss_taps(int ks, double sigma, float* out) {
    double g[CLIP_MAX_KS], sum = 0.0;
    for (int k = 0; k < ks; ++k) {
        const double d = (double)(k - ks / 2);
        g[k] = exp(-d * d / (2.0 * sigma * sigma));
        sum += g[k];
    }
    for (int k = 0; k < ks; ++k) out[k] = (float)(g[k] / sum);
}

extern "C" int vcx_clip_preprocess_f32(const float* x, float* y, int B, int C, int H, int W, int size, int antialias,
                                       const float* mean_host, const float* std_host, void* stream) {
    VCX_REQUIRE(x && y && mean_host && std_host, "vcx_clip_preprocess_f32: null pointer");
    VCX_REQUIRE(B > 0 && C > 0 && C <= 4 && H > 1 && W > 1 && size > 1, "vcx_clip_preprocess_f32: bad shape B=%d C=%d H=%d W=%d size=%d", B, C, H, W, size);
    VCX_REQUIRE((int64_t)(size - 1) * (H > W ? H : W) < (1ll << 31), "vcx_clip_preprocess_f32: image too large");
    ClipPreArgs a;
    a.B = B; a.C = C; a.H = H; a.W = W; a.S = size;
    const double fy = (double)H / size, fx = (double)W / size;
    a.ksy = a.ksx = 1;
    a.gy[0] = a.gx[0] = 1.f;
    if (antialias && (fy > 1.0 || fx > 1.0) && !(H == size && W == size)) {       // kornia: blur only when an axis shrinks, sigma = (factor - 1) / 2
        const double sy = fmax((fy - 1.0) / 2.0, 0.001), sx = fmax((fx - 1.0) / 2.0, 0.001);
        int ky = (int)fmax(4.0 * sy, 3.0), kx = (int)fmax(4.0 * sx, 3.0);
        ky += 1 - (ky & 1);
        kx += 1 - (kx & 1);
        VCX_REQUIRE(ky <= CLIP_MAX_KS && kx <= CLIP_MAX_KS, "vcx_clip_preprocess_f32: shrink factors beyond 33 are not supported (%dx%d -> %d)", H, W, size);
        VCX_REQUIRE(ky / 2 < H && kx / 2 < W, "vcx_clip_preprocess_f32: blur kernel %dx%d exceeds the image %dx%d", ky, kx, H, W);
        a.ksy = ky; a.ksx = kx;
        clip_gauss_taps(ky, sy, a.gy);
        clip_gauss_taps(kx, sx, a.gx);
    }
    for (int c = 0; c < 4; ++c) {
        a.mean[c] = c < C ? mean_host[c] : 0.f;
        a.istd[c] = c < C ? 1.f / std_host[c] : 1.f;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * C * size * size;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * ((double)B * C * H * W + (double)total));
    hipLaunchKernelGGL(clip_preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, a);
    return vcx_check_launch("vcx_clip_preprocess_f32");
}

// h[n][p][c] += src[n][c][p]: the T2I-adapter feature maps the reference adds behind every third input block
// (openaimodel3d.py:582-585), handed over in the reference's own [(b t), C, h, w] layout.  Never on the ViewCrafter path proper.
__global__ void add_nchw_to_nhwc_kernel(const float* __restrict__ src, half_t* __restrict__ h, int n, int C, int64_t HW) {
    const int64_t total = (int64_t)n * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int64_t p = r % HW;
        const int64_t b = r / HW;
        h[i] = (half_t)((float)h[i] + src[(b * C + c) * HW + p]);
    }
}

extern "C" int vcx_add_nchw_f32_to_nhwc_f16(const float* src, void* h, int n, int C, int64_t HW, void* stream) {
    VCX_REQUIRE(src && h && n > 0 && C > 0 && HW > 0, "vcx_add_nchw_f32_to_nhwc_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * C * HW;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 8.0 * total);
    hipLaunchKernelGGL(add_nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, (half_t*)h, n, C, HW);
    return vcx_check_launch("vcx_add_nchw_f32_to_nhwc_f16");
}

extern "C" int vcx_gelu_f16(const void* x, void* y, int64_t n, void* stream) {
    VCX_REQUIRE(x && y && n > 0 && n % 8 == 0, "vcx_gelu_f16: bad arguments (n must be a multiple of 8)");
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "vcx_gelu_f16: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * n);
    hipLaunchKernelGGL(gelu_f16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const half_t*)x, (half_t*)y, n / 8);
    return vcx_check_launch("vcx_gelu_f16");
}

namespace {

// 2x2 average pool / nearest 2x of a channels-last fp16 image batch [n][H][W][C], eight channels (16 bytes) per thread.  The pool sums in fp32 and
// rounds once (what torch's avg_pool2d does for half inputs); odd H / W drop the last row / column like AvgPool2d(2, 2).
__global__ void avgpool2x2_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int H, int W, int C8, int64_t total) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        int64_t r = i / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int64_t img = r / Ho;
        const half_t* src = x + (((img * H + 2 * yo) * W + 2 * xo) * C8 + c) * 8;
        const h8 a = *reinterpret_cast<const h8*>(src), b = *reinterpret_cast<const h8*>(src + (int64_t)C8 * 8);
        const h8 d = *reinterpret_cast<const h8*>(src + (int64_t)W * C8 * 8), e = *reinterpret_cast<const h8*>(src + ((int64_t)W + 1) * C8 * 8);
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)(((float)a[k] + (float)b[k] + (float)d[k] + (float)e[k]) * 0.25f);
        *reinterpret_cast<h8*>(y + i * 8) = o;
    }
}

__global__ void upsample2x_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int H, int W, int C8, int64_t total) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        int64_t r = i / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int64_t img = r / Ho;
        *reinterpret_cast<h8*>(y + i * 8) = *reinterpret_cast<const h8*>(x + (((img * H + (yo >> 1)) * W + (xo >> 1)) * C8 + c) * 8);
    }
}

}  // namespace

extern "C" int vcx_avgpool2x2_f16(const void* x, void* y, int n, int H, int W, int C, void* stream) {
    VCX_REQUIRE(x && y && n > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "vcx_avgpool2x2_f16: need H, W >= 2 and C %% 8 == 0 (H=%d W=%d C=%d)", H, W, C);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "vcx_avgpool2x2_f16: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * (H / 2) * (W / 2) * (C / 8);
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 2.0 * n * (double)H * W * C + 16.0 * total);
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, s, (const half_t*)x, (half_t*)y, H, W, C / 8, total);
    return vcx_check_launch("vcx_avgpool2x2_f16");
}

extern "C" int vcx_upsample2x_f16(const void* x, void* y, int n, int H, int W, int C, void* stream) {
    VCX_REQUIRE(x && y && n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "vcx_upsample2x_f16: need C %% 8 == 0 (C=%d)", C);
    VCX_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "vcx_upsample2x_f16: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * (2 * H) * (2 * W) * (C / 8);
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 2.0 * n * (double)H * W * C + 16.0 * total);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, s, (const half_t*)x, (half_t*)y, H, W, C / 8, total);
    return vcx_check_launch("vcx_upsample2x_f16");
}

extern "C" int vcx_silu_f32(const float* x, float* y, int64_t n, void* stream) {
    VCX_REQUIRE(x && y && n > 0, "vcx_silu_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 8.0 * n);
    hipLaunchKernelGGL(silu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    return vcx_check_launch("vcx_silu_f32");
}

extern "C" int vcx_timestep_embedding_f32(const int64_t* t, float* out, int B, int dim, float max_period, void* stream) {
    VCX_REQUIRE(t && out && B > 0 && dim >= 2, "vcx_timestep_embedding_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * B * dim);
    const int total = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, out, B, dim,
                       logf(max_period));
    return vcx_check_launch("vcx_timestep_embedding_f32");
}

extern "C" int vcx_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
    VCX_REQUIRE(x && y && n > 0, "vcx_cast_f32_to_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 6.0 * n);
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, (half_t*)y, n);
    return vcx_check_launch("vcx_cast_f32_to_f16");
}

extern "C" int vcx_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream) {
    VCX_REQUIRE(x && y && n > 0, "vcx_cast_f16_to_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 6.0 * n);
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, (const half_t*)x, y, n);
    return vcx_check_launch("vcx_cast_f16_to_f32");
}

extern "C" int vcx_copy2d_f16(const void* src, void* dst, int64_t rows, int cols, int64_t lds_, int64_t ldd, void* stream) {
    VCX_REQUIRE(src && dst && rows > 0 && cols > 0, "vcx_copy2d_f16: bad arguments");
    VCX_REQUIRE(cols % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                "vcx_copy2d_f16: cols/strides must be multiples of 8 and pointers 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * rows * (double)cols);
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * (cols / 8), 16384)), dim3(256), 0, s, (const half_t*)src,
                       (half_t*)dst, rows, cols / 8, lds_, ldd);
    return vcx_check_launch("vcx_copy2d_f16");
}

extern "C" int vcx_ncthw_f32_to_nthwc_f16(const float* src, void* dst, int B, int C, int T, int64_t HW, int ldc, int c_off,
                                          float scale, void* stream) {
    VCX_REQUIRE(src && dst && B > 0 && C > 0 && T > 0 && HW > 0 && c_off >= 0 && c_off + C <= ldc,
                "vcx_ncthw_f32_to_nthwc_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * C * T * HW;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 6.0 * n);
    hipLaunchKernelGGL(ncthw_to_nthwc_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, src, (half_t*)dst, B, C, T, HW, ldc,
                       c_off, scale);
    return vcx_check_launch("vcx_ncthw_f32_to_nthwc_f16");
}

extern "C" int vcx_nthwc_to_ncthw_f32(const void* src, float* dst, int B, int C, int T, int64_t HW, int ldc, int src_f32,
                                      void* stream) {
    VCX_REQUIRE(src && dst && B > 0 && C > 0 && T > 0 && HW > 0 && C <= ldc, "vcx_nthwc_to_ncthw_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * C * T * HW;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 6.0 * n);
    if (src_f32)
        hipLaunchKernelGGL(nthwc_to_ncthw_kernel<true>, dim3(grid_for(n, 16384)), dim3(256), 0, s, src, dst, B, C, T, HW, ldc);
    else
        hipLaunchKernelGGL(nthwc_to_ncthw_kernel<false>, dim3(grid_for(n, 16384)), dim3(256), 0, s, src, dst, B, C, T, HW, ldc);
    return vcx_check_launch("vcx_nthwc_to_ncthw_f32");
}

extern "C" size_t vcx_ddim_ws_bytes(int B, int64_t n) {        // four fp64 partial sums per block of the reduction grid
    return B > 0 && n > 0 ? (size_t)32 * grid_for(n, 256) * (size_t)B : 0;
}

extern "C" int vcx_ddim_step_f32(const float* x, const float* v_cond, const float* v_uncond, const float* noise,
                                 float* x_prev, float* pred_x0, void* ws, size_t ws_bytes, int B, int64_t n,
                                 const float* coef_host, void* stream) {
    float c9[9];
    if (!coef_host) {
        vcx_set_error("vcx_ddim_step_f32: null pointer");
        return VCX_EINVAL;
    }
    for (int i = 0; i < 8; ++i) c9[i] = coef_host[i];
    c9[8] = 0.f;
    return vcx_ddim_step3_f32(x, v_cond, v_uncond, nullptr, noise, x_prev, pred_x0, ws, ws_bytes, B, n, c9, stream);
}

extern "C" int vcx_ddim_step3_f32(const float* x, const float* v_cond, const float* v_uncond, const float* v_img,
                                  const float* noise, float* x_prev, float* pred_x0, void* ws, size_t ws_bytes, int B,
                                  int64_t n, const float* coef_host, void* stream) {
    VCX_REQUIRE(x && v_cond && x_prev && pred_x0 && ws && coef_host, "vcx_ddim_step_f32: null pointer");
    VCX_REQUIRE(!v_img || v_uncond, "vcx_ddim_step3_f32: v_img needs v_uncond");
    VCX_REQUIRE(B > 0 && B <= 65535 && n > 1, "vcx_ddim_step_f32: bad sizes");
    VCX_REQUIRE(((uintptr_t)ws & 7) == 0, "vcx_ddim_step_f32: ws must be 8-byte aligned");
    VCX_REQUIRE(ws_bytes >= vcx_ddim_ws_bytes(B, n), "vcx_ddim_step_f32: workspace of %zu bytes, vcx_ddim_ws_bytes(%d, %lld) = %zu", ws_bytes, B,
                (long long)n, vcx_ddim_ws_bytes(B, n));
    DdimCoef k;
    const float a_prev = coef_host[2], sigma = coef_host[3];
    k.sqrt_acp = coef_host[0];
    k.sqrt_1m_acp = coef_host[1];
    k.sqrt_a_prev = sqrtf(a_prev);
    const float dir2 = 1.0f - a_prev - sigma * sigma;
    k.dir_coef = sqrtf(dir2 > 0.f ? dir2 : 0.f);
    k.sigma = sigma;
    k.scale_ratio = coef_host[4];
    k.cfg = coef_host[5];
    k.rescale = coef_host[6];
    k.is_v = coef_host[7] != 0.f;
    k.has_uncond = v_uncond != nullptr;
    k.has_img = v_img != nullptr;
    k.cfg_img = coef_host[8];
    k.has_noise = (noise != nullptr) && sigma != 0.f;
    hipStream_t s = (hipStream_t)stream;
    VcxProfScope prof(VCX_FAM_ELT, s, 0.0, 4.0 * B * (double)n * 8);
    const unsigned gx = grid_for(n, 256);
    if (k.has_uncond && k.rescale > 0.f) {
        hipLaunchKernelGGL(ddim_reduce_kernel, dim3(gx, B), dim3(256), 0, s, v_cond, v_uncond, v_img, (double*)ws, n, k);
        int rc = vcx_check_launch("vcx_ddim_step_f32(reduce)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ddim_update_kernel, dim3(gx, B), dim3(256), 0, s, x, v_cond, v_uncond, v_img, noise, x_prev, pred_x0,
                       (const double*)ws, n, k, (int)gx);
    return vcx_check_launch("vcx_ddim_step_f32");
}
