// Shared between the generic (register-staged) and the DMA (buffer_load ... lds) GEMM kernels.
#pragma once
#include "vcx_common.h"

namespace vcxgemm {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct GemmArgs {
    const half_t* A;
    const half_t* W;
    void* C;
    const float* bias;
    const float* rowadd;
    const half_t* R;
    int64_t lda;
    int M, N, K;
    int ldw, ldc, ldr;
    int in_h, in_w, out_h, out_w, cin, kh, kw, stride, pad_h, pad_w, ups;
    int rowadd_div;
    int rowadd_ld;      // elements between consecutive rows of rowadd (>= N)
    int flags;
    float alpha;
    int tiles_m, tiles_n;
    unsigned a_bytes, w_bytes;   // operand extents for the DMA kernel's buffer descriptors (whole problem, not the launch's rows)
    unsigned c_bytes, r_bytes;   // output / residual extents (the DMA kernel's epilogue addresses them through descriptors too)
    int m_begin;   // first output row of this launch (tail split of large-tile launches); rows are < M
    const float* ln_stats;    // VCX_GEMM_LNFOLD[_T]: (mean, rstd) pairs
    const float* ln_colsum;   // VCX_GEMM_LNFOLD[_T]: row sums of the folded weight
    float* colstats;          // VCX_GEMM_COLSTATS: (mean, M2) per 64-row strip and output column
    int64_t ldcs;             // columns per strip of colstats (>= N: the buffer may hold a concatenated partner's columns too)
    // vcx_gemm_units_f16 (weight-stationary kernel only): one weight / bias set per unit_rows consecutive rows; 0 = one set for all
    int unit_rows, units;
    int64_t w_unit_stride, bias_unit_stride;      // elements between consecutive units' weights / biases
    float* rowstats;          // VCX_GEMM_ROWSTATS (gemm_ws320_pipe_kernel only): (mean, rstd) of every output row
    float rowstats_eps;
    // K tail of a convolution (gemm_dma_kernel<..., TAIL>): the last (k2 + k3) / 64 K-steps read rows of A2, then A3, linearly
    const half_t* A2;
    const half_t* A3;
    int64_t lda2, lda3;
    int k2, k3;
    unsigned a2_bytes, a3_bytes;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // element offset of a 16-byte chunk inside a [rows][64] fp16 tile
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// exact-erf GELU (F.gelu default): gelu(x) = x Phi(x) with the normal tail Phi(-|x|) = exp2(q(|x|)), q a degree-6 polynomial.
// log2 of the Gaussian tail is almost a parabola, so six Horner steps reproduce it on [0, 9] (weighted minimax fit against
// scipy's erfc; beyond 9 the tail is < 1e-19 and |x| is clamped): evaluated in fp32, max |gelu error| is 3.7e-7 over EVERY finite
// fp16 input and the fp16 result is within one unit in the last place of the correctly rounded value everywhere
// (tests/test_kernels_gpu.py::test_gelu_every_fp16_input...) - the accuracy of the Abramowitz-Stegun 7.1.26 form used in rounds
// 1-2 (4.5e-7; the A/B builds of rounds 3-5 - that form, the compare / select tail, other panel heights of the tile walk - are in the
// history: commit e530759 and before), with ONE transcendental (v_exp) instead of two (v_rcp + v_exp), 10 instead of 14 VALU operations and no
// reciprocal at the head of the dependent chain.  The GEGLU epilogue runs this once per output element and was 28 % of the
// level-0 GEGLU layer (profiles/r02_experiments.md section 9); same-box effect: GEGLU layers -4 ... -6 % (profiles/r03_experiments.md).
// in pieces, so that a caller can spread them over several MFMA shadows (gemm_ws320_geglu_kernel); gelu_erf() below is their
// composition - ONE definition of the arithmetic
constexpr float GELU_Q[7] = {3.3093042e-05f, -7.6922239e-04f, 8.0807274e-03f, -5.3412125e-02f, -4.5877096e-01f, -1.1512017e+00f, -9.9999309e-01f};
template <int I>
__device__ __forceinline__ float gelu_q_step(float q, float a) { return __builtin_fmaf(q, a, GELU_Q[I]); }       // Horner step I = 1 .. 6, q0 = GELU_Q[0]
__device__ __forceinline__ float gelu_clamp(float x) { return fminf(fabsf(x), 9.0f); }
__device__ __forceinline__ float gelu_finish(float x, float a, float mx, float e) {       // e = exp2(q6) = Phi(-|x|), mx = max(x, 0)
    // x Phi(x) = max(x, 0) - |x| Phi(-|x|): one max and one multiply-add instead of subtract / compare / select / multiply - three
    // VALU issue slots fewer per GEGLU output and one rounding instead of two (max error over every fp16 input 2.8e-7 against
    // 3.7e-7, still within one fp16 unit in the last place everywhere).  The clamped |x| serves: beyond 9 the product is < 1e-17.
    return __builtin_fmaf(-a, e, mx);
}
__device__ __forceinline__ void gelu_erf_head(float x, float& a, float& mx, float& q) {
    a = gelu_clamp(x);
    mx = __builtin_fmaxf(x, 0.f);
    q = gelu_q_step<3>(gelu_q_step<2>(gelu_q_step<1>(GELU_Q[0], a), a), a);
}
__device__ __forceinline__ float gelu_erf_tail(float x, float a, float mx, float q) {
    q = gelu_q_step<6>(gelu_q_step<5>(gelu_q_step<4>(q, a), a), a);
    return gelu_finish(x, a, mx, __builtin_amdgcn_exp2f(q));
}
__device__ __forceinline__ float gelu_erf(float x) {
    float a, mx, q;
    gelu_erf_head(x, a, mx, q);
    return gelu_erf_tail(x, a, mx, q);
}


// XCD-aware persistent tile walk: tile ids congruent mod 8 form a contiguous band of (tile_m, tile_n).
__device__ __forceinline__ void tile_coords(int t, int ntiles, int tiles_n, int& tm, int& tn) {
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xcd = t & 7, idx = t >> 3;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // Wide layers (>= 16 column tiles: the GEGLU projections of levels 1-3): inside a band, tiles are numbered down 8-row
    // panels (tm fastest), so the 32 tiles an XCD has in flight form an 8 x 4 block that shares 8 activation row tiles and
    // 4 weight tiles in its L2 instead of 1-2 and 32 (+8-10 % on those layers; with few column tiles the row-major walk,
    // which completes whole output rows at a time, is 1-2 % better)
    if (tiles_n < 16) {
        tn = vid % tiles_n;
        tm = vid / tiles_n;
        return;
    }
    const int tiles_m = ntiles / tiles_n, full = tiles_m >> 3, split = full * 8 * tiles_n;
    if (vid < split) {
        const int pnl = vid / (8 * tiles_n), r = vid - pnl * 8 * tiles_n;
        tn = r >> 3;
        tm = pnl * 8 + (r & 7);
    } else {
        const int rem = tiles_m - full * 8, r = vid - split;
        tn = r / rem;
        tm = full * 8 + r % rem;
    }
}

int persistent_grid(int ntiles, int blocks_per_cu = 2);
int launch_dma(GemmArgs& a, int cfg, bool conv, bool geglu, bool f32, hipStream_t s);      // (a.k2 + a.k3 > 0: the K-tail instantiations)
int launch_ws320_geglu(GemmArgs& a, hipStream_t s);  // ... GEGLU projection, K = 320, N % 256 == 0
int launch_ws320_lnfold(GemmArgs& a, hipStream_t s); // ... LayerNorm-folded projection (VCX_GEMM_LNFOLD), K = 320, N % 64 == 0
int launch_ws320_units(GemmArgs& a, hipStream_t s);  // ... with one weight / bias set per unit of rows (vcx_gemm_units_f16)
int launch_ws320(GemmArgs& a, hipStream_t s);        // gemm_ws.hip: weight-stationary linear layer, N = K = 320 (plain / COLSTATS epilogues)   // a.flags & VCX_GEMM_LNFOLD[_T] selects the folded-LayerNorm epilogue

}  // namespace vcxgemm
