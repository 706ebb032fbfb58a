// REJECTED in round 4 (kept for the record; not part of libvcx.so): temporal attention + output projection + residual as one kernel.
//
// Built, correct at the first run (7 shapes vs fp32 torch, in place, bit-reproducible; the whole model suite with it switched on) - and
// SLOWER than the pair it replaces, same box, interleaved (profiles/r04i_tattn_fused_ab.txt):
//     level 0 (C = 320, 5 heads, B = 2)     attention 0.234 ms   attention + K = 320 GEMM 0.459 ms   fused 0.588 ms
//     init_attn (8 heads -> 320, B = 1)     0.188                0.342                               0.435
//     320x512x25 level 0                    0.072                0.136                               0.190
// Why: the output tile of a pixel ([320, T] fp32 = 160 registers per lane) leaves no room to prefetch the next head's q / k / v rows,
// the Wo slice of a head is staged through LDS for the block's 8 pixels behind two block barriers per head, and 156 KB of LDS allow one
// block per CU - so every head pays its HBM round trip, its Wo load and its barrier skew in sequence (13 us per head against ~2 us of
// MFMA work).  A version that hides them (Wo by LDS-DMA with a source-side swizzle under the attention phase, q / k / v prefetched
// under the projection) is bounded by the 1.47 GB the fused kernel still moves: ~0.33 ms against 0.459 ms = at most 1.3 ms per DDIM
// step.  Not pursued.  To build it again: paste the kernel and the entry point back into csrc/attention.hip, declare
// vcx_attn_temporal_proj_d64_f16 in include/vcx.h (git show 1bbd792 has the binding, the ops wrapper and the test).
#include "../../viewcrafter_amd/csrc/vcx_common.h"

// =======================================================================================
// Temporal attention + output projection + residual in one kernel (round 4): out = x + bias + concat_h(O_h) Wo^T.
// The unfused pair writes O [tokens, inner] and reads it back in an HBM-bound K = inner GEMM; here a wave keeps one pixel's
// out^T [C_out, T] in accumulators (C_out = 32 NM <= 320: 16 NM registers) while it walks the heads: per head the attention of
// tattn_d64_kernel (same code, same LDS patch), then O_h^T - still in the accumulator layout, used as the B operand exactly like P
// in the PV product - times the head's slice of Wo, whose rows the block stages in LDS once per head for its 8 pixels (A fragments
// gathered with the accumulator's d permutation: two 8-byte reads).  HBM traffic: qkv + residual in, out written once.
// =======================================================================================
struct TAttnProjArgs {
    const half_t* qkv;
    half_t* out;
    const half_t* wo;        // [C_out][heads * 64]
    const float* bias;       // [C_out] or null
    const half_t* res;       // [tokens][ldr] or null
    int B, T, heads, cout;
    int64_t P, ld, ldo, ldr;
    int k_off, v_off;
    float scale;
    int64_t npix;            // B * P
};

template <int NM>
__global__ void __launch_bounds__(512) tattn_proj_d64_kernel(TAttnProjArgs p) {
    constexpr int VLD = 72;
    constexpr int WAVES = 8;
    constexpr int WLD = 72;                                            // Wo slice row pitch (halfs): 144 B, conflict-free 8-byte gathers
    __shared__ __attribute__((aligned(16))) half_t sVt[WAVES][3 * 32 * VLD];
    __shared__ __attribute__((aligned(16))) half_t sW[NM * 32 * WLD];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const int64_t pixg = (int64_t)blockIdx.x * WAVES + wave;
    const bool pvalid = pixg < p.npix;                                 // (an inactive wave still takes part in the block barriers)
    const int64_t pixc = pvalid ? pixg : p.npix - 1;
    const int64_t pix = pixc % p.P;
    const int b = (int)(pixc / p.P);
    const int64_t fstride = p.P * p.ld;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int inner = p.heads * 64;
    half_t* sv = sVt[wave];
    half_t* sq = sv + 32 * VLD;
    half_t* sk = sv + 64 * VLD;

    f16v acc[NM];                                                      // out^T[c = 32 m + (r&3) + 8 (r>>2) + 4 hi][frame lq]
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = zero16;

    for (int h = 0; h < p.heads; ++h) {
        // ---- this head's Wo slice [C_out][64] -> LDS (all 512 threads; 16-byte chunks, 8 per row)
        __syncthreads();                                               // every wave is done with the previous head's slice
        for (int c = tid; c < NM * 32 * 8; c += 512) {
            const int row = c >> 3, ch = c & 7;
            *reinterpret_cast<h8*>(sW + row * WLD + ch * 8) = *reinterpret_cast<const h8*>(p.wo + (int64_t)row * inner + h * 64 + ch * 8);
        }
        // ---- attention of (pixel, head): identical to tattn_d64_kernel
        const half_t* base = p.qkv + ((int64_t)b * p.T * p.P + pix) * p.ld + h * 64;
        h8 qf[4], kf[4];
        {
            h8 q4[4], k4[4], v4[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = (lane >> 3) + 8 * it, ch = lane & 7;
                const half_t* src = base + (int64_t)row * fstride + ch * 8;
                const bool ok = row < p.T;
                q4[it] = ok ? *reinterpret_cast<const h8*>(src) : zero8;
                k4[it] = ok ? *reinterpret_cast<const h8*>(src + p.k_off) : zero8;
                v4[it] = ok ? *reinterpret_cast<const h8*>(src + p.v_off) : zero8;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = (lane >> 3) + 8 * it, ch = lane & 7;
                *reinterpret_cast<h8*>(sq + row * VLD + ch * 8) = q4[it];
                *reinterpret_cast<h8*>(sk + row * VLD + ch * 8) = k4[it];
                *reinterpret_cast<h8*>(sv + row * VLD + ch * 8) = v4[it];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = *reinterpret_cast<const h8*>(sq + lq * VLD + s * 16 + hi * 8);
                kf[s] = *reinterpret_cast<const h8*>(sk + lq * VLD + s * 16 + hi * 8);
            }
        }
        f16v sacc = zero16;
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], s == 0 ? zero16 : sacc, 0, 0, 0);
        const float c = p.scale * 1.4426950408889634f;
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.T) sacc[r] = -1e30f;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * c;
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c, -mx));
            sacc[r] = e;
            l += e;
        }
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        h8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s][j] = (half_t)(sacc[8 * s + j] * inv);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        f16v oacc[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int d = db * 32 + lq;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 vf;
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = sv[(16 * s + (j & 3) + 8 * (j >> 2) + 4 * hi) * VLD + d];
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], s == 0 ? zero16 : oacc[db], 0, 0, 0);
            }
        }
        // ---- out^T += Wo_h O_h^T.  O_h^T is rounded to fp16 here exactly where the unfused path stores it.  k-step (db, s) feeds
        // the accumulator registers 8 s .. 8 s + 7 of oacc[db] as the B operand: logical d = 32 db + 16 s + (j&3) + 8 (j>>2) + 4 hi.
        h8 of[4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) of[db * 2 + s][j] = (half_t)oacc[db][8 * s + j];
        __syncthreads();                                               // the Wo slice is in LDS
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const half_t* wrow = sW + (m * 32 + lq) * WLD + 4 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h4 lo = *reinterpret_cast<const h4*>(wrow + 16 * ks), up = *reinterpret_cast<const h4*>(wrow + 16 * ks + 8);
                const h8 wf = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, of[ks], acc[m], 0, 0, 0);
            }
        }
    }
    if (!pvalid || lq >= p.T) return;
    // ---- epilogue: + bias + residual, one rounding; row = frame lq of this pixel.  Channel group g = 4 m + (r >> 2) holds the four
    // channels 8 g + 4 hi .. + 3 in this lane: lanes l and l + 32 together hold 8 consecutive channels of the row (16 bytes).
    const int64_t row = ((int64_t)b * p.T + lq) * p.P + pix;
    half_t* orow = p.out + row * p.ldo;
    const half_t* rrow = p.res ? p.res + row * p.ldr : nullptr;
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c0 = 32 * m + 8 * gq + 4 * hi;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[m][gq * 4 + r];
            if (p.bias) {
                const f4 bv = *reinterpret_cast<const f4*>(p.bias + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bv[r];
            }
            if (rrow) {
                const h4 rv = *reinterpret_cast<const h4*>(rrow + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            }
            *reinterpret_cast<h4*>(orow + c0) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
}

extern "C" int vcx_attn_temporal_proj_d64_f16(const void* qkv, void* out, const void* wo, const float* bias, const void* residual, int B,
                                              int T, int64_t P, int heads, int cout, int64_t ld, int k_off, int v_off, int64_t ldo,
                                              int64_t ldr, float scale, void* stream) {
    VCX_REQUIRE(qkv && out && wo, "vcx_attn_temporal_proj_d64_f16: null pointer");
    VCX_REQUIRE(B > 0 && T > 0 && T <= 32 && P > 0 && heads > 0, "vcx_attn_temporal_proj_d64_f16: need 0 < T <= 32 (T=%d)", T);
    VCX_REQUIRE(cout % 32 == 0 && cout >= 32 && cout <= 320, "vcx_attn_temporal_proj_d64_f16: C_out must be a multiple of 32 up to 320 (%d): "
                "the output tile lives in accumulators; wider layers take vcx_attn_temporal_d64_f16 + vcx_gemm_f16", cout);
    VCX_REQUIRE(ld % 8 == 0 && ldo % 4 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && (!residual || ldr % 4 == 0),
                "vcx_attn_temporal_proj_d64_f16: strides/offsets must be multiples of 8 (ldo / ldr: 4)");
    VCX_REQUIRE((((uintptr_t)qkv | (uintptr_t)wo) & 15) == 0 && (((uintptr_t)out | (uintptr_t)residual) & 7) == 0 && ((uintptr_t)bias & 15) == 0,
                "vcx_attn_temporal_proj_d64_f16: pointer alignment");
    TAttnProjArgs a;
    a.qkv = (const half_t*)qkv; a.out = (half_t*)out; a.wo = (const half_t*)wo; a.bias = bias; a.res = (const half_t*)residual;
    a.B = B; a.T = T; a.heads = heads; a.cout = cout; a.P = P; a.ld = ld; a.ldo = ldo; a.ldr = ldr;
    a.k_off = k_off; a.v_off = v_off; a.scale = scale;
    a.npix = (int64_t)B * P;
    hipStream_t s = (hipStream_t)stream;
    const double tok = (double)a.npix * T;
    VcxProfScope prof(VCX_FAM_TATTN, s, 4.0 * a.npix * heads * (double)T * T * 64 + 2.0 * tok * cout * heads * 64,
                      2.0 * tok * (3.0 * heads * 64 + 2.0 * cout));
    const int64_t nblk = (a.npix + 7) / 8;
    VCX_REQUIRE(nblk < (1ll << 31), "vcx_attn_temporal_proj_d64_f16: grid too large");
    const dim3 grid((unsigned)nblk);
    switch (cout / 32) {
#define TP_CASE(n) case n: hipLaunchKernelGGL(tattn_proj_d64_kernel<n>, grid, dim3(512), 0, s, a); break;
        TP_CASE(1) TP_CASE(2) TP_CASE(3) TP_CASE(4) TP_CASE(5) TP_CASE(6) TP_CASE(7) TP_CASE(8) TP_CASE(9) TP_CASE(10)
#undef TP_CASE
    }
    return vcx_check_launch("vcx_attn_temporal_proj_d64_f16");
}


/* the GPU test it passed (tests/test_kernels_gpu.py at that commit):
@pytest.mark.parametrize("B,T,P,heads,cout", [(2, 25, 300, 5, 320), (1, 16, 64, 8, 320), (1, 1, 9, 1, 64), (2, 32, 17, 2, 128), (1, 3, 1000, 4, 256),
                                              (1, 25, 13, 3, 192), (1, 7, 8, 1, 32)])
def test_temporal_attention_with_fused_output_projection(B, T, P, heads, cout):
    """vcx_attn_temporal_proj_d64_f16: x + bias + temporal_attention(qkv) Wo^T in one launch (reference attention.py:81-126 + to_out.0
    + the residual of :243-245) - against fp32 torch, against the unfused pair (attention kernel -> GEMM), in place on the residual,
    with pixel counts that leave waves of the last block idle, T = 1 and T = 32, 1 to 8 heads, every supported output width class."""
    from viewcrafter_amd import ops
    C = heads * 64
    tokens = B * T * P
    qkv = rnd(tokens, 3 * C, seed=801).to(DEV).half()
    wo = (rnd(cout, C, seed=802) / math.sqrt(C)).to(DEV).half()
    bias = rnd(cout, seed=803).to(DEV)
    res = rnd(tokens, cout, seed=804).to(DEV).half()
    scale = 0.125
    x = qkv.float().view(B, T, P, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)
    att = (torch.softmax(x[0] @ x[1].transpose(-1, -2) * scale, -1) @ x[2]).permute(0, 3, 1, 2, 4).reshape(tokens, C)
    ref = att.half().float() @ wo.float().t() + bias + res.float()
    out = ops.temporal_attn_proj(qkv, wo, bias, res, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, scale=scale)
    check(out, ref, tol=3e-3, name="temporal attention + projection")
    o = torch.empty((tokens, C), dtype=torch.float16, device=DEV)
    ops.temporal_attn(qkv, o, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, ldo=C, scale=scale)
    unfused = ops.linear(o, wo, bias, residual=res)
    assert rel_l2(out, unfused) <= 1e-3
    for _ in range(5):
        assert torch.equal(ops.temporal_attn_proj(qkv, wo, bias, res, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, scale=scale), out)
    inplace = res.clone()
    ops.temporal_attn_proj(qkv, wo, bias, inplace, out=inplace, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, scale=scale)
    assert torch.equal(inplace, out)
    nobias = ops.temporal_attn_proj(qkv, wo, None, None, B=B, T=T, P=P, heads=heads, ld=3 * C, k_off=C, v_off=2 * C, scale=scale)
    check(nobias, att.half().float() @ wo.float().t(), tol=3e-3, name="temporal attention + projection, no bias / residual")


*/
