// GEMM / implicit-GEMM convolution engine for gfx950 (CDNA4), fp16 in, fp32 accumulate.
//
//   out[m, n] = epilogue(alpha * sum_k X[m, k] * W[n, k])
//
// Tile: BM=128 output rows x BN (128 or 160) output columns x BK=64, 256 threads = 4 waves
// arranged 2 (m) x 2 (n); a wave owns 64 x BN/2 outputs as 4 x (BN/32) fragments of
// v_mfma_f32_16x16x32_f16.  The MFMA is issued with the WEIGHT fragment as operand A and the
// ACTIVATION fragment as operand B, so D[i][j] = out[m = j][n = i]: a lane then holds 4
// consecutive output columns of one output row (row = lane&15, cols = 4*(lane>>4)+r) and the
// epilogue stores 8 contiguous bytes per fragment instead of four 2-byte scatters.
//
// Staging is global -> registers -> LDS (ds_write_b128) with the 16-byte chunk index XORed by
// ((row>>1)&7): 128-byte rows would otherwise put every lane of a ds_read_b128 group on the
// same bank slots.  LDS is double buffered; the next tile's global loads are issued before the
// current tile's MFMAs and written to the other buffer after them (one barrier per K-step).
//
// X rows are either linear (mode 0) or an im2col gather over a channels-last image (mode 1):
// every 16-byte load is 8 consecutive input channels of one tap, so a 3x3 / (3,1,1) / strided /
// nearest-upsampled convolution is the same main loop with a different address function and
// zero fill outside the image.
#include "vcx_common.h"

#include "gemm_args.h"
#include <stdlib.h>

using namespace vcxgemm;

namespace {

template <int BN, bool CONV, bool GEGLU, bool OUT_F32>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_kernel(GemmArgs p) {
    constexpr int NFRAG = BN / 32;          // 16-wide n fragments per wave
    constexpr int WROWS = BN / 32;          // weight rows staged per thread (BN*8 chunks / 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sX = reinterpret_cast<half_t*>(smem_raw);              // [2][BM*BK]
    half_t* sW = sX + 2 * BM * BK;                                  // [2][BN*BK]

    // ---- persistent blocks: block b walks tiles b, b+G, b+2G, ... (G = gridDim.x, a multiple of 8 whenever a block
    // owns more than one tile).  Tile ids are mapped XCD-aware: ids congruent mod 8 (= the XCD the dispatcher puts this
    // block on) form a contiguous band of (tile_m, tile_n), so tiles sharing an activation panel hit the same L2.
    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    auto tile_coords = [&](int t, int& tm, int& tn) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7;
        const int xcd = t & 7, idx = t >> 3;
        const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        tn = vid % p.tiles_n;
        tm = vid / p.tiles_n;
    };

    const int tid = threadIdx.x;
    const int chunk = tid & 7;   // 16-byte chunk inside the BK=64 slice
    const int r0 = tid >> 3;     // 0..31

    // ---- per-thread gather state of the tile being LOADED (4 activation rows, WROWS weight rows)
    int64_t xbase[4];  // linear: element offset of the row; conv: image base pixel index
    int xoy[4], xox[4];
    bool xvalid[4];
    int ci = 0, ky = 0, kx = 0;   // conv tap walker: k = (ky*kw + kx)*cin + ci, advanced by BK per K-step
    const half_t* wptr[WROWS];
    bool wvalid[WROWS];
    auto init_load = [&](int t) {
        int tile_m, tile_n;
        tile_coords(t, tile_m, tile_n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = tile_m * BM + r0 + 32 * i;
            xvalid[i] = m < p.M;
            if (CONV) {
                const int hw = p.out_h * p.out_w;
                const int mm = xvalid[i] ? m : 0;
                const int img = mm / hw;
                const int rem = mm - img * hw;
                const int oy = rem / p.out_w;
                const int ox = rem - oy * p.out_w;
                xbase[i] = (int64_t)img * p.in_h * p.in_w;
                xoy[i] = oy * p.stride - p.pad_h;
                xox[i] = ox * p.stride - p.pad_w;
            } else {
                xbase[i] = (int64_t)m * p.lda;
                xoy[i] = xox[i] = 0;
            }
        }
        ci = chunk * 8; ky = 0; kx = 0;
        if (CONV && !(p.flags & VCX_GEMM_CONV_SLABK)) {
            while (ci >= p.cin) {
                ci -= p.cin;
                if (++kx == p.kw) { kx = 0; ++ky; }
            }
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int n = tile_n * BN + r0 + 32 * i;
            wvalid[i] = n < p.N;
            wptr[i] = p.W + (int64_t)(wvalid[i] ? n : 0) * p.ldw + chunk * 8;
        }
    };

    h8 xreg[4];
    h8 wreg[WROWS];
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lim_h = p.in_h << p.ups, lim_w = p.in_w << p.ups;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK + chunk * 8;
        const bool kin = k0 < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h8 v = zero8;
            if (CONV) {
                const int iy = xoy[i] + ky, ix = xox[i] + kx;
                if (kin && xvalid[i] && iy >= 0 && iy < lim_h && ix >= 0 && ix < lim_w) {
                    const int64_t pix = xbase[i] + (int64_t)(iy >> p.ups) * p.in_w + (ix >> p.ups);
                    v = *reinterpret_cast<const h8*>(p.A + pix * p.lda + ci);
                }
            } else {
                if (kin && xvalid[i]) v = *reinterpret_cast<const h8*>(p.A + xbase[i] + k0);
            }
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            h8 v = zero8;
            if (kin && wvalid[i]) v = *reinterpret_cast<const h8*>(wptr[i] + (int64_t)kt * BK);
            wreg[i] = v;
        }
        if (CONV) {  // advance the tap walker to the next K-step
            if (p.flags & VCX_GEMM_CONV_SLABK) {      // k = ((c / 64) * taps + tap) * 64 + c % 64: a K-step is one tap of one slab
                if (++kx == p.kw) {
                    kx = 0;
                    if (++ky == p.kh) { ky = 0; ci += BK; }
                }
            } else {
                ci += BK;
                while (ci >= p.cin) {
                    ci -= p.cin;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        half_t* dx = sX + buf * BM * BK;
        half_t* dw = sW + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<h8*>(dx + lds_off(r0 + 32 * i, chunk)) = xreg[i];
#pragma unroll
        for (int i = 0; i < WROWS; ++i) *reinterpret_cast<h8*>(dw + lds_off(r0 + 32 * i, chunk)) = wreg[i];
    };

    // ---- wave / lane decomposition
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int lr = lane & 15, lg = lane >> 4;

    f4 acc[NFRAG][4];
#pragma unroll
    for (int a = 0; a < NFRAG; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    // ---- software pipeline over the flattened (tile, k-step) sequence of this block: the loads of step s+1 are in
    // flight during the MFMAs (and, at a tile boundary, the epilogue) of step s, so neither the first-load latency nor
    // the epilogue of a tile is exposed after the block's first tile.
    const int nk = (p.K + BK - 1) / BK;
    int ltile = blockIdx.x, lkt = 0;     // load cursor
    int ctile = blockIdx.x, ckt = 0;     // compute cursor
    int tile_m, tile_n;
    tile_coords(ctile, tile_m, tile_n);
    init_load(ltile);
    load_tile(0);
    store_tile(0);
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
        if (more) load_tile(lkt);
        const half_t* cx = sX + cur * BM * BK;
        const half_t* cw = sW + cur * BN * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 wf[NFRAG], xf[4];
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
                wf[a] = *reinterpret_cast<const h8*>(cw + lds_off(wn * (BN / 2) + a * 16 + lr, kk * 4 + lg));
#pragma unroll
            for (int b = 0; b < 4; ++b)
                xf[b] = *reinterpret_cast<const h8*>(cx + lds_off(wm * 64 + b * 16 + lr, kk * 4 + lg));
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (ckt == nk - 1) {
    // ---- epilogue.  acc[a][b][r] = out[m][n], m = tile_m*BM + wm*64 + b*16 + lr,
    //      n = tile_n*BN + wn*(BN/2) + a*16 + lg*4 + r.
    // Fast path (N % 4 == 0, no GEGLU): all residual / bias loads are issued together, waited for once, then the
    // math and the 8-byte stores follow; addresses of out-of-range rows/columns are clamped so the loads need no branch.
            if (!GEGLU && (p.N & 3) == 0) {
                const int mbase = tile_m * BM + wm * 64 + lr;
                const int nbase = tile_n * BN + wn * (BN / 2) + lg * 4;
                h4 rr[NFRAG][4];
                f4 bv[NFRAG];
                if (flags & VCX_GEMM_RESIDUAL) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int mc = min(mbase + b * 16, p.M - 1);
#pragma unroll
                        for (int a = 0; a < NFRAG; ++a) {
                            const int nc = min(nbase + a * 16, p.N - 4);
                            rr[a][b] = *reinterpret_cast<const h4*>(p.R + (int64_t)mc * p.ldr + nc);
                        }
                    }
                }
                if (flags & VCX_GEMM_BIAS_N) {
#pragma unroll
                    for (int a = 0; a < NFRAG; ++a) bv[a] = *reinterpret_cast<const f4*>(p.bias + min(nbase + a * 16, p.N - 4));
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int m = mbase + b * 16;
                    const int mc = min(m, p.M - 1);
                    const float bm = (flags & VCX_GEMM_BIAS_M) ? p.bias[mc] : 0.f;
                    const float* radd = (flags & VCX_GEMM_ROWADD) ? p.rowadd + (int64_t)(mc / p.rowadd_div) * p.rowadd_ld : nullptr;
#pragma unroll
                    for (int a = 0; a < NFRAG; ++a) {
                        const int n0 = nbase + a * 16;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha + bm;
                        if (flags & VCX_GEMM_BIAS_N) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += bv[a][r];
                        }
                        if (radd) {
                            const f4 rv = *reinterpret_cast<const f4*>(radd + min(n0, p.N - 4));
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += rv[r];
                        }
                        if (flags & VCX_GEMM_RESIDUAL) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)rr[a][b][r];
                        }
                        if (m < p.M && n0 < p.N) {
                            if (OUT_F32) {
                                float* dst = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n0;
                                *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]};
                            } else {
                                half_t* dst = reinterpret_cast<half_t*>(p.C) + (int64_t)m * p.ldc + n0;
                                *reinterpret_cast<h4*>(dst) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            }
                        }
                    }
                }
            } else {
    // generic path (GEGLU, or N not a multiple of 4): per-fragment guards, scalar tail
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int m = tile_m * BM + wm * 64 + b * 16 + lr;
        if (m >= p.M) continue;
        const float* radd = (flags & VCX_GEMM_ROWADD) ? p.rowadd + (int64_t)(m / p.rowadd_div) * p.rowadd_ld : nullptr;
        if (GEGLU) {
#pragma unroll
            for (int a = 0; a < NFRAG / 2; ++a) {
                const int nx = tile_n * BN + wn * (BN / 2) + a * 16 + lg * 4;  // packed-space column of x
                const int ng = nx + 32;                                        // its gate
                const int j = tile_n * (BN / 2) + wn * (BN / 4) + a * 16 + lg * 4;  // output column
                if (nx + 4 > p.N) continue;
                half_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xv = acc[a][b][r] * p.alpha, gv = acc[a + NFRAG / 2][b][r] * p.alpha;
                    if (flags & VCX_GEMM_BIAS_N) { xv += p.bias[nx + r]; gv += p.bias[ng + r]; }
                    o[r] = (half_t)(xv * gelu_erf(gv));
                }
                half_t* dst = reinterpret_cast<half_t*>(p.C) + (int64_t)m * p.ldc + j;
                *reinterpret_cast<h4*>(dst) = h4{o[0], o[1], o[2], o[3]};
            }
        } else {
#pragma unroll
            for (int a = 0; a < NFRAG; ++a) {
                const int n0 = tile_n * BN + wn * (BN / 2) + a * 16 + lg * 4;
                if (n0 >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha;
                if (flags & VCX_GEMM_BIAS_M) {   // loaded where it is consumed: a load left pending on a skipped path
                    const float bm = p.bias[m];  // would make the compiler drain vmcnt(0) in front of the next MFMAs
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bm;
                }
                const bool full = (n0 + 4 <= p.N);
                if (full) {
                    if (flags & VCX_GEMM_BIAS_N) {
                        const f4 bv = *reinterpret_cast<const f4*>(p.bias + n0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += bv[r];
                    }
                    if (radd) {
                        const f4 rv = *reinterpret_cast<const f4*>(radd + n0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    }
                    if (flags & VCX_GEMM_RESIDUAL) {
                        const h4 rr = *reinterpret_cast<const h4*>(p.R + (int64_t)m * p.ldr + n0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                    }
                    if (OUT_F32) {
                        float* dst = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n0;
                        *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]};
                    } else {
                        half_t* dst = reinterpret_cast<half_t*>(p.C) + (int64_t)m * p.ldc + n0;
                        *reinterpret_cast<h4*>(dst) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    }
                } else {
                    for (int r = 0; r < 4 && n0 + r < p.N; ++r) {
                        float x = v[r];
                        if (flags & VCX_GEMM_BIAS_N) x += p.bias[n0 + r];
                        if (radd) x += radd[n0 + r];
                        if (flags & VCX_GEMM_RESIDUAL) x += (float)p.R[(int64_t)m * p.ldr + n0 + r];
                        if (OUT_F32)
                            reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n0 + r] = x;
                        else
                            reinterpret_cast<half_t*>(p.C)[(int64_t)m * p.ldc + n0 + r] = (half_t)x;
                    }
                }
            }
        }
    }
            }
#pragma unroll
            for (int a = 0; a < NFRAG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        }   // end of the tile's epilogue
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        if (++ckt == nk) {
            ckt = 0;
            ctile += G;
            if (ctile >= ntiles) break;
            tile_coords(ctile, tile_m, tile_n);
        }
    }
}

}  // namespace
int vcxgemm::persistent_grid(int ntiles, int blocks_per_cu) {
    static std::atomic<int> cached{0};       // every GPU of a node is the same part: one query per process
    int ncu = cached.load(std::memory_order_relaxed);
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ncu = prop.multiProcessorCount;
        cached.store(ncu, std::memory_order_relaxed);
    }
    const int slots = blocks_per_cu * ncu;   // resident blocks chip-wide (LDS- and VGPR-limited)
    if (ntiles <= slots) return ntiles;
    const int rounds = (ntiles + slots - 1) / slots;          // balance: every block gets rounds or rounds-1 tiles
    int g = (ntiles + rounds - 1) / rounds;
    g = (g + 7) & ~7;                                         // multiple of 8 keeps a block's tiles on one XCD band
    return g < slots ? g : slots;
}
namespace {

template <int BN, bool CONV, bool GEGLU, bool OUT_F32>
int launch(const GemmArgs& a, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
    static VcxLdsAttr lds;
    auto kern = gemm_kernel<BN, CONV, GEGLU, OUT_F32>;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), (int)smem, "vcx_gemm_f16")) return VCX_ELAUNCH;
    const int nb = persistent_grid(a.tiles_m * a.tiles_n);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(NTHREADS), smem, s, a);
    return vcx_check_launch("vcx_gemm_f16");
}

template <int BN>
int dispatch(const GemmArgs& a, bool conv, bool geglu, bool f32, hipStream_t s) {
    if (geglu) return conv ? launch<BN, true, true, false>(a, s) : launch<BN, false, true, false>(a, s);
    if (f32) return conv ? launch<BN, true, false, true>(a, s) : launch<BN, false, false, true>(a, s);
    return conv ? launch<BN, true, false, false>(a, s) : launch<BN, false, false, false>(a, s);
}

}  // namespace

static bool force_cfg_unset() { return vcx_tune(VCX_TUNE_GEMM_CFG) < 0; }      // a forced tile configuration (A/B tools, soak) means the tiled engine

extern "C" int vcx_gemm_f16(const vcx_gemm_desc* d, void* stream) {
    VCX_REQUIRE(d != nullptr, "vcx_gemm_f16: null descriptor");
    // ABI 5: the caller states the size of the struct it filled in.  A binding written against another header version (the
    // 144-byte ABI-1 or 168-byte ABI-4 layouts started with the A pointer) is refused here instead of being read past its end.
    VCX_REQUIRE(d->struct_size == sizeof(vcx_gemm_desc), "vcx_gemm_f16: descriptor struct_size %zu != %zu (sizeof(vcx_gemm_desc), ABI %d)",
                (size_t)d->struct_size, sizeof(vcx_gemm_desc), VCX_ABI_VERSION);
    VCX_REQUIRE(d->A && d->W && d->C, "vcx_gemm_f16: null A/W/C");
    VCX_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "vcx_gemm_f16: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    VCX_REQUIRE(d->K % 8 == 0 && d->ldw % 8 == 0 && d->lda % 8 == 0,
                "vcx_gemm_f16: K (%d), ldw (%d), lda (%lld) must be multiples of 8", d->K, d->ldw, (long long)d->lda);
    VCX_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->W & 15) == 0 && ((uintptr_t)d->C & 15) == 0,
                "vcx_gemm_f16: A/W/C must be 16-byte aligned");
    const int flags = d->flags;
    VCX_REQUIRE(!(flags & (VCX_GEMM_BIAS_N | VCX_GEMM_BIAS_M)) || d->bias, "vcx_gemm_f16: bias flag without bias");
    VCX_REQUIRE(!(flags & VCX_GEMM_ROWADD) || (d->rowadd && d->rowadd_div > 0 && (d->rowadd_ld == 0 || (d->rowadd_ld >= d->N && d->rowadd_ld % 4 == 0))),
                "vcx_gemm_f16: bad rowadd (rowadd_ld must be 0 or a multiple of 4 >= N)");
    VCX_REQUIRE(!(flags & VCX_GEMM_RESIDUAL) || d->residual, "vcx_gemm_f16: residual flag without pointer");
    const bool conv = d->mode == 1;
    const bool geglu = flags & VCX_GEMM_GEGLU;
    const bool f32 = flags & VCX_GEMM_OUT_F32;
    VCX_REQUIRE(!(flags & VCX_GEMM_CONV_SLABK) || (d->mode == 1 && d->cin % 64 == 0),
                "vcx_gemm_f16: VCX_GEMM_CONV_SLABK needs a convolution with cin %% 64 == 0 (cin=%d)", d->cin);
    VCX_REQUIRE(d->mode == 0 || d->mode == 1, "vcx_gemm_f16: unknown mode %d", d->mode);
    VCX_REQUIRE(!(geglu && (f32 || (flags & (VCX_GEMM_ROWADD | VCX_GEMM_RESIDUAL | VCX_GEMM_BIAS_M)))),
                "vcx_gemm_f16: GEGLU combines only with BIAS_N");
    VCX_REQUIRE(!geglu || d->N % 64 == 0, "vcx_gemm_f16: GEGLU needs N %% 64 == 0 (N=%d)", d->N);
    const int lnf = (flags & VCX_GEMM_LNFOLD) ? 1 : (flags & VCX_GEMM_LNFOLD_T) ? 2 : 0;
    if (lnf) {
        VCX_REQUIRE(!((flags & VCX_GEMM_LNFOLD) && (flags & VCX_GEMM_LNFOLD_T)), "vcx_gemm_f16: LNFOLD and LNFOLD_T are exclusive");
        VCX_REQUIRE(d->ln_stats && d->ln_colsum && ((uintptr_t)d->ln_stats & 15) == 0 && ((uintptr_t)d->ln_colsum & 15) == 0,
                    "vcx_gemm_f16: LNFOLD needs 16-byte aligned ln_stats and ln_colsum");
        VCX_REQUIRE(!conv && !f32 && !(flags & VCX_GEMM_ROWADD), "vcx_gemm_f16: LNFOLD is for linear layers with fp16 output, without ROWADD");
        VCX_REQUIRE(lnf == 1 ? !(flags & VCX_GEMM_BIAS_M) : !(flags & VCX_GEMM_BIAS_N) && !geglu,
                    "vcx_gemm_f16: LNFOLD takes BIAS_N (and GEGLU), LNFOLD_T takes BIAS_M");
        VCX_REQUIRE((lnf == 1 ? d->N : d->M) % 4 == 0 && (lnf == 1 ? d->N : d->M) >= 4, "vcx_gemm_f16: LNFOLD needs the colsum side to be a multiple of 4");
    }
    if (flags & VCX_GEMM_COLSTATS) {
        VCX_REQUIRE(d->colstats && ((uintptr_t)d->colstats & 15) == 0, "vcx_gemm_f16: COLSTATS needs a 16-byte aligned colstats buffer");
        VCX_REQUIRE(!geglu && !f32 && !lnf && d->M % 64 == 0 && d->N % 8 == 0,
                    "vcx_gemm_f16: COLSTATS is for fp16 outputs (no GEGLU / LNFOLD) with M %% 64 == 0 and N %% 8 == 0 (M=%d N=%d)", d->M, d->N);
        VCX_REQUIRE(d->ldcs == 0 || (d->ldcs >= d->N && d->ldcs % 2 == 0), "vcx_gemm_f16: COLSTATS ldcs (%lld) must be 0 or an even number >= N", (long long)d->ldcs);
    }
    if (conv) {
        VCX_REQUIRE(d->cin > 0 && d->cin % 8 == 0 && d->kh > 0 && d->kw > 0 && d->tail_k0 >= 0 && d->tail_k1 >= 0 &&
                        d->K == d->kh * d->kw * d->cin + d->tail_k0 + d->tail_k1,
                    "vcx_gemm_f16: conv needs cin %% 8 == 0 and K == kh*kw*cin + tail (cin=%d kh=%d kw=%d K=%d tail=%d+%d)", d->cin,
                    d->kh, d->kw, d->K, d->tail_k0, d->tail_k1);
        VCX_REQUIRE(d->out_h > 0 && d->out_w > 0 && d->in_h > 0 && d->in_w > 0 && d->stride > 0 &&
                        (d->ups == 0 || d->ups == 1) && d->M % (d->out_h * d->out_w) == 0,
                    "vcx_gemm_f16: bad conv geometry");
    }
    const int tail = conv ? d->tail_k0 + d->tail_k1 : 0;
    VCX_REQUIRE(conv || (d->tail_k0 == 0 && d->tail_k1 == 0), "vcx_gemm_f16: a K tail (tail_k0 / tail_k1) belongs to a convolution (mode 1)");
    if (tail) {
        VCX_REQUIRE(d->tail_k0 % 64 == 0 && d->tail_k1 % 64 == 0 && (d->tail_k0 == 0 || (d->tail_a0 && d->tail_lda0 >= d->tail_k0 && d->tail_lda0 % 8 == 0)) &&
                        (d->tail_k1 == 0 || (d->tail_a1 && d->tail_lda1 >= d->tail_k1 && d->tail_lda1 % 8 == 0)) && !(d->tail_k0 == 0 && d->tail_k1 != 0),
                    "vcx_gemm_f16: K tail: tail_k %% 64 == 0, sources with tail_lda >= tail_k, tail_lda %% 8 == 0, tail_a0 first (tail=%d+%d)", d->tail_k0, d->tail_k1);
        VCX_REQUIRE((((uintptr_t)d->tail_a0 | (uintptr_t)d->tail_a1) & 15) == 0 && !geglu && !f32 && !lnf, "vcx_gemm_f16: K tail: 16-byte aligned sources, fp16 output, no GEGLU / LNFOLD");
    }
    // vector epilogue alignment
    if (!geglu) {
        VCX_REQUIRE(d->N < 4 || d->N % 4 != 0 || d->ldc % 4 == 0, "vcx_gemm_f16: ldc must be a multiple of 4");
        if (flags & VCX_GEMM_RESIDUAL)
            VCX_REQUIRE(d->ldr % 4 == 0 && ((uintptr_t)d->residual & 7) == 0, "vcx_gemm_f16: residual alignment");
        if ((flags & VCX_GEMM_BIAS_N) && d->N >= 4) VCX_REQUIRE(((uintptr_t)d->bias & 15) == 0, "vcx_gemm_f16: bias alignment");
        if (flags & VCX_GEMM_ROWADD) VCX_REQUIRE(((uintptr_t)d->rowadd & 15) == 0 && d->N % 4 == 0, "vcx_gemm_f16: rowadd alignment");
    } else {
        VCX_REQUIRE(d->ldc % 4 == 0, "vcx_gemm_f16: ldc must be a multiple of 4");
    }

    GemmArgs a;
    a.A = (const half_t*)d->A;
    a.W = (const half_t*)d->W;
    a.C = d->C;
    a.bias = d->bias;
    a.rowadd = d->rowadd;
    a.R = (const half_t*)d->residual;
    a.lda = d->lda;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.ldw = d->ldw; a.ldc = d->ldc; a.ldr = d->ldr;
    a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w;
    a.cin = d->cin; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.ups = d->ups;
    a.rowadd_div = d->rowadd_div > 0 ? d->rowadd_div : 1;
    a.rowadd_ld = d->rowadd_ld > 0 ? d->rowadd_ld : d->N;
    a.flags = flags;
    a.alpha = d->alpha;
    const bool use160 = !geglu && (d->N % 160 == 0);
    const int bn = use160 ? 160 : 128;
    a.tiles_m = (d->M + BM - 1) / BM;
    a.tiles_n = (d->N + bn - 1) / bn;
    a.m_begin = 0;
    a.ln_stats = d->ln_stats;
    a.ln_colsum = d->ln_colsum;
    a.colstats = d->colstats;
    a.ldcs = d->ldcs > 0 ? d->ldcs : d->N;
    a.unit_rows = 0; a.units = 1; a.w_unit_stride = 0; a.bias_unit_stride = 0;
    a.rowstats = d->rowstats; a.rowstats_eps = d->rowstats_eps;
    a.A2 = (const half_t*)d->tail_a0; a.A3 = (const half_t*)d->tail_a1; a.lda2 = d->tail_lda0; a.lda3 = d->tail_lda1;
    a.k2 = conv ? d->tail_k0 : 0; a.k3 = conv ? d->tail_k1 : 0; a.a2_bytes = a.a3_bytes = 0;
    if (flags & VCX_GEMM_ROWSTATS)
        VCX_REQUIRE(d->rowstats && ((uintptr_t)d->rowstats & 7) == 0 && d->rowstats_eps >= 0.f, "vcx_gemm_f16: ROWSTATS needs an 8-byte aligned rowstats buffer and eps >= 0");
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * d->M * (double)d->N * d->K;
    const double bytes = 2.0 * ((double)d->M * d->K / (conv ? d->kh * d->kw : 1) + (double)d->N * d->K + (double)d->M * d->N);
    VcxProfScope prof(VCX_FAM_GEMM, s, flops, bytes);
    // DMA kernel (gemm_dma.hip) whenever its addressing assumptions hold; the register-staged kernel otherwise.
    const bool dma_enabled = vcx_tune(VCX_TUNE_GEMM_DMA) != 0;
    const unsigned long long lim = 0xFFFF0000ull;
    const unsigned long long a_ext = conv ? 2ull * (unsigned long long)(d->M / (d->out_h * d->out_w)) * d->in_h * d->in_w * d->lda
                                          : 2ull * ((unsigned long long)(d->M - 1) * d->lda + d->K);
    const unsigned long long w_ext = 2ull * ((unsigned long long)(d->N - 1) * d->ldw + d->K);
    // output / residual: 32-bit byte offsets up to 256 rows past the end must not wrap (rows >= M are dropped by the
    // descriptor's range check, which only works if their offset is still >= the extent)
    const unsigned long long esz = f32 ? 4 : 2;
    const unsigned long long c_ext = esz * ((unsigned long long)(d->M - 1) * d->ldc + (geglu ? d->N / 2 : d->N));
    const unsigned long long r_ext = d->residual ? 2ull * ((unsigned long long)(d->M - 1) * d->ldr + d->N) : 0;
    const bool out_ok = esz * (unsigned long long)(d->M + 256) * d->ldc < lim && 2ull * (unsigned long long)(d->M + 256) * d->ldr < lim;
    const unsigned long long a2_ext = a.k2 ? 2ull * ((unsigned long long)(d->M - 1) * d->tail_lda0 + a.k2) : 0;
    const unsigned long long a3_ext = a.k3 ? 2ull * ((unsigned long long)(d->M - 1) * d->tail_lda1 + a.k3) : 0;
    const bool dma_ok = dma_enabled && d->K % 64 == 0 && d->N % ((geglu || f32) ? 4 : 8) == 0 && (!conv || d->cin % 64 == 0) &&   // fp16 output goes out in dwordx4 pieces of 8 columns
                        a_ext < lim && w_ext < lim && (!geglu || d->N >= 64) && out_ok && a2_ext < lim && a3_ext < lim;
    a.a2_bytes = (unsigned)a2_ext; a.a3_bytes = (unsigned)a3_ext;
    VCX_REQUIRE(!tail || dma_ok, "vcx_gemm_f16: a K tail needs the DMA kernel (cin %% 64 == 0, K %% 64 == 0, N %% 8 == 0, extents < 4 GiB); cin=%d K=%d N=%d", d->cin, d->K, d->N);
    a.a_bytes = (unsigned)a_ext;
    a.w_bytes = (unsigned)w_ext;
    a.c_bytes = (unsigned)c_ext;
    a.r_bytes = (unsigned)r_ext;
    // Weight-stationary kernel (gemm_ws.hip) for the memory-bound K = 320 linear layers of level 0 (N = 320, 640, 960): the weight stays
    // in the register file, only the activation rows stream.  From 128 tiles of 64 rows on (below that the tiled engine's small
    // configuration is as good).  Not the LayerNorm-folded projections: a lean folded epilogue was built and measured level with the
    // tiled engine (0.407 vs 0.416 ms; profiles/r05_experiments.md section 3), so they stay there.
    // ... and for the GEGLU projection of level 0 (N = 2560 packed columns: ten 256-column blocks per row stream on one XCD), whose GELU
    // epilogue rides in the next tile's MFMA stream there (knob GEMM_WS = 2: everything weight-stationary but this; 3: this without its
    // cross-XCD streams on the spare CUs).  For EVERY M: its
    // bias rides in the accumulators, so its last bits differ from the tiled engine's, and a row's bits must not depend on how many
    // rows the call has (B = 2 equals two B = 1 forwards bit for bit).  Beyond ten column blocks per row stream the tiled engine wins.
    if (dma_ok && !conv && geglu && !f32 && !lnf && d->K == 320 && d->N % 256 == 0 && d->N <= 2560 && !(flags & ~(VCX_GEMM_GEGLU | VCX_GEMM_BIAS_N)) &&
        d->alpha == 1.0f && ((vcx_tune(VCX_TUNE_GEMM_WS) | 2) == 3 || vcx_tune(VCX_TUNE_GEMM_WS) == 5) && force_cfg_unset())
        return launch_ws320_geglu(a, s);
    // ... and for the LayerNorm-folded projections of level 0 (q | k | v of the spatial self-attention, N = 960, and the 640-column ones) on
    // the same skeleton with a lighter epilogue (gemm_ws320_lnf_kernel; knob GEMM_WS = 4: everything weight-stationary but this).  For
    // every M, like the GEGLU kernel: its bits differ from the tiled engine's (32x32x16 sums K in another order).
    // Where: measured against the tiled engine at M = 460800 (profiles/r05am_ws_lnf_ab.txt) it wins when the last 256-column block is at
    // least three quarters full - N = 512 (-15 %), 960 (-3 %), 1280 (-10 %); 640 (a half-empty third block) +5 %, 1920 level.  The kernel
    // is bound by what a CU can pull through LDS-DMA (every column block streams all activation rows), not by its matrix work
    // (profiles/r05an_ws_lnf_ablate.txt).  Knob GEMM_WS = 5 sends every N % 64 == 0 up to 2560 there (tests).
    {
        const int wsk = vcx_tune(VCX_TUNE_GEMM_WS);
        const int pad = (d->N + 255) / 256 * 256 - d->N;
        if (dma_ok && !conv && !geglu && !f32 && lnf == 1 && d->K == 320 && d->N % 64 == 0 && d->N <= 2560 &&
            ((wsk >= 1 && wsk <= 3 && pad <= 64 && d->N >= 512 && d->N <= 1536) || wsk == 5) &&
            !(flags & ~(VCX_GEMM_LNFOLD | VCX_GEMM_BIAS_N)) && 8ull * (unsigned long long)d->M < lim && force_cfg_unset())
            return launch_ws320_lnfold(a, s);
    }
    // ROWSTATS (LayerNorm statistics of the output rows) exists where one block owns whole rows: the pipelined weight-stationary kernel, N = 320
    const bool rs_ok = !(flags & VCX_GEMM_ROWSTATS) || (d->N == 320 && !(flags & ~(VCX_GEMM_ROWSTATS | VCX_GEMM_BIAS_N | VCX_GEMM_RESIDUAL)) && 8ull * (unsigned long long)d->M < lim);
    if (dma_ok && !conv && !geglu && !f32 && !lnf && d->K == 320 && d->N % 320 == 0 && d->N <= 1280 && d->M >= 8192 && rs_ok &&
        !(flags & VCX_GEMM_BIAS_M) && vcx_tune(VCX_TUNE_GEMM_WS) != 0 && force_cfg_unset())
        return launch_ws320(a, s);
    VCX_REQUIRE(!(flags & VCX_GEMM_ROWSTATS), "vcx_gemm_f16: ROWSTATS needs the weight-stationary kernel (linear, N = K = 320, M >= 8192, BIAS_N / RESIDUAL at most, knob GEMM_WS on); M=%d N=%d K=%d flags=0x%x",
                d->M, d->N, d->K, flags);
    if (dma_ok) {
        // tile choice: the large (256-row, 8-wave) tiles halve the LDS traffic per MFMA but need >= ~1.5 waves of 256 tiles
        const int force = vcx_tune(VCX_TUNE_GEMM_CFG);      // -1 in production; tools/gemm_quick.py A/Bs tile configurations
        int cfg = use160 ? 1 : 0;
        const int big_bn = (d->N % 320 == 0 && !geglu) ? 320 : ((d->N % 256 == 0 || d->N >= 1024) ? 256 : 0);
        if (big_bn) {
            const long long tiles = (long long)((d->M + 255) / 256) * ((d->N + big_bn - 1) / big_bn);
            if (tiles >= 384) cfg = big_bn == 320 ? 3 : 2;
        }
        if (force >= 0 && !(geglu && (force == 1 || force == 3))) cfg = force;
        const int tbm = cfg >= 2 ? 256 : 128, tbn = cfg == 0 ? 128 : cfg == 1 ? 160 : cfg == 2 ? 256 : 320;
        a.tiles_m = (d->M + tbm - 1) / tbm;
        a.tiles_n = (d->N + tbn - 1) / tbn;
        auto big = [&](GemmArgs& g, int c) { return launch_dma(g, c, conv, geglu, f32, s); };
        if (cfg >= 2) {
            // Large tiles run one block per CU: a partial last round of 256-row tiles costs a full tile time.  When the
            // remainder is small, finish the full rounds with large tiles and hand the tail rows to the small-tile config.
            const int slots = persistent_grid(1 << 30, 1);
            const long long tiles = (long long)a.tiles_m * a.tiles_n;
            const long long full = tiles / slots, rem = tiles % slots;
            if (full >= 1 && rem > 0 && rem * 10 < slots * 7) {
                const int tm1 = (int)(full * slots / a.tiles_n);
                const int m1 = tm1 * 256;
                if (m1 > 0 && m1 < d->M) {
                    GemmArgs b = a;
                    b.M = m1;
                    b.tiles_m = tm1;
                    int rc = big(b, cfg);
                    if (rc) return rc;
                    GemmArgs c = a;
                    const int scfg = (geglu || d->N % 160 != 0) ? 0 : 1;
                    const int sbn = scfg ? 160 : 128;
                    c.m_begin = m1;
                    c.tiles_m = (d->M - m1 + 127) / 128;
                    c.tiles_n = (d->N + sbn - 1) / sbn;
                    return launch_dma(c, scfg, conv, geglu, f32, s);
                }
            }
        }
        return cfg >= 2 ? big(a, cfg) : launch_dma(a, cfg, conv, geglu, f32, s);
    }
    VCX_REQUIRE(!(flags & VCX_GEMM_COLSTATS), "vcx_gemm_f16: COLSTATS needs the DMA kernel (K / cin %% 64 == 0, extents < 4 GiB); K=%d cin=%d", d->K, d->cin);
    VCX_REQUIRE(!lnf, "vcx_gemm_f16: LNFOLD needs the DMA kernel (K %% 64 == 0, N %% 8 == 0, extents < 4 GiB); K=%d N=%d", d->K, d->N);
    return use160 ? dispatch<160>(a, conv, geglu, f32, s) : dispatch<128>(a, conv, geglu, f32, s);
}

// One weight / bias set per unit of rows (include/vcx.h): the weight-stationary kernel in ONE launch where it applies (N = K = 320, the
// level-0 projections: a block keeps its unit's weights in registers anyway), otherwise unit by unit through vcx_gemm_f16.
extern "C" int vcx_gemm_units_f16(const vcx_gemm_desc* d, int unit_rows, int64_t w_unit_stride, int64_t bias_unit_stride, void* stream) {
    VCX_REQUIRE(d != nullptr && d->struct_size == sizeof(vcx_gemm_desc), "vcx_gemm_units_f16: null descriptor or wrong struct_size");
    VCX_REQUIRE(d->A && d->W && d->C && d->M > 0 && d->N > 0 && d->K > 0, "vcx_gemm_units_f16: null A/W/C or empty problem");
    VCX_REQUIRE(d->mode == 0 && !(d->flags & ~(VCX_GEMM_BIAS_N | VCX_GEMM_ROWSTATS)), "vcx_gemm_units_f16: linear layers with a per-column bias (and ROWSTATS) at most (mode %d flags 0x%x)", d->mode, d->flags);
    VCX_REQUIRE(!(d->flags & VCX_GEMM_ROWSTATS) || (d->rowstats && ((uintptr_t)d->rowstats & 7) == 0 && d->rowstats_eps >= 0.f), "vcx_gemm_units_f16: ROWSTATS needs an 8-byte aligned rowstats buffer and eps >= 0");
    VCX_REQUIRE(unit_rows > 0 && d->M % unit_rows == 0, "vcx_gemm_units_f16: M (%d) must be a whole number of units of %d rows", d->M, unit_rows);
    VCX_REQUIRE(w_unit_stride % 8 == 0 && bias_unit_stride % 4 == 0, "vcx_gemm_units_f16: unit strides must keep W 16-byte and bias 16-byte aligned");
    VCX_REQUIRE(!(d->flags & VCX_GEMM_BIAS_N) || d->bias, "vcx_gemm_units_f16: bias flag without bias");
    VCX_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && d->ldc % 8 == 0 && (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C) & 15) == 0 && ((uintptr_t)d->bias & 15) == 0,
                "vcx_gemm_units_f16: strides must be multiples of 8, pointers 16-byte aligned");
    const int units = d->M / unit_rows;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long lim = 0xFFFF0000ull;
    const unsigned long long a_ext = 2ull * ((unsigned long long)(d->M - 1) * d->lda + d->K), c_ext = 2ull * ((unsigned long long)(d->M - 1) * d->ldc + d->N);
    if (units > 1 && d->K == 320 && d->N == 320 && unit_rows % 32 == 0 && unit_rows >= 1024 && d->M >= 8192 && a_ext < lim &&
        2ull * (unsigned long long)(d->M + 256) * d->ldc < lim && units <= 65535 && vcx_tune(VCX_TUNE_GEMM_DMA) != 0 && vcx_tune(VCX_TUNE_GEMM_WS) != 0 &&
        force_cfg_unset()) {
        GemmArgs a{};
        a.A = (const half_t*)d->A; a.W = (const half_t*)d->W; a.C = d->C; a.bias = d->bias;
        a.lda = d->lda; a.M = d->M; a.N = d->N; a.K = d->K; a.ldw = d->ldw; a.ldc = d->ldc; a.ldr = 0;
        a.rowadd_div = 1; a.rowadd_ld = d->N; a.flags = d->flags; a.alpha = d->alpha; a.m_begin = 0;
        a.ldcs = d->N;
        a.a_bytes = (unsigned)a_ext; a.c_bytes = (unsigned)c_ext; a.w_bytes = 0; a.r_bytes = 0;
        a.unit_rows = unit_rows; a.units = units; a.w_unit_stride = w_unit_stride; a.bias_unit_stride = bias_unit_stride;
        a.rowstats = d->rowstats; a.rowstats_eps = d->rowstats_eps;
        VcxProfScope prof(VCX_FAM_GEMM, s, 2.0 * d->M * (double)d->N * d->K, 2.0 * ((double)d->M * d->K + (double)units * d->N * d->K + (double)d->M * d->N));
        return launch_ws320_units(a, s);
    }
    VCX_REQUIRE(!(d->flags & VCX_GEMM_ROWSTATS), "vcx_gemm_units_f16: ROWSTATS needs the one-launch weight-stationary form (N = K = 320, unit_rows %% 32 == 0, >= 1024, M >= 8192); M=%d N=%d K=%d unit_rows=%d",
                d->M, d->N, d->K, unit_rows);
    for (int u = 0; u < units; ++u) {
        vcx_gemm_desc du = *d;
        du.A = (const half_t*)d->A + (int64_t)u * unit_rows * d->lda;
        du.C = (half_t*)d->C + (int64_t)u * unit_rows * d->ldc;
        du.W = (const half_t*)d->W + (int64_t)u * w_unit_stride;
        if (d->bias) du.bias = d->bias + (int64_t)u * bias_unit_stride;
        du.M = unit_rows;
        const int rc = vcx_gemm_f16(&du, stream);
        if (rc) return rc;
    }
    return VCX_OK;
}
