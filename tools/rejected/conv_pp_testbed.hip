// Implicit-GEMM 3x3 convolution (NHWC, pad 1, stride 1) on the phase-split main loop of tools/gemm_pp.hip - test bed.
//   hipcc --offload-arch=gfx950 -O3 tools/conv_pp.hip -o /tmp/conv_pp && /tmp/conv_pp [frames H W Cin Cout]
//   C[M][N] (fp16) = X[M][K] . W[N][K]^T, 256x256 tile, BK = 64, 8 waves: group g = wave >> 2 owns rows 128 g .. 128 g + 127,
//   wave & 3 picks a 64-column strip.  A K-tile is four phases of 16 MFMAs; every phase is a LOAD slot (ds_reads of the phase's
//   fragments + two DMA pieces) and an MFMA slot, separated by raw s_barriers, and group 1 runs one slot behind group 0 - while one
//   wave of a SIMD issues MFMAs, the other issues LDS reads and DMA.  X has three LDS stages (rows of tile T+2 in flight), W two;
//   the only vmcnt wait of a K-tile is vmcnt(4): it never drains the queue.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_pp.hip -o /tmp/gemm_pp && /tmp/gemm_pp [M N K]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half_t;
typedef half_t h8 __attribute__((ext_vector_type(8)));
typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BK = 64;
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt((((n) >> 4) << 14) | 0x0f70 | ((n) & 15))
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int TBM, int TBN, int XST>
__global__ void __launch_bounds__(512, 2) conv_pp(const half_t* X, const half_t* W, half_t* C, int M, int N, int K, int H, int Wd, int Cin) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WN = TBN / 4, NF = WN / 16;                // wave strip (64 or 80 columns) and its 16-column fragments
    constexpr int WP = TBN / 64;                           // DMA pieces of the weight tile per wave (4 or 5)
    constexpr int XP = TBM / 64;                           // DMA pieces of the row tile per wave (4 or 3)
    constexpr int GM = TBM / 2, MF = GM / 16, MH = MF / 2;  // rows per wave group, its 16-row fragments, fragments per phase
    half_t* sX = reinterpret_cast<half_t*>(smem);          // [XST][TBM * 64]
    half_t* sW = sX + XST * TBM * BK;                      // [2][TBN * 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    const int tiles_n = N / TBN, tiles_m = M / TBM;
    // XCD-aware order: blocks b, b+8, b+16, ... (same XCD) walk one contiguous band of tiles
    const int nt = tiles_m * tiles_n, q8 = nt >> 3, r8 = nt & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tile_m = vid / tiles_n, tile_n = vid % tiles_n;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(X), 0, (int)((long long)M * Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W), 0, (int)((long long)N * K * 2), 0x00020000);
    // DMA piece q (0..3) of a 256-row tile for this wave: rows 64 q + 8 wave + (lane >> 3), source chunk swizzled
    unsigned xoff[4], xmask[4], woff[WP];
#pragma unroll
    for (int q = 0; q < WP; ++q) {
        const int r = 64 * q + 8 * wave + (lane >> 3);
        const unsigned csrc = (unsigned)((lane & 7) ^ ((r >> 1) & 7)) * 16u;
        if (q < XP) {
            const int m = tile_m * TBM + r, mm = m < M ? m : 0;
            const int img = mm / (H * Wd), rem = mm - img * H * Wd, oy = rem / Wd, ox = rem - oy * Wd;
            const long long pix0 = ((long long)img * H + (oy - 1)) * Wd + (ox - 1);        // tap (0,0); may be negative at the border
            xoff[q] = (unsigned)(pix0 * Cin * 2) + csrc;                                   // wraps; valid taps un-wrap it
            unsigned mask = 0;
            if (m < M)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        if (oy + ky - 1 >= 0 && oy + ky - 1 < H && ox + kx - 1 >= 0 && ox + kx - 1 < Wd) mask |= 1u << (ky * 3 + kx);
            xmask[q] = mask;
        }
        woff[q] = (unsigned)((long long)(tile_n * TBN + r) * K * 2) + csrc;
    }
    int tap = 0, ci0 = 0, tkx = 0, tky = 0;     // block-uniform K walker of the row loads: k = tap * Cin + ci0
    unsigned tap_off = 0;
    auto dma_x = [&](int kt, int q) {   // piece q of the X rows of K-tile kt -> stage kt % XST
        half_t* d = sX + (kt % XST) * TBM * BK + (64 * q + 8 * wave) * BK;
        const unsigned v = ((xmask[q] >> tap) & 1u) ? xoff[q] + tap_off : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_x, (lds_ptr_t)d, 16, v, 0, 0, 0);
    };
    auto advance_x = [&]() {
        ci0 += BK; tap_off += BK * 2;
        if (ci0 == Cin) { ci0 = 0; ++tap; if (++tkx == 3) { tkx = 0; ++tky; } tap_off = (unsigned)((tky * Wd + tkx) * Cin * 2); }
    };
    auto dma_w = [&](int kt, int q) {
        half_t* d = sW + (kt & 1) * TBN * BK + (64 * q + 8 * wave) * BK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_ptr_t)d, 16, woff[q], (unsigned)kt * 128u, 0, 0);
    };
    f4 acc[NF][MF];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    // prologue: W(0), X(0), X(1); rows of tile T+2 and weights of tile T+1 are issued during tile T
    constexpr int XAHEAD = XST - 1;        // the rows are fetched XAHEAD K-tiles ahead, the weights one
#pragma unroll
    for (int q = 0; q < WP; ++q) dma_w(0, q);
#pragma unroll
    for (int q = 0; q < XP; ++q) dma_x(0, q);
    advance_x();
    if (XAHEAD == 2 && nk > 1) {
#pragma unroll
        for (int q = 0; q < XP; ++q) dma_x(1, q);
        advance_x();
        WAIT_VM(XP);
    } else {
        WAIT_VM(0);
    }
    BARRIER();
    if (grp == 1) BARRIER();          // group 1 runs one slot behind group 0
    for (int kt = 0; kt < nk; ++kt) {
        const half_t* cx = sX + (kt % XST) * TBM * BK + (grp * GM) * BK;
        const half_t* cw = sW + (kt & 1) * TBN * BK + (wc * WN) * BK;
        const bool w_next = kt + 1 < nk, x_next = kt + XAHEAD < nk;
        h8 wf[NF], xf[MH];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int kk = ph >> 1, mh = ph & 1;
            // ---- LOAD slot
            if (mh == 0) {
#pragma unroll
                for (int a = 0; a < NF; ++a) wf[a] = *reinterpret_cast<const h8*>(cw + lds_off(a * 16 + lr, kk * 4 + lg));
            }
#pragma unroll
            for (int b = 0; b < MH; ++b) xf[b] = *reinterpret_cast<const h8*>(cx + lds_off((mh * MH + b) * 16 + lr, kk * 4 + lg));
            if (ph < 2) {
                if (w_next) {
                    dma_w(kt + 1, 2 * ph); dma_w(kt + 1, 2 * ph + 1);
                    if (WP == 5 && ph == 1) dma_w(kt + 1, 4);
                }
            } else if (x_next) { dma_x(kt + XAHEAD, 2 * (ph - 2)); if (2 * (ph - 2) + 1 < XP) dma_x(kt + XAHEAD, 2 * (ph - 2) + 1); }
            if (ph == 3 && x_next) advance_x();
            if (ph == 3) {             // the weights of tile kt+1 (and everything older) must have landed; with three row stages the rows of kt+2 may fly
                if (XAHEAD == 2 && x_next) WAIT_VM(XP); else WAIT_VM(0);
            }
            BARRIER();
            // ---- MFMA slot
            WAIT_LDS();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < MH; ++b)
                    acc[a][mh * MH + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][mh * MH + b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            BARRIER();
        }
    }
    if (grp == 0) BARRIER();          // match group 1's extra barrier
    // epilogue: plain fp16 store (acc[a][b][r] = C[m = 16 b + lr][n = 16 a + 4 lg + r] inside the wave tile)
    half_t* cbase = C + (long long)(tile_m * TBM + grp * GM + lr) * N + tile_n * TBN + wc * WN + lg * 4;
#pragma unroll
    for (int b = 0; b < MF; ++b)
#pragma unroll
        for (int a = 0; a < NF; ++a)
            *reinterpret_cast<h4*>(cbase + (long long)b * 16 * N + a * 16) =
                h4{(half_t)acc[a][b][0], (half_t)acc[a][b][1], (half_t)acc[a][b][2], (half_t)acc[a][b][3]};
#endif
}

int main(int argc, char** argv) {
    const int F = argc > 5 ? atoi(argv[1]) : 50, H = argc > 5 ? atoi(argv[2]) : 72, Wd = argc > 5 ? atoi(argv[3]) : 128;
    const int Cin = argc > 5 ? atoi(argv[4]) : 320, N = argc > 5 ? atoi(argv[5]) : 320;
    const int M = F * H * Wd, K = 9 * Cin;
    std::vector<half_t> hx((size_t)M * Cin), hw((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hx) v = (half_t)rnd();
    for (auto& v : hw) v = (half_t)(rnd() * 0.05f);
    half_t *dx, *dw, *dc;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dc, (size_t)M * N * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    auto run = [&](auto kern, int tbm, int tbn, int xst, const char* name) {
        if (N % tbn || M % tbm) { printf("%s: shape not a multiple of the tile, skipped\n", name); return; }
        const size_t smem = (size_t)(xst * tbm + 2 * tbn) * BK * 2;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const int grid = (M / tbm) * (N / tbn);
        hipMemset(dc, 0, (size_t)M * N * 2);
        kern<<<grid, 512, smem>>>(dx, dw, dc, M, N, K, H, Wd, Cin);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return; }
        std::vector<half_t> hc((size_t)M * N);
        hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int t = 0; t < 3000; ++t) {
            int m = (int)(((unsigned long long)t * 2654435761ull) % M); const int n = (int)(((unsigned long long)t * 40503ull + 17) % N);
            if (t < 400) m = (m / Wd) * Wd + (t & 1 ? 0 : Wd - 1);          // image borders
            if (t >= 400 && t < 600) m = (m / (H * Wd)) * H * Wd + (t & 1 ? 0 : (H - 1) * Wd) + m % Wd;
            const int img = m / (H * Wd), oy = (m / Wd) % H, ox = m % Wd;
            double ref = 0;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = oy + ky - 1, ix = ox + kx - 1;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
                    const half_t* xp = &hx[(((size_t)img * H + iy) * Wd + ix) * Cin];
                    const half_t* wp = &hw[(size_t)n * K + (ky * 3 + kx) * Cin];
                    for (int c = 0; c < Cin; ++c) ref += (double)(float)xp[c] * (double)(float)wp[c];
                }
            const double err = fabs(ref - (double)(float)hc[(size_t)m * N + n]);
            if (err > maxerr) maxerr = err;
            if (fabs(ref) > maxref) maxref = fabs(ref);
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 10;
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) kern<<<grid, 512, smem>>>(dx, dw, dc, M, N, K, H, Wd, Cin);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
        printf("%-22s conv3x3 %dx%dx%d Cin %d Cout %d (M %d K %d): %.3f ms  %.0f TF/s   (max |err| %.3g, max |ref| %.3g)\n", name, F, H, Wd, Cin, N, M, K, ms,
               2.0 * M * N * K / ms / 1e9, maxerr, maxref);
    };
    run(conv_pp<256, 256, 3>, 256, 256, 3, "256x256, 3 row stages");
    run(conv_pp<256, 320, 2>, 256, 320, 2, "256x320, 2 row stages");
    run(conv_pp<192, 320, 3>, 192, 320, 3, "192x320, 3 row stages");
    return 0;
}
