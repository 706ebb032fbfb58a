// Store-pattern micro-benchmark: how fast can a CU write a 256 x 320 fp16 tile (row stride ldc) when one wave store
// instruction covers (a) 16 rows x 64 B (the MFMA-fragment pattern of the GEMM epilogue after permlane widening),
// (b) 16 rows x 32 B (dwordx2 per lane), (c) whole 320-byte row pieces: 64 lanes x 16 B walking rows contiguously
// (what an LDS-staged epilogue would issue)?  8 waves per block, one block per CU, persistent over `tiles` tiles.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o /tmp/ubench_store && /tmp/ubench_store
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned short* C, int ldc, int tiles_m, int tiles_n, int ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
    const u4v val = {threadIdx.x, 1u, 2u, 3u};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        if (MODE == 0 || MODE == 1) {
            // wave (wm = wave & 1, wn = wave >> 1) owns rows 128 wm .. +127, columns 80 wn .. +79 (as the shipped 2x4 layout)
            const int wm = wave & 1, wn = wave >> 1;
            const unsigned base = ((unsigned)(tm * 256 + wm * 128 + lr) * ldc + tn * 320 + wn * 80) * 2u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned row = base + (unsigned)b * 16u * ldc * 2u;
                if (MODE == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, row + ((lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, row + (32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(u2v{val[0], val[1]}, srd, row + (64 + lg * 4) * 2u, 0, 0);
                } else {
#pragma unroll
                    for (int a = 0; a < 5; ++a) __builtin_amdgcn_raw_buffer_store_b64(u2v{val[0], val[1]}, srd, row + (a * 16 + lg * 4) * 2u, 0, 0);
                }
            }
        } else if (MODE == 2) {
            // wave (wm = wave & 3, wn = wave >> 2) owns rows 64 wm .. +63, columns 160 wn .. +159; per 16-row group 5 stores of
            // 64 lanes x 16 B walking the 16 x 320-byte block row-major (20 chunks per row)
            const int wm = wave & 3, wn = wave >> 2;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int c = j * 64 + lane, r = c / 20, ch = c % 20;
                    const unsigned off = ((unsigned)(tm * 256 + wm * 64 + b * 16 + r) * ldc + tn * 320 + wn * 160) * 2u + ch * 16u;
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, 0, 0);
                }
            }
        } else if (MODE == 3) {
            // 4x2 layout, fragment pattern: 16 rows x 64 B per store, 5 per row group
            const int wm = wave & 3, wn = wave >> 2;
            const unsigned base = ((unsigned)(tm * 256 + wm * 64 + lr) * ldc + tn * 320 + wn * 160) * 2u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, base + (unsigned)b * 16u * ldc * 2u + (u * 32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
        } else if (MODE == 5) {
            // 8x1 waves (32 rows x 320 columns each), fragment pattern: 16 rows x 64 B per store, 10 per row group
            const unsigned base = ((unsigned)(tm * 256 + wave * 32 + lr) * ldc + tn * 320) * 2u;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int u = 0; u < 10; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, base + (unsigned)b * 16u * ldc * 2u + (u * 32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
        } else if (MODE == 6) {
            // 4x2 waves, fragment pattern, but the two waves of a row group interleave 64-byte pieces (wave wn takes units 2u + wn)
            const int wm = wave & 3, wn = wave >> 2;
            const unsigned base = ((unsigned)(tm * 256 + wm * 64 + lr) * ldc + tn * 320) * 2u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, base + (unsigned)b * 16u * ldc * 2u + ((2 * u + wn) * 32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
        } else if (MODE == 7) {
            // whole 640-byte rows, but each wave takes every 8th row (row r = 8 i + wave) instead of 32 consecutive rows
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const int c = j * 64 + lane, rr = c / 40, ch = c % 40;
                const unsigned off = ((unsigned)(tm * 256 + rr * 8 + wave) * ldc + tn * 320) * 2u + ch * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, 0, 0);
            }
        } else if (MODE == 8) {
            // 256 x 256 tile (tiles_n recomputed by the caller), 4x2 waves: 64 rows x 256 B per wave = two whole lines per row
            const int wm = wave & 3, wn = wave >> 2;
            const unsigned base = ((unsigned)(tm * 256 + wm * 64 + lr) * ldc + tn * 256 + wn * 128) * 2u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(val, srd, base + (unsigned)b * 16u * ldc * 2u + (u * 32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
        } else if (MODE == 9) {
            // 256 x 320 tile, 4x2 waves with an uneven split on line boundaries: wn = 0 takes columns 0..191, wn = 1 192..319
            const int wm = wave & 3, wn = wave >> 2;
            const unsigned base = ((unsigned)(tm * 256 + wm * 64 + lr) * ldc + tn * 320 + wn * 192) * 2u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (u < (wn ? 4 : 6))
                        __builtin_amdgcn_raw_buffer_store_b128(val, srd, base + (unsigned)b * 16u * ldc * 2u + (u * 32 + (lg & 1) * 16 + (lg >> 1) * 8) * 2u, 0, 0);
        } else if (MODE == 4) {
            // whole tile rows: the block writes 256 rows x 640 B, each wave 32 rows, each store 64 lanes x 16 B = 1.6 rows
            const int c0 = wave * 32 * 40;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const int c = c0 + j * 64 + lane, r = c / 40, ch = c % 40;
                const unsigned off = ((unsigned)(tm * 256 + r) * ldc + tn * 320) * 2u + ch * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, 0, 0);
            }
        }
    }
}

template <int MODE>
void run(const char* name, unsigned short* C, int M, int N) {
    const int tiles_m = M / 256, tiles_n = N / (MODE == 8 ? 256 : 320), nt = tiles_m * tiles_n;
    if (N % (MODE == 8 ? 256 : 320)) return;
    k<MODE><<<256, 512>>>(C, N, tiles_m, tiles_n, nt);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE><<<256, 512>>>(C, N, tiles_m, tiles_n, nt);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double bytes = 2.0 * M * N;
    printf("%-62s M=%d N=%d: %.3f ms  %.2f TB/s  %.1f us per tile round\n", name, M, N, ms, bytes / ms / 1e9, ms * 1e3 / ((nt + 255) / 256));
}

int main() {
    unsigned short* C; hipMalloc(&C, (size_t)460800 * 1920 * 2);
    for (int N : {320, 1280, 1920}) {
        run<0>("2x4 waves, fragment pattern widened (2 x b128 + b64 / group)", C, 460800, N);
        run<1>("2x4 waves, fragment pattern narrow (5 x b64 / group)", C, 460800, N);
        run<3>("4x2 waves, fragment pattern widened (5 x b128 / group)", C, 460800, N);
        run<2>("4x2 waves, row-major 320-byte pieces (LDS-staged epilogue)", C, 460800, N);
        run<4>("whole 640-byte tile rows", C, 460800, N);
        run<5>("8x1 waves, fragment pattern (10 x b128 / group)", C, 460800, N);
        run<6>("4x2 waves, fragment pattern, pieces interleaved between the waves", C, 460800, N);
        run<7>("whole 640-byte rows, wave takes every 8th row", C, 460800, N);
        run<8>("256x256 tile, 4x2 waves: 2 whole lines per wave and row", C, 460800, N);
        run<9>("256x320 tile, 4x2 waves split 192 | 128 columns (whole lines)", C, 460800, N);
    }
    return 0;
}
