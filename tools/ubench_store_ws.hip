// Store patterns of the weight-stationary kernels (gemm_ws.hip): 256 blocks of 4 waves, a block owns a 256-column block of a 64-row
// tile, the tiles_n column blocks of a row stream sit on one XCD and walk the row tiles together.  What one wave store instruction covers:
//   0  32 rows x 32 B  (the MFMA layout after the half swap: lanes (lq, hi) hold 16 B each of row lq) - what the kernels issue
//   1  8 rows x 128 B  (a wave's 64 columns, row-major: what a wave-private LDS transpose would issue)
//   2  2 rows x 512 B  (the block's 256 columns, row-major, wave w rows 16 w ..: what a block-wide LDS transpose would issue)
//   3  2 rows x 512 B, wave w takes rows 4 j + w (the four waves complete neighbouring rows together)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store_ws.hip -o /tmp/ubench_store_ws && /tmp/ubench_store_ws
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned short* C, int N, int tiles_m, int tiles_n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lq = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
    const u4v val = {threadIdx.x, 1u, 2u, 3u};
    const int G = gridDim.x / tiles_n;
    const int cb = (blockIdx.x >> 3) % tiles_n;
    const int t_first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) / tiles_n;
    const unsigned OOB = 0xFFFFFFFFu;
    for (int t = t_first; t < tiles_m; t += G) {
        if (MODE == 0) {
            const int wcol = cb * 256 + wave * 64;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const unsigned off = wcol < N ? ((unsigned)(t * 64 + 32 * mb + lq) * N + wcol + 8 * hi) * 2u : OOB;
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, j * 32, 0);
            }
        } else if (MODE == 1) {
            const int wcol = cb * 256 + wave * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned off = wcol < N ? ((unsigned)(t * 64 + 8 * j + (lane >> 3)) * N + wcol + 8 * (lane & 7)) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = MODE == 2 ? 16 * wave + 2 * j + (lane >> 5) : 8 * j + 2 * wave + (lane >> 5);
                const int col = cb * 256 + 8 * (lane & 31);
                const unsigned off = col < N ? ((unsigned)(t * 64 + row) * N + col) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(val, srd, off, 0, 0);
            }
        }
    }
}

template <int MODE>
void run(unsigned short* C, int M, int N, const char* what) {
    const int tiles_n = (N + 255) / 256, tiles_m = M / 64;
    const int nb = 8 * (32 / tiles_n) * tiles_n;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, C, N, tiles_m, tiles_n);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms / 5 < best) best = ms / 5;
    }
    printf("N = %4d  %-34s %7.3f ms  %5.2f TB/s\n", N, what, best, 2.0 * M * N / best / 1e9);
}

int main() {
    const int M = 460800;
    unsigned short* C;
    hipMalloc(&C, (size_t)M * 2560 * 2);
    for (int N : {960, 1280, 512, 2560}) {
        run<0>(C, M, N, "32 rows x 32 B (shipped)");
        run<1>(C, M, N, "8 rows x 128 B (wave transpose)");
        run<2>(C, M, N, "2 rows x 512 B (block, 16-row bands)");
        run<3>(C, M, N, "2 rows x 512 B (block, interleaved)");
    }
    return 0;
}
