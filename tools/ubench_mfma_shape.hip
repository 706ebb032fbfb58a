// Which MFMA shape does more work per joule under the socket's power cap?  The same 64 x 160 x 64 wave-tile K-step (the tiled GEMM engine's) as
// 80 x v_mfma_f32_16x16x32_f16 (4 x 10 fragments, 2 K halves) or 40 x v_mfma_f32_32x32x16_f16 (2 x 5 fragments, 4 K slices), operands held in
// registers (random fp16 data, re-used across iterations: no LDS, no memory), 256 blocks x 8 waves = two waves per SIMD on every CU, looped for
// seconds while tools/telemetry.py samples clock and power (tools/mfma_shape_probe.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench_mfma_shape.hip -o tools/_abl/libmfma_shape.so
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ h8 rnd8(unsigned& s) {
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        v[j] = (_Float16)(((int)(s >> 9) % 2048 - 1024) * (1.0f / 512.0f));      // ~uniform in [-2, 2): every mantissa bit toggles
    }
    return v;
}

template <int SHAPE>
__global__ void __launch_bounds__(512, 1) mfma_shape_kernel(float* out, int iters) {
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float sink = 0.f;
    if (SHAPE == 16) {
        h8 w[1][10], x[1][4];      // (one K half of operands, used for both: 160 accumulators + 112 operand registers would spill)
#pragma unroll
        for (int k = 0; k < 1; ++k) {
#pragma unroll
            for (int a = 0; a < 10; ++a) w[k][a] = rnd8(seed);
#pragma unroll
            for (int b = 0; b < 4; ++b) x[k][b] = rnd8(seed);
        }
        f4 acc[10][4];
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int a = 0; a < 10; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0][a], x[0][b], acc[a][b], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) sink += acc[a][b][0] + acc[a][b][3];
    } else {
        h8 w[2][5], x[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int a = 0; a < 5; ++a) w[k][a] = rnd8(seed);
#pragma unroll
            for (int b = 0; b < 2; ++b) x[k][b] = rnd8(seed);
        }
        f16v acc[5][2];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[k & 1][a], x[k & 1][b], acc[a][b], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) sink += acc[a][b][0] + acc[a][b][15];
    }
    if (sink == 123.456f) out[0] = sink;
}

extern "C" int mfma_shape_run(int shape, int blocks, int iters, float* out, void* stream) {
    if (shape == 16) hipLaunchKernelGGL(mfma_shape_kernel<16>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(mfma_shape_kernel<32>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
