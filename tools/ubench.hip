// Micro-benchmarks for gfx950 issue rates: how many cycles do v_exp_f32 / v_fma_f32 / MFMA cost per wave instruction,
// and do MFMA and VALU work overlap (inside one wave, and between two waves of one SIMD)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define ITERS 2000

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int role_split) {
    const int wave = threadIdx.x >> 6;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f4 acc16[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc16[i] = f4{0.f, 0.f, 0.f, 0.f};
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f); b[j] = (_Float16)(j * 1e-2f); }
    // role: 0 = VALU work, 1 = MFMA work, 2 = both interleaved
    int role = MODE == 5 ? (wave < role_split ? 1 : 0) : 0;
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0) {            // v_exp_f32 only
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        }
    } else if (MODE == 1) {     // v_fma_f32 only
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
    } else if (MODE == 2) {     // MFMA 32x32x16 only (4 independent accumulators)
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
    } else if (MODE == 3) {     // one wave: 1 MFMA 32x32x16 + 8 fma, interleaved
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
            }
        }
    } else if (MODE == 4) {     // one wave: 1 MFMA + 4 exp, interleaved
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
            }
        }
    } else if (MODE == 5) {     // waves [0, role_split) do MFMA only, the rest fma only (same amount of each as modes 2 / 1)
        if (role == 1) {
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            }
        }
    } else if (MODE == 6) {     // MFMA 16x16x32 only
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc16[i], 0, 0, 0);
        }
    } else if (MODE == 7) {     // exp + fma interleaved (does the transcendental unit run beside the main VALU?)
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = __builtin_amdgcn_exp2f(v[i]);
                v[4 + i] = __builtin_fmaf(v[4 + i], 1.0001f, 0.5f);
                v[4 + i] = __builtin_fmaf(v[4 + i], 1.0001f, 0.5f);
                v[4 + i] = __builtin_fmaf(v[4 + i], 1.0001f, 0.5f);
            }
        }
    } else if (MODE == 8) {     // packed fp32 fma
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2{v[2 * i], v[2 * i + 1]};
        const f2 c1 = {1.0001f, 1.0001f}, c2 = {0.5f, 0.5f};
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = __builtin_elementwise_fma(w[i], c1, c2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = w[i][0]; v[2 * i + 1] = w[i][1]; }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7] + acc16[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, double instr_per_iter, int role_split = 0) {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
    hipMemset(cyc, 0, 64);
    k<MODE><<<1, threads>>>(out, cyc, role_split);
    k<MODE><<<1, threads>>>(out, cyc, role_split);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-58s waves=%d  ticks/iter:", name, threads / 64);
    for (int w = 0; w < threads / 64; ++w) printf(" %.1f", (double)h[w] / ITERS);
    printf("   (%.0f wave-instr per iter per wave)\n", instr_per_iter);
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("clock64() ticks (s_memtime, 100 MHz constant clock on gfx9: multiply by shader_clk/100MHz for cycles)\n");
    for (int threads : {64, 256, 512}) {
        run<1>("v_fma_f32 x8", threads, 8);
        run<8>("v_pk_fma_f32 x8 (16 fmas)", threads, 8);
        run<0>("v_exp_f32 x8", threads, 8);
        run<7>("4 x (exp + 3 fma)", threads, 16);
        run<2>("mfma 32x32x16 x4", threads, 4);
        run<6>("mfma 16x16x32 x4", threads, 4);
        run<3>("4 x (mfma32 + 8 fma) one wave", threads, 36);
        run<4>("4 x (mfma32 + 4 exp) one wave", threads, 20);
    }
    run<5>("8 waves: 0-3 mfma32 x4, 4-7 fma x32", 512, 0, 4);
    run<5>("8 waves: all fma x32 (role_split 0)", 512, 0, 0);
    run<5>("8 waves: all mfma (role_split 8)", 512, 0, 8);
    return 0;
}
