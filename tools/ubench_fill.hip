// How many VALU "fillers" ride for free between the MFMAs of ONE wave's stream - with the wave alone on its SIMD and with a second
// wave of the same stream beside it?  The question behind a deferred (software-pipelined) GEMM epilogue: the GELU / pack / store
// work of tile i issued between the MFMAs of tile i+1.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o /tmp/ubench_fill && /tmp/ubench_fill
// Every stream is inline asm (hipcc neither reorders nor packs it).  Groups: 1 MFMA + NF fillers, 8 groups per loop iteration on 8
// independent accumulators.  Output: shader cycles (s_memtime) per group, per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define ITERS 1000

// KIND 0: v_fma_f32 fillers, 1: v_exp_f32, 2: half fma half exp (1 exp per 4 fillers), 3: v_cvt_pk_f16_f32 ... ; MF 16 or 32
template <int MF, int NF, int KIND>
__global__ void __launch_bounds__(512) kfill(float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f4 a16[8];
    f16v a32[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a16[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a32[i][j] = 0.f;
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f); b[j] = (_Float16)(j * 1e-2f); }
    const float c1 = 1.0001f, c2 = 0.5f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v vp[4], cp = {1.0001f, 0.5f};
    for (int i = 0; i < 4; ++i) vp[i] = f2v{threadIdx.x * 1e-3f, (float)i};
    f2v ones4 = {1.875f, 1.875f};      // (bit pattern irrelevant for timing)
    asm volatile("" : "+v"(cp), "+v"(ones4));
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MF == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a16[g]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a32[g & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int r = (g * NF + f) & 7;
                const bool ex = KIND == 1 || (KIND == 2 && (f & 3) == 0);
                if (ex) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
                else if (KIND == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[r]) : "v"(c1), "v"(c2));      // round 5: row sums of packed fp16 probabilities
                else if (KIND == 5) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[r]) : "v"(c1), "v"(c2));
                else if (KIND == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
                // round 6 (what the flash kernel's softmax stream could be made of): packed fp32 add, an add with a compiler-style s_nop 0 behind
                // it, a 4x4x4 MFMA with a ones operand (sums four packed fp16 values per lane on the matrix pipe in 2 passes), max3, half swap
                else if (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(vp[r & 3]) : "v"(cp));
                else if (KIND == 8) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(v[r]) : "v"(c1));
                else if (KIND == 9) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(a16[r]) : "v"(ones4), "v"(cp));
                else if (KIND == 10) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(c1), "v"(c2));
                else if (KIND == 11) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 4) & 7]));
                else if (KIND == 12) asm volatile("s_nop 0");
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(c1), "v"(c2));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + a16[i][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += a32[i][0] + a32[i][9] + vp[i][0] + vp[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// MFMA-only wave beside a filler-only wave on the same SIMD (waves w and w + 4 of a 512-thread block), fillers given priority or not
template <int PRIO>
__global__ void __launch_bounds__(512) ksplit(float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f4 a16[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a16[i] = f4{0.f, 0.f, 0.f, 0.f};
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f); b[j] = (_Float16)(j * 1e-2f); }
    const float c1 = 1.0001f, c2 = 0.5f;
    __syncthreads();
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(2);
    if (PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(2);
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < ITERS; ++it)
#pragma unroll
            for (int g = 0; g < 8; ++g) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a16[g]) : "v"(a), "v"(b));
    } else {
        for (int it = 0; it < ITERS; ++it)
#pragma unroll
            for (int g = 0; g < 32; ++g) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[g & 7]) : "v"(c1), "v"(c2));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + a16[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <class K>
void run(K kern, const char* name, int threads, int groups) {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
    hipMemset(cyc, 0, 64);
    kern<<<1, threads>>>(out, cyc);
    kern<<<1, threads>>>(out, cyc);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-44s waves=%d  cycles/group:", name, threads / 64);
    for (int w = 0; w < threads / 64; w += (threads > 256 ? 4 : 1)) printf(" %.1f", (double)h[w] / (ITERS * groups));
    printf("\n");
    hipFree(out); hipFree(cyc);
}

#define SWEEP(MF, KIND, label)                                                                      \
    for (int threads : {256, 512}) {                                                                \
        run(kfill<MF, 0, KIND>, label " + 0", threads, 8); run(kfill<MF, 1, KIND>, label " + 1", threads, 8); \
        run(kfill<MF, 2, KIND>, label " + 2", threads, 8); run(kfill<MF, 3, KIND>, label " + 3", threads, 8); \
        run(kfill<MF, 4, KIND>, label " + 4", threads, 8); run(kfill<MF, 6, KIND>, label " + 6", threads, 8); \
        run(kfill<MF, 8, KIND>, label " + 8", threads, 8);                                          \
    }

#define SWEEP1(MF, KIND, label)                                                                     \
    for (int threads : {256}) {                                                                     \
        run(kfill<MF, 0, KIND>, label " + 0", threads, 8); run(kfill<MF, 2, KIND>, label " + 2", threads, 8); \
        run(kfill<MF, 4, KIND>, label " + 4", threads, 8); run(kfill<MF, 6, KIND>, label " + 6", threads, 8); \
        run(kfill<MF, 8, KIND>, label " + 8", threads, 8); run(kfill<MF, 12, KIND>, label " + 12", threads, 8); \
    }

int main(int argc, char** argv) {
    printf("cycles per group = 1 MFMA + NF fillers; 256 threads = one wave per SIMD, 512 = two (first wave of each SIMD pair shown: w0, w4)\n");
    if (argc > 1) {      // round 6: the candidates for the flash kernel's softmax stream, one wave per SIMD, beside 32x32x16 MFMAs
        SWEEP1(32, 6, "mfma32x32x16 + v_add_f32")
        SWEEP1(32, 7, "mfma32x32x16 + v_pk_add_f32")
        SWEEP1(32, 8, "mfma32x32x16 + (v_add_f32, s_nop 0)")
        SWEEP1(32, 12, "mfma32x32x16 + s_nop 0")
        SWEEP1(32, 9, "mfma32x32x16 + v_mfma_4x4x4 (ones)")
        SWEEP1(32, 10, "mfma32x32x16 + v_max3_f32")
        SWEEP1(32, 11, "mfma32x32x16 + v_permlane32_swap")
        SWEEP1(32, 1, "mfma32x32x16 + v_exp_f32")
        SWEEP1(32, 3, "mfma32x32x16 + v_cvt_pk")
        return 0;
    }
    SWEEP(16, 0, "mfma16x16x32 + fma")
    SWEEP(16, 2, "mfma16x16x32 + (1 exp : 3 fma)")
    SWEEP(16, 3, "mfma16x16x32 + cvt_pk")
    SWEEP(32, 0, "mfma32x32x16 + fma")
    SWEEP(32, 2, "mfma32x32x16 + (1 exp : 3 fma)")
    SWEEP(32, 6, "mfma32x32x16 + v_add_f32")
    SWEEP(32, 4, "mfma32x32x16 + v_dot2c_f32_f16")
    SWEEP(32, 5, "mfma32x32x16 + v_dot2_f32_f16")
    SWEEP(32, 3, "mfma32x32x16 + cvt_pk")
    run(ksplit<0>, "w0-3 mfma16 x8 | w4-7 fma x32, prio 0", 512, 8);
    run(ksplit<1>, "same, fma waves s_setprio 2", 512, 8);
    run(ksplit<2>, "same, mfma waves s_setprio 2", 512, 8);
    return 0;
}
