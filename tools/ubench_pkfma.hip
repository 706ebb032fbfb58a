// Minimal reproducer hunt for the run-dependent `v_pk_fma_f32 ... op_sel:[0,1,1]` (profiles/r04_pkfma_rootcause.md).
//
// tools/pkfma_variants.py showed on the pre-fix library: padding the four instructions with s_nop (before / after) changes nothing,
// replacing them by scalar v_fma_f32 on the SAME registers removes the error.  This program takes the instruction out of the GEMM:
// the exact register picture and instruction sequence of the failing epilogue fragment (four ds_read_b128 of two wave-private LDS
// strips, `s_waitcnt lgkmcnt(3)`, eight packed multiply-adds of which four carry op_sel:[0,1,1], dst = src0 = the registers the
// LDS read just filled) as ONE asm block with hard-wired registers, fed with random operands, every result compared bit for bit
// with scalar fmaf of the same operands.  Modes:
//   0  the sequence alone, 1 wave per SIMD              3  mode 1 with the op_sel form replaced by explicit (x, x) pairs (control)
//   1  the sequence, 2 waves per SIMD (both testing)    4  registers only: no LDS reads in front (operands moved in with v_mov)
//   2  2 waves per SIMD, the partner wave runs MFMAs + LDS traffic (what the GEMM's other waves do)
// Output: mismatching results per mode, split by 16-lane quarter and by low / high half of the packed result.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_pkfma.hip -o /tmp/ubench_pkfma && /tmp/ubench_pkfma [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float rnd(unsigned& s) { return (float)(int)(lcg(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }   // [-1, 1)

// counts[mode][quarter][half]
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* counts, int iters, int noisy_partner) {
    __shared__ __attribute__((aligned(16))) float strips[4][2][128];      // per wave: two strips of 128 floats (like sB / sS)
    __shared__ __attribute__((aligned(16))) _Float16 junk[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lg = lane >> 4;
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    if (MODE == 2 && noisy_partner && (blockIdx.x & 1)) {
        // partner: MFMA + LDS read/write stream, never checked
        h8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)rnd(seed); b[j] = (_Float16)rnd(seed); }
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 8; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            *reinterpret_cast<h8*>(&junk[((threadIdx.x + it) & 1023) * 8]) = a;
            a = *reinterpret_cast<const h8*>(&junk[((threadIdx.x * 7 + it) & 1023) * 8]);
        }
        if (acc[0] == 123.456f) counts[63] = 1;
        return;
    }
    float* sB = strips[wave][0];
    float* sS = strips[wave][1];
    unsigned long long bad[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        // wave-private strips, rewritten every iteration (lanes < 32 write 4 floats each, as the epilogue's lanes < WN / 4 do)
        if (lane < 32) {
            *reinterpret_cast<f4*>(sB + lane * 4) = f4{rnd(seed), rnd(seed), rnd(seed), rnd(seed)};
            *reinterpret_cast<f4*>(sS + lane * 4) = f4{rnd(seed), rnd(seed), rnd(seed), rnd(seed)};
        }
        const float c_even = rnd(seed), c_odd = rnd(seed), b_even = rnd(seed), b_odd = rnd(seed);   // (colsum, bias') of rows b = 0 / 1
        float acc_in[8], out[8];
        for (int i = 0; i < 8; ++i) acc_in[i] = rnd(seed) * 4.f;
        const unsigned a0 = (unsigned)(size_t)(sS + lg * 4) & 0xffff, a1 = (unsigned)(size_t)(sB + lg * 4) & 0xffff;   // LDS byte addresses
        // reference: v[r] = fma(acc, t[r], fma(u[r], colsum_odd, bias_odd)) with t from sB, u from sS; fragments a = 0 (offset 0) and a = 1 (offset 64 B)
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): strip writes are done (same wave, in order - belt and braces)
        float want[8];
        for (int f = 0; f < 2; ++f)
            for (int r = 0; r < 4; ++r) {
                const float t = sB[lg * 4 + f * 16 + r], u = sS[lg * 4 + f * 16 + r];
                want[f * 4 + r] = __builtin_fmaf(acc_in[f * 4 + r], t, __builtin_fmaf(u, c_odd, b_odd));
            }
        if (MODE == 4) {
            float u0[8], t0[8];
            for (int f = 0; f < 2; ++f)
                for (int r = 0; r < 4; ++r) { t0[f * 4 + r] = sB[lg * 4 + f * 16 + r]; u0[f * 4 + r] = sS[lg * 4 + f * 16 + r]; }
            asm volatile(
                "v_mov_b32 v4, %[ce]\n v_mov_b32 v5, %[co]\n v_mov_b32 v0, %[be]\n v_mov_b32 v1, %[bo]\n"
                "v_mov_b32 v52, %[i0]\n v_mov_b32 v53, %[i1]\n v_mov_b32 v54, %[i2]\n v_mov_b32 v55, %[i3]\n"
                "v_mov_b32 v48, %[i4]\n v_mov_b32 v49, %[i5]\n v_mov_b32 v50, %[i6]\n v_mov_b32 v51, %[i7]\n"
                "v_mov_b32 v60, %[u0]\n v_mov_b32 v61, %[u1]\n v_mov_b32 v62, %[u2]\n v_mov_b32 v63, %[u3]\n"
                "v_mov_b32 v120, %[u4]\n v_mov_b32 v121, %[u5]\n v_mov_b32 v122, %[u6]\n v_mov_b32 v123, %[u7]\n"
                "v_mov_b32 v66, %[t0]\n v_mov_b32 v67, %[t1]\n v_mov_b32 v68, %[t2]\n v_mov_b32 v69, %[t3]\n"
                "v_mov_b32 v124, %[t4]\n v_mov_b32 v125, %[t5]\n v_mov_b32 v126, %[t6]\n v_mov_b32 v127, %[t7]\n"
                "s_nop 4\n"
                "v_pk_fma_f32 v[60:61], v[60:61], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                "v_add_u32_e32 v57, v4, v5\n"
                "v_pk_fma_f32 v[52:53], v[52:53], v[66:67], v[60:61]\n"
                "v_pk_fma_f32 v[60:61], v[62:63], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[54:55], v[54:55], v[68:69], v[60:61]\n"
                "v_pk_fma_f32 v[60:61], v[120:121], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[48:49], v[48:49], v[124:125], v[60:61]\n"
                "v_pk_fma_f32 v[60:61], v[122:123], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[50:51], v[50:51], v[126:127], v[60:61]\n"
                "s_nop 1\n"
                "v_mov_b32 %[o0], v52\n v_mov_b32 %[o1], v53\n v_mov_b32 %[o2], v54\n v_mov_b32 %[o3], v55\n"
                "v_mov_b32 %[o4], v48\n v_mov_b32 %[o5], v49\n v_mov_b32 %[o6], v50\n v_mov_b32 %[o7], v51\n"
                : [o0] "=&v"(out[0]), [o1] "=&v"(out[1]), [o2] "=&v"(out[2]), [o3] "=&v"(out[3]), [o4] "=&v"(out[4]), [o5] "=&v"(out[5]),
                  [o6] "=&v"(out[6]), [o7] "=&v"(out[7])
                : [ce] "v"(c_even), [co] "v"(c_odd), [be] "v"(b_even), [bo] "v"(b_odd), [i0] "v"(acc_in[0]), [i1] "v"(acc_in[1]),
                  [i2] "v"(acc_in[2]), [i3] "v"(acc_in[3]), [i4] "v"(acc_in[4]), [i5] "v"(acc_in[5]), [i6] "v"(acc_in[6]), [i7] "v"(acc_in[7]),
                  [u0] "v"(u0[0]), [u1] "v"(u0[1]), [u2] "v"(u0[2]), [u3] "v"(u0[3]), [u4] "v"(u0[4]), [u5] "v"(u0[5]), [u6] "v"(u0[6]),
                  [u7] "v"(u0[7]), [t0] "v"(t0[0]), [t1] "v"(t0[1]), [t2] "v"(t0[2]), [t3] "v"(t0[3]), [t4] "v"(t0[4]), [t5] "v"(t0[5]),
                  [t6] "v"(t0[6]), [t7] "v"(t0[7])
                : "v0", "v1", "v4", "v5", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v57", "v60", "v61", "v62", "v63", "v66", "v67",
                  "v68", "v69", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        } else {
#define PKFMA_HEAD                                                                                                            \
            "v_mov_b32 v4, %[ce]\n v_mov_b32 v5, %[co]\n v_mov_b32 v0, %[be]\n v_mov_b32 v1, %[bo]\n"                          \
            "v_mov_b32 v52, %[i0]\n v_mov_b32 v53, %[i1]\n v_mov_b32 v54, %[i2]\n v_mov_b32 v55, %[i3]\n"                      \
            "v_mov_b32 v48, %[i4]\n v_mov_b32 v49, %[i5]\n v_mov_b32 v50, %[i6]\n v_mov_b32 v51, %[i7]\n"                      \
            "v_mov_b32 v58, %[a0]\n v_mov_b32 v59, %[a1]\n"                                                                   \
            "s_nop 1\n"                                                                                                       \
            "ds_read_b128 v[60:63], v58\n"                                                                                    \
            "ds_read_b128 v[120:123], v58 offset:64\n"                                                                        \
            "ds_read_b128 v[66:69], v59\n"                                                                                    \
            "ds_read_b128 v[124:127], v59 offset:64\n"
#define PKFMA_TAIL                                                                                                            \
            "s_nop 1\n"                                                                                                       \
            "v_mov_b32 %[o0], v52\n v_mov_b32 %[o1], v53\n v_mov_b32 %[o2], v54\n v_mov_b32 %[o3], v55\n"                      \
            "v_mov_b32 %[o4], v48\n v_mov_b32 %[o5], v49\n v_mov_b32 %[o6], v50\n v_mov_b32 %[o7], v51\n"
#define PKFMA_OPS                                                                                                              \
            : [o0] "=&v"(out[0]), [o1] "=&v"(out[1]), [o2] "=&v"(out[2]), [o3] "=&v"(out[3]), [o4] "=&v"(out[4]), [o5] "=&v"(out[5]),  \
              [o6] "=&v"(out[6]), [o7] "=&v"(out[7])                                                                           \
            : [ce] "v"(c_even), [co] "v"(c_odd), [be] "v"(b_even), [bo] "v"(b_odd), [i0] "v"(acc_in[0]), [i1] "v"(acc_in[1]),     \
              [i2] "v"(acc_in[2]), [i3] "v"(acc_in[3]), [i4] "v"(acc_in[4]), [i5] "v"(acc_in[5]), [i6] "v"(acc_in[6]),           \
              [i7] "v"(acc_in[7]), [a0] "v"(a0), [a1] "v"(a1)                                                                  \
            : "v0", "v1", "v4", "v5", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v57", "v58", "v59", "v60", "v61", "v62",   \
              "v63", "v66", "v67", "v68", "v69", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", \
              "v131", "memory"
            if (MODE == 3) {
                asm volatile(PKFMA_HEAD
                             "v_mov_b32 v128, v5\n v_mov_b32 v129, v5\n v_mov_b32 v130, v1\n v_mov_b32 v131, v1\n"
                             "s_waitcnt lgkmcnt(3)\n"
                             "v_pk_fma_f32 v[60:61], v[60:61], v[128:129], v[130:131]\n"
                             "v_add_u32_e32 v57, v4, v5\n"
                             "s_waitcnt lgkmcnt(1)\n"
                             "v_pk_fma_f32 v[52:53], v[52:53], v[66:67], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[62:63], v[128:129], v[130:131]\n"
                             "s_nop 0\n"
                             "v_pk_fma_f32 v[54:55], v[54:55], v[68:69], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[120:121], v[128:129], v[130:131]\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             "v_pk_fma_f32 v[48:49], v[48:49], v[124:125], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[122:123], v[128:129], v[130:131]\n"
                             "s_nop 0\n"
                             "v_pk_fma_f32 v[50:51], v[50:51], v[126:127], v[60:61]\n" PKFMA_TAIL PKFMA_OPS);
            } else {
                asm volatile(PKFMA_HEAD
                             "s_waitcnt lgkmcnt(3)\n"
                             "v_pk_fma_f32 v[60:61], v[60:61], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                             "v_add_u32_e32 v57, v4, v5\n"
                             "s_waitcnt lgkmcnt(1)\n"
                             "v_pk_fma_f32 v[52:53], v[52:53], v[66:67], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[62:63], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                             "s_nop 0\n"
                             "v_pk_fma_f32 v[54:55], v[54:55], v[68:69], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[120:121], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             "v_pk_fma_f32 v[48:49], v[48:49], v[124:125], v[60:61]\n"
                             "v_pk_fma_f32 v[60:61], v[122:123], v[4:5], v[0:1] op_sel:[0,1,1]\n"
                             "s_nop 0\n"
                             "v_pk_fma_f32 v[50:51], v[50:51], v[126:127], v[60:61]\n" PKFMA_TAIL PKFMA_OPS);
            }
        }
        for (int i = 0; i < 8; ++i)
            if (__builtin_bit_cast(unsigned, out[i]) != __builtin_bit_cast(unsigned, want[i])) ++bad[i & 1];
    }
    if (bad[0]) atomicAdd(&counts[(MODE * 4 + lg) * 2 + 0], bad[0]);
    if (bad[1]) atomicAdd(&counts[(MODE * 4 + lg) * 2 + 1], bad[1]);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* d;
    hipMalloc(&d, 64 * sizeof(unsigned long long));
    hipMemset(d, 0, 64 * sizeof(unsigned long long));
    const int cus = 256;
    // mode 0: one block of 256 threads per CU (1 wave per SIMD); modes 1-4: 2 blocks per CU (2 waves per SIMD)
    hipLaunchKernelGGL(k<0>, dim3(cus), dim3(256), 0, 0, d, iters, 0);
    hipLaunchKernelGGL(k<1>, dim3(cus * 2), dim3(256), 0, 0, d, iters, 0);
    hipLaunchKernelGGL(k<2>, dim3(cus * 2), dim3(256), 0, 0, d, iters, 1);
    hipLaunchKernelGGL(k<3>, dim3(cus * 2), dim3(256), 0, 0, d, iters, 0);
    hipLaunchKernelGGL(k<4>, dim3(cus * 2), dim3(256), 0, 0, d, iters, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    unsigned long long h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"0 sequence, 1 wave/SIMD", "1 sequence, 2 waves/SIMD", "2 sequence, partner wave in MFMA + LDS traffic", "3 control: explicit (x, x) pairs", "4 registers only (no LDS reads in front)"};
    printf("v_pk_fma_f32 op_sel:[0,1,1] in isolation: %d iterations x 8 results per lane; mismatches vs scalar fmaf of the same operands\n", iters);
    for (int m = 0; m < 5; ++m) {
        unsigned long long tot = 0;
        for (int q = 0; q < 8; ++q) tot += h[m * 8 + q];
        printf("mode %-48s total %10llu   by lane quarter (low half, high half):", names[m], tot);
        for (int q = 0; q < 4; ++q) printf("  q%d (%llu, %llu)", q, h[(m * 4 + q) * 2], h[(m * 4 + q) * 2 + 1]);
        printf("\n");
    }
    return 0;
}
