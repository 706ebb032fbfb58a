// HBM efficiency of the GEMM's activation fetch pattern: every K-step a block pulls 128 bytes from each of its 256 rows
// (row stride ld bytes), next K-step the next 128 bytes, ... versus the same bytes fetched as longer runs per row.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rows.hip -o /tmp/ubench_rows && /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// RUN = bytes fetched contiguously per row per step (128, 256, 512); a step moves 256 rows x RUN bytes per block
template <int RUN>
__global__ void __launch_bounds__(512) k(const char* src, unsigned bytes_total, int ld, int ksteps, int rounds, int pace, int mode, int busy, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    h8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(threadIdx.x * 1e-3f); fb[j] = (_Float16)(j * 1e-2f); }
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes_total, 0x00020000);
    constexpr int LPR = RUN / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int NI = 256 / (RPI * 8);      // instructions per wave per step
    for (int rd = 0; rd < rounds; ++rd) {
        const unsigned tile = (unsigned)(rd * gridDim.x + blockIdx.x);
        const unsigned row0 = tile * 256u;
        const int nks = ksteps * 128 / RUN;
        auto issue = [&](int ks) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const unsigned row = row0 + (unsigned)((i * 8 + wave) * RPI + lane / LPR);
                const unsigned o = row * (unsigned)ld + (unsigned)ks * RUN + (unsigned)(lane % LPR) * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + ((wave * NI + i + (ks & 1) * 32) & 63) * 1024), 16, o, 0, 0, 0);
            }
        };
        // mode 0: every wave on its own (issue, wait, repeat)          mode 1: block-wide lockstep (barrier per step, as a GEMM K-step)
        // mode 2: lockstep, but the next step's loads are issued before waiting for this step's (two stages in flight)
        auto compute = [&]() {
            if (busy == 0) { for (int z = 0; z < pace; ++z) __builtin_amdgcn_s_sleep(1); }
            else {           // the same time spent issuing MFMAs (64 cycles per pace unit: 4 x 16x16x32)
                for (int z = 0; z < pace; ++z) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[q], 0, 0, 0);
                }
            }
        };
        if (mode == 3) {
            // register-staged variant: the next step's rows are fetched into VGPRs (plain buffer loads) before this step's
            // compute and written to LDS with ds_write afterwards - one stage in flight, like a classic double-buffered GEMM
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            u4v st[NI];
            auto fetch = [&](int ks) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const unsigned row = row0 + (unsigned)((i * 8 + wave) * RPI + lane / LPR);
                    st[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, row * (unsigned)ld + (unsigned)ks * RUN + (unsigned)(lane % LPR) * 16u, 0, 0);
                }
            };
            fetch(0);
            for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    *reinterpret_cast<u4v*>(smem + ((wave * NI + i + (ks & 1) * 32) & 63) * 1024 + lane * 16) = st[i];
                if (ks + 1 < nks) fetch(ks + 1);
                __syncthreads();
                compute();
            }
        } else {
        if (mode == 2) issue(0);
        for (int ks = 0; ks < nks; ++ks) {
            if (mode == 2) {
                if (ks + 1 < nks) { issue(ks + 1); __builtin_amdgcn_s_waitcnt(0x0f70 | NI); }   // all but the youngest NI
                else __builtin_amdgcn_s_waitcnt(0x0f70);
            } else {
                issue(ks);
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
            if (mode >= 1) __syncthreads();
            compute();
        }
        }
    }
    if (busy) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#endif
}

template <int RUN>
void run(const char* src, unsigned bytes, int ld, int ksteps, int pace, int mode, int busy = 0) {
    static float* sink = nullptr; if (!sink) hipMalloc(&sink, 256 * 512 * 4);
    auto kern = k<RUN>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int rows = (int)(bytes / ld);
    const int rounds = rows / 256 / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 512, 65536>>>(src, bytes, ld, ksteps, rounds, pace, mode, busy, sink);
    hipEventRecord(e0);
    kern<<<256, 512, 65536>>>(src, bytes, ld, ksteps, rounds, pace, mode, busy, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)rounds * 256 * 256 * ksteps * 128.0;
    printf("run=%4d B/row/step  ld=%5d  pace=%2d %s mode=%d : %.3f ms  %.2f TB/s  (%.2f us per step)\n", RUN, ld, pace, busy ? "mfma " : "sleep", mode, ms, moved / (ms * 1e-3) / 1e12, ms * 1e3 / (rounds * ksteps * 128.0 / RUN));
}

int main() {
    const unsigned bytes = 0x7f000000u;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    // pace: units of ~64 cycles of "compute" per step (s_sleep(1), or 4 MFMA 16x16x32 per wave = 2 waves/SIMD x 64 cycles);
    // a GEMM K-step is ~80 MFMAs per wave = pace 20 in MFMA units
    for (int busy = 1; busy < 2; ++busy)
        for (int pace : {0, 10, 20, 40})
            for (int mode = 0; mode < 4; ++mode) run<128>(src, bytes, 5120, 40, pace, mode, busy);
    return 0;
}
