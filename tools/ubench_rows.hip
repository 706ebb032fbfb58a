// HBM efficiency of the GEMM's activation fetch pattern: every K-step a block pulls 128 bytes from each of its 256 rows
// (row stride ld bytes), next K-step the next 128 bytes, ... versus the same bytes fetched as longer runs per row.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rows.hip -o /tmp/ubench_rows && /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// RUN = bytes fetched contiguously per row per step (128, 256, 512); a step moves 256 rows x RUN bytes per block
template <int RUN>
__global__ void __launch_bounds__(512) k(const char* src, unsigned bytes_total, int ld, int ksteps, int rounds, int pace) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes_total, 0x00020000);
    constexpr int LPR = RUN / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int NI = 256 / (RPI * 8);      // instructions per wave per step
    for (int rd = 0; rd < rounds; ++rd) {
        const unsigned tile = (unsigned)(rd * gridDim.x + blockIdx.x);
        const unsigned row0 = tile * 256u;
        for (int ks = 0; ks < ksteps * 128 / RUN; ++ks) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const unsigned row = row0 + (unsigned)((i * 8 + wave) * RPI + lane / LPR);
                const unsigned o = row * (unsigned)ld + (unsigned)ks * RUN + (unsigned)(lane % LPR) * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + ((wave * NI + i) & 63) * 1024), 16, o, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
            for (int z = 0; z < pace; ++z) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
}

template <int RUN>
void run(const char* src, unsigned bytes, int ld, int ksteps, int pace) {
    auto kern = k<RUN>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int rows = (int)(bytes / ld);
    const int rounds = rows / 256 / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 512, 65536>>>(src, bytes, ld, ksteps, rounds, pace);
    hipEventRecord(e0);
    kern<<<256, 512, 65536>>>(src, bytes, ld, ksteps, rounds, pace);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)rounds * 256 * 256 * ksteps * 128.0;
    printf("run=%4d B/row/step  ld=%5d  pace=%d : %.3f ms  %.2f TB/s\n", RUN, ld, pace, ms, moved / (ms * 1e-3) / 1e12);
}

int main() {
    const unsigned bytes = 0x7f000000u;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    for (int pace : {0, 2}) {
        for (int ld : {5120, 2560, 640}) {
            const int ksteps = ld / 128;
            run<128>(src, bytes, ld, ksteps, pace);
            run<256>(src, bytes, ld, ksteps, pace);
            run<512>(src, bytes, ld, ksteps, pace);
        }
    }
    return 0;
}
