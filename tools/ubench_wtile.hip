// L2 behaviour of the GEMM's weight-tile fetch: every block re-reads the same 320 x 128-byte slice per K-step.
// Row-major weights put those 320 lines 2*K bytes apart (few L2 channels); a K-step-major packing makes them contiguous.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_wtile.hip -o /tmp/ubench_wtile && /tmp/ubench_wtile
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void __launch_bounds__(512) k(const char* src, unsigned bytes_total, int ld, int ksteps, int reps, int packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes_total, 0x00020000);
    for (int rp = 0; rp < reps; ++rp)
        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned row = (unsigned)((i * 8 + wave) * 8 + lane / 8);            // 0..319
                const unsigned o = packed ? ((unsigned)ks * 320u + row) * 128u + (unsigned)(lane % 8) * 16u
                                          : row * (unsigned)ld + (unsigned)ks * 128u + (unsigned)(lane % 8) * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + ((wave * 5 + i) & 63) * 1024), 16, o, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
#endif
}

int main() {
    const unsigned bytes = 64u << 20;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int ld : {5120, 2560, 1280, 640, 10240}) {
        const int ksteps = ld / 128;
        for (int packed = 0; packed < 2; ++packed) {
            const int reps = 400 / ksteps + 1;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<256, 512, 65536>>>(src, bytes, ld, ksteps, reps, packed);
            hipEventRecord(e0);
            k<<<256, 512, 65536>>>(src, bytes, ld, ksteps, reps, packed);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double steps = (double)reps * ksteps;
            printf("K=%5d (row stride %5d B) %s: %.3f us per K-step (40 KB per CU, 256 CUs)  %.1f TB/s from L2\n", ld / 2, ld,
                   packed ? "packed  " : "rowmajor", ms * 1e3 / steps, 256.0 * 40960 * steps / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
