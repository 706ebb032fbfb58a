// How fast can a CU pull data with `buffer_load_dwordx4 ... lds` (DMA into LDS) vs. plain buffer loads into VGPRs?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dma.hip -o /tmp/ubench_dma && /tmp/ubench_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// every wave fetches `iters` x `batch` KB; footprint per block = fp_kb KB (re-read cyclically -> L2 resident when small)
template <int MODE>
__global__ void __launch_bounds__(512) k(const char* src, long long bytes_total, int fp_bytes, int iters, int batch, unsigned* sink, long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)(bytes_total > 0x7fffffffll ? 0x7fffffff : bytes_total), 0x00020000);
    const unsigned base = (unsigned)(((long long)blockIdx.x * fp_bytes) % (bytes_total - fp_bytes));
    unsigned off = 0;
    u4v acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        for (int b = 0; b < batch; ++b) {
            const unsigned o = base + (off + (unsigned)(wave * 1024 + lane * 16)) % (unsigned)fp_bytes;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + ((wave * batch + b) & 63) * 1024), 16, o, 0, 0, 0);
            } else {
                const u4v v = __builtin_amdgcn_raw_buffer_load_b128(srd, o, 0, 0);
                acc += v;
            }
            off += nw * 1024;
        }
        if (MODE == 0) __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    const long long t1 = clock64();
    if (MODE == 1) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int MODE>
void run(const char* name, const char* src, long long bytes_total, int blocks, int threads, int fp_kb, int iters, int batch) {
    unsigned* sink; long long* cyc;
    hipMalloc(&sink, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    auto kern = k<MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads, 65536>>>(src, bytes_total, fp_kb * 1024, iters, batch, sink, cyc);
    hipEventRecord(e0);
    kern<<<blocks, threads, 65536>>>(src, bytes_total, fp_kb * 1024, iters, batch, sink, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    const double per_block = (double)(threads / 64) * iters * batch * 1024.0;
    printf("%-26s blocks=%3d waves=%2d footprint/block=%5d KB batch=%2d : %6.1f B/clk/CU  (%.2f TB/s aggregate, %.3f ms)\n", name, blocks,
           threads / 64, fp_kb, batch, per_block / avg, per_block * blocks / (ms * 1e-3) / 1e12, ms);
    hipFree(sink); hipFree(cyc);
}

int main() {
    const long long bytes = 2ll << 30;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    for (int waves : {4, 8, 16}) {
        const int thr = waves * 64;
        for (int batch : {4, 9, 18}) {
            run<0>("DMA->LDS, L2 resident", src, bytes, 256, thr, 64, 4000 / batch, batch);
            run<1>("load->VGPR, L2 resident", src, bytes, 256, thr, 64, 4000 / batch, batch);
        }
        run<0>("DMA->LDS, streaming", src, bytes, 256, thr, 4096, 400, 9);
        run<1>("load->VGPR, streaming", src, bytes, 256, thr, 4096, 400, 9);
    }
    return 0;
}
