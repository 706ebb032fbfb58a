// Shared device/host helpers for libvcx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/vcx.h"
#include "vcx_ablate.h"

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// ---- error plumbing (api.hip) ----
void vcx_set_error(const char* fmt, ...);
int vcx_check_launch(const char* what);

#define VCX_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            vcx_set_error(__VA_ARGS__);   \
            return VCX_EINVAL;            \
        }                                 \
    } while (0)

// ---- kernels that need more than the default 64 KB of dynamic LDS ----
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), callable from any host thread: one instance per
// kernel (function-local static), a bit per device already prepared.  A process that drives several GPUs prepares each of them.
struct VcxLdsAttr {
    std::atomic<unsigned> done{0};
    // returns false (and sets the error text) if the runtime refuses the reservation
    bool ensure(const void* kernel, int bytes, const char* who) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (done.load(std::memory_order_acquire) & bit) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            vcx_set_error("%s: cannot reserve %d bytes of LDS", who, bytes);
            return false;
        }
        done.fetch_or(bit, std::memory_order_release);
        return true;
    }
};

// ---- experiment knobs (api.hip; include/vcx.h VCX_TUNE_*) ----
int vcx_tune(int knob);

// ---- profiling (api.hip) ----
enum { VCX_FAM_GEMM = 0, VCX_FAM_FLASH = 1, VCX_FAM_TATTN = 2, VCX_FAM_GN = 3, VCX_FAM_LN = 4, VCX_FAM_ELT = 5 };
struct VcxProfScope {
    int rec;
    hipStream_t s;
    VcxProfScope(int family, hipStream_t stream, double flops, double bytes);
    ~VcxProfScope();
};

// ---- device helpers ----
__device__ __forceinline__ float vcx_silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float vcx_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float vcx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
