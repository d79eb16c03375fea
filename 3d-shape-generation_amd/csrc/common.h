// Shared helpers for the gfx950 library (internal; the public ABI is include/pcd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pcd_hip.h"

namespace pcd {

void set_error(const char* fmt, ...);

#define PCD_CHECK_ARG(cond)                                                        \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ::pcd::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
            return PCD_ERR_ARG;                                                    \
        }                                                                          \
    } while (0)

#define PCD_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::pcd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
                             hipGetErrorString(e__));                                    \
            return PCD_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define PCD_CHECK_LAUNCH() PCD_CHECK_HIP(hipGetLastError())

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ half_t to_half_sat(float v) {
    // keep fp16 activations finite: clamp to the largest normal instead of +-inf
    v = __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
    return (half_t)v;
}

// LDS-DMA of 16 bytes per lane (1 KiB per wave-instruction): global `g` (per-lane address) -> LDS at the wave-uniform
// address `lds_wave_base` + lane * 16.  Issued from inline asm so the compiler does not serialise later LDS reads behind
// it; the caller owns the `s_waitcnt vmcnt` and the barrier that make the bytes visible.
__device__ __forceinline__ void lds_dma16(const void* g, const void* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory", "m0");
}


// Dynamic LDS above 64 KB has to be allowed per kernel AND per device: once per (device, kernel), whichever device is current at the launch.
struct PcdLdsOnce { unsigned long long done = 0; };          // one bit per device ordinal below 64 (above: set every time)
static inline hipError_t pcd_allow_lds(PcdLdsOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((once.done >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) once.done |= 1ull << dev;
    return e;
}

}  // namespace pcd
