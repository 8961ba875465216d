// Shared host/device helpers for libtmf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/tmf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void tmf_set_error(const char* fmt, ...);

#define TMF_REQUIRE_PTR(p)                                                     \
    do {                                                                       \
        if ((p) == nullptr) {                                                  \
            tmf_set_error("%s: argument '%s' is NULL", __func__, #p);          \
            return TMF_E_NULL;                                                 \
        }                                                                      \
    } while (0)

#define TMF_REQUIRE_ALIGNED(p)                                                 \
    do {                                                                       \
        if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) {                     \
            tmf_set_error("%s: argument '%s' is not 16-byte aligned", __func__, #p); \
            return TMF_E_ALIGN;                                                \
        }                                                                      \
    } while (0)

#define TMF_REQUIRE(cond, code, ...)                                           \
    do {                                                                       \
        if (!(cond)) {                                                         \
            tmf_set_error(__VA_ARGS__);                                        \
            return (code);                                                     \
        }                                                                      \
    } while (0)

// Returns 0 or the (positive) hipError_t of the launch that just happened.
static inline int tmf_launch_result(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tmf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return TMF_OK;
}

// Opt a kernel into > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU).
template <typename K>
static inline int tmf_allow_lds(K kernel, size_t bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        tmf_set_error("%s: cannot reserve %zu B of LDS: %s", what, bytes, hipGetErrorString(e));
        return (int)e;
    }
    return TMF_OK;
}

static inline int tmf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// MI355X dispatches workgroup b to XCD b % 8 (speed hint only, MI355X_MICROARCH.md).
// Map the launch index so that each XCD (= one private L2) walks a contiguous range
// of tiles: neighbouring tiles share their halo in that L2.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
#endif
