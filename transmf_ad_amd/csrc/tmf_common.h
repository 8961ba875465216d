// Shared host/device helpers for libtmf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/tmf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef __HIPCC__
typedef __bf16 tmf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tmf_bf16x8 __attribute__((ext_vector_type(8)));
// two floats -> two bf16 (round-to-nearest-even, a in the low half): one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned int tmf_pack_bf16(float a, float b) {
    const tmf_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
// Typed activation I/O: float tensors, or bf16 tensors (unsigned short storage) widened to / rounded from fp32 registers.
typedef unsigned short tmf_bf16_t;
typedef unsigned int tmf_u32x2 __attribute__((ext_vector_type(2)));
typedef float tmf_f32x1 __attribute__((ext_vector_type(1)));
template <typename T, int VEC> struct TmfIO;
template <> struct TmfIO<float, 4> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    // streaming (read-once / write-once) forms: keep the lines out of L2 so that re-used data (halos, weights) stay
    static __device__ __forceinline__ f32x4 ld_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
    static __device__ __forceinline__ void st_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
};
template <> struct TmfIO<float, 1> {
    static __device__ __forceinline__ tmf_f32x1 ld(const float* p) { return *reinterpret_cast<const tmf_f32x1*>(p); }
    static __device__ __forceinline__ void st(float* p, tmf_f32x1 v) { *p = v[0]; }
};
template <> struct TmfIO<tmf_bf16_t, 4> {
    static __device__ __forceinline__ f32x4 ld(const tmf_bf16_t* p) {
        const tmf_u32x2 r = *reinterpret_cast<const tmf_u32x2*>(p);
        return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xFFFF0000u),
                     __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xFFFF0000u)};
    }
    static __device__ __forceinline__ void st(tmf_bf16_t* p, f32x4 v) {
        *reinterpret_cast<tmf_u32x2*>(p) = tmf_u32x2{tmf_pack_bf16(v[0], v[1]), tmf_pack_bf16(v[2], v[3])};
    }
};
typedef float tmf_f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tmf_u32x4 __attribute__((ext_vector_type(4)));
template <> struct TmfIO<float, 8> {
    static __device__ __forceinline__ tmf_f32x8 ld(const float* p) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return tmf_f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    static __device__ __forceinline__ void st(float* p, tmf_f32x8 v) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};
template <> struct TmfIO<tmf_bf16_t, 8> {          // one 16-byte access per lane
    static __device__ __forceinline__ tmf_f32x8 ld(const tmf_bf16_t* p) {
        const tmf_u32x4 r = *reinterpret_cast<const tmf_u32x4*>(p);
        tmf_f32x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xFFFF0000u);
        }
        return v;
    }
    static __device__ __forceinline__ void st(tmf_bf16_t* p, tmf_f32x8 v) {
        *reinterpret_cast<tmf_u32x4*>(p) = tmf_u32x4{tmf_pack_bf16(v[0], v[1]), tmf_pack_bf16(v[2], v[3]),
                                                      tmf_pack_bf16(v[4], v[5]), tmf_pack_bf16(v[6], v[7])};
    }
};
template <> struct TmfIO<tmf_bf16_t, 1> {
    static __device__ __forceinline__ tmf_f32x1 ld(const tmf_bf16_t* p) {
        return tmf_f32x1{__builtin_bit_cast(float, (unsigned int)(*p) << 16)};
    }
    static __device__ __forceinline__ void st(tmf_bf16_t* p, tmf_f32x1 v) {
        *p = (tmf_bf16_t)(tmf_pack_bf16(v[0], 0.f) & 0xFFFFu);
    }
};
#endif

void tmf_set_error(const char* fmt, ...);
extern int tmf_g_debug;          // conv3d_bf16.hip: timing-ablation bits (tmf_set_option("debug", ..))
extern int tmf_g_wgrad_tr;       // conv3d_bf16.hip: bf16 weight-gradient kernel choice, tmf_set_option("wgrad_tr", 0 never | 1 where faster | 2 wherever possible: the transposing-read kernel)
// conv3d_mfma.hip: tmf_conv3d_fwd / tmf_conv3d_stat_blocks with a per-call minimum for the "conv_rt" mode (snet_path.hip: TMF_SNET_ALONE)
int tmf_conv3d_fwd_mode(const float* x, const float* w, float* z, float* stat_partial, int B, int D, int H, int W, int cin,
                        int cout, int ksize, int rt_min, void* stream);
int tmf_conv3d_stat_blocks_mode(int B, int D, int H, int W, int cin, int cout, int ksize, int rt_min);
int tmf_c1_split_set(int v);     // conv1_fused.hip: tmf_set_option("c1_split", 0 | 1): z of the first block (fp32) as exact 3-way bf16 splits
int tmf_c1_gram_set(int v);      // conv1_gram.hip: tmf_set_option("c1_gram", 0 | 1): the first block through the tap Gram matrix of its input
int tmf_wino_p_set(int v);       // conv3d_wino.hip: tmf_set_option("wino_p", 0 | 1): two-waves-per-SIMD / persistent one-wave-per-SIMD forward kernel
// per-call algorithm choice (tmf_snet_desc.flags & TMF_SNET_ALGO): the whole-encoder entries set it for the calling thread while they
// plan and enqueue; the option getters (tmf_conv_wino_mode, wino_p_mode, wino_x_mode, c1_gram_mode) look here first
int  tmf_algo_override(void);            // 0, or a flags word with TMF_SNET_ALGO set
void tmf_algo_override_set(int flags);
struct TmfAlgoScope {
    int prev;
    explicit TmfAlgoScope(int flags) : prev(tmf_algo_override()) { if (flags & TMF_SNET_ALGO) tmf_algo_override_set(flags); }
    ~TmfAlgoScope() { tmf_algo_override_set(prev); }
    TmfAlgoScope(const TmfAlgoScope&) = delete;
    TmfAlgoScope& operator=(const TmfAlgoScope&) = delete;
};
int tmf_wino_x_set(int v);       // conv3d_winox.hip: tmf_set_option("wino_x", 0 | 1): Winograd forward / data gradient as exact 3-way bf16 splits on the bf16 matrix pipe
int tmf_winox_takes(int B, int D, int H, int W, int cin, int cout, int geom);
long tmf_winox_items(int D, int H, int W, int* swap);     // items per sample (the better of the two item orientations)
int tmf_winox_launch(const char* what, const float* x, const unsigned short* u3, float* z, float* stat_partial, int B, int D, int H,
                     int W, int cin, int cout, int ncu, hipStream_t stream, const float* scale = nullptr, const float* shift = nullptr,
                     float slope = 0.f, int pool = 0);     // scale != NULL: the eval-mode block (y = LeakyReLU(scale z + shift), pool none | max)
int tmf_conv_wino_set(int v);    // conv3d_wino.hip: tmf_set_option("conv_wino", 0 | 1 | 2)
extern int tmf_g_bf16_dma;       // conv3d_bf16.hip: LDS-DMA form of the large-brick bf16 forward kernel (tmf_set_option("bf16_dma", 0 | 1))
extern int tmf_g_bf16_v2;        // conv3d_bf16.hip: kernel choice of the bf16 forward (tmf_set_option("bf16_v2", ..))

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
    // once per kernel instantiation and size (K is a distinct function-pointer VALUE per call site, so key on it)
    // ... and on the device: the attribute belongs to the function ON ONE DEVICE, so a second GPU driven from the
    // same thread needs its own call
    static thread_local const void* last_fn[8] = {};
    static thread_local size_t last_sz[8] = {};
    static thread_local int last_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    const void* fn = reinterpret_cast<const void*>(kernel);
    const unsigned slot = (unsigned)((reinterpret_cast<uintptr_t>(fn) >> 4) & 7u);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (last_fn[slot] == fn && last_dev[slot] == dev && last_sz[slot] >= bytes) return TMF_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        tmf_set_error("%s: cannot reserve %zu B of LDS: %s", what, bytes, hipGetErrorString(e));
        return (int)e;
    }
    last_fn[slot] = fn;
    last_sz[slot] = bytes;
    last_dev[slot] = dev;
    return TMF_OK;
}

static inline int tmf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Two-stage slab reduction plan: `groups` first-stage groups (1 = single stage straight into `out`).
static inline int tmf_reduce_groups(int nsplit) {
    if (nsplit <= 64) return 1;
    int g = nsplit / 16;
    return g > 32 ? 32 : g;
}

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

// out[g][e] = sum over the slabs s of group g of partial[s][e]   (fixed order -> deterministic; fp64
// accumulation: the sums cancel heavily).  Block = 64 consecutive elements x TMF_RED_LANES slab lanes.
#define TMF_RED_LANES 16
// tcin > 0: the n = T * cin * cout sums are a tap-major weight gradient [t][ci][co] and are stored in the reference's
// nn.Conv3d layout (Cout, Cin, k, k, k) = [co][ci][t] instead (T = n / (tcin * tcout)); <= 3.5 MB, the scattered
// 4-byte stores are noise next to the slab reads.
static __global__ __launch_bounds__(64 * TMF_RED_LANES) void tmf_slab_reduce_kernel(
    const float* __restrict__ partial, float* __restrict__ out, int nsplit, long n, int slabs_per_group,
    int tcin = 0, int tcout = 0) {
    __shared__ double red[TMF_RED_LANES][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + tx;
    const int g = blockIdx.y;
    const int s0 = g * slabs_per_group;
    int s1 = s0 + slabs_per_group;
    if (s1 > nsplit) s1 = nsplit;
    double a = 0.0;
    if (e < n) {
        int s = s0 + ty;
        for (; s + 3 * TMF_RED_LANES < s1; s += 4 * TMF_RED_LANES) {
            const float v0 = partial[(size_t)s * n + e];
            const float v1 = partial[(size_t)(s + TMF_RED_LANES) * n + e];
            const float v2 = partial[(size_t)(s + 2 * TMF_RED_LANES) * n + e];
            const float v3 = partial[(size_t)(s + 3 * TMF_RED_LANES) * n + e];
            a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; s < s1; s += TMF_RED_LANES) a += (double)partial[(size_t)s * n + e];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && e < n) {
#pragma unroll
        for (int k = 1; k < TMF_RED_LANES; ++k) a += red[k][tx];
        size_t o = (size_t)g * n + e;
        if (tcin > 0) {
            const int co = (int)(e % tcout), ci = (int)((e / tcout) % tcin), t = (int)(e / ((long)tcout * tcin));
            const int T = (int)(n / ((long)tcin * tcout));
            o = ((size_t)co * tcin + ci) * T + t;
        }
        out[o] = (float)a;
    }
}

// The same sums, four consecutive elements per lane (16-byte loads: a quarter of the load instructions; 256 slabs of
// 221 KB went 26.6 -> ~14 us).  Per element the order of additions is exactly the scalar kernel's, so the two are bitwise
// interchangeable; used when n % 4 == 0 and the slabs are 16-byte aligned.
static __global__ __launch_bounds__(64 * TMF_RED_LANES) void tmf_slab_reduce4_kernel(
    const float* __restrict__ partial, float* __restrict__ out, int nsplit, long n, int slabs_per_group,
    int tcin = 0, int tcout = 0) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ double red[TMF_RED_LANES][64][4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long e = ((long)blockIdx.x * 64 + tx) * 4;
    const int g = blockIdx.y;
    const int s0 = g * slabs_per_group;
    int s1 = s0 + slabs_per_group;
    if (s1 > nsplit) s1 = nsplit;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (e < n) {
        int s = s0 + ty;
        for (; s + 3 * TMF_RED_LANES < s1; s += 4 * TMF_RED_LANES) {
            const f4 v0 = *reinterpret_cast<const f4*>(partial + (size_t)s * n + e);
            const f4 v1 = *reinterpret_cast<const f4*>(partial + (size_t)(s + TMF_RED_LANES) * n + e);
            const f4 v2 = *reinterpret_cast<const f4*>(partial + (size_t)(s + 2 * TMF_RED_LANES) * n + e);
            const f4 v3 = *reinterpret_cast<const f4*>(partial + (size_t)(s + 3 * TMF_RED_LANES) * n + e);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] += ((double)v0[c] + (double)v1[c]) + ((double)v2[c] + (double)v3[c]);
        }
        for (; s < s1; s += TMF_RED_LANES) {
            const f4 v = *reinterpret_cast<const f4*>(partial + (size_t)s * n + e);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] += (double)v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[ty][tx][c] = a[c];
    __syncthreads();
    if (ty == 0 && e < n) {
#pragma unroll
        for (int k = 1; k < TMF_RED_LANES; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] += red[k][tx][c];
        if (tcin > 0) {
            const int T = (int)(n / ((long)tcin * tcout));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long ec = e + c;
                const int co = (int)(ec % tcout), ci = (int)((ec / tcout) % tcin), t = (int)(ec / ((long)tcout * tcin));
                out[((size_t)co * tcin + ci) * T + t] = (float)a[c];
            }
        } else {
            *reinterpret_cast<f4*>(out + (size_t)g * n + e) = f4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
        }
    }
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

// Reduce `nsplit` slabs of `n` floats into out[n].  `scratch` (>= tmf_reduce_groups(nsplit)*n floats) is only
// touched when the plan has two stages.
// tcin / tcout > 0 (weight gradients, dw_layout == TMF_DW_REFERENCE): the final stage stores in the nn.Conv3d layout.
static inline int tmf_reduce_slabs(const float* partial, int nsplit, long n, float* scratch, float* out,
                                   hipStream_t s, const char* what, int tcin = 0, int tcout = 0) {
    const int G = tmf_reduce_groups(nsplit);
    const dim3 block(64 * TMF_RED_LANES);
    const bool v4 = n % 4 == 0 && n >= 1024 && ((size_t)partial & 15) == 0 && ((size_t)out & 15) == 0 &&
                    (G == 1 || ((size_t)scratch & 15) == 0);
    const int gx = v4 ? (int)((n / 4 + 63) / 64) : (int)((n + 63) / 64);
    auto k = v4 ? tmf_slab_reduce4_kernel : tmf_slab_reduce_kernel;
    if (G == 1) {
        hipLaunchKernelGGL(k, dim3(gx, 1), block, 0, s, partial, out, nsplit, n, nsplit, tcin, tcout);
        return tmf_launch_result(what);
    }
    const int spg = (nsplit + G - 1) / G;
    hipLaunchKernelGGL(k, dim3(gx, G), block, 0, s, partial, scratch, nsplit, n, spg, 0, 0);
    int rc = tmf_launch_result(what);
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(gx, 1), block, 0, s, (const float*)scratch, out, G, n, G, tcin, tcout);
    return tmf_launch_result(what);
}
#endif
