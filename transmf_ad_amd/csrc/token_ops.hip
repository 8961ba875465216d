// token_ops.hip — LayerNorm (fwd/bwd) and the token pooling head of
// CrossTransformer_MOD_AVG, plus the library-level entry points.  gfx950.
//
// LayerNorm: one wavefront per token row (dim <= 2048), wave-level shuffles only, no LDS in
// the forward; the backward also accumulates the dgamma / dbeta column sums in registers and
// reduces them once per workgroup.
// Replaces F.layer_norm at /root/reference/models/networks.py:117,219 and
// AdaptiveAvgPool1d/AdaptiveMaxPool1d + cat at :264-269, 276-281.
#include "tmf_common.h"

static thread_local char g_err[512] = "";

void tmf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" int tmf_version(void) { return 1; }
extern "C" const char* tmf_last_error_string(void) { return g_err; }

namespace {

constexpr int LN_MAXV = 8;   // float4 groups per lane: dim <= 64*4*8 = 2048

template <int VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
    int rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane * VEC; i < dim; i += 64 * VEC) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) s += xr[i + q];
    }
    const float mu = wave_sum(s) / dim;
    float v = 0.f;
    for (int i = lane * VEC; i < dim; i += 64 * VEC) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) { const float d = xr[i + q] - mu; v += d * d; }
    }
    const float rs = 1.f / sqrtf(wave_sum(v) / dim + eps);
    float* yr = y + (size_t)row * dim;
    for (int i = lane * VEC; i < dim; i += 64 * VEC) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float v = (xr[i + q] - mu) * rs * gamma[i + q] + beta[i + q];
            if (residual != nullptr) v += residual[(size_t)row * dim + i + q];
            yr[i + q] = v;
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// partial[blk][0][c] = sum_rows dy*xhat, partial[blk][1][c] = sum_rows dy
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ dy, float* __restrict__ dx,
    float* __restrict__ partial, int rows, int dim, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][2][dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[LN_MAXV * VEC], ab[LN_MAXV * VEC];
#pragma unroll
    for (int i = 0; i < LN_MAXV * VEC; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int row = r0 + wave; row < r1; row += 4) {
        const float* xr = x + (size_t)row * dim;
        const float* gr = dy + (size_t)row * dim;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < LN_MAXV; ++u) {
            const int i = (u * 64 + lane) * VEC;
            if (i < dim) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float xh = (xr[i + q] - mu) * rs;
                    const float g = gr[i + q] * gamma[i + q];
                    s1 += g;
                    s2 += g * xh;
                    ag[u * VEC + q] += gr[i + q] * xh;
                    ab[u * VEC + q] += gr[i + q];
                }
            }
        }
        s1 = wave_sum(s1) / dim;
        s2 = wave_sum(s2) / dim;
        float* dr = dx + (size_t)row * dim;
#pragma unroll
        for (int u = 0; u < LN_MAXV; ++u) {
            const int i = (u * 64 + lane) * VEC;
            if (i < dim) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float xh = (xr[i + q] - mu) * rs;
                    dr[i + q] = rs * (gr[i + q] * gamma[i + q] - s1 - xh * s2);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < LN_MAXV; ++u) {
        const int i = (u * 64 + lane) * VEC;
        if (i < dim) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                red[(wave * 2 + 0) * dim + i + q] = ag[u * VEC + q];
                red[(wave * 2 + 1) * dim + i + q] = ab[u * VEC + q];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += 256) {
        const float a = red[e] + red[2 * dim + e] + red[4 * dim + e] + red[6 * dim + e];
        partial[(size_t)blockIdx.x * 2 * dim + e] = a;
    }
}

// cls[b] = [mean_n mri | mean_n pet | max_n mri | max_n pet].  One workgroup per (b, modality, 64-channel group): 4
// token lanes per channel walk the N tokens with stride 4 (coalesced 256-B rows), then combine in token order so that
// the FIRST maximum wins, as AdaptiveMaxPool1d (the one-thread-per-channel form took 55 us on the critical path).
__global__ __launch_bounds__(256) void token_pool_fwd_kernel(const float* __restrict__ mri, const float* __restrict__ pet,
                                                             float* __restrict__ cls, int32_t* __restrict__ argmax,
                                                             int B, int N, int dim) {
    __shared__ float ssum[4][64], smax[4][64];
    __shared__ int sarg[4][64];
    const int cgroups = (dim + 63) / 64;
    const int cg = blockIdx.x % cgroups, mod = (blockIdx.x / cgroups) % 2, b = blockIdx.x / (2 * cgroups);
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int c = cg * 64 + cl;
    float s = 0.f, mx = -INFINITY;
    int am = 0;
    if (c < dim) {
        const float* src = (mod == 0 ? mri : pet) + (size_t)b * N * dim + c;
        for (int n = tl; n < N; n += 4) {
            const float v = src[(size_t)n * dim];
            s += v;
            if (v > mx || v != v) { mx = v; am = n; }     // NaN propagates, as in ATen's adaptive max pool
        }
    }
    ssum[tl][cl] = s; smax[tl][cl] = mx; sarg[tl][cl] = am;
    __syncthreads();
    if (tl == 0 && c < dim) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            s += ssum[k][cl];
            const float v = smax[k][cl];
            const int a = sarg[k][cl];
            if (v != v) {                                               // NaN wins (ATen: the LAST NaN token)
                if (mx == mx || a > am) { mx = v; am = a; }
            } else if (mx == mx && (v > mx || (v == mx && a < am))) { mx = v; am = a; }      // ties: the earliest token
        }
        cls[(size_t)b * 4 * dim + mod * dim + c] = s / N;
        cls[(size_t)b * 4 * dim + (2 + mod) * dim + c] = mx;
        argmax[((size_t)b * 2 + mod) * dim + c] = am;
    }
}

__global__ void token_pool_bwd_kernel(const float* __restrict__ dcls, const int32_t* __restrict__ argmax,
                                      float* __restrict__ dmri, float* __restrict__ dpet, int B, int N, int dim) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * 2 * N * dim;
    if (i >= total) return;
    const int c = i % dim;
    const int n = (i / dim) % N;
    const int mod = (i / ((long)dim * N)) % 2;
    const int b = i / ((long)dim * N * 2);
    const float gavg = dcls[(size_t)b * 4 * dim + mod * dim + c];
    const float gmax = dcls[(size_t)b * 4 * dim + (2 + mod) * dim + c];
    const int am = argmax[((size_t)b * 2 + mod) * dim + c];
    float* dst = (mod == 0 ? dmri : dpet);
    dst[((size_t)b * N + n) * dim + c] = gavg / N + (n == am ? gmax : 0.f);
}

// Conv weights, reference layout (Cout, Cin, k, k, k) -> the two layouts the kernels consume, in ONE launch:
//   fwd  [t][ci][co]       = w[co][ci][t]          (tmf_conv3d_fwd / tmf_conv3d_wgrad order)
//   dgrad[T-1-t][co][ci]   = w[co][ci][t]          (the same kernel computing the data gradient)
// Writes are coalesced (one thread per output element); the strided reads hit L2 (<= 3.5 MB of weights).
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                                                float* __restrict__ dgrad, int cout, int cin, int T) {
    const long n = (long)cout * cin * T;
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (fwd != nullptr) {
        if (e < n) {
            const int co = e % cout, ci = (e / cout) % cin, t = e / ((long)cout * cin);
            fwd[e] = w[((long)co * cin + ci) * T + t];
            return;
        }
        e -= n;
    }
    if (dgrad != nullptr && e < n) {
        const int ci = e % cin, co = (e / cin) % cout, tr = e / ((long)cout * cin);
        dgrad[e] = w[((long)co * cin + ci) * T + (T - 1 - tr)];
    }
}

// The bf16 matrix-core kernels' weight layouts from the reference tensor, in ONE launch (what the host side did with
// permute / flip / contiguous / to(bfloat16) copies — several launches per layer and step):
//   fwd16  [t][co][ci]       = bf16(w[co][ci][t])     (B operand rows = output channel, K = input channel contiguous)
//   dgrad16[T-1-t][ci][co]   = bf16(w[co][ci][t])     (the same kernel computing the data gradient)
// One thread per PAIR of consecutive output elements (one 4-byte store); cin and cout are even.
__global__ __launch_bounds__(256) void pack_conv_weights_bf16_kernel(const float* __restrict__ w, unsigned int* __restrict__ fwd,
                                                                     unsigned int* __restrict__ dgrad, int cout, int cin, int T) {
    const long n2 = (long)cout * cin * T / 2;
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p < n2) {
        const long e = 2 * p;
        const int ci = e % cin, co = (e / cin) % cout, t = e / ((long)cin * cout);
        const float* src = w + ((long)co * cin + ci) * T + t;
        fwd[p] = tmf_pack_bf16(src[0], src[T]);
        return;
    }
    p -= n2;
    if (dgrad != nullptr && p < n2) {
        const long e = 2 * p;
        const int co = e % cout, ci = (e / cout) % cin, tr = e / ((long)cin * cout);
        const float* src = w + ((long)co * cin + ci) * T + (T - 1 - tr);
        dgrad[p] = tmf_pack_bf16(src[0], src[(long)cin * T]);
    }
}

// The fp32x mode's weight layouts (conv3d_fwd_split_kernel: every fp32 number as hi + mid + lo, three bf16 numbers, EXACT):
//   fwd3  [p][t][co][ci]     = part p of w[co][ci][t]     dgrad3[p][T-1-t][ci][co] = part p of w[co][ci][t]
// (what the host side did with permute / flip / three casts / two subtractions / stack: ~8 launches per layer and pass).
__device__ __forceinline__ unsigned short tmf_bf16_bits(float a) { return __builtin_bit_cast(unsigned short, (__bf16)a); }
__global__ __launch_bounds__(256) void pack_conv_weights_split3_kernel(const float* __restrict__ w, unsigned short* __restrict__ fwd,
                                                                       unsigned short* __restrict__ dgrad, int cout, int cin, int T) {
    const long n = (long)cout * cin * T;
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned short* dst = fwd;
    float a;
    if (e < n) {
        const int ci = e % cin, co = (e / cin) % cout, t = e / ((long)cin * cout);
        a = w[((long)co * cin + ci) * T + t];
    } else {
        e -= n;
        if (dgrad == nullptr || e >= n) return;
        const int co = e % cout, ci = (e / cout) % cin, tr = e / ((long)cin * cout);
        a = w[((long)co * cin + ci) * T + (T - 1 - tr)];
        dst = dgrad;
    }
    const unsigned short h = tmf_bf16_bits(a);
    const float r1 = a - __builtin_bit_cast(float, (unsigned)h << 16);           // exact
    const unsigned short m = tmf_bf16_bits(r1);
    const float r2 = r1 - __builtin_bit_cast(float, (unsigned)m << 16);          // exact, fits bf16
    dst[e] = h; dst[n + e] = m; dst[2 * n + e] = tmf_bf16_bits(r2);
}

// [B][R][C] <-> [B][C][R] through a 32 x 33 LDS tile (both sides coalesced); R = D*H*W voxels, C channels
__global__ __launch_bounds__(256) void transpose_tile_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int rows, int cols, int tiles_c) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.y * rows * cols;
    const int r0 = (blockIdx.x / tiles_c) * 32, c0 = (blockIdx.x % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[base + (size_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[base + (size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

}  // namespace

static int launch_transpose(const float* src, float* dst, int B, long rows, long cols, void* stream, const char* what) {
    TMF_REQUIRE_PTR(src); TMF_REQUIRE_PTR(dst);
    TMF_REQUIRE(B > 0 && B <= 65535 && rows > 0 && cols > 0 && rows < (1L << 31) && cols < (1L << 31), TMF_E_SHAPE,
                "%s: B=%d rows=%ld cols=%ld", what, B, rows, cols);
    const long tr = tmf_cdiv(rows, 32L), tc = tmf_cdiv(cols, 32L);
    TMF_REQUIRE(tr * tc < (1L << 31), TMF_E_SHAPE, "%s: %ld tiles exceed the grid limit", what, tr * tc);
    hipLaunchKernelGGL(transpose_tile_kernel, dim3((unsigned)(tr * tc), B), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (int)rows, (int)cols, (int)tc);
    return tmf_launch_result(what);
}

extern "C" int tmf_pack_conv_weights(const float* w, float* w_fwd, float* w_dgrad, int cout, int cin, int taps, void* stream) {
    TMF_REQUIRE_PTR(w);
    TMF_REQUIRE(w_fwd != nullptr || w_dgrad != nullptr, TMF_E_NULL, "tmf_pack_conv_weights: both outputs are NULL");
    TMF_REQUIRE(cout > 0 && cin > 0 && (taps == 1 || taps == 27), TMF_E_SHAPE,
                "tmf_pack_conv_weights: cout=%d cin=%d taps=%d", cout, cin, taps);
    const long n = (long)cout * cin * taps;
    const long total = (w_fwd != nullptr && w_dgrad != nullptr) ? 2 * n : n;
    hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)tmf_cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream,
                       w, w_fwd, w_dgrad, cout, cin, taps);
    return tmf_launch_result("tmf_pack_conv_weights");
}

extern "C" int tmf_pack_conv_weights_bf16(const float* w, void* w_fwd_bf16, void* w_dgrad_bf16, int cout, int cin, int taps,
                                          void* stream) {
    TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(w_fwd_bf16);
    TMF_REQUIRE(cout > 0 && cin > 0 && (taps == 1 || taps == 27), TMF_E_SHAPE,
                "tmf_pack_conv_weights_bf16: cout=%d cin=%d taps=%d", cout, cin, taps);
    TMF_REQUIRE(cin % 2 == 0 && (w_dgrad_bf16 == nullptr || cout % 2 == 0), TMF_E_SHAPE,
                "tmf_pack_conv_weights_bf16: channel counts must be even (cin=%d cout=%d)", cin, cout);
    const long n2 = (long)cout * cin * taps / 2;
    const long total = w_dgrad_bf16 != nullptr ? 2 * n2 : n2;
    hipLaunchKernelGGL(pack_conv_weights_bf16_kernel, dim3((unsigned)tmf_cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream,
                       w, (unsigned int*)w_fwd_bf16, (unsigned int*)w_dgrad_bf16, cout, cin, taps);
    return tmf_launch_result("tmf_pack_conv_weights_bf16");
}

extern "C" int tmf_pack_conv_weights_split3(const float* w, void* w3_fwd, void* w3_dgrad, int cout, int cin, int taps, void* stream) {
    TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(w3_fwd);
    TMF_REQUIRE(cout > 0 && cin > 0 && (taps == 1 || taps == 27), TMF_E_SHAPE,
                "tmf_pack_conv_weights_split3: cout=%d cin=%d taps=%d", cout, cin, taps);
    const long n = (long)cout * cin * taps;
    const long total = w3_dgrad != nullptr ? 2 * n : n;
    hipLaunchKernelGGL(pack_conv_weights_split3_kernel, dim3((unsigned)tmf_cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream,
                       w, (unsigned short*)w3_fwd, (unsigned short*)w3_dgrad, cout, cin, taps);
    return tmf_launch_result("tmf_pack_conv_weights_split3");
}

extern "C" int tmf_layout_ncdhw_to_ndhwc(const float* src, float* dst, int B, int C, long voxels, void* stream) {
    return launch_transpose(src, dst, B, C, voxels, stream, "tmf_layout_ncdhw_to_ndhwc");
}

extern "C" int tmf_layout_ndhwc_to_ncdhw(const float* src, float* dst, int B, int C, long voxels, void* stream) {
    return launch_transpose(src, dst, B, voxels, C, stream, "tmf_layout_ndhwc_to_ncdhw");
}

extern "C" int tmf_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                 float* mean, float* rstd, int rows, int dim, float eps, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(gamma); TMF_REQUIRE_PTR(beta); TMF_REQUIRE_PTR(y);
    TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(rstd);
    TMF_REQUIRE(rows > 0 && dim > 0, TMF_E_SHAPE, "tmf_layernorm_fwd: rows=%d dim=%d", rows, dim);
    dim3 grid(tmf_cdiv(rows, 4)), block(256);
    if (dim % 4 == 0) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, block, 0, (hipStream_t)stream, x, gamma, beta, residual, y, mean, rstd, rows, dim, eps);
    else              hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, block, 0, (hipStream_t)stream, x, gamma, beta, residual, y, mean, rstd, rows, dim, eps);
    return tmf_launch_result("tmf_layernorm_fwd");
}

static int ln_rows_per_block(int rows) {
    int rpb = tmf_cdiv(rows, 128);
    if (rpb < 4) rpb = 4;
    return rpb;
}

extern "C" int tmf_layernorm_bwd_blocks(int rows, int dim) {
    (void)dim;
    if (rows <= 0) return 0;
    return tmf_cdiv(rows, ln_rows_per_block(rows));
}

extern "C" int tmf_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* dy, float* dx, float* partial, int rows, int dim, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(gamma); TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(rstd);
    TMF_REQUIRE_PTR(dy); TMF_REQUIRE_PTR(dx); TMF_REQUIRE_PTR(partial);
    TMF_REQUIRE(rows > 0 && dim > 0, TMF_E_SHAPE, "tmf_layernorm_bwd: rows=%d dim=%d", rows, dim);
    const int vec = dim % 4 == 0 ? 4 : 1;
    TMF_REQUIRE(dim <= 64 * vec * LN_MAXV, TMF_E_SHAPE, "tmf_layernorm_bwd: dim=%d exceeds %d", dim, 64 * vec * LN_MAXV);
    const int rpb = ln_rows_per_block(rows);
    dim3 grid(tmf_cdiv(rows, rpb)), block(256);
    const size_t lds = (size_t)8 * dim * 4;
    if (vec == 4) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, lds, (hipStream_t)stream, x, gamma, mean, rstd, dy, dx, partial, rows, dim, rpb);
    else          hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, lds, (hipStream_t)stream, x, gamma, mean, rstd, dy, dx, partial, rows, dim, rpb);
    return tmf_launch_result("tmf_layernorm_bwd");
}

extern "C" int tmf_token_pool_fwd(const float* mri, const float* pet, float* cls, int32_t* argmax,
                                  int B, int N, int dim, void* stream) {
    TMF_REQUIRE_PTR(mri); TMF_REQUIRE_PTR(pet); TMF_REQUIRE_PTR(cls); TMF_REQUIRE_PTR(argmax);
    TMF_REQUIRE(B > 0 && N > 0 && dim > 0, TMF_E_SHAPE, "tmf_token_pool_fwd: B=%d N=%d dim=%d", B, N, dim);
    hipLaunchKernelGGL(token_pool_fwd_kernel, dim3(B * 2 * tmf_cdiv(dim, 64)), dim3(256), 0, (hipStream_t)stream,
                       mri, pet, cls, argmax, B, N, dim);
    return tmf_launch_result("tmf_token_pool_fwd");
}

extern "C" int tmf_token_pool_bwd(const float* dcls, const int32_t* argmax, float* dmri, float* dpet,
                                  int B, int N, int dim, void* stream) {
    TMF_REQUIRE_PTR(dcls); TMF_REQUIRE_PTR(argmax); TMF_REQUIRE_PTR(dmri); TMF_REQUIRE_PTR(dpet);
    TMF_REQUIRE(B > 0 && N > 0 && dim > 0, TMF_E_SHAPE, "tmf_token_pool_bwd: B=%d N=%d dim=%d", B, N, dim);
    const long total = (long)B * 2 * N * dim;
    hipLaunchKernelGGL(token_pool_bwd_kernel, dim3(tmf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       dcls, argmax, dmri, dpet, B, N, dim);
    return tmf_launch_result("tmf_token_pool_bwd");
}
