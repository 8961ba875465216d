// bn_act_pool.hip — BatchNorm3d (batch statistics) + LeakyReLU + 2x2x2 pool as streaming
// HBM-bound passes over channels-last activations (gfx950).
//
// Train-mode BatchNorm needs the global per-channel statistics before anything can be
// normalised, so the layer is two passes by construction: the conv kernel's
// epilogue produces sum / sum-of-squares partials, tmf_bn_finalize turns them into
// scale/shift, and ONE fused pass here does normalise + LeakyReLU + pool.  Backward mirrors
// it: one pass reduces sum(dy), sum(dy*xhat) straight from (z, dout) — the pool routing and
// the LeakyReLU mask are recomputed, never stored — and one pass writes dz.
//
// Thread mapping (all kernels): a workgroup is a [ROWS][CQ] grid, CQ = C/VEC channel groups;
// a thread keeps its channel group for the whole launch (so per-channel partials live in
// registers) and walks pooling windows / voxels with a grid stride.  Consecutive lanes touch
// consecutive 16-B channel groups of one voxel, then the next voxel: fully coalesced.
//
// Replaces F.batch_norm, F.leaky_relu, max_pool3d, avg_pool3d (+ their backward) at
// /root/reference/models/networks.py:23-25, 29-30, 32-34, 38-39, 41-43, 47-48, 50-52.
#include <type_traits>
#include "tmf_common.h"

namespace {

typedef tmf_f32x1 f32x1;
template <int VEC> struct Vec;
template <> struct Vec<4> { typedef f32x4 T; };
template <> struct Vec<1> { typedef f32x1 T; };
template <> struct Vec<8> { typedef tmf_f32x8 T; };

// activations are float or bf16 tensors (TmfIO widens / rounds); per-channel vectors are always float
template <int VEC, typename T>
__device__ __forceinline__ typename Vec<VEC>::T ldv(const T* p) { return TmfIO<T, VEC>::ld(p); }
template <int VEC, typename T>
__device__ __forceinline__ void stv(T* p, typename Vec<VEC>::T v) { TmfIO<T, VEC>::st(p, v); }
// streaming accesses of the big activation tensors (only the fp32 x 4 form has a nontemporal variant)
template <int VEC, typename T>
__device__ __forceinline__ typename Vec<VEC>::T ldv_s(const T* p) {
    if constexpr (VEC == 4 && std::is_same<T, float>::value) return TmfIO<float, 4>::ld_nt(p);
    else return TmfIO<T, VEC>::ld(p);
}
template <int VEC, typename T>
__device__ __forceinline__ void stv_s(T* p, typename Vec<VEC>::T v) {
    if constexpr (VEC == 4 && std::is_same<T, float>::value) TmfIO<float, 4>::st_nt(p, v);
    else TmfIO<T, VEC>::st(p, v);
}

struct Geo {
    int B, D, H, W, C;
    int WD, WH, WW;     // windows per axis (ceil(D/2).. for pools, D.. without)
    long nwin;          // B*WD*WH*WW
};

__host__ __device__ inline Geo make_geo(int B, int D, int H, int W, int C, int pool) {
    Geo g;
    g.B = B; g.D = D; g.H = H; g.W = W; g.C = C;
    if (pool == TMF_POOL_NONE) { g.WD = D; g.WH = H; g.WW = W; }
    else { g.WD = (D + 1) / 2; g.WH = (H + 1) / 2; g.WW = (W + 1) / 2; }
    g.nwin = (long)B * g.WD * g.WH * g.WW;
    return g;
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int VEC, int POOL, typename ZT = float, typename YT = float>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(
    const ZT* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
    YT* __restrict__ out, Geo g, int CQ, int ROWS, float slope) {
    typedef typename Vec<VEC>::T V;
    const int cq = threadIdx.x % CQ, prow = threadIdx.x / CQ;
    if (prow >= ROWS) return;
    const int c = cq * VEC;
    const V sc = ldv<VEC>(scale + c), sh = ldv<VEC>(shift + c);
    const int OD = POOL ? g.D / 2 : g.D, OH = POOL ? g.H / 2 : g.H, OW = POOL ? g.W / 2 : g.W;
    const long nout = (long)g.B * OD * OH * OW;
    for (long o = (long)blockIdx.x * ROWS + prow; o < nout; o += (long)gridDim.x * ROWS) {
        long t = o;
        const int ow = t % OW; t /= OW;
        const int oh = t % OH; t /= OH;
        const int od = t % OD;
        const int b = t / OD;
        V res;
        if (POOL == TMF_POOL_NONE) {
            const V v = ldv<VEC>(z + (((long)(b * g.D + od) * g.H + oh) * g.W + ow) * g.C + c);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float y = v[q] * sc[q] + sh[q];
                res[q] = y > 0.f ? y : y * slope;
            }
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) res[q] = POOL == TMF_POOL_MAX2 ? -INFINITY : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dd = 2 * od + (k >> 2), hh = 2 * oh + ((k >> 1) & 1), ww = 2 * ow + (k & 1);
                const V v = ldv<VEC>(z + (((long)(b * g.D + dd) * g.H + hh) * g.W + ww) * g.C + c);
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float y = v[q] * sc[q] + sh[q];
                    const float a = y > 0.f ? y : y * slope;
                    if (POOL == TMF_POOL_MAX2) res[q] = fmaxf(res[q], a);
                    else res[q] += a;
                }
            }
            if (POOL == TMF_POOL_AVG2) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) res[q] *= 0.125f;
            }
        }
        stv<VEC>(out + o * g.C + c, res);
    }
}

// ---------------------------------------------------------------------------------
// backward: shared window evaluation
// ---------------------------------------------------------------------------------
// For one pooling window (or one voxel when POOL == NONE) and one channel group, produce for
// each of the (up to 8) voxels: validity, xhat and dy = dLoss/d(BN output).
template <int VEC, int POOL, typename ZT = float, typename YT = float>
struct Window {
    typedef typename Vec<VEC>::T V;
    static constexpr int NV = POOL == TMF_POOL_NONE ? 1 : 8;
    bool valid[NV];
    long off[NV];        // element offset of voxel k (channel group included)
    V xhat[NV], dy[NV];

    __device__ __forceinline__ void eval(const ZT* __restrict__ z, const YT* __restrict__ dout,
                                         const Geo& g, long win, int c, const V& sc, const V& sh,
                                         const V& mu, const V& is, float slope) {
        long t = win;
        const int ww0 = t % g.WW; t /= g.WW;
        const int wh0 = t % g.WH; t /= g.WH;
        const int wd0 = t % g.WD;
        const int b = t / g.WD;
        if (POOL == TMF_POOL_NONE) {
            off[0] = (((long)(b * g.D + wd0) * g.H + wh0) * g.W + ww0) * g.C + c;
            valid[0] = true;
            const V v = ldv<VEC>(z + off[0]);
            const V go = ldv<VEC>(dout + off[0]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float zz = v[q];
                const float y = zz * sc[q] + sh[q];
                xhat[0][q] = (zz - mu[q]) * is[q];
                dy[0][q] = go[q] * (y > 0.f ? 1.f : slope);
            }
            return;
        }
        const int OD = g.D / 2, OH = g.H / 2, OW = g.W / 2;
        const bool pooled = wd0 < OD && wh0 < OH && ww0 < OW;   // a full 2x2x2 window (floor mode)
        V go;
#pragma unroll
        for (int q = 0; q < VEC; ++q) go[q] = 0.f;
        if (pooled) go = ldv<VEC>(dout + ((((long)b * OD + wd0) * OH + wh0) * OW + ww0) * g.C + c);
        V best, lr[NV];
        int arg[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) { best[q] = -INFINITY; arg[q] = 0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dd = 2 * wd0 + (k >> 2), hh = 2 * wh0 + ((k >> 1) & 1), ww = 2 * ww0 + (k & 1);
            valid[k] = dd < g.D && hh < g.H && ww < g.W;
            off[k] = (((long)(b * g.D + dd) * g.H + hh) * g.W + ww) * g.C + c;
            V v;
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[q] = 0.f;
            if (valid[k]) v = ldv<VEC>(z + off[k]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float zz = v[q];
                const float y = zz * sc[q] + sh[q];
                const float a = y > 0.f ? y : y * slope;
                xhat[k][q] = (zz - mu[q]) * is[q];
                lr[k][q] = y > 0.f ? 1.f : slope;
                if (POOL == TMF_POOL_MAX2 && a > best[q]) {   // strict '>' keeps the FIRST maximum (torch)
                    best[q] = a;
                    arg[q] = k;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                float d;
                if (POOL == TMF_POOL_MAX2) d = (pooled && arg[q] == k) ? go[q] : 0.f;
                else d = pooled ? go[q] * 0.125f : 0.f;
                dy[k][q] = d * lr[k][q];
            }
        }
    }
};

template <int VEC, int POOL, typename ZT = float, typename YT = float>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const ZT* __restrict__ z, const YT* __restrict__ dout, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ partial, Geo g, int CQ, int ROWS, float slope) {
    typedef typename Vec<VEC>::T V;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [ROWS][2][C]
    const int cq = threadIdx.x % CQ, prow = threadIdx.x / CQ;
    const bool live = prow < ROWS;
    const int c = cq * VEC;
    V s_dy, s_dyx;
#pragma unroll
    for (int q = 0; q < VEC; ++q) { s_dy[q] = 0.f; s_dyx[q] = 0.f; }
    if (live && POOL == TMF_POOL_MAX2) {
        // Max pool: the gradient reaches ONE voxel per window and channel — the first maximum of y = scale*z + shift
        // (LeakyReLU is increasing) — so the pass only needs that voxel's z: a streaming maximum, no per-voxel
        // arrays (the array form needs 200 registers at 8 channels per lane).  Border windows that are not pooled
        // (floor mode) carry no gradient and are skipped.
        const V sc = ldv<VEC>(scale + c), sh = ldv<VEC>(shift + c), mu = ldv<VEC>(mean + c), is = ldv<VEC>(invstd + c);
        const int OD = g.D / 2, OH = g.H / 2, OW = g.W / 2;
        for (long win = (long)blockIdx.x * ROWS + prow; win < g.nwin; win += (long)gridDim.x * ROWS) {
            long t = win;
            const int ww0 = t % g.WW; t /= g.WW;
            const int wh0 = t % g.WH; t /= g.WH;
            const int wd0 = t % g.WD;
            const int b = t / g.WD;
            if (!(wd0 < OD && wh0 < OH && ww0 < OW)) continue;
            const V go = ldv<VEC>(dout + ((((long)b * OD + wd0) * OH + wh0) * OW + ww0) * g.C + c);
            V ymax, zs;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dd = 2 * wd0 + (k >> 2), hh = 2 * wh0 + ((k >> 1) & 1), ww = 2 * ww0 + (k & 1);
                const V v = ldv<VEC>(z + (((long)(b * g.D + dd) * g.H + hh) * g.W + ww) * g.C + c);
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float y = v[q] * sc[q] + sh[q];
                    const bool up = k == 0 || y > ymax[q];          // strict '>' keeps the FIRST maximum (torch)
                    ymax[q] = up ? y : ymax[q];
                    zs[q] = up ? v[q] : zs[q];
                }
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float d = go[q] * (ymax[q] > 0.f ? 1.f : slope);
                s_dy[q] += d;
                s_dyx[q] += d * ((zs[q] - mu[q]) * is[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            red[(prow * 2 + 0) * g.C + c + q] = s_dy[q];
            red[(prow * 2 + 1) * g.C + c + q] = s_dyx[q];
        }
    } else if (live) {
        const V sc = ldv<VEC>(scale + c), sh = ldv<VEC>(shift + c), mu = ldv<VEC>(mean + c), is = ldv<VEC>(invstd + c);
        Window<VEC, POOL, ZT, YT> wn;
        for (long win = (long)blockIdx.x * ROWS + prow; win < g.nwin; win += (long)gridDim.x * ROWS) {
            wn.eval(z, dout, g, win, c, sc, sh, mu, is, slope);
#pragma unroll
            for (int k = 0; k < Window<VEC, POOL, ZT, YT>::NV; ++k) {
                if (wn.valid[k]) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        s_dy[q] += wn.dy[k][q];
                        s_dyx[q] += wn.dy[k][q] * wn.xhat[k][q];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            red[(prow * 2 + 0) * g.C + c + q] = s_dy[q];
            red[(prow * 2 + 1) * g.C + c + q] = s_dyx[q];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * g.C; e += blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < ROWS; ++r) a += red[r * 2 * g.C + e];
        partial[(size_t)blockIdx.x * 2 * g.C + e] = a;
    }
}

template <int VEC, int POOL, typename ZT = float, typename YT = float>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const ZT* __restrict__ z, const YT* __restrict__ dout, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coef, ZT* __restrict__ dz, Geo g, int CQ, int ROWS, float slope) {
    typedef typename Vec<VEC>::T V;
    const int cq = threadIdx.x % CQ, prow = threadIdx.x / CQ;
    if (prow >= ROWS) return;
    const int c = cq * VEC;
    const V sc = ldv<VEC>(scale + c), sh = ldv<VEC>(shift + c), mu = ldv<VEC>(mean + c), is = ldv<VEC>(invstd + c);
    const V k0 = ldv<VEC>(coef + c), k1 = ldv<VEC>(coef + g.C + c);
    if (POOL == TMF_POOL_MAX2 && VEC == 8) {      // (at 4 channels per lane the array form below measured faster: 88 vs 101 us)
        // dz = scale*(dy - k0 - xhat*k1) = z*K1 + K0 (+ scale*dy at the routed voxel of a pooled window)
        V K1, K0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            K1[q] = -sc[q] * k1[q] * is[q];
            K0[q] = sc[q] * (k1[q] * mu[q] * is[q] - k0[q]);
        }
        const int OD = g.D / 2, OH = g.H / 2, OW = g.W / 2;
        for (long win = (long)blockIdx.x * ROWS + prow; win < g.nwin; win += (long)gridDim.x * ROWS) {
            long t = win;
            const int ww0 = t % g.WW; t /= g.WW;
            const int wh0 = t % g.WH; t /= g.WH;
            const int wd0 = t % g.WD;
            const int b = t / g.WD;
            const bool pooled = wd0 < OD && wh0 < OH && ww0 < OW;
            V zv[8], ymax, add;
            int arg[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) { add[q] = 0.f; arg[q] = -1; ymax[q] = 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dd = 2 * wd0 + (k >> 2), hh = 2 * wh0 + ((k >> 1) & 1), ww = 2 * ww0 + (k & 1);
                const bool valid = dd < g.D && hh < g.H && ww < g.W;
#pragma unroll
                for (int q = 0; q < VEC; ++q) zv[k][q] = 0.f;
                if (valid) zv[k] = ldv<VEC>(z + (((long)(b * g.D + dd) * g.H + hh) * g.W + ww) * g.C + c);
                if (pooled) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float y = zv[k][q] * sc[q] + sh[q];
                        const bool up = k == 0 || y > ymax[q];
                        ymax[q] = up ? y : ymax[q];
                        arg[q] = up ? k : arg[q];
                    }
                }
            }
            if (pooled) {
                const V go = ldv<VEC>(dout + ((((long)b * OD + wd0) * OH + wh0) * OW + ww0) * g.C + c);
#pragma unroll
                for (int q = 0; q < VEC; ++q) add[q] = sc[q] * go[q] * (ymax[q] > 0.f ? 1.f : slope);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dd = 2 * wd0 + (k >> 2), hh = 2 * wh0 + ((k >> 1) & 1), ww = 2 * ww0 + (k & 1);
                if (dd < g.D && hh < g.H && ww < g.W) {
                    V r;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) r[q] = zv[k][q] * K1[q] + K0[q] + (arg[q] == k ? add[q] : 0.f);
                    stv<VEC>(dz + (((long)(b * g.D + dd) * g.H + hh) * g.W + ww) * g.C + c, r);
                }
            }
        }
        return;
    }
    Window<VEC, POOL, ZT, YT> wn;
    for (long win = (long)blockIdx.x * ROWS + prow; win < g.nwin; win += (long)gridDim.x * ROWS) {
        wn.eval(z, dout, g, win, c, sc, sh, mu, is, slope);
#pragma unroll
        for (int k = 0; k < Window<VEC, POOL, ZT, YT>::NV; ++k) {
            if (wn.valid[k]) {
                V r;
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    r[q] = sc[q] *
                                    (wn.dy[k][q] - k0[q] - wn.xhat[k][q] * k1[q]);
                stv_s<VEC>(dz + wn.off[k], r);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// finalize kernels (tiny)
// ---------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ conv_bias, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        s1 += (double)part[((size_t)i * 2 + 0) * C + c];
        s2 += (double)part[((size_t)i * 2 + 1) * C + c];
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double m = r1[0] / count;
        double var = r2[0] / count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * is;
        mean[c] = (float)m;
        invstd[c] = is;
        scale[c] = sc;
        shift[c] = beta[c] - (float)m * sc;
        if (rmean != nullptr) {
            const float mb = (float)m + (conv_bias ? conv_bias[c] : 0.f);
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mb;
        }
        if (rvar != nullptr) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* conv_bias,
                                      const float* rmean, const float* rvar, float eps, int C,
                                      float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rvar[c] + eps);
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rmean[c]) * sc;
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        s1 += (double)part[((size_t)i * 2 + 0) * C + c];
        s2 += (double)part[((size_t)i * 2 + 1) * C + c];
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)r1[0];
        if (dgamma) dgamma[c] = (float)r2[0];
        coef[c] = (float)(r1[0] / count);
        coef[C + c] = (float)(r2[0] / count);
    }
}

struct EwPlan { int vec, cq, rows, nblk; };
EwPlan plan_ew(long nwork, int C, bool wide = false) {
    EwPlan p;
    p.vec = (wide && C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : 1;      // wide: 8 bf16 channels = one 16-byte access per lane
    p.cq = C / p.vec;
    p.rows = 256 / p.cq;
    if (p.rows < 1) p.rows = 0;      // C/vec > 256 unsupported
    long nb = p.rows ? (nwork + p.rows - 1) / p.rows : 0;
    // 8 workgroups per CU, grid-stride over the rest.  (Round 5, five alternating processes of 30 steps per variant on one box: 4096 ->
    // 923 pairs/s, 2048 -> 939, 1024 -> 938: half the rows for the finalize kernels, the passes themselves level.)
#ifndef TMF_EW_MAX_BLOCKS
#define TMF_EW_MAX_BLOCKS 2048
#endif
    if (nb > TMF_EW_MAX_BLOCKS) nb = TMF_EW_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    p.nblk = (int)nb;
    return p;
}

int check_geo(const char* fn, int B, int D, int H, int W, int C, int pool) {
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, TMF_E_SHAPE, "%s: non-positive dimension", fn);
    TMF_REQUIRE(pool == TMF_POOL_NONE || pool == TMF_POOL_MAX2 || pool == TMF_POOL_AVG2, TMF_E_ARG,
                "%s: unknown pool mode %d", fn, pool);
    TMF_REQUIRE((C % 4 == 0 ? C / 4 : C) <= 256, TMF_E_SHAPE, "%s: C=%d too large for one workgroup row", fn, C);
    return TMF_OK;
}

}  // namespace

#define TMF_DISPATCH_VP(KERNEL, vec, pool, ...)                                              \
    do {                                                                                     \
        if (vec == 8 && pool == TMF_POOL_NONE) { KERNEL(8, TMF_POOL_NONE, __VA_ARGS__); }    \
        else if (vec == 8 && pool == TMF_POOL_MAX2) { KERNEL(8, TMF_POOL_MAX2, __VA_ARGS__); } \
        else if (vec == 8) { KERNEL(8, TMF_POOL_AVG2, __VA_ARGS__); }                        \
        else if (vec == 4 && pool == TMF_POOL_NONE) { KERNEL(4, TMF_POOL_NONE, __VA_ARGS__); }    \
        else if (vec == 4 && pool == TMF_POOL_MAX2) { KERNEL(4, TMF_POOL_MAX2, __VA_ARGS__); } \
        else if (vec == 4) { KERNEL(4, TMF_POOL_AVG2, __VA_ARGS__); }                        \
        else if (pool == TMF_POOL_NONE) { KERNEL(1, TMF_POOL_NONE, __VA_ARGS__); }           \
        else if (pool == TMF_POOL_MAX2) { KERNEL(1, TMF_POOL_MAX2, __VA_ARGS__); }           \
        else { KERNEL(1, TMF_POOL_AVG2, __VA_ARGS__); }                                      \
    } while (0)

extern "C" int tmf_bn_finalize(const float* stat_partial, int nblk, int C, double count,
                               const float* gamma, const float* beta, const float* conv_bias,
                               float* running_mean, float* running_var, float momentum, float eps,
                               float* mean, float* invstd, float* scale, float* shift, void* stream) {
    TMF_REQUIRE_PTR(stat_partial); TMF_REQUIRE_PTR(gamma); TMF_REQUIRE_PTR(beta);
    TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(invstd); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift);
    TMF_REQUIRE(nblk > 0 && C > 0 && count > 0, TMF_E_SHAPE, "tmf_bn_finalize: nblk=%d C=%d count=%g", nblk, C, count);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stat_partial, nblk, C, count,
                       gamma, beta, conv_bias, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    return tmf_launch_result("tmf_bn_finalize");
}

extern "C" int tmf_bn_eval_coeffs(const float* gamma, const float* beta, const float* conv_bias,
                                  const float* running_mean, const float* running_var, float eps, int C,
                                  float* scale, float* shift, void* stream) {
    TMF_REQUIRE_PTR(gamma); TMF_REQUIRE_PTR(beta); TMF_REQUIRE_PTR(running_mean); TMF_REQUIRE_PTR(running_var);
    TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift);
    TMF_REQUIRE(C > 0, TMF_E_SHAPE, "tmf_bn_eval_coeffs: C=%d", C);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(tmf_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, conv_bias, running_mean, running_var, eps, C, scale, shift);
    return tmf_launch_result("tmf_bn_eval_coeffs");
}

// io: bit 0 = z / dz are bf16 tensors, bit 1 = out / dout are bf16 tensors (0 = everything float)
#define TMF_DISPATCH_IO(LAUNCH, io, fn)                                                                   \
    do {                                                                                                  \
        if ((io) == 0) { LAUNCH(float, float); }                                                          \
        else if ((io) == 3) { LAUNCH(tmf_bf16_t, tmf_bf16_t); }                                           \
        else if ((io) == 1) { LAUNCH(tmf_bf16_t, float); }                                                \
        else { tmf_set_error("%s: unsupported io mode %d (0, 1 or 3)", fn, (int)(io)); return TMF_E_ARG; } \
    } while (0)

extern "C" int tmf_bn_act_pool_fwd_t(const void* z, const float* scale, const float* shift, void* out,
                                     int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream) {
    TMF_REQUIRE_PTR(z); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(out);
    int rc = check_geo("tmf_bn_act_pool_fwd", B, D, H, W, C, pool);
    if (rc) return rc;
    TMF_REQUIRE_ALIGNED(z); TMF_REQUIRE_ALIGNED(out);
    const Geo g = make_geo(B, D, H, W, C, pool);
    const long nout = pool ? (long)B * (D / 2) * (H / 2) * (W / 2) : (long)B * D * H * W;
    if (nout == 0) return TMF_OK;
    const EwPlan p = plan_ew(nout, C, io == 3);
#define K_FWD(V, P, ...) hipLaunchKernelGGL((bn_act_pool_fwd_kernel<V, P, ZT_, YT_>), dim3(p.nblk), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)
#define L_FWD(ZT, YT) { typedef ZT ZT_; typedef YT YT_; \
        TMF_DISPATCH_VP(K_FWD, p.vec, pool, (const ZT_*)z, scale, shift, (YT_*)out, g, p.cq, p.rows, slope); }
    TMF_DISPATCH_IO(L_FWD, io, "tmf_bn_act_pool_fwd_t");
#undef L_FWD
#undef K_FWD
    return tmf_launch_result("tmf_bn_act_pool_fwd");
}

extern "C" int tmf_bn_act_pool_fwd(const float* z, const float* scale, const float* shift, float* out,
                                   int B, int D, int H, int W, int C, int pool, float slope, void* stream) {
    return tmf_bn_act_pool_fwd_t(z, scale, shift, out, B, D, H, W, C, pool, slope, 0, stream);
}

extern "C" int tmf_bn_act_pool_bwd_blocks(int B, int D, int H, int W, int C, int pool) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const Geo g = make_geo(B, D, H, W, C, pool);
    return plan_ew(g.nwin, C).nblk;
}

extern "C" int tmf_bn_act_pool_bwd_reduce_t(const void* z, const void* dout, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, float* partial,
                                            int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream) {
    TMF_REQUIRE_PTR(z); TMF_REQUIRE_PTR(dout); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift);
    TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(invstd); TMF_REQUIRE_PTR(partial);
    int rc = check_geo("tmf_bn_act_pool_bwd_reduce", B, D, H, W, C, pool);
    if (rc) return rc;
    TMF_REQUIRE_ALIGNED(z); TMF_REQUIRE_ALIGNED(dout);
    const Geo g = make_geo(B, D, H, W, C, pool);
    EwPlan p = plan_ew(g.nwin, C, io == 3 && pool != TMF_POOL_AVG2);
    p.nblk = plan_ew(g.nwin, C).nblk;            // the slab count callers size `partial` with (tmf_bn_act_pool_bwd_blocks)
    const size_t lds = (size_t)p.rows * 2 * C * 4;
#define K_RED(V, P, ...) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, P, ZT_, YT_>), dim3(p.nblk), dim3(256), lds, (hipStream_t)stream, __VA_ARGS__)
#define L_RED(ZT, YT) { typedef ZT ZT_; typedef YT YT_; \
        TMF_DISPATCH_VP(K_RED, p.vec, pool, (const ZT_*)z, (const YT_*)dout, scale, shift, mean, invstd, partial, g, p.cq, p.rows, slope); }
    TMF_DISPATCH_IO(L_RED, io, "tmf_bn_act_pool_bwd_reduce_t");
#undef L_RED
#undef K_RED
    return tmf_launch_result("tmf_bn_act_pool_bwd_reduce");
}

extern "C" int tmf_bn_act_pool_bwd_reduce(const float* z, const float* dout, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, float* partial,
                                          int B, int D, int H, int W, int C, int pool, float slope, void* stream) {
    return tmf_bn_act_pool_bwd_reduce_t(z, dout, scale, shift, mean, invstd, partial, B, D, H, W, C, pool, slope, 0, stream);
}

extern "C" int tmf_bn_bwd_finalize(const float* partial, int nblk, int C, double count,
                                   float* dgamma, float* dbeta, float* coef, void* stream) {
    TMF_REQUIRE_PTR(partial); TMF_REQUIRE_PTR(coef);
    TMF_REQUIRE(nblk > 0 && C > 0 && count > 0, TMF_E_SHAPE, "tmf_bn_bwd_finalize: nblk=%d C=%d count=%g", nblk, C, count);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, count,
                       dgamma, dbeta, coef);
    return tmf_launch_result("tmf_bn_bwd_finalize");
}

extern "C" int tmf_bn_act_pool_bwd_apply_t(const void* z, const void* dout, const float* scale, const float* shift,
                                           const float* mean, const float* invstd, const float* coef, void* dz,
                                           int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream) {
    TMF_REQUIRE_PTR(z); TMF_REQUIRE_PTR(dout); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift);
    TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(invstd); TMF_REQUIRE_PTR(coef); TMF_REQUIRE_PTR(dz);
    int rc = check_geo("tmf_bn_act_pool_bwd_apply", B, D, H, W, C, pool);
    if (rc) return rc;
    TMF_REQUIRE_ALIGNED(z); TMF_REQUIRE_ALIGNED(dout); TMF_REQUIRE_ALIGNED(dz);
    const Geo g = make_geo(B, D, H, W, C, pool);
    const EwPlan p = plan_ew(g.nwin, C, io == 3 && pool != TMF_POOL_AVG2);
#define K_APP(V, P, ...) hipLaunchKernelGGL((bn_bwd_apply_kernel<V, P, ZT_, YT_>), dim3(p.nblk), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)
#define L_APP(ZT, YT) { typedef ZT ZT_; typedef YT YT_; \
        TMF_DISPATCH_VP(K_APP, p.vec, pool, (const ZT_*)z, (const YT_*)dout, scale, shift, mean, invstd, coef, (ZT_*)dz, g, p.cq, p.rows, slope); }
    TMF_DISPATCH_IO(L_APP, io, "tmf_bn_act_pool_bwd_apply_t");
#undef L_APP
#undef K_APP
    return tmf_launch_result("tmf_bn_act_pool_bwd_apply");
}

extern "C" int tmf_bn_act_pool_bwd_apply(const float* z, const float* dout, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const float* coef, float* dz,
                                         int B, int D, int H, int W, int C, int pool, float slope, void* stream) {
    return tmf_bn_act_pool_bwd_apply_t(z, dout, scale, shift, mean, invstd, coef, dz, B, D, H, W, C, pool, slope, 0, stream);
}

extern "C" int tmf_colsum_finalize(const float* partial, int nblk, int ncol, float* out, void* stream) {
    TMF_REQUIRE_PTR(partial); TMF_REQUIRE_PTR(out);
    TMF_REQUIRE(nblk > 0 && ncol > 0, TMF_E_SHAPE, "tmf_colsum_finalize: nblk=%d ncol=%d", nblk, ncol);
    const dim3 block(64 * TMF_RED_LANES);
    hipLaunchKernelGGL(tmf_slab_reduce_kernel, dim3(tmf_cdiv(ncol, 64), 1), block, 0, (hipStream_t)stream,
                       partial, out, nblk, (long)ncol, nblk, 0, 0);
    return tmf_launch_result("tmf_colsum_finalize");
}
