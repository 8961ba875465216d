// conv1_gram.hip — the first sNet block (Conv3d(1 -> C, 3x3x3), networks.py:21-26) through the tap Gram matrix of its INPUT:
// BatchNorm statistics without a convolution pass, and the backward of the block in ONE pass over the volume.  gfx950, fp32 mode.
//
// z_c(v) = sum_t w[t][c] x~(v + t) over the 27 taps t in {-1, 0, 1}^3, x~ = the volume zero-padded.  Everything quadratic in z is a
// form in the 27 x 27 matrix  G[t][t'] = sum_{v in V} x~(v + t) x~(v + t')  and the 27 sums  S_t = sum_{v in V} x~(v + t):
//     sum_v z_c = sum_t w[t][c] S_t,      sum_v z_c^2 = w_c^T G w_c,      sum_v x~(v + t) z_c(v) = (G w_c)[t].
// Let V+ be the volume grown by one voxel on every side.  Over V+ every in-volume voxel u meets every tap exactly once (u - t is
// in V+ for every t), so the sums over V+ depend on the OFFSET only:
//     sum_{v in V+} x~(v + t) x~(v + t') = R(t' - t),   R(d) = sum_u x(u) x~(u + d),  d in [-2, 2]^3,  R(d) = R(-d): 63 numbers,
// and G = R(t' - t) - Hs[t][t'], S_t = S - Es[t] with Hs / Es the same sums over the one-voxel SHELL V+ \ V (6 % of the voxels at
// 96^3), where a voxel of face class (axis, side) sees only the 9 taps that point inward: 81 + 9 sums per class.
//   c1_gram_kernel         R and S: one thread = 4 voxels along w, 63 double-precision accumulators (the products of two fp32
//                          numbers are exact in fp64, so filters that cancel on smooth volumes lose nothing): 63 FMAs per voxel
//                          instead of 27 C = 864 multiply-adds on the matrix pipe; HBM: the input once (28 MB at B = 8, 96^3)
//   c1_shell_gram_kernel   Hs, Es per face class
//   c1_gram2_finish_kernel G, S_t -> the caller's buffer (kept for backward); rows 0 / 1 of stat_partial = high / low float
//                          halves of sum z, sum z^2 (tmf_bn_finalize over 2 rows, unchanged)
//   c1_bwd_fused_finish    dw, dgamma, dbeta from the sums of conv1_fused_kernel<MODE_RD> (conv1_fused.hip) and G — see below
// tmf_set_option("c1_gram", 0 | 1) / TMF_C1_GRAM (default 1): 0 -> tmf_c1_gram_bytes() = 0, callers take the recomputing passes.
#include "tmf_common.h"

namespace {

constexpr int GT_D = 8, GT_H = 8, GT_W = 32;                       // voxels per brick: 512 threads x 4 along w
constexpr int GP_D = GT_D + 2, GP_H = GT_H + 4, GP_W = GT_W + 4;   // tile: d 0 .. +2, h -2 .. +2, w -2 .. +2 (the half space of offsets)
constexpr int GTHR = 512, GWG = 256, NACC = 64;         // 63 offsets + S

// index of an offset of the half space {d > 0} u {d = 0, h > 0} u {d = h = 0, w >= 0}
__host__ __device__ constexpr int kidx(int dd, int dh, int dw) {
    return dd == 0 ? (dh == 0 ? dw : 3 + (dh - 1) * 5 + (dw + 2)) : 13 + ((dd - 1) * 5 + (dh + 2)) * 5 + (dw + 2);
}
static_assert(kidx(2, 2, 2) == 62, "63 offsets");

typedef double f64x2 __attribute__((ext_vector_type(2)));
// the bf16 mode (conv1_fused_kernel<.., true>) multiplies the volume and the taps ROUNDED to bf16 (round-to-nearest-even): the same
// rounding here makes G, S_t and the forms in them the exact statistics of what that kernel computes
__device__ __forceinline__ float rbf16(float v) { return __builtin_bit_cast(float, tmf_pack_bf16(v, 0.f) << 16); }

__global__ __launch_bounds__(GTHR) void c1_gram_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                        int D, int H, int W, int tilesD, int tilesH, int tilesW, int ntiles, int round16) {
    // the tile as DOUBLES (converted once where it is written: one conversion per voxel instead of one per use), 34 KB;
    // the reduction scratch of the epilogue shares the allocation
    constexpr int NT = GP_D * GP_H * GP_W, NS = NT > 8 * GTHR ? NT : 8 * GTHR;
    __shared__ __attribute__((aligned(16))) double smem[NS];
    double* tile = smem;
    double* red = smem;
    const int tid = threadIdx.x;
    const int g = tid & 7, hh = (tid >> 3) & 7, dd0 = tid >> 6;
    double acc[63];
#pragma unroll
    for (int k = 0; k < 63; ++k) acc[k] = 0.0;
    double ssum = 0.0;
    // the tile elements this thread copies (the same of every brick): coordinates inside the tile, packed
    constexpr int NE = (NT + GTHR - 1) / GTHR;
    int pe[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + j * GTHR;
        pe[j] = e < NT ? (e % GP_W) | (((e / GP_W) % GP_H) << 8) | ((e / (GP_W * GP_H)) << 16) : -1;
    }
    float pre[NE];                                                  // the NEXT brick's elements, in flight while this one is summed
    auto fetch = [&](int t) {
        int r = t;
        const int bw = r % tilesW; r /= tilesW;
        const int bh = r % tilesH; r /= tilesH;
        const int bd = r % tilesD;
        const int b = r / tilesD;
        const int d0 = bd * GT_D, h0 = bh * GT_H - 2, w0 = bw * GT_W - 2;
        const float* xb = x + (size_t)b * D * H * W;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int gd = d0 + (pe[j] >> 16), gh = h0 + ((pe[j] >> 8) & 255), gw = w0 + (pe[j] & 255);
            float v = 0.f;
            if (pe[j] >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) v = xb[((size_t)gd * H + gh) * W + gw];
            pre[j] = round16 ? rbf16(v) : v;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                            // (the previous brick's reads are done)
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (pe[j] >= 0) tile[tid + j * GTHR] = (double)pre[j];
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        // this thread's voxels: (d0 + dd0, h0 + 2 + hh, w0 + 2 + 4 g + i), i = 0 .. 3; a row = the 8 values w - 2 .. w + 5
        const double* base = tile + (dd0 * GP_H + hh + 2) * GP_W + 4 * g;
        double a[4];
        {
            double rd[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f64x2 p = *reinterpret_cast<const f64x2*>(base + 2 * j); rd[2 * j] = p[0]; rd[2 * j + 1] = p[1]; }
            a[0] = rd[2]; a[1] = rd[3]; a[2] = rd[4]; a[3] = rd[5];
            ssum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
            for (int dw = 0; dw <= 2; ++dw)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[kidx(0, 0, dw)] = fma(a[i], rd[i + 2 + dw], acc[kidx(0, 0, dw)]);
        }
#pragma unroll
        for (int dd = 0; dd <= 2; ++dd)
#pragma unroll
            for (int dh = -2; dh <= 2; ++dh) {
                if (dd == 0 && dh <= 0) continue;
                const double* row = base + (dd * GP_H + dh) * GP_W;
                double rd[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const f64x2 p = *reinterpret_cast<const f64x2*>(row + 2 * j); rd[2 * j] = p[0]; rd[2 * j + 1] = p[1]; }
#pragma unroll
                for (int dw = -2; dw <= 2; ++dw)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[kidx(dd, dh, dw)] = fma(a[i], rd[i + 2 + dw], acc[kidx(dd, dh, dw)]);
            }
    }
    // workgroup sums in a fixed order, 8 accumulators at a time: thread (k = tid / 64, j = tid % 64) adds 8 threads' values,
    // then the 64 lanes of a wave fold theirs
    const int lane = tid & 63, kq = tid >> 6;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = c * 8 + q;
            red[q * GTHR + tid] = k < 63 ? acc[k < 63 ? k : 0] : ssum;
        }
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) s += red[kq * GTHR + m * 64 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) part[(size_t)blockIdx.x * NACC + c * 8 + kq] = s;
    }
}

constexpr int GMAXC = 64;                                          // channels of the first block this path takes (sNet: dim / 4 = 32)

// ------------------------------------------------------------------------------------------------------------
// The exact tap Gram matrix of the volume, for the BACKWARD of the block as well (round 5, second half):
//     G[t][t'] = sum_{v in V} x~(v + t) x~(v + t')  = R(t' - t) - Hs[t][t'],      S_t = sum_{v in V} x~(v + t) = S - Es[t],
// Hs / Es = the same sums over the one-voxel shell V+ \ V.  A shell voxel of the face class (axis a, side) sees only the 9 taps with
// t_a pointing inward: 81 pair sums and 9 sums per class.  With G and S_t the statistics are quadratic / linear forms of the
// weights (sum z_c^2 = w_c^T G w_c, sum z_c = sum_t w[t][c] S_t), and the weight gradient of the block needs NO second pass over
// the volume: with dz = s (dy - c0 - xhat c1), xhat = (z - mu) / sigma,
//     dw[t][c] = s_c [ D[t][c] - c0_c S_t - c1_c / sigma_c ( V[t][c] - mu_c S_t ) ],   V[t][c] = sum_{t'} w[t'][c] G[t][t'],
// where D[t][c] = sum_v x~(v + t) dy_c(v) is the only term that needs dy — and dy is ONE element per pooling window: 27 reads and
// multiply-adds per window in the pass that already computes the BatchNorm sums (conv1_fused_kernel<MODE_RD>).
// Layout of the caller's buffer (doubles): [0, 729) G, [729, 756) S_t, 756 S, 757 spare, then the scratch of the three kernels.
// ------------------------------------------------------------------------------------------------------------
constexpr int NGRAM = 760;                                         // doubles ahead of the scratch
constexpr int HWG = 64, HTHR = 256, HNB = 288, HPART = 96;          // shell classes: 6 x HWG workgroups, batches of 288 voxels
constexpr int HNE = (HNB * 9 + HTHR - 1) / HTHR;                   // tile elements per thread and batch          // shell classes: 6 x HWG workgroups; 90 numbers per partial row

// tap index of in-class tap j (0 .. 8) of face class cls: the axis cls / 2 is fixed at +1 (side 0: v = -1) or -1 (side 1: v = n)
__host__ __device__ constexpr int class_tap(int cls, int j) {
    const int ax = cls >> 1, fixed = (cls & 1) ? 0 : 2, a = j / 3, b = j % 3;
    return ax == 0 ? fixed * 9 + a * 3 + b : (ax == 1 ? a * 9 + fixed * 3 + b : a * 9 + b * 3 + fixed);
}

__global__ __launch_bounds__(HTHR) void c1_shell_gram_kernel(const float* __restrict__ x, double* __restrict__ part, int B, int D, int H, int W,
                                                              int round16) {
    __shared__ double xs[HNB * 9];                              // (converted once where it is written)
    __shared__ int vbase[HNB], vcrd[HNB];                      // per voxel of the batch: sample offset, packed (vd + 1, vh + 1, vw + 1) or -1
    __shared__ double red[3 * 81];
    const int tid = threadIdx.x, cls = blockIdx.y, ax = cls >> 1, side = cls & 1;
    const int n1 = ax == 0 ? H + 2 : D, n2 = ax == 2 ? H : W + 2;          // the face as an n1 x n2 grid per sample
    const int per = n1 * n2, total = B * per;                  // (< 2^31: checked by the caller)
    const int p = tid % 81, slice = tid / 81, pi = p / 9, pj = p % 9;
    // this thread's copies of a batch: elements e = tid + 256 k -> (voxel e / 9, in-class tap e % 9): the tap's offset is fixed
    int cv_[HNE], cdd[HNE], cdh[HNE], cdw[HNE];
#pragma unroll
    for (int k = 0; k < HNE; ++k) {
        const int e = tid + k * HTHR, vi = e / 9, j = e - vi * 9, tap = class_tap(cls, j);
        cv_[k] = e < HNB * 9 ? vi : -1;
        cdd[k] = tap / 9 - 2; cdh[k] = (tap / 3) % 3 - 2; cdw[k] = tap % 3 - 2;      // (relative to the +1-biased coordinates)
    }
    double acc = 0.0, eacc = 0.0;
    for (int base = blockIdx.x * HNB; base < total; base += gridDim.x * HNB) {
        __syncthreads();
        for (int vt = tid; vt < HNB; vt += HTHR) {
            const int s_ = base + vt;
            int c = -1, off = 0;
            if (s_ < total) {
                const int b = s_ / per, q = s_ - b * per, q1 = q / n2, q2 = q - q1 * n2;
                int vd, vh, vw;
                if (ax == 0) { vd = side ? D : -1; vh = q1 - 1; vw = q2 - 1; }
                else if (ax == 1) { vh = side ? H : -1; vd = q1; vw = q2 - 1; }
                else { vw = side ? W : -1; vd = q1; vh = q2; }
                c = (vd + 1) | (vh + 1) << 10 | (vw + 1) << 20;
                off = b * D * H * W;
            }
            vcrd[vt] = c; vbase[vt] = off;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HNE; ++k) {
            if (cv_[k] < 0) continue;
            const int c = vcrd[cv_[k]];
            float v = 0.f;
            if (c >= 0) {
                const int ud = (c & 1023) + cdd[k], uh = ((c >> 10) & 1023) + cdh[k], uw = (c >> 20) + cdw[k];
                if (ud >= 0 && ud < D && uh >= 0 && uh < H && uw >= 0 && uw < W) v = x[(size_t)vbase[cv_[k]] + ((size_t)ud * H + uh) * W + uw];
            }
            xs[tid + k * HTHR] = (double)(round16 ? rbf16(v) : v);
        }
        __syncthreads();
        if (slice < 3) {                                           // (the threads of column j = 0 also sum their tap: E_i)
#pragma unroll 8
            for (int vi = slice * (HNB / 3); vi < (slice + 1) * (HNB / 3); ++vi) {
                const double xa = xs[vi * 9 + pi];
                acc = fma(xa, xs[vi * 9 + pj], acc);
                eacc += pj == 0 ? xa : 0.0;
            }
        }
    }
    __shared__ double rede[3 * 9];
    __syncthreads();
    if (slice < 3) {
        red[slice * 81 + p] = acc;
        if (pj == 0) rede[slice * 9 + pi] = eacc;
    }
    __syncthreads();
    double* row = part + ((size_t)cls * gridDim.x + blockIdx.x) * HPART;
    if (tid < 81) row[tid] = (red[tid] + red[81 + tid]) + red[162 + tid];
    else if (tid < 90) row[tid] = (rede[tid - 81] + rede[9 + tid - 81]) + rede[18 + tid - 81];
}

// G, S_t, S -> gram[0 .. 756]; statistic rows 0 / 1 of stat_partial (high / low halves) from them
__global__ __launch_bounds__(1024) void c1_gram2_finish_kernel(const float* __restrict__ w, const double* __restrict__ rpart, int ngram,
                                                                const double* __restrict__ hpart, int nh, double* __restrict__ gram,
                                                                float* __restrict__ out, int C, int round16) {
    __shared__ double ra[16 * NACC];
    __shared__ double R[NACC];
    __shared__ double Hc[6 * HPART];
    __shared__ double G[27 * 27 + 27];
    __shared__ float wsh[27 * GMAXC];
    __shared__ double rows[27 * GMAXC];
    const int tid = threadIdx.x;
    {
        const int k = tid & 63, rg = tid >> 6;
        double s = 0.0;
#pragma unroll 8
        for (int i = rg; i < ngram; i += 16) s += rpart[(size_t)i * NACC + k];
        ra[rg * NACC + k] = s;
    }
    for (int e = tid; e < 27 * C; e += 1024) wsh[e] = round16 ? rbf16(w[e]) : w[e];
    for (int e = tid; e < 6 * HPART; e += 1024) {                    // class cls, number k: fixed-order sum over the class's workgroups
        const int cls = e / HPART, k = e - cls * HPART;
        double s0 = 0.0, s1_ = 0.0, s2_ = 0.0, s3 = 0.0;
        if (k < 90) {
            int i = 0;
            for (; i + 3 < nh; i += 4) {
                s0 += hpart[((size_t)cls * nh + i) * HPART + k]; s1_ += hpart[((size_t)cls * nh + i + 1) * HPART + k];
                s2_ += hpart[((size_t)cls * nh + i + 2) * HPART + k]; s3 += hpart[((size_t)cls * nh + i + 3) * HPART + k];
            }
            for (; i < nh; ++i) s0 += hpart[((size_t)cls * nh + i) * HPART + k];
        }
        Hc[e] = (s0 + s1_) + (s2_ + s3);
    }
    __syncthreads();
    if (tid < NACC) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += ra[i * NACC + tid];
        R[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < 27 * 27 + 27; e += 1024) {
        double v;
        if (e < 729) {
            const int t = e / 27, u = e - t * 27;
            int dd = u / 9 - t / 9, dh = (u / 3) % 3 - (t / 3) % 3, dw = u % 3 - t % 3;
            if (dd < 0 || (dd == 0 && (dh < 0 || (dh == 0 && dw < 0)))) { dd = -dd; dh = -dh; dw = -dw; }
            v = R[kidx(dd, dh, dw)];
            for (int cls = 0; cls < 6; ++cls)                       // the classes both taps belong to
                for (int i = 0; i < 9; ++i) {
                    if (class_tap(cls, i) != t) continue;
                    for (int j = 0; j < 9; ++j)
                        if (class_tap(cls, j) == u) v -= Hc[cls * HPART + i * 9 + j];
                }
        } else {
            const int t = e - 729;
            v = R[63];
            for (int cls = 0; cls < 6; ++cls)
                for (int i = 0; i < 9; ++i)
                    if (class_tap(cls, i) == t) v -= Hc[cls * HPART + 81 + i];
        }
        G[e] = v;
        gram[e] = v;
    }
    if (tid == 0) { gram[756] = R[63]; gram[757] = 0.0; }
    __syncthreads();
    for (int e = tid; e < 27 * C; e += 1024) {                       // rows[t][c] = w[t][c] * sum_u w[u][c] G[t][u]
        const int t = e / C, c = e - t * C;
        double row = 0.0;
        for (int u = 0; u < 27; ++u) row = fma((double)wsh[u * C + c], G[t * 27 + u], row);
        rows[e] = (double)wsh[e] * row;
    }
    __syncthreads();
    if (tid < C) {
        double q = 0.0, s1 = 0.0;
        for (int t = 0; t < 27; ++t) { q += rows[t * C + tid]; s1 = fma((double)wsh[t * C + tid], G[729 + t], s1); }
        const float h1 = (float)s1, h2 = (float)q;
        out[0 * C + tid] = h1;
        out[1 * C + tid] = h2;
        out[2 * C + tid] = (float)(s1 - (double)h1);
        out[3 * C + tid] = (float)(q - (double)h2);
    }
}

// dw, dgamma, dbeta of the first block from the sums of conv1_fused_kernel<MODE_RD>: part [nblk][2][C] (sum dy, sum dy xhat),
// dred [27][C] (D, reduced over the workgroups), G / S_t of the forward
__global__ __launch_bounds__(1024) void c1_bwd_fused_finish_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ dred,
                                                                    const float* __restrict__ w, const double* __restrict__ gram,
                                                                    const float* __restrict__ scale, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, double count,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    float* __restrict__ dw, int C, int dw_ref, int round16) {
    __shared__ double pa[16 * 2 * GMAXC];
    __shared__ double c01[2 * GMAXC];
    __shared__ float wsh[27 * GMAXC];
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * 2 * C; e += 1024) {                   // column j = which * C + c, 16 row groups
        const int j = e % (2 * C), rg = e / (2 * C);
        double s = 0.0;
#pragma unroll 8
        for (int i = rg; i < nblk; i += 16) s += (double)part[(size_t)i * 2 * C + j];
        pa[e] = s;
    }
    for (int e = tid; e < 27 * C; e += 1024) wsh[e] = round16 ? rbf16(w[e]) : w[e];
    __syncthreads();
    for (int j = tid; j < 2 * C; j += 1024) {
        double s = 0.0;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) s += pa[rg * 2 * C + j];
        c01[j] = s / count;
        if (j < C) { if (dbeta) dbeta[j] = (float)s; }
        else if (dgamma) dgamma[j - C] = (float)s;
    }
    __syncthreads();
    for (int e = tid; e < 27 * C; e += 1024) {
        const int t = e / C, c = e - t * C;
        double v = 0.0;                                             // V[t][c] = sum_u w[u][c] G[t][u]
        for (int u = 0; u < 27; ++u) v = fma((double)wsh[u * C + c], gram[t * 27 + u], v);
        const double st = gram[729 + t], mu = mean[c], is = invstd[c];
        const double r = (double)scale[c] * ((double)dred[e] - c01[c] * st - c01[C + c] * is * (v - mu * st));
        if (dw_ref) dw[(size_t)c * 27 + t] = (float)r;
        else dw[e] = (float)r;
    }
}

int g_c1_gram = -1;
int c1_gram_mode() {
    if (const int o = tmf_algo_override()) return (o & TMF_SNET_ALGO_C1_GRAM) ? ((o & TMF_SNET_ALGO_C1_GRAM_BF16) ? 2 : 1) : 0;
    if (g_c1_gram < 0) {
        const char* e = getenv("TMF_C1_GRAM");
        g_c1_gram = e ? (atoi(e) <= 0 ? 0 : atoi(e) >= 2 ? 2 : 1) : 1;
    }
    return g_c1_gram;
}

}  // namespace

int tmf_c1_gram_set(int v) { g_c1_gram = v <= 0 ? 0 : (v >= 2 ? 2 : 1); return TMF_OK; }
int tmf_c1_gram_mode(void) { return c1_gram_mode(); }

// ---- the statistics + Gram data of the forward, and the one-pass backward (fp32; conv1_fused.hip launches the MODE_RD kernel) ----
extern "C" size_t tmf_c1_gram_bytes(int B, int D, int H, int W, int C) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || !c1_gram_mode() || C > GMAXC) return 0;
    // (32-bit voxel offsets and 10-bit packed coordinates in the shell kernel: larger volumes take the recomputing passes)
    if ((long)B * D * H * W >= (1L << 31) || D >= 1022 || H >= 1022 || W >= 1022) return 0;
    if ((long)B * tmf_cdiv(D, GT_D) * tmf_cdiv(H, GT_H) * tmf_cdiv(W, GT_W) >= (1L << 31)) return 0;
    return ((size_t)NGRAM + (size_t)GWG * NACC + (size_t)6 * HWG * HPART) * 8;
}
// the bf16 mode takes this path only under "c1_gram" 2 (measured, round 6: its recomputing passes run two bf16 MFMAs per 32 voxels
// and are bound by the pooled tensors they read and write — the fp64 pair sums cost more than the statistics pass they replace)
extern "C" size_t tmf_c1_gram_bytes_bf16(int B, int D, int H, int W, int C) {
    return c1_gram_mode() == 2 ? tmf_c1_gram_bytes(B, D, H, W, C) : 0;
}

static int c1_stats_g(int round16, const float* x, const float* w, float* stat_partial, void* gram, size_t gram_bytes,
                      int B, int D, int H, int W, int C, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(stat_partial); TMF_REQUIRE_PTR(gram);
    const size_t need = tmf_c1_gram_bytes(B, D, H, W, C);       // (a direct call in the bf16 mode needs "c1_gram" >= 1 only)
    TMF_REQUIRE(need > 0, TMF_E_SHAPE, "tmf_c1_stats_g: not available for this shape / option (tmf_c1_gram_bytes() = 0)");
    TMF_REQUIRE(gram_bytes >= need, TMF_E_WORKSPACE, "tmf_c1_stats_g: gram buffer %zu B < required %zu B", gram_bytes, need);
    TMF_REQUIRE((long)B * D * H * W < (1L << 31) && D < 1022 && H < 1022 && W < 1022, TMF_E_SHAPE,
                "tmf_c1_stats_g: the batch exceeds 2^31 voxels or an axis 1021");
    const int tilesD = tmf_cdiv(D, GT_D), tilesH = tmf_cdiv(H, GT_H), tilesW = tmf_cdiv(W, GT_W);
    const long ntiles = (long)B * tilesD * tilesH * tilesW;
    TMF_REQUIRE(ntiles < (1L << 31), TMF_E_SHAPE, "tmf_c1_stats_g: too many bricks");
    const int ngram = (int)(ntiles < GWG ? ntiles : GWG);
    double* g = (double*)gram;
    double* rpart = g + NGRAM;
    double* hpart = rpart + (size_t)GWG * NACC;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(c1_gram_kernel, dim3(ngram), dim3(GTHR), 0, s, x, rpart, D, H, W, tilesD, tilesH, tilesW, (int)ntiles, round16);
    hipLaunchKernelGGL(c1_shell_gram_kernel, dim3(HWG, 6), dim3(HTHR), 0, s, x, hpart, B, D, H, W, round16);
    hipLaunchKernelGGL(c1_gram2_finish_kernel, dim3(1), dim3(1024), 0, s, w, (const double*)rpart, ngram, (const double*)hpart, HWG, g,
                       stat_partial, C, round16);
    return tmf_launch_result("tmf_c1_stats_g");
}
extern "C" int tmf_c1_stats_g(const float* x, const float* w, float* stat_partial, void* gram, size_t gram_bytes,
                              int B, int D, int H, int W, int C, void* stream) {
    return c1_stats_g(0, x, w, stat_partial, gram, gram_bytes, B, D, H, W, C, stream);
}
// the bf16 mode's form: the volume and the taps rounded to bf16 first, as conv1_fused_kernel<.., true> multiplies them
extern "C" int tmf_c1_stats_g_bf16(const float* x, const float* w, float* stat_partial, void* gram, size_t gram_bytes,
                                   int B, int D, int H, int W, int C, void* stream) {
    return c1_stats_g(1, x, w, stat_partial, gram, gram_bytes, B, D, H, W, C, stream);
}

int tmf_c1_bwd_fused_finish(const float* part, int nblk, const float* dred, const float* w, const void* gram, const float* scale,
                            const float* mean, const float* invstd, double count, float* dgamma, float* dbeta, float* dw, int C,
                            int dw_ref, int round16, void* stream) {
    hipLaunchKernelGGL(c1_bwd_fused_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nblk, dred, w, (const double*)gram,
                       scale, mean, invstd, count, dgamma, dbeta, dw, C, dw_ref, round16);
    return tmf_launch_result("tmf_c1_bwd_fused(finish)");
}
