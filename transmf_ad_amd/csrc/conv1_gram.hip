// conv1_gram.hip — the BatchNorm statistics of the first sNet block (Conv3d(1 -> C, 3x3x3), networks.py:21-26) WITHOUT computing
// the convolution: sum z and sum z^2 per channel from pair sums of the INPUT.  gfx950.
//
// z_c(v) = sum_t w[t][c] x~(v + t) over the 27 taps t in {-1, 0, 1}^3, x~ = the volume zero-padded.  Let V+ be the volume grown by
// one voxel on every side ([-1, D] x [-1, H] x [-1, W]).  Over V+ every in-volume voxel u meets every tap exactly once
// (u - t is in V+ for every t), so
//     sum_{v in V+} z_c(v)   = (sum_t w[t][c]) * S,                        S    = sum_u x(u)
//     sum_{v in V+} z_c(v)^2 = sum_{t, t'} w[t][c] w[t'][c] R(t' - t),     R(d) = sum_u x(u) x~(u + d)
// — the 27 x 27 Gram matrix of the shifted volumes depends on the offset d = t' - t in [-2, 2]^3 only, R(d) = R(-d): 63 numbers.
// What BatchNorm wants is the sum over the volume itself = the sum over V+ minus the sum over the SHELL V+ \ V (6 % of the voxels
// at 96^3), where z is evaluated directly (27 taps x C channels on the vector ALU; only boundary layers of x are read).
//   c1_gram_kernel    R and S: one thread = 4 voxels along w, 63 double-precision accumulators (the products of two fp32 numbers
//                     are exact in fp64, so wide filters that cancel on smooth volumes lose nothing): 63 FMAs per voxel instead
//                     of 27 C = 864 multiply-adds on the matrix pipe; HBM: the input once (28 MB at B = 8, 96^3)
//   c1_shell_kernel   sum z, sum z^2 over the shell, one thread per shell voxel
//   c1_gram_finish    the quadratic forms in fp64 -> rows 0 / 1 of stat_partial = the high / low float halves of the two sums,
//                     all other rows zero: tmf_bn_finalize (fp64 sum over the rows) needs no change.
// The statistics then differ from "the sums of the fp32 z the forward pass produces" by fp32 rounding of z (1e-7 relative).
// Scratch: the tail of stat_partial itself ([tmf_c1_blocks()][2][C] floats); a volume whose partial buffer is too small for it
// (a few bricks) keeps the direct pass.  The bf16 mode keeps its own pass too: there the convolution is 2 MFMAs per tile instead
// of 14 (92 us at 128^3 against 184 us for these three kernels, whose fp64 work does not depend on the operand type).
// tmf_set_option("c1_gram", 0 | 1) / TMF_C1_GRAM (default 1).
#include "tmf_common.h"

namespace {

constexpr int GT_D = 8, GT_H = 8, GT_W = 32;                       // voxels per brick: 512 threads x 4 along w
constexpr int GP_D = GT_D + 2, GP_H = GT_H + 4, GP_W = GT_W + 4;   // tile: d 0 .. +2, h -2 .. +2, w -2 .. +2 (the half space of offsets)
constexpr int GTHR = 512, GWG = 256, SWG = 256, NACC = 64;         // 63 offsets + S

// index of an offset of the half space {d > 0} u {d = 0, h > 0} u {d = h = 0, w >= 0}
__host__ __device__ constexpr int kidx(int dd, int dh, int dw) {
    return dd == 0 ? (dh == 0 ? dw : 3 + (dh - 1) * 5 + (dw + 2)) : 13 + ((dd - 1) * 5 + (dh + 2)) * 5 + (dw + 2);
}
static_assert(kidx(2, 2, 2) == 62, "63 offsets");

typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(GTHR) void c1_gram_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                        int D, int H, int W, int tilesD, int tilesH, int tilesW, int ntiles) {
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
            pre[j] = v;
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

// Shell voxel s of a sample: the two d faces as full (H + 2) x (W + 2) planes, then the h faces over d in [0, D), then the w faces
// over d, h inside.
__device__ __forceinline__ void shell_voxel(int r, int D, int H, int W, int& vd, int& vh, int& vw) {
    const int pd = (H + 2) * (W + 2), ph = D * (W + 2), pw = D * H;
    if (r < 2 * pd) {
        const int q = r % pd;
        vd = r < pd ? -1 : D; vh = q / (W + 2) - 1; vw = q % (W + 2) - 1;
    } else if ((r -= 2 * pd) < 2 * ph) {
        const int q = r % ph;
        vh = r < ph ? -1 : H; vd = q / (W + 2); vw = q % (W + 2) - 1;
    } else {
        r -= 2 * ph;
        const int q = r % pw;
        vw = r < pw ? -1 : W; vd = q / H; vh = q % H;
    }
}

constexpr int STHR = 1024;                                         // shell: few rows of partials, many waves per workgroup
__global__ __launch_bounds__(STHR) void c1_shell_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                         int B, int D, int H, int W, int C) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 32];
    __shared__ float red[16 * (STHR + 1)];
    const int tid = threadIdx.x, n0 = blockIdx.y * 32;
    for (int e = tid; e < 27 * 32; e += STHR) {
        const int c = n0 + (e & 31);
        float v = c < C ? w[(e >> 5) * C + c] : 0.f;
        wl[e] = v;
    }
    __syncthreads();
    const int ns = 2 * (H + 2) * (W + 2) + 2 * D * (W + 2) + 2 * D * H;
    const long total = (long)B * ns;
    float s12[64];                                                  // [sum z | sum z^2] x 32 channels
#pragma unroll
    for (int c = 0; c < 64; ++c) s12[c] = 0.f;
    for (long s = (long)blockIdx.x * STHR + tid; s < total; s += (long)gridDim.x * STHR) {
        const int b = (int)(s / ns);
        int vd, vh, vw;
        shell_voxel((int)(s - (long)b * ns), D, H, W, vd, vh, vw);
        const float* xb = x + (size_t)b * D * H * W;
        float z[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) z[c] = 0.f;
#pragma unroll 1
        for (int tap = 0; tap < 27; ++tap) {                       // (a rolled loop: 27 x 32 unrolled multiply-adds spill)
            const int ud = vd + tap / 9 - 1, uh = vh + (tap / 3) % 3 - 1, uw = vw + tap % 3 - 1;
            if (ud >= 0 && ud < D && uh >= 0 && uh < H && uw >= 0 && uw < W) {      // (a shell voxel sees at most 9 taps inside)
                float xv = xb[((size_t)ud * H + uh) * W + uw];
                const f32x4* wr = reinterpret_cast<const f32x4*>(wl + tap * 32);
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const f32x4 wv = wr[c4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[4 * c4 + e] = fmaf(wv[e], xv, z[4 * c4 + e]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) { s12[c] += z[c]; s12[32 + c] = fmaf(z[c], z[c], s12[32 + c]); }
    }
    // 64 sums over the 1024 threads, 16 at a time: thread (j = tid / 64, lane) adds 16 threads' values, its wave folds the lanes
    const int lane = tid & 63, j = tid >> 6;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) red[q * (STHR + 1) + tid] = s12[ch * 16 + q];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) s += red[j * (STHR + 1) + m * 64 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const int col = ch * 16 + j, c = n0 + (col & 31);           // col = which * 32 + channel
        if (lane == 0 && c < C) part[((size_t)blockIdx.x * 2 + (col >> 5)) * C + c] = s;
    }
}

constexpr int GMAXC = 64;                                          // channels of the first block this path takes (sNet: dim / 4 = 32)

__global__ __launch_bounds__(1024) void c1_gram_finish_kernel(const float* __restrict__ w, const double* __restrict__ gram, int ngram,
                                                               const float* __restrict__ shell, int nshell, float* __restrict__ out,
                                                               int C) {
    __shared__ double ra[16 * NACC];
    __shared__ double R[NACC];
    __shared__ double sh[2 * GMAXC];
    __shared__ float wsh[27 * GMAXC];
    __shared__ double rows[27 * GMAXC];
    const int tid = threadIdx.x;
    {
        const int k = tid & 63, rg = tid >> 6;
        double s = 0.0;
#pragma unroll 8
        for (int i = rg; i < ngram; i += 16) s += gram[(size_t)i * NACC + k];
        ra[rg * NACC + k] = s;
    }
    for (int e = tid; e < 27 * C; e += 1024) {
        float v = w[e];
        wsh[e] = v;
    }
    __shared__ double shp[8 * 2 * GMAXC];
    for (int e = tid; e < 8 * 2 * C; e += 1024) {                    // column j = which * C + c of the shell partials, 8 row groups
        const int j = e % (2 * C), rg = e / (2 * C);
        double s = 0.0;
#pragma unroll 8
        for (int i = rg; i < nshell; i += 8) s += (double)shell[(size_t)i * 2 * C + j];
        shp[e] = s;
    }
    __syncthreads();
    if (tid < NACC) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += ra[i * NACC + tid];
        R[tid] = s;
    }
    for (int j = tid; j < 2 * C; j += 1024) {
        double s = 0.0;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) s += shp[rg * 2 * C + j];
        sh[j] = s;
    }
    __syncthreads();
    // rows[t][c] = w[t][c] * sum_u w[u][c] R(u - t): one (t, c) per thread
    for (int e = tid; e < 27 * C; e += 1024) {
        const int t = e / C, c = e - t * C;
        const int td = t / 9, th = (t / 3) % 3, tw = t % 3;
        double row = 0.0;
        for (int u = 0; u < 27; ++u) {
            int dd = u / 9 - td, dh = (u / 3) % 3 - th, dw = u % 3 - tw;
            if (dd < 0 || (dd == 0 && (dh < 0 || (dh == 0 && dw < 0)))) { dd = -dd; dh = -dh; dw = -dw; }
            row = fma((double)wsh[u * C + c], R[kidx(dd, dh, dw)], row);
        }
        rows[e] = (double)wsh[e] * row;
    }
    __syncthreads();
    if (tid < C) {
        double q = 0.0, sw = 0.0;
        for (int t = 0; t < 27; ++t) { q += rows[t * C + tid]; sw += (double)wsh[t * C + tid]; }
        const double s1 = sw * R[63] - sh[tid], s2 = q - sh[C + tid];
        const float h1 = (float)s1, h2 = (float)s2;
        out[0 * C + tid] = h1;                          // row 0: [sum z | sum z^2] high halves (the scratch behind row 1 is read: done above)
        out[1 * C + tid] = h2;
        out[2 * C + tid] = (float)(s1 - (double)h1);    // row 1: the low halves
        out[3 * C + tid] = (float)(s2 - (double)h2);
    }
}

int g_c1_gram = -1;
int c1_gram_mode() {
    if (g_c1_gram < 0) {
        const char* e = getenv("TMF_C1_GRAM");
        g_c1_gram = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_c1_gram;
}

}  // namespace

int tmf_c1_gram_set(int v) { g_c1_gram = v ? 1 : 0; return TMF_OK; }

// rows of stat_partial that carry the sums after tmf_c1_stats: 2 (high / low halves) where the pair-sum path runs, else all
// tmf_c1_blocks() rows — the nblk to hand to tmf_bn_finalize
static bool gram_plan(int nblk, int B, int D, int H, int W, int C, int& tilesD, int& tilesH, int& tilesW, long& ntiles, int& ngram, int& nshell,
                      size_t& off) {
    if (!c1_gram_mode() || C > GMAXC || nblk < 2) return false;
    tilesD = tmf_cdiv(D, GT_D); tilesH = tmf_cdiv(H, GT_H); tilesW = tmf_cdiv(W, GT_W);
    ntiles = (long)B * tilesD * tilesH * tilesW;
    ngram = (int)(ntiles < GWG ? ntiles : GWG);
    const long nshell_vox = (long)B * (2L * (H + 2) * (W + 2) + 2L * D * (W + 2) + 2L * D * H);
    nshell = (int)(tmf_cdiv(nshell_vox, (long)STHR) < SWG ? tmf_cdiv(nshell_vox, (long)STHR) : SWG);
    off = ((size_t)4 * C * 4 + 255) / 256 * 256;                                    // bytes: the two result rows, then the scratch
    const size_t have = (size_t)nblk * 2 * C * 4, fixed = off + (size_t)ngram * NACC * 8, row = (size_t)2 * C * 4;
    if (have < fixed + row || ntiles >= (1L << 31)) return false;
    if ((have - fixed) / row < (size_t)nshell) nshell = (int)((have - fixed) / row);        // (fewer, longer shell workgroups)
    return true;
}
extern "C" int tmf_c1_stat_rows(int B, int D, int H, int W, int C, int nblk) {
    int a, b, c, ng, nsh; long nt; size_t off;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || nblk <= 0) return 0;
    return gram_plan(nblk, B, D, H, W, C, a, b, c, nt, ng, nsh, off) ? 2 : nblk;
}

// -> 1: the statistics are enqueued (rows 0, 1 of stat_partial: tmf_c1_stat_rows() = 2); 0: not taken (option off, C > 64, or the
// partial buffer too small for the scratch) — the caller runs the direct pass; < 0: an error code
int tmf_c1_stats_gram(const float* x, const float* w, float* stat_partial, int nblk,
                      int B, int D, int H, int W, int C, void* stream) {
    int tilesD, tilesH, tilesW, ngram, nshell; long ntiles; size_t off;
    if (!gram_plan(nblk, B, D, H, W, C, tilesD, tilesH, tilesW, ntiles, ngram, nshell, off)) return 0;
    double* gram = (double*)((char*)stat_partial + off);
    float* shell = (float*)(gram + (size_t)ngram * NACC);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(c1_gram_kernel, dim3(ngram), dim3(GTHR), 0, s, x, gram, D, H, W, tilesD, tilesH, tilesW, (int)ntiles);
    hipLaunchKernelGGL(c1_shell_kernel, dim3(nshell, tmf_cdiv(C, 32)), dim3(STHR), 0, s, x, w, shell, B, D, H, W, C);
    hipLaunchKernelGGL(c1_gram_finish_kernel, dim3(1), dim3(1024), 0, s, w, (const double*)gram, ngram, (const float*)shell, nshell, stat_partial, C);
    const int rc = tmf_launch_result("tmf_c1_stats(gram)");
    return rc ? rc : 1;
}
