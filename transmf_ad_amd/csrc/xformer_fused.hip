// xformer_fused.hip — one Transformer(dim, depth=1) instance of the fusion block per launch.  gfx950, exact fp32.
//
// reference: models/networks.py:215-230 (Transformer.forward: x = attn(x, context) + x; x = ff(x) + x; norm(x)), built
// from PreNorm (:114-121; the context passes through UN-normalised), Attention (:141-175), FeedForward (:125-137) and the
// caller's "+ tokens" (:274-275).  The op-per-launch path (token_gemm.hip / attention.hip / token_ops.hip) needs 7
// launches forward and 13 backward per instance, each 5-20 us for ~1 us of matrix work on 1 728 token rows: the chain of a
// row tile is short, its launches are not.  Here a workgroup owns 16 token rows of ONE batch element and walks the whole
// chain with the tile in LDS:
//
//   xf_fwd_kernel     LN1 -> to_q -> attention over the context's K / V (one head per wavefront, scores of all keys in
//                     registers, exact softmax) -> to_out + bias (+ mask) + x -> LN2 -> Linear + bias -> GELU (+ mask)
//                     -> Linear + bias (+ mask) + x1 -> final LayerNorm + x -> to_kv of the NEXT instance (whose context
//                     this output is), stored row-major and transposed.
//   xf_bwd_q_kernel   the same chain backwards for a tile of QUERY rows: LNf', FF', LN2', to_out', the dQ half of the
//                     attention backward, to_q', LN1' + the two residual gradients; emits the per-tile column sums of every
//                     bias / LayerNorm parameter gradient and the dy operands of the weight gradients.
//   xf_bwd_kv_kernel  for a tile of KEY rows of the context: dK, dV over all queries of the batch element, then
//                     to_kv' + what the context tensor has collected so far.
//
// Batch elements never mix before the weight gradients, and a launch boundary costs less than a hand-rolled grid barrier
// (MI355X_MICROARCH.md: 1.5 us vs 4+ us), so the only cross-tile dependency — attention needs K / V (dK / dV: Q, dO) of the
// whole batch element — is a launch boundary.  The weight gradients dy^T x of ALL instances run as one table-driven launch
// at the end (tmf_tok_wgrad_multi).
//
// GEMMs: v_mfma_f32_16x16x4_f32, A = the 16-row tile from LDS (one ds_read_b128 per 4 MFMAs through the K-permutation
// k = 16 s + 4 (lane >> 4) + j), B straight from L2: one 16-byte load per 4 MFMAs in the y = x W^T form (W rows = output
// features), G-float loads in the dx = dy W form.  A lane's accumulator tiles are INTERLEAVED columns (tile u of a lane
// with m = lane & 15 is column G m + u of its group), so a lane owns G consecutive columns of a row: wide row-major
// stores, and the dx form reads G consecutive weights at once.
//
// Attention operands in both orders: K row-major [token][feature] and V transposed [feature][token] for the forward
// (S^T = K Q^T, O^T = V^T P^T: lane = one query, softmax reductions are two shuffles), K^T and V row-major for dQ, Q^T
// and dO^T for dK / dV — every producer writes the transposed copy from its accumulator layout for free (a lane holds 4
// consecutive tokens of one feature = one 16-byte store).  Transposed buffers are [B][features][Npad], Npad = 16 ceil(N /
// 16), zero in the padding.
//
// Dropout (options/option.py:39 `--dropout`; networks.py:131,133,153): the three keep-masks of an instance (after to_out,
// after GELU, after the second Linear), already scaled by 1 / (1 - p), are INPUTS (NULL = inactive) — the random draw
// stays with the caller, as in heads.hip.
#include "tmf_common.h"

namespace {

constexpr int XT = 16;             // token rows per workgroup
constexpr int XD = 128;            // model dim = heads * dim_head
constexpr int XH = 4;              // heads = wavefronts per workgroup
constexpr int XDH = 32;            // dim_head
constexpr int XP = XD + 4;         // LDS row pitch (floats) of a 128-wide tile
constexpr int XTHR = 256;
constexpr float XLOG2E = 1.4426950408889634f;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// phase time stamps of every wave (shader clock) for tools/xf_trace.py: only in builds with -DTMF_XF_TRACE
// (TMF_EXTRA_FLAGS=-DTMF_XF_TRACE python -m transmf_ad_amd.build); the product kernels carry none of it
#ifdef TMF_XF_TRACE
#define XF_STAMP(k)                                                                                                   \
    do {                                                                                                              \
        if (p.trace != nullptr && lane == 0)                                                                          \
            p.trace[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define XF_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float xgelu(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float xgelu_grad(float h) {
    return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}
__device__ __forceinline__ float xhalf_sum(float v) {        // sum over the 32 lanes of a half-wave
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

// Columns of a lane's accumulator tiles: groups of G = min(TPW, 4) interleaved tiles, 16 G columns per group.
template <int TPW> struct TileMap {
    static constexpr int G = TPW < 4 ? TPW : 4;
    static constexpr int NG = TPW / G;
    static __device__ __forceinline__ int col(int cbase, int u, int m) { return cbase + (u / G) * (16 * G) + G * m + (u % G); }
    static __device__ __forceinline__ int col0(int cbase, int gq, int m) { return cbase + gq * (16 * G) + G * m; }
};

// acc[u] += A[16 x 16 KS] . B for the lane's TPW column tiles.
//   NN == false:  B[k][c] = W[c * ldw + k]   (y = x W^T;  W [N][K] row-major)
//   NN == true:   B[k][c] = W[k * ldw + c]   (dx = dy W;  W [K][N] row-major)
// A tile in LDS, row pitch KP floats.  B operands of PF steps are in flight ahead of the MFMAs (register ring).
template <int TPW, int KS, bool NN>
__device__ __forceinline__ void gemm_tile(f32x4 (&acc)[TPW], const float* As, int KP, const float* __restrict__ W,
                                          int ldw, int cbase, int m, int kb) {
    constexpr int G = TileMap<TPW>::G, NG = TileMap<TPW>::NG;
    constexpr int PF = (TPW <= 2) ? 3 : 2;
    float b[PF + 1][TPW][4];
    auto load = [&](int s, float (&dst)[TPW][4]) {
        if (!NN) {
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(W + (size_t)TileMap<TPW>::col(cbase, u, m) * ldw + 16 * s + 4 * kb);
                dst[u][0] = v[0]; dst[u][1] = v[1]; dst[u][2] = v[2]; dst[u][3] = v[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const float* src = W + (size_t)(16 * s + 4 * kb + j) * ldw + TileMap<TPW>::col0(cbase, gq, m);
                    if (G == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                        dst[gq * 4 + 0][j] = v[0]; dst[gq * 4 + 1][j] = v[1]; dst[gq * 4 + 2][j] = v[2]; dst[gq * 4 + 3][j] = v[3];
                    } else {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(src);
                        dst[gq * 2 + 0][j] = v[0]; dst[gq * 2 + 1][j] = v[1];
                    }
                }
        }
    };
#pragma unroll
    for (int s = 0; s < PF && s < KS; ++s) load(s, b[s]);
    const float* arow = As + m * KP + 4 * kb;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + PF < KS) load(s + PF, b[(s + PF) % (PF + 1)]);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < TPW; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b[s % (PF + 1)][u][j], acc[u], 0, 0, 0);
    }
}

template <int TPW>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TPW]) {
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// column sums over the 16 rows of an LDS tile -> dst[0 .. ncols)
__device__ __forceinline__ void tile_colsum(const float* T, int pitch, int ncols, float* __restrict__ dst, int tid) {
    for (int c = tid; c < ncols; c += XTHR) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < XT; ++r) s += T[r * pitch + c];
        dst[c] = s;
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct XfFwdArgs {
    const float* x;                 // [B*N][128] input tokens of the instance (only_kv: the context tokens)
    const float* kv;                // [B*N][256] K | V of the context, row-major
    const float* kvT;               // [B][256][Npad] the same, transposed
    const float *ln1_g, *ln1_b, *wq, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2, *lnf_g, *lnf_b;
    const float* wkv_next;          // to_kv weight of the instance that takes this output as context (NULL: none)
    const float *mask_o, *mask_g, *mask_f;       // scaled Dropout keep-masks [R][128], [R][mlp], [R][128] or NULL
    float eps1, eps2, epsf, scale;
    float *a, *q, *qT, *out, *lse, *x1, *f, *h, *g, *x2, *y, *m1, *r1, *m2, *r2, *mf, *rf;     // saved for backward
    float *kv_next, *kvT_next;
    int B, N, Npad, tiles, only_kv;
    unsigned long long* trace;
};

constexpr int XMLP = 512;
constexpr int XGP = XMLP + 4;

// LayerNorm of a 128-wide row held as 4 values per lane of a half-wave (lane li owns columns 4 li .. 4 li + 3)
__device__ __forceinline__ f32x4 ln_row(const f32x4 v, const float* __restrict__ g, const float* __restrict__ b, float eps,
                                        int li, float& mu, float& rs) {
    mu = xhalf_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 128.f);
    const f32x4 d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
    const float var = xhalf_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 128.f);
    rs = 1.f / sqrtf(var + eps);
    const f32x4 gg = ld4(g + li * 4), bb = ld4(b + li * 4);
    return f32x4{d[0] * rs * gg[0] + bb[0], d[1] * rs * gg[1] + bb[1], d[2] * rs * gg[2] + bb[2], d[3] * rs * gg[3] + bb[3]};
}

template <int MT>
__global__ __launch_bounds__(XTHR, 1) void xf_fwd_kernel(const XfFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // input rows
    float* As = Xs + XT * XP;               // LN1(x), later LN2(x1)
    float* Qs = As + XT * XP;               // q
    float* Os = Qs + XT * XP;               // attention output
    float* X1s = Os + XT * XP;
    float* Ys = X1s + XT * XP;              // x2, later y
    float* Gs = Ys + XT * XP;               // [16][XGP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4, li = lane & 31;
    const int tile = blockIdx.x, bz = blockIdx.y;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;                 // valid rows of this tile
    const size_t row0 = (size_t)bz * N + t0;

    XF_STAMP(0);
    // ---- P0: x tile, LayerNorm 1 ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = ld4(p.x + gr * XD + li * 4);
        if (p.only_kv) {
            st4(Ys + row * XP + li * 4, v);
        } else {
            st4(Xs + row * XP + li * 4, v);
            float mu, rs;
            f32x4 a = ln_row(v, p.ln1_g, p.ln1_b, p.eps1, li, mu, rs);
            if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(As + row * XP + li * 4, a);
            if (ok) {
                st4(p.a + gr * XD + li * 4, a);
                if (li == 0) { p.m1[gr] = mu; p.r1[gr] = rs; }
            }
        }
    }
    __syncthreads();

    XF_STAMP(1);
    if (!p.only_kv) {
        // ---- P1: q = LN1(x) Wq^T ----
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            gemm_tile<2, 8, false>(acc, As, XP, p.wq, XD, 32 * wave, m, kb);
            const int c0 = 32 * wave + 2 * m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const f32x2 v = {acc[0][r], acc[1][r]};
                *reinterpret_cast<f32x2*>(Qs + row * XP + c0) = v;
                if (row < nv) *reinterpret_cast<f32x2*>(p.q + (row0 + row) * XD + c0) = v;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)           // rows past nv are exact zeros (their A rows are)
                st4(p.qT + ((size_t)bz * XD + c0 + u) * Npad + t0 + 4 * kb, acc[u]);
        }
        __syncthreads();

        XF_STAMP(2);
        // ---- P2: attention, head = wave.  S^T[key][query] = K Q^T: a lane holds 4 keys per 16-key tile of ONE query ----
        {
            const int h = wave;
            const float c = p.scale * XLOG2E;
            float qreg[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 v = ld4(Qs + m * XP + XDH * h + 16 * s + 4 * kb);
                qreg[s][0] = v[0] * c; qreg[s][1] = v[1] * c; qreg[s][2] = v[2] * c; qreg[s][3] = v[3] * c;
            }
            const int ntiles = Npad >> 4;
            const float* Kb = p.kv + (size_t)bz * N * (2 * XD) + XDH * h + 4 * kb;
            f32x4 sT[MT];
            float mx = -INFINITY;
            f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0;
            {
                const int key = m < N ? m : N - 1;
                k0 = ld4(Kb + (size_t)key * (2 * XD));
                k1 = ld4(Kb + (size_t)key * (2 * XD) + 16);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t < ntiles) {
                    f32x4 n0 = k0, n1 = k1;
                    if (t + 1 < ntiles) {
                        int key = 16 * (t + 1) + m;
                        key = key < N ? key : N - 1;
                        n0 = ld4(Kb + (size_t)key * (2 * XD));
                        n1 = ld4(Kb + (size_t)key * (2 * XD) + 16);
                    }
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[j], qreg[0][j], s, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[j], qreg[1][j], s, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * t + 4 * kb + r;
                        s[r] = key < N ? s[r] : -INFINITY;
                        mx = fmaxf(mx, s[r]);
                    }
                    sT[t] = s;
                    k0 = n0; k1 = n1;
                } else {
                    sT[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
            XF_STAMP(3);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t < ntiles) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = exp2f(sT[t][r] - mx);
                        sT[t][r] = e;
                        l += e;
                    }
                }
            }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            XF_STAMP(4);
            // O^T[d][query] = V^T P^T:  A = V^T (transposed copy: 4 consecutive keys of feature d per lane), B = P (own registers)
            const float* Vt = p.kvT + ((size_t)bz * (2 * XD) + XD + XDH * h + m) * Npad + 4 * kb;
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            f32x4 v0 = ld4(Vt), v1 = ld4(Vt + (size_t)16 * Npad);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t < ntiles) {
                    f32x4 n0 = v0, n1 = v1;
                    if (t + 1 < ntiles) {
                        n0 = ld4(Vt + 16 * (t + 1));
                        n1 = ld4(Vt + (size_t)16 * Npad + 16 * (t + 1));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[r], sT[t][r], o[0], 0, 0, 0);
                        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[r], sT[t][r], o[1], 0, 0, 0);
                    }
                    v0 = n0; v1 = n1;
                }
            }
            // lane (m = query, kb): o[dt][r] = O[query m][feature 32 h + 16 dt + 4 kb + r]
            const float inv = 1.f / l;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4 w4 = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                const int cc = XDH * h + 16 * dt + 4 * kb;
                st4(Os + m * XP + cc, w4);
                if (m < nv) st4(p.out + (row0 + m) * XD + cc, w4);
            }
            if (kb == 0) p.lse[((size_t)bz * XH + h) * Npad + t0 + m] = m < nv ? mx + log2f(l) : INFINITY;
        }
        __syncthreads();

        XF_STAMP(5);
        // ---- P4: x1 = mask_o (out Wo^T + bo) + x ----
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            gemm_tile<2, 8, false>(acc, Os, XP, p.wo, XD, 32 * wave, m, kb);
            const int c0 = 32 * wave + 2 * m;
            const f32x2 bo = *reinterpret_cast<const f32x2*>(p.bo + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                const f32x2 xr = *reinterpret_cast<const f32x2*>(Xs + row * XP + c0);
                f32x2 v = {acc[0][r] + bo[0], acc[1][r] + bo[1]};
                if (p.mask_o != nullptr && ok) {
                    const f32x2 mk = *reinterpret_cast<const f32x2*>(p.mask_o + (row0 + row) * XD + c0);
                    v[0] *= mk[0]; v[1] *= mk[1];
                }
                v[0] += xr[0]; v[1] += xr[1];
                *reinterpret_cast<f32x2*>(X1s + row * XP + c0) = v;
                if (ok) *reinterpret_cast<f32x2*>(p.x1 + (row0 + row) * XD + c0) = v;
            }
        }
        __syncthreads();

        XF_STAMP(6);
        // ---- LayerNorm 2 ----
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 8 + wave * 2 + (lane >> 5);
            const bool ok = row < nv;
            const size_t gr = row0 + row;
            const f32x4 v = ld4(X1s + row * XP + li * 4);
            float mu, rs;
            f32x4 f = ln_row(v, p.ln2_g, p.ln2_b, p.eps2, li, mu, rs);
            if (!ok) f = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(As + row * XP + li * 4, f);
            if (ok) {
                st4(p.f + gr * XD + li * 4, f);
                if (li == 0) { p.m2[gr] = mu; p.r2[gr] = rs; }
            }
        }
        __syncthreads();

        XF_STAMP(7);
        // ---- P5: h = LN2(x1) W1^T + b1;  g = mask_g GELU(h) ----
        {
            f32x4 acc[8];
            zero_acc<8>(acc);
            gemm_tile<8, 8, false>(acc, As, XP, p.w1, XD, 128 * wave, m, kb);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int c0 = 128 * wave + 64 * gq + 4 * m;
                const f32x4 b1 = ld4(p.b1 + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kb + r;
                    const bool ok = row < nv;
                    const f32x4 hv = {acc[gq * 4 + 0][r] + b1[0], acc[gq * 4 + 1][r] + b1[1], acc[gq * 4 + 2][r] + b1[2],
                                      acc[gq * 4 + 3][r] + b1[3]};
                    f32x4 gv = {xgelu(hv[0]), xgelu(hv[1]), xgelu(hv[2]), xgelu(hv[3])};
                    if (p.mask_g != nullptr && ok) {
                        const f32x4 mk = ld4(p.mask_g + (row0 + row) * XMLP + c0);
                        gv[0] *= mk[0]; gv[1] *= mk[1]; gv[2] *= mk[2]; gv[3] *= mk[3];
                    }
                    if (!ok) gv = f32x4{0.f, 0.f, 0.f, 0.f};
                    st4(Gs + row * XGP + c0, gv);
                    if (ok) {
                        st4(p.h + (row0 + row) * XMLP + c0, hv);
                        st4(p.g + (row0 + row) * XMLP + c0, gv);
                    }
                }
            }
        }
        __syncthreads();

        XF_STAMP(8);
        // ---- P6: x2 = mask_f (g W2^T + b2) + x1 ----
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            gemm_tile<2, 32, false>(acc, Gs, XGP, p.w2, XMLP, 32 * wave, m, kb);
            const int c0 = 32 * wave + 2 * m;
            const f32x2 b2 = *reinterpret_cast<const f32x2*>(p.b2 + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                const f32x2 xr = *reinterpret_cast<const f32x2*>(X1s + row * XP + c0);
                f32x2 v = {acc[0][r] + b2[0], acc[1][r] + b2[1]};
                if (p.mask_f != nullptr && ok) {
                    const f32x2 mk = *reinterpret_cast<const f32x2*>(p.mask_f + (row0 + row) * XD + c0);
                    v[0] *= mk[0]; v[1] *= mk[1];
                }
                v[0] += xr[0]; v[1] += xr[1];
                *reinterpret_cast<f32x2*>(Ys + row * XP + c0) = v;
                if (ok) *reinterpret_cast<f32x2*>(p.x2 + (row0 + row) * XD + c0) = v;
            }
        }
        __syncthreads();

        XF_STAMP(9);
        // ---- final LayerNorm + the caller's "+ tokens" ----
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 8 + wave * 2 + (lane >> 5);
            const bool ok = row < nv;
            const size_t gr = row0 + row;
            const f32x4 v = ld4(Ys + row * XP + li * 4);
            const f32x4 xr = ld4(Xs + row * XP + li * 4);
            float mu, rs;
            f32x4 y = ln_row(v, p.lnf_g, p.lnf_b, p.epsf, li, mu, rs);
            y[0] += xr[0]; y[1] += xr[1]; y[2] += xr[2]; y[3] += xr[3];
            if (!ok) y = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(Ys + row * XP + li * 4, y);             // the same lanes read and write this row
            if (ok) {
                st4(p.y + gr * XD + li * 4, y);
                if (li == 0) { p.mf[gr] = mu; p.rf[gr] = rs; }
            }
        }
        __syncthreads();
    }

    XF_STAMP(10);
    // ---- P7: K | V of the next instance (its context = this output): row-major and transposed ----
    if (p.wkv_next != nullptr) {
        f32x4 acc[4];
        zero_acc<4>(acc);
        gemm_tile<4, 8, false>(acc, Ys, XP, p.wkv_next, XD, 64 * wave, m, kb);
        const int c0 = 64 * wave + 4 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r;
            if (row < nv) st4(p.kv_next + (row0 + row) * (2 * XD) + c0, f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]});
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) st4(p.kvT_next + ((size_t)bz * (2 * XD) + c0 + u) * Npad + t0 + 4 * kb, acc[u]);
    }
    XF_STAMP(11);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, query side
// ---------------------------------------------------------------------------------------------------------------------
struct XfBwdQArgs {
    const float* dy;                // [R][128] gradient w.r.t. the instance's output
    const float* x;                 // the instance's input tokens
    const float *kv, *kvT;          // K | V of the context (row-major, transposed)
    const float *ln1_g, *wq, *wo, *ln2_g, *w1, *w2, *lnf_g;
    const float *mask_o, *mask_g, *mask_f;
    float scale;
    const float *q, *out, *lse, *x1, *h, *x2, *m1, *r1, *m2, *r2, *mf, *rf;      // saved by the forward
    float *dx2, *dh, *dx1, *dq;     // dy operands of the weight gradients ([R][128], [R][mlp], [R][128], [R][128])
    float *dout, *doutT, *delta;    // for the key-side kernel: dO row-major, transposed; delta [B][heads][Npad]
    float* dx;                      // [R][128] gradient w.r.t. the instance's input tokens
    float* part;                    // [B*tiles][stride]: b2 | b1 | bo | ln2 g | ln2 b | ln1 g | ln1 b | lnf g | lnf b
    int stride;
    int B, N, Npad, tiles;
    unsigned long long* trace;
};

__global__ __launch_bounds__(XTHR, 1) void xf_bwd_q_kernel(const XfBwdQArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DY = smem;                       // dy
    float* T1 = DY + XT * XP;               // dy * xhat products for the column sums
    float* DX2 = T1 + XT * XP;              // gradient w.r.t. x2 (residual path)
    float* DX2M = DX2 + XT * XP;            // ... times mask_f (the Linear's output gradient)
    float* DF = DX2M + XT * XP;             // gradient w.r.t. LN2's output
    float* DX1 = DF + XT * XP;              // gradient w.r.t. x1
    float* DX1M = DX1 + XT * XP;            // ... times mask_o
    float* DO = DX1M + XT * XP;             // gradient w.r.t. the attention output
    float* DQ = DO + XT * XP;
    float* DA = DQ + XT * XP;               // gradient w.r.t. LN1's output
    float* DH = DA + XT * XP;               // [16][XGP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4, li = lane & 31;
    const int tile = blockIdx.x, bz = blockIdx.y;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;
    const size_t row0 = (size_t)bz * N + t0;
    float* part = p.part + ((size_t)bz * p.tiles + tile) * p.stride;
    const int o_b2 = 0, o_b1 = XD, o_bo = XD + XMLP, o_ln2 = 2 * XD + XMLP, o_ln1 = 4 * XD + XMLP, o_lnf = 6 * XD + XMLP;

    XF_STAMP(0);
    // ---- S1: final LayerNorm backward (row-wise) ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        f32x4 dyv = {0.f, 0.f, 0.f, 0.f}, xv = dyv;
        float mu = 0.f, rs = 0.f;
        if (ok) { dyv = ld4(p.dy + gr * XD + li * 4); xv = ld4(p.x2 + gr * XD + li * 4); mu = p.mf[gr]; rs = p.rf[gr]; }
        const f32x4 gam = ld4(p.lnf_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dyv[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, dm, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2); px[c] = dyv[c] * xh[c]; }
        dm = d;
        if (p.mask_f != nullptr && ok) {
            const f32x4 mk = ld4(p.mask_f + gr * XD + li * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dm[c] *= mk[c];
        }
        st4(DY + row * XP + li * 4, dyv);
        st4(T1 + row * XP + li * 4, px);
        st4(DX2 + row * XP + li * 4, d);
        st4(DX2M + row * XP + li * 4, dm);
        if (ok) st4(p.dx2 + gr * XD + li * 4, dm);
    }
    __syncthreads();
    tile_colsum(T1, XP, XD, part + o_lnf, tid);
    tile_colsum(DY, XP, XD, part + o_lnf + XD, tid);
    tile_colsum(DX2M, XP, XD, part + o_b2, tid);

    XF_STAMP(1);
    // ---- S2: dg = dx2m W2;  dh = dg * mask_g * GELU'(h) ----
    {
        f32x4 acc[8];
        zero_acc<8>(acc);
        gemm_tile<8, 8, true>(acc, DX2M, XP, p.w2, XMLP, 128 * wave, m, kb);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int c0 = 128 * wave + 64 * gq + 4 * m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                f32x4 dv = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const f32x4 hv = ld4(p.h + (row0 + row) * XMLP + c0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) dv[c] = acc[gq * 4 + c][r] * xgelu_grad(hv[c]);
                    if (p.mask_g != nullptr) {
                        const f32x4 mk = ld4(p.mask_g + (row0 + row) * XMLP + c0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) dv[c] *= mk[c];
                    }
                    st4(p.dh + (row0 + row) * XMLP + c0, dv);
                }
                st4(DH + row * XGP + c0, dv);
            }
        }
    }
    __syncthreads();
    tile_colsum(DH, XGP, XMLP, part + o_b1, tid);

    XF_STAMP(2);
    // ---- S3: df = dh W1 ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gemm_tile<2, 32, true>(acc, DH, XGP, p.w1, XD, 32 * wave, m, kb);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(DF + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
    }
    __syncthreads();

    XF_STAMP(3);
    // ---- S4: LayerNorm 2 backward + the residual gradient dx2 ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        float mu = 0.f, rs = 0.f;
        if (ok) { xv = ld4(p.x1 + gr * XD + li * 4); mu = p.m2[gr]; rs = p.r2[gr]; }
        const f32x4 dfv = ld4(DF + row * XP + li * 4);
        const f32x4 r2v = ld4(DX2 + row * XP + li * 4);
        const f32x4 gam = ld4(p.ln2_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dfv[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, dm, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2) + r2v[c]; px[c] = dfv[c] * xh[c]; }
        dm = d;
        if (p.mask_o != nullptr && ok) {
            const f32x4 mk = ld4(p.mask_o + gr * XD + li * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dm[c] *= mk[c];
        }
        st4(T1 + row * XP + li * 4, px);
        st4(DX1 + row * XP + li * 4, d);
        st4(DX1M + row * XP + li * 4, dm);
        if (ok) st4(p.dx1 + gr * XD + li * 4, dm);
    }
    __syncthreads();
    tile_colsum(T1, XP, XD, part + o_ln2, tid);
    tile_colsum(DF, XP, XD, part + o_ln2 + XD, tid);
    tile_colsum(DX1M, XP, XD, part + o_bo, tid);

    XF_STAMP(4);
    // ---- S5: dout = dx1m Wo ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gemm_tile<2, 8, true>(acc, DX1M, XP, p.wo, XD, 32 * wave, m, kb);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r;
            const f32x2 v = {acc[0][r], acc[1][r]};
            *reinterpret_cast<f32x2*>(DO + row * XP + c0) = v;
            if (row < nv) *reinterpret_cast<f32x2*>(p.dout + (row0 + row) * XD + c0) = v;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) st4(p.doutT + ((size_t)bz * XD + c0 + u) * Npad + t0 + 4 * kb, acc[u]);
    }
    __syncthreads();

    XF_STAMP(5);
    // ---- S6: dQ half of the attention backward, head = wave ----
    {
        const int h = wave;
        const float c = p.scale * XLOG2E;
        const size_t grm = row0 + (m < nv ? m : nv - 1);
        float qreg[2][4], doreg[2][4];
        float delta = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cc = XDH * h + 16 * s + 4 * kb;
            const f32x4 qv = ld4(p.q + grm * XD + cc);
            const f32x4 ov = ld4(p.out + grm * XD + cc);
            const f32x4 dv = ld4(DO + m * XP + cc);                 // zero rows past nv
#pragma unroll
            for (int j = 0; j < 4; ++j) { qreg[s][j] = qv[j] * c; doreg[s][j] = dv[j]; delta += dv[j] * ov[j]; }
        }
        delta += __shfl_xor(delta, 16);
        delta += __shfl_xor(delta, 32);
        const size_t sidx = ((size_t)bz * XH + h) * Npad + t0 + m;
        if (kb == 0) p.delta[sidx] = delta;
        const float lse2 = p.lse[sidx];                             // +inf past nv: p = 0 there
        const int ntiles = Npad >> 4;
        const float* Kb = p.kv + (size_t)bz * N * (2 * XD) + XDH * h + 4 * kb;
        const float* Kt = p.kvT + ((size_t)bz * (2 * XD) + XDH * h + m) * Npad + 4 * kb;
        f32x4 dqT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 k0, k1, v0, v1, kt0, kt1;
        {
            const int key = m < N ? m : N - 1;
            const float* kr = Kb + (size_t)key * (2 * XD);
            k0 = ld4(kr); k1 = ld4(kr + 16); v0 = ld4(kr + XD); v1 = ld4(kr + XD + 16);
            kt0 = ld4(Kt); kt1 = ld4(Kt + (size_t)16 * Npad);
        }
        for (int t = 0; t < ntiles; ++t) {
            f32x4 nk0 = k0, nk1 = k1, nv0 = v0, nv1 = v1, nt0 = kt0, nt1 = kt1;
            if (t + 1 < ntiles) {
                int key = 16 * (t + 1) + m;
                key = key < N ? key : N - 1;
                const float* kr = Kb + (size_t)key * (2 * XD);
                nk0 = ld4(kr); nk1 = ld4(kr + 16); nv0 = ld4(kr + XD); nv1 = ld4(kr + XD + 16);
                nt0 = ld4(Kt + 16 * (t + 1)); nt1 = ld4(Kt + (size_t)16 * Npad + 16 * (t + 1));
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[j], qreg[0][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[j], doreg[0][j], dp, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[j], qreg[1][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[j], doreg[1][j], dp, 0, 0, 0);
            }
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * kb + r;
                const float pr = key < N ? exp2f(s[r] - lse2) : 0.f;
                ds[r] = pr * (dp[r] - delta) * p.scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dqT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt0[r], ds[r], dqT[0], 0, 0, 0);
                dqT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt1[r], ds[r], dqT[1], 0, 0, 0);
            }
            k0 = nk0; k1 = nk1; v0 = nv0; v1 = nv1; kt0 = nt0; kt1 = nt1;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int cc = XDH * h + 16 * dt + 4 * kb;
            f32x4 w4 = dqT[dt];
            if (m >= nv) w4 = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(DQ + m * XP + cc, w4);
            if (m < nv) st4(p.dq + (row0 + m) * XD + cc, w4);
        }
    }
    __syncthreads();

    XF_STAMP(6);
    // ---- S7: da = dq Wq ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gemm_tile<2, 8, true>(acc, DQ, XP, p.wq, XD, 32 * wave, m, kb);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(DA + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
    }
    __syncthreads();

    XF_STAMP(7);
    // ---- S8: LayerNorm 1 backward + the residual gradients dx1 (attention block) and dy (the caller's "+ tokens") ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        float mu = 0.f, rs = 0.f;
        if (ok) { xv = ld4(p.x + gr * XD + li * 4); mu = p.m1[gr]; rs = p.r1[gr]; }
        const f32x4 dav = ld4(DA + row * XP + li * 4);
        const f32x4 r1v = ld4(DX1 + row * XP + li * 4);
        const f32x4 ryv = ld4(DY + row * XP + li * 4);
        const f32x4 gam = ld4(p.ln1_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dav[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2) + r1v[c] + ryv[c]; px[c] = dav[c] * xh[c]; }
        st4(T1 + row * XP + li * 4, px);
        if (ok) st4(p.dx + gr * XD + li * 4, d);
    }
    __syncthreads();
    tile_colsum(T1, XP, XD, part + o_ln1, tid);
    tile_colsum(DA, XP, XD, part + o_ln1 + XD, tid);
    XF_STAMP(8);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, key side
// ---------------------------------------------------------------------------------------------------------------------
struct XfBwdKvArgs {
    const float *q, *qT;            // queries of the instance: row-major [R][128], transposed [B][128][Npad]
    const float* kv;                // K | V of the context, row-major
    const float *lse, *delta;       // [B][heads][Npad]
    const float *dout, *doutT;
    const float* wkv;
    const float* dctx_acc;          // what the context tensor's gradient has collected so far, or NULL
    float scale;
    float* dkv;                     // [R][256] (dy operand of to_kv's weight gradient)
    float* dctx;                    // [R][128]
    int B, N, Npad, tiles;
    unsigned long long* trace;
};

constexpr int XKP = 2 * XD + 4;

__global__ __launch_bounds__(XTHR, 1) void xf_bwd_kv_kernel(const XfBwdKvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DKV = smem;                      // [16][XKP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4;
    const int tile = blockIdx.x, bz = blockIdx.y;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;
    const size_t row0 = (size_t)bz * N + t0;
    XF_STAMP(0);
    {
        const int h = wave;
        const float c = p.scale * XLOG2E;
        // this lane's key (B operand column j = m): K row scaled, V row
        float kreg[2][4], vreg[2][4];
        {
            const float* kr = p.kv + (row0 + (m < nv ? m : nv - 1)) * (2 * XD) + XDH * h + 4 * kb;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 kv4 = ld4(kr + 16 * s), vv4 = ld4(kr + XD + 16 * s);
#pragma unroll
                for (int j = 0; j < 4; ++j) { kreg[s][j] = kv4[j] * c; vreg[s][j] = vv4[j]; }
            }
        }
        const int ntiles = Npad >> 4;
        const float* Qb = p.q + (size_t)bz * N * XD + XDH * h + 4 * kb;
        const float* Db = p.dout + (size_t)bz * N * XD + XDH * h + 4 * kb;
        const float* Qt = p.qT + ((size_t)bz * XD + XDH * h + m) * Npad + 4 * kb;
        const float* Dt = p.doutT + ((size_t)bz * XD + XDH * h + m) * Npad + 4 * kb;
        const float* Ls = p.lse + ((size_t)bz * XH + h) * Npad + 4 * kb;
        const float* Dl = p.delta + ((size_t)bz * XH + h) * Npad + 4 * kb;
        f32x4 dkT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dvT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 qa0, qa1, da0, da1, qt0, qt1, dt0, dt1, ls4, dl4;
        {
            const int qr = m < N ? m : N - 1;
            qa0 = ld4(Qb + (size_t)qr * XD); qa1 = ld4(Qb + (size_t)qr * XD + 16);
            da0 = ld4(Db + (size_t)qr * XD); da1 = ld4(Db + (size_t)qr * XD + 16);
            qt0 = ld4(Qt); qt1 = ld4(Qt + (size_t)16 * Npad);
            dt0 = ld4(Dt); dt1 = ld4(Dt + (size_t)16 * Npad);
            ls4 = ld4(Ls); dl4 = ld4(Dl);
        }
        for (int t = 0; t < ntiles; ++t) {
            f32x4 nqa0 = qa0, nqa1 = qa1, nda0 = da0, nda1 = da1, nqt0 = qt0, nqt1 = qt1, ndt0 = dt0, ndt1 = dt1, nls = ls4, ndl = dl4;
            if (t + 1 < ntiles) {
                int qr = 16 * (t + 1) + m;
                qr = qr < N ? qr : N - 1;
                nqa0 = ld4(Qb + (size_t)qr * XD); nqa1 = ld4(Qb + (size_t)qr * XD + 16);
                nda0 = ld4(Db + (size_t)qr * XD); nda1 = ld4(Db + (size_t)qr * XD + 16);
                nqt0 = ld4(Qt + 16 * (t + 1)); nqt1 = ld4(Qt + (size_t)16 * Npad + 16 * (t + 1));
                ndt0 = ld4(Dt + 16 * (t + 1)); ndt1 = ld4(Dt + (size_t)16 * Npad + 16 * (t + 1));
                nls = ld4(Ls + 16 * (t + 1)); ndl = ld4(Dl + 16 * (t + 1));
            }
            // s[r] = S[query 16 t + 4 kb + r][key m];  dp likewise
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[j], kreg[0][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(da0[j], vreg[0][j], dp, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(qa1[j], kreg[1][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(da1[j], vreg[1][j], dp, 0, 0, 0);
            }
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = exp2f(s[r] - ls4[r]);                   // lse = +inf on padded queries: 0
                ds[r] = pr[r] * (dp[r] - dl4[r]) * p.scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dvT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dt0[r], pr[r], dvT[0], 0, 0, 0);
                dvT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dt1[r], pr[r], dvT[1], 0, 0, 0);
                dkT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt0[r], ds[r], dkT[0], 0, 0, 0);
                dkT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt1[r], ds[r], dkT[1], 0, 0, 0);
            }
            qa0 = nqa0; qa1 = nqa1; da0 = nda0; da1 = nda1; qt0 = nqt0; qt1 = nqt1; dt0 = ndt0; dt1 = ndt1; ls4 = nls; dl4 = ndl;
        }
        // lane (m = key, kb): dkT[dt][r] = dK[key m][feature 32 h + 16 dt + 4 kb + r]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int cc = XDH * h + 16 * dt + 4 * kb;
            f32x4 a = dkT[dt], b = dvT[dt];
            if (m >= nv) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
            st4(DKV + m * XKP + cc, a);
            st4(DKV + m * XKP + XD + cc, b);
            if (m < nv) {
                st4(p.dkv + (row0 + m) * (2 * XD) + cc, a);
                st4(p.dkv + (row0 + m) * (2 * XD) + XD + cc, b);
            }
        }
    }
    XF_STAMP(1);
    __syncthreads();
    // ---- dctx = dkv Wkv + what the context has collected ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gemm_tile<2, 16, true>(acc, DKV, XKP, p.wkv, XD, 32 * wave, m, kb);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r;
            if (row < nv) {
                f32x2 v = {acc[0][r], acc[1][r]};
                if (p.dctx_acc != nullptr) {
                    const f32x2 a = *reinterpret_cast<const f32x2*>(p.dctx_acc + (row0 + row) * XD + c0);
                    v[0] += a[0]; v[1] += a[1];
                }
                *reinterpret_cast<f32x2*>(p.dctx + (row0 + row) * XD + c0) = v;
            }
        }
    }
    XF_STAMP(2);
}

// Column sums of the per-tile partials of ALL instances in one launch: out = sum over the tile rows, fp64 accumulation in
// fixed order.  Columns [0, nsmall) go to small[i], the rest to lnf[i].
constexpr int XF_MAX_INST = 2 * TMF_FUSION_MAX_DEPTH;
struct XfColsumArgs {
    const float* part[XF_MAX_INST];
    float* small[XF_MAX_INST];
    float* lnf[XF_MAX_INST];
    int nblk, stride, nsmall;
};

__global__ __launch_bounds__(64 * 16) void xf_colsum_kernel(const XfColsumArgs p) {
    __shared__ double red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, i = blockIdx.y;
    const float* src = p.part[i];
    double a = 0.0;
    if (c < p.stride)
        for (int r = ty; r < p.nblk; r += 16) a += (double)src[(size_t)r * p.stride + c];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < p.stride) {
#pragma unroll
        for (int k = 1; k < 16; ++k) a += red[k][tx];
        if (c < p.nsmall) p.small[i][c] = (float)a;
        else p.lnf[i][c - p.nsmall] = (float)a;
    }
}

constexpr size_t XF_FWD_LDS = (size_t)(6 * XT * XP + XT * XGP) * 4;
constexpr size_t XF_BWDQ_LDS = (size_t)(10 * XT * XP + XT * XGP) * 4;
constexpr size_t XF_BWDKV_LDS = (size_t)(XT * XKP) * 4;

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side: launchers used by fusion_path.hip (C++ linkage, declared there)
// ---------------------------------------------------------------------------------------------------------------------
static unsigned long long* g_xf_trace[3] = {nullptr, nullptr, nullptr};
// debugging hook (tools/xf_trace.py): per-wave phase time stamps of the fused kernels, [workgroups][4 waves][16] uint64 each
extern "C" void tmf_debug_xf_trace(void* fwd, void* bwd_q, void* bwd_kv) {
    g_xf_trace[0] = (unsigned long long*)fwd; g_xf_trace[1] = (unsigned long long*)bwd_q; g_xf_trace[2] = (unsigned long long*)bwd_kv;
}

bool tmf_xf_supported(int N, int dim, int heads, int dim_head, int mlp) {
    return dim == XD && heads == XH && dim_head == XDH && mlp == XMLP && N >= 1 && N <= 512;
}
int tmf_xf_npad(int N) { return (N + XT - 1) / XT * XT; }
int tmf_xf_tiles(int N) { return (N + XT - 1) / XT; }
int tmf_xf_part_stride(void) { return 8 * XD + XMLP; }

struct tmf_xf_fwd_io {
    const float *x, *kv, *kvT, *wkv_next, *mask_o, *mask_g, *mask_f;
    float *a, *q, *qT, *out, *lse, *x1, *f, *h, *g, *x2, *y, *m1, *r1, *m2, *r2, *mf, *rf, *kv_next, *kvT_next;
};

int tmf_xf_launch_fwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_fwd_io* io, float scale, int only_kv,
                      hipStream_t s) {
    XfFwdArgs a = {};
    a.x = io->x; a.kv = io->kv; a.kvT = io->kvT; a.wkv_next = io->wkv_next;
    a.mask_o = io->mask_o; a.mask_g = io->mask_g; a.mask_f = io->mask_f;
    if (w != nullptr) {
        a.ln1_g = w->ln1_g; a.ln1_b = w->ln1_b; a.wq = w->wq; a.wo = w->wo; a.bo = w->bo; a.ln2_g = w->ln2_g; a.ln2_b = w->ln2_b;
        a.w1 = w->w1; a.b1 = w->b1; a.w2 = w->w2; a.b2 = w->b2; a.lnf_g = w->lnf_g; a.lnf_b = w->lnf_b;
        a.eps1 = w->eps1; a.eps2 = w->eps2; a.epsf = w->epsf;
    }
    a.scale = scale;
    a.a = io->a; a.q = io->q; a.qT = io->qT; a.out = io->out; a.lse = io->lse; a.x1 = io->x1; a.f = io->f; a.h = io->h; a.g = io->g;
    a.x2 = io->x2; a.y = io->y; a.m1 = io->m1; a.r1 = io->r1; a.m2 = io->m2; a.r2 = io->r2; a.mf = io->mf; a.rf = io->rf;
    a.kv_next = io->kv_next; a.kvT_next = io->kvT_next;
    a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N); a.only_kv = only_kv;
    a.trace = only_kv ? nullptr : g_xf_trace[0];
    const dim3 grid(a.tiles, B), block(XTHR);
    const int mt = a.Npad / 16;
    int rc;
#define XF_LAUNCH(MT)                                                                         \
    {                                                                                         \
        auto kf = xf_fwd_kernel<MT>;                                                          \
        if ((rc = tmf_allow_lds(kf, XF_FWD_LDS, "tmf_fusion_train_fwd(fused)"))) return rc;   \
        hipLaunchKernelGGL(kf, grid, block, XF_FWD_LDS, s, a);                                \
    }
    if (mt <= 8) XF_LAUNCH(8)
    else if (mt <= 16) XF_LAUNCH(16)
    else XF_LAUNCH(32)
#undef XF_LAUNCH
    return tmf_launch_result("tmf_fusion_train_fwd(fused)");
}

struct tmf_xf_bwd_io {
    const float *dy, *x, *kv, *kvT, *mask_o, *mask_g, *mask_f;
    const float *q, *qT, *out, *lse, *x1, *h, *x2, *m1, *r1, *m2, *r2, *mf, *rf;
    float *dx2, *dh, *dx1, *dq, *dout, *doutT, *delta, *dx, *part, *dkv, *dctx;
    const float* dctx_acc;
};

int tmf_xf_launch_bwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_bwd_io* io, float scale, hipStream_t s) {
    int rc;
    {
        XfBwdQArgs a = {};
        a.dy = io->dy; a.x = io->x; a.kv = io->kv; a.kvT = io->kvT;
        a.ln1_g = w->ln1_g; a.wq = w->wq; a.wo = w->wo; a.ln2_g = w->ln2_g; a.w1 = w->w1; a.w2 = w->w2; a.lnf_g = w->lnf_g;
        a.mask_o = io->mask_o; a.mask_g = io->mask_g; a.mask_f = io->mask_f;
        a.scale = scale;
        a.q = io->q; a.out = io->out; a.lse = io->lse; a.x1 = io->x1; a.h = io->h; a.x2 = io->x2;
        a.m1 = io->m1; a.r1 = io->r1; a.m2 = io->m2; a.r2 = io->r2; a.mf = io->mf; a.rf = io->rf;
        a.dx2 = io->dx2; a.dh = io->dh; a.dx1 = io->dx1; a.dq = io->dq; a.dout = io->dout; a.doutT = io->doutT; a.delta = io->delta;
        a.dx = io->dx; a.part = io->part; a.stride = tmf_xf_part_stride();
        a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N);
        a.trace = g_xf_trace[1];
        auto kf = xf_bwd_q_kernel;
        if ((rc = tmf_allow_lds(kf, XF_BWDQ_LDS, "tmf_fusion_train_bwd(fused q)"))) return rc;
        hipLaunchKernelGGL(kf, dim3(a.tiles, B), dim3(XTHR), XF_BWDQ_LDS, s, a);
        if ((rc = tmf_launch_result("tmf_fusion_train_bwd(fused q)"))) return rc;
    }
    {
        XfBwdKvArgs a = {};
        a.q = io->q; a.qT = io->qT; a.kv = io->kv; a.lse = io->lse; a.delta = io->delta; a.dout = io->dout; a.doutT = io->doutT;
        a.wkv = w->wkv; a.dctx_acc = io->dctx_acc; a.scale = scale; a.dkv = io->dkv; a.dctx = io->dctx;
        a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N);
        a.trace = g_xf_trace[2];
        hipLaunchKernelGGL(xf_bwd_kv_kernel, dim3(a.tiles, B), dim3(XTHR), XF_BWDKV_LDS, s, a);
        if ((rc = tmf_launch_result("tmf_fusion_train_bwd(fused kv)"))) return rc;
    }
    return TMF_OK;
}

int tmf_xf_launch_colsum(int n_inst, const float* const* part, float* const* small, float* const* lnf, int nblk, hipStream_t s) {
    XfColsumArgs a = {};
    for (int i = 0; i < n_inst; ++i) { a.part[i] = part[i]; a.small[i] = small[i]; a.lnf[i] = lnf[i]; }
    a.nblk = nblk; a.stride = tmf_xf_part_stride(); a.nsmall = 6 * XD + XMLP;
    hipLaunchKernelGGL(xf_colsum_kernel, dim3(tmf_cdiv(a.stride, 64), n_inst), dim3(64 * 16), 0, s, a);
    return tmf_launch_result("tmf_fusion_train_bwd(fused colsum)");
}
