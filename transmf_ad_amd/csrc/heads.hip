// heads.hip — the dense heads of model_ad in ONE launch per direction.
//
// reference: models/mymodel.py:190-194 (`fc_cls` = Linear(4*dim, 512)-BatchNorm1d-ReLU-Dropout(.5)-Linear(512, 64)-
// BatchNorm1d-ReLU-Dropout(.5)-Linear(64, 2);  `D` = Linear(dim, 128)-BatchNorm1d-ReLU-Linear(128, 2)), driven by
// model_ad.forward :209-215, 221: D is applied TWICE — to revgrad(mean over tokens of the MRI embedding, 2.0) and to the
// PET one — each call with its own batch statistics, the running statistics updated twice in that order.
//
// Why a kernel for (B, <= 512) matrices: as stock torch modules these heads are ~25 forward and ~40 backward launches of
// a few microseconds, and backward STARTS with them, right after the reference step's two loss.item() host syncs
// (kfold_train_adversarial.py:127-128) have drained the queue: the GPU idles for the 1.3 ms the host needs to issue
// them (tools/host_probe.py) before the first real kernel of backward can start.  One workgroup does the whole thing
// in tens of microseconds; a thread owns an output feature, so the BatchNorm1d batch statistics (over B <= 16 rows) are
// register-local and nothing is reduced across threads except the K-split of the 512 -> 64 layer.
//
// Numerics: fp32 fma chains in k order per output; BatchNorm1d as torch (biased variance for normalisation, unbiased
// into running_var, momentum, eps); Dropout by a caller-supplied keep-mask already scaled by 1 / (1 - p) (NULL = none).
#include "tmf_common.h"

namespace {

constexpr int HT = 512;          // threads
constexpr int MAXB = 32;          // batch rows the kernels hold in registers: instances MB = 16 (B <= 16) and MB = 32
constexpr int ND = 4;            // workgroups of the discriminator's backward (j-ranges of D.0)

struct HeadsArgs {
    // inputs
    const float* cls;            // [B][C4]
    const float* tok[2];         // [B][N][dim] MRI, PET tokens (the D inputs are their means over N)
    const float* mask1;          // [B][H1] or NULL
    const float* mask2;          // [B][H2] or NULL
    // fc_cls parameters
    const float *w0, *b0, *g1, *be1; float *rm1, *rv1;     // Linear(C4, H1), BatchNorm1d(H1)
    const float *w4, *b4, *g5, *be5; float *rm5, *rv5;     // Linear(H1, H2), BatchNorm1d(H2)
    const float *w8, *b8;                                   // Linear(H2, NC)
    // D parameters
    const float *dw0, *db0, *dg1, *dbe1; float *drm1, *drv1;   // Linear(dim, HD), BatchNorm1d(HD)
    const float *dw3, *db3;                                     // Linear(HD, NC)
    // outputs
    float* logits;               // [B][NC]
    float* dlog[2];              // [B][NC] D(MRI), D(PET)
    float* saved;                // tmf_heads_saved_floats()
    int B, N, dim, C4, H1, H2, HD, NC, training;
    float mom1, eps1, mom5, eps5, dmom, deps;
};

struct SavedPlan { int v, xhD, isD, xh1, is1, a1, xh2, is2, a2, total; };
__host__ __device__ inline SavedPlan saved_plan(int B, int dim, int H1, int H2, int HD) {
    SavedPlan p;
    int o = 0;
    p.v = o; o += 2 * B * dim;          // token means (the D inputs)
    p.xhD = o; o += 2 * B * HD;         // normalised pre-activations of D's BatchNorm, per call
    p.isD = o; o += 2 * HD;
    p.xh1 = o; o += B * H1;
    p.is1 = o; o += H1;
    p.a1 = o; o += B * H1;              // input of Linear(H1, H2): after ReLU and dropout
    p.xh2 = o; o += B * H2;
    p.is2 = o; o += H2;
    p.a2 = o; o += B * H2;
    p.total = o;
    return p;
}

// BatchNorm1d of one feature over the batch held in acc[0..B): returns xhat in place, writes y = gamma*xhat + beta to yv
template <int MB>
__device__ __forceinline__ void bn1d(float (&acc)[MB], float (&yv)[MB], int B, bool training, float g, float be,
                                     float* rmean, float* rvar, float mom, float eps, float& invstd, bool update) {
    float mean, var;
    if (training) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) s += acc[b];
        mean = s / B;
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) { const float d = acc[b] - mean; q += d * d; }
        var = q / B;
        if (update && rmean != nullptr) {
            *rmean = (1.f - mom) * *rmean + mom * mean;
            *rvar = (1.f - mom) * *rvar + mom * (B > 1 ? q / (B - 1) : var);
        }
    } else {
        mean = *rmean;
        var = *rvar;
    }
    invstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < B) {
            acc[b] = (acc[b] - mean) * invstd;
            yv[b] = acc[b] * g + be;
        }
}

// v[m][b][c] = mean over the N tokens of tok_m[b][n][c]: one workgroup per (modality, sample), 4 token slices per channel
// (one thread walking 216-512 dependent-latency loads is what made a single-workgroup version of the heads slow)
__global__ __launch_bounds__(HT) void token_mean_kernel(const float* __restrict__ mri, const float* __restrict__ pet,
                                                        float* __restrict__ v, int B, int N, int dim) {
    __shared__ float part[HT];
    const int m = blockIdx.x / B, b = blockIdx.x % B;
    const float* src = (m == 0 ? mri : pet) + (size_t)b * N * dim;
    const int t = threadIdx.x;
    for (int c0 = 0; c0 < dim; c0 += 128) {
        const int c = c0 + (t & 127), sl = t >> 7;            // 4 slices of tokens
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < dim) {
            int n = sl;
            for (; n + 12 < N; n += 16) {
                s0 += src[(size_t)n * dim + c];
                s1 += src[(size_t)(n + 4) * dim + c];
                s2 += src[(size_t)(n + 8) * dim + c];
                s3 += src[(size_t)(n + 12) * dim + c];
            }
            for (; n < N; n += 4) s0 += src[(size_t)n * dim + c];
        }
        part[t] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl == 0 && c < dim) v[((size_t)m * B + b) * dim + c] = (part[t] + part[t + 128] + part[t + 256] + part[t + 384]) / N;
        __syncthreads();
    }
}

// fc_cls.0 / .1 / .2 / .3 on H1 / 32 workgroups: 32 output features per workgroup, 16 threads per feature split K in
// interleaved 16-byte pieces (the 16 threads of a feature read 256 contiguous bytes of its weight row per step; all of a
// thread's loads are in flight at once).  A feature's batch column ends up in one thread: BatchNorm1d is register-local.
template <int MB>
__global__ __launch_bounds__(HT) void heads_fc0_kernel(HeadsArgs a) {
    extern __shared__ float lds[];
    const int t = threadIdx.x, B = a.B;
    const SavedPlan sp = saved_plan(B, a.dim, a.H1, a.H2, a.HD);
    for (int e = t; e < B * a.C4; e += HT) lds[e] = a.cls[e];
    __syncthreads();
    const int j = blockIdx.x * 32 + (t >> 4), part = t & 15;
    float acc[MB], y[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = 0.f;
    if (j < a.H1) {
        const float* wr = a.w0 + (size_t)j * a.C4;
#pragma unroll 8
        for (int k = part * 4; k < a.C4; k += 64) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    const float* x = lds + b * a.C4 + k;
                    acc[b] = fmaf(w[0], x[0], acc[b]); acc[b] = fmaf(w[1], x[1], acc[b]);
                    acc[b] = fmaf(w[2], x[2], acc[b]); acc[b] = fmaf(w[3], x[3], acc[b]);
                }
        }
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        acc[b] += __shfl_xor(acc[b], 1);
        acc[b] += __shfl_xor(acc[b], 2);
        acc[b] += __shfl_xor(acc[b], 4);
        acc[b] += __shfl_xor(acc[b], 8);
    }
    if (j < a.H1 && part == 0) {
        const float bias = a.b0[j];
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] += bias;
        float is;
        bn1d<MB>(acc, y, B, a.training != 0, a.g1[j], a.be1[j], a.rm1 ? a.rm1 + j : nullptr, a.rv1 ? a.rv1 + j : nullptr, a.mom1,
             a.eps1, is, true);
        a.saved[sp.is1 + j] = is;
#pragma unroll
        for (int b = 0; b < MB; ++b)
            if (b < B) {
                a.saved[sp.xh1 + b * a.H1 + j] = acc[b];
                float r = y[b] > 0.f ? y[b] : 0.f;
                if (a.mask1 != nullptr) r *= a.mask1[b * a.H1 + j];
                a.saved[sp.a1 + b * a.H1 + j] = r;
            }
    }
}

// The discriminator D = Linear(dim, HD)-BatchNorm1d-ReLU-Linear(HD, NC) applied to BOTH token means l_v [2][B][dim] by one
// workgroup (mymodel.py:150-153 / :209-215: two calls, each with its own batch statistics, the running statistics updated
// MRI call first, then PET).  l_r [2][B][HD], l_st [2][HD][2], l_w3 [NC][HD] are LDS scratch; o_isD / o_xhD = offsets of
// the saved inverse standard deviations / normalised pre-activations in a.saved.
template <int MB>
__device__ __forceinline__ void disc_fwd(const HeadsArgs& a, int o_isD, int o_xhD, const float* l_v, float* l_r, float* l_st,
                                         float* l_w3) {
    const int t = threadIdx.x, B = a.B;
    // ---- D.0 / .1 / .2 on both token means: 2 threads per (call, output feature) split K ----
    for (int e0 = 0; e0 < 2 * a.HD; e0 += HT / 2) {
        const int e = e0 + (t >> 1), part = t & 1;
        const bool live = e < 2 * a.HD;
        const int j = live ? e % a.HD : 0, m = live ? e / a.HD : 0;
        float acc[MB], y[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] = 0.f;
        if (live) {
            const float* wr = a.dw0 + (size_t)j * a.dim;
            const int kper = a.dim / 2;                       // dim % 8 == 0: whole float4s
#pragma unroll 8
            for (int k = part * kper; k < (part + 1) * kper; k += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    if (b < B) {
                        const float* x = l_v + (m * B + b) * a.dim + k;
                        acc[b] = fmaf(w[0], x[0], acc[b]); acc[b] = fmaf(w[1], x[1], acc[b]);
                        acc[b] = fmaf(w[2], x[2], acc[b]); acc[b] = fmaf(w[3], x[3], acc[b]);
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] += __shfl_xor(acc[b], 1);
        if (!live || part != 0) continue;
        {
            const float bias = a.db0[j];
#pragma unroll
            for (int b = 0; b < MB; ++b) acc[b] += bias;
        }
        // batch statistics of THIS call; the running buffers are updated below, MRI call first, then PET
        float mean = 0.f, q = 0.f;
        if (a.training) {
#pragma unroll
            for (int b = 0; b < MB; ++b) if (b < B) mean += acc[b];
            mean /= B;
#pragma unroll
            for (int b = 0; b < MB; ++b) if (b < B) { const float d = acc[b] - mean; q += d * d; }
            l_st[(m * a.HD + j) * 2] = mean;
            l_st[(m * a.HD + j) * 2 + 1] = B > 1 ? q / (B - 1) : q / B;
        }
        float is;
        bn1d<MB>(acc, y, B, a.training != 0, a.dg1[j], a.dbe1[j], a.drm1 ? a.drm1 + j : nullptr, a.drv1 ? a.drv1 + j : nullptr,
             a.dmom, a.deps, is, false);
        a.saved[o_isD + m * a.HD + j] = is;
#pragma unroll
        for (int b = 0; b < MB; ++b)
            if (b < B) {
                a.saved[o_xhD + (m * B + b) * a.HD + j] = acc[b];
                l_r[(m * B + b) * a.HD + j] = y[b] > 0.f ? y[b] : 0.f;
            }
    }
    __syncthreads();
    if (a.training && a.drm1 != nullptr)
        for (int j = t; j < a.HD; j += HT) {
            float rm = a.drm1[j], rv = a.drv1[j];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                rm = (1.f - a.dmom) * rm + a.dmom * l_st[(m * a.HD + j) * 2];
                rv = (1.f - a.dmom) * rv + a.dmom * l_st[(m * a.HD + j) * 2 + 1];
            }
            a.drm1[j] = rm;
            a.drv1[j] = rv;
        }
    // ---- D.3 (its small weight matrix staged in LDS) ----
    for (int e = t; e < a.NC * a.HD; e += HT) l_w3[e] = a.dw3[e];
    __syncthreads();
    for (int e = t; e < 2 * B * a.NC; e += HT) {
        const int c = e % a.NC, b = (e / a.NC) % B, m = e / (a.NC * B);
        float s = a.db3[c];
        for (int k = 0; k < a.HD; ++k) s = fmaf(l_w3[c * a.HD + k], l_r[(m * B + b) * a.HD + k], s);
        a.dlog[m][b * a.NC + c] = s;
    }
}

template <int MB>
__global__ __launch_bounds__(HT) void heads_fwd_kernel(HeadsArgs a) {
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    const int B = a.B;
    const SavedPlan sp = saved_plan(B, a.dim, a.H1, a.H2, a.HD);
    // (the two roles run in different workgroups and share the allocation: fwd_lds() = the larger plan)
    float* l_a1 = lds;                           // fc role: [B][H1]            D role: D.3's weights [NC][HD]
    float* l_a2 = l_a1 + B * a.H1;               //          [B][H2]
    const int wD = a.NC * a.HD;
    float* l_v = lds + wD;                       // D role:  [2][B][dim]
    float* l_r = l_v + 2 * B * a.dim;            //          [2][B][HD]  (ReLU output of D's hidden layer)
    float* l_st = l_r + 2 * B * a.HD;            //          [2][HD][2]  batch mean / unbiased var of the two D calls
    // two workgroups, two independent chains: block 0 = fc_cls.4 .. fc_cls.8 (logits), block 1 = D on both token means
    const bool fc_role = blockIdx.x == 0;
    // ---- token means (token_mean_kernel) and the first hidden layer's output (heads_fc0_kernel) from `saved` ----
    if (fc_role) { for (int e = t; e < B * a.H1; e += HT) l_a1[e] = a.saved[sp.a1 + e]; }
    else { for (int e = t; e < 2 * B * a.dim; e += HT) l_v[e] = a.saved[sp.v + e]; }
    __syncthreads();
    if (fc_role) {
    // ---- fc_cls.4 / .5 / .6 / .7 : 8 threads per output feature split K ----
    for (int j0 = 0; j0 < a.H2; j0 += HT / 8) {
        const int j = j0 + (t >> 3), part = t & 7;
        float acc[MB], y[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] = 0.f;
        if (j < a.H2) {
            const int kper = (a.H1 + 7) / 8;
            const int k0 = part * kper, k1 = k0 + kper < a.H1 ? k0 + kper : a.H1;
            const float* wr = a.w4 + (size_t)j * a.H1;
#pragma unroll 8
            for (int k = k0; k < k1; k += 4) {                  // H1 % 32 == 0: every part is a whole number of float4s
                const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    if (b < B) {
                        const float* x = l_a1 + b * a.H1 + k;
                        acc[b] = fmaf(w[0], x[0], acc[b]); acc[b] = fmaf(w[1], x[1], acc[b]);
                        acc[b] = fmaf(w[2], x[2], acc[b]); acc[b] = fmaf(w[3], x[3], acc[b]);
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            acc[b] += __shfl_xor(acc[b], 1);
            acc[b] += __shfl_xor(acc[b], 2);
            acc[b] += __shfl_xor(acc[b], 4);
        }
        if (j < a.H2 && part == 0) {
            const float bias = a.b4[j];
#pragma unroll
            for (int b = 0; b < MB; ++b) acc[b] += bias;
            float is;
            bn1d<MB>(acc, y, B, a.training != 0, a.g5[j], a.be5[j], a.rm5 ? a.rm5 + j : nullptr, a.rv5 ? a.rv5 + j : nullptr,
                 a.mom5, a.eps5, is, true);
            a.saved[sp.is2 + j] = is;
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    a.saved[sp.xh2 + b * a.H2 + j] = acc[b];
                    float r = y[b] > 0.f ? y[b] : 0.f;
                    if (a.mask2 != nullptr) r *= a.mask2[b * a.H2 + j];
                    l_a2[b * a.H2 + j] = r;
                    a.saved[sp.a2 + b * a.H2 + j] = r;
                }
        }
    }
        // ---- fc_cls.8 (its small weight matrix staged in LDS: l_a1 is free by now) ----
        __syncthreads();
        float* l_w8 = l_a1;                         // [NC][H2]
        for (int e = t; e < a.NC * a.H2; e += HT) l_w8[e] = a.w8[e];
        __syncthreads();
        for (int e = t; e < B * a.NC; e += HT) {
            const int c = e % a.NC, b = e / a.NC;
            float s = a.b8[c];
            for (int k = 0; k < a.H2; ++k) s = fmaf(l_w8[c * a.H2 + k], l_a2[b * a.H2 + k], s);
            a.logits[e] = s;
        }
        return;
    }
    disc_fwd<MB>(a, sp.isD, sp.xhD, l_v, l_r, l_st, l_a1);           // block 1 (l_a1 is unused by this block: D.3's weights)
}

struct HeadsBwdArgs {
    HeadsArgs f;                 // the forward's arguments (inputs, parameters, saved)
    const float* d_logits;       // [B][NC]
    const float* d_dlog[2];      // [B][NC]
    // parameter gradients (same shapes as the parameters)
    float *gw0, *gb0, *gg1, *gbe1, *gw4, *gb4, *gg5, *gbe5, *gw8, *gb8;
    float *gdw0, *gdb0, *gdg1, *gdbe1, *gdw3, *gdb3;
    float* d_cls;                // [B][C4]
    float* d_tok[2];             // [B][N][dim]
    float alpha;                 // gradient reversal factor (mymodel.py:209: 2.0)
    float* s_dz1;                // scratch [B][H1], [B][H2], [2][B][dim]: inputs of heads_bwd_outer_kernel
    float* s_dz2;
    float* s_dv;
};

// backward of y = gamma*xhat + beta followed by ReLU (and an optional scaled keep-mask) for one feature:
// in: dr[b] = gradient w.r.t. the masked ReLU output; out: dz[b] w.r.t. the BatchNorm input; dgamma, dbeta
template <int MB>
__device__ __forceinline__ void bn1d_relu_bwd(float (&dr)[MB], const float* xhat, int stride, const float* mask, int B,
                                              bool training, float g, float be, float invstd, float& dgamma, float& dbeta) {
    float sg = 0.f, sx = 0.f;
    float xh[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < B) {
            xh[b] = xhat[b * stride];
            const float y = xh[b] * g + be;
            float gr = y > 0.f ? dr[b] : 0.f;
            if (mask != nullptr) gr *= mask[b * stride];
            dr[b] = gr;
            sg += gr;
            sx += gr * xh[b];
        }
    dgamma = sx;
    dbeta = sg;
    const float k = g * invstd;
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < B) dr[b] = training ? k * (dr[b] - sg / B - xh[b] * (sx / B)) : k * dr[b];
}

// Backward of disc_fwd on ND workgroups (dq = 0 .. ND-1; `first` = the one that stores the small results): every one
// recomputes the D.3 / BatchNorm1d(HD) part, then takes the dq-th j-range of the D.0 backward — its rows of dW0 and its
// partial of the token-mean gradient, a.s_dv[dq][2][B][dim] (summed, scaled by -alpha / N and broadcast over the tokens by the
// caller's second launch).  l_dzD [2][B][HD]; l_x: LDS scratch of max(6 HD, 2 B dim NS) floats.
template <int MB>
__device__ __forceinline__ void disc_bwd(const HeadsBwdArgs& a, int o_xhD, int o_isD, int o_v, bool first, int dq, float* l_dzD,
                                         float* l_x) {
    const HeadsArgs& f = a.f;
    const int t = threadIdx.x, B = f.B;
    const bool tr = f.training != 0;
    // ---- D.3 backward and through ReLU / BatchNorm1d(HD), both calls ----
    for (int e = t; first && e < f.NC * f.HD; e += HT) { // dW3[c][k] = sum_m sum_b dd[m][b][c] r[m][b][k]
        const int k = e % f.HD, c = e / f.HD;
        float s = 0.f;
        for (int m = 0; m < 2; ++m)
            for (int b = 0; b < B; ++b) {
                const float y = f.saved[o_xhD + (m * B + b) * f.HD + k] * f.dg1[k] + f.dbe1[k];
                s = fmaf(a.d_dlog[m][b * f.NC + c], y > 0.f ? y : 0.f, s);
            }
        a.gdw3[e] = s;
    }
    for (int c = t; first && c < f.NC; c += HT) {
        float s = 0.f;
        for (int m = 0; m < 2; ++m) for (int b = 0; b < B; ++b) s += a.d_dlog[m][b * f.NC + c];
        a.gdb3[c] = s;
    }
    float* l_pg = l_x;                                   // [2][HD][3]: per-call dgamma, dbeta, dbias partials
    __syncthreads();                                     // l_x (cls) is free from here on
    for (int e = t; e < 2 * f.HD; e += HT) {
        const int j = e % f.HD, m = e / f.HD;
        float dr[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            float s = 0.f;
            if (b < B) for (int c = 0; c < f.NC; ++c) s = fmaf(a.d_dlog[m][b * f.NC + c], f.dw3[c * f.HD + j], s);
            dr[b] = s;
        }
        float dg, db;
        bn1d_relu_bwd<MB>(dr, f.saved + o_xhD + m * B * f.HD + j, f.HD, nullptr, B, tr, f.dg1[j], f.dbe1[j],
                      f.saved[o_isD + m * f.HD + j], dg, db);
        float sb = 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) { l_dzD[(m * B + b) * f.HD + j] = dr[b]; sb += dr[b]; }
        l_pg[(m * f.HD + j) * 3] = dg;
        l_pg[(m * f.HD + j) * 3 + 1] = db;
        l_pg[(m * f.HD + j) * 3 + 2] = sb;
    }
    __syncthreads();
    for (int j = t; first && j < f.HD; j += HT) {        // the shared D parameters collect both calls
        a.gdg1[j] = l_pg[j * 3] + l_pg[(f.HD + j) * 3];
        a.gdbe1[j] = l_pg[j * 3 + 1] + l_pg[(f.HD + j) * 3 + 1];
        a.gdb0[j] = l_pg[j * 3 + 2] + l_pg[(f.HD + j) * 3 + 2];
    }
    // ---- D.0 backward: dW0D[j][i] = sum_m sum_b dzD[m][b][j] v[m][b][i];  dv[m][b][i] = sum_j dzD[m][b][j] W0D[j][i] ----
    //      a thread = (input channel i, one of NS j-slices): all of its weight loads are in flight at once
    float* l_dvp = l_x;                                 // [NS][2][B][dim] partial dv: NS <= 2 fits the B x 4*dim region
    __syncthreads();                                    // (its previous contents, the per-call partials, are consumed)
    {
        const int NS = HT / f.dim >= 2 ? 2 : 1;
        const int i = t % f.dim, sl = t / f.dim;
        if (sl < NS && t < NS * f.dim) {
            float v[2 * MB], dv[2 * MB];
#pragma unroll
            for (int q = 0; q < 2 * MB; ++q) {
                const int m = q / MB, b = q % MB;
                v[q] = b < B ? f.saved[o_v + (m * B + b) * f.dim + i] : 0.f;
                dv[q] = 0.f;
            }
            const int jblk = (f.HD + ND - 1) / ND;          // this block's j-range, NS thread slices inside it
            const int jb0 = dq * jblk, jb1 = jb0 + jblk < f.HD ? jb0 + jblk : f.HD;
            const int jper = (jblk + NS - 1) / NS;
            const int j0 = jb0 + sl * jper, j1 = j0 + jper < jb1 ? j0 + jper : jb1;
#pragma unroll 8
            for (int j = j0; j < j1; ++j) {
                const float w = f.dw0[(size_t)j * f.dim + i];
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 2 * MB; ++q) {
                    const int m = q / MB, b = q % MB;
                    if (b < B) {
                        const float dz = l_dzD[(m * B + b) * f.HD + j];
                        s = fmaf(dz, v[q], s);
                        dv[q] = fmaf(dz, w, dv[q]);
                    }
                }
                a.gdw0[(size_t)j * f.dim + i] = s;
            }
#pragma unroll
            for (int q = 0; q < 2 * MB; ++q) {
                const int m = q / MB, b = q % MB;
                if (b < B) l_dvp[((sl * 2 + m) * B + b) * f.dim + i] = dv[q];
            }
        }
        __syncthreads();
        for (int e = t; e < 2 * B * f.dim; e += HT) {   // broadcast over the tokens: heads_bwd_outer_kernel
            float sum = 0.f;
            for (int q = 0; q < NS; ++q) sum += l_dvp[q * 2 * B * f.dim + e];
            a.s_dv[(size_t)dq * 2 * B * f.dim + e] = sum;
        }
    }
}

// Roles by workgroup: blocks 0 .. nA-1 run the fc_cls chain (every one recomputes the small fc_cls.8 / BatchNorm1d(H2) part,
// block 0 stores its results; then block i takes H1 / nA of the fc_cls.4 / BatchNorm1d(H1) features, 8 j-slices per feature
// so that a thread's chain of dependent weight loads is H2 / 8 long instead of H2); blocks nA .. nA + ND - 1 run the D path
// (every one recomputes the small D.3 / BatchNorm1d(HD) part, the first stores its results; block q takes the q-th j-range
// of the D.0 backward and writes its partial of the token-mean gradient, summed by heads_bwd_outer_kernel).
template <int MB>
__global__ __launch_bounds__(HT) void heads_bwd_kernel(HeadsBwdArgs a, int nA) {
    extern __shared__ float lds[];
    const HeadsArgs& f = a.f;
    const int t = threadIdx.x;
    const int B = f.B;
    const SavedPlan sp = saved_plan(B, f.dim, f.H1, f.H2, f.HD);
    const bool tr = f.training != 0;
    const int blk = blockIdx.x;
    const bool d_role = blk >= nA, first = blk == 0 || blk == nA;
    const int dq = blk - nA;                             // D role: which j-range of D.0
    // (the two roles run in different workgroups and share the allocation: bwd_lds() = the larger plan)
    float* l_dz2 = lds;                                            // fc role: [B][H2], then l_x = the j-slice partials [8][B][64]
    float* l_dzD = lds;                                            // D role:  [2][B][HD], then l_x = disc_bwd's scratch
    float* l_x = d_role ? l_dzD + 2 * B * f.HD : l_dz2 + B * f.H2;
    if (!d_role) {
    // ---- fc_cls.8 backward, then through Dropout / ReLU / BatchNorm1d(H2) ----
    for (int e = t; first && e < f.NC * f.H2; e += HT) { // dW8[c][k] = sum_b dlogits[b][c] a2[b][k]
        const int k = e % f.H2, c = e / f.H2;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a.d_logits[b * f.NC + c], f.saved[sp.a2 + b * f.H2 + k], s);
        a.gw8[e] = s;
    }
    for (int c = t; first && c < f.NC; c += HT) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += a.d_logits[b * f.NC + c];
        a.gb8[c] = s;
    }
    for (int j = t; j < f.H2; j += HT) {
        float dr[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            float s = 0.f;
            if (b < B) for (int c = 0; c < f.NC; ++c) s = fmaf(a.d_logits[b * f.NC + c], f.w8[c * f.H2 + j], s);
            dr[b] = s;
        }
        float dg, db;
        bn1d_relu_bwd<MB>(dr, f.saved + sp.xh2 + j, f.H2, f.mask2 ? f.mask2 + j : nullptr, B, tr, f.g5[j], f.be5[j],
                      f.saved[sp.is2 + j], dg, db);
        float sb = 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) { l_dz2[b * f.H2 + j] = dr[b]; sb += dr[b]; }
        if (first) {
            a.gg5[j] = dg;
            a.gbe5[j] = db;
            a.gb4[j] = sb;
#pragma unroll
            for (int b = 0; b < MB; ++b) if (b < B) a.s_dz2[b * f.H2 + j] = dr[b];
        }
    }
    __syncthreads();
    // ---- fc_cls.4 backward: da1[b][k] = sum_j dz2[b][j] W4[j][k]  (dW4 = dz2^T a1: heads_bwd_outer_kernel) ----
    //      this block's features k in chunks of 64: thread = (feature t & 63, j-slice t >> 6), partials through LDS
    float* l_part = l_x;                                 // [8][B][64]
    const int kper = (f.H1 + nA - 1) / nA;
    const int kbeg = blk * kper, kend = kbeg + kper < f.H1 ? kbeg + kper : f.H1;
    for (int kc = kbeg; kc < kend; kc += 64) {
        const int kk = t & 63, sl = t >> 6, k = kc + kk;
        const int jper = (f.H2 + 7) / 8;
        const int j0 = sl * jper, j1 = j0 + jper < f.H2 ? j0 + jper : f.H2;
        float dr[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) dr[b] = 0.f;
        if (k < kend) {
#pragma unroll 8
            for (int j = j0; j < j1; ++j) {
                const float w = f.w4[(size_t)j * f.H1 + k];
#pragma unroll
                for (int b = 0; b < MB; ++b) if (b < B) dr[b] = fmaf(l_dz2[b * f.H2 + j], w, dr[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) l_part[(sl * B + b) * 64 + kk] = dr[b];
        __syncthreads();
        if (sl == 0 && k < kend) {
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += l_part[(q * B + b) * 64 + kk];
                    dr[b] = sum;
                }
            float dg, db;
            bn1d_relu_bwd<MB>(dr, f.saved + sp.xh1 + k, f.H1, f.mask1 ? f.mask1 + k : nullptr, B, tr, f.g1[k], f.be1[k],
                          f.saved[sp.is1 + k], dg, db);
            a.gg1[k] = dg;
            a.gbe1[k] = db;
            float sb = 0.f;
#pragma unroll
            for (int b = 0; b < MB; ++b) if (b < B) { a.s_dz1[b * f.H1 + k] = dr[b]; sb += dr[b]; }
            a.gb0[k] = sb;
        }
        __syncthreads();
    }
    return;
    }
    // (fc_cls.0 backward — dcls = dz1 W0 and dW0 = dz1^T cls — runs on many workgroups: heads_bwd_outer_kernel)
    disc_bwd<MB>(a, sp.xhD, sp.isD, sp.v, first, dq, l_dzD, l_x);       // blocks nA .. nA + ND - 1
}

// The wide parts of the backward on many workgroups:
//   blocks [0, nbc)        : dcls[b][k] = sum_j dz1[b][j] W0[j][k]        (32 columns per block, 16 j-slices)
//   next nb0 blocks        : dW0[j][k] = sum_b dz1[b][j] cls[b][k]        (8 rows j per block, a thread per column k)
//   next nb4 blocks        : dW4[j][k] = sum_b dz2[b][j] a1[b][k]
//   the rest               : mean over tokens + gradient reversal: d tok[m][b][n][c] = -alpha * dv[m][b][c] / N
template <int MB>
__global__ __launch_bounds__(HT) void heads_bwd_outer_kernel(HeadsBwdArgs a, int nb0, int nb4, int nbc) {
    __shared__ float part[16][MB][32];
    const HeadsArgs& f = a.f;
    const int t = threadIdx.x, B = f.B;
    int blk = blockIdx.x;
    if (blk < nbc) {
        // dcls[b][k] = sum_j dz1[b][j] W0[j][k]: 32 columns per workgroup, 16 j-slices of H1 / 16 rows each (a thread's
        // chain of dependent weight loads is what this launch waits for)
        const int kk = t & 31, sl = t >> 5, k = blk * 32 + kk;
        float dc[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) dc[b] = 0.f;
        const int jper = (f.H1 + 15) / 16;
        const int j0 = sl * jper, j1 = j0 + jper < f.H1 ? j0 + jper : f.H1;
        if (k < f.C4) {
#pragma unroll 8
            for (int j = j0; j < j1; ++j) {
                const float w = f.w0[(size_t)j * f.C4 + k];
#pragma unroll
                for (int b = 0; b < MB; ++b) if (b < B) dc[b] = fmaf(a.s_dz1[b * f.H1 + j], w, dc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) part[sl][b][kk] = dc[b];
        __syncthreads();
        for (int e = t; e < B * 32; e += HT) {
            const int b = e >> 5, c = e & 31;
            if (blk * 32 + c < f.C4) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) sum += part[q][b][c];
                a.d_cls[b * f.C4 + blk * 32 + c] = sum;
            }
        }
        return;
    }
    blk -= nbc;
    if (blk < nb0 + nb4) {
        const bool first = blk < nb0;
        if (!first) blk -= nb0;
        const int rows = first ? f.H1 : f.H2, cols = first ? f.C4 : f.H1;
        const float* dz = first ? a.s_dz1 : a.s_dz2;                         // [B][rows]
        const SavedPlan sp = saved_plan(B, f.dim, f.H1, f.H2, f.HD);
        const float* x = first ? f.cls : f.saved + sp.a1;                    // [B][cols]
        float* gw = first ? a.gw0 : a.gw4;
        for (int k = t; k < cols; k += HT) {
            float xv[MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) xv[b] = b < B ? x[b * cols + k] : 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int j = blk * 8 + r;
                if (j < rows) {
                    float s = 0.f;
#pragma unroll
                    for (int b = 0; b < MB; ++b) if (b < B) s = fmaf(dz[b * rows + j], xv[b], s);
                    gw[(size_t)j * cols + k] = s;
                }
            }
        }
        return;
    }
    blk -= nb0 + nb4;
    const float sc = -a.alpha / f.N;
    const size_t per = (size_t)B * f.N * f.dim, total = 2 * per;
    for (size_t e = ((size_t)blk * HT + t) * 4; e < total; e += (size_t)(gridDim.x - nb0 - nb4 - nbc) * HT * 4) {
        const int m = e >= per ? 1 : 0;
        const size_t o = e - m * per;
        const int c = o % f.dim, b = o / ((size_t)f.N * f.dim);              // dim % 4 == 0: the 4 elements share (m, b)
        const float* dv = a.s_dv + ((size_t)m * B + b) * f.dim + c;
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < ND; ++q) {                        // the ND j-range partials of heads_bwd_kernel, fixed order
            const float* dq_ = dv + (size_t)q * 2 * B * f.dim;
            acc4[0] += dq_[0]; acc4[1] += dq_[1]; acc4[2] += dq_[2]; acc4[3] += dq_[3];
        }
        *reinterpret_cast<f32x4*>(a.d_tok[m] + o) = f32x4{sc * acc4[0], sc * acc4[1], sc * acc4[2], sc * acc4[3]};
    }
}

// ------------------------------------------------------------------------------------------------------------
// The heads of the CNN-only models in one launch per direction (+ the token mean / the broadcast over the tokens):
//   model_CNN_ad (mymodel.py:143-178): fc_cls = Linear(2 dim, H)-ReLU-Linear(H, NC) on cat[gap(mri), gap(pet)], and D on
//                                      revgrad(gap(mri), 2), revgrad(gap(pet), 2)  — M = 2, HD > 0;
//   model_single (mymodel.py:13-41):   fc = Linear(dim, H)-ReLU-Linear(H, NC) on gap(img)                 — M = 1, HD = 0.
// HeadsArgs is reused: tok[], w0 / b0 = the hidden layer [H1][C4] with C4 = M dim, w8 / b8 = the output layer [NC][H1], the
// d* fields = D; saved = saved_plan(B, dim, H1, 0, HD) with `a1` = the ReLU output and `v` = the token means.
// ------------------------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(HT) void cnn_heads_fwd_kernel(HeadsArgs a) {
    extern __shared__ float lds[];
    const int t = threadIdx.x, B = a.B, M = a.C4 / a.dim;
    const SavedPlan sp = saved_plan(B, a.dim, a.H1, 0, a.HD);
    const int wide = a.H1 > 2 * a.HD ? a.H1 : 2 * a.HD;
    float* l_v = lds;                            // [M][B][dim]
    float* l_h = l_v + 2 * B * a.dim;            // [B][H1]            (D role: [2][B][HD])
    float* l_w = l_h + B * wide;                 // [NC][H1]           (D role: [NC][HD])
    float* l_st = l_w + a.NC * wide;             // [2][HD][2]
    for (int e = t; e < M * B * a.dim; e += HT) l_v[e] = a.saved[sp.v + e];
    __syncthreads();
    if (blockIdx.x == 1) { disc_fwd<MB>(a, sp.isD, sp.xhD, l_v, l_h, l_st, l_w); return; }
    // ---- hidden layer + ReLU: 4 threads per feature split K; x[b][k] = v[k / dim][b][k % dim] is the concatenation ----
    for (int j0 = 0; j0 < a.H1; j0 += HT / 4) {
        const int j = j0 + (t >> 2), part = t & 3;
        float acc[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] = 0.f;
        if (j < a.H1) {
            const float* wr = a.w0 + (size_t)j * a.C4;
            const int kper = a.C4 / 4;                          // C4 % 16 == 0: whole float4s, none straddles a modality
#pragma unroll 8
            for (int k = part * kper; k < (part + 1) * kper; k += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
                const float* xm = l_v + (size_t)(k / a.dim) * B * a.dim + k % a.dim;
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    if (b < B) {
                        const float* x = xm + b * a.dim;
                        acc[b] = fmaf(w[0], x[0], acc[b]); acc[b] = fmaf(w[1], x[1], acc[b]);
                        acc[b] = fmaf(w[2], x[2], acc[b]); acc[b] = fmaf(w[3], x[3], acc[b]);
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            acc[b] += __shfl_xor(acc[b], 1);
            acc[b] += __shfl_xor(acc[b], 2);
        }
        if (j < a.H1 && part == 0) {
            const float bias = a.b0[j];
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    const float r = acc[b] + bias > 0.f ? acc[b] + bias : 0.f;
                    l_h[b * a.H1 + j] = r;
                    a.saved[sp.a1 + b * a.H1 + j] = r;
                }
        }
    }
    // ---- output layer ----
    for (int e = t; e < a.NC * a.H1; e += HT) l_w[e] = a.w8[e];
    __syncthreads();
    for (int e = t; e < B * a.NC; e += HT) {
        const int c = e % a.NC, b = e / a.NC;
        float s = a.b8[c];
        for (int k = 0; k < a.H1; ++k) s = fmaf(l_w[c * a.H1 + k], l_h[b * a.H1 + k], s);
        a.logits[e] = s;
    }
}

// block 0: the fc chain (output layer, ReLU, hidden layer: its weight gradient and the gradient of the concatenated means,
// a.s_dz1 [B][C4]); blocks 1 .. ND: disc_bwd.  cnn_heads_bwd_tok_kernel then spreads both over the tokens.
template <int MB>
__global__ __launch_bounds__(HT) void cnn_heads_bwd_kernel(HeadsBwdArgs a) {
    extern __shared__ float lds[];
    const HeadsArgs& f = a.f;
    const int t = threadIdx.x, B = f.B;
    const SavedPlan sp = saved_plan(B, f.dim, f.H1, 0, f.HD);
    float* l_dz = lds;                        // [B][H1]
    float* l_dzD = l_dz + B * f.H1;           // [2][B][HD]
    float* l_x = l_dzD + 2 * B * f.HD;        // scratch: [NS][B][C4] | disc_bwd's
    if (blockIdx.x > 0) { disc_bwd<MB>(a, sp.xhD, sp.isD, sp.v, blockIdx.x == 1, (int)blockIdx.x - 1, l_dzD, l_x); return; }
    const float* h = f.saved + sp.a1;
    for (int e = t; e < f.NC * f.H1; e += HT) {          // dW2[c][k] = sum_b dlogits[b][c] h[b][k]
        const int k = e % f.H1, c = e / f.H1;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a.d_logits[b * f.NC + c], h[b * f.H1 + k], s);
        a.gw8[e] = s;
    }
    for (int c = t; c < f.NC; c += HT) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += a.d_logits[b * f.NC + c];
        a.gb8[c] = s;
    }
    for (int k = t; k < f.H1; k += HT) {                 // through the ReLU
        float sb = 0.f;
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int c = 0; c < f.NC; ++c) s = fmaf(a.d_logits[b * f.NC + c], f.w8[c * f.H1 + k], s);
            s = h[b * f.H1 + k] > 0.f ? s : 0.f;
            l_dz[b * f.H1 + k] = s;
            sb += s;
        }
        a.gb0[k] = sb;
    }
    __syncthreads();
    // hidden layer: dW0[k][i] = sum_b dz[b][k] x[b][i];  dx[b][i] = sum_k dz[b][k] W0[k][i]; thread = (column i, k-slice)
    const int NS = HT / f.C4 >= 2 ? 2 : 1;
    const int i = t % f.C4, sl = t / f.C4;
    if (sl < NS) {
        float x[MB], dx[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            x[b] = b < B ? f.saved[sp.v + ((size_t)(i / f.dim) * B + b) * f.dim + i % f.dim] : 0.f;
            dx[b] = 0.f;
        }
        const int kper = (f.H1 + NS - 1) / NS;
        const int k0 = sl * kper, k1 = k0 + kper < f.H1 ? k0 + kper : f.H1;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float w = f.w0[(size_t)k * f.C4 + i];
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    const float dz = l_dz[b * f.H1 + k];
                    s = fmaf(dz, x[b], s);
                    dx[b] = fmaf(dz, w, dx[b]);
                }
            a.gw0[(size_t)k * f.C4 + i] = s;
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) if (b < B) l_x[(sl * B + b) * f.C4 + i] = dx[b];
    }
    __syncthreads();
    for (int e = t; e < B * f.C4; e += HT) {
        float sum = 0.f;
        for (int q = 0; q < NS; ++q) sum += l_x[q * B * f.C4 + e];
        a.s_dz1[e] = sum;
    }
}

// d tok[m][b][n][c] = (dx[b][m dim + c] - alpha * sum_q dvD[q][m][b][c]) / N: the mean over the tokens backwards, the
// discriminator's part through the gradient reversal (mymodel.py:150-151)
__global__ __launch_bounds__(HT) void cnn_heads_bwd_tok_kernel(HeadsBwdArgs a) {
    const HeadsArgs& f = a.f;
    const int B = f.B, M = f.C4 / f.dim;
    const float inv = 1.0f / f.N, sc = -a.alpha / f.N;
    const size_t per = (size_t)B * f.N * f.dim, total = (size_t)M * per;
    for (size_t e = ((size_t)blockIdx.x * HT + threadIdx.x) * 4; e < total; e += (size_t)gridDim.x * HT * 4) {
        const int m = e >= per ? 1 : 0;
        const size_t o = e - m * per;
        const int c = o % f.dim, b = o / ((size_t)f.N * f.dim);              // dim % 4 == 0: the 4 elements share (m, b)
        const float* dx = a.s_dz1 + (size_t)b * f.C4 + m * f.dim + c;
        f32x4 r = {inv * dx[0], inv * dx[1], inv * dx[2], inv * dx[3]};
        if (f.HD > 0) {
            const float* dv = a.s_dv + ((size_t)m * B + b) * f.dim + c;
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const float* dq_ = dv + (size_t)q * 2 * B * f.dim;
                acc4[0] += dq_[0]; acc4[1] += dq_[1]; acc4[2] += dq_[2]; acc4[3] += dq_[3];
            }
            r = f32x4{fmaf(sc, acc4[0], r[0]), fmaf(sc, acc4[1], r[1]), fmaf(sc, acc4[2], r[2]), fmaf(sc, acc4[3], r[3])};
        }
        *reinterpret_cast<f32x4*>(a.d_tok[m] + o) = r;
    }
}

// the kernel instance for a batch: 16 rows in registers where that is enough, 32 above
#define HEADS_K(K, B) ((B) <= 16 ? K<16> : K<32>)

int check_cnn_heads(const char* fn, const tmf_heads_cnn_desc* d) {
    TMF_REQUIRE_PTR(d);
    TMF_REQUIRE(d->B > 0 && d->B <= MAXB, TMF_E_SHAPE, "%s: batch %d (1..%d)", fn, d->B, MAXB);
    TMF_REQUIRE(d->M == 1 || d->M == 2, TMF_E_SHAPE, "%s: M=%d modalities (1 or 2)", fn, d->M);
    TMF_REQUIRE(d->N > 0 && d->dim > 0 && d->H > 0 && d->HD >= 0 && d->NC > 0 && d->NC <= 16, TMF_E_SHAPE,
                "%s: non-positive dimension", fn);
    TMF_REQUIRE(d->dim % 16 == 0 && d->M * d->dim <= HT, TMF_E_SHAPE, "%s: dim=%d must be a multiple of 16 and M*dim <= %d", fn,
                d->dim, HT);
    TMF_REQUIRE(d->HD == 0 || d->M == 2, TMF_E_SHAPE, "%s: the discriminator takes both modalities (M=2)", fn);
    return TMF_OK;
}
HeadsArgs make_cnn_args(const tmf_heads_cnn_desc& d, const float* mri_tok, const float* pet_tok, const tmf_heads_cnn_params& p,
                        float* saved) {
    HeadsArgs a = {};
    a.tok[0] = mri_tok; a.tok[1] = pet_tok;
    a.w0 = p.fc0_w; a.b0 = p.fc0_b; a.w8 = p.fc2_w; a.b8 = p.fc2_b;
    a.dw0 = p.d0_w; a.db0 = p.d0_b; a.dg1 = p.dbn_g; a.dbe1 = p.dbn_b; a.drm1 = p.dbn_rm; a.drv1 = p.dbn_rv;
    a.dw3 = p.d3_w; a.db3 = p.d3_b;
    a.saved = saved;
    a.B = d.B; a.N = d.N; a.dim = d.dim; a.C4 = d.M * d.dim; a.H1 = d.H; a.H2 = 0; a.HD = d.HD; a.NC = d.NC;
    a.training = d.training; a.dmom = d.momentum; a.deps = d.eps;
    return a;
}
size_t cnn_fwd_lds(const tmf_heads_cnn_desc& d) {
    const int wide = d.H > 2 * d.HD ? d.H : 2 * d.HD;
    return (size_t)(2 * d.B * d.dim + d.B * wide + d.NC * wide + 4 * d.HD) * 4;
}
size_t cnn_bwd_lds(const tmf_heads_cnn_desc& d) {
    size_t lx = (size_t)2 * d.B * d.M * d.dim;                       // [NS][B][C4]
    if (lx < (size_t)6 * d.HD) lx = (size_t)6 * d.HD;                // disc_bwd: per-call partials, then [NS][2][B][dim]
    if (lx < (size_t)4 * d.B * d.dim) lx = (size_t)4 * d.B * d.dim;
    return ((size_t)d.B * d.H + 2 * d.B * d.HD + lx) * 4;
}

int check_heads(const char* fn, const tmf_heads_desc* d) {
    TMF_REQUIRE_PTR(d);
    TMF_REQUIRE(d->B > 0 && d->B <= MAXB, TMF_E_SHAPE, "%s: batch %d (1..%d)", fn, d->B, MAXB);
    TMF_REQUIRE(d->N > 0 && d->dim > 0 && d->H1 > 0 && d->H2 > 0 && d->HD > 0 && d->NC > 0 && d->NC <= 16, TMF_E_SHAPE,
                "%s: non-positive dimension", fn);
    TMF_REQUIRE(d->dim % 8 == 0 && d->H1 % 32 == 0, TMF_E_SHAPE, "%s: dim=%d must be a multiple of 8 and H1=%d of 32", fn,
                d->dim, d->H1);
    return TMF_OK;
}
size_t fwd_lds(const tmf_heads_desc& d) {
    const size_t fc = (size_t)d.B * d.H1 + (size_t)d.B * d.H2;
    const size_t fc8 = (size_t)d.NC * d.H2;                                     // fc_cls.8's weights re-use l_a1
    const size_t dd = (size_t)d.NC * d.HD + 2 * d.B * d.dim + 2 * d.B * d.HD + 4 * d.HD;
    size_t n = fc > dd ? fc : dd;
    if (n < fc8) n = fc8;
    return n * 4;
}
size_t bwd_lds(const tmf_heads_desc& d) {
    const size_t fc = (size_t)d.B * d.H2 + (size_t)8 * d.B * 64;               // dz2 + the j-slice partials of the fc_cls.4 backward
    size_t dx = (size_t)2 * 2 * d.B * d.dim;                                    // disc_bwd: [NS <= 2][2][B][dim] partial dv ...
    if (dx < (size_t)6 * d.HD) dx = (size_t)6 * d.HD;                           // ... after the per-call partials [2][HD][3]
    const size_t dd = (size_t)2 * d.B * d.HD + dx;
    return (fc > dd ? fc : dd) * 4;
}
HeadsArgs make_args(const tmf_heads_desc& d, const float* cls, const float* mri_tok, const float* pet_tok, const float* mask1,
                    const float* mask2, const tmf_heads_params& p, float* saved) {
    HeadsArgs a = {};
    a.cls = cls; a.tok[0] = mri_tok; a.tok[1] = pet_tok; a.mask1 = mask1; a.mask2 = mask2;
    a.w0 = p.fc0_w; a.b0 = p.fc0_b; a.g1 = p.bn1_g; a.be1 = p.bn1_b; a.rm1 = p.bn1_rm; a.rv1 = p.bn1_rv;
    a.w4 = p.fc4_w; a.b4 = p.fc4_b; a.g5 = p.bn5_g; a.be5 = p.bn5_b; a.rm5 = p.bn5_rm; a.rv5 = p.bn5_rv;
    a.w8 = p.fc8_w; a.b8 = p.fc8_b;
    a.dw0 = p.d0_w; a.db0 = p.d0_b; a.dg1 = p.dbn_g; a.dbe1 = p.dbn_b; a.drm1 = p.dbn_rm; a.drv1 = p.dbn_rv;
    a.dw3 = p.d3_w; a.db3 = p.d3_b;
    a.saved = saved;
    a.B = d.B; a.N = d.N; a.dim = d.dim; a.C4 = 4 * d.dim; a.H1 = d.H1; a.H2 = d.H2; a.HD = d.HD; a.NC = d.NC;
    a.training = d.training;
    a.mom1 = d.momentum[0]; a.eps1 = d.eps[0]; a.mom5 = d.momentum[1]; a.eps5 = d.eps[1]; a.dmom = d.momentum[2]; a.deps = d.eps[2];
    return a;
}
int check_params(const char* fn, const tmf_heads_params* p) {
    TMF_REQUIRE_PTR(p);
    TMF_REQUIRE(p->fc0_w && p->fc0_b && p->bn1_g && p->bn1_b && p->fc4_w && p->fc4_b && p->bn5_g && p->bn5_b && p->fc8_w &&
                p->fc8_b && p->d0_w && p->d0_b && p->dbn_g && p->dbn_b && p->d3_w && p->d3_b, TMF_E_NULL,
                "%s: a parameter pointer is NULL", fn);
    return TMF_OK;
}

}  // namespace

namespace {
// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words.  Counter-based, so a keep-mask
// is a pure function of (seed, call offset, element index): no generator state on the device, reproducible under a seed.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Scaled Dropout keep-masks of up to TMF_MASK_SEGMENTS tensors in ONE launch: element e of segment s is 1 / keep_s with
// probability keep_s, else 0 (u = 24 random bits; keep iff u < keep * 2^24).  Four elements per Philox call.
struct MaskSeg { float* out; long n; float keep, inv; };
struct MaskArgs { MaskSeg seg[TMF_MASK_SEGMENTS]; int nseg; unsigned long long seed, offset; };
__global__ __launch_bounds__(256) void dropout_masks_kernel(MaskArgs a) {
    long q = (long)blockIdx.x * 256 + threadIdx.x;            // quad index over all segments
    for (int s = 0; s < a.nseg; ++s) {
        const long nq = (a.seg[s].n + 3) >> 2;
        if (q < nq) {
            unsigned r[4];
            // key = torch's seed XOR a domain constant: torch / curand put the offset in c0, c1 and the thread in c2, c3 — the other way
            // round — so with the bare seed as key the two counter spaces would intersect ((q = X, offset = O) here is torch's
            // thread O at offset 4 X) and mask bits could coincide with numbers another torch kernel draws in the same run
            philox4x32_10((unsigned)q, (unsigned)(q >> 32) ^ ((unsigned)s << 24), (unsigned)a.offset, (unsigned)(a.offset >> 32),
                          (unsigned)a.seed ^ 0x746D666Du, (unsigned)(a.seed >> 32) ^ 0x6B736D5Fu, r);
            const float thr = a.seg[s].keep * 16777216.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long e = 4 * q + j;
                if (e < a.seg[s].n) a.seg[s].out[e] = (float)(r[j] >> 8) < thr ? a.seg[s].inv : 0.f;
            }
            return;
        }
        q -= nq;
    }
}
}  // namespace

extern "C" int tmf_dropout_keep_masks(int nseg, float* const* out, const long* numel, const float* keep,
                                      unsigned long long seed, unsigned long long offset, void* stream) {
    TMF_REQUIRE(nseg > 0 && nseg <= TMF_MASK_SEGMENTS, TMF_E_ARG, "tmf_dropout_keep_masks: %d segments (1 .. %d)", nseg,
                TMF_MASK_SEGMENTS);
    TMF_REQUIRE_PTR(out); TMF_REQUIRE_PTR(numel); TMF_REQUIRE_PTR(keep);
    MaskArgs a;
    a.nseg = nseg; a.seed = seed; a.offset = offset;
    long quads = 0;
    for (int s = 0; s < nseg; ++s) {
        TMF_REQUIRE(out[s] != nullptr && numel[s] > 0, TMF_E_ARG, "tmf_dropout_keep_masks: segment %d is empty", s);
        TMF_REQUIRE(keep[s] > 0.f && keep[s] <= 1.f, TMF_E_ARG, "tmf_dropout_keep_masks: keep probability %g of segment %d",
                    (double)keep[s], s);
        a.seg[s] = MaskSeg{out[s], numel[s], keep[s], 1.0f / keep[s]};
        quads += (numel[s] + 3) >> 2;
    }
    TMF_REQUIRE(quads < (1L << 31) * 256L, TMF_E_SHAPE, "tmf_dropout_keep_masks: too many elements");
    hipLaunchKernelGGL(dropout_masks_kernel, dim3((unsigned)tmf_cdiv(quads, 256L)), dim3(256), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_dropout_keep_masks");
}

extern "C" size_t tmf_heads_saved_bytes(const tmf_heads_desc* d) {
    if (check_heads("tmf_heads_saved_bytes", d) != TMF_OK) return 0;
    return (size_t)saved_plan(d->B, d->dim, d->H1, d->H2, d->HD).total * 4;
}

extern "C" int tmf_heads_fwd(const tmf_heads_desc* d, const float* cls, const float* mri_tok, const float* pet_tok,
                             const float* mask1, const float* mask2, const tmf_heads_params* p, float* logits,
                             float* d_mri_logits, float* d_pet_logits, void* saved, size_t saved_bytes, void* stream) {
    int rc;
    if ((rc = check_heads("tmf_heads_fwd", d))) return rc;
    if ((rc = check_params("tmf_heads_fwd", p))) return rc;
    TMF_REQUIRE_PTR(cls); TMF_REQUIRE_PTR(mri_tok); TMF_REQUIRE_PTR(pet_tok); TMF_REQUIRE_PTR(logits);
    TMF_REQUIRE_PTR(d_mri_logits); TMF_REQUIRE_PTR(d_pet_logits); TMF_REQUIRE_PTR(saved);
    TMF_REQUIRE(d->training || (p->bn1_rm && p->bn1_rv && p->bn5_rm && p->bn5_rv && p->dbn_rm && p->dbn_rv), TMF_E_NULL,
                "tmf_heads_fwd: eval mode needs the running statistics");
    TMF_REQUIRE(saved_bytes >= tmf_heads_saved_bytes(d), TMF_E_WORKSPACE, "tmf_heads_fwd: saved %zu B < required %zu B",
                saved_bytes, tmf_heads_saved_bytes(d));
    TMF_REQUIRE_ALIGNED(cls); TMF_REQUIRE_ALIGNED(p->fc0_w);
    HeadsArgs a = make_args(*d, cls, mri_tok, pet_tok, mask1, mask2, *p, (float*)saved);
    a.logits = logits; a.dlog[0] = d_mri_logits; a.dlog[1] = d_pet_logits;
    const size_t lds = fwd_lds(*d);
    TMF_REQUIRE(lds <= 150 * 1024, TMF_E_SHAPE, "tmf_heads_fwd: %zu B of LDS needed (batch x widths too large)", lds);
    if ((rc = tmf_allow_lds(HEADS_K(heads_fwd_kernel, d->B), lds, "tmf_heads_fwd"))) return rc;
    hipLaunchKernelGGL(token_mean_kernel, dim3(2 * d->B), dim3(HT), 0, (hipStream_t)stream, mri_tok, pet_tok,
                       (float*)saved + saved_plan(d->B, d->dim, d->H1, d->H2, d->HD).v, d->B, d->N, d->dim);
    if ((rc = tmf_launch_result("tmf_heads_fwd(token mean)"))) return rc;
    const size_t lds0 = (size_t)d->B * 4 * d->dim * 4;
    if ((rc = tmf_allow_lds(HEADS_K(heads_fc0_kernel, d->B), lds0, "tmf_heads_fwd(fc0)"))) return rc;
    hipLaunchKernelGGL(HEADS_K(heads_fc0_kernel, d->B), dim3(tmf_cdiv(d->H1, 32)), dim3(HT), lds0, (hipStream_t)stream, a);
    if ((rc = tmf_launch_result("tmf_heads_fwd(fc0)"))) return rc;
    hipLaunchKernelGGL(HEADS_K(heads_fwd_kernel, d->B), dim3(2), dim3(HT), lds, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_heads_fwd");
}

extern "C" size_t tmf_heads_bwd_scratch_bytes(const tmf_heads_desc* d) {
    if (check_heads("tmf_heads_bwd_scratch_bytes", d) != TMF_OK) return 0;
    return (size_t)(d->B * d->H1 + d->B * d->H2 + ND * 2 * d->B * d->dim) * 4;
}

extern "C" int tmf_heads_bwd(const tmf_heads_desc* d, const float* cls, const float* mask1, const float* mask2,
                             const tmf_heads_params* p, const void* saved, size_t saved_bytes, const float* d_logits,
                             const float* d_d_mri_logits, const float* d_d_pet_logits, const tmf_heads_grads* g,
                             float* d_cls, float* d_mri_tok, float* d_pet_tok, float revgrad_alpha,
                             void* scratch, size_t scratch_bytes, void* stream) {
    int rc;
    if ((rc = check_heads("tmf_heads_bwd", d))) return rc;
    if ((rc = check_params("tmf_heads_bwd", p))) return rc;
    TMF_REQUIRE_PTR(cls); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(d_logits); TMF_REQUIRE_PTR(d_d_mri_logits);
    TMF_REQUIRE_PTR(d_d_pet_logits); TMF_REQUIRE_PTR(g); TMF_REQUIRE_PTR(d_cls); TMF_REQUIRE_PTR(d_mri_tok); TMF_REQUIRE_PTR(d_pet_tok);
    TMF_REQUIRE(g->fc0_w && g->fc0_b && g->bn1_g && g->bn1_b && g->fc4_w && g->fc4_b && g->bn5_g && g->bn5_b && g->fc8_w &&
                g->fc8_b && g->d0_w && g->d0_b && g->dbn_g && g->dbn_b && g->d3_w && g->d3_b, TMF_E_NULL,
                "tmf_heads_bwd: a gradient pointer is NULL");
    TMF_REQUIRE(saved_bytes >= tmf_heads_saved_bytes(d), TMF_E_WORKSPACE, "tmf_heads_bwd: saved %zu B < required %zu B",
                saved_bytes, tmf_heads_saved_bytes(d));
    HeadsBwdArgs a = {};
    a.f = make_args(*d, cls, nullptr, nullptr, mask1, mask2, *p, (float*)const_cast<void*>(saved));
    a.d_logits = d_logits; a.d_dlog[0] = d_d_mri_logits; a.d_dlog[1] = d_d_pet_logits;
    a.gw0 = g->fc0_w; a.gb0 = g->fc0_b; a.gg1 = g->bn1_g; a.gbe1 = g->bn1_b; a.gw4 = g->fc4_w; a.gb4 = g->fc4_b;
    a.gg5 = g->bn5_g; a.gbe5 = g->bn5_b; a.gw8 = g->fc8_w; a.gb8 = g->fc8_b;
    a.gdw0 = g->d0_w; a.gdb0 = g->d0_b; a.gdg1 = g->dbn_g; a.gdbe1 = g->dbn_b; a.gdw3 = g->d3_w; a.gdb3 = g->d3_b;
    a.d_cls = d_cls; a.d_tok[0] = d_mri_tok; a.d_tok[1] = d_pet_tok; a.alpha = revgrad_alpha;
    TMF_REQUIRE_PTR(scratch);
    TMF_REQUIRE(scratch_bytes >= tmf_heads_bwd_scratch_bytes(d), TMF_E_WORKSPACE, "tmf_heads_bwd: scratch %zu B < required %zu B",
                scratch_bytes, tmf_heads_bwd_scratch_bytes(d));
    TMF_REQUIRE_ALIGNED(d_mri_tok); TMF_REQUIRE_ALIGNED(d_pet_tok); TMF_REQUIRE_ALIGNED(scratch);
    a.s_dz1 = (float*)scratch; a.s_dz2 = a.s_dz1 + (size_t)d->B * d->H1; a.s_dv = a.s_dz2 + (size_t)d->B * d->H2;
    const size_t lds = bwd_lds(*d);
    TMF_REQUIRE(lds <= 150 * 1024, TMF_E_SHAPE, "tmf_heads_bwd: %zu B of LDS needed (batch x widths too large)", lds);
    if ((rc = tmf_allow_lds(HEADS_K(heads_bwd_kernel, d->B), lds, "tmf_heads_bwd"))) return rc;
    const int nA = d->H1 % 64 == 0 ? d->H1 / 64 : 1;              // workgroups of the fc_cls chain (+ one for the D path)
    hipLaunchKernelGGL(HEADS_K(heads_bwd_kernel, d->B), dim3(nA + ND), dim3(HT), lds, (hipStream_t)stream, a, nA);
    if ((rc = tmf_launch_result("tmf_heads_bwd"))) return rc;
    const int nb0 = tmf_cdiv(d->H1, 8), nb4 = tmf_cdiv(d->H2, 8);
    int nbt = tmf_cdiv((long)2 * d->B * d->N * d->dim, (long)HT * 4 * 4);
    if (nbt > 256) nbt = 256;
    if (nbt < 1) nbt = 1;
    const int nbc = tmf_cdiv(4 * d->dim, 32);
    hipLaunchKernelGGL(HEADS_K(heads_bwd_outer_kernel, d->B), dim3(nbc + nb0 + nb4 + nbt), dim3(HT), 0, (hipStream_t)stream, a, nb0, nb4, nbc);
    return tmf_launch_result("tmf_heads_bwd(outer)");
}

extern "C" size_t tmf_heads_cnn_saved_bytes(const tmf_heads_cnn_desc* d) {
    if (check_cnn_heads("tmf_heads_cnn_saved_bytes", d) != TMF_OK) return 0;
    return (size_t)saved_plan(d->B, d->dim, d->H, 0, d->HD).total * 4;
}

extern "C" size_t tmf_heads_cnn_bwd_scratch_bytes(const tmf_heads_cnn_desc* d) {
    if (check_cnn_heads("tmf_heads_cnn_bwd_scratch_bytes", d) != TMF_OK) return 0;
    return (size_t)(d->B * d->M * d->dim + ND * 2 * d->B * d->dim) * 4;
}

extern "C" int tmf_heads_cnn_fwd(const tmf_heads_cnn_desc* d, const float* mri_tok, const float* pet_tok,
                                 const tmf_heads_cnn_params* p, float* logits, float* d_mri_logits, float* d_pet_logits,
                                 void* saved, size_t saved_bytes, void* stream) {
    int rc;
    if ((rc = check_cnn_heads("tmf_heads_cnn_fwd", d))) return rc;
    TMF_REQUIRE_PTR(p); TMF_REQUIRE_PTR(mri_tok); TMF_REQUIRE_PTR(logits); TMF_REQUIRE_PTR(saved);
    TMF_REQUIRE(p->fc0_w && p->fc0_b && p->fc2_w && p->fc2_b, TMF_E_NULL, "tmf_heads_cnn_fwd: a parameter pointer is NULL");
    if (d->M == 2) TMF_REQUIRE_PTR(pet_tok);
    if (d->HD > 0) {
        TMF_REQUIRE(p->d0_w && p->d0_b && p->dbn_g && p->dbn_b && p->d3_w && p->d3_b, TMF_E_NULL,
                    "tmf_heads_cnn_fwd: a discriminator parameter pointer is NULL");
        TMF_REQUIRE(d->training || (p->dbn_rm && p->dbn_rv), TMF_E_NULL, "tmf_heads_cnn_fwd: eval mode needs the running statistics");
        TMF_REQUIRE_PTR(d_mri_logits); TMF_REQUIRE_PTR(d_pet_logits);
    }
    TMF_REQUIRE(saved_bytes >= tmf_heads_cnn_saved_bytes(d), TMF_E_WORKSPACE, "tmf_heads_cnn_fwd: saved %zu B < required %zu B",
                saved_bytes, tmf_heads_cnn_saved_bytes(d));
    TMF_REQUIRE_ALIGNED(p->fc0_w); TMF_REQUIRE_ALIGNED(saved);
    HeadsArgs a = make_cnn_args(*d, mri_tok, pet_tok, *p, (float*)saved);
    a.logits = logits; a.dlog[0] = d_mri_logits; a.dlog[1] = d_pet_logits;
    const size_t lds = cnn_fwd_lds(*d);
    TMF_REQUIRE(lds <= 150 * 1024, TMF_E_SHAPE, "tmf_heads_cnn_fwd: %zu B of LDS needed (batch x widths too large)", lds);
    if ((rc = tmf_allow_lds(HEADS_K(cnn_heads_fwd_kernel, d->B), lds, "tmf_heads_cnn_fwd"))) return rc;
    hipLaunchKernelGGL(token_mean_kernel, dim3(d->M * d->B), dim3(HT), 0, (hipStream_t)stream, mri_tok, pet_tok,
                       (float*)saved + saved_plan(d->B, d->dim, d->H, 0, d->HD).v, d->B, d->N, d->dim);
    if ((rc = tmf_launch_result("tmf_heads_cnn_fwd(token mean)"))) return rc;
    hipLaunchKernelGGL(HEADS_K(cnn_heads_fwd_kernel, d->B), dim3(d->HD > 0 ? 2 : 1), dim3(HT), lds, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_heads_cnn_fwd");
}

extern "C" int tmf_heads_cnn_bwd(const tmf_heads_cnn_desc* d, const tmf_heads_cnn_params* p, const void* saved, size_t saved_bytes,
                                 const float* d_logits, const float* d_d_mri_logits, const float* d_d_pet_logits,
                                 const tmf_heads_cnn_grads* g, float* d_mri_tok, float* d_pet_tok, float revgrad_alpha,
                                 void* scratch, size_t scratch_bytes, void* stream) {
    int rc;
    if ((rc = check_cnn_heads("tmf_heads_cnn_bwd", d))) return rc;
    TMF_REQUIRE_PTR(p); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(d_logits); TMF_REQUIRE_PTR(g); TMF_REQUIRE_PTR(d_mri_tok);
    TMF_REQUIRE_PTR(scratch);
    TMF_REQUIRE(p->fc0_w && p->fc0_b && p->fc2_w && p->fc2_b, TMF_E_NULL, "tmf_heads_cnn_bwd: a parameter pointer is NULL");
    TMF_REQUIRE(g->fc0_w && g->fc0_b && g->fc2_w && g->fc2_b, TMF_E_NULL, "tmf_heads_cnn_bwd: a gradient pointer is NULL");
    if (d->M == 2) TMF_REQUIRE_PTR(d_pet_tok);
    if (d->HD > 0) {
        TMF_REQUIRE(p->d0_w && p->d0_b && p->dbn_g && p->dbn_b && p->d3_w && p->d3_b, TMF_E_NULL,
                    "tmf_heads_cnn_bwd: a discriminator parameter pointer is NULL");
        TMF_REQUIRE(g->d0_w && g->d0_b && g->dbn_g && g->dbn_b && g->d3_w && g->d3_b, TMF_E_NULL,
                    "tmf_heads_cnn_bwd: a discriminator gradient pointer is NULL");
        TMF_REQUIRE_PTR(d_d_mri_logits); TMF_REQUIRE_PTR(d_d_pet_logits);
    }
    TMF_REQUIRE(saved_bytes >= tmf_heads_cnn_saved_bytes(d), TMF_E_WORKSPACE, "tmf_heads_cnn_bwd: saved %zu B < required %zu B",
                saved_bytes, tmf_heads_cnn_saved_bytes(d));
    TMF_REQUIRE(scratch_bytes >= tmf_heads_cnn_bwd_scratch_bytes(d), TMF_E_WORKSPACE,
                "tmf_heads_cnn_bwd: scratch %zu B < required %zu B", scratch_bytes, tmf_heads_cnn_bwd_scratch_bytes(d));
    TMF_REQUIRE_ALIGNED(d_mri_tok); TMF_REQUIRE_ALIGNED(scratch);
    if (d->M == 2) TMF_REQUIRE_ALIGNED(d_pet_tok);
    HeadsBwdArgs a = {};
    a.f = make_cnn_args(*d, nullptr, nullptr, *p, (float*)const_cast<void*>(saved));
    a.d_logits = d_logits; a.d_dlog[0] = d_d_mri_logits; a.d_dlog[1] = d_d_pet_logits;
    a.gw0 = g->fc0_w; a.gb0 = g->fc0_b; a.gw8 = g->fc2_w; a.gb8 = g->fc2_b;
    a.gdw0 = g->d0_w; a.gdb0 = g->d0_b; a.gdg1 = g->dbn_g; a.gdbe1 = g->dbn_b; a.gdw3 = g->d3_w; a.gdb3 = g->d3_b;
    a.d_tok[0] = d_mri_tok; a.d_tok[1] = d_pet_tok; a.alpha = revgrad_alpha;
    a.s_dz1 = (float*)scratch; a.s_dv = a.s_dz1 + (size_t)d->B * d->M * d->dim;
    const size_t lds = cnn_bwd_lds(*d);
    TMF_REQUIRE(lds <= 150 * 1024, TMF_E_SHAPE, "tmf_heads_cnn_bwd: %zu B of LDS needed (batch x widths too large)", lds);
    if ((rc = tmf_allow_lds(HEADS_K(cnn_heads_bwd_kernel, d->B), lds, "tmf_heads_cnn_bwd"))) return rc;
    hipLaunchKernelGGL(HEADS_K(cnn_heads_bwd_kernel, d->B), dim3(d->HD > 0 ? 1 + ND : 1), dim3(HT), lds, (hipStream_t)stream, a);
    if ((rc = tmf_launch_result("tmf_heads_cnn_bwd"))) return rc;
    int nbt = tmf_cdiv((long)d->M * d->B * d->N * d->dim, (long)HT * 4 * 4);
    if (nbt > 256) nbt = 256;
    if (nbt < 1) nbt = 1;
    hipLaunchKernelGGL(cnn_heads_bwd_tok_kernel, dim3(nbt), dim3(HT), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_heads_cnn_bwd(tokens)");
}
