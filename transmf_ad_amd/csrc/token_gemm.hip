// token_gemm.hip — the Linear layers of the fusion transformer, fused with what surrounds them.  gfx950.
//
// The token side of model_ad is tiny (B*216 = 1728 rows of 128 floats per modality) but long: per
// Transformer(dim, depth=1) the reference issues LayerNorm -> to_q, to_kv, attention, to_out + bias, + x,
// LayerNorm -> Linear + bias -> GELU -> Linear + bias, + x, LayerNorm, + tokens (networks.py:117-121, 152-175,
// 219-230, 262-263) — 13 kernels forward and ~40 backward when each op is its own launch, all of them 3-15 us
// and serial.  Here every Linear is ONE launch that also does its neighbours' element-wise work:
//
//   forward   y  = [GELU]( LayerNorm?(x) . W^T + bias ) + residual        tok_gemm_kernel<false, LN, EPI>
//   backward  dx = LayerNormBackward?( (dy . W) [* GELU'(h)] ) + add1 + add2   tok_gemm_kernel<true, false, EPI>
//             + the bias-gradient column sums of dy and the LayerNorm dgamma / dbeta block partials
//
// Tile: 16 token rows x 128 output columns per workgroup of 4 wavefronts (2 MFMA column tiles each),
// v_mfma_f32_16x16x4_f32 (exact fp32).  The A tile (after the LayerNorm prologue) sits in LDS and is read
// with one ds_read_b128 per 4 MFMAs through a K-permutation (lane group kb takes k = 16 s + 4 kb + j); the
// weights come straight from L2 in the matching order (one 16-B load per 4 MFMAs in the W^T form).
// Weight gradients stay plain GEMMs (dy^T . saved activations).
#include "tmf_common.h"

namespace {

constexpr int TM = 16, TN = 128, TTHR = 256;

struct TokArgs {
    const float* A; const float* W; float* Y;
    int R, K, N;                                   // A [R][K], Y [R][N]; W is [N][K] (forward) or [K][N] (backward)
    // LayerNorm prologue (forward)
    const float* ln_g; const float* ln_b; float eps; float* ln_mean; float* ln_rstd; float* ln_out;
    // forward epilogue
    const float* bias; const float* res; float* pre;
    // backward epilogues
    const float* gelu_h;
    const float* lnb_x; const float* lnb_mean; const float* lnb_rstd; const float* lnb_g;
    const float* add1; const float* add2;
    float* lnb_partial; float* colsum_partial; int partial_stride;
};

__device__ __forceinline__ float gelu_f(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float h) {
    return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}
__device__ __forceinline__ float half_sum(float v) {         // sum over the 32 lanes of a half-wave
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

enum { EPI_PLAIN = 0, EPI_GELU = 1, EPI_GELU_GRAD = 2, EPI_LN_BWD = 3 };

template <bool NN, bool LN, int EPI>
__global__ __launch_bounds__(TTHR, 2) void tok_gemm_kernel(const TokArgs p) {   // <= 256 registers: MFMA results stay in VGPRs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KP = p.K + 4;
    float* As = smem;                              // [16][K + 4]
    float* red = smem + TM * KP;                   // [4 waves][16 rows][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4;
    const int r0 = blockIdx.x * TM;
    const int c0 = blockIdx.y * TN + wave * 32;

    // ---- A tile -> LDS ----
    if (LN) {                                      // K == 128: a half-wave owns one row
        const int li = lane & 31;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = wave * 4 + pass * 2 + (lane >> 5);
            const int gr = r0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < p.R) v = *reinterpret_cast<const f32x4*>(p.A + (size_t)gr * p.K + li * 4);
            const float mu = half_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 128.f);
            const f32x4 d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
            const float var = half_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 128.f);
            const float rs = 1.f / sqrtf(var + p.eps);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_g + li * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + li * 4);
            f32x4 a = {d[0] * rs * g[0] + b[0], d[1] * rs * g[1] + b[1], d[2] * rs * g[2] + b[2], d[3] * rs * g[3] + b[3]};
            if (gr >= p.R) a = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(As + row * KP + li * 4) = a;
            if (blockIdx.y == 0 && gr < p.R) {
                if (p.ln_out) *reinterpret_cast<f32x4*>(p.ln_out + (size_t)gr * p.K + li * 4) = a;
                if (li == 0) { p.ln_mean[gr] = mu; p.ln_rstd[gr] = rs; }
            }
        }
    } else {
        const int k4 = p.K >> 2;
        for (int e = tid; e < TM * k4; e += TTHR) {
            const int row = e / k4, c4 = e - row * k4;
            const int gr = r0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < p.R) v = *reinterpret_cast<const f32x4*>(p.A + (size_t)gr * p.K + c4 * 4);
            *reinterpret_cast<f32x4*>(As + row * KP + c4 * 4) = v;
        }
    }
    __syncthreads();

    if (NN && p.colsum_partial != nullptr && blockIdx.y == 0) {      // bias gradient of the layer that made dy
        for (int c = tid; c < p.K; c += TTHR) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < TM; ++r) s += As[r * KP + c];
            p.colsum_partial[(size_t)blockIdx.x * p.partial_stride + c] = s;
        }
    }

    // ---- MFMA: acc[t] = A[16 x K] . B[K x 16] for this wave's two column tiles ----
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* arow = As + m * KP + 4 * kb;
    const int nsteps = p.K >> 4;
    if (!NN) {
        const float* w0 = p.W + (size_t)(c0 + m) * p.K + 4 * kb;
        const float* w1 = w0 + (size_t)16 * p.K;
#pragma unroll 4
        for (int s = 0; s < nsteps; ++s) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * s);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(w0 + 16 * s);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(w1 + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b1[j], acc[1], 0, 0, 0);
            }
        }
    } else {
        const float* w0 = p.W + (size_t)(4 * kb) * p.N + c0 + m;
#pragma unroll 4
        for (int s = 0; s < nsteps; ++s) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * s);
            float b0[4], b1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b0[j] = w0[(size_t)(16 * s + j) * p.N];
                b1[j] = w0[(size_t)(16 * s + j) * p.N + 16];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b1[j], acc[1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  D fragment: column = lane & 15, row = 4 * (lane >> 4) + r ----
    if (EPI == EPI_PLAIN || EPI == EPI_GELU || EPI == EPI_GELU_GRAD) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = c0 + t * 16 + m;
            const float bv = (EPI != EPI_GELU_GRAD && p.bias != nullptr) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = r0 + 4 * kb + r;
                if (gr < p.R) {
                    const size_t o = (size_t)gr * p.N + col;
                    float v = acc[t][r] + bv;
                    if (EPI == EPI_PLAIN) {
                        if (p.res != nullptr) v += p.res[o];
                    } else if (EPI == EPI_GELU) {
                        p.pre[o] = v;
                        v = gelu_f(v);
                    } else {
                        v *= gelu_grad_f(p.gelu_h[o]);
                    }
                    p.Y[o] = v;
                }
            }
        }
    } else {                                        // LayerNorm backward over the 128 output columns (gridDim.y == 1)
        float g[2][4], xh[2][4], s1[4], s2[4], pg[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
        float rsr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = r0 + 4 * kb + r;
            const bool ok = gr < p.R;
            const float mu = ok ? p.lnb_mean[gr] : 0.f;
            rsr[r] = ok ? p.lnb_rstd[gr] : 0.f;
            s1[r] = 0.f; s2[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = c0 + t * 16 + m;
                const float xv = ok ? p.lnb_x[(size_t)gr * 128 + col] : 0.f;
                xh[t][r] = (xv - mu) * rsr[r];
                g[t][r] = acc[t][r] * p.lnb_g[col];
                s1[r] += g[t][r];
                s2[r] += g[t][r] * xh[t][r];
                pg[t] += acc[t][r] * xh[t][r];
                pb[t] += acc[t][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[r] += __shfl_xor(s1[r], 8); s2[r] += __shfl_xor(s2[r], 8);
            s1[r] += __shfl_xor(s1[r], 4); s2[r] += __shfl_xor(s2[r], 4);
            s1[r] += __shfl_xor(s1[r], 2); s2[r] += __shfl_xor(s2[r], 2);
            s1[r] += __shfl_xor(s1[r], 1); s2[r] += __shfl_xor(s2[r], 1);
            if (m == 0) { red[(wave * 16 + 4 * kb + r) * 2] = s1[r]; red[(wave * 16 + 4 * kb + r) * 2 + 1] = s2[r]; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pg[t] += __shfl_xor(pg[t], 16); pg[t] += __shfl_xor(pg[t], 32);
            pb[t] += __shfl_xor(pb[t], 16); pb[t] += __shfl_xor(pb[t], 32);
            if (kb == 0 && p.lnb_partial != nullptr) {
                const int col = c0 + t * 16 + m;
                p.lnb_partial[(size_t)blockIdx.x * p.partial_stride + col] = pg[t];
                p.lnb_partial[(size_t)blockIdx.x * p.partial_stride + 128 + col] = pb[t];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r, gr = r0 + row;
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { S1 += red[(w * 16 + row) * 2]; S2 += red[(w * 16 + row) * 2 + 1]; }
            S1 *= (1.f / 128.f); S2 *= (1.f / 128.f);
            if (gr < p.R) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const size_t o = (size_t)gr * 128 + c0 + t * 16 + m;
                    float v = rsr[r] * (g[t][r] - S1 - xh[t][r] * S2);
                    if (p.add1 != nullptr) v += p.add1[o];
                    if (p.add2 != nullptr) v += p.add2[o];
                    p.Y[o] = v;
                }
            }
        }
    }
}

template <bool NN, bool LN, int EPI>
int launch_tok(const TokArgs& a, hipStream_t s, const char* what) {
    const size_t lds = (size_t)(TM * (a.K + 4) + 4 * 16 * 2) * 4;
    hipLaunchKernelGGL((tok_gemm_kernel<NN, LN, EPI>), dim3(tmf_cdiv(a.R, TM), a.N / TN), dim3(TTHR), lds, s, a);
    return tmf_launch_result(what);
}

}  // namespace

extern "C" int tmf_tok_row_blocks(int R) { return R > 0 ? tmf_cdiv(R, TM) : 0; }

extern "C" int tmf_tok_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                  int R, int K, int Nout, const float* ln_gamma, const float* ln_beta, float eps,
                                  float* ln_mean, float* ln_rstd, float* ln_out, float* gelu_pre, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(y);
    TMF_REQUIRE(R > 0 && K > 0 && Nout > 0, TMF_E_SHAPE, "tmf_tok_linear_fwd: non-positive dimension");
    TMF_REQUIRE(K % 16 == 0 && K <= 2048 && Nout % TN == 0, TMF_E_SHAPE,
                "tmf_tok_linear_fwd: K=%d must be a multiple of 16 (<= 2048) and Nout=%d a multiple of 128", K, Nout);
    TokArgs a = {};
    a.A = x; a.W = w; a.Y = y; a.R = R; a.K = K; a.N = Nout;
    a.bias = bias; a.res = residual; a.pre = gelu_pre;
    hipStream_t s = (hipStream_t)stream;
    if (ln_gamma != nullptr) {
        TMF_REQUIRE(K == 128, TMF_E_SHAPE, "tmf_tok_linear_fwd: the LayerNorm prologue needs K == 128 (got %d)", K);
        TMF_REQUIRE_PTR(ln_beta); TMF_REQUIRE_PTR(ln_mean); TMF_REQUIRE_PTR(ln_rstd);
        a.ln_g = ln_gamma; a.ln_b = ln_beta; a.eps = eps; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_out = ln_out;
        if (gelu_pre != nullptr) {
            TMF_REQUIRE(residual == nullptr, TMF_E_SHAPE, "tmf_tok_linear_fwd: GELU epilogue takes no residual");
            return launch_tok<false, true, EPI_GELU>(a, s, "tmf_tok_linear_fwd(ln,gelu)");
        }
        return launch_tok<false, true, EPI_PLAIN>(a, s, "tmf_tok_linear_fwd(ln)");
    }
    if (gelu_pre != nullptr) {
        TMF_REQUIRE(residual == nullptr, TMF_E_SHAPE, "tmf_tok_linear_fwd: GELU epilogue takes no residual");
        return launch_tok<false, false, EPI_GELU>(a, s, "tmf_tok_linear_fwd(gelu)");
    }
    return launch_tok<false, false, EPI_PLAIN>(a, s, "tmf_tok_linear_fwd");
}

extern "C" int tmf_tok_linear_bwd_input(const float* dy, const float* w, float* dx, int R, int Nout, int K,
                                        const float* gelu_pre, const float* ln_x, const float* ln_mean,
                                        const float* ln_rstd, const float* ln_gamma, const float* add1,
                                        const float* add2, float* ln_partial, float* bias_partial,
                                        int partial_stride, void* stream) {
    TMF_REQUIRE_PTR(dy); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(dx);
    TMF_REQUIRE(R > 0 && K > 0 && Nout > 0, TMF_E_SHAPE, "tmf_tok_linear_bwd_input: non-positive dimension");
    TMF_REQUIRE(Nout % 16 == 0 && Nout <= 2048 && K % TN == 0, TMF_E_SHAPE,
                "tmf_tok_linear_bwd_input: Nout=%d must be a multiple of 16 (<= 2048) and K=%d a multiple of 128", Nout, K);
    TMF_REQUIRE((ln_partial == nullptr && bias_partial == nullptr) || partial_stride > 0, TMF_E_SHAPE,
                "tmf_tok_linear_bwd_input: partial_stride must be positive");
    TokArgs a = {};
    a.A = dy; a.W = w; a.Y = dx; a.R = R; a.K = Nout; a.N = K;       // contraction over dy's columns
    a.gelu_h = gelu_pre; a.add1 = add1; a.add2 = add2;
    a.lnb_partial = ln_partial; a.colsum_partial = bias_partial; a.partial_stride = partial_stride;
    hipStream_t s = (hipStream_t)stream;
    if (ln_x != nullptr) {
        TMF_REQUIRE(K == 128 && gelu_pre == nullptr, TMF_E_SHAPE,
                    "tmf_tok_linear_bwd_input: the LayerNorm-backward epilogue needs K == 128 and no GELU (K=%d)", K);
        TMF_REQUIRE_PTR(ln_mean); TMF_REQUIRE_PTR(ln_rstd); TMF_REQUIRE_PTR(ln_gamma);
        a.lnb_x = ln_x; a.lnb_mean = ln_mean; a.lnb_rstd = ln_rstd; a.lnb_g = ln_gamma;
        return launch_tok<true, false, EPI_LN_BWD>(a, s, "tmf_tok_linear_bwd_input(ln)");
    }
    if (gelu_pre != nullptr) return launch_tok<true, false, EPI_GELU_GRAD>(a, s, "tmf_tok_linear_bwd_input(gelu)");
    a.res = add1;                                                    // plain epilogue: + add1 (one residual gradient)
    TMF_REQUIRE(add2 == nullptr, TMF_E_SHAPE, "tmf_tok_linear_bwd_input: add2 needs the LayerNorm epilogue");
    return launch_tok<true, false, EPI_PLAIN>(a, s, "tmf_tok_linear_bwd_input");
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradients of the block's Linears, all in one launch:  dW_p[n][k] = sum_r dy_p[r][n] * x_p[r][k]  for up to
// 32 problems p (the five of a Transformer block: to_q, to_kv, to_out, the two FeedForward layers).  One workgroup =
// one 32 x 32 tile of one dW and one of NSPLIT row ranges; its 4 waves take a quarter of the range each (both
// operands straight from L2: a lane reads dy[row][n0 + lane&31] / x[row][k0 + lane&31], rows 2 s + lane>>5 — coalesced
// 128-B segments, no LDS staging), v_mfma_f32_32x32x2_f32, cross-wave sum in LDS, partial tile to the workspace; a
// second small kernel adds the NSPLIT partials in fixed order (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int WG_MAXP = 32, WG_NSPLIT = 8;
struct WgMulti {
    const float* dy[WG_MAXP]; const float* x[WG_MAXP]; float* dw[WG_MAXP];
    int R[WG_MAXP], N[WG_MAXP], K[WG_MAXP], tile0[WG_MAXP + 1], elem0[WG_MAXP + 1];
    int nprob;
};

__global__ __launch_bounds__(256, 2) void tok_wgrad_multi_kernel(const WgMulti p, float* __restrict__ partial, long total_elems) {
    __shared__ float red[3][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    int q = 0;
    while (q + 1 < p.nprob && (int)blockIdx.x >= p.tile0[q + 1]) ++q;
    const int t = blockIdx.x - p.tile0[q];
    const int tiles_k = p.K[q] >> 5;
    const int n0 = (t / tiles_k) * 32, k0 = (t % tiles_k) * 32;
    const int R = p.R[q], N = p.N[q], K = p.K[q];
    int rows = (R + WG_NSPLIT - 1) / WG_NSPLIT;
    rows = (rows + 7) & ~7;                                  // 4 waves x pairs of rows
    const int rb = blockIdx.y * rows + wave * (rows >> 2);
    const float* dy = p.dy[q] + n0 + l31;
    const float* xx = p.x[q] + k0 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int steps = rows >> 3;                             // row pairs per wave
    // Operands straight from L2, one dword per lane and MFMA: the loop is a chain of load latencies, not of matrix work
    // (a workgroup took ~12 us for 1.7 us of MFMA with 4 steps in flight).  16 steps are requested at once — rows past
    // the range are clamped to a valid row and multiplied by zero, so the batch has no branches around its loads.
    constexpr int UB = 16;
    for (int s0 = 0; s0 < steps; s0 += UB) {
        float a[UB], b[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int row = rb + 2 * (s0 + u) + hsel;
            const bool ok = (s0 + u) < steps && row < R;
            const int rr = ok ? row : 0;
            a[u] = dy[(size_t)rr * N];
            b[u] = xx[(size_t)rr * K];
            a[u] = ok ? a[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    // D[i = n][j = k]: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 hsel
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][((r & 3) + 8 * (r >> 2) + 4 * hsel) * 32 + l31] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = partial + (size_t)blockIdx.y * total_elems + p.elem0[q];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hsel;
            const float v = acc[r] + red[0][i * 32 + l31] + red[1][i * 32 + l31] + red[2][i * 32 + l31];
            dst[(size_t)(n0 + i) * K + k0 + l31] = v;
        }
    }
}

__global__ __launch_bounds__(256) void tok_wgrad_multi_reduce_kernel(const WgMulti p, const float* __restrict__ partial,
                                                                     long total_elems) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total_elems) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < WG_NSPLIT; ++k) s += partial[(size_t)k * total_elems + e];
    int q = 0;
    while (q + 1 < p.nprob && e >= p.elem0[q + 1]) ++q;
    p.dw[q][e - p.elem0[q]] = s;
}

int fill_multi(WgMulti& m, int nprob, const float* const* dy, const float* const* x, float* const* dw,
               const int* R, const int* N, const int* K, const char* fn) {
    TMF_REQUIRE(nprob > 0 && nprob <= WG_MAXP, TMF_E_SHAPE, "%s: 1..%d problems, got %d", fn, WG_MAXP, nprob);
    m.nprob = nprob;
    m.tile0[0] = 0; m.elem0[0] = 0;
    for (int q = 0; q < nprob; ++q) {
        TMF_REQUIRE(R[q] > 0 && N[q] > 0 && K[q] > 0 && N[q] % 32 == 0 && K[q] % 32 == 0, TMF_E_SHAPE,
                    "%s: problem %d: R=%d N=%d K=%d (N and K must be multiples of 32)", fn, q, R[q], N[q], K[q]);
        m.R[q] = R[q]; m.N[q] = N[q]; m.K[q] = K[q];
        if (dy) { m.dy[q] = dy[q]; m.x[q] = x[q]; m.dw[q] = dw[q]; }
        m.tile0[q + 1] = m.tile0[q] + (N[q] / 32) * (K[q] / 32);
        m.elem0[q + 1] = m.elem0[q] + N[q] * K[q];
    }
    return TMF_OK;
}

}  // namespace

extern "C" size_t tmf_tok_wgrad_multi_workspace_bytes(int nprob, const int* N, const int* K) {
    if (nprob <= 0 || nprob > WG_MAXP || N == nullptr || K == nullptr) return 0;
    size_t e = 0;
    for (int q = 0; q < nprob; ++q) e += (size_t)N[q] * K[q];
    return e * WG_NSPLIT * 4;
}

extern "C" int tmf_tok_wgrad_multi(int nprob, const float* const* dy, const float* const* x, float* const* dw,
                                   const int* R, const int* N, const int* K, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    TMF_REQUIRE_PTR(dy); TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(R); TMF_REQUIRE_PTR(N); TMF_REQUIRE_PTR(K);
    TMF_REQUIRE_PTR(workspace);
    WgMulti m = {};
    int rc = fill_multi(m, nprob, dy, x, dw, R, N, K, "tmf_tok_wgrad_multi");
    if (rc) return rc;
    for (int q = 0; q < nprob; ++q) { TMF_REQUIRE_PTR(dy[q]); TMF_REQUIRE_PTR(x[q]); TMF_REQUIRE_PTR(dw[q]); }
    const size_t need = tmf_tok_wgrad_multi_workspace_bytes(nprob, N, K);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_tok_wgrad_multi: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const long total = m.elem0[nprob];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tok_wgrad_multi_kernel, dim3(m.tile0[nprob], WG_NSPLIT), dim3(256), 0, s, m, (float*)workspace, total);
    if ((rc = tmf_launch_result("tmf_tok_wgrad_multi"))) return rc;
    hipLaunchKernelGGL(tok_wgrad_multi_reduce_kernel, dim3((unsigned)tmf_cdiv(total, 256L)), dim3(256), 0, s, m,
                       (const float*)workspace, total);
    return tmf_launch_result("tmf_tok_wgrad_multi(reduce)");
}
