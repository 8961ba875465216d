/*
 * tmf_oracle.c — plain-C restatement of the hot path's primitives (TEST INFRASTRUCTURE).
 *
 * Naive loops, fp32 storage, fp64 accumulation, reference tensor layout (NCDHW / row-major).
 * It exists so that the torch-functional oracle (oracle/tmf_oracle.py) is itself cross-checked by an
 * implementation that shares no library code with it (tests/test_oracle_c.py).  Nothing in the product
 * package links or loads this file.  Parity status: pinned transitively — tmf_oracle.py is pinned to the
 * reference's golden vectors (tests/test_oracle_golden.py) and this file is pinned to tmf_oracle.py.
 *
 * Each function cites the reference call site whose ATen op it restates
 * (/root/reference/models/networks.py unless stated).
 */
#include <math.h>
#include <stddef.h>

/* F.conv3d, stride 1, zero padding k/2, cross-correlation, with bias (networks.py:22,28,31,37,40,46,49). */
void orc_conv3d(const float* x, const float* w, const float* bias, float* y,
                int B, int Cin, int Cout, int D, int H, int W, int k) {
    const int p = k / 2;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int ww = 0; ww < W; ++ww) {
                        double acc = bias ? bias[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int kd = 0; kd < k; ++kd) {
                                const int zd = d + kd - p;
                                if (zd < 0 || zd >= D) continue;
                                for (int kh = 0; kh < k; ++kh) {
                                    const int zh = h + kh - p;
                                    if (zh < 0 || zh >= H) continue;
                                    for (int kw = 0; kw < k; ++kw) {
                                        const int zw = ww + kw - p;
                                        if (zw < 0 || zw >= W) continue;
                                        acc += (double)x[(((size_t)(b * Cin + ci) * D + zd) * H + zh) * W + zw] *
                                               (double)w[((((size_t)co * Cin + ci) * k + kd) * k + kh) * k + kw];
                                    }
                                }
                            }
                        y[(((size_t)(b * Cout + co) * D + d) * H + h) * W + ww] = (float)acc;
                    }
}

/* BatchNorm (train): batch statistics over (B, spatial), biased variance for normalisation, unbiased into
 * running_var, momentum update (networks.py:23,...; torch.nn.BatchNorm3d defaults). In place on x. */
void orc_batchnorm_train(float* x, const float* gamma, const float* beta, float* rmean, float* rvar,
                         int B, int C, long S, float momentum, float eps) {
    const double n = (double)B * (double)S;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int b = 0; b < B; ++b)
            for (long i = 0; i < S; ++i) s += x[((size_t)b * C + c) * S + i];
        const double m = s / n;
        double v = 0.0;
        for (int b = 0; b < B; ++b)
            for (long i = 0; i < S; ++i) { const double d = x[((size_t)b * C + c) * S + i] - m; v += d * d; }
        v /= n;
        const double is = 1.0 / sqrt(v + (double)eps);
        for (int b = 0; b < B; ++b)
            for (long i = 0; i < S; ++i) {
                float* q = &x[((size_t)b * C + c) * S + i];
                *q = (float)(((double)*q - m) * is * gamma[c] + beta[c]);
            }
        if (rmean) rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
        if (rvar) rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * (n > 1 ? v * n / (n - 1) : v));
    }
}

/* LeakyReLU(0.01) (networks.py:24). */
void orc_leaky_relu(float* x, long n, float slope) {
    for (long i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

/* MaxPool3d(2,2) / AvgPool3d(2,2), floor mode (networks.py:25,34,43,52).  mode 1 = max, 2 = avg. */
void orc_pool2(const float* x, float* y, int BC, int D, int H, int W, int mode) {
    const int OD = D / 2, OH = H / 2, OW = W / 2;
    for (int c = 0; c < BC; ++c)
        for (int d = 0; d < OD; ++d)
            for (int h = 0; h < OH; ++h)
                for (int w = 0; w < OW; ++w) {
                    double acc = mode == 1 ? -INFINITY : 0.0;
                    for (int k = 0; k < 8; ++k) {
                        const float v = x[(((size_t)c * D + 2 * d + (k >> 2)) * H + 2 * h + ((k >> 1) & 1)) * W + 2 * w + (k & 1)];
                        if (mode == 1) { if (v > acc) acc = v; } else acc += v;
                    }
                    y[(((size_t)c * OD + d) * OH + h) * OW + w] = (float)(mode == 1 ? acc : acc / 8.0);
                }
}

/* LayerNorm over the last dim, eps 1e-5 (networks.py:117,219). */
void orc_layernorm(const float* x, const float* g, const float* b, float* y, long rows, int dim, float eps) {
    for (long r = 0; r < rows; ++r) {
        double m = 0.0, v = 0.0;
        for (int i = 0; i < dim; ++i) m += x[r * dim + i];
        m /= dim;
        for (int i = 0; i < dim; ++i) { const double d = x[r * dim + i] - m; v += d * d; }
        const double is = 1.0 / sqrt(v / dim + (double)eps);
        for (int i = 0; i < dim; ++i) y[r * dim + i] = (float)((x[r * dim + i] - m) * is * g[i] + b[i]);
    }
}

/* y = x W^T + b (nn.Linear; networks.py:129,132,149,150,153). */
void orc_linear(const float* x, const float* w, const float* bias, float* y, long rows, int cin, int cout) {
    for (long r = 0; r < rows; ++r)
        for (int o = 0; o < cout; ++o) {
            double acc = bias ? bias[o] : 0.0;
            for (int i = 0; i < cin; ++i) acc += (double)x[r * cin + i] * (double)w[(size_t)o * cin + i];
            y[r * cout + o] = (float)acc;
        }
}

/* exact (erf) GELU (networks.py:130). */
void orc_gelu(float* x, long n) {
    for (long i = 0; i < n; ++i) x[i] = (float)(0.5 * x[i] * (1.0 + erf(x[i] / sqrt(2.0))));
}

/* Multi-head attention core (networks.py:166-174): q (B,N,h*dh), k/v (B,M,h*dh) -> out (B,N,h*dh),
 * softmax over keys of q.k * scale. */
void orc_attention(const float* q, const float* k, const float* v, float* out,
                   int B, int heads, int N, int M, int dh, float scale, double* scratch /* M doubles */) {
    const int inner = heads * dh;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < heads; ++h)
            for (int i = 0; i < N; ++i) {
                const float* qi = q + ((size_t)b * N + i) * inner + h * dh;
                double mx = -INFINITY;
                for (int j = 0; j < M; ++j) {
                    const float* kj = k + ((size_t)b * M + j) * inner + h * dh;
                    double s = 0.0;
                    for (int d = 0; d < dh; ++d) s += (double)qi[d] * (double)kj[d];
                    scratch[j] = s * scale;
                    if (scratch[j] > mx) mx = scratch[j];
                }
                double den = 0.0;
                for (int j = 0; j < M; ++j) { scratch[j] = exp(scratch[j] - mx); den += scratch[j]; }
                for (int d = 0; d < dh; ++d) {
                    double acc = 0.0;
                    for (int j = 0; j < M; ++j) acc += scratch[j] * (double)v[((size_t)b * M + j) * inner + h * dh + d];
                    out[((size_t)b * N + i) * inner + h * dh + d] = (float)(acc / den);
                }
            }
}
