// adam.hip — the optimizer step of the reference's train_step as ONE launch.
//
// reference: kfold_train_adversarial.py:135 (`optimizer.step()`), utils/utils.py:38-39 (getOptimizer: torch.optim.Adam,
// lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight decay 0) over the 156 parameter tensors of model_ad (4.17 M floats).
// torch's multi-tensor Adam needs ~10 launches for them (its per-launch tensor table is a kernel argument of limited
// size); here every tensor's pointers ride in ONE kernel-argument table (the two moment estimates live in two flat
// buffers owned by the optimizer, so a tensor costs 8 + 8 + 4 + 4 + 4 = 28 bytes of table) and a workgroup finds its tensor by a binary
// search over the chunk prefix sums.  SURVEY.md §8f rank 2.
//
// Update (torch.optim.Adam, amsgrad = False, maximize = False; weight_decay is L2 as in torch):
//   g' = g + wd * p;  m = m + (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "tmf_common.h"

namespace {

constexpr int CHUNK = 2048;                    // elements per workgroup: 256 threads x 2 float4

struct AdamTable {
    float* p[TMF_ADAM_MAX_TENSORS];
    const float* g[TMF_ADAM_MAX_TENSORS];
    int off[TMF_ADAM_MAX_TENSORS];             // element offset of the tensor's moments in the flat buffers
    int first[TMF_ADAM_MAX_TENSORS + 1];       // first chunk of tensor i (prefix sums); first[n] = number of chunks
    int numel[TMF_ADAM_MAX_TENSORS];
    int n;
};
static_assert(sizeof(AdamTable) <= 6144, "the table is a kernel argument (AMD kernarg segments are not limited to 4 KB)");

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTable t, float* __restrict__ m_, float* __restrict__ v_,
                                                        float step_size, float inv_sqrt_bc2, float omb1, float b2, float omb2,
                                                        float eps, float wd) {
    const int chunk = blockIdx.x;
    int lo = 0, hi = t.n - 1;                  // the tensor whose chunk range contains `chunk`
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.first[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    const int ti = lo;
    const int base = (chunk - t.first[ti]) * CHUNK;
    const int n = t.numel[ti];
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = m_ + t.off[ti];
    float* __restrict__ v = v_ + t.off[ti];
    auto upd = [&](float& pe, float ge, float& me, float& ve) {
        ge += wd * pe;
        me += omb1 * (ge - me);
        ve = b2 * ve + omb2 * ge * ge;
        pe -= step_size * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
    };
    // 16-byte accesses where the tensor allows (storage offsets of views are only 4-byte aligned in general)
    const bool vec = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
#pragma unroll
    for (int it = 0; it < CHUNK / 1024; ++it) {
        const int e = base + it * 1024 + threadIdx.x * 4;
        if (e >= n) break;
        if (vec && e + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + e), mv = *reinterpret_cast<f32x4*>(m + e), vv = *reinterpret_cast<f32x4*>(v + e);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pe = pv[j], me = mv[j], ve = vv[j];
                upd(pe, gv[j], me, ve);
                pv[j] = pe; mv[j] = me; vv[j] = ve;
            }
            *reinterpret_cast<f32x4*>(p + e) = pv; *reinterpret_cast<f32x4*>(m + e) = mv; *reinterpret_cast<f32x4*>(v + e) = vv;
        } else {
            for (int j = 0; j < 4 && e + j < n; ++j) upd(p[e + j], g[e + j], m[e + j], v[e + j]);
        }
    }
}

}  // namespace

extern "C" int tmf_adam_step(int n, float* const* params, const float* const* grads, const long* numel,
                             float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int step, void* stream) {
    TMF_REQUIRE_PTR(params); TMF_REQUIRE_PTR(grads); TMF_REQUIRE_PTR(numel); TMF_REQUIRE_PTR(exp_avg); TMF_REQUIRE_PTR(exp_avg_sq);
    TMF_REQUIRE(n > 0 && n <= TMF_ADAM_MAX_TENSORS, TMF_E_SHAPE, "tmf_adam_step: %d tensors (1 .. %d per call)", n,
                TMF_ADAM_MAX_TENSORS);
    TMF_REQUIRE(step >= 1, TMF_E_ARG, "tmf_adam_step: step=%d (the count INCLUDING this update, from 1)", step);
    TMF_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0.,
                TMF_E_ARG, "tmf_adam_step: lr=%g betas=(%g, %g) eps=%g weight_decay=%g", lr, beta1, beta2, eps, weight_decay);
    TMF_REQUIRE_ALIGNED(exp_avg); TMF_REQUIRE_ALIGNED(exp_avg_sq);
    AdamTable t;
    long off = 0;
    int chunks = 0, k = 0;
    for (int i = 0; i < n; ++i) {
        TMF_REQUIRE(numel[i] >= 0 && numel[i] < (1L << 31), TMF_E_SHAPE, "tmf_adam_step: tensor %d has %ld elements", i, numel[i]);
        if (grads[i] != nullptr && numel[i] > 0) {             // a parameter without a gradient is skipped, as in torch
            TMF_REQUIRE(params[i] != nullptr, TMF_E_NULL, "tmf_adam_step: parameter %d is NULL", i);
            t.p[k] = params[i]; t.g[k] = grads[i]; t.off[k] = (int)off; t.numel[k] = (int)numel[i]; t.first[k] = chunks;
            chunks += (int)((numel[i] + CHUNK - 1) / CHUNK);
            ++k;
        }
        off += (numel[i] + 3) & ~3L;                           // every tensor's moments start 16-byte aligned
        TMF_REQUIRE(off < (1L << 31), TMF_E_SHAPE, "tmf_adam_step: more than 2^31 moment elements");
    }
    if (k == 0) return TMF_OK;
    t.first[k] = chunks;
    t.n = k;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adam_step_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, t, exp_avg, exp_avg_sq,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)eps, (float)weight_decay);      // hyper-parameters arrive as doubles: 1 - beta is formed as torch forms it
    return tmf_launch_result("tmf_adam_step");
}

// elements of the flat moment buffers tmf_adam_step addresses for these tensors (each tensor padded to a multiple of 4)
extern "C" long tmf_adam_state_elems(int n, const long* numel) {
    if (n <= 0 || numel == nullptr) return 0;
    long off = 0;
    for (int i = 0; i < n; ++i) off += (numel[i] + 3) & ~3L;
    return off;
}
