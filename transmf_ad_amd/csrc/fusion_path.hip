// Whole-fusion entries: ONE host call enqueues every launch of CrossTransformer_MOD_AVG's train-mode forward (or of its
// backward).
//
// reference: models/networks.py:255-281 (CrossTransformer_MOD_AVG: depth x [mri <- Transformer(mri | pet) + mri;
// pet <- Transformer(pet | NEW mri) + pet], then cat[mean, mean, max, max] over the tokens), built from
// Transformer(depth=1) (:215-230), PreNorm (:114-121), Attention (:141-175) and FeedForward (:125-137).
//
// Why: the six Transformer instances are 6 x (7 forward + ~13 backward) launches of 5-25 us each.  Issued from Python
// (one autograd node, a dozen ctypes calls and ~25 small tensor allocations per instance and direction) the host needs
// longer per launch than the GPU — and this is the part of backward that runs FIRST, right after the reference step's
// two loss.item() syncs have drained the queue (kfold_train_adversarial.py:127-128), so the GPU idles behind the host.
// Here a pass is one call; the per-launch entries are the library's own (token_gemm.hip, attention.hip, token_ops.hip).
//
// Pure host code.
#include "tmf_common.h"

// fused per-instance kernels (xformer_fused.hip)
struct tmf_xf_fwd_io {
    const float *x, *KR, *VC, *pk, *pkv_next, *mask_o, *mask_g, *mask_f;
    float *a, *QR, *QC, *out, *lse, *x1, *f, *h, *g, *x2, *y, *m1, *r1, *m2, *r2, *mf, *rf, *KRn, *KCn, *VRn, *VCn;
};
struct tmf_xf_bwd_io {
    const float *dy, *x, *KR, *KC, *VR, *pk, *mask_o, *mask_g, *mask_f;
    const float *QR, *QC, *out, *lse, *x1, *h, *x2, *m1, *r1, *m2, *r2, *mf, *rf;
    float *dx2, *dh, *dx1, *dq, *DR, *DC, *delta, *dx, *part, *dkv, *dctx;
    const float* dctx_acc;
};
bool tmf_xf_supported(int N, int dim, int heads, int dim_head, int mlp);
int tmf_xf_npad(int N);
int tmf_xf_tiles(int N);
int tmf_xf_part_stride(void);
int tmf_xf_pack_floats(void);
int tmf_xf_pack_kv_offset(void);
int tmf_xf_launch_pack(int n_inst, const tmf_xformer_params* inst, float* const* pk_fwd, float* const* pk_bwd, hipStream_t s);
int tmf_xf_launch_fwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_fwd_io* io, float scale, int only_kv, int h2, hipStream_t s);
int tmf_xf_launch_bwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_bwd_io* io, float scale, int h2, hipStream_t s);
int tmf_xf_launch_colsum(int n_inst, const float* const* part, float* const* small, float* const* lnf, int nblk, hipStream_t s);

namespace {

inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

#define TMF_TRY(call) do { int rc__ = (call); if (rc__ != TMF_OK) return rc__; } while (0)

int check_desc(const char* fn, const tmf_fusion_desc* d) {
    TMF_REQUIRE_PTR(d);
    TMF_REQUIRE(d->B > 0 && d->N > 0 && d->depth >= 0 && d->depth <= TMF_FUSION_MAX_DEPTH, TMF_E_SHAPE,
                "%s: B=%d N=%d depth=%d (depth <= %d)", fn, d->B, d->N, d->depth, TMF_FUSION_MAX_DEPTH);
    TMF_REQUIRE(d->dim == 128 && d->heads > 0 && d->dim_head > 0 && (d->heads * d->dim_head) % 128 == 0 && d->mlp % 128 == 0,
                TMF_E_SHAPE, "%s: needs dim == 128 and heads*dim_head, mlp multiples of 128 (dim=%d inner=%d mlp=%d)", fn,
                d->dim, d->heads * d->dim_head, d->mlp);
    TMF_REQUIRE(d->dim_head == 8 || d->dim_head == 16 || d->dim_head == 32 || d->dim_head == 64, TMF_E_SHAPE,
                "%s: dim_head=%d must be 8, 16, 32 or 64", fn, d->dim_head);
    return TMF_OK;
}

// Saved tensors of ONE Transformer instance (floats), in this order inside the instance's slab.
struct InstPlan {
    size_t a, q, kv, out, lse, x1, f, h, g, x2, y, m1, r1, m2, r2, mf, rf, total;      // byte offsets
    size_t QR, QC, KR, KC, VR, VC, pkf, pkb;      // fused kernels: fragment-order panels, weight packs
};

constexpr int WG_CHUNK = 30;    // weight-gradient problems per tmf_tok_wgrad_multi launch (6 instances)

struct Plan {
    int R, inner, nblk, nblk_ln, stride;
    bool fused;                 // one launch per instance and direction (xformer_fused.hip)
    int npad, tiles;
    InstPlan I;                 // identical for every instance
    size_t off_arg, saved_bytes;
    // backward scratch (byte offsets)
    size_t s_dx2, s_dh, s_dx1, s_dout, s_dq, s_dkv, s_part, s_lnpart, s_G[4], s_ws, ws_bytes, scratch_bytes;
    // fused backward: s_dx2 .. s_dkv and s_part are the first of `2 * depth` per-instance regions of inst_stride bytes
    size_t s_DR, s_DC, s_delta, inst_stride;
};

Plan make_plan(const tmf_fusion_desc& d) {
    Plan p;
    p.R = d.B * d.N;
    p.inner = d.heads * d.dim_head;
    const size_t R = (size_t)p.R, dim = d.dim, inner = p.inner, mlp = d.mlp;
    size_t o = 0;
    auto take = [&](size_t floats) { const size_t at = o; o += up256(floats * 4); return at; };
    InstPlan& I = p.I;
    p.fused = !(d.flags & TMF_FUSION_PER_OP) && tmf_xf_supported(d.N, d.dim, d.heads, d.dim_head, d.mlp);
    p.npad = tmf_xf_npad(d.N);
    p.tiles = tmf_xf_tiles(d.N);
    I.a = take(R * dim); I.q = take(R * inner); I.kv = take(R * 2 * inner); I.out = take(R * inner);
    I.lse = take((size_t)d.B * d.heads * p.npad); I.x1 = take(R * dim); I.f = take(R * dim); I.h = take(R * mlp);
    I.g = take(R * mlp); I.x2 = take(R * dim); I.y = take(R * dim);
    I.m1 = take(R); I.r1 = take(R); I.m2 = take(R); I.r2 = take(R); I.mf = take(R); I.rf = take(R);
    I.QR = I.QC = I.KR = I.KC = I.VR = I.VC = I.pkf = I.pkb = o;
    if (p.fused) {
        const size_t panel = (size_t)d.B * p.npad * inner;
        I.QR = take(panel); I.QC = take(panel); I.KR = take(panel); I.KC = take(panel); I.VR = take(panel); I.VC = take(panel);
        I.pkf = take(tmf_xf_pack_floats()); I.pkb = take(tmf_xf_pack_floats());
    }
    I.total = o;
    p.off_arg = 2 * (size_t)d.depth * I.total;
    p.saved_bytes = p.off_arg + up256((size_t)d.B * 2 * dim * 4);        // int32 argmax of the max pools
    p.nblk = tmf_tok_row_blocks(p.R);
    p.nblk_ln = tmf_layernorm_bwd_blocks(p.R, d.dim);
    p.stride = 6 * d.dim + d.mlp;
    o = 0;
    const int N5[5] = {d.dim, d.mlp, d.dim, 2 * p.inner, p.inner}, K5[5] = {d.mlp, d.dim, p.inner, d.dim, d.dim};
    if (p.fused) {
        // per instance (kept until the one weight-gradient launch at the end): dx2, dh, dx1, dq, dkv, column-sum partials
        p.stride = tmf_xf_part_stride();
        p.s_dx2 = take(R * dim); p.s_dh = take(R * mlp); p.s_dx1 = take(R * dim); p.s_dq = take(R * inner);
        p.s_dkv = take(R * 2 * inner); p.s_part = take((size_t)d.B * p.tiles * p.stride);
        p.inst_stride = o;
        o = p.inst_stride * (size_t)(2 * d.depth > 0 ? 2 * d.depth : 1);
        p.s_dout = o;
        p.s_DR = take((size_t)d.B * p.npad * inner); p.s_DC = take((size_t)d.B * p.npad * inner);
        p.s_delta = take((size_t)d.B * d.heads * p.npad);
        p.s_lnpart = o;
        for (int i = 0; i < 4; ++i) p.s_G[i] = take(R * dim);
        int n_inst = 2 * d.depth;
        if (n_inst > WG_CHUNK / 5) n_inst = WG_CHUNK / 5;
        if (n_inst < 1) n_inst = 1;
        int Nc[WG_CHUNK], Kc[WG_CHUNK];
        for (int i = 0; i < n_inst; ++i)
            for (int j = 0; j < 5; ++j) { Nc[5 * i + j] = N5[j]; Kc[5 * i + j] = K5[j]; }
        p.ws_bytes = tmf_tok_wgrad_multi_workspace_bytes(5 * n_inst, Nc, Kc);
    } else {
        p.s_dx2 = take(R * dim); p.s_dh = take(R * mlp); p.s_dx1 = take(R * dim); p.s_dout = take(R * inner);
        p.s_dq = take(R * inner); p.s_dkv = take(R * 2 * inner);
        p.s_part = take((size_t)p.nblk * p.stride); p.s_lnpart = take((size_t)p.nblk_ln * 2 * dim);
        p.s_DR = p.s_DC = p.s_delta = p.inst_stride = 0;
        for (int i = 0; i < 4; ++i) p.s_G[i] = take(R * dim);
        p.ws_bytes = tmf_tok_wgrad_multi_workspace_bytes(5, N5, K5);
    }
    if (p.ws_bytes < 16) p.ws_bytes = 16;
    p.s_ws = take(p.ws_bytes / 4 + 1);
    p.scratch_bytes = o;
    return p;
}

inline float* F(char* base, size_t off) { return (float*)(base + off); }

}  // namespace

extern "C" size_t tmf_fusion_saved_bytes(const tmf_fusion_desc* d) {
    if (check_desc("tmf_fusion_saved_bytes", d) != TMF_OK) return 0;
    return make_plan(*d).saved_bytes;
}

// 1: the calls of this descriptor run the fused per-instance kernels (csrc/xformer_fused.hip); 0: one launch per op
extern "C" int tmf_fusion_uses_fused(const tmf_fusion_desc* d) {
    if (check_desc("tmf_fusion_uses_fused", d) != TMF_OK) return 0;
    return make_plan(*d).fused ? 1 : 0;
}

extern "C" size_t tmf_fusion_bwd_scratch_bytes(const tmf_fusion_desc* d) {
    if (check_desc("tmf_fusion_bwd_scratch_bytes", d) != TMF_OK) return 0;
    return make_plan(*d).scratch_bytes;
}

// One Transformer(depth=1) instance:  y = LayerNorm_f( FF(LN2(x1)) + x1 ) + x,   x1 = Attention(LN1(x), ctx) + x
static int instance_fwd(const tmf_fusion_desc& d, const Plan& p, const tmf_xformer_params& w, const float* x, const float* c,
                        char* sv, void* stream) {
    const int R = p.R, dim = d.dim, inner = p.inner, mlp = d.mlp;
    const InstPlan& I = p.I;
    const float scale = 1.0f / sqrtf((float)d.dim_head);
    TMF_TRY(tmf_tok_linear_fwd(x, w.wq, nullptr, nullptr, F(sv, I.q), R, dim, inner, w.ln1_g, w.ln1_b, w.eps1, F(sv, I.m1),
                               F(sv, I.r1), F(sv, I.a), nullptr, stream));
    TMF_TRY(tmf_tok_linear_fwd(c, w.wkv, nullptr, nullptr, F(sv, I.kv), R, dim, 2 * inner, nullptr, nullptr, 0.f, nullptr,
                               nullptr, nullptr, nullptr, stream));
    TMF_TRY(tmf_xattn_fwd(F(sv, I.q), F(sv, I.kv), F(sv, I.kv) + inner, F(sv, I.out), F(sv, I.lse), d.B, d.heads, d.N, d.N,
                          d.dim_head, inner, 2 * inner, scale, stream));
    TMF_TRY(tmf_tok_linear_fwd(F(sv, I.out), w.wo, w.bo, x, F(sv, I.x1), R, inner, dim, nullptr, nullptr, 0.f, nullptr, nullptr,
                               nullptr, nullptr, stream));
    TMF_TRY(tmf_tok_linear_fwd(F(sv, I.x1), w.w1, w.b1, nullptr, F(sv, I.g), R, dim, mlp, w.ln2_g, w.ln2_b, w.eps2, F(sv, I.m2),
                               F(sv, I.r2), F(sv, I.f), F(sv, I.h), stream));
    TMF_TRY(tmf_tok_linear_fwd(F(sv, I.g), w.w2, w.b2, F(sv, I.x1), F(sv, I.x2), R, mlp, dim, nullptr, nullptr, 0.f, nullptr,
                               nullptr, nullptr, nullptr, stream));
    // block-final LayerNorm with the caller's "+ tokens" (networks.py:274-275) folded into the same pass
    TMF_TRY(tmf_layernorm_fwd(F(sv, I.x2), w.lnf_g, w.lnf_b, x, F(sv, I.y), F(sv, I.mf), F(sv, I.rf), R, dim, w.epsf, stream));
    return TMF_OK;
}

static int check_params(const char* fn, const tmf_xformer_params* w, int n) {
    for (int i = 0; i < n; ++i) {
        const tmf_xformer_params& q = w[i];
        TMF_REQUIRE(q.ln1_g && q.ln1_b && q.wq && q.wkv && q.wo && q.bo && q.ln2_g && q.ln2_b && q.w1 && q.b1 && q.w2 && q.b2 &&
                    q.lnf_g && q.lnf_b, TMF_E_NULL, "%s: a parameter pointer of Transformer instance %d is NULL", fn, i);
    }
    return TMF_OK;
}

extern "C" int tmf_fusion_train_fwd(const tmf_fusion_desc* d, const float* mri_tok, const float* pet_tok,
                                    const tmf_xformer_params* inst, void* saved, size_t saved_bytes, float* cls, void* stream) {
    TMF_TRY(check_desc("tmf_fusion_train_fwd", d));
    TMF_REQUIRE_PTR(mri_tok); TMF_REQUIRE_PTR(pet_tok); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(cls);
    TMF_REQUIRE(d->depth == 0 || inst != nullptr, TMF_E_NULL, "tmf_fusion_train_fwd: argument 'inst' is NULL");
    TMF_REQUIRE_ALIGNED(mri_tok); TMF_REQUIRE_ALIGNED(pet_tok); TMF_REQUIRE_ALIGNED(saved);
    const Plan p = make_plan(*d);
    TMF_REQUIRE(saved_bytes >= p.saved_bytes, TMF_E_WORKSPACE, "tmf_fusion_train_fwd: saved workspace %zu B < required %zu B",
                saved_bytes, p.saved_bytes);
    TMF_TRY(check_params("tmf_fusion_train_fwd", inst, 2 * d->depth));
    char* base = (char*)saved;
    const float* m = mri_tok;
    const float* q = pet_tok;
    for (int i = 0; i < 2 * d->depth; ++i)
        TMF_REQUIRE(p.fused || (!inst[i].mask_o && !inst[i].mask_g && !inst[i].mask_f), TMF_E_SHAPE,
                    "tmf_fusion_train_fwd: Dropout masks need the fused kernels (dim 128, 4 heads of 32 or 8 of 16, mlp 512, N <= 512)");
    if (p.fused && d->depth > 0) {
        const float scale = 1.0f / sqrtf((float)d->dim_head);
        hipStream_t s = (hipStream_t)stream;
        const int n_inst = 2 * d->depth;
        const InstPlan& I = p.I;
        {   // this step's weights in fragment order (forward and backward forms)
            float* pf[2 * TMF_FUSION_MAX_DEPTH];
            float* pb[2 * TMF_FUSION_MAX_DEPTH];
            for (int i = 0; i < n_inst; ++i) { pf[i] = F(base + (size_t)i * I.total, I.pkf); pb[i] = F(base + (size_t)i * I.total, I.pkb); }
            TMF_TRY(tmf_xf_launch_pack(n_inst, inst, pf, pb, s));
        }
        {   // K | V of the first instance's context (the pet tokens)
            tmf_xf_fwd_io io = {};
            io.x = pet_tok; io.pkv_next = F(base, I.pkf) + tmf_xf_pack_kv_offset();
            io.KRn = F(base, I.KR); io.KCn = F(base, I.KC); io.VRn = F(base, I.VR); io.VCn = F(base, I.VC);
            TMF_TRY(tmf_xf_launch_fwd(d->B, d->N, nullptr, &io, scale, 1, d->heads == 8, s));
        }
        for (int i = 0; i < n_inst; ++i) {
            char* sv = base + (size_t)i * I.total;
            tmf_xf_fwd_io io = {};
            io.x = (i & 1) ? q : m;
            io.KR = F(sv, I.KR); io.VC = F(sv, I.VC); io.pk = F(sv, I.pkf);
            io.mask_o = inst[i].mask_o; io.mask_g = inst[i].mask_g; io.mask_f = inst[i].mask_f;
            io.a = F(sv, I.a); io.QR = F(sv, I.QR); io.QC = F(sv, I.QC); io.out = F(sv, I.out); io.lse = F(sv, I.lse);
            io.x1 = F(sv, I.x1); io.f = F(sv, I.f); io.h = F(sv, I.h); io.g = F(sv, I.g); io.x2 = F(sv, I.x2); io.y = F(sv, I.y);
            io.m1 = F(sv, I.m1); io.r1 = F(sv, I.r1); io.m2 = F(sv, I.m2); io.r2 = F(sv, I.r2); io.mf = F(sv, I.mf); io.rf = F(sv, I.rf);
            if (i + 1 < n_inst) {           // this output is the next instance's context
                char* sn = sv + I.total;
                io.pkv_next = F(sn, I.pkf) + tmf_xf_pack_kv_offset();
                io.KRn = F(sn, I.KR); io.KCn = F(sn, I.KC); io.VRn = F(sn, I.VR); io.VCn = F(sn, I.VC);
            }
            TMF_TRY(tmf_xf_launch_fwd(d->B, d->N, &inst[i], &io, scale, 0, d->heads == 8, s));
            if (i & 1) q = F(sv, I.y); else m = F(sv, I.y);
        }
        return tmf_token_pool_fwd(m, q, cls, (int32_t*)(base + p.off_arg), d->B, d->N, d->dim, stream);
    }
    for (int l = 0; l < d->depth; ++l) {
        char* sm = base + (size_t)(2 * l) * p.I.total;
        char* sp = base + (size_t)(2 * l + 1) * p.I.total;
        TMF_TRY(instance_fwd(*d, p, inst[2 * l], m, q, sm, stream));           // mri <- T(mri | pet) + mri
        m = F(sm, p.I.y);
        TMF_TRY(instance_fwd(*d, p, inst[2 * l + 1], q, m, sp, stream));       // pet <- T(pet | NEW mri) + pet
        q = F(sp, p.I.y);
    }
    return tmf_token_pool_fwd(m, q, cls, (int32_t*)(base + p.off_arg), d->B, d->N, d->dim, stream);
}

// Backward of one instance.  dy = gradient w.r.t. its output y.  Writes
//   dx_out   = dy + d(LN_f path)/dx                       (gradient w.r.t. the instance's own input tokens)
//   dctx_out = dctx_acc + d/d(context tokens)             (dctx_acc: what the context tensor has collected so far)
static int instance_bwd(const tmf_fusion_desc& d, const Plan& p, const tmf_xformer_params& w, const tmf_xformer_grads& g,
                        const float* x, const float* c, const char* sv_, const float* dy, const float* dctx_acc,
                        float* dx_out, float* dctx_out, char* sc, void* stream) {
    char* sv = const_cast<char*>(sv_);
    const int R = p.R, dim = d.dim, inner = p.inner, mlp = d.mlp, stride = p.stride;
    const InstPlan& I = p.I;
    const float scale = 1.0f / sqrtf((float)d.dim_head);
    float *dx2 = F(sc, p.s_dx2), *dh = F(sc, p.s_dh), *dx1 = F(sc, p.s_dx1), *dout = F(sc, p.s_dout), *dq = F(sc, p.s_dq),
          *dkv = F(sc, p.s_dkv), *part = F(sc, p.s_part), *lnpart = F(sc, p.s_lnpart);
    // small-parameter gradients share one [row blocks][stride] partial workspace; column order = g.small's layout
    const int o_b2 = 0, o_b1 = dim, o_bo = dim + mlp, o_ln2 = 2 * dim + mlp, o_ln1 = 4 * dim + mlp;
    // block-final LayerNorm
    TMF_TRY(tmf_layernorm_bwd(F(sv, I.x2), w.lnf_g, F(sv, I.mf), F(sv, I.rf), dy, dx2, lnpart, R, dim, stream));
    TMF_TRY(tmf_colsum_finalize(lnpart, p.nblk_ln, 2 * dim, g.lnf, stream));
    // FeedForward
    TMF_TRY(tmf_tok_linear_bwd_input(dx2, w.w2, dh, R, dim, mlp, F(sv, I.h), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, part + o_b2, stride, stream));
    TMF_TRY(tmf_tok_linear_bwd_input(dh, w.w1, dx1, R, mlp, dim, nullptr, F(sv, I.x1), F(sv, I.m2), F(sv, I.r2), w.ln2_g, dx2,
                                     nullptr, part + o_ln2, part + o_b1, stride, stream));
    // Attention
    TMF_TRY(tmf_tok_linear_bwd_input(dx1, w.wo, dout, R, dim, inner, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, part + o_bo, stride, stream));
    TMF_TRY(tmf_xattn_bwd(F(sv, I.q), F(sv, I.kv), F(sv, I.kv) + inner, F(sv, I.out), F(sv, I.lse), dout, dq, dkv, dkv + inner,
                          d.B, d.heads, d.N, d.N, d.dim_head, inner, 2 * inner, 2 * inner, scale, stream));
    TMF_TRY(tmf_tok_linear_bwd_input(dkv, w.wkv, dctx_out, R, 2 * inner, dim, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     dctx_acc, nullptr, nullptr, nullptr, 0, stream));
    TMF_TRY(tmf_tok_linear_bwd_input(dq, w.wq, dx_out, R, inner, dim, nullptr, x, F(sv, I.m1), F(sv, I.r1), w.ln1_g, dx1, dy,
                                     part + o_ln1, nullptr, stride, stream));
    TMF_TRY(tmf_colsum_finalize(part, p.nblk, stride, g.small, stream));
    const float* dys[5] = {dx2, dh, dx1, dkv, dq};
    const float* xs[5] = {F(sv, I.g), F(sv, I.f), F(sv, I.out), c, F(sv, I.a)};
    float* dws[5] = {g.dw2, g.dw1, g.dwo, g.dwkv, g.dwq};
    const int Rs[5] = {R, R, R, R, R}, Ns[5] = {dim, mlp, dim, 2 * inner, inner}, Ks[5] = {mlp, dim, inner, dim, dim};
    return tmf_tok_wgrad_multi(5, dys, xs, dws, Rs, Ns, Ks, sc + p.s_ws, p.ws_bytes, stream);
}

// Fused form of instance_bwd: two launches (query side, key side); weight gradients and column sums are left to the caller.
static int instance_bwd_fused(const tmf_fusion_desc& d, const Plan& p, const tmf_xformer_params& w, int idx, const float* x,
                              const char* sv_, const float* dy, const float* dctx_acc, float* dx_out, float* dctx_out,
                              char* sc, void* stream) {
    char* sv = const_cast<char*>(sv_);
    const InstPlan& I = p.I;
    char* si = sc + (size_t)idx * p.inst_stride;
    tmf_xf_bwd_io io = {};
    io.dy = dy; io.x = x; io.KR = F(sv, I.KR); io.KC = F(sv, I.KC); io.VR = F(sv, I.VR); io.pk = F(sv, I.pkb);
    io.mask_o = w.mask_o; io.mask_g = w.mask_g; io.mask_f = w.mask_f;
    io.QR = F(sv, I.QR); io.QC = F(sv, I.QC); io.out = F(sv, I.out); io.lse = F(sv, I.lse); io.x1 = F(sv, I.x1); io.h = F(sv, I.h);
    io.x2 = F(sv, I.x2); io.m1 = F(sv, I.m1); io.r1 = F(sv, I.r1); io.m2 = F(sv, I.m2); io.r2 = F(sv, I.r2);
    io.mf = F(sv, I.mf); io.rf = F(sv, I.rf);
    io.dx2 = F(si, p.s_dx2); io.dh = F(si, p.s_dh); io.dx1 = F(si, p.s_dx1); io.dq = F(si, p.s_dq); io.dkv = F(si, p.s_dkv);
    io.part = F(si, p.s_part);
    io.DR = F(sc, p.s_DR); io.DC = F(sc, p.s_DC); io.delta = F(sc, p.s_delta);
    io.dx = dx_out; io.dctx = dctx_out; io.dctx_acc = dctx_acc;
    return tmf_xf_launch_bwd(d.B, d.N, &w, &io, 1.0f / sqrtf((float)d.dim_head), d.heads == 8, (hipStream_t)stream);
}

extern "C" int tmf_fusion_train_bwd(const tmf_fusion_desc* d, const float* mri_tok, const float* pet_tok,
                                    const tmf_xformer_params* inst, const void* saved, size_t saved_bytes, const float* dcls,
                                    const tmf_xformer_grads* grads, float* dmri_tok, float* dpet_tok,
                                    void* scratch, size_t scratch_bytes, void* stream) {
    TMF_TRY(check_desc("tmf_fusion_train_bwd", d));

    TMF_REQUIRE_PTR(mri_tok); TMF_REQUIRE_PTR(pet_tok); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(dcls);
    TMF_REQUIRE_PTR(dmri_tok); TMF_REQUIRE_PTR(dpet_tok); TMF_REQUIRE_PTR(scratch);
    TMF_REQUIRE(d->depth == 0 || (inst != nullptr && grads != nullptr), TMF_E_NULL,
                "tmf_fusion_train_bwd: argument 'inst' or 'grads' is NULL");
    TMF_REQUIRE_ALIGNED(saved); TMF_REQUIRE_ALIGNED(scratch); TMF_REQUIRE_ALIGNED(dmri_tok); TMF_REQUIRE_ALIGNED(dpet_tok);
    const Plan p = make_plan(*d);
    TMF_REQUIRE(saved_bytes >= p.saved_bytes, TMF_E_WORKSPACE, "tmf_fusion_train_bwd: saved workspace %zu B < required %zu B",
                saved_bytes, p.saved_bytes);
    TMF_REQUIRE(scratch_bytes >= p.scratch_bytes, TMF_E_WORKSPACE, "tmf_fusion_train_bwd: scratch %zu B < required %zu B",
                scratch_bytes, p.scratch_bytes);
    TMF_TRY(check_params("tmf_fusion_train_bwd", inst, 2 * d->depth));
    for (int i = 0; i < 2 * d->depth; ++i) {
        const tmf_xformer_grads& g = grads[i];
        TMF_REQUIRE(g.small && g.lnf && g.dwq && g.dwkv && g.dwo && g.dw1 && g.dw2, TMF_E_NULL,
                    "tmf_fusion_train_bwd: a gradient pointer of Transformer instance %d is NULL", i);
    }
    const char* base = (const char*)saved;
    char* sc = (char*)scratch;
    const int depth = d->depth;
    float* G[4] = {F(sc, p.s_G[0]), F(sc, p.s_G[1]), F(sc, p.s_G[2]), F(sc, p.s_G[3])};
    // gradient w.r.t. the FINAL mri / pet tokens from the pooling head
    float* Gm = depth == 0 ? dmri_tok : G[0];
    float* Gp = depth == 0 ? dpet_tok : G[1];
    TMF_TRY(tmf_token_pool_bwd(dcls, (const int32_t*)(base + p.off_arg), Gm, Gp, d->B, d->N, d->dim, stream));
    for (int l = depth - 1; l >= 0; --l) {
        const char* sm = base + (size_t)(2 * l) * p.I.total;
        const char* sp = base + (size_t)(2 * l + 1) * p.I.total;
        const float* m_in = l == 0 ? mri_tok : (const float*)(base + (size_t)(2 * (l - 1)) * p.I.total + p.I.y);
        const float* p_in = l == 0 ? pet_tok : (const float*)(base + (size_t)(2 * (l - 1) + 1) * p.I.total + p.I.y);
        const float* m_new = (const float*)(sm + p.I.y);
        // pet instance l: own input p_in, context m_new; its output gradient is Gp, the context has collected Gm so far
        float* Gp_own = G[2];                       // dy + d/d(p_in) through the pet instance
        float* Gm_full = G[3];                      // Gm + d/d(m_new) through the pet instance's keys / values
        if (p.fused) {
            TMF_TRY(instance_bwd_fused(*d, p, inst[2 * l + 1], 2 * l + 1, p_in, sp, Gp, Gm, Gp_own, Gm_full, sc, stream));
            float* Gm_nx = l == 0 ? dmri_tok : G[0];
            float* Gp_nx = l == 0 ? dpet_tok : G[1];
            TMF_TRY(instance_bwd_fused(*d, p, inst[2 * l], 2 * l, m_in, sm, Gm_full, Gp_own, Gm_nx, Gp_nx, sc, stream));
            Gm = Gm_nx;
            Gp = Gp_nx;
            continue;
        }
        TMF_TRY(instance_bwd(*d, p, inst[2 * l + 1], grads[2 * l + 1], p_in, m_new, sp, Gp, Gm, Gp_own, Gm_full, sc, stream));
        // mri instance l: own input m_in, context p_in; output gradient Gm_full, the context has collected Gp_own
        float* Gm_next = l == 0 ? dmri_tok : G[0];
        float* Gp_next = l == 0 ? dpet_tok : G[1];
        TMF_TRY(instance_bwd(*d, p, inst[2 * l], grads[2 * l], m_in, p_in, sm, Gm_full, Gp_own, Gm_next, Gp_next, sc, stream));
        Gm = Gm_next;
        Gp = Gp_next;
    }
    if (p.fused && depth > 0) {
        // every bias / LayerNorm gradient of every instance: one column-sum launch over the per-tile partials
        const int n_inst = 2 * depth;
        const float* parts[2 * TMF_FUSION_MAX_DEPTH];
        float* smalls[2 * TMF_FUSION_MAX_DEPTH];
        float* lnfs[2 * TMF_FUSION_MAX_DEPTH];
        for (int i = 0; i < n_inst; ++i) {
            parts[i] = F(sc + (size_t)i * p.inst_stride, p.s_part);
            smalls[i] = grads[i].small;
            lnfs[i] = grads[i].lnf;
        }
        TMF_TRY(tmf_xf_launch_colsum(n_inst, parts, smalls, lnfs, d->B * p.tiles, (hipStream_t)stream));
        // the five weight gradients of every instance: table-driven launches of up to WG_CHUNK problems
        const int R = p.R, dim = d->dim, inner = p.inner, mlp = d->mlp;
        for (int i0 = 0; i0 < n_inst; i0 += WG_CHUNK / 5) {
            const int ni = (n_inst - i0) < WG_CHUNK / 5 ? (n_inst - i0) : WG_CHUNK / 5;
            const float* dys[WG_CHUNK];
            const float* xs[WG_CHUNK];
            float* dws[WG_CHUNK];
            int Rs[WG_CHUNK], Ns[WG_CHUNK], Ks[WG_CHUNK];
            for (int k = 0; k < ni; ++k) {
                const int i = i0 + k, l = i >> 1;
                char* sv = const_cast<char*>(base) + (size_t)i * p.I.total;
                char* si = sc + (size_t)i * p.inst_stride;
                // the context of instance i: pet tokens entering layer l (mri instance) / the NEW mri tokens (pet instance)
                const float* ctx = (i & 1) ? (const float*)(base + (size_t)(2 * l) * p.I.total + p.I.y)
                                           : (l == 0 ? pet_tok : (const float*)(base + (size_t)(2 * (l - 1) + 1) * p.I.total + p.I.y));
                const float* dy5[5] = {F(si, p.s_dx2), F(si, p.s_dh), F(si, p.s_dx1), F(si, p.s_dkv), F(si, p.s_dq)};
                const float* x5[5] = {F(sv, p.I.g), F(sv, p.I.f), F(sv, p.I.out), ctx, F(sv, p.I.a)};
                float* dw5[5] = {grads[i].dw2, grads[i].dw1, grads[i].dwo, grads[i].dwkv, grads[i].dwq};
                const int N5[5] = {dim, mlp, dim, 2 * inner, inner}, K5[5] = {mlp, dim, inner, dim, dim};
                for (int j = 0; j < 5; ++j) {
                    dys[5 * k + j] = dy5[j]; xs[5 * k + j] = x5[j]; dws[5 * k + j] = dw5[j];
                    Rs[5 * k + j] = R; Ns[5 * k + j] = N5[j]; Ks[5 * k + j] = K5[j];
                }
            }
            TMF_TRY(tmf_tok_wgrad_multi(5 * ni, dys, xs, dws, Rs, Ns, Ks, sc + p.s_ws, p.ws_bytes, stream));
        }
    }
    return TMF_OK;
}
