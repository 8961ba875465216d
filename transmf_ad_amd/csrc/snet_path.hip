// Whole-encoder entries: ONE host call enqueues every launch of an sNet forward (or backward) pass.
//
// reference: models/networks.py:18-61 (sNet: seven Conv3d -> BatchNorm3d -> LeakyReLU blocks, max pools after blocks
// 1, 3, 5, an average pool after block 7), train mode, driven once per modality by models/mymodel.py:206-207 and
// differentiated by kfold_train_adversarial.py:131-132.
//
// Why: issued op by op from Python an encoder is ~45 forward and ~75 backward launches behind ~100 ctypes calls, tensor
// allocations and autograd nodes — ~4 ms of host time per encoder and step, which is on the critical path the moment
// the reference step's two loss.item() syncs (kfold_train_adversarial.py:127-128) drain the queue between forward
// and backward.  Here the host side of a pass is one call: the same kernels, in the same order, with the same
// arguments as the op-by-op path (results are bit-identical — tested), all intermediate tensors carved out of ONE
// caller-allocated workspace, weight gradients reduced straight into the reference's nn.Conv3d layout.
//
// Pure host code: it only calls the library's own extern "C" entries (include/tmf_hip.h) — and the two *_mode forms of
// tmf_conv3d_fwd / tmf_conv3d_stat_blocks (tmf_common.h) that carry tmf_snet_desc.flags' TMF_SNET_ALONE to the conv plan.
#include "tmf_common.h"

namespace {

constexpr int NL = TMF_SNET_BLOCKS;

inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

struct LayerPlan {
    int cin, cout, k, pool;
    int D, H, W;            // conv input = conv output dims of the block
    int oD, oH, oW;         // block output dims (after the pool)
    bool bf;                // bf16 matrix-core kernels for this block's conv / dgrad / wgrad
    bool sp;                // fp32x: forward / data gradient as six bf16 partial products per fp32 product, weight gradient exact fp32
    bool wgf, wgd, wgw;     // fp32: forward / data gradient / weight gradient in the Winograd form (tmf_set_option("conv_wino", ..), conv3d_wino.hip)
    bool x16, z16, o16;     // bf16 storage of the block input, the raw conv output (and dz), the block output
    size_t off_z, off_out, off_vec, off_wf, off_wd;     // in the saved workspace
    size_t x_bytes, z_bytes, out_bytes;
    int cpad;               // floats between the per-channel vectors (mean | invstd | scale | shift)
};

struct Plan {
    LayerPlan L[NL];
    size_t off_part;        // forward statistics partials (max over the blocks)
    size_t off_c1gram = 0, c1gram_bytes = 0;    // fp32: Gram data of the first block (statistics forward, one-pass backward)
    size_t saved_bytes;
    // backward scratch
    size_t off_bpart, off_coef, off_dz, off_gA, off_gB, off_ws, scratch_bytes, ws_bytes;
};

int check_desc(const char* fn, const tmf_snet_desc* d) {
    TMF_REQUIRE_PTR(d);
    TMF_REQUIRE(d->B > 0 && d->D >= 16 && d->H >= 16 && d->W >= 16, TMF_E_SHAPE,
                "%s: volume %dx%dx%d (batch %d): every edge must be >= 16 (four 2x2x2 pools)", fn, d->D, d->H, d->W, d->B);
    TMF_REQUIRE(d->dim >= 32 && d->dim % 32 == 0, TMF_E_SHAPE, "%s: dim=%d must be a positive multiple of 32", fn, d->dim);
    TMF_REQUIRE(d->precision == TMF_PREC_FP32 || d->precision == TMF_PREC_BF16 || d->precision == TMF_PREC_FP32X, TMF_E_ARG,
                "%s: unknown precision %d", fn, d->precision);
    TMF_REQUIRE(!d->storage_bf16 || d->precision == TMF_PREC_BF16, TMF_E_ARG,
                "%s: bf16 activation storage needs the bf16 precision", fn);
    return TMF_OK;
}

Plan make_plan(const tmf_snet_desc& d) {
    Plan p;
    const int q = d.dim / 4, h = d.dim / 2, dm = d.dim, d2 = d.dim * 2;
    const int cin[NL] = {1, q, q, h, h, dm, d2}, cout[NL] = {q, q, h, h, dm, d2, dm};
    const int ks[NL] = {3, 3, 3, 3, 3, 3, 1};
    const int pool[NL] = {TMF_POOL_MAX2, TMF_POOL_NONE, TMF_POOL_MAX2, TMF_POOL_NONE, TMF_POOL_MAX2, TMF_POOL_NONE, TMF_POOL_AVG2};
    const bool b16 = d.precision == TMF_PREC_BF16, s16 = b16 && d.storage_bf16;
    int D = d.D, H = d.H, W = d.W;
    size_t off = 0;
    size_t part_max = 0, bpart_max = 0, z_max = 0, x_max = 0, ws_max = 0;
    for (int l = 0; l < NL; ++l) {
        LayerPlan& L = p.L[l];
        L.cin = cin[l]; L.cout = cout[l]; L.k = ks[l]; L.pool = pool[l];
        L.D = D; L.H = H; L.W = W;
        if (pool[l] != TMF_POOL_NONE) { D /= 2; H /= 2; W /= 2; }
        L.oD = D; L.oH = H; L.oW = W;
        L.bf = b16 && ks[l] == 3 && cin[l] > 1 && cin[l] % 8 == 0;
        // (both channel counts multiples of 8: the data gradient is the same kernel with the roles swapped)
        L.sp = d.precision == TMF_PREC_FP32X && ks[l] == 3 && cin[l] > 1 && cin[l] % 8 == 0 && cout[l] % 8 == 0;
        const int wino = (d.precision == TMF_PREC_FP32 && ks[l] == 3 && cin[l] > 1) ? tmf_conv_wino_mode() : 0;
        L.wgf = wino >= 2 && tmf_conv3d_wino_ok(cin[l], cout[l]);
        L.wgd = wino >= 1 && tmf_conv3d_wino_ok(cout[l], cin[l]);
        L.wgw = wino >= 3 && tmf_conv3d_wgrad_wino_ok(cin[l], cout[l]);
    }
    for (int l = 0; l < NL; ++l) {
        LayerPlan& L = p.L[l];
        // a block hands a bf16 tensor to the next one iff both run on the bf16 kernels (the first block's fused
        // bf16 passes count): networks.py sNet.forward_channels_last
        const bool mine = l == 0 ? b16 : L.bf;
        L.o16 = s16 && l + 1 < NL && mine && p.L[l + 1].bf;
        L.z16 = s16 && L.bf;
        L.x16 = l > 0 && p.L[l - 1].o16;
        const size_t vox = (size_t)d.B * L.D * L.H * L.W, ovox = (size_t)d.B * L.oD * L.oH * L.oW;
        L.x_bytes = vox * L.cin * (L.x16 ? 2 : 4);
        L.z_bytes = l == 0 ? 0 : vox * L.cout * (L.z16 ? 2 : 4);            // the first block never stores z
        L.out_bytes = l == NL - 1 ? 0 : ovox * L.cout * (L.o16 ? 2 : 4);    // the last output is the caller's tensor
        L.cpad = (int)(up256((size_t)L.cout * 4) / 4);
        L.off_z = off; off += up256(L.z_bytes);
        L.off_out = off; off += up256(L.out_bytes);
        L.off_vec = off; off += (size_t)4 * L.cpad * 4;
        const size_t wn = (size_t)L.k * L.k * L.k * L.cin * L.cout;
        const size_t wino_b = tmf_conv3d_wino_weight_bytes(L.cin, L.cout);
        L.off_wf = off; off += up256(L.wgf ? wino_b : wn * (L.sp ? 6 : L.bf ? 2 : 4));
        L.off_wd = off; off += l == 0 ? 0 : up256(L.wgd ? wino_b : wn * (L.sp ? 6 : L.bf ? 2 : 4));
        int nblk, nb2;
        size_t ws;
        if (l == 0) {
            nblk = tmf_c1_blocks(d.B, L.D, L.H, L.W, L.cout);
            nb2 = nblk;
            ws = tmf_c1_bwd_wgrad_workspace_bytes(d.B, L.D, L.H, L.W, L.cout);
            // fp32 and bf16: pair-sum statistics + the tap Gram matrix of the volume, kept for the one-pass backward (conv1_gram.hip)
            p.c1gram_bytes = d.precision == TMF_PREC_FP32 ? tmf_c1_gram_bytes(d.B, L.D, L.H, L.W, L.cout)
                             : b16 ? tmf_c1_gram_bytes_bf16(d.B, L.D, L.H, L.W, L.cout) : 0;   // (bf16: under "c1_gram" 2 only)
            if (p.c1gram_bytes) {
                const size_t wf_ = tmf_c1_bwd_fused_workspace_bytes(d.B, L.D, L.H, L.W, L.cout);
                if (wf_ > ws) ws = wf_;
            }
        } else {
            nblk = L.bf ? tmf_conv3d_bf16_stat_blocks(d.B, L.D, L.H, L.W)
                   : L.sp ? tmf_conv3d_split_stat_blocks(d.B, L.D, L.H, L.W)
                   : L.wgf ? tmf_conv3d_wino_stat_blocks(d.B, L.D, L.H, L.W)
                        : tmf_conv3d_stat_blocks_mode(d.B, L.D, L.H, L.W, L.cin, L.cout, L.k, (d.flags & TMF_SNET_ALONE) ? 1 : 0);
            nb2 = tmf_bn_act_pool_bwd_blocks(d.B, L.D, L.H, L.W, L.cout, L.pool);
            ws = L.bf ? tmf_conv3d_wgrad_bf16_workspace_bytes(d.B, L.D, L.H, L.W, L.cin, L.cout)
                 : L.wgw ? tmf_conv3d_wgrad_wino_workspace_bytes(d.B, L.D, L.H, L.W, L.cin, L.cout)
                      : tmf_conv3d_wgrad_workspace_bytes(d.B, L.D, L.H, L.W, L.cin, L.cout, L.k);
        }
        const size_t pb = (size_t)nblk * 2 * L.cout * 4, pb2 = (size_t)nb2 * 2 * L.cout * 4;
        if (pb > part_max) part_max = pb;
        if (pb2 > bpart_max) bpart_max = pb2;
        if (L.z_bytes > z_max) z_max = L.z_bytes;
        if (l > 0 && L.x_bytes > x_max) x_max = L.x_bytes;
        if (ws > ws_max) ws_max = ws;
    }
    p.off_part = off; off += up256(part_max);
    p.off_c1gram = off; off += up256(p.c1gram_bytes);
    p.saved_bytes = off;
    size_t so = 0;
    p.off_bpart = so; so += up256(bpart_max);
    p.off_coef = so; so += up256((size_t)2 * 2 * d.dim * 4);
    p.off_dz = so; so += up256(z_max);
    p.off_gA = so; so += up256(x_max);
    p.off_gB = so; so += up256(x_max);
    p.off_ws = so; so += up256(ws_max > 16 ? ws_max : 16);
    p.ws_bytes = ws_max > 16 ? ws_max : 16;
    p.scratch_bytes = so;
    return p;
}

struct Vecs { float *mean, *invstd, *scale, *shift; };
inline Vecs vecs_of(char* base, const LayerPlan& L) {
    float* v = (float*)(base + L.off_vec);
    return Vecs{v, v + L.cpad, v + 2 * L.cpad, v + 3 * L.cpad};
}

#define TMF_TRY(call) do { int rc__ = (call); if (rc__ != TMF_OK) return rc__; } while (0)

}  // namespace

extern "C" size_t tmf_snet_saved_bytes(const tmf_snet_desc* d) {
    if (check_desc("tmf_snet_saved_bytes", d) != TMF_OK) return 0;
    const TmfAlgoScope algo_scope(d->flags);
    return make_plan(*d).saved_bytes;
}

extern "C" size_t tmf_snet_bwd_scratch_bytes(const tmf_snet_desc* d) {
    if (check_desc("tmf_snet_bwd_scratch_bytes", d) != TMF_OK) return 0;
    const TmfAlgoScope algo_scope(d->flags);
    return make_plan(*d).scratch_bytes;
}

extern "C" int tmf_snet_train_fwd(const tmf_snet_desc* d, const float* vol, const tmf_snet_params* prm,
                                  void* saved, size_t saved_bytes, float* out, void* stream) {
    TMF_TRY(check_desc("tmf_snet_train_fwd", d));
    const TmfAlgoScope algo_scope(d->flags);          // this call's algorithm choice, if the descriptor carries one
    TMF_REQUIRE_PTR(vol); TMF_REQUIRE_PTR(prm); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(out);
    TMF_REQUIRE_ALIGNED(vol); TMF_REQUIRE_ALIGNED(saved); TMF_REQUIRE_ALIGNED(out);
    const Plan p = make_plan(*d);
    TMF_REQUIRE(saved_bytes >= p.saved_bytes, TMF_E_WORKSPACE, "tmf_snet_train_fwd: saved workspace %zu B < required %zu B",
                saved_bytes, p.saved_bytes);
    for (int l = 0; l < NL; ++l) {
        TMF_REQUIRE(prm->weight[l] && prm->gamma[l] && prm->beta[l], TMF_E_NULL,
                    "tmf_snet_train_fwd: weight / gamma / beta of block %d is NULL", l);
    }
    char* base = (char*)saved;
    float* part = (float*)(base + p.off_part);
    const bool b16 = d->precision == TMF_PREC_BF16;
    const void* x = vol;
    {   // the transformed weights of every Winograd block (forward and data-gradient layouts) in one launch
        const float* pw[8]; float* pf[8]; float* pd[8]; int pco[8], pci[8];
        int n = 0;
        for (int l = 0; l < NL; ++l) {
            const LayerPlan& L = p.L[l];
            if (l == 0 || L.bf || L.sp || !(L.wgf || L.wgd)) continue;
            pw[n] = prm->weight[l]; pf[n] = L.wgf ? (float*)(base + L.off_wf) : nullptr; pd[n] = L.wgd ? (float*)(base + L.off_wd) : nullptr;
            pco[n] = L.cout; pci[n] = L.cin; ++n;
        }
        if (n > 0) TMF_TRY(tmf_pack_conv_weights_wino_multi(n, pw, pf, pd, pco, pci, stream));
    }
    for (int l = 0; l < NL; ++l) {
        const LayerPlan& L = p.L[l];
        const Vecs v = vecs_of(base, L);
        void* o = l == NL - 1 ? (void*)out : (void*)(base + L.off_out);
        void* wf = base + L.off_wf;
        void* wd = base + L.off_wd;
        void* z = base + L.off_z;
        const double count = (double)d->B * L.D * L.H * L.W;
        const bool has_out = (size_t)L.oD * L.oH * L.oW > 0;
        int nblk;
        if (l == 0) {
            // fused first block: the conv output is never stored (csrc/conv1_fused.hip)
            TMF_TRY(tmf_pack_conv_weights(prm->weight[l], (float*)wf, nullptr, L.cout, 1, 27, stream));
            nblk = tmf_c1_blocks(d->B, L.D, L.H, L.W, L.cout);
            // fp32, bf16: pair sums + the tap Gram matrix (bf16: of the rounded volume); fp32x keeps its recomputing passes
            if (p.c1gram_bytes) {
                if (b16) TMF_TRY(tmf_c1_stats_g_bf16(vol, (const float*)wf, part, base + p.off_c1gram, p.c1gram_bytes, d->B, L.D, L.H, L.W,
                                                     L.cout, stream));
                else     TMF_TRY(tmf_c1_stats_g(vol, (const float*)wf, part, base + p.off_c1gram, p.c1gram_bytes, d->B, L.D, L.H, L.W, L.cout,
                                                stream));
                nblk = 2;
            } else if (b16) TMF_TRY(tmf_c1_stats_bf16(vol, (const float*)wf, part, d->B, L.D, L.H, L.W, L.cout, stream));
            else     TMF_TRY(tmf_c1_stats(vol, (const float*)wf, part, d->B, L.D, L.H, L.W, L.cout, stream));
        } else if (L.bf) {
            TMF_TRY(tmf_pack_conv_weights_bf16(prm->weight[l], wf, wd, L.cout, L.cin, 27, stream));
            nblk = tmf_conv3d_bf16_stat_blocks(d->B, L.D, L.H, L.W);
            TMF_TRY(tmf_conv3d_fwd_bf16_t(x, wf, z, part, d->B, L.D, L.H, L.W, L.cin, L.cout,
                                          (L.x16 ? 1 : 0) | (L.z16 ? 2 : 0), stream));
        } else if (L.sp) {
            TMF_TRY(tmf_pack_conv_weights_split3(prm->weight[l], wf, wd, L.cout, L.cin, 27, stream));
            nblk = tmf_conv3d_split_stat_blocks(d->B, L.D, L.H, L.W);
            TMF_TRY(tmf_conv3d_fwd_split((const float*)x, wf, (float*)z, part, d->B, L.D, L.H, L.W, L.cin, L.cout, stream));
        } else {
            if (!L.wgf || !L.wgd)                           // (the Winograd layouts of this block: packed above, all blocks in one launch)
                TMF_TRY(tmf_pack_conv_weights(prm->weight[l], L.wgf ? nullptr : (float*)wf, L.wgd ? nullptr : (float*)wd, L.cout,
                                              L.cin, L.k * L.k * L.k, stream));
            if (L.wgf) {
                nblk = tmf_conv3d_wino_stat_blocks(d->B, L.D, L.H, L.W);
                TMF_TRY(tmf_conv3d_fwd_wino((const float*)x, (const float*)wf, (float*)z, part, d->B, L.D, L.H, L.W, L.cin, L.cout, stream));
            } else {
                const int rt_min = (d->flags & TMF_SNET_ALONE) ? 1 : 0;
                nblk = tmf_conv3d_stat_blocks_mode(d->B, L.D, L.H, L.W, L.cin, L.cout, L.k, rt_min);
                TMF_TRY(tmf_conv3d_fwd_mode((const float*)x, (const float*)wf, (float*)z, part, d->B, L.D, L.H, L.W, L.cin, L.cout,
                                            L.k, rt_min, stream));
            }
        }
        TMF_TRY(tmf_bn_finalize(part, nblk, L.cout, count, prm->gamma[l], prm->beta[l], prm->bias[l], prm->running_mean[l],
                                prm->running_var[l], d->momentum[l], d->eps[l], v.mean, v.invstd, v.scale, v.shift, stream));
        if (has_out) {
            if (l == 0) {
                if (b16) TMF_TRY(tmf_c1_bn_pool_fwd_bf16(vol, (const float*)wf, v.scale, v.shift, o, d->B, L.D, L.H, L.W, L.cout,
                                                         d->slope[l], L.o16 ? 1 : 0, stream));
                else     TMF_TRY(tmf_c1_bn_pool_fwd(vol, (const float*)wf, v.scale, v.shift, (float*)o, d->B, L.D, L.H, L.W,
                                                    L.cout, d->slope[l], stream));
            } else {
                TMF_TRY(tmf_bn_act_pool_fwd_t(z, v.scale, v.shift, o, d->B, L.D, L.H, L.W, L.cout, L.pool, d->slope[l],
                                              (L.z16 ? 1 : 0) | (L.o16 ? 2 : 0), stream));
            }
        }
        x = o;
    }
    return TMF_OK;
}

// Inference (val_step, kfold_train_adversarial.py:144-161): BatchNorm is affine in eval mode, so in the fp32 precision a block
// is ONE kernel (conv + folded BatchNorm + LeakyReLU + pool, tmf_conv3d_fwd_affine; the fused first-block forward for block
// 0) — seven launches + seven weight packs + seven coefficient kernels per encoder, enqueued by one call.  In the bf16
// precisions (round 4) the blocks keep the train path's two kernels — the bf16 matrix-core convolution without statistics,
// then normalise + LeakyReLU + pool with the folded coefficients — i.e. exactly the launches of the block-by-block path
// (bit-identical, tested), issued by one call as well.
extern "C" size_t tmf_snet_eval_workspace_bytes(const tmf_snet_desc* d) {
    if (check_desc("tmf_snet_eval_workspace_bytes", d) != TMF_OK) return 0;
    const TmfAlgoScope algo_scope(d->flags);
    tmf_snet_desc e = *d;
    if (e.precision == TMF_PREC_FP32X) { e.precision = TMF_PREC_FP32; e.storage_bf16 = 0; }
    return make_plan(e).saved_bytes;            // same carving: raw conv outputs (bf16 modes), block outputs, vectors, packed weights
}

extern "C" int tmf_snet_eval_fwd(const tmf_snet_desc* d, const float* vol, const tmf_snet_params* prm,
                                 void* workspace, size_t workspace_bytes, float* out, void* stream) {
    TMF_TRY(check_desc("tmf_snet_eval_fwd", d));
    const TmfAlgoScope algo_scope(d->flags);
    TMF_REQUIRE(d->precision == TMF_PREC_FP32 || d->precision == TMF_PREC_BF16, TMF_E_ARG,
                "tmf_snet_eval_fwd: fp32 and bf16 precisions only");
    TMF_REQUIRE_PTR(vol); TMF_REQUIRE_PTR(prm); TMF_REQUIRE_PTR(workspace); TMF_REQUIRE_PTR(out);
    TMF_REQUIRE_ALIGNED(vol); TMF_REQUIRE_ALIGNED(workspace); TMF_REQUIRE_ALIGNED(out);
    const Plan p = make_plan(*d);
    TMF_REQUIRE(workspace_bytes >= p.saved_bytes, TMF_E_WORKSPACE, "tmf_snet_eval_fwd: workspace %zu B < required %zu B",
                workspace_bytes, p.saved_bytes);
    for (int l = 0; l < NL; ++l)
        TMF_REQUIRE(prm->weight[l] && prm->gamma[l] && prm->beta[l] && prm->running_mean[l] && prm->running_var[l], TMF_E_NULL,
                    "tmf_snet_eval_fwd: weight / gamma / beta / running statistics of block %d is NULL", l);
    char* base = (char*)workspace;
    const bool b16 = d->precision == TMF_PREC_BF16;
    const void* x = vol;
    for (int l = 0; l < NL; ++l) {
        const LayerPlan& L = p.L[l];
        const Vecs v = vecs_of(base, L);
        void* o = l == NL - 1 ? (void*)out : (void*)(base + L.off_out);
        void* wf = base + L.off_wf;
        const bool wino = L.wgf && L.pool != TMF_POOL_AVG2;       // the Winograd form of the one-kernel block (fp32 only: make_plan)
        if (L.bf) TMF_TRY(tmf_pack_conv_weights_bf16(prm->weight[l], wf, nullptr, L.cout, L.cin, 27, stream));
        else if (wino) TMF_TRY(tmf_pack_conv_weights_wino(prm->weight[l], (float*)wf, nullptr, L.cout, L.cin, stream));
        else TMF_TRY(tmf_pack_conv_weights(prm->weight[l], (float*)wf, nullptr, L.cout, L.cin, L.k * L.k * L.k, stream));
        TMF_TRY(tmf_bn_eval_coeffs(prm->gamma[l], prm->beta[l], prm->bias[l], prm->running_mean[l], prm->running_var[l],
                                   d->eps[l], L.cout, v.scale, v.shift, stream));
        if ((size_t)L.oD * L.oH * L.oW == 0) continue;
        if (l == 0) {
            if (b16) TMF_TRY(tmf_c1_bn_pool_fwd_bf16(vol, (const float*)wf, v.scale, v.shift, o, d->B, L.D, L.H, L.W, L.cout,
                                                     d->slope[l], L.o16 ? 1 : 0, stream));
            else     TMF_TRY(tmf_c1_bn_pool_fwd(vol, (const float*)wf, v.scale, v.shift, (float*)o, d->B, L.D, L.H, L.W, L.cout,
                                                d->slope[l], stream));
        } else if (L.bf) {
            void* z = base + L.off_z;
            TMF_TRY(tmf_conv3d_fwd_bf16_t(x, wf, z, nullptr, d->B, L.D, L.H, L.W, L.cin, L.cout,
                                          (L.x16 ? 1 : 0) | (L.z16 ? 2 : 0), stream));
            TMF_TRY(tmf_bn_act_pool_fwd_t(z, v.scale, v.shift, o, d->B, L.D, L.H, L.W, L.cout, L.pool, d->slope[l],
                                          (L.z16 ? 1 : 0) | (L.o16 ? 2 : 0), stream));
        } else if (b16) {       // a block the bf16 kernels do not take (the 1x1x1 layer) inside a bf16 encoder: two kernels, fp32
            float* z = (float*)(base + L.off_z);
            TMF_TRY(tmf_conv3d_fwd((const float*)x, (const float*)wf, z, nullptr, d->B, L.D, L.H, L.W, L.cin, L.cout, L.k, stream));
            TMF_TRY(tmf_bn_act_pool_fwd_t(z, v.scale, v.shift, o, d->B, L.D, L.H, L.W, L.cout, L.pool, d->slope[l], 0, stream));
        } else if (wino) {
            TMF_TRY(tmf_conv3d_fwd_wino_affine((const float*)x, (const float*)wf, v.scale, v.shift, (float*)o, d->B, L.D, L.H, L.W,
                                               L.cin, L.cout, L.pool, d->slope[l], stream));
        } else {
            TMF_TRY(tmf_conv3d_fwd_affine((const float*)x, (const float*)wf, v.scale, v.shift, (float*)o, d->B, L.D, L.H, L.W,
                                          L.cin, L.cout, L.k, L.pool, d->slope[l], stream));
        }
        x = o;
    }
    return TMF_OK;
}

extern "C" int tmf_snet_train_bwd(const tmf_snet_desc* d, const float* vol, const void* saved, size_t saved_bytes,
                                  const float* dout, const tmf_snet_grads* g, void* scratch, size_t scratch_bytes,
                                  void* stream) {
    TMF_TRY(check_desc("tmf_snet_train_bwd", d));
    const TmfAlgoScope algo_scope(d->flags);          // (the forward's: the plan of `saved` is the one the forward laid out)
    TMF_REQUIRE_PTR(vol); TMF_REQUIRE_PTR(saved); TMF_REQUIRE_PTR(dout); TMF_REQUIRE_PTR(g); TMF_REQUIRE_PTR(scratch);
    TMF_REQUIRE_ALIGNED(vol); TMF_REQUIRE_ALIGNED(saved); TMF_REQUIRE_ALIGNED(dout); TMF_REQUIRE_ALIGNED(scratch);
    const Plan p = make_plan(*d);
    TMF_REQUIRE(saved_bytes >= p.saved_bytes, TMF_E_WORKSPACE, "tmf_snet_train_bwd: saved workspace %zu B < required %zu B",
                saved_bytes, p.saved_bytes);
    TMF_REQUIRE(scratch_bytes >= p.scratch_bytes, TMF_E_WORKSPACE, "tmf_snet_train_bwd: scratch %zu B < required %zu B",
                scratch_bytes, p.scratch_bytes);
    char* base = (char*)saved;          // read-only here (the const is dropped for pointer arithmetic only)
    char* sc = (char*)scratch;
    float* part = (float*)(sc + p.off_bpart);
    float* coef = (float*)(sc + p.off_coef);
    void* dz = sc + p.off_dz;
    void* ws = sc + p.off_ws;
    hipStream_t s = (hipStream_t)stream;
    const bool b16 = d->precision == TMF_PREC_BF16;
    // a bias ahead of a batch-statistics BatchNorm has an exactly-zero gradient (the reference returns noise): one fill
    // when the caller laid the bias gradients out back to back (ops.SNetTrain does), one per block otherwise
    bool bias_filled = false;
    {
        float* first = nullptr;
        const float* expect = nullptr;
        size_t total = 0;
        bool contiguous = true;
        for (int l = 0; l < NL; ++l) {
            if (g->dbias[l] == nullptr) continue;
            if (first == nullptr) first = g->dbias[l];
            else if (g->dbias[l] != expect) contiguous = false;
            expect = g->dbias[l] + p.L[l].cout;
            total += (size_t)p.L[l].cout;
        }
        if (first != nullptr && contiguous) {
            hipError_t e = hipMemsetAsync(first, 0, total * 4, s);
            TMF_REQUIRE(e == hipSuccess, (int)e, "tmf_snet_train_bwd: memset failed: %s", hipGetErrorString(e));
            bias_filled = true;
        }
    }
    const void* go = dout;              // gradient w.r.t. the current block's output
    for (int l = NL - 1; l >= 0; --l) {
        const LayerPlan& L = p.L[l];
        const Vecs v = vecs_of(base, L);
        const void* x = l == 0 ? (const void*)vol : (const void*)(base + p.L[l - 1].off_out);
        const void* wf = base + L.off_wf;
        const void* wd = base + L.off_wd;
        const void* z = base + L.off_z;
        const double count = (double)d->B * L.D * L.H * L.W;
        if (g->dbias[l] != nullptr && !bias_filled) {
            hipError_t e = hipMemsetAsync(g->dbias[l], 0, (size_t)L.cout * 4, s);
            TMF_REQUIRE(e == hipSuccess, (int)e, "tmf_snet_train_bwd: memset failed: %s", hipGetErrorString(e));
        }
        if ((size_t)L.oD * L.oH * L.oW == 0) continue;
        if (l == 0 && p.c1gram_bytes && g->dweight[l] != nullptr) {
            // one pass over the volume: BatchNorm sums and D = x (*) dy together; dw from the forward's Gram data (conv1_gram.hip)
            if (b16) TMF_TRY(tmf_c1_bwd_fused_bf16(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, go, base + p.off_c1gram,
                                                   g->dweight[l], g->dgamma[l], g->dbeta[l], ws, p.ws_bytes, d->B, L.D, L.H, L.W, L.cout,
                                                   d->slope[l], L.o16 ? 1 : 0, TMF_DW_REFERENCE, stream));
            else     TMF_TRY(tmf_c1_bwd_fused(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, (const float*)go,
                                              base + p.off_c1gram, g->dweight[l], g->dgamma[l], g->dbeta[l], ws, p.ws_bytes, d->B, L.D,
                                              L.H, L.W, L.cout, d->slope[l], TMF_DW_REFERENCE, stream));
            break;
        }
        if (l == 0) {
            const int nblk = tmf_c1_blocks(d->B, L.D, L.H, L.W, L.cout);
            if (b16) TMF_TRY(tmf_c1_bwd_reduce_bf16(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, go, part, d->B, L.D,
                                                    L.H, L.W, L.cout, d->slope[l], L.o16 ? 1 : 0, stream));
            else     TMF_TRY(tmf_c1_bwd_reduce(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, (const float*)go, part,
                                               d->B, L.D, L.H, L.W, L.cout, d->slope[l], stream));
            TMF_TRY(tmf_bn_bwd_finalize(part, nblk, L.cout, count, g->dgamma[l], g->dbeta[l], coef, stream));
            if (g->dweight[l] != nullptr) {
                if (b16) TMF_TRY(tmf_c1_bwd_wgrad_bf16(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, coef, go,
                                                       g->dweight[l], ws, p.ws_bytes, d->B, L.D, L.H, L.W, L.cout, d->slope[l],
                                                       L.o16 ? 1 : 0, TMF_DW_REFERENCE, stream));
                else     TMF_TRY(tmf_c1_bwd_wgrad(vol, (const float*)wf, v.scale, v.shift, v.mean, v.invstd, coef,
                                                  (const float*)go, g->dweight[l], ws, p.ws_bytes, d->B, L.D, L.H, L.W, L.cout,
                                                  d->slope[l], TMF_DW_REFERENCE, stream));
            }
            break;
        }
        const int io = (L.z16 ? 1 : 0) | (L.o16 ? 2 : 0);
        const int nblk = tmf_bn_act_pool_bwd_blocks(d->B, L.D, L.H, L.W, L.cout, L.pool);
        TMF_TRY(tmf_bn_act_pool_bwd_reduce_t(z, go, v.scale, v.shift, v.mean, v.invstd, part, d->B, L.D, L.H, L.W, L.cout, L.pool,
                                             d->slope[l], io, stream));
        TMF_TRY(tmf_bn_bwd_finalize(part, nblk, L.cout, count, g->dgamma[l], g->dbeta[l], coef, stream));
        TMF_TRY(tmf_bn_act_pool_bwd_apply_t(z, go, v.scale, v.shift, v.mean, v.invstd, coef, dz, d->B, L.D, L.H, L.W, L.cout,
                                            L.pool, d->slope[l], io, stream));
        void* dx = sc + ((l & 1) ? p.off_gA : p.off_gB);
        if (L.bf) {
            if (g->dweight[l] != nullptr)
                TMF_TRY(tmf_conv3d_wgrad_bf16_t(x, dz, g->dweight[l], ws, p.ws_bytes, d->B, L.D, L.H, L.W, L.cin, L.cout,
                                                L.x16 ? 1 : 0, TMF_DW_REFERENCE, stream));
            TMF_TRY(tmf_conv3d_fwd_bf16_t(dz, wd, dx, nullptr, d->B, L.D, L.H, L.W, L.cout, L.cin,
                                          (L.z16 ? 1 : 0) | (L.x16 ? 2 : 0), stream));
        } else {
            if (g->dweight[l] != nullptr) {
                if (L.wgw) TMF_TRY(tmf_conv3d_wgrad_wino((const float*)x, (const float*)dz, g->dweight[l], ws, p.ws_bytes, d->B, L.D, L.H,
                                                         L.W, L.cin, L.cout, TMF_DW_REFERENCE, stream));
                else TMF_TRY(tmf_conv3d_wgrad((const float*)x, (const float*)dz, g->dweight[l], ws, p.ws_bytes, d->B, L.D, L.H, L.W,
                                              L.cin, L.cout, L.k, TMF_DW_REFERENCE, stream));
            }
            if (L.sp) TMF_TRY(tmf_conv3d_fwd_split((const float*)dz, wd, (float*)dx, nullptr, d->B, L.D, L.H, L.W, L.cout, L.cin, stream));
            else if (L.wgd) TMF_TRY(tmf_conv3d_fwd_wino((const float*)dz, (const float*)wd, (float*)dx, nullptr, d->B, L.D, L.H, L.W,
                                                        L.cout, L.cin, stream));
            else TMF_TRY(tmf_conv3d_fwd_mode((const float*)dz, (const float*)wd, (float*)dx, nullptr, d->B, L.D, L.H, L.W, L.cout,
                                             L.cin, L.k, (d->flags & TMF_SNET_ALONE) ? 1 : 0, stream));
        }
        go = dx;
        if (l == TMF_SNET_DEEP_FROM && g->deep_event != nullptr) {       // blocks 6 .. l: every gradient is queued behind this point
            hipError_t e = hipEventRecord((hipEvent_t)g->deep_event, s);
            TMF_REQUIRE(e == hipSuccess, (int)e, "tmf_snet_train_bwd: hipEventRecord failed: %s", hipGetErrorString(e));
        }
    }
    return TMF_OK;
}
