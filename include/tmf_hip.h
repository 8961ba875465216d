/*
 * tmf_hip.h — C ABI of libtmf_hip.so: the MI355X (gfx950) kernels behind the
 * TransMF_AD forward/backward hot path.
 *
 * The reference (Kateridge/TransMF_AD) has no FFI of its own: the whole path is a
 * sequence of ATen calls issued by Python nn.Modules.  Each entry point below
 * therefore replaces a group of ATen calls at a cited reference call site; the
 * Python package `transmf_ad_amd` binds them with ctypes (transmf_ad_amd/_lib.py)
 * from torch.autograd.Functions, and INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (hipMalloc'd / torch.Tensor.data_ptr()),
 *    fp32 unless stated; all buffers, including workspaces, are caller-allocated.
 *  - Activations are channels-last:  x[b][d][h][w][c]  ("NDHWC").
 *  - 3x3x3 weights are tap-major:    w[kd*9+kh*3+kw][cin][cout]; 1x1x1: w[cin][cout].
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it,
 *    never synchronises, never allocates, never throws.  Stateless and re-entrant.
 *  - Return: 0 = launched; <0 = TMF_E_* argument error (nothing launched);
 *    >0 = hipError_t from the launch.  tmf_last_error_string() describes the last
 *    non-zero return on the calling thread.
 */
#ifndef TMF_HIP_H
#define TMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMF_OK              0
#define TMF_E_NULL         -1   /* a required pointer is NULL */
#define TMF_E_SHAPE        -2   /* unsupported / inconsistent shape */
#define TMF_E_ALIGN        -3   /* pointer not 16-byte aligned */
#define TMF_E_WORKSPACE    -4   /* workspace smaller than *_workspace_bytes() */
#define TMF_E_ARG          -5   /* bad enum / scalar argument */

/* layout of a convolution weight gradient: tap-major dw[t][cin][cout] (the kernels' own weight layout) or the
 * reference's nn.Conv3d layout (Cout, Cin, k, k, k) = dw[cout][cin][t], written directly by the final reduction */
#define TMF_DW_TAPMAJOR  0
#define TMF_DW_REFERENCE 1

#define TMF_POOL_NONE 0
#define TMF_POOL_MAX2 1        /* MaxPool3d(2, stride=2), floor mode */
#define TMF_POOL_AVG2 2        /* AvgPool3d(2, stride=2), floor mode */

int         tmf_version(void);                 /* ABI version, currently 1 */
const char* tmf_last_error_string(void);
/* Process-wide tuning knobs (never change results).  "conv_waves" = 2 | 4 | 8 | 16: workgroup shape of the
 * convolution kernels (16, the default: two 8-wave workgroups per CU).  "conv_rt" = 0 | 1 | 2: the register-tiled fp32
 * forward / data-gradient kernel (6x6x12 bricks, csrc/conv3d_mfma.hip conv3d_fwd_rt_kernel) never (default) / for volumes of
 * at most 24^3 voxels that its bricks tile exactly, cin and cout multiples of 16 / wherever the bricks fit (results equal
 * up to fp32 summation order; tmf_conv3d_stat_blocks and tmf_conv3d_fwd_kernel_name follow the choice).  "bf16_v2" = 0 | 1 | 2:
 * bf16 forward / data-gradient kernel with 8x8x8 bricks and 2 x 2 register tiles never / by brick count (default) /
 * always (results equal up to fp32 summation order; tmf_conv3d_bf16_stat_blocks follows the choice).  "wgrad_tr" = 0 | 1 | 2:
 * bf16 weight-gradient kernel with LDS transposing reads never / where it is the faster one (default) / wherever its
 * shape rule allows (cin, cout multiples of 8); equal up to fp32 summation order.  "debug": timing ablations only, live in
 * -DTMF_ABLATE builds (results are garbage when set).
 * Size limits of the convolution entries: one sample of a layer (D*H*W*max(cin, cout)) and one weight tensor stay below
 * 2^29 elements — offsets inside a sample are 32-bit byte offsets of buffer resources; violating shapes return TMF_E_SHAPE. */
int         tmf_set_option(const char* name, int value);

/* ------------------------------------------------------------------------------
 * 3-D convolution, stride 1, "same" zero padding, cross-correlation, NO bias
 * (a bias ahead of BatchNorm is folded into the BN shift / running mean by
 * tmf_bn_finalize).  Replaces F.conv3d at networks.py:22,28,31,37,40,46,49.
 *
 * ksize = 3 or 1.  fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32).
 * If `stat_partial` != NULL the kernel also writes per-workgroup partial sums
 * [nblk][2][cout] (sum z, sum z^2) for the BatchNorm statistics, nblk =
 * tmf_conv3d_stat_blocks(); reduce them with tmf_bn_finalize.
 * The data-gradient of the layer is the SAME entry called on dz with the weight
 * tensor flipped and transposed (w'[26-t][co][ci] = w[t][ci][co]).
 * ---------------------------------------------------------------------------- */
int  tmf_conv3d_fwd(const float* x, const float* w, float* z, float* stat_partial,
                    int B, int D, int H, int W, int cin, int cout, int ksize, void* stream);
int  tmf_conv3d_stat_blocks(int B, int D, int H, int W, int cin, int cout, int ksize);
/* Template-argument text ("FwdCfg<3, 16, 1, 1, 8, 1, 4, 8, 8, 3>" / "WgCfg<1, 4, 8, 8, 8, 32>") of the kernel instance
 * tmf_conv3d_fwd / tmf_conv3d_wgrad launch for a shape, as a kernel trace prints it (measurement aid: bench.py groups
 * its live launch timings by it).  Thread-local static storage; "?" for shapes without a kernel. */
const char* tmf_conv3d_fwd_kernel_name(int B, int D, int H, int W, int cin, int cout, int ksize);
const char* tmf_conv3d_wgrad_kernel_name(int B, int D, int H, int W, int cin, int cout, int ksize);
/* Eval-mode block in ONE pass (BatchNorm is affine in eval mode; val_step, kfold_train_adversarial.py:144-161):
 *   y = pool( LeakyReLU( scale[c] * conv(x)[c] + shift[c] ) ),   pool = TMF_POOL_NONE | MAX2 | AVG2 (floor mode),
 * scale / shift from tmf_bn_eval_coeffs (they absorb the conv bias).  The raw conv output is never written.
 * y: [B][D][H][W][cout] or [B][D/2][H/2][W/2][cout].  cin and cout must be multiples of 4.  Forward only. */
int  tmf_conv3d_fwd_affine(const float* x, const float* w, const float* scale, const float* shift, float* y,
                           int B, int D, int H, int W, int cin, int cout, int ksize, int pool, float slope, void* stream);

/* Weight gradient: dw[t][ci][co] = sum_{b,pos} x[b,pos+t-1][ci] * dz[b,pos][co].
 * Two stages: split-K partial slabs into `workspace`, then a deterministic reduce.
 * Replaces the weight half of aten::convolution_backward for the call sites above. */
size_t tmf_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int cin, int cout, int ksize);
int    tmf_conv3d_wgrad(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                        int B, int D, int H, int W, int cin, int cout, int ksize, int dw_layout, void* stream);

/* bf16 matrix-core variant (BASELINE configs[2]: "bf16 with MFMA 3D conv"), 3x3x3 only: fp32 tensors in HBM,
 * operands rounded to bf16 (RNE) on the way into LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32
 * output and statistics.  w_bf16: bf16 [27][cout][cin] (cin contiguous); cin % 8 == 0.  The data gradient is the
 * same entry on dz with w'[26-t][ci][co] = w[t][co][ci].  stat_partial: [tmf_conv3d_bf16_stat_blocks()][2][cout]. */
int  tmf_conv3d_fwd_bf16(const float* x, const void* w_bf16, float* z, float* stat_partial,
                         int B, int D, int H, int W, int cin, int cout, void* stream);
int  tmf_conv3d_bf16_stat_blocks(int B, int D, int H, int W);
/* Weight gradient on the bf16 matrix cores (x and dz rounded to bf16 while staged, fp32 accumulation, fp32 dw
 * [27][cin][cout]); workspace as tmf_conv3d_wgrad. */
size_t tmf_conv3d_wgrad_bf16_workspace_bytes(int B, int D, int H, int W, int cin, int cout);
int    tmf_conv3d_wgrad_bf16(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                             int B, int D, int H, int W, int cin, int cout, void* stream);
/* bf16 ACTIVATION STORAGE (configs[2]: "bf16 storage + bf16 MFMA with fp32 accumulate"): the same kernels reading /
 * writing bf16 tensors directly (no conversion in the staging path, half the bytes).  io bits for the forward: 1 = x is
 * bf16, 2 = z is bf16 (statistics still from the fp32 accumulators); for the weight gradient io = 1: x and dz are bf16. */
int    tmf_conv3d_fwd_bf16_t(const void* x, const void* w_bf16, void* z, float* stat_partial,
                             int B, int D, int H, int W, int cin, int cout, int io, void* stream);
const char* tmf_conv3d_fwd_bf16_kernel_name(int B, int D, int H, int W, int cin, int cout, int io);   /* as tmf_conv3d_fwd_kernel_name */
const char* tmf_conv3d_wgrad_bf16_kernel_name(int B, int D, int H, int W, int cin, int cout, int io); /* io: 1 = bf16 tensors */
int    tmf_conv3d_wgrad_bf16_t(const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes,
                               int B, int D, int H, int W, int cin, int cout, int io, int dw_layout, void* stream);
/* fp32-ACCURATE variant on the bf16 matrix cores: operands split exactly into three bf16 numbers (hi+mid+lo), the
 * six partial products of order >= 2^-16 accumulated in fp32 (the dropped terms are below one fp32 ulp of the
 * product).  w3_bf16: bf16 [3 parts][27][cout][cin]; same shapes as tmf_conv3d_fwd_bf16; stat_partial:
 * [tmf_conv3d_split_stat_blocks()][2][cout]. */
int  tmf_conv3d_fwd_split(const float* x, const void* w3_bf16, float* z, float* stat_partial,
                          int B, int D, int H, int W, int cin, int cout, void* stream);
int  tmf_conv3d_split_stat_blocks(int B, int D, int H, int W);        /* rows of its stat_partial */
/* WINOGRAD form F(2x2x2, 3x3x3) of the same convolution (csrc/conv3d_wino.hip), exact-fp32 arithmetic on the fp32 matrix
 * pipe: 64 products per 2x2x2 output tile, input and output channel instead of 216.  Replaces aten::conv3d and the input
 * gradient of convolution_backward at networks.py:28,31,37,40,46 for layers with cin % 8 == 0 and cout % 32 == 0
 * (tmf_conv3d_wino_ok).  x [B][D][H][W][cin], z [B][D][H][W][cout] fp32 channels-last; u = the transformed weights from
 * tmf_pack_conv_weights_wino (forward: u_fwd; data gradient: the same entry called with dz, u_dgrad and the channel
 * counts swapped); stat_partial (may be NULL): [tmf_conv3d_wino_stat_blocks()][2][cout].  Results differ from the direct
 * kernels' by fp32 rounding only (about twice their distance to the fp64 value).
 * tmf_conv_wino_mode(): tmf_set_option("conv_wino", 0 | 1 | 2 | 3) / TMF_CONV_WINO — 0 never (the direct kernels), 1 the data
 * gradients, 2 forward and data gradients, 3 (default) forward, data and weight gradients of the train-mode encoder blocks
 * that qualify. */
int    tmf_conv3d_fwd_wino(const float* x, const float* u, float* z, float* stat_partial,
                           int B, int D, int H, int W, int cin, int cout, void* stream);
int    tmf_conv3d_wino_ok(int cin, int cout);
/* ... and the eval-mode block in one pass (tmf_conv3d_fwd_affine's Winograd form, val_step: kfold_train_adversarial.py:144-161):
 * y = LeakyReLU(scale * conv(x, w) + shift), pool TMF_POOL_NONE | TMF_POOL_MAX2 (floor mode) applied before the store. */
int    tmf_conv3d_fwd_wino_affine(const float* x, const float* u, const float* scale, const float* shift, float* y,
                                  int B, int D, int H, int W, int cin, int cout, int pool, float slope, void* stream);
int    tmf_conv3d_wino_stat_blocks(int B, int D, int H, int W);       /* rows of stat_partial: one per workgroup of the persistent
                                                                       * kernel (= compute units; all rows are written), one per
                                                                       * 4x8x8 brick with "wino_p" 0 */
int    tmf_conv3d_wino_bricks(int B, int D, int H, int W);            /* bricks the forward kernel walks per 32 output channels */
int    tmf_conv3d_wino_bricks2(int B, int D, int H, int W, int cin, int cout);   /* ... of a launch with these channel counts: where the split
                                                                         kernel can take it, the one-sample bricks are kept unless the folded
                                                                         geometry saves more than 15 % of them (round 6) */
const char* tmf_conv3d_wino_kernel_name(int B, int D, int H, int W, int stats);        /* the instance a kernel trace shows (as */
const char* tmf_conv3d_wgrad_wino_kernel_name(int B, int D, int H, int W, int cin, int cout);   /* tmf_conv3d_fwd_kernel_name) */
long   tmf_conv3d_wgrad_wino_tiles(int B, int D, int H, int W, int cin, int cout);     /* 2x2x2 tiles (padded: 16 per stage) the weight-
                                                                       * gradient launch multiplies per channel pair */
const char* tmf_conv3d_wino_kernel_name2(int B, int D, int H, int W, int cin, int cout, int stats);   /* ... with the channel counts: the
                                                                       * split kernel conv3d_winox_kernel where it takes the launch */
size_t tmf_conv3d_wino_weight_bytes(int cin, int cout);             /* 64 * cin * cout floats + the same numbers as three bf16 parts
                                                                       * (6 bytes each) behind them */
/* tmf_wino_x_mode(): tmf_set_option("wino_x", 0 | 1) / TMF_WINO_X — 1 (default): tmf_conv3d_fwd_wino (train forward and data gradient)
 * runs csrc/conv3d_winox.hip where cin % 32 == 0, cout % 32 == 0 and the volume takes 4x8x8 bricks: the SAME fp32 products, each
 * operand split exactly into three bf16 numbers (no rounding: 8 + 8 + 8 significand bits), six of the nine partial products on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation — the dropped three are below 2^-24 of the product, under the rounding of the
 * fp32 product itself.  0: the fp32 matrix pipe (conv3d_wino_p_kernel) everywhere. */
int    tmf_wino_x_mode(void);
int    tmf_conv_wino_mode(void);
/* tmf_wino_p_mode(): tmf_set_option("wino_p", 0 | 1) / TMF_WINO_P — 1 (default): the three Winograd entries run their persistent
 * one-wave-per-SIMD kernels (conv3d_wino_p_kernel, conv3d_wino_wgrad_p_kernel; the forward picks per volume between 4x8x8
 * bricks of one sample and 4x4x4 bricks of four samples — tmf_conv3d_wino_bricks() follows it); 0: the two-waves-per-
 * SIMD kernels of round 4.  Same results up to fp32 rounding of the output transform's order of additions. */
int    tmf_wino_p_mode(void);
/* Weight gradient in the same form: dU_p = V_p^T Z_p per position of the transformed tile (V = the forward's input transform of
 * x, Z = A dz A^T), summed over all tiles on the fp32 matrix pipe, then dw = G^T dU G (fp64) — replaces the weight gradient of
 * convolution_backward at networks.py:28,31,37,40,46 for cin % 32 == 0 and cout % 32 == 0.  Arguments as tmf_conv3d_wgrad. */
int    tmf_conv3d_wgrad_wino_ok(int cin, int cout);
size_t tmf_conv3d_wgrad_wino_workspace_bytes(int B, int D, int H, int W, int cin, int cout);
int    tmf_conv3d_wgrad_wino(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                             int B, int D, int H, int W, int cin, int cout, int dw_layout, void* stream);

/* First layer, cin == 1 (networks.py:22): x[b][d][h][w], w[27][cout]. */
int    tmf_conv3d_c1_fwd(const float* x, const float* w, float* z, float* stat_partial,
                         int B, int D, int H, int W, int cout, void* stream);
int    tmf_conv3d_c1_stat_blocks(int B, int D, int H, int W, int cout);
size_t tmf_conv3d_c1_wgrad_workspace_bytes(int B, int D, int H, int W, int cout);
int    tmf_conv3d_c1_wgrad(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                           int B, int D, int H, int W, int cout, int dw_layout, void* stream);

/* ------------------------------------------------------------------------------
 * Fused first block: Conv3d(1->C,3x3x3) -> BatchNorm3d -> LeakyReLU -> MaxPool3d(2) (networks.py:21-26)
 * without materialising the conv output: every pass recomputes it from x (28 MB) instead of
 * reading/writing the 906 MB tensor.  x[b][d][h][w], w[27][C], pooled/dpool [b][D/2][H/2][W/2][C].
 *   tmf_c1_stats        -> stat_partial [tmf_c1_blocks()][2][C]  (reduce with tmf_bn_finalize)
 *   tmf_c1_bn_pool_fwd  -> pooled
 *   tmf_c1_bwd_reduce   -> partial [tmf_c1_blocks()][2][C]       (reduce with tmf_bn_bwd_finalize)
 *   tmf_c1_bwd_wgrad    -> dw[27][C]   (coef from tmf_bn_bwd_finalize)
 * ---------------------------------------------------------------------------- */
int    tmf_c1_blocks(int B, int D, int H, int W, int C);
int    tmf_c1_stats(const float* x, const float* w, float* stat_partial, int B, int D, int H, int W, int C, void* stream);
/* Round 5: the first block through the tap Gram matrix of its INPUT (csrc/conv1_gram.hip; tmf_set_option("c1_gram", 0 | 1) /
 * TMF_C1_GRAM, default 1; fp32).  tmf_c1_stats_g: the statistics WITHOUT a convolution pass and the exact 27 x 27 matrix
 * G[t][t'] = sum_v x~(v + t) x~(v + t') of the zero-padded volume (+ the 27 shifted sums S_t), from 63 offset pair sums of the
 * volume minus per-face-class sums over the one-voxel shell around it, all in fp64 —
 * gram: tmf_c1_gram_bytes() bytes (0: not available — option "c1_gram" off or C > 64), doubles [0,729) G, [729,756) S_t, 756 S, then
 * scratch; stat_partial rows 0 / 1 = the high / low float halves of sum z, sum z^2 (tmf_bn_finalize over 2 rows).  With G the
 * backward of the block is ONE pass over the
 * volume: tmf_c1_bwd_fused = tmf_c1_bwd_reduce + tmf_bn_bwd_finalize + tmf_c1_bwd_wgrad (train mode, fp32):
 *   dw[t][c] = scale_c [ D[t][c] - c0_c S_t - c1_c invstd_c (sum_t' w[t'][c] G[t][t'] - mean_c S_t) ],  D = sum_v x(v + t) dy_c(v)
 * (dy is one element per pooling window: 27 multiply-adds per window beside the BatchNorm sums).  workspace:
 * tmf_c1_bwd_fused_workspace_bytes(). */
size_t tmf_c1_gram_bytes(int B, int D, int H, int W, int C);
size_t tmf_c1_gram_bytes_bf16(int B, int D, int H, int W, int C);   /* what a bf16-mode caller asks: 0 unless "c1_gram" is 2 */
int    tmf_c1_stats_g(const float* x, const float* w, float* stat_partial, void* gram, size_t gram_bytes,
                      int B, int D, int H, int W, int C, void* stream);
size_t tmf_c1_bwd_fused_workspace_bytes(int B, int D, int H, int W, int C);
int    tmf_c1_bwd_fused(const float* x, const float* w, const float* scale, const float* shift, const float* mean,
                        const float* invstd, const float* dpool, const void* gram, float* dw, float* dgamma, float* dbeta,
                        void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, float slope,
                        int dw_layout, void* stream);
/* Round 6: the same two entries for the bf16 mode (TMF_PREC_BF16; conv1_fused_kernel<.., true> multiplies the volume and the taps
 * ROUNDED to bf16): G, S_t of the rounded volume, forms in the rounded taps — the exact statistics of that kernel's z; the backward
 * pass accumulates D from the rounded volume and the unrounded dy (pooled_bf16: dpool is a bf16 tensor). */
int    tmf_c1_stats_g_bf16(const float* x, const float* w, float* stat_partial, void* gram, size_t gram_bytes,
                           int B, int D, int H, int W, int C, void* stream);
int    tmf_c1_bwd_fused_bf16(const float* x, const float* w, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const void* dpool, const void* gram, float* dw, float* dgamma, float* dbeta,
                             void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, float slope,
                             int pooled_bf16, int dw_layout, void* stream);
int    tmf_c1_bn_pool_fwd(const float* x, const float* w, const float* scale, const float* shift, float* pooled,
                          int B, int D, int H, int W, int C, float slope, void* stream);
int    tmf_c1_bwd_reduce(const float* x, const float* w, const float* scale, const float* shift,
                         const float* mean, const float* invstd, const float* dpool, float* partial,
                         int B, int D, int H, int W, int C, float slope, void* stream);
size_t tmf_c1_bwd_wgrad_workspace_bytes(int B, int D, int H, int W, int C);
int    tmf_c1_bwd_wgrad(const float* x, const float* w, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* coef, const float* dpool,
                        float* dw, void* workspace, size_t workspace_bytes,
                        int B, int D, int H, int W, int C, float slope, int dw_layout, void* stream);
/* The same four passes with both products on the bf16 matrix cores (operands rounded to bf16, fp32 accumulation):
 * 2 MFMAs per tile instead of 14 / 16.  Same arguments, same workspace / block counts; pooled_bf16 != 0: the pooled
 * output / its gradient dpool are bf16 tensors (bf16 activation storage, BASELINE configs[2]). */
int    tmf_c1_stats_bf16(const float* x, const float* w, float* stat_partial, int B, int D, int H, int W, int C, void* stream);
int    tmf_c1_bn_pool_fwd_bf16(const float* x, const float* w, const float* scale, const float* shift, void* pooled,
                               int B, int D, int H, int W, int C, float slope, int pooled_bf16, void* stream);
int    tmf_c1_bwd_reduce_bf16(const float* x, const float* w, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const void* dpool, float* partial,
                              int B, int D, int H, int W, int C, float slope, int pooled_bf16, void* stream);
int    tmf_c1_bwd_wgrad_bf16(const float* x, const float* w, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* coef, const void* dpool,
                             float* dw, void* workspace, size_t workspace_bytes,
                             int B, int D, int H, int W, int C, float slope, int pooled_bf16, int dw_layout, void* stream);

/* ------------------------------------------------------------------------------
 * BatchNorm3d (training statistics) + LeakyReLU + 2x2x2 pool, two passes.
 * Replaces F.batch_norm / leaky_relu / max_pool3d / avg_pool3d at
 * networks.py:23-25,29-30,32-34,38-39,41-43,47-48,50-52.
 * ---------------------------------------------------------------------------- */

/* Reduce the conv's partial sums -> per-channel mean / invstd / scale / shift and
 * update the running buffers (momentum, unbiased running_var, torch semantics).
 * conv_bias (may be NULL) is added to the mean that goes into running_mean.
 * count = B*D*H*W.  running_* may be NULL (no update). */
int tmf_bn_finalize(const float* stat_partial, int nblk, int C, double count,
                    const float* gamma, const float* beta, const float* conv_bias,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Eval mode: scale/shift from the running statistics (and the folded conv bias). */
int tmf_bn_eval_coeffs(const float* gamma, const float* beta, const float* conv_bias,
                       const float* running_mean, const float* running_var, float eps, int C,
                       float* scale, float* shift, void* stream);

/* out = pool(leaky_relu(z*scale + shift, slope)); out dims are floor(D/2).. for pools. */
int tmf_bn_act_pool_fwd(const float* z, const float* scale, const float* shift, float* out,
                        int B, int D, int H, int W, int C, int pool, float slope, void* stream);

/* Backward pass 1: per-workgroup partials [nblk][2][C] of sum(dy) and sum(dy*xhat),
 * dy = d(loss)/d(BN output) obtained from `dout` through pool and LeakyReLU. */
int tmf_bn_act_pool_bwd_blocks(int B, int D, int H, int W, int C, int pool);
int tmf_bn_act_pool_bwd_reduce(const float* z, const float* dout, const float* scale, const float* shift,
                               const float* mean, const float* invstd, float* partial,
                               int B, int D, int H, int W, int C, int pool, float slope, void* stream);
/* Reduce partials -> dgamma, dbeta and the two per-channel constants the apply pass needs
 * (coef[0][c] = sum(dy)/count, coef[1][c] = sum(dy*xhat)/count). */
int tmf_bn_bwd_finalize(const float* partial, int nblk, int C, double count,
                        float* dgamma, float* dbeta, float* coef, void* stream);
/* Backward pass 2: dz = scale * (dy - coef0 - xhat*coef1). */
int tmf_bn_act_pool_bwd_apply(const float* z, const float* dout, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* coef, float* dz,
                              int B, int D, int H, int W, int C, int pool, float slope, void* stream);

/* The three passes on typed tensors.  io bit 0: z and dz are bf16 tensors; bit 1: out and dout are bf16 tensors
 * (io = 0 all float, 1 = bf16 z with a float output (the block feeding the fp32 1x1x1 layer), 3 = all bf16).  Arithmetic,
 * per-channel vectors and partials are fp32 in every mode. */
int tmf_bn_act_pool_fwd_t(const void* z, const float* scale, const float* shift, void* out,
                          int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream);
int tmf_bn_act_pool_bwd_reduce_t(const void* z, const void* dout, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, float* partial,
                                 int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream);
int tmf_bn_act_pool_bwd_apply_t(const void* z, const void* dout, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* coef, void* dz,
                                int B, int D, int H, int W, int C, int pool, float slope, int io, void* stream);

/* Sum `nblk` partial rows of `ncol` floats (fp64 accumulation) into out[ncol]. */
int tmf_colsum_finalize(const float* partial, int nblk, int ncol, float* out, void* stream);

/* ------------------------------------------------------------------------------
 * Fused multi-head cross attention:  out = softmax(q k^T * scale) v
 * (networks.py:169-173: einsum -> softmax -> einsum, plus the two rearranges).
 * q: [B][N][*] rows of stride q_stride floats, head h at columns [h*dh, (h+1)*dh);
 * k, v: [B][M][*] rows of stride kv_stride (so they can alias the to_kv output);
 * out: [B][N][heads*dh] ('b n (h d)'); lse: [B][heads][N] log-sum-exp (saved for bwd).
 * dh must be 8, 16, 32 or 64.
 * ---------------------------------------------------------------------------- */
int tmf_xattn_fwd(const float* q, const float* k, const float* v, float* out, float* lse,
                  int B, int heads, int N, int M, int dh, int q_stride, int kv_stride, float scale,
                  void* stream);
/* dq: [B][N][heads*dh] (stride heads*dh); dk, dv: rows of stride dkv_stride. */
int tmf_xattn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse,
                  const float* dout, float* dq, float* dk, float* dv,
                  int B, int heads, int N, int M, int dh, int q_stride, int kv_stride, int dkv_stride,
                  float scale, void* stream);

/* ------------------------------------------------------------------------------
 * LayerNorm over the last dim (networks.py:117,219) and the token pooling of
 * CrossTransformer_MOD_AVG.forward (networks.py:276-281).
 * ---------------------------------------------------------------------------- */
/* y = LayerNorm(x) * gamma + beta (+ residual, [rows][dim], may be NULL: the "+ tokens" of networks.py:262-263). */
int tmf_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                      float* mean, float* rstd, int rows, int dim, float eps, void* stream);
int tmf_layernorm_bwd_blocks(int rows, int dim);
/* partial: [nblk][2][dim] (dgamma, dbeta partials; reduce with tmf_colsum_finalize, ncol = 2*dim). */
int tmf_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* dy, float* dx, float* partial, int rows, int dim, void* stream);

/* ------------------------------------------------------------------------------
 * Linear layers of the fusion transformer fused with their neighbours (token_gemm.hip): one launch per
 * nn.Linear, 16-row token tiles, exact-fp32 MFMA.  Replaces, per Transformer block, F.layer_norm + F.linear
 * (networks.py:117-121 with 158-159, 219), to_out + bias + residual (:160-163, 226), Linear + GELU + Linear +
 * residual (:131-141, 227), and the matching backward passes.
 *
 * forward:  y = [GELU]( LayerNorm?(x) . w^T + bias ) + residual
 *   x [R][K], w [Nout][K] (nn.Linear layout), y [R][Nout]; K % 16 == 0, Nout % 128 == 0.
 *   ln_gamma != NULL: LayerNorm over K first (K must be 128); writes ln_mean / ln_rstd [R] and, when
 *   ln_out != NULL, the normalised rows [R][K].  bias [Nout] and residual [R][Nout] may be NULL.
 *   gelu_pre != NULL: gelu_pre = (..) . w^T + bias and y = GELU(gelu_pre) (erf form); no residual then.
 * backward (input gradient):  dx = E( dy . w ),  dy [R][Nout], w [Nout][K], dx [R][K]; Nout % 16 == 0, K % 128 == 0.
 *   gelu_pre != NULL: E(v) = v * GELU'(gelu_pre), gelu_pre [R][K].
 *   ln_x != NULL (K == 128): E(v) = LayerNormBackward(v; ln_x, ln_mean, ln_rstd, ln_gamma) + add1 + add2, and the
 *   row-block partials of dgamma / dbeta go to ln_partial[blk * partial_stride + {0..127 | 128..255}].
 *   otherwise E(v) = v + add1.   add1 / add2 [R][K] may be NULL.
 *   bias_partial != NULL: column sums of the block's dy rows -> bias_partial[blk * partial_stride + c], c < Nout.
 *   blk < tmf_tok_row_blocks(R); reduce the partials with tmf_colsum_finalize(partial, nblk, partial_stride, ..).
 * ---------------------------------------------------------------------------- */
int tmf_tok_row_blocks(int R);
int tmf_tok_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                       int R, int K, int Nout, const float* ln_gamma, const float* ln_beta, float eps,
                       float* ln_mean, float* ln_rstd, float* ln_out, float* gelu_pre, void* stream);
int tmf_tok_linear_bwd_input(const float* dy, const float* w, float* dx, int R, int Nout, int K,
                             const float* gelu_pre, const float* ln_x, const float* ln_mean, const float* ln_rstd,
                             const float* ln_gamma, const float* add1, const float* add2, float* ln_partial,
                             float* bias_partial, int partial_stride, void* stream);

/* Conv weights from the reference layout (Cout, Cin, k, k, k) of nn.Conv3d (networks.py:22,28,31,37,40,46,49;
 * taps = k^3 = 1 | 27) into the forward / weight-gradient
 * layout w_fwd[t][ci][co] and, when w_dgrad != NULL, the data-gradient layout w_dgrad[taps-1-t][co][ci], in one launch
 * (what the host side otherwise does with permute / flip copies on every step).  w_fwd may be NULL when only the
 * data-gradient layout is wanted (the forward runs in the Winograd form). */
int tmf_pack_conv_weights(const float* w, float* w_fwd, float* w_dgrad, int cout, int cin, int taps, void* stream);
/* The same for the bf16 matrix-core kernels: w_fwd_bf16[t][co][ci] = bf16(w[co][ci][t]) (the layout tmf_conv3d_fwd_bf16
 * reads) and, when w_dgrad_bf16 != NULL, w_dgrad_bf16[taps-1-t][ci][co] (its data-gradient call).  RNE rounding,
 * bit-identical to torch's .to(bfloat16).  cin (and cout, with a dgrad copy) even. */
int tmf_pack_conv_weights_bf16(const float* w, void* w_fwd_bf16, void* w_dgrad_bf16, int cout, int cin, int taps, void* stream);
/* The same for tmf_conv3d_fwd_split (the fp32x mode): every weight as its EXACT three-way bf16 decomposition,
 * w3_fwd[p][t][co][ci] = part p (hi, mid, lo) of w[co][ci][t] and, when w3_dgrad != NULL, w3_dgrad[p][taps-1-t][ci][co];
 * 3 * taps * cout * cin bf16 numbers each — the operand layout torch's split3 (.to(bfloat16), subtract, repeat) gives. */
int tmf_pack_conv_weights_split3(const float* w, void* w3_fwd, void* w3_dgrad, int cout, int cin, int taps, void* stream);
/* The same for tmf_conv3d_fwd_wino: U = G g G^T along the three axes of every 3x3x3 filter (fp64 inside, rounded once),
 * u_fwd[p][cin / 8][2][cout][4] (position p = (pd, ph, pw) of the 4x4x4 transformed tile; input channel 8 g + 4 hs + s) and
 * u_dgrad[p][cout / 8][2][cin][4] (the flipped filter with the channel roles swapped).  Either may be NULL; the forward
 * form needs tmf_conv3d_wino_ok(cin, cout), the data-gradient form tmf_conv3d_wino_ok(cout, cin). */
int tmf_pack_conv_weights_wino(const float* w, float* u_fwd, float* u_dgrad, int cout, int cin, void* stream);
/* ... the same for up to 8 layers in one launch (an encoder's Winograd blocks: one launch per forward instead of one per block) */
int tmf_pack_conv_weights_wino_multi(int n, const float* const* w, float* const* u_fwd, float* const* u_dgrad,
                                     const int* cout, const int* cin, void* stream);

/* Layout conversion between the reference's NCDHW tensors and the channels-last tensors every kernel here uses
 * (voxels = D*H*W).  The model itself never needs it — its input has C == 1 (same bytes either way) and its output
 * is handed back as a strided view — it is here for callers that feed or read intermediate activations. */
int tmf_layout_ncdhw_to_ndhwc(const float* src, float* dst, int B, int C, long voxels, void* stream);
int tmf_layout_ndhwc_to_ncdhw(const float* src, float* dst, int B, int C, long voxels, void* stream);

/* Weight gradients of up to 32 Linears in one launch (the backward of to_q / to_kv / to_out, networks.py:149-155, and of
 * the two FeedForward Linears, :129-132): dw[p][n][k] = sum_r dy[p][r][n] * x[p][r][k]
 * (dy[p]: [R[p]][N[p]], x[p]: [R[p]][K[p]], dw[p]: [N[p]][K[p]] = nn.Linear.weight layout; N, K multiples of 32).
 * dy / x / dw are HOST arrays of device pointers, R / N / K host arrays.  Split over 8 row ranges into `workspace`,
 * then summed in fixed order (deterministic). */
size_t tmf_tok_wgrad_multi_workspace_bytes(int nprob, const int* N, const int* K);
int    tmf_tok_wgrad_multi(int nprob, const float* const* dy, const float* const* x, float* const* dw,
                           const int* R, const int* N, const int* K, void* workspace, size_t workspace_bytes, void* stream);

/* cls[b] = [mean_n mri | mean_n pet | max_n mri | max_n pet]  (4*dim); argmax: int32 [B][2][dim]. */
int tmf_token_pool_fwd(const float* mri, const float* pet, float* cls, int32_t* argmax,
                       int B, int N, int dim, void* stream);
int tmf_token_pool_bwd(const float* dcls, const int32_t* argmax, float* dmri, float* dpet,
                       int B, int N, int dim, void* stream);

/* ------------------------------------------------------------------------------
 * Input pipeline step ahead of the path, on the device (csrc/input_pipeline.hip).  Replaces MONAI's ScaleIntensityd and
 * RandFlipd(spatial_axis=0) of datasets/ADNI.py:64-66 (host, num_workers=0) for volumes already copied to the device
 * (kfold_train_adversarial.py:106-108).  Exact fp32 formulas (bit-identical to the numpy restatement in
 * oracle/input_oracle.py):  minmax[b] = (min, max) of volume b;  dst = (src - min) / (max - min)  (a constant volume
 * gives zeros, as monai.transforms.utils.rescale_array);  flip_d[b] != 0 reverses the first spatial axis of volume b
 * (the random decision itself stays with the caller: MONAI draws it from its own RandomState).  vol / src / dst:
 * [B][D][H][W] fp32;  workspace >= tmf_scale_intensity_workspace_bytes(B).
 * ---------------------------------------------------------------------------- */
size_t tmf_scale_intensity_workspace_bytes(int B);
/* RandRotated(range_x=0.05, prob=0.3) / RandZoomd(min_zoom=0.95, max_zoom=1, prob=0.3) of datasets/ADNI.py:67-68 with the
 * random decisions as inputs: dst[b] = Rotate(angle about the first spatial axis; bilinear, border padding, keep_size)
 * of src[b] where do_rot[b] != 0 (cos_sin[b] = {cos, sin} of the angle in fp32), a copy otherwise;  dst[b] = Zoom(mode
 * "area", edge padding, keep_size) to out_size[b] = {floor(D z), floor(H z), floor(W z)} (1 <= out <= size) where
 * do_zoom[b] != 0, a copy otherwise.  Bit-identical to oracle/input_oracle.py rotate_x / zoom_area.  Not in place. */
int    tmf_rotate_x(const float* src, float* dst, const float* cos_sin, const unsigned char* do_rot,
                    int B, int D, int H, int W, void* stream);
int    tmf_zoom_area(const float* src, float* dst, const int* out_size, const unsigned char* do_zoom,
                     int B, int D, int H, int W, void* stream);
int    tmf_volume_minmax(const float* vol, float* minmax, void* workspace, size_t workspace_bytes, int B, long voxels,
                         void* stream);
int    tmf_scale_flip(const float* src, float* dst, const float* minmax, const unsigned char* flip_d,
                      int B, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------
 * Whole-encoder entries (csrc/snet_path.hip): ONE call enqueues every launch of an sNet train-mode forward, or of its
 * backward.  Replaces, per modality, `self.mri_cnn(mri)` / `self.pet_cnn(pet)` (models/mymodel.py:206-207 ->
 * networks.py:55-61) and the matching slice of `all_loss.backward()` (kfold_train_adversarial.py:131-132).  Same kernels,
 * same order and arguments as calling the per-block entries above one by one — bit-identical results — but the host
 * side of a pass is one call instead of ~100, and every intermediate tensor lives in ONE caller-allocated workspace.
 *
 * Network: channels 1 -> dim/4 -> dim/4 -> dim/2 -> dim/2 -> dim -> 2*dim -> dim, 3x3x3 except the last (1x1x1), max
 * pools after blocks 0, 2, 4 and an average pool after block 6 (TMF_SNET_BLOCKS = 7 blocks).  dim % 32 == 0, every
 * volume edge >= 16.  precision: TMF_PREC_FP32 (exact fp32 MFMA) or TMF_PREC_BF16 (operands rounded to bf16, fp32
 * accumulation; storage_bf16 != 0 keeps the activations between the 3x3x3 blocks as bf16 tensors).
 *
 * Parameters are the reference's state_dict tensors as they are: weight[l] (Cout, Cin, k, k, k), bias[l] (may be NULL),
 * gamma / beta = BatchNorm3d weight / bias, running_mean / running_var (updated in place, torch semantics; may be NULL).
 *   tmf_snet_train_fwd : vol [B][D][H][W] fp32 -> out [B][D/16][H/16][W/16][dim] fp32 (channels-last); `saved`
 *                        (>= tmf_snet_saved_bytes) receives everything backward needs and must stay untouched until then.
 *   tmf_snet_train_bwd : dout (same shape as out) -> gradients written into g: dweight[l] in the nn.Conv3d layout,
 *                        dbias[l] (exact zeros: a bias ahead of batch-statistics BatchNorm), dgamma[l], dbeta[l].  NULL
 *                        dweight / dbias entries are skipped.  `scratch` >= tmf_snet_bwd_scratch_bytes, free afterwards.
 * There is no gradient for vol (the network input needs none: kfold_train_adversarial.py:106-107).
 * ---------------------------------------------------------------------------- */
#define TMF_SNET_BLOCKS 7
#define TMF_PREC_FP32 0
#define TMF_PREC_BF16 1
#define TMF_PREC_FP32X 2          /* fp32-accurate forward / data gradient on the bf16 matrix cores (exact 3-way bf16 split of both
                                     operands, six partial products: tmf_conv3d_fwd_split); first block, 1x1x1 layer and weight
                                     gradients exact fp32 */
/* tmf_snet_desc.flags.  TMF_SNET_ALONE: the caller runs this encoder with nothing beside it on the device, so the partial
 * workgroup rounds of the pooled layers are not filled by a second stream: the fp32 forward / data-gradient convolutions
 * of those layers take the register-tiled kernel (as tmf_set_option("conv_rt", 1) would, for this call only; same results
 * up to fp32 summation order).  model_single at B = 16: 13.72 -> 13.51 ms per step. */
#define TMF_SNET_ALONE 1
/* The ALGORITHM of a call travels in its descriptor (round 6): with TMF_SNET_ALGO set, the fields below decide which kernels the
 * call's fp32 convolutions take — for this call only, whatever tmf_set_option says, so that two models with different settings
 * live in one process and a forward / backward pair cannot disagree about the plan of the `saved` workspace (the backward gets
 * the descriptor of its forward).  Without the bit the process options (tmf_set_option / TMF_* environment) are the default.
 * tmf_snet_algo_flags() encodes the process options of the moment into such a word (the Python binding pins them at forward). */
#define TMF_SNET_ALGO          0x100
#define TMF_SNET_ALGO_WINO(m)  (((m) & 3) << 9)   /* conv_wino: 0 direct kernels .. 3 forward, data and weight gradients in the Winograd form */
#define TMF_SNET_ALGO_WINO_P   0x800              /* wino_p: the persistent one-wave-per-SIMD Winograd kernels */
#define TMF_SNET_ALGO_WINO_X   0x1000             /* wino_x: forward / data gradient as exact 3-way bf16 splits (conv3d_winox.hip) */
#define TMF_SNET_ALGO_C1_GRAM  0x2000             /* c1_gram: the first block through the tap Gram matrix of its input */
#define TMF_SNET_ALGO_C1_GRAM_BF16 0x4000         /* c1_gram 2: in the bf16 mode as well (off by default: slower there) */
#define TMF_SNET_ALGO_C1_SPLIT 0x8000             /* c1_split: z of the first block (fp32) as exact 3-way bf16 splits on the bf16 pipe */
int  tmf_snet_algo_flags(void);
int  tmf_c1_split_mode(void);                     /* the process option "c1_split" (TMF_C1_SPLIT, default 1) or the calling entry's flags */
typedef struct tmf_snet_desc {
    int   B, D, H, W;                /* input volumes (B, 1, D, H, W) */
    int   dim;                       /* sNet(dim) */
    int   precision;                 /* TMF_PREC_* */
    int   storage_bf16;              /* bf16 activation storage between the 3x3x3 blocks (TMF_PREC_BF16 only) */
    float momentum[TMF_SNET_BLOCKS]; /* BatchNorm3d.momentum, .eps and LeakyReLU.negative_slope per block */
    float eps[TMF_SNET_BLOCKS];
    float slope[TMF_SNET_BLOCKS];
    int   flags;                     /* TMF_SNET_ALONE: no other encoder runs beside this one (model_single, TMF_STREAMS=1);
                                      * TMF_SNET_ALGO | ...: this call's algorithm choice (above) */
} tmf_snet_desc;
typedef struct tmf_snet_params {
    const float* weight[TMF_SNET_BLOCKS];
    const float* bias[TMF_SNET_BLOCKS];
    const float* gamma[TMF_SNET_BLOCKS];
    const float* beta[TMF_SNET_BLOCKS];
    float*       running_mean[TMF_SNET_BLOCKS];
    float*       running_var[TMF_SNET_BLOCKS];
} tmf_snet_params;
/* deep_event (optional, a hipEvent_t): recorded on `stream` as soon as every gradient of blocks TMF_SNET_DEEP_FROM .. 6
 * (conv3.0 .. conv4.3: 95 % of an encoder's gradient bytes) has been written — a data-parallel wrapper starts their
 * all-reduce there, under the backward of conv2 / conv1 (60 % of the backward's time), instead of after the call. */
#define TMF_SNET_DEEP_FROM 3
typedef struct tmf_snet_grads {
    float* dweight[TMF_SNET_BLOCKS];
    float* dbias[TMF_SNET_BLOCKS];
    float* dgamma[TMF_SNET_BLOCKS];
    float* dbeta[TMF_SNET_BLOCKS];
    void*  deep_event;
} tmf_snet_grads;
size_t tmf_snet_saved_bytes(const tmf_snet_desc* d);
size_t tmf_snet_bwd_scratch_bytes(const tmf_snet_desc* d);
int    tmf_snet_train_fwd(const tmf_snet_desc* d, const float* vol, const tmf_snet_params* params,
                          void* saved, size_t saved_bytes, float* out, void* stream);
int    tmf_snet_train_bwd(const tmf_snet_desc* d, const float* vol, const void* saved, size_t saved_bytes,
                          const float* dout, const tmf_snet_grads* grads, void* scratch, size_t scratch_bytes, void* stream);
/* Inference form (val_step, kfold_train_adversarial.py:144-161; eval-mode BatchNorm folded into each block's single
 * conv + BN + LeakyReLU + pool kernel): one call per encoder, fp32 precision, running statistics required; `workspace`
 * (>= tmf_snet_eval_workspace_bytes) is scratch. */
size_t tmf_snet_eval_workspace_bytes(const tmf_snet_desc* d);
int    tmf_snet_eval_fwd(const tmf_snet_desc* d, const float* vol, const tmf_snet_params* params,
                         void* workspace, size_t workspace_bytes, float* out, void* stream);

/* ------------------------------------------------------------------------------
 * Whole-fusion entries (csrc/fusion_path.hip): ONE call enqueues every launch of CrossTransformer_MOD_AVG's train-mode
 * forward, or of its backward.  Replaces `self.fuse_transformer(mri_embeddings, pet_embeddings)` (models/mymodel.py:220
 * -> networks.py:272-281: depth x [mri <- Transformer(mri | pet) + mri; pet <- Transformer(pet | NEW mri) + pet], then
 * cat[mean, mean, max, max] over tokens) and its slice of `all_loss.backward()`.  Dropout (options/option.py:39) enters as
 * keep-masks in tmf_xformer_params.  dim == 128; heads*dim_head and mlp multiples of 128.
 *
 * inst[2*l] / inst[2*l + 1] = the mri / pet Transformer(depth=1) of layer l, parameters = the reference's state_dict
 * tensors (nn.Linear weights (out, in)).  Gradients: small = [b2 (dim) | b1 (mlp) | bo (dim) | ln2 gamma | ln2 beta |
 * ln1 gamma | ln1 beta] contiguous (6*dim + mlp floats), lnf = [gamma | beta] of the block-final LayerNorm, dw* in the
 * nn.Linear layout.  mri_tok / pet_tok: [B][N][dim]; cls: [B][4*dim]; dmri_tok / dpet_tok: [B][N][dim].
 * ---------------------------------------------------------------------------- */
#define TMF_FUSION_MAX_DEPTH 16
/* flags: TMF_FUSION_PER_OP = enqueue one launch per Linear / attention / LayerNorm (token_gemm.hip, attention.hip,
 * token_ops.hip: 7 forward + 13 backward launches per instance) even where the fused per-instance kernels of
 * csrc/xformer_fused.hip apply (dim 128, 4 heads of 32 or — round 6 — 8 heads of 16, mlp 512, N <= 512: 1 forward + 2 backward launches per instance,
 * all weight gradients in one launch at the end).  Forward and backward of one pass must see the same desc. */
#define TMF_FUSION_PER_OP 1
typedef struct tmf_fusion_desc { int B, N, dim, heads, dim_head, mlp, depth, flags; } tmf_fusion_desc;
typedef struct tmf_xformer_params {
    const float *ln1_g, *ln1_b;         /* layers.0.0.norm                          (networks.py:117) */
    const float *wq, *wkv, *wo, *bo;    /* layers.0.0.fn.to_q / to_kv / to_out.0    (:149-155) */
    const float *ln2_g, *ln2_b;         /* layers.0.1.norm */
    const float *w1, *b1, *w2, *b2;     /* layers.0.1.fn.net.0 / net.3              (:129-132) */
    const float *lnf_g, *lnf_b;         /* norm                                     (:219) */
    float eps1, eps2, epsf;
    /* Dropout keep-masks of the instance ALREADY scaled by 1 / (1 - p), or NULL (eval, or p = 0): after to_out
     * (networks.py:153) [B*N][dim], after GELU (:131) [B*N][mlp], after the second Linear (:133) [B*N][dim].  The random
     * draw stays with the caller; the same pointers must be passed to forward and backward.  Fused kernels only. */
    const float *mask_o, *mask_g, *mask_f;
} tmf_xformer_params;
typedef struct tmf_xformer_grads {
    float *small, *lnf, *dwq, *dwkv, *dwo, *dw1, *dw2;
} tmf_xformer_grads;
/* Debugging hook (tools/xf_trace.py): when non-NULL, every wave of the fused kernels writes its phase time stamps
 * (shader clock) to fwd / bwd_q / bwd_kv, each [workgroups][4][16] uint64.  NULL (the default) turns it off. */
void   tmf_debug_xf_trace(void* fwd, void* bwd_q, void* bwd_kv);
size_t tmf_fusion_saved_bytes(const tmf_fusion_desc* d);
int    tmf_fusion_uses_fused(const tmf_fusion_desc* d);     /* 1: this descriptor's calls run the fused per-instance kernels */
size_t tmf_fusion_bwd_scratch_bytes(const tmf_fusion_desc* d);
int    tmf_fusion_train_fwd(const tmf_fusion_desc* d, const float* mri_tok, const float* pet_tok,
                            const tmf_xformer_params* inst, void* saved, size_t saved_bytes, float* cls, void* stream);
int    tmf_fusion_train_bwd(const tmf_fusion_desc* d, const float* mri_tok, const float* pet_tok,
                            const tmf_xformer_params* inst, const void* saved, size_t saved_bytes, const float* dcls,
                            const tmf_xformer_grads* grads, float* dmri_tok, float* dpet_tok,
                            void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------
 * The dense heads of model_ad in one launch per direction (csrc/heads.hip).  Replaces `self.D(D_MRI_inp)`,
 * `self.D(D_PET_inp)` on the reversed-gradient token means and `self.fc_cls(fused_embeds)` (models/mymodel.py:209-215,
 * 221; modules :190-194) and their backward.  fc_cls = Linear(4*dim, H1)-BatchNorm1d-ReLU-Dropout-Linear(H1, H2)-
 * BatchNorm1d-ReLU-Dropout-Linear(H2, NC);  D = Linear(dim, HD)-BatchNorm1d-ReLU-Linear(HD, NC), applied to the MRI then
 * the PET mean with separate batch statistics (running statistics updated twice, in that order).  B <= 32 (round 6; instances holding 16 / 32 batch rows in registers).
 *   cls [B][4*dim]; mri_tok / pet_tok [B][N][dim] (their means over N are D's inputs); mask1 [B][H1], mask2 [B][H2]:
 *   Dropout keep-masks ALREADY scaled by 1 / (1 - p), NULL = no dropout (eval, or p = 0); the random draw stays with the
 *   caller.  training != 0: batch statistics + running update (momentum / eps index 0 = fc_cls.1, 1 = fc_cls.5, 2 = D.1);
 *   else running statistics.  Outputs: logits, d_mri_logits, d_pet_logits [B][NC]; `saved` (tmf_heads_saved_bytes) for
 *   backward.  Backward: parameter gradients into g (same shapes as the parameters; D's collect both calls), d_cls,
 *   and d_mri_tok / d_pet_tok = -revgrad_alpha * dmean / N broadcast over the tokens (gradient reversal, mymodel.py:209).
 * ---------------------------------------------------------------------------- */
typedef struct tmf_heads_desc { int B, N, dim, H1, H2, HD, NC, training; float momentum[3], eps[3]; } tmf_heads_desc;
typedef struct tmf_heads_params {
    const float *fc0_w, *fc0_b, *bn1_g, *bn1_b, *fc4_w, *fc4_b, *bn5_g, *bn5_b, *fc8_w, *fc8_b;   /* fc_cls.0/.1/.4/.5/.8 */
    const float *d0_w, *d0_b, *dbn_g, *dbn_b, *d3_w, *d3_b;                                         /* D.0/.1/.3 */
    float *bn1_rm, *bn1_rv, *bn5_rm, *bn5_rv, *dbn_rm, *dbn_rv;                                     /* running statistics */
} tmf_heads_params;
typedef struct tmf_heads_grads {
    float *fc0_w, *fc0_b, *bn1_g, *bn1_b, *fc4_w, *fc4_b, *bn5_g, *bn5_b, *fc8_w, *fc8_b, *d0_w, *d0_b, *dbn_g, *dbn_b, *d3_w, *d3_b;
} tmf_heads_grads;
size_t tmf_heads_saved_bytes(const tmf_heads_desc* d);
size_t tmf_heads_bwd_scratch_bytes(const tmf_heads_desc* d);
int    tmf_heads_fwd(const tmf_heads_desc* d, const float* cls, const float* mri_tok, const float* pet_tok,
                     const float* mask1, const float* mask2, const tmf_heads_params* params, float* logits,
                     float* d_mri_logits, float* d_pet_logits, void* saved, size_t saved_bytes, void* stream);
int    tmf_heads_bwd(const tmf_heads_desc* d, const float* cls, const float* mask1, const float* mask2,
                     const tmf_heads_params* params, const void* saved, size_t saved_bytes, const float* d_logits,
                     const float* d_d_mri_logits, const float* d_d_pet_logits, const tmf_heads_grads* grads,
                     float* d_cls, float* d_mri_tok, float* d_pet_tok, float revgrad_alpha,
                     void* scratch, size_t scratch_bytes, void* stream);

/* Scaled keep-masks of several nn.Dropout modules in ONE launch (mymodel.py:190-191: the two Dropout(0.5) of fc_cls; the
 * Dropout modules of the fusion block, networks.py:131,133,153): out[s][e] = 1 / keep[s] with probability keep[s], else 0,
 * from a counter-based generator (Philox4x32-10 keyed by `seed`, counter = (element / 4, segment, offset)): the masks are a
 * pure function of (seed, offset), `offset` distinguishes the calls of a run.  nseg <= TMF_MASK_SEGMENTS. */
#define TMF_MASK_SEGMENTS 20
int    tmf_dropout_keep_masks(int nseg, float* const* out, const long* numel, const float* keep,
                              unsigned long long seed, unsigned long long offset, void* stream);

/* The heads of the CNN-only models, one launch per direction (csrc/heads.hip).
 * reference: models/mymodel.py:143-178 model_CNN_ad — `fc_cls` = Linear(2 dim, H)-ReLU-Linear(H, NC) on
 * cat[gap(mri), gap(pet)] and `D` (as in model_ad) on revgrad(gap(mri), 2), revgrad(gap(pet), 2): M = 2, HD = 128;
 * models/mymodel.py:13-41 model_single — `fc` = Linear(dim, H)-ReLU-Linear(H, NC) on avgpool(cnn(img)): M = 1, HD = 0
 * (pet_tok, the D parameters / outputs / gradients are then NULL).  Tokens [B][N][dim] = the encoder output, channels last;
 * gap == mean over N.  B <= 32, dim % 16 == 0, M dim <= 512.  momentum / eps are D's BatchNorm1d's. */
typedef struct tmf_heads_cnn_desc { int B, N, dim, M, H, HD, NC, training; float momentum, eps; } tmf_heads_cnn_desc;
typedef struct tmf_heads_cnn_params {
    const float *fc0_w, *fc0_b, *fc2_w, *fc2_b;              /* [H][M dim], [H], [NC][H], [NC] */
    const float *d0_w, *d0_b, *dbn_g, *dbn_b;                /* [HD][dim], [HD], [HD], [HD] */
    float *dbn_rm, *dbn_rv;                                  /* running statistics: updated twice per training call, or NULL */
    const float *d3_w, *d3_b;                                /* [NC][HD], [NC] */
} tmf_heads_cnn_params;
typedef struct tmf_heads_cnn_grads { float *fc0_w, *fc0_b, *fc2_w, *fc2_b, *d0_w, *d0_b, *dbn_g, *dbn_b, *d3_w, *d3_b; } tmf_heads_cnn_grads;
size_t tmf_heads_cnn_saved_bytes(const tmf_heads_cnn_desc* d);
size_t tmf_heads_cnn_bwd_scratch_bytes(const tmf_heads_cnn_desc* d);
int    tmf_heads_cnn_fwd(const tmf_heads_cnn_desc* d, const float* mri_tok, const float* pet_tok,
                         const tmf_heads_cnn_params* params, float* logits, float* d_mri_logits, float* d_pet_logits,
                         void* saved, size_t saved_bytes, void* stream);
int    tmf_heads_cnn_bwd(const tmf_heads_cnn_desc* d, const tmf_heads_cnn_params* params, const void* saved, size_t saved_bytes,
                         const float* d_logits, const float* d_d_mri_logits, const float* d_d_pet_logits,
                         const tmf_heads_cnn_grads* grads, float* d_mri_tok, float* d_pet_tok, float revgrad_alpha,
                         void* scratch, size_t scratch_bytes, void* stream);

/* ---- optimizer step: torch.optim.Adam over ALL parameter tensors in one launch --------------------------------------
 * reference: kfold_train_adversarial.py:135 (optimizer.step()), utils/utils.py:38-39 (Adam, lr 1e-4, weight decay 0).
 * params[i] / grads[i]: device pointers of tensor i (numel[i] contiguous floats); grads[i] == NULL skips the tensor, as
 * torch does for a parameter without a gradient.  exp_avg / exp_avg_sq: the two moment estimates of ALL tensors in two
 * flat caller-owned buffers of tmf_adam_state_elems(n, numel) floats, zero before the first step (tensor i's slice
 * starts at the sum of the preceding numel, each rounded up to a multiple of 4).  step: the update count INCLUDING this
 * one (1 on the first call).  weight_decay is torch's L2 form (added to the gradient).  n <= TMF_ADAM_MAX_TENSORS per
 * call (the tensor table is a kernel argument); larger sets take several calls with their own state slices. */
#define TMF_ADAM_MAX_TENSORS 160
long   tmf_adam_state_elems(int n, const long* numel);
int    tmf_adam_step(int n, float* const* params, const float* const* grads, const long* numel,
                     float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TMF_HIP_H */
