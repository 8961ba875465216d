// conv3d_mfma.hip — im2col-free fp32 MFMA 3-D convolution for gfx950 (MI355X).
//
// Forward / data-gradient:  z[pos][co] = sum_{tap,ci} x[pos + tap - 1][ci] * w[tap][ci][co]
// as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak):
//   * a workgroup owns a TD x TH x TW brick of output voxels and NB output channels;
//   * the (TD+2)(TH+2)(TW+2) x CINC input halo is staged ONCE per input-channel chunk in
//     LDS (channels-last rows, 16-B padded) and re-used by all 27 taps -> no im2col,
//     HBM/L2 sees each input voxel ~2.3x instead of 27x;
//   * weights stream through a double-buffered LDS ring, one (kd,kh) row of 3 taps per
//     stage, prefetched into registers while the previous stage computes (one barrier
//     per stage);
//   * A operand (voxels x cin) is read with one ds_read_b128 per 4 MFMAs using a
//     K-permutation: within a group of 8 input channels lanes 0-31 take channels 0-3
//     and lanes 32-63 channels 4-7 (the GEMM K order is free as long as B agrees);
//   * epilogue: 128-B row segments of NDHWC output per half-wave + per-workgroup
//     sum / sum-of-squares partials for the BatchNorm statistics.
//
// Weight-gradient: dw[tap][ci][co] = sum_pos x[pos+tap-1][ci] dz[pos][co] with
// M = ci, N = co, K = voxels; the 27 taps are spread over the 4 waves (7 accumulators
// each), the workgroup walks a contiguous range of bricks (split-K) and writes one
// partial slab; a second kernel reduces the slabs deterministically.
//
// Replaces aten::conv3d / convolution_backward at /root/reference/models/networks.py:
// 22,28,31,37,40,46,49.
#include <type_traits>
#include "tmf_common.h"

// kernel A/B switches (builds for TMF_LIB=...): plane-long partial sums, buffer-resource staging
#ifndef TMF_CONV_ACC
#define TMF_CONV_ACC 1
#endif
#ifndef TMF_CONV_BUF
#define TMF_CONV_BUF 1
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------
template <int KS_, int CINC_, int MT_, int NT_, int WM_, int WN_, int TD_, int TH_, int TW_, int TPS_>
struct FwdCfg {
    static constexpr int KS = KS_, CINC = CINC_, MT = MT_, NT = NT_, WM = WM_, WN = WN_;
    static constexpr int TD = TD_, TH = TH_, TW = TW_, TPS = TPS_;
    static constexpr int PAD = KS / 2;
    static constexpr int HD = TD + 2 * PAD, HH = TH + 2 * PAD, HW = TW + 2 * PAD;
    static constexpr int NHALO = HD * HH * HW;
    static constexpr int CP = CINC + 4;                  // padded halo row (floats), keeps 16-B alignment
    static constexpr int NPOS = TD * TH * TW;
    static constexpr int NB = 32 * NT * WN;              // output channels per workgroup
    static constexpr int NTAPS = KS * KS * KS;
    static constexpr int NSTAGES = NTAPS / TPS;
    static constexpr int BSTAGE = TPS * CINC * NB;       // floats per weight stage
    static constexpr int NW = WM * WN;                   // wavefronts per workgroup (4, or 8 = 2 per SIMD)
    static constexpr int NTHR = 64 * NW;
    static constexpr int BV = (BSTAGE / 4 + NTHR - 1) / NTHR;   // float4 prefetch registers per thread
    static constexpr int RED = NW * 32 * NT * 2;         // cross-wave stat scratch (floats)
    static constexpr size_t LDS_BYTES = (size_t)(NHALO * CP + 2 * BSTAGE + RED) * 4;
    static_assert(NPOS == 32 * MT * WM, "brick must be MT*WM tiles of 32 voxels");
    static_assert(TD == 4 && (TW == 8 || TW == 4) && TH * TW == 8 * MT * WM, "lane mapping: 4 planes x 8 (h, w) per tile");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(CINC % 8 == 0, "cin chunk is a multiple of 8");
    static_assert(NTAPS % TPS == 0, "taps per stage must divide the tap count");
    static_assert(KS == 1 || TPS == 1 || TPS == 3, "stage = 1 tap or one kw row");
    // two 8-wave workgroups per CU need <= 128 registers per wave: ask for 4 waves per SIMD where the LDS allows two
    // (the kernel sits at 117-123 without the bound; an innocent-looking change once pushed it to 130-156 and silently
    // halved the occupancy — the bound turns that into a visible spill instead)
    static constexpr int MIN_WAVES = (NW == 8 && LDS_BYTES * 2 <= 160 * 1024) ? 4 : 1;
};

// FUSED = true is the eval-mode block in ONE pass (BatchNorm is affine there): the epilogue applies
// y = LeakyReLU(scale * z + shift) to the accumulators and, for a pooled block, reduces the 2x2x2 windows before
// anything is stored — d pairs sit in one lane's registers (r & 3), w pairs in lane / lane + 32, h pairs in the same
// lane (4x4x4 bricks, MT = 2) or in the neighbouring wave (through LDS).  z is then the (pooled) output tensor.
template <class C, bool VEC, bool FUSED = false>
__global__ __launch_bounds__(C::NTHR, C::MIN_WAVES) void conv3d_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z,
    float* __restrict__ stat_partial, int D, int H, int W, int Cin, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles, int dbg,
    const float* __restrict__ aff_scale = nullptr, const float* __restrict__ aff_shift = nullptr,
    float slope = 0.f, int pool = 0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;
    float* Bs = smem + C::NHALO * C::CP;
    float* red = Bs + 2 * C::BSTAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hsel = lane >> 5;
    const int wm = wave / C::WN, wn = wave % C::WN;

    const int tile = xcd_contiguous(blockIdx.x, ntiles);
    int t = tile;
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; t /= tilesH;
    const int td = t % tilesD;
    const int b = t / tilesD;
    const int d0 = td * C::TD, h0 = th * C::TH, w0 = tw * C::TW;
    const int n0 = blockIdx.y * C::NB;
    // a wave is idle when its output channels lie beyond Cout, or when ALL brick rows h of its M-tiles lie beyond H
    // (wave = brick row: an odd pooled size like 27 leaves 5 of the last brick's 8 rows outside the volume) — it still
    // takes part in staging and barriers, but leaves the matrix pipe to the other waves
    const bool wave_active = (n0 + wn * C::NT * 32) < Cout && h0 + (wm * C::MT) * (8 / C::TW) < H;

    // per-lane halo float index of the voxel this lane feeds to M-tile i (tap (0,0,0))
    int a_lane[C::MT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i) {
        // M-tile T covers all TD = 4 planes of 8 / TW brick rows; lane = (d in the two LOW bits, then w, then h).
        // With the odd 16-B-slot row pitch this is the mapping whose ds_read_b128 service groups ({0-3,12-15,20-27},
        // {4-11,16-19,28-31} per half-wave) hit 16 distinct slots for every tap offset; the natural (h, w) raster
        // is 3-way conflicted (exhaustive search over bit permutations).
        const int T = wm * C::MT + i, rest = l31 >> 2;
        const int pd = l31 & 3, pw = rest % C::TW, ph = T * (8 / C::TW) + rest / C::TW;
        a_lane[i] = ((pd * C::HH + ph) * C::HW + pw) * C::CP + hsel * 4;
    }
    const int b_lane = hsel * 4 * C::NB + wn * C::NT * 32 + l31;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* xb = x + (size_t)b * D * H * W * Cin;

    // ---- brick-invariant addressing, computed once ----
    // VEC path: x (one sample) and w are read through buffer resources: per-lane BYTE offsets are computed once per
    // brick, the chunk / stage base travels in the scalar offset, and an out-of-volume (or out-of-tile) piece carries
    // the offset 2^31 >= num_records, which the hardware answers with zeros: no compare / select / 64-bit address per
    // piece inside the chunk and stage loops (every v_* instruction here is paid in matrix time).
    constexpr int C4 = C::CINC / 4;
    constexpr int HV = (C::NHALO * C4 + C::NTHR - 1) / C::NTHR;
    constexpr int OOB = (int)0x80000000u;
    constexpr bool BUF = VEC && TMF_CONV_BUF;
    const int esz = BUF ? 4 : 1;        // offsets in bytes (buffer path) or elements (scalar path)
    int hoff[HV];                       // offset of the halo position in the sample (+ c4 * 4 channels); < 0 = zero fill
    {
        // piece q of this thread is halo position tid / C4 + q * (NTHR / C4), same channel quad: the position is taken
        // apart into (hd, hh, hw) once and then WALKED by the constant step with two carries — the three divisions per
        // piece were most of this per-brick prologue, and every vector instruction here is paid in matrix time
        static_assert(C::NTHR % C4 == 0, "all pieces of a thread share their channel quad");
        constexpr int STEP = C::NTHR / C4;
        constexpr int SW = STEP % C::HW, SH = (STEP / C::HW) % C::HH, SD = STEP / (C::HW * C::HH);
        const int hp0 = tid / C4, c4 = tid % C4;
        int hw = hp0 % C::HW, hh = (hp0 / C::HW) % C::HH, hd = hp0 / (C::HW * C::HH);
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            const int gd = d0 + hd - C::PAD, gh = h0 + hh - C::PAD, gw = w0 + hw - C::PAD;
            const bool ok = hd < C::HD && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            hoff[q] = ok ? (((gd * H + gh) * W + gw) * Cin + c4 * 4) * esz : OOB;
            hw += SW;
            if (hw >= C::HW) { hw -= C::HW; ++hh; }
            hh += SH;
            if (hh >= C::HH) { hh -= C::HH; ++hd; }
            hd += SD;
        }
    }
    int boff[C::BV];                    // weight offset inside a (stage, chunk) slab; < 0 = outside
    int bci[C::BV];                     // input channel inside the chunk (to test against Cin)
#pragma unroll
    for (int q = 0; q < C::BV; ++q) {
        const int e = tid + q * C::NTHR;
        const int row = e / (C::NB / 4), col = (e % (C::NB / 4)) * 4;
        const int co = n0 + col;
        bci[q] = row % C::CINC;
        boff[q] = (e < C::BSTAGE / 4 && (VEC ? co < Cout : true)) ? (((row / C::CINC) * Cin + bci[q]) * Cout + co) * esz : OOB;
    }
    const bool ragged = Cin % C::CINC != 0;               // the last chunk is partly beyond Cin
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, D * H * W * Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, C::NTAPS * Cin * Cout * 4, 0x00020000);

#ifndef TMF_CONV_PIPE
#define TMF_CONV_PIPE 1
#endif
    // PIPE: the NEXT chunk's halo and first weight stage are requested during the last stage of the current chunk (the
    // registers of both prefetches are free there).  Measured: -1.1 % on the one-tile kernels with 4 chunks (conv2.3
    // data gradient), nothing with 2 chunks, +3-4 % (slower) on the 4x4x4-brick kernels, and the two-tile kernels
    // spill under their 128-register bound — so the exposed global-load latency per chunk is not what the remaining
    // ~7 % of waits are made of.  Enabled where it measured faster.
    constexpr bool PIPE = BUF && TMF_CONV_PIPE && C::NT == 1 && C::TPS == 3;
    f32x4 hreg[HV];
    f32x4 breg[C::BV];
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            const int c = c0 + ((tid + q * C::NTHR) % C4) * 4;
            if constexpr (BUF) {
                const int off = (ragged && c >= Cin) ? OOB : hoff[q];
                hreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, c0 * 4, 0));
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (hoff[q] >= 0) {
                    const float* src = xb + hoff[q] + c0;
                    if (VEC) {
                        if (c < Cin) v = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < Cin) v[u] = src[u];
                    }
                }
                hreg[q] = v;
            }
        }
    };
    auto load_b = [&](int st, int c0) {
        if constexpr (BUF) {
            const int sbase = ((st * C::TPS * Cin + c0) * Cout) * 4;              // wave-uniform
#pragma unroll
            for (int q = 0; q < C::BV; ++q) {
                const int off = (ragged && c0 + bci[q] >= Cin) ? OOB : boff[q];
                breg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, sbase, 0));
            }
        } else {
            const float* wst = w + (size_t)(st * C::TPS * Cin + c0) * Cout;       // wave-uniform
#pragma unroll
            for (int q = 0; q < C::BV; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (boff[q] >= 0 && c0 + bci[q] < Cin) {
                    const float* src = wst + boff[q];
                    if (VEC) {
                        v = *reinterpret_cast<const f32x4*>(src);
                    } else {
                        const int co = n0 + ((tid + q * C::NTHR) % (C::NB / 4)) * 4;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (co + u < Cout) v[u] = src[u];
                    }
                }
                breg[q] = v;
            }
        }
    };

    for (int c0 = 0; c0 < Cin; c0 += C::CINC) {
        if (c0 > 0) __syncthreads();   // everyone is done with the previous halo and weight ring
        if (!PIPE || c0 == 0) {
            load_halo(c0);
            load_b(0, c0);
        }
        // loads issued after stage st has started: the next stage's weights, or (last stage) the next chunk's first loads
        auto next_loads = [&](int st) {
            if (st + 1 < C::NSTAGES) load_b(st + 1, c0);
            else if (PIPE && c0 + C::CINC < Cin) { load_halo(c0 + C::CINC); load_b(0, c0 + C::CINC); }
        };
        auto store_b = [&](int buf) {
#pragma unroll
            for (int q = 0; q < C::BV; ++q) {
                const int e = tid + q * C::NTHR;
                if (e < C::BSTAGE / 4)
                    *reinterpret_cast<f32x4*>(&Bs[buf * C::BSTAGE + e * 4]) = breg[q];
            }
        };

#pragma unroll
        for (int q = 0; q < HV; ++q) {
            const int e = tid + q * C::NTHR;
            if (e < C::NHALO * C4) *reinterpret_cast<f32x4*>(&halo[(e / C4) * C::CP + (e % C4) * 4]) = hreg[q];
        }
        // Partial sums: a stage (NT = 1) or a whole kd plane (NT = 2: 9 taps x CINC <= 288 products per output)
        // accumulates from zero and is then added to the running sum.  The fp32 MFMA is a strict k-ordered fma chain,
        // and one 864..3456-term chain would carry ~sqrt(K) ulp of drift; short chains + 3..9 adds per channel chunk keep
        // it 2-5x smaller.  The adds are vector-ALU work = matrix time: plane-long chains cut them to a
        // third.  The one-N-tile kernels on 4x8x8 bricks keep per-stage partials: with plane-long sums (both loops
        // unrolled, TMF_CONV_ACC1=1) they gain 2 %, but the error of the 512-d feature vector against the fp64 reference
        // doubles (4e-6 -> 8e-6), and the ill-conditioned batch-2 head of the ad_full_b2 fixture amplifies that to
        // 1.02e-3 on one logit — over the 1e-3 gate.  Parity wins.
#ifndef TMF_CONV_ACC1
#define TMF_CONV_ACC1 0
#endif
        constexpr bool UNROLL_ALL = C::NT == 1 && C::TPS == 3 && C::KS == 3;     // small stage body: 9 stages unrolled
        constexpr int ACC = (C::KS == 1 || (UNROLL_ALL && !(TMF_CONV_ACC1 && VEC)) || !TMF_CONV_ACC) ? 1 : 9 / C::TPS;      // stages per partial sum
        static_assert(C::NSTAGES % ACC == 0, "partial sums cover whole kd planes");
        auto mma = [&](int st, int buf, f32x16 (&part)[C::MT][C::NT]) {       // this wave's MFMAs of stage st
            int stage_off;                                                     // halo offset of the stage's first tap
            if (C::KS == 1) stage_off = 0;
            else if (C::TPS == 3) stage_off = ((st / 3) * C::HH + (st % 3)) * C::HW * C::CP;
            else stage_off = (((st / 9) * C::HH + (st / 3) % 3) * C::HW + st % 3) * C::CP;
            // B reads of the 64-channel tiles: two base registers 128 B apart (the two N-tiles), which
            // the compiler must not recognise as one base: every read is then a multiple of 256 B away from its base and
            // pairs up as ds_read2st64 with instruction offsets that reach the whole weight ring.  Pairing the two
            // N-tiles of a row instead (plain ds_read2, 1-KB reach) cost a vector add per 4 rows.
            const float* bs0 = Bs + buf * C::BSTAGE + b_lane;
            int delta = 32;                                // (an opaque OFFSET: laundering the pointer would lose the LDS
            asm volatile("" : "+v"(delta));                //  address space and turn the reads into flat loads)
            const float* bs1 = bs0 + delta;
#pragma unroll
            for (int tp = 0; tp < C::TPS; ++tp) {
#pragma unroll
                for (int g = 0; g < C::CINC / 8; ++g) {
                    f32x4 a[C::MT];
#pragma unroll
                    for (int i = 0; i < C::MT; ++i)
                        a[i] = *reinterpret_cast<const f32x4*>(&halo[a_lane[i] + stage_off + tp * C::CP + g * 8]);
#pragma unroll
                    for (int j = 0; j < C::NT; ++j) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int row = tp * C::CINC + g * 8 + s;
                            const float bv = C::NB == 64 ? (j ? bs1 : bs0)[row * C::NB] : bs0[row * C::NB + j * 32];
#pragma unroll
                            for (int i = 0; i < C::MT; ++i)
                                part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], bv, part[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        };
        if constexpr (ACC == 1) {
            for (int st = 0; st < C::NSTAGES; ++st) {                  // unrolling left to the compiler
                const int buf = st & 1;
                store_b(buf);
                __syncthreads();
                next_loads(st);                            // in flight while this stage computes
                if (wave_active && !(dbg & 2)) {
                    f32x16 part[C::MT][C::NT];
#pragma unroll
                    for (int i = 0; i < C::MT; ++i)
#pragma unroll
                        for (int j = 0; j < C::NT; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
                    mma(st, buf, part);
#pragma unroll
                    for (int i = 0; i < C::MT; ++i)
#pragma unroll
                        for (int j = 0; j < C::NT; ++j) acc[i][j] += part[i][j];
                }
            }
        } else if constexpr (UNROLL_ALL) {
#pragma unroll
            for (int sg = 0; sg < C::NSTAGES / ACC; ++sg) {
                f32x16 part[C::MT][C::NT];
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
#pragma unroll
                for (int ss = 0; ss < ACC; ++ss) {
                    const int st = sg * ACC + ss, buf = st & 1;
                    store_b(buf);
                    __syncthreads();
                    next_loads(st);
                    if (wave_active && !(dbg & 2)) mma(st, buf, part);
                }
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j) acc[i][j] += part[i][j];
            }
        } else {
#pragma unroll 1
            for (int sg = 0; sg < C::NSTAGES / ACC; ++sg) {
                f32x16 part[C::MT][C::NT];
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
                // the group's first stage is peeled: its MFMAs take the literal 0 as C, so the partial sums are never
                // zeroed with 32 moves per group
                auto one = [&](int st) {
                    const int buf = st & 1;
                    store_b(buf);
                    __syncthreads();
                    next_loads(st);
                    if (wave_active && !(dbg & 2)) mma(st, buf, part);
                };
                one(sg * ACC);
#pragma unroll 1
                for (int ss = 1; ss < ACC; ++ss) one(sg * ACC + ss);
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j) acc[i][j] += part[i][j];
            }
        }
    }

    if constexpr (FUSED) {
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
            const int co = n0 + (wn * C::NT + j) * 32 + l31;
            const float sc = co < Cout ? aff_scale[co] : 0.f, sh = co < Cout ? aff_shift[co] : 0.f;
#pragma unroll
            for (int i = 0; i < C::MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] * sc + sh;
                    acc[i][j][r] = v > 0.f ? v : v * slope;
                }
        }
        if (pool != 0) {
            const bool is_max = pool == 1;
            auto op = [&](float a_, float b_) { return is_max ? fmaxf(a_, b_) : a_ + b_; };
            const int OD = D / 2, OH = H / 2, OW = W / 2;
            float* yb = z + (size_t)b * OD * OH * OW * Cout;
            // d pairs (in-lane) and w pairs (lane <-> lane + 32); q = r >> 2.  (A helper, not an array: an array of
            // these partials indexed through hsel-dependent selects ended up in scratch memory.)
            auto dw = [&](int i, int j, int dp, int q) {
                const float t = op(acc[i][j][2 * dp + 4 * q], acc[i][j][2 * dp + 1 + 4 * q]);
                return op(t, __shfl_xor(t, 32));
            };
            auto put = [&](int j, int php, int pwp, float v) {        // this half-wave stores pooled plane dp = hsel
                const int od = d0 / 2 + hsel, oh = h0 / 2 + php, ow = w0 / 2 + pwp;
                const int co = n0 + (wn * C::NT + j) * 32 + l31;
                if (od < OD && oh < OH && ow < OW && co < Cout)
                    yb[((size_t)(od * OH + oh) * OW + ow) * Cout + co] = is_max ? v : v * 0.125f;
            };
            if constexpr (C::TW == 4) {                               // ph = 2 T + (q >> 1), pw = 2 (q & 1) + hsel
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float v0 = op(dw(i, j, 0, q), dw(i, j, 0, q + 2)), v1 = op(dw(i, j, 1, q), dw(i, j, 1, q + 2));
                            put(j, wm * C::MT + i, q, hsel ? v1 : v0);
                        }
            } else if constexpr (C::MT == 2) {                        // ph = 2 wm + i, pw = 2 q + hsel
#pragma unroll
                for (int j = 0; j < C::NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = op(dw(0, j, 0, q), dw(1, j, 0, q)), v1 = op(dw(0, j, 1, q), dw(1, j, 1, q));
                        put(j, wm, q, hsel ? v1 : v0);
                    }
            } else {                                                  // ph = wm: the partner row is wave wm ^ 1
                __syncthreads();                                      // every wave is done with the halo
                float* ex = halo;
                const int slot = (((wm >> 1) * C::WN + wn) * C::NT) * 4;
#pragma unroll
                for (int j = 0; j < C::NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m0 = dw(0, j, 0, q), m1 = dw(0, j, 1, q);
                        const float mine = hsel ? m1 : m0;
                        if (wm & 1) ex[(slot + j * 4 + q) * 64 + lane] = mine;
                    }
                __syncthreads();
                if (!(wm & 1)) {
#pragma unroll
                    for (int j = 0; j < C::NT; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float m0 = dw(0, j, 0, q), m1 = dw(0, j, 1, q);
                            const float mine = hsel ? m1 : m0;
                            put(j, wm >> 1, q, op(mine, ex[(slot + j * 4 + q) * 64 + lane]));
                        }
                }
            }
            return;
        }
    }

    // ---- epilogue: NDHWC store (+ BatchNorm statistic partials) ----
    // C/D fragment: column = lane & 31 (output channel), row r -> voxel (r&3) + 8*(r>>2) + 4*(lane>>5)
    float s1[C::NT], s2[C::NT];
#pragma unroll
    for (int j = 0; j < C::NT; ++j) s1[j] = s2[j] = 0.f;
    float* zb = z + (size_t)b * D * H * W * Cout;
    // 32-bit offsets inside the sample (checked on the host); a brick that lies inside the volume with a full channel
    // tile takes the branch-free path (no per-voxel predicates around the stores)
    if constexpr (VEC && TMF_CONV_BUF) {
        // Buffer stores: the lane part of the address (channel, w parity, brick row) is one VGPR per (M-tile, w pair),
        // the plane d travels in the scalar offset, the second N-tile in the instruction offset; a lane outside the
        // volume or beyond Cout carries an out-of-range offset and its store is dropped by the hardware.
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(zb, 0, D * H * W * Cout * 4, 0x00020000);
        const int co0 = n0 + wn * C::NT * 32 + l31;
        const bool full = d0 + C::TD <= D && h0 + C::TH <= H && w0 + C::TW <= W && n0 + C::NB <= Cout;
        const int plane = H * W * Cout * 4;
        auto epilogue = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            f32x2 p1[C::NT], p2[C::NT];
#pragma unroll
            for (int j = 0; j < C::NT; ++j) p1[j] = p2[j] = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < C::MT; ++i) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int T = wm * C::MT + i, rest = 2 * q4 + hsel;                 // fragment row -> (d, w, h) as a_lane
                    const int pw = rest % C::TW, ph = T * (8 / C::TW) + rest / C::TW;
                    const int gh = h0 + ph, gw = w0 + pw;
                    const bool hw_ok = FULL || (gh < H && gw < W);
                    const int vo = ((gh * W + gw) * Cout + co0) * 4;
                    int voj[C::NT];
#pragma unroll
                    for (int j = 0; j < C::NT; ++j) voj[j] = (FULL || (hw_ok && co0 + j * 32 < Cout)) ? vo + j * 128 : OOB;
#pragma unroll
                    for (int pd = 0; pd < 4; pd += 2) {
                        const int r = q4 * 4 + pd;
                        const bool d_ok0 = FULL || d0 + pd < D, d_ok1 = FULL || d0 + pd + 1 < D;      // wave-uniform
#pragma unroll
                        for (int j = 0; j < C::NT; ++j) {
                            // (scalars, not elements of the pair: __builtin_bit_cast of v[1] reads v[0] with this compiler)
                            float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                            if (!(dbg & 4)) {
                                if (d_ok0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), zr, voj[j], (d0 + pd) * plane, 2);
                                if (d_ok1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), zr, voj[j], (d0 + pd + 1) * plane, 2);
                            }
                            if (!FULL) {
                                const bool lane_ok = voj[j] >= 0;
                                v0 = (lane_ok && d_ok0) ? v0 : 0.f;
                                v1 = (lane_ok && d_ok1) ? v1 : 0.f;
                            }
                            const f32x2 v = {v0, v1};
                            p1[j] += v;
                            p2[j] += v * v;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < C::NT; ++j) { s1[j] = p1[j][0] + p1[j][1]; s2[j] = p2[j][0] + p2[j][1]; }
        };
        if (full) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    } else {
        auto epilogue = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int i = 0; i < C::MT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int T = wm * C::MT + i, rest = 2 * (r >> 2) + hsel;            // fragment row -> (d, w, h) as a_lane
                    const int pd = r & 3, pw = rest % C::TW, ph = T * (8 / C::TW) + rest / C::TW;
                    const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
                    const bool pv = FULL || (gd < D && gh < H && gw < W);
                    const int off = ((gd * H + gh) * W + gw) * Cout;
#pragma unroll
                    for (int j = 0; j < C::NT; ++j) {
                        const int co = n0 + (wn * C::NT + j) * 32 + l31;
                        if (FULL || (pv && co < Cout)) {
                            const float v = acc[i][j][r];
                            if (!(dbg & 4)) __builtin_nontemporal_store(v, &zb[off + co]);
                            s1[j] += v;
                            s2[j] += v * v;
                        }
                    }
                }
            }
        };
        if (d0 + C::TD <= D && h0 + C::TH <= H && w0 + C::TW <= W && n0 + C::NB <= Cout) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    if (stat_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
        }
        // red[wave][j*32 + l31][2]
        if (hsel == 0) {
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                red[(wave * C::NT * 32 + j * 32 + l31) * 2 + 0] = s1[j];
                red[(wave * C::NT * 32 + j * 32 + l31) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < C::NB) {
            const int wn_t = tid / (C::NT * 32), col = tid % (C::NT * 32);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int m = 0; m < C::WM; ++m) {
                const int wv = m * C::WN + wn_t;
                a1 += red[(wv * C::NT * 32 + col) * 2 + 0];
                a2 += red[(wv * C::NT * 32 + col) * 2 + 1];
            }
            const int co = n0 + tid;
            if (co < Cout) {
                stat_partial[((size_t)tile * 2 + 0) * Cout + co] = a1;
                stat_partial[((size_t)tile * 2 + 1) * Cout + co] = a2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// forward / dgrad, REGISTER-TILED form (round 3): 6x6x12 bricks, 7 x NT tiles of 16x16 per wave
// ------------------------------------------------------------------------------------
// Why a second form.  (i) Balance: the pooled volumes are 27 * 2^k bricks of any power-of-two size, so a launch of the
// ring kernel above fills 27/32 of its last workgroup round at 24^3 and 12^3 whatever brick is chosen (0.76 / 0.70 of
// the peak in isolation against 0.86 at 48^3).  432-voxel bricks (6 x 6 x 12) divide 48^3, 24^3 and 12^3 into 2048 / 256 /
// 32 bricks per batch of 8 — multiples of the chip's 256 CUs — and a brick is 27 M-tiles of 16 voxels, 28 with one tile
// of padding = SEVEN per wave of a 4-wave workgroup: every SIMD of a CU carries the same number of tiles (3.6 % padding
// instead of 16 % idle).  (ii) Instructions per MFMA: beside fp32 MFMAs every LDS / vector / memory instruction of any
// wave on the SIMD is paid in matrix time; the ring kernel issues 2.6 of them per 64-cycle MFMA.  Here a
// wave holds a 7 x NT register tile of v_mfma_f32_16x16x4_f32 accumulators (112 x 16 NT outputs): one ds_read_b128 per
// M-tile and one per N-tile feed 4 k-steps = 28 NT MFMAs, 0.16 operand reads per 32-cycle MFMA at NT = 2, and staging a
// 16-channel chunk (halo 57 KB + 27 taps of weights) is amortised over 6 048 MFMAs per wave.
//   * K order inside a 16-channel chunk: lane group g = lane >> 4 owns channels 4 g .. 4 g + 3; k-step s of a tap takes
//     element s of every lane's 16-byte fragment (A: [g][halo position][4], B: [tap][g][co][4] in LDS) — any order is
//     right as long as A and B agree.
//   * weights arrive tap-major [tap][ci][co] (the ring kernel's pack): a staged 16-byte piece is 4 output channels of one
//     input channel and is written as four dwords into the [g][co][ci % 4] image (54 dword writes per thread and chunk
//     against 1 512 MFMAs per wave and chunk).
//   * two 4-wave workgroups per CU (70 KB of LDS each): 2 waves per SIMD, so the kernel may use 256 registers.
template <int NT_>
struct RtCfg {
    static constexpr int NT = NT_, NB = 16 * NT;
    static constexpr int BD = 6, BH = 6, BW = 12, NVOX = BD * BH * BW;              // 432 voxels
    static constexpr int HD = BD + 2, HH = BH + 2, HW = BW + 2, NPOS = HD * HH * HW; // 896 halo positions
    static constexpr int CINC = 16, NG = CINC / 4;
    static constexpr int NW = 4, NTHR = 64 * NW, MT = 7;                             // 28 M-tiles of 16 voxels
    static constexpr int TPS = 3, NSTAGES = 9;
    static constexpr int HALO_F = NG * NPOS * 4;                                     // floats
    static constexpr int WST_F = TPS * NG * NB * 4;                                  // floats per weight stage
    static constexpr int RED_F = NW * NB * 2;
    static constexpr size_t LDS_BYTES = (size_t)(HALO_F + 2 * WST_F + RED_F) * 4;
    static constexpr int HV = NPOS * NG / NTHR;                                      // 14 halo pieces per thread
    static constexpr int WPIECES = TPS * CINC * (NB / 4);                            // 16-byte pieces per weight stage
    static constexpr int WV = (WPIECES + NTHR - 1) / NTHR;                           // 2 | 1 per thread
    static_assert(NPOS * NG % NTHR == 0, "halo pieces divide evenly");
    static_assert(MT * NW * 16 >= NVOX, "tiles cover the brick");
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <class C>
__global__ __launch_bounds__(C::NTHR, 2) void conv3d_fwd_rt_kernel(
    const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z,
    float* __restrict__ stat_partial, int D, int H, int W, int Cin, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;
    float* Ws = smem + C::HALO_F;
    float* red = Ws + 2 * C::WST_F;
    constexpr int NT = C::NT, NB = C::NB, MT = C::MT, NPOS = C::NPOS, NG = C::NG;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;

    const int tile = xcd_contiguous(blockIdx.x, ntiles);
    int t = tile;
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; t /= tilesH;
    const int td = t % tilesD;
    const int b = t / tilesD;
    const int d0 = td * C::BD, h0 = th * C::BH, w0 = tw * C::BW;
    const int n0 = blockIdx.y * NB;

    // LDS float index of the fragment this lane feeds to its M-tile m at tap (0, 0, 0): voxel 16 (wave + 4 m) + i16 of the
    // brick (raster d, h, w), channel group g.  Voxels past the brick (the 28th tile) read voxel 431: results dropped.
    int a_lane[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int v = 16 * (wave + C::NW * m) + i16;
        v = v < C::NVOX ? v : C::NVOX - 1;
        const int pd = v / (C::BH * C::BW), ph = (v / C::BW) % C::BH, pw = v % C::BW;
        a_lane[m] = (g * NPOS + (pd * C::HH + ph) * C::HW + pw) * 4;
    }
    const int b_lane = (g * NB + i16) * 4;

    f32x4v acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[m][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const float* xb = x + (size_t)b * D * H * W * Cin;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, D * H * W * Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, C::TPS * C::NSTAGES * Cin * Cout * 4, 0x00020000);
    constexpr int OOB = (int)0x80000000u;

    // halo pieces: e = tid + 256 q -> channel group e & 3 (fastest: a position's 64 bytes are read by 4 neighbouring
    // lanes), position e >> 2.  Byte offset inside the sample once per brick; the chunk base travels in the scalar offset.
    int hoff[C::HV];
#pragma unroll
    for (int q = 0; q < C::HV; ++q) {
        const int e = tid + q * C::NTHR;
        const int hp = e >> 2, gg = e & 3;
        const int hw = hp % C::HW, hh = (hp / C::HW) % C::HH, hd = hp / (C::HW * C::HH);
        const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
        const bool ok = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
        hoff[q] = ok ? (((gd * H + gh) * W + gw) * Cin + gg * 4) * 4 : OOB;
    }
    // weight pieces of a 3-tap stage: e -> 4 output channels co4 .. co4 + 3 of (tap tp, input channel ci)
    int woff[C::WV], wlds[C::WV];
#pragma unroll
    for (int q = 0; q < C::WV; ++q) {
        const int e = tid + q * C::NTHR;
        const int co4 = (e % (NB / 4)) * 4, ci = (e / (NB / 4)) % C::CINC, tp = e / ((NB / 4) * C::CINC);
        const bool ok = e < C::WPIECES && n0 + co4 < Cout;
        woff[q] = ok ? ((tp * Cin + ci) * Cout + n0 + co4) * 4 : OOB;
        wlds[q] = ((tp * NG + (ci >> 2)) * NB + co4) * 4 + (ci & 3);
    }

    f32x4v wreg[2][C::WV];
    auto load_w = [&](int c0, int st, int slot) {
        const int sbase = (st * C::TPS * Cin + c0) * Cout * 4;
#pragma unroll
        for (int q = 0; q < C::WV; ++q)
            wreg[slot][q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, woff[q], sbase, 0));
    };
    auto store_w = [&](int buf, int slot) {
        float* dst = Ws + buf * C::WST_F;
#pragma unroll
        for (int q = 0; q < C::WV; ++q) {
            if (C::WPIECES % C::NTHR == 0 || tid + q * C::NTHR < C::WPIECES) {
#pragma unroll
                for (int u = 0; u < 4; ++u) dst[wlds[q] + 4 * u] = wreg[slot][q][u];
            }
        }
    };

    struct Frag { f32x4v a[MT]; f32x4v b[NT]; };
    // The halo of chunk c + 1 is requested into registers ahead of the LAST 56 MFMAs of chunk c (where one of the two fragment
    // sets is dead: the kernel stays under 256 registers) and written to LDS behind the barrier that ends the chunk: what
    // stays exposed per chunk is the rest of the load round trip, 14 LDS writes and a barrier.
    f32x4v hreg[C::HV];
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int q = 0; q < C::HV; ++q)
            hreg[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, hoff[q], c0 * 4, 0));
    };
    // AHEAD (one N-tile, 212 registers): the next chunk's halo and its first two weight stages are requested before the current
    // chunk ends.  With two N-tiles that costs 40 bytes of scratch and 2 % of speed: everything is requested at the chunk start.
    constexpr bool AHEAD = NT == 1;
    if (AHEAD) { load_halo(0); load_w(0, 0, 0); load_w(0, 1, 1); }
    for (int c0 = 0; c0 < Cin; c0 += C::CINC) {
        if (c0 > 0) __syncthreads();                       // the previous chunk's halo is read out
        if (!AHEAD) { load_halo(c0); load_w(c0, 0, 0); load_w(c0, 1, 1); }
        else if (c0 > 0) {
            // nine stages per chunk: the two weight slots change roles at every chunk boundary (both requests are a
            // stage or more old by now)
#pragma unroll
            for (int q = 0; q < C::WV; ++q) { const f32x4v tmp = wreg[0][q]; wreg[0][q] = wreg[1][q]; wreg[1][q] = tmp; }
        }
#pragma unroll
        for (int q = 0; q < C::HV; ++q) {
            const int e = tid + q * C::NTHR;
            *reinterpret_cast<f32x4v*>(halo + ((e & 3) * NPOS + (e >> 2)) * 4) = hreg[q];
        }
        const bool more = c0 + C::CINC < Cin;
        // Two fragment sets P, Q.  A stage multiplies taps kw = 0, 1, 2 from (P, Q, P); the A fragments of the NEXT stage's
        // first tap do not depend on the weight ring (the halo stays for the whole chunk), so they are requested into Q ahead
        // of the stage's last products, i.e. AHEAD of the barrier: behind it only the NT weight fragments are waited for.
        // The sets swap roles every stage (the stage loop is unrolled: all of this is static).
        Frag fr[2];
        auto load_a = [&](int st, int kw, Frag& f) {
            const int tapoff = (((st / 3) * C::HH + st % 3) * C::HW + kw) * 4;
#pragma unroll
            for (int m = 0; m < MT; ++m) f.a[m] = *reinterpret_cast<const f32x4v*>(halo + a_lane[m] + tapoff);
        };
        auto load_b = [&](int st, int kw, Frag& f) {
            const float* wsb = Ws + (st & 1) * C::WST_F + b_lane;
#pragma unroll
            for (int j = 0; j < NT; ++j) f.b[j] = *reinterpret_cast<const f32x4v*>(wsb + (kw * NG * NB + j * 16) * 4);
        };
        auto mul = [&](const Frag& f) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[m][s4], f.b[j][s4], acc[m][j], 0, 0, 0);
        };
#pragma unroll
        for (int st = 0; st < C::NSTAGES; ++st) {
            Frag& P = fr[st & 1];
            Frag& Q = fr[(st & 1) ^ 1];
            store_w(st & 1, st & 1);
            __syncthreads();
            // weight stages run two ahead, across the chunk boundary (stage 9 / 10 = stage 0 / 1 of the next chunk)
            if (st + 2 < C::NSTAGES) load_w(c0, st + 2, st & 1);
            else if (AHEAD && more) load_w(c0 + C::CINC, st + 2 - C::NSTAGES, st & 1);
            if (st == 0) load_a(st, 0, P);                  // (first stage of a chunk: the halo was written just now)
            load_b(st, 0, P);
            load_a(st, 1, Q);
            load_b(st, 1, Q);
            __builtin_amdgcn_sched_barrier(0);
            mul(P);
            load_a(st, 2, P);
            load_b(st, 2, P);
            __builtin_amdgcn_sched_barrier(0);
            mul(Q);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < C::NSTAGES) load_a(st + 1, 0, Q);  // next stage's first tap, ahead of the barrier
            if (AHEAD && st == C::NSTAGES - 1 && more) load_halo(c0 + C::CINC);      // (Q's registers are free from here on)
            mul(P);
        }
    }

    // ---- epilogue: NDHWC store + BatchNorm statistic partials ----
    // D fragment: column = lane & 15 (output channel), row 4 g + r -> voxel 16 (wave + 4 m) + 4 g + r of the brick
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    float* zb = z + (size_t)b * D * H * W * Cout;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = 16 * (wave + C::NW * m) + 4 * g + r;
            const int pd = v / (C::BH * C::BW), ph = (v / C::BW) % C::BH, pw = v % C::BW;
            const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
            const bool pv = v < C::NVOX && gd < D && gh < H && gw < W;
            const int off = ((gd * H + gh) * W + gw) * Cout;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = n0 + j * 16 + i16;
                if (pv && co < Cout) {
                    const float val = acc[m][j][r];
                    __builtin_nontemporal_store(val, &zb[off + co]);
                    s1[j] += val;
                    s2[j] += val * val;
                }
            }
        }
    }
    if (stat_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
        }
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                red[(wave * NB + j * 16 + i16) * 2 + 0] = s1[j];
                red[(wave * NB + j * 16 + i16) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < NB) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int q = 0; q < C::NW; ++q) { a1 += red[(q * NB + tid) * 2]; a2 += red[(q * NB + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < Cout) {
                stat_partial[((size_t)tile * 2 + 0) * Cout + co] = a1;
                stat_partial[((size_t)tile * 2 + 1) * Cout + co] = a2;
            }
        }
    }
}

template <int KS, int CINC> using CfgL32 = FwdCfg<KS, CINC, 2, 1, 4, 1, 4, 8, 8, (KS == 3 ? 3 : 1)>;
template <int KS, int CINC> using CfgL64 = FwdCfg<KS, CINC, 2, 2, 4, 1, 4, 8, 8, (KS == 3 ? 3 : 1)>;
template <int KS, int CINC> using CfgS128 = FwdCfg<KS, CINC, 1, 2, 2, 2, 4, 4, 4, 1>;
// 8-wave variants (two wavefronts per SIMD share the matrix pipe: one computes while the other waits on LDS)
template <int KS, int CINC> using CfgL32w8 = FwdCfg<KS, CINC, 1, 1, 8, 1, 4, 8, 8, (KS == 3 ? 3 : 1)>;
template <int KS, int CINC> using CfgL64w8 = FwdCfg<KS, CINC, 1, 2, 8, 1, 4, 8, 8, (KS == 3 ? 3 : 1)>;
template <int KS, int CINC> using CfgS128w8 = FwdCfg<KS, CINC, 1, 1, 2, 4, 4, 4, 4, 1>;
// 4x4x4 brick x 64 output channels, 4 waves, 48 KB of LDS: three workgroups per CU — for launches whose 128-channel
// workgroups would leave most of a round empty (plan_fwd)
template <int KS, int CINC> using CfgS64 = FwdCfg<KS, CINC, 1, 1, 2, 2, 4, 4, 4, 1>;
// 4x4x8 bricks, 4 waves, <= 78 KiB of LDS: TWO workgroups per CU whose phases (halo staging, weight-stage
// barriers, epilogue) are independent, so one computes while the other stages or stores
template <int KS, int CINC> using CfgM32 = FwdCfg<KS, CINC, 1, 1, 4, 1, 4, 4, 8, (KS == 3 ? 3 : 1)>;
template <int KS, int CINC> using CfgM64 = FwdCfg<KS, CINC, 1, 2, 4, 1, 4, 4, 8, 1>;

// tuning knob (tmf_set_option("conv_waves", v) or TMF_CONV_WAVES): forward/dgrad workgroup shape for the large
// layers: 4 = 4 waves, 1 workgroup per CU; 8 = 8 waves (two per SIMD); 2 = 4x4x8 bricks, two 4-wave workgroups
// per CU; 16 (default) = 8 waves with 16-channel chunks (77 KB of LDS): two 8-wave workgroups per CU, so one
// stages / stores while the other multiplies (measured +3..7 % over 8 on the 48^3 layers)
// tmf_set_option("conv_rt", 0 | 1 | 2) / TMF_CONV_RT: the register-tiled forward kernel never (default) / for the pooled
// volumes / wherever its bricks fit.  Off by default: in isolation it is 9 % faster than the ring kernel at 24^3 and 12^3
// (no 27/32 round), but the two encoder streams of a train step already fill those rounds with each other's kernels —
// model_ad steps at 14.75 ms either way; what has ONE stream gains (model_single, B = 16: 13.73 -> 13.59 ms; model_ad
// with TMF_STREAMS=1: 15.84 -> 15.37).  DESIGN.md 3.1.
int g_conv_rt = -1;
int conv_rt() {
    if (g_conv_rt < 0) {
        const char* e = getenv("TMF_CONV_RT");
        g_conv_rt = e == nullptr ? 0 : (atoi(e) == 2 ? 2 : (atoi(e) == 1 ? 1 : 0));
    }
    return g_conv_rt;
}
int g_debug = 0;          // timing ablations only (tmf_set_option("debug", bits)); results are garbage when set
int g_conv_waves = 0;
int conv_waves() {
    if (g_conv_waves == 0) {
        const char* e = getenv("TMF_CONV_WAVES");
        const int v = e ? atoi(e) : 0;
        g_conv_waves = (v == 4 || v == 2 || v == 8) ? v : 16;
    }
    return g_conv_waves;
}

struct FwdPlan {
    int cfg;      // 0 = L32, 1 = L64, 2 = S128  (+3: 8-wave variant); 6 = M32, 7 = M64
    int cinc;     // 8, 16, 32
    int tilesD, tilesH, tilesW, ntiles, nby;
};

FwdPlan plan_fwd(int B, int D, int H, int W, int cin, int cout, int ks, bool allow_rt = true, int rt_min = 0) {
    FwdPlan p;
    // Register-tiled form (cfg 12: 32 channels per workgroup, 13: 16): volumes its 6 x 6 x 12 bricks tile exactly, channel
    // counts in whole 16-channel chunks / tiles.  16-channel workgroups where 32-channel ones would leave CUs without one.
    // Taken for the pooled volumes (<= 24^3 voxels per sample: +9 % over the ring kernel at 24^3 and 12^3, B = 8, measured per
    // layer with tools/conv_ab.py --opt conv_rt); at 48^3 the ring kernel's last round is 3/8 full and cheap, and its
    // 0.81-0.86 stands against 0.79-0.81 here.  tmf_set_option("conv_rt", 2) takes this form wherever its bricks fit.
    const int rt_mode = conv_rt() > rt_min ? conv_rt() : rt_min;       // rt_min: the caller's per-call request (TMF_SNET_ALONE)
    const bool rt_size = rt_mode == 2 || (long)D * H * W <= 24L * 24 * 24;
    if (allow_rt && rt_mode && rt_size && ks == 3 && cin % 16 == 0 && cout % 16 == 0 && D % 6 == 0 && H % 6 == 0 && W % 12 == 0) {
        p.tilesD = D / 6; p.tilesH = H / 6; p.tilesW = W / 12;
        p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
        const bool nt2 = cout % 32 == 0 && (long)p.ntiles * (cout / 32) >= 512;
        p.cfg = nt2 ? 12 : 13;
        p.cinc = 16;
        p.nby = cout / (nt2 ? 32 : 16);
        return p;
    }
    p.cinc = cin <= 8 ? 8 : (cin <= 16 ? 16 : 32);
    const int mind = D < H ? (D < W ? D : W) : (H < W ? H : W);
    int td, th, tw, nb;
    if (mind >= 16 || (long)D * H * W >= 4096) {
        if (cout <= 32) { p.cfg = 0; nb = 32; } else { p.cfg = 1; nb = 64; }
        td = 4; th = 8; tw = 8;
    } else {
        p.cfg = 2; nb = 128; td = 4; th = 4; tw = 4;
    }
    // Equal workgroups spread over 256 CUs: a launch lasts as long as the CU with the most of them, ceil(n / 256) x the
    // work of one — 576 workgroups (the reference's 22x27x22 level in 4x8x8 bricks) cost 3 units where 432 (24^3) cost 2.
    // The half-size brick (4x4x8, four waves) costs ~8 % more per voxel but divides twice as finely and pads H to a
    // multiple of 4 instead of 8: 1 008 workgroups = 4 half units (conv3.0 at that size: 0.31 -> 0.22 ms; conv3.3 0.51 ->
    // 0.44, the model says 5 -> 4.3).  Chosen per launch by that model, the established brick on a tie; TMF_CONV_AUTO=0
    // keeps the 8-wave brick everywhere.
    static const bool auto_brick = [] { const char* e = getenv("TMF_CONV_AUTO"); return e == nullptr || atoi(e) != 0; }();
    bool half_brick = false;
    if (auto_brick && ks == 3 && p.cinc == 32 && conv_waves() == 16 && p.cfg < 2) {
        const long nby = tmf_cdiv(cout, nb);
        const long nL = (long)B * tmf_cdiv(D, 4) * tmf_cdiv(H, 8) * tmf_cdiv(W, 8) * nby;
        const long nM = (long)B * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 8) * nby;
        const double cL = (double)((nL + 255) / 256), cM = 0.54 * (double)((nM + 255) / 256);
        half_brick = cM < 0.9 * cL;        // (0.945 at 24^3, 64 -> 128, measured 2 % slower: co-resident workgroups do overlap a little)
    }
    if (auto_brick && ks == 3 && p.cinc == 32 && conv_waves() == 16 && p.cfg == 2 && cout % 64 == 0) {
        // small volumes: 128-channel workgroups against 64-channel ones of half the work (11x13x11, 128 -> 256: 576 vs
        // 1 152 workgroups = 3 vs 2.5 units, 0.34 -> 0.28 ms; 256 -> 128: 2 vs 1.5, 0.43 -> 0.33)
        const long n = (long)B * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 4);
        const double c128 = (double)((n * tmf_cdiv(cout, 128) + 255) / 256), c64 = 0.5 * (double)((n * (cout / 64) + 255) / 256);
        if (c64 < 0.9 * c128) { p.cfg = 11; nb = 64; }
    }
    if (half_brick) { p.cfg += 6; th = 4; }
    else if (p.cinc == 32 && conv_waves() == 16 && p.cfg < 2) { p.cfg += 8; p.cinc = 16; }
    else if (p.cinc == 32 && conv_waves() >= 8 && p.cfg < 11) p.cfg += 3;
    else if (p.cinc == 32 && conv_waves() == 2) {
        if (p.cfg == 2) p.cfg = 5;
        else { p.cfg += 6; th = 4; }
    }
    p.tilesD = tmf_cdiv(D, td); p.tilesH = tmf_cdiv(H, th); p.tilesW = tmf_cdiv(W, tw);
    p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
    p.nby = tmf_cdiv(cout, nb);
    return p;
}

struct Affine { const float* scale; const float* shift; float slope; int pool; };

template <class C>
int launch_fwd_cfg(const FwdPlan& p, const float* x, const float* w, float* z, float* sp,
                   int D, int H, int W, int cin, int cout, hipStream_t s, const Affine* aff) {
    const bool vec = (cin % 4 == 0) && (cout % 4 == 0);
    dim3 grid(p.ntiles, p.nby), block(C::NTHR);
    int rc;
    if (aff != nullptr) {
        TMF_REQUIRE(vec, TMF_E_SHAPE, "tmf_conv3d_fwd_affine: cin=%d and cout=%d must be multiples of 4", cin, cout);
        auto k = conv3d_fwd_kernel<C, true, true>;
        if ((rc = tmf_allow_lds(k, C::LDS_BYTES, "tmf_conv3d_fwd_affine"))) return rc;
        hipLaunchKernelGGL(k, grid, block, C::LDS_BYTES, s, x, w, z, (float*)nullptr, D, H, W, cin, cout,
                           p.tilesD, p.tilesH, p.tilesW, p.ntiles, 0, aff->scale, aff->shift, aff->slope, aff->pool);
        return tmf_launch_result("tmf_conv3d_fwd_affine");
    }
    if (vec) {
        auto k = conv3d_fwd_kernel<C, true, false>;
        if ((rc = tmf_allow_lds(k, C::LDS_BYTES, "tmf_conv3d_fwd"))) return rc;
        hipLaunchKernelGGL(k, grid, block, C::LDS_BYTES, s, x, w, z, sp, D, H, W, cin, cout,
                           p.tilesD, p.tilesH, p.tilesW, p.ntiles, g_debug, (const float*)nullptr, (const float*)nullptr, 0.f, 0);
    } else {
        auto k = conv3d_fwd_kernel<C, false, false>;
        if ((rc = tmf_allow_lds(k, C::LDS_BYTES, "tmf_conv3d_fwd"))) return rc;
        hipLaunchKernelGGL(k, grid, block, C::LDS_BYTES, s, x, w, z, sp, D, H, W, cin, cout,
                           p.tilesD, p.tilesH, p.tilesW, p.ntiles, g_debug, (const float*)nullptr, (const float*)nullptr, 0.f, 0);
    }
    return tmf_launch_result("tmf_conv3d_fwd");
}

template <class C>
int launch_fwd_rt(const FwdPlan& p, const float* x, const float* w, float* z, float* sp,
                  int D, int H, int W, int cin, int cout, hipStream_t s) {
    auto k = conv3d_fwd_rt_kernel<C>;
    int rc;
    if ((rc = tmf_allow_lds(k, C::LDS_BYTES, "tmf_conv3d_fwd"))) return rc;
    hipLaunchKernelGGL(k, dim3(p.ntiles, p.nby), dim3(C::NTHR), C::LDS_BYTES, s, x, w, z, sp, D, H, W, cin, cout,
                       p.tilesD, p.tilesH, p.tilesW, p.ntiles);
    return tmf_launch_result("tmf_conv3d_fwd");
}

template <int KS>
int launch_fwd(const FwdPlan& p, const float* x, const float* w, float* z, float* sp,
               int D, int H, int W, int cin, int cout, hipStream_t s, const Affine* aff = nullptr) {
    if (p.cfg == 12 || p.cfg == 13) {
        TMF_REQUIRE(aff == nullptr, TMF_E_ARG, "tmf_conv3d_fwd_affine: internal error: register-tiled plan");
        return p.cfg == 12 ? launch_fwd_rt<RtCfg<2>>(p, x, w, z, sp, D, H, W, cin, cout, s)
                           : launch_fwd_rt<RtCfg<1>>(p, x, w, z, sp, D, H, W, cin, cout, s);
    }
#define TMF_FWD_CASE(CFG, CINC)                                                                   \
    if (p.cinc == CINC) return launch_fwd_cfg<CFG<KS, CINC>>(p, x, w, z, sp, D, H, W, cin, cout, s, aff);
    if (p.cfg == 0) { TMF_FWD_CASE(CfgL32, 8) TMF_FWD_CASE(CfgL32, 16) TMF_FWD_CASE(CfgL32, 32) }
    if (p.cfg == 1) { TMF_FWD_CASE(CfgL64, 8) TMF_FWD_CASE(CfgL64, 16) TMF_FWD_CASE(CfgL64, 32) }
    if (p.cfg == 2) { TMF_FWD_CASE(CfgS128, 8) TMF_FWD_CASE(CfgS128, 16) TMF_FWD_CASE(CfgS128, 32) }
    if (p.cfg == 3) { TMF_FWD_CASE(CfgL32w8, 32) }
    if (p.cfg == 4) { TMF_FWD_CASE(CfgL64w8, 32) }
    if (p.cfg == 5) { TMF_FWD_CASE(CfgS128w8, 32) }
    if (p.cfg == 8) { TMF_FWD_CASE(CfgL32w8, 16) }     // 16-channel chunks: 77 KB of LDS -> two 8-wave workgroups per CU
    if (p.cfg == 9) { TMF_FWD_CASE(CfgL64w8, 16) }
    if (p.cfg == 6) { TMF_FWD_CASE(CfgM32, 32) }
    if (p.cfg == 7) { TMF_FWD_CASE(CfgM64, 32) }
    if (p.cfg == 11) { TMF_FWD_CASE(CfgS64, 32) }
#undef TMF_FWD_CASE
    tmf_set_error("tmf_conv3d_fwd: no kernel for plan cfg=%d cinc=%d", p.cfg, p.cinc);
    return TMF_E_SHAPE;
}

// Template-argument text of the kernel a plan selects, as rocprofv3 prints it ("FwdCfg<3, 16, 1, 1, 8, 1, 4, 8, 8, 3>"):
// bench.py groups its live per-launch timings by this name so that they can be checked against a kernel-trace.
template <int KS, int CINC, int MT, int NT, int WM, int WN, int TD, int TH, int TW, int TPS>
const char* cfg_name(const FwdCfg<KS, CINC, MT, NT, WM, WN, TD, TH, TW, TPS>*) {
    static thread_local char buf[96];
    snprintf(buf, sizeof buf, "FwdCfg<%d, %d, %d, %d, %d, %d, %d, %d, %d, %d>", KS, CINC, MT, NT, WM, WN, TD, TH, TW, TPS);
    return buf;
}
template <int KS>
const char* fwd_kernel_name(const FwdPlan& p) {
    if (p.cfg == 12) return "RtCfg<2>";
    if (p.cfg == 13) return "RtCfg<1>";
#define TMF_FWD_CASE(CFG, CINC) if (p.cinc == CINC) return cfg_name((const CFG<KS, CINC>*)nullptr);
    if (p.cfg == 0) { TMF_FWD_CASE(CfgL32, 8) TMF_FWD_CASE(CfgL32, 16) TMF_FWD_CASE(CfgL32, 32) }
    if (p.cfg == 1) { TMF_FWD_CASE(CfgL64, 8) TMF_FWD_CASE(CfgL64, 16) TMF_FWD_CASE(CfgL64, 32) }
    if (p.cfg == 2) { TMF_FWD_CASE(CfgS128, 8) TMF_FWD_CASE(CfgS128, 16) TMF_FWD_CASE(CfgS128, 32) }
    if (p.cfg == 3) { TMF_FWD_CASE(CfgL32w8, 32) }
    if (p.cfg == 4) { TMF_FWD_CASE(CfgL64w8, 32) }
    if (p.cfg == 5) { TMF_FWD_CASE(CfgS128w8, 32) }
    if (p.cfg == 8) { TMF_FWD_CASE(CfgL32w8, 16) }
    if (p.cfg == 9) { TMF_FWD_CASE(CfgL64w8, 16) }
    if (p.cfg == 6) { TMF_FWD_CASE(CfgM32, 32) }
    if (p.cfg == 7) { TMF_FWD_CASE(CfgM64, 32) }
    if (p.cfg == 11) { TMF_FWD_CASE(CfgS64, 32) }
#undef TMF_FWD_CASE
    return "?";
}

// ------------------------------------------------------------------------------------
// weight gradient, 3x3x3
// ------------------------------------------------------------------------------------
template <int NT_, int TD_, int TH_, int TW_, int NW_ = 4, int CI_ = 32>
struct WgCfg {
    static constexpr int NT = NT_, TD = TD_, TH = TH_, TW = TW_, NW = NW_;
    static constexpr int NTHR = 64 * NW;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int NHALO = HD * HH * HW;
    // input channels per workgroup.  32 = one MFMA M-tile per tap.  16 (layers with Cin <= 16) PACKS TWO TAPS into
    // one M-tile: rows 0-15 are tap 2p, rows 16-31 tap 2p + 1, each lane reading its own tap's shifted halo address —
    // 14 tile-taps instead of 27 half-empty ones.
    static constexpr int CI = CI_;
    static constexpr bool PACK = CI_ == 16;
    static constexpr int NB = 32 * NT;            // output channels per workgroup
    static constexpr int NPOS = TD * TH * TW;
    // accumulator tiles per wave: 4 waves x 7 taps, or 8 waves as 4 SIMD pairs of (4 + 3) -> 7 taps per SIMD either
    // way; packed: tap pairs p = wave, wave + 8 (p < 14) -> 4, 4, 3, 3 tiles per SIMD
    static constexpr int TPW = PACK ? 2 : (NW == 4 ? 7 : 4);
    static_assert(CI_ == 32 || (CI_ == 16 && NW_ == 8), "packed taps are an 8-wave configuration");
    static constexpr size_t LDS_BYTES = (size_t)(NHALO * CI + NPOS * NB) * 4;
    static_assert(TW % 2 == 0, "voxel pairs must not straddle a row");
};

template <class C, bool VEC>
__global__ __launch_bounds__(C::NTHR) void conv3d_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int ntiles,
    int tiles_per_split, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // dz rows first: their ds_read_b32 offsets then fit the 16-bit instruction field (behind the 77 KB halo every
    // read needed its own v_add, and vector-ALU instructions are paid in matrix time)
    float* dzs = smem;                      // [NPOS][NB]
    float* xh = smem + C::NPOS * C::NB;     // [NHALO][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hsel = lane >> 5;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * C::CI;
    const int co0 = blockIdx.z * C::NB;

    // This wave's taps.  Waves w and w+4 share a SIMD (dispatch order), each SIMD owns 7 consecutive taps.
    const int tap_base = C::NW == 4 ? 7 * wave : 7 * (wave & 3) + 4 * (wave >> 2);
    const int tap_cnt = C::PACK ? (wave < 6 ? 2 : 1) : C::NW == 4 ? 7 : (wave < 4 ? 4 : 3);
    // halo float offset of each tap (clamped; surplus taps are neither computed nor written)
    int tapoff[C::TPW];
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) {
        int tap = C::PACK ? 2 * (wave + 8 * t) + (l31 >> 4) : tap_base + t;
        tap = tap > 26 ? 26 : tap;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        tapoff[t] = ((kd * C::HH + kh) * C::HW + kw + hsel) * C::CI + (C::PACK ? (l31 & 15) : l31);
    }
    const int b_lane = hsel * C::NB + l31;

    f32x16 acc[C::TPW][C::NT];
#pragma unroll
    for (int t = 0; t < C::TPW; ++t)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > ntiles) tile_end = ntiles;

    // Software pipeline over bricks: the NEXT brick's halo travels HBM -> registers while the current brick is
    // being multiplied and is written to LDS after the compute phase (one barrier pair per brick); the dz rows
    // (shorter, and the register file is full of accumulators) are fetched in one batch at the brick boundary.
    constexpr int C4 = C::CI / 4;                  // float4 pieces per halo position
    constexpr int HV = (C::NHALO * C4 + C::NTHR - 1) / C::NTHR;
    constexpr int DV = (C::NPOS * (C::NB / 4) + C::NTHR - 1) / C::NTHR;
    f32x4 hreg[HV], dreg[DV];
    auto locate = [&](int tile, int& b, int& d0, int& h0, int& w0) {
        int tt = tile;
        const int tw = tt % tilesW; tt /= tilesW;
        const int th = tt % tilesH; tt /= tilesH;
        const int td = tt % tilesD;
        b = tt / tilesD;
        d0 = td * C::TD; h0 = th * C::TH; w0 = tw * C::TW;
    };
    // slots [0, HVP) of the halo are prefetched a whole brick ahead, the rest rides with the dz batch
    // (register budget: 2 waves per SIMD -> 256 VGPRs, half of them accumulators when NT == 2)
    constexpr int HVP = (C::NW == 8 && C::NT == 2) ? (HV > 3 ? HV - 3 : 0) : HV;
    // VEC path: buffer loads.  The per-lane byte offset of every piece RELATIVE to the brick's halo origin and its halo
    // coordinates (one byte each) are computed once per kernel; per brick only the scalar base moves.  A brick whose
    // halo lies inside the volume loads with no vector arithmetic at all; at the volume faces a piece is valid iff
    // lo <= (hd, hh, hw) <= hi per byte (5 vector instructions), and an invalid piece carries the offset 2^31 >=
    // num_records, which the hardware answers with zeros.
    constexpr int OOB = (int)0x80000000u;
    constexpr bool BUF = VEC && TMF_CONV_BUF;
    int hrel[BUF ? HV : 1], hcrd[BUF ? HV : 1], drel[BUF ? DV : 1], dcrd[BUF ? DV : 1];
    if constexpr (BUF) {
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            const int e = tid + q * C::NTHR;
            const int hp = e / C4, c = ci0 + (e % C4) * 4;
            const int hw = hp % C::HW, hh = (hp / C::HW) % C::HH, hd = hp / (C::HW * C::HH);
            hrel[q] = (e < C::NHALO * C4 && c < Cin) ? (((hd * H + hh) * W + hw) * Cin + c) * 4 : OOB;
            hcrd[q] = hd | hh << 8 | hw << 16;
        }
#pragma unroll
        for (int q = 0; q < DV; ++q) {
            const int e = tid + q * C::NTHR;
            const int p = e / (C::NB / 4), c = co0 + (e % (C::NB / 4)) * 4;
            const int pw = p % C::TW, ph = (p / C::TW) % C::TH, pd = p / (C::TW * C::TH);
            drel[q] = (e < C::NPOS * (C::NB / 4) && c < Cout) ? (((pd * H + ph) * W + pw) * Cout + c) * 4 : OOB;
            dcrd[q] = pd | ph << 8 | pw << 16;
        }
    }
    // valid iff every byte of crd lies in [lo, hi] (all bytes < 128): bit 7 of each byte of (crd + 0x808080 - lo) and of
    // ((hi | 0x808080) - crd) survives exactly when there is no borrow
    auto in_box = [](int crd, int lo_bias, int hi_bias) {
        const unsigned t = (unsigned)crd + (unsigned)lo_bias, u = (unsigned)hi_bias - (unsigned)crd;
        return ((t & u) | ~0x808080u) == 0xFFFFFFFFu;
    };
    auto min_i = [](int a_, int b_) { return a_ < b_ ? a_ : b_; };
    auto fetch_halo = [&](int tile, const int q0, const int q1) {
        int b, d0, h0, w0;
        locate(tile, b, d0, h0, w0);
        const float* xb = x + (size_t)b * D * H * W * Cin;
        if constexpr (BUF) {
            // origin = halo position (0, 0, 0); at a low face it points before the sample (never dereferenced there)
            const float* org = xb + ((long)((d0 - 1) * H + (h0 - 1)) * W + (w0 - 1)) * Cin;
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(org), 0, 0x7FFFFFFF, 0x00020000);
            const bool interior = d0 >= 1 && d0 + C::TD < D && h0 >= 1 && h0 + C::TH < H && w0 >= 1 && w0 + C::TW < W;
            if (interior) {
#pragma unroll
                for (int q = q0; q < q1; ++q)
                    hreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, hrel[q], 0, 0));
            } else {
                const int lo = (d0 == 0 ? 1 : 0) | (h0 == 0 ? 1 : 0) << 8 | (w0 == 0 ? 1 : 0) << 16;
                const int hi = min_i(C::HD - 1, D - d0) | min_i(C::HH - 1, H - h0) << 8 | min_i(C::HW - 1, W - w0) << 16;
                const int lo_bias = 0x808080 - lo, hi_bias = hi | 0x808080;
#pragma unroll
                for (int q = q0; q < q1; ++q) {
                    const int off = in_box(hcrd[q], lo_bias, hi_bias) ? hrel[q] : OOB;
                    hreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int e = tid + q * C::NTHR;
                const int hp = e / C4, c4 = e % C4;
                const int hw = hp % C::HW, hh = (hp / C::HW) % C::HH, hd = hp / (C::HW * C::HH);
                const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
                const int c = ci0 + c4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e < C::NHALO * C4 && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) {
                    const float* src = xb + ((size_t)(gd * H + gh) * W + gw) * Cin + c;
                    if (VEC) {
                        if (c < Cin) v = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < Cin) v[u] = src[u];
                    }
                }
                hreg[q] = v;
            }
        }
    };
    auto fetch_dz = [&](int tile) {
        int b, d0, h0, w0;
        locate(tile, b, d0, h0, w0);
        const float* dzb = dz + (size_t)b * D * H * W * Cout;
        if constexpr (BUF) {
            const float* org = dzb + ((long)(d0 * H + h0) * W + w0) * Cout;
            const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(org), 0, 0x7FFFFFFF, 0x00020000);
            if (d0 + C::TD <= D && h0 + C::TH <= H && w0 + C::TW <= W) {
#pragma unroll
                for (int q = 0; q < DV; ++q)
                    dreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, drel[q], 0, 0));
            } else {
                const int hi = min_i(C::TD, D - d0) - 1 | (min_i(C::TH, H - h0) - 1) << 8 | (min_i(C::TW, W - w0) - 1) << 16;
                const int hi_bias = hi | 0x808080;
#pragma unroll
                for (int q = 0; q < DV; ++q) {
                    const int off = in_box(dcrd[q], 0x808080, hi_bias) ? drel[q] : OOB;
                    dreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, off, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < DV; ++q) {
                const int e = tid + q * C::NTHR;
                const int p = e / (C::NB / 4), c4 = e % (C::NB / 4);
                const int pw = p % C::TW, ph = (p / C::TW) % C::TH, pd = p / (C::TW * C::TH);
                const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
                const int c = co0 + c4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e < C::NPOS * (C::NB / 4) && gd < D && gh < H && gw < W) {
                    const float* src = dzb + ((size_t)(gd * H + gh) * W + gw) * Cout + c;
                    if (VEC) {
                        if (c < Cout) v = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < Cout) v[u] = src[u];
                    }
                }
                dreg[q] = v;
            }
        }
    };

    constexpr bool DZ_AHEAD = C::NT == 1;      // 64 accumulator registers: room to carry the dz rows as well
    if (tile_begin < tile_end) {
        fetch_halo(tile_begin, 0, HVP);
        if (DZ_AHEAD) fetch_dz(tile_begin);
    }
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        if ((dbg & 1) && tile > tile_begin) goto compute;       // timing ablation: no staging after brick 0
        fetch_halo(tile, HVP, HV);
        if (!DZ_AHEAD) fetch_dz(tile);   // short-lived registers: issued here, landed by the time the halo is written
        __syncthreads();   // previous brick fully consumed
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            const int e = tid + q * C::NTHR;
            if (e < C::NHALO * C4) *reinterpret_cast<f32x4*>(&xh[e * 4]) = hreg[q];
        }
#pragma unroll
        for (int q = 0; q < DV; ++q) {
            const int e = tid + q * C::NTHR;
            if (e < C::NPOS * (C::NB / 4)) *reinterpret_cast<f32x4*>(&dzs[e * 4]) = dreg[q];
        }
        __syncthreads();
        if (tile + 1 < tile_end) {                               // in flight during the whole compute phase
            fetch_halo(tile + 1, 0, HVP);
            if (DZ_AHEAD) fetch_dz(tile + 1);
        }

    compute:
        if (dbg & 2) continue;                                  // timing ablation: staging only
        // K loop over voxel pairs: row = (pd, ph), pairs along w.  The tap count is a compile-time constant of
        // the (wave-uniform) role, so the loop body is branch-free: all LDS reads of a voxel pair are issued
        // before its MFMAs and the compiler pipelines them across pairs.
        auto mma_bricks = [&](auto ntaps_c) {
            constexpr int NTAPS = decltype(ntaps_c)::value;
            // one plane of the brick per iteration, its TH rows unrolled: the row and pair offsets of the x reads are
            // instruction offsets (ds_read2st64: 256-B units), so the only vector adds are one per tap and plane
#pragma unroll 1
            for (int pd = 0; pd < C::TD; ++pd) {
                const float* xp[NTAPS];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) xp[t] = xh + tapoff[t] + pd * C::HH * C::HW * C::CI;
                const float* bsrc = dzs + b_lane + pd * C::TH * C::TW * C::NB;
#pragma unroll
                for (int ph = 0; ph < C::TH; ++ph) {
#pragma unroll
                    for (int q = 0; q < C::TW / 2; ++q) {
                        float bv[C::NT], av[NTAPS];
#pragma unroll
                        for (int j = 0; j < C::NT; ++j) bv[j] = bsrc[(ph * C::TW + 2 * q) * C::NB + j * 32];
#pragma unroll
                        for (int t = 0; t < NTAPS; ++t) av[t] = xp[t][ph * C::HW * C::CI + 2 * q * C::CI];
#pragma unroll
                        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                            for (int j = 0; j < C::NT; ++j)
                                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[j], acc[t][j], 0, 0, 0);
                    }
                }
            }
        };
        if (C::PACK) {
            if (wave < 6) mma_bricks(std::integral_constant<int, 2>{});
            else mma_bricks(std::integral_constant<int, 1>{});
        }
        else if (C::NW == 4) mma_bricks(std::integral_constant<int, C::TPW>{});
        else if (wave < 4) mma_bricks(std::integral_constant<int, 4>{});
        else mma_bricks(std::integral_constant<int, 3>{});     // tap 27 (wave 7) is clamped: computed, not stored
    }

    // partial[split][tap][ci][co];  D fragment: row = ci, column = co
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) {
        const int tap0 = C::PACK ? 2 * (wave + 8 * t) : tap_base + t;
        if (t < tap_cnt && tap0 < 27) {
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                const int co = co0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;            // M-tile row (>= 16 iff r >= 8)
                    const int tap = C::PACK ? tap0 + (r >> 3) : tap0;
                    const int ci = ci0 + (C::PACK ? (m & 15) : m);
                    if (tap < 27 && ci < Cin && co < Cout)
                        partial[(((size_t)split * 27 + tap) * Cin + ci) * Cout + co] = acc[t][j][r];
                }
            }
        }
    }
}

// 1x1x1 weight gradient: dw[ci][co] = sum_pos x[pos][ci] dz[pos][co]; the 4 waves split the
// voxels of each 256-voxel slab and own one partial slab each (split index = block*4 + wave).
template <int NT>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    long npos, int Cin, int Cout, int slabs_per_block) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int ci = blockIdx.y * 32 + l31;
    const int co0 = blockIdx.z * 32 * NT;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long p_begin = ((long)blockIdx.x * slabs_per_block) * 256 + wave * 64;
    for (int sl = 0; sl < slabs_per_block; ++sl) {
        const long pb = p_begin + (long)sl * 256;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
            const long p = pb + 2 * q + hsel;
            const bool pv = p < npos;
            const float av = (pv && ci < Cin) ? x[p * Cin + ci] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = co0 + j * 32 + l31;
                const float bv = (pv && co < Cout) ? dz[p * Cout + co] : 0.f;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
    }
    const int split = blockIdx.x * 4 + wave;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
            if (cr < Cin && co < Cout) partial[((size_t)split * Cin + cr) * Cout + co] = acc[j][r];
        }
    }
}

struct WgPlan {
    int nt, tilesD, tilesH, tilesW, ntiles, tps, nsplit, gy, gz;
    int small;   // 1 = 4x4x4 bricks
    int pack;    // 1 = Cin <= 16: two taps per MFMA M-tile
};

WgPlan plan_wgrad(int B, int D, int H, int W, int cin, int cout) {
    WgPlan p;
    const int mind = D < H ? (D < W ? D : W) : (H < W ? H : W);
    p.small = !(mind >= 16 || (long)D * H * W >= 4096);
    const int td = 4, th = p.small ? 4 : 8, tw = p.small ? 4 : 8;
    // 8-wave kernels: one 32-channel output tile per workgroup (64 accumulator registers) leaves room for the
    // cross-brick halo prefetch; two tiles (128) spill.  Wider layers simply use more workgroup columns.
    const bool w8 = (cin % 4 == 0) && (cout % 4 == 0) && conv_waves() >= 8;
    p.nt = (cout <= 32 || w8) ? 1 : 2;
    p.tilesD = tmf_cdiv(D, td); p.tilesH = tmf_cdiv(H, th); p.tilesW = tmf_cdiv(W, tw);
    p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
    p.pack = w8 && cin <= 16;            // two taps per MFMA M-tile (WgCfg::PACK)
    p.gy = p.pack ? 1 : tmf_cdiv(cin, 32);
    p.gz = tmf_cdiv(cout, 32 * p.nt);
    const int groups = p.gy * p.gz;
    int want = 256 / groups;             // one workgroup per CU in total: one round, half the partial slabs of two
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tps = tmf_cdiv(p.ntiles, want);
    p.nsplit = tmf_cdiv(p.ntiles, p.tps);
    return p;
}

struct Wg1Plan { int nt, gy, gz, spb, nblk, nsplit; };
Wg1Plan plan_wgrad1(long npos, int cin, int cout) {
    Wg1Plan p;
    p.nt = cout <= 32 ? 1 : 2;
    p.gy = tmf_cdiv(cin, 32);
    p.gz = tmf_cdiv(cout, 32 * p.nt);
    const long slabs = (npos + 255) / 256;
    int want = 1024 / (p.gy * p.gz);
    if (want < 1) want = 1;
    if (want > slabs) want = (int)slabs;
    p.spb = (int)((slabs + want - 1) / want);
    p.nblk = (int)((slabs + p.spb - 1) / p.spb);
    p.nsplit = p.nblk * 4;
    return p;
}

// ------------------------------------------------------------------------------------
// first layer: cin == 1
// ------------------------------------------------------------------------------------
// z[pos][co] = sum_tap x[pos+tap-1] w[tap][co]:  A[i = voxel][k = tap] read from a scalar halo
// brick in LDS, B[k = tap][j = co] held in 14 registers (28 taps, the 28th is zero).
constexpr int C1_TD = 4, C1_TH = 8, C1_TW = 8;
constexpr int C1_HD = C1_TD + 2, C1_HH = C1_TH + 2, C1_HW = C1_TW + 2;
constexpr int C1_NHALO = C1_HD * C1_HH * C1_HW;

__device__ __forceinline__ constexpr int c1_tapoff(int tap) {
    return tap >= 27 ? 0 : ((tap / 9) * C1_HH + (tap / 3) % 3) * C1_HW + tap % 3;
}

__global__ __launch_bounds__(256) void conv3d_c1_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z,
    float* __restrict__ stat_partial, int D, int H, int W, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles) {
    __shared__ float halo[C1_NHALO];
    __shared__ float red[4 * 32 * 2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int tile = xcd_contiguous(blockIdx.x, ntiles);
    int t = tile;
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; t /= tilesH;
    const int td = t % tilesD;
    const int b = t / tilesD;
    const int d0 = td * C1_TD, h0 = th * C1_TH, w0 = tw * C1_TW;
    const int n0 = blockIdx.y * 32;
    const int co = n0 + l31;

    const float* xb = x + (size_t)b * D * H * W;
    for (int e = tid; e < C1_NHALO; e += 256) {
        const int hw = e % C1_HW, hh = (e / C1_HW) % C1_HH, hd = e / (C1_HW * C1_HH);
        const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
        float v = 0.f;
        if (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
            v = xb[(size_t)(gd * H + gh) * W + gw];
        halo[e] = v;
    }
    float bw[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = 2 * s + hsel;
        bw[s] = (tap < 27 && co < Cout) ? w[tap * Cout + co] : 0.f;
    }
    __syncthreads();

    f32x16 acc[2];
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int p = (wave * 2 + i) * 32 + l31;
        const int pw = p % C1_TW, ph = (p / C1_TW) % C1_TH, pd = p / (C1_TW * C1_TH);
        abase[i] = (pd * C1_HH + ph) * C1_HW + pw;
    }
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int off = hsel ? c1_tapoff(2 * s + 1) : c1_tapoff(2 * s);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(halo[abase[i] + off], bw[s], acc[i], 0, 0, 0);
    }

    float s1 = 0.f, s2 = 0.f;
    float* zb = z + (size_t)b * D * H * W * Cout;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = (wave * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
            const int pw = p % C1_TW, ph = (p / C1_TW) % C1_TH, pd = p / (C1_TW * C1_TH);
            const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
            if (gd < D && gh < H && gw < W && co < Cout) {
                const float v = acc[i][r];
                zb[((size_t)(gd * H + gh) * W + gw) * Cout + co] = v;
                s1 += v;
                s2 += v * v;
            }
        }
    }
    if (stat_partial != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hsel == 0) {
            red[(wave * 32 + l31) * 2 + 0] = s1;
            red[(wave * 32 + l31) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < 32 && n0 + tid < Cout) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a1 += red[(m * 32 + tid) * 2 + 0];
                a2 += red[(m * 32 + tid) * 2 + 1];
            }
            stat_partial[((size_t)tile * 2 + 0) * Cout + n0 + tid] = a1;
            stat_partial[((size_t)tile * 2 + 1) * Cout + n0 + tid] = a2;
        }
    }
}

// dw[tap][co] = sum_pos x[pos+tap-1] dz[pos][co]:  A[i = tap (27 of 32)][k = voxel], B[k = voxel][j = co].
// The 4 waves split the brick's voxels; partial[split*4 + wave][27][Cout].
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    int D, int H, int W, int Cout, int tilesD, int tilesH, int tilesW, int ntiles, int tiles_per_split) {
    constexpr int NPOS = C1_TD * C1_TH * C1_TW;
    __shared__ float halo[C1_NHALO];
    extern __shared__ __attribute__((aligned(16))) float dzs[];   // [NPOS][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int n0 = blockIdx.y * 32;
    const int tapo = c1_tapoff(l31) + hsel;   // lanes >= 27 read tap 0's voxels; their rows are dropped

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int tile_begin = blockIdx.x * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > ntiles) tile_end = ntiles;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        int t = tile;
        const int tw = t % tilesW; t /= tilesW;
        const int th = t % tilesH; t /= tilesH;
        const int td = t % tilesD;
        const int b = t / tilesD;
        const int d0 = td * C1_TD, h0 = th * C1_TH, w0 = tw * C1_TW;
        const float* xb = x + (size_t)b * D * H * W;
        const float* dzb = dz + (size_t)b * D * H * W * Cout;
        __syncthreads();
        for (int e = tid; e < C1_NHALO; e += 256) {
            const int hw = e % C1_HW, hh = (e / C1_HW) % C1_HH, hd = e / (C1_HW * C1_HH);
            const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
            float v = 0.f;
            if (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                v = xb[(size_t)(gd * H + gh) * W + gw];
            halo[e] = v;
        }
        for (int e = tid; e < NPOS * 32; e += 256) {
            const int p = e >> 5, c = e & 31;
            const int pw = p % C1_TW, ph = (p / C1_TW) % C1_TH, pd = p / (C1_TW * C1_TH);
            const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
            float v = 0.f;
            if (gd < D && gh < H && gw < W && n0 + c < Cout)
                v = dzb[((size_t)(gd * H + gh) * W + gw) * Cout + n0 + c];
            dzs[e] = v;
        }
        __syncthreads();
        // this wave's voxels: pd = wave (C1_TD == 4 waves), all (ph, pw)
#pragma unroll 4
        for (int ph = 0; ph < C1_TH; ++ph) {
#pragma unroll
            for (int q = 0; q < C1_TW / 2; ++q) {
                const int hp = (wave * C1_HH + ph) * C1_HW + 2 * q;
                const int p = (wave * C1_TH + ph) * C1_TW + 2 * q + hsel;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(halo[hp + tapo], dzs[p * 32 + l31], acc, 0, 0, 0);
            }
        }
    }
    const int split = blockIdx.x * 4 + wave;
    const int co = n0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tap = (r & 3) + 8 * (r >> 2) + 4 * hsel;
        if (tap < 27 && co < Cout) partial[((size_t)split * 27 + tap) * Cout + co] = acc[r];
    }
}

struct C1Plan { int tilesD, tilesH, tilesW, ntiles, nby, tps, nblk; };
C1Plan plan_c1(int B, int D, int H, int W, int cout) {
    C1Plan p;
    p.tilesD = tmf_cdiv(D, C1_TD); p.tilesH = tmf_cdiv(H, C1_TH); p.tilesW = tmf_cdiv(W, C1_TW);
    p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
    p.nby = tmf_cdiv(cout, 32);
    int want = 1024 / p.nby;
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tps = tmf_cdiv(p.ntiles, want);
    p.nblk = tmf_cdiv(p.ntiles, p.tps);
    return p;
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
static thread_local int t_algo_override = 0;
int tmf_algo_override(void) { return t_algo_override; }
void tmf_algo_override_set(int flags) { t_algo_override = (flags & TMF_SNET_ALGO) ? flags : 0; }
int tmf_c1_gram_mode(void);      // conv1_gram.hip
extern "C" int tmf_snet_algo_flags(void) {
    return TMF_SNET_ALGO | TMF_SNET_ALGO_WINO(tmf_conv_wino_mode()) | (tmf_wino_p_mode() ? TMF_SNET_ALGO_WINO_P : 0) |
           (tmf_wino_x_mode() ? TMF_SNET_ALGO_WINO_X : 0) | (tmf_c1_gram_mode() ? TMF_SNET_ALGO_C1_GRAM : 0) |
           (tmf_c1_gram_mode() == 2 ? TMF_SNET_ALGO_C1_GRAM_BF16 : 0) | (tmf_c1_split_mode() ? TMF_SNET_ALGO_C1_SPLIT : 0);
}

extern "C" int tmf_set_option(const char* name, int value) {
    TMF_REQUIRE_PTR(name);
    if (strcmp(name, "conv_waves") == 0) {
        TMF_REQUIRE(value == 4 || value == 8 || value == 2 || value == 16, TMF_E_ARG,
                    "tmf_set_option: conv_waves must be 2, 4, 8 or 16, got %d", value);
        g_conv_waves = value;
        return TMF_OK;
    }
    if (strcmp(name, "conv_rt") == 0) {
        TMF_REQUIRE(value >= 0 && value <= 2, TMF_E_ARG, "tmf_set_option: conv_rt must be 0, 1 or 2, got %d", value);
        g_conv_rt = value;
        return TMF_OK;
    }
    if (strcmp(name, "wino_p") == 0) return tmf_wino_p_set(value);
    if (strcmp(name, "wino_x") == 0) return tmf_wino_x_set(value);
    if (strcmp(name, "c1_gram") == 0) return tmf_c1_gram_set(value);
    if (strcmp(name, "c1_split") == 0) return tmf_c1_split_set(value);
    if (strcmp(name, "conv_wino") == 0) {
        TMF_REQUIRE(value >= 0 && value <= 3, TMF_E_ARG, "tmf_set_option: conv_wino must be 0, 1, 2 or 3, got %d", value);
        return tmf_conv_wino_set(value);
    }
    if (strcmp(name, "debug") == 0) { g_debug = value; tmf_g_debug = value; return TMF_OK; }
    if (strcmp(name, "bf16_v2") == 0) {
        TMF_REQUIRE(value >= 0 && value <= 2, TMF_E_ARG, "tmf_set_option: bf16_v2 must be 0, 1 or 2, got %d", value);
        tmf_g_bf16_v2 = value;
        return TMF_OK;
    }
    if (strcmp(name, "bf16_dma") == 0) {
        TMF_REQUIRE(value == 0 || value == 1, TMF_E_ARG, "tmf_set_option: bf16_dma must be 0 or 1, got %d", value);
        tmf_g_bf16_dma = value;
        return TMF_OK;
    }
    if (strcmp(name, "wgrad_tr") == 0) {
        TMF_REQUIRE(value >= 0 && value <= 2, TMF_E_ARG, "tmf_set_option: wgrad_tr must be 0, 1 or 2, got %d", value);
        tmf_g_wgrad_tr = value;
        return TMF_OK;
    }
    tmf_set_error("tmf_set_option: unknown option '%s'", name);
    return TMF_E_ARG;
}

extern "C" const char* tmf_conv3d_fwd_kernel_name(int B, int D, int H, int W, int cin, int cout, int ksize) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return "?";
    const FwdPlan p = plan_fwd(B, D, H, W, cin, cout, ksize);
    return ksize == 3 ? fwd_kernel_name<3>(p) : fwd_kernel_name<1>(p);
}

extern "C" const char* tmf_conv3d_wgrad_kernel_name(int B, int D, int H, int W, int cin, int cout, int ksize) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return "?";
    if (ksize != 3) return "conv1x1_wgrad_kernel";
    const WgPlan p = plan_wgrad(B, D, H, W, cin, cout);
    const bool vec = (cin % 4 == 0) && (cout % 4 == 0);
    const bool w8 = vec && conv_waves() >= 8;
    static thread_local char buf[64];
    const int edge = p.small ? 4 : 8;
    const int nt = (w8 && p.pack) ? 1 : p.nt;
    snprintf(buf, sizeof buf, "WgCfg<%d, 4, %d, %d, %d, %d>", nt, edge, edge, w8 ? 8 : 4, (w8 && p.pack) ? 16 : 32);
    return buf;
}

extern "C" int tmf_conv3d_stat_blocks(int B, int D, int H, int W, int cin, int cout, int ksize) {
    return tmf_conv3d_stat_blocks_mode(B, D, H, W, cin, cout, ksize, 0);
}
// (internal, tmf_common.h) rt_min = 1: at least tmf_set_option("conv_rt", 1) for this call — the one-call encoder with TMF_SNET_ALONE
int tmf_conv3d_stat_blocks_mode(int B, int D, int H, int W, int cin, int cout, int ksize, int rt_min) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
    return plan_fwd(B, D, H, W, cin, cout, ksize, true, rt_min).ntiles;
}

extern "C" int tmf_conv3d_fwd(const float* x, const float* w, float* z, float* stat_partial,
                              int B, int D, int H, int W, int cin, int cout, int ksize, void* stream) {
    return tmf_conv3d_fwd_mode(x, w, z, stat_partial, B, D, H, W, cin, cout, ksize, 0, stream);
}
int tmf_conv3d_fwd_mode(const float* x, const float* w, float* z, float* stat_partial,
                        int B, int D, int H, int W, int cin, int cout, int ksize, int rt_min, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(z);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_fwd: non-positive dimension (B=%d D=%d H=%d W=%d cin=%d cout=%d)", B, D, H, W, cin, cout);
    TMF_REQUIRE(ksize == 1 || ksize == 3, TMF_E_ARG, "tmf_conv3d_fwd: ksize must be 1 or 3, got %d", ksize);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_fwd: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE((long)ksize * ksize * ksize * cin * cout < (1L << 29), TMF_E_SHAPE, "tmf_conv3d_fwd: weight tensor exceeds 2^29 elements");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(w); TMF_REQUIRE_ALIGNED(z);
    const FwdPlan p = plan_fwd(B, D, H, W, cin, cout, ksize, true, rt_min);
    hipStream_t s = (hipStream_t)stream;
    return ksize == 3 ? launch_fwd<3>(p, x, w, z, stat_partial, D, H, W, cin, cout, s)
                      : launch_fwd<1>(p, x, w, z, stat_partial, D, H, W, cin, cout, s);
}

extern "C" int tmf_conv3d_fwd_affine(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                     int B, int D, int H, int W, int cin, int cout, int ksize, int pool, float slope,
                                     void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(y);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_fwd_affine: non-positive dimension (B=%d D=%d H=%d W=%d cin=%d cout=%d)", B, D, H, W, cin, cout);
    TMF_REQUIRE(ksize == 1 || ksize == 3, TMF_E_ARG, "tmf_conv3d_fwd_affine: ksize must be 1 or 3, got %d", ksize);
    TMF_REQUIRE(pool == TMF_POOL_NONE || pool == TMF_POOL_MAX2 || pool == TMF_POOL_AVG2, TMF_E_ARG,
                "tmf_conv3d_fwd_affine: unknown pool mode %d", pool);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_fwd_affine: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE((long)ksize * ksize * ksize * cin * cout < (1L << 29), TMF_E_SHAPE, "tmf_conv3d_fwd_affine: weight tensor exceeds 2^29 elements");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(w); TMF_REQUIRE_ALIGNED(y);
    if (pool != TMF_POOL_NONE && (D / 2 == 0 || H / 2 == 0 || W / 2 == 0)) return TMF_OK;      // empty output
    const FwdPlan p = plan_fwd(B, D, H, W, cin, cout, ksize, false);      // (the fused epilogue lives in the ring kernel)
    const Affine aff = {scale, shift, slope, pool};
    hipStream_t s = (hipStream_t)stream;
    return ksize == 3 ? launch_fwd<3>(p, x, w, y, nullptr, D, H, W, cin, cout, s, &aff)
                      : launch_fwd<1>(p, x, w, y, nullptr, D, H, W, cin, cout, s, &aff);
}

extern "C" size_t tmf_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int cin, int cout, int ksize) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
    if (ksize == 3) {
        const WgPlan p = plan_wgrad(B, D, H, W, cin, cout);
        return (size_t)(p.nsplit + tmf_reduce_groups(p.nsplit)) * 27 * cin * cout * 4;
    }
    const Wg1Plan p = plan_wgrad1((long)B * D * H * W, cin, cout);
    return (size_t)(p.nsplit + tmf_reduce_groups(p.nsplit)) * cin * cout * 4;
}

extern "C" int tmf_conv3d_wgrad(const float* x, const float* dz, float* dw, void* workspace,
                                size_t workspace_bytes, int B, int D, int H, int W, int cin, int cout,
                                int ksize, int dw_layout, void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG,
                "tmf_conv3d_wgrad: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(dz); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_wgrad: non-positive dimension");
    TMF_REQUIRE(ksize == 1 || ksize == 3, TMF_E_ARG, "tmf_conv3d_wgrad: ksize must be 1 or 3, got %d", ksize);
    TMF_REQUIRE((long)(D > 6 ? D : 6) * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_wgrad: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(dz); TMF_REQUIRE_ALIGNED(dw); TMF_REQUIRE_ALIGNED(workspace);
    const size_t need = tmf_conv3d_wgrad_workspace_bytes(B, D, H, W, cin, cout, ksize);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_conv3d_wgrad: workspace %zu B < required %zu B",
                workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const bool vec = (cin % 4 == 0) && (cout % 4 == 0);
    int rc;
    long nelem;
    int nsplit;
    if (ksize == 3) {
        const WgPlan p = plan_wgrad(B, D, H, W, cin, cout);
        dim3 grid(p.nsplit, p.gy, p.gz);
#define TMF_WG_LAUNCH(CFG, V)                                                                       \
    do {                                                                                            \
        auto k = conv3d_wgrad_kernel<CFG, V>;                                                       \
        if ((rc = tmf_allow_lds(k, CFG::LDS_BYTES, "tmf_conv3d_wgrad"))) return rc;                 \
        hipLaunchKernelGGL(k, grid, dim3(CFG::NTHR), CFG::LDS_BYTES, s, x, dz, partial, D, H, W, cin, cout,   \
                           p.tilesD, p.tilesH, p.tilesW, p.ntiles, p.tps, g_debug);                 \
    } while (0)
        using L1 = WgCfg<1, 4, 8, 8>;
        using L2 = WgCfg<2, 4, 8, 8>;
        using S1 = WgCfg<1, 4, 4, 4>;
        using S2 = WgCfg<2, 4, 4, 4>;
        using L1w8 = WgCfg<1, 4, 8, 8, 8>;
        using L2w8 = WgCfg<2, 4, 8, 8, 8>;
        using S1w8 = WgCfg<1, 4, 4, 4, 8>;
        using S2w8 = WgCfg<2, 4, 4, 4, 8>;
        using L1w8p = WgCfg<1, 4, 8, 8, 8, 16>;
        using S1w8p = WgCfg<1, 4, 4, 4, 8, 16>;
        const bool w8 = vec && conv_waves() >= 8;
        if (w8 && p.pack) {
            if (!p.small) TMF_WG_LAUNCH(L1w8p, true);
            else          TMF_WG_LAUNCH(S1w8p, true);
        } else if (w8) {
            if (!p.small && p.nt == 1) TMF_WG_LAUNCH(L1w8, true);
            else if (!p.small)         TMF_WG_LAUNCH(L2w8, true);
            else if (p.nt == 1)        TMF_WG_LAUNCH(S1w8, true);
            else                       TMF_WG_LAUNCH(S2w8, true);
        }
        else if (!p.small && p.nt == 1) { if (vec) TMF_WG_LAUNCH(L1, true); else TMF_WG_LAUNCH(L1, false); }
        else if (!p.small)         { if (vec) TMF_WG_LAUNCH(L2, true); else TMF_WG_LAUNCH(L2, false); }
        else if (p.nt == 1)        { if (vec) TMF_WG_LAUNCH(S1, true); else TMF_WG_LAUNCH(S1, false); }
        else                       { if (vec) TMF_WG_LAUNCH(S2, true); else TMF_WG_LAUNCH(S2, false); }
#undef TMF_WG_LAUNCH
        if ((rc = tmf_launch_result("tmf_conv3d_wgrad"))) return rc;
        nelem = 27L * cin * cout;
        nsplit = p.nsplit;
    } else {
        const long npos = (long)B * D * H * W;
        const Wg1Plan p = plan_wgrad1(npos, cin, cout);
        dim3 grid(p.nblk, p.gy, p.gz), block(256);
        if (p.nt == 1) hipLaunchKernelGGL(conv1x1_wgrad_kernel<1>, grid, block, 0, s, x, dz, partial, npos, cin, cout, p.spb);
        else           hipLaunchKernelGGL(conv1x1_wgrad_kernel<2>, grid, block, 0, s, x, dz, partial, npos, cin, cout, p.spb);
        if ((rc = tmf_launch_result("tmf_conv3d_wgrad(1x1)"))) return rc;
        nelem = (long)cin * cout;
        nsplit = p.nsplit;
    }
    return tmf_reduce_slabs(partial, nsplit, nelem, partial + (size_t)nsplit * nelem, dw, s, "tmf_conv3d_wgrad(reduce)",
                            dw_layout == TMF_DW_REFERENCE ? cin : 0, cout);
}

extern "C" int tmf_conv3d_c1_stat_blocks(int B, int D, int H, int W, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cout <= 0) return 0;
    return plan_c1(B, D, H, W, cout).ntiles;
}

extern "C" int tmf_conv3d_c1_fwd(const float* x, const float* w, float* z, float* stat_partial,
                                 int B, int D, int H, int W, int cout, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(z);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cout > 0, TMF_E_SHAPE, "tmf_conv3d_c1_fwd: non-positive dimension");
    TMF_REQUIRE((long)D * H * W * cout < (1L << 31), TMF_E_SHAPE, "tmf_conv3d_c1_fwd: one sample exceeds 2^31 elements");
    const C1Plan p = plan_c1(B, D, H, W, cout);
    hipLaunchKernelGGL(conv3d_c1_fwd_kernel, dim3(p.ntiles, p.nby), dim3(256), 0, (hipStream_t)stream,
                       x, w, z, stat_partial, D, H, W, cout, p.tilesD, p.tilesH, p.tilesW, p.ntiles);
    return tmf_launch_result("tmf_conv3d_c1_fwd");
}

extern "C" size_t tmf_conv3d_c1_wgrad_workspace_bytes(int B, int D, int H, int W, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cout <= 0) return 0;
    const C1Plan p = plan_c1(B, D, H, W, cout);
    return (size_t)(p.nblk * 4 + tmf_reduce_groups(p.nblk * 4)) * 27 * cout * 4;
}

extern "C" int tmf_conv3d_c1_wgrad(const float* x, const float* dz, float* dw, void* workspace,
                                   size_t workspace_bytes, int B, int D, int H, int W, int cout, int dw_layout,
                                   void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG,
                "tmf_conv3d_c1_wgrad: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(dz); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cout > 0, TMF_E_SHAPE, "tmf_conv3d_c1_wgrad: non-positive dimension");
    const size_t need = tmf_conv3d_c1_wgrad_workspace_bytes(B, D, H, W, cout);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_conv3d_c1_wgrad: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const C1Plan p = plan_c1(B, D, H, W, cout);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const size_t lds = (size_t)C1_TD * C1_TH * C1_TW * 32 * 4;
    hipLaunchKernelGGL(conv3d_c1_wgrad_kernel, dim3(p.nblk, p.nby), dim3(256), lds, s, x, dz, partial,
                       D, H, W, cout, p.tilesD, p.tilesH, p.tilesW, p.ntiles, p.tps);
    int rc;
    if ((rc = tmf_launch_result("tmf_conv3d_c1_wgrad"))) return rc;
    const long nelem = 27L * cout;
    return tmf_reduce_slabs(partial, p.nblk * 4, nelem, partial + (size_t)p.nblk * 4 * nelem, dw, s,
                            "tmf_conv3d_c1_wgrad(reduce)", dw_layout == TMF_DW_REFERENCE ? 1 : 0, cout);
}
