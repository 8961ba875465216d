// conv1_fused.hip — the first sNet block (Conv3d(1->C, 3x3x3) -> BatchNorm3d -> LeakyReLU -> MaxPool3d(2))
// WITHOUT ever materialising the conv output in HBM.  gfx950.
//
// At B = 8, 96^3 the raw output z of this layer is 906 MB per stream and, stored, it is read or written six
// times per training step (the layer is HBM-bound, AI = 13 flop/B).  The input is only 28 MB
// and the 27-tap convolution costs 14 MFMAs per 32 voxels x 32 channels, so every pass RECOMPUTES z in
// registers from an LDS halo brick instead:
//   stats   : z -> per-workgroup sum / sum-of-squares partials                    (reads x)
//   forward : z -> scale/shift -> LeakyReLU -> 2x2x2 max -> pooled output         (reads x, writes P)
//   reduce  : z, dP -> dy (first-maximum routing, LeakyReLU mask) -> sum dy, sum dy*xhat partials
//   wgrad   : z, dP -> dz = scale (dy - c0 - xhat c1) -> dw[tap][c] += x[voxel+tap] dz[voxel][c]  (MFMA)
// z is bit-identical in all four (same instruction sequence), so the statistics, the forward activations and
// the recomputed backward masks are mutually consistent.
//
// Fragment trick: the 32 voxels of an MFMA M-tile are a 2x4x4 block ordered so that the accumulator registers
// of ONE lane hold two complete 2x2x2 pooling windows of one channel (row bits: w0,h0 | lane half = h1 | d0,w1):
// pooling, its argmax and the BN algebra are per-lane register work, and in the wgrad pass the dz registers are
// directly the MFMA B operand (k = lane half) — no LDS round trip, no shuffles.
//
// Replaces, for networks.py:21-26 (sNet.conv1): aten::conv3d, batch_norm, leaky_relu, max_pool3d and their
// backward (weight gradient only: the network input needs no gradient, kfold_train_adversarial.py:106).
#include <type_traits>
#include "tmf_common.h"

namespace {

#ifndef TMF_C1_TD
#define TMF_C1_TD 4
#endif
constexpr int TD = TMF_C1_TD, TH = 8, TW = 8;
constexpr int NTI = TD / 2;                           // M-tiles per wave and brick (4 waves, TD * 2 tiles of 32 voxels)
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int NHALO = HD * HH * HW;

enum { MODE_STATS = 0, MODE_FWD = 1, MODE_REDUCE = 2, MODE_WGRAD = 3, MODE_RD = 4 };
#ifndef TMF_C1X_ABL
#define TMF_C1X_ABL 0              // timing ablations of the SPLIT forward (wrong results): 1 = no stores, 2 = no MFMAs
#endif
#ifndef TMF_C1X_LAZY
#define TMF_C1X_LAZY 0x12          // bit MODE: the SPLIT form of that pass computes one M-tile's z at a time — the one-pass backward
                                   // needs it to stay below 256 registers (226), the forward drops to 128 (4 waves per SIMD:
                                   // 111.6 -> 105.0 us at B = 8, 96^3; requesting the next image's rows ahead of the MFMAs: no gain)
#endif
// MODE_RD (round 5): MODE_REDUCE plus D[tap][c] = sum_voxel x[voxel + tap] dy[voxel][c] — dy is ONE element per pooling window, so D is 27
// reads and multiply-adds per window; with the tap Gram matrix of the forward (conv1_gram.hip) the weight gradient follows without
// a second pass (tmf_c1_bwd_fused).

// bf16 passes: the halo brick lives in LDS as bf16, TWICE — copy c stores element e at index e + c — so that the pair
// (x[w], x[w + 1]) is one aligned dword for every w (even w: copy 0, odd w: copy 1).  A lane then fetches two taps per
// ds_read_b32 with no conversion: 9 reads per M-tile instead of 16 fp32 reads + 8 packs.  With the fp32 halo these passes
// were bound by the LDS port (PMC: half of all LDS cycles bank conflicts; no gather at all: 99 -> 37 us, stats, 128^3).
// Pitches (dwords) are chosen so that the five lane bits of a fragment row land on five different address bits:
//   w0 -> copy offset + 1 = 2 (mod 32), w1 -> 1, h0 -> 8, h1 -> 16, d0 -> 100 = 4 (mod 32): conflict-free ds_read_b32.
constexpr int BROW = 16, BPLANE = 200;                          // elements: row, plane
constexpr int BCP_DW = (HD * BPLANE / 2 + 1 + 31) / 32 * 32 + 1;    // copy pitch in dwords: = 1 (mod 32); TD = 4: 609 = 19 * 32 + 1
constexpr int BCOPY = 2 * BCP_DW;                               // ... in elements
constexpr int NHB_DW = 2 * BCP_DW;                              // dwords of LDS for both copies (copy 1 ends at 609 + 600 + 1)
static_assert(HD * BPLANE / 2 + 1 <= BCP_DW && HH * BROW <= BPLANE && HW + 2 <= BROW, "bf16 halo layout");
__device__ __forceinline__ constexpr int brow_off(int r) { return (r / 3) * BPLANE + (r % 3) * BROW; }   // tap row r = 3 dz + dy

__device__ __forceinline__ constexpr int tapoff(int tap) {
    return tap >= 27 ? 0 : ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3;
}
// halo index (tap (0,0,0) corner) of M-tile t's origin, and of fragment row r (lane half 0) relative to it
__device__ __forceinline__ constexpr int tile_org(int t) { return ((2 * (t >> 2)) * HH + 4 * ((t >> 1) & 1)) * HW + 4 * (t & 1); }
__device__ __forceinline__ constexpr int row_off(int r) {   // r bits: b0 -> w0, b1 -> h0, b2 -> d0, b3 -> w1
    return (((r >> 2) & 1) * HH + ((r >> 1) & 1)) * HW + 2 * ((r >> 3) & 1) + (r & 1);
}

struct Args {
    const float* x;        // [B][D][H][W]
    const float* w;        // [27][C]
    const float* scale;    // [C]   (FWD, REDUCE, WGRAD)
    const float* shift;
    const float* mean;     // (REDUCE, WGRAD)
    const float* invstd;
    const float* coef;     // [2][C] (WGRAD)
    const void* dpool;     // [B][D/2][H/2][W/2][C] (REDUCE, WGRAD); float, or bf16 when P16
    void* pooled;          // (FWD)
    float* partial;        // STATS/REDUCE/RD: [nblk][2][C];  WGRAD: [nblk][27][C]
    float* partial2;       // RD: [nblk][27][C]
    int D, H, W, C;
    int tilesD, tilesH, tilesW, ntiles, tiles_per_block;
    float slope;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tmf_bf16x8 pack8(const float (&v)[8]) {
    const u32x4 p = {tmf_pack_bf16(v[0], v[1]), tmf_pack_bf16(v[2], v[3]), tmf_pack_bf16(v[4], v[5]), tmf_pack_bf16(v[6], v[7])};
    return __builtin_bit_cast(tmf_bf16x8, p);
}

// x = h + m + l exactly, each part a bf16 number (low 16 bits zero): h / m by truncation, l = the remaining <= 8 bits
__device__ __forceinline__ void split3(float x, float& h, float& m, float& l) {
    h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
    const float r = x - h;
    m = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r) & 0xFFFF0000u);
    l = r - m;
}

// BF16 = true: both products run on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16, fp32 accumulation) — the
// 27-tap convolution is 2 MFMAs instead of 14 and the tap-gradient product 2 instead of 16, which turns the four
// passes from matrix-bound into LDS / HBM-bound (the opt-in bf16 mode of BASELINE configs[2]).
//
// SPLIT = true (round 6, fp32 mode): z of the fp32 precision on the bf16 matrix pipe — the volume and the taps as EXACT sums of three
// bf16 parts each (h = the top 8 mantissa bits, m = the next 8, l = the rest; three bf16 halo images in the layout above), z = the six
// partial products wh xl + wl xh + wm xm + wh xm + wm xh + wh xh, accumulated in fp32 from the small terms up: 18 MFMAs of 32 cycles per
// 32 voxels x 32 channels instead of 14 of 64, the dropped products are below 2^-24 of |w||x| (conv3d_winox.hip has the same
// arithmetic), and — what counts as much — the bf16 MFMAs do not share the vector ALU's issue port, so the BatchNorm / pooling /
// routing arithmetic of these passes runs beside them instead of in between.  The fp32 halo stays next to the images for the passes
// that multiply the inputs themselves (D of MODE_RD, the tap-gradient product of MODE_WGRAD: fp32 as before).
template <int MODE, bool BF16, bool P16 = false, bool SPLIT = false>     // P16: pooled / dpool are bf16 tensors (bf16 activation storage)
__global__ __launch_bounds__(256, MODE == 3 && !SPLIT ? 4 : 2) void conv1_fused_kernel(Args a) {   // <= 256 registers: MFMA results in VGPRs (no v_accvgpr_read copies)
    static_assert(!(BF16 && SPLIT), "SPLIT is the fp32 mode's variant");
    constexpr bool B16L = BF16 || SPLIT;                    // bf16 halo image(s) in LDS
    constexpr int NIMG = SPLIT ? 3 : 1;
    constexpr bool KEEP32 = SPLIT && (MODE == MODE_RD || MODE == MODE_WGRAD);
    __shared__ float halo[B16L ? NIMG * NHB_DW : NHALO];
    __shared__ float halo32[KEEP32 ? NHALO : 1];
    const unsigned hb_base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)halo;   // LDS byte address of the bf16 copies (BF16)
    __shared__ float red[4 * 32 * 32];      // cross-wave reduction scratch (16 KB)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int n0 = blockIdx.y * 32;
    const int co = n0 + l31;
    const bool cv = co < a.C;
    const int OD = a.D / 2, OH = a.H / 2, OW = a.W / 2;

    float bw[B16L ? 1 : 14];
    // BF16: K = 48 = 3 MFMAs x (2 lane halves x 4 tap rows x 2 taps): k = 16 m + 8 hsel + 2 s + t is tap row r = 4 m + s
    // (= 3 dz + dy), dx = 2 hsel + t — the lane half picks the pair (dx 0, 1) or (dx 2, pad); rows >= 9 and dx = 3 are zero
    tmf_bf16x8 bwb[NIMG][3];                // [part h / m / l][MFMA]
    if (B16L) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 4 * m + (j >> 1), dx = 2 * hsel + (j & 1);
                v[j] = (r < 9 && dx < 3 && cv) ? a.w[(3 * r + dx) * a.C + co] : 0.f;
            }
            if (SPLIT) {
                float vh[8], vm[8], vl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) split3(v[j], vh[j], vm[j], vl[j]);
                bwb[0][m] = pack8(vh); bwb[NIMG > 1 ? 1 : 0][m] = pack8(vm); bwb[NIMG > 2 ? 2 : 0][m] = pack8(vl);
            } else {
                bwb[0][m] = pack8(v);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int tap = 2 * s + hsel;
            bw[s] = (tap < 27 && cv) ? a.w[tap * a.C + co] : 0.f;
        }
    }
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, c0 = 0.f, c1 = 0.f;
    if (MODE != MODE_STATS && cv) { sc = a.scale[co]; sh = a.shift[co]; }
    if ((MODE == MODE_REDUCE || MODE == MODE_WGRAD || MODE == MODE_RD) && cv) { mu = a.mean[co]; is = a.invstd[co]; }
    if (MODE == MODE_WGRAD && cv) { c0 = a.coef[co]; c1 = a.coef[a.C + co]; }

    float s1 = 0.f, s2 = 0.f;               // STATS: sum z, sum z^2;  REDUCE / RD: sum dy, sum dy*xhat
    float dacc[MODE == MODE_RD ? 27 : 1];   // RD: D[tap][this lane's channel]
#pragma unroll
    for (int t = 0; t < (MODE == MODE_RD ? 27 : 1); ++t) dacc[t] = 0.f;
    f32x16 accw;                            // WGRAD: dw[tap = row][co = column]
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    const int a_tap = tapoff(l31);          // lanes >= 27 read tap 0's voxels; their rows are dropped
    // BF16 wgrad gather: byte address of (this lane's tap, lane half's brick rows) in the copy of the tap's parity
    const int wg_tap = l31 < 27 ? l31 : 0, wg_c = (wg_tap % 3) & 1;
    const unsigned wg_b = hb_base + 2u * (unsigned)((wg_tap / 9) * BPLANE + ((wg_tap / 3) % 3) * BROW + wg_tap % 3 + wg_c +
                                                    wg_c * BCOPY + 2 * hsel * BROW);

    const int tile_begin = blockIdx.x * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.ntiles) tile_end = a.ntiles;

    // halo brick: 600 scalars / 256 threads -> 3 registers per thread, fetched one brick ahead through a buffer
    // resource whose base is the brick's halo origin: the per-lane byte offsets and halo coordinates are computed once
    // per kernel, an interior brick loads with no vector arithmetic, a brick at a volume face checks (hd, hh, hw) of a
    // piece with one packed-byte range test, and an invalid piece carries the offset 2^31 >= num_records (hardware
    // zero fill).  fp32 MFMA and the vector ALU share one issue port: the index arithmetic this
    // replaces (~30 instructions per piece) and the 64-bit pooled addresses below were a third of these passes.
    constexpr int OOB = (int)0x80000000u;
    constexpr int HVN = (NHALO + 255) / 256;
    float hv[HVN];
    int hrel[HVN], hcrd[HVN], hdst[HVN];
#pragma unroll
    for (int q = 0; q < HVN; ++q) {
        const int e = tid + q * 256;
        const int hw = e % HW, hh = (e / HW) % HH, hd = e / (HW * HH);
        hrel[q] = e < NHALO ? ((hd * a.H + hh) * a.W + hw) * 4 : OOB;
        hcrd[q] = hd | hh << 8 | hw << 16;
        hdst[q] = hd * BPLANE + hh * BROW + hw;          // BF16: element index in copy 0
    }
    if (B16L) {                                          // pad elements (row tails, plane gaps) are read against zero weights
        for (int e = tid; e < NIMG * NHB_DW; e += 256) halo[e] = 0.f;
        __syncthreads();
    }
    auto min_i = [](int x_, int y_) { return x_ < y_ ? x_ : y_; };
    // Brick coordinates advance incrementally (a workgroup walks consecutive bricks): decoding the linear index with
    // three runtime divisions per brick — once for the brick, once for its prefetch — was ~100 of the ~200 scalar
    // instructions of an iteration, and the scalar unit (one issue per SIMD turn, in order with the wave's vector
    // work) was 60 % busy in the bf16 passes (PMC, profiles/r02_pmc_c1_bf16.txt).
    // ... and so do the halo origin (64-bit) and the "brick row is interior / complete" tests: per brick that is an add and
    // two compares; everything else is redone only when a brick row wraps.
    struct Crd { int tw, th, td, b; long xoff; bool in_dh, full_dh; };
    const bool even_dims = (a.D % 2 == 0) && (a.H % 2 == 0) && (a.W % 2 == 0);
    auto row_setup = [&](Crd& c) {          // the parts of a brick that depend on (b, td, th) only
        const int d0 = c.td * TD, h0 = c.th * TH;
        c.xoff = (long)c.b * a.D * a.H * a.W + ((long)((d0 - 1) * a.H + (h0 - 1)) * a.W + (c.tw * TW - 1));
        c.in_dh = d0 >= 1 && d0 + TD < a.D && h0 >= 1 && h0 + TH < a.H;
        c.full_dh = d0 + TD <= a.D && h0 + TH <= a.H && even_dims;
    };
    auto next_crd = [&](Crd c) {
        c.xoff += TW;
        if (++c.tw == a.tilesW) {
            c.tw = 0;
            if (++c.th == a.tilesH) {
                c.th = 0;
                if (++c.td == a.tilesD) { c.td = 0; ++c.b; }
            }
            row_setup(c);
        }
        return c;
    };
    auto fetch = [&](const Crd& c) {
        const int d0 = c.td * TD, h0 = c.th * TH, w0 = c.tw * TW;
        const float* org = a.x + c.xoff;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(org), 0, 0x7FFFFFFF, 0x00020000);
        if (c.in_dh && w0 >= 1 && w0 + TW < a.W) {
#pragma unroll
            for (int q = 0; q < HVN; ++q)
                hv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, hrel[q], 0, 0));
        } else {
            // valid iff lo <= (hd, hh, hw) <= hi per byte: bit 7 of every byte of (crd + 0x808080 - lo) and of
            // ((hi | 0x808080) - crd) survives exactly when there is no borrow
            const int lo = (d0 == 0 ? 1 : 0) | (h0 == 0 ? 1 : 0) << 8 | (w0 == 0 ? 1 : 0) << 16;
            const int hi = min_i(HD - 1, a.D - d0) | min_i(HH - 1, a.H - h0) << 8 | min_i(HW - 1, a.W - w0) << 16;
            const unsigned lo_bias = 0x808080u - (unsigned)lo, hi_bias = (unsigned)hi | 0x808080u;
#pragma unroll
            for (int q = 0; q < HVN; ++q) {
                const unsigned tt = (unsigned)hcrd[q] + lo_bias, uu = hi_bias - (unsigned)hcrd[q];
                const int off = (((tt & uu) | ~0x808080u) == 0xFFFFFFFFu) ? hrel[q] : OOB;
                hv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
            }
        }
    };
    // pooled / dpool addresses: lane part = (second brick row of the lane half, channel); everything else is scalar
    constexpr int PSZ = P16 ? 2 : 4;
    const int pl_lane = cv ? (hsel * OW * a.C + co) * PSZ : OOB;
    Crd cur;
    {
        int t = tile_begin;
        cur.tw = t % a.tilesW; t /= a.tilesW;
        cur.th = t % a.tilesH; t /= a.tilesH;
        cur.td = t % a.tilesD;
        cur.b = t / a.tilesD;
        row_setup(cur);
    }
    if (tile_begin < tile_end) fetch(cur);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int b = cur.b;
        const int d0 = cur.td * TD, h0 = cur.th * TH, w0 = cur.tw * TW;
        const bool full = cur.full_dh && w0 + TW <= a.W;
        cur = next_crd(cur);                 // from here on: the NEXT brick (prefetch target, and the next iteration's own)
        if (tile > tile_begin) __syncthreads();
#pragma unroll
        for (int q = 0; q < HVN; ++q) {
            const int e = tid + q * 256;
            if (SPLIT) {
                if (e < NHALO) {
                    float ph, pm, pl;
                    split3(hv[q], ph, pm, pl);
                    const unsigned short p16[3] = {(unsigned short)(__builtin_bit_cast(unsigned, ph) >> 16),
                                                   (unsigned short)(__builtin_bit_cast(unsigned, pm) >> 16),
                                                   (unsigned short)(__builtin_bit_cast(unsigned, pl) >> 16)};
#pragma unroll
                    for (int im = 0; im < 3; ++im) {
                        unsigned short* dst = reinterpret_cast<unsigned short*>(halo) + im * (2 * NHB_DW) + hdst[q];
                        dst[0] = p16[im];
                        dst[BCOPY + 1] = p16[im];
                    }
                    if (KEEP32) halo32[e] = hv[q];
                }
            } else if (BF16) {
                if (e < NHALO) {
                    const unsigned short h16 = (unsigned short)(tmf_pack_bf16(hv[q], 0.f) & 0xFFFFu);
                    unsigned short* dst = reinterpret_cast<unsigned short*>(halo) + hdst[q];
                    dst[0] = h16;
                    dst[BCOPY + 1] = h16;
                }
            } else {
                if (e < NHALO) halo[e] = hv[q];
            }
        }
        __syncthreads();
        if (tile + 1 < tile_end) fetch(cur);

        auto process = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;   // brick entirely inside the volume: no per-voxel checks
        // ---- z = conv(x) for BOTH M-tiles of this wave, interleaved (two independent MFMA chains) ----
        //      A[i = voxel][k = tap], voxel i = lane & 31 in fragment-row order
        f32x16 zt[NTI];
        constexpr bool LAZY = (BF16 && MODE == MODE_WGRAD) || (SPLIT && ((TMF_C1X_LAZY >> MODE) & 1));   // register budget: one M-tile's z at a time
        const int i = l31;
        // BF16: A[i = voxel][k]: per MFMA four dwords = the lane half's tap pair of four tap rows (rows >= 9 repeat row 8
        // against zero weights); the lane's byte address is loop-invariant up to the M-tile origin
        const unsigned lane_b = hb_base + 2u * (unsigned)(((i >> 3) & 1) * BPLANE + (2 * ((i >> 2) & 1) + ((i >> 1) & 1)) * BROW +
                                                        2 * ((i >> 4) & 1) + 2 * (i & 1) + (i & 1) * BCOPY + 2 * hsel);
        auto load_rows = [&](int ti, unsigned (&pr)[9], int img = 0) {
            const int mt = wave * NTI + ti;
            const unsigned tb = lane_b + 2u * (unsigned)((2 * (mt >> 2)) * BPLANE + 4 * ((mt >> 1) & 1) * BROW + 4 * (mt & 1)) +
                                (unsigned)(img * NHB_DW * 4);
#pragma unroll
            for (int r = 0; r < 9; ++r)
                pr[r] = *reinterpret_cast<const __attribute__((address_space(3))) unsigned*>((size_t)(tb + 2u * (unsigned)brow_off(r)));
        };
        auto mma = [&](int ti, int m, const unsigned (&pr)[9], int wpart = 0) {
            const u32x4 av = {pr[4 * m < 9 ? 4 * m : 8], pr[4 * m + 1 < 9 ? 4 * m + 1 : 8],
                              pr[4 * m + 2 < 9 ? 4 * m + 2 : 8], pr[4 * m + 3 < 9 ? 4 * m + 3 : 8]};
#if TMF_C1X_ABL & 2
            asm volatile("" :: "v"(av));
            zt[ti][m] += __builtin_bit_cast(float, av[0]);
#else
            zt[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tmf_bf16x8, av), bwb[wpart < NIMG ? wpart : 0][m], zt[ti], 0, 0, 0);
#endif
        };
        auto conv_one = [&](int ti) {
            unsigned pr[9];
#pragma unroll
            for (int r = 0; r < 16; ++r) zt[ti][r] = 0.f;
            if (SPLIT) {                                    // (the order of the interleaved form below: bit-identical z)
#pragma unroll
                for (int img = 2; img >= 0; --img) {
                    load_rows(ti, pr, img);
#pragma unroll
                    for (int wp = 2 - img; wp >= 0; --wp)
#pragma unroll
                        for (int m = 0; m < 3; ++m) mma(ti, m, pr, wp);
                }
                return;
            }
            load_rows(ti, pr);
#pragma unroll
            for (int m = 0; m < 3; ++m) mma(ti, m, pr);
        };
        if (!LAZY) {
            const int vox = (((i >> 3) & 1) * HH + 2 * ((i >> 2) & 1) + ((i >> 1) & 1)) * HW + 2 * ((i >> 4) & 1) + (i & 1);
            int a_vox[NTI];
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) {
                const int mt = wave * NTI + ti;
                a_vox[ti] = ((2 * (mt >> 2)) * HH + 4 * ((mt >> 1) & 1)) * HW + 4 * (mt & 1) + vox;
#pragma unroll
                for (int r = 0; r < 16; ++r) zt[ti][r] = 0.f;
            }
            if (SPLIT) {
                // image l, then m, then h — the products in ascending size: wh xl | wm xm, wh xm | wl xh, wm xh, wh xh
#pragma unroll
                for (int img = 2; img >= 0; --img) {
                    unsigned pr[NTI][9];
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti) load_rows(ti, pr[ti], img);
#pragma unroll
                    for (int wp = 2 - img; wp >= 0; --wp)
#pragma unroll
                        for (int m = 0; m < 3; ++m)
#pragma unroll
                            for (int ti = 0; ti < NTI; ++ti) mma(ti, m, pr[ti], wp);
                }
            } else if (BF16) {
                unsigned pr[NTI][9];
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti) load_rows(ti, pr[ti]);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti) mma(ti, m, pr[ti]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 14; ++s) {
                    const int off = hsel ? tapoff(2 * s + 1) : tapoff(2 * s);
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti)
                        zt[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(halo[a_vox[ti] + off], bw[s], zt[ti], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti) {
            const int mt = wave * NTI + ti;                     // M-tile of the brick (wave-uniform)
            const int org = ((2 * (mt >> 2)) * HH + 4 * ((mt >> 1) & 1)) * HW + 4 * (mt & 1);
            if (LAZY) conv_one(ti);
            f32x16& z = zt[ti];
            // voxel coordinates of this lane's 16 rows (row r, lane half hsel), relative to the brick:
            //   d = 2*(mt>>2) + r[2],  h = 4*((mt>>1)&1) + 2*hsel + r[1],  w = 4*(mt&1) + 2*r[3] + r[0]
            const int bd = d0 + 2 * (mt >> 2), bh = h0 + 4 * ((mt >> 1) & 1) + 2 * hsel, bwid = w0 + 4 * (mt & 1);

            if (MODE == MODE_STATS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gd = bd + ((r >> 2) & 1), gh = bh + ((r >> 1) & 1), gw = bwid + 2 * ((r >> 3) & 1) + (r & 1);
                    if (FULL || (gd < a.D && gh < a.H && gw < a.W)) { s1 += z[r]; s2 += z[r] * z[r]; }
                }
                continue;
            }

            // ---- per pooling window q (= r >> 3): LeakyReLU (slope > 0) is increasing, so the window maximum of the
            //      activation is the activation of the maximum of y = scale*z + shift: one fma per element and a max
            //      tree instead of activation + mask + running arg-max per element (the passes were VALU-bound).  The
            //      routed element is the FIRST k (torch scan order k = 4 d + 2 h + w) with y[k] == max. ----
            float K0 = 0.f, K1 = 0.f;
            if (MODE == MODE_WGRAD) {        // dz = sc*(dy - c0 - xhat*c1) = sc*dy + z*K1 + K0
                K1 = -sc * c1 * is;
                K0 = sc * (c1 * mu * is - c0);
            }
            // pooled window of this lane: (od, ohb + hsel, owb + q); the sample's pooled tensor is one buffer resource
            const int od = (d0 >> 1) + (mt >> 2), ohb = (h0 >> 1) + 2 * ((mt >> 1) & 1), owb = (w0 >> 1) + 2 * (mt & 1);
            void* pbase = MODE == MODE_FWD ? a.pooled : const_cast<void*>(a.dpool);
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(pbase) + (size_t)b * OD * OH * OW * a.C * PSZ, 0, OD * OH * OW * a.C * PSZ, 0x00020000);
            const int pv = (FULL || ohb + hsel < OH) ? pl_lane : OOB;      // this lane's voffset (out of range = no window)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ow = owb + q;
                const bool win_u = FULL || (od < OD && ow < OW);           // wave-uniform part of "the window exists"
                const int psoff = ((od * OH + ohb) * OW + ow) * a.C * PSZ;
                float y[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) y[k] = z[8 * q + k] * sc + sh;
                const float ymax = fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7])));
                const float lrm = ymax > 0.f ? 1.f : a.slope;
                if (MODE == MODE_FWD) {
                    const float best = ymax * lrm;
                    if ((TMF_C1X_ABL & 1) && best == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, best), pr, pv, psoff, 0);
                    if (win_u && !(TMF_C1X_ABL & 1)) {
                        if (P16) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(tmf_pack_bf16(best, 0.f) & 0xFFFFu), pr, pv, psoff, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, best), pr, pv, psoff, 0);
                    }
                    continue;
                }
                float g = 0.f;                                             // lanes without a window read zeros
                if (win_u)
                    g = P16 ? __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(pr, pv, psoff, 0) << 16)
                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, pv, psoff, 0));
                const float gl = g * lrm;                       // dLoss/dy at the routed element
                if (MODE == MODE_REDUCE || MODE == MODE_RD) {
                    float zs = z[8 * q + 7];                    // z of the first maximum (a pooled window is all-valid)
#pragma unroll
                    for (int k = 6; k >= 0; --k) zs = (y[k] == ymax) ? z[8 * q + k] : zs;
                    s1 += gl;
                    s2 += gl * ((zs - mu) * is);
                    if (MODE == MODE_RD) {
                        // the routed voxel's 27 input taps (halo index of fragment row r = 8 q + k: row_off(r) + the lane half's rows)
                        int arg = 7;
#pragma unroll
                        for (int k = 6; k >= 0; --k) arg = (y[k] == ymax) ? k : arg;
                        if (BF16) {
                            // the bf16 halo (copy 0: element index = hd * BPLANE + hh * BROW + hw): the inputs as the MFMAs of z saw them
                            const int hd = 2 * (mt >> 2) + ((arg >> 2) & 1), hh = 4 * ((mt >> 1) & 1) + 2 * hsel + ((arg >> 1) & 1);
                            const unsigned short* hb = reinterpret_cast<const unsigned short*>(halo) + hd * BPLANE + hh * BROW +
                                                       4 * (mt & 1) + 2 * q + (arg & 1);
#pragma unroll
                            for (int t = 0; t < 27; ++t) {
                                const float xv = __builtin_bit_cast(float, (unsigned int)hb[(t / 9) * BPLANE + ((t / 3) % 3) * BROW + t % 3] << 16);
                                dacc[t] = fmaf(xv, gl, dacc[t]);
                            }
                        } else {
                        const int vx = org + hsel * 2 * HW + ((arg >> 2) & 1) * HH * HW + ((arg >> 1) & 1) * HW + 2 * q + (arg & 1);
                        const float* h32 = KEEP32 ? halo32 : halo;
#pragma unroll
                        for (int t = 0; t < 27; ++t) dacc[t] = fmaf(h32[vx + tapoff(t)], gl, dacc[t]);
                        }
                    }
                } else {
                    int arg = 7;
#pragma unroll
                    for (int k = 6; k >= 0; --k) arg = (y[k] == ymax) ? k : arg;
                    const float add = sc * gl;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = 8 * q + k;
                        const int gd = bd + ((r >> 2) & 1), gh = bh + ((r >> 1) & 1), gw = bwid + 2 * q + (r & 1);
                        const bool vv = FULL || (gd < a.D && gh < a.H && gw < a.W);
                        const float dzv = z[r] * K1 + K0 + (k == arg ? add : 0.f);
                        z[r] = (vv && cv) ? dzv : 0.f;          // dz, in place
                    }
                }
            }
            if (MODE == MODE_WGRAD) {
                // dw[tap][co] += sum_voxel x[voxel + tap] * dz[voxel][co]:  A[i = tap][k = lane half], B = dz regs
                const int a_row = org + hsel * 2 * HW + a_tap;
                if (BF16) {
                    // K = the 16 voxels of fragment rows 8 m .. 8 m + 7 of both lane halves: B is the lane's own dz
                    // registers, A the same voxels' inputs shifted by the lane's tap — rows 2 u, 2 u + 1 are w-neighbours,
                    // i.e. one dword of the copy whose parity matches the tap's dx
                    const unsigned wt = wg_b + 2u * (unsigned)((2 * (mt >> 2)) * BPLANE + 4 * ((mt >> 1) & 1) * BROW + 4 * (mt & 1));
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float bv[8];
                        unsigned ad[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            ad[u] = *reinterpret_cast<const __attribute__((address_space(3))) unsigned*>(
                                (size_t)(wt + 2u * (unsigned)((u >> 1) * BPLANE + (u & 1) * BROW + 2 * m)));
#pragma unroll
                        for (int j = 0; j < 8; ++j) bv[j] = z[8 * m + j];
                        const u32x4 av = {ad[0], ad[1], ad[2], ad[3]};
                        accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tmf_bf16x8, av), pack8(bv), accw, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        accw = __builtin_amdgcn_mfma_f32_32x32x2f32((KEEP32 ? halo32 : halo)[a_row + row_off(r)], z[r], accw, 0, 0, 0);
                }
            }
        }
        };
        if (full)
            process(std::true_type{});
        else
            process(std::false_type{});
    }

    // ---- workgroup reduction and partial slab ----
    if (MODE == MODE_STATS || MODE == MODE_REDUCE || MODE == MODE_RD) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        __syncthreads();
        if (hsel == 0) { red[(wave * 32 + l31) * 2] = s1; red[(wave * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 32 && n0 + tid < a.C) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) { t1 += red[(m * 32 + tid) * 2]; t2 += red[(m * 32 + tid) * 2 + 1]; }
            a.partial[((size_t)blockIdx.x * 2 + 0) * a.C + n0 + tid] = t1;
            a.partial[((size_t)blockIdx.x * 2 + 1) * a.C + n0 + tid] = t2;
        }
    }
    if (MODE == MODE_RD) {                  // D: the two lane halves, then the four waves
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const float v = dacc[t] + __shfl_xor(dacc[t], 32);
            if (hsel == 0) red[(wave * 32 + t) * 32 + l31] = v;
        }
        __syncthreads();
        for (int e = tid; e < 27 * 32; e += 256) {
            const int tap = e >> 5, c = e & 31;
            if (n0 + c < a.C) {
                const float v = red[(0 * 32 + tap) * 32 + c] + red[(1 * 32 + tap) * 32 + c] +
                                red[(2 * 32 + tap) * 32 + c] + red[(3 * 32 + tap) * 32 + c];
                a.partial2[((size_t)blockIdx.x * 27 + tap) * a.C + n0 + c] = v;
            }
        }
    } else if (MODE == MODE_WGRAD) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tap = (r & 3) + 8 * (r >> 2) + 4 * hsel;
            red[(wave * 32 + tap) * 32 + l31] = accw[r];
        }
        __syncthreads();
        for (int e = tid; e < 27 * 32; e += 256) {
            const int tap = e >> 5, c = e & 31;
            if (n0 + c < a.C) {
                const float v = red[(0 * 32 + tap) * 32 + c] + red[(1 * 32 + tap) * 32 + c] +
                                red[(2 * 32 + tap) * 32 + c] + red[(3 * 32 + tap) * 32 + c];
                a.partial[((size_t)blockIdx.x * 27 + tap) * a.C + n0 + c] = v;
            }
        }
    }
}

struct Plan { int tilesD, tilesH, tilesW, ntiles, nby, tpb, nblk; };
Plan make_plan(int B, int D, int H, int W, int C, int target_blocks) {
    Plan p;
    p.tilesD = tmf_cdiv(D, TD); p.tilesH = tmf_cdiv(H, TH); p.tilesW = tmf_cdiv(W, TW);
    p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
    p.nby = tmf_cdiv(C, 32);
    int want = target_blocks / p.nby;
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tpb = tmf_cdiv(p.ntiles, want);
    p.nblk = tmf_cdiv(p.ntiles, p.tpb);
    return p;
}
// STATS / REDUCE / WGRAD write one slab per workgroup: keep them few (4 per CU: measured 0.772 ms for the four passes
// against 0.796 at 8 per CU and 0.81 at 14-20; TMF_C1_BLOCKS overrides); FWD has no slab.
static int slab_blocks() {
    static int v = 0;
    if (v == 0) { const char* e = getenv("TMF_C1_BLOCKS"); v = e ? atoi(e) : 1024; if (v < 64) v = 1024; }
    return v;
}
#define SLAB_BLOCKS slab_blocks()

int check(const char* fn, int B, int D, int H, int W, int C) {
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, TMF_E_SHAPE, "%s: non-positive dimension", fn);
    TMF_REQUIRE((long)D * H * W * C < (1L << 31), TMF_E_SHAPE, "%s: one sample exceeds 2^31 elements", fn);
    TMF_REQUIRE((long)(D > 6 ? D : 6) * H * W < (1L << 29), TMF_E_SHAPE, "%s: the input volume exceeds 2^29 voxels", fn);
    return TMF_OK;
}

Args base_args(const float* x, const float* w, int D, int H, int W, int C, const Plan& p, float slope) {
    Args a = {};
    a.x = x; a.w = w; a.D = D; a.H = H; a.W = W; a.C = C;
    a.tilesD = p.tilesD; a.tilesH = p.tilesH; a.tilesW = p.tilesW; a.ntiles = p.ntiles; a.tiles_per_block = p.tpb;
    a.slope = slope;
    return a;
}

int g_c1_split = -1;
// tmf_set_option("c1_split", 0 | 1) / TMF_C1_SPLIT (default 1): the fp32 passes compute z as exact 3-way bf16 splits (SPLIT above)
int c1_split_mode() {
    if (const int o = tmf_algo_override()) return (o & TMF_SNET_ALGO_C1_SPLIT) ? 1 : 0;
    if (g_c1_split < 0) {
        const char* e = getenv("TMF_C1_SPLIT");
        g_c1_split = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_c1_split;
}

}  // namespace

int tmf_c1_split_set(int v) { g_c1_split = v ? 1 : 0; return TMF_OK; }
extern "C" int tmf_c1_split_mode(void) { return c1_split_mode(); }

extern "C" int tmf_c1_blocks(int B, int D, int H, int W, int C) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return make_plan(B, D, H, W, C, SLAB_BLOCKS).nblk;
}

static int c1_stats(bool bf16, const float* x, const float* w, float* stat_partial,
                    int B, int D, int H, int W, int C, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(stat_partial);
    int rc = check("tmf_c1_stats", B, D, H, W, C);
    if (rc) return rc;
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    Args a = base_args(x, w, D, H, W, C, p, 0.f);
    a.partial = stat_partial;
    if (bf16) hipLaunchKernelGGL((conv1_fused_kernel<MODE_STATS, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else if (c1_split_mode()) hipLaunchKernelGGL((conv1_fused_kernel<MODE_STATS, false, false, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else      hipLaunchKernelGGL((conv1_fused_kernel<MODE_STATS, false>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_c1_stats");
}
extern "C" int tmf_c1_stats(const float* x, const float* w, float* stat_partial,
                            int B, int D, int H, int W, int C, void* stream) {
    return c1_stats(false, x, w, stat_partial, B, D, H, W, C, stream);
}
extern "C" int tmf_c1_stats_bf16(const float* x, const float* w, float* stat_partial,
                                 int B, int D, int H, int W, int C, void* stream) {
    return c1_stats(true, x, w, stat_partial, B, D, H, W, C, stream);
}

static int c1_bn_pool_fwd(bool bf16, bool p16, const float* x, const float* w, const float* scale, const float* shift,
                          void* pooled, int B, int D, int H, int W, int C, float slope, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(pooled);
    int rc = check("tmf_c1_bn_pool_fwd", B, D, H, W, C);
    if (rc) return rc;
    static const int fwd_mult = getenv("TMF_C1_FWD_MULT") ? atoi(getenv("TMF_C1_FWD_MULT")) : 4;
    const Plan p = make_plan(B, D, H, W, C, fwd_mult * SLAB_BLOCKS);     // a few bricks per workgroup (halo prefetch)
    Args a = base_args(x, w, D, H, W, C, p, slope);
    a.scale = scale; a.shift = shift; a.pooled = pooled;
    TMF_REQUIRE(bf16 || !p16, TMF_E_ARG, "tmf_c1_bn_pool_fwd: bf16 tensors only with the bf16 kernels");
    if (bf16 && p16) hipLaunchKernelGGL((conv1_fused_kernel<MODE_FWD, true, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else if (bf16)   hipLaunchKernelGGL((conv1_fused_kernel<MODE_FWD, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else if (c1_split_mode()) hipLaunchKernelGGL((conv1_fused_kernel<MODE_FWD, false, false, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else             hipLaunchKernelGGL((conv1_fused_kernel<MODE_FWD, false>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_c1_bn_pool_fwd");
}
extern "C" int tmf_c1_bn_pool_fwd(const float* x, const float* w, const float* scale, const float* shift,
                                  float* pooled, int B, int D, int H, int W, int C, float slope, void* stream) {
    return c1_bn_pool_fwd(false, false, x, w, scale, shift, pooled, B, D, H, W, C, slope, stream);
}
extern "C" int tmf_c1_bn_pool_fwd_bf16(const float* x, const float* w, const float* scale, const float* shift,
                                       void* pooled, int B, int D, int H, int W, int C, float slope, int pooled_bf16,
                                       void* stream) {
    return c1_bn_pool_fwd(true, pooled_bf16 != 0, x, w, scale, shift, pooled, B, D, H, W, C, slope, stream);
}

static int c1_bwd_reduce(bool bf16, bool p16, const float* x, const float* w, const float* scale, const float* shift,
                         const float* mean, const float* invstd, const void* dpool, float* partial,
                         int B, int D, int H, int W, int C, float slope, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift);
    TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(invstd); TMF_REQUIRE_PTR(dpool); TMF_REQUIRE_PTR(partial);
    int rc = check("tmf_c1_bwd_reduce", B, D, H, W, C);
    if (rc) return rc;
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    Args a = base_args(x, w, D, H, W, C, p, slope);
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.dpool = dpool; a.partial = partial;
    TMF_REQUIRE(bf16 || !p16, TMF_E_ARG, "tmf_c1_bwd_reduce: bf16 tensors only with the bf16 kernels");
    if (bf16 && p16) hipLaunchKernelGGL((conv1_fused_kernel<MODE_REDUCE, true, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else if (bf16)   hipLaunchKernelGGL((conv1_fused_kernel<MODE_REDUCE, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else if (c1_split_mode()) hipLaunchKernelGGL((conv1_fused_kernel<MODE_REDUCE, false, false, true>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    else             hipLaunchKernelGGL((conv1_fused_kernel<MODE_REDUCE, false>), dim3(p.nblk, p.nby), dim3(256), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_c1_bwd_reduce");
}
extern "C" int tmf_c1_bwd_reduce(const float* x, const float* w, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* dpool, float* partial,
                                 int B, int D, int H, int W, int C, float slope, void* stream) {
    return c1_bwd_reduce(false, false, x, w, scale, shift, mean, invstd, dpool, partial, B, D, H, W, C, slope, stream);
}
extern "C" int tmf_c1_bwd_reduce_bf16(const float* x, const float* w, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, const void* dpool, float* partial,
                                      int B, int D, int H, int W, int C, float slope, int pooled_bf16, void* stream) {
    return c1_bwd_reduce(true, pooled_bf16 != 0, x, w, scale, shift, mean, invstd, dpool, partial, B, D, H, W, C, slope, stream);
}

extern "C" size_t tmf_c1_bwd_wgrad_workspace_bytes(int B, int D, int H, int W, int C) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    return (size_t)(p.nblk + tmf_reduce_groups(p.nblk)) * 27 * C * 4;
}

static int c1_bwd_wgrad(bool bf16, bool p16, const float* x, const float* w, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* coef, const void* dpool,
                        float* dw, void* workspace, size_t workspace_bytes,
                        int B, int D, int H, int W, int C, float slope, int dw_layout, void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG,
                "tmf_c1_bwd_wgrad: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(mean);
    TMF_REQUIRE_PTR(invstd); TMF_REQUIRE_PTR(coef); TMF_REQUIRE_PTR(dpool); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    int rc = check("tmf_c1_bwd_wgrad", B, D, H, W, C);
    if (rc) return rc;
    const size_t need = tmf_c1_bwd_wgrad_workspace_bytes(B, D, H, W, C);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_c1_bwd_wgrad: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    Args a = base_args(x, w, D, H, W, C, p, slope);
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.coef = coef; a.dpool = dpool;
    a.partial = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    TMF_REQUIRE(bf16 || !p16, TMF_E_ARG, "tmf_c1_bwd_wgrad: bf16 tensors only with the bf16 kernels");
    if (bf16 && p16) hipLaunchKernelGGL((conv1_fused_kernel<MODE_WGRAD, true, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else if (bf16)   hipLaunchKernelGGL((conv1_fused_kernel<MODE_WGRAD, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else if (c1_split_mode()) hipLaunchKernelGGL((conv1_fused_kernel<MODE_WGRAD, false, false, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else             hipLaunchKernelGGL((conv1_fused_kernel<MODE_WGRAD, false>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    if ((rc = tmf_launch_result("tmf_c1_bwd_wgrad"))) return rc;
    const long n = 27L * C;
    return tmf_reduce_slabs((const float*)workspace, p.nblk, n, (float*)workspace + (size_t)p.nblk * n, dw, s,
                            "tmf_c1_bwd_wgrad(reduce)", dw_layout == TMF_DW_REFERENCE ? 1 : 0, C);
}

extern "C" int tmf_c1_bwd_wgrad(const float* x, const float* w, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* coef, const float* dpool,
                                float* dw, void* workspace, size_t workspace_bytes,
                                int B, int D, int H, int W, int C, float slope, int dw_layout, void* stream) {
    return c1_bwd_wgrad(false, false, x, w, scale, shift, mean, invstd, coef, dpool, dw, workspace, workspace_bytes, B, D, H, W,
                        C, slope, dw_layout, stream);
}
extern "C" int tmf_c1_bwd_wgrad_bf16(const float* x, const float* w, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, const float* coef, const void* dpool,
                                     float* dw, void* workspace, size_t workspace_bytes,
                                     int B, int D, int H, int W, int C, float slope, int pooled_bf16, int dw_layout,
                                     void* stream) {
    return c1_bwd_wgrad(true, pooled_bf16 != 0, x, w, scale, shift, mean, invstd, coef, dpool, dw, workspace, workspace_bytes,
                        B, D, H, W, C, slope, dw_layout, stream);
}

// ---- reduce + weight gradient in ONE pass over the volume (fp32; needs the Gram data tmf_c1_stats_g left in `gram`) ----
// workspace: [nblk][2][C] sums | [nblk][27][C] D slabs | their reduction scratch | [27][C] reduced D
int tmf_c1_bwd_fused_finish(const float* part, int nblk, const float* dred, const float* w, const void* gram, const float* scale,
                            const float* mean, const float* invstd, double count, float* dgamma, float* dbeta, float* dw, int C,
                            int dw_ref, int round16, void* stream);
extern "C" size_t tmf_c1_bwd_fused_workspace_bytes(int B, int D, int H, int W, int C) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    return ((size_t)p.nblk * 2 * C + (size_t)(p.nblk + tmf_reduce_groups(p.nblk) + 1) * 27 * C) * 4;
}
static int c1_bwd_fused(bool bf16, bool p16, const float* x, const float* w, const float* scale, const float* shift, const float* mean,
                        const float* invstd, const void* dpool, const void* gram, float* dw, float* dgamma, float* dbeta,
                        void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, float slope,
                        int dw_layout, void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG, "tmf_c1_bwd_fused: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(mean); TMF_REQUIRE_PTR(invstd);
    TMF_REQUIRE_PTR(dpool); TMF_REQUIRE_PTR(gram); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    int rc = check("tmf_c1_bwd_fused", B, D, H, W, C);
    if (rc) return rc;
    TMF_REQUIRE(C <= 64, TMF_E_SHAPE, "tmf_c1_bwd_fused: C = %d > 64", C);
    const size_t need = tmf_c1_bwd_fused_workspace_bytes(B, D, H, W, C);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_c1_bwd_fused: workspace %zu B < required %zu B", workspace_bytes, need);
    const Plan p = make_plan(B, D, H, W, C, SLAB_BLOCKS);
    Args a = base_args(x, w, D, H, W, C, p, slope);
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.dpool = dpool;
    float* part = (float*)workspace;
    float* slabs = part + (size_t)p.nblk * 2 * C;
    const long n = 27L * C;
    float* scratch = slabs + (size_t)p.nblk * n;
    float* dred = scratch + (size_t)tmf_reduce_groups(p.nblk) * n;
    a.partial = part; a.partial2 = slabs;
    hipStream_t s = (hipStream_t)stream;
    TMF_REQUIRE(bf16 || !p16, TMF_E_ARG, "tmf_c1_bwd_fused: bf16 tensors only with the bf16 kernels");
    if (bf16 && p16) hipLaunchKernelGGL((conv1_fused_kernel<MODE_RD, true, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else if (bf16)   hipLaunchKernelGGL((conv1_fused_kernel<MODE_RD, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else if (c1_split_mode()) hipLaunchKernelGGL((conv1_fused_kernel<MODE_RD, false, false, true>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    else             hipLaunchKernelGGL((conv1_fused_kernel<MODE_RD, false>), dim3(p.nblk, p.nby), dim3(256), 0, s, a);
    if ((rc = tmf_launch_result("tmf_c1_bwd_fused"))) return rc;
    if ((rc = tmf_reduce_slabs(slabs, p.nblk, n, scratch, dred, s, "tmf_c1_bwd_fused(reduce)"))) return rc;
    return tmf_c1_bwd_fused_finish(part, p.nblk, dred, w, gram, scale, mean, invstd, (double)B * D * H * W, dgamma, dbeta, dw, C,
                                   dw_layout == TMF_DW_REFERENCE ? 1 : 0, bf16 ? 1 : 0, stream);
}
extern "C" int tmf_c1_bwd_fused(const float* x, const float* w, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* dpool, const void* gram, float* dw, float* dgamma, float* dbeta,
                                void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, float slope,
                                int dw_layout, void* stream) {
    return c1_bwd_fused(false, false, x, w, scale, shift, mean, invstd, dpool, gram, dw, dgamma, dbeta, workspace, workspace_bytes,
                        B, D, H, W, C, slope, dw_layout, stream);
}
// the bf16 mode's form (gram from tmf_c1_stats_g_bf16): z and D from the volume and the taps rounded to bf16, dy unrounded — the
// weight gradient is that of the bf16 forward WITHOUT the second rounding of dz that tmf_c1_bwd_wgrad_bf16's product performs
extern "C" int tmf_c1_bwd_fused_bf16(const float* x, const float* w, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const void* dpool, const void* gram, float* dw, float* dgamma, float* dbeta,
                                     void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, float slope,
                                     int pooled_bf16, int dw_layout, void* stream) {
    return c1_bwd_fused(true, pooled_bf16 != 0, x, w, scale, shift, mean, invstd, dpool, gram, dw, dgamma, dbeta, workspace,
                        workspace_bytes, B, D, H, W, C, slope, dw_layout, stream);
}
