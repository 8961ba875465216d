// conv3d_winox.hip — the Winograd form F(2x2x2, 3x3x3) of the encoders' 3x3x3 convolutions (forward and data gradient) with the
// fp32 products carried by the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 numbers and six of the nine
// partial products are accumulated in fp32 (round 6).  gfx950 (MI355X).
//
// Why.  The fp32 matrix instructions run on the vector ALU (v_mfma_f32_32x32x2_f32: 64 cycles, one issue resource with every v_*
// instruction), so the persistent fp32 kernel of conv3d_wino.hip pays each of its ~4.7 vector instructions per MFMA in matrix
// time: 0.52 of the pipe, purely additive (profiles/r05_wino_ablations.txt).  v_mfma_f32_32x32x16_bf16 is a separate pipe at 16 x
// the rate, and vector instructions of a SECOND wave on the SIMD run beside it.  tools/microbench/mix_cost_bf16.hip
// (profiles/r06_mix_cost_bf16.txt): the instruction mix of one 16-channel chunk — 96 bf16 MFMAs, the input transform, the operand
// split, LDS reads, weight loads, halo copies — takes 5 795 cycles with two waves per SIMD against 10 178 for the fp32 mix.
//
// Exactness.  x = h + m + l with h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = (x - h) - m: three bf16 numbers, no rounding
// anywhere (8 + 8 + 8 significand bits).  x u = (hh + hm + mh + hl + lh + mm) + (ml + lm + ll); the dropped terms are below 2^-24
// of the product — under the rounding of the fp32 product itself.  The transformed weights U are the SAME fp32 numbers the fp32
// kernel multiplies (fp64 inside the pack kernel, rounded once), split by the pack kernel; the transformed input is split in
// the main loop.  Accumulation is fp32 (the MFMA's).  dtype of the path stays f32; tests hold it to the fp32 kernel's tolerances.
//
// Structure (8 waves, TWO per SIMD, one persistent workgroup per CU):
//   * wave = (pd, half of ph): the 8 positions (pd, ph = 2 mhh + {0, 1}, pw = 0..3) of the 4x4x4 transformed tile, one 32 x 32
//     accumulator each (128 registers) — 64 positions x 32 tiles x 32 output channels per item as in the fp32 kernel;
//   * chunks of 16 input channels = two 8-channel halo buffers in the parity-sorted slot map of conv3d_wino.hip (the lane (tile,
//     quad) reads its quad of both: K index j < 4 -> channel 4 quad + j, j >= 4 -> 8 + 4 quad + (j - 4)); halo by LDS-DMA one chunk
//     ahead (4 buffers), one barrier per chunk;
//   * per position: the d / h / w input transform in registers (shared rows kept across the two ph of the wave), the split (11
//     vector instructions per pair of values), six MFMAs with M = output channels (the weights are the A operand: a lane then
//     holds FOUR CONSECUTIVE channels of one tile per accumulator quad — 16-byte exchange writes and stores in the epilogue);
//   * the three bf16 parts of the transformed weights go global -> registers two positions ahead (one buffer_load_dwordx4 per
//     part: the lane's 8 K values), counted waits (s_waitcnt vmcnt(n): loads, LDS-DMA copies and stores retire in order);
//   * no wave specialisation: the two waves of a SIMD run the same code out of phase — one splits while the other multiplies;
//   * epilogue per item: w and half of h in registers, exchange through LDS in two passes (ho = 0, 1; 68 KB each, aliasing the
//     halo pair just consumed), d on the reading side, 16-byte stores, BatchNorm partials per workgroup (one row per launch).
//
// Replaces aten::conv3d / the data-gradient half of convolution_backward at /root/reference/models/networks.py:28,31,37,40,46.
#include <type_traits>
#include "tmf_common.h"

#ifndef X_PF                // weights requested this many positions ahead (1 | 2)
#define X_PF 2
#endif
#ifndef X_NGFAST
#define X_NGFAST 1
#endif
#ifndef X_SKEW              // 1: the waves 4-7 meet the barriers half a phase later than the waves 0-3 (measured level: off)
#define X_SKEW 0
#endif
#ifndef X_RB                // 1: all 16 LDS reads of a (group, buffer) are requested before the first is used (measured level: off)
#define X_RB 0
#endif
#ifndef X_ABL               // timing ablations (tools/build_variant.py --flags=-DX_ABL=n; results are wrong with any bit set):
#define X_ABL 0             // 1 no input transform, 2 no weight loads, 4 no halo copies, 8 no epilogue, 16 no MFMAs, 32 no split
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef TMF_WINOX_TRACE
// instrumented build (tools/winox_trace.py): shader-clock stamps of the phases of the SECOND item of workgroup 77, every wave.  The
// stamps go to LDS (the item table shrinks to make room) and leave at the end of the kernel: a global store per stamp would sit in
// the vmcnt queue the kernel's counted waits rely on and serialise the very thing that is measured.
__device__ long long g_winox_phases[8 * 64];
#define XTR(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 77 && it == 1 && lane == 0) xtr_lds[wave * 64 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define XTRC(i) XTR((i) + 20 * (c & 1))
#else
#define XTR(i)
#define XTRC(i)
#endif

constexpr int XN = 512;                               // 8 waves
constexpr int XCK = 16;                               // input channels per chunk
// the 4x8x8 brick and its halo in the slot map of conv3d_wino.hip (PGeom<0>): group = parity class * CP + (hd >> 1) * SD + (hh >> 1) * SH
// + (hw >> 1), slot = 2 group + (quad ^ (hh >> 1 & 1)), 16 bytes per slot
constexpr int BD = 4, BH = 8, BW = 8;
constexpr int CP = 96, SD = 32, SH = 6;
constexpr int SLOTS_G = 16 * CP;                      // 1 536 slots = 24 KB per 8-channel buffer
constexpr int RAWB = SLOTS_G * 16;
constexpr int XDMA = SLOTS_G / XN;                    // 3 LDS-DMA instructions per wave and 8-channel buffer
constexpr int TS = 68;                                // exchange: floats per (source wave, tile): [wo 2][channel 32] + 4 (bank spread)
constexpr int EX_OFF = 2 * RAWB;                      // the exchange aliases halo buffers 2, 3 (the pair of every item's LAST chunk)
constexpr int EX_BYTES = 8 * 32 * TS * 4;             // 69 632
constexpr int RED_OFF = EX_OFF + EX_BYTES;            // statistic sums [which 2][source 64][33] floats: every lane adds its own 4 + 4 per item
constexpr int PLAN_OFF = RED_OFF + 2 * 64 * 33 * 4 + 64;      // halo copy plan [9][512] ints: per lane hrel[3], hm[3], hoff[3] (registers are scarce)
constexpr int TAB_OFF = PLAN_OFF + 9 * XN * 4;
#ifdef TMF_WINOX_TRACE
constexpr int TAB = 96;
#else
constexpr int TAB = 256;                              // item table entries per workgroup (2 x int4 each)
#endif
constexpr size_t X_LDS_BYTES = (size_t)TAB_OFF + 256 * 32;
static_assert(EX_OFF + EX_BYTES >= 4 * RAWB && X_LDS_BYTES <= 160 * 1024, "LDS carving");

__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// LDS-DMA of 16 bytes per lane: LDS byte = lds_wave_base + 16 * lane <- base + voff + soff; a lane outside the range delivers zeros
__device__ __forceinline__ void blds16(int voff, i32x4 rsrc, int soff, unsigned lds_wave_base) {
    asm volatile("" : "+s"(soff));                  // (a register, not a literal the instruction cannot encode)
    rsrc = i32x4{__builtin_amdgcn_readfirstlane(rsrc[0]), __builtin_amdgcn_readfirstlane(rsrc[1]), __builtin_amdgcn_readfirstlane(rsrc[2]),
                 __builtin_amdgcn_readfirstlane(rsrc[3])};  // (folds away where the descriptor already sits in scalar registers)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base) : "memory");
}
// 16 bytes per lane to registers, invisible to the compiler's wait-count pass (the kernel counts its waits itself)
__device__ __forceinline__ void bload16(i32x4& dst, int voff, i32x4 rsrc, int soff) {
    asm volatile("" : "+s"(soff));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin3(i32x4& a, i32x4& b, i32x4& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
// a wide store reads its data registers after it has issued (conv3d_wino.hip: store_guard)
__device__ __forceinline__ void store_guard() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0");
    __builtin_amdgcn_sched_barrier(0);
}
// workgroup barrier that leaves the weight loads in flight (the LDS traffic of this wave has been consumed by then)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ... for the waves whose flag is set, as ONE opaque statement: a branch in the source splits the loop into blocks and costs the
// register allocator two accumulators (spilled), although the two roles run the same instructions
__device__ __forceinline__ void wg_barrier_if(int flag) {
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .LWGB%=\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n.LWGB%=:" ::"s"(__builtin_amdgcn_readfirstlane(flag)) : "memory", "scc");
}

// exact 3-way bf16 split of 8 fp32 values (K order: y0[0..3], y1[0..3]); element 2 j in the low half of register j
__device__ __forceinline__ void split8(const float (&y0)[4], const float (&y1)[4], i32x4& H, i32x4& M, i32x4& L) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = j < 2 ? y0[2 * j] : y1[2 * j - 4], b = j < 2 ? y0[2 * j + 1] : y1[2 * j - 3];
        const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
        const float ra = a - __builtin_bit_cast(float, ua & 0xFFFF0000u), rb = b - __builtin_bit_cast(float, ub & 0xFFFF0000u);
        const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
        const float la = ra - __builtin_bit_cast(float, va & 0xFFFF0000u), lb = rb - __builtin_bit_cast(float, vb & 0xFFFF0000u);
        H[j] = (int)__builtin_amdgcn_perm(ub, ua, 0x07060302u);
        M[j] = (int)__builtin_amdgcn_perm(vb, va, 0x07060302u);
        L[j] = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, lb), __builtin_bit_cast(unsigned, la), 0x07060302u);
    }
}
__device__ __forceinline__ f32x16 mfma_bf16(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tmf_bf16x8, a), __builtin_bit_cast(tmf_bf16x8, b), c, 0, 0, 0);
}

// MODE 0: z only (data gradient); 1: z + BatchNorm statistic partials (train forward); 2 / 3: the eval-mode block in ONE pass —
// y = LeakyReLU(scale z + shift) (BatchNorm is affine in eval mode), 3: 2x2x2 max-pooled (floor mode) before anything is stored: the
// window's d pair sits in one reader lane, its h pair in the two passes of the epilogue, its w pair in lane ^ 8
template <int MODE>
__global__ __launch_bounds__(XN) void conv3d_winox_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ u3, float* __restrict__ z, float* __restrict__ stat_partial,
    int B, int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int nbricks, int item0, int nitems,
    int stat_rows, int stat_accum, const float* __restrict__ aff_scale = nullptr, const float* __restrict__ aff_shift = nullptr,
    float slope = 0.f, int swap = 0) {
    // swap (round 6): the kernel's d axis is the TENSOR's h axis and vice versa — D, H, tilesD, tilesH arrive in kernel space, the
    // voxel strides of the two axes trade places and so do pd / ph in the weight positions (U is separable: U_k[pd][ph] = U[ph][pd]).
    // The items' 4-voxel side then lies along the tensor's h: 22x27x22 is 7 x 3 x 3 = 63 items per sample instead of 6 x 4 x 3 = 72.
    const int sd = swap ? W : H * W, sh = swap ? D * W : W;                 // voxel strides of the kernel's d and h axes
    constexpr bool STATS = MODE == 1;
    constexpr bool AFFINE = MODE >= 2, POOL = MODE == 3;
    constexpr int EPI_STORES = POOL ? 1 : 4;                // vector-memory stores of an item's epilogue (static: the counted waits)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ex = smem + EX_OFF / 4;
    float* red = smem + RED_OFF / 4;
    i32x4* tab = reinterpret_cast<i32x4*>(smem + TAB_OFF / 4);
#ifdef TMF_WINOX_TRACE
    long long* xtr_lds = reinterpret_cast<long long*>(smem + TAB_OFF / 4 + TAB * 8);     // [8][64] stamps behind the (shrunk) table
    for (int i = threadIdx.x; i < 512; i += XN) xtr_lds[i] = 0;
#endif
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int mpd = wave >> 1, mhh = wave & 1;
    // The two waves of a SIMD (w and w + 4) run the SAME program — rows(a) pos(a) rows(b) pos(b) per chunk — half a phase apart: the
    // waves 0-3 meet the workgroup's barriers behind their positions, the waves 4-7 ("late") behind their rows, so that between two
    // barriers one wave of every SIMD reads LDS and transforms (latency-bound, few vector instructions) while the other splits and
    // multiplies (vector- and matrix-bound).  The late waves carry the transformed rows across the barrier in registers and copy
    // the halo like the others (the vector-memory sequence of both roles is the same: only the barriers sit elsewhere).
    const int late = (X_SKEW && wave >= 4) ? 1 : 0, early = late ^ 1;
    const int nchunk = Cin / XCK;                           // even (Cin % 32 == 0): every item ends on halo pair 1
    constexpr int OOB = (int)0x80000000u;

    // ---- this workgroup's items (the persistent fp32 kernel's order: item0 + jb + i G, XCD-contiguous within a window) ----
    const int G = gridDim.x;
    const int jb = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int my_items = jb < nitems ? (nitems - jb + G - 1) / G : 0;
    if (my_items == 0) return;
    for (int i = tid; i < my_items; i += XN) {
        const int item = item0 + jb + i * G;
#if X_NGFAST
        // the channel group is the FAST index: the groups of one brick are neighbouring items, i.e. workgroups of one XCD at about the
        // same time (jb above), and share the brick's halo through that XCD's L2 instead of fetching it again a whole pass later
        const int ngroups = Cout >> 5;
        int t = item / ngroups;
        const int ng = item - t * ngroups;
#else
        const int ng = item / nbricks;
        int t = item - ng * nbricks;
#endif
        const int bw = t % tilesW; t /= tilesW;
        const int bh = t % tilesH; t /= tilesH;
        const int bd = t % tilesD;
        const int b0 = t / tilesD;
        tab[2 * i] = i32x4{0, ng, b0, bd | (bh << 10) | (bw << 20)};
        const int d0 = bd * BD, h0 = bh * BH, w0 = bw * BW;
        auto range = [](int lo, int hi) { return ((1 << hi) - 1) & ~((1 << lo) - 1); };
        auto mn = [](int a, int b_) { return a < b_ ? a : b_; };
        // byte offset of the halo's voxel (0, 0, 0) = (d0 - 1, h0 - 1, w0 - 1) in the sample and the valid halo coordinates per axis
        tab[2 * i + 1] = i32x4{((d0 - 1) * sd + (h0 - 1) * sh + (w0 - 1)) * Cin * 4,
                               range(d0 == 0 ? 1 : 0, mn(BD + 2, D - d0 + 1)) | (range(h0 == 0 ? 1 : 0, mn(BH + 2, H - h0 + 1)) << 6) |
                                   (range(w0 == 0 ? 1 : 0, mn(BW + 2, W - w0 + 1)) << 16), 0, 0};
    }
    __syncthreads();

    // ---- halo staging: DMA instruction q of wave w fills the slots (q * 8 + w) * 64 + lane of an 8-channel buffer ----
    // (scalar arithmetic throughout the main loop: packed fp32 instructions are slow beside bf16 MFMAs, mix_cost_bf16.hip)
    int* plan = reinterpret_cast<int*>(smem + PLAN_OFF / 4) + tid;      // [k * XN]: k = q (hrel), 3 + q (hm), 6 + q (hoff)
#pragma unroll
    for (int q = 0; q < XDMA; ++q) {
        const int e = (q * 8 + wave) * 64 + lane, gg = e >> 1, par = gg / CP, g = gg % CP;
        const int a_ = g / 32, bb = (g % 32) / 6, c_ = (g % 32) % 6;
        const bool real = bb < 5 && c_ < 5;
        const int hd = 2 * a_ + (par >> 2), hh = 2 * bb + ((par >> 1) & 1), hw = 2 * c_ + (par & 1);
        const int quad = (e & 1) ^ (bb & 1);
        plan[q * XN] = ((hd * sd + hh * sh + hw) * Cin + quad * 4) * 4;
        plan[(3 + q) * XN] = real ? (1 << hd) | (1 << (6 + hh)) | (1 << (16 + hw)) : (1 << 30);
    }
    i32x4 xr = make_rsrc(x, 0);
    int dm_i = 0, dm_c = 0;                                 // the (item, chunk) the next halo copy belongs to
    auto dma_plan = [&](int i) {                            // i < 0: behind the stream — every lane out of range
        int corner = 0, vm = 0;
        if (i >= 0) {
            const i32x4 e1 = tab[2 * i + 1];
            const int b = __builtin_amdgcn_readfirstlane(tab[2 * i][2]);
            corner = __builtin_amdgcn_readfirstlane(e1[0]);
            vm = __builtin_amdgcn_readfirstlane(e1[1]);
            xr = make_rsrc(x + (size_t)b * D * H * W * Cin, (unsigned)(D * H * W * Cin * 4));
        }
#pragma unroll
        for (int q = 0; q < XDMA; ++q) {
            const int m = plan[(3 + q) * XN];
            plan[(6 + q) * XN] = (m & vm) == m ? plan[q * XN] + corner : OOB;
        }
    };
    // a third of the halo of (dm_i, dm_c) -> pair `pr` (both 8-channel buffers); the last third advances the cursor.  Behind the
    // end of the stream the copies stay (static wait counts): every lane out of range, zeros into the pair nobody reads any more.
    auto dma_part = [&](int pr, int part) {
        const unsigned base = lds0 + pr * 2 * RAWB + wave * 1024 + part * 8192;
        const int ho = plan[(6 + part) * XN];
        if (!(X_ABL & 4)) {
            blds16(ho, xr, dm_c * (XCK * 4), base);
            blds16(ho, xr, dm_c * (XCK * 4) + 32, base + RAWB);
        }
        if (part == XDMA - 1) {
            if (++dm_c == nchunk) {
                dm_c = 0;
                ++dm_i;
                dma_plan(dm_i < my_items ? dm_i : -1);
            }
        }
    };

    // ---- transformed weights, split: u3[position 64][chunk][part 3][K half 2][cout][8] bf16; this lane's 16 bytes of a part ----
    const i32x4 ur = make_rsrc(u3, (unsigned)(64 * Cin * Cout * 6));
    const int b_lane = (hsel * Cout + l31) * 16;
    const int part_b = 2 * Cout * 16, chunk_b = 3 * part_b, pos_b = nchunk * chunk_b;
    // this wave's positions: p_first + (phl * 4 + pw) * pos_b; swap: the tensor's position (ph, pd, pw) = p_first + (phl * 16 + pw) * pos_b
    const int p_first = (swap ? (2 * mhh * 4 + mpd) * 4 : (mpd * 4 + 2 * mhh) * 4) * pos_b;
    const int q_hi = swap ? 16 : 4;
    i32x4 Bq[4][3];                                         // [stream position & 3][part h, m, l]
    auto load_b = [&](int slot, int q8, int c, int n0) {
        if (X_ABL & 2) return;
        const int so = p_first + ((q8 >> 2) * q_hi + (q8 & 3)) * pos_b + c * chunk_b + n0 * 16;
        bload16(Bq[slot][0], b_lane, ur, so);
        bload16(Bq[slot][1], b_lane, ur, so + part_b);
        bload16(Bq[slot][2], b_lane, ur, so + 2 * part_b);
    };

    // ---- input transform of this wave: planes da, db (d row pd of B^T), the h rows of its two ph, all four w columns ----
    const int da = mpd == 0 ? 0 : (mpd == 2 ? 2 : 1);
    const int db = mpd == 0 ? 2 : (mpd == 1 ? 2 : (mpd == 2 ? 1 : 3));
    const float sgn = mpd == 1 ? 1.f : -1.f;
    // mhh = 0: ph 0 = x0 - x2, ph 1 = x1 + x2 (keeper x2);  mhh = 1: ph 2 = x2 - x1, ph 3 = x1 - x3 (keeper x1)
    const int ik = mhh == 0 ? 2 : 1, ia = mhh == 0 ? 0 : 2, ib = mhh == 0 ? 1 : 3;
    const float c1 = mhh == 0 ? 1.f : -1.f;
    const int td = l31 >> 4, th = (l31 >> 2) & 3, tw = l31 & 3;
    const int g0 = td * SD + th * SH + tw;
    const int e0 = hsel ^ (th & 1);
    auto pgd = [](int dd) { return (dd & 1) * 4 * CP + (dd >> 1) * SD; };
    auto pgh = [](int i) { return (i & 1) * 2 * CP + (i >> 1) * SH; };
    auto pgw = [](int k) { return (k & 1) * CP + (k >> 1); };
    auto row_base = [&](int dd, int i) { return (2 * (g0 + pgd(dd) + pgh(i)) + (e0 ^ (i >> 1))) * 4; };
    const int rka = row_base(da, ik), rkb = row_base(db, ik);
    const int raa = row_base(da, ia), rab = row_base(db, ia);
    const int rba = row_base(da, ib), rbb = row_base(db, ib);
    // the four positions (ph of group GA, pw = 0..3) of one 8-channel buffer, 4 channels of this lane's quad: per tap the d
    // combination of the group's row and of the keeper row, their h combination (group a: x_ia - x_k, group b: c1 x_ib + x_k), then
    // the w transform.  (No w-transformed keeper is carried from group a to group b: 32 registers this kernel does not have; the
    // vector instruction count is the same, the keeper's 8 LDS reads per buffer are repeated.)
    auto group_rows = [&](const float* R, int xa, int xb, auto ga_c, float (&Y)[4][4]) {
        constexpr bool GA = decltype(ga_c)::value;
        if (X_ABL & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[k][e] = sgn + (float)(k + e);
            return;
        }
        float T[4][4];
        f32x4 ax[4], bx[4], ak[4], bk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ax[k] = *reinterpret_cast<const f32x4*>(&R[xa + pgw(k) * 8]); bx[k] = *reinterpret_cast<const f32x4*>(&R[xb + pgw(k) * 8]);
            ak[k] = *reinterpret_cast<const f32x4*>(&R[rka + pgw(k) * 8]); bk[k] = *reinterpret_cast<const f32x4*>(&R[rkb + pgw(k) * 8]);
        }
        if (X_RB) __builtin_amdgcn_sched_barrier(0);       // (16 reads in flight, then the arithmetic as they arrive: one latency per buffer)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tx = ax[k][e] + sgn * bx[k][e], tk = ak[k][e] + sgn * bk[k][e];
                T[k][e] = GA ? tx - tk : c1 * tx + tk;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Y[0][e] = T[0][e] - T[2][e]; Y[1][e] = T[1][e] + T[2][e]; Y[2][e] = T[2][e] - T[1][e]; Y[3][e] = T[1][e] - T[3][e];
        }
    };

    f32x16 acc[8];                                          // [phl * 4 + pw]: rows = output channels 8 j + 4 hsel + e (register 4 j + e), column = tile l31

    // ---- epilogue addressing: the reader lane (channel quad cq, wo, tile 4 w' + vq) ----
    // (recomputed per item from an opaque copy of the lane index: hoisted out of the item loop these addresses are ~10 live registers
    //  across the main loop, which runs at the 256-register limit)
    const int r_td = wave >> 2, r_th = wave & 3;            // reader tile 4 w' + vq = td * 16 + th * 4 + tw, tw = vq
    auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    // BatchNorm statistic partials (MODE 1): ONE row [2][Cout] per workgroup.  Every reader lane adds its 4 + 4 sums of an item into
    // its OWN cells of `red` (no registers across the main loop); the reduction over the workgroup and the store happen only where
    // the channel group changes (the items of a workgroup are ordered by group) and at the end.
    int st_n0 = -1;
    unsigned st_seen = 0;
    auto stat_zero = [&]() {
        const int le = lane_now();
        float* red_l = red + (wave * 8 + (le >> 3)) * 33 + 4 * (le & 7);
#pragma unroll
        for (int e = 0; e < 4; ++e) { red_l[e] = 0.f; red_l[64 * 33 + e] = 0.f; }
    };
    auto stat_flush = [&]() {                               // the sums of channel group st_n0 -> this workgroup's row
        __syncthreads();
        if (tid < 256) {                                    // thread (which, channel, part) adds 16 sources in a fixed order, a quad its four parts
            const int part = tid & 3, ch = (tid >> 2) & 31, which = tid >> 7;
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) a += red[(which * 64 + part * 16 + m) * 33 + ch];
            a += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
            a += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));
            if (part == 0) {
                float* dst = stat_partial + ((size_t)blockIdx.x * 2 + which) * Cout + st_n0 + ch;
                *dst = (stat_accum || ((st_seen >> (st_n0 >> 5)) & 1)) ? *dst + a : a;
            }
        }
        st_seen |= 1u << (st_n0 >> 5);
        __syncthreads();
        stat_zero();
    };
    if (STATS) stat_zero();

    // ---- prologue of the stream: the first halo, the first two positions' weights ----
    int ci_n0 = __builtin_amdgcn_readfirstlane(tab[0][1]) * 32;    // channel group of the item being multiplied
    dma_plan(0);
#pragma unroll
    for (int i = 0; i < XDMA; ++i) dma_part(0, i);
    load_b(0, 0, 0, ci_n0);
    if (X_PF == 2) load_b(1, 1, 0, ci_n0);
    vm_wait<0>();
    __syncthreads();

    int g = 0;                                              // chunk of the stream
    for (int it = 0; it < my_items; ++it) {
        const i32x4 ce = tab[2 * it];
        const int b = __builtin_amdgcn_readfirstlane(ce[2]);
        const int cpk = __builtin_amdgcn_readfirstlane(ce[3]);
        const int d0 = (cpk & 1023) * BD, h0 = ((cpk >> 10) & 1023) * BH, w0 = (cpk >> 20) * BW;
        const int n0 = ci_n0;
        auto chunk = [&](int c, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            const int pr = g & 1;
            const float* R0 = smem + pr * (2 * RAWB / 4);
            const float* R1 = R0 + RAWB / 4;
            // the chunk the weights two positions ahead of positions 6, 7 belong to
            int nc = c + 1, n0n = n0;
            if (nc == nchunk) {
                nc = 0;
                n0n = __builtin_amdgcn_readfirstlane(tab[2 * (it + 1 < my_items ? it + 1 : it)][1]) * 32;
                ci_n0 = n0n;
            }
            float Y0[4][4], Y1[4][4];                           // [pw][channel of the quad]: the transformed rows of a group, both buffers
            auto position = [&](auto q8_c, const float (&y0)[4], const float (&y1)[4]) {
                constexpr int q8 = decltype(q8_c)::value;
                // weights X_PF positions ahead (the slot's last reader was the position before the running one)
                if (q8 + X_PF < 8) load_b((q8 + X_PF) & 3, q8 + X_PF, c, n0);
                else load_b((q8 + X_PF) & 3, q8 + X_PF - 8, nc, n0n);
                // in flight behind this position's weights: those of the X_PF positions after it (3 each) and, as long as the six halo
                // copies issued at the chunk's start lie behind them (positions < X_PF), those and — at the start of an item — the 4
                // stores of the epilogue before it
                constexpr int cnt = 3 * X_PF + (q8 < X_PF ? ((X_ABL & 4) ? 0 : 6) + (FIRST && !(X_ABL & 8) ? EPI_STORES : 0) : 0);
                vm_wait<cnt>();
                pin3(Bq[q8 & 3][0], Bq[q8 & 3][1], Bq[q8 & 3][2]);
                i32x4 vh, vm_, vl;
                if (X_ABL & 32) {
                    vh = i32x4{__builtin_bit_cast(int, y0[0]), __builtin_bit_cast(int, y0[1]), __builtin_bit_cast(int, y0[2]), __builtin_bit_cast(int, y0[3])};
                    vm_ = i32x4{__builtin_bit_cast(int, y1[0]), __builtin_bit_cast(int, y1[1]), __builtin_bit_cast(int, y1[2]), __builtin_bit_cast(int, y1[3])};
                    vl = vh;
                } else
                    split8(y0, y1, vh, vm_, vl);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const i32x4 wh = Bq[q8 & 3][0], wm = Bq[q8 & 3][1], wl = Bq[q8 & 3][2];
#if X_ABL & 16
                f32x16 a = FIRST ? zero : acc[q8];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += __builtin_bit_cast(float, wh[e] ^ vl[e] ^ wl[e] ^ vh[e] ^ wm[e] ^ vm_[e]);
                acc[q8] = a;
#else
                f32x16 a = mfma_bf16(wh, vl, FIRST ? zero : acc[q8]);      // small terms first
                a = mfma_bf16(wl, vh, a);
                a = mfma_bf16(wm, vm_, a);
                a = mfma_bf16(wh, vm_, a);
                a = mfma_bf16(wm, vh, a);
                acc[q8] = mfma_bf16(wh, vh, a);
#endif
            };
            // (the scheduler is fenced between the phases: left alone it pulls the next group's LDS reads into the positions — the other
            //  wave of the SIMD provides that overlap here — and spills accumulators to make room)
            XTRC(0);
            __builtin_amdgcn_sched_barrier(0);
            // the next chunk's halo (its pair is free: every wave is past rows(b) of the chunk before); the copies have this phase and
            // position 0 to land before the weights of position 2 queue up behind them
#pragma unroll
            for (int i = 0; i < XDMA; ++i) dma_part(pr ^ 1, i);
            group_rows(R0, raa, rab, std::true_type{}, Y0);
            group_rows(R1, raa, rab, std::true_type{}, Y1);
            __builtin_amdgcn_sched_barrier(0);
            XTRC(1);
            wg_barrier_if(late);
            position(std::integral_constant<int, 0>{}, Y0[0], Y1[0]);
            position(std::integral_constant<int, 1>{}, Y0[1], Y1[1]);
            position(std::integral_constant<int, 2>{}, Y0[2], Y1[2]);
            position(std::integral_constant<int, 3>{}, Y0[3], Y1[3]);
            XTRC(2);
            __builtin_amdgcn_sched_barrier(0);
            if (X_SKEW) wg_barrier_if(early);
            group_rows(R0, rba, rbb, std::false_type{}, Y0);
            group_rows(R1, rba, rbb, std::false_type{}, Y1);
            __builtin_amdgcn_sched_barrier(0);
            XTRC(3);
            wg_barrier_if(late);                            // (every wave is done with this chunk's halo)
            position(std::integral_constant<int, 4>{}, Y0[0], Y1[0]);
            position(std::integral_constant<int, 5>{}, Y0[1], Y1[1]);
            position(std::integral_constant<int, 6>{}, Y0[2], Y1[2]);
            position(std::integral_constant<int, 7>{}, Y0[3], Y1[3]);
            __builtin_amdgcn_sched_barrier(0);
            XTRC(4);
            wg_barrier_if(early);                           // (... and the next chunk's has landed: this wave's copies by the waits above)
            XTRC(5);
            ++g;
        };
        chunk(0, std::true_type{});
        for (int c = 1; c < nchunk; ++c) chunk(c, std::false_type{});

#if X_ABL & 8
        {
            float sm = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sm += acc[q][0] + acc[q][7];
            if (sm == 12345.678f) z[tid] = sm + (float)(b + d0 + h0 + w0 + n0);
            wg_barrier();
            continue;
        }
#endif
        // ---- output transform: w and this wave's half of h in registers, d across the waves through LDS, one pass per ho ----
        //   mhh = 0 (ph 0, 1): P[ho 0] = y(ph 0) + y(ph 1), P[ho 1] = y(ph 1);  mhh = 1 (ph 2, 3): P[ho 0] = y(ph 2), P[ho 1] = y(ph 2) + y(ph 3)
        //   out_h[0] = P0[0] + P1[0], out_h[1] = P0[1] - P1[1];  S_pd = that;  out[do 0] = S_0 + S_1 + S_2, out[do 1] = S_1 - S_2 - S_3
        f32x4 P1[4][2];                                     // [j][wo]: the ho = 1 partials wait for the second pass
        const int le = lane_now();
        const int cq = le & 7, rwo = (le >> 3) & 1, vq = le >> 4, r_tw = vq;
        const int st_lane = ((2 * r_td * sd + 2 * r_th * sh + 2 * r_tw + rwo) * Cout + 4 * cq) * 4;
        float* red_l = red + (wave * 8 + (le >> 3)) * 33 + 4 * cq;
        float* exw = ex + (wave * 32 + (le & 31)) * TS + 4 * (le >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 y[2][2];
#pragma unroll
            for (int phl = 0; phl < 2; ++phl) {
                f32x4 m[4];
#pragma unroll
                for (int pw = 0; pw < 4; ++pw)
                    m[pw] = f32x4{acc[phl * 4 + pw][4 * j], acc[phl * 4 + pw][4 * j + 1], acc[phl * 4 + pw][4 * j + 2], acc[phl * 4 + pw][4 * j + 3]};
                y[phl][0] = (m[0] + m[1]) + m[2];
                y[phl][1] = (m[1] - m[2]) - m[3];
            }
#pragma unroll
            for (int wo = 0; wo < 2; ++wo) {
                const f32x4 sum = y[0][wo] + y[1][wo];
                const f32x4 p0 = mhh == 0 ? sum : y[0][wo];
                P1[j][wo] = mhh == 0 ? y[1][wo] : sum;
                *reinterpret_cast<f32x4*>(&exw[wo * 32 + 8 * j]) = p0;
            }
        }
        XTR(40);
        const int OD = D / 2, OH = H / 2, OW = W / 2;        // (MODE 3: the pooled tensor)
        float* zb = z + (size_t)b * (POOL ? OD * OH * OW : D * H * W) * Cout;
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(zb, 0, (POOL ? OD * OH * OW : D * H * W) * Cout * 4, 0x00020000);
        const bool full = d0 + BD <= D && h0 + BH <= H && w0 + BW <= W;
        if (STATS && stat_partial != nullptr && st_n0 != n0) {
            if (st_n0 >= 0) stat_flush();
            st_n0 = n0;
        }
        if (AFFINE && st_n0 != n0) {                         // a new channel group: its scale / shift into LDS (the statistics' cells are free here)
            __syncthreads();
            if (tid < 64) red[tid] = tid < 32 ? aff_scale[n0 + tid] : aff_shift[n0 + tid - 32];
            __syncthreads();
            st_n0 = n0;
        }
        f32x4 a_sc = {1.f, 1.f, 1.f, 1.f}, a_sh = {0.f, 0.f, 0.f, 0.f}, pmax = {0.f, 0.f, 0.f, 0.f};
        if (AFFINE) {
            a_sc = *reinterpret_cast<const f32x4*>(&red[4 * cq]);
            a_sh = *reinterpret_cast<const f32x4*>(&red[32 + 4 * cq]);
        }
        const int gd0 = d0 + 2 * r_td, gh0 = h0 + 2 * r_th, gw = w0 + 2 * r_tw + rwo;
        const int st_item = ((d0 * sd + h0 * sh + w0) * Cout + n0) * 4;
        const float* exr = ex + (4 * wave + vq) * TS + rwo * 32 + 4 * cq;
        auto pass = [&](int ho) {
            wg_barrier();                                   // (lgkmcnt(0) first: this wave's exchange writes)
            XTR(41 + 4 * ho);
            f32x4 S[4];
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&exr[(2 * pd) * 32 * TS]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&exr[(2 * pd + 1) * 32 * TS]);
                S[pd] = ho == 0 ? lo + hi : lo - hi;
            }
            f32x4 o[2];
            o[0] = (S[0] + S[1]) + S[2];
            o[1] = (S[1] - S[2]) - S[3];
            const int gh = gh0 + ho;
            if (AFFINE) {
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = o[dd][e] * a_sc[e] + a_sh[e];
                        o[dd][e] = yv > 0.f ? yv : yv * slope;
                    }
            }
            if (POOL) {
                // the window = the tile: d pair here, h pair = the two passes, w pair = lane ^ 8; ONE store per item (pass 1, wo = 0 lanes)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(o[0][e], o[1][e]);
                    pmax[e] = ho == 0 ? v : fmaxf(pmax[e], v);
                }
                if (ho == 1) {
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // row_ror:8 = lane ^ 8, as assembly: with the builtin (mov_dpp and update_dpp alike) the compiler emitted ONE
                        // v_mov_b32_dpp and used its result for all four channels.  (s_nop: a DPP read needs two wait states behind the
                        // VALU write of its source, which the hazard pass does not see inside an asm block.)
                        float other;
                        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(other) : "v"(pmax[e]));
                        m[e] = fmaxf(pmax[e], other);
                    }
                    const int od = (d0 >> 1) + r_td, oh = (h0 >> 1) + r_th, ow = (w0 >> 1) + r_tw;
                    const bool okp = rwo == 0 && od < OD && oh < OH && ow < OW;
                    const int osd = swap ? OW : OH * OW, osh = swap ? OD * OW : OW;
                    const int voff = ((r_td * osd + r_th * osh + r_tw) * Cout + 4 * cq) * 4;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tmf_u32x4, m), zr, okp ? voff : OOB,
                                                           (((d0 >> 1) * osd + (h0 >> 1) * osh + (w0 >> 1)) * Cout + n0) * 4, 0);
                    store_guard();
                }
                return;
            }
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const bool ok = full || (gd0 + dd < D && gh < H && gw < W);
                // (always issued — the counted waits of the next item rely on exactly four stores per epilogue; a lane outside the
                // volume is out of the resource's range and dropped)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tmf_u32x4, o[dd]), zr, ok ? st_lane : OOB,
                                                       st_item + (dd * sd + ho * sh) * Cout * 4, 0);
                store_guard();                              // (its data registers die right behind the store)
                if (STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ok ? o[dd][e] : 0.f;
                        red_l[e] += v;
                        red_l[64 * 33 + e] += v * v;
                    }
                }
            }
        };
        pass(0);
        XTR(42);
        wg_barrier();                                       // every reader is done with the ho = 0 partials
        XTR(43);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int wo = 0; wo < 2; ++wo) *reinterpret_cast<f32x4*>(&exw[wo * 32 + 8 * j]) = P1[j][wo];
        XTR(44);
        pass(1);
        XTR(46);
        wg_barrier();                                       // the exchange (= halo pair 1) is free for the next item's copies
        XTR(47);
    }
    vm_wait<0>();
#ifdef TMF_WINOX_TRACE
    __syncthreads();
    if (blockIdx.x == 77) for (int i = threadIdx.x; i < 512; i += XN) g_winox_phases[i] = xtr_lds[i];
#endif
    if constexpr (STATS) {
        if (stat_partial != nullptr) {
            if (st_n0 >= 0) stat_flush();
            if (!stat_accum) {
                for (int ng = 0; ng < Cout / 32; ++ng)
                    if (!((st_seen >> ng) & 1) && tid < 64) stat_partial[((size_t)blockIdx.x * 2 + (tid >> 5)) * Cout + ng * 32 + (tid & 31)] = 0.f;
                if (blockIdx.x == 0)
                    for (int i = (int)gridDim.x * 2 * Cout + tid; i < stat_rows * 2 * Cout; i += XN) stat_partial[i] = 0.f;
            }
        }
    }
}

}  // namespace
#ifdef TMF_WINOX_TRACE
extern "C" int tmf_winox_trace_read(long long* phases) {
    return (int)hipMemcpyFromSymbol(phases, HIP_SYMBOL(g_winox_phases), sizeof(long long) * 8 * 64);
}
#endif
namespace {
int g_wino_x = -1;
int wino_x_mode() {
    if (const int o = tmf_algo_override()) return (o & TMF_SNET_ALGO_WINO_X) ? 1 : 0;
    if (g_wino_x < 0) {
        const char* e = getenv("TMF_WINO_X");
        g_wino_x = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_wino_x;
}

}  // namespace

int tmf_wino_x_set(int v) { g_wino_x = v ? 1 : 0; return TMF_OK; }
extern "C" int tmf_wino_x_mode(void) { return wino_x_mode(); }

// does the split kernel take this launch?  (4x8x8 bricks of one sample only: the folded four-sample geometry of the small deep
// volumes stays on the fp32 kernel; two 8-channel buffers per chunk and an even chunk count: cin % 32 == 0)
int tmf_winox_takes(int B, int D, int H, int W, int cin, int cout, int geom) {
    return wino_x_mode() && geom == 0 && cin % 32 == 0 && cout % 32 == 0 && cout <= 1024;
}

// items per sample of a volume, and whether the transposed form (4-voxel side along the tensor's h) has fewer
long tmf_winox_items(int D, int H, int W, int* swap) {
    const long t_n = (long)tmf_cdiv(D, BD) * tmf_cdiv(H, BH) * tmf_cdiv(W, BW);
    const long t_s = (long)tmf_cdiv(H, BD) * tmf_cdiv(D, BH) * tmf_cdiv(W, BW);
    static const bool allow = !(getenv("TMF_WINOX_SWAP") && atoi(getenv("TMF_WINOX_SWAP")) == 0);
    const int sw = allow && t_s < t_n ? 1 : 0;
    if (swap) *swap = sw;
    return sw ? t_s : t_n;
}

int tmf_winox_launch(const char* what, const float* x, const unsigned short* u3, float* z, float* stat_partial, int B, int Dt, int Ht,
                     int W, int cin, int cout, int ncu, hipStream_t stream, const float* scale, const float* shift, float slope, int pool) {
    int swap = 0;
    tmf_winox_items(Dt, Ht, W, &swap);
    const int D = swap ? Ht : Dt, H = swap ? Dt : Ht;        // kernel space
    const int tilesD = tmf_cdiv(D, BD), tilesH = tmf_cdiv(H, BH), tilesW = tmf_cdiv(W, BW);
    TMF_REQUIRE(tilesD < 1024 && tilesH < 1024 && tilesW < 1024, TMF_E_SHAPE, "%s: more than 1023 bricks along one axis", what);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "%s: one sample exceeds 2^29 elements (32-bit byte offsets)", what);
    TMF_REQUIRE((long)64 * cin * cout * 6 < (1L << 31), TMF_E_SHAPE, "%s: weight tensor exceeds 2^31 bytes", what);
    const long nbricks = (long)B * tilesD * tilesH * tilesW;
    const long nitems = nbricks * (cout / 32);
    TMF_REQUIRE(nitems < (1L << 30), TMF_E_SHAPE, "%s: too many bricks", what);
    int rc;
    auto launch = [&](auto k, int mode_stats) -> int {
        if ((rc = tmf_allow_lds(k, X_LDS_BYTES, what))) return rc;
        const long per_launch = (long)ncu * TAB;
        for (long i0 = 0; i0 < nitems; i0 += per_launch) {
            const long n = nitems - i0 < per_launch ? nitems - i0 : per_launch;
            int grid = (int)(n < ncu ? n : ncu);
            if (n > ncu) {                                  // as many workgroups as the number of rounds needs
                const long rounds = (n + ncu - 1) / ncu;
                grid = (int)((n + rounds - 1) / rounds);
            }
            hipLaunchKernelGGL(k, dim3(grid), dim3(XN), X_LDS_BYTES, stream, x, u3, z, stat_partial, B, D, H, W, cin, cout,
                               tilesD, tilesH, tilesW, (int)nbricks, (int)i0, (int)n, ncu, i0 > 0 ? 1 : 0, scale, shift, slope, swap);
            if ((rc = tmf_launch_result(what))) return rc;
        }
        return TMF_OK;
    };
    if (scale != nullptr) return pool == TMF_POOL_MAX2 ? launch(conv3d_winox_kernel<3>, 3) : launch(conv3d_winox_kernel<2>, 2);
    if (stat_partial != nullptr) return launch(conv3d_winox_kernel<1>, 1);
    return launch(conv3d_winox_kernel<0>, 0);
}
