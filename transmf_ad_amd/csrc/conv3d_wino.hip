// conv3d_wino.hip — 3x3x3 convolution (forward / data gradient) in the Winograd form F(2x2x2, 3x3x3), exact-fp32
// arithmetic on v_mfma_f32_32x32x2_f32, for gfx950 (MI355X).
//
// Why.  The fp32 train step is bound by the fp32 matrix pipe (0.78 / 0.83 busy over a whole step, every
// vector instruction paid in matrix time), and the direct kernels of conv3d_mfma.hip already run at 0.72-0.86 of it: the
// only thing left to remove is the matrix work itself.  With 2x2x2 output tiles a 3x3x3 convolution needs 64 products per
// tile, input and output channel instead of 216 (x 3.375 less), at the price of 192 additions per (tile, input channel) and
// 112 per (tile, output channel) — 10-15 % of the remaining matrix time at 32-64 channels.
//
//   y = A^T [ (G g G^T) .* (B^T d B) ] A   along each of the three axes,
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
//
// One workgroup (8 waves, one per CU: the accumulators are half the CU's register file) owns a 4x8x8 brick of output
// voxels = 32 tiles and 32 output channels:
//   * per chunk of 8 input channels the 6x10x10 halo goes global -> registers -> LDS (zero fill from the buffer range
//     check), is transformed to V[64 positions][32 tiles][8 channels] in LDS (thread = (tile, channel pair, row pd of the
//     d transform): 32 ds_read_b64, 48 packed adds, 16 ds_write_b64), and then multiplied: wave w owns the 8 positions
//     (pd = w / 2, ph in {2 (w & 1), +1}, pw = 0..3) with one 32 x 32 accumulator each; A operand = one ds_read_b128 per
//     position (the K permutation of conv3d_mfma.hip: lanes 0-31 channels 0-3, lanes 32-63 channels 4-7), B operand =
//     the transformed weights straight from L2 (each position's weights are used by exactly one wave of the workgroup: a
//     ring in LDS would share nothing), one buffer_load_b128 per position, requested before the transform phase;
//   * epilogue: the w transform and this wave's half of the h transform in registers, one exchange through LDS (128 KB),
//     the d transform on the reading side, 128-byte channel rows to z, BatchNorm statistic partials as in the direct
//     kernels (fixed order: results are bit-reproducible run to run).
//
// The transformed weights U[p][cin/8][2][cout][4] come from tmf_pack_conv_weights_wino (computed in fp64, rounded once).
// Numerics: transforms, products and sums in fp32; against the fp64 reference the error of z is about twice the direct
// kernels' (cancellation in the output transform), measured on the fixtures in tools/winograd_numerics.py and gated by the
// golden tests at the same tolerances as the direct path.
//
// Replaces aten::conv3d / convolution_backward (input gradient) at /root/reference/models/networks.py:28,31,37,40,46.
#include <type_traits>
#include "tmf_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TD = 4, TH = 8, TW = 8;                 // output brick
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;  // input halo
constexpr int CK = 8;                                 // input channels per chunk
constexpr int NTHR = 512;
// Halo of one chunk in LDS, in 16-byte slots (4 channels of one voxel).  The two channel quads of a voxel sit side by side (one
// 32-byte piece of global memory = two neighbouring DMA lanes), the voxels are sorted by the PARITY of their halo coordinates
// and then by their halves, and every second h row stores its two quads swapped:
//     G(hd, hh, hw) = ((hd & 1) * 4 + (hh & 1) * 2 + (hw & 1)) * 96 + (hd >> 1) * 32 + (hh >> 1) * 6 + (hw >> 1)
//     slot(voxel, quad) = 2 G + (quad ^ ((hh >> 1) & 1))
// The 32 tiles of the brick (origins 2 apart on every axis) then read, for any tap, groups G that are consecutive along w, 6
// apart along h and 32 apart along d: in both 16-lane service groups of a ds_read_b128 ({0-3,12-15,20-27} / {4-11,16-19,
// 28-31}: tile rows th {0,3,1,2} / {1,2,0,3}) the even rows hit the 8 even-or-odd slots 2 (6 th + tw mod 8) + e and the odd rows
// the other 8 — conflict-free (tests/test_host_cpu.py checks the map exhaustively).
constexpr int GROUPS = 768, SLOTS = 2 * GROUPS;                   // 1 536 slots = 24 KB per buffer (1 200 of them real)
constexpr int RAW_BYTES = SLOTS * 16;
constexpr int NDMA = SLOTS / NTHR;                                // 3 LDS-DMA instructions per wave and chunk
constexpr int B_OFF = 2 * RAW_BYTES;                              // transformed weights of the running chunk: [wave][position 8][lane 64][4]
constexpr int B_BYTES = 8 * 8 * 64 * 16;                          // 64 KB (every wave reads only its own 8 KB: no barrier for it)
constexpr int EX_FLOATS = 8 * 8 * 2 * 64 * 4;                     // exchange [wave][r pair][ho][lane][wo, rr] — aliases halo + weights
constexpr int RED_OFF = EX_FLOATS;                                // statistic scratch [wave][32][2]
constexpr int LDS_FLOATS = RED_OFF + 8 * 32 * 2;
constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
static_assert(B_OFF + B_BYTES <= EX_FLOATS * 4, "LDS carving");

// group offsets of a tap (dd, i, k) relative to the tile's own group td * 32 + th * 6 + tw
__device__ __forceinline__ constexpr int ogd(int dd) { return (dd & 1) * 384 + (dd >> 1) * 32; }
__device__ __forceinline__ constexpr int ogh(int i) { return (i & 1) * 192 + (i >> 1) * 6; }
__device__ __forceinline__ constexpr int ogw(int k) { return (k & 1) * 96 + (k >> 1); }

// LDS-DMA of 16 bytes per lane through a buffer resource: LDS byte = lds_wave_base + 16 * lane <- base + voff + soff; a
// lane outside the range delivers zeros (the same helper and the same reasons as conv3d_bf16.hip: the compiler does not
// see these copies, the kernel waits for them itself before the barrier that publishes the buffer)
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void blds16(int voff, i32x4 rsrc, int soff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef TMF_WINO_TRACE
// instrumented build (tools/wino_trace.py): shader-clock stamps of every block (start, end, hardware ids) and of the phases of one
__device__ long long g_wino_blocks[8192 * 4];
__device__ long long g_wino_phases[8 * 64];
#define TRB(i, v) do { if (blockIdx.y == 0 && tid == 0 && blockIdx.x < 8192) g_wino_blocks[blockIdx.x * 4 + (i)] = (v); } while (0)
#define TRP(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 1500 && blockIdx.y == 0 && lane == 0) g_wino_phases[wave * 64 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TRB(i, v)
#define TRP(i)
#endif

template <bool STATS>
__global__ __launch_bounds__(NTHR) void conv3d_wino_kernel(
    const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ z, float* __restrict__ stat_partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ex = smem;
    float* red = smem + RED_OFF;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    TRB(0, (long long)__builtin_readcyclecounter());
    TRB(2, (long long)__builtin_amdgcn_s_getreg(63492));
    TRB(3, (long long)__builtin_amdgcn_s_getreg(63508));
    TRP(0);

    const int tile = xcd_contiguous(blockIdx.x, ntiles);
    int t = tile;
    const int bw = t % tilesW; t /= tilesW;
    const int bh = t % tilesH; t /= tilesH;
    const int bd = t % tilesD;
    const int b = t / tilesD;
    const int d0 = bd * TD, h0 = bh * TH, w0 = bw * TW;
    const int n0 = blockIdx.y * 32;
    const int nchunk = Cin / CK;

    constexpr int OOB = (int)0x80000000u;
    const float* xb = x + (size_t)b * D * H * W * Cin;
    const i32x4 xr = make_rsrc(xb, (unsigned)(D * H * W * Cin * 4));

    // ---- halo staging: DMA instruction q of wave w fills the slots (q * 8 + w) * 64 + lane, i.e. the groups (q * 8 + w) * 32 +
    // (lane >> 1): row (hh >> 1, hw >> 1) = the lane's own (lane >> 1) / 6, % 6 in every instruction, parity class and d half
    // = (q * 8 + w) / 3, % 3 — wave-uniform ----
    int hoff[NDMA];
    {
        const int r = lane >> 1, bb = r / 6, c = r % 6;
        const int quad = (lane & 1) ^ (bb & 1);
        const bool real = bb < 5 && c < 5;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int g32 = q * 8 + wave, par = g32 / 3, a = g32 % 3;
            const int hd = 2 * a + (par >> 2), hh = 2 * bb + ((par >> 1) & 1), hw = 2 * c + (par & 1);
            const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
            const bool ok = real && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            hoff[q] = ok ? (((gd * H + gh) * W + gw) * Cin + quad * 4) * 4 : OOB;
        }
    }
    auto stage = [&](int c) {                       // halo of chunk c -> buffer c & 1
        const unsigned base = lds0 + (c & 1) * RAW_BYTES + wave * 1024;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) blds16(hoff[q], xr, c * (CK * 4), base + q * 8192);
    };

    // ---- this wave's part of the input transform and of the products ----
    // wave = (pd, half of ph): the 8 positions (pd, ph = 2 mhh + {0, 1}, pw = 0..3), one 32 x 32 accumulator each.  The lane
    // computes the A operands of ITS tile (l31) and channel quad (hsel) in registers, straight from the halo:
    //   along d: row pd of B^T = d_a + sgn d_b;   along w: all four;   along h: the two rows this wave multiplies —
    //   mhh = 0: ph0 = x0 - x2, ph1 = x1 + x2 (keeper x2);  mhh = 1: ph2 = x2 - x1, ph3 = x1 - x3 = -(x3 - x1) (keeper x1)
    const int mpd = wave >> 1, mhh = wave & 1;
    const int da = mpd == 0 ? 0 : (mpd == 2 ? 2 : 1);
    const int db = mpd == 0 ? 2 : (mpd == 1 ? 2 : (mpd == 2 ? 1 : 3));
    const float sgn = mpd == 1 ? 1.f : -1.f;
    const int ik = mhh == 0 ? 2 : 1, ia = mhh == 0 ? 0 : 2, ib = mhh == 0 ? 1 : 3;
    const float c1 = mhh == 0 ? 1.f : -1.f;                                  // A[1] = c1 * x_ib + x_keeper
    const int td = l31 >> 4, th = (l31 >> 2) & 3, tw = l31 & 3;
    // slot of the tile's voxel at tap (dd, i, k) = 2 (G0 + ogd + ogh + ogw) + (hsel ^ ((th + (i >> 1)) & 1)); in floats x 4
    const int g0 = td * 32 + th * 6 + tw;
    const int e0 = hsel ^ (th & 1);
    auto row_base = [&](int dd, int i) { return (2 * (g0 + ogd(dd) + ogh(i)) + (e0 ^ (i >> 1))) * 4; };
    const int rka = row_base(da, ik), rkb = row_base(db, ik);
    const int raa = row_base(da, ia), rab = row_base(db, ia);
    const int rba = row_base(da, ib), rbb = row_base(db, ib);
    const int p0 = mpd * 16 + mhh * 8;
    const int b_lane = (hsel * Cout + n0 + l31) * 16;        // bytes
    const i32x4 ur = make_rsrc(u, (unsigned)(64 * Cin * Cout * 4));
    const unsigned bl0 = lds0 + B_OFF + wave * 8192;
    auto stage_b = [&](int c, int q) {              // this wave's weights of (chunk c, position q) -> its own 8 KB
        blds16(b_lane, ur, (((p0 + q) * nchunk + c) * 2 * Cout) * 16, bl0 + q * 1024);
    };
    const float* Bl = smem + (B_OFF + wave * 8192) / 4 + lane * 4;
    f32x16 acc[8];

    stage(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_b(0, q);
    TRP(1);
    dma_wait();
    __syncthreads();
    TRP(2);

    // Two waves share a SIMD (w and w + 4) and the matrix pipe is the resource to keep busy, so the two halves of the workgroup
    // run the chunk loop half a period apart: between two barriers the waves 0-3 transform chunk c and THEN multiply it, the
    // waves 4-7 FIRST multiply chunk c - 1 and then transform chunk c — on every SIMD one wave's loads / adds run beside the
    // other wave's MFMAs (one more barrier interval per brick; measured on the in-kernel timeline: tools/wino_trace.py).
    f32x4 A0[4], A1[4];
    auto transform = [&](int c) {                   // halo buffer c & 1 -> this wave's A operands of chunk c
        if (c + 1 < nchunk) stage(c + 1);           // (its buffer was last read before the barrier two intervals ago)
        const float* R = smem + (c & 1) * (RAW_BYTES / 4);
        const f32x4 s4 = {sgn, sgn, sgn, sgn};
        auto wrow = [&](int pa, int pb, f32x4 (&wv)[4]) {                    // one h row: d combination, then the w transform
            f32x4 tv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tv[k] = *reinterpret_cast<const f32x4*>(&R[pa + ogw(k) * 8]) + s4 * *reinterpret_cast<const f32x4*>(&R[pb + ogw(k) * 8]);
            wv[0] = tv[0] - tv[2]; wv[1] = tv[1] + tv[2]; wv[2] = tv[2] - tv[1]; wv[3] = tv[1] - tv[3];
        };
        f32x4 wk[4];
        wrow(rka, rkb, wk);
        wrow(raa, rab, A0);
        wrow(rba, rbb, A1);
        const f32x4 c4 = {c1, c1, c1, c1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            A0[k] = A0[k] - wk[k];
            A1[k] = c4 * A1[k] + wk[k];
        }
    };
    auto multiply = [&](int c, auto first_c) {      // chunk c: A operands x the weights in this wave's LDS region
        constexpr bool FIRST = decltype(first_c)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bl[k * 256]), b1 = *reinterpret_cast<const f32x4*>(&Bl[(4 + k) * 256]);
            if (c + 1 < nchunk) {                   // both reads have returned: the next chunk's weights for these two positions
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stage_b(c + 1, k);
                stage_b(c + 1, 4 + k);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {           // (the first chunk starts its sums from the literal 0: no zeroing moves)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[k][s], b0[s], (FIRST && s == 0) ? zero : acc[k], 0, 0, 0);
                acc[4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[k][s], b1[s], (FIRST && s == 0) ? zero : acc[4 + k], 0, 0, 0);
            }
        }
    };
    auto sync = [&]() {
        dma_wait();
        __syncthreads();                            // the next halo and weights are in LDS; every wave is done with the running halo
    };
    if (wave < 4) {
        TRP(3);
        transform(0);
        TRP(4);
        multiply(0, std::true_type{});
        TRP(5);
        sync();
        for (int c = 1; c < nchunk; ++c) {
            TRP(3 + 5 * (c & 3));
            transform(c);
            TRP(4 + 5 * (c & 3));
            multiply(c, std::false_type{});
            TRP(5 + 5 * (c & 3));
            sync();
        }
        TRP(23);
        sync();
    } else {
        TRP(3);
        transform(0);
        TRP(4);
        sync();
        TRP(5);
        multiply(0, std::true_type{});
        TRP(6);
        if (nchunk > 1) transform(1);
        TRP(7);
        sync();
        for (int c = 2; c <= nchunk; ++c) {
            TRP(3 + 5 * ((c - 1) & 3));
            multiply(c - 1, std::false_type{});
            TRP(4 + 5 * ((c - 1) & 3));
            if (c < nchunk) transform(c);
            TRP(5 + 5 * ((c - 1) & 3));
            sync();
        }
    }

    // ---- output transform ----
    // in registers: along w (complete), along h (this wave's two rows ph: the reader adds / subtracts the halves)
    //   mhh = 0 (ph 0, 1): P[ho = 0] = y(ph0) + y(ph1), P[ho = 1] = y(ph1)
    //   mhh = 1 (ph 2, 3): P[ho = 0] = y(ph2),          P[ho = 1] = y(ph2) + y(ph3)    (enters out_h1 with a minus)
    // exchange float4 (wave, r pair rp, ho, lane) = {P[ho][wo 0][2 rp], P[ho][0][2 rp + 1], P[ho][1][2 rp], P[ho][1][2 rp + 1]}
    auto exchange = [&](auto first_half) {
        constexpr bool H0 = decltype(first_half)::value;
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            f32x2 y[2][2];                                       // [phl][wo], the pair = accumulator rows 2 rp, 2 rp + 1
#pragma unroll
            for (int phl = 0; phl < 2; ++phl) {
                f32x2 m[4];
#pragma unroll
                for (int pw = 0; pw < 4; ++pw) m[pw] = f32x2{acc[phl * 4 + pw][2 * rp], acc[phl * 4 + pw][2 * rp + 1]};
                y[phl][0] = (m[0] + m[1]) + m[2];
                y[phl][1] = (m[1] - m[2]) - m[3];
            }
            const f32x2 sum0 = y[0][0] + y[1][0], sum1 = y[0][1] + y[1][1];
            const f32x2 p00 = H0 ? sum0 : y[0][0], p01 = H0 ? sum1 : y[0][1];       // P[ho 0][wo]
            const f32x2 p10 = H0 ? y[1][0] : sum0, p11 = H0 ? y[1][1] : sum1;       // P[ho 1][wo]
            *reinterpret_cast<f32x4*>(&ex[(((wave * 8 + rp) * 2 + 0) * 64 + lane) * 4]) = f32x4{p00[0], p00[1], p01[0], p01[1]};
            *reinterpret_cast<f32x4*>(&ex[(((wave * 8 + rp) * 2 + 1) * 64 + lane) * 4]) = f32x4{p10[0], p10[1], p11[0], p11[1]};
        }
    };
    TRP(30);
    if (mhh == 0) exchange(std::true_type{});
    else exchange(std::false_type{});
    TRP(31);
    __syncthreads();
    TRP(32);

    // reader: wave w' takes the accumulator rows r = 2 w' + rr; S_pd[ho] = P(pd, 0)[ho] +- P(pd, 1)[ho];
    // out[do 0] = S_0 + S_1 + S_2, out[do 1] = S_1 - S_2 - S_3
    f32x4 outv[2][2];                                        // [do][ho] -> {wo 0 rr 0, wo 0 rr 1, wo 1 rr 0, wo 1 rr 1}
    {
        f32x4 S[4][2];
#pragma unroll
        for (int pd = 0; pd < 4; ++pd)
#pragma unroll
            for (int ho = 0; ho < 2; ++ho) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&ex[((((2 * pd) * 8 + wave) * 2 + ho) * 64 + lane) * 4]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&ex[((((2 * pd + 1) * 8 + wave) * 2 + ho) * 64 + lane) * 4]);
                S[pd][ho] = ho == 0 ? lo + hi : lo - hi;
            }
#pragma unroll
        for (int ho = 0; ho < 2; ++ho) {
            outv[0][ho] = (S[0][ho] + S[1][ho]) + S[2][ho];
            outv[1][ho] = (S[1][ho] - S[2][ho]) - S[3][ho];
        }
    }
    TRP(33);
    // accumulator row r -> tile (r & 3) + 8 (r >> 2) + 4 hsel; with r = 2 w' + rr:
    //   tile w = 2 (w' & 1) + rr, tile h = hsel + 2 ((w' >> 1) & 1), tile d = w' >> 2
    // voxel w = w0 + 4 (w' & 1) + 2 rr + wo (four consecutive), h = h0 + 2 hsel + 4 ((w' >> 1) & 1) + ho, d = d0 + 2 (w' >> 2) + do
    float s1 = 0.f, s2 = 0.f;
    {
        float* zb = z + (size_t)b * D * H * W * Cout;
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(zb, 0, D * H * W * Cout * 4, 0x00020000);
        const int co = n0 + l31;
        const int gwb = w0 + 4 * (wave & 1), ghb = h0 + 2 * hsel + 4 * ((wave >> 1) & 1), gdb = d0 + 2 * (wave >> 2);
        const bool full = d0 + TD <= D && h0 + TH <= H && w0 + TW <= W;
        auto put = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int ho = 0; ho < 2; ++ho) {
                    const int gd = gdb + dd, gh = ghb + ho;
                    const bool row_ok = FULL || (gd < D && gh < H);
                    const int rowoff = ((gd * H + gh) * W + gwb) * Cout + co;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int wi = 2 * (e & 1) + (e >> 1);          // e = wo * 2 + rr -> w offset 2 rr + wo
                        const bool ok = row_ok && (FULL || gwb + wi < W);
                        float v = outv[dd][ho][e];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), zr, ok ? (rowoff + wi * Cout) * 4 : OOB, 0, 2);
                        if (STATS) {
                            if (!FULL) v = ok ? v : 0.f;
                            s1 += v;
                            s2 += v * v;
                        }
                    }
                }
        };
        if (full) put(std::true_type{});
        else put(std::false_type{});
    }
    TRP(34);
    if constexpr (STATS) {
        if (stat_partial != nullptr) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (hsel == 0) {
                red[(wave * 32 + l31) * 2 + 0] = s1;
                red[(wave * 32 + l31) * 2 + 1] = s2;
            }
            __syncthreads();
            if (tid < 32) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) { a1 += red[(m * 32 + tid) * 2]; a2 += red[(m * 32 + tid) * 2 + 1]; }
                stat_partial[((size_t)tile * 2 + 0) * Cout + n0 + tid] = a1;
                stat_partial[((size_t)tile * 2 + 1) * Cout + n0 + tid] = a2;
            }
        }
    }
    TRP(35);
    TRB(1, (long long)__builtin_readcyclecounter());
}

// Transformed weights from the reference tensor w[cout][cin][3][3][3], one thread per (co, ci), fp64 inside:
//   fwd  [p][cin / 8][2][cout][4]  = U_p(w[co][ci])        input channel ci = 8 g + 4 hs + s
//   dgrad[p'][cout / 8][2][cin][4] = U_p(w[co][ci])        p' = p with every axis index mapped 0 <-> 3 (the flipped
//                                                           kernel: G's rows 0 / 3 swap, rows 1 / 2 are symmetric),
//                                                           input channel co, output channel ci
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                                        float* __restrict__ dgrad, int cout, int cin) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cout * cin) return;
    const int co = e % cout, ci = e / cout;
    const float* src = w + ((size_t)co * cin + ci) * 27;
    double g[3][3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[a][i][j] = (double)src[(a * 3 + i) * 3 + j];
    auto G = [](double x0, double x1, double x2, int row) {
        return row == 0 ? x0 : (row == 1 ? 0.5 * (x0 + x1 + x2) : (row == 2 ? 0.5 * (x0 - x1 + x2) : x2));
    };
    double t1[4][3][3], t2[4][4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) t1[p][i][j] = G(g[0][i][j], g[1][i][j], g[2][i][j], p);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) t2[p][q][j] = G(t1[p][0][j], t1[p][1][j], t1[p][2][j], q);
    const int fl[4] = {3, 1, 2, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (float)G(t2[p][q][0], t2[p][q][1], t2[p][q][2], r);
                if (fwd != nullptr) {
                    const int pos = (p * 4 + q) * 4 + r;
                    fwd[((((size_t)pos * (cin / 8) + ci / 8) * 2 + (ci >> 2 & 1)) * cout + co) * 4 + (ci & 3)] = v;
                }
                if (dgrad != nullptr) {
                    const int pos = (fl[p] * 4 + fl[q]) * 4 + fl[r];
                    dgrad[((((size_t)pos * (cout / 8) + co / 8) * 2 + (co >> 2 & 1)) * cin + ci) * 4 + (co & 3)] = v;
                }
            }
}

int g_conv_wino = -1;

}  // namespace

#ifdef TMF_WINO_TRACE
extern "C" int tmf_wino_trace_read(long long* blocks, long long* phases) {
    hipError_t e = hipMemcpyFromSymbol(blocks, HIP_SYMBOL(g_wino_blocks), sizeof(long long) * 8192 * 4);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(phases, HIP_SYMBOL(g_wino_phases), sizeof(long long) * 8 * 64);
    return (int)e;
}
#endif

// tmf_set_option("conv_wino", 0 | 1 | 2) / TMF_CONV_WINO: the Winograd form never / for the data gradients / (default) for forward
// and data gradients of the encoder's 3x3x3 blocks that qualify (tmf_conv3d_wino_ok); consulted by the whole-encoder entries
// (snet_path.hip) and, through tmf_conv_wino_mode(), by the op-by-op path (ops.py)
extern "C" int tmf_conv_wino_mode(void) {
    if (g_conv_wino < 0) {
        const char* e = getenv("TMF_CONV_WINO");
        const int v = e ? atoi(e) : 2;
        g_conv_wino = (v == 0 || v == 1) ? v : 2;
    }
    return g_conv_wino;
}
int tmf_conv_wino_set(int v) { g_conv_wino = v; return TMF_OK; }

extern "C" int tmf_conv3d_wino_ok(int cin, int cout) { return cin > 0 && cout > 0 && cin % 8 == 0 && cout % 32 == 0; }

extern "C" int tmf_conv3d_wino_stat_blocks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);
}

extern "C" size_t tmf_conv3d_wino_weight_bytes(int cin, int cout) { return (size_t)64 * cin * cout * 4; }

extern "C" int tmf_pack_conv_weights_wino(const float* w, float* u_fwd, float* u_dgrad, int cout, int cin, void* stream) {
    TMF_REQUIRE_PTR(w);
    TMF_REQUIRE(u_fwd != nullptr || u_dgrad != nullptr, TMF_E_NULL, "tmf_pack_conv_weights_wino: both outputs are NULL");
    TMF_REQUIRE(cout > 0 && cin > 0, TMF_E_SHAPE, "tmf_pack_conv_weights_wino: cout=%d cin=%d", cout, cin);
    TMF_REQUIRE(u_fwd == nullptr || tmf_conv3d_wino_ok(cin, cout), TMF_E_SHAPE,
                "tmf_pack_conv_weights_wino: forward form needs cin %% 8 == 0 and cout %% 32 == 0 (cin=%d cout=%d)", cin, cout);
    TMF_REQUIRE(u_dgrad == nullptr || tmf_conv3d_wino_ok(cout, cin), TMF_E_SHAPE,
                "tmf_pack_conv_weights_wino: data-gradient form needs cout %% 8 == 0 and cin %% 32 == 0 (cin=%d cout=%d)", cin, cout);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)tmf_cdiv((long)cout * cin, 256L)), dim3(256), 0, (hipStream_t)stream,
                       w, u_fwd, u_dgrad, cout, cin);
    return tmf_launch_result("tmf_pack_conv_weights_wino");
}

extern "C" int tmf_conv3d_fwd_wino(const float* x, const float* u, float* z, float* stat_partial,
                                   int B, int D, int H, int W, int cin, int cout, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(u); TMF_REQUIRE_PTR(z);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, TMF_E_SHAPE, "tmf_conv3d_fwd_wino: non-positive dimension (B=%d D=%d H=%d W=%d)", B, D, H, W);
    TMF_REQUIRE(tmf_conv3d_wino_ok(cin, cout), TMF_E_SHAPE,
                "tmf_conv3d_fwd_wino: needs cin %% 8 == 0 and cout %% 32 == 0 (cin=%d cout=%d)", cin, cout);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_fwd_wino: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE((long)64 * cin * cout < (1L << 29), TMF_E_SHAPE, "tmf_conv3d_fwd_wino: weight tensor exceeds 2^29 elements");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(u); TMF_REQUIRE_ALIGNED(z);
    const int tilesD = tmf_cdiv(D, TD), tilesH = tmf_cdiv(H, TH), tilesW = tmf_cdiv(W, TW);
    const int ntiles = B * tilesD * tilesH * tilesW;
    dim3 grid(ntiles, cout / 32), block(NTHR);
    int rc;
    if (stat_partial != nullptr) {
        auto k = conv3d_wino_kernel<true>;
        if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino"))) return rc;
        hipLaunchKernelGGL(k, grid, block, LDS_BYTES, (hipStream_t)stream, x, u, z, stat_partial, D, H, W, cin, cout,
                           tilesD, tilesH, tilesW, ntiles);
    } else {
        auto k = conv3d_wino_kernel<false>;
        if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino"))) return rc;
        hipLaunchKernelGGL(k, grid, block, LDS_BYTES, (hipStream_t)stream, x, u, z, stat_partial, D, H, W, cin, cout,
                           tilesD, tilesH, tilesW, ntiles);
    }
    return tmf_launch_result("tmf_conv3d_fwd_wino");
}
