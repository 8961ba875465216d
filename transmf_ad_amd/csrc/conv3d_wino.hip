// conv3d_wino.hip — the encoders' 3x3x3 convolutions (forward, data gradient, weight gradient, eval-mode block) in the
// Winograd form F(2x2x2, 3x3x3), exact-fp32 arithmetic on v_mfma_f32_32x32x2_f32, for gfx950 (MI355X).  DESIGN.md 3.15.
//
// Why.  The fp32 train step is bound by the fp32 matrix pipe (0.78 / 0.83 busy over a whole step, every vector instruction
// paid in matrix time), and the direct kernels of conv3d_mfma.hip already run at 0.72-0.86 of it: the only thing left to
// remove is the matrix work itself.  With 2x2x2 output tiles a 3x3x3 convolution needs 64 products per tile, input and
// output channel instead of 216 (x 3.375 less), at the price of 192 additions per (tile, input channel) and 112 per (tile,
// output channel).
//
//   y = A^T [ (G g G^T) .* (B^T d B) ] A   along each of the three axes,
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
//
// Forward / data gradient (conv3d_wino_kernel<MODE>).  One workgroup (8 waves, ONE per CU: its 64 x 32 x 32 accumulators
// are half the CU's register file) owns a 4x8x8 brick of output voxels = 32 tiles and 32 output channels:
//   * per chunk of 8 input channels the 6x10x10 halo comes in by LDS-DMA (zero fill from the buffer range check), double-
//     buffered, in a parity-sorted slot map whose tile-strided ds_read_b128 is conflict-free (see RAW layout below);
//   * wave w owns the 8 positions (pd = w / 2, ph in {2 (w & 1), +1}, pw = 0..3) of the transformed tile with one 32 x 32
//     accumulator each; NO transformed tensor exists: the lane (tile, channel quad) computes the A operands of its tile for
//     those positions in registers straight from the halo (24 ds_read_b128 + 64 packed adds per chunk); the K permutation is
//     conv3d_mfma.hip's (lanes 0-31 channels 0-3, lanes 32-63 channels 4-7);
//   * B operand = the transformed weights U[p][cin/8][2][cout][4] (tmf_pack_conv_weights_wino: fp64 inside, rounded once):
//     each position's weights are used by exactly one wave, so they go global -> that wave's private 8 KB of LDS by DMA, the
//     next chunk's copies issued right behind the reads of the running one;
//   * the waves 0-3 transform a chunk and then multiply it while the waves 4-7 first multiply the previous chunk and then
//     transform (w and w + 4 share a SIMD: one's loads / adds run beside the other's MFMAs); one barrier per chunk;
//   * epilogue: the w transform and this wave's half of the h transform in registers, one exchange through LDS (128 KB), the
//     other half and the d transform on the reading side, 128-byte channel rows to z, BatchNorm statistic partials as the
//     direct kernels produce them (MODE 1) or the eval-mode affine + LeakyReLU + in-lane 2x2x2 max pool (MODE 2); fixed
//     order everywhere: results are bit-reproducible run to run.
// Weight gradient (conv3d_wino_wgrad_kernel): dU_p = V_p^T Z_p per position over all tiles, dW = G^T dU G — further down.
//
// Numerics: transforms, products and sums in fp32 (U and the final G^T . G in fp64); against fp64 the error of z, dx and dw
// is level with the direct kernels' (2-8e-7 of the maximum: Cin-long fp32 chains + the cancellation of the output transform
// against 864..3456-term chains), measured in tools/wino_check.py / wino_wgrad_check.py and gated by the golden tests at the
// same tolerances as the direct path.
//
// Replaces aten::conv3d / convolution_backward at /root/reference/models/networks.py:28,31,37,40,46.
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

// MODE 0: z only (data gradient, eval-mode z); 1: z + BatchNorm statistic partials (train forward); 2: the eval-mode block in
// ONE pass — y = LeakyReLU(scale z + shift), optionally 2x2x2 max-pooled, is what gets stored (BatchNorm is affine in eval mode;
// a pooling window — d pair x h pair x w pair — lies inside ONE lane's 16 outputs, so the pool costs no exchange at all)
template <int MODE>
__global__ __launch_bounds__(NTHR) void conv3d_wino_kernel(
    const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ z, float* __restrict__ stat_partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int ntiles,
    const float* __restrict__ aff_scale = nullptr, const float* __restrict__ aff_shift = nullptr, float slope = 0.f, int pool = 0) {
    constexpr bool STATS = MODE == 1;
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
    // this wave's first weights are requested before anything else is computed (one workgroup per CU: nothing else hides the round trip)
    const int mpd = wave >> 1, mhh = wave & 1;
    const int p0 = mpd * 16 + mhh * 8;
    const int b_lane = (hsel * Cout + n0 + l31) * 16;        // bytes
    const i32x4 ur = make_rsrc(u, (unsigned)(64 * Cin * Cout * 4));
    const unsigned bl0 = lds0 + B_OFF + wave * 8192;
    auto stage_b = [&](int c, int q) {              // this wave's weights of (chunk c, position q) -> its own 8 KB
        blds16(b_lane, ur, (((p0 + q) * nchunk + c) * 2 * Cout) * 16, bl0 + q * 1024);
    };
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_b(0, q);
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
    stage(0);                                       // (the rest of the address plan below runs under this round trip)

    // ---- this wave's part of the input transform and of the products ----
    // wave = (pd, half of ph): the 8 positions (pd, ph = 2 mhh + {0, 1}, pw = 0..3), one 32 x 32 accumulator each.  The lane
    // computes the A operands of ITS tile (l31) and channel quad (hsel) in registers, straight from the halo:
    //   along d: row pd of B^T = d_a + sgn d_b;   along w: all four;   along h: the two rows this wave multiplies —
    //   mhh = 0: ph0 = x0 - x2, ph1 = x1 + x2 (keeper x2);  mhh = 1: ph2 = x2 - x1, ph3 = x1 - x3 = -(x3 - x1) (keeper x1)
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
    const float* Bl = smem + (B_OFF + wave * 8192) / 4 + lane * 4;
    f32x16 acc[8];
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
        // (float2 halves on purpose: <2 x float> arithmetic is one v_pk_*_f32 per pair, <4 x float> is split into scalars)
        const f32x2 s2 = {sgn, sgn}, c2 = {c1, c1};
        auto wrow = [&](int pa, int pb, f32x2 (&lo)[4], f32x2 (&hi)[4]) {     // one h row: d combination, then the w transform
            f32x2 tl[4], th2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&R[pa + ogw(k) * 8]), bq = *reinterpret_cast<const f32x4*>(&R[pb + ogw(k) * 8]);
                tl[k] = f32x2{a[0], a[1]} + s2 * f32x2{bq[0], bq[1]};
                th2[k] = f32x2{a[2], a[3]} + s2 * f32x2{bq[2], bq[3]};
            }
            lo[0] = tl[0] - tl[2]; lo[1] = tl[1] + tl[2]; lo[2] = tl[2] - tl[1]; lo[3] = tl[1] - tl[3];
            hi[0] = th2[0] - th2[2]; hi[1] = th2[1] + th2[2]; hi[2] = th2[2] - th2[1]; hi[3] = th2[1] - th2[3];
        };
        f32x2 kl[4], kh[4], al[4], ah[4], bl[4], bh[4];
        wrow(rka, rkb, kl, kh);
        wrow(raa, rab, al, ah);
        wrow(rba, rbb, bl, bh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            al[k] = al[k] - kl[k]; ah[k] = ah[k] - kh[k];
            bl[k] = c2 * bl[k] + kl[k]; bh[k] = c2 * bh[k] + kh[k];
            A0[k] = f32x4{al[k][0], al[k][1], ah[k][0], ah[k][1]};
            A1[k] = f32x4{bl[k][0], bl[k][1], bh[k][0], bh[k][1]};
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
    if constexpr (MODE == 2) {
        const int co = n0 + l31;
        const float sc = aff_scale[co], sh = aff_shift[co];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = outv[dd][ho][e] * sc + sh;
                    outv[dd][ho][e] = y > 0.f ? y : y * slope;
                }
        const int gwb = w0 + 4 * (wave & 1), ghb = h0 + 2 * hsel + 4 * ((wave >> 1) & 1), gdb = d0 + 2 * (wave >> 2);
        if (pool == TMF_POOL_MAX2) {
            const int OD = D / 2, OH = H / 2, OW = W / 2;
            float* yb = z + (size_t)b * OD * OH * OW * Cout;
            const int od = gdb >> 1, oh = ghb >> 1, ow = gwb >> 1;
#pragma unroll
            for (int wp = 0; wp < 2; ++wp) {               // w pair wp = outputs e = wp (w offset 2 wp) and e = wp + 2 (2 wp + 1)
                float m = fmaxf(fmaxf(outv[0][0][wp], outv[0][0][wp + 2]), fmaxf(outv[0][1][wp], outv[0][1][wp + 2]));
                m = fmaxf(m, fmaxf(fmaxf(outv[1][0][wp], outv[1][0][wp + 2]), fmaxf(outv[1][1][wp], outv[1][1][wp + 2])));
                if (od < OD && oh < OH && ow + wp < OW) yb[((size_t)(od * OH + oh) * OW + ow + wp) * Cout + co] = m;
            }
        } else {
            float* yb = z + (size_t)b * D * H * W * Cout;
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gd = gdb + dd, gh = ghb + ho, gw = gwb + 2 * (e & 1) + (e >> 1);
                        if (gd < D && gh < H && gw < W) yb[((size_t)(gd * H + gh) * W + gw) * Cout + co] = outv[dd][ho][e];
                    }
        }
        return;
    }
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

// ------------------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd form.  y = A^T [(G g) .* (B^T d)] gives  dL/d(G g)_p = (B^T d)_p (A dy)_p: per position p of
// the 4x4x4 transformed tile a plain product of the TRANSFORMED INPUT V (the forward's own input transform) and the
// transformed output gradient Z = A dz A^T (1-D: z0 = y0, z1 = y0 + y1, z2 = y0 - y1, z3 = -y1), summed over all tiles:
//     dU_p[ci][co] = sum_tiles V_p[tile][ci] Z_p[tile][co]        (64 GEMMs with K = tiles instead of 27 with K = voxels)
//     dW = G^T dU G along the three axes                          (64 -> 27, once, in the reduction kernel)
// — 64 products per 2x2x2 tile, input and output channel instead of 216, like the forward.
// A workgroup (8 waves) owns a 32 x 32 block of (ci, co) and walks a contiguous range of 4x4x8 half bricks (16 tiles):
// per stage the 6x6x10 halo of x (32 channels, 46 KB) and the dz half brick (16 KB) come in by LDS-DMA (whole 128-byte voxel
// rows), double-buffered, one barrier per stage; wave w owns the 8 positions (pd = w / 2, ph = 2 (w & 1) + {0, 1}, pw) as in
// the forward, the lane = (channel, tile parity) transforms ITS tile's taps in registers — A operand V_p[tile 2 j + hsel][ci],
// B operand Z_p[tile 2 j + hsel][co] — and every step j is 8 MFMAs (K = 2 tiles).  No barrier inside a stage: the two waves
// of a SIMD drift apart and one's loads / adds run beside the other's MFMAs.  Each workgroup leaves one partial slab
// [64][32][32]; tmf_reduce_slabs (fp64, fixed order) and wino_wgrad_finish_kernel (G^T . G in fp64) turn them into dw.
constexpr int WX_SLOTS = 3072, WX_REAL = 360 * 8, WZ_SLOTS = 1024;        // 16-byte pieces: x halo 6x6x10 voxels x 8, dz 4x4x8 x 8
constexpr int WBUF_BYTES = (WX_SLOTS + WZ_SLOTS) * 16;                      // 64 KB per stage buffer
constexpr int WZ_OFF = WX_SLOTS * 4;                                        // floats
constexpr size_t WG_LDS_BYTES = 2 * (size_t)WBUF_BYTES;

__global__ __launch_bounds__(NTHR) void conv3d_wino_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int nbricks, int per_split, int ncob) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int split = blockIdx.x, blk = blockIdx.y;
    const int ci0 = (blk / ncob) * 32, co0 = (blk % ncob) * 32;
    const int st0 = split * per_split;
    const int st1 = st0 + per_split < nbricks ? st0 + per_split : nbricks;
    constexpr int OOB = (int)0x80000000u;

    // ---- staging plan of this thread: 6 pieces of the x halo, 2 of dz; everything that does not depend on the brick ----
    int relx[6], pcx[6], relz[2], pcz[2];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int e = (q * 8 + wave) * 64 + lane, v = e >> 3, piece = e & 7;
        const int hd = v / 60, hh = (v / 10) % 6, hw = v % 10;
        relx[q] = (((hd - 1) * H + (hh - 1)) * W + (hw - 1)) * Cin * 4 + piece * 16;
        pcx[q] = e < WX_REAL ? (hd | (hh << 8) | (hw << 16)) : -1;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = (q * 8 + wave) * 64 + lane, v = e >> 3, piece = e & 7;
        const int od = v >> 5, oh = (v >> 3) & 3, ow = v & 7;
        relz[q] = ((od * H + oh) * W + ow) * Cout * 4 + piece * 16;
        pcz[q] = od | (oh << 8) | (ow << 16);
    }
    auto stage = [&](int st) {                      // half brick st -> buffer (st - st0) & 1
        int t = st;
        const int bw = t % tilesW; t /= tilesW;
        const int bh = t % tilesH; t /= tilesH;
        const int bd = t % tilesD;
        const int b = t / tilesD;
        const int d0 = bd * 4, h0 = bh * 4, w0 = bw * 8;
        const i32x4 xr = make_rsrc(x + (size_t)b * D * H * W * Cin, (unsigned)(D * H * W * Cin * 4));
        const i32x4 zr = make_rsrc(dz + (size_t)b * D * H * W * Cout, (unsigned)(D * H * W * Cout * 4));
        const int xbase = ((d0 * H + h0) * W + w0) * Cin * 4 + ci0 * 4, zbase = ((d0 * H + h0) * W + w0) * Cout * 4 + co0 * 4;
        const unsigned base = lds0 + ((st - st0) & 1) * WBUF_BYTES + wave * 1024;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int hd = pcx[q] & 255, hh = (pcx[q] >> 8) & 255, hw = (pcx[q] >> 16) & 255;
            const bool ok = pcx[q] >= 0 && (unsigned)(d0 - 1 + hd) < (unsigned)D && (unsigned)(h0 - 1 + hh) < (unsigned)H &&
                            (unsigned)(w0 - 1 + hw) < (unsigned)W;
            blds16(ok ? relx[q] + xbase : OOB, xr, 0, base + q * 8192);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int od = pcz[q] & 255, oh = (pcz[q] >> 8) & 255, ow = (pcz[q] >> 16) & 255;
            const bool ok = d0 + od < D && h0 + oh < H && w0 + ow < W;
            blds16(ok ? relz[q] + zbase : OOB, zr, 0, base + WX_SLOTS * 16 + q * 8192);
        }
    };

    // ---- this wave's positions and transform rows (as the forward kernel) ----
    const int mpd = wave >> 1, mhh = wave & 1;
    const int da = mpd == 0 ? 0 : (mpd == 2 ? 2 : 1);
    const int db = mpd == 0 ? 2 : (mpd == 1 ? 2 : (mpd == 2 ? 1 : 3));
    const float sgn = mpd == 1 ? 1.f : -1.f;
    const int ik = mhh == 0 ? 2 : 1, ia = mhh == 0 ? 0 : 2, ib = mhh == 0 ? 1 : 3;
    const float c1 = mhh == 0 ? 1.f : -1.f;
    const float ca = mpd == 3 ? 0.f : 1.f, cb = mpd == 0 ? 0.f : (mpd == 1 ? 1.f : -1.f);      // z_d = ca y[do 0] + cb y[do 1]
    const float e0 = mhh == 0 ? 0.f : -1.f, f0 = mhh == 0 ? 1.f : 0.f, f1 = mhh == 0 ? 1.f : -1.f;
    const int lx = l31 + hsel * 64;                                                          // tile 2 j + hsel: w origin 2 hsel more
    auto xrow = [&](int dd, int i) { return lx + (dd * 60 + i * 10) * 32; };
    const int rka = xrow(da, ik), rkb = xrow(db, ik), raa = xrow(da, ia), rab = xrow(db, ia), rba = xrow(da, ib), rbb = xrow(db, ib);
    const int lz = WZ_OFF + l31 + hsel * 64;

    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    if (st0 < st1) {
        stage(st0);
        dma_wait();
        __syncthreads();
    }
    for (int st = st0; st < st1; ++st) {
        if (st + 1 < st1) stage(st + 1);            // (its buffer was last read before the barrier that ended stage st - 1)
        const float* R = smem + ((st - st0) & 1) * (WBUF_BYTES / 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int td = j >> 2, th = (j >> 1) & 1, twb = (j & 1) * 2;
            const int xo = ((2 * td * 6 + 2 * th) * 10 + 2 * twb) * 32, zo = ((2 * td * 4 + 2 * th) * 8 + 2 * twb) * 32;
            // input transform of the lane's tile and channel: rows keeper / a / b, all four w positions
            auto wrow = [&](int pa, int pb, float (&wv)[4]) {
                float tv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) tv[k] = R[pa + xo + k * 32] + sgn * R[pb + xo + k * 32];
                wv[0] = tv[0] - tv[2]; wv[1] = tv[1] + tv[2]; wv[2] = tv[2] - tv[1]; wv[3] = tv[1] - tv[3];
            };
            float wk[4], A0[4], A1[4];
            wrow(rka, rkb, wk);
            wrow(raa, rab, A0);
            wrow(rba, rbb, A1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                A0[k] = A0[k] - wk[k];
                A1[k] = c1 * A1[k] + wk[k];
            }
            // transform of the lane's dz tile and channel: Z[pd][ph = 2 mhh + phl][pw]
            float u[2][2];
#pragma unroll
            for (int wo = 0; wo < 2; ++wo) {
                float zd[2];
#pragma unroll
                for (int ho = 0; ho < 2; ++ho)
                    zd[ho] = ca * R[lz + zo + ((0 * 4 + ho) * 8 + wo) * 32] + cb * R[lz + zo + ((1 * 4 + ho) * 8 + wo) * 32];
                u[0][wo] = zd[0] + e0 * zd[1];
                u[1][wo] = f0 * zd[0] + f1 * zd[1];
            }
#pragma unroll
            for (int phl = 0; phl < 2; ++phl) {
                const float b0 = u[phl][0], b1 = u[phl][0] + u[phl][1], b2 = u[phl][0] - u[phl][1], b3 = -u[phl][1];
                const float* Ap = phl == 0 ? A0 : A1;
                acc[phl * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[0], b0, acc[phl * 4 + 0], 0, 0, 0);
                acc[phl * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[1], b1, acc[phl * 4 + 1], 0, 0, 0);
                acc[phl * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[2], b2, acc[phl * 4 + 2], 0, 0, 0);
                acc[phl * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[3], b3, acc[phl * 4 + 3], 0, 0, 0);
            }
        }
        dma_wait();
        __syncthreads();                            // the next stage is in LDS; every wave is done with this one
    }

    // ---- partial slab [split][blk][p][ci 32][co 32]: accumulator row r -> ci = (r & 3) + 8 (r >> 2) + 4 hsel, column co = l31 ----
    float* out = partial + (((size_t)split * gridDim.y + blk) * 64 + (mpd * 16 + mhh * 8)) * 1024;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[q * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hsel) * 32 + l31] = acc[q][r];
}

// dU (reduced over the slabs) [blk][p][ci 32][co 32] -> dw = G^T dU G along the three axes, one thread per (ci, co), fp64:
// 1-D: w0 = u0 + (u1 + u2) / 2, w1 = (u1 - u2) / 2, w2 = (u1 + u2) / 2 + u3.  dw_ref: nn.Conv3d's [co][ci][27], else [27][ci][co].
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ du, float* __restrict__ dw,
                                                                int cin, int cout, int ncob, int dw_ref) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cin * cout) return;
    const int co = e % cout, ci = e / cout;
    const int blk = (ci / 32) * ncob + co / 32;
    const float* src = du + (size_t)blk * 64 * 1024 + (ci % 32) * 32 + (co % 32);
    double u[4][4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[p][q][r] = (double)src[((p * 4 + q) * 4 + r) * 1024];
    auto GT = [](double u0, double u1, double u2, double u3, int a) {
        return a == 0 ? u0 + 0.5 * (u1 + u2) : (a == 1 ? 0.5 * (u1 - u2) : 0.5 * (u1 + u2) + u3);
    };
    double t1[3][4][4], t2[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) t1[a][q][r] = GT(u[0][q][r], u[1][q][r], u[2][q][r], u[3][q][r], a);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bq = 0; bq < 3; ++bq)
#pragma unroll
            for (int r = 0; r < 4; ++r) t2[a][bq][r] = GT(t1[a][0][r], t1[a][1][r], t1[a][2][r], t1[a][3][r], bq);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bq = 0; bq < 3; ++bq)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (float)GT(t2[a][bq][0], t2[a][bq][1], t2[a][bq][2], t2[a][bq][3], c);
                const int tap = (a * 3 + bq) * 3 + c;
                if (dw_ref) dw[((size_t)co * cin + ci) * 27 + tap] = v;
                else dw[((size_t)tap * cin + ci) * cout + co] = v;
            }
}

struct WinoWgPlan { int tilesD, tilesH, tilesW, nbricks, nblk, ncob, nsplit, per; };
WinoWgPlan plan_wino_wgrad(int B, int D, int H, int W, int cin, int cout) {
    WinoWgPlan p;
    p.tilesD = tmf_cdiv(D, 4); p.tilesH = tmf_cdiv(H, 4); p.tilesW = tmf_cdiv(W, 8);
    p.nbricks = B * p.tilesD * p.tilesH * p.tilesW;
    p.ncob = cout / 32;
    p.nblk = (cin / 32) * p.ncob;
    int ns = p.nblk >= 256 ? 1 : 256 / p.nblk;             // one round of workgroups over the 256 CUs
    if (ns > p.nbricks) ns = p.nbricks;
    p.per = tmf_cdiv(p.nbricks, ns);
    p.nsplit = tmf_cdiv(p.nbricks, p.per);
    return p;
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

// tmf_set_option("conv_wino", 0 | 1 | 2 | 3) / TMF_CONV_WINO: the Winograd form never / for the data gradients / for forward and
// data gradients / (default) for forward, data and weight gradients of the encoder's 3x3x3 blocks that qualify (tmf_conv3d_wino_ok); consulted by the whole-encoder entries
// (snet_path.hip) and, through tmf_conv_wino_mode(), by the op-by-op path (ops.py)
extern "C" int tmf_conv_wino_mode(void) {
    if (g_conv_wino < 0) {
        const char* e = getenv("TMF_CONV_WINO");
        const int v = e ? atoi(e) : 3;
        g_conv_wino = (v >= 0 && v <= 2) ? v : 3;
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
        auto k = conv3d_wino_kernel<1>;
        if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino"))) return rc;
        hipLaunchKernelGGL(k, grid, block, LDS_BYTES, (hipStream_t)stream, x, u, z, stat_partial, D, H, W, cin, cout,
                           tilesD, tilesH, tilesW, ntiles, (const float*)nullptr, (const float*)nullptr, 0.f, 0);
    } else {
        auto k = conv3d_wino_kernel<0>;
        if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino"))) return rc;
        hipLaunchKernelGGL(k, grid, block, LDS_BYTES, (hipStream_t)stream, x, u, z, stat_partial, D, H, W, cin, cout,
                           tilesD, tilesH, tilesW, ntiles, (const float*)nullptr, (const float*)nullptr, 0.f, 0);
    }
    return tmf_launch_result("tmf_conv3d_fwd_wino");
}

extern "C" int tmf_conv3d_wgrad_wino_ok(int cin, int cout) { return cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0; }

extern "C" size_t tmf_conv3d_wgrad_wino_workspace_bytes(int B, int D, int H, int W, int cin, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || !tmf_conv3d_wgrad_wino_ok(cin, cout)) return 0;
    const WinoWgPlan p = plan_wino_wgrad(B, D, H, W, cin, cout);
    const size_t n = (size_t)p.nblk * 64 * 1024;
    return (size_t)(p.nsplit + tmf_reduce_groups(p.nsplit) + 1) * n * 4;
}

extern "C" int tmf_conv3d_wgrad_wino(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                                     int B, int D, int H, int W, int cin, int cout, int dw_layout, void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG,
                "tmf_conv3d_wgrad_wino: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(dz); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, TMF_E_SHAPE, "tmf_conv3d_wgrad_wino: non-positive dimension");
    TMF_REQUIRE(tmf_conv3d_wgrad_wino_ok(cin, cout), TMF_E_SHAPE,
                "tmf_conv3d_wgrad_wino: needs cin %% 32 == 0 and cout %% 32 == 0 (cin=%d cout=%d)", cin, cout);
    TMF_REQUIRE((long)(D + 2) * (H + 2) * (W + 2) * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_wgrad_wino: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(dz); TMF_REQUIRE_ALIGNED(dw); TMF_REQUIRE_ALIGNED(workspace);
    const size_t need = tmf_conv3d_wgrad_wino_workspace_bytes(B, D, H, W, cin, cout);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_conv3d_wgrad_wino: workspace %zu B < required %zu B", workspace_bytes, need);
    const WinoWgPlan p = plan_wino_wgrad(B, D, H, W, cin, cout);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const long n = (long)p.nblk * 64 * 1024;
    int rc;
    auto k = conv3d_wino_wgrad_kernel;
    if ((rc = tmf_allow_lds(k, WG_LDS_BYTES, "tmf_conv3d_wgrad_wino"))) return rc;
    hipLaunchKernelGGL(k, dim3(p.nsplit, p.nblk), dim3(NTHR), WG_LDS_BYTES, s, x, dz, partial, D, H, W, cin, cout,
                       p.tilesD, p.tilesH, p.tilesW, p.nbricks, p.per, p.ncob);
    if ((rc = tmf_launch_result("tmf_conv3d_wgrad_wino"))) return rc;
    float* scratch = partial + (size_t)p.nsplit * n;
    float* du = scratch + (size_t)tmf_reduce_groups(p.nsplit) * n;
    if ((rc = tmf_reduce_slabs(partial, p.nsplit, n, scratch, du, s, "tmf_conv3d_wgrad_wino(reduce)"))) return rc;
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)tmf_cdiv((long)cin * cout, 256L)), dim3(256), 0, s,
                       (const float*)du, dw, cin, cout, p.ncob, dw_layout == TMF_DW_REFERENCE ? 1 : 0);
    return tmf_launch_result("tmf_conv3d_wgrad_wino(finish)");
}

// Eval-mode block in one pass (tmf_conv3d_fwd_affine's Winograd form): y = LeakyReLU(scale * conv(x, w) + shift), optionally
// 2x2x2 max-pooled (floor mode).  pool: TMF_POOL_NONE | TMF_POOL_MAX2 (the average pool of the network follows its 1x1x1 block).
extern "C" int tmf_conv3d_fwd_wino_affine(const float* x, const float* u, const float* scale, const float* shift, float* y,
                                          int B, int D, int H, int W, int cin, int cout, int pool, float slope, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(u); TMF_REQUIRE_PTR(scale); TMF_REQUIRE_PTR(shift); TMF_REQUIRE_PTR(y);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, TMF_E_SHAPE, "tmf_conv3d_fwd_wino_affine: non-positive dimension");
    TMF_REQUIRE(tmf_conv3d_wino_ok(cin, cout), TMF_E_SHAPE,
                "tmf_conv3d_fwd_wino_affine: needs cin %% 8 == 0 and cout %% 32 == 0 (cin=%d cout=%d)", cin, cout);
    TMF_REQUIRE(pool == TMF_POOL_NONE || pool == TMF_POOL_MAX2, TMF_E_ARG, "tmf_conv3d_fwd_wino_affine: pool must be none or max, got %d", pool);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "tmf_conv3d_fwd_wino_affine: one sample exceeds 2^29 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE((long)64 * cin * cout < (1L << 29), TMF_E_SHAPE, "tmf_conv3d_fwd_wino_affine: weight tensor exceeds 2^29 elements");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(u); TMF_REQUIRE_ALIGNED(y);
    if (pool != TMF_POOL_NONE && (D / 2 == 0 || H / 2 == 0 || W / 2 == 0)) return TMF_OK;      // empty output
    const int tilesD = tmf_cdiv(D, TD), tilesH = tmf_cdiv(H, TH), tilesW = tmf_cdiv(W, TW);
    const int ntiles = B * tilesD * tilesH * tilesW;
    auto k = conv3d_wino_kernel<2>;
    int rc;
    if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino_affine"))) return rc;
    hipLaunchKernelGGL(k, dim3(ntiles, cout / 32), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, x, u, y, (float*)nullptr, D, H, W, cin, cout,
                       tilesD, tilesH, tilesW, ntiles, scale, shift, slope, pool);
    return tmf_launch_result("tmf_conv3d_fwd_wino_affine");
}
