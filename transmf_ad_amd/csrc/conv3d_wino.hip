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
// A buffer_store_dwordx4 reads its data registers AFTER it has issued: a v_pk_* that overwrites them in the very next slot
// corrupted the second register of the pair in lanes 12-15 of every row of 16 (measured on gfx950 with an SGPR soffset, the
// case LLVM's hazard recognizer exempts; tools/asm_checks.py finds the pattern in a listing).  One wait state after a store
// whose data dies right behind it:
__device__ __forceinline__ void store_guard() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef TMF_WINO_TRACE
// instrumented build (tools/wino_trace.py): shader-clock stamps of every block (start, end, hardware ids) and of the phases of one
__device__ long long g_wino_blocks[8192 * 4];
__device__ long long g_wino_phases[8 * 64];
#define TRB(i, v) do { if (blockIdx.y == 0 && tid == 0 && blockIdx.x < 8192) g_wino_blocks[blockIdx.x * 4 + (i)] = (v); } while (0)
#define TRP(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 1500 && blockIdx.y == 0 && lane == 0) g_wino_phases[wave * 64 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
// persistent kernel: phases of the SECOND item of one workgroup (steady state of the stream)
#define TRQ(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 77 && it == 1 && lane == 0) g_wino_phases[wave * 64 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TRB(i, v)
#define TRP(i)
#define TRQ(i)
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
// The same convolution as ONE WAVE PER SIMD and a persistent workgroup (conv3d_wino_p_kernel<MODE>; round 5).
// Why: on gfx950 the fp32 matrix instructions execute on the vector ALU itself — a partner wave's VALU / LDS instructions
// do not run beside them, they wait (tools/microbench/valu_cost.hip: a wave that only transforms beside a wave that only
// multiplies costs BOTH, +35 % on the multiplier for 2 VALU per MFMA, where the same instructions interleaved in the
// multiplying wave itself cost +9 %), so the half-period stagger of the kernel above hides nothing, and with one workgroup
// per CU its prologue / epilogue (8 000 of 33 000 cycles per brick with all arithmetic removed) are exposed.  Here:
//   * 4 waves, wave = pd (the d position of the 4x4x4 transformed tile): 16 positions x 32 tiles x 32 output channels =
//     256 accumulator registers per lane (the AGPR half of the 512-register budget of a lone wave);
//   * the workgroup walks a strided list of (brick, channel group) items as ONE continuous stream of 8-channel chunks:
//     the halo of chunk g + 2 is in flight (LDS-DMA, two buffers), the A operands of chunk g + 1 are transformed and its
//     weights loaded while chunk g is multiplied — across item boundaries too, so only the first item of a workgroup
//     pays a prologue;
//   * a chunk is two phases of 32 MFMAs: P0 multiplies the positions ph 1, 2 (they need only the halo rows 1, 2:
//     u1 + u2, u2 - u1) while the rows 0, 3 are read and turned into ph 0, 3 (u0 - u2, u1 - u3); P1 multiplies ph 0, 3
//     while the rows 1, 2 of the NEXT chunk are transformed — no second copy of the A operands, 32 ds_read_b128 and 96
//     packed adds per 64 MFMAs (the two-waves-per-SIMD kernel: 64 and 128);
//   * the transformed weights go global -> registers (one buffer_load_dwordx4 per position and chunk: the 16 bytes of a
//     lane are its B operand of the four K steps), a phase ahead: no LDS traffic for them at all;
//   * epilogue: w and h output transforms in registers (all 16 positions of a pd are in the wave), one 64 KB exchange
//     for the d transform, stores and BatchNorm partials as above.
#ifndef P_BSTAGE
#define P_BSTAGE 2
#endif
#ifndef P_ABL               // timing ablations (tools/build_variant.py --flags=-DP_ABL=n; results are wrong with any bit set):
#define P_ABL 0             // 1 no input transform, 2 no weight loads, 4 no halo copies, 8 no epilogue, 16 no MFMAs, 128 no statistic sums, 256 no store_guard
#endif
constexpr int PN = 256;                                           // threads: 4 waves, wave = pd
// The 32 tiles of an item, two geometries:
//   GEOM 0: one sample, 2 x 4 x 4 tiles = the 4x8x8 brick and slot map of the kernel above (tile l31 = td * 16 + th * 4 + tw);
//   GEOM 1: FOUR samples x 2 x 2 x 2 tiles = a 4x4x4 brick of each (tile l31 = s * 8 + td * 4 + th * 2 + tw) for the small deep
//     volumes: 12^3 in 4x8x8 bricks pads to 12x16x16 (1.78 x the products), 11x13x11 likewise; 4x4x4 bricks fit exactly / pad
//     to 12x16x12.  Slot map of its 4 x 6x6x6 halo: group = parity * 144 + s * 36 + (hd >> 1) * 10 + (hh >> 1) * 3 + (hw >> 1),
//     slot = 2 group + (quad ^ (hh >> 1 & 1)) as in GEOM 0 — the strides 36 and 10 make the four lane quads of either service
//     group of a ds_read_b128 start at the bank residues {0, 2, 4, 6} mod 8 (tests/test_host_cpu.py checks both maps
//     exhaustively: bijective, every tap conflict-free).
template <int GEOM> struct PGeom;
template <> struct PGeom<0> { static constexpr int BS = 1, BD = 4, BH = 8, BW = 8, CP = 96, SS = 0, SD = 32, SH = 6, TAB = 768; };
template <> struct PGeom<1> { static constexpr int BS = 4, BD = 4, BH = 4, BW = 4, CP = 144, SS = 36, SD = 10, SH = 3, TAB = 448; };
template <int GEOM> struct PLds {
    using GM = PGeom<GEOM>;
    static constexpr int SLOTS_G = 16 * GM::CP;                   // 8 parity classes x 2 channel quads
    static constexpr int RAWB = SLOTS_G * 16;                     // one halo buffer: 24 KB / 36 KB
    static constexpr int NDMA_G = SLOTS_G / PN;                   // 6 / 9 LDS-DMA instructions per wave and chunk
    static constexpr int EX_OFF = 2 * RAWB;                       // bytes; exchange [pd 4][tile 32][ho 2][wo 2][channel 32] floats
    static constexpr int EX_BYTES = 4 * 32 * 4 * 32 * 4;
    static constexpr int RED_OFF = EX_OFF + EX_BYTES;             // statistic scratch [which 2][source 32][33] floats
    static constexpr int TAB_OFF = RED_OFF + 2 * 32 * 33 * 4 + 64;      // item table [TAB] x 2 int4: {brick, channel group, first sample,
                                                                  // bd | bh << 10 | bw << 20}, {halo corner byte offset, valid-coordinate mask, -, -}
    static constexpr size_t BYTES = (size_t)TAB_OFF + GM::TAB * 32;     // 145 KB / 155 KB
    static_assert(SLOTS_G % PN == 0 && BYTES <= 160 * 1024, "LDS carving");
};

__device__ __forceinline__ void bload16(f32x4& dst, int voff, i32x4 rsrc, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// the loads above are invisible to the compiler's wait-count pass: the kernel waits itself (vmcnt(0)) and then re-defines
// the registers, so that every use is ordered behind the wait
__device__ __forceinline__ void bpin(f32x4 (&b)[8]) {
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}

// Packed fp32 adds as opaque instructions: beside MFMAs the compiler splits <2 x float> arithmetic into scalar instructions
// (a post-RA peephole written for the bf16 matrix pipe, where VALU instructions run beside the matrix unit).  The fp32 matrix
// instructions run ON the vector ALU and a lone wave issues one instruction per ~5 cycles: a packed add costs the matrix stream
// what a scalar one costs (tools/microbench/valu_cost.hip: 5.4 against 5.3 cycles), i.e. half per channel.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {          // a * b + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// d combination + 1-D input transform of one row of four taps as ONE opaque block: t_k = a_k + s b_k (in place), then
// u = [t0 - t2, t1 + t2, t2 - t1, t1 - t3].  (Between separate inline-asm instructions that depend on each other the compiler pads
// with s_nop — an issue slot each, ~5 matrix cycles for a lone wave; inside a block there is nothing to pad: VALU results are
// interlocked.)
__device__ __forceinline__ void pk_row4(f32x2 (&u)[4], f32x2 (&a)[4], const f32x2 (&b)[4], f32x2 s) {
    asm("v_pk_fma_f32 %4, %12, %8, %4\n\t"
        "v_pk_fma_f32 %5, %12, %9, %5\n\t"
        "v_pk_fma_f32 %6, %12, %10, %6\n\t"
        "v_pk_fma_f32 %7, %12, %11, %7\n\t"
        "v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %5, %6\n\t"
        "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(s));
}
// o = [y0, y0 + y1, y0 - y1, y1] (the 1-D transform of Z without its last sign) for two rows at once
__device__ __forceinline__ void pk_z2(f32x2& a1, f32x2& a2, f32x2& b1, f32x2& b2, f32x2 ya0, f32x2 ya1, f32x2 yb0, f32x2 yb1) {
    asm("v_pk_add_f32 %0, %4, %5\n\t"
        "v_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %6, %7\n\t"
        "v_pk_add_f32 %3, %6, %7 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(a1), "=&v"(a2), "=&v"(b1), "=&v"(b2) : "v"(ya0), "v"(ya1), "v"(yb0), "v"(yb1));
}
// d = a + b and e = a - b for four pairs (the h transform of two positions rows; operands may not alias the results)
__device__ __forceinline__ void pk_addsub4(f32x2 (&d)[4], f32x2 (&e)[4], const f32x2 (&a)[4], const f32x2 (&b)[4]) {
    asm("v_pk_add_f32 %0, %8, %12\n\t"
        "v_pk_add_f32 %1, %9, %13\n\t"
        "v_pk_add_f32 %2, %10, %14\n\t"
        "v_pk_add_f32 %3, %11, %15\n\t"
        "v_pk_add_f32 %4, %8, %12 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %9, %13 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %6, %10, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %7, %11, %15 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
// d = a - b for four pairs
__device__ __forceinline__ void pk_sub4(f32x2 (&d)[4], const f32x2 (&a)[4], const f32x2 (&b)[4]) {
    asm("v_pk_add_f32 %0, %4, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %5, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %6, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %7, %11 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// one accumulator element AGPR -> VGPR, in program order (left to the compiler, the copies of all 256 elements are hoisted to
// the top of the epilogue and the next item's operands are spilled to make room)
__device__ __forceinline__ float acc_read(float a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

template <int MODE, int GEOM>
__global__ __launch_bounds__(PN) void conv3d_wino_p_kernel(
    const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ z, float* __restrict__ stat_partial,
    int B, int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int nbricks, int item0, int nitems,
    const float* __restrict__ aff_scale = nullptr, const float* __restrict__ aff_shift = nullptr, float slope = 0.f, int pool = 0,
    int stat_rows = 0, int stat_accum = 0) {
    constexpr bool STATS = MODE == 1;
    using GM = PGeom<GEOM>;
    using PL = PLds<GEOM>;
    constexpr int BS = GM::BS, BD = GM::BD, BH = GM::BH, BW = GM::BW;                 // samples and voxels of a brick
    constexpr int CP = GM::CP, SS = GM::SS, SD = GM::SD, SH = GM::SH;                 // slot map strides (groups)
    constexpr int RAWB = PL::RAWB, PDMA = PL::NDMA_G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ex = smem + PL::EX_OFF / 4;
    float* red = smem + PL::RED_OFF / 4;
    i32x4* tab = reinterpret_cast<i32x4*>(smem + PL::TAB_OFF / 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int nchunk = Cin / CK;
    constexpr int OOB = (int)0x80000000u;

    // ---- this workgroup's items: item0 + jb + i G; within a window of G consecutive items every XCD (blockIdx % 8) takes a
    // contiguous range (neighbouring bricks share their halo in that XCD's L2) ----
    const int G = gridDim.x;
    const int jb = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int my_items = jb < nitems ? (nitems - jb + G - 1) / G : 0;
    if (my_items == 0) return;
    for (int i = tid; i < my_items; i += PN) {
        const int item = item0 + jb + i * G;
        const int ng = item / nbricks;
        int t = item - ng * nbricks;
        const int brick = t;
        const int bw = t % tilesW; t /= tilesW;
        const int bh = t % tilesH; t /= tilesH;
        const int bd = t % tilesD;
        const int b0 = (t / tilesD) * BS;
        tab[2 * i] = i32x4{brick, ng, b0, bd | (bh << 10) | (bw << 20)};
        // the halo of this brick: byte offset of its voxel (0, 0, 0) = (d0 - 1, h0 - 1, w0 - 1) in the sample (negative at the faces)
        // and the valid halo coordinates [lo, hi) per axis (the volume's faces and a ragged last brick cut them; a last brick of
        // fewer than BS samples cuts the sample range) as one mask: d bits 0-5, h bits 6-15, w bits 16-25, sample bits 26-29
        const int d0 = bd * BD, h0 = bh * BH, w0 = bw * BW;
        auto range = [](int lo, int hi) { return ((1 << hi) - 1) & ~((1 << lo) - 1); };
        auto mn = [](int a, int b_) { return a < b_ ? a : b_; };
        tab[2 * i + 1] = i32x4{(((d0 - 1) * H + (h0 - 1)) * W + (w0 - 1)) * Cin * 4,
                               range(d0 == 0 ? 1 : 0, mn(BD + 2, D - d0 + 1)) | (range(h0 == 0 ? 1 : 0, mn(BH + 2, H - h0 + 1)) << 6) |
                                   (range(w0 == 0 ? 1 : 0, mn(BW + 2, W - w0 + 1)) << 16) | (range(0, mn(BS, B - b0)) << 26), 0, 0};
    }
    __syncthreads();
    const int total = my_items * nchunk;                    // chunks of the whole stream

    // ---- halo staging (the slot map of the kernel above): DMA instruction q of wave w fills the slots (q * 4 + w) * 64 + lane ----
    // Per lane and instruction only what does not depend on the brick: the voxel's offset from the halo's corner and one-hot masks
    // of its halo h / w coordinates; the brick's part sits in the item table; per item (dma_plan: every instruction of a lone wave
    // is ~5 matrix cycles) the lanes outside the volume get the out-of-range offset by one AND + compare per copy.
    int hrel[PDMA], hm[PDMA], hoff[PDMA];
#pragma unroll
    for (int q = 0; q < PDMA; ++q) {
        const int e = (q * 4 + wave) * 64 + lane, gg = e >> 1, par = gg / CP, g = gg % CP;
        int s_, a_, bb, c_;
        bool real;
        if (GEOM == 0) { s_ = 0; a_ = g / 32; bb = (g % 32) / 6; c_ = (g % 32) % 6; real = bb < 5 && c_ < 5; }
        else { s_ = g / 36; a_ = (g % 36) / 10; bb = ((g % 36) % 10) / 3; c_ = ((g % 36) % 10) % 3; real = a_ < 3 && (g % 36) % 10 < 9; }
        const int hd = 2 * a_ + (par >> 2), hh = 2 * bb + ((par >> 1) & 1), hw = 2 * c_ + (par & 1);
        const int quad = (e & 1) ^ (bb & 1);
        hrel[q] = ((((s_ * D + hd) * H + hh) * W + hw) * Cin + quad * 4) * 4;
        hm[q] = real ? (1 << hd) | (1 << (6 + hh)) | (1 << (16 + hw)) | (1 << (26 + s_)) : (1 << 30);      // (bit 30: in no brick's mask)
    }
    i32x4 xr;
    int dm_i = 0, dm_c = 0;                                 // the (item, chunk) the next halo copy belongs to
    auto dma_plan = [&](int i) {
        const i32x4 e1 = tab[2 * i + 1];
        const int b = __builtin_amdgcn_readfirstlane(tab[2 * i][2]);
        const int corner = __builtin_amdgcn_readfirstlane(e1[0]), vm = __builtin_amdgcn_readfirstlane(e1[1]);
        const int ns = B - b < BS ? B - b : BS;
        xr = make_rsrc(x + (size_t)b * D * H * W * Cin, (unsigned)(ns * D * H * W * Cin * 4));
#pragma unroll
        for (int q = 0; q < PDMA; ++q) {
            const int off = hrel[q] + corner;               // (absolute and non-negative where valid: the range check sees vector + scalar offset)
            hoff[q] = (hm[q] & vm) == hm[q] ? off : OOB;
#if P_ABL & 32          // timing only: 8 / 16 cache lines per copy instruction instead of 32 (contiguous 1 KB / 16 x 64 B pieces)
            hoff[q] = (q * 4 + wave) * 1024 + lane * 16;
#elif P_ABL & 64
            hoff[q] = ((q * 4 + wave) * 16 + (lane >> 2)) * Cin * 4 + (lane & 3) * 16;
#endif
        }
    };
    // halo of (dm_i, dm_c) -> buffer buf in three parts (a third of the copies each); the third part advances the cursor
    auto dma_part = [&](int buf, int part) {
        const unsigned base = lds0 + buf * RAWB + wave * 1024;
#pragma unroll
        for (int q = 0; q < ((P_ABL & 4) ? 0 : PDMA); ++q)
            if (q / (PDMA / 3) == part) blds16(hoff[q], xr, dm_c * (CK * 4), base + q * 4096);
        if (part == 2 && ++dm_c == nchunk) {
            dm_c = 0;
            if (++dm_i < my_items) dma_plan(dm_i);
        }
    };
    auto dma_issue = [&](int buf) {                         // all of it (prologue)
#pragma unroll
        for (int i = 0; i < 3; ++i) dma_part(buf, i);
    };

    // ---- transformed weights: position p = (pd * 4 + ph) * 4 + pw, chunk c, channel group n0 -> this lane's 16 bytes ----
    const i32x4 ur = make_rsrc(u, (unsigned)(64 * Cin * Cout * 4));
    const int b_lane = (hsel * Cout + l31) * 16;
    f32x4 B12[8], B03[8];                                   // [ph slot][pw]: ph 1, 2 / ph 0, 3
    auto load_b = [&](f32x4 (&b)[8], int pha, int phb, int c, int n0) {
        if (P_ABL & 2) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bload16(b[k], b_lane, ur, ((((wave * 4 + pha) * 4 + k) * nchunk + c) * 2 * Cout + n0) * 16);
            bload16(b[4 + k], b_lane, ur, ((((wave * 4 + phb) * 4 + k) * nchunk + c) * 2 * Cout + n0) * 16);
        }
    };

    // ---- input transform of this wave's pd: planes da, db (d row of B^T), all four h rows and w columns ----
    const int da = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int db = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.f : -1.f;
    const int ts = GEOM == 0 ? 0 : l31 >> 3, td = GEOM == 0 ? l31 >> 4 : (l31 >> 2) & 1;
    const int th = GEOM == 0 ? (l31 >> 2) & 3 : (l31 >> 1) & 1, tw = GEOM == 0 ? l31 & 3 : l31 & 1;
    const int g0 = ts * SS + td * SD + th * SH + tw;
    const int e0 = hsel ^ (th & 1);
    // group offsets of a tap (dd, i, k) relative to the tile's own group
    auto pgd = [](int dd) { return (dd & 1) * 4 * CP + (dd >> 1) * SD; };
    auto pgh = [](int i) { return (i & 1) * 2 * CP + (i >> 1) * SH; };
    auto pgw = [](int k) { return (k & 1) * CP + (k >> 1); };
    auto row_base = [&](int dd, int i) { return (2 * (g0 + pgd(dd) + pgh(i)) + (e0 ^ (i >> 1))) * 4; };
    int rba[4], rbb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rba[i] = row_base(da, i); rbb[i] = row_base(db, i); }
    f32x4 A[16];                                            // [ph][pw]
    f32x2 u1l[4], u1h[4], u2l[4], u2h[4];                   // the w-transformed rows 1, 2 (channels 0-1 / 2-3), kept for ph 0, 3
    const f32x2 s2 = {sgn, sgn};
    auto pack4 = [](f32x2 l, f32x2 h) { return f32x4{l[0], l[1], h[0], h[1]}; };
    // The transform of a phase is cut into four stages, one per group of 8 MFMAs (the compiler's scheduler is fenced between the
    // groups: left alone it gathers the 32 MFMAs of a phase behind the barrier and waits for the LDS in front of it):
    //   stage 0 reads one h row (8 ds_read_b128), stage 1 combines / transforms it and reads the second row, stages 2, 3 finish.
    f32x4 ra[4], rb[4];                                     // the row in flight: planes da / db, the four w taps
    auto row_read = [&](const float* R, int i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ra[k] = *reinterpret_cast<const f32x4*>(&R[rba[i] + pgw(k) * 8]);
            rb[k] = *reinterpret_cast<const f32x4*>(&R[rbb[i] + pgw(k) * 8]);
        }
    };
    auto row_xform = [&](f32x2 (&lo)[4], f32x2 (&hi)[4]) {  // d combination, then the w transform (channels 0-1 / 2-3 of the quad)
        f32x2 al[4], ah[4], bl[4], bh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            al[k] = f32x2{ra[k][0], ra[k][1]}; ah[k] = f32x2{ra[k][2], ra[k][3]};
            bl[k] = f32x2{rb[k][0], rb[k][1]}; bh[k] = f32x2{rb[k][2], rb[k][3]};
        }
        pk_row4(lo, al, bl, s2);
        pk_row4(hi, ah, bh, s2);
    };
    f32x2 vl[4], vh[4];
    // (the stage results are pinned where they are computed: they are used a barrier later, and the optimiser sinks pure
    // arithmetic into the block of its first use otherwise — out of the MFMA shadow it was placed in)
    auto pin2 = [](f32x2 (&a)[4], f32x2 (&b)[4]) {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    };
    auto pin4 = [](f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); };
    auto T1 = [&](int buf, int stage) {                     // rows 1, 2 -> ph 1 = u1 + u2, ph 2 = u2 - u1
        if (P_ABL & 1) return;
        const float* R = smem + buf * (RAWB / 4);
        if (stage == 0) row_read(R, 1);
        if (stage == 1) { row_xform(u1l, u1h); row_read(R, 2); pin2(u1l, u1h); }
        if (stage == 2) { row_xform(u2l, u2h); pin2(u2l, u2h); }
        if (stage == 3) {
            f32x2 sl[4], dl[4], sh[4], dh[4];
            pk_addsub4(sl, dl, u2l, u1l);                   // ph 1 = u1 + u2, ph 2 = u2 - u1
            pk_addsub4(sh, dh, u2h, u1h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                A[4 + k] = pack4(sl[k], sh[k]);
                A[8 + k] = pack4(dl[k], dh[k]);
            }
            pin4(A[4], A[5], A[6], A[7]);
            pin4(A[8], A[9], A[10], A[11]);
        }
    };
    auto T2 = [&](int buf, int stage) {                     // rows 0, 3 -> ph 0 = u0 - u2, ph 3 = u1 - u3
        if (P_ABL & 1) return;
        const float* R = smem + buf * (RAWB / 4);
        if (stage == 0) row_read(R, 0);
        if (stage == 1) {
            row_xform(vl, vh);
            row_read(R, 3);
            {
                f32x2 dl[4], dh[4];
                pk_sub4(dl, vl, u2l);
                pk_sub4(dh, vh, u2h);
#pragma unroll
                for (int k = 0; k < 4; ++k) A[k] = pack4(dl[k], dh[k]);
            }
            pin4(A[0], A[1], A[2], A[3]);
        }
        if (stage == 2) { row_xform(vl, vh); pin2(vl, vh); }
        if (stage == 3) {
            {
                f32x2 dl[4], dh[4];
                pk_sub4(dl, u1l, vl);
                pk_sub4(dh, u1h, vh);
#pragma unroll
                for (int k = 0; k < 4; ++k) A[12 + k] = pack4(dl[k], dh[k]);
            }
            pin4(A[12], A[13], A[14], A[15]);
        }
    };

    f32x16 acc[16];
    // 8 MFMAs: K step s (channels s and 4 + s of the chunk) of the phase's 8 positions — every accumulator once per group, so a
    // dependent pair is 8 instructions apart
    auto mfma_k = [&](int pha, int phb, const f32x4 (&b)[8], int s, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#if P_ABL & 16
            if (FIRST && s == 0) { acc[pha * 4 + k] = zero; acc[phb * 4 + k] = zero; }
            acc[pha * 4 + k][s] += A[pha * 4 + k][s] * b[k][s]; acc[phb * 4 + k][s] += A[phb * 4 + k][s] * b[4 + k][s];
#else
            acc[pha * 4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[pha * 4 + k][s], b[k][s], (FIRST && s == 0) ? zero : acc[pha * 4 + k], 0, 0, 0);
            acc[phb * 4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[phb * 4 + k][s], b[4 + k][s], (FIRST && s == 0) ? zero : acc[phb * 4 + k], 0, 0, 0);
#endif
        }
    };

    // store side of the epilogue: the reader lane (channel quad lane & 7, wo = lane >> 3 & 1, vq = lane >> 4) of wave w' holds the tiles
    // 8 w' + 4 j + vq, j = 0, 1.  GEOM 0: tile d = w' >> 1, tile h = 2 (w' & 1) + j, tile w = vq;  GEOM 1: sample w', tile d = j,
    // tile h = vq >> 1, tile w = vq & 1.  st_lane: this lane's (and wave's) byte offset inside a brick; st_j: the step of j
    const int rvq = lane >> 4;
    const int st_ts = GEOM == 0 ? 0 : wave, st_td = GEOM == 0 ? wave >> 1 : 0;                  // (+ j for GEOM 1)
    const int st_th = GEOM == 0 ? 2 * (wave & 1) : rvq >> 1, st_tw = GEOM == 0 ? rvq : rvq & 1; // (+ j for GEOM 0)
    const int st_lane = (((((st_ts * D + 2 * st_td) * H + 2 * st_th) * W + 2 * st_tw + ((lane >> 3) & 1)) * Cout + 4 * (lane & 7))) * 4;
    const int st_j = (GEOM == 0 ? 2 * W : 2 * H * W) * Cout * 4;
    // ---- prologue of the stream: two halos in flight, the first weights, the first half transform ----
    dma_plan(0);
    dma_issue(0);
    if (total > 1) dma_issue(1);
    int ci_n0;                                              // channel group of the item being multiplied
    {
        const i32x4 e = tab[0];
        ci_n0 = __builtin_amdgcn_readfirstlane(e[1]) * 32;
    }
    load_b(B12, 1, 2, 0, ci_n0);
    dma_wait();
    __syncthreads();
    bpin(B12);
#pragma unroll
    for (int k = 0; k < 4; ++k) T1(0, k);

    // BatchNorm statistic partials (MODE 1): ONE row [2][Cout] per workgroup.  A lane keeps its 4 + 4 sums over the items of a
    // channel group in registers; the LDS reduction over the workgroup and the store happen only where the group changes (the
    // items of a workgroup are ordered by group) and at the end — per item this is 32 packed adds, no barrier.
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    int st_n0 = -1;                                         // channel group the sums belong to (-1: none yet)
    unsigned st_seen = 0;                                   // groups this workgroup has written
    auto stat_flush = [&]() {                               // sums of group st_n0 -> this workgroup's row
        const int cq_ = lane & 7, src = wave * 8 + (lane >> 3);
        __syncthreads();                                    // (red may still be read by the previous flush)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[src * 33 + 4 * cq_ + e] = st1[e];
            red[(32 + src) * 33 + 4 * cq_ + e] = st2[e];
        }
        __syncthreads();
        // thread (which, channel, part) adds 8 sources in a fixed order, a quad of lanes its four parts with two DPP steps
        const int part = tid & 3, ch = (tid >> 2) & 31, which = tid >> 7;
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) a += red[(which * 32 + part * 8 + m) * 33 + ch];
        a += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        a += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        if (part == 0) {
            float* dst = stat_partial + ((size_t)blockIdx.x * 2 + which) * Cout + st_n0 + ch;
            *dst = (stat_accum || ((st_seen >> (st_n0 >> 5)) & 1)) ? *dst + a : a;
        }
        st_seen |= 1u << (st_n0 >> 5);
        st1 = f32x4{0.f, 0.f, 0.f, 0.f};
        st2 = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    int g = 0;                                              // chunk of the stream
    for (int it = 0; it < my_items; ++it) {
        const i32x4 ce = tab[2 * it];
        const int brick = __builtin_amdgcn_readfirstlane(ce[0]), b = __builtin_amdgcn_readfirstlane(ce[2]);
        const int cpk = __builtin_amdgcn_readfirstlane(ce[3]);
        const int d0 = (cpk & 1023) * BD, h0 = ((cpk >> 10) & 1023) * BH, w0 = (cpk >> 20) * BW;
        const int n0 = ci_n0;
        TRQ(0);
        auto chunk = [&](int c, auto first_c) {
            // P0: ph 1, 2 of chunk g; meanwhile rows 0, 3 -> ph 0, 3 and their weights
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                if (k == P_BSTAGE) load_b(B03, 0, 3, c, n0);       // (half a phase ahead: their registers are free for the row in flight until here)
                T2(g & 1, k);
                mfma_k(1, 2, B12, k, first_c);
            }
            __builtin_amdgcn_sched_barrier(0);
            TRQ(1 + 4 * (c & 3));
            dma_wait();
            __syncthreads();                                // every wave is done with the halo of chunk g; chunk g + 1's has landed
            TRQ(2 + 4 * (c & 3));
            bpin(B03);
            const bool dma_go = dm_i < my_items;            // halo of chunk g + 2: two copies per group of MFMAs below
            // (behind the last chunk of the stream the "next chunk" is a repeat of valid addresses: its transform and weights are never used)
            int nc = c + 1;
            if (nc == nchunk) {
                nc = 0;
                const i32x4 e = tab[2 * (it + 1 < my_items ? it + 1 : it)];
                ci_n0 = __builtin_amdgcn_readfirstlane(e[1]) * 32;
            }
            // P1: ph 0, 3 of chunk g; meanwhile rows 1, 2 of chunk g + 1 -> its ph 1, 2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                if (k == P_BSTAGE) load_b(B12, 1, 2, nc, ci_n0);
                if (k == 0 && dma_go) dma_issue(g & 1);   // (spread over the groups, two copies each, it measured 2 % slower here; the
                                                          //  weight-gradient kernel's sixteen copies per stage gain 8 % from being spread)
                T1((g + 1) & 1, k);
                mfma_k(0, 3, B03, k, first_c);
            }
            __builtin_amdgcn_sched_barrier(0);
            TRQ(3 + 4 * (c & 3));
            dma_wait();
            TRQ(4 + 4 * (c & 3));
            bpin(B12);
            ++g;
        };
        chunk(0, std::true_type{});
        for (int c = 1; c < nchunk; ++c) chunk(c, std::false_type{});

#if P_ABL & 8
        {
            float sm = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sm += acc_read(acc[q][0]) + acc_read(acc[q][7]);
            if (sm == 12345.678f) z[tid] = sm + (float)(b + d0 + h0 + w0 + brick + n0);
            continue;
        }
#endif
        // ---- output transform: w and h in registers (accumulator-row pairs), d through LDS ----
        // The exchange is TRANSPOSED on the way: the writer lane (channel l31, row half hsel) stores single floats at
        // ex[pd][tile][ho][wo][channel]; the reader lane (channel quad cq = lane & 7, wo = lane >> 3 & 1, vq = lane >> 4) reads float4s
        // of 4 channels for its two tiles 8 w' + 4 j + vq: 16-byte stores to z (8 instead of 32 per lane; one instruction covers 8
        // consecutive w voxels x 128 bytes) and a conflict-free ds_read_b128 (the two wo rows of a tile are 32 floats apart).
        {
            float* exw = ex + (wave * 32 + 4 * hsel) * 128 + l31;        // accumulator row r -> tile (r & 3) + 8 (r >> 2) + 4 hsel
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                f32x2 y[4][2];
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    f32x2 m[4];
#pragma unroll
                    for (int pw = 0; pw < 4; ++pw) m[pw] = f32x2{acc_read(acc[ph * 4 + pw][2 * rp]), acc_read(acc[ph * 4 + pw][2 * rp + 1])};
                    y[ph][0] = (m[0] + m[1]) + m[2];
                    y[ph][1] = (m[1] - m[2]) - m[3];
                }
#pragma unroll
                for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                    for (int wo = 0; wo < 2; ++wo) {
                        const f32x2 pv = ho == 0 ? (y[0][wo] + y[1][wo]) + y[2][wo] : (y[1][wo] - y[2][wo]) - y[3][wo];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * rp + e, trow = (r & 3) + 8 * (r >> 2);
                            exw[((trow * 2 + ho) * 2 + wo) * 32] = pv[e];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);          // (one row pair at a time: the next item's operands stay in registers)
            }
        }
        TRQ(20);
        __syncthreads();
        TRQ(21);
        // reader: out[do 0] = S_0 + S_1 + S_2, out[do 1] = S_1 - S_2 - S_3 over the four pd
        const int cq = lane & 7, rwo = (lane >> 3) & 1, vq = lane >> 4;
        f32x4 outv[2][2][2];                                // [j][ho][do]: tile 8 w' + 4 j + vq, channels 4 cq .. 4 cq + 3
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ho = 0; ho < 2; ++ho) {
                f32x4 S[4];
#pragma unroll
                for (int pd = 0; pd < 4; ++pd)
                    S[pd] = *reinterpret_cast<const f32x4*>(&ex[((((pd * 32 + 8 * wave + 4 * j + vq) * 2 + ho) * 2 + rwo) * 32) + 4 * cq]);
                outv[j][ho][0] = (S[0] + S[1]) + S[2];
                outv[j][ho][1] = (S[1] - S[2]) - S[3];
            }
        TRQ(22);
        // voxel of (tile 8 w' + 4 j + vq, do, ho, wo): d = gdb + 2 j jd + do, h = ghb + 2 j jh + ho, w = gw (jd, jh: which axis j steps)
        const int co = n0 + 4 * cq;
        constexpr int jd = GEOM == 0 ? 0 : 1, jh = GEOM == 0 ? 1 : 0;
        const int gdb = d0 + 2 * st_td, ghb = h0 + 2 * st_th, gw = w0 + 2 * st_tw + rwo;
        const int bs = b + st_ts;                           // this wave's sample
        if constexpr (MODE == 2) {
            f32x4 sc, sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = aff_scale[co + e]; sh[e] = aff_shift[co + e]; }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = outv[j][ho][dd][e] * sc[e] + sh[e];
                            outv[j][ho][dd][e] = y > 0.f ? y : y * slope;
                        }
            if (pool == TMF_POOL_MAX2) {
                const int OD = D / 2, OH = H / 2, OW = W / 2;
                float* yb = z + (size_t)bs * OD * OH * OW * Cout;
                const int ow = (w0 >> 1) + st_tw;
#pragma unroll
                for (int j = 0; j < 2; ++j) {                  // the window = the tile: d and h pairs in this lane, the w pair in lane ^ 8
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaxf(fmaxf(outv[j][0][0][e], outv[j][0][1][e]), fmaxf(outv[j][1][0][e], outv[j][1][1][e]));
                        m[e] = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)));       // row_ror:8 = lane ^ 8
                    }
                    const int od = (gdb >> 1) + j * jd, oh = (ghb >> 1) + j * jh;
                    if (rwo == 0 && bs < B && od < OD && oh < OH && ow < OW) *reinterpret_cast<f32x4*>(&yb[((size_t)(od * OH + oh) * OW + ow) * Cout + co]) = m;
                }
            } else {
                float* yb = z + (size_t)bs * D * H * W * Cout;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                        for (int dd = 0; dd < 2; ++dd) {
                            const int gd = gdb + 2 * j * jd + dd, gh = ghb + 2 * j * jh + ho;
                            if (bs < B && gd < D && gh < H && gw < W) *reinterpret_cast<f32x4*>(&yb[((size_t)(gd * H + gh) * W + gw) * Cout + co]) = outv[j][ho][dd];
                        }
            }
        } else {
            // (the resource covers the brick's samples that exist: a lane of a sample beyond the batch is out of range and dropped)
            float* zb = z + (size_t)b * D * H * W * Cout;
            const int ns = B - b < BS ? B - b : BS;
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(zb, 0, ns * D * H * W * Cout * 4, 0x00020000);
            const bool full = d0 + BD <= D && h0 + BH <= H && w0 + BW <= W && b + BS <= B;
            if (STATS && stat_partial != nullptr && st_n0 != n0) {    // a new channel group: the previous one's sums leave
                if (st_n0 >= 0) stat_flush();
                st_n0 = n0;
            }
            // address = this lane's part (st_lane: tile and channel quad, fixed for the kernel) + the brick's corner and the row
            // (d, h) of the store in the scalar offset: one s_add per store
            const int st_item = (((d0 * H + h0) * W + w0) * Cout + n0) * 4;
            auto put = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
                const bool w_ok = FULL || (gw < W && bs < B);
                const int voff = w_ok ? st_lane : OOB;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                        for (int dd = 0; dd < 2; ++dd) {
                            const int gd = gdb + 2 * j * jd + dd, gh = ghb + 2 * j * jh + ho;      // (wave-uniform for GEOM 0; per lane for GEOM 1)
                            const bool row_ok = FULL || (gd < D && gh < H);
                            f32x4 v = outv[j][ho][dd];
                            if (GEOM == 0) {
                                if (row_ok)
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tmf_u32x4, v), zr, voff, st_item + j * st_j + (dd * H + ho) * W * Cout * 4, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tmf_u32x4, v), zr, row_ok ? voff : OOB, st_item + j * st_j + (dd * H + ho) * W * Cout * 4, 0);
                            }
                            if (STATS && !(P_ABL & 128)) {
                                if (!(P_ABL & 256)) store_guard();
                                if (!FULL) v = (row_ok && w_ok) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                                st1 += v;
                                st2 += v * v;
                            }
                        }
            };
            if (full) put(std::true_type{});
            else put(std::false_type{});
            TRQ(23);
        }
        TRQ(24);
    }
    if constexpr (STATS) {
        if (stat_partial != nullptr) {
            if (st_n0 >= 0) stat_flush();
            if (!stat_accum) {
                // the channel groups this workgroup never saw, and (workgroup 0) the rows no workgroup owns: zeros
                for (int ng = 0; ng < Cout / 32; ++ng)
                    if (!((st_seen >> ng) & 1) && tid < 64) stat_partial[((size_t)blockIdx.x * 2 + (tid >> 5)) * Cout + ng * 32 + (tid & 31)] = 0.f;
                if (blockIdx.x == 0)
                    for (int i = (int)gridDim.x * 2 * Cout + tid; i < stat_rows * 2 * Cout; i += PN) stat_partial[i] = 0.f;
            }
        }
    }
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

// ------------------------------------------------------------------------------------------------------------
// The weight gradient as ONE WAVE PER SIMD (conv3d_wino_wgrad_p_kernel; round 5) — the reasons of conv3d_wino_p_kernel:
// the fp32 matrix instructions run on the vector ALU and a lone wave issues one instruction per ~5 cycles, so what counts is
// the NUMBER of instructions beside the MFMAs.  The kernel above spends 9.6 of them per MFMA (one channel and one tile per
// lane and instruction); here:
//   * 4 waves, wave = pd: 16 positions x 32 ci x 32 co = 256 accumulator registers (AGPRs);
//   * a lane (channel l31, tile parity hsel) transforms TWO tiles at once — w origins 4 apart, i.e. 512 bytes apart in the
//     stage buffer: one ds_read2st64_b32 fetches both into a register pair and the transform runs on v_pk_*_f32
//     (32 + 8 paired reads and ~64 packed adds per 32 MFMAs: 3.3 instructions per MFMA);
//   * the minus signs of Z = A dz A^T (z3 = -y1 on every axis) are left out of the main loop: position p carries the sign
//     (-1)^[pd = 3] (-1)^[ph = 3] (-1)^[pw = 3], applied where the kernel turns its sums into dw (G^T dU G per workgroup, once);
//   * the step of four tiles is two phases of 16 MFMAs: P0 multiplies ph 1, 2 while the rows 0, 3 are read and turned into
//     ph 0, 3; P1 multiplies ph 0, 3 while the rows 1, 2 of the NEXT step become its ph 1, 2 — no second copy of the operands,
//     across stage boundaries too (one barrier per stage of 16 tiles, the stage after next in flight by LDS-DMA); every row
//     is read a group of MFMAs before it is used.
template <int C, int PAIR = 2>
__device__ __forceinline__ f32x2 lds_pair(unsigned base_even, unsigned base_odd) {      // floats at 128 C and 128 C + 256 PAIR bytes
    f32x2 d;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"((C & 1) ? base_odd : base_even), "n"(C >> 1), "n"((C >> 1) + PAIR));
    return d;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
constexpr int WPN = 256;                                          // threads: 4 waves, wave = pd
// The stage of 16 tiles (a lane's pair = two tiles one ds_read2st64 apart) in two geometries, as in the forward:
//   GEOM 0: 2 x 2 x 4 tiles of one sample — a 4x4x8 half brick, halo [6][6][10] voxels, the pair 4 voxels apart in w;
//   GEOM 1: 2 x 2 x 2 tiles of TWO samples — a 4x4x4 brick of each, halo [6][6][sample 2][6] (the pair = the two samples, 6 voxel
//           rows apart), dz [4][4][sample 2][4]: a 12^3 volume is 27 such bricks instead of 18 half bricks a third of whose w
//           range lies outside (conv4.0 of the 96^3 input).  The main loop differs in three constants.
template <int GEOM> struct WGeomP {
    static constexpr int ROWW = GEOM == 0 ? 10 : 12, PLANE = 6 * ROWW, PAIR = GEOM == 0 ? 2 : 3, BW = GEOM == 0 ? 8 : 4, BS = GEOM == 0 ? 1 : 2;
    static constexpr int XREAL = 36 * ROWW * 8;                           // 16-byte pieces of the halo
    static constexpr int XDMA = (XREAL + 4 * 64 - 1) / (4 * 64);          // LDS-DMA instructions per wave and stage: 12 / 14 (+ 4 for dz)
    static constexpr int XSLOTS = XDMA * WPN;
    static constexpr int BUF_BYTES = (XSLOTS + WZ_SLOTS) * 16;            // 64 KB / 72 KB per stage buffer
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES;
};
constexpr int WP_ZDMA = WZ_SLOTS / WPN;

template <int GEOM>
__global__ __launch_bounds__(WPN) void conv3d_wino_wgrad_p_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    int B, int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int nbricks, int per_split, int ncob) {
    using WG = WGeomP<GEOM>;
    constexpr int WP_XDMA = WG::XDMA, WX_SLOTS = WG::XSLOTS, WX_REAL = WG::XREAL, WBUF_BYTES = WG::BUF_BYTES;
    constexpr int ROWW = WG::ROWW, PLANE = WG::PLANE, PAIR = WG::PAIR;
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

    // ---- staging plan: per lane and instruction the voxel's offset from the stage's corner and one-hot masks of its coordinates;
    // per stage one AND + compare + add + select per copy ----
    int relx[WP_XDMA], mx[WP_XDMA], relz[WP_ZDMA], mz[WP_ZDMA];
#pragma unroll
    for (int q = 0; q < WP_XDMA; ++q) {
        const int e = (q * 4 + wave) * 64 + lane, v = e >> 3, piece = e & 7;
        const int hd = v / PLANE, hh = (v / ROWW) % 6, hr = v % ROWW, s_ = GEOM == 0 ? 0 : hr / 6, hw = GEOM == 0 ? hr : hr % 6;
        relx[q] = (((s_ * D + hd) * H + hh) * W + hw) * Cin * 4 + piece * 16;
        mx[q] = e < WX_REAL ? (1 << hd) | (1 << (8 + hh)) | (1 << (16 + hw)) | (1 << (26 + s_)) : (1 << 30);
    }
#pragma unroll
    for (int q = 0; q < WP_ZDMA; ++q) {
        const int e = (q * 4 + wave) * 64 + lane, v = e >> 3, piece = e & 7;
        const int od = v >> 5, oh = (v >> 3) & 3, s_ = GEOM == 0 ? 0 : (v >> 2) & 1, ow = GEOM == 0 ? v & 7 : v & 3;
        relz[q] = (((s_ * D + od) * H + oh) * W + ow) * Cout * 4 + piece * 16;
        mz[q] = (1 << od) | (1 << (8 + oh)) | (1 << (16 + ow)) | (1 << (26 + s_));
    }
    // the half brick the next copy belongs to: decoded once (stage_first), then advanced by one (stage_next: no divisions)
    int sbw = 0, sbh = 0, sbd = 0, sb = 0;
    auto stage_first = [&](int st) {
        int t = st;
        sbw = t % tilesW; t /= tilesW;
        sbh = t % tilesH; t /= tilesH;
        sbd = t % tilesD;
        sb = t / tilesD;
    };
    auto stage_next = [&]() {
        if (++sbw == tilesW) { sbw = 0; if (++sbh == tilesH) { sbh = 0; if (++sbd == tilesD) { sbd = 0; ++sb; } } }
    };
    // The 16 copies of a stage go out in four parts of four, one part per phase: sixteen back-to-back LDS-DMA instructions
    // fill the CU's address queue and every further one waits in the issue slot (the whole plan issued at the barrier cost 15 %
    // of the kernel).  stage_setup: the scalars of the half brick (sb, sbd, sbh, sbw); stage_part(i): copies 4 i .. 4 i + 3.
    i32x4 s_xr, s_zr;
    int s_xcorner = 0, s_zcorner = 0, s_vmx = 0, s_vmz = 0;
    unsigned s_base = 0;
    auto stage_setup = [&](int buf) {
        const int b = sb * WG::BS, d0 = sbd * 4, h0 = sbh * 4, w0 = sbw * WG::BW;
        const int ns = B - b < WG::BS ? B - b : WG::BS;                  // (a last pair of one sample: the other's lanes are masked)
        s_xr = make_rsrc(x + (size_t)b * D * H * W * Cin, (unsigned)(ns * D * H * W * Cin * 4));
        s_zr = make_rsrc(dz + (size_t)b * D * H * W * Cout, (unsigned)(ns * D * H * W * Cout * 4));
        s_xcorner = (((d0 - 1) * H + (h0 - 1)) * W + (w0 - 1)) * Cin * 4 + ci0 * 4;       // (negative at the faces)
        s_zcorner = ((d0 * H + h0) * W + w0) * Cout * 4 + co0 * 4;
        auto range = [](int lo, int hi) { return ((1 << hi) - 1) & ~((1 << lo) - 1); };
        auto mn = [](int a, int b_) { return a < b_ ? a : b_; };
        s_vmx = range(d0 == 0 ? 1 : 0, mn(6, D - d0 + 1)) | (range(h0 == 0 ? 1 : 0, mn(6, H - h0 + 1)) << 8) |
                (range(w0 == 0 ? 1 : 0, mn(WG::BW + 2, W - w0 + 1)) << 16) | (range(0, ns) << 26);
        s_vmz = range(0, mn(4, D - d0)) | (range(0, mn(4, H - h0)) << 8) | (range(0, mn(WG::BW, W - w0)) << 16) | (range(0, ns) << 26);
        s_base = lds0 + buf * WBUF_BYTES + wave * 1024;
    };
    auto stage_part = [&](int part) {
        if (P_ABL & 4) return;
#pragma unroll
        for (int q = 0; q < WP_XDMA; ++q)
            if (q * 3 / WP_XDMA == part) blds16((mx[q] & s_vmx) == mx[q] ? relx[q] + s_xcorner : OOB, s_xr, 0, s_base + q * 4096);
#pragma unroll
        for (int q = 0; q < WP_ZDMA; ++q)
            if (3 == part) blds16((mz[q] & s_vmz) == mz[q] ? relz[q] + s_zcorner : OOB, s_zr, 0, s_base + WX_SLOTS * 16 + q * 4096);
    };
    auto stage_issue = [&](int buf) {               // the whole stage at once (prologue)
        stage_setup(buf);
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_part(i);
    };

    // ---- this wave's d row of the transforms ----
    const int da = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int db = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.f : -1.f;
    const f32x2 s2 = {sgn, sgn};
    const float cz = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
    const f32x2 cz2 = {cz, cz};
    const unsigned zfirst = wave == 3 ? 4 * 8 * 32 * 4 : 0;          // pd 3: the plane do 1 (4 x 8 voxels x 128 bytes further)
    // LDS byte addresses of this lane's tile pair (w origins 2 hsel and 2 hsel + 4) in buffer 0: x plane da / db and dz
    // (ds_read2st64 offsets count 256 bytes: the odd 128-byte rows go through a base 128 bytes further)
    const unsigned lx = lds0 + (2 * hsel * 32 + l31) * 4;
    const unsigned xa0 = lx + da * PLANE * 128, xb0 = lx + db * PLANE * 128, zz0 = lx + WX_SLOTS * 16;
    f32x2 A[16], Bv[16];                            // [ph][pw] -> {tile set 0, tile set 1}
#if P_ABL & 1
    for (int i = 0; i < 16; ++i) { A[i] = f32x2{(float)tid, 1.f}; Bv[i] = f32x2{(float)lane, 2.f}; }
#endif
    f32x2 u1[4], u2[4], zd[2][2];                   // kept between the two halves of a step: w-transformed rows 1, 2; d-combined dz [ho][wo]
    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    auto pin4 = [](f32x2& a, f32x2& b, f32x2& c, f32x2& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); };
    // one h row (2 th + i) of the step (td, th): planes da / db, the four w taps — issued a group of MFMAs ahead of its use
    // (row_issue), then d combination -> w transform (row_finish).  The reads are opaque to the compiler: wait + pin before use.
    f32x2 ra[4], rb[4];                             // the row in flight
    auto row_issue = [&](unsigned ea, unsigned eb, auto c_c) {
        constexpr int Cb = decltype(c_c)::value;    // row offset in 128-byte units: ((2 td) * PLANE + (2 th + i) * ROWW)
        static_for<4>([&](auto k_c) {
            constexpr int K = decltype(k_c)::value;
            ra[K] = lds_pair<Cb + K, PAIR>(ea, ea + 128);
            rb[K] = lds_pair<Cb + K, PAIR>(eb, eb + 128);
        });
    };
    auto row_finish = [&](f32x2 (&u)[4]) {
        lds_wait();
        pin4(ra[0], ra[1], ra[2], ra[3]);
        pin4(rb[0], rb[1], rb[2], rb[3]);
        pk_row4(u, ra, rb, s2);
    };
    // rows pha, phb of B = [y0, y0 + y1, y0 - y1, y1] each: the 1-D transform of Z without its last sign
    auto zrows = [&](f32x2 ya0, f32x2 ya1, int pha, f32x2 yb0, f32x2 yb1, int phb) {
        Bv[pha * 4 + 0] = ya0; Bv[pha * 4 + 3] = ya1; Bv[phb * 4 + 0] = yb0; Bv[phb * 4 + 3] = yb1;
        pk_z2(Bv[pha * 4 + 1], Bv[pha * 4 + 2], Bv[phb * 4 + 1], Bv[phb * 4 + 2], ya0, ya1, yb0, yb1);
    };
    f32x2 yq[2][2][2];                              // dz in flight: [slot][ho][wo]
    f32x2 v0[4];
    // T1 (three parts, one per group of MFMAs of the phase it rides under): rows 1, 2 of step (td, th) in buffer `par` -> A, B of
    // ph 1, 2; keeps u1, u2 and zd for T2.  dz: z_d = y[do 0] | y0 + y1 | y0 - y1 | y[do 1] for pd = 0 .. 3 = first + cz * second,
    // where pd 3 reads the plane do 1 as its "first" (a wave-uniform address) and cz = 0, 1, -1, 0: no branch
    auto T1 = [&](int par, auto step_c, int part) {
        constexpr int STEP = decltype(step_c)::value, TDs = STEP >> 1, THs = STEP & 1;
        if (P_ABL & 1) return;
        const unsigned off = par * WBUF_BYTES;
        if (part == 0) {
            row_issue(xa0 + off, xb0 + off, std::integral_constant<int, (2 * TDs) * PLANE + (2 * THs + 1) * ROWW>{});
            static_for<2>([&](auto ho_c) {
                static_for<2>([&](auto wo_c) {
                    constexpr int HO = decltype(ho_c)::value, WO = decltype(wo_c)::value;
                    constexpr int C0 = ((2 * TDs + 0) * 4 + (2 * THs + HO)) * 8 + WO, C1 = ((2 * TDs + 1) * 4 + (2 * THs + HO)) * 8 + WO;
                    yq[0][HO][WO] = lds_pair<C0>(zz0 + zfirst + off, zz0 + zfirst + off + 128);
                    yq[1][HO][WO] = lds_pair<C1>(zz0 + off, zz0 + off + 128);
                });
            });
        } else if (part == 1) {
            row_finish(u1);
            pin4(yq[0][0][0], yq[0][0][1], yq[0][1][0], yq[0][1][1]);
            pin4(yq[1][0][0], yq[1][0][1], yq[1][1][0], yq[1][1][1]);
            row_issue(xa0 + off, xb0 + off, std::integral_constant<int, (2 * TDs) * PLANE + (2 * THs + 2) * ROWW>{});
#pragma unroll
            for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                for (int wo = 0; wo < 2; ++wo) zd[ho][wo] = pk_fma(cz2, yq[1][ho][wo], yq[0][ho][wo]);
            {
                f32x2 zs0, zdf0, zs1, zdf1;             // ho sum / difference per wo
                pk_z2(zs0, zdf0, zs1, zdf1, zd[0][0], zd[1][0], zd[0][1], zd[1][1]);
                zrows(zs0, zs1, 1, zdf0, zdf1, 2);
            }
            pin4(u1[0], u1[1], u1[2], u1[3]);
            pin4(Bv[4], Bv[5], Bv[6], Bv[7]); pin4(Bv[8], Bv[9], Bv[10], Bv[11]);
        } else {
            row_finish(u2);
            {
                f32x2 sm[4], df[4];
                pk_addsub4(sm, df, u2, u1);             // ph 1 = u1 + u2, ph 2 = u2 - u1
#pragma unroll
                for (int k = 0; k < 4; ++k) { A[4 + k] = sm[k]; A[8 + k] = df[k]; }
            }
            pin4(A[4], A[5], A[6], A[7]); pin4(A[8], A[9], A[10], A[11]);
        }
    };
    // T2: rows 0, 3 of the same step -> A, B of ph 0, 3
    auto T2 = [&](int par, auto step_c, int part) {
        constexpr int STEP = decltype(step_c)::value, TDs = STEP >> 1, THs = STEP & 1;
        if (P_ABL & 1) return;
        const unsigned off = par * WBUF_BYTES;
        if (part == 0) {
            row_issue(xa0 + off, xb0 + off, std::integral_constant<int, (2 * TDs) * PLANE + (2 * THs + 0) * ROWW>{});
            zrows(zd[0][0], zd[0][1], 0, zd[1][0], zd[1][1], 3);
            pin4(Bv[0], Bv[1], Bv[2], Bv[3]); pin4(Bv[12], Bv[13], Bv[14], Bv[15]);
        } else if (part == 1) {
            row_finish(v0);
            row_issue(xa0 + off, xb0 + off, std::integral_constant<int, (2 * TDs) * PLANE + (2 * THs + 3) * ROWW>{});
            {
                f32x2 d[4];
                pk_sub4(d, v0, u2);
#pragma unroll
                for (int k = 0; k < 4; ++k) A[k] = d[k];
            }
            pin4(A[0], A[1], A[2], A[3]);
        } else {
            row_finish(v0);
            {
                f32x2 d[4];
                pk_sub4(d, u1, v0);
#pragma unroll
                for (int k = 0; k < 4; ++k) A[12 + k] = d[k];
            }
            pin4(A[12], A[13], A[14], A[15]);
        }
    };
    // a phase's 16 MFMAs (both tile sets of the 8 positions ph in {pha, phb}) in three groups of 4, 4, 8: part g of the other
    // half's transform rides under group g
    auto mfma_g = [&](int pha, int phb, int g) {
        const int k0 = g == 2 ? 0 : 2 * g, k1 = g == 2 ? 4 : 2 * g + 2, s0 = g == 2 ? 1 : 0;
#pragma unroll
        for (int k = k0; k < k1; ++k) {
#if P_ABL & 16
            acc[pha * 4 + k][s0] += A[pha * 4 + k][s0] * Bv[pha * 4 + k][s0]; acc[phb * 4 + k][s0] += A[phb * 4 + k][s0] * Bv[phb * 4 + k][s0];
#else
            acc[pha * 4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[pha * 4 + k][s0], Bv[pha * 4 + k][s0], acc[pha * 4 + k], 0, 0, 0);
            acc[phb * 4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[phb * 4 + k][s0], Bv[phb * 4 + k][s0], acc[phb * 4 + k], 0, 0, 0);
#endif
        }
    };

    if (st0 < st1) {
        stage_first(st0);
        stage_issue(0);
        if (st0 + 1 < st1) { stage_next(); stage_issue(1); }
        dma_wait();
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 3; ++g) T1(0, std::integral_constant<int, 0>{}, g);
        bool pend = false;                              // a stage's copies are going out (parts 1 .. 3 in the phases behind the barrier's)
        for (int st = st0; st < st1; ++st) {
            const int par = (st - st0) & 1;
            static_for<4>([&](auto step_c) {
                constexpr int STEP = decltype(step_c)::value;
                // P0: ph 1, 2 of this step; meanwhile rows 0, 3 -> ph 0, 3
                __builtin_amdgcn_sched_barrier(0);
                if (STEP < 2 && pend) { stage_part(1 + 2 * STEP); if (STEP == 1) pend = false; }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    T2(par, step_c, g);
                    mfma_g(1, 2, g);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (STEP == 3) {                        // the last read of this stage's buffer is done
                    dma_wait();
                    __syncthreads();                    // every wave is done with stage st; stage st + 1 has landed
                    if (st + 2 < st1) { stage_next(); stage_setup(par); stage_part(0); pend = true; }
                }
                if (STEP == 0 && pend) stage_part(2);
                // P1: ph 0, 3 of this step; meanwhile rows 1, 2 of the next step (the next stage's first step behind the last one;
                // behind the last stage the "next step" re-reads valid LDS: its operands are never used)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    T1(STEP < 3 ? par : par ^ 1, std::integral_constant<int, (STEP + 1) & 3>{}, g);
                    mfma_g(0, 3, g);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }

    // ---- this workgroup's share of dw = G^T dU G, straight into its slab [split][27][cin][cout] (tap-major, the full tensor: the
    // blocks of one split fill disjoint parts): G^T along w and h on this wave's 16 positions in registers, along d across the
    // four waves through LDS (two halves of the accumulator rows: 72 KB each).  1-D: w0 = u0 + (u1 + u2) / 2, w1 = (u1 - u2) / 2,
    // w2 = (u1 + u2) / 2 + u3, where u3 carries the sign the Z transform left out (so: ... - u3).  The slabs are 27 / 64 of the
    // 64-position ones and tmf_reduce_slabs writes dw itself: no finish launch.
    // accumulator row r -> ci = (r & 3) + 8 (r >> 2) + 4 hsel, column co = l31 ----
    __syncthreads();                                    // (every wave is done with the stage buffers: the exchange re-uses them)
    float* slab = partial + (size_t)split * 27 * Cin * Cout + (size_t)ci0 * Cout + co0 + l31;
    auto gt3 = [](float u0, float u1, float u2, float u3, float (&o)[3]) {
        const float sm = u1 + u2, df = u1 - u2;
        o[0] = u0 + 0.5f * sm; o[1] = 0.5f * df; o[2] = 0.5f * sm - u3;
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = half * 8 + rr;
            float t[4][3], v[3][3];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
                gt3(acc_read(acc[ph * 4 + 0][r]), acc_read(acc[ph * 4 + 1][r]), acc_read(acc[ph * 4 + 2][r]), acc_read(acc[ph * 4 + 3][r]), t[ph]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float o[3];
                gt3(t[0][j], t[1][j], t[2][j], t[3][j], o);
                v[0][j] = o[0]; v[1][j] = o[1]; v[2][j] = o[2];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) smem[((wave * 8 + rr) * 9 + i * 3 + j) * 64 + lane] = v[i][j];
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // wave w' combines the four pd of the (row, h-w tap) pairs e = 18 w' .. 18 w' + 17 of this half
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int e = wave * 18 + q, rr = e / 9, ij = e % 9;
            float V[4], o[3];
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) V[pd] = smem[((pd * 8 + rr) * 9 + ij) * 64 + lane];
            gt3(V[0], V[1], V[2], V[3], o);
            const int r = half * 8 + rr, ci = (r & 3) + 8 * (r >> 2) + 4 * hsel;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) slab[((size_t)(kd * 9 + ij) * Cin + ci) * Cout] = o[kd];
        }
        __syncthreads();
    }
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

int wino_p_mode();
struct WinoWgPlan { int geom, tilesD, tilesH, tilesW, nbricks, nblk, ncob, nsplit, per; };
WinoWgPlan plan_wino_wgrad(int B, int D, int H, int W, int cin, int cout) {
    WinoWgPlan p;
    p.tilesD = tmf_cdiv(D, 4); p.tilesH = tmf_cdiv(H, 4); p.tilesW = tmf_cdiv(W, 8);
    p.nbricks = B * p.tilesD * p.tilesH * p.tilesW;
    p.geom = 0;
    // the persistent kernel's second geometry (stages of two samples x 4x4x4 voxels) where that is fewer stages; both samples sit
    // behind one buffer resource with 32-bit byte offsets
    const long two = 2L * D * H * W * (cin > cout ? cin : cout) * 4;
    const int nb1 = tmf_cdiv(B, 2) * p.tilesD * p.tilesH * tmf_cdiv(W, 4);
    if (wino_p_mode() && nb1 < p.nbricks && two < (1L << 31)) {
        p.geom = 1;
        p.tilesW = tmf_cdiv(W, 4);
        p.nbricks = nb1;
    }
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
// thread e -> (co, ci).  The bf16 parts are 2-byte elements, eight K values (16 bytes) per (position, part, K half, N): with the K
// index in the low bits of the thread index a wave's 64 stores are 128 contiguous bytes — with N there (the fp32 layouts' natural
// order) they were 64 separate 2-byte requests 16 bytes apart and the pack launch took 88 us instead of 25.  K is ci for the
// forward layout and co for the data-gradient layout, so a layer with both split layouts is packed in two passes (which = 1, 2),
// each writing one of them; which = 0: one pass, N-major (layers without a split layout).
__device__ __forceinline__ void wino_pack_one(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgrad,
                                              int cout, int cin, int e, int which = 0) {
    if (e >= cout * cin) return;
    int co, ci;
    if (which == 1 && cin % 16 == 0) {          // forward: j = e & 7 -> ci = 16 c + 8 (j >> 2) + 4 h + (j & 3), then co, then (c, h)
        const int j = e & 7, r = e >> 3;
        co = r % cout;
        const int ch = r / cout;
        ci = 16 * (ch >> 1) + 8 * (j >> 2) + 4 * (ch & 1) + (j & 3);
    } else if (which == 2 && cout % 16 == 0) {  // data gradient: the same with the roles of co and ci swapped
        const int j = e & 7, r = e >> 3;
        ci = r % cin;
        const int ch = r / cin;
        co = 16 * (ch >> 1) + 8 * (j >> 2) + 4 * (ch & 1) + (j & 3);
    } else {
        co = e % cout; ci = e / cout;
    }
    if (which == 1) dgrad = nullptr;
    if (which == 2) fwd = nullptr;
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
                // the same number as three bf16 parts (exact: 8 + 8 + 8 significand bits) for conv3d_winox.hip, behind the fp32 tensor:
                //   [p][K / 16][part h, m, l][K half 2][N][8] bf16,  K index 16 c + 8 s + 4 half + e  ->  element s * 4 + e of its half
                const unsigned uv = __builtin_bit_cast(unsigned, v);
                const float r1 = v - __builtin_bit_cast(float, uv & 0xFFFF0000u);
                const unsigned ur1 = __builtin_bit_cast(unsigned, r1);
                const float r2 = r1 - __builtin_bit_cast(float, ur1 & 0xFFFF0000u);
                const unsigned short part[3] = {(unsigned short)(uv >> 16), (unsigned short)(ur1 >> 16),
                                                (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16)};
                if (fwd != nullptr) {
                    const int pos = (p * 4 + q) * 4 + r;
                    fwd[((((size_t)pos * (cin / 8) + ci / 8) * 2 + (ci >> 2 & 1)) * cout + co) * 4 + (ci & 3)] = v;
                    if (cin % 32 == 0) {
                        unsigned short* f3 = reinterpret_cast<unsigned short*>(fwd + (size_t)64 * cin * cout);
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            f3[(((((size_t)pos * (cin / 16) + ci / 16) * 3 + t) * 2 + (ci >> 2 & 1)) * cout + co) * 8 + (ci >> 3 & 1) * 4 + (ci & 3)] = part[t];
                    }
                }
                if (dgrad != nullptr) {
                    const int pos = (fl[p] * 4 + fl[q]) * 4 + fl[r];
                    dgrad[((((size_t)pos * (cout / 8) + co / 8) * 2 + (co >> 2 & 1)) * cin + ci) * 4 + (co & 3)] = v;
                    if (cout % 32 == 0) {
                        unsigned short* d3 = reinterpret_cast<unsigned short*>(dgrad + (size_t)64 * cin * cout);
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            d3[(((((size_t)pos * (cout / 16) + co / 16) * 3 + t) * 2 + (co >> 2 & 1)) * cin + ci) * 8 + (co >> 3 & 1) * 4 + (co & 3)] = part[t];
                    }
                }
            }
}

// one layer: both layouts with their split copies in two K-major passes, a single layout in one, anything else N-major
__device__ __forceinline__ void wino_pack_layer(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgrad,
                                                int cout, int cin, int e) {
    if (fwd != nullptr && dgrad != nullptr) {
        if (cin % 32 == 0 && cout % 32 == 0) {
            wino_pack_one(w, fwd, dgrad, cout, cin, e, 1);
            wino_pack_one(w, fwd, dgrad, cout, cin, e, 2);
        } else {
            wino_pack_one(w, fwd, dgrad, cout, cin, e, 0);
        }
    } else if (fwd != nullptr) {
        wino_pack_one(w, fwd, nullptr, cout, cin, e, cin % 32 == 0 ? 1 : 0);
    } else {
        wino_pack_one(w, nullptr, dgrad, cout, cin, e, cout % 32 == 0 ? 2 : 0);
    }
}

__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                                        float* __restrict__ dgrad, int cout, int cin) {
    wino_pack_layer(w, fwd, dgrad, cout, cin, blockIdx.x * 256 + threadIdx.x);
}
// the transformed weights of several layers in ONE launch (blockIdx.y = layer): an encoder's five Winograd blocks need ten
// of these small transforms per step, each its own ~10 us launch otherwise
struct WinoPackMulti { const float* w[8]; float* fwd[8]; float* dgrad[8]; int cout[8], cin[8]; };
__global__ __launch_bounds__(256) void wino_pack_multi_kernel(WinoPackMulti a) {
    const int l = blockIdx.y;
    wino_pack_layer(a.w[l], a.fwd[l], a.dgrad[l], a.cout[l], a.cin[l], blockIdx.x * 256 + threadIdx.x);
}

int g_conv_wino = -1;
int g_wino_p = -1;          // forward / data gradient: 1 the persistent one-wave-per-SIMD kernel (default), 0 the two-waves-per-SIMD one

int wino_p_mode() {
    if (const int o = tmf_algo_override()) return (o & TMF_SNET_ALGO_WINO_P) ? 1 : 0;
    if (g_wino_p < 0) {
        const char* e = getenv("TMF_WINO_P");
        g_wino_p = (e && atoi(e) == 0) ? 0 : 1;
    }
    return g_wino_p;
}

int wino_cu_count() {
    static thread_local int n[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) dev = 0;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        const char* e = getenv("TMF_WINO_CUS");             // experiments: fewer persistent workgroups than compute units
        if (e && atoi(e) > 0 && atoi(e) < v) v = atoi(e);
        n[dev] = v;
    }
    return n[dev];
}

// Brick geometry of the persistent kernel for a volume: 1 (four samples x 4x4x4) where that executes fewer tiles than 0 (4x8x8)
// (... and only for volumes below 2^17 voxels: the four samples of such a brick sit behind ONE buffer resource with 32-bit byte
//  offsets, 4 x D x H x W x channels x 4 B < 2^31 for up to 1 024 channels — a large ragged volume keeps the one-sample bricks, which
//  only need one sample below 2^29 elements; the advisor's finding, round 5.  The geometry depends on the volume alone, so every
//  query — bricks, kernel names, statistic rows — and the launch agree.)
int wino_p_geom(int B, int D, int H, int W) {
    const long t0 = (long)B * tmf_cdiv(D, 4) * tmf_cdiv(H, 8) * tmf_cdiv(W, 8);
    const long t1 = (long)tmf_cdiv(B, 4) * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 4);
    return (t1 < t0 && (long)D * H * W < (1L << 17)) ? 1 : 0;
}
// ... and with the channel counts (round 6): where the split kernel (conv3d_winox.hip: one-sample bricks only, ~1.2 x the fp32
// kernel per brick) can take the launch, the folded geometry must save more than that to be chosen — 22x27x22 at B = 8 (the
// reference's 91x109x91 volume two levels down): 576 one-sample bricks against 504 folded ones, and the split kernel on 576 is faster
int wino_fwd_geom(int B, int D, int H, int W, int cin, int cout) {
    const int g = wino_p_geom(B, D, H, W);
    if (g == 0 || !tmf_winox_takes(B, D, H, W, cin, cout, 0)) return g;
    const long t0 = (long)B * tmf_winox_items(D, H, W, nullptr);       // (the better of the split kernel's two item orientations)
    const long t1 = (long)tmf_cdiv(B, 4) * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 4);
    return 100 * t0 <= 115 * t1 ? 0 : 1;
}
long wino_p_bricks(int geom, int B, int D, int H, int W) {
    return geom ? (long)tmf_cdiv(B, 4) * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 4)
                : (long)B * tmf_cdiv(D, 4) * tmf_cdiv(H, 8) * tmf_cdiv(W, 8);
}

template <int MODE, int GEOM>
int launch_wino_p_g(const char* what, const float* x, const float* u, float* z, float* stat_partial, int B, int D, int H, int W,
                    int cin, int cout, const float* scale, const float* shift, float slope, int pool, hipStream_t stream) {
    using GM = PGeom<GEOM>;
    const int tilesD = tmf_cdiv(D, GM::BD), tilesH = tmf_cdiv(H, GM::BH), tilesW = tmf_cdiv(W, GM::BW);
    TMF_REQUIRE(tilesD < 1024 && tilesH < 1024 && tilesW < 1024, TMF_E_SHAPE, "%s: more than 1023 bricks along one axis", what);
    TMF_REQUIRE((long)GM::BS * D * H * W * (cin > cout ? cin : cout) < (1L << 29), TMF_E_SHAPE,
                "%s: the samples of one brick exceed 2^29 elements (32-bit byte offsets)", what);
    const long nbricks = wino_p_bricks(GEOM, B, D, H, W);
    const long nitems = nbricks * (cout / 32);
    TMF_REQUIRE(nitems < (1L << 30), TMF_E_SHAPE, "%s: too many bricks", what);
    TMF_REQUIRE(cout <= 1024, TMF_E_SHAPE, "%s: more than 32 groups of output channels", what);
    auto k = conv3d_wino_p_kernel<MODE, GEOM>;
    int rc;
    if ((rc = tmf_allow_lds(k, PLds<GEOM>::BYTES, what))) return rc;
    const int ncu = wino_cu_count();
    const long per_launch = (long)ncu * GM::TAB;            // a workgroup's item table holds TAB entries
    for (long i0 = 0; i0 < nitems; i0 += per_launch) {
        const long n = nitems - i0 < per_launch ? nitems - i0 : per_launch;
        // as many workgroups as the launch needs for its number of ROUNDS (864 items on 256 CUs are 4 rounds: 216 workgroups
        // of exactly 4 items take as long as 256 of 3 or 4 and leave 40 CUs to the other encoder's stream)
        int grid = (int)(n < ncu ? n : ncu);
        static const bool even = !(getenv("TMF_WINO_EVEN") && atoi(getenv("TMF_WINO_EVEN")) == 0);
        if (even && n > ncu) {
            const long rounds = (n + ncu - 1) / ncu;
            grid = (int)((n + rounds - 1) / rounds);
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(PN), PLds<GEOM>::BYTES, stream, x, u, z, stat_partial, B, D, H, W, cin, cout,
                           tilesD, tilesH, tilesW, (int)nbricks, (int)i0, (int)n, scale, shift, slope, pool, ncu, i0 > 0 ? 1 : 0);
        if ((rc = tmf_launch_result(what))) return rc;
    }
    return TMF_OK;
}
template <int MODE>
int launch_wino_p(const char* what, const float* x, const float* u, float* z, float* stat_partial, int B, int D, int H, int W,
                  int cin, int cout, const float* scale, const float* shift, float slope, int pool, hipStream_t stream) {
    // train forward / data gradient on the bf16 matrix pipe through exact 3-way splits (conv3d_winox.hip) where it takes the launch
    const int geom = wino_fwd_geom(B, D, H, W, cin, cout);
    if (tmf_winox_takes(B, D, H, W, cin, cout, geom))
        return tmf_winox_launch(what, x, reinterpret_cast<const unsigned short*>(u + (size_t)64 * cin * cout), z, stat_partial, B, D, H, W,
                                cin, cout, wino_cu_count(), stream, MODE == 2 ? scale : nullptr, MODE == 2 ? shift : nullptr, slope, pool);
    if (geom && (long)4 * D * H * W * (cin > cout ? cin : cout) < (1L << 29)) return launch_wino_p_g<MODE, 1>(what, x, u, z, stat_partial, B, D, H, W, cin, cout, scale, shift, slope, pool, stream);
    return launch_wino_p_g<MODE, 0>(what, x, u, z, stat_partial, B, D, H, W, cin, cout, scale, shift, slope, pool, stream);
}

}  // namespace

int tmf_wino_p_set(int v) { g_wino_p = v ? 1 : 0; return TMF_OK; }
extern "C" int tmf_wino_p_mode(void) { return wino_p_mode(); }

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
    if (const int o = tmf_algo_override()) return (o >> 9) & 3;
    if (g_conv_wino < 0) {
        const char* e = getenv("TMF_CONV_WINO");
        const int v = e ? atoi(e) : 3;
        g_conv_wino = (v >= 0 && v <= 2) ? v : 3;
    }
    return g_conv_wino;
}
int tmf_conv_wino_set(int v) { g_conv_wino = v; return TMF_OK; }

extern "C" int tmf_conv3d_wino_ok(int cin, int cout) { return cin > 0 && cout > 0 && cin % 8 == 0 && cout % 32 == 0; }

// rows of the statistic partials of tmf_conv3d_fwd_wino: one per workgroup of the persistent kernel (= compute units of the
// device: every row is written, those of workgroups a small launch does not have with zeros), one per brick with the
// two-waves-per-SIMD kernel
extern "C" int tmf_conv3d_wino_stat_blocks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (wino_p_mode()) return wino_cu_count();
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);
}
// bricks (items per group of 32 output channels) of the persistent forward kernel for a volume
// ... of a launch with these channel counts (the split kernel's eligibility enters the choice of the geometry: wino_fwd_geom)
extern "C" int tmf_conv3d_wino_bricks2(int B, int D, int H, int W, int cin, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (wino_p_mode()) {
        const int g = wino_fwd_geom(B, D, H, W, cin, cout);
        if (tmf_winox_takes(B, D, H, W, cin, cout, g)) return (int)(B * tmf_winox_items(D, H, W, nullptr));
        return (int)wino_p_bricks(g, B, D, H, W);
    }
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);
}
extern "C" int tmf_conv3d_wino_bricks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (wino_p_mode()) return (int)wino_p_bricks(wino_p_geom(B, D, H, W), B, D, H, W);
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);
}

// the kernel instance tmf_conv3d_fwd_wino (stats: with statistic partials) / tmf_conv3d_wgrad_wino launch for a volume, as a kernel
// trace prints it (bench.py's roofline rows and tools/pmc_traffic.py's keys)
extern "C" const char* tmf_conv3d_wino_kernel_name2(int B, int D, int H, int W, int cin, int cout, int stats) {
    if (wino_p_mode() && B > 0 && D > 0 && H > 0 && W > 0 && tmf_winox_takes(B, D, H, W, cin, cout, wino_fwd_geom(B, D, H, W, cin, cout)))
        return stats ? "conv3d_winox_kernel<1>" : "conv3d_winox_kernel<0>";
    return tmf_conv3d_wino_kernel_name(B, D, H, W, stats);
}
extern "C" const char* tmf_conv3d_wino_kernel_name(int B, int D, int H, int W, int stats) {
    if (!wino_p_mode()) return stats ? "conv3d_wino_kernel<1>" : "conv3d_wino_kernel<0>";
    const int g = (B > 0 && D > 0 && H > 0 && W > 0) ? wino_p_geom(B, D, H, W) : 0;
    return stats ? (g ? "conv3d_wino_p_kernel<1, 1>" : "conv3d_wino_p_kernel<1, 0>") : (g ? "conv3d_wino_p_kernel<0, 1>" : "conv3d_wino_p_kernel<0, 0>");
}
// tiles (of 2x2x2 output voxels, padded) the weight-gradient launch multiplies per (input, output) channel pair: 16 per stage
extern "C" long tmf_conv3d_wgrad_wino_tiles(int B, int D, int H, int W, int cin, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || !tmf_conv3d_wgrad_wino_ok(cin, cout)) return 0;
    if (!wino_p_mode()) return 16L * B * tmf_cdiv(D, 4) * tmf_cdiv(H, 4) * tmf_cdiv(W, 8);
    return 16L * plan_wino_wgrad(B, D, H, W, cin, cout).nbricks;
}
extern "C" const char* tmf_conv3d_wgrad_wino_kernel_name(int B, int D, int H, int W, int cin, int cout) {
    if (!wino_p_mode()) return "conv3d_wino_wgrad_kernel";
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return "conv3d_wino_wgrad_p_kernel<0>";
    return plan_wino_wgrad(B, D, H, W, cin, cout).geom ? "conv3d_wino_wgrad_p_kernel<1>" : "conv3d_wino_wgrad_p_kernel<0>";
}

// the fp32 tensor U[64][cin][cout] and, behind it, its exact 3-way bf16 split (conv3d_winox.hip; written when the K dimension is a
// multiple of 32, reserved always: which kernel reads the buffer is decided per launch)
extern "C" size_t tmf_conv3d_wino_weight_bytes(int cin, int cout) { return (size_t)64 * cin * cout * (4 + 6); }

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

extern "C" int tmf_pack_conv_weights_wino_multi(int n, const float* const* w, float* const* u_fwd, float* const* u_dgrad,
                                                const int* cout, const int* cin, void* stream) {
    TMF_REQUIRE(n > 0 && n <= 8, TMF_E_ARG, "tmf_pack_conv_weights_wino_multi: %d layers (1 .. 8)", n);
    TMF_REQUIRE_PTR(w); TMF_REQUIRE_PTR(u_fwd); TMF_REQUIRE_PTR(u_dgrad); TMF_REQUIRE_PTR(cout); TMF_REQUIRE_PTR(cin);
    WinoPackMulti a;
    long most = 0;
    for (int l = 0; l < n; ++l) {
        TMF_REQUIRE(w[l] != nullptr && (u_fwd[l] != nullptr || u_dgrad[l] != nullptr), TMF_E_NULL,
                    "tmf_pack_conv_weights_wino_multi: layer %d: weight or both outputs NULL", l);
        TMF_REQUIRE(cout[l] > 0 && cin[l] > 0 && (u_fwd[l] == nullptr || tmf_conv3d_wino_ok(cin[l], cout[l])) &&
                    (u_dgrad[l] == nullptr || tmf_conv3d_wino_ok(cout[l], cin[l])), TMF_E_SHAPE,
                    "tmf_pack_conv_weights_wino_multi: layer %d: cin=%d cout=%d", l, cin[l], cout[l]);
        a.w[l] = w[l]; a.fwd[l] = u_fwd[l]; a.dgrad[l] = u_dgrad[l]; a.cout[l] = cout[l]; a.cin[l] = cin[l];
        if ((long)cout[l] * cin[l] > most) most = (long)cout[l] * cin[l];
    }
    for (int l = n; l < 8; ++l) { a.w[l] = nullptr; a.fwd[l] = nullptr; a.dgrad[l] = nullptr; a.cout[l] = 0; a.cin[l] = 0; }
    hipLaunchKernelGGL(wino_pack_multi_kernel, dim3((unsigned)tmf_cdiv(most, 256L), n), dim3(256), 0, (hipStream_t)stream, a);
    return tmf_launch_result("tmf_pack_conv_weights_wino_multi");
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
    if (wino_p_mode()) {
        if (stat_partial != nullptr)
            return launch_wino_p<1>("tmf_conv3d_fwd_wino", x, u, z, stat_partial, B, D, H, W, cin, cout, nullptr, nullptr, 0.f, 0, (hipStream_t)stream);
        return launch_wino_p<0>("tmf_conv3d_fwd_wino", x, u, z, nullptr, B, D, H, W, cin, cout, nullptr, nullptr, 0.f, 0, (hipStream_t)stream);
    }
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
    if (wino_p_mode())                                      // slabs of dw itself [27][cin][cout] (+ the first-stage sums of a two-stage reduction)
        return (size_t)(p.nsplit + tmf_reduce_groups(p.nsplit)) * 27 * cin * cout * 4;
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
    if (wino_p_mode()) {
        auto k = p.geom ? conv3d_wino_wgrad_p_kernel<1> : conv3d_wino_wgrad_p_kernel<0>;
        const size_t lds = p.geom ? WGeomP<1>::LDS_BYTES : WGeomP<0>::LDS_BYTES;
        if ((rc = tmf_allow_lds(k, lds, "tmf_conv3d_wgrad_wino"))) return rc;
        hipLaunchKernelGGL(k, dim3(p.nsplit, p.nblk), dim3(WPN), lds, s, x, dz, partial, B, D, H, W, cin, cout,
                           p.tilesD, p.tilesH, p.tilesW, p.nbricks, p.per, p.ncob);
        if ((rc = tmf_launch_result("tmf_conv3d_wgrad_wino"))) return rc;
        // the slabs are dw's own [27][cin][cout]: the (fp64, fixed-order) reduction writes the result in either layout
        const long nw = (long)27 * cin * cout;
        const bool ref = dw_layout == TMF_DW_REFERENCE;
        return tmf_reduce_slabs(partial, p.nsplit, nw, partial + (size_t)p.nsplit * nw, dw, s, "tmf_conv3d_wgrad_wino(reduce)",
                                ref ? cin : 0, ref ? cout : 0);
    }
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
    if (wino_p_mode())
        return launch_wino_p<2>("tmf_conv3d_fwd_wino_affine", x, u, y, nullptr, B, D, H, W, cin, cout, scale, shift, slope, pool, (hipStream_t)stream);
    const int tilesD = tmf_cdiv(D, TD), tilesH = tmf_cdiv(H, TH), tilesW = tmf_cdiv(W, TW);
    const int ntiles = B * tilesD * tilesH * tilesW;
    auto k = conv3d_wino_kernel<2>;
    int rc;
    if ((rc = tmf_allow_lds(k, LDS_BYTES, "tmf_conv3d_fwd_wino_affine"))) return rc;
    hipLaunchKernelGGL(k, dim3(ntiles, cout / 32), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, x, u, y, (float*)nullptr, D, H, W, cin, cout,
                       tilesD, tilesH, tilesW, ntiles, scale, shift, slope, pool);
    return tmf_launch_result("tmf_conv3d_fwd_wino_affine");
}
