// conv3d_bf16.hip — 3x3x3 convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 2.5 PF dense) with
// fp32 tensors in HBM and fp32 accumulation: the "bf16 with MFMA 3D conv" mode of BASELINE.json configs[2].
//
// Same brick / halo / weight-ring structure as conv3d_mfma.hip; what changes is the operand path:
//   * activations are read as fp32, rounded to bf16 (RNE) ONCE while the halo is staged in LDS (80-B rows of
//     32 channels), weights arrive pre-packed as bf16 [tap][cout][cin] (cin contiguous);
//   * one ds_read_b128 per operand fragment: the 32x32x16 MFMA takes 8 consecutive k (= input channels) per
//     lane, which is exactly the channels-last row;
//   * 8 waves of 32 voxels x (32|64) channels, <= 128 registers: two workgroups (16 waves) per CU hide the staging;
//   * accumulators, BatchNorm statistic partials and the stored output stay fp32.
// At 1/16 of the fp32 MFMA time the kernel is bound by halo staging and the fp32 output stream, not by the
// matrix pipe.
//
// Forward and data-gradient (flipped / transposed weights) of networks.py:28,31,37,40,46.
#include <type_traits>
#include "tmf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// round-to-nearest-even, a -> low half: one v_cvt_pk_bf16_f32 on gfx950 (the integer emulation cost 8 VALU
// instructions per pair and made the staging of the bf16 kernels VALU-bound, PMC)
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}

constexpr int TD = 4, TH = 8, TW = 8;                 // brick: 256 voxels = 8 waves x one 32-voxel M-tile
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int NHALO = HD * HH * HW;
constexpr int CINC = 32;                              // input channels per chunk
constexpr int RP = 40;                                // LDS row pitch in bf16 (80 B: 16-B aligned, odd multiple of 16 B)
constexpr int TPS = 3;                                // taps per weight stage
constexpr int NSTAGES = 9;
constexpr int NTHR = 512;                             // 8 waves, one 32-voxel M-tile each

template <int NT>                                     // NT output-channel tiles of 32 per wave / workgroup
struct BfCfg {
    static constexpr int NB = 32 * NT;
    static constexpr int WSTAGE = TPS * NB * RP;      // bf16 elements per weight stage
    // NT = 2: the cross-wave statistics scratch aliases the halo (one extra barrier) so that two workgroups still fit a CU
    static constexpr bool RED_ALIAS = NT == 2;
    static constexpr size_t LDS_BYTES = (size_t)(NHALO * RP + 2 * WSTAGE) * 2 + (RED_ALIAS ? 0 : 8 * NB * 2 * 4);
};

// IN16 / OUT16: the activation tensors themselves are bf16 (configs[2] "bf16 storage"): the halo is then a plain
// 8-byte copy per 4 channels (no conversion through the VALU) and z is rounded once on the way out — the BatchNorm
// statistics still come from the fp32 accumulators.
template <int NT, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(NTHR, 4) void conv3d_fwd_bf16_kernel(
    const void* __restrict__ x_, const u16* __restrict__ w, void* __restrict__ z_,
    float* __restrict__ stat_partial, int D, int H, int W, int Cin, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int NB = BfCfg<NT>::NB, WSTAGE = BfCfg<NT>::WSTAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* halo = reinterpret_cast<u16*>(smem_raw);
    u16* Ws = halo + NHALO * RP;
    float* red = BfCfg<NT>::RED_ALIAS ? reinterpret_cast<float*>(smem_raw) : reinterpret_cast<float*>(Ws + 2 * WSTAGE);

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
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int n0 = blockIdx.y * NB;

    // LDS element offsets (bf16 units) of this lane's fragments
    int a_lane;
    {
        // wave = brick row h, lane = (d, w) with d in the two LOW bits: with the 80-B row pitch this is the mapping
        // for which every 16-lane service group of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}) lands on
        // 16 distinct 16-B slots, for every tap offset (the 4 x 8 (h, w) mapping is 2-3-way conflicted)
        const int pd = l31 & 3, pw = l31 >> 2, ph = wave;
        a_lane = ((pd * HH + ph) * HW + pw) * RP + hsel * 8;
    }
    const int b_lane = l31 * RP + hsel * 8;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const float* xb = reinterpret_cast<const float*>(x_) + (IN16 ? 0 : (size_t)b * D * H * W * Cin);
    const u16* xb16 = reinterpret_cast<const u16*>(x_) + (IN16 ? (size_t)b * D * H * W * Cin : 0);

    // brick-invariant halo addressing, computed once (the div/mod chain per piece was a third of the VALU work)
    constexpr int HV = (NHALO * 8 + NTHR - 1) / NTHR;
    constexpr int HB = (HV + 1) / 2;
    int hoff[HV];                                             // element offset of the position in the sample, -1 = zero fill
#pragma unroll
    for (int q = 0; q < HV; ++q) {
        const int hp = (tid + q * NTHR) >> 3;
        const int hw = hp % HW, hh = (hp / HW) % HH, hd = hp / (HW * HH);
        const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
        const bool ok = hp < NHALO && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
        hoff[q] = ok ? ((gd * H + gh) * W + gw) * Cin : -1;
    }

    for (int c0 = 0; c0 < Cin; c0 += CINC) {
        if (c0 > 0) __syncthreads();
        // ---- halo: fp32 from HBM, rounded to bf16 on the way into LDS.  The loads of a batch are all issued
        //      before its LDS writes; two batches keep the kernel under 128 registers (two workgroups per CU) ----
        auto stage_halo = [&](const int q0, const int q1) {
            f32x4 hreg[IN16 ? 1 : HB];
            u32x2 hreg16[IN16 ? (NT == 1 ? HV : HB) : 1];
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int c = c0 + ((tid + q * NTHR) & 7) * 4;
                const bool ok = hoff[q] >= 0 && c < Cin;
                if (IN16) {
                    u32x2 v = {0u, 0u};
                    if (ok) v = *reinterpret_cast<const u32x2*>(xb16 + hoff[q] + c);
                    hreg16[q - q0] = v;
                } else {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (ok) v = *reinterpret_cast<const f32x4*>(xb + hoff[q] + c);
                    hreg[q - q0] = v;
                }
            }
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int e = tid + q * NTHR;
                if (e < NHALO * 8) {
                    u32x2 pk;
                    if (IN16) pk = hreg16[q - q0];
                    else {
                        const f32x4 v = hreg[q - q0];
                        pk = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                    }
                    *reinterpret_cast<u32x2*>(halo + (e >> 3) * RP + (e & 7) * 4) = pk;
                }
            }
        };
        // ---- weight stage loader: [tap][co][ci] bf16, one 16-B piece = 8 input channels of one output channel ----
        constexpr int WV = (TPS * NB * 4 + NTHR - 1) / NTHR;         // 4 pieces per (tap, co) row of 32 channels
        // A stage is only 6 NT MFMAs (~200 NT cycles): with the next stage's weights requested one stage ahead, every
        // stage waited a full L2 round trip (9 per chunk: the kernel ran at 5 % matrix utilisation per wave).  The
        // loads now run PW stages ahead in registers (in-flight global loads survive the barriers: __syncthreads only
        // waits for LDS traffic), and the stage loop is unrolled so the register slots are static.
        constexpr int PW = NT == 1 ? 6 : 1;           // (two tiles: no registers to spare, classic one-stage prefetch)
        u32x4 wreg[PW][WV];
        auto load_w = [&](int st, int slot) {
#pragma unroll
            for (int q = 0; q < WV; ++q) {
                const int e = tid + q * NTHR;
                const int row = e >> 2, piece = e & 3;             // row = tap_in_stage * NB + co
                const int tap = st * TPS + row / NB, co = n0 + row % NB, ci = c0 + piece * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (e < TPS * NB * 4 && co < Cout && ci < Cin)
                    v = *reinterpret_cast<const u32x4*>(w + ((size_t)tap * Cout + co) * Cin + ci);
                wreg[slot][q] = v;
            }
        };
        auto store_w = [&](int buf, int slot) {
#pragma unroll
            for (int q = 0; q < WV; ++q) {
                const int e = tid + q * NTHR;
                if (e < TPS * NB * 4)
                    *reinterpret_cast<u32x4*>(Ws + buf * WSTAGE + (e >> 2) * RP + (e & 3) * 8) = wreg[slot][q];
            }
        };
        load_w(0, 0);
        if constexpr (IN16 && NT == 1) {          // bf16 tensors: the whole halo is 20 registers, one latency instead of two
#pragma unroll
            for (int st = 1; st < PW; ++st) load_w(st, st);
            stage_halo(0, HV);
        } else {
            stage_halo(0, HB);
#pragma unroll
            for (int st = 1; st < PW; ++st) load_w(st, st);
            stage_halo(HB, HV);
        }
        constexpr int UNR = NT == 1 ? NSTAGES : 1;        // static register slots need the unrolled form
#pragma unroll UNR
        for (int st = 0; st < NSTAGES; ++st) {
            const int buf = st & 1;
            store_w(buf, st % PW);
            __syncthreads();
            if (st + PW < NSTAGES) load_w(st + PW, st % PW);
            const int stage_off = ((st / 3) * HH + (st % 3)) * HW * RP;
            const u16* ws = Ws + buf * WSTAGE + b_lane;
#pragma unroll
            for (int tp = 0; tp < TPS; ++tp) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {                      // two k-blocks of 16 channels
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(halo + a_lane + stage_off + tp * RP + s * 16);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const bf16x8 bb = *reinterpret_cast<const bf16x8*>(ws + (tp * NB + j * 32) * RP + s * 16);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: fp32 NDHWC store + BatchNorm statistic partials (same as the fp32 kernel) ----
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    float* zb = reinterpret_cast<float*>(z_) + (OUT16 ? 0 : (size_t)b * D * H * W * Cout);
    u16* zb16 = reinterpret_cast<u16*>(z_) + (OUT16 ? (size_t)b * D * H * W * Cout : 0);
    auto epilogue = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        if constexpr (OUT16) {
            // bf16 tensor: rows are taken in pairs (r, r + 1 = planes d, d + 1) and neighbouring channel lanes swap one
            // value, so that the even lane stores the channel PAIR of row r and the odd lane that of row r + 1 as one
            // dword each (sub-dword stores run at a fraction of the rate).  Cout is even (checked on the host).
            const int odd = lane & 1;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int pw = 2 * (r >> 2) + hsel, ph = wave;
                const int gh = h0 + ph, gw = w0 + pw;
                const int gd = d0 + (r & 3) + odd;                        // the row this lane stores
                const bool pv = FULL || (gd < D && gh < H && gw < W);
                const int off = ((gd * H + gh) * W + gw) * Cout;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = n0 + j * 32 + l31;
                    const float a = acc[j][r], b = acc[j][r + 1];
                    const float other = __shfl_xor(odd ? a : b, 1);       // even lane gets the odd lane's a, odd gets b
                    const unsigned int pk = odd ? pack_bf16(other, b) : pack_bf16(a, other);
                    if (FULL || (pv && co < Cout))
                        *reinterpret_cast<unsigned int*>(zb16 + off + (co & ~1)) = pk;
                    // statistics: each lane's own channel, both rows (validity of each row separately)
                    const bool va = FULL || (d0 + (r & 3) < D && gh < H && gw < W && co < Cout);
                    const bool vb = FULL || (d0 + (r & 3) + 1 < D && gh < H && gw < W && co < Cout);
                    if (va) { s1[j] += a; s2[j] += a * a; }
                    if (vb) { s1[j] += b; s2[j] += b * b; }
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pd = r & 3, pw = 2 * (r >> 2) + hsel, ph = wave;      // fragment row i -> (d = i & 3, w = i >> 2)
            const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
            const bool pv = FULL || (gd < D && gh < H && gw < W);
            const int off = ((gd * H + gh) * W + gw) * Cout;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = n0 + j * 32 + l31;
                if (FULL || (pv && co < Cout)) {
                    const float v = acc[j][r];
                    zb[off + co] = v;
                    s1[j] += v;
                    s2[j] += v * v;
                }
            }
        }
        }
    };
    if (d0 + TD <= D && h0 + TH <= H && w0 + TW <= W && n0 + NB <= Cout) epilogue(std::true_type{});
    else epilogue(std::false_type{});

    if (stat_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
        }
        if (BfCfg<NT>::RED_ALIAS) __syncthreads();            // every wave is done reading the halo
        if (hsel == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                red[(wave * NB + j * 32 + l31) * 2 + 0] = s1[j];
                red[(wave * NB + j * 32 + l31) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < NB) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) { a1 += red[(m * NB + tid) * 2]; a2 += red[(m * NB + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < Cout) {
                stat_partial[((size_t)tile * 2 + 0) * Cout + co] = a1;
                stat_partial[((size_t)tile * 2 + 1) * Cout + co] = a2;
            }
        }
    }
}


namespace dma {
// LDS-DMA of 16 bytes per lane: LDS byte = lds_wave_base (wave-uniform LDS address) + 16 * lane.  Written as inline
// assembly on purpose: with __builtin_amdgcn_global_load_lds the compiler cannot tell that the copy fills the OTHER
// buffer and waits vmcnt(0) before the first fragment read of every brick, which serialises copy and multiply (the
// first build did: 61 us of 240 exposed).  The price is that the compiler does not see the copies at all: the kernel
// waits for them itself (dma_wait) before the barrier that publishes the buffer.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "memory");
}
// The same through a buffer resource: LDS byte = lds_wave_base + 16 * lane <- resource base + voff (per lane) + soff (scalar).
// A lane whose voff + soff is not below the resource's num_records delivers ZEROS to its LDS bytes (gfx950: the scalar
// offset is part of the range check, tools/microbench/blds_probe.hip) — zero fill costs no pointer select, no compare.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void blds16(int voff, i32x4 rsrc, int soff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
}  // namespace dma
__device__ __attribute__((aligned(16))) const unsigned int tmf_zero16[4] = {0u, 0u, 0u, 0u};

// ------------------------------------------------------------------------------------------------------------
// Large-layer variant: 8x8x8 bricks, a 2 x NT REGISTER TILE per wave.
//
// The kernel above is bound by its LDS traffic, not by the matrix pipe (PMC: pipe 24-30 % busy, LDS > 40 %): a wave
// owns ONE 32-voxel M-tile, so every MFMA needs 1.5-2 ds_read_b128 (one A and one B fragment per 32x32x16 product;
// at the bf16 rate the CU's LDS port delivers exactly two per MFMA slot), and a 256-voxel brick re-stages the full
// 27-tap weight set (135 KB per 32-channel chunk) for only 864 MFMAs.  Here
//   * a wave owns TWO M-tiles (planes 0-3 and 4-7 of its brick row) x NT N-tiles: 2 + NT reads feed 2 NT MFMAs
//     (1.0 read per MFMA at NT = 2), and a 512-voxel brick halves the weight staging per MFMA;
//   * input channels go in chunks of 16 (one MFMA k-block): halo rows are 48 B (32 B + 16 B pad: the same lane ->
//     voxel map stays conflict-free — 12 pd + 3 pw (mod 16) is a bijection on each 16-lane service group), the
//     10x10x10 halo is 48 KB and the 3-tap weight ring 2 x 9 KB: 66 KB per workgroup, two workgroups per CU;
//   * everything else (zero-filled halo, weight stages prefetched in registers across the barriers, fp32 accumulation,
//     statistics from the accumulators, dword stores of bf16 outputs) is the scheme of the kernel above.
// Used when the launch has enough 512-voxel bricks to fill the chip (host: use_v2()); results are identical to
// the small-brick kernel up to fp32 summation order (k runs 16-channel chunk -> tap here, 32-channel chunk -> tap -> half
// there).
// ------------------------------------------------------------------------------------------------------------
// -DTMF_TRACE=<workgroup>: waves 0 and 4 of that workgroup of the LDS-DMA forward kernel log shader-clock stamps at their
// phase boundaries (tools/v2_trace.py reads them through tmf_debug_trace_read); never in the shipped library.
#ifdef TMF_TRACE
__device__ unsigned long long tmf_trace_buf[2][256];
#define TR(id) do { if (tr_on) { if (lane == 0) tmf_trace_buf[tr_slot][tr_n] = ((unsigned long long)(id) << 48) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFull); ++tr_n; } } while (0)
#else
#define TR(id) do { } while (0)
#endif
namespace v2 {
constexpr int TD = 8, TH = 8, TW = 8;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int NHALO = HD * HH * HW;                   // 1000
constexpr int CINC = 16;
constexpr int RP = 24;                                // bf16 per LDS row: 48 B
constexpr int TPS = 3, NSTAGES = 9;
constexpr int NTHR = 512;
constexpr int MT2 = 4 * HH * HW * RP;                 // LDS offset of the wave's second M-tile (planes 4-7)
template <int NT>
struct Cfg {
    static constexpr int NB = 32 * NT;
    static constexpr int WSTAGE = TPS * NB * RP;      // bf16 elements per weight stage
    static constexpr size_t LDS_BYTES = (size_t)(NHALO * RP + 2 * WSTAGE) * 2;     // statistics scratch aliases the halo
    // DMA form: unpadded 32-byte rows, two halo buffers, two weight stages
    static constexpr int TPSD = 9, NSTD = 3;           // taps per weight stage, stages per chunk
    static constexpr int HBYTES = NHALO * 32, WBYTES = TPSD * NB * 32;
    static constexpr size_t LDS_BYTES_DMA = (size_t)HBYTES + 2 * WBYTES;           // 68 864 B at NT = 2: two workgroups per CU
};
}  // namespace v2

// DMA = true (bf16 input tensors only): the operands are byte copies of global memory, so both are filled by LDS-DMA
// instead of through registers, and the halo of input-channel chunk c + 1 streams into a second buffer while chunk c is
// multiplied (with register staging the kernel spent a third of a launch waiting for the halo and weight loads of the
// chunk it was about to multiply: tools/bf16_ablate.py).  An LDS-DMA image is lane-linear, so the 48-byte padded rows
// become 32-byte rows and the bank spread comes from a swizzle on the SOURCE side instead: the two 16-byte halves of a
// halo row are swapped where bit 1 of the row's plane index is set, of a weight row where bit 3 of its output channel
// is set (both maps verified conflict-free for ds_read_b128's 16-lane groups: tests/test_swizzle_maps.py).  vmcnt
// retires in order per wave, so the two streams are issued by different waves: waves 0-3 copy the weight stage one
// stage ahead (L2-warm, lands within a stage) and wait for it at every stage barrier; waves 4-7 copy the next chunk's
// halo and wait once per chunk.
template <int NT, bool IN16, bool OUT16, bool DMA = false>
__global__ __launch_bounds__(v2::NTHR, 4) void conv3d_fwd_bf16_v2_kernel(
    const void* __restrict__ x_, const u16* __restrict__ w, void* __restrict__ z_,
    float* __restrict__ stat_partial, int D, int H, int W, int Cin, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles, int dbg_arg) {
    // dbg (tmf_set_option("debug", bits), timing ablations only — results are garbage when set): 1 = no weight loads,
    // 2 = no halo loads, 4 = no stage barriers, 8 = no MFMAs, 16 = no output stores.  Live only in a -DTMF_ABLATE
    // build (tools/bf16_ablate.py makes one); the shipped kernel folds every test away.
#ifdef TMF_ABLATE
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    constexpr int TD = v2::TD, TH = v2::TH, TW = v2::TW, HH = v2::HH, HW = v2::HW, NHALO = v2::NHALO, CINC = v2::CINC,
                  RP = v2::RP, TPS = v2::TPS, NSTAGES = v2::NSTAGES, NTHR = v2::NTHR, MT2 = v2::MT2;
    constexpr int NB = v2::Cfg<NT>::NB, WSTAGE = v2::Cfg<NT>::WSTAGE;
    // PERM (two N-tiles, bf16 outputs): weight row l of tile j holds output channel n0 + 2 l + j, so a lane's two
    // accumulator tiles are the two halves of one stored dword (a channel pair): no exchange between lanes in the epilogue
    // and 128 contiguous bytes per voxel and half-wave.  Only the SOURCE channel of a staged weight row changes.
    constexpr bool PERM = NT == 2 && OUT16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* halo = reinterpret_cast<u16*>(smem_raw);
    u16* Ws = halo + NHALO * RP;
    float* red = reinterpret_cast<float*>(smem_raw);

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
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int n0 = blockIdx.y * NB;

    // wave = brick row h; lane = (d & 3 in the two LOW bits, w); M-tile m = planes 4m .. 4m + 3
    const int a_lane = (((l31 & 3) * HH + wave) * HW + (l31 >> 2)) * RP + hsel * 8;
    const int b_lane = l31 * RP + hsel * 8;

    f32x16 acc[2][NT];
    if (!DMA || (dbg & 8)) {                                  // (the DMA form's first products start from the literal 0)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    }

    const float* xb = reinterpret_cast<const float*>(x_) + (IN16 ? 0 : (size_t)b * D * H * W * Cin);
    const u16* xb16 = reinterpret_cast<const u16*>(x_) + (IN16 ? (size_t)b * D * H * W * Cin : 0);

#ifdef TMF_TRACE
    const bool tr_on = DMA && blockIdx.x == TMF_TRACE && (wave == 0 || wave == 4);
    const int tr_slot = wave >> 2;
    int tr_n = 0;
#endif
    TR(1);
    if constexpr (DMA) {
        static_assert(IN16, "the DMA form copies bf16 tensors");
        using namespace dma;
        constexpr int HBYTES = v2::Cfg<NT>::HBYTES, WBYTES = v2::Cfg<NT>::WBYTES;
        constexpr int MT2B = 4 * HH * HW * 32;                 // bytes to the wave's second M-tile (planes 4-7)
        const unsigned lds0 = lds_addr(smem_raw);
        const bool wrole = wave < 4;
        const int rt = tid & 255;                              // thread index inside its role
        // fragment addresses: a halo row keeps channels 8 (hd >> 1 & 1) .. first; plane hd = (l31 & 3) + kd (+ 4)
        const int pd = l31 & 3;
        const int a_vox = ((pd * HH + wave) * HW + (l31 >> 2)) * 32;
        int a_kd[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) a_kd[kd] = a_vox + ((hsel ^ (((pd + kd) >> 1) & 1)) * 16);
        const int b_row = l31 * 32 + ((hsel ^ ((l31 >> 3) & 1)) * 16);
        // copy descriptors.  Halo: piece e = rt + 256 i is LDS bytes 16 e .. of the buffer: row e / 2, half e % 2, which
        // holds channel group (e % 2) ^ (hd >> 1 & 1).  Weights: piece e is row e / 2 = tap * NB + co, channel group
        // (e % 2) ^ (co >> 3 & 1).
        constexpr int TPSD = v2::Cfg<NT>::TPSD, NSTD = v2::Cfg<NT>::NSTD;
        // Copy descriptors.  The kernel is ISSUE-bound (PMC: 6.7 vector instructions per MFMA; a 32-cycle MFMA slot hides
        // about five other issues), so the pieces are laid out to need almost no arithmetic per copy:
        //  * halo: one instruction per halo PLANE - thread rt < 200 of the role copies piece rt of the plane's 10 rows x
        //    10 voxels x 2 halves (LDS bytes 16 (200 hd + rt): lane-linear).  Row / column / half of a thread never
        //    change: in-plane offset and validity are computed once; the plane's validity and base are scalars, the
        //    swizzle bit (hd >> 1 & 1) is a compile-time constant of the unrolled plane loop.
        //  * weights: piece e = rt + 256 i is row e / 2 = tap * NB + co, half e % 2: column, half and tap-in-instruction
        //    of a thread never change either; a stage / chunk / instruction step is a scalar offset.
        constexpr int PPL = HH * HW * 2;                       // 200 pieces per halo plane
        constexpr int WQ = (TPSD * NB * 2 + 255) / 256;        // 5 | 3
        // (one register set for both roles: a wave is a halo copier or a weight copier for the whole kernel)
        // Byte offsets inside the sample / the weight tensor; a lane that must deliver zeros (outside the volume, beyond
        // Cout, an idle lane of the last plane piece) carries OOBV, a plane outside the volume an out-of-range SCALAR offset.
        constexpr int OOBV = 0x7FFFFFF0;
        int r_vo, r_vox = OOBV, r_q8, r_tp = 0;                 // r_vox: the halo row's OTHER 16-byte half (swapped planes)
        if (wrole) {
            const int col = (rt >> 1) % NB;                     // LDS row inside the tap
            const int cs = PERM ? 2 * (col & 31) + (col >> 5) : col;   // the output channel it holds
            r_tp = (rt >> 1) / NB;
            r_q8 = 8 * ((rt & 1) ^ ((col >> 3) & 1));
            r_vo = n0 + cs < Cout ? ((r_tp * Cout + n0 + cs) * Cin + r_q8) * 2 : OOBV;
        } else {
            const int hh = rt / (HW * 2), hw = (rt % (HW * 2)) >> 1;
            const int gh = h0 + hh - 1, gw = w0 + hw - 1;
            const bool ok = rt < PPL && gh >= 0 && gh < H && gw >= 0 && gw < W;
            r_q8 = 8 * (rt & 1);
            r_vo = ok ? ((gh * W + gw) * Cin + r_q8) * 2 : OOBV;           // channel group q; the swapped planes take the other
            r_vox = ok ? ((gh * W + gw) * Cin + (r_q8 ^ 8)) * 2 : OOBV;    // (a row starts at an odd multiple of 8 when Cin % 16 == 8)
        }
        const i32x4 xr = make_rsrc(xb16, (unsigned)(D * H * W * Cin * 2));
        const i32x4 wr = make_rsrc(w, (unsigned)(27 * Cout * Cin * 2));
        const bool ragged_c = (Cin & (CINC - 1)) != 0;         // the last chunk is partial: its missing channel groups read zeros
        const int plane_b = H * W * Cin * 2;
        constexpr int TPI = 256 / (2 * NB);                    // taps per copy instruction: 2 | 4
        auto issue_h = [&](int c0) {                           // waves 4-7
            if (dbg & 2) return;
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (wave - 4) * 1024);
            int vo = r_vo, vox = r_vox;
            if (ragged_c) { vo = c0 + r_q8 < Cin ? vo : OOBV; vox = c0 + (r_q8 ^ 8) < Cin ? vox : OOBV; }
            if (rt < PPL) {
#pragma unroll
                for (int hd = 0; hd < v2::HD; ++hd) {
                    const int gd = d0 + hd - 1;                    // scalar
                    const int so = (gd >= 0 && gd < D) ? gd * plane_b + c0 * 2 : OOBV;
                    blds16(((hd >> 1) & 1) ? vox : vo, xr, so, base + hd * (PPL * 16));   // the swizzle bit is compile-time
                }
            }
        };
        auto issue_w = [&](int c0, int st, int wbuf) {         // waves 0-3
            if (dbg & 1) return;
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + HBYTES + wbuf * WBYTES + wave * 1024);
            int vo = r_vo;
            if (ragged_c) vo = c0 + r_q8 < Cin ? vo : OOBV;
            const int so0 = (st * TPSD * Cout * Cin + c0) * 2;
#pragma unroll
            for (int i = 0; i < WQ; ++i) {
                const bool in_stage = i * TPI + TPI <= TPSD || i * TPI + r_tp < TPSD;      // the last instruction is half empty
                if (in_stage) blds16(vo, wr, so0 + i * TPI * Cout * Cin * 2, base + i * 4096);
            }
        };
        TR(2);
        if (wrole) issue_w(0, 0, 0);
        TR(3);
        // one 16-channel chunk; FIRST: the very first product of every accumulator takes the literal 0 as its addend (no
        // 32 / 64 zeroing moves per brick: every instruction of every wave is paid in SIMD issue time here)
        auto chunk = [&](int c0, int par, auto first_c) {      // par: parity of the chunk index, 3 stages flip the weight ring
            constexpr bool FIRST = decltype(first_c)::value;
            if (!FIRST) __syncthreads();                       // the halo of the previous chunk is read out
            TR(10);
            if (!wrole) { issue_h(c0); TR(11); dma_wait(); TR(12); }
#pragma unroll
            for (int st = 0; st < NSTD; ++st) {
                const int wb = (st & 1) ^ par;
                TR(20);
                if (wrole) dma_wait();                         // this wave's share of stage st has landed ...
                TR(21);
                __syncthreads();                               // ... everybody's has (and the halo), and stage st - 1 is read out
                TR(22);
                if (wrole) {
                    if (st + 1 < NSTD) issue_w(c0, st + 1, wb ^ 1);
                    else if (c0 + CINC < Cin) issue_w(c0 + CINC, 0, wb ^ 1);
                }
                TR(23);
                if (dbg & 8) continue;
                const unsigned char* wsb = smem_raw + HBYTES + wb * WBYTES + b_row;
                // software pipeline over the 9 taps: the fragments of tap tp + 1 are requested BEFORE the products of tap tp
                // are issued (left to the compiler a fragment was requested one product ahead of its use and every product
                // had its own s_waitcnt: the waves of a SIMD then sit out the LDS latency together)
                struct Frag { bf16x8 a0, a1, b[NT]; };
                auto load = [&](int tp, Frag& f) {                                                     // kh = tp / 3, kw = tp % 3
                    const unsigned char* ap = smem_raw + ((st * HH + tp / 3) * HW + tp % 3) * 32 + a_kd[st];   // kd = st
                    f.a0 = *reinterpret_cast<const bf16x8*>(ap);
                    f.a1 = *reinterpret_cast<const bf16x8*>(ap + MT2B);
#pragma unroll
                    for (int j = 0; j < NT; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(wsb + (tp * NB + j * 32) * 32);
                };
                auto mul = [&](const Frag& f, bool zero_c) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (zero_c) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0, f.b[j], zero, 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1, f.b[j], zero, 0, 0, 0);
                        } else {
                            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0, f.b[j], acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1, f.b[j], acc[1][j], 0, 0, 0);
                        }
                    }
                };
                Frag f0, f1;
                load(0, f0);
#pragma unroll
                for (int tp = 0; tp < TPSD; tp += 2) {
                    if (tp + 1 < TPSD) load(tp + 1, f1);
                    __builtin_amdgcn_sched_barrier(0);
                    mul(f0, FIRST && st == 0 && tp == 0);
                    if (tp + 2 < TPSD) load(tp + 2, f0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tp + 1 < TPSD) mul(f1, false);
                }
            }
        };
        chunk(0, 0, std::true_type{});
        int par = 1;
        for (int c0 = CINC; c0 < Cin; c0 += CINC, par ^= 1) chunk(c0, par, std::false_type{});
    } else {
    // halo pieces of 16 B: bf16 tensors 2 per position (8 channels each), fp32 tensors 4 per position (4 channels each)
    constexpr int PPP = IN16 ? 2 : 4;                         // pieces per position
    constexpr int PSH = IN16 ? 1 : 2;
    constexpr int HV = (NHALO * PPP + NTHR - 1) / NTHR;       // 4 | 8
    int hoff[HV];                                             // element offset of the position in the sample, -1 = zero fill
#pragma unroll
    for (int q = 0; q < HV; ++q) {
        const int hp = (tid + q * NTHR) >> PSH;
        const int hw = hp % HW, hh = (hp / HW) % HH, hd = hp / (HW * HH);
        const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
        const bool ok = hp < NHALO && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
        hoff[q] = ok ? ((gd * H + gh) * W + gw) * Cin : -1;
    }

    for (int c0 = 0; c0 < Cin; c0 += CINC) {
        if (c0 > 0) __syncthreads();
        // ---- weight stages: [tap][co][ci] bf16; a stage = 3 taps x NB rows x 16 channels = 2 pieces of 16 B per row ----
        constexpr int NPIECE = TPS * NB * 2;                  // 384 | 192 <= NTHR: one piece per thread
        constexpr int PW = 2;                                 // stages requested ahead, in registers
        u32x4 wreg[PW];
        auto load_w = [&](int st, int slot) {
            const int row = tid >> 1, piece = tid & 1;        // row = tap_in_stage * NB + co
            const int col = row % NB;
            const int tap = st * TPS + row / NB, co = n0 + (PERM ? 2 * (col & 31) + (col >> 5) : col), ci = c0 + piece * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (tid < NPIECE && co < Cout && ci < Cin && !(dbg & 1))
                v = *reinterpret_cast<const u32x4*>(w + ((size_t)tap * Cout + co) * Cin + ci);
            wreg[slot] = v;
        };
        auto store_w = [&](int buf, int slot) {
            if (tid < NPIECE)
                *reinterpret_cast<u32x4*>(Ws + buf * WSTAGE + (tid >> 1) * RP + (tid & 1) * 8) = wreg[slot];
        };
#pragma unroll
        for (int st = 0; st < PW; ++st) load_w(st, st);
        // ---- halo: all loads of the chunk are issued before the first LDS write ----
        if constexpr (IN16) {
            u32x4 hreg[HV];
#pragma unroll
            for (int q = 0; q < HV; ++q) {
                const int c = c0 + ((tid + q * NTHR) & 1) * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (hoff[q] >= 0 && c < Cin && !(dbg & 2)) v = *reinterpret_cast<const u32x4*>(xb16 + hoff[q] + c);
                hreg[q] = v;
            }
#pragma unroll
            for (int q = 0; q < HV; ++q) {
                const int e = tid + q * NTHR;
                if (e < NHALO * PPP) *reinterpret_cast<u32x4*>(halo + (e >> 1) * RP + (e & 1) * 8) = hreg[q];
            }
        } else {
            // fp32 tensors: batches of HB pieces keep the kernel under 128 registers
            constexpr int HB = NT == 2 ? 2 : 4;
#pragma unroll
            for (int q0 = 0; q0 < HV; q0 += HB) {
                f32x4 hreg[HB];
#pragma unroll
                for (int q = q0; q < q0 + HB; ++q) {
                    const int c = c0 + ((tid + q * NTHR) & 3) * 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (hoff[q] >= 0 && c < Cin) v = *reinterpret_cast<const f32x4*>(xb + hoff[q] + c);
                    hreg[q - q0] = v;
                }
#pragma unroll
                for (int q = q0; q < q0 + HB; ++q) {
                    const int e = tid + q * NTHR;
                    if (e < NHALO * PPP)
                        *reinterpret_cast<u32x2*>(halo + (e >> 2) * RP + (e & 3) * 4) =
                            u32x2{pack_bf16(hreg[q - q0][0], hreg[q - q0][1]), pack_bf16(hreg[q - q0][2], hreg[q - q0][3])};
                }
            }
        }
#pragma unroll
        for (int st = 0; st < NSTAGES; ++st) {
            const int buf = st & 1;
            store_w(buf, st % PW);
            if (!(dbg & 4)) __syncthreads();
            if (st + PW < NSTAGES) load_w(st + PW, st % PW);
            const u16* ap = halo + a_lane + ((st / 3) * HH + (st % 3)) * HW * RP;          // kd = st / 3, kh = st % 3
            if (dbg & 8) continue;
            const u16* ws = Ws + buf * WSTAGE + b_lane;
#pragma unroll
            for (int tp = 0; tp < TPS; ++tp) {                                            // kw
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap + tp * RP);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + tp * RP + MT2);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 bb = *reinterpret_cast<const bf16x8*>(ws + (tp * NB + j * 32) * RP);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1][j], 0, 0, 0);
                }
            }
        }
    }

    }   // register-staged form

    // ---- epilogue: NDHWC store (fp32, or bf16 as channel-pair dwords) + BatchNorm statistic partials ----
    // bf16 outputs: buffer stores — the lane part of the address is one VGPR per (M-tile, w pair), the plane travels in the
    // scalar offset, a lane outside the volume or beyond Cout carries an out-of-range offset and is dropped by the hardware.
    // Per stored dword: NT = 2 one v_cvt_pk (the lane holds both channels of the pair, PERM); NT = 1 one v_cvt_pk of the
    // lane's planes (d, d + 1), one DPP move from the neighbouring channel and one v_perm — the even lane stores plane d,
    // the odd one plane d + 1.  (Before: ≈ 15 vector instructions and a ds_bpermute per dword.)
    TR(30);
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    float* zb = reinterpret_cast<float*>(z_) + (OUT16 ? 0 : (size_t)b * D * H * W * Cout);
    u16* zb16 = reinterpret_cast<u16*>(z_) + (OUT16 ? (size_t)b * D * H * W * Cout : 0);
    auto epilogue = [&](auto full_c, auto stats_c) {
        constexpr bool FULL = decltype(full_c)::value, STATS = decltype(stats_c)::value;
        if constexpr (OUT16) {
            constexpr int OOB16 = 0x7FFFFFF0;
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(zb16, 0, D * H * W * Cout * 2, 0x00020000);
            const int plane = H * W * Cout * 2;                               // bytes
            const int odd = lane & 1;
            const int co = PERM ? n0 + 2 * l31 : n0 + (l31 & ~1);             // first channel of the stored pair
            const int gh = h0 + wave;
            const unsigned sel = odd ? 0x03020706u : 0x05040100u;             // v_perm: (other.hi, own.hi) | (own.lo, other.lo)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int gw = w0 + 2 * q4 + hsel;
                    const bool lane_ok = FULL || (gh < H && gw < W && co < Cout);
                    const int vo = ((gh * W + gw) * Cout + co) * 2;
                    if constexpr (PERM) {
                        const int vof = lane_ok ? vo : OOB16;
#pragma unroll
                        for (int pd = 0; pd < 4; ++pd) {
                            const int r = q4 * 4 + pd, gd = d0 + 4 * m + pd;
                            const bool d_ok = FULL || gd < D;                 // wave-uniform
                            float v0 = acc[m][0][r], v1 = acc[m][NT - 1][r];
                            if (d_ok && !(dbg & 16)) __builtin_amdgcn_raw_buffer_store_b32(pack_bf16(v0, v1), zr, vof, gd * plane, 0);
                            if constexpr (STATS) {
                                if (!FULL) { v0 = (lane_ok && d_ok) ? v0 : 0.f; v1 = (lane_ok && d_ok) ? v1 : 0.f; }
                                s1[0] += v0; s2[0] += v0 * v0;
                                s1[NT - 1] += v1; s2[NT - 1] += v1 * v1;
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bool lane_okj = FULL || (lane_ok && co + j * 32 < Cout);
#pragma unroll
                            for (int pd = 0; pd < 4; pd += 2) {
                                const int r = q4 * 4 + pd, gd = d0 + 4 * m + pd;      // this lane stores plane gd + odd
                                float a = acc[m][j][r], bq = acc[m][j][r + 1];
                                const unsigned own = pack_bf16(a, bq);
                                const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                                const unsigned pk = __builtin_amdgcn_perm(oth, own, sel);
                                int vof = vo + j * 64 + odd * plane;
                                if (!FULL) vof = (lane_okj && gd + odd < D) ? vof : OOB16;
                                if ((FULL || gd < D) && !(dbg & 16)) __builtin_amdgcn_raw_buffer_store_b32(pk, zr, vof, gd * plane, 0);
                                if constexpr (STATS) {
                                    if (!FULL) {
                                        const bool cv = gh < H && gw < W && n0 + j * 32 + l31 < Cout;
                                        a = (cv && gd < D) ? a : 0.f;
                                        bq = (cv && gd + 1 < D) ? bq : 0.f;
                                    }
                                    s1[j] += a; s2[j] += a * a;
                                    s1[j] += bq; s2[j] += bq * bq;
                                }
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gd = d0 + 4 * m + (r & 3), gh = h0 + wave, gw = w0 + 2 * (r >> 2) + hsel;
                    const bool pv = FULL || (gd < D && gh < H && gw < W);
                    const int off = ((gd * H + gh) * W + gw) * Cout;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int co = n0 + j * 32 + l31;
                        if (FULL || (pv && co < Cout)) {
                            const float v = acc[m][j][r];
                            zb[off + co] = v;
                            s1[j] += v;
                            s2[j] += v * v;
                        }
                    }
                }
            }
        }
    };
    {
        const bool full = d0 + TD <= D && h0 + TH <= H && w0 + TW <= W && n0 + NB <= Cout;
        if (stat_partial != nullptr) {
            if (full) epilogue(std::true_type{}, std::true_type{});
            else epilogue(std::false_type{}, std::true_type{});
        } else {
            if (full) epilogue(std::true_type{}, std::false_type{});
            else epilogue(std::false_type{}, std::false_type{});
        }
    }
    TR(31);

    if (stat_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
        }
        __syncthreads();                                      // every wave is done reading the halo (red aliases it)
        if (hsel == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cr = PERM ? 2 * l31 + j : j * 32 + l31;            // channel of (tile j, lane) inside the workgroup's tile
                red[(wave * NB + cr) * 2 + 0] = s1[j];
                red[(wave * NB + cr) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < NB) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) { a1 += red[(m * NB + tid) * 2]; a2 += red[(m * NB + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < Cout) {
                stat_partial[((size_t)tile * 2 + 0) * Cout + co] = a1;
                stat_partial[((size_t)tile * 2 + 1) * Cout + co] = a2;
            }
        }
    }
    TR(40);
}


// ------------------------------------------------------------------------------------------------------------
// Round 4 — two more structures for the 32-channel layers were built, measured against the kernel above and REMOVED
// (profiles/r04_bf16_r4_ab.txt, r04_bf16_r5_ab.txt, r04_bf16_data_probe.txt):
//   * four waves per brick, a 4 x 1 register tile whose A fragments are shared between the kh taps of a brick column (0.75
//     operand reads per MFMA instead of 1.5, twice the MFMAs per wave and brick, three workgroups per CU; bitwise-equal z):
//     106 / 103 / 232 us against 112 / 102 / 234;
//   * the halo's WHOLE channel depth fetched once per brick in 64-byte rows ((kd, kw) weight columns, output tiles walked
//     inside the kernel; half the fabric traffic — the 16-channel chunks above request every 128-byte line once per chunk):
//     103-112 us against 106-110 on conv2.0, 220 against 200 on conv2.3.
// What the three have in common is the matrix work, and on random operands that is what the chip's POWER budget prices: the
// same launches on zero operands run 18-34 % faster (conv2.0 110 -> 82 us = 0.57 of the nominal peak): the chip clocks to its
// power budget (MI355X_MICROARCH.md, DVFS give-back; there a GEMM template gains 15-21 % the same way).
// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
// fp32-ACCURATE convolution on the bf16 matrix cores ("3-way split"): every fp32 operand is written as
// hi + mid + lo with three bf16 numbers (8 + 8 + 8 significand bits: the decomposition is EXACT), and the
// product a*b is evaluated as the six partial products of order >= 2^-16 — (h,h) (h,m) (m,h) (h,l) (l,h) (m,m) —
// each exact in fp32, accumulated in fp32 by the MFMA.  The three dropped terms are <= 2^-24 |a b|, i.e. below
// the rounding of the fp32 product itself, so the result carries fp32 accuracy (tests hold it to the same 1e-5 as
// the exact-fp32 MFMA kernel) at 6/16 of the fp32-MFMA matrix time.  Opt-in (set_conv_precision("fp32x")).
// ------------------------------------------------------------------------------------------------------------
constexpr int SC = 16;                                // input channels per chunk (one MFMA k-block)
constexpr int SRP = 16;                               // LDS row = 16 bf16 = 32 B, two 16-B halves XOR-swizzled by row bit 3
constexpr int STPS = 9;                               // taps per weight stage: one kd plane, 3 stages per chunk
constexpr int SNST = 3;
constexpr int SHALO = NHALO * SRP;                    // bf16 elements per halo part
constexpr int SWSTAGE = STPS * 32 * SRP;              // bf16 elements per weight stage and part
constexpr size_t SPLIT_LDS_BYTES = (size_t)(3 * SHALO + 2 * 3 * SWSTAGE) * 2 + 8 * 32 * 2 * 4;
__device__ __forceinline__ int swz(int row, int half) { return row * SRP + ((half ^ ((row >> 3) & 1)) << 3); }

__device__ __forceinline__ unsigned int rne_bf16_bits(float a) {          // bf16 bit pattern in the low half
    return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)a);
}
__device__ __forceinline__ void split3(float a, unsigned int& h, unsigned int& m, unsigned int& l) {
    h = rne_bf16_bits(a);
    const float r1 = a - __builtin_bit_cast(float, h << 16);             // exact
    m = rne_bf16_bits(r1);
    const float r2 = r1 - __builtin_bit_cast(float, m << 16);            // exact, fits bf16
    l = rne_bf16_bits(r2);
}

__global__ __launch_bounds__(NTHR) void conv3d_fwd_split_kernel(
    const float* __restrict__ x, const u16* __restrict__ w3, float* __restrict__ z,
    float* __restrict__ stat_partial, int D, int H, int W, int Cin, int Cout,
    int tilesD, int tilesH, int tilesW, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* halo = reinterpret_cast<u16*>(smem_raw);             // [3 parts][NHALO][SRP]
    u16* Ws = halo + 3 * SHALO;                                // [2 buffers][3 parts][TPS][32 co][SRP]
    float* red = reinterpret_cast<float*>(Ws + 2 * 3 * SWSTAGE);

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
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int n0 = blockIdx.y * 32;

    int a_lane;
    {
        // wave = brick row h, lane = (d low bits, w): the mapping whose ds_read_b128 service groups are
        // conflict-free with the bit-3 swizzle for every tap (searched exhaustively; see conv3d_fwd_bf16_kernel)
        const int pd = l31 & 3, pw = l31 >> 2, ph = wave;
        a_lane = (pd * HH + ph) * HW + pw;                      // halo ROW of this lane's voxel (tap (0,0,0))
    }
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const float* xb = x + (size_t)b * D * H * W * Cin;

    // ---- brick-invariant addressing, computed ONCE (the staging VALU work was the bottleneck: 970 VALU
    //      instructions per wave and chunk in the first version, PMC) ----
    constexpr int HV = (NHALO * 4 + NTHR - 1) / NTHR;       // halo float4 pieces per thread (4 per position)
    int hoff[HV], hdst[HV];                                   // global element offset (-1 = zero fill), LDS element offset
#pragma unroll
    for (int q = 0; q < HV; ++q) {
        const int e = tid + q * NTHR;
        const int hp = e >> 2, c4 = e & 3;
        const int hw = hp % HW, hh = (hp / HW) % HH, hd = hp / (HW * HH);
        const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
        const bool ok = e < NHALO * 4 && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
        hoff[q] = ok ? ((gd * H + gh) * W + gw) * Cin + c4 * 4 : -1;
        hdst[q] = e < NHALO * 4 ? swz(hp, c4 >> 1) + (c4 & 1) * 4 : -1;
    }
    constexpr int WV = (3 * STPS * 32 * 2 + NTHR - 1) / NTHR;   // weight 16-B pieces per thread and stage
    int woff[WV], wdst[WV];
#pragma unroll
    for (int q = 0; q < WV; ++q) {
        const int e = tid + q * NTHR;
        const int piece = e & 1, row = e >> 1;                // row = (part * STPS + tap_in_stage) * 32 + co
        const int co = n0 + (row & 31), tp = (row >> 5) % STPS, part = row / (32 * STPS);
        const bool ok = e < 3 * STPS * 32 * 2 && co < Cout;
        woff[q] = ok ? part * (27 * Cout * Cin) + (tp * Cout + co) * Cin + piece * 8 : -1;
        wdst[q] = e < 3 * STPS * 32 * 2 ? swz(row, piece) : -1;
    }
    const int wstage_stride = STPS * Cout * Cin;

    for (int c0 = 0; c0 < Cin; c0 += SC) {
        if (c0 > 0) __syncthreads();
        // ---- halo: fp32 -> (hi, mid, lo) bf16, three LDS images ----
        f32x4 hreg[HV];
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hoff[q] >= 0 && c0 + (tid & 3) * 4 < Cin) v = *reinterpret_cast<const f32x4*>(xb + hoff[q] + c0);
            hreg[q] = v;
        }
        // ---- weights: pre-split on the host, [part][tap][co][ci]; one 16-B piece = 8 input channels ----
        u32x4 wreg[WV];
        auto load_w = [&](int st) {
#pragma unroll
            for (int q = 0; q < WV; ++q) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (woff[q] >= 0 && c0 + (tid & 1) * 8 < Cin)
                    v = *reinterpret_cast<const u32x4*>(w3 + woff[q] + st * wstage_stride + c0);
                wreg[q] = v;
            }
        };
        auto store_w = [&](int buf) {
#pragma unroll
            for (int q = 0; q < WV; ++q)
                if (wdst[q] >= 0) *reinterpret_cast<u32x4*>(Ws + buf * 3 * SWSTAGE + wdst[q]) = wreg[q];
        };
        load_w(0);
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            if (hdst[q] >= 0) {
                // exact 3-way split by truncation: a = hi + mid + lo, each with <= 8 significand bits
                unsigned int hh_[4], mm_[4], ll_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = hreg[q][u];
                    const unsigned int hb = __builtin_bit_cast(unsigned int, a) & 0xFFFF0000u;
                    const float r1 = a - __builtin_bit_cast(float, hb);
                    const unsigned int mb = __builtin_bit_cast(unsigned int, r1) & 0xFFFF0000u;
                    const float r2 = r1 - __builtin_bit_cast(float, mb);
                    hh_[u] = hb; mm_[u] = mb; ll_[u] = __builtin_bit_cast(unsigned int, r2);
                }
                u16* dst = halo + hdst[q];
                u32x2 ph = {(hh_[0] >> 16) | hh_[1], (hh_[2] >> 16) | hh_[3]};
                u32x2 pm = {(mm_[0] >> 16) | mm_[1], (mm_[2] >> 16) | mm_[3]};
                u32x2 pl = {(ll_[0] >> 16) | (ll_[1] & 0xFFFF0000u), (ll_[2] >> 16) | (ll_[3] & 0xFFFF0000u)};
                *reinterpret_cast<u32x2*>(dst) = ph;
                *reinterpret_cast<u32x2*>(dst + SHALO) = pm;
                *reinterpret_cast<u32x2*>(dst + 2 * SHALO) = pl;
            }
        }
        for (int st = 0; st < SNST; ++st) {                     // stage = kd plane (9 taps)
            const int buf = st & 1;
            store_w(buf);
            __syncthreads();
            if (st + 1 < SNST) load_w(st + 1);
            const u16* ws = Ws + buf * 3 * SWSTAGE;
            // explicit software pipeline: the six fragments of tap t+1 are in flight while tap t multiplies
            bf16x8 fa[2][3], fb[2][3];
            auto fetch = [&](const int tp, const int slot) {
                const int arow = a_lane + (st * HH + tp / 3) * HW + tp % 3;
                const u16* ap = halo + swz(arow, hsel);
                const u16* bp = ws + swz(tp * 32 + l31, hsel);
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    fa[slot][part] = *reinterpret_cast<const bf16x8*>(ap + part * SHALO);
                    fb[slot][part] = *reinterpret_cast<const bf16x8*>(bp + part * SWSTAGE);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int tp = 0; tp < STPS; ++tp) {
                const int cur = tp & 1;
                if (tp + 1 < STPS) fetch(tp + 1, cur ^ 1);
                const bf16x8 ah = fa[cur][0], am = fa[cur][1], al = fa[cur][2];
                const bf16x8 bh = fb[cur][0], bm = fb[cur][1], bl = fb[cur][2];
                // two independent chains: the 2^-16..2^-8 cross terms, and the leading terms
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
    }
    acc += acc2;

    float s1 = 0.f, s2 = 0.f;
    float* zb = z + (size_t)b * D * H * W * Cout;
    const int co = n0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pd = r & 3, pw = 2 * (r >> 2) + hsel, ph = wave;
        const int gd = d0 + pd, gh = h0 + ph, gw = w0 + pw;
        if (gd < D && gh < H && gw < W && co < Cout) {
            const float v = acc[r];
            zb[((gd * H + gh) * W + gw) * Cout + co] = v;
            s1 += v;
            s2 += v * v;
        }
    }
    if (stat_partial != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hsel == 0) { red[(wave * 32 + l31) * 2] = s1; red[(wave * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 32 && n0 + tid < Cout) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) { a1 += red[(m * 32 + tid) * 2]; a2 += red[(m * 32 + tid) * 2 + 1]; }
            stat_partial[((size_t)tile * 2 + 0) * Cout + n0 + tid] = a1;
            stat_partial[((size_t)tile * 2 + 1) * Cout + n0 + tid] = a2;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores:  dw[tap][ci][co] = sum_v x[v + tap][ci] * dz[v][co],  M = ci, N = co,
// K = voxels.  The 32x32x16 MFMA wants 8 consecutive k per lane, i.e. 8 VOXELS of one channel, so both operands are
// staged TRANSPOSED (channel-major, one brick row of 8 voxels = one 16-B fragment) while they are rounded to bf16.
// A tap shift along w would misalign those 16-B runs, so x is kept as THREE images pre-shifted by kw = 0, 1, 2
// (the thread that transposes a halo row holds its 10 values and writes the three 8-value windows); shifts along
// d / h are plain row offsets.  Per-channel pitches of 61 / 33 sixteen-byte slots (odd) make both fragment reads
// conflict-free.  Taps are spread over the waves exactly as in the fp32 kernel (7 per SIMD, 4 + 3 per wave pair).
// ------------------------------------------------------------------------------------------------------------
constexpr int WG_XP = 61 * 8;                         // bf16 per input channel in one shifted image (60 halo rows + pad)
constexpr int WG_XIMG = 32 * WG_XP;
constexpr int WG_DP = 33 * 8;                         // bf16 per output channel (32 brick rows + pad)
constexpr size_t WG_LDS_BYTES = (size_t)(3 * WG_XIMG + 32 * WG_DP) * 2;

template <bool IN16>
__global__ __launch_bounds__(NTHR) void conv3d_wgrad_bf16_kernel(
    const void* __restrict__ x_, const void* __restrict__ dz_, float* __restrict__ partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int ntiles, int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* xT = reinterpret_cast<u16*>(smem_raw);               // [3 shifts][32 ci][61 rows][8]
    u16* dzT = xT + 3 * WG_XIMG;                              // [32 co][33 rows][8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;

    const int tap_base = 7 * (wave & 3) + 4 * (wave >> 2);    // waves w and w+4 share a SIMD: 7 taps per SIMD
    const int tap_cnt = wave < 4 ? 4 : 3;
    int a_off[4];                                             // element offset of tap t's fragment at k-step 0
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int tap = tap_base + t;
        tap = tap > 26 ? 26 : tap;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        a_off[t] = kw * WG_XIMG + l31 * WG_XP + (kd * HH + kh + hsel) * 8;
    }
    const int b_off = l31 * WG_DP + hsel * 8;

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > ntiles) tile_end = ntiles;
    const bool ci_ok = ci0 + l31 < Cin, co_ok = co0 + l31 < Cout;

    // Global loads of brick n+1 are issued into registers before the MFMA loop of brick n (the kernel was latency-
    // bound on its staging: 19k cycles per brick against 3.5k of MFMA); conversion + LDS writes follow the loop.
    constexpr int XQ = (HD * HH * 32 + NTHR - 1) / NTHR;      // 4 (halo row, ci) tasks per thread
    constexpr int DQ = TD * TH * 32 / NTHR;                   // 2 (brick row, co) tasks per thread
    // bf16 tensors: a lane loads a DWORD = one channel PAIR (sub-dword loads run at a fraction of the rate: the first
    // version with 2-byte loads was 3x slower than the fp32-tensor path), so tasks are (row, channel pair)
    constexpr int XQ2 = (HD * HH * 16 + NTHR - 1) / NTHR;     // 2
    float xv[IN16 ? 1 : XQ][HW], dv[IN16 ? 1 : DQ][TW];
    unsigned int xw[IN16 ? XQ2 : 1][HW], dw2[TW];
    const int pr = tid & 15;
    auto fetch = [&](int tile) {
        int tt = tile;
        const int tw = tt % tilesW; tt /= tilesW;
        const int th = tt % tilesH; tt /= tilesH;
        const int td = tt % tilesD;
        const int b = tt / tilesD;
        const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
        if constexpr (IN16) {
            const u16* xb16 = reinterpret_cast<const u16*>(x_) + (size_t)b * D * H * W * Cin + ci0 + 2 * pr;
            const u16* dzb16 = reinterpret_cast<const u16*>(dz_) + (size_t)b * D * H * W * Cout + co0 + 2 * pr;
            const bool cp_ok = ci0 + 2 * pr < Cin, op_ok = co0 + 2 * pr < Cout;
#pragma unroll
            for (int i = 0; i < XQ2; ++i) {
                const int hrow = (tid + i * NTHR) >> 4;
                const int hd = hrow / HH, hh = hrow % HH;
                const int gd = d0 + hd - 1, gh = h0 + hh - 1;
                const bool rv = cp_ok && hrow < HD * HH && gd >= 0 && gd < D && gh >= 0 && gh < H;
                const size_t src = (size_t)(gd * H + gh) * W * Cin;
#pragma unroll
                for (int q = 0; q < HW; ++q) {
                    const int gw = w0 + q - 1;
                    xw[i][q] = (rv && gw >= 0 && gw < W) ? *reinterpret_cast<const unsigned int*>(xb16 + src + (size_t)gw * Cin) : 0u;
                }
            }
            {
                const int row = tid >> 4;
                const int gd = d0 + row / TH, gh = h0 + row % TH;
                const bool rv = op_ok && gd < D && gh < H;
                const size_t src = (size_t)(gd * H + gh) * W * Cout;
#pragma unroll
                for (int q = 0; q < TW; ++q)
                    dw2[q] = (rv && w0 + q < W) ? *reinterpret_cast<const unsigned int*>(dzb16 + src + (size_t)(w0 + q) * Cout) : 0u;
            }
        } else {
            const float* xb = reinterpret_cast<const float*>(x_) + (size_t)b * D * H * W * Cin + ci0 + l31;
            const float* dzb = reinterpret_cast<const float*>(dz_) + (size_t)b * D * H * W * Cout + co0 + l31;
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int hrow = (tid + i * NTHR) >> 5;
                const int hd = hrow / HH, hh = hrow % HH;
                const int gd = d0 + hd - 1, gh = h0 + hh - 1;
                const bool rv = ci_ok && hrow < HD * HH && gd >= 0 && gd < D && gh >= 0 && gh < H;
                const float* src = xb + (size_t)(gd * H + gh) * W * Cin;
#pragma unroll
                for (int q = 0; q < HW; ++q) {
                    const int gw = w0 + q - 1;
                    xv[i][q] = (rv && gw >= 0 && gw < W) ? src[(size_t)gw * Cin] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < DQ; ++i) {
                const int row = (tid + i * NTHR) >> 5;
                const int gd = d0 + row / TH, gh = h0 + row % TH;
                const bool rv = co_ok && gd < D && gh < H;
                const float* src = dzb + (size_t)(gd * H + gh) * W * Cout;
#pragma unroll
                for (int q = 0; q < TW; ++q) dv[i][q] = (rv && w0 + q < W) ? src[(size_t)(w0 + q) * Cout] : 0.f;
            }
        }
    };
    // channel h (0 = low half, 1 = high half) of two pair-dwords, as one dword of two consecutive voxels
    auto pair16 = [](unsigned int a, unsigned int b_, int h) {
        return h ? ((a >> 16) | (b_ & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b_ << 16));
    };
    auto commit = [&]() {
        if constexpr (IN16) {
#pragma unroll
            for (int i = 0; i < XQ2; ++i) {
                const int hrow = (tid + i * NTHR) >> 4;
                if (hrow < HD * HH) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned int pk[HW - 1];
#pragma unroll
                        for (int q = 0; q < HW - 1; ++q) pk[q] = pair16(xw[i][q], xw[i][q + 1], h);
                        u16* dst = xT + (2 * pr + h) * WG_XP + hrow * 8;
#pragma unroll
                        for (int sft = 0; sft < 3; ++sft) {
                            u32x4 o = {pk[sft], pk[sft + 2], pk[sft + 4], pk[sft + 6]};
                            *reinterpret_cast<u32x4*>(dst + sft * WG_XIMG) = o;
                        }
                    }
                }
            }
            const int row = tid >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o = {pair16(dw2[0], dw2[1], h), pair16(dw2[2], dw2[3], h), pair16(dw2[4], dw2[5], h),
                           pair16(dw2[6], dw2[7], h)};
                *reinterpret_cast<u32x4*>(dzT + (2 * pr + h) * WG_DP + row * 8) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int hrow = (tid + i * NTHR) >> 5;
                if (hrow < HD * HH) {
                    unsigned int pk[HW - 1];                       // pk[q] = bf16(v[q]) | bf16(v[q+1]) << 16
#pragma unroll
                    for (int q = 0; q < HW - 1; ++q) pk[q] = pack_bf16(xv[i][q], xv[i][q + 1]);
                    u16* dst = xT + l31 * WG_XP + hrow * 8;
#pragma unroll
                    for (int sft = 0; sft < 3; ++sft) {
                        u32x4 o = {pk[sft], pk[sft + 2], pk[sft + 4], pk[sft + 6]};
                        *reinterpret_cast<u32x4*>(dst + sft * WG_XIMG) = o;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < DQ; ++i) {
                const int row = (tid + i * NTHR) >> 5;
                u32x4 o = {pack_bf16(dv[i][0], dv[i][1]), pack_bf16(dv[i][2], dv[i][3]), pack_bf16(dv[i][4], dv[i][5]),
                           pack_bf16(dv[i][6], dv[i][7])};
                *reinterpret_cast<u32x4*>(dzT + l31 * WG_DP + row * 8) = o;
            }
        }
    };

    if (tile_begin < tile_end) fetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();                                       // previous brick fully consumed
        commit();
        __syncthreads();
        if (tile + 1 < tile_end) fetch(tile + 1);

        auto mma = [&](auto ntaps_c) {
            constexpr int NTAPS = decltype(ntaps_c)::value;
#pragma unroll 4
            for (int ks = 0; ks < TD * TH / 2; ++ks) {        // 16 voxels per step: brick rows 2*ks and 2*ks + 1
                const int d = ks / (TH / 2), hp = (ks % (TH / 2)) * 2;
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(dzT + b_off + (d * TH + hp) * 8);
                const int xrow = (d * HH + hp) * 8;
                bf16x8 av[NTAPS];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) av[t] = *reinterpret_cast<const bf16x8*>(xT + a_off[t] + xrow);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bv, acc[t], 0, 0, 0);
            }
        };
        if (wave < 4) mma(std::integral_constant<int, 4>{});
        else mma(std::integral_constant<int, 3>{});
    }

    // partial[split][tap][ci][co]; D fragment: row = ci, column = co
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = tap_base + t;
        if (t < tap_cnt && tap < 27) {
            const int co = co0 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (ci < Cin && co < Cout)
                    partial[(((size_t)split * 27 + tap) * Cin + ci) * Cout + co] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient, second form: the operands stay in their NATURAL channels-last order in LDS and the transposition
// the matrix instruction needs (8 consecutive VOXELS of one channel per lane) is done by the LDS unit itself:
// ds_read_b64_tr_b16 hands lane l channel l % 32 of four voxels when the 16 lanes of a group address a [4 voxel][16
// channel] block (tools/microbench/tr16.hip prints the mapping).  So
//   * staging is a plain 16-byte copy (halo voxels are 64-B rows of 32 channels; no register transposes, no three
//     pre-shifted images: a tap shift along w is a 64-B address offset like the shifts along d / h), 38 KB + 16 KB per
//     32 output channels per brick instead of 110 KB;
//   * the brick is double-buffered in LDS with ONE barrier per brick, and the two waves that share a SIMD run the
//     staging of brick n+1 at opposite ends of their matrix loops (waves 0-3: stage, then multiply; waves 4-7:
//     multiply, then stage), so the matrix pipe always has a wave feeding it;
//   * a workgroup covers 32 x (32 * NH) channels: with NH = 2 every x fragment feeds two MFMAs.
// Needs cin % 8 == 0 and cout % 8 == 0 (16-byte channel groups); other shapes use the kernel above.
// ------------------------------------------------------------------------------------------------------------
namespace wtr {
constexpr int XB = HD * HH * HW * 64;                  // bytes of one halo image: [600 voxels][32 ci] bf16
constexpr int DZB = TD * TH * TW * 64;                 // bytes of one 32-channel dz image: [256 voxels][32 co] bf16
constexpr int XCH = HD * HH * HW * 4;                  // 16-byte chunks of the halo image
constexpr int XQ = (XCH + NTHR - 1) / NTHR;            // 5 per thread
constexpr size_t lds_bytes(int nh) { return (size_t)2 * (XB + nh * DZB); }
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr8(const unsigned char* p) {      // voxels 0-3 at p, voxels 4-7 at p + 4 * 64
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 256));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
using namespace dma;
}  // namespace wtr

template <int NH, bool IN16>
__global__ __launch_bounds__(NTHR) void conv3d_wgrad_bf16_tr_kernel(
    const void* __restrict__ x_, const void* __restrict__ dz_, float* __restrict__ partial,
    int D, int H, int W, int Cin, int Cout, int tilesD, int tilesH, int tilesW, int ntiles, int tiles_per_split) {
    using namespace wtr;
    // timing ablations (compile-time: -DTMF_ABLATE_WG=bits builds made by tools/wgrad_ablate.py; results are garbage
    // when set): 1 = no staging copies, 2 = no LDS fragment reads, 4 = no MFMAs, 8 = no barriers
#ifdef TMF_ABLATE_WG
    constexpr int dbg = TMF_ABLATE_WG;
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int BUFB = XB + NH * DZB;
    constexpr int DQ = NH * 2;                                // dz chunks per thread (NH images x 256 voxels x 4 chunks)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hsel = lane >> 5;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32 * NH;

    // Seven taps per SIMD, shared by its two waves (w and w + 4).  NH = 2 (two 32-channel halves of dz): each wave owns
    // three taps with both halves and ONE half of the SIMD's seventh tap - seven MFMAs per wave and k-step, identical
    // instruction streams (SIMD 3 has only six taps: its seventh is a repeat whose result is dropped; the other SIMDs
    // take seven anyway).  NH = 1: four taps for wave w, three for wave w + 4.
    constexpr int NACC = NH == 2 ? 7 : 4;
    const int simd = wave & 3, pair = wave >> 2;
    const bool four = NH == 2 || pair == 0;                   // the fourth fragment slot is in use
    int tap_of[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        tap_of[t] = NH == 2 ? (t < 3 ? 7 * simd + 3 * pair + t : 7 * simd + 6) : 7 * simd + 4 * pair + t;
    // the lane's 8 bytes inside a [4 voxel][32 channel] block: voxel i16 / 4, channels 16 * g16 + 4 * (i16 % 4) ...
    const int lane_off = (i16 >> 2) * 64 + (i16 & 3) * 8 + g16 * 32;
    int a_off[4];                                             // byte offset of tap t's fragment at k-step 0
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = tap_of[t] > 26 ? 26 : tap_of[t];
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        a_off[t] = (((kd * HH + kh + hsel) * HW) + kw) * 64 + lane_off;
    }
    const int b_off = XB + hsel * TW * 64 + lane_off;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Which bricks: with a multiple of 8 workgroups per channel block, workgroup s runs on XCD s % 8 (round-robin
    // dispatch), so XCD x takes the x-th eighth of the bricks and its workgroups walk it INTERLEAVED (brick j, j + n,
    // j + 2n, ...): at any moment the workgroups that share an L2 hold neighbouring bricks, whose halos overlap (a
    // brick's halo is 2.3x its voxels; with contiguous ranges per workgroup every overlap was fetched again from HBM).
    int tile_begin, tile_end, tile_step;
    if ((gridDim.x & 7) == 0) {
        const int per_xcd = (ntiles + 7) >> 3, xcd = split & 7;
        tile_begin = xcd * per_xcd + (split >> 3);
        tile_step = gridDim.x >> 3;
        tile_end = (xcd + 1) * per_xcd;
    } else {
        tile_begin = split * tiles_per_split;
        tile_step = 1;
        tile_end = tile_begin + tiles_per_split;
    }
    if (tile_end > ntiles) tile_end = ntiles;

    // One staging chunk = 16 bytes of the LDS image = 8 channels of one voxel.  Chunk c of the halo image is voxel c / 4,
    // channels 8 * (c % 4); chunk c of the dz images is image c / 1024, voxel (c % 1024) / 4 — LDS byte 16 * c in both,
    // which is what an LDS-DMA wants (wave-uniform base + 16 * lane).  Per thread and chunk, brick-independent: the
    // element offset relative to the brick origin and the packed halo coordinates for the bounds test.
    // The bounds test is two operations per chunk: each chunk carries one-hot bits of its halo coordinates (bit 31 for a
    // chunk that never exists), each brick a mask of the coordinates that fall inside the volume.
    int xrel[XQ], drel[DQ];
    unsigned xoh[XQ], doh[DQ];
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
        const int c = tid + i * NTHR;
        const int hv = c >> 2, part = c & 3;
        const int hd = hv / (HH * HW), r2 = hv - hd * (HH * HW);
        const int hh = r2 / HW, hw = r2 - hh * HW;
        const int ch = ci0 + part * 8;
        xrel[i] = (((hd - 1) * H + (hh - 1)) * W + (hw - 1)) * Cin + ch;
        xoh[i] = (c < XCH && ch < Cin) ? ((1u << hd) | (1u << (8 + hh)) | (1u << (20 + hw))) : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < DQ; ++i) {
        const int c = tid + i * NTHR;
        const int img = c >> 10, v = (c & 1023) >> 2, part = c & 3;
        const int vd = v >> 6, vh = (v >> 3) & 7, vw = v & 7;
        const int ch = co0 + img * 32 + part * 8;
        drel[i] = ((vd * H + vh) * W + vw) * Cout + ch;
        doh[i] = ch < Cout ? ((1u << vd) | (1u << (8 + vh)) | (1u << (20 + vw))) : 0x80000000u;
    }
    struct Origin { size_t vox; int b, vin; unsigned xmask, dmask; };   // vox = b * D H W + vin
    auto bits = [](int lo, int hi) { return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u; };   // bits lo .. hi-1
    auto imin = [](int a_, int b_) { return a_ < b_ ? a_ : b_; };
    // The bricks of a workgroup are tile_begin, + tile_step, ...: their coordinates advance incrementally (the step's own
    // (b, d, h, w) digits with carries) — three runtime divisions per brick were ~100 scalar instructions, and a scalar
    // instruction costs a SIMD about as much issue time as a vector one.  origin_next() returns the cursor's
    // brick and advances the cursor; both staging forms ask for the bricks strictly in order.
    int cw, ch, cd, cb, sw, sh, sd, sb;
    {
        int tt = tile_begin;
        cw = tt % tilesW; tt /= tilesW; ch = tt % tilesH; tt /= tilesH; cd = tt % tilesD; cb = tt / tilesD;
        tt = tile_step;
        sw = tt % tilesW; tt /= tilesW; sh = tt % tilesH; tt /= tilesH; sd = tt % tilesD; sb = tt / tilesD;
    }
    auto origin_next = [&]() {
        const int d0 = cd * TD, h0 = ch * TH, w0 = cw * TW;
        Origin o;
        o.vox = (((size_t)cb * D + d0) * H + h0) * W + w0;
        o.b = cb;
        o.vin = (d0 * H + h0) * W + w0;
        // halo coordinate q is voxel origin + q - 1: inside the volume for 1 - origin <= q < extent - origin + 1
        o.xmask = bits(d0 ? 0 : 1, imin(HD, D - d0 + 1)) | (bits(h0 ? 0 : 1, imin(HH, H - h0 + 1)) << 8) |
                  (bits(w0 ? 0 : 1, imin(HW, W - w0 + 1)) << 20);
        o.dmask = bits(0, imin(TD, D - d0)) | (bits(0, imin(TH, H - h0)) << 8) | (bits(0, imin(TW, W - w0)) << 20);
        cw += sw; if (cw >= tilesW) { cw -= tilesW; ++ch; }
        ch += sh; if (ch >= tilesH) { ch -= tilesH; ++cd; }
        cd += sd; if (cd >= tilesD) { cd -= tilesD; ++cb; }
        cb += sb;
        return o;
    };
    auto x_ok = [&](const Origin& o, int i) { return (xoh[i] & o.xmask) == xoh[i]; };
    auto dz_ok = [&](const Origin& o, int i) { return (doh[i] & o.dmask) == doh[i]; };

    // The k-loop of one brick, fully unrolled and software-pipelined by hand: the fragments of k-step ks + 1 are
    // requested BEFORE the MFMAs of k-step ks are issued (left to the compiler they were requested behind all but the
    // last two: both waves of a SIMD then sat out the LDS latency together, 50 us of a 240 us launch).  `mid` runs after
    // the second k-step: the copies of the next brick are issued in the shadow of queued MFMAs, not in front of them.
    struct Frag { bf16x8 a[4]; bf16x8 b[NH]; };
    auto load = [&](const unsigned char* base, int ks, Frag& f) {
        if constexpr ((dbg & 2) != 0) {
            bf16x8 c;
#pragma unroll
            for (int e = 0; e < 8; ++e) c[e] = (__bf16)(float)(ks + e);
#pragma unroll
            for (int n = 0; n < NH; ++n) f.b[n] = c;
#pragma unroll
            for (int t = 0; t < 4; ++t) f.a[t] = c;
            return;
        }
        const int d = ks / (TH / 2), hp = (ks % (TH / 2)) * 2;       // 16 voxels per step: brick rows 2*ks and 2*ks + 1
        const unsigned char* bb = base + b_off + (d * TH + hp) * TW * 64;
        const unsigned char* ab = base + (d * HH + hp) * HW * 64;
#pragma unroll
        for (int n = 0; n < NH; ++n) f.b[n] = tr8(bb + n * DZB);
#pragma unroll
        for (int t = 0; t < 3; ++t) f.a[t] = tr8(ab + a_off[t]);
        if (four) f.a[3] = tr8(ab + a_off[3]);
    };
    auto mul = [&](const Frag& f) {
        if constexpr ((dbg & 4) != 0) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t][0] += (float)f.a[t & 3][0] * (float)f.b[t % NH][1];
            return;
        }
        if constexpr (NH == 2) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[2 * t + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[t], f.b[n], acc[2 * t + n], 0, 0, 0);
            const bf16x8 bs = pair ? f.b[1] : f.b[0];
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[3], bs, acc[6], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[t], f.b[0], acc[t], 0, 0, 0);
            if (four) acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[3], f.b[0], acc[3], 0, 0, 0);
        }
    };
    auto mma = [&](int buf, auto&& mid) {
        const unsigned char* base = smem_raw + buf * BUFB;
        constexpr int KS = TD * TH / 2;
        Frag f0, f1;
        load(base, 0, f0);
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            load(base, ks + 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mul(f0);
            if (ks + 2 < KS) load(base, ks + 2, f0);
            __builtin_amdgcn_sched_barrier(0);
            mul(f1);
            if (ks == 0) mid();
        }
    };

    if constexpr (IN16) {
        // bf16 tensors: the images are byte copies of global memory -> LDS-DMA, no staging registers; voxels outside the
        // volume (and channels past the end) are copied from 16 zero bytes.  Brick n+1 streams into the other buffer
        // while brick n is multiplied; the barrier at the end of the brick waits for it.
        const u16* xg = reinterpret_cast<const u16*>(x_);
        const u16* dg = reinterpret_cast<const u16*>(dz_);
        // Copies through buffer resources (blds16): the per-thread byte offsets xvo / dvo are fixed for the whole kernel, a
        // brick contributes one scalar offset per tensor, and a chunk outside the volume (or past the channels) takes the
        // out-of-range offset, which the hardware answers with zeros: and + compare + select per copy instead of a 64-bit
        // address, a pointer select against a zero block and the exec juggling around it.  The x resource starts one halo
        // margin AHEAD of the tensor (the first brick's halo offsets are negative; those chunks are masked anyway).
        constexpr int OOBV = 0x7FFFFFF0;
        const int xmargin = (H * W + W + 1) * Cin;                                   // elements
        int xvo[XQ], dvo[DQ];
#pragma unroll
        for (int i = 0; i < XQ; ++i) xvo[i] = (xoh[i] & 0x80000000u) ? OOBV : (xrel[i] + xmargin) * 2;
#pragma unroll
        for (int i = 0; i < DQ; ++i) dvo[i] = (doh[i] & 0x80000000u) ? OOBV : drel[i] * 2;
        const size_t sample_x = (size_t)D * H * W * Cin, sample_d = (size_t)D * H * W * Cout;
        auto issue = [&](int buf) {
            if (dbg & 1) return;
            const Origin o = origin_next();
            const unsigned wbase = __builtin_amdgcn_readfirstlane(lds_addr(smem_raw) + buf * BUFB + wave * 1024);   // this wave's 64 chunks of pass 0
            const size_t vs = (size_t)o.b;                                          // the brick's sample (scalar)
            const int vin = o.vin;                                                   // its origin voxel inside the sample
            const i32x4 xr = make_rsrc(xg + vs * sample_x - xmargin, (unsigned)((sample_x + 2 * (size_t)xmargin) * 2));
            const i32x4 dr = make_rsrc(dg + vs * sample_d, (unsigned)(sample_d * 2));
            const int xso = vin * Cin * 2, dso = vin * Cout * 2;
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int vo = x_ok(o, i) ? xvo[i] : OOBV;
                if ((i + 1) * NTHR <= XCH || tid + i * NTHR < XCH) blds16(vo, xr, xso, wbase + i * (NTHR * 16));
            }
#pragma unroll
            for (int i = 0; i < DQ; ++i) {
                const int vo = dz_ok(o, i) ? dvo[i] : OOBV;
                blds16(vo, dr, dso, wbase + XB + i * (NTHR * 16));
            }
        };
        int cur = 0;
        if (tile_begin < tile_end) issue(0);
        dma_wait();
        __syncthreads();
        for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
            mma(cur, [&]() { if (tile + tile_step < tile_end) issue(cur ^ 1); });
            dma_wait();                          // this wave's copies of brick n+1 have landed ...
            if (!(dbg & 8)) __syncthreads();     // ... and so have everybody else's
            cur ^= 1;
        }
    } else {
        // fp32 tensors: rounded to bf16 on the way (two 16-byte loads -> one 16-byte LDS write per chunk).  The two waves
        // that share a SIMD stage brick n+1 at opposite ends of their matrix loops (waves 0-3: stage, then multiply;
        // waves 4-7: multiply, then stage), so the matrix pipe always has a wave feeding it.
        const float* xg = reinterpret_cast<const float*>(x_);
        const float* dg = reinterpret_cast<const float*>(dz_);
        u32x4 xr[XQ], dr[DQ];
        auto load8 = [&](const float* p8, bool ok) -> u32x4 {
            if (!ok) return u32x4{0u, 0u, 0u, 0u};
            const float4 a = *reinterpret_cast<const float4*>(p8);
            const float4 b_ = *reinterpret_cast<const float4*>(p8 + 4);
            return u32x4{pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b_.x, b_.y), pack_bf16(b_.z, b_.w)};
        };
        auto fetch = [&]() {
            const Origin o = origin_next();
            const size_t xo = o.vox * Cin, dzo = o.vox * Cout;
#pragma unroll
            for (int i = 0; i < XQ; ++i) xr[i] = load8(xg + (ptrdiff_t)xo + xrel[i], x_ok(o, i));
#pragma unroll
            for (int i = 0; i < DQ; ++i) dr[i] = load8(dg + (ptrdiff_t)dzo + drel[i], dz_ok(o, i));
        };
        auto commit = [&](int buf) {
            unsigned char* base = smem_raw + buf * BUFB + tid * 16;
#pragma unroll
            for (int i = 0; i < XQ; ++i)
                if (tid + i * NTHR < XCH) *reinterpret_cast<u32x4*>(base + i * (NTHR * 16)) = xr[i];
#pragma unroll
            for (int i = 0; i < DQ; ++i) *reinterpret_cast<u32x4*>(base + XB + i * (NTHR * 16)) = dr[i];
        };
        int cur = 0;
        if (tile_begin < tile_end) {
            fetch();
            commit(0);
            if (tile_begin + tile_step < tile_end) fetch();
        }
        __syncthreads();
        for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
            const bool more = tile + tile_step < tile_end;
            auto stage = [&]() {
                commit(cur ^ 1);
                if (tile + 2 * tile_step < tile_end) fetch();
            };
            if (pair == 0 && more) stage();
            mma(cur, []() {});
            if (pair != 0 && more) stage();
            __syncthreads();
            cur ^= 1;
        }
    }

    // partial[split][tap][ci][co]; D fragment: row = ci, column = co
#pragma unroll
    for (int u = 0; u < NACC; ++u) {
        const int t = NH == 2 ? (u < 6 ? u >> 1 : 3) : u;
        const int n = NH == 2 ? (u < 6 ? (u & 1) : pair) : 0;
        const int tap = tap_of[t];
        if (tap < 27 && (t < 3 || four)) {
            const int co = co0 + n * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
                if (ci < Cin && co < Cout)
                    partial[(((size_t)split * 27 + tap) * Cin + ci) * Cout + co] = acc[u][r];
            }
        }
    }
}

struct WgBfPlan { int tilesD, tilesH, tilesW, ntiles, gy, gz, tps, nsplit, nh; };
// nh = 0: the register-transposing kernel (any channel counts); 1 / 2: the transposing-read kernel with 32 / 64 output
// channels per workgroup (cin % 8 == 0 and cout % 8 == 0)
int wgrad_tr_nh(int cin, int cout, int io) {
    if (tmf_g_wgrad_tr == 0 || cin % 8 != 0 || cout % 8 != 0) return 0;
    if (io) return cout > 32 ? 2 : 1;
    // fp32 tensors are rounded on the way and so staged through registers (32 output channels per workgroup): measured
    // ahead of the register-transposing kernel from 64 input channels on, level with it or behind below
    return (cin >= 64 || tmf_g_wgrad_tr == 2) ? 1 : 0;
}
WgBfPlan plan_wgrad_bf16(int B, int D, int H, int W, int cin, int cout, int io) {
    WgBfPlan p;
    p.nh = wgrad_tr_nh(cin, cout, io);
    p.tilesD = tmf_cdiv(D, TD); p.tilesH = tmf_cdiv(H, TH); p.tilesW = tmf_cdiv(W, TW);
    p.ntiles = B * p.tilesD * p.tilesH * p.tilesW;
    p.gy = tmf_cdiv(cin, 32); p.gz = tmf_cdiv(cout, p.nh == 2 ? 64 : 32);
    int want = 256 / (p.gy * p.gz);                     // one workgroup per CU and group; each walks ntiles / want bricks
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tps = tmf_cdiv(p.ntiles, want);
    p.nsplit = tmf_cdiv(p.ntiles, p.tps);
    return p;
}

}  // namespace

// 8x8x8-brick kernel (2 x NT register tiles) when the launch has enough of those bricks to fill the chip's 512 workgroup
// slots reasonably; the choice depends on the geometry only (tmf_conv3d_bf16_stat_blocks has no channel arguments).
// tmf_set_option("bf16_v2", 0) / TMF_BF_V2=0 selects the small-brick kernel everywhere, 2 the large-brick kernel
// everywhere (A/B runs, tests), 1 (default) by the brick count.
int tmf_g_debug = 0;             // tmf_set_option("debug", bits): timing ablations (results are garbage when set)
int tmf_g_wgrad_tr = 1;          // tmf_set_option("wgrad_tr", 0 | 1 | 2)
int tmf_g_bf16_dma = 1;          // tmf_set_option("bf16_dma", 0 | 1): LDS-DMA form of the 8x8x8-brick bf16 forward (bf16 tensors)
int tmf_g_bf16_v2 = -1;          // tmf_set_option("bf16_v2", 0 | 1 | 2); -1 = not set yet: TMF_BF_V2 or 1
static bool use_v2(int B, int D, int H, int W) {
    if (tmf_g_bf16_v2 < 0) { const char* e = getenv("TMF_BF_V2"); tmf_g_bf16_v2 = e == nullptr ? 1 : atoi(e); }
    const int mode = tmf_g_bf16_v2;
    if (mode == 0) return false;
    if (mode == 2) return true;
    return (long)B * tmf_cdiv(D, v2::TD) * tmf_cdiv(H, v2::TH) * tmf_cdiv(W, v2::TW) >= 384;
}

#ifdef TMF_TRACE
extern "C" int tmf_debug_trace_read(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(tmf_trace_buf), bytes < sizeof(tmf_trace_buf) ? bytes : sizeof(tmf_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int tmf_conv3d_split_stat_blocks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);          // the split kernel always uses 4x8x8 bricks
}

extern "C" int tmf_conv3d_bf16_stat_blocks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (use_v2(B, D, H, W)) return B * tmf_cdiv(D, v2::TD) * tmf_cdiv(H, v2::TH) * tmf_cdiv(W, v2::TW);
    return B * tmf_cdiv(D, TD) * tmf_cdiv(H, TH) * tmf_cdiv(W, TW);
}

// kernel-trace name of the instance tmf_conv3d_fwd_bf16_t launches for a shape (measurement aid, as tmf_conv3d_fwd_kernel_name)
extern "C" const char* tmf_conv3d_fwd_bf16_kernel_name(int B, int D, int H, int W, int cin, int cout, int io) {
    static thread_local char buf[64];
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return "?";
    const char* i16 = (io & 1) ? "true" : "false";
    const char* o16 = (io & 2) ? "true" : "false";
    if (use_v2(B, D, H, W))
        snprintf(buf, sizeof buf, "conv3d_fwd_bf16_v2_kernel<%d, %s, %s, %s>", cout > 32 ? 2 : 1, i16, o16,
                 ((io & 1) && tmf_g_bf16_dma) ? "true" : "false");
    else {
        static const bool nt2 = [] { const char* e = getenv("TMF_BF_NT2"); return e == nullptr || atoi(e) != 0; }();
        snprintf(buf, sizeof buf, "conv3d_fwd_bf16_kernel<%d, %s, %s>", (nt2 && cout % 64 == 0) ? 2 : 1, i16, o16);
    }
    return buf;
}

// io: bit 0 = x is a bf16 tensor, bit 1 = z is a bf16 tensor
extern "C" int tmf_conv3d_fwd_bf16_t(const void* x, const void* w_bf16, void* z, float* stat_partial,
                                     int B, int D, int H, int W, int cin, int cout, int io, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w_bf16); TMF_REQUIRE_PTR(z);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_fwd_bf16: non-positive dimension");
    TMF_REQUIRE(cin % 8 == 0, TMF_E_SHAPE, "tmf_conv3d_fwd_bf16: cin=%d must be a multiple of 8", cin);
    TMF_REQUIRE(io >= 0 && io <= 3, TMF_E_ARG, "tmf_conv3d_fwd_bf16: io mode %d", io);
    TMF_REQUIRE(!(io & 2) || cout % 2 == 0, TMF_E_SHAPE, "tmf_conv3d_fwd_bf16: a bf16 output needs an even cout (%d)", cout);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 30), TMF_E_SHAPE,
                "tmf_conv3d_fwd_bf16: one sample exceeds 2^30 elements (32-bit byte offsets inside a sample)");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(w_bf16); TMF_REQUIRE_ALIGNED(z);
    int rc;
    hipStream_t s = (hipStream_t)stream;
    if (use_v2(B, D, H, W)) {
        const int tD = tmf_cdiv(D, v2::TD), tH = tmf_cdiv(H, v2::TH), tW = tmf_cdiv(W, v2::TW);
        const int ntiles = B * tD * tH * tW;
#define TMF_BF2_LAUNCH(NT, I16, O16, DMA_)                                                                           \
    do {                                                                                                             \
        auto k = conv3d_fwd_bf16_v2_kernel<NT, I16, O16, DMA_>;                                                      \
        const size_t ldsb = DMA_ ? v2::Cfg<NT>::LDS_BYTES_DMA : v2::Cfg<NT>::LDS_BYTES;                              \
        if ((rc = tmf_allow_lds(k, ldsb, "tmf_conv3d_fwd_bf16"))) return rc;                                         \
        hipLaunchKernelGGL(k, dim3(ntiles, tmf_cdiv(cout, 32 * NT)), dim3(v2::NTHR), ldsb, s, x,                      \
                           (const u16*)w_bf16, z, stat_partial, D, H, W, cin, cout, tD, tH, tW, ntiles, tmf_g_debug); \
    } while (0)
#define TMF_BF2_IO(NT)                                                                                               \
    do {                                                                                                             \
        if (io == 0) TMF_BF2_LAUNCH(NT, false, false, false);                                                        \
        else if (io == 1 && tmf_g_bf16_dma) TMF_BF2_LAUNCH(NT, true, false, true);                                   \
        else if (io == 1) TMF_BF2_LAUNCH(NT, true, false, false);                                                    \
        else if (io == 2) TMF_BF2_LAUNCH(NT, false, true, false);                                                    \
        else if (tmf_g_bf16_dma) TMF_BF2_LAUNCH(NT, true, true, true);                                               \
        else TMF_BF2_LAUNCH(NT, true, true, false);                                                                  \
    } while (0)
        if (cout > 32) TMF_BF2_IO(2);
        else TMF_BF2_IO(1);
#undef TMF_BF2_IO
#undef TMF_BF2_LAUNCH
        return tmf_launch_result("tmf_conv3d_fwd_bf16(v2)");
    }
    const int tD = tmf_cdiv(D, TD), tH = tmf_cdiv(H, TH), tW = tmf_cdiv(W, TW);
    const int ntiles = B * tD * tH * tW;
    // 64 output channels per workgroup where Cout allows it (TMF_BF_NT2=0 selects the 32-channel kernel everywhere): half
    // the workgroups (the per-workgroup overhead — offsets, LDS writes, 9 barriers per chunk, epilogue — is 40 % of this
    // kernel), half the halo loads, one A fragment per two MFMAs; both variants stay under 128 registers / 80 KB, two
    // workgroups per CU.  -20 % per launch (conv2.3 at 64^3, bf16 tensors: 0.395 -> 0.317 ms).
    static const bool nt2 = [] { const char* e = getenv("TMF_BF_NT2"); return e == nullptr || atoi(e) != 0; }();
    const bool two = nt2 && cout % 64 == 0;
#define TMF_BF_LAUNCH(I16, O16)                                                                                      \
    if (two) {                                                                                                       \
        auto k = conv3d_fwd_bf16_kernel<2, I16, O16>;                                                                \
        if ((rc = tmf_allow_lds(k, BfCfg<2>::LDS_BYTES, "tmf_conv3d_fwd_bf16"))) return rc;                          \
        hipLaunchKernelGGL(k, dim3(ntiles, cout / 64), dim3(NTHR), BfCfg<2>::LDS_BYTES, s, x,                        \
                           (const u16*)w_bf16, z, stat_partial, D, H, W, cin, cout, tD, tH, tW, ntiles);             \
    } else {                                                                                                         \
        auto k = conv3d_fwd_bf16_kernel<1, I16, O16>;                                                                \
        if ((rc = tmf_allow_lds(k, BfCfg<1>::LDS_BYTES, "tmf_conv3d_fwd_bf16"))) return rc;                          \
        hipLaunchKernelGGL(k, dim3(ntiles, tmf_cdiv(cout, 32)), dim3(NTHR), BfCfg<1>::LDS_BYTES, s, x,               \
                           (const u16*)w_bf16, z, stat_partial, D, H, W, cin, cout, tD, tH, tW, ntiles);             \
    }
    if (io == 0) TMF_BF_LAUNCH(false, false)
    else if (io == 1) TMF_BF_LAUNCH(true, false)
    else if (io == 2) TMF_BF_LAUNCH(false, true)
    else TMF_BF_LAUNCH(true, true)
#undef TMF_BF_LAUNCH
    return tmf_launch_result("tmf_conv3d_fwd_bf16");
}

extern "C" int tmf_conv3d_fwd_bf16(const float* x, const void* w_bf16, float* z, float* stat_partial,
                                   int B, int D, int H, int W, int cin, int cout, void* stream) {
    return tmf_conv3d_fwd_bf16_t(x, w_bf16, z, stat_partial, B, D, H, W, cin, cout, 0, stream);
}

extern "C" int tmf_conv3d_fwd_split(const float* x, const void* w3_bf16, float* z, float* stat_partial,
                                    int B, int D, int H, int W, int cin, int cout, void* stream) {
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(w3_bf16); TMF_REQUIRE_PTR(z);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_fwd_split: non-positive dimension");
    TMF_REQUIRE(cin % 8 == 0, TMF_E_SHAPE, "tmf_conv3d_fwd_split: cin=%d must be a multiple of 8", cin);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 31), TMF_E_SHAPE,
                "tmf_conv3d_fwd_split: one sample exceeds 2^31 elements");
    TMF_REQUIRE_ALIGNED(x); TMF_REQUIRE_ALIGNED(w3_bf16); TMF_REQUIRE_ALIGNED(z);
    const int tD = tmf_cdiv(D, TD), tH = tmf_cdiv(H, TH), tW = tmf_cdiv(W, TW);
    const int ntiles = B * tD * tH * tW;
    int rc;
    if ((rc = tmf_allow_lds(conv3d_fwd_split_kernel, SPLIT_LDS_BYTES, "tmf_conv3d_fwd_split"))) return rc;
    hipLaunchKernelGGL(conv3d_fwd_split_kernel, dim3(ntiles, tmf_cdiv(cout, 32)), dim3(NTHR), SPLIT_LDS_BYTES,
                       (hipStream_t)stream, x, (const u16*)w3_bf16, z, stat_partial, D, H, W, cin, cout,
                       tD, tH, tW, ntiles);
    return tmf_launch_result("tmf_conv3d_fwd_split");
}

// kernel-trace name of the instance tmf_conv3d_wgrad_bf16_t launches for a shape (measurement aid)
extern "C" const char* tmf_conv3d_wgrad_bf16_kernel_name(int B, int D, int H, int W, int cin, int cout, int io) {
    static thread_local char buf[64];
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return "?";
    const int nh = wgrad_tr_nh(cin, cout, io);
    if (nh) snprintf(buf, sizeof buf, "conv3d_wgrad_bf16_tr_kernel<%d, %s>", nh, io ? "true" : "false");
    else snprintf(buf, sizeof buf, "conv3d_wgrad_bf16_kernel<%s>", io ? "true" : "false");
    return buf;
}

extern "C" size_t tmf_conv3d_wgrad_bf16_workspace_bytes(int B, int D, int H, int W, int cin, int cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
    size_t need = 0;                                    // no tensor-type argument: enough for either kernel choice
    for (int io = 0; io < 2; ++io) {
        const WgBfPlan p = plan_wgrad_bf16(B, D, H, W, cin, cout, io);
        const size_t n = (size_t)(p.nsplit + tmf_reduce_groups(p.nsplit)) * 27 * cin * cout * 4;
        if (n > need) need = n;
    }
    return need;
}

extern "C" int tmf_conv3d_wgrad_bf16_t(const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes,
                                       int B, int D, int H, int W, int cin, int cout, int io, int dw_layout, void* stream) {
    TMF_REQUIRE(dw_layout == TMF_DW_TAPMAJOR || dw_layout == TMF_DW_REFERENCE, TMF_E_ARG,
                "tmf_conv3d_wgrad_bf16: unknown dw_layout %d", dw_layout);
    TMF_REQUIRE_PTR(x); TMF_REQUIRE_PTR(dz); TMF_REQUIRE_PTR(dw); TMF_REQUIRE_PTR(workspace);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, TMF_E_SHAPE,
                "tmf_conv3d_wgrad_bf16: non-positive dimension");
    TMF_REQUIRE(io == 0 || io == 1, TMF_E_ARG, "tmf_conv3d_wgrad_bf16: io mode %d (0 = float tensors, 1 = bf16 tensors)", io);
    TMF_REQUIRE(io == 0 || (cin % 2 == 0 && cout % 2 == 0), TMF_E_SHAPE,
                "tmf_conv3d_wgrad_bf16: bf16 tensors need even channel counts (cin=%d cout=%d)", cin, cout);
    TMF_REQUIRE((long)D * H * W * (cin > cout ? cin : cout) < (1L << 31), TMF_E_SHAPE,
                "tmf_conv3d_wgrad_bf16: one sample exceeds 2^31 elements");
    const size_t need = tmf_conv3d_wgrad_bf16_workspace_bytes(B, D, H, W, cin, cout);
    TMF_REQUIRE(workspace_bytes >= need, TMF_E_WORKSPACE, "tmf_conv3d_wgrad_bf16: workspace %zu B < required %zu B",
                workspace_bytes, need);
    const WgBfPlan p = plan_wgrad_bf16(B, D, H, W, cin, cout, io);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    float* partial = (float*)workspace;
#define TMF_WG_LAUNCH(KERNEL, LDSB)                                                                                      \
    do {                                                                                                                \
        if ((rc = tmf_allow_lds(KERNEL, LDSB, "tmf_conv3d_wgrad_bf16"))) return rc;                                     \
        hipLaunchKernelGGL(KERNEL, dim3(p.nsplit, p.gy, p.gz), dim3(NTHR), LDSB, s, x, dz, partial, D, H, W, cin, cout, \
                           p.tilesD, p.tilesH, p.tilesW, p.ntiles, p.tps);                                              \
    } while (0)
    if (p.nh == 2) {
        TMF_WG_LAUNCH((conv3d_wgrad_bf16_tr_kernel<2, true>), wtr::lds_bytes(2));
    } else if (p.nh == 1) {
        if (io == 0) TMF_WG_LAUNCH((conv3d_wgrad_bf16_tr_kernel<1, false>), wtr::lds_bytes(1));
        else TMF_WG_LAUNCH((conv3d_wgrad_bf16_tr_kernel<1, true>), wtr::lds_bytes(1));
    } else {
        if (io == 0) TMF_WG_LAUNCH(conv3d_wgrad_bf16_kernel<false>, WG_LDS_BYTES);
        else TMF_WG_LAUNCH(conv3d_wgrad_bf16_kernel<true>, WG_LDS_BYTES);
    }
#undef TMF_WG_LAUNCH
    if ((rc = tmf_launch_result("tmf_conv3d_wgrad_bf16"))) return rc;
    const long n = 27L * cin * cout;
    return tmf_reduce_slabs(partial, p.nsplit, n, partial + (size_t)p.nsplit * n, dw, s, "tmf_conv3d_wgrad_bf16(reduce)",
                            dw_layout == TMF_DW_REFERENCE ? cin : 0, cout);
}

extern "C" int tmf_conv3d_wgrad_bf16(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                                     int B, int D, int H, int W, int cin, int cout, void* stream) {
    return tmf_conv3d_wgrad_bf16_t(x, dz, dw, workspace, workspace_bytes, B, D, H, W, cin, cout, 0, TMF_DW_TAPMAJOR, stream);
}
