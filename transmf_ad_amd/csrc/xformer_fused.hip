// xformer_fused.hip — one Transformer(dim, depth=1) instance of the fusion block per launch.  gfx950, exact fp32.
//
// reference: models/networks.py:215-230 (Transformer.forward: x = attn(x, context) + x; x = ff(x) + x; norm(x)), built
// from PreNorm (:114-121; the context passes through UN-normalised), Attention (:141-175), FeedForward (:125-137) and the
// caller's "+ tokens" (:274-275).  The op-per-launch path (token_gemm.hip / attention.hip / token_ops.hip) needs 7
// launches forward and 13 backward per instance, each 5-20 us for ~1 us of matrix work on 1 728 token rows: the chain of a
// row tile is short, its launches are not.  Here a workgroup owns 16 token rows of ONE batch element and walks the whole
// chain with the tile in LDS:
//
//   xf_fwd_kernel     LN1 -> to_q -> attention over the context's K / V (one head per wavefront, scores of all keys in
//                     registers, exact softmax) -> to_out + bias (+ mask) + x -> LN2 -> Linear + bias -> GELU (+ mask)
//                     -> Linear + bias (+ mask) + x1 -> final LayerNorm + x -> to_kv of the NEXT instance (whose context
//                     this output is), stored row-major and transposed.
//   xf_bwd_q_kernel   the same chain backwards for a tile of QUERY rows: LNf', FF', LN2', to_out', the dQ half of the
//                     attention backward, to_q', LN1' + the two residual gradients; emits the per-tile column sums of every
//                     bias / LayerNorm parameter gradient and the dy operands of the weight gradients.
//   xf_bwd_kv_kernel  for a tile of KEY rows of the context: dK, dV over all queries of the batch element, then
//                     to_kv' + what the context tensor has collected so far.
//
// Batch elements never mix before the weight gradients, and a launch boundary costs less than a hand-rolled grid barrier
// (MI355X_MICROARCH.md: 1.5 us vs 4+ us), so the only cross-tile dependency — attention needs K / V (dK / dV: Q, dO) of the
// whole batch element — is a launch boundary.  The weight gradients dy^T x of ALL instances run as one table-driven launch
// at the end (tmf_tok_wgrad_multi).
//
// GEMMs: v_mfma_f32_16x16x4_f32, A = the 16-row tile from LDS (one ds_read_b128 per 4 MFMAs through the K-permutation
// k = 16 s + 4 (lane >> 4) + j).  Every B operand comes from memory that is laid out in FRAGMENT ORDER — the 64 lanes of a
// load instruction read 1 KB of consecutive bytes, lane l its own 16.  Measured on the first version (tools/xf_trace.py),
// which read nn.Linear weights / row-major K, V panels in place: there a lane's 16 bytes sit in a cache line of their own
// (rows are >= 512 bytes apart), such a gather costs the wave ~150 cycles where a lane-linear load costs ~30, and every
// GEMM phase ran at 2-4x its matrix time with the weights already in L2.  So
//   * the weights of an instance are re-packed once per step (xf_pack_kernel: [wave][k step][tile][lane][4], one pack for
//     the y = x W^T products of the forward, one for the dx = dy W products of the backward);
//   * the attention operands are WRITTEN in fragment order by their producers (a lane's accumulator registers are 4
//     consecutive tokens of one feature, resp. — across its interleaved tiles — 4 consecutive features of one token: one
//     16-byte store either way): per (batch element, head) a "row" fragment R[tile][s][lane][4] = X[token 16 tile + (lane
//     & 15)][feature 16 s + 4 (lane >> 4) + j] (operand of products that contract over features: S = K Q^T, dP = V dO^T,
//     and with Q / dO in that role for dK / dV) and a "column" fragment C[dt][tile][lane][4] = X[token 16 tile + 4 (lane
//     >> 4) + j][feature 16 dt + (lane & 15)] (products that contract over tokens: O^T = V^T P^T, dQ^T = K^T dS^T, dV^T =
//     dO^T P, dK^T = Q^T dS).  Rows past N are exact zeros.
// A lane's accumulator tiles are INTERLEAVED columns (tile u of a lane with m = lane & 15 is column G m + u of its group),
// so a lane owns G consecutive columns of a row: wide row-major stores.
//
// Workgroup -> XCD: all tiles of a batch element run on one XCD (xf_who), whose L2 then holds that element's panels and
// one copy of the weights; the cold part of a launch's working set (weights: written by the optimizer; in backward the
// saved panels) is requested cooperatively at kernel start (xf_prefetch).
//
// Dropout (options/option.py:39 `--dropout`; networks.py:131,133,153): the three keep-masks of an instance (after to_out,
// after GELU, after the second Linear), already scaled by 1 / (1 - p), are INPUTS (NULL = inactive) — the random draw
// stays with the caller, as in heads.hip.
#include "tmf_common.h"

namespace {

constexpr int XT = 16;             // token rows per workgroup
constexpr int XD = 128;            // model dim = heads * dim_head
constexpr int XH = 4;              // heads = wavefronts per workgroup
constexpr int XDH = 32;            // dim_head
constexpr int XP = XD + 4;         // LDS row pitch (floats) of a 128-wide tile
constexpr int XTHR = 256;
constexpr float XLOG2E = 1.4426950408889634f;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// phase time stamps of every wave (shader clock) for tools/xf_trace.py: only in builds with -DTMF_XF_TRACE
// (TMF_EXTRA_FLAGS=-DTMF_XF_TRACE python -m transmf_ad_amd.build); the product kernels carry none of it
#ifdef TMF_XF_TRACE
#define XF_STAMP(k)                                                                                                   \
    do {                                                                                                              \
        if (p.trace != nullptr && lane == 0)                                                                          \
            p.trace[(((size_t)bz * p.tiles + tile) * 4 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define XF_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float xgelu(float h) { return 0.5f * h * (1.f + erff(h * 0.70710678118654752f)); }
__device__ __forceinline__ float xgelu_grad(float h) {
    return 0.5f * (1.f + erff(h * 0.70710678118654752f)) + h * 0.3989422804014327f * expf(-0.5f * h * h);
}
__device__ __forceinline__ float xhalf_sum(float v) {        // sum over the 32 lanes of a half-wave
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

// Columns of a lane's accumulator tiles: groups of G = min(TPW, 4) interleaved tiles, 16 G columns per group.
template <int TPW> struct TileMap {
    static constexpr int G = TPW < 4 ? TPW : 4;
    static constexpr int NG = TPW / G;
    static __device__ __forceinline__ int col(int cbase, int u, int m) { return cbase + (u / G) * (16 * G) + G * m + (u % G); }
    static __device__ __forceinline__ int col0(int cbase, int gq, int m) { return cbase + gq * (16 * G) + G * m; }
};

// acc[u] += A[16 x 16 KS] . B for the lane's TPW column tiles; B from a weight pack in fragment order
// P[wave][s][u][lane][j] = B[k = 16 s + 4 (lane >> 4) + j][TileMap<TPW>::col(16 TPW wave, u, lane & 15)]  (xf_pack_kernel).
// A tile in LDS, row pitch KP floats.  B operands of PF steps are in flight ahead of the MFMAs (register ring).
// prime() issues the first PF steps' loads and is called BEFORE the previous phase's epilogue: on gfx9 stores count in
// vmcnt like loads and retire in order, so operand loads issued behind an epilogue's 10-30 global stores would wait
// for every one of them.
template <int TPW, int KS> struct Gemm {
    static constexpr int PF = (TPW <= 2) ? 4 : 2;
    f32x4 b[PF + 1][TPW];
    const float* src;
    __device__ __forceinline__ void load(int s, f32x4 (&dst)[TPW]) {
#pragma unroll
        for (int u = 0; u < TPW; ++u) dst[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s * TPW + u) * 256);
    }
    __device__ __forceinline__ void prime(const float* __restrict__ P, int wave, int lane) {
        src = P + ((size_t)wave * KS * TPW * 64 + lane) * 4;
#pragma unroll
        for (int s = 0; s < PF && s < KS; ++s) load(s, b[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[TPW], const float* As, int KP, int lane) {
        const float* arow = As + (lane & 15) * KP + 4 * (lane >> 4);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + PF < KS) {
                load(s + PF, b[(s + PF) % (PF + 1)]);
                __builtin_amdgcn_sched_barrier(0);       // keep the request AHEAD of this step's MFMAs (the scheduler sinks it otherwise)
            }
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < TPW; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b[s % (PF + 1)][u][j], acc[u], 0, 0, 0);
        }
    }
};

template <int TPW>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TPW]) {
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// column sums over the 16 rows of an LDS tile -> dst[0 .. ncols)
__device__ __forceinline__ void tile_colsum(const float* T, int pitch, int ncols, float* __restrict__ dst, int tid) {
    for (int c = tid; c < ncols; c += XTHR) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < XT; ++r) s += T[r * pitch + c];
        dst[c] = s;
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Fragment-order panels of the attention operands (floats; add 4 * lane):  T = tiles of 16 tokens
__device__ __forceinline__ size_t frag_r(int b, int h, int t, int s, int T) { return ((((size_t)(b * XH + h) * T + t) * 2 + s) * 64) * 4; }
__device__ __forceinline__ size_t frag_c(int b, int h, int dt, int t, int T) { return ((((size_t)(b * XH + h) * 2 + dt) * T + t) * 64) * 4; }

// Producer side, TPW = 2 GEMM whose wave = head (to_q, to_out'): acc[u][r] = X[token 4 kb + r][feature 32 h + 2 m + u]
__device__ __forceinline__ void store_frag_head(float* __restrict__ XR, float* __restrict__ XC, const f32x4 (&acc)[2], int b, int h,
                                                int tile, int T, int m, int kb) {
#pragma unroll
    for (int u = 0; u < 2; ++u)             // column fragment: 4 consecutive tokens of feature 2 m + u
        st4(XC + frag_c(b, h, m >> 3, tile, T) + (kb * 16 + 2 * (m & 7) + u) * 4, acc[u]);
    float* dst = XR + frag_r(b, h, tile, m >> 3, T) + (((m & 7) >> 1) * 16 + 4 * kb) * 4 + 2 * (m & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r)             // row fragment: features 2 m, 2 m + 1 of token 4 kb + r
        *reinterpret_cast<f32x2*>(dst + r * 4) = f32x2{acc[0][r], acc[1][r]};
}
// Producer side, the TPW = 4 to_kv GEMM: wave w holds features 64 w + 4 m + u of [K | V]: waves 0, 1 -> K, 2, 3 -> V
__device__ __forceinline__ void store_frag_kv(float* __restrict__ KR, float* __restrict__ KC, float* __restrict__ VR,
                                              float* __restrict__ VC, const f32x4 (&acc)[4], int b, int w, int tile, int T, int m, int kb) {
    float* XR = w < 2 ? KR : VR;
    float* XC = w < 2 ? KC : VC;
    const int h = 2 * (w & 1) + (m >> 3), sd = (m & 7) >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) st4(XC + frag_c(b, h, sd, tile, T) + (kb * 16 + 4 * (m & 3) + u) * 4, acc[u]);
    float* dst = XR + frag_r(b, h, tile, sd, T) + ((m & 3) * 16 + 4 * kb) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) st4(dst + r * 4, f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]});
}

// Workgroup id -> (batch element, tile).  The grid is 1-D, 8 * ns workgroups with ns = tiles * ceil(B / 8): workgroup id
// runs on XCD id % 8 (observed dispatch order; a speed matter only), and ALL tiles of a batch element are given to one
// XCD — its private L2 then holds that element's K / V / Q panels (read by every tile) and one copy of the instance's
// weights, instead of every XCD pulling every panel.  `slot` numbers the workgroups of an XCD (0 .. ns - 1).
struct XfWho { int bz, tile, slot, ns; bool live; };
__device__ __forceinline__ XfWho xf_who(int B, int tiles) {
    XfWho w;
    const int id = blockIdx.x, xcd = id & 7;
    w.slot = id >> 3;
    w.ns = tiles * ((B + 7) >> 3);
    w.bz = xcd + 8 * (w.slot / tiles);
    w.tile = w.slot % tiles;
    w.live = w.bz < B;
    return w;
}

// Cooperative L2 warm-up.  What a launch reads that its XCD has not touched recently (the instance's weights: written by
// the optimizer; in backward the saved panels of the batch element) comes from the Infinity Cache / HBM at ~2 us per
// dependent round trip, and the GEMM operand rings look only a few hundred ns ahead: measured (tools/xf_trace.py), every
// GEMM phase ran at 3-4x its matrix time, one cold round trip per ring refill.  So the ns workgroups of an XCD first
// request one slice each of everything the launch will read (16-byte loads into registers that are only "used" by an
// empty asm later): the cold latency is paid ONCE, overlapped with the tile's own prologue, and the phases then hit L2.
constexpr int XPF = 20;                          // 16-byte pieces per thread (80 KB per workgroup)
struct XfRegion { const float* p; int n4; };     // n4 = float4 pieces
template <int NR>
__device__ __forceinline__ void xf_prefetch(const XfRegion (&reg)[NR], int slot, int ns, int tid, f32x4 (&pf)[XPF]) {
    int n4[NR];
    long total = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { n4[r] = reg[r].p != nullptr ? reg[r].n4 : 0; total += n4[r]; }
#pragma unroll
    for (int k = 0; k < XPF; ++k) pf[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (total == 0) return;                              // (uniform)
    const long lo = total * slot / ns, hi = total * (slot + 1) / ns;
#pragma unroll
    for (int k = 0; k < XPF; ++k) {
        // always a load (no divergent branches: the waits around them stay exact): pieces past the slice re-read its last one
        long g = lo + tid + (long)XTHR * k;
        g = g < hi ? g : hi - 1;
        g = g < lo ? lo : g;
        const float* base = reg[0].p;
        bool done = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool here = !done && g < n4[r];
            base = here ? reg[r].p : base;
            done = done || here;
            g = done ? g : g - n4[r];
        }
        pf[k] = ld4(base + 4 * g);
    }
}
__device__ __forceinline__ void xf_prefetch_done(f32x4 (&pf)[XPF]) {
#pragma unroll
    for (int k = 0; k < XPF; ++k) asm volatile("" :: "v"(pf[k]));
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct XfFwdArgs {
    const float* x;                 // [B*N][128] input tokens of the instance (only_kv: the context tokens)
    const float *KR, *VC;           // K row fragments / V column fragments of the context
    const float *ln1_g, *ln1_b, *bo, *ln2_g, *ln2_b, *b1, *b2, *lnf_g, *lnf_b;
    const float *pq, *po, *p1, *p2; // weight packs (y = x W^T form) of to_q, to_out, the two FeedForward Linears
    const float* pkv_next;          // pack of the to_kv weight of the instance that takes this output as context (NULL: none)
    const float *mask_o, *mask_g, *mask_f;       // scaled Dropout keep-masks [R][128], [R][mlp], [R][128] or NULL
    float eps1, eps2, epsf, scale;
    float *a, *QR, *QC, *out, *lse, *x1, *f, *h, *g, *x2, *y, *m1, *r1, *m2, *r2, *mf, *rf;     // saved for backward
    float *KRn, *KCn, *VRn, *VCn;   // K | V fragments of the next instance
    int B, N, Npad, tiles, only_kv;
    unsigned long long* trace;
};

constexpr int XMLP = 512;
constexpr int XGP = XMLP + 4;

// LayerNorm of a 128-wide row held as 4 values per lane of a half-wave (lane li owns columns 4 li .. 4 li + 3)
__device__ __forceinline__ f32x4 ln_row(const f32x4 v, const float* __restrict__ g, const float* __restrict__ b, float eps,
                                        int li, float& mu, float& rs) {
    mu = xhalf_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 128.f);
    const f32x4 d = {v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};
    const float var = xhalf_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 128.f);
    rs = 1.f / sqrtf(var + eps);
    const f32x4 gg = ld4(g + li * 4), bb = ld4(b + li * 4);
    return f32x4{d[0] * rs * gg[0] + bb[0], d[1] * rs * gg[1] + bb[1], d[2] * rs * gg[2] + bb[2], d[3] * rs * gg[3] + bb[3]};
}

// MT: key tiles held in registers (16 MT >= N).  EXACT: the launch has exactly MT key tiles — the attention loops then
// carry no run-time guards; guarded loops make the compiler drain every outstanding load at each branch join, which
// turned the operand rings into one L2 round trip per tile (measured: 890 cycles per 16-key tile against 320 of MFMA).
// H2 (round 6): EIGHT heads of 16 features (train_adversarial.py:30-31) — a wave owns the two heads 2 wave, 2 wave + 1, which are
// exactly the two 16-feature blocks s = 0, 1 of its 32 features: the fragment panels are the same memory (head' = 2 h + s), only
// the contractions change (a head's scores sum ONE block, its output is ONE dt tile), lse / delta are [B][8][Npad].
template <int MT, bool EXACT, bool H2>
__global__ __launch_bounds__(XTHR, 1) void xf_fwd_kernel(const XfFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // input rows
    float* As = Xs + XT * XP;               // LN1(x), later LN2(x1)
    float* Qs = As + XT * XP;               // q
    float* Os = Qs + XT * XP;               // attention output
    float* X1s = Os + XT * XP;
    float* Ys = X1s + XT * XP;              // x2, later y
    float* Gs = Ys + XT * XP;               // [16][XGP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4, li = lane & 31;
    const XfWho who = xf_who(p.B, p.tiles);
    if (!who.live) return;
    const int tile = who.tile, bz = who.bz;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;                 // valid rows of this tile
    const size_t row0 = (size_t)bz * N + t0;
    const int ntiles = EXACT ? MT : (Npad >> 4);

    XF_STAMP(0);
    // ---- P0: x tile (requested first), L2 warm-up of the instance's weights, LayerNorm 1 ----
    f32x4 xin[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        xin[pass] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < nv) xin[pass] = ld4(p.x + (row0 + row) * XD + li * 4);
    }
    f32x4 pf[XPF];
    {
        const XfRegion reg[5] = {{p.only_kv ? nullptr : p.pq, XD * XD / 4}, {p.only_kv ? nullptr : p.po, XD * XD / 4},
                                 {p.only_kv ? nullptr : p.p1, XMLP * XD / 4}, {p.only_kv ? nullptr : p.p2, XMLP * XD / 4},
                                 {p.pkv_next, 2 * XD * XD / 4}};
        xf_prefetch<5>(reg, who.slot, who.ns, tid, pf);
    }
    Gemm<2, 8> gq;
    if (!p.only_kv) gq.prime(p.pq, wave, lane);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        const f32x4 v = xin[pass];
        if (p.only_kv) {
            st4(Ys + row * XP + li * 4, v);
        } else {
            st4(Xs + row * XP + li * 4, v);
            float mu, rs;
            f32x4 a = ln_row(v, p.ln1_g, p.ln1_b, p.eps1, li, mu, rs);
            if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(As + row * XP + li * 4, a);
            if (ok) {
                st4(p.a + gr * XD + li * 4, a);
                if (li == 0) { p.m1[gr] = mu; p.r1[gr] = rs; }
            }
        }
    }
    __syncthreads();
    xf_prefetch_done(pf);

    XF_STAMP(1);
    Gemm<4, 8> gkv;
    if (!p.only_kv) {
        // ---- P1: q = LN1(x) Wq^T ----
        constexpr int RD = 4;                    // key tiles in flight (register ring, static indices: t is unrolled)
        const int h = wave;
        const float* Kb = p.KR + frag_r(bz, h, 0, 0, ntiles) + lane * 4;       // + 512 floats per tile, 256 per s
        const float* Vt = p.VC + frag_c(bz, h, 0, 0, ntiles) + lane * 4;       // + 256 floats per tile, 256 * tiles per dt
        f32x4 kr[RD][2], vr[RD][2];
        auto load_k = [&](int t, f32x4 (&dst)[2]) {
            dst[0] = ld4(Kb + t * 512);
            dst[1] = ld4(Kb + t * 512 + 256);
        };
        auto load_v = [&](int t, f32x4 (&dst)[2]) {
            dst[0] = ld4(Vt + t * 256);
            dst[1] = ld4(Vt + (size_t)(ntiles + t) * 256);
        };
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            gq.run(acc, As, XP, lane);
            // the attention's first key tiles are requested ahead of this epilogue's stores
            if constexpr (!H2) {
#pragma unroll
                for (int t = 0; t < RD - 1; ++t) {
                    kr[t][0] = kr[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EXACT || t < ntiles) load_k(t < ntiles ? t : 0, kr[t]);
                }
            }
            const int c0 = 32 * wave + 2 * m;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x2*>(Qs + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
            store_frag_head(p.QR, p.QC, acc, bz, wave, tile, p.tiles, m, kb);     // rows past nv are exact zeros (their A rows are)
        }
        __syncthreads();

        XF_STAMP(2);
        // ---- P2: attention, head = wave.  S^T[key][query] = K Q^T: a lane holds 4 keys per 16-key tile of ONE query ----
        Gemm<2, 8> go;
        if constexpr (H2) {
            // two heads of 16 features, one after the other (the scores of ONE head's keys in registers at a time)
            const float c = p.scale * XLOG2E;
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                f32x4 k1[RD], v1[RD];
                auto load_k1 = [&](int t, f32x4& dst) { dst = ld4(Kb + t * 512 + hs * 256); };
                auto load_v1 = [&](int t, f32x4& dst) { dst = ld4(Vt + (size_t)(hs * ntiles + t) * 256); };
#pragma unroll
                for (int t = 0; t < RD - 1; ++t) {
                    k1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EXACT || t < ntiles) load_k1(t < ntiles ? t : 0, k1[t]);
                }
                float qreg[4];
                {
                    const f32x4 v = ld4(Qs + m * XP + XDH * h + 16 * hs + 4 * kb);
                    qreg[0] = v[0] * c; qreg[1] = v[1] * c; qreg[2] = v[2] * c; qreg[3] = v[3] * c;
                }
                f32x4 sT[MT];
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    if (EXACT || t < ntiles) {
                        if (t + RD - 1 < MT && (EXACT || t + RD - 1 < ntiles)) {
                            load_k1(t + RD - 1, k1[(t + RD - 1) % RD]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const f32x4 k0 = k1[t % RD];
                        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 4; ++j) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[j], qreg[j], sc, 0, 0, 0);
                        sT[t] = sc;
                    } else {
                        sT[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                }
#pragma unroll
                for (int t = 0; t < RD - 1; ++t) {
                    v1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EXACT || t < ntiles) load_v1(t < ntiles ? t : 0, v1[t]);
                }
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * t + 4 * kb + r;
                        sT[t][r] = key < N ? sT[t][r] : -INFINITY;
                        mx = fmaxf(mx, sT[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = exp2f(sT[t][r] - mx);
                        sT[t][r] = e;
                        l += e;
                    }
                }
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    if (EXACT || t < ntiles) {
                        if (t + RD - 1 < MT && (EXACT || t + RD - 1 < ntiles)) {
                            load_v1(t + RD - 1, v1[(t + RD - 1) % RD]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const f32x4 v0 = v1[t % RD];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[r], sT[t][r], o, 0, 0, 0);
                    }
                }
                if (hs == 1) go.prime(p.po, wave, lane);
                const float inv = 1.f / l;
                const f32x4 w4 = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                const int cc = XDH * h + 16 * hs + 4 * kb;
                st4(Os + m * XP + cc, w4);
                if (m < nv) st4(p.out + (row0 + m) * XD + cc, w4);
                if (kb == 0) p.lse[((size_t)bz * (2 * XH) + 2 * h + hs) * Npad + t0 + m] = m < nv ? mx + log2f(l) : INFINITY;
            }
        } else {
            const float c = p.scale * XLOG2E;
            float qreg[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 v = ld4(Qs + m * XP + XDH * h + 16 * s + 4 * kb);
                qreg[s][0] = v[0] * c; qreg[s][1] = v[1] * c; qreg[s][2] = v[2] * c; qreg[s][3] = v[3] * c;
            }
            f32x4 sT[MT];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (EXACT || t < ntiles) {
                    if (t + RD - 1 < MT && (EXACT || t + RD - 1 < ntiles)) {
                        load_k(t + RD - 1, kr[(t + RD - 1) % RD]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4 k0 = kr[t % RD][0], k1 = kr[t % RD][1];
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[j], qreg[0][j], s, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[j], qreg[1][j], s, 0, 0, 0);
                    sT[t] = s;
                } else {
                    sT[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
            XF_STAMP(3);
            // first V tiles requested before the softmax arithmetic
#pragma unroll
            for (int t = 0; t < RD - 1; ++t) {
                vr[t][0] = vr[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (EXACT || t < ntiles) load_v(t < ntiles ? t : 0, vr[t]);
            }
            // keys past N (only in the last tile) do not take part
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * t + 4 * kb + r;
                    sT[t][r] = key < N ? sT[t][r] : -INFINITY;
                    mx = fmaxf(mx, sT[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(sT[t][r] - mx);
                    sT[t][r] = e;
                    l += e;
                }
            }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            XF_STAMP(4);
            // O^T[d][query] = V^T P^T:  A = V column fragments (4 consecutive keys of feature d per lane), B = P (own registers)
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (EXACT || t < ntiles) {
                    if (t + RD - 1 < MT && (EXACT || t + RD - 1 < ntiles)) {
                        load_v(t + RD - 1, vr[(t + RD - 1) % RD]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4 v0 = vr[t % RD][0], v1 = vr[t % RD][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[r], sT[t][r], o[0], 0, 0, 0);
                        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[r], sT[t][r], o[1], 0, 0, 0);
                    }
                }
            }
            go.prime(p.po, wave, lane);
            // lane (m = query, kb): o[dt][r] = O[query m][feature 32 h + 16 dt + 4 kb + r]
            const float inv = 1.f / l;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4 w4 = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                const int cc = XDH * h + 16 * dt + 4 * kb;
                st4(Os + m * XP + cc, w4);
                if (m < nv) st4(p.out + (row0 + m) * XD + cc, w4);
            }
            if (kb == 0) p.lse[((size_t)bz * XH + h) * Npad + t0 + m] = m < nv ? mx + log2f(l) : INFINITY;
        }
        __syncthreads();

        XF_STAMP(5);
        // ---- P4: x1 = mask_o (out Wo^T + bo) + x ----
        Gemm<8, 8> g1;
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            go.run(acc, Os, XP, lane);
            g1.prime(p.p1, wave, lane);
            const int c0 = 32 * wave + 2 * m;
            const f32x2 bo = *reinterpret_cast<const f32x2*>(p.bo + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                const f32x2 xr = *reinterpret_cast<const f32x2*>(Xs + row * XP + c0);
                f32x2 v = {acc[0][r] + bo[0], acc[1][r] + bo[1]};
                if (p.mask_o != nullptr && ok) {
                    const f32x2 mk = *reinterpret_cast<const f32x2*>(p.mask_o + (row0 + row) * XD + c0);
                    v[0] *= mk[0]; v[1] *= mk[1];
                }
                v[0] += xr[0]; v[1] += xr[1];
                *reinterpret_cast<f32x2*>(X1s + row * XP + c0) = v;
                if (ok) *reinterpret_cast<f32x2*>(p.x1 + (row0 + row) * XD + c0) = v;
            }
        }
        __syncthreads();

        XF_STAMP(6);
        // ---- LayerNorm 2 ----
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 8 + wave * 2 + (lane >> 5);
            const bool ok = row < nv;
            const size_t gr = row0 + row;
            const f32x4 v = ld4(X1s + row * XP + li * 4);
            float mu, rs;
            f32x4 f = ln_row(v, p.ln2_g, p.ln2_b, p.eps2, li, mu, rs);
            if (!ok) f = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(As + row * XP + li * 4, f);
            if (ok) {
                st4(p.f + gr * XD + li * 4, f);
                if (li == 0) { p.m2[gr] = mu; p.r2[gr] = rs; }
            }
        }
        __syncthreads();

        XF_STAMP(7);
        // ---- P5: h = LN2(x1) W1^T + b1;  g = mask_g GELU(h) ----
        Gemm<2, 32> g2;
        {
            f32x4 acc[8];
            zero_acc<8>(acc);
            g1.run(acc, As, XP, lane);
            g2.prime(p.p2, wave, lane);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int c0 = 128 * wave + 64 * gq + 4 * m;
                const f32x4 b1 = ld4(p.b1 + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kb + r;
                    const bool ok = row < nv;
                    const f32x4 hv = {acc[gq * 4 + 0][r] + b1[0], acc[gq * 4 + 1][r] + b1[1], acc[gq * 4 + 2][r] + b1[2],
                                      acc[gq * 4 + 3][r] + b1[3]};
                    f32x4 gv = {xgelu(hv[0]), xgelu(hv[1]), xgelu(hv[2]), xgelu(hv[3])};
                    if (p.mask_g != nullptr && ok) {
                        const f32x4 mk = ld4(p.mask_g + (row0 + row) * XMLP + c0);
                        gv[0] *= mk[0]; gv[1] *= mk[1]; gv[2] *= mk[2]; gv[3] *= mk[3];
                    }
                    if (!ok) gv = f32x4{0.f, 0.f, 0.f, 0.f};
                    st4(Gs + row * XGP + c0, gv);
                    if (ok) {
                        st4(p.h + (row0 + row) * XMLP + c0, hv);
                        st4(p.g + (row0 + row) * XMLP + c0, gv);
                    }
                }
            }
        }
        __syncthreads();

        XF_STAMP(8);
        // ---- P6: x2 = mask_f (g W2^T + b2) + x1 ----
        {
            f32x4 acc[2];
            zero_acc<2>(acc);
            g2.run(acc, Gs, XGP, lane);
            if (p.pkv_next != nullptr) gkv.prime(p.pkv_next, wave, lane);
            const int c0 = 32 * wave + 2 * m;
            const f32x2 b2 = *reinterpret_cast<const f32x2*>(p.b2 + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                const f32x2 xr = *reinterpret_cast<const f32x2*>(X1s + row * XP + c0);
                f32x2 v = {acc[0][r] + b2[0], acc[1][r] + b2[1]};
                if (p.mask_f != nullptr && ok) {
                    const f32x2 mk = *reinterpret_cast<const f32x2*>(p.mask_f + (row0 + row) * XD + c0);
                    v[0] *= mk[0]; v[1] *= mk[1];
                }
                v[0] += xr[0]; v[1] += xr[1];
                *reinterpret_cast<f32x2*>(Ys + row * XP + c0) = v;
                if (ok) *reinterpret_cast<f32x2*>(p.x2 + (row0 + row) * XD + c0) = v;
            }
        }
        __syncthreads();

        XF_STAMP(9);
        // ---- final LayerNorm + the caller's "+ tokens" ----
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 8 + wave * 2 + (lane >> 5);
            const bool ok = row < nv;
            const size_t gr = row0 + row;
            const f32x4 v = ld4(Ys + row * XP + li * 4);
            const f32x4 xr = ld4(Xs + row * XP + li * 4);
            float mu, rs;
            f32x4 y = ln_row(v, p.lnf_g, p.lnf_b, p.epsf, li, mu, rs);
            y[0] += xr[0]; y[1] += xr[1]; y[2] += xr[2]; y[3] += xr[3];
            if (!ok) y = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(Ys + row * XP + li * 4, y);             // the same lanes read and write this row
            if (ok) {
                st4(p.y + gr * XD + li * 4, y);
                if (li == 0) { p.mf[gr] = mu; p.rf[gr] = rs; }
            }
        }
        __syncthreads();
    } else if (p.pkv_next != nullptr) {
        gkv.prime(p.pkv_next, wave, lane);
    }

    XF_STAMP(10);
    // ---- P7: K | V of the next instance (its context = this output), in fragment order ----
    if (p.pkv_next != nullptr) {
        f32x4 acc[4];
        zero_acc<4>(acc);
        gkv.run(acc, Ys, XP, lane);
        store_frag_kv(p.KRn, p.KCn, p.VRn, p.VCn, acc, bz, wave, tile, p.tiles, m, kb);       // rows past nv: zeros
    }
    XF_STAMP(11);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, query side
// ---------------------------------------------------------------------------------------------------------------------
struct XfBwdQArgs {
    const float* dy;                // [R][128] gradient w.r.t. the instance's output
    const float* x;                 // the instance's input tokens
    const float *KR, *KC, *VR;      // fragments of the context's K (row, column) and V (row)
    const float *ln1_g, *ln2_g, *lnf_g;
    const float *b2, *b1, *bo, *bq; // weight packs (dx = dy W form) of Linear 2, Linear 1, to_out, to_q
    const float *mask_o, *mask_g, *mask_f;
    float scale;
    const float *QR, *out, *lse, *x1, *h, *x2, *m1, *r1, *m2, *r2, *mf, *rf;      // saved by the forward
    float *dx2, *dh, *dx1, *dq;     // dy operands of the weight gradients ([R][128], [R][mlp], [R][128], [R][128])
    float *DR, *DC, *delta;         // for the key-side kernel: dO fragments (row, column); delta [B][heads][Npad]
    float* dx;                      // [R][128] gradient w.r.t. the instance's input tokens
    float* part;                    // [B*tiles][stride]: b2 | b1 | bo | ln2 g | ln2 b | ln1 g | ln1 b | lnf g | lnf b
    int stride;
    int B, N, Npad, tiles;
    unsigned long long* trace;
};

template <bool H2>                      // H2: eight heads of 16 (see xf_fwd_kernel)
__global__ __launch_bounds__(XTHR, 1) void xf_bwd_q_kernel(const XfBwdQArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DY = smem;                       // dy
    float* T1 = DY + XT * XP;               // dy * xhat products for the column sums
    float* DX2 = T1 + XT * XP;              // gradient w.r.t. x2 (residual path)
    float* DX2M = DX2 + XT * XP;            // ... times mask_f (the Linear's output gradient)
    float* DF = DX2M + XT * XP;             // gradient w.r.t. LN2's output
    float* DX1 = DF + XT * XP;              // gradient w.r.t. x1
    float* DX1M = DX1 + XT * XP;            // ... times mask_o
    float* DO = DX1M + XT * XP;             // gradient w.r.t. the attention output
    float* DQ = DO + XT * XP;
    float* DA = DQ + XT * XP;               // gradient w.r.t. LN1's output
    float* DH = DA + XT * XP;               // [16][XGP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4, li = lane & 31;
    const XfWho who = xf_who(p.B, p.tiles);
    if (!who.live) return;
    const int tile = who.tile, bz = who.bz;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;
    const size_t row0 = (size_t)bz * N + t0;
    float* part = p.part + ((size_t)bz * p.tiles + tile) * p.stride;
    const int o_b2 = 0, o_b1 = XD, o_bo = XD + XMLP, o_ln2 = 2 * XD + XMLP, o_ln1 = 4 * XD + XMLP, o_lnf = 6 * XD + XMLP;

    XF_STAMP(0);
    // ---- everything the tile reads from the forward pass is requested NOW (cold: written a whole encoder backward ago),
    //      in the register layout of the phase that uses it; then the L2 warm-up of the weights and the K / V panels ----
    f32x4 dy_r[2], x2_r[2], x1_r[2], x_r[2];
    float mf_r[2], rf_r[2], m2_r[2], r2_r[2], m1_r[2], r1_r[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const size_t gr = row0 + row;
        dy_r[pass] = x2_r[pass] = x1_r[pass] = x_r[pass] = f32x4{0.f, 0.f, 0.f, 0.f};
        mf_r[pass] = rf_r[pass] = m2_r[pass] = r2_r[pass] = m1_r[pass] = r1_r[pass] = 0.f;
        if (row < nv) {
            dy_r[pass] = ld4(p.dy + gr * XD + li * 4); x2_r[pass] = ld4(p.x2 + gr * XD + li * 4);
            x1_r[pass] = ld4(p.x1 + gr * XD + li * 4); x_r[pass] = ld4(p.x + gr * XD + li * 4);
            mf_r[pass] = p.mf[gr]; rf_r[pass] = p.rf[gr]; m2_r[pass] = p.m2[gr]; r2_r[pass] = p.r2[gr];
            m1_r[pass] = p.m1[gr]; r1_r[pass] = p.r1[gr];
        }
    }
    f32x4 h_r[2][4];                        // GELU pre-activations in the layout of S2's epilogue
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r;
            h_r[gq][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < nv) h_r[gq][r] = ld4(p.h + (row0 + row) * XMLP + 128 * wave + 64 * gq + 4 * m);
        }
    f32x4 q_r[2], o_r[2];                   // this lane's query / attention-output features (head = wave) for S6
    {
        const size_t grm = row0 + (m < nv ? m : nv - 1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            q_r[s2] = ld4(p.QR + frag_r(bz, wave, tile, s2, p.tiles) + lane * 4);
            o_r[s2] = ld4(p.out + grm * XD + XDH * wave + 16 * s2 + 4 * kb);
        }
    }
    f32x4 pf[XPF];
    {
        const int panel = Npad * XD / 4;            // one fragment panel of a batch element: [heads][tiles][2][64] float4
        const XfRegion reg[7] = {{p.b2, XMLP * XD / 4}, {p.b1, XMLP * XD / 4}, {p.bo, XD * XD / 4}, {p.bq, XD * XD / 4},
                                 {p.KR + (size_t)bz * Npad * XD, panel}, {p.VR + (size_t)bz * Npad * XD, panel},
                                 {p.KC + (size_t)bz * Npad * XD, panel}};
        xf_prefetch<7>(reg, who.tile, p.tiles, tid, pf);
    }
    Gemm<8, 8> g2;
    g2.prime(p.b2, wave, lane);             // S2's first operands: ahead of S1's stores

    // ---- S1: final LayerNorm backward (row-wise) ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        const f32x4 dyv = dy_r[pass], xv = x2_r[pass];
        const float mu = mf_r[pass], rs = rf_r[pass];
        const f32x4 gam = ld4(p.lnf_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dyv[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, dm, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2); px[c] = dyv[c] * xh[c]; }
        dm = d;
        if (p.mask_f != nullptr && ok) {
            const f32x4 mk = ld4(p.mask_f + gr * XD + li * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dm[c] *= mk[c];
        }
        st4(DY + row * XP + li * 4, dyv);
        st4(T1 + row * XP + li * 4, px);
        st4(DX2 + row * XP + li * 4, d);
        st4(DX2M + row * XP + li * 4, dm);
        if (ok) st4(p.dx2 + gr * XD + li * 4, dm);
    }
    __syncthreads();
    xf_prefetch_done(pf);
    tile_colsum(T1, XP, XD, part + o_lnf, tid);
    tile_colsum(DY, XP, XD, part + o_lnf + XD, tid);
    tile_colsum(DX2M, XP, XD, part + o_b2, tid);

    XF_STAMP(1);
    // ---- S2: dg = dx2m W2;  dh = dg * mask_g * GELU'(h) ----
    Gemm<2, 32> g1;
    {
        f32x4 acc[8];
        zero_acc<8>(acc);
        g2.run(acc, DX2M, XP, lane);
        g1.prime(p.b1, wave, lane);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int c0 = 128 * wave + 64 * gq + 4 * m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kb + r;
                const bool ok = row < nv;
                f32x4 dv = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const f32x4 hv = h_r[gq][r];
#pragma unroll
                    for (int c = 0; c < 4; ++c) dv[c] = acc[gq * 4 + c][r] * xgelu_grad(hv[c]);
                    if (p.mask_g != nullptr) {
                        const f32x4 mk = ld4(p.mask_g + (row0 + row) * XMLP + c0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) dv[c] *= mk[c];
                    }
                    st4(p.dh + (row0 + row) * XMLP + c0, dv);
                }
                st4(DH + row * XGP + c0, dv);
            }
        }
    }
    __syncthreads();
    tile_colsum(DH, XGP, XMLP, part + o_b1, tid);

    XF_STAMP(2);
    // ---- S3: df = dh W1 ----
    Gemm<2, 8> go;
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        g1.run(acc, DH, XGP, lane);
        go.prime(p.bo, wave, lane);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(DF + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
    }
    __syncthreads();

    XF_STAMP(3);
    // ---- S4: LayerNorm 2 backward + the residual gradient dx2 ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        const f32x4 xv = x1_r[pass];
        const float mu = m2_r[pass], rs = r2_r[pass];
        const f32x4 dfv = ld4(DF + row * XP + li * 4);
        const f32x4 r2v = ld4(DX2 + row * XP + li * 4);
        const f32x4 gam = ld4(p.ln2_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dfv[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, dm, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2) + r2v[c]; px[c] = dfv[c] * xh[c]; }
        dm = d;
        if (p.mask_o != nullptr && ok) {
            const f32x4 mk = ld4(p.mask_o + gr * XD + li * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dm[c] *= mk[c];
        }
        st4(T1 + row * XP + li * 4, px);
        st4(DX1 + row * XP + li * 4, d);
        st4(DX1M + row * XP + li * 4, dm);
        if (ok) st4(p.dx1 + gr * XD + li * 4, dm);
    }
    __syncthreads();
    tile_colsum(T1, XP, XD, part + o_ln2, tid);
    tile_colsum(DF, XP, XD, part + o_ln2 + XD, tid);
    tile_colsum(DX1M, XP, XD, part + o_bo, tid);

    XF_STAMP(4);
    // ---- S5: dout = dx1m Wo ----
    const int ntiles = Npad >> 4;
    const float* Kb = p.KR + frag_r(bz, wave, 0, 0, ntiles) + lane * 4;
    const float* Vb = p.VR + frag_r(bz, wave, 0, 0, ntiles) + lane * 4;
    const float* Kt = p.KC + frag_c(bz, wave, 0, 0, ntiles) + lane * 4;
    // operands of key tile t: K row (2), V row (2), K^T columns (2).  Three buffers in fixed roles (the loop below is
    // unrolled by three): a tile is requested two tiles before its use and nothing is ever copied — a register ring that
    // shifts its contents reads the newest load and so waits for it, one L2 round trip per tile (measured).
    f32x4 A0[6], A1[6], A2[6];
    auto load_t = [&](int t, f32x4 (&d)[6]) {
        t = t < ntiles ? t : ntiles - 1;
        d[0] = ld4(Kb + t * 512); d[1] = ld4(Kb + t * 512 + 256);
        d[2] = ld4(Vb + t * 512); d[3] = ld4(Vb + t * 512 + 256);
        d[4] = ld4(Kt + t * 256); d[5] = ld4(Kt + (size_t)(ntiles + t) * 256);
    };
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        go.run(acc, DX1M, XP, lane);
        load_t(0, A0);                      // ahead of this epilogue's stores
        load_t(1, A1);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x2*>(DO + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
        store_frag_head(p.DR, p.DC, acc, bz, wave, tile, p.tiles, m, kb);      // rows past nv are exact zeros
    }
    __syncthreads();

    XF_STAMP(5);
    // ---- S6: dQ half of the attention backward, head = wave ----
    Gemm<2, 8> gq;
    {
        const int h = wave;
        const float c = p.scale * XLOG2E;
        float qreg[2][4], doreg[2][4];
        float delta = 0.f, delta_b[2] = {0.f, 0.f};                 // (delta_b / lse_b: per 16-feature block = per head of H2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cc = XDH * h + 16 * s + 4 * kb;
            const f32x4 qv = q_r[s];
            const f32x4 ov = o_r[s];
            const f32x4 dv = ld4(DO + m * XP + cc);                 // zero rows past nv
#pragma unroll
            for (int j = 0; j < 4; ++j) { qreg[s][j] = qv[j] * c; doreg[s][j] = dv[j]; delta_b[s] += dv[j] * ov[j]; }
        }
        float lse_b[2] = {0.f, 0.f};
        if constexpr (H2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                delta_b[s] += __shfl_xor(delta_b[s], 16);
                delta_b[s] += __shfl_xor(delta_b[s], 32);
                const size_t si = ((size_t)bz * (2 * XH) + 2 * h + s) * Npad + t0 + m;
                if (kb == 0) p.delta[si] = delta_b[s];
                lse_b[s] = p.lse[si];
            }
        } else {
            delta = delta_b[0] + delta_b[1];
            delta += __shfl_xor(delta, 16);
            delta += __shfl_xor(delta, 32);
        }
        const size_t sidx = ((size_t)bz * XH + h) * Npad + t0 + m;
        if (!H2 && kb == 0) p.delta[sidx] = delta;
        const float lse2 = H2 ? 0.f : p.lse[sidx];                  // +inf past nv: p = 0 there
        f32x4 dqT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto step = [&](int t, const f32x4 (&cur)[6], f32x4 (&nxt)[6]) {
            load_t(t + 2, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const bool on = t < ntiles;                             // the last group of three may overhang: zero weight
            if constexpr (H2) {
                f32x4 sb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dpb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sb[hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[hs][j], qreg[hs][j], sb[hs], 0, 0, 0);
                        dpb[hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[2 + hs][j], doreg[hs][j], dpb[hs], 0, 0, 0);
                    }
                f32x4 dsb[2];
#pragma unroll
                for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * t + 4 * kb + r;
                        const float pr = (on && key < N) ? exp2f(sb[hs][r] - lse_b[hs]) : 0.f;
                        dsb[hs][r] = pr * (dpb[hs][r] - delta_b[hs]) * p.scale;
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dqT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[4][r], dsb[0][r], dqT[0], 0, 0, 0);
                    dqT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[5][r], dsb[1][r], dqT[1], 0, 0, 0);
                }
                return;
            }
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0][j], qreg[0][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[2][j], doreg[0][j], dp, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1][j], qreg[1][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[3][j], doreg[1][j], dp, 0, 0, 0);
            }
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * kb + r;
                const float pr = (on && key < N) ? exp2f(s[r] - lse2) : 0.f;
                ds[r] = pr * (dp[r] - delta) * p.scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dqT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[4][r], ds[r], dqT[0], 0, 0, 0);
                dqT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[5][r], ds[r], dqT[1], 0, 0, 0);
            }
        };
        for (int t = 0; t < ntiles; t += 3) {
            step(t, A0, A2);
            step(t + 1, A1, A0);
            step(t + 2, A2, A1);
        }
        gq.prime(p.bq, wave, lane);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int cc = XDH * h + 16 * dt + 4 * kb;
            f32x4 w4 = dqT[dt];
            if (m >= nv) w4 = f32x4{0.f, 0.f, 0.f, 0.f};
            st4(DQ + m * XP + cc, w4);
            if (m < nv) st4(p.dq + (row0 + m) * XD + cc, w4);
        }
    }
    __syncthreads();

    XF_STAMP(6);
    // ---- S7: da = dq Wq ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gq.run(acc, DQ, XP, lane);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(DA + (4 * kb + r) * XP + c0) = f32x2{acc[0][r], acc[1][r]};
    }
    __syncthreads();

    XF_STAMP(7);
    // ---- S8: LayerNorm 1 backward + the residual gradients dx1 (attention block) and dy (the caller's "+ tokens") ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + wave * 2 + (lane >> 5);
        const bool ok = row < nv;
        const size_t gr = row0 + row;
        const f32x4 xv = x_r[pass];
        const float mu = m1_r[pass], rs = r1_r[pass];
        const f32x4 dav = ld4(DA + row * XP + li * 4);
        const f32x4 r1v = ld4(DX1 + row * XP + li * 4);
        const f32x4 ryv = ld4(DY + row * XP + li * 4);
        const f32x4 gam = ld4(p.ln1_g + li * 4);
        f32x4 xh, gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) { xh[c] = (xv[c] - mu) * rs; gg[c] = dav[c] * gam[c]; }
        const float s1 = xhalf_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 128.f);
        const float s2 = xhalf_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 128.f);
        f32x4 d, px;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = rs * (gg[c] - s1 - xh[c] * s2) + r1v[c] + ryv[c]; px[c] = dav[c] * xh[c]; }
        st4(T1 + row * XP + li * 4, px);
        if (ok) st4(p.dx + gr * XD + li * 4, d);
    }
    __syncthreads();
    tile_colsum(T1, XP, XD, part + o_ln1, tid);
    tile_colsum(DA, XP, XD, part + o_ln1 + XD, tid);
    XF_STAMP(8);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, key side
// ---------------------------------------------------------------------------------------------------------------------
struct XfBwdKvArgs {
    const float *QR, *QC;           // fragments of the instance's queries
    const float *KR, *VR;           // row fragments of the context's K, V (this tile's own keys)
    const float *lse, *delta;       // [B][heads][Npad]
    const float *DR, *DC;           // dO fragments
    const float* bkv;               // weight pack (dx = dy W form) of to_kv
    const float* dctx_acc;          // what the context tensor's gradient has collected so far, or NULL
    float scale;
    float* dkv;                     // [R][256] (dy operand of to_kv's weight gradient)
    float* dctx;                    // [R][128]
    int B, N, Npad, tiles;
    unsigned long long* trace;
};

constexpr int XKP = 2 * XD + 4;

template <bool H2>                      // H2: eight heads of 16 (see xf_fwd_kernel)
__global__ __launch_bounds__(XTHR, 1) void xf_bwd_kv_kernel(const XfBwdKvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DKV = smem;                      // [16][XKP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kb = lane >> 4;
    const XfWho who = xf_who(p.B, p.tiles);
    if (!who.live) return;
    const int tile = who.tile, bz = who.bz;
    const int N = p.N, Npad = p.Npad;
    const int t0 = tile * XT;
    const int nv = (N - t0) < XT ? (N - t0) : XT;
    const size_t row0 = (size_t)bz * N + t0;
    const int ntiles = Npad >> 4;
    XF_STAMP(0);
    const int h = wave;
    const float* Qb = p.QR + frag_r(bz, h, 0, 0, ntiles) + lane * 4;
    const float* Db = p.DR + frag_r(bz, h, 0, 0, ntiles) + lane * 4;
    const float* Qt = p.QC + frag_c(bz, h, 0, 0, ntiles) + lane * 4;
    const float* Dt = p.DC + frag_c(bz, h, 0, 0, ntiles) + lane * 4;
    // (H2: the wave's two heads 2 h, 2 h + 1 of eight: rows of lse / delta Npad apart)
    const float* Ls = p.lse + ((size_t)bz * (H2 ? 2 * XH : XH) + (H2 ? 2 * h : h)) * Npad + 4 * kb;
    const float* Dl = p.delta + ((size_t)bz * (H2 ? 2 * XH : XH) + (H2 ? 2 * h : h)) * Npad + 4 * kb;
    // operands of query tile t: Q row (2), dO row (2), Q^T columns (2), dO^T columns (2), lse, delta (H2: of both heads) — three
    // buffers in fixed roles, a tile is requested two tiles before its use (see xf_bwd_q_kernel)
    constexpr int NOP = H2 ? 12 : 10;
    f32x4 A0[NOP], A1[NOP], A2[NOP];
    auto load_t = [&](int t, f32x4 (&d)[NOP]) {
        t = t < ntiles ? t : ntiles - 1;
        d[0] = ld4(Qb + t * 512); d[1] = ld4(Qb + t * 512 + 256);
        d[2] = ld4(Db + t * 512); d[3] = ld4(Db + t * 512 + 256);
        d[4] = ld4(Qt + t * 256); d[5] = ld4(Qt + (size_t)(ntiles + t) * 256);
        d[6] = ld4(Dt + t * 256); d[7] = ld4(Dt + (size_t)(ntiles + t) * 256);
        d[8] = ld4(Ls + 16 * t); d[9] = ld4(Dl + 16 * t);
        if constexpr (H2) { d[10] = ld4(Ls + Npad + 16 * t); d[11] = ld4(Dl + Npad + 16 * t); }
    };
    // this lane's key (B operand column j = m): K row scaled, V row
    float kreg[2][4], vreg[2][4];
    {
        const float c = p.scale * XLOG2E;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 kv4 = ld4(p.KR + frag_r(bz, h, tile, s, ntiles) + lane * 4);
            const f32x4 vv4 = ld4(p.VR + frag_r(bz, h, tile, s, ntiles) + lane * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { kreg[s][j] = kv4[j] * c; vreg[s][j] = vv4[j]; }
        }
    }
    load_t(0, A0);
    load_t(1, A1);
    f32x4 pf[XPF];
    {   // L2 warm-up: to_kv's weight pack and the batch element's Q panels (from the forward pass: cold); dO / delta were
        // written by the query-side launch just before this one, on this XCD
        const int panel = Npad * XD / 4;
        const XfRegion reg[3] = {{p.bkv, 2 * XD * XD / 4}, {p.QR + (size_t)bz * Npad * XD, panel},
                                 {p.QC + (size_t)bz * Npad * XD, panel}};
        xf_prefetch<3>(reg, who.tile, p.tiles, tid, pf);
    }
    Gemm<2, 16> gkv;
    {
        f32x4 dkT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dvT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto step = [&](int t, const f32x4 (&cur)[NOP], f32x4 (&nxt)[NOP]) {
            load_t(t + 2, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const bool on = t < ntiles;                             // the last group of three may overhang: zero weight
            if constexpr (H2) {
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, dpb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sb = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[hs][j], kreg[hs][j], sb, 0, 0, 0);
                        dpb = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[2 + hs][j], vreg[hs][j], dpb, 0, 0, 0);
                    }
                    f32x4 prb, dsb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        prb[r] = on ? exp2f(sb[r] - cur[hs ? (NOP - 2) : 8][r]) : 0.f;
                        dsb[r] = prb[r] * (dpb[r] - cur[hs ? (NOP - 1) : 9][r]) * p.scale;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dvT[hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[6 + hs][r], prb[r], dvT[hs], 0, 0, 0);
                        dkT[hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[4 + hs][r], dsb[r], dkT[hs], 0, 0, 0);
                    }
                }
                return;
            }
            // s[r] = S[query 16 t + 4 kb + r][key m];  dp likewise
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0][j], kreg[0][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[2][j], vreg[0][j], dp, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1][j], kreg[1][j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[3][j], vreg[1][j], dp, 0, 0, 0);
            }
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = on ? exp2f(s[r] - cur[8][r]) : 0.f;         // lse = +inf on padded queries: 0
                ds[r] = pr[r] * (dp[r] - cur[9][r]) * p.scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dvT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[6][r], pr[r], dvT[0], 0, 0, 0);
                dvT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[7][r], pr[r], dvT[1], 0, 0, 0);
                dkT[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[4][r], ds[r], dkT[0], 0, 0, 0);
                dkT[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[5][r], ds[r], dkT[1], 0, 0, 0);
            }
        };
        for (int t = 0; t < ntiles; t += 3) {
            step(t, A0, A2);
            step(t + 1, A1, A0);
            step(t + 2, A2, A1);
        }
        xf_prefetch_done(pf);
        gkv.prime(p.bkv, wave, lane);
        // lane (m = key, kb): dkT[dt][r] = dK[key m][feature 32 h + 16 dt + 4 kb + r]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int cc = XDH * h + 16 * dt + 4 * kb;
            f32x4 a = dkT[dt], b = dvT[dt];
            if (m >= nv) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
            st4(DKV + m * XKP + cc, a);
            st4(DKV + m * XKP + XD + cc, b);
            if (m < nv) {
                st4(p.dkv + (row0 + m) * (2 * XD) + cc, a);
                st4(p.dkv + (row0 + m) * (2 * XD) + XD + cc, b);
            }
        }
    }
    XF_STAMP(1);
    __syncthreads();
    // ---- dctx = dkv Wkv + what the context has collected ----
    {
        f32x4 acc[2];
        zero_acc<2>(acc);
        gkv.run(acc, DKV, XKP, lane);
        const int c0 = 32 * wave + 2 * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kb + r;
            if (row < nv) {
                f32x2 v = {acc[0][r], acc[1][r]};
                if (p.dctx_acc != nullptr) {
                    const f32x2 a = *reinterpret_cast<const f32x2*>(p.dctx_acc + (row0 + row) * XD + c0);
                    v[0] += a[0]; v[1] += a[1];
                }
                *reinterpret_cast<f32x2*>(p.dctx + (row0 + row) * XD + c0) = v;
            }
        }
    }
    XF_STAMP(2);
}

// Weight packs in fragment order (see gemm_tile): dst[wave][s][u][lane][j] = B[k = 16 s + 4 (lane >> 4) + j][c],
// c = TileMap::col(16 tpw wave, u, lane & 15);  nn == 0: B[k][c] = src[c K + k] (y = x W^T, src = W [N][K]);
// nn == 1: B[k][c] = src[k N + c] (dx = dy W, src = W [K][N]).  One thread per 16 bytes of a pack; a few MB per step.
struct XfPackDesc { const float* src; float* dst; int N, K, tpw, ks, nn; };
constexpr int XF_PACK_MAX = 60;
struct XfPackArgs { XfPackDesc d[XF_PACK_MAX]; };

__global__ __launch_bounds__(256) void xf_pack_kernel(const XfPackArgs p) {
    const XfPackDesc d = p.d[blockIdx.y];
    const int q4 = blockIdx.x * 256 + threadIdx.x;
    if (q4 >= d.N * d.K / 4) return;
    const int lane = q4 & 63;
    int r = q4 >> 6;
    const int u = r % d.tpw;
    r /= d.tpw;
    const int s = r % d.ks, w = r / d.ks;
    const int m = lane & 15, kb = lane >> 4;
    const int G = d.tpw < 4 ? d.tpw : 4;
    const int c = w * 16 * d.tpw + (u / G) * (16 * G) + G * m + (u % G);
    const int k = 16 * s + 4 * kb;
    f32x4 v;
    if (!d.nn) {
        v = ld4(d.src + (size_t)c * d.K + k);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d.src[(size_t)(k + j) * d.N + c];
    }
    st4(d.dst + (size_t)q4 * 4, v);
}

// Column sums of the per-tile partials of ALL instances in one launch: out = sum over the tile rows, fp64 accumulation in
// fixed order.  Columns [0, nsmall) go to small[i], the rest to lnf[i].
constexpr int XF_MAX_INST = 2 * TMF_FUSION_MAX_DEPTH;
struct XfColsumArgs {
    const float* part[XF_MAX_INST];
    float* small[XF_MAX_INST];
    float* lnf[XF_MAX_INST];
    int nblk, stride, nsmall;
};

__global__ __launch_bounds__(64 * 16) void xf_colsum_kernel(const XfColsumArgs p) {
    __shared__ double red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, i = blockIdx.y;
    const float* src = p.part[i];
    double a = 0.0;
    if (c < p.stride)
        for (int r = ty; r < p.nblk; r += 16) a += (double)src[(size_t)r * p.stride + c];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < p.stride) {
#pragma unroll
        for (int k = 1; k < 16; ++k) a += red[k][tx];
        if (c < p.nsmall) p.small[i][c] = (float)a;
        else p.lnf[i][c - p.nsmall] = (float)a;
    }
}

constexpr size_t XF_FWD_LDS = (size_t)(6 * XT * XP + XT * XGP) * 4;
constexpr size_t XF_BWDQ_LDS = (size_t)(10 * XT * XP + XT * XGP) * 4;
constexpr size_t XF_BWDKV_LDS = (size_t)(XT * XKP) * 4;

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side: launchers used by fusion_path.hip (C++ linkage, declared there)
// ---------------------------------------------------------------------------------------------------------------------
static unsigned long long* g_xf_trace[3] = {nullptr, nullptr, nullptr};
// debugging hook (tools/xf_trace.py): per-wave phase time stamps of the fused kernels, [workgroups][4 waves][16] uint64 each
extern "C" void tmf_debug_xf_trace(void* fwd, void* bwd_q, void* bwd_kv) {
    g_xf_trace[0] = (unsigned long long*)fwd; g_xf_trace[1] = (unsigned long long*)bwd_q; g_xf_trace[2] = (unsigned long long*)bwd_kv;
}

bool tmf_xf_supported(int N, int dim, int heads, int dim_head, int mlp) {
    // 4 heads of 32 (kfold_train_adversarial.py:78-79), or 8 heads of 16 (train_adversarial.py:30-31: the H2 instances)
    return dim == XD && ((heads == XH && dim_head == XDH) || (heads == 2 * XH && dim_head == XDH / 2)) && mlp == XMLP && N >= 1 && N <= 512;
}
int tmf_xf_npad(int N) { return (N + XT - 1) / XT * XT; }
int tmf_xf_tiles(int N) { return (N + XT - 1) / XT; }
int tmf_xf_part_stride(void) { return 8 * XD + XMLP; }

// float offsets of the packs inside an instance's forward / backward pack buffer (XF_PACK_FLOATS each)
enum { XF_PK_Q = 0, XF_PK_O = XD * XD, XF_PK_1 = 2 * XD * XD, XF_PK_2 = 2 * XD * XD + XMLP * XD, XF_PK_KV = 2 * XD * XD + 2 * XMLP * XD,
       XF_PACK_FLOATS = 4 * XD * XD + 2 * XMLP * XD };
enum { XF_BK_2 = 0, XF_BK_1 = XMLP * XD, XF_BK_O = 2 * XMLP * XD, XF_BK_Q = 2 * XMLP * XD + XD * XD, XF_BK_KV = 2 * XMLP * XD + 2 * XD * XD };
int tmf_xf_pack_floats(void) { return XF_PACK_FLOATS; }
int tmf_xf_pack_kv_offset(void) { return XF_PK_KV; }

// Re-pack the weights of n_inst instances: pk_fwd[i] / pk_bwd[i] = XF_PACK_FLOATS floats each.
int tmf_xf_launch_pack(int n_inst, const tmf_xformer_params* inst, float* const* pk_fwd, float* const* pk_bwd, hipStream_t s) {
    for (int i0 = 0; i0 < n_inst; i0 += XF_PACK_MAX / 10) {
        const int ni = (n_inst - i0) < XF_PACK_MAX / 10 ? (n_inst - i0) : XF_PACK_MAX / 10;
        XfPackArgs a = {};
        for (int k = 0; k < ni; ++k) {
            const tmf_xformer_params& w = inst[i0 + k];
            float* f = pk_fwd[i0 + k];
            float* b = pk_bwd[i0 + k];
            XfPackDesc* d = a.d + 10 * k;
            d[0] = {w.wq, f + XF_PK_Q, XD, XD, 2, 8, 0};
            d[1] = {w.wo, f + XF_PK_O, XD, XD, 2, 8, 0};
            d[2] = {w.w1, f + XF_PK_1, XMLP, XD, 8, 8, 0};
            d[3] = {w.w2, f + XF_PK_2, XD, XMLP, 2, 32, 0};
            d[4] = {w.wkv, f + XF_PK_KV, 2 * XD, XD, 4, 8, 0};
            d[5] = {w.w2, b + XF_BK_2, XMLP, XD, 8, 8, 1};             // dg = dx2 W2:  W2 [dim][mlp]
            d[6] = {w.w1, b + XF_BK_1, XD, XMLP, 2, 32, 1};            // df = dh W1:   W1 [mlp][dim]
            d[7] = {w.wo, b + XF_BK_O, XD, XD, 2, 8, 1};               // dout = dx1 Wo
            d[8] = {w.wq, b + XF_BK_Q, XD, XD, 2, 8, 1};               // da = dq Wq
            d[9] = {w.wkv, b + XF_BK_KV, XD, 2 * XD, 2, 16, 1};        // dctx = dkv Wkv: Wkv [2 inner][dim]
        }
        hipLaunchKernelGGL(xf_pack_kernel, dim3(XMLP * XD / 4 / 256, 10 * ni), dim3(256), 0, s, a);
        int rc = tmf_launch_result("tmf_fusion_train_fwd(weight packs)");
        if (rc) return rc;
    }
    return TMF_OK;
}

struct tmf_xf_fwd_io {
    const float *x, *KR, *VC, *pk, *pkv_next, *mask_o, *mask_g, *mask_f;      // pk: this instance's forward pack buffer
    float *a, *QR, *QC, *out, *lse, *x1, *f, *h, *g, *x2, *y, *m1, *r1, *m2, *r2, *mf, *rf, *KRn, *KCn, *VRn, *VCn;
};

int tmf_xf_launch_fwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_fwd_io* io, float scale, int only_kv, int h2,
                      hipStream_t s) {
    XfFwdArgs a = {};
    a.x = io->x; a.KR = io->KR; a.VC = io->VC; a.pkv_next = io->pkv_next;
    a.mask_o = io->mask_o; a.mask_g = io->mask_g; a.mask_f = io->mask_f;
    if (w != nullptr) {
        a.ln1_g = w->ln1_g; a.ln1_b = w->ln1_b; a.bo = w->bo; a.ln2_g = w->ln2_g; a.ln2_b = w->ln2_b;
        a.b1 = w->b1; a.b2 = w->b2; a.lnf_g = w->lnf_g; a.lnf_b = w->lnf_b;
        a.eps1 = w->eps1; a.eps2 = w->eps2; a.epsf = w->epsf;
        a.pq = io->pk + XF_PK_Q; a.po = io->pk + XF_PK_O; a.p1 = io->pk + XF_PK_1; a.p2 = io->pk + XF_PK_2;
    }
    a.scale = scale;
    a.a = io->a; a.QR = io->QR; a.QC = io->QC; a.out = io->out; a.lse = io->lse; a.x1 = io->x1; a.f = io->f; a.h = io->h; a.g = io->g;
    a.x2 = io->x2; a.y = io->y; a.m1 = io->m1; a.r1 = io->r1; a.m2 = io->m2; a.r2 = io->r2; a.mf = io->mf; a.rf = io->rf;
    a.KRn = io->KRn; a.KCn = io->KCn; a.VRn = io->VRn; a.VCn = io->VCn;
    a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N); a.only_kv = only_kv;
    a.trace = only_kv ? nullptr : g_xf_trace[0];
    const dim3 grid(8 * a.tiles * ((B + 7) / 8)), block(XTHR);
    const int mt = a.Npad / 16;
    int rc;
#define XF_LAUNCH(MT, EX)                                                                     \
    {                                                                                         \
        auto kf = h2 ? xf_fwd_kernel<MT, EX, true> : xf_fwd_kernel<MT, EX, false>;            \
        if ((rc = tmf_allow_lds(kf, XF_FWD_LDS, "tmf_fusion_train_fwd(fused)"))) return rc;   \
        hipLaunchKernelGGL(kf, grid, block, XF_FWD_LDS, s, a);                                \
    }
    // exact instances for the token counts of the benchmark volumes (96^3 -> 216, 91x109x91 -> 150, 128^3 -> 512)
    if (mt == 14) XF_LAUNCH(14, true)
    else if (mt == 10) XF_LAUNCH(10, true)
    else if (mt == 32) XF_LAUNCH(32, true)
    else if (mt <= 8) XF_LAUNCH(8, false)
    else if (mt <= 16) XF_LAUNCH(16, false)
    else XF_LAUNCH(32, false)
#undef XF_LAUNCH
    return tmf_launch_result("tmf_fusion_train_fwd(fused)");
}

struct tmf_xf_bwd_io {
    const float *dy, *x, *KR, *KC, *VR, *pk, *mask_o, *mask_g, *mask_f;        // pk: this instance's backward pack buffer
    const float *QR, *QC, *out, *lse, *x1, *h, *x2, *m1, *r1, *m2, *r2, *mf, *rf;
    float *dx2, *dh, *dx1, *dq, *DR, *DC, *delta, *dx, *part, *dkv, *dctx;
    const float* dctx_acc;
};

int tmf_xf_launch_bwd(int B, int N, const tmf_xformer_params* w, const tmf_xf_bwd_io* io, float scale, int h2, hipStream_t s) {
    int rc;
    const int grid = 8 * tmf_xf_tiles(N) * ((B + 7) / 8);
    {
        XfBwdQArgs a = {};
        a.dy = io->dy; a.x = io->x; a.KR = io->KR; a.KC = io->KC; a.VR = io->VR;
        a.ln1_g = w->ln1_g; a.ln2_g = w->ln2_g; a.lnf_g = w->lnf_g;
        a.b2 = io->pk + XF_BK_2; a.b1 = io->pk + XF_BK_1; a.bo = io->pk + XF_BK_O; a.bq = io->pk + XF_BK_Q;
        a.mask_o = io->mask_o; a.mask_g = io->mask_g; a.mask_f = io->mask_f;
        a.scale = scale;
        a.QR = io->QR; a.out = io->out; a.lse = io->lse; a.x1 = io->x1; a.h = io->h; a.x2 = io->x2;
        a.m1 = io->m1; a.r1 = io->r1; a.m2 = io->m2; a.r2 = io->r2; a.mf = io->mf; a.rf = io->rf;
        a.dx2 = io->dx2; a.dh = io->dh; a.dx1 = io->dx1; a.dq = io->dq; a.DR = io->DR; a.DC = io->DC; a.delta = io->delta;
        a.dx = io->dx; a.part = io->part; a.stride = tmf_xf_part_stride();
        a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N);
        a.trace = g_xf_trace[1];
        auto kf = h2 ? xf_bwd_q_kernel<true> : xf_bwd_q_kernel<false>;
        if ((rc = tmf_allow_lds(kf, XF_BWDQ_LDS, "tmf_fusion_train_bwd(fused q)"))) return rc;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(XTHR), XF_BWDQ_LDS, s, a);
        if ((rc = tmf_launch_result("tmf_fusion_train_bwd(fused q)"))) return rc;
    }
    {
        XfBwdKvArgs a = {};
        a.QR = io->QR; a.QC = io->QC; a.KR = io->KR; a.VR = io->VR; a.lse = io->lse; a.delta = io->delta; a.DR = io->DR; a.DC = io->DC;
        a.bkv = io->pk + XF_BK_KV; a.dctx_acc = io->dctx_acc; a.scale = scale; a.dkv = io->dkv; a.dctx = io->dctx;
        a.B = B; a.N = N; a.Npad = tmf_xf_npad(N); a.tiles = tmf_xf_tiles(N);
        a.trace = g_xf_trace[2];
        auto kkv = h2 ? xf_bwd_kv_kernel<true> : xf_bwd_kv_kernel<false>;
        hipLaunchKernelGGL(kkv, dim3(grid), dim3(XTHR), XF_BWDKV_LDS, s, a);
        if ((rc = tmf_launch_result("tmf_fusion_train_bwd(fused kv)"))) return rc;
    }
    return TMF_OK;
}

int tmf_xf_launch_colsum(int n_inst, const float* const* part, float* const* small, float* const* lnf, int nblk, hipStream_t s) {
    XfColsumArgs a = {};
    for (int i = 0; i < n_inst; ++i) { a.part[i] = part[i]; a.small[i] = small[i]; a.lnf[i] = lnf[i]; }
    a.nblk = nblk; a.stride = tmf_xf_part_stride(); a.nsmall = 6 * XD + XMLP;
    hipLaunchKernelGGL(xf_colsum_kernel, dim3(tmf_cdiv(a.stride, 64), n_inst), dim3(64 * 16), 0, s, a);
    return tmf_launch_result("tmf_fusion_train_bwd(fused colsum)");
}
