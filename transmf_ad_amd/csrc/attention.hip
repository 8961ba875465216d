// attention.hip — fused multi-head cross attention  out = softmax(q k^T * scale) v  for
// gfx950, forward and backward, fp32 on v_mfma_f32_32x32x2_f32.
//
// One wavefront owns a 32-row tile of queries (forward, dQ) or of keys (dK/dV); the other
// side of the product is staged whole in LDS (rows padded to dh+1 floats: conflict-free both
// for "one row per lane" and for "one column per lane" reads).  Scores are computed
// TRANSPOSED — S^T = K Q^T — so that after the MFMA a lane holds 16 keys of ONE query
// (column = lane & 31): the softmax row reduction is a per-lane loop plus one
// lane <-> lane+32 exchange (wavefront-reduced, no LDS round trip), and the probability
// tile is already in the B-operand layout of the next product (O^T = V^T P^T), so it never
// leaves the registers.  Keys stream in chunks of 128 with an online softmax, the score
// matrix is never materialised (networks.py:169-173 materialises (B,h,N,N)).
//
// Round 2: a workgroup is 4 waves = 2 row tiles x 2 SPLITS of the streamed side (key chunks in the forward and dQ, query
// tiles in dK/dV).  At 216 / 512 tokens a launch is only 224 / 512 row tiles for the chip's 1 024 SIMDs and a wave's
// serial chain (MFMAs + softmax arithmetic, one wave per SIMD) is the critical path: splitting it puts every tile on two
// SIMDs (the two splits of a tile are adjacent waves, i.e. different SIMDs of the CU).  The two partial results of a tile
// are merged through LDS (online-softmax merge in the forward, plain sums in the backward), in a fixed order.
//
// Replaces, at /root/reference/models/networks.py:166-174: the three rearranges, einsum
// 'bhid,bhjd->bhij' * scale, Softmax(dim=-1), einsum 'bhij,bhjd->bhid' and their backward.
#include "tmf_common.h"

namespace {

constexpr int KC = 4;            // key tiles (of 32) per online-softmax chunk
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int frag_row(int r, int hsel) { return (r & 3) + 8 * (r >> 2) + 4 * hsel; }

// acc[i = LDS row][j] += sum_d lds[row0 + (lane&31)][d] * regs[d/2]   (lane>>5 selects d parity)
template <int DH>
__device__ __forceinline__ void mma_rows(f32x16& acc, const float* lds, int row0, const float (&regs)[DH / 2],
                                         int l31, int hsel) {
    const float* p = lds + (row0 + l31) * (DH + 1) + hsel;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[2 * s], regs[s], acc, 0, 0, 0);
}

// acc[i = d][j] += sum_r lds[row0 + frag_row(r)][dt*32 + (lane&31)] * tile[r]
template <int DH>
__device__ __forceinline__ void mma_cols(f32x16& acc, const float* lds, int row0, const f32x16& tile, int dt,
                                         int l31, int hsel) {
    const float* p = lds + (row0 + 4 * hsel) * (DH + 1) + dt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[((r & 3) + 8 * (r >> 2)) * (DH + 1)], tile[r], acc, 0, 0, 0);
}

// Stage `nrows` rows (zero-filled past `limit`) of a [*, stride] matrix, columns [col0, col0+DH), into lds[(DH+1)].
// 16-byte global loads, eight in flight per thread before the first LDS write (the whole K/V panel of a head is
// two or three such batches: the staging is bandwidth- not latency-bound).  Needs stride % 4 == 0, col0 % 4 == 0.
template <int DH>
__device__ __forceinline__ void stage_rows(float* lds, const float* src, int stride, int col0, int first, int nrows,
                                           int limit, int tid, int nthreads) {
    constexpr int Q = DH / 4;                 // float4 per row
    constexpr int BATCH = 8;
    const int total = nrows * Q;
    for (int base = 0; base < total; base += nthreads * BATCH) {
        f32x4 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = base + u * nthreads + tid;
            const int r = e / Q, c = (e % Q) * 4;
            const int g = first + r;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (e < total && g < limit) t = *reinterpret_cast<const f32x4*>(src + (size_t)g * stride + col0 + c);
            v[u] = t;
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = base + u * nthreads + tid;
            if (e < total) {
                float* d = lds + (e / Q) * (DH + 1) + (e % Q) * 4;
                d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
            }
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256, DH <= 32 ? 2 : 1) void xattn_fwd_kernel(   // <= 256 registers: MFMA results stay in VGPRs
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, float* __restrict__ lse, int heads, int N, int M, int q_stride, int kv_stride,
    float scale, int sb) {
    constexpr int DT = (DH + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (sb + 1) * (DH + 1);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int ksp = wave & 1;                                // which half of the key chunks this wave takes
    const int q0 = (blockIdx.x * 2 + (wave >> 1)) * 32;
    const float* qb = q + (size_t)b * N * q_stride;
    const float* kb = k + (size_t)b * M * kv_stride;
    const float* vb = v + (size_t)b * M * kv_stride;

    float qreg[DH / 2];
    {
        const int qi = q0 + l31;
        const float c = scale * LOG2E;
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) qreg[s] = qi < N ? qb[(size_t)qi * q_stride + h * DH + 2 * s + hsel] * c : 0.f;
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int chunk = 0;                                           // key chunks are dealt out alternately

    for (int sb0 = 0; sb0 < M; sb0 += sb) {
        const int nrows = (M - sb0) < sb ? (M - sb0) : sb;
        const int nrows_pad = (nrows + 31) & ~31;
        if (sb0 > 0) __syncthreads();
        stage_rows<DH>(Ks, kb, kv_stride, h * DH, sb0, nrows_pad, M, tid, 256);
        stage_rows<DH>(Vs, vb, kv_stride, h * DH, sb0, nrows_pad, M, tid, 256);
        __syncthreads();
        if (q0 < N) {
            for (int kt0 = 0; kt0 < nrows_pad / 32; kt0 += KC) {
                if ((chunk++ & 1) != ksp) continue;
                f32x16 sT[KC];
                float mx = m_run;
#pragma unroll
                for (int t = 0; t < KC; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sT[t][r] = 0.f;
                    if (kt0 + t < nrows_pad / 32) {
                        mma_rows<DH>(sT[t], Ks, (kt0 + t) * 32, qreg, l31, hsel);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = sb0 + (kt0 + t) * 32 + frag_row(r, hsel);
                            sT[t][r] = key < M ? sT[t][r] : -INFINITY;
                            mx = fmaxf(mx, sT[t][r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sT[t][r] = -INFINITY;
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float alpha = exp2f(m_run - mx);      // m_run = -inf on the first chunk -> 0
                m_run = mx;
                float psum = 0.f;
#pragma unroll
                for (int t = 0; t < KC; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = exp2f(sT[t][r] - mx);
                        sT[t][r] = p;
                        psum += p;
                    }
                l_run = l_run * alpha + psum;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
                for (int t = 0; t < KC; ++t) {
                    if (kt0 + t < nrows_pad / 32) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) mma_cols<DH>(o[dt], Vs, (kt0 + t) * 32, sT[t], dt, l31, hsel);
                    }
                }
            }
        }
    }
    // merge the two key halves of a query tile: (o, m, l) of the second through LDS (the panels are dead by now)
    constexpr int XW = 16 * DT + 2;
    __syncthreads();
    float* xch = smem + ((wave >> 1) * 64 + lane) * XW;
    if (ksp == 1) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[dt * 16 + r] = o[dt][r];
        xch[16 * DT] = m_run;
        xch[16 * DT + 1] = l_run;
    }
    __syncthreads();
    if (ksp == 1) return;
    {
        const float m1 = xch[16 * DT], l1 = xch[16 * DT + 1];
        const float mx = fmaxf(m_run, m1);                   // this half always holds chunk 0: finite
        const float a0 = exp2f(m_run - mx), a1 = exp2f(m1 - mx);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = o[dt][r] * a0 + xch[dt * 16 + r] * a1;
        l_run = l_run * a0 + l1 * a1;
        m_run = mx;
    }
    const int qi = q0 + l31;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qi < N) {
        const float inv = 1.f / l_tot;
        float* ob = out + ((size_t)b * N + qi) * (heads * DH) + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hsel;
                if (d < DH) {
                    f32x4 w4 = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                    *reinterpret_cast<f32x4*>(ob + d) = w4;
                }
            }
        if (hsel == 0) lse[((size_t)b * heads + h) * N + qi] = m_run + log2f(l_tot);   // base-2 log-sum-exp of scaled scores
    }
}

// dQ: same geometry as the forward (K, V resident in LDS; a wave owns 32 queries).
template <int DH>
__global__ __launch_bounds__(256, DH <= 32 ? 2 : 1) void xattn_bwd_dq_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ dout,
    float* __restrict__ dq, int heads, int N, int M, int q_stride, int kv_stride, float scale, int sb) {
    constexpr int DT = (DH + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (sb + 1) * (DH + 1);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int ksp = wave & 1;                                // which half of the key tiles this wave takes
    const int q0 = (blockIdx.x * 2 + (wave >> 1)) * 32;
    const int qi = q0 + l31;
    const float* qb = q + (size_t)b * N * q_stride;
    const float* kb = k + (size_t)b * M * kv_stride;
    const float* vb = v + (size_t)b * M * kv_stride;
    const int inner = heads * DH;

    float qreg[DH / 2], doreg[DH / 2];
    float delta = 0.f;
    {
        const float c = scale * LOG2E;
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) {
            const int d = h * DH + 2 * s + hsel;
            qreg[s] = qi < N ? qb[(size_t)qi * q_stride + d] * c : 0.f;
            doreg[s] = qi < N ? dout[((size_t)b * N + qi) * inner + d] : 0.f;
            const float ov = qi < N ? out[((size_t)b * N + qi) * inner + d] : 0.f;
            delta += doreg[s] * ov;
        }
        delta += __shfl_xor(delta, 32);
    }
    const float lse2 = qi < N ? lse[((size_t)b * heads + h) * N + qi] : 0.f;
    f32x16 dqT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[dt][r] = 0.f;

    for (int sb0 = 0; sb0 < M; sb0 += sb) {
        const int nrows = (M - sb0) < sb ? (M - sb0) : sb;
        const int nrows_pad = (nrows + 31) & ~31;
        if (sb0 > 0) __syncthreads();
        stage_rows<DH>(Ks, kb, kv_stride, h * DH, sb0, nrows_pad, M, tid, 256);
        stage_rows<DH>(Vs, vb, kv_stride, h * DH, sb0, nrows_pad, M, tid, 256);
        __syncthreads();
        if (q0 < N) {
            for (int kt = ksp; kt < nrows_pad / 32; kt += 2) {
                f32x16 sT, dpT;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
                mma_rows<DH>(sT, Ks, kt * 32, qreg, l31, hsel);
                mma_rows<DH>(dpT, Vs, kt * 32, doreg, l31, hsel);
                // zero-filled key rows give dS * 0 below, so no key mask is needed here
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = exp2f(sT[r] - lse2);
                    sT[r] = p * (dpT[r] - delta) * scale;
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) mma_cols<DH>(dqT[dt], Ks, kt * 32, sT, dt, l31, hsel);
            }
        }
    }
    // sum the two key halves of a query tile through LDS (the panels are dead by now)
    __syncthreads();
    {
        float* xch = smem + ((wave >> 1) * 64 + lane) * (16 * DT);
        if (ksp == 1) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[dt * 16 + r] = dqT[dt][r];
        }
        __syncthreads();
        if (ksp == 1) return;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqT[dt][r] += xch[dt * 16 + r];
    }
    if (qi < N) {
        float* ob = dq + ((size_t)b * N + qi) * inner + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hsel;
                if (d < DH) {
                    f32x4 w4 = {dqT[dt][4 * g], dqT[dt][4 * g + 1], dqT[dt][4 * g + 2], dqT[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ob + d) = w4;
                }
            }
    }
}

// dK, dV: a wave owns 32 keys; Q and dO (all queries) are resident in LDS with lse / delta.
template <int DH>
__global__ __launch_bounds__(256, DH <= 32 ? 2 : 1) void xattn_bwd_dkv_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ dout,
    float* __restrict__ dk, float* __restrict__ dv, int heads, int N, int M, int q_stride, int kv_stride,
    int dkv_stride, float scale, int sb) {
    constexpr int DT = (DH + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ds = smem + (sb + 1) * (DH + 1);
    float* lses = Ds + (sb + 1) * (DH + 1);
    float* dels = lses + sb;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int ksp = wave & 1;                                // which half of the query tiles this wave takes
    const int k0 = (blockIdx.x * 2 + (wave >> 1)) * 32;
    const int ki = k0 + l31;
    const float* qb = q + (size_t)b * N * q_stride;
    const float* kb = k + (size_t)b * M * kv_stride;
    const float* vb = v + (size_t)b * M * kv_stride;
    const int inner = heads * DH;
    const float* dob = dout + (size_t)b * N * inner;
    const float* ob = out + (size_t)b * N * inner;
    const float c2 = scale * LOG2E;

    float kreg[DH / 2], vreg[DH / 2];
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) {
        const int d = h * DH + 2 * s + hsel;
        kreg[s] = ki < M ? kb[(size_t)ki * kv_stride + d] : 0.f;
        vreg[s] = ki < M ? vb[(size_t)ki * kv_stride + d] : 0.f;
    }
    f32x16 dkT[DT], dvT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[dt][r] = 0.f; dvT[dt][r] = 0.f; }

    for (int sb0 = 0; sb0 < N; sb0 += sb) {
        const int nrows = (N - sb0) < sb ? (N - sb0) : sb;
        const int nrows_pad = (nrows + 31) & ~31;
        if (sb0 > 0) __syncthreads();
        stage_rows<DH>(Qs, qb, q_stride, h * DH, sb0, nrows_pad, N, tid, 256);
        stage_rows<DH>(Ds, dob, inner, h * DH, sb0, nrows_pad, N, tid, 256);
        for (int r = tid; r < nrows_pad; r += 256) {
            const int g = sb0 + r;
            float dl = 0.f, ls = 0.f;
            if (g < N) {
                ls = lse[((size_t)b * heads + h) * N + g];
                const f32x4* pa = reinterpret_cast<const f32x4*>(dob + (size_t)g * inner + h * DH);
                const f32x4* pb = reinterpret_cast<const f32x4*>(ob + (size_t)g * inner + h * DH);
#pragma unroll
                for (int d = 0; d < DH / 4; ++d) {
                    const f32x4 u = pa[d], w4 = pb[d];
                    dl += u[0] * w4[0] + u[1] * w4[1] + u[2] * w4[2] + u[3] * w4[3];
                }
            }
            lses[r] = ls;
            dels[r] = dl;
        }
        __syncthreads();
        if (k0 < M) {
            for (int qt = ksp; qt < nrows_pad / 32; qt += 2) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                mma_rows<DH>(s, Qs, qt * 32, kreg, l31, hsel);      // s[query][key]
                mma_rows<DH>(dp, Ds, qt * 32, vreg, l31, hsel);     // dP[query][key]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = qt * 32 + frag_row(r, hsel);
                    const float p = exp2f(s[r] * c2 - lses[qr]);    // zero-filled query rows: dO = 0, delta = 0
                    s[r] = p;
                    dp[r] = p * (dp[r] - dels[qr]) * scale;
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    mma_cols<DH>(dvT[dt], Ds, qt * 32, s, dt, l31, hsel);    // dV^T += dO^T P
                    mma_cols<DH>(dkT[dt], Qs, qt * 32, dp, dt, l31, hsel);   // dK^T += Q^T dS
                }
            }
        }
    }
    // sum the two query halves of a key tile through LDS (the panels are dead by now)
    __syncthreads();
    {
        float* xch = smem + ((wave >> 1) * 64 + lane) * (32 * DT);
        if (ksp == 1) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { xch[dt * 32 + r] = dkT[dt][r]; xch[dt * 32 + 16 + r] = dvT[dt][r]; }
        }
        __syncthreads();
        if (ksp == 1) return;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dkT[dt][r] += xch[dt * 32 + r]; dvT[dt][r] += xch[dt * 32 + 16 + r]; }
    }
    if (ki < M) {
        float* dkb = dk + ((size_t)b * M + ki) * dkv_stride + h * DH;
        float* dvb = dv + ((size_t)b * M + ki) * dkv_stride + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hsel;
                if (d < DH) {
                    f32x4 a = {dkT[dt][4 * g], dkT[dt][4 * g + 1], dkT[dt][4 * g + 2], dkT[dt][4 * g + 3]};
                    f32x4 c = {dvT[dt][4 * g], dvT[dt][4 * g + 1], dvT[dt][4 * g + 2], dvT[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dkb + d) = a;
                    *reinterpret_cast<f32x4*>(dvb + d) = c;
                }
            }
    }
}

// rows of the LDS-resident side per super-block: everything if it fits in ~135 KiB, else 512 (256 for dh = 64)
int resident_rows(int n, int dh) {
    const int cap = dh == 64 ? 256 : 512;
    const int need = (n + 31) & ~31;
    return need < cap ? need : cap;
}
// panels, or the merge area of the two splits (2 row tiles x 64 lanes x the accumulators of a lane) if that is larger
size_t lds_two(int sb, int dh) {
    const size_t panels = (size_t)(2 * (sb + 1) * (dh + 1) + 64) * 4, merge = (size_t)2 * 64 * (16 * ((dh + 31) / 32) + 2) * 4;
    return panels > merge ? panels : merge;
}
size_t lds_dkv(int sb, int dh) {
    const size_t panels = (size_t)(2 * (sb + 1) * (dh + 1) + 2 * sb + 64) * 4, merge = (size_t)2 * 64 * (32 * ((dh + 31) / 32)) * 4;
    return panels > merge ? panels : merge;
}

int check_attn(const char* fn, int B, int heads, int N, int M, int dh, int q_stride, int kv_stride) {
    TMF_REQUIRE(B > 0 && heads > 0 && N > 0 && M > 0, TMF_E_SHAPE, "%s: non-positive dimension", fn);
    TMF_REQUIRE(dh == 8 || dh == 16 || dh == 32 || dh == 64, TMF_E_SHAPE, "%s: dim_head %d not in {8,16,32,64}", fn, dh);
    TMF_REQUIRE(q_stride >= heads * dh && kv_stride >= heads * dh, TMF_E_SHAPE, "%s: row stride smaller than heads*dh", fn);
    TMF_REQUIRE(heads <= 65535 && B <= 65535, TMF_E_SHAPE, "%s: grid dimension overflow", fn);
    TMF_REQUIRE(q_stride % 4 == 0 && kv_stride % 4 == 0, TMF_E_SHAPE, "%s: row strides must be multiples of 4 floats", fn);
    return TMF_OK;
}

}  // namespace

#define TMF_DH_SWITCH(dh, CALL)            \
    switch (dh) {                          \
        case 8:  { CALL(8);  break; }      \
        case 16: { CALL(16); break; }      \
        case 32: { CALL(32); break; }      \
        default: { CALL(64); break; }      \
    }

extern "C" int tmf_xattn_fwd(const float* q, const float* k, const float* v, float* out, float* lse,
                             int B, int heads, int N, int M, int dh, int q_stride, int kv_stride, float scale,
                             void* stream) {
    TMF_REQUIRE_PTR(q); TMF_REQUIRE_PTR(k); TMF_REQUIRE_PTR(v); TMF_REQUIRE_PTR(out); TMF_REQUIRE_PTR(lse);
    int rc = check_attn("tmf_xattn_fwd", B, heads, N, M, dh, q_stride, kv_stride);
    if (rc) return rc;
    TMF_REQUIRE_ALIGNED(out); TMF_REQUIRE_ALIGNED(q); TMF_REQUIRE_ALIGNED(k); TMF_REQUIRE_ALIGNED(v);
    TMF_REQUIRE((heads * dh) % 4 == 0, TMF_E_SHAPE, "tmf_xattn_fwd: heads*dh must be a multiple of 4");
    const int sb = resident_rows(M, dh);
    const size_t lds = lds_two(sb, dh);
    dim3 grid(tmf_cdiv(N, 64), heads, B), block(256);
#define CALL(DH)                                                                                     \
    auto kf = xattn_fwd_kernel<DH>;                                                                  \
    if ((rc = tmf_allow_lds(kf, lds, "tmf_xattn_fwd"))) return rc;                                   \
    hipLaunchKernelGGL(kf, grid, block, lds, (hipStream_t)stream, q, k, v, out, lse, heads, N, M,    \
                       q_stride, kv_stride, scale, sb);
    TMF_DH_SWITCH(dh, CALL)
#undef CALL
    return tmf_launch_result("tmf_xattn_fwd");
}

extern "C" int tmf_xattn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse,
                             const float* dout, float* dq, float* dk, float* dv,
                             int B, int heads, int N, int M, int dh, int q_stride, int kv_stride, int dkv_stride,
                             float scale, void* stream) {
    TMF_REQUIRE_PTR(q); TMF_REQUIRE_PTR(k); TMF_REQUIRE_PTR(v); TMF_REQUIRE_PTR(out); TMF_REQUIRE_PTR(lse);
    TMF_REQUIRE_PTR(dout); TMF_REQUIRE_PTR(dq); TMF_REQUIRE_PTR(dk); TMF_REQUIRE_PTR(dv);
    int rc = check_attn("tmf_xattn_bwd", B, heads, N, M, dh, q_stride, kv_stride);
    if (rc) return rc;
    TMF_REQUIRE(dkv_stride >= heads * dh && dkv_stride % 4 == 0 && (heads * dh) % 4 == 0, TMF_E_SHAPE,
                "tmf_xattn_bwd: dkv_stride must be a multiple of 4 and >= heads*dh");
    TMF_REQUIRE_ALIGNED(dq); TMF_REQUIRE_ALIGNED(dk); TMF_REQUIRE_ALIGNED(dv);
    const int sbk = resident_rows(M, dh), sbq = resident_rows(N, dh);
    const size_t lds_q = lds_two(sbk, dh), lds_kv = lds_dkv(sbq, dh);
    hipStream_t s = (hipStream_t)stream;
#define CALL(DH)                                                                                          \
    auto k1 = xattn_bwd_dq_kernel<DH>;                                                                    \
    auto k2 = xattn_bwd_dkv_kernel<DH>;                                                                   \
    if ((rc = tmf_allow_lds(k1, lds_q, "tmf_xattn_bwd"))) return rc;                                      \
    if ((rc = tmf_allow_lds(k2, lds_kv, "tmf_xattn_bwd"))) return rc;                                     \
    hipLaunchKernelGGL(k1, dim3(tmf_cdiv(N, 64), heads, B), dim3(256), lds_q, s, q, k, v, out, lse, dout, dq, \
                       heads, N, M, q_stride, kv_stride, scale, sbk);                                     \
    if ((rc = tmf_launch_result("tmf_xattn_bwd(dq)"))) return rc;                                         \
    hipLaunchKernelGGL(k2, dim3(tmf_cdiv(M, 64), heads, B), dim3(256), lds_kv, s, q, k, v, out, lse, dout, dk, dv, \
                       heads, N, M, q_stride, kv_stride, dkv_stride, scale, sbq);
    TMF_DH_SWITCH(dh, CALL)
#undef CALL
    return tmf_launch_result("tmf_xattn_bwd(dkv)");
}
