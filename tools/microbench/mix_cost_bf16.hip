// mix_cost_bf16.hip — would the Winograd forward's chunk run faster on the bf16 matrix pipe with EXACT 3-way operand splits?
// (gfx950; VERDICT r05 item 1: "measure before any kernel work")
//
// An fp32 operand x splits exactly into three bf16 numbers (8 + 8 + 8 significand bits): h = x & 0xffff0000, r = x - h, m = r & 0xffff0000,
// l = r - m.  x·u = six partial products of order >= 2^-16 (hh, hm, mh, hl, lh, mm) + three dropped ones <= 2^-24.  The weights are split once
// per step by the pack kernel; the transformed INPUT must be split in the main loop: per PAIR of values 4 v_and + 4 v_sub + 3 v_perm = 11
// vector instructions.  One 16-channel chunk of a 32-tile x 32-channel x 16-position item and wave:
//   C: 96 x v_mfma_f32_32x32x16_bf16 (3 072 matrix cycles) + 384 v_add_f32 (input transform; packed fp32 is an anti-lever beside bf16 MFMAs,
//      bit 16 selects 192 v_pk_add_f32 instead) + 64 ds_read_b128 + 48 buffer_load_dwordx4 (3 weight parts) + 12 LDS-DMA copies
//      + 704 split instructions, one barrier — one wave per SIMD;
//   E: the same chunk over TWO waves per SIMD, each 8 positions (128 accumulator registers): 48 MFMAs + half of everything else per wave;
//   F: an fp16 2-way split (22-bit operands — NOT fp32-exact; for the table only): 48 x v_mfma_f32_32x32x16_f16, split = 2 v_cvt_pkrtz +
//      2 v_fma_mix per pair (256 per chunk), 32 weight loads.
// Reference: the fp32 chunk of mix_cost.hip covers 8 channels in 5 089 cycles, i.e. 10 178 cycles per 16 channels.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mix_cost_bf16.hip -o tools/microbench/mix_cost_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// WHAT bits: 1 transform adds, 2 LDS reads, 4 weight loads, 8 LDS-DMA, 16 transform adds PACKED, 32 split, 64 MFMAs
template <int MODE, int WHAT>       // MODE 0 = C (1 wave/SIMD), 1 = E (2 waves/SIMD), 2 = F (fp16 2-way, 1 wave/SIMD)
__global__ __launch_bounds__(MODE == 1 ? 512 : 256) void k(float* out, long long* cyc, const float* gsrc, int iters) {
    extern __shared__ float lds[];
    constexpr int NTHR = MODE == 1 ? 512 : 256;
    constexpr int NPOS = MODE == 1 ? 8 : 16;              // positions per wave and chunk
    constexpr int NMF = MODE == 2 ? 3 : 6;                // MFMAs per position
    constexpr int NWL = MODE == 2 ? 2 : 3;                // weight loads per position
    for (int i = threadIdx.x; i < 16384; i += NTHR) lds[i] = i * 1e-6f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds;
    const unsigned ldsaddr = lds0 + wave * 2048 + lane * 16;
    const unsigned ldsdma = __builtin_amdgcn_readfirstlane(lds0 + 65536 + wave * 4096);
    const unsigned long long ga = (unsigned long long)(gsrc + (blockIdx.x & 63) * 65536);
    const i32x4 rsrc = {(int)(unsigned)ga, (int)((unsigned)(ga >> 32) & 0xFFFFu), 262144, 0x00020000};
    const int voff = lane * 16;
    f32x16 acc[NPOS];
    for (int i = 0; i < NPOS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float t[8]; f32x2 p[8]; f32x4 q[8]; i32x4 bw[6];
    unsigned ah[4], am[4], al[4];                          // the three parts of 8 transformed values, packed pairs
    for (int i = 0; i < 8; ++i) { t[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{t[i], 1.f}; q[i] = f32x4{1.f, 2.f, 3.f, (float)i}; }
    for (int i = 0; i < 6; ++i) bw[i] = i32x4{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    for (int i = 0; i < 4; ++i) { ah[i] = 0x3f803f80u; am[i] = 0x3b803b80u; al[i] = 0x37803780u; }
    const unsigned msk = 0xffff0000u, sel = 0x07060302u;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NPOS; ++g) {                   // one position: its operands are produced while the previous one multiplies
            if (WHAT & 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(g * 4 + j) & 7]) : "v"(ldsaddr), "n"(0));
            }
            if (WHAT & 4) {
#pragma unroll
                for (int j = 0; j < NWL; ++j)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bw[(g & 1) * 3 + j]) : "v"(voff), "s"(rsrc), "s"((g * 3 + j) * 1024 + (it & 3) * 65536) : "memory");
            }
            if ((WHAT & 8) && (MODE == 1 ? g < 6 : g < 12))
                asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(g * 1024 + (it & 15) * 16384), "s"(ldsdma) : "memory");
            // the rest in NMF slices, one ahead of each MFMA
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                if (WHAT & 16) {
#pragma unroll
                    for (int j = 0; j < 12 / NMF; ++j) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[(m * 2 + j) & 7]) : "v"(p[(m * 2 + j) & 7]), "v"(p[(m * 2 + j + 3) & 7]));
                } else if (WHAT & 1) {
#pragma unroll
                    for (int j = 0; j < 24 / NMF; ++j) asm volatile("v_add_f32 %0, %1, %2" : "=v"(t[(m * 4 + j) & 7]) : "v"(t[(m * 4 + j) & 7]), "v"(t[(m * 4 + j + 3) & 7]));
                }
                if (WHAT & 32) {
                    if (MODE != 2) {
                        // bf16 3-way: 4 pairs per position = 44 instructions; slices: pairs 0..3 over the first four MFMA slots
                        if (m < 4) {
                            float a = t[2 * m], b = t[2 * m + 1], ha, hb, ra, rb, ma, mb, la, lb;
                            asm volatile("v_and_b32 %3, %11, %13\n\tv_and_b32 %4, %12, %13\n\tv_perm_b32 %0, %12, %11, %14\n\t"
                                         "v_sub_f32 %5, %11, %3\n\tv_sub_f32 %6, %12, %4\n\t"
                                         "v_and_b32 %7, %5, %13\n\tv_and_b32 %8, %6, %13\n\tv_perm_b32 %1, %6, %5, %14\n\t"
                                         "v_sub_f32 %9, %5, %7\n\tv_sub_f32 %10, %6, %8\n\tv_perm_b32 %2, %10, %9, %14"
                                         : "=&v"(ah[m]), "=&v"(am[m]), "=&v"(al[m]), "=&v"(ha), "=&v"(hb), "=&v"(ra), "=&v"(rb), "=&v"(ma), "=&v"(mb), "=&v"(la), "=&v"(lb)
                                         : "v"(a), "v"(b), "v"(msk), "v"(sel));
                        }
                    } else {
                        // fp16 2-way: 4 pairs per position = 16 instructions; slices: pairs {0,1}, {2,3}, none
                        if (m < 2) {
#pragma unroll
                            for (int pp = 2 * m; pp < 2 * m + 2; ++pp) {
                                float a = t[2 * pp], b = t[2 * pp + 1], ra, rb;
                                asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n\ts_nop 0\n\tv_fma_mix_f32 %2, -%0, 1.0, %4 op_sel_hi:[1,0,0]\n\t"
                                             "v_fma_mix_f32 %3, -%0, 1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_cvt_pkrtz_f16_f32 %1, %2, %3"
                                             : "=&v"(ah[pp]), "=&v"(al[pp]), "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
                            }
                        }
                    }
                }
                if (WHAT & 64) {
                    if (MODE != 2) {
                        // hh, hm, mh, hl, lh, mm
                        const i32x4 av = m == 0 || m == 1 || m == 3 ? i32x4{(int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]}
                                       : m == 2 || m == 5           ? i32x4{(int)am[0], (int)am[1], (int)am[2], (int)am[3]}
                                                                     : i32x4{(int)al[0], (int)al[1], (int)al[2], (int)al[3]};
                        const i32x4 bv = bw[(g & 1) * 3 + (m == 0 || m == 2 || m == 4 ? 0 : m == 1 || m == 5 ? 1 : 2)];
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[g], 0, 0, 0);
                    } else {
                        const i32x4 av = m == 0 || m == 1 ? i32x4{(int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]} : i32x4{(int)al[0], (int)al[1], (int)al[2], (int)al[3]};
                        const i32x4 bv = bw[(g & 1) * 3 + (m == 1 ? 1 : 0)];
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc[g], 0, 0, 0);
                    }
                }
            }
            if (WHAT & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < NPOS; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += t[i] + p[i][0] + p[i][1] + q[i][0] + q[i][3];
    for (int i = 0; i < 6; ++i) sum += (float)bw[i][0] + (float)bw[i][3];
    for (int i = 0; i < 4; ++i) sum += (float)(ah[i] ^ am[i] ^ al[i]);
    out[blockIdx.x * NTHR + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int WHAT>
void run(float* out, long long* cyc, float* gsrc) {
    const int iters = 300, blocks = 256; const size_t ldsb = 120 * 1024;
    constexpr int NTHR = MODE == 1 ? 512 : 256;
    auto kern = k<MODE, WHAT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), ldsb, 0, out, cyc, gsrc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), ldsb, 0, out, cyc, gsrc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; int n = 0;
    for (int bq = 0; bq < 256; ++bq) for (int w = 0; w < NTHR / 64; ++w) { avg += h[bq * 8 + w]; ++n; }
    avg /= n;
    const double matrix = (WHAT & 64) ? (MODE == 2 ? 1536.0 : 3072.0) : 0.0;
    printf("%s  what %3d : %7.0f cycles per 16-channel chunk (%4.0f matrix cycles per SIMD; fp32 kernel's mix: 10178)   %.1f ns per chunk   x%.2f vs fp32\n",
           MODE == 0 ? "C bf16x3 1 wave/SIMD " : MODE == 1 ? "E bf16x3 2 waves/SIMD" : "F fp16x2 1 wave/SIMD ", WHAT, avg / iters, matrix, ms * 1e6 / iters,
           10178.0 / (avg / iters));
}
#define ALL(MODE) run<MODE, 64>(out, cyc, gsrc); run<MODE, 64 + 1>(out, cyc, gsrc); run<MODE, 64 + 16>(out, cyc, gsrc); run<MODE, 64 + 2>(out, cyc, gsrc); \
    run<MODE, 64 + 4>(out, cyc, gsrc); run<MODE, 64 + 8>(out, cyc, gsrc); run<MODE, 64 + 32>(out, cyc, gsrc); run<MODE, 32>(out, cyc, gsrc); \
    run<MODE, 32 + 1>(out, cyc, gsrc); run<MODE, 64 + 32 + 1>(out, cyc, gsrc); run<MODE, 64 + 15>(out, cyc, gsrc); run<MODE, 64 + 32 + 15>(out, cyc, gsrc); \
    run<MODE, 64 + 32 + 16 + 14>(out, cyc, gsrc); run<MODE, 32 + 15>(out, cyc, gsrc);
int main() {
    float *out, *gsrc; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8); hipMalloc(&gsrc, 64 * 65536 * 4);
    hipMemset(gsrc, 0, 64 * 65536 * 4);
    ALL(0) ALL(1) ALL(2)
    return 0;
}
