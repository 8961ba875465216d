// valu_cost.hip — what does one more vector / LDS / LDS-DMA instruction cost a wave that streams fp32 MFMAs (gfx950)?
// One wave per SIMD (256 threads, 1 workgroup per CU) or two (512 threads): per iteration 32 x v_mfma_f32_32x32x2_f32 (or 64 x
// v_mfma_f32_16x16x4_f32 — the same 2048 matrix cycles) on 8 accumulators, with NPER extra instructions of kind X after every
// MFMA (PLACE 0) or all of them in one block in front of the MFMAs (PLACE 1).  Output: shader cycles per iteration (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_cost.hip -o tools/microbench/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { X_NONE, X_ADD, X_PKADD, X_FMA, X_PKFMA, X_MOV, X_DSR128, X_DSR64, X_DSR32, X_DMA, X_DSW128, X_PKMUL, X_GLD, X_NKINDS };
static const char* xname[] = {"none", "v_add_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "ds_read_b128", "ds_read_b64",
                              "ds_read_b32", "buffer_load_dwordx4 lds", "ds_write_b128", "v_pk_mul_f32", "buffer_load_dwordx4"};

template <int X>
__device__ __forceinline__ void extra(float (&s)[8], f32x2 (&p)[8], f32x4 (&q)[4], int j, unsigned ldsaddr, i32x4 rsrc, int voff, unsigned ldsdma) {
    const int r = j & 7;
    if (X == X_ADD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s[r]) : "v"(s[r]), "v"(s[(r + 3) & 7]));
    if (X == X_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[r]) : "v"(s[(r + 3) & 7]), "v"(s[(r + 5) & 7]));
    if (X == X_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(s[r]) : "v"(s[(r + 3) & 7]));
    if (X == X_PKADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 3) & 7]));
    if (X == X_PKMUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 3) & 7]));
    if (X == X_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[r]) : "v"(p[(r + 3) & 7]), "v"(p[(r + 5) & 7]));
    if (X == X_DSR128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[j & 3]) : "v"(ldsaddr), "n"(0));
    if (X == X_DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(p[r]) : "v"(ldsaddr));
    if (X == X_DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(s[r]) : "v"(ldsaddr));
    if (X == X_DSW128) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsaddr), "v"(q[j & 3]) : "memory");
    if (X == X_GLD) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q[j & 3]) : "v"(voff), "s"(rsrc) : "memory");
    if (X == X_DMA) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(0), "s"(ldsdma) : "memory");
}

// MT 0: 32x32x2 (32 per iteration), 1: 16x16x4 (64 per iteration, 4-register accumulators)
template <int MT, int X, int NPER, int PLACE, int NTHR, int ROLE>
__global__ __launch_bounds__(NTHR) void k(float* out, long long* cyc, const float* gsrc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += NTHR) lds[i] = i * 1e-6f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds;
    const unsigned ldsaddr = lds0 + wave * 4096 + lane * 16;
    const unsigned ldsdma = __builtin_amdgcn_readfirstlane(lds0 + 65536 + wave * 1024);
    const unsigned long long ga = (unsigned long long)(gsrc + blockIdx.x * 4096);
    const i32x4 rsrc = {(int)(unsigned)ga, (int)((unsigned)(ga >> 32) & 0xFFFFu), 16384, 0x00020000};
    const int voff = lane * 16;
    f32x16 acc[8];
    f32x4 acc4[32];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 32; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s[8]; f32x2 p[8]; f32x4 q[4];
    for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{s[i], s[i] + 1.f}; }
    for (int i = 0; i < 4; ++i) q[i] = f32x4{s[i], 1.f, 2.f, 3.f};
    float a = threadIdx.x, b = 1.f;
    // ROLE 0: every wave does MFMAs + extras.  ROLE 1 (512 threads): waves 0-3 only MFMAs, waves 4-7 only the extras (same count)
    const bool do_m = ROLE == 0 || wave < 4, do_x = ROLE == 0 || wave >= 4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (PLACE == 1 && do_x) {
#pragma unroll
            for (int j = 0; j < 32 * NPER; ++j) extra<X>(s, p, q, j, ldsaddr, rsrc, voff, ldsdma);
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (do_m) {
                if (MT == 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
                else {
                    acc4[(2 * m) & 31] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * m) & 31], 0, 0, 0);
                    acc4[(2 * m + 1) & 31] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * m + 1) & 31], 0, 0, 0);
                }
            }
            if (PLACE == 0 && do_x) {
#pragma unroll
                for (int j = 0; j < NPER; ++j) extra<X>(s, p, q, m * NPER + j, ldsaddr, rsrc, voff, ldsdma);
            }
        }
        if (X == X_DSR128 || X == X_DSR64 || X == X_DSR32 || X == X_DSW128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (X == X_DMA || X == X_GLD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 32; ++i) sum += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    for (int i = 0; i < 8; ++i) sum += s[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i) sum += q[i][0] + q[i][3];
    out[blockIdx.x * NTHR + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MT, int X, int NPER, int PLACE, int NTHR, int ROLE>
void run(float* out, long long* cyc, float* gsrc) {
    const int iters = 400, blocks = 256; const size_t ldsb = 100 * 1024;
    auto kern = k<MT, X, NPER, PLACE, NTHR, ROLE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), ldsb, 0, out, cyc, gsrc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), ldsb, 0, out, cyc, gsrc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = NTHR / 64;
    double avg = 0; int n = 0;
    for (int bq = 0; bq < 256; ++bq) for (int w = 0; w < nw; ++w) { avg += h[bq * 8 + w]; ++n; }
    avg /= n;
    const double per = avg / iters;
    const int waves_per_simd = NTHR / 256;
    const int mf_waves = ROLE == 0 ? waves_per_simd : 1;
    // per SIMD and iteration: matrix cycles 2048 x mf_waves
    printf("%-14s %-24s n/MFMA %d  %s  %d thr %s : %8.0f cyc/iter/wave = %6.1f extra per instruction (per SIMD: %5.0f, matrix %d)  %.3f ms\n",
           MT == 0 ? "32x32x2" : "16x16x4(x2)", xname[X], NPER, PLACE ? "block " : "spread", NTHR, ROLE ? "split" : "same ",
           per, X == X_NONE ? 0.0 : (per * (ROLE == 0 ? 1 : 1) - 2048.0 * mf_waves) / (32.0 * NPER) / (ROLE == 0 ? waves_per_simd : 1),
           per, 2048 * mf_waves, ms);
}

#define ROWS(MT, X) \
    run<MT, X, 1, 0, 256, 0>(out, cyc, gsrc); run<MT, X, 2, 0, 256, 0>(out, cyc, gsrc); run<MT, X, 4, 0, 256, 0>(out, cyc, gsrc); \
    run<MT, X, 2, 1, 256, 0>(out, cyc, gsrc); run<MT, X, 2, 0, 512, 0>(out, cyc, gsrc); run<MT, X, 2, 1, 512, 0>(out, cyc, gsrc); \
    run<MT, X, 2, 1, 512, 1>(out, cyc, gsrc);

int main() {
    float *out, *gsrc; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8); hipMalloc(&gsrc, 256 * 4096 * 4);
    hipMemset(gsrc, 0, 256 * 4096 * 4);
    run<0, X_NONE, 1, 0, 256, 0>(out, cyc, gsrc);
    run<0, X_NONE, 1, 0, 512, 0>(out, cyc, gsrc);
    run<0, X_GLD, 1, 0, 256, 0>(out, cyc, gsrc); run<0, X_GLD, 1, 1, 256, 0>(out, cyc, gsrc); run<0, X_GLD, 1, 0, 512, 0>(out, cyc, gsrc);
    run<0, X_DMA, 1, 0, 256, 0>(out, cyc, gsrc); run<0, X_DMA, 1, 1, 256, 0>(out, cyc, gsrc); run<0, X_DMA, 1, 0, 512, 0>(out, cyc, gsrc);
#ifdef FULL_TABLE
    run<1, X_NONE, 1, 0, 256, 0>(out, cyc, gsrc);
    run<1, X_NONE, 1, 0, 512, 0>(out, cyc, gsrc);
    ROWS(0, X_ADD) ROWS(0, X_PKADD) ROWS(0, X_FMA) ROWS(0, X_PKFMA) ROWS(0, X_PKMUL) ROWS(0, X_MOV)
    ROWS(0, X_DSR128) ROWS(0, X_DSR64) ROWS(0, X_DSR32) ROWS(0, X_DSW128)
    run<0, X_DMA, 1, 1, 512, 1>(out, cyc, gsrc);
    ROWS(1, X_ADD) ROWS(1, X_PKADD) ROWS(1, X_DSR128)
#endif
    return 0;
}
