// mix_cost.hip — the instruction mix of one 8-channel chunk of the Winograd forward kernel beside its MFMAs, two ways (gfx950):
//   B: one wave per SIMD (256 threads): 64 x v_mfma_f32_32x32x2_f32 + 96 packed adds + 32 ds_read_b128 + 16 buffer_load_dwordx4
//      + 6 LDS-DMA copies, one barrier                                   (what conv3d_wino_p_kernel issues per chunk and wave)
//   D: two waves per SIMD (512 threads), each 64 x v_mfma_f32_16x16x4_f32 + 48 packed adds + 32 ds_read_b64 + 16 buffer_load_dwordx4
//      + 3 LDS-DMA copies, one barrier                                   (the same chunk of a 32-tile x 32-channel item split by tiles)
// Both: 4096 matrix cycles per SIMD and chunk.  Output: shader cycles and time per chunk.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mix_cost.hip -o tools/microbench/mix_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WHAT>       // WHAT bits: 1 packed adds, 2 LDS reads, 4 weight loads, 8 LDS-DMA
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void k(float* out, long long* cyc, const float* gsrc, int iters) {
    extern __shared__ float lds[];
    constexpr int NTHR = MODE == 0 ? 256 : 512;
    for (int i = threadIdx.x; i < 16384; i += NTHR) lds[i] = i * 1e-6f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds;
    const unsigned ldsaddr = lds0 + wave * 2048 + lane * 16;
    const unsigned ldsdma = __builtin_amdgcn_readfirstlane(lds0 + 65536 + wave * 4096);
    const unsigned long long ga = (unsigned long long)(gsrc + (blockIdx.x & 63) * 65536);
    const i32x4 rsrc = {(int)(unsigned)ga, (int)((unsigned)(ga >> 32) & 0xFFFFu), 262144, 0x00020000};
    const int voff = lane * 16;
    f32x16 acc[16];
    f32x4 acc4[32];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 32; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 p[8]; f32x4 q[8]; f32x4 bw[16];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f}; q[i] = f32x4{1.f, 2.f, 3.f, (float)i}; }
    for (int i = 0; i < 16; ++i) bw[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    float a = threadIdx.x, b = 1.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // 8 groups of 8 MFMAs, the other instructions spread over them
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (WHAT & 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(g * 4 + j) & 7]) : "v"(ldsaddr), "n"(0));
                    else asm volatile("ds_read_b64 %0, %1" : "=v"(p[(g * 4 + j) & 7]) : "v"(ldsaddr));
                }
            }
            if (WHAT & 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bw[g * 2 + j]) : "v"(voff), "s"(rsrc), "s"((g * 2 + j) * 1024 + (it & 15) * 16384) : "memory");
            }
            if ((WHAT & 8) && (MODE == 0 ? g < 6 : g < 3))
                asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(g * 1024 + (it & 15) * 16384), "s"(ldsdma) : "memory");
            if (WHAT & 1) {
#pragma unroll
                for (int j = 0; j < (MODE == 0 ? 12 : 6); ++j)
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[j & 7]) : "v"(p[j & 7]), "v"(p[(j + 3) & 7]));
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MODE == 0) acc[(g * 8 + m) & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(g * 8 + m) & 15], 0, 0, 0);
                else acc4[(g * 8 + m) & 31] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(g * 8 + m) & 31], 0, 0, 0);
            }
            if (WHAT & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 32; ++i) sum += acc4[i][0] + acc4[i][3];
    for (int i = 0; i < 8; ++i) sum += p[i][0] + p[i][1] + q[i][0] + q[i][3];
    for (int i = 0; i < 16; ++i) sum += bw[i][0] + bw[i][3];
    out[blockIdx.x * NTHR + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int WHAT>
void run(float* out, long long* cyc, float* gsrc) {
    const int iters = 300, blocks = 256; const size_t ldsb = 120 * 1024;
    constexpr int NTHR = MODE == 0 ? 256 : 512;
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
    printf("%s  what %2d : %7.0f cycles per chunk (4096 matrix cycles per SIMD)   %.1f ns per chunk   matrix busy %.2f\n",
           MODE == 0 ? "B 1 wave/SIMD  32x32x2" : "D 2 waves/SIMD 16x16x4", WHAT, avg / iters, ms * 1e6 / iters, 4096.0 / (avg / iters));
}
#define ALL(MODE) run<MODE, 0>(out, cyc, gsrc); run<MODE, 1>(out, cyc, gsrc); run<MODE, 2>(out, cyc, gsrc); run<MODE, 4>(out, cyc, gsrc); \
    run<MODE, 8>(out, cyc, gsrc); run<MODE, 7>(out, cyc, gsrc); run<MODE, 15>(out, cyc, gsrc);
int main() {
    float *out, *gsrc; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8); hipMalloc(&gsrc, 64 * 65536 * 4);
    hipMemset(gsrc, 0, 64 * 65536 * 4);
    ALL(0) ALL(1)
    return 0;
}
