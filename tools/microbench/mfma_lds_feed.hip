#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: registers only; 1: + one ds_read_b128 (A) and 2 ds_read_b32... per 8 MFMAs, prefetched one group ahead
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 24000; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* pa = lds + (threadIdx.x & 63) * 12;
    const float* pb = lds + 12000 + (threadIdx.x & 31);
    f32x4 a[2][4]; float b[2][8];
    auto fetch = [&](int buf, int it) {
        const int o = it * 12;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[buf][i] = *reinterpret_cast<const f32x4*>(pa + i * 480 + o);
#pragma unroll
        for (int q = 0; q < 8; ++q) b[buf][q] = pb[it * 512 + q * 32];
    };
    if (MODE == 1) fetch(0, 0);
    else { for (int i = 0; i < 4; ++i) a[0][i] = f32x4{1.f, 2.f, 3.f, 4.f}; for (int q = 0; q < 8; ++q) b[0][q] = q; }
    auto mm = [&](int cur) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][kk], b[cur][kk * 2 + j], acc[i * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int t9 = 0;
    for (int it = 0; it < iters; it += 2) {
        if (MODE == 1) fetch(1, t9 + 1);
        mm(0);
        if (MODE == 1) fetch(0, t9 + 2);
        mm(MODE == 1 ? 1 : 0);
        t9 += 2; if (t9 >= 7) t9 = 0;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, size_t ldsb, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 3000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), ldsb, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), ldsb, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-60s %.3f ms  %.1f TF\n", name, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0>("regs only, 1 WG/CU (150 KB LDS), 4 waves = 1 per SIMD", 150 * 1024, 256);
    run<0>("regs only, 2 WG/CU (76 KB LDS)", 76 * 1024, 512);
    run<1>("LDS operands prefetched, 1 WG/CU, 1 wave per SIMD", 150 * 1024, 256);
    run<1>("LDS operands prefetched, 2 WG/CU", 96 * 1024 > 76 * 1024 ? 76 * 1024 + 24000 * 0 : 0, 512);
    return 0;
}
