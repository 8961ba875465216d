#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// waves 0-3: MFMA stream (VAR = what sits between two MFMAs); waves 4-7: a timed side loop of KIND.
// VAR 0: nothing; 1: s_nop 0; 2: compute at prio 0, side at prio 3; 3: s_nop 4 every 8 MFMAs; 4: s_sleep 1 every 32; 5 = no MFMAs at all (side alone)
template <int VAR, int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* tim, const float* gsrc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 1e-6f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (VAR == 5) return;
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
                if (VAR >= 16 && (m % (VAR & 15)) == (VAR & 15) - 1) { if ((VAR >> 4) == 1) asm volatile("s_nop 1"); if ((VAR >> 4) == 2) asm volatile("s_nop 2"); if ((VAR >> 4) == 3) asm volatile("s_nop 3"); if ((VAR >> 4) == 4) asm volatile("s_nop 4"); if ((VAR >> 4) == 8) asm volatile("s_nop 8"); if ((VAR >> 4) == 15) asm volatile("s_nop 15"); if ((VAR >> 4) == 16) asm volatile("s_nop 15\n s_nop 15"); }
                if (VAR == 1) asm volatile("s_nop 0");
                if (VAR == 3 && (m & 7) == 7) asm volatile("s_nop 4");
            }
            if (VAR == 4) __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        if (VAR == 2) __builtin_amdgcn_s_setprio(3);
        const long long t0 = __builtin_readcyclecounter();
        float v = threadIdx.x * 0.001f;
        int si = blockIdx.x;
        if (KIND == 0) {
#pragma unroll 8
            for (int i = 0; i < 2000; ++i) v = __builtin_fmaf(v, 0.999f, 0.001f);
        } else if (KIND == 1) {
#pragma unroll 8
            for (int i = 0; i < 2000; ++i) { si = si * 3 + 1; asm volatile("" : "+s"(si)); }
            v = si;
        } else if (KIND == 2) {
            int idx = threadIdx.x & 63;
            for (int i = 0; i < 200; ++i) { idx = (int)(lds[idx] * 1e6f + 0.5f) & 4095; }
            v = idx;
        } else {
            int idx = threadIdx.x & 63;
            for (int i = 0; i < 50; ++i) { idx = (int)(gsrc[idx + blockIdx.x * 4096]) & 4095; }
            v = idx;
        }
        const long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) tim[blockIdx.x * 4 + wave - 4] = t1 - t0;
        out[65536 * 4 + blockIdx.x * 256 + threadIdx.x - 256] = v;
    }
}
template <int VAR, int KIND>
void run(const char* name, float* out, long long* tim, float* gsrc) {
    const int iters = 3000, blocks = 256; const size_t ldsb = 150 * 1024;
    hipFuncSetAttribute((const void*)k<VAR, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<VAR, KIND>), dim3(blocks), dim3(512), ldsb, 0, out, tim, gsrc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<VAR, KIND>), dim3(blocks), dim3(512), ldsb, 0, out, tim, gsrc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    long long h[1024]; hipMemcpy(h, tim, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += h[i]; avg /= 1024;
    double fl = (double)blocks * 4 * iters * 32 * 32768.0;
    printf("%-46s MFMA %.1f TF   side loop %.0f cycles\n", name, VAR == 5 ? 0.0 : fl / ms / 1e9, avg);
}
int main() {
    float *out, *gsrc; long long* tim;
    hipMalloc(&out, 65536 * 8 * 4); hipMalloc(&tim, 1024 * 8); hipMalloc(&gsrc, 256 * 4096 * 4);
    float* h = new float[256 * 4096]; for (int i = 0; i < 256 * 4096; ++i) h[i] = (i * 7 + 13) & 4095;
    hipMemcpy(gsrc, h, 256 * 4096 * 4, hipMemcpyHostToDevice);
    const char* kinds[4] = {"VALU x2000", "SALU x2000", "LDS chase x200", "global chase x50"};
#define ROW(KIND) \
    printf("-- side = %s\n", kinds[KIND]); \
    run<5, KIND>(" side alone", out, tim, gsrc); \
    run<0, KIND>(" bf16 MFMA back-to-back", out, tim, gsrc); \
    run<(1 << 4) | 1, KIND>(" s_nop 1 after each MFMA", out, tim, gsrc); \
    run<(4 << 4) | 1, KIND>(" s_nop 4 after each MFMA", out, tim, gsrc);
    ROW(0) ROW(2) ROW(3)
    return 0;
}
