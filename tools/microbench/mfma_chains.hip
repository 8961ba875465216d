#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// PAT: 0 = rotate over NACC accumulators every MFMA; 1 = runs of 4 on one accumulator, then the next
template <int NACC, int PAT>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ float lds[];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x, b = 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int j = PAT == 0 ? m % NACC : (m / 4) % NACC;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int PAT>
void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 3000, blocks = 256; const size_t ldsb = 150 * 1024;
    hipFuncSetAttribute((const void*)k<NACC, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NACC, PAT>), dim3(blocks), dim3(threads), ldsb, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NACC, PAT>), dim3(blocks), dim3(threads), ldsb, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = (double)blocks * (threads / 64) * iters * 32 * 4096.0;
    printf("%-44s %d waves/SIMD  %.1f TF\n", name, threads / 256, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int t : {256, 512}) {
        run<1, 0>("1 accumulator", t);
        run<2, 0>("2 accumulators alternating", t);
        run<2, 1>("2 accumulators, runs of 4", t);
        run<4, 0>("4 accumulators rotating", t);
        run<4, 1>("4 accumulators, runs of 4", t);
        run<8, 0>("8 accumulators rotating", t);
    }
    return 0;
}
