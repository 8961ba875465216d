// pk_waw.hip — does gfx950 deliver a stale half when a single-pass VALU instruction overwrites ONE half of a register
// pair right behind the packed-fp32 instruction that wrote the pair?
//
// The sequence clang's SLP vectoriser produced in the first block's reduce pass:
//     v_pk_mul_f32 v[46:47], v[40:41], v[42:43]      ; (t0, t1) = (a0*b0, a1*b1)
//     v_mov_b32    v46, v41                          ; t0 <- a1          (half-overwrite, t0 never read in between)
//     v_pk_add_f32 v[44:45], v[44:45], v[46:47]      ; (s0, s1) += (a1, a1*b1)
// A stale half shows as s0 picking up a0*b0 (= 1000) instead of a1 (= 1) in some lanes.  Variants: NOPS s_nop states
// between the pk_mul and the mov; one wave per SIMD up to eight; alone or with a memory-streaming kernel on a second
// stream (the reduce pass failed at the TAIL of its grid, next to the other encoder's kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/pk_waw.hip -o tools/microbench/pk_waw && gpurun -- ./tools/microbench/pk_waw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 128
#define STR2(x) #x
#define STR(x) STR2(x)

template <int NOPS>
__global__ void pk_waw_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = in[4 * i + 0], a1 = in[4 * i + 1], b0 = in[4 * i + 2], b1 = in[4 * i + 3];
    float s0 = 0.f, s1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t"
            "v_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\t"
            ".rept " STR(REP) "\n\t"
            "v_pk_mul_f32 v[46:47], v[40:41], v[42:43]\n\t"
            ".rept %6\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 v46, v41\n\t"
            "v_pk_add_f32 v[44:45], v[44:45], v[46:47]\n\t"
            ".endr\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\t"
            : "+v"(s0), "+v"(s1)
            : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "n"(NOPS)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    }
    out[2 * i] = s0;
    out[2 * i + 1] = s1;
}

// Variant B: the instruction neighbourhood of the kernel that misbehaved (hipcc -S of conv1_fused_kernel<MODE_REDUCE>
// WITHOUT -fno-slp-vectorize, 14 lines around the second accumulation):
//     v_sub_f32    v54, v50, v51                        ; zs - mu
//     v_pk_mul_f32 v[48:49], v[40:41], v[54:55]         ; (xhat, gl) = (is * (zs - mu), lrm * g)
//     v_fma_f32    v41, v56, v57, v58                   ; the next window's y overwrites v41 (a SOURCE half of the pk_mul)
//     v_pk_mul_f32 v[46:47], v[48:49], v[48:49] op_sel_hi:[0,1]   ; (xhat*xhat, xhat*gl) — depends on the pk_mul above
//     v_mov_b32    v46, v49                             ; low half <- gl  (the half-overwrite; reads a fresh pk result)
//     v_pk_add_f32 v[44:45], v[44:45], v[46:47]         ; (s1, s2) += (gl, xhat*gl)
// Exact small integers: is = 2, zs - mu = 3 -> xhat = 6; lrm = 1, g = 4 -> gl = 4; expected += (4, 24).
// A stale low half gives += 36; the source overwrite (WAR) would give gl = 7 * 4 = 28.
template <int NOPS>
__global__ void pk_waw_dep_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float is = in[4 * i + 0] * 0.f + 2.f, lrm = in[4 * i + 1], g = in[4 * i + 1] * 4.f, zs = in[4 * i + 3] - 1.f, mu = 2.f;
    float s0 = 0.f, s1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v55, %4\n\tv_mov_b32 v50, %5\n\tv_mov_b32 v51, %6\n\t"
            "v_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\tv_mov_b32 v59, %3\n\t"
            "v_mov_b32 v56, 1.0\n\tv_mov_b32 v57, 2.0\n\tv_mov_b32 v58, 5.0\n\t"
            ".rept " STR(REP) "\n\t"
            "v_sub_f32 v54, v50, v51\n\t"
            "v_pk_mul_f32 v[48:49], v[40:41], v[54:55]\n\t"
            "v_fma_f32 v41, v56, v57, v58\n\t"
            "v_pk_mul_f32 v[46:47], v[48:49], v[48:49] op_sel_hi:[0,1]\n\t"
            ".rept %7\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 v46, v49\n\t"
            "v_pk_add_f32 v[44:45], v[44:45], v[46:47]\n\t"
            "v_mov_b32 v41, v59\n\t"
            ".endr\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\t"
            : "+v"(s0), "+v"(s1)
            : "v"(is), "v"(lrm), "v"(g), "v"(zs), "v"(mu), "n"(NOPS)
            : "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v54", "v55", "v56", "v57", "v58", "v59");
    }
    out[2 * i] = s0;
    out[2 * i + 1] = s1;
}

// Variant C: variant B's sequence in waves 0-3 of a 512-thread workgroup while waves 4-7 (the second wave of every SIMD)
// stream bf16 MFMAs back to back — the first block's kernels recompute the conv output on the matrix cores, so in the
// real kernel packed VALU work always executes beside other waves' MFMAs.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int NOPS>
__global__ __launch_bounds__(512) void pk_waw_mfma_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int i = blockIdx.x * 256 + (threadIdx.x & 255);
    if (threadIdx.x >= 256) {
        f32x16_t acc0 = {}, acc1 = {};
        bf16x8_t a = {}, b = {};
        for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        }
        if (acc0[0] + acc1[3] == 12345.f) out[0] = acc0[1];          // keep the loop alive
        return;
    }
    float is = in[4 * i + 0] * 0.f + 2.f, lrm = in[4 * i + 1], g = in[4 * i + 1] * 4.f, zs = in[4 * i + 3] - 1.f, mu = 2.f;
    float s0 = 0.f, s1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v55, %4\n\tv_mov_b32 v50, %5\n\tv_mov_b32 v51, %6\n\t"
            "v_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\tv_mov_b32 v59, %3\n\t"
            "v_mov_b32 v56, 1.0\n\tv_mov_b32 v57, 2.0\n\tv_mov_b32 v58, 5.0\n\t"
            ".rept " STR(REP) "\n\t"
            "v_sub_f32 v54, v50, v51\n\t"
            "v_pk_mul_f32 v[48:49], v[40:41], v[54:55]\n\t"
            "v_fma_f32 v41, v56, v57, v58\n\t"
            "v_pk_mul_f32 v[46:47], v[48:49], v[48:49] op_sel_hi:[0,1]\n\t"
            ".rept %7\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 v46, v49\n\t"
            "v_pk_add_f32 v[44:45], v[44:45], v[46:47]\n\t"
            "v_mov_b32 v41, v59\n\t"
            ".endr\n\t"
            "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\t"
            : "+v"(s0), "+v"(s1)
            : "v"(is), "v"(lrm), "v"(g), "v"(zs), "v"(mu), "n"(NOPS)
            : "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v54", "v55", "v56", "v57", "v58", "v59");
    }
    out[2 * i] = s0;
    out[2 * i + 1] = s1;
}

__global__ void stream_kernel(float* buf, size_t n, int rounds) {        // keeps the memory system and the other SIMDs busy
    for (int r = 0; r < rounds; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            buf[i] = buf[i] * 1.0001f + 1.f;
}

template <int NOPS, bool DEP = false>
long run(int blocks, int threads, int iters, bool neighbour, float* in, float* out, float* h_out, float* big, size_t nbig,
         hipStream_t s0, hipStream_t s1) {
    const int n = blocks * threads;
    long bad = 0;
    for (int rep = 0; rep < 20; ++rep) {
        if (neighbour) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s1, big, nbig, 2);
        if (DEP) hipLaunchKernelGGL(pk_waw_dep_kernel<NOPS>, dim3(blocks), dim3(threads), 0, s0, in, out, iters);
        else hipLaunchKernelGGL(pk_waw_kernel<NOPS>, dim3(blocks), dim3(threads), 0, s0, in, out, iters);
        hipStreamSynchronize(s0);
        hipMemcpy(h_out, out, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost);
        const float want0 = (DEP ? 4.0f : 1.0f) * REP * iters, want1 = (DEP ? 24.0f : 6.0f) * REP * iters;   // exact in fp32
        for (int i = 0; i < n; ++i) bad += (h_out[2 * i] != want0) + (h_out[2 * i + 1] != want1);
    }
    hipDeviceSynchronize();
    return bad;
}

int main() {
    const int maxn = 2048 * 512;
    float *in, *out, *big;
    const size_t nbig = (size_t)1 << 28;
    hipMalloc(&in, (size_t)maxn * 4 * sizeof(float));
    hipMalloc(&out, (size_t)maxn * 2 * sizeof(float));
    hipMalloc(&big, nbig * sizeof(float));
    hipMemset(big, 0, nbig * sizeof(float));
    float* h = (float*)malloc((size_t)maxn * 4 * sizeof(float));
    for (int i = 0; i < maxn; ++i) { h[4 * i] = 100.f; h[4 * i + 1] = 1.f; h[4 * i + 2] = 10.f; h[4 * i + 3] = 6.f; }   // a0*b0 = 1000
    hipMemcpy(in, h, (size_t)maxn * 4 * sizeof(float), hipMemcpyHostToDevice);
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    printf("%-28s %10s %10s %10s %10s\n", "configuration", "nops=0", "nops=1", "nops=2", "nops=4");
    struct { const char* name; int blocks, threads; bool nb; } cfg[] = {
        {"1 wave / SIMD, 256 CUs", 256, 256, false},      {"2 waves / SIMD", 256, 512, false},
        {"8 waves / SIMD", 1024, 512, false},              {"tail: 8 workgroups only", 8, 256, false},
        {"1 wave / SIMD + neighbour", 256, 256, true},     {"8 waves / SIMD + neighbour", 1024, 512, true},
        {"tail: 8 workgroups + neighbour", 8, 256, true},  {"single wave + neighbour", 1, 64, true}};
    long total = 0;
    for (auto& c : cfg) {
        const long b0 = run<0>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b1 = run<1>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b2 = run<2>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b4 = run<4>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        printf("%-28s %10ld %10ld %10ld %10ld   (wrong lanes of %d x 20 launches x %d sequences)\n", c.name, b0, b1, b2, b4,
               c.blocks * c.threads, 64 * REP);
        total += b0 + b1 + b2 + b4;
    }
    printf("variant B (dependent pk_mul chain, op_sel_hi, source overwritten behind the first pk_mul):\n");
    for (auto& c : cfg) {
        const long b0 = run<0, true>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b1 = run<1, true>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b2 = run<2, true>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        const long b4 = run<4, true>(c.blocks, c.threads, 64, c.nb, in, out, h, big, nbig, s0, s1);
        printf("%-28s %10ld %10ld %10ld %10ld\n", c.name, b0, b1, b2, b4);
        total += b0 + b1 + b2 + b4;
    }
    printf("variant C (variant B beside bf16 MFMA waves on the same SIMDs; 256 checked lanes per 512-thread workgroup):\n");
    {
        long tc = 0;
        for (int nb = 0; nb < 2; ++nb)
            for (int blocks : {256, 1024, 8}) {
                long bad[2] = {0, 0};
                for (int v = 0; v < 2; ++v)
                    for (int rep = 0; rep < 20; ++rep) {
                        if (nb) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s1, big, nbig, 2);
                        if (v == 0) hipLaunchKernelGGL(pk_waw_mfma_kernel<0>, dim3(blocks), dim3(512), 0, s0, in, out, 64);
                        else hipLaunchKernelGGL(pk_waw_mfma_kernel<2>, dim3(blocks), dim3(512), 0, s0, in, out, 64);
                        hipStreamSynchronize(s0);
                        hipMemcpy(h, out, (size_t)blocks * 256 * 2 * sizeof(float), hipMemcpyDeviceToHost);
                        for (int i = 0; i < blocks * 256; ++i)
                            bad[v] += (h[2 * i] != 4.0f * REP * 64) + (h[2 * i + 1] != 24.0f * REP * 64);
                    }
                hipDeviceSynchronize();
                printf("%4d workgroups%s   nops=0: %ld   nops=2: %ld\n", blocks, nb ? " + neighbour" : "            ", bad[0], bad[1]);
                tc += bad[0] + bad[1];
            }
        total += tc;
    }
    printf("total wrong results: %ld\n", total);
    return 0;
}
