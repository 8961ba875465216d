// Probe of `buffer_load_dwordx4 ... offen lds` on gfx950: what lands in LDS for lanes whose voffset is out of range, and is
// the scalar offset part of the range check?   hipcc --offload-arch=gfx950 -O2 blds_probe.hip -o blds_probe && ./blds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void blds16(int voff, i32x4 rsrc, int soff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base) : "memory");
}
// mode 0: voffset = 16 lane, soffset 0;  1: odd lanes voffset = 0x7FFFFFF0;  2: voffset in range, soffset = bytes (past the end)
// 3: voffset = 16 lane, soffset = 64 (shifted window, last lanes run past num_records)
__global__ void k(const unsigned* x, unsigned* out, int bytes, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned sm[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) sm[i] = 0xDEADBEEFu;
    __syncthreads();
    i32x4 rr = make_rsrc(x, bytes);
    unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm;
    int vo = 16 * l, so = 0;
    if (mode == 1 && (l & 1)) vo = 0x7FFFFFF0;
    if (mode == 2) so = bytes;
    if (mode == 3) so = 64;
    blds16(vo, rr, so, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = sm[i];
}
int main() {
    const int n = 256;                         // dwords in the buffer = 1024 B = exactly one wave copy
    std::vector<unsigned> h(n + 64);
    for (int i = 0; i < n + 64; ++i) h[i] = 1000 + i;
    unsigned *x, *o;
    hipMalloc(&x, (n + 64) * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(x, h.data(), (n + 64) * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o, n * 4, mode);
        std::vector<unsigned> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i : {0, 1, 3, 4, 5, 7, 8, 12, 236, 240, 244, 248, 252, 255}) printf(" [%d]=%u", i, r[i]);
        printf("\n");
    }
    return 0;
}
