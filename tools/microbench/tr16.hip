// What ds_read_b64_tr_b16 returns, lane by lane (gfx950): every lane hands in the LDS address of 4 consecutive 16-bit
// elements; the program prints which of the elements each lane receives.  LDS element i holds the value i.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/tr16.hip -o tools/microbench/tr16 && tools/microbench/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    (void)hipMalloc(&d_addr, sizeof h_addr); (void)hipMalloc(&d_out, sizeof h_out);
    for (int pat = 0; pat < 2; ++pat) {
        // pattern 0: lane l reads elements 100*l .. 100*l+3 (who gets what);
        // pattern 1: a [voxel][32 channels] image, pitch 32 elements: 16-lane group g of half h reads voxels 4h..4h+3 (rows), channels 16g..16g+15
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = 100 * l;
            else { int i = l & 15, g = (l >> 4) & 1, h = l >> 5; h_addr[l] = (8 * h + (i >> 2)) * 32 + 16 * g + 4 * (i & 3); }
        }
        (void)hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        (void)hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr %5d ->", l, h_addr[l]);
            for (int j = 0; j < 4; ++j) {
                if (pat == 0) printf("  (lane %2d, e%d)", h_out[l * 4 + j] / 100, h_out[l * 4 + j] % 100);
                else printf("  (vox %2d, ch %2d)", h_out[l * 4 + j] / 32, h_out[l * 4 + j] % 32);
            }
            printf("\n");
        }
    }
    return 0;
}
