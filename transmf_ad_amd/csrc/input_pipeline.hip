// Input pipeline step ahead of the hot path, on the device.
//
// reference: datasets/ADNI.py:59-84 — per subject and modality `ScaleIntensityd` (min-max scaling of the whole volume to
// [0, 1]) and `RandFlipd(prob=0.3, spatial_axis=0)` from MONAI, run on the host by a DataLoader with num_workers=0
// (datasets/__init__.py:56), followed by `batch['MRI'].to(device)` (kfold_train_adversarial.py:106-108).  At > 400 pairs/s
// that host pipeline starves the GPU by orders of magnitude; here the raw volumes are copied to the device as they are
// (pinned staging + a copy stream: transmf_ad_amd/pipeline.py) and transformed there.
//
// MONAI (pinned by the reference's requirements.txt) is not vendored in the reference tree, so these kernels follow its
// PUBLISHED formulas — monai.transforms.ScaleIntensity(minv=0, maxv=1) = utils.rescale_array:
//     mina, maxa = arr.min(), arr.max();  mina == maxa -> arr * minv;  else (arr - mina) / (maxa - mina) * (maxv - minv) + minv
// and monai.transforms.Flip(spatial_axis=0) = torch.flip along the first spatial axis — restated in oracle/input_oracle.py.
// Both are exact in fp32 (min / max are order-independent, the division is IEEE-correctly rounded), so the GPU result is
// BIT-identical to the numpy restatement.  RandRotated / RandZoomd (ADNI.py:67-68) interpolate (MONAI: bilinear grid
// sample / trilinear zoom with its own align-corners and padding conventions) and are NOT implemented: their parity
// cannot be pinned without the library.
//
// HBM-bound streaming passes: 4 B read (min/max) + 4 B read + 4 B write (scale) per voxel.
#include "tmf_common.h"

namespace {

constexpr int MM_THREADS = 256;
constexpr int MM_BLOCKS = 64;          // partial blocks per volume

// partial[b][blk] = (min, max) over a contiguous slice of volume b.  NaNs propagate as in numpy (min / max return NaN).
__global__ __launch_bounds__(MM_THREADS) void volume_minmax_partial_kernel(const float* __restrict__ vol,
                                                                           float* __restrict__ partial, long voxels) {
    __shared__ float smin[MM_THREADS / 64], smax[MM_THREADS / 64];
    __shared__ int snan[MM_THREADS / 64];
    const int b = blockIdx.y;
    const float* src = vol + (size_t)b * voxels;
    const long per = (voxels + MM_BLOCKS - 1) / MM_BLOCKS;
    const long lo = (long)blockIdx.x * per;
    long hi = lo + per;
    if (hi > voxels) hi = voxels;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    // coalesced dword loads (a (b, slice) base is not 16-byte aligned for odd volume sizes such as 91x109x91); the
    // volumes are a few tens of MB per batch — this pass is microseconds either way
    for (long j = lo + threadIdx.x; j < hi; j += MM_THREADS) {
        const float v = src[j];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
        nan |= (v != v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
        nan |= __shfl_xor(nan, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smin[wave] = mn; smax[wave] = mx; snan[wave] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < MM_THREADS / 64; ++k) { mn = fminf(mn, smin[k]); mx = fmaxf(mx, smax[k]); nan |= snan[k]; }
        float* p = partial + ((size_t)b * MM_BLOCKS + blockIdx.x) * 2;
        p[0] = nan ? NAN : mn;
        p[1] = nan ? NAN : mx;
    }
}

// minmax[b] = reduction of the MM_BLOCKS partials (one wave per volume)
__global__ __launch_bounds__(64) void volume_minmax_final_kernel(const float* __restrict__ partial, float* __restrict__ minmax) {
    const int b = blockIdx.x;
    const float* p = partial + ((size_t)b * MM_BLOCKS + threadIdx.x) * 2;
    float mn = p[0], mx = p[1];
    int nan = (mn != mn) || (mx != mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
        nan |= __shfl_xor(nan, o);
    }
    if (threadIdx.x == 0) { minmax[2 * b] = nan ? NAN : mn; minmax[2 * b + 1] = nan ? NAN : mx; }
}

// dst[b][d][h][w] = scale(src[b][flip_b ? D-1-d : d][h][w]);  one thread per 4 consecutive w-run elements of a (b, d) plane
template <bool VEC>
__global__ __launch_bounds__(256) void scale_flip_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const float* __restrict__ minmax, const unsigned char* __restrict__ flip,
                                                         int D, long plane) {
    const int b = blockIdx.z, d = blockIdx.y;
    const int sd = (flip != nullptr && flip[b]) ? D - 1 - d : d;
    const float mn = minmax[2 * b], mx = minmax[2 * b + 1];
    const bool flat = mn == mx;                      // constant volume: arr * minv = arr * 0 (MONAI rescale_array)
    const float den = mx - mn;
    const float* s = src + ((size_t)b * D + sd) * plane;
    float* o = dst + ((size_t)b * D + d) * plane;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= plane) return;
    if (VEC) {
        f32x4 v = *reinterpret_cast<const f32x4*>(s + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = flat ? v[k] * 0.0f : __fdiv_rn(v[k] - mn, den);
        *reinterpret_cast<f32x4*>(o + i) = v;
    } else {
        const float v = s[i];
        o[i] = flat ? v * 0.0f : __fdiv_rn(v - mn, den);
    }
}

}  // namespace

extern "C" size_t tmf_scale_intensity_workspace_bytes(int B) {
    return B > 0 ? (size_t)B * MM_BLOCKS * 2 * 4 : 0;
}

extern "C" int tmf_volume_minmax(const float* vol, float* minmax, void* workspace, size_t workspace_bytes, int B, long voxels,
                                 void* stream) {
    TMF_REQUIRE_PTR(vol); TMF_REQUIRE_PTR(minmax); TMF_REQUIRE_PTR(workspace);
    TMF_REQUIRE(B > 0 && voxels > 0, TMF_E_SHAPE, "tmf_volume_minmax: B=%d voxels=%ld", B, voxels);
    TMF_REQUIRE(workspace_bytes >= tmf_scale_intensity_workspace_bytes(B), TMF_E_WORKSPACE,
                "tmf_volume_minmax: workspace %zu B < required %zu B", workspace_bytes, tmf_scale_intensity_workspace_bytes(B));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(volume_minmax_partial_kernel, dim3(MM_BLOCKS, B), dim3(MM_THREADS), 0, s, vol, (float*)workspace, voxels);
    int rc = tmf_launch_result("tmf_volume_minmax");
    if (rc) return rc;
    hipLaunchKernelGGL(volume_minmax_final_kernel, dim3(B), dim3(64), 0, s, (const float*)workspace, minmax);
    return tmf_launch_result("tmf_volume_minmax(final)");
}

extern "C" int tmf_scale_flip(const float* src, float* dst, const float* minmax, const unsigned char* flip_d,
                              int B, int D, int H, int W, void* stream) {
    TMF_REQUIRE_PTR(src); TMF_REQUIRE_PTR(dst); TMF_REQUIRE_PTR(minmax);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, TMF_E_SHAPE, "tmf_scale_flip: non-positive dimension");
    TMF_REQUIRE(src != dst || flip_d == nullptr, TMF_E_ARG, "tmf_scale_flip: in-place only without a flip");
    const long plane = (long)H * W;
    const bool vec = plane % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    hipStream_t s = (hipStream_t)stream;
    if (vec) {
        hipLaunchKernelGGL(scale_flip_kernel<true>, dim3((unsigned)tmf_cdiv(plane, 1024L), D, B), dim3(256), 0, s, src, dst,
                           minmax, flip_d, D, plane);
    } else {
        hipLaunchKernelGGL(scale_flip_kernel<false>, dim3((unsigned)tmf_cdiv(plane, 256L), D, B), dim3(256), 0, s, src, dst,
                           minmax, flip_d, D, plane);
    }
    return tmf_launch_result("tmf_scale_flip");
}
