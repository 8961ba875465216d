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
// BIT-identical to the numpy restatement.
//
// Round 3: RandRotated(range_x=0.05, prob=0.3) and RandZoomd(min_zoom=0.95, max_zoom=1, prob=0.3) (ADNI.py:67-68) as
// device kernels too.  Their published algorithm (MONAI Rotate: pull-direction rotation about the first spatial axis
// around the volume centre, bilinear, border padding; Zoom: area interpolation = adaptive average to floor(S z) per axis,
// edge padding back to the original size) is restated with a fixed fp32 operation order in oracle/input_oracle.py
// (`rotate_x`, `zoom_area`; the latter bit-identical to torch's CPU interpolate(mode="area"), the former within 2e-5 of
// torch's affine_grid + grid_sample route) and evaluated here with un-contracted __f*_rn operations in that order: the
// device results are BIT-identical to that restatement.  The random decisions (apply?, angle, factor) are inputs.
//
// HBM-bound streaming passes: 4 B read (min/max) + 4 B read + 4 B write (scale) per voxel; rotate / zoom 4 B + 4 B.
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

// dst[b][d][h][w] = bilinear sample of plane d of src[b] at the rotated point (oracle/input_oracle.py rotate_x), or a
// copy when do_rot[b] == 0.  cos_sin[b] = (cos, sin) of the angle, rounded to fp32 on the host.
__global__ __launch_bounds__(256) void rotate_x_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       const float* __restrict__ cos_sin, const unsigned char* __restrict__ do_rot,
                                                       int D, int H, int W) {
    // one rounding per written operation, as in the numpy restatement: this file is compiled with -ffp-contract=off (build.py)
    const int b = blockIdx.z, d = blockIdx.y;
    const long plane = (long)H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    const float* s = src + ((size_t)b * D + d) * plane;
    float* o = dst + ((size_t)b * D + d) * plane;
    if (!do_rot[b]) { o[i] = s[i]; return; }
    const int h = (int)(i / W), w = (int)(i - (long)h * W);
    const float cs = cos_sin[2 * b], sn = cos_sin[2 * b + 1];
    const float c1 = 0.5f * (float)(H - 1), c2 = 0.5f * (float)(W - 1);
    const float o1 = (float)h - c1, o2 = (float)w - c2;
    const float p11 = cs * o1, p12 = sn * o2, p21 = sn * o1, p22 = cs * o2;
    const float d1 = p11 - p12, d2 = p21 + p22;
    float s1 = c1 + d1;
    float s2 = c2 + d2;
    s1 = fminf(fmaxf(s1, 0.f), (float)(H - 1));
    s2 = fminf(fmaxf(s2, 0.f), (float)(W - 1));
    const float f1 = floorf(s1), f2 = floorf(s2);
    const float t1 = s1 - f1, t2 = s2 - f2;
    const float a = 1.f - t1, bb = 1.f - t2;
    const int i1 = (int)f1, i2 = (int)f2;
    const int j1 = i1 + 1 < H ? i1 + 1 : H - 1, j2 = i2 + 1 < W ? i2 + 1 : W - 1;
    const float v00 = s[(long)i1 * W + i2], v01 = s[(long)i1 * W + j2], v10 = s[(long)j1 * W + i2], v11 = s[(long)j1 * W + j2];
    const float w00 = a * bb, w01 = a * t2, w10 = t1 * bb, w11 = t1 * t2;
    const float q00 = v00 * w00, q01 = v01 * w01, q10 = v10 * w10, q11 = v11 * w11;
    float r = q00 + q01;
    r = r + q10;
    r = r + q11;
    o[i] = r;
}

// dst[b] = zoom_area(src[b]) (oracle/input_oracle.py): adaptive average to out_size[b] = (Od, Oh, Ow), edge padding back
// to (D, H, W); a copy when do_zoom[b] == 0.
__global__ __launch_bounds__(256) void zoom_area_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int* __restrict__ out_size, const unsigned char* __restrict__ do_zoom,
                                                        int D, int H, int W) {
    const int b = blockIdx.z, d = blockIdx.y;
    const long plane = (long)H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    const float* s = src + (size_t)b * D * plane;
    float* o = dst + ((size_t)b * D + d) * plane;
    if (!do_zoom[b]) { o[i] = s[(size_t)d * plane + i]; return; }
    const int h = (int)(i / W), w = (int)(i - (long)h * W);
    const int Od = out_size[3 * b], Oh = out_size[3 * b + 1], Ow = out_size[3 * b + 2];
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    const int zd = clampi(d - (D - Od) / 2, Od - 1), zh = clampi(h - (H - Oh) / 2, Oh - 1), zw = clampi(w - (W - Ow) / 2, Ow - 1);
    const int d0 = (int)(((long)zd * D) / Od), d1 = (int)((((long)zd + 1) * D + Od - 1) / Od);
    const int h0 = (int)(((long)zh * H) / Oh), h1 = (int)((((long)zh + 1) * H + Oh - 1) / Oh);
    const int w0 = (int)(((long)zw * W) / Ow), w1 = (int)((((long)zw + 1) * W + Ow - 1) / Ow);
    float acc = 0.f;
    for (int a = d0; a < d1; ++a)
        for (int bq = h0; bq < h1; ++bq)
            for (int c = w0; c < w1; ++c) acc = acc + s[((size_t)a * H + bq) * W + c];
    acc = __fdiv_rn(acc, (float)(d1 - d0));
    acc = __fdiv_rn(acc, (float)(h1 - h0));
    acc = __fdiv_rn(acc, (float)(w1 - w0));
    o[i] = acc;
}

}  // namespace

extern "C" int tmf_rotate_x(const float* src, float* dst, const float* cos_sin, const unsigned char* do_rot,
                            int B, int D, int H, int W, void* stream) {
    TMF_REQUIRE_PTR(src); TMF_REQUIRE_PTR(dst); TMF_REQUIRE_PTR(cos_sin); TMF_REQUIRE_PTR(do_rot);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535 && D <= 65535, TMF_E_SHAPE, "tmf_rotate_x: B=%d D=%d H=%d W=%d", B, D, H, W);
    TMF_REQUIRE(src != dst, TMF_E_ARG, "tmf_rotate_x: not in place");
    hipLaunchKernelGGL(rotate_x_kernel, dim3((unsigned)tmf_cdiv((long)H * W, 256L), D, B), dim3(256), 0, (hipStream_t)stream,
                       src, dst, cos_sin, do_rot, D, H, W);
    return tmf_launch_result("tmf_rotate_x");
}

extern "C" int tmf_zoom_area(const float* src, float* dst, const int* out_size, const unsigned char* do_zoom,
                             int B, int D, int H, int W, void* stream) {
    TMF_REQUIRE_PTR(src); TMF_REQUIRE_PTR(dst); TMF_REQUIRE_PTR(out_size); TMF_REQUIRE_PTR(do_zoom);
    TMF_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535 && D <= 65535, TMF_E_SHAPE, "tmf_zoom_area: B=%d D=%d H=%d W=%d", B, D, H, W);
    TMF_REQUIRE(src != dst, TMF_E_ARG, "tmf_zoom_area: not in place");
    hipLaunchKernelGGL(zoom_area_kernel, dim3((unsigned)tmf_cdiv((long)H * W, 256L), D, B), dim3(256), 0, (hipStream_t)stream,
                       src, dst, out_size, do_zoom, D, H, W);
    return tmf_launch_result("tmf_zoom_area");
}

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
