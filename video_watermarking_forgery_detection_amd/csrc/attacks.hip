// Stencil / resampling attack kernels on NCHW f32 planes (gfx950), forward + backward.
//   stencil3  : depthwise 3x3, zero pad 1     noise_layers/gaussian_blur.py:44-56
//   median    : k x k median, zero padding    noise_layers/middle_filter.py:5-13 (kornia MedianBlur)
//   resample  : F.interpolate bilinear / bicubic (align_corners=False, no antialias) of a sub-rectangle
//               noise_layers/resize.py:42-53, noise_layers/crop.py:46-53
//   quant     : round(255 x)/255               models/modules/Quantization.py:7-14
// All are HBM-bound (24 B/px for a 3-channel image: one read + one write); neighbouring taps are
// served by L1/L2, a thread owns one output pixel, consecutive lanes consecutive pixels of a row.
// Backward passes are written in gather form (each input pixel collects from the outputs that used
// it), so they are deterministic: no float atomics.
#include "wm_common.h"

namespace {

struct W9 { float w[9]; };

__global__ __launch_bounds__(256) void stencil3_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                       int W, W9 k) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const float* p = x + (i - (size_t)h * W - w);  // plane base
        float acc = 0.f;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int hh = h + dh;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int ww = w + dw;
                if (ww < 0 || ww >= W) continue;
                acc += k.w[(dh + 1) * 3 + (dw + 1)] * p[(size_t)hh * W + ww];
            }
        }
        y[i] = acc;
    }
}

// ---- median: rank by counting (stable tie-break by tap index), K2 = k*k taps
template <int K>
__global__ __launch_bounds__(256) void median_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int8_t* __restrict__ idx, int N, int H, int W) {
    constexpr int K2 = K * K, R = K / 2;
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const float* p = x + (i - (size_t)h * W - w);
        float v[K2];
#pragma unroll
        for (int t = 0; t < K2; ++t) {
            const int hh = h + t / K - R, ww = w + t % K - R;
            v[t] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? p[(size_t)hh * W + ww] : 0.f;
        }
        float med = 0.f;
        int sel = 0;
#pragma unroll
        for (int a = 0; a < K2; ++a) {
            int rank = 0;
#pragma unroll
            for (int b = 0; b < K2; ++b) rank += (v[b] < v[a] || (v[b] == v[a] && b < a)) ? 1 : 0;
            if (rank == K2 / 2) { med = v[a]; sel = a; }
        }
        y[i] = med;
        if (idx) idx[i] = (int8_t)sel;
    }
}

template <int K>
__global__ __launch_bounds__(256) void median_bwd_kernel(const float* __restrict__ gy, const int8_t* __restrict__ idx,
                                                         float* __restrict__ gx, int N, int H, int W) {
    constexpr int R = K / 2;
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const size_t base = i - (size_t)h * W - w;
        float acc = 0.f;
        // output pixel (oh,ow) selected tap t  <=>  it points at input (oh + t/K - R, ow + t%K - R)
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const int oh = h - (t / K - R), ow = w - (t % K - R);
            if (oh >= 0 && oh < H && ow >= 0 && ow < W) {
                const size_t o = base + (size_t)oh * W + ow;
                if (idx[o] == t) acc += gy[o];
            }
        }
        gx[i] = acc;
    }
}

// ---- resampling (ATen upsample_bilinear2d / upsample_bicubic2d, align_corners=False)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// taps of output index o along one axis: up to 4 (index, weight) pairs; indices already clamped to [0,in)
template <int KIND>
__device__ __forceinline__ int axis_taps(int o, int in, float scale, int (&ix)[4], float (&wt)[4]) {
    if (KIND == WM_BILINEAR) {
        float src = ((float)o + 0.5f) * scale - 0.5f;
        if (src < 0.f) src = 0.f;
        int i0 = (int)src;
        if (i0 > in - 1) i0 = in - 1;
        const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
        const float l1 = src - (float)i0;
        ix[0] = i0; wt[0] = 1.f - l1;
        ix[1] = i1; wt[1] = l1;
        return 2;
    } else {
        const float A = -0.75f;
        const float src = ((float)o + 0.5f) * scale - 0.5f;
        const float fl = floorf(src);
        const int i0 = (int)fl;
        const float t = src - fl;
        wt[0] = cubic2(t + 1.f, A);
        wt[1] = cubic1(t, A);
        wt[2] = cubic1(1.f - t, A);
        wt[3] = cubic2(2.f - t, A);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int j = i0 - 1 + k;
            j = j < 0 ? 0 : (j > in - 1 ? in - 1 : j);
            ix[k] = j;
        }
        return 4;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void resample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                           int W, int h0, int hs, int w0, int ws, int OH, int OW,
                                                           int clamp01) {
    const float sh = (float)hs / (float)OH, sw = (float)ws / (float)OW;
    const size_t total = (size_t)N * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % OW);
        const int oh = (int)((i / OW) % OH);
        const size_t n = i / ((size_t)OW * OH);
        const float* p = x + n * (size_t)H * W;
        int iy[4], ixx[4];
        float wy[4], wx[4];
        const int ny = axis_taps<KIND>(oh, hs, sh, iy, wy);
        const int nx = axis_taps<KIND>(ow, ws, sw, ixx, wx);
        float acc = 0.f;
        for (int a = 0; a < ny; ++a) {
            float row = 0.f;
            for (int b = 0; b < nx; ++b) row += wx[b] * p[(size_t)(h0 + iy[a]) * W + w0 + ixx[b]];
            acc += wy[a] * row;
        }
        if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
        y[i] = acc;
    }
}

// weight with which output o uses input i along one axis (sum over its taps that clamp onto i)
template <int KIND>
__device__ __forceinline__ float axis_weight(int o, int i, int in, float scale) {
    int ix[4];
    float wt[4];
    const int n = axis_taps<KIND>(o, in, scale, ix, wt);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += (ix[k] == i) ? wt[k] : 0.f;
    return s;
}

template <int KIND>
__device__ __forceinline__ void axis_range(int i, int in, int out, float scale, int& lo, int& hi) {
    // outputs whose source coordinate lies within [i-2, i+2] can touch input i (clamped taps included:
    // a tap clamps onto a border pixel only from at most 2 pixels outside)
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 2.f + 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 2.f + 0.5f) * inv - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

template <int KIND>
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ yc,
                                                           float* __restrict__ gx, int N, int H, int W, int h0, int hs,
                                                           int w0, int ws, int OH, int OW) {
    const float sh = (float)hs / (float)OH, sw = (float)ws / (float)OW;
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const size_t n = i / ((size_t)W * H);
        const int ih = h - h0, iw = w - w0;
        float acc = 0.f;
        if (ih >= 0 && ih < hs && iw >= 0 && iw < ws) {
            int ylo, yhi, xlo, xhi;
            axis_range<KIND>(ih, hs, OH, sh, ylo, yhi);
            axis_range<KIND>(iw, ws, OW, sw, xlo, xhi);
            const float* g = gy + n * (size_t)OH * OW;
            const float* yy = yc ? yc + n * (size_t)OH * OW : nullptr;
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = axis_weight<KIND>(oy, ih, hs, sh);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = xlo; ox <= xhi; ++ox) {
                    const float wx = axis_weight<KIND>(ox, iw, ws, sw);
                    if (wx == 0.f) continue;
                    float gg = g[(size_t)oy * OW + ox];
                    if (yy) {  // clamp(0,1) passed the gradient only strictly inside the interval
                        const float v = yy[(size_t)oy * OW + ox];
                        if (!(v > 0.f && v < 1.f)) gg = 0.f;
                    }
                    row += wx * gg;
                }
                acc += wy * row;
            }
        }
        gx[i] = acc;
    }
}

__global__ __launch_bounds__(256) void quant_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = rintf(x[i] * 255.f) / 255.f;
}

inline int grid_for(size_t n) {
    const size_t g = (n + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int wm_stencil3_fwd(const float* x, float* y, int N, int H, int W, const float* w9, void* stream) {
    WM_REQUIRE(x && y && w9 && N > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_stencil3_fwd: bad arguments");
    W9 k;
    for (int i = 0; i < 9; ++i) k.w[i] = w9[i];
    hipLaunchKernelGGL(stencil3_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, k);
    WM_LAUNCH_CHECK("wm_stencil3_fwd");
    return WM_OK;
}

extern "C" int wm_median_fwd(const float* x, float* y, int8_t* idx, int N, int H, int W, int k, void* stream) {
    WM_REQUIRE(x && y && N > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_median_fwd: bad arguments");
    WM_REQUIRE(k == 3 || k == 5, WM_E_SHAPE, "wm_median_fwd: kernel size must be 3 or 5 (got %d)", k);
    const dim3 grid(grid_for((size_t)N * H * W)), block(256);
    if (k == 3) hipLaunchKernelGGL(median_fwd_kernel<3>, grid, block, 0, (hipStream_t)stream, x, y, idx, N, H, W);
    else hipLaunchKernelGGL(median_fwd_kernel<5>, grid, block, 0, (hipStream_t)stream, x, y, idx, N, H, W);
    WM_LAUNCH_CHECK("wm_median_fwd");
    return WM_OK;
}

extern "C" int wm_median_bwd(const float* gy, const int8_t* idx, float* gx, int N, int H, int W, int k, void* stream) {
    WM_REQUIRE(gy && idx && gx && N > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_median_bwd: bad arguments");
    WM_REQUIRE(k == 3 || k == 5, WM_E_SHAPE, "wm_median_bwd: kernel size must be 3 or 5 (got %d)", k);
    const dim3 grid(grid_for((size_t)N * H * W)), block(256);
    if (k == 3) hipLaunchKernelGGL(median_bwd_kernel<3>, grid, block, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
    else hipLaunchKernelGGL(median_bwd_kernel<5>, grid, block, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
    WM_LAUNCH_CHECK("wm_median_bwd");
    return WM_OK;
}

static int resample_check(const char* name, int N, int H, int W, int h0, int hs, int w0, int ws, int OH, int OW, int kind) {
    WM_REQUIRE(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, WM_E_BADARG, "%s: bad shape", name);
    WM_REQUIRE(h0 >= 0 && w0 >= 0 && hs > 0 && ws > 0 && h0 + hs <= H && w0 + ws <= W, WM_E_BADARG,
               "%s: rectangle [%d,%d)x[%d,%d) outside %dx%d", name, h0, h0 + hs, w0, w0 + ws, H, W);
    WM_REQUIRE(kind == WM_BILINEAR || kind == WM_BICUBIC, WM_E_BADARG, "%s: kind must be 0 (bilinear) or 1 (bicubic)", name);
    return WM_OK;
}

extern "C" int wm_resample_fwd(const float* x, float* y, int N, int H, int W, int h0, int hs, int w0, int ws, int OH,
                               int OW, int kind, int clamp01, void* stream) {
    WM_REQUIRE(x && y, WM_E_BADARG, "wm_resample_fwd: null pointer");
    int rc = resample_check("wm_resample_fwd", N, H, W, h0, hs, w0, ws, OH, OW, kind);
    if (rc) return rc;
    const dim3 grid(grid_for((size_t)N * OH * OW)), block(256);
    if (kind == WM_BILINEAR) hipLaunchKernelGGL(resample_fwd_kernel<WM_BILINEAR>, grid, block, 0, (hipStream_t)stream, x, y, N, H, W, h0, hs, w0, ws, OH, OW, clamp01);
    else hipLaunchKernelGGL(resample_fwd_kernel<WM_BICUBIC>, grid, block, 0, (hipStream_t)stream, x, y, N, H, W, h0, hs, w0, ws, OH, OW, clamp01);
    WM_LAUNCH_CHECK("wm_resample_fwd");
    return WM_OK;
}

extern "C" int wm_resample_bwd(const float* gy, const float* y_clamped, float* gx, int N, int H, int W, int h0, int hs,
                               int w0, int ws, int OH, int OW, int kind, void* stream) {
    WM_REQUIRE(gy && gx, WM_E_BADARG, "wm_resample_bwd: null pointer");
    int rc = resample_check("wm_resample_bwd", N, H, W, h0, hs, w0, ws, OH, OW, kind);
    if (rc) return rc;
    const dim3 grid(grid_for((size_t)N * H * W)), block(256);
    if (kind == WM_BILINEAR) hipLaunchKernelGGL(resample_bwd_kernel<WM_BILINEAR>, grid, block, 0, (hipStream_t)stream, gy, y_clamped, gx, N, H, W, h0, hs, w0, ws, OH, OW);
    else hipLaunchKernelGGL(resample_bwd_kernel<WM_BICUBIC>, grid, block, 0, (hipStream_t)stream, gy, y_clamped, gx, N, H, W, h0, hs, w0, ws, OH, OW);
    WM_LAUNCH_CHECK("wm_resample_bwd");
    return WM_OK;
}

extern "C" int wm_quant_fwd(const float* x, float* y, size_t n, void* stream) {
    WM_REQUIRE(x && y && n > 0, WM_E_BADARG, "wm_quant_fwd: bad arguments");
    hipLaunchKernelGGL(quant_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    WM_LAUNCH_CHECK("wm_quant_fwd");
    return WM_OK;
}
