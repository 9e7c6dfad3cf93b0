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

// grid = (column blocks of 256, rows, planes): no 64-bit div/mod per pixel, a wave reads 3 x 256 B of contiguous row data
__global__ __launch_bounds__(256) void stencil3_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                       int W, W9 k) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        const float* p = x + (size_t)n * H * W;
        for (int h = blockIdx.y; h < H; h += gridDim.y) {
            float acc = 0.f;
#pragma unroll
            for (int dh = -1; dh <= 1; ++dh) {
                const int hh = h + dh;
                if (hh < 0 || hh >= H) continue;
                const float* r = p + (size_t)hh * W;
                const float l = w > 0 ? r[w - 1] : 0.f, c = r[w], rt = w + 1 < W ? r[w + 1] : 0.f;
                acc += k.w[(dh + 1) * 3] * l + k.w[(dh + 1) * 3 + 1] * c + k.w[(dh + 1) * 3 + 2] * rt;
            }
            y[(size_t)n * H * W + (size_t)h * W + w] = acc;
        }
    }
}

// ---- median of K2 = k*k taps (zero padded).  The VALUE comes from a min/max selection network (19 exchanges for 9 taps, 99 for 25:
// Devillard's opt_med9 / opt_med25 orders) instead of K2^2 rank comparisons; the tap INDEX the backward needs keeps the stable
// tie-break of the rank form (among equal values the one with the lowest tap index ranks first): with lt = #{v < med}, the
// selected tap is the (K2/2 - lt)-th tap equal to med.
#define WM_CE(a, b) { const float lo_ = fminf(v[a], v[b]); v[b] = fmaxf(v[a], v[b]); v[a] = lo_; }
template <int K2> __device__ __forceinline__ float median_select(float (&v)[K2]);
template <> __device__ __forceinline__ float median_select<9>(float (&v)[9]) {
    WM_CE(1, 2) WM_CE(4, 5) WM_CE(7, 8) WM_CE(0, 1) WM_CE(3, 4) WM_CE(6, 7) WM_CE(1, 2) WM_CE(4, 5) WM_CE(7, 8) WM_CE(0, 3)
    WM_CE(5, 8) WM_CE(4, 7) WM_CE(3, 6) WM_CE(1, 4) WM_CE(2, 5) WM_CE(4, 7) WM_CE(4, 2) WM_CE(6, 4) WM_CE(4, 2)
    return v[4];
}
template <> __device__ __forceinline__ float median_select<25>(float (&v)[25]) {
    WM_CE(0, 1) WM_CE(3, 4) WM_CE(2, 4) WM_CE(2, 3) WM_CE(6, 7) WM_CE(5, 7) WM_CE(5, 6) WM_CE(9, 10) WM_CE(8, 10) WM_CE(8, 9)
    WM_CE(12, 13) WM_CE(11, 13) WM_CE(11, 12) WM_CE(15, 16) WM_CE(14, 16) WM_CE(14, 15) WM_CE(18, 19) WM_CE(17, 19) WM_CE(17, 18)
    WM_CE(21, 22) WM_CE(20, 22) WM_CE(20, 21) WM_CE(23, 24) WM_CE(2, 5) WM_CE(3, 6) WM_CE(0, 6) WM_CE(0, 3) WM_CE(4, 7) WM_CE(1, 7)
    WM_CE(1, 4) WM_CE(11, 14) WM_CE(8, 14) WM_CE(8, 11) WM_CE(12, 15) WM_CE(9, 15) WM_CE(9, 12) WM_CE(13, 16) WM_CE(10, 16)
    WM_CE(10, 13) WM_CE(20, 23) WM_CE(17, 23) WM_CE(17, 20) WM_CE(21, 24) WM_CE(18, 24) WM_CE(18, 21) WM_CE(19, 22) WM_CE(8, 17)
    WM_CE(9, 18) WM_CE(0, 18) WM_CE(0, 9) WM_CE(10, 19) WM_CE(1, 19) WM_CE(1, 10) WM_CE(11, 20) WM_CE(2, 20) WM_CE(2, 11)
    WM_CE(12, 21) WM_CE(3, 21) WM_CE(3, 12) WM_CE(13, 22) WM_CE(4, 22) WM_CE(4, 13) WM_CE(14, 23) WM_CE(5, 23) WM_CE(5, 14)
    WM_CE(15, 24) WM_CE(6, 24) WM_CE(6, 15) WM_CE(7, 16) WM_CE(7, 19) WM_CE(13, 21) WM_CE(15, 23) WM_CE(7, 13) WM_CE(7, 15)
    WM_CE(1, 9) WM_CE(3, 11) WM_CE(5, 17) WM_CE(11, 17) WM_CE(9, 17) WM_CE(4, 10) WM_CE(6, 12) WM_CE(7, 14) WM_CE(4, 6) WM_CE(4, 7)
    WM_CE(12, 14) WM_CE(10, 14) WM_CE(6, 7) WM_CE(10, 12) WM_CE(6, 10) WM_CE(6, 17) WM_CE(12, 17) WM_CE(7, 17) WM_CE(7, 10)
    WM_CE(12, 18) WM_CE(7, 12) WM_CE(10, 18) WM_CE(12, 20) WM_CE(10, 20) WM_CE(10, 12)
    return v[12];
}
#undef WM_CE

template <int K>
__global__ __launch_bounds__(256) void median_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int8_t* __restrict__ idx, int N, int H, int W) {
    constexpr int K2 = K * K, R = K / 2;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        const float* p = x + (size_t)n * H * W;
        for (int h = blockIdx.y; h < H; h += gridDim.y) {
            float v[K2], s[K2];
#pragma unroll
            for (int t = 0; t < K2; ++t) {
                const int hh = h + t / K - R, ww = w + t % K - R;
                v[t] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? p[(size_t)hh * W + ww] : 0.f;
                s[t] = v[t];
            }
            const float med = median_select<K2>(s);
            const size_t o = (size_t)n * H * W + (size_t)h * W + w;
            y[o] = med;
            if (idx) {
                int lt = 0;
#pragma unroll
                for (int t = 0; t < K2; ++t) lt += v[t] < med ? 1 : 0;
                int want = K2 / 2 - lt, sel = 0, seen = 0;
#pragma unroll
                for (int t = 0; t < K2; ++t) {
                    const bool eq = v[t] == med;
                    if (eq && seen == want) sel = t;
                    seen += eq ? 1 : 0;
                }
                idx[o] = (int8_t)sel;
            }
        }
    }
}

// The same result, four adjacent output pixels per thread (W % 4 == 0, 16-byte aligned planes): a window row is one float4 and the R columns
// either side of it (bounds decided per row, not per tap: the form above spends 25 exec-masked loads on one pixel), the four windows
// are views of those K x (4 + 2R) registers, and the tap index takes the short way when exactly one tap equals the median (then no
// value below it is missing from the count: lt = K2/2, the wanted tap is that one) -- ties (zero padding, saturated images) fall back
// to the rank form above, so the index plane is the same byte for byte.
template <int K>
__global__ __launch_bounds__(256) void median_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, int8_t* __restrict__ idx,
                                                          int N, int H, int W) {
    constexpr int K2 = K * K, R = K / 2, RW = 4 + 2 * R;
    const int W4 = W >> 2;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= H * W4) return;
    const int h = item / W4, w4 = (item - h * W4) * 4;
    const bool lo = w4 > 0, hi = w4 + 4 < W;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float* p = x + (size_t)n * H * W;
        float r[K][RW];
#pragma unroll
        for (int dh = 0; dh < K; ++dh) {
            const int hh = h + dh - R;
#pragma unroll
            for (int q = 0; q < RW; ++q) r[dh][q] = 0.f;
            if (hh >= 0 && hh < H) {
                const float* q = p + (size_t)hh * W + w4;
                const float4 c = *reinterpret_cast<const float4*>(q);
                r[dh][R] = c.x; r[dh][R + 1] = c.y; r[dh][R + 2] = c.z; r[dh][R + 3] = c.w;
                if constexpr (R == 2) {
                    if (lo) { const float2 l = *reinterpret_cast<const float2*>(q - 2); r[dh][0] = l.x; r[dh][1] = l.y; }
                    if (hi) { const float2 g = *reinterpret_cast<const float2*>(q + 4); r[dh][6] = g.x; r[dh][7] = g.y; }
                } else {
                    if (lo) r[dh][0] = q[-1];
                    if (hi) r[dh][5] = q[4];
                }
            }
        }
        float med[4];
        unsigned selw = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s[K2];
#pragma unroll
            for (int t = 0; t < K2; ++t) s[t] = r[t / K][j + t % K];
            med[j] = median_select<K2>(s);
            if (idx) {
                int cnt = 0, sel = 0;
#pragma unroll
                for (int t = K2 - 1; t >= 0; --t) {
                    const bool eq = r[t / K][j + t % K] == med[j];
                    sel = eq ? t : sel;
                    cnt += eq ? 1 : 0;
                }
                if (cnt != 1) {
                    int lt = 0;
#pragma unroll
                    for (int t = 0; t < K2; ++t) lt += r[t / K][j + t % K] < med[j] ? 1 : 0;
                    const int want = K2 / 2 - lt;
                    int seen = 0;
                    sel = 0;
#pragma unroll
                    for (int t = 0; t < K2; ++t) {
                        const bool eq = r[t / K][j + t % K] == med[j];
                        if (eq && seen == want) sel = t;
                        seen += eq ? 1 : 0;
                    }
                }
                selw |= (unsigned)sel << (8 * j);
            }
        }
        const size_t o = (size_t)n * H * W + (size_t)h * W + w4;
        *reinterpret_cast<float4*>(y + o) = float4{med[0], med[1], med[2], med[3]};
        if (idx) *reinterpret_cast<unsigned*>(idx + o) = selw;
    }
}

template <int K>
__global__ __launch_bounds__(256) void median_bwd_kernel(const float* __restrict__ gy, const int8_t* __restrict__ idx,
                                                         float* __restrict__ gx, int N, int H, int W) {
    constexpr int R = K / 2;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        const size_t base = (size_t)n * H * W;
        for (int h = blockIdx.y; h < H; h += gridDim.y) {
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
            gx[base + (size_t)h * W + w] = acc;
        }
    }
}

// The same sums, four adjacent input pixels per thread (W % 4 == 0, 16-byte aligned planes): per window row three 4-byte words of the index
// plane and three float4 of gy cover the 4 + 2 R columns every one of the four pixels looks at -- 30 loads for 4 pixels where the form above
// issues 4 x 2 K^2 predicated ones (90 -> 25 us for K = 5 at B=16, 3 x 256 x 256).  Taps are visited in the same order: bit-identical results.
template <int K>
__global__ __launch_bounds__(256) void median_bwd4_kernel(const float* __restrict__ gy, const int8_t* __restrict__ idx,
                                                          float* __restrict__ gx, int N, int H, int W) {
    constexpr int R = K / 2;
    const int W4 = W >> 2;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= H * W4) return;
    const int h = item / W4, w4 = (item - h * W4) * 4;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t base = (size_t)n * H * W;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dh = -R; dh <= R; ++dh) {
            const int oh = h - dh;            // the output row whose window row dh + R lies on input row h
            if (oh < 0 || oh >= H) continue;
            const size_t o = base + (size_t)oh * W + w4;
            const bool lo = w4 >= 4, hi = w4 + 4 < W;
            const unsigned iw[3] = {lo ? *reinterpret_cast<const unsigned*>(idx + o - 4) : 0x7f7f7f7fu, *reinterpret_cast<const unsigned*>(idx + o),
                                    hi ? *reinterpret_cast<const unsigned*>(idx + o + 4) : 0x7f7f7f7fu};
            const float4 z = {0.f, 0.f, 0.f, 0.f};
            const float4 gw[3] = {lo ? *reinterpret_cast<const float4*>(gy + o - 4) : z, *reinterpret_cast<const float4*>(gy + o),
                                  hi ? *reinterpret_cast<const float4*>(gy + o + 4) : z};
            const float* g = reinterpret_cast<const float*>(gw);     // g[q], b(q): column w4 - 4 + q
            auto b = [&](int q) { return (int)((iw[q >> 2] >> (8 * (q & 3))) & 0xffu); };
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int dw = -R; dw <= R; ++dw) {
                    const int q = 4 + j - dw, t = (dh + R) * K + (dw + R);
                    if (b(q) == t) acc[j] += g[q];
                }
        }
        *reinterpret_cast<float4*>(gx + base + (size_t)h * W + w4) = float4{acc[0], acc[1], acc[2], acc[3]};
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

// gather-form backward (deterministic): input pixel (ih, iw) collects from the outputs that used it.  The interpolation is
// separable, so the weights along x of the candidate output columns are evaluated ONCE per input pixel (not once per candidate
// output row as before: 10x fewer cubic evaluations at the 0.7x resize).  2-D launch: no 64-bit div/mod per pixel.
template <int KIND>
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ yc,
                                                           float* __restrict__ gx, int N, int H, int W, int h0, int hs,
                                                           int w0, int ws, int OH, int OW) {
    constexpr int MAXC = 24;   // candidate output columns kept in registers; wider footprints (scale < 0.2) evaluate on the fly
    const float sh = (float)hs / (float)OH, sw = (float)ws / (float)OW;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    const int iw = w - w0;
    int xlo = 0, xhi = -1;
    float wxs[MAXC];
    const bool in_x = iw >= 0 && iw < ws;
    if (in_x) {
        axis_range<KIND>(iw, ws, OW, sw, xlo, xhi);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) wxs[c] = (xlo + c <= xhi) ? axis_weight<KIND>(xlo + c, iw, ws, sw) : 0.f;
    }
    const bool wide = xhi - xlo + 1 > MAXC;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        const float* g = gy + (size_t)n * OH * OW;
        const float* yy = yc ? yc + (size_t)n * OH * OW : nullptr;
        for (int h = blockIdx.y; h < H; h += gridDim.y) {
            const int ih = h - h0;
            float acc = 0.f;
            if (in_x && ih >= 0 && ih < hs) {
                int ylo, yhi;
                axis_range<KIND>(ih, hs, OH, sh, ylo, yhi);
                for (int oy = ylo; oy <= yhi; ++oy) {
                    const float wy = axis_weight<KIND>(oy, ih, hs, sh);
                    if (wy == 0.f) continue;
                    float row = 0.f;
                    const float* gr = g + (size_t)oy * OW;
                    const float* yr = yy ? yy + (size_t)oy * OW : nullptr;
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int ox = xlo + c;
                        if (ox > xhi || wxs[c] == 0.f) continue;
                        float gg = gr[ox];
                        if (yr) {  // clamp(0,1) passed the gradient only strictly inside the interval
                            const float v = yr[ox];
                            if (!(v > 0.f && v < 1.f)) gg = 0.f;
                        }
                        row += wxs[c] * gg;
                    }
                    if (wide)
                        for (int ox = xlo + MAXC; ox <= xhi; ++ox) {
                            const float wx = axis_weight<KIND>(ox, iw, ws, sw);
                            if (wx == 0.f) continue;
                            float gg = gr[ox];
                            if (yr) {
                                const float v = yr[ox];
                                if (!(v > 0.f && v < 1.f)) gg = 0.f;
                            }
                            row += wx * gg;
                        }
                    acc += wy * row;
                }
            }
            gx[(size_t)n * H * W + (size_t)h * W + w] = acc;
        }
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
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)(H < 65535 ? H : 65535), (unsigned)(N < 65535 ? N : 65535));
    hipLaunchKernelGGL(stencil3_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, k);
    WM_LAUNCH_CHECK("wm_stencil3_fwd");
    return WM_OK;
}

extern "C" int wm_median_fwd(const float* x, float* y, int8_t* idx, int N, int H, int W, int k, void* stream) {
    WM_REQUIRE(x && y && N > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_median_fwd: bad arguments");
    WM_REQUIRE(k == 3 || k == 5, WM_E_SHAPE, "wm_median_fwd: kernel size must be 3 or 5 (got %d)", k);
    if (W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (((uintptr_t)idx) & 3) == 0 && (long long)H * (W / 4) < (1LL << 31)) {
        const dim3 grid4((unsigned)(((long long)H * (W / 4) + 255) / 256), (unsigned)(N < 65535 ? N : 65535)), block4(256);
        if (k == 3) hipLaunchKernelGGL(median_fwd4_kernel<3>, grid4, block4, 0, (hipStream_t)stream, x, y, idx, N, H, W);
        else hipLaunchKernelGGL(median_fwd4_kernel<5>, grid4, block4, 0, (hipStream_t)stream, x, y, idx, N, H, W);
        WM_LAUNCH_CHECK("wm_median_fwd");
        return WM_OK;
    }
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)(H < 65535 ? H : 65535), (unsigned)(N < 65535 ? N : 65535)), block(256);
    if (k == 3) hipLaunchKernelGGL(median_fwd_kernel<3>, grid, block, 0, (hipStream_t)stream, x, y, idx, N, H, W);
    else hipLaunchKernelGGL(median_fwd_kernel<5>, grid, block, 0, (hipStream_t)stream, x, y, idx, N, H, W);
    WM_LAUNCH_CHECK("wm_median_fwd");
    return WM_OK;
}

extern "C" int wm_median_bwd(const float* gy, const int8_t* idx, float* gx, int N, int H, int W, int k, void* stream) {
    WM_REQUIRE(gy && idx && gx && N > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_median_bwd: bad arguments");
    WM_REQUIRE(k == 3 || k == 5, WM_E_SHAPE, "wm_median_bwd: kernel size must be 3 or 5 (got %d)", k);
    if (W % 4 == 0 && (((uintptr_t)gy | (uintptr_t)gx) & 15) == 0 && (((uintptr_t)idx) & 3) == 0 && (long long)H * (W / 4) < (1LL << 31)) {
        const dim3 grid4((unsigned)(((long long)H * (W / 4) + 255) / 256), (unsigned)(N < 65535 ? N : 65535)), block4(256);
        if (k == 3) hipLaunchKernelGGL(median_bwd4_kernel<3>, grid4, block4, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
        else hipLaunchKernelGGL(median_bwd4_kernel<5>, grid4, block4, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
        WM_LAUNCH_CHECK("wm_median_bwd");
        return WM_OK;
    }
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)(H < 65535 ? H : 65535), (unsigned)(N < 65535 ? N : 65535)), block(256);
    if (k == 3) hipLaunchKernelGGL(median_bwd_kernel<3>, grid, block, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
    else hipLaunchKernelGGL(median_bwd_kernel<5>, grid, block, 0, (hipStream_t)stream, gy, idx, gx, N, H, W);
    WM_LAUNCH_CHECK("wm_median_bwd");
    return WM_OK;
}

namespace {

// ---- the same backward in two separable passes (x then y) through a workspace tmp[N][OH][W]: each pass gathers the ~4 / scale + 3 taps of
// ONE axis, where the gather form above walks their product for every input pixel (36-64 taps at the Resize attack's 0.7x / 1.43x, behind a
// 24-wide predicated column loop per candidate row): 83 -> 25 us for bicubic at B=16 3x256x256.  Same weights (axis_weight), a different
// summation order than the gather form (agreement ~1e-7 relative; both are compared with autograd of F.interpolate in tests/).
template <int KIND, int MAXC>   // MAXC: candidate output columns kept in registers (the host picks 8 / 12 / 24 from the scale; wider footprints evaluate on the fly)
__global__ __launch_bounds__(256) void resample_bwd_x_kernel(const float* __restrict__ gy, const float* __restrict__ yc, float* __restrict__ tmp,
                                                             int N, int W, int w0, int ws, int OH, int OW) {
    const float sw = (float)ws / (float)OW;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    const int iw = w - w0;
    int xlo = 0, xhi = -1;
    float wxs[MAXC];
    const bool in_x = iw >= 0 && iw < ws;
    if (in_x) {
        axis_range<KIND>(iw, ws, OW, sw, xlo, xhi);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) wxs[c] = (xlo + c <= xhi) ? axis_weight<KIND>(xlo + c, iw, ws, sw) : 0.f;
    }
    const bool wide = xhi - xlo + 1 > MAXC;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        for (int oy = blockIdx.y; oy < OH; oy += gridDim.y) {
            const float* gr = gy + ((size_t)n * OH + oy) * OW;
            const float* yr = yc ? yc + ((size_t)n * OH + oy) * OW : nullptr;
            float row = 0.f;
            if (in_x) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int ox = xlo + c;
                    if (ox > xhi || wxs[c] == 0.f) continue;
                    float gg = gr[ox];
                    if (yr) {  // clamp(0,1) passed the gradient only strictly inside the interval
                        const float v = yr[ox];
                        if (!(v > 0.f && v < 1.f)) gg = 0.f;
                    }
                    row += wxs[c] * gg;
                }
                if (wide)
                    for (int ox = xlo + MAXC; ox <= xhi; ++ox) {
                        const float wx = axis_weight<KIND>(ox, iw, ws, sw);
                        if (wx == 0.f) continue;
                        float gg = gr[ox];
                        if (yr) {
                            const float v = yr[ox];
                            if (!(v > 0.f && v < 1.f)) gg = 0.f;
                        }
                        row += wx * gg;
                    }
            }
            tmp[((size_t)n * OH + oy) * W + w] = row;
        }
    }
}

// The x pass with the output row staged in the LDS (round 4): the block loads a row of gy once, coalesced -- the clamp's mask applied while it
// does -- and every thread then reads its MAXC candidates from there: 1-2 global loads per thread and row instead of up to 2 MAXC predicated
// ones behind a branch each (31.7 -> us for bicubic 0.7x at B=16, 3 x 256 x 256).  Same weights, same order of the sum: bit-identical.
constexpr int RS_MAXOW = 1024;
template <int KIND, int MAXC>
__global__ __launch_bounds__(256) void resample_bwd_x_lds_kernel(const float* __restrict__ gy, const float* __restrict__ yc, float* __restrict__ tmp,
                                                                 int N, int W, int w0, int ws, int OH, int OW) {
    __shared__ float srow[2][RS_MAXOW];
    const float sw = (float)ws / (float)OW;
    const int w = blockIdx.x * 256 + threadIdx.x;
    const bool active = w < W;
    const int iw = w - w0;
    int xlo = 0, xhi = -1;
    float wxs[MAXC];
    const bool in_x = active && iw >= 0 && iw < ws;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) wxs[c] = 0.f;
    if (in_x) {
        axis_range<KIND>(iw, ws, OW, sw, xlo, xhi);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) wxs[c] = (xlo + c <= xhi) ? axis_weight<KIND>(xlo + c, iw, ws, sw) : 0.f;
    }
    const bool wide = xhi - xlo + 1 > MAXC;
    int it = 0;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        for (int oy = blockIdx.y; oy < OH; oy += gridDim.y, ++it) {
            const float* gr = gy + ((size_t)n * OH + oy) * OW;
            const float* yr = yc ? yc + ((size_t)n * OH + oy) * OW : nullptr;
            float* sr = srow[it & 1];     // two buffers: the barrier of row k + 1 separates the reads of row k from the writes of row k + 2
            for (int i = threadIdx.x; i < OW; i += 256) {
                float gg = gr[i];
                if (yr) {  // clamp(0,1) passed the gradient only strictly inside the interval
                    const float v = yr[i];
                    if (!(v > 0.f && v < 1.f)) gg = 0.f;
                }
                sr[i] = gg;
            }
            __syncthreads();
            if (!active) continue;
            float row = 0.f;
            if (in_x) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int ox = min(xlo + c, OW - 1);
                    const float gg = wxs[c] != 0.f ? sr[ox] : 0.f;   // (a term the gather form skips contributes exactly nothing here either)
                    row += wxs[c] * gg;
                }
                if (wide)
                    for (int ox = xlo + MAXC; ox <= xhi; ++ox) {
                        const float wx = axis_weight<KIND>(ox, iw, ws, sw);
                        if (wx == 0.f) continue;
                        row += wx * sr[ox];
                    }
            }
            tmp[((size_t)n * OH + oy) * W + w] = row;
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void resample_bwd_y_kernel(const float* __restrict__ tmp, float* __restrict__ gx, int N, int H, int W, int h0,
                                                             int hs, int OH) {
    const float sh = (float)hs / (float)OH;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= W) return;
    // rows outside, planes inside (round 4): a row's tap range and weights are evaluated ONCE and serve every plane the block walks -- with
    // one (row, plane) per block (12-18 thousand blocks of one row each) the weight polynomials were most of the kernel
    constexpr int MAXT = 12;
    for (int h = blockIdx.y; h < H; h += gridDim.y) {
        const int ih = h - h0;
        int ylo = 0, yhi = -1;
        float wys[MAXT];
        const bool in_y = ih >= 0 && ih < hs;
        if (in_y) {
            axis_range<KIND>(ih, hs, OH, sh, ylo, yhi);           // (wave-uniform bounds and weights: every lane of the row has the same ih)
#pragma unroll
            for (int c = 0; c < MAXT; ++c) wys[c] = (ylo + c <= yhi) ? axis_weight<KIND>(ylo + c, ih, hs, sh) : 0.f;
        }
        for (int n = blockIdx.z; n < N; n += gridDim.z) {
            const float* t = tmp + (size_t)n * OH * W + w;
            float acc = 0.f;
            if (in_y) {
#pragma unroll
                for (int c = 0; c < MAXT; ++c)
                    if (ylo + c <= yhi && wys[c] != 0.f) acc += wys[c] * t[(size_t)(ylo + c) * W];
                for (int oy = ylo + MAXT; oy <= yhi; ++oy) {     // (footprints wider than MAXT taps: evaluated on the fly, as before)
                    const float wy = axis_weight<KIND>(oy, ih, hs, sh);
                    if (wy != 0.f) acc += wy * t[(size_t)oy * W];
                }
            }
            gx[((size_t)n * H + h) * W + w] = acc;
        }
    }
}

}  // namespace

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
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)(H < 65535 ? H : 65535), (unsigned)(N < 65535 ? N : 65535)), block(256);
    if (kind == WM_BILINEAR) hipLaunchKernelGGL(resample_bwd_kernel<WM_BILINEAR>, grid, block, 0, (hipStream_t)stream, gy, y_clamped, gx, N, H, W, h0, hs, w0, ws, OH, OW);
    else hipLaunchKernelGGL(resample_bwd_kernel<WM_BICUBIC>, grid, block, 0, (hipStream_t)stream, gy, y_clamped, gx, N, H, W, h0, hs, w0, ws, OH, OW);
    WM_LAUNCH_CHECK("wm_resample_bwd");
    return WM_OK;
}

extern "C" int wm_resample_bwd_sep(const float* gy, const float* y_clamped, float* gx, float* tmp, int N, int H, int W, int h0, int hs,
                                   int w0, int ws, int OH, int OW, int kind, void* stream) {
    WM_REQUIRE(gy && gx && tmp, WM_E_BADARG, "wm_resample_bwd_sep: null pointer");
    int rc = resample_check("wm_resample_bwd_sep", N, H, W, h0, hs, w0, ws, OH, OW, kind);
    if (rc) return rc;
    // x pass: a thread keeps its column's weights in registers and walks rows: ~1024 blocks in all instead of one per (row, plane), so that
    // the weights are evaluated once per ~OH / 16 rows.  y pass: rows outside, planes inside (its kernel): a block per row and plane GROUP
    const unsigned gz = (unsigned)(N < 65535 ? N : 65535);
    const unsigned bx = (unsigned)((W + 255) / 256);
    unsigned ry = (unsigned)(1024u / (bx * gz > 0 ? bx * gz : 1u));
    if (ry < 1) ry = 1;
    if (ry > (unsigned)OH) ry = (unsigned)OH;
    unsigned pz = (unsigned)(1024u / (bx * (unsigned)(H < 65535 ? H : 65535)));
    if (pz < 1) pz = 1;
    if (pz > gz) pz = gz;
    const dim3 gx_grid(bx, ry, gz), gy_grid(bx, (unsigned)(H < 65535 ? H : 65535), pz), block(256);
    hipStream_t s = (hipStream_t)stream;
    // candidate columns of one input pixel: axis_range's [lo, hi] spans at most 4 / scale + 5 outputs (scale = ws / OW)
    const int span = (int)(4.0 * (double)OW / (double)ws) + 5;
#define WM_RS_X(KIND_, MAXC_)                                                                                                                   \
    do {                                                                                                                                       \
        if (OW <= RS_MAXOW) hipLaunchKernelGGL((resample_bwd_x_lds_kernel<KIND_, MAXC_>), gx_grid, block, 0, s, gy, y_clamped, tmp, N, W, w0, ws, OH, OW); \
        else hipLaunchKernelGGL((resample_bwd_x_kernel<KIND_, MAXC_>), gx_grid, block, 0, s, gy, y_clamped, tmp, N, W, w0, ws, OH, OW);           \
    } while (0)
    if (kind == WM_BILINEAR) {
        if (span <= 8) WM_RS_X(WM_BILINEAR, 8); else if (span <= 12) WM_RS_X(WM_BILINEAR, 12); else WM_RS_X(WM_BILINEAR, 24);
        hipLaunchKernelGGL(resample_bwd_y_kernel<WM_BILINEAR>, gy_grid, block, 0, s, (const float*)tmp, gx, N, H, W, h0, hs, OH);
    } else {
        if (span <= 8) WM_RS_X(WM_BICUBIC, 8); else if (span <= 12) WM_RS_X(WM_BICUBIC, 12); else WM_RS_X(WM_BICUBIC, 24);
        hipLaunchKernelGGL(resample_bwd_y_kernel<WM_BICUBIC>, gy_grid, block, 0, s, (const float*)tmp, gx, N, H, W, h0, hs, OH);
    }
#undef WM_RS_X
    WM_LAUNCH_CHECK("wm_resample_bwd_sep");
    return WM_OK;
}

extern "C" int wm_quant_fwd(const float* x, float* y, size_t n, void* stream) {
    WM_REQUIRE(x && y && n > 0, WM_E_BADARG, "wm_quant_fwd: bad arguments");
    hipLaunchKernelGGL(quant_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    WM_LAUNCH_CHECK("wm_quant_fwd");
    return WM_OK;
}
