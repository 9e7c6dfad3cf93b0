// Elementwise / small kernels of the general layer family (gconv.hip) for the networks of SURVEY §8f row 1:
//   activations  nn.ReLU, nn.LeakyReLU(0.2), nn.GELU (erf), nn.ELU, nn.Sigmoid, nn.Tanh       conditional_jpeg_generator.py:37-58,292-334; networks.py:667-713
//   residual add / QFAttention  x + gamma * res(x) + beta                                       conditional_jpeg_generator.py:185-200
//   AdaptiveAvgPool2d((1,1)) over NHWC                                                           conditional_jpeg_generator.py:790-812
//   symmetric (symm_pad) / replication padding of an NCHW image into the NHWC conv input, and back    conditional_jpeg_generator.py:865-885,306-309,369
//   torch.nn.utils.spectral_norm: one power iteration + W / sigma, and its backward              networks.py:1381-1385
//   the Bayar constraint on a 5x5 filter                                                         conditional_jpeg_generator.py:814-817
// All tensors NHWC with dtype T (f32 / bf16 / f16); per-sample vectors and parameters f32.
#include "wm_common.h"

namespace {

enum { ACT_RELU = 0, ACT_LRELU = 1, ACT_GELU = 2, ACT_ELU = 3, ACT_SIGMOID = 4, ACT_TANH = 5,
       ACT_ELU_OUT = 6 };   // backward only: the ELU derivative from the layer's OUTPUT (out > 0 ? 1 : out + 1) -- the fused conv + ELU keeps no pre-activation

__device__ __forceinline__ float act_f(int kind, float x) {
    switch (kind) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_LRELU: return x > 0.f ? x : 0.2f * x;
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return tanhf(x);
    }
}
__device__ __forceinline__ float act_d(int kind, float x) {
    switch (kind) {
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_LRELU: return x > 0.f ? 1.f : 0.2f;
        case ACT_GELU: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
        case ACT_ELU: return x > 0.f ? 1.f : expf(x);
        case ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
        case ACT_ELU_OUT: return x > 0.f ? 1.f : x + 1.f;
        default: { const float t = tanhf(x); return 1.f - t * t; }
    }
}

inline int grid1(size_t n, int cap = 4096) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

// element-wise kernels: 16-byte vectors when the length and the pointers allow (NHWC tensors with a stride-16 channel dimension always
// do), the scalar loop for the rest (VEC = false)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void unary_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, int kind) {
    if constexpr (VEC) {
        constexpr int VE = vec16<T>::N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VE; i += (size_t)gridDim.x * blockDim.x) {
            const vec16<T> v = reinterpret_cast<const vec16<T>*>(x)[i];
            vec16<T> o;
#pragma unroll
            for (int e = 0; e < VE; ++e) o.set(e, act_f(kind, v.get(e)));
            reinterpret_cast<vec16<T>*>(y)[i] = o;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = from_f32<T>(act_f(kind, to_f32(x[i])));
    }
}
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void unary_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, size_t n, int kind) {
    if constexpr (VEC) {
        constexpr int VE = vec16<T>::N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VE; i += (size_t)gridDim.x * blockDim.x) {
            const vec16<T> v = reinterpret_cast<const vec16<T>*>(x)[i], g = reinterpret_cast<const vec16<T>*>(gy)[i];
            vec16<T> o;
#pragma unroll
            for (int e = 0; e < VE; ++e) o.set(e, g.get(e) * act_d(kind, v.get(e)));
            reinterpret_cast<vec16<T>*>(gx)[i] = o;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            gx[i] = from_f32<T>(to_f32(gy[i]) * act_d(kind, to_f32(x[i])));
    }
}
inline bool vec_ok(size_t n, int elem, const void* a, const void* b, const void* c = nullptr) {
    return n % (16 / elem) == 0 && !(((size_t)a | (size_t)b | (size_t)c) & 15);
}
// gx = gy * act'(x) AND the split partials of its column sums as stored (the bias gradient of the convolution that fed the activation) in
// one pass -- two launches (this + the reduce) instead of three, and gx is not read back: grid (64-channel blocks, pixel splits), thread
// layout of gconv.hip's gcolsum; part [nsplit][C].  (A one-launch form, the last workgroup to arrive summing the partials, was measured
// and dropped: on this chip an agent-scope release is a write-back of the XCD's L2 -- 256 workgroups fencing after 17 MB of stores each
// made the kernel slower than the three launches it replaced: 160 -> 184 ms on the embedder's step.)
template <typename T>
__global__ __launch_bounds__(256) void unary_bwd_colsum_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, size_t npix, int C, int kind,
                                                               float* __restrict__ part) {
    constexpr int VE = 16 / sizeof(T), MAXPL = 256 / (16 / VE);
    const int cw = min(64, C - (int)blockIdx.x * 64), nv = cw / VE, PL = 256 / nv;
    const int v = threadIdx.x % nv, pl = threadIdx.x / nv;
    const int c0 = blockIdx.x * 64 + v * VE;
    const size_t per = (npix + gridDim.y - 1) / gridDim.y, p0 = (size_t)blockIdx.y * per, p1 = p0 + per < npix ? p0 + per : npix;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    auto one = [&](const vec16<T>& tx, const vec16<T>& tg, size_t p) {
        vec16<T> o;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            o.set(e, tg.get(e) * act_d(kind, tx.get(e)));
            acc[e] += o.get(e);
        }
        *reinterpret_cast<vec16<T>*>(gx + p * C + c0) = o;
    };
    if (pl < PL) {
        size_t p = p0 + pl;
        for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {   // eight loads in flight per thread
            vec16<T> tx[4], tg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                tx[u] = *reinterpret_cast<const vec16<T>*>(x + (p + (size_t)u * PL) * C + c0);
                tg[u] = *reinterpret_cast<const vec16<T>*>(gy + (p + (size_t)u * PL) * C + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(tx[u], tg[u], p + (size_t)u * PL);
        }
        for (; p < p1; p += PL) one(*reinterpret_cast<const vec16<T>*>(x + p * C + c0), *reinterpret_cast<const vec16<T>*>(gy + p * C + c0), p);
    }
    __shared__ float s[MAXPL][65];
    if (pl < PL) {
#pragma unroll
        for (int e = 0; e < VE; ++e) s[pl][v * VE + e] = acc[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < cw) {
        float t = 0.f;
        for (int k = 0; k < PL; ++k) t += s[k][threadIdx.x];
        part[(size_t)blockIdx.y * C + blockIdx.x * 64 + threadIdx.x] = t;
    }
}
// out[c] (+)= the split partials in a fixed order: 64 channels x 16 split lanes per workgroup
__global__ __launch_bounds__(1024) void ubc_reduce_kernel(const float* __restrict__ part, int nsplit, int C, float* __restrict__ out, int Creal, int accumulate) {
    const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float t = 0.f;
    if (c < Creal) {
        int k = sub;
        for (; k + 48 < nsplit; k += 64) {   // four loads in flight (one workgroup reads every split of its 64 channels: latency, not bandwidth)
            const float r0 = part[(size_t)k * C + c], r1 = part[(size_t)(k + 16) * C + c], r2 = part[(size_t)(k + 32) * C + c], r3 = part[(size_t)(k + 48) * C + c];
            t += r0; t += r1; t += r2; t += r3;
        }
        for (; k < nsplit; k += 16) t += part[(size_t)k * C + c];
    }
    __shared__ float sh[16][64];
    sh[sub][cl] = t;
    __syncthreads();
    if (sub == 0 && c < Creal) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += sh[k][cl];
        out[c] = (accumulate ? out[c] : 0.f) + r;
    }
}
inline int ubc_nsplit(size_t npix) {   // >= 128 pixels per split, at most 512 splits (2 workgroups per CU; the reduce reads them all from one)
    const size_t n = (npix + 127) / 128;
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}
// out = a + alpha * b
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n, float alpha) {
    if constexpr (VEC) {
        constexpr int VE = vec16<T>::N;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VE; i += (size_t)gridDim.x * blockDim.x) {
            const vec16<T> va = reinterpret_cast<const vec16<T>*>(a)[i], vb = reinterpret_cast<const vec16<T>*>(b)[i];
            vec16<T> o;
#pragma unroll
            for (int e = 0; e < VE; ++e) o.set(e, va.get(e) + alpha * vb.get(e));
            reinterpret_cast<vec16<T>*>(out)[i] = o;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            out[i] = from_f32<T>(to_f32(a[i]) + alpha * to_f32(b[i]));
    }
}
// QFAttention: out[b,p,c] = x + gamma[b,c] * res + beta[b,c]
template <typename T>
__global__ __launch_bounds__(256) void qfatt_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, size_t hw, int C, int ldv, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t b = i / ((size_t)C * hw);
        out[i] = from_f32<T>(to_f32(x[i]) + gamma[b * ldv + c] * to_f32(res[i]) + beta[b * ldv + c]);
    }
}
// backward: gres = gamma * g (elementwise), ggamma[b,c] = sum_p g * res, gbeta[b,c] = sum_p g; grid = (C/64 blocks, B, pixel splits), 4 pixel
// lanes; part [nsplit][B][2][ldv], summed in a fixed order by qfatt_bwd_reduce_kernel
template <typename T>
__global__ __launch_bounds__(256) void qfatt_bwd_kernel(const T* __restrict__ g, const T* __restrict__ res, const float* __restrict__ gamma,
                                                        T* __restrict__ gres, float* __restrict__ part, size_t hw, int C, int ldv) {
    // a thread owns one 16-byte vector (VE channels) of the block's cw <= 64 channels and every PL-th pixel of the split (gcolsum's layout)
    constexpr int VE = 16 / sizeof(T), MAXPL = 256 / (16 / VE);
    const int cw = min(64, C - (int)blockIdx.x * 64), nv = cw / VE, PL = 256 / nv;
    const int v = threadIdx.x % nv, pl = threadIdx.x / nv, b = blockIdx.y;
    const int c0 = blockIdx.x * 64 + v * VE;
    const size_t per = (hw + gridDim.z - 1) / gridDim.z, p0 = (size_t)blockIdx.z * per, p1 = p0 + per < hw ? p0 + per : hw;
    float s1[VE], s2[VE], gm[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { s1[e] = 0.f; s2[e] = 0.f; gm[e] = pl < PL ? gamma[(size_t)b * ldv + c0 + e] : 0.f; }
    auto one = [&](const vec16<T>& gv, const vec16<T>& rv, size_t p) {
        vec16<T> o;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float gg = gv.get(e);
            o.set(e, gm[e] * gg);
            s1[e] += gg * rv.get(e);
            s2[e] += gg;
        }
        *reinterpret_cast<vec16<T>*>(gres + ((size_t)b * hw + p) * C + c0) = o;
    };
    if (pl < PL) {
        size_t p = p0 + pl;
        for (; p + PL < p1; p += 2 * (size_t)PL) {   // four loads in flight per thread
            const size_t i0 = ((size_t)b * hw + p) * C + c0, i1 = ((size_t)b * hw + p + PL) * C + c0;
            const vec16<T> g0 = *reinterpret_cast<const vec16<T>*>(g + i0), r0 = *reinterpret_cast<const vec16<T>*>(res + i0);
            const vec16<T> g1 = *reinterpret_cast<const vec16<T>*>(g + i1), r1 = *reinterpret_cast<const vec16<T>*>(res + i1);
            one(g0, r0, p); one(g1, r1, p + PL);
        }
        for (; p < p1; p += PL) {
            const size_t i0 = ((size_t)b * hw + p) * C + c0;
            one(*reinterpret_cast<const vec16<T>*>(g + i0), *reinterpret_cast<const vec16<T>*>(res + i0), p);
        }
    }
    __shared__ float sh[MAXPL][65];
    float* o = part + (((size_t)blockIdx.z * gridDim.y + b) * 2) * ldv;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (pl < PL) {
#pragma unroll
            for (int e = 0; e < VE; ++e) sh[pl][v * VE + e] = which ? s2[e] : s1[e];
        }
        __syncthreads();
        if ((int)threadIdx.x < cw) {
            float t = 0.f;
            for (int k = 0; k < PL; ++k) t += sh[k][threadIdx.x];
            o[which * ldv + blockIdx.x * 64 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void qfatt_bwd_reduce_kernel(const float* __restrict__ part, int nsplit, int B, int C, int ldv, float* __restrict__ ggamma,
                                                               float* __restrict__ gbeta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nsplit; ++k) {
        const float* o = part + (((size_t)k * B + b) * 2) * ldv;
        s1 += o[c]; s2 += o[ldv + c];
    }
    ggamma[(size_t)b * ldv + c] = s1; gbeta[(size_t)b * ldv + c] = s2;
}
inline int qfatt_nsplit(size_t hw) {
    const size_t n = (hw + 1023) / 1024;
    return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}
// global average pool x [B,hw,C] -> out f32 [B,C]; backward: gx[b,p,c] = g[b,c] / hw
template <typename T>
__global__ __launch_bounds__(256) void gpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, size_t hw, int C) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, b = blockIdx.y;
    float s = 0.f;
    if (c < C)
        for (size_t p = part; p < hw; p += 4) s += to_f32(x[((size_t)b * hw + p) * C + c]);
    __shared__ float sh[4][64];
    sh[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C) out[(size_t)b * C + c] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x])) / (float)hw;
}
template <typename T>
__global__ __launch_bounds__(256) void gpool_bwd_kernel(const float* __restrict__ g, T* __restrict__ gx, size_t hw, int C, size_t n) {
    const float inv = 1.f / (float)hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t b = i / ((size_t)C * hw);
        gx[i] = from_f32<T>(g[b * C + c] * inv);
    }
}

// NCHW f32 image -> padded NHWC tensor in one gather.  mode 0: symmetric (edge-including reflect, symm_pad of
// conditional_jpeg_generator.py:865-885); mode 1: replicate (nn.ReplicationPad2d, :306-309); pads may be zero (plain layout change).
__device__ __forceinline__ int pad_src(int j, int lo, int n, int mode) {
    int t = j - lo;
    if (mode == 1) return t < 0 ? 0 : (t >= n ? n - 1 : t);
    const int period = 2 * n;
    t %= period; if (t < 0) t += period;
    return t < n ? t : period - 1 - t;
}
struct PadGeo { int B, C, H, W, left, right, top, bottom, mode, CP; };
// x [B,C,H,W] f32 -> out [B,PH,PW,CP] T (channels >= C zero)
template <typename T>
__global__ __launch_bounds__(256) void pad_fwd_kernel(const float* __restrict__ x, T* __restrict__ out, PadGeo g) {
    const int PH = g.H + g.top + g.bottom, PW = g.W + g.left + g.right;
    const size_t n = (size_t)g.B * PH * PW * g.CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.CP);
        const int px = (int)((i / g.CP) % PW), py = (int)((i / ((size_t)g.CP * PW)) % PH);
        const size_t b = i / ((size_t)g.CP * PW * PH);
        float v = 0.f;
        if (c < g.C) v = x[((b * g.C + c) * g.H + pad_src(py, g.top, g.H, g.mode)) * g.W + pad_src(px, g.left, g.W, g.mode)];
        out[i] = from_f32<T>(v);
    }
}
// gx [B,C,H,W] f32 = sum over the padded positions that read each pixel of gp [B,PH,PW,CP]: the direct position plus the border
// rows / columns whose source is this pixel
template <typename T>
__global__ __launch_bounds__(256) void pad_bwd_kernel(const T* __restrict__ gp, float* __restrict__ gx, PadGeo g) {
    const int PH = g.H + g.top + g.bottom, PW = g.W + g.left + g.right;
    const size_t n = (size_t)g.B * g.C * g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % g.W), h = (int)((i / g.W) % g.H), c = (int)((i / ((size_t)g.W * g.H)) % g.C);
        const size_t b = i / ((size_t)g.W * g.H * g.C);
        float acc = 0.f;
        for (int iy = -1; iy < g.top + g.bottom; ++iy) {
            const int py = iy < 0 ? h + g.top : (iy < g.top ? iy : g.H + iy);
            if (iy >= 0 && pad_src(py, g.top, g.H, g.mode) != h) continue;
            for (int ix = -1; ix < g.left + g.right; ++ix) {
                const int px = ix < 0 ? w + g.left : (ix < g.left ? ix : g.W + ix);
                if (ix >= 0 && pad_src(px, g.left, g.W, g.mode) != w) continue;
                acc += to_f32(gp[((b * PH + py) * PW + px) * g.CP + c]);
            }
        }
        gx[i] = acc;
    }
}
// x [B,PH,PW,CP] T -> out [B,C,H,W] f32 (the top-left H x W window, the first C channels) and its adjoint (everything else zero)
template <typename T>
__global__ __launch_bounds__(256) void unpack_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int C, int H, int W, int PH, int PW, int CP) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        out[i] = to_f32(x[((b * PH + h) * PW + w) * CP + c]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void unpack_bwd_kernel(const float* __restrict__ g, T* __restrict__ gx, int B, int C, int H, int W, int PH, int PW, int CP) {
    const size_t n = (size_t)B * PH * PW * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CP);
        const int px = (int)((i / CP) % PW), py = (int)((i / ((size_t)CP * PW)) % PH);
        const size_t b = i / ((size_t)CP * PW * PH);
        gx[i] = from_f32<T>((c < C && py < H && px < W) ? g[((b * C + c) * H + py) * W + px] : 0.f);
    }
}

// ---- spectral norm (torch.nn.utils.spectral_norm, one power iteration): W [M][N] f32 = weight_orig.view(Cout, -1)
//   v = normalize(W^T u), u = normalize(W v), sigma = u . (W v); eps = 1e-12.  Five small launches (the matrix is read twice, by many
//   workgroups): column pass (slices, their sum), its norm, row pass, its norm + sigma.  scratch: wm_spectral_norm_scratch_floats(M, N)
// column pass in two stages (a thread per column walking all M rows left 18 workgroups with 512 dependent loads each: 75 us for a 9 MB
// matrix): row slices of SN_ROWS rows x column blocks -> slice partials [RS][N]; then the slices are summed per column in a fixed order
constexpr int SN_ROWS = 32;
__global__ __launch_bounds__(256) void sn_cols_part_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ partial, int M, int N) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * SN_ROWS, r1 = min(M, r0 + SN_ROWS);
    if (j >= N) return;
    float s = 0.f;
    int i = r0;
    for (; i + 7 < r1; i += 8) {
        float w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = W[(size_t)(i + k) * N + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += w[k] * u[i + k];
    }
    for (; i < r1; ++i) s += W[(size_t)i * N + j] * u[i];
    partial[(size_t)blockIdx.y * N + j] = s;
}
__global__ __launch_bounds__(256) void sn_cols_sum_kernel(const float* __restrict__ partial, int RS, float* __restrict__ vraw, float* __restrict__ part, int N) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (j < N) {
        for (int k = 0; k < RS; ++k) s += partial[(size_t)k * N + j];
        vraw[j] = s;
    }
    float q = wave_sum(s * s);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// v = vraw / max(||vraw||, eps)   (one workgroup)
__global__ __launch_bounds__(256) void sn_vnorm_kernel(const float* __restrict__ vraw, const float* __restrict__ part, int nparts, float* __restrict__ v, int N,
                                                       float eps) {
    __shared__ float sn;
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < nparts; ++k) t += part[k];
        sn = fmaxf(sqrtf(t), eps);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) v[j] = vraw[j] / sn;
}
// wv[i] = W[i,:] . v  (a wave per row)
__global__ __launch_bounds__(256) void sn_rows_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ wv, int M, int N) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    float s = 0.f;
    for (int j = threadIdx.x & 63; j < N; j += 64) s += W[(size_t)i * N + j] * v[j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wv[i] = s;
}
// do_iter: u = wv / max(||wv||, eps); sigma = u . wv   (one workgroup of 1024 threads)
__global__ __launch_bounds__(1024) void sn_sigma_kernel(const float* __restrict__ wv, float* __restrict__ u, float* __restrict__ sigma, int M, int do_iter, float eps) {
    __shared__ float red[1024];
    const int tid = threadIdx.x;
    auto block_sum = [&](float x) {
        red[tid] = x;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    float nu = 0.f;
    for (int i = tid; i < M; i += 1024) nu += wv[i] * wv[i];
    const float un = fmaxf(sqrtf(block_sum(nu)), eps);
    float dot = 0.f;
    for (int i = tid; i < M; i += 1024) {
        const float ui = do_iter ? wv[i] / un : u[i];
        if (do_iter) u[i] = ui;
        dot += ui * wv[i];
    }
    const float sg = block_sum(dot);
    if (tid == 0) sigma[0] = sg;
}
__global__ __launch_bounds__(256) void sn_apply_kernel(const float* __restrict__ W, const float* __restrict__ sigma, float* __restrict__ Wsn, size_t n) {
    const float inv = 1.f / sigma[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) Wsn[i] = W[i] * inv;
}
// gW = (G - <G, Wsn> u v^T) / sigma : partial dot products per workgroup, then the elementwise pass
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ G, const float* __restrict__ Wsn, float* __restrict__ partial, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += G[i] * Wsn[i];
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ G, const float* __restrict__ partial, int nparts, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma, float* __restrict__ gW, int N, size_t n,
                                                     int accumulate) {
    __shared__ float sdot;
    if (threadIdx.x == 0) {
        float d = 0.f;
        for (int k = 0; k < nparts; ++k) d += partial[k];
        sdot = d;
    }
    __syncthreads();
    const float d = sdot, inv = 1.f / sigma[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / N, c = i - r * N;
        const float val = (G[i] - d * u[r] * v[c]) * inv;
        gW[i] = (accumulate ? gW[i] : 0.f) + val;
    }
}

// Bayar constraint on w [Co][Ci][5][5] in place (conditional_jpeg_generator.py:814-817): centre := 0, every 5x5 filter divided by its
// sum, centre := -1
__global__ void bayar_kernel(float* __restrict__ w, int nfilters) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfilters) return;
    float* p = w + (size_t)f * 25;
    p[12] = 0.f;
    float s = 0.f;
    for (int i = 0; i < 25; ++i) s += p[i];
    const float inv = 1.f / s;
    for (int i = 0; i < 25; ++i) p[i] *= inv;
    p[12] += -1.f;
}

}  // namespace

extern "C" int wm_unary_fwd(const void* x, void* y, size_t n, int kind, int dtype, void* stream) {
    WM_REQUIRE(x && y && n > 0 && kind >= 0 && kind <= 5, WM_E_BADARG, "wm_unary_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_unary_fwd", {
        if (vec_ok(n, (int)sizeof(T), x, y)) hipLaunchKernelGGL((unary_fwd_kernel<T, true>), dim3(grid1(n / (16 / sizeof(T)))), dim3(256), 0, s, (const T*)x, (T*)y, n, kind);
        else hipLaunchKernelGGL((unary_fwd_kernel<T, false>), dim3(grid1(n)), dim3(256), 0, s, (const T*)x, (T*)y, n, kind);
    });
    WM_LAUNCH_CHECK("wm_unary_fwd");
    return WM_OK;
}
extern "C" int wm_unary_bwd(const void* x, const void* gy, void* gx, size_t n, int kind, int dtype, void* stream) {
    WM_REQUIRE(x && gy && gx && n > 0 && kind >= 0 && kind <= 6, WM_E_BADARG, "wm_unary_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_unary_bwd", {
        if (vec_ok(n, (int)sizeof(T), x, gy, gx))
            hipLaunchKernelGGL((unary_bwd_kernel<T, true>), dim3(grid1(n / (16 / sizeof(T)))), dim3(256), 0, s, (const T*)x, (const T*)gy, (T*)gx, n, kind);
        else hipLaunchKernelGGL((unary_bwd_kernel<T, false>), dim3(grid1(n)), dim3(256), 0, s, (const T*)x, (const T*)gy, (T*)gx, n, kind);
    });
    WM_LAUNCH_CHECK("wm_unary_bwd");
    return WM_OK;
}
extern "C" size_t wm_unary_bwd_colsum_scratch_floats(size_t npix, int C) { return (size_t)ubc_nsplit(npix) * C; }
// gx = gy * act'(x) over x [npix][C]; out [Creal] f32 (+)= column sums of gx.  part: wm_unary_bwd_colsum_scratch_floats(npix, C) floats.
extern "C" int wm_unary_bwd_colsum(const void* x, const void* gy, void* gx, size_t npix, int C, int kind, float* part, float* out, int Creal, int accumulate,
                                   int dtype, void* stream) {
    WM_REQUIRE(x && gy && gx && part && out && npix > 0 && C > 0 && C % 16 == 0 && Creal > 0 && Creal <= C && kind >= 0 && kind <= 6, WM_E_BADARG,
               "wm_unary_bwd_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int ns = ubc_nsplit(npix);
    const dim3 grid((unsigned)((C + 63) / 64), (unsigned)ns);
    WM_DISPATCH_DTYPE(dtype, "wm_unary_bwd_colsum",
        hipLaunchKernelGGL(unary_bwd_colsum_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (const T*)gy, (T*)gx, npix, C, kind, part));
    hipLaunchKernelGGL(ubc_reduce_kernel, dim3((unsigned)((Creal + 63) / 64)), dim3(1024), 0, s, part, ns, C, out, Creal, accumulate);
    WM_LAUNCH_CHECK("wm_unary_bwd_colsum");
    return WM_OK;
}
extern "C" int wm_add_scaled(const void* a, const void* b, void* out, size_t n, float alpha, int dtype, void* stream) {
    WM_REQUIRE(a && b && out && n > 0, WM_E_BADARG, "wm_add_scaled: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_add_scaled", {
        if (vec_ok(n, (int)sizeof(T), a, b, out))
            hipLaunchKernelGGL((add_kernel<T, true>), dim3(grid1(n / (16 / sizeof(T)))), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)out, n, alpha);
        else hipLaunchKernelGGL((add_kernel<T, false>), dim3(grid1(n)), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)out, n, alpha);
    });
    WM_LAUNCH_CHECK("wm_add_scaled");
    return WM_OK;
}
extern "C" int wm_qfatt_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* out, int B, size_t hw, int C, int ldv, int dtype,
                            void* stream) {
    WM_REQUIRE(x && res && gamma && beta && out && B > 0 && hw > 0 && C > 0 && ldv >= C, WM_E_BADARG, "wm_qfatt_fwd: bad arguments");
    const size_t n = (size_t)B * hw * C;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_qfatt_fwd",
        hipLaunchKernelGGL(qfatt_fwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, (const T*)x, (const T*)res, gamma, beta, (T*)out, hw, C, ldv, n));
    WM_LAUNCH_CHECK("wm_qfatt_fwd");
    return WM_OK;
}
extern "C" size_t wm_qfatt_bwd_scratch_floats(int B, size_t hw, int ldv) { return (size_t)qfatt_nsplit(hw) * B * 2 * ldv; }
extern "C" int wm_qfatt_bwd(const void* g, const void* res, const float* gamma, void* gres, float* ggamma, float* gbeta, float* scratch, int B, size_t hw,
                            int C, int ldv, int dtype, void* stream) {
    WM_REQUIRE(g && res && gamma && gres && ggamma && gbeta && scratch && B > 0 && hw > 0 && C > 0 && ldv >= C, WM_E_BADARG, "wm_qfatt_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int ns = qfatt_nsplit(hw);
    WM_DISPATCH_DTYPE(dtype, "wm_qfatt_bwd",
        hipLaunchKernelGGL(qfatt_bwd_kernel<T>, dim3((unsigned)((C + 63) / 64), (unsigned)B, (unsigned)ns), dim3(256), 0, s, (const T*)g, (const T*)res, gamma,
                           (T*)gres, scratch, hw, C, ldv));
    hipLaunchKernelGGL(qfatt_bwd_reduce_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, scratch, ns, B, C, ldv, ggamma, gbeta);
    WM_LAUNCH_CHECK("wm_qfatt_bwd");
    return WM_OK;
}
extern "C" int wm_gpool_fwd(const void* x, float* out, int B, size_t hw, int C, int dtype, void* stream) {
    WM_REQUIRE(x && out && B > 0 && hw > 0 && C > 0, WM_E_BADARG, "wm_gpool_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_gpool_fwd",
        hipLaunchKernelGGL(gpool_fwd_kernel<T>, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0, s, (const T*)x, out, hw, C));
    WM_LAUNCH_CHECK("wm_gpool_fwd");
    return WM_OK;
}
extern "C" int wm_gpool_bwd(const float* g, void* gx, int B, size_t hw, int C, int dtype, void* stream) {
    WM_REQUIRE(g && gx && B > 0 && hw > 0 && C > 0, WM_E_BADARG, "wm_gpool_bwd: bad arguments");
    const size_t n = (size_t)B * hw * C;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_gpool_bwd", hipLaunchKernelGGL(gpool_bwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, g, (T*)gx, hw, C, n));
    WM_LAUNCH_CHECK("wm_gpool_bwd");
    return WM_OK;
}
static int pad_check(const char* name, const PadGeo& g) {
    WM_REQUIRE(g.B > 0 && g.C > 0 && g.H > 0 && g.W > 0 && g.left >= 0 && g.right >= 0 && g.top >= 0 && g.bottom >= 0 && g.CP >= g.C, WM_E_BADARG,
               "%s: bad arguments", name);
    WM_REQUIRE(g.mode == 0 || g.mode == 1, WM_E_BADARG, "%s: mode %d (0 symmetric, 1 replicate)", name, g.mode);
    return WM_OK;
}
extern "C" int wm_pad_nchw_to_nhwc(const float* x, void* out, int B, int C, int H, int W, int left, int right, int top, int bottom, int mode, int CP,
                                   int dtype, void* stream) {
    WM_REQUIRE(x && out, WM_E_BADARG, "wm_pad_nchw_to_nhwc: null pointer");
    const PadGeo g{B, C, H, W, left, right, top, bottom, mode, CP};
    int rc = pad_check("wm_pad_nchw_to_nhwc", g);
    if (rc) return rc;
    const size_t n = (size_t)B * (H + top + bottom) * (W + left + right) * CP;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_pad_nchw_to_nhwc", hipLaunchKernelGGL(pad_fwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, x, (T*)out, g));
    WM_LAUNCH_CHECK("wm_pad_nchw_to_nhwc");
    return WM_OK;
}
extern "C" int wm_pad_nchw_to_nhwc_bwd(const void* gp, float* gx, int B, int C, int H, int W, int left, int right, int top, int bottom, int mode, int CP,
                                       int dtype, void* stream) {
    WM_REQUIRE(gp && gx, WM_E_BADARG, "wm_pad_nchw_to_nhwc_bwd: null pointer");
    const PadGeo g{B, C, H, W, left, right, top, bottom, mode, CP};
    int rc = pad_check("wm_pad_nchw_to_nhwc_bwd", g);
    if (rc) return rc;
    const size_t n = (size_t)B * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_pad_nchw_to_nhwc_bwd", hipLaunchKernelGGL(pad_bwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, (const T*)gp, gx, g));
    WM_LAUNCH_CHECK("wm_pad_nchw_to_nhwc_bwd");
    return WM_OK;
}
extern "C" int wm_gunpack_nchw(const void* x, float* out, int B, int C, int H, int W, int PH, int PW, int CP, int dtype, void* stream) {
    WM_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0 && PH >= H && PW >= W && CP >= C, WM_E_BADARG, "wm_gunpack_nchw: bad arguments");
    const size_t n = (size_t)B * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_gunpack_nchw", hipLaunchKernelGGL(unpack_fwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, (const T*)x, out, B, C, H, W, PH, PW, CP));
    WM_LAUNCH_CHECK("wm_gunpack_nchw");
    return WM_OK;
}
extern "C" int wm_gunpack_nchw_bwd(const float* g, void* gx, int B, int C, int H, int W, int PH, int PW, int CP, int dtype, void* stream) {
    WM_REQUIRE(g && gx && B > 0 && C > 0 && H > 0 && W > 0 && PH >= H && PW >= W && CP >= C, WM_E_BADARG, "wm_gunpack_nchw_bwd: bad arguments");
    const size_t n = (size_t)B * PH * PW * CP;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_gunpack_nchw_bwd", hipLaunchKernelGGL(unpack_bwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, g, (T*)gx, B, C, H, W, PH, PW, CP));
    WM_LAUNCH_CHECK("wm_gunpack_nchw_bwd");
    return WM_OK;
}

// u [M], v [N] updated in place when do_iter != 0 (training); sigma [1]; then Wsn = W / sigma.  scratch: wm_spectral_norm_scratch_floats(M, N)
extern "C" size_t wm_spectral_norm_scratch_floats(int M, int N) {
    return (size_t)N + M + (size_t)((N + 255) / 256) + 8 + (size_t)((M + SN_ROWS - 1) / SN_ROWS) * N;
}
extern "C" int wm_spectral_norm_fwd(const float* W, float* u, float* v, float* sigma, float* Wsn, float* scratch, int M, int N, int do_iter, void* stream) {
    WM_REQUIRE(W && u && v && sigma && Wsn && scratch && M > 0 && N > 0, WM_E_BADARG, "wm_spectral_norm_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* vraw = scratch;
    float* wv = scratch + N;
    float* part = scratch + N + M;
    const int nb = (N + 255) / 256;
    if (do_iter) {
        const int RS = (M + SN_ROWS - 1) / SN_ROWS;
        float* slices = scratch + N + M + nb + 8;
        hipLaunchKernelGGL(sn_cols_part_kernel, dim3(nb, RS), dim3(256), 0, s, W, u, slices, M, N);
        hipLaunchKernelGGL(sn_cols_sum_kernel, dim3(nb), dim3(256), 0, s, slices, RS, vraw, part, N);
        hipLaunchKernelGGL(sn_vnorm_kernel, dim3(1), dim3(256), 0, s, vraw, part, nb, v, N, 1e-12f);
    }
    hipLaunchKernelGGL(sn_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, W, v, wv, M, N);
    hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(1024), 0, s, wv, u, sigma, M, do_iter, 1e-12f);
    WM_LAUNCH_CHECK("wm_spectral_norm_fwd(power iteration)");
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(sn_apply_kernel, dim3(grid1(n)), dim3(256), 0, s, W, sigma, Wsn, n);
    WM_LAUNCH_CHECK("wm_spectral_norm_fwd");
    return WM_OK;
}
// gW (+)= (G - <G, Wsn> u v^T) / sigma; partial: f32[256] scratch
extern "C" int wm_spectral_norm_bwd(const float* G, const float* Wsn, const float* u, const float* v, const float* sigma, float* partial, float* gW, int M,
                                    int N, int accumulate, void* stream) {
    WM_REQUIRE(G && Wsn && u && v && sigma && partial && gW && M > 0 && N > 0, WM_E_BADARG, "wm_spectral_norm_bwd: bad arguments");
    const size_t n = (size_t)M * N;
    hipStream_t s = (hipStream_t)stream;
    const int nparts = grid1(n, 256);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(nparts), dim3(256), 0, s, G, Wsn, partial, n);
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid1(n)), dim3(256), 0, s, G, partial, nparts, u, v, sigma, gW, N, n, accumulate);
    WM_LAUNCH_CHECK("wm_spectral_norm_bwd");
    return WM_OK;
}
extern "C" int wm_bayar_constrain(float* w, int nfilters, void* stream) {
    WM_REQUIRE(w && nfilters > 0, WM_E_BADARG, "wm_bayar_constrain: bad arguments");
    hipLaunchKernelGGL(bayar_kernel, dim3((nfilters + 63) / 64), dim3(64), 0, (hipStream_t)stream, w, nfilters);
    WM_LAUNCH_CHECK("wm_bayar_constrain");
    return WM_OK;
}
