// Invertible-embedder pieces (models/invertible_net.py of the reference) that are not convolutions:
//   HaarDownsampling / HaarUpsampling  :178-292  fixed 2x2 stride-2 depthwise wavelet analysis / synthesis (HBM-bound: 1 read + 1 write)
//   RNVPCouplingBlock's affine         :122-175  y = e(s) * x + t  /  y = (x - t) / e(s),  e(s) = exp(clamp * (2 sigmoid(s) - 1)) + 1e-4
//   channel narrow / cat               :150-151,175 and the subnets' torch.cat (:318-322,363) as strided channel copies
// NHWC tensors of dtype T (f32 / bf16 / f16), channel stride a multiple of 16, padding channels written as zero.
#include "wm_common.h"

namespace {

inline int grid1(size_t n, int cap = 8192) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

// sign of haar_weights[k][0][dy][dx] (:187-197): k=0 all +; k=1 dx; k=2 dy; k=3 dy^dx
__device__ __forceinline__ float haar_sign(int k, int dy, int dx) {
    const int neg = k == 0 ? 0 : (k == 1 ? dx : (k == 2 ? dy : (dy ^ dx)));
    return neg ? -1.f : 1.f;
}

// analysis: in [B,2H,2W,CPin] (C real channels) -> out [B,H,W,CPout], out channel 4c+k = fac * sum_{dy,dx} sign_k * in(2y+dy, 2x+dx, c)
template <typename T>
__global__ __launch_bounds__(256) void haar_down_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int CPin, int CPout,
                                                        float fac, int bywav) {
    const int CQ = CPout / 4;
    const size_t n = (size_t)B * H * W * CQ;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CQ);
        const int x = (int)((i / CQ) % W), y = (int)((i / ((size_t)CQ * W)) % H);
        const size_t b = i / ((size_t)CQ * W * H);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            float v[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) v[dy][dx] = to_f32(in[((b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * CPin + c]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = fac * ((haar_sign(k, 0, 0) * v[0][0] + haar_sign(k, 0, 1) * v[0][1]) + (haar_sign(k, 1, 0) * v[1][0] + haar_sign(k, 1, 1) * v[1][1]));
        }
        // order_by_wavelet (invertible_net.py:207-218): wavelet k of channel c lands at k C + c instead of 4 c + k (padding groups c >= C
        // keep their 4 c + k slots, which lie beyond the 4 C real channels either way)
        T* op = out + ((b * H + y) * W + x) * CPout;
#pragma unroll
        for (int k = 0; k < 4; ++k) op[(bywav && c < C) ? k * C + c : 4 * c + k] = from_f32<T>(o[k]);
    }
}
// synthesis: in [B,H,W,CPin] (4C real channels) -> out [B,2H,2W,CPout], out(2y+dy, 2x+dx, c) = fac * sum_k sign_k(dy,dx) * in(y, x, 4c+k)
// (bywav: in(y, x, k C + c))
template <typename T>
__global__ __launch_bounds__(256) void haar_up_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int CPin, int CPout,
                                                      float fac, int bywav) {
    const size_t n = (size_t)B * H * W * CPout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CPout);
        const int x = (int)((i / CPout) % W), y = (int)((i / ((size_t)CPout * W)) % H);
        const size_t b = i / ((size_t)CPout * W * H);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const T* ip = in + ((b * H + y) * W + x) * CPin;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = to_f32(ip[bywav ? k * C + c : 4 * c + k]);
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float o = fac * ((haar_sign(0, dy, dx) * v[0] + haar_sign(1, dy, dx) * v[1]) + (haar_sign(2, dy, dx) * v[2] + haar_sign(3, dy, dx) * v[3]));
                out[((b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * CPout + c] = from_f32<T>(o);
            }
    }
}

// dst[p][doff + c] = src[p][soff + c], c < n
template <typename T>
__global__ __launch_bounds__(256) void chan_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t npix, int sstride, int soff, int dstride, int doff,
                                                        int n) {
    const size_t total = npix * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n);
        const size_t p = i / n;
        dst[p * dstride + doff + c] = src[p * sstride + soff + c];
    }
}

// dst[p][c], every c < dstride, in one launch: a[p][aoff + c - adst] for c in [adst, adst + na), b[p][boff + c - bdst] for c in [bdst, bdst + nb)
// (b may be null), zero elsewhere -- a channel slice, its adjoint, a cat and the cat's two adjoints each as ONE fully written tensor instead
// of a zero fill plus one or two strided copies.  VEC: every offset / count / stride is a multiple of the 16-byte vector.
struct PlaceArgs {
    const void* a; const void* b; void* dst;
    size_t npix;
    int astride, aoff, adst, na, bstride, boff, bdst, nb, dstride;
};
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void chan_place_kernel(PlaceArgs g) {
    constexpr int VE = VEC ? 16 / (int)sizeof(T) : 1;
    const T* a = (const T*)g.a;
    const T* b = (const T*)g.b;
    T* dst = (T*)g.dst;
    const int per = g.dstride / VE;
    const size_t total = g.npix * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per) * VE;
        const size_t p = i / per;
        if constexpr (VEC) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 v = {0u, 0u, 0u, 0u};
            if (c >= g.adst && c < g.adst + g.na) v = *reinterpret_cast<const u4*>(a + p * g.astride + g.aoff + (c - g.adst));
            else if (b && c >= g.bdst && c < g.bdst + g.nb) v = *reinterpret_cast<const u4*>(b + p * g.bstride + g.boff + (c - g.bdst));
            *reinterpret_cast<u4*>(dst + p * g.dstride + c) = v;
        } else {
            T v = from_f32<T>(0.f);
            if (c >= g.adst && c < g.adst + g.na) v = a[p * g.astride + g.aoff + (c - g.adst)];
            else if (b && c >= g.bdst && c < g.bdst + g.nb) v = b[p * g.bstride + g.boff + (c - g.bdst)];
            dst[p * g.dstride + c] = v;
        }
    }
}

__device__ __forceinline__ float coupling_e(float s, float clamp, float eps) { return expf(clamp * (2.f / (1.f + expf(-s)) - 1.f)) + eps; }

template <typename T>
__global__ __launch_bounds__(256) void affine_fwd_kernel(const T* __restrict__ x, const T* __restrict__ s, const T* __restrict__ t, T* __restrict__ y, size_t n,
                                                         float clamp, float eps, int rev) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float e = coupling_e(to_f32(s[i]), clamp, eps), xv = to_f32(x[i]), tv = to_f32(t[i]);
        y[i] = from_f32<T>(rev ? (xv - tv) / e : e * xv + tv);
    }
}
// v = x (rev 0) or the output y (rev 1)
template <typename T>
__global__ __launch_bounds__(256) void affine_bwd_kernel(const T* __restrict__ g, const T* __restrict__ v, const T* __restrict__ s, T* __restrict__ gx,
                                                         T* __restrict__ gs, T* __restrict__ gt, size_t n, float clamp, float eps, int rev) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float sv = to_f32(s[i]), gv = to_f32(g[i]), vv = to_f32(v[i]);
        const float sg = 1.f / (1.f + expf(-sv));
        const float ex = expf(clamp * (2.f * sg - 1.f)), e = ex + eps;
        const float de = ex * clamp * 2.f * sg * (1.f - sg);
        if (!rev) {
            gx[i] = from_f32<T>(e * gv);
            gt[i] = from_f32<T>(gv);
            gs[i] = from_f32<T>(gv * vv * de);
        } else {
            const float ge = gv / e;
            gx[i] = from_f32<T>(ge);
            gt[i] = from_f32<T>(-ge);
            gs[i] = from_f32<T>(-ge * vv * de);
        }
    }
}

}  // namespace

// up == 0: analysis, in [B,2H,2W,CPin] -> out [B,H,W,CPout] (C = input channels, CPout >= 4C); up != 0: synthesis, in [B,H,W,CPin] ->
// out [B,2H,2W,CPout] (C = output channels, CPin >= 4C).  The two are each other's adjoint for equal fac.
extern "C" int wm_haar(const void* in, void* out, int B, int H, int W, int C, int CPin, int CPout, float fac, int up, int dtype, void* stream) {
    WM_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0, WM_E_BADARG, "wm_haar: bad arguments");
    WM_REQUIRE(up >= 0 && up <= 3, WM_E_BADARG, "wm_haar: up is a 2-bit field (bit 0 synthesis, bit 1 wavelet-major channel order)");
    WM_REQUIRE(CPin % 4 == 0 && CPout % 4 == 0 && ((up & 1) ? (CPin >= 4 * C && CPout >= C) : (CPin >= C && CPout >= 4 * C)), WM_E_SHAPE,
               "wm_haar: channel strides %d -> %d do not hold %d x4 channels", CPin, CPout, C);
    hipStream_t s = (hipStream_t)stream;
    const int bywav = (up >> 1) & 1;   // bit 1 of `up`: the 4C channels ordered by wavelet (k C + c) instead of by channel (4 c + k)
    if (up & 1) {
        const size_t n = (size_t)B * H * W * CPout;
        WM_DISPATCH_DTYPE(dtype, "wm_haar", hipLaunchKernelGGL(haar_up_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, (const T*)in, (T*)out, B, H, W, C, CPin, CPout, fac, bywav));
    } else {
        const size_t n = (size_t)B * H * W * (CPout / 4);
        WM_DISPATCH_DTYPE(dtype, "wm_haar", hipLaunchKernelGGL(haar_down_kernel<T>, dim3(grid1(n)), dim3(256), 0, s, (const T*)in, (T*)out, B, H, W, C, CPin, CPout, fac, bywav));
    }
    WM_LAUNCH_CHECK("wm_haar");
    return WM_OK;
}

extern "C" int wm_chan_copy(const void* src, void* dst, size_t npix, int sstride, int soff, int dstride, int doff, int n, int dtype, void* stream) {
    WM_REQUIRE(src && dst && npix > 0 && n > 0 && soff >= 0 && doff >= 0 && soff + n <= sstride && doff + n <= dstride, WM_E_BADARG,
               "wm_chan_copy: channel window [%d,%d) / [%d,%d) outside strides %d / %d", soff, soff + n, doff, doff + n, sstride, dstride);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_chan_copy",
        hipLaunchKernelGGL(chan_copy_kernel<T>, dim3(grid1(npix * n)), dim3(256), 0, s, (const T*)src, (T*)dst, npix, sstride, soff, dstride, doff, n));
    WM_LAUNCH_CHECK("wm_chan_copy");
    return WM_OK;
}

// dst [npix][dstride], written whole: a's window [aoff, aoff + na) lands at channel adst, b's (b may be NULL) at bdst, zero elsewhere
extern "C" int wm_chan_place(const void* a, int astride, int aoff, int adst, int na, const void* b, int bstride, int boff, int bdst, int nb, void* dst,
                             int dstride, size_t npix, int dtype, void* stream) {
    WM_REQUIRE(a && dst && npix > 0 && na > 0 && aoff >= 0 && adst >= 0 && aoff + na <= astride && adst + na <= dstride, WM_E_BADARG,
               "wm_chan_place: window a [%d,%d) -> [%d,%d) outside strides %d / %d", aoff, aoff + na, adst, adst + na, astride, dstride);
    WM_REQUIRE(!b || (nb > 0 && boff >= 0 && bdst >= 0 && boff + nb <= bstride && bdst + nb <= dstride && (bdst >= adst + na || bdst + nb <= adst)), WM_E_BADARG,
               "wm_chan_place: window b [%d,%d) -> [%d,%d) outside strides %d / %d or over window a", boff, boff + nb, bdst, bdst + nb, bstride, dstride);
    PlaceArgs g;
    g.a = a; g.b = b; g.dst = dst; g.npix = npix; g.astride = astride; g.aoff = aoff; g.adst = adst; g.na = na;
    g.bstride = b ? bstride : 0; g.boff = b ? boff : 0; g.bdst = b ? bdst : 0; g.nb = b ? nb : 0; g.dstride = dstride;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_chan_place", {
        const int ve = 16 / (int)sizeof(T);
        const bool vec = !((astride | aoff | adst | na | g.bstride | g.boff | g.bdst | g.nb | dstride) % ve);
        const size_t total = npix * (size_t)(vec ? dstride / ve : dstride);
        if (vec) hipLaunchKernelGGL((chan_place_kernel<T, true>), dim3(grid1(total)), dim3(256), 0, s, g);
        else hipLaunchKernelGGL((chan_place_kernel<T, false>), dim3(grid1(total)), dim3(256), 0, s, g);
    });
    WM_LAUNCH_CHECK("wm_chan_place");
    return WM_OK;
}

extern "C" int wm_coupling_fwd(const void* x, const void* s, const void* t, void* y, size_t n, float clamp, float eps, int rev, int dtype, void* stream) {
    WM_REQUIRE(x && s && t && y && n > 0, WM_E_BADARG, "wm_coupling_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_coupling_fwd",
        hipLaunchKernelGGL(affine_fwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, st, (const T*)x, (const T*)s, (const T*)t, (T*)y, n, clamp, eps, rev));
    WM_LAUNCH_CHECK("wm_coupling_fwd");
    return WM_OK;
}
extern "C" int wm_coupling_bwd(const void* g, const void* v, const void* s, void* gx, void* gs, void* gt, size_t n, float clamp, float eps, int rev, int dtype,
                               void* stream) {
    WM_REQUIRE(g && v && s && gx && gs && gt && n > 0, WM_E_BADARG, "wm_coupling_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_coupling_bwd",
        hipLaunchKernelGGL(affine_bwd_kernel<T>, dim3(grid1(n)), dim3(256), 0, st, (const T*)g, (const T*)v, (const T*)s, (T*)gx, (T*)gs, (T*)gt, n, clamp, eps, rev));
    WM_LAUNCH_CHECK("wm_coupling_bwd");
    return WM_OK;
}
