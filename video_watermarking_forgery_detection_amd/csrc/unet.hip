// UNet-specific pieces (network/UNet.py:7-97 of the reference) around the shared conv3x3 path:
//   MaxPool2d(2,2)            on the fused relu(scale*y+shift) of an encoder block (UNet.py:14-20)
//   ConvTranspose2d(k2,s2)    the four up-convolutions (UNet.py:24-38): non-overlapping, i.e. four
//                             independent 1x1 GEMMs with a pixel-shuffle store straight into the
//                             channel slice of the concat buffer the decoder block reads
// Both are bandwidth-bound next to the 3x3 convolutions (the up-convs are <2% of the UNet FLOPs),
// so they are plain coalesced kernels: 16-byte channel vectors per lane, LDS-staged pixel tiles for
// the small GEMMs (f32 VALU accumulate), deterministic partial sums for the weight gradients.
#include "wm_common.h"

namespace {

// ------------------------------------------------------------------ max pool 2x2 of relu(scale*y+shift)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ pooled, int ldp,
                                                          T* __restrict__ act, int lda, int c0a, int B, int H, int W, int C) {
    constexpr int VE = vec16<T>::N;
    const int VPP = C / VE, OH = H / 2, OW = W / 2;
    const size_t total = (size_t)B * OH * OW * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int vv = (int)(i % VPP);
        size_t p = i / VPP;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        const int c = vv * VE;
        float sc[VE], sh[VE], best[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) { sc[e] = scale[c + e]; sh[e] = shift[c + e]; best[e] = 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t pin = ((size_t)b * H + 2 * oh + (q >> 1)) * W + 2 * ow + (q & 1);
            const vec16<T> v = *reinterpret_cast<const vec16<T>*>(y + pin * ldy + c);
            vec16<T> a;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float t = fmaxf(sc[e] * v.get(e) + sh[e], 0.f);
                a.set(e, t);
                best[e] = fmaxf(best[e], a.get(e));  // compare the stored (rounded) activations
            }
            if (act) *reinterpret_cast<vec16<T>*>(act + pin * lda + c0a + c) = a;
        }
        vec16<T> o;
#pragma unroll
        for (int e = 0; e < VE; ++e) o.set(e, best[e]);
        *reinterpret_cast<vec16<T>*>(pooled + (((size_t)b * OH + oh) * OW + ow) * ldp + c) = o;
    }
}

// g[full] = g_skip[full] + (first arg-max of the 2x2 window ? gpooled : 0)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ gp, int ldgp,
                                                          const T* __restrict__ gs, int ldgs, T* __restrict__ g, int ldg,
                                                          int B, int H, int W, int C) {
    constexpr int VE = vec16<T>::N;
    const int VPP = C / VE, OH = H / 2, OW = W / 2;
    const size_t total = (size_t)B * OH * OW * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int vv = (int)(i % VPP);
        size_t p = i / VPP;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        const int c = vv * VE;
        float a[4][VE];
        size_t pin[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pin[q] = ((size_t)b * H + 2 * oh + (q >> 1)) * W + 2 * ow + (q & 1);
            const vec16<T> v = *reinterpret_cast<const vec16<T>*>(y + pin[q] * ldy + c);
#pragma unroll
            for (int e = 0; e < VE; ++e) a[q][e] = to_f32(from_f32<T>(fmaxf(scale[c + e] * v.get(e) + shift[c + e], 0.f)));
        }
        const vec16<T> gpv = *reinterpret_cast<const vec16<T>*>(gp + (((size_t)b * OH + oh) * OW + ow) * ldgp + c);
        int sel[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            int s = 0;
            float m = a[0][e];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (a[q][e] > m) { m = a[q][e]; s = q; }  // strict >: first maximum in row-major order (ATen max_pool2d)
            sel[e] = s;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vec16<T> o;
            if (gs) o = *reinterpret_cast<const vec16<T>*>(gs + pin[q] * ldgs + c);
#pragma unroll
            for (int e = 0; e < VE; ++e) o.set(e, (gs ? o.get(e) : 0.f) + (sel[e] == q ? gpv.get(e) : 0.f));
            *reinterpret_cast<vec16<T>*>(g + pin[q] * ldg + c) = o;
        }
    }
}

// ------------------------------------------------------------------ up-convolution as a small GEMM
// out[p][n] = sum_k in[p][k] * w[k][n]   (f32 accumulate), 16 pixels per workgroup staged in LDS.
//   MODE 0 (forward) : in = relu(scale*x+shift) (or x), n = (ij, co) -> store y[b,2h+i,2w+j,c0+co] + bias[co]
//   MODE 1 (backward): in[p][k=(ij,co)] gathered from gy[b,2h+i,2w+j,c0+co], w = weight^T [4Cout][Cin], store gx[p][n]
constexpr int UP_PIX = 16;
template <typename T, int MODE>
__global__ __launch_bounds__(256) void upconv_gemm_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ out, int ldout,
                                                          int c0, int B, int H, int W, int K, int N, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float s_in[];  // [UP_PIX][K]
    const size_t npix = (size_t)B * H * W;
    const size_t p0 = (size_t)blockIdx.x * UP_PIX;
    for (int i = threadIdx.x; i < UP_PIX * K; i += 256) {
        const int pp = i / K, k = i - pp * K;
        const size_t p = p0 + pp;
        float v = 0.f;
        if (p < npix) {
            if (MODE == 0) {
                v = to_f32(in[p * ldin + k]);
                if (scale) v = fmaxf(scale[k] * v + shift[k], 0.f);
            } else {
                const int ij = k / Cout, co = k - ij * Cout;
                const int w_ = (int)(p % W);
                const int h_ = (int)((p / W) % H);
                const size_t b = p / ((size_t)W * H);
                v = to_f32(in[((b * 2 * H + 2 * h_ + (ij >> 1)) * 2 * W + 2 * w_ + (ij & 1)) * ldin + c0 + co]);
            }
        }
        s_in[i] = v;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float acc[UP_PIX];
#pragma unroll
        for (int pp = 0; pp < UP_PIX; ++pp) acc[pp] = 0.f;
        // forward weight layout [Cin][Cout][2][2]: column n = (ij, co) lives at k*4Cout + co*4 + ij
        const int ij = (MODE == 0) ? n / Cout : 0, co = (MODE == 0) ? n - ij * Cout : 0;
        const float* wcol = (MODE == 0) ? w + co * 4 + ij : w + n;
        const int wstride = (MODE == 0) ? 4 * Cout : N;
        for (int k = 0; k < K; ++k) {
            const float wv = wcol[(size_t)k * wstride];
#pragma unroll
            for (int pp = 0; pp < UP_PIX; ++pp) acc[pp] += s_in[pp * K + k] * wv;
        }
        const float bv = (MODE == 0 && bias) ? bias[co] : 0.f;
#pragma unroll
        for (int pp = 0; pp < UP_PIX; ++pp) {
            const size_t p = p0 + pp;
            if (p >= npix) break;
            if (MODE == 0) {
                const int w_ = (int)(p % W);
                const int h_ = (int)((p / W) % H);
                const size_t b = p / ((size_t)W * H);
                out[((b * 2 * H + 2 * h_ + (ij >> 1)) * 2 * W + 2 * w_ + (ij & 1)) * ldout + c0 + co] = from_f32<T>(acc[pp] + bv);
            } else {
                out[p * ldout + n] = from_f32<T>(acc[pp]);
            }
        }
    }
}

// dW partials: grid (pixel chunks, ci tiles of 16, n tiles of 16); thread (ci, n) of the 16x16 tile.
// partial[chunk][ci][n(co*4+ij order of the PyTorch weight)] ; bias partial in the extra row Cin.
constexpr int DW_PIX = 64;
template <typename T>
__global__ __launch_bounds__(256) void upconv_dw_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const T* __restrict__ gy, int ldgy,
                                                        int c0, float* __restrict__ partials, int B, int H, int W, int Cin,
                                                        int Cout, int chunk_pix) {
    __shared__ float sa[DW_PIX][17];
    __shared__ float sg[DW_PIX][17];
    const int N = 4 * Cout;
    const int ci_t = blockIdx.y * 16, n_t = blockIdx.z * 16;
    const int tci = threadIdx.x >> 4, tn = threadIdx.x & 15;
    const size_t npix = (size_t)B * H * W;
    const size_t pbeg = (size_t)blockIdx.x * chunk_pix;
    const size_t pend = pbeg + chunk_pix < npix ? pbeg + chunk_pix : npix;
    float acc = 0.f, bacc = 0.f;
    for (size_t pb = pbeg; pb < pend; pb += DW_PIX) {
        __syncthreads();
        for (int i = threadIdx.x; i < DW_PIX * 16; i += 256) {
            const int pp = i >> 4, j = i & 15;
            const size_t p = pb + pp;
            float av = 0.f, gv = 0.f;
            if (p < pend) {
                const int ci = ci_t + j;
                if (ci < Cin) {
                    av = to_f32(x[p * ldx + ci]);
                    if (scale) av = fmaxf(scale[ci] * av + shift[ci], 0.f);
                }
                const int n = n_t + j;  // PyTorch order: n = co*4 + ij
                if (n < N) {
                    const int co = n >> 2, ij = n & 3;
                    const int w_ = (int)(p % W);
                    const int h_ = (int)((p / W) % H);
                    const size_t b = p / ((size_t)W * H);
                    gv = to_f32(gy[((b * 2 * H + 2 * h_ + (ij >> 1)) * 2 * W + 2 * w_ + (ij & 1)) * ldgy + c0 + co]);
                }
            }
            sa[pp][j] = av;
            sg[pp][j] = gv;
        }
        __syncthreads();
#pragma unroll 8
        for (int pp = 0; pp < DW_PIX; ++pp) {
            acc += sa[pp][tci] * sg[pp][tn];
            if (tci == 0) bacc += sg[pp][tn];
        }
    }
    float* prow = partials + (size_t)blockIdx.x * (Cin + 1) * N;
    if (ci_t + tci < Cin && n_t + tn < N) prow[(size_t)(ci_t + tci) * N + n_t + tn] = acc;
    if (blockIdx.y == 0 && tci == 0 && n_t + tn < N) prow[(size_t)Cin * N + n_t + tn] = bacc;
}

inline int grid_for(size_t n) {
    const size_t g = (n + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

bool vec_ok(int C, int ld, int dtype) {
    const int ve = dtype != WM_F32 ? 8 : 4;
    return C > 0 && C % ve == 0 && ld % ve == 0;
}

}  // namespace

extern "C" int wm_bnrelu_maxpool2(const void* y, int ldy, const float* scale, const float* shift, void* pooled, int ldp,
                                  void* act_out, int lda, int c0a, int B, int H, int W, int C, int dtype, void* stream) {
    WM_REQUIRE(y && scale && shift && pooled, WM_E_BADARG, "wm_bnrelu_maxpool2: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, WM_E_SHAPE, "wm_bnrelu_maxpool2: H, W must be even (got %dx%d)", H, W);
    WM_REQUIRE(vec_ok(C, ldy, dtype) && vec_ok(C, ldp, dtype) && (!act_out || (vec_ok(C, lda, dtype) && c0a % (dtype != WM_F32 ? 8 : 4) == 0)),
               WM_E_SHAPE, "wm_bnrelu_maxpool2: channel counts / strides must be 16-byte multiples");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / (dtype != WM_F32 ? 8 : 4));
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_bnrelu_maxpool2",
        hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)y, ldy, scale, shift,
                           (T*)pooled, ldp, (T*)act_out, lda, c0a, B, H, W, C));
    WM_LAUNCH_CHECK("wm_bnrelu_maxpool2");
    return WM_OK;
}

extern "C" int wm_maxpool2_bwd(const void* y, int ldy, const float* scale, const float* shift, const void* gpooled, int ldgp,
                               const void* g_skip, int ldgs, void* g, int ldg, int B, int H, int W, int C, int dtype,
                               void* stream) {
    WM_REQUIRE(y && scale && shift && gpooled && g, WM_E_BADARG, "wm_maxpool2_bwd: null pointer");
    WM_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0, WM_E_SHAPE, "wm_maxpool2_bwd: H, W must be even");
    WM_REQUIRE(vec_ok(C, ldy, dtype) && vec_ok(C, ldgp, dtype) && vec_ok(C, ldg, dtype) && (!g_skip || vec_ok(C, ldgs, dtype)), WM_E_SHAPE,
               "wm_maxpool2_bwd: channel counts / strides must be 16-byte multiples");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / (dtype != WM_F32 ? 8 : 4));
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_maxpool2_bwd",
        hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)y, ldy, scale, shift,
                           (const T*)gpooled, ldgp, (const T*)g_skip, ldgs, (T*)g, ldg, B, H, W, C));
    WM_LAUNCH_CHECK("wm_maxpool2_bwd");
    return WM_OK;
}

extern "C" int wm_upconv2x2_fwd(const void* x, int ldx, const float* scale, const float* shift, const float* w,
                                const float* bias, void* y, int ldy, int c0, int B, int H, int W, int Cin, int Cout,
                                int dtype, void* stream) {
    WM_REQUIRE(x && w && y, WM_E_BADARG, "wm_upconv2x2_fwd: null pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_upconv2x2_fwd: scale/shift must come together");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldy >= c0 + Cout, WM_E_BADARG, "wm_upconv2x2_fwd: bad shape");
    WM_REQUIRE((size_t)UP_PIX * Cin * 4 <= 64 * 1024, WM_E_SHAPE, "wm_upconv2x2_fwd: Cin=%d too large for the LDS tile", Cin);
    const size_t npix = (size_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((npix + UP_PIX - 1) / UP_PIX)), block(256);
    const size_t lds = (size_t)UP_PIX * Cin * sizeof(float);
    WM_DISPATCH_DTYPE(dtype, "wm_upconv2x2_fwd",
        hipLaunchKernelGGL((upconv_gemm_kernel<T, 0>), grid, block, lds, s, (const T*)x, ldx, scale, shift, w, bias, (T*)y, ldy, c0,
                           B, H, W, Cin, 4 * Cout, Cout));
    WM_LAUNCH_CHECK("wm_upconv2x2_fwd");
    return WM_OK;
}

extern "C" int wm_upconv2x2_dw_chunks(int B, int H, int W) {
    const size_t npix = (size_t)B * H * W;
    const size_t c = (npix + 1023) / 1024;
    return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}

extern "C" int wm_upconv2x2_bwd(const void* x, int ldx, const float* scale, const float* shift, const float* w_t,
                                const void* gy, int ldgy, int c0, void* gx, int ldgx, float* dw_partials, int B, int H,
                                int W, int Cin, int Cout, int dtype, void* stream) {
    WM_REQUIRE(x && w_t && gy && gx && dw_partials, WM_E_BADARG, "wm_upconv2x2_bwd: null pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_upconv2x2_bwd: scale/shift must come together");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldgx >= Cin && ldgy >= c0 + Cout, WM_E_BADARG, "wm_upconv2x2_bwd: bad shape");
    const int N = 4 * Cout;
    WM_REQUIRE((size_t)UP_PIX * N * 4 <= 64 * 1024, WM_E_SHAPE, "wm_upconv2x2_bwd: Cout=%d too large for the LDS tile", Cout);
    const size_t npix = (size_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((npix + UP_PIX - 1) / UP_PIX)), block(256);
    const size_t lds = (size_t)UP_PIX * N * sizeof(float);
    const int chunks = wm_upconv2x2_dw_chunks(B, H, W);
    const int chunk_pix = (int)(((npix + chunks - 1) / chunks + DW_PIX - 1) / DW_PIX * DW_PIX);
    const dim3 gdw((unsigned)chunks, (unsigned)wm_cdiv(Cin, 16), (unsigned)wm_cdiv(N, 16));
    WM_DISPATCH_DTYPE(dtype, "wm_upconv2x2_bwd",
        hipLaunchKernelGGL((upconv_gemm_kernel<T, 1>), grid, block, lds, s, (const T*)gy, ldgy, (const float*)nullptr, (const float*)nullptr,
                           w_t, (const float*)nullptr, (T*)gx, ldgx, c0, B, H, W, N, Cin, Cout);
        hipLaunchKernelGGL((upconv_dw_kernel<T>), gdw, block, 0, s, (const T*)x, ldx, scale, shift, (const T*)gy, ldgy, c0, dw_partials,
                           B, H, W, Cin, Cout, chunk_pix));
    WM_LAUNCH_CHECK("wm_upconv2x2_bwd");
    return WM_OK;
}
