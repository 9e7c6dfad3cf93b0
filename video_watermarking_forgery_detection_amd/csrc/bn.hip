// BatchNorm2d (training mode) + ReLU: statistics finalisation, backward (two streaming passes),
// and the small consumers that apply the fused relu(scale*y+shift) on the fly
// (global average pool, channel-slice copy).  gfx950, HBM-bound streaming kernels:
// every thread moves 16-byte vectors of one pixel's channels, a wave covers whole pixels
// (128 contiguous bytes per 64 bf16 channels), reductions go wave -> LDS -> per-workgroup
// partial rows that a tiny second kernel sums in double (deterministic, no float atomics).
//
// Reference ops: nn.BatchNorm2d + nn.ReLU in hidden_models/conv_bn_relu.py:12-14 and
// network/UNet.py:67-97; nn.AdaptiveAvgPool2d in hidden_models/decoder.py:24,
// hidden_models/discriminator.py:16.
#include "wm_common.h"

namespace {

constexpr int RED_THREADS = 256;

// ------------------------------------------------------------------ partial-row reduction helpers
// rows[n][W] -> rows[0..R)[W] in place: output row j = sum of rows j, j+R, j+2R, ...  (block j only ever
// touches rows congruent to j, so reading and writing the same buffer is race-free).  Keeps the
// finalisation kernels short: they then walk at most R = 64 rows.
constexpr int TREE_ROWS = 64;
__global__ __launch_bounds__(256) void tree_reduce_rows_kernel(float* __restrict__ rows, int n, int W) {
    // thread -> (16-byte column, row sub-slice): with W = 128 a workgroup keeps 8 sub-slices x 4 loads in flight per column
    const int j = blockIdx.x, W4 = W >> 2;
    __shared__ float4 red[256];
    for (int cb = 0; cb < W4; cb += 256) {
        const int ncol = min(256, W4 - cb), nsub = 256 / ncol;
        const int cl = threadIdx.x % ncol, sub = threadIdx.x / ncol;
        float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (sub < nsub) {
            const float4* col = reinterpret_cast<const float4*>(rows) + cb + cl;
            const size_t st = (size_t)nsub * TREE_ROWS;
            size_t r = (size_t)j + (size_t)sub * TREE_ROWS;
            auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
            for (; r + 3 * st < (size_t)n; r += 4 * st) {
                const float4 v0 = col[r * W4], v1 = col[(r + st) * W4], v2 = col[(r + 2 * st) * W4], v3 = col[(r + 3 * st) * W4];
                add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
            }
            for (; r < (size_t)n; r += st) add(a0, col[r * W4]);
            a0.x = (a0.x + a1.x) + (a2.x + a3.x); a0.y = (a0.y + a1.y) + (a2.y + a3.y);
            a0.z = (a0.z + a1.z) + (a2.z + a3.z); a0.w = (a0.w + a1.w) + (a2.w + a3.w);
        }
        red[threadIdx.x] = a0;
        __syncthreads();   // every read of row j (sub-slice 0) is done before it is overwritten
        if (sub == 0) {
            float4 t = red[cl];
            for (int k = 1; k < nsub; ++k) { const float4 v = red[k * ncol + cl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            reinterpret_cast<float4*>(rows)[(size_t)j * W4 + cb + cl] = t;
        }
        __syncthreads();
    }
}
// row pitch not a multiple of 4 floats (the 1x1 head's 3 x (Cin+1) rows): one column per thread
__global__ __launch_bounds__(256) void tree_reduce_rows_scalar_kernel(float* __restrict__ rows, int n, int W) {
    const int j = blockIdx.x;
    for (int c = threadIdx.x; c < W; c += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = j;
        for (; r + 3 * TREE_ROWS < n; r += 4 * TREE_ROWS) {
            a0 += rows[(size_t)r * W + c];
            a1 += rows[(size_t)(r + TREE_ROWS) * W + c];
            a2 += rows[(size_t)(r + 2 * TREE_ROWS) * W + c];
            a3 += rows[(size_t)(r + 3 * TREE_ROWS) * W + c];
        }
        for (; r < n; r += TREE_ROWS) a0 += rows[(size_t)r * W + c];
        rows[(size_t)j * W + c] = (a0 + a1) + (a2 + a3);
    }
}
// returns the number of rows left
inline int tree_reduce_rows(float* rows, int n, int W, hipStream_t s) {
    if (n <= 4 * TREE_ROWS) return n;  // the finalisation kernels walk up to 256 rows (8 slices x 32) themselves
    if ((W & 3) == 0 && ((uintptr_t)rows & 15) == 0) hipLaunchKernelGGL(tree_reduce_rows_kernel, dim3(TREE_ROWS), dim3(256), 0, s, rows, n, W);
    else hipLaunchKernelGGL(tree_reduce_rows_scalar_kernel, dim3(TREE_ROWS), dim3(256), 0, s, rows, n, W);
    return TREE_ROWS;
}

// ------------------------------------------------------------------ forward statistics
// partials: [nparts][2][CP] (sum, sum of squares).  One workgroup per 8 channels, 32 slices of parts.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, int C, int CP,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* running_mean,
                                                          float* running_var, float momentum, float eps,
                                                          float* scale, float* shift, float* mean_out,
                                                          float* invstd_out) {
    // 8 channels x 32 row slices per workgroup: every thread's <= 8 row loads are independent and in flight together
    __shared__ double s1[32][8], s2[32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < CP) {
#pragma unroll 8
        for (int p = sl; p < nparts; p += 32) {
            a1 += (double)partials[((size_t)p * 2 + 0) * CP + c];
            a2 += (double)partials[((size_t)p * 2 + 1) * CP + c];
        }
    }
    s1[sl][cl] = a1; s2[sl][cl] = a2;
    __syncthreads();
    if (sl == 0 && c < CP) {
        for (int k = 1; k < 32; ++k) { a1 += s1[k][cl]; a2 += s2[k][cl]; }
        if (c < C) {
            const double m = a1 / count;
            double var = a2 / count - m * m;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * invstd;
            scale[c] = sc;
            shift[c] = beta[c] - (float)m * sc;
            mean_out[c] = (float)m;
            invstd_out[c] = invstd;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        } else {  // padded channel: contributes nothing downstream
            scale[c] = 0.f; shift[c] = 0.f; mean_out[c] = 0.f; invstd_out[c] = 0.f;
        }
    }
}

// ------------------------------------------------------------------ backward, pass 1
// thread -> (pixel slot, 16-byte channel vector).  VPP vectors per pixel, PPB pixels per block-iteration.
// BT threads per workgroup: the passes that emit partial rows run 1024-thread workgroups on a grid of <= 256, so that the
// finalisation walks 256 rows directly (no intermediate tree reduction); the pure apply pass streams with 256.
template <typename T, bool APPLY, int BT, bool GVEC>
__global__ __launch_bounds__(BT) void bn_bwd_kernel(const T* __restrict__ g, int ldg, const float* __restrict__ gvec,
                                                             const T* __restrict__ y, int ldy,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ coef, T* __restrict__ dy, int lddy,
                                                             float* __restrict__ partials, size_t npix, size_t hw, int CP,
                                                             int reverse) {
    constexpr int VE = vec16<T>::N;
    const int VPP = CP / VE;
    const int PPB = BT / VPP;  // pixels per block iteration (VPP divides 256 for CP in {32,64,128,256,512})
    const int vv = threadIdx.x % VPP, ps = threadIdx.x / VPP;
    const int c0 = vv * VE;
    float sc[VE], sh[VE], mu[VE], is[VE], ca[VE], c1[VE], c2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
        if (APPLY) { ca[e] = coef[c0 + e]; c1[e] = coef[CP + c0 + e]; c2[e] = coef[2 * CP + c0 + e]; }
    }
    // bf16 apply pass: the folded form of wm_common.h (shared with the kernels that fuse this pass)
    constexpr bool FOLD = APPLY && sizeof(T) == 2;
    float k2[VE], k3[VE];
    if (FOLD) {
#pragma unroll
        for (int e = 0; e < VE; ++e) wm_bn_fold(mu[e], is[e], ca[e], c1[e], c2[e], k2[e], k3[e]);
    }
    float a1[VE], a2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    // GVEC: the gradient is one vector per sample (a globally pooled output): its values, not a tensor, are streamed
    auto load_g = [&](size_t p, float (&gg)[VE]) {
        if (GVEC) {
            const float* gb = gvec + (p / hw) * CP + c0;
#pragma unroll
            for (int e = 0; e < VE; ++e) gg[e] = gb[e];
        } else {
            const vec16<T> gv = *reinterpret_cast<const vec16<T>*>(g + p * ldg + c0);
#pragma unroll
            for (int e = 0; e < VE; ++e) gg[e] = gv.get(e);
        }
    };
    auto body = [&](size_t p, const vec16<T>& yv, const float (&gg)[VE]) {
        vec16<T> out;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float yy = yv.get(e);
            if (FOLD) {
                const float d = GVEC ? wm_bn_fold_dy(yy, sc[e], sh[e], k2[e], k3[e], wm_bn_fold_g(ca[e], gg[e], k3[e]))
                                     : wm_bn_fold_dyg(yy, gg[e], sc[e], sh[e], ca[e], k2[e], k3[e]);
                out.set(e, d);
                a1[e] += d;
                continue;
            }
            const float z = sc[e] * yy + sh[e];
            const float gz = z > 0.f ? gg[e] : 0.f;
            const float xh = (yy - mu[e]) * is[e];
            if (APPLY) {
                const float d = ca[e] * (gz - c1[e] - xh * c2[e]);
                out.set(e, d);
                a1[e] += d;
            } else {
                a1[e] += gz;
                a2[e] += gz * xh;
            }
        }
        if (APPLY) *reinterpret_cast<vec16<T>*>(dy + p * lddy + c0) = out;
    };
    if (ps < PPB) {
        // reverse: sweep the tensors from the end, where the previous pass of the chain stopped (Infinity Cache reuse).
        // Two pixels per trip, all four loads issued before the arithmetic: a straight-line, branch-free body
        const size_t stride = (size_t)gridDim.x * PPB;
        size_t q = (size_t)blockIdx.x * PPB + ps;
        for (; q + stride < npix; q += 2 * stride) {
            const size_t p0 = reverse ? npix - 1 - q : q, p1 = reverse ? npix - 1 - (q + stride) : q + stride;
            const vec16<T> y0 = *reinterpret_cast<const vec16<T>*>(y + p0 * ldy + c0);
            const vec16<T> y1 = *reinterpret_cast<const vec16<T>*>(y + p1 * ldy + c0);
            float g0[VE], g1[VE];
            load_g(p0, g0);
            load_g(p1, g1);
            body(p0, y0, g0);
            body(p1, y1, g1);
        }
        if (q < npix) {
            const size_t p0 = reverse ? npix - 1 - q : q;
            const vec16<T> y0 = *reinterpret_cast<const vec16<T>*>(y + p0 * ldy + c0);
            float g0[VE];
            load_g(p0, g0);
            body(p0, y0, g0);
        }
    }
    if (!partials) return;
    // block reduction over the PPB pixel slots that share a channel vector
    __shared__ float red[APPLY ? 1 : 2][BT][vec16<T>::N + 1];
#pragma unroll
    for (int e = 0; e < VE; ++e) { red[0][threadIdx.x][e] = a1[e]; if (!APPLY) red[APPLY ? 0 : 1][threadIdx.x][e] = a2[e]; }
    __syncthreads();
    const int nwhich = APPLY ? 1 : 2;
    for (int i = threadIdx.x; i < nwhich * CP; i += BT) {
        const int which = i / CP, c = i - which * CP;
        const int v2 = c / VE, e = c - v2 * VE;
        float s = 0.f;
        for (int q = 0; q < PPB; ++q) s += red[which][q * VPP + v2][e];
        partials[((size_t)blockIdx.x * nwhich + which) * CP + c] = s;
    }
}

// partials [nparts][2][CP] -> dgamma, dbeta, coef[3][CP] (the body lives in wm_common.h: the weight-gradient slab reduction can carry
// it as extra workgroups).  raw_mean != nullptr: the second row holds sum(gz*y) (reduced by a dgrad epilogue, conv3x3_ws.hip BWDST)
// instead of sum(gz*xhat): xhat = (y-mean)*invstd is applied there, in double
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts, int C, int CP,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd, float* dgamma,
                                                              float* dbeta, int accumulate, float* coef,
                                                              const float* __restrict__ raw_mean,
                                                              const float* __restrict__ pgv = nullptr,
                                                              const float* __restrict__ pn = nullptr,
                                                              const float* __restrict__ ps = nullptr) {
    __shared__ double sh[2 * 32 * 8];
    WmBnBwdFin j;
    j.partials = partials; j.nparts = nparts; j.C = C; j.CP = CP; j.count = count; j.gamma = gamma; j.mean = raw_mean; j.invstd = invstd;
    j.dgamma = dgamma; j.dbeta = dbeta; j.accumulate = accumulate; j.coef = coef;
    wm_bn_bwd_finalize_block(j, (int)blockIdx.x, sh, pgv, pn, ps);
}

// out[c] (+)= sum_p partials[p*ldp + c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ partials, int nparts, int C, int ldp,
                                                     float* out, int accumulate, float mul, size_t in_bstride,
                                                     size_t out_bstride) {
    partials += (size_t)blockIdx.y * in_bstride;
    out += (size_t)blockIdx.y * out_bstride;
    __shared__ double s1[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a1 = 0.0;
    if (c < C)
        for (int p = sl; p < nparts; p += 8) a1 += (double)partials[(size_t)p * ldp + c];
    s1[sl][cl] = a1;
    __syncthreads();
    if (sl == 0 && c < C) {
        for (int k = 1; k < 8; ++k) a1 += s1[k][cl];
        out[c] = (accumulate ? out[c] : 0.f) + (float)(a1 * (double)mul);
    }
}

// ------------------------------------------------------------------ avg pool of relu(scale*y+shift)
// grid (S slices, B).  ws [B][S][NPL][CP]; NPL = 1: the pooled sums; NPL = 3 (training): also, per (sample, channel), the
// number of active pixels N+ = #[z > 0] and S+ = sum of y over them.  With the pooled layer's gradient one value gv[b,c] per
// (sample, channel), its BatchNorm-backward sums are sum(gz) = sum_b gv N+ and sum(gz*y) = sum_b gv S+: the backward needs no
// pass over y (pooled_bwd_rows_kernel below).
template <typename T, int NPL>
__global__ __launch_bounds__(RED_THREADS) void avgpool_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ ws,
                                                              size_t hw, int CP) {
    constexpr int VE = vec16<T>::N;
    const int VPP = CP / VE, PPB = RED_THREADS / VPP;
    const int vv = threadIdx.x % VPP, ps = threadIdx.x / VPP;
    const int c0 = vv * VE;
    const int b = blockIdx.y, S = gridDim.x;
    float sc[VE], sh[VE], a1[VE], a2[VE], a3[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
    if (ps < PPB) {
        auto body = [&](const vec16<T>& yv) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float yy = yv.get(e);
                const float z = sc[e] * yy + sh[e];
                a1[e] += fmaxf(z, 0.f);
                if (NPL == 3) { a2[e] += z > 0.f ? 1.f : 0.f; a3[e] += z > 0.f ? yy : 0.f; }
            }
        };
        // four pixels per trip, the loads issued before the arithmetic
        const size_t st = (size_t)S * PPB;
        const T* yb = y + (size_t)b * hw * ldy + c0;
        size_t p = (size_t)blockIdx.x * PPB + ps;
        for (; p + 3 * st < hw; p += 4 * st) {
            const vec16<T> v0 = *reinterpret_cast<const vec16<T>*>(yb + p * ldy);
            const vec16<T> v1 = *reinterpret_cast<const vec16<T>*>(yb + (p + st) * ldy);
            const vec16<T> v2 = *reinterpret_cast<const vec16<T>*>(yb + (p + 2 * st) * ldy);
            const vec16<T> v3 = *reinterpret_cast<const vec16<T>*>(yb + (p + 3 * st) * ldy);
            body(v0); body(v1); body(v2); body(v3);
        }
        for (; p < hw; p += st) body(*reinterpret_cast<const vec16<T>*>(yb + p * ldy));
    }
    __shared__ float red[NPL][RED_THREADS][vec16<T>::N + 1];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        red[0][threadIdx.x][e] = a1[e];
        if (NPL == 3) { red[NPL - 2][threadIdx.x][e] = a2[e]; red[NPL - 1][threadIdx.x][e] = a3[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NPL * CP; i += RED_THREADS) {
        const int pl = i / CP, c = i - pl * CP;
        const int v2 = c / VE, e = c - v2 * VE;
        float s = 0.f;
        for (int q = 0; q < PPB; ++q) s += red[pl][q * VPP + v2][e];
        ws[(((size_t)b * S + blockIdx.x) * NPL + pl) * CP + c] = s;
    }
}
// ws [B][S][3][CP] -> out3 [3][B][CP]: plane 0 = pooled mean (x inv_hw), planes 1, 2 = N+, S+.  grid (3*CP/32, B): 32 columns x 8
// slice groups per workgroup
__global__ __launch_bounds__(256) void avgpool_stats_finalize_kernel(const float* __restrict__ ws, int S, int B, int CP, float inv_hw,
                                                                     float* __restrict__ out3) {
    __shared__ double s1[8][32];
    const int b = blockIdx.y, cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;   // column of the [3*CP] row
    double a = 0.0;
    if (i < 3 * CP)
        for (int s = sl; s < S; s += 8) a += (double)ws[((size_t)b * S + s) * 3 * CP + i];
    s1[sl][cl] = a;
    __syncthreads();
    if (sl == 0 && i < 3 * CP) {
        for (int k = 1; k < 8; ++k) a += s1[k][cl];
        const int pl = i / CP, c = i - pl * CP;
        out3[((size_t)pl * B + b) * CP + c] = (float)(pl == 0 ? a * (double)inv_hw : a);
    }
}
// partial rows for wm_bn_bwd_finalize_raw from the pooled statistics: rows[b] = (gv[b,:] * N+[b,:], gv[b,:] * S+[b,:])
__global__ __launch_bounds__(256) void pooled_bwd_rows_kernel(const float* __restrict__ gvec, const float* __restrict__ npos,
                                                              const float* __restrict__ ysum, int B, int CP, float* __restrict__ rows) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * CP; i += gridDim.x * 256) {
        const int b = i / CP, c = i - b * CP;
        rows[((size_t)b * 2 + 0) * CP + c] = gvec[i] * npos[i];
        rows[((size_t)b * 2 + 1) * CP + c] = gvec[i] * ysum[i];
    }
}

// ------------------------------------------------------------------ channel-slice copy with fused BN+ReLU
template <typename T>
__global__ __launch_bounds__(256) void bnrelu_copy_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ y, int ldy, int c0,
                                                          size_t npix, int C) {
    constexpr int VE = vec16<T>::N;
    const int VPP = C / VE;
    const size_t total = npix * VPP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / VPP;
        const int c = (int)(i - p * VPP) * VE;
        vec16<T> v = *reinterpret_cast<const vec16<T>*>(x + p * ldx + c);
        if (scale) {
#pragma unroll
            for (int e = 0; e < VE; ++e) v.set(e, fmaxf(scale[c + e] * v.get(e) + shift[c + e], 0.f));
        }
        *reinterpret_cast<vec16<T>*>(y + p * ldy + c0 + c) = v;
    }
}

bool cp_ok(int CP, int dtype) {
    const int ve = dtype != WM_F32 ? 8 : 4;
    if (CP <= 0 || CP % ve) return false;
    const int vpp = CP / ve;
    return vpp <= 256 && (256 % vpp) == 0;
}

}  // namespace

extern "C" int wm_bn_finalize(const float* partials, int nparts, int C, int CP, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              float* scale, float* shift, float* mean, float* invstd, void* stream) {
    WM_REQUIRE(partials && gamma && beta && scale && shift && mean && invstd, WM_E_BADARG, "wm_bn_finalize: null pointer");
    WM_REQUIRE(nparts > 0 && C > 0 && CP >= C && count > 0, WM_E_BADARG, "wm_bn_finalize: bad sizes");
    WM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), WM_E_BADARG, "wm_bn_finalize: running stats must come together");
    nparts = tree_reduce_rows(const_cast<float*>(partials), nparts, 2 * CP, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(wm_cdiv(CP, 8)), dim3(256), 0, (hipStream_t)stream, partials, nparts, C, CP,
                       count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd);
    WM_LAUNCH_CHECK("wm_bn_finalize");
    return WM_OK;
}

constexpr int BWD_BT = 1024;
WM_KNOB_INT(g_bn_reverse, "WM_BN_REVERSE", 1);   // bit 0: reduce pass sweeps backwards (it follows a forward-sweeping dgrad), bit 1: apply pass
WM_KNOB_SETTER(wm_debug_bn_reverse, g_bn_reverse)
extern "C" int wm_bn_bwd_nparts(size_t npix) {
    const size_t n = (npix + BWD_BT - 1) / BWD_BT;
    return (int)(n < 1 ? 1 : (n > 256 ? 256 : n));
}

extern "C" int wm_bn_bwd_reduce(const void* g, int ldg, const float* gvec, const void* y, int ldy, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float* partials, int B,
                                size_t hw, int CP, int dtype, void* stream) {
    WM_REQUIRE((g != nullptr) != (gvec != nullptr), WM_E_BADARG, "wm_bn_bwd_reduce: exactly one of g / gvec");
    WM_REQUIRE(y && scale && shift && mean && invstd && partials, WM_E_BADARG, "wm_bn_bwd_reduce: null pointer");
    WM_REQUIRE(cp_ok(CP, dtype), WM_E_SHAPE, "wm_bn_bwd_reduce: unsupported channel count CP=%d", CP);
    const size_t npix = (size_t)B * hw;
    const int nparts = wm_bn_bwd_nparts(npix);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_bn_bwd_reduce",
        if (g) hipLaunchKernelGGL((bn_bwd_kernel<T, false, BWD_BT, false>), dim3(nparts), dim3(BWD_BT), 0, s, (const T*)g, ldg, gvec,
                           (const T*)y, ldy, scale, shift, mean, invstd, (const float*)nullptr, (T*)nullptr, 0, partials,
                           npix, hw, CP, g_bn_reverse & 1);
        else hipLaunchKernelGGL((bn_bwd_kernel<T, false, BWD_BT, true>), dim3(nparts), dim3(BWD_BT), 0, s, (const T*)g, ldg, gvec,
                           (const T*)y, ldy, scale, shift, mean, invstd, (const float*)nullptr, (T*)nullptr, 0, partials,
                           npix, hw, CP, g_bn_reverse & 1));
    WM_LAUNCH_CHECK("wm_bn_bwd_reduce");
    return WM_OK;
}

extern "C" int wm_bn_bwd_finalize(const float* partials, int nparts, int C, int CP, double count, const float* gamma,
                                  const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef,
                                  void* stream) {
    WM_REQUIRE(partials && gamma && invstd && coef, WM_E_BADARG, "wm_bn_bwd_finalize: null pointer");
    nparts = tree_reduce_rows(const_cast<float*>(partials), nparts, 2 * CP, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(wm_cdiv(CP, 8)), dim3(256), 0, (hipStream_t)stream, partials, nparts, C,
                       CP, count, gamma, invstd, dgamma, dbeta, accumulate, coef, (const float*)nullptr);
    WM_LAUNCH_CHECK("wm_bn_bwd_finalize");
    return WM_OK;
}

extern "C" int wm_bn_bwd_finalize_raw(const float* partials, int nparts, int C, int CP, double count, const float* gamma,
                                      const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                      float* coef, void* stream) {
    WM_REQUIRE(partials && gamma && mean && invstd && coef, WM_E_BADARG, "wm_bn_bwd_finalize_raw: null pointer");
    nparts = tree_reduce_rows(const_cast<float*>(partials), nparts, 2 * CP, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(wm_cdiv(CP, 8)), dim3(256), 0, (hipStream_t)stream, partials, nparts, C,
                       CP, count, gamma, invstd, dgamma, dbeta, accumulate, coef, mean);
    WM_LAUNCH_CHECK("wm_bn_bwd_finalize_raw");
    return WM_OK;
}

extern "C" int wm_bn_bwd_apply(const void* g, int ldg, const float* gvec, const void* y, int ldy, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* coef, void* dy,
                               int lddy, float* dbias_partials, int B, size_t hw, int CP, int dtype, void* stream) {
    WM_REQUIRE((g != nullptr) != (gvec != nullptr), WM_E_BADARG, "wm_bn_bwd_apply: exactly one of g / gvec");
    WM_REQUIRE(y && scale && shift && mean && invstd && coef && dy, WM_E_BADARG, "wm_bn_bwd_apply: null pointer");
    WM_REQUIRE(cp_ok(CP, dtype), WM_E_SHAPE, "wm_bn_bwd_apply: unsupported channel count CP=%d", CP);
    const size_t npix = (size_t)B * hw;
    const int nparts = wm_bn_bwd_nparts(npix);
    hipStream_t s = (hipStream_t)stream;
    if (dbias_partials) {   // rows for wm_colsum_finalize: same grid as the reduce pass
        WM_DISPATCH_DTYPE(dtype, "wm_bn_bwd_apply",
            if (g) hipLaunchKernelGGL((bn_bwd_kernel<T, true, BWD_BT, false>), dim3(nparts), dim3(BWD_BT), 0, s, (const T*)g, ldg, gvec,
                               (const T*)y, ldy, scale, shift, mean, invstd, coef, (T*)dy, lddy, dbias_partials, npix, hw, CP, (g_bn_reverse >> 1) & 1);
            else hipLaunchKernelGGL((bn_bwd_kernel<T, true, BWD_BT, true>), dim3(nparts), dim3(BWD_BT), 0, s, (const T*)g, ldg, gvec,
                               (const T*)y, ldy, scale, shift, mean, invstd, coef, (T*)dy, lddy, dbias_partials, npix, hw, CP, (g_bn_reverse >> 1) & 1));
    } else {
        const size_t nb = (npix + 255) / 256;
        const int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
        WM_DISPATCH_DTYPE(dtype, "wm_bn_bwd_apply",
            if (g) hipLaunchKernelGGL((bn_bwd_kernel<T, true, 256, false>), dim3(grid), dim3(256), 0, s, (const T*)g, ldg, gvec,
                               (const T*)y, ldy, scale, shift, mean, invstd, coef, (T*)dy, lddy, dbias_partials, npix, hw, CP, (g_bn_reverse >> 1) & 1);
            else hipLaunchKernelGGL((bn_bwd_kernel<T, true, 256, true>), dim3(grid), dim3(256), 0, s, (const T*)g, ldg, gvec,
                               (const T*)y, ldy, scale, shift, mean, invstd, coef, (T*)dy, lddy, dbias_partials, npix, hw, CP, (g_bn_reverse >> 1) & 1));
    }
    WM_LAUNCH_CHECK("wm_bn_bwd_apply");
    return WM_OK;
}

extern "C" int wm_colsum_finalize(const float* partials, int nparts, int C, int ldp, float* out, int accumulate,
                                  void* stream) {
    WM_REQUIRE(partials && out && nparts > 0 && C > 0 && ldp >= C, WM_E_BADARG, "wm_colsum_finalize: bad arguments");
    nparts = tree_reduce_rows(const_cast<float*>(partials), nparts, ldp, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_kernel, dim3(wm_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, partials, nparts, C, ldp, out,
                       accumulate, 1.0f, (size_t)0, (size_t)0);
    WM_LAUNCH_CHECK("wm_colsum_finalize");
    return WM_OK;
}

extern "C" int wm_avgpool_slices(size_t hw) {
    const size_t n = (hw + 1023) / 1024;
    return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}

extern "C" int wm_bnrelu_avgpool(const void* y, int ldy, const float* scale, const float* shift, float* out, float* ws,
                                 int B, size_t hw, int CP, int dtype, void* stream) {
    WM_REQUIRE(y && scale && shift && out && ws, WM_E_BADARG, "wm_bnrelu_avgpool: null pointer");
    WM_REQUIRE(cp_ok(CP, dtype), WM_E_SHAPE, "wm_bnrelu_avgpool: unsupported channel count CP=%d", CP);
    const int S = wm_avgpool_slices(hw);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_bnrelu_avgpool",
        hipLaunchKernelGGL((avgpool_kernel<T, 1>), dim3(S, B), dim3(RED_THREADS), 0, s, (const T*)y, ldy, scale, shift, ws, hw, CP));
    WM_LAUNCH_CHECK("wm_bnrelu_avgpool");
    // ws viewed as [B][S*CP]: per sample reduce S rows of CP -> out[b][CP], scaled by 1/hw
    hipLaunchKernelGGL(colsum_kernel, dim3(wm_cdiv(CP, 32), B), dim3(256), 0, s, ws, S, CP, CP, out, 0,
                       (float)(1.0 / (double)hw), (size_t)S * CP, (size_t)CP);
    WM_LAUNCH_CHECK("wm_bnrelu_avgpool(finalize)");
    return WM_OK;
}

WM_KNOB_ON(g_pool_stats, "WM_NO_POOL_STATS");
WM_KNOB_SETTER(wm_debug_pool_stats, g_pool_stats)   // A/B knob (tools/ab_step.py, debug build only)
extern "C" int wm_pool_stats_enabled(void) { return g_pool_stats; }

extern "C" int wm_bnrelu_avgpool_stats(const void* y, int ldy, const float* scale, const float* shift, float* out3, float* ws, int B,
                                       size_t hw, int CP, int dtype, void* stream) {
    WM_REQUIRE(y && scale && shift && out3 && ws, WM_E_BADARG, "wm_bnrelu_avgpool_stats: null pointer");
    WM_REQUIRE(cp_ok(CP, dtype), WM_E_SHAPE, "wm_bnrelu_avgpool_stats: unsupported channel count CP=%d", CP);
    const int S = wm_avgpool_slices(hw);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_bnrelu_avgpool_stats",
        hipLaunchKernelGGL((avgpool_kernel<T, 3>), dim3(S, B), dim3(RED_THREADS), 0, s, (const T*)y, ldy, scale, shift, ws, hw, CP));
    WM_LAUNCH_CHECK("wm_bnrelu_avgpool_stats");
    hipLaunchKernelGGL(avgpool_stats_finalize_kernel, dim3(wm_cdiv(3 * CP, 32), B), dim3(256), 0, s, ws, S, B, CP, (float)(1.0 / (double)hw), out3);
    WM_LAUNCH_CHECK("wm_bnrelu_avgpool_stats(finalize)");
    return WM_OK;
}

// the pooled layer's whole reduce + finalisation in one launch: rows formed from (gvec, N+, S+) on the fly
extern "C" int wm_bn_bwd_finalize_pooled(const float* gvec, const float* npos, const float* ysum, int B, int C, int CP, double count,
                                         const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                         int accumulate, float* coef, void* stream) {
    WM_REQUIRE(gvec && npos && ysum && gamma && mean && invstd && coef && B > 0, WM_E_BADARG, "wm_bn_bwd_finalize_pooled: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(wm_cdiv(CP, 8)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, B, C, CP,
                       count, gamma, invstd, dgamma, dbeta, accumulate, coef, mean, gvec, npos, ysum);
    WM_LAUNCH_CHECK("wm_bn_bwd_finalize_pooled");
    return WM_OK;
}

extern "C" int wm_pooled_bn_bwd_rows(const float* gvec, const float* npos, const float* ysum, int B, int CP, float* rows, void* stream) {
    WM_REQUIRE(gvec && npos && ysum && rows && B > 0 && CP > 0, WM_E_BADARG, "wm_pooled_bn_bwd_rows: bad arguments");
    hipLaunchKernelGGL(pooled_bwd_rows_kernel, dim3(wm_cdiv(B * CP, 256)), dim3(256), 0, (hipStream_t)stream, gvec, npos, ysum, B, CP, rows);
    WM_LAUNCH_CHECK("wm_pooled_bn_bwd_rows");
    return WM_OK;
}

extern "C" int wm_bnrelu_copy(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy, int c0,
                              size_t npix, int C, int dtype, void* stream) {
    WM_REQUIRE(x && y, WM_E_BADARG, "wm_bnrelu_copy: null pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_bnrelu_copy: scale/shift must come together");
    const int ve = dtype != WM_F32 ? 8 : 4;
    WM_REQUIRE(C > 0 && C % ve == 0 && c0 % ve == 0 && ldx % ve == 0 && ldy % ve == 0, WM_E_SHAPE,
               "wm_bnrelu_copy: C=%d c0=%d ldx=%d ldy=%d must be multiples of %d", C, c0, ldx, ldy, ve);
    const size_t total = npix * (C / ve);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_bnrelu_copy",
        hipLaunchKernelGGL((bnrelu_copy_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)x, ldx, scale, shift, (T*)y, ldy, c0, npix, C));
    WM_LAUNCH_CHECK("wm_bnrelu_copy");
    return WM_OK;
}
