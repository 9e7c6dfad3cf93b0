// 1x1 convolution heads on the fused relu(scale*y+shift) of the last ConvBNRelu block:
//   hidden_models/encoder.py:28,42  final_layer = nn.Conv2d(64, 3, kernel_size=1)  -> encoded image (NCHW f32)
//   network/UNet.py:41-43,65        conv = nn.Conv2d(32, 1, 1) + sigmoid           -> tamper mask   (NCHW f32)
// Cout <= 4, so this is a bandwidth-bound reduction, not a GEMM: VPP = Cin/VE lanes share one pixel,
// each lane loads one 16-byte channel vector (a wave reads 64/VPP whole pixels, fully contiguous),
// forms its partial dot products and the lanes of a pixel combine with wave shuffles.
#include "wm_common.h"

namespace {

template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, size_t npix,
                                                       size_t hw, int Cin, int act, T* __restrict__ act16) {
    constexpr int VE = vec16<T>::N;
    const int VPP = Cin / VE, PPB = 256 / VPP;
    const int vv = threadIdx.x % VPP, ps = threadIdx.x / VPP;
    const int c0 = vv * VE;
    float sc[VE], sh[VE], wr[COUT][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sc[e] = scale ? scale[c0 + e] : 1.f;
        sh[e] = scale ? shift[c0 + e] : 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co) wr[co][e] = w[co * Cin + c0 + e];
    }
    // UN pixels per thread and trip, all loads issued before the arithmetic (one load in flight per thread ran at 2.7 TB/s)
    constexpr int UN = 4;
    for (size_t base = (size_t)blockIdx.x * PPB * UN; base < npix; base += (size_t)gridDim.x * PPB * UN) {
        vec16<T> yv[UN];
        bool valid[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t p = base + (size_t)u * PPB + ps;
            valid[u] = p < npix;
            yv[u] = *reinterpret_cast<const vec16<T>*>(y + (valid[u] ? p : npix - 1) * ldy + c0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t p = base + (size_t)u * PPB + ps;
            float part[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) part[co] = 0.f;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float a = sc[e] * yv[u].get(e) + sh[e];
                if (scale) a = fmaxf(a, 0.f);
#pragma unroll
                for (int co = 0; co < COUT; ++co) part[co] += wr[co][e] * a;
            }
            for (int o = VPP >> 1; o > 0; o >>= 1) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) part[co] += __shfl_xor(part[co], o, 64);
            }
            if (valid[u] && vv < COUT) {
                float v = 0.f;
#pragma unroll
                for (int co = 0; co < COUT; ++co) v = (vv == co) ? part[co] + bias[co] : v;
                if (act == 1) v = 1.f / (1.f + __expf(-v));
                const size_t b = p / hw, q = p - b * hw;
                out[(b * COUT + vv) * hw + q] = v;
            }
            // act16 (round 4): the same pixel once more as the 16-channel NHWC pixel the image-fed first layers read (channels 0..COUT-1,
            // zero tail) -- what wm_nchw_to_nhwc would make of `out` in a launch of its own; every lane of the pixel's group holds all sums
            if (act16 && valid[u] && vv < 16 / VE) {   // lane vv of the pixel's group writes the pixel's vv-th 16-byte piece: one store
                vec16<T> o;                            // instruction covers the wave's 8 adjacent pixels (256 contiguous bytes at 16 bits)
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    float val = 0.f;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) val = (vv * VE + e == co) ? part[co] + bias[co] : val;
                    o.set(e, val);
                }
                *reinterpret_cast<vec16<T>*>(act16 + p * 16 + vv * VE) = o;
            }
        }
    }
}

template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ gout, T* __restrict__ g, int ldg,
                                                       float* __restrict__ partials, size_t npix, size_t hw, int Cin,
                                                       float* __restrict__ bn_partials) {
    constexpr int VE = vec16<T>::N;
    const int VPP = Cin / VE, PPB = 256 / VPP;
    const int vv = threadIdx.x % VPP, ps = threadIdx.x / VPP;
    const int c0 = vv * VE;
    float sc[VE], sh[VE], wr[COUT][VE], dw[COUT][VE], db[COUT];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sc[e] = scale ? scale[c0 + e] : 1.f;
        sh[e] = scale ? shift[c0 + e] : 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co) { wr[co][e] = w[co * Cin + c0 + e]; dw[co][e] = 0.f; }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) db[co] = 0.f;
    // bn_partials: the BatchNorm-backward sums of the ConvBNRelu that produced y -- sum(gz), sum(gz*y), gz = g (as stored) * [z > 0] --
    // gathered while y and g are in registers anyway: the separate reduce pass over (g, y) disappears (rows for wm_bn_bwd_finalize_raw)
    float b1[VE], b2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { b1[e] = 0.f; b2[e] = 0.f; }
    for (size_t p = (size_t)blockIdx.x * PPB + ps; p < npix; p += (size_t)gridDim.x * PPB) {
        const size_t b = p / hw, q = p - b * hw;
        float go[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) go[co] = gout[(b * COUT + co) * hw + q];
        const vec16<T> yv = *reinterpret_cast<const vec16<T>*>(y + p * ldy + c0);
        vec16<T> gv;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float a = sc[e] * yv.get(e) + sh[e];
            if (scale) a = fmaxf(a, 0.f);
            float gg = 0.f;
#pragma unroll
            for (int co = 0; co < COUT; ++co) { gg += go[co] * wr[co][e]; dw[co][e] += go[co] * a; }
            gv.set(e, gg);
            if (bn_partials) {
                const float gz = a > 0.f ? gv.get(e) : 0.f;
                b1[e] += gz;
                b2[e] = __builtin_fmaf(gz, yv.get(e), b2[e]);
            }
        }
        if (vv == 0) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) db[co] += go[co];
        }
        *reinterpret_cast<vec16<T>*>(g + p * ldg + c0) = gv;
    }
    constexpr int RW = (COUT * vec16<T>::N + COUT + 1) > (2 * vec16<T>::N + 1) ? (COUT * vec16<T>::N + COUT + 1) : (2 * vec16<T>::N + 1);
    __shared__ float red[256][RW];   // (wide enough for the 2 x VE BatchNorm sums of the second reduction too)
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int e = 0; e < VE; ++e) red[threadIdx.x][co * VE + e] = dw[co][e];
        red[threadIdx.x][COUT * VE + co] = db[co];
    }
    __syncthreads();
    float* prow = partials + (size_t)blockIdx.x * COUT * (Cin + 1);
    for (int i = threadIdx.x; i < COUT * Cin; i += 256) {
        const int co = i / Cin, c = i - co * Cin;
        const int v2 = c / VE, e = c - v2 * VE;
        float s = 0.f;
        for (int qq = 0; qq < PPB; ++qq) s += red[qq * VPP + v2][co * VE + e];
        prow[i] = s;
    }
    if (threadIdx.x < COUT) {
        float s = 0.f;
        for (int qq = 0; qq < PPB; ++qq) s += red[qq * VPP][COUT * VE + threadIdx.x];
        prow[COUT * Cin + threadIdx.x] = s;
    }
    if (bn_partials) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VE; ++e) { red[threadIdx.x][e] = b1[e]; red[threadIdx.x][VE + e] = b2[e]; }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Cin; i += 256) {
            const int which = i / Cin, c = i - which * Cin;
            const int v2 = c / VE, e = c - v2 * VE;
            float s = 0.f;
            for (int qq = 0; qq < PPB; ++qq) s += red[qq * VPP + v2][which * VE + e];
            bn_partials[((size_t)blockIdx.x * 2 + which) * Cin + c] = s;
        }
    }
}

bool cin_ok(int Cin, int dtype) {
    const int ve = dtype != WM_F32 ? 8 : 4;
    if (Cin <= 0 || Cin % ve) return false;
    const int vpp = Cin / ve;
    return vpp <= 64 && (vpp & (vpp - 1)) == 0;
}

inline int head_parts(size_t npix) {
    const size_t n = (npix + 511) / 512;
    return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

}  // namespace

extern "C" int wm_conv1x1_head_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                                   const float* bias, float* out, int B, size_t hw, int Cin, int Cout, int act, int dtype,
                                   void* stream) {
    return wm_conv1x1_head_fwd_act(y, ldy, scale, shift, w, bias, out, nullptr, B, hw, Cin, Cout, act, dtype, stream);
}

extern "C" int wm_conv1x1_head_fwd_act(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                                       const float* bias, float* out, void* act16, int B, size_t hw, int Cin, int Cout, int act, int dtype,
                                       void* stream) {
    WM_REQUIRE(y && w && bias && out, WM_E_BADARG, "wm_conv1x1_head_fwd: null pointer");
    WM_REQUIRE(!act16 || (act == 0 && ((uintptr_t)act16 & 15) == 0), WM_E_BADARG, "wm_conv1x1_head_fwd_act: act16 goes with act == 0 and a 16-byte aligned pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_conv1x1_head_fwd: scale/shift must come together");
    WM_REQUIRE(cin_ok(Cin, dtype), WM_E_SHAPE, "wm_conv1x1_head_fwd: unsupported Cin=%d", Cin);
    WM_REQUIRE(Cout == 1 || Cout == 3, WM_E_SHAPE, "wm_conv1x1_head_fwd: Cout must be 1 or 3 (got %d)", Cout);
    const size_t npix = (size_t)B * hw;
    const int grid = (int)((npix + 127) / 128 > 2048 ? 2048 : (npix + 127) / 128);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_conv1x1_head_fwd",
        if (Cout == 3) hipLaunchKernelGGL((head_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, scale, shift, w, bias, out, npix, hw, Cin, act, (T*)act16);
        else hipLaunchKernelGGL((head_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, scale, shift, w, bias, out, npix, hw, Cin, act, (T*)act16));
    WM_LAUNCH_CHECK("wm_conv1x1_head_fwd");
    return WM_OK;
}

extern "C" int wm_conv1x1_head_nparts(size_t npix) { return head_parts(npix); }

extern "C" int wm_conv1x1_head_bwd(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                                   const float* gout, void* g, int ldg, float* partials, float* bn_partials, int B, size_t hw, int Cin,
                                   int Cout, int dtype, void* stream) {
    WM_REQUIRE(y && w && gout && g && partials, WM_E_BADARG, "wm_conv1x1_head_bwd: null pointer");
    WM_REQUIRE(!bn_partials || scale, WM_E_BADARG, "wm_conv1x1_head_bwd: bn_partials need the producing layer's scale / shift");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_conv1x1_head_bwd: scale/shift must come together");
    WM_REQUIRE(cin_ok(Cin, dtype), WM_E_SHAPE, "wm_conv1x1_head_bwd: unsupported Cin=%d", Cin);
    WM_REQUIRE(Cout == 1 || Cout == 3, WM_E_SHAPE, "wm_conv1x1_head_bwd: Cout must be 1 or 3 (got %d)", Cout);
    const size_t npix = (size_t)B * hw;
    const int grid = head_parts(npix);
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_conv1x1_head_bwd",
        if (Cout == 3) hipLaunchKernelGGL((head_bwd_kernel<T, 3>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, scale, shift, w, gout, (T*)g, ldg, partials, npix, hw, Cin, bn_partials);
        else hipLaunchKernelGGL((head_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, (const T*)y, ldy, scale, shift, w, gout, (T*)g, ldg, partials, npix, hw, Cin, bn_partials));
    WM_LAUNCH_CHECK("wm_conv1x1_head_bwd");
    return WM_OK;
}

// ------------------------------------------------------------------ nn.Linear after the global average pool
// (hidden_models/decoder.py:26,32-34, discriminator.py:18,24-26): [B,I] x [O,I]^T with B = 16, I,O <= 64 -- one small
// launch each way instead of a handful of library GEMM / elementwise launches.
namespace {
__global__ __launch_bounds__(256) void linear_head_fwd_kernel(const float* __restrict__ pooled, int ldp, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out, int B, int I,
                                                              int O) {
    for (int idx = threadIdx.x; idx < B * O; idx += 256) {
        const int b = idx / O, o = idx - b * O;
        float acc = bias ? bias[o] : 0.f;
        for (int i = 0; i < I; ++i) acc = fmaf(pooled[(size_t)b * ldp + i], w[(size_t)o * I + i], acc);
        out[idx] = acc;
    }
}
// dW[o,i] (+)= sum_b g[b,o] pooled[b,i];  db[o] (+)= sum_b g[b,o];  gvec[b, :CP] = (g[b,:] @ W) * inv_hw, zero padded
__global__ __launch_bounds__(256) void linear_head_bwd_kernel(const float* __restrict__ pooled, int ldp, const float* __restrict__ w,
                                                              const float* __restrict__ g, float* __restrict__ dw,
                                                              float* __restrict__ db, int accumulate, float* __restrict__ gvec,
                                                              int CP, float inv_hw, int B, int I, int O) {
    // grid-stride over the three independent outputs (one workgroup took 10-22 us on the serial loops)
    const int t0 = blockIdx.x * 256 + threadIdx.x, ts = gridDim.x * 256;
    for (int idx = t0; idx < O * I; idx += ts) {
        const int o = idx / I, i = idx - o * I;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(g[(size_t)b * O + o], pooled[(size_t)b * ldp + i], acc);
        dw[idx] = (accumulate ? dw[idx] : 0.f) + acc;
    }
    for (int o = t0; o < O; o += ts) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += g[(size_t)b * O + o];
        db[o] = (accumulate ? db[o] : 0.f) + acc;
    }
    for (int idx = t0; idx < B * CP; idx += ts) {
        const int b = idx / CP, i = idx - b * CP;
        float acc = 0.f;
        if (i < I)
            for (int o = 0; o < O; ++o) acc = fmaf(g[(size_t)b * O + o], w[(size_t)o * I + i], acc);
        gvec[idx] = acc * inv_hw;
    }
}

// ------------------------------------------------------------------ the whole pooled head in ONE launch (round 4)
// After the global average pool a discriminator / decoder pass ran four tiny launches in a row, each waiting for the one before
// (nn.Linear forward, the loss, nn.Linear backward, the pooled layer's BatchNorm-backward finalisation; hidden.py:68-101 over
// discriminator.py:24-26 / decoder.py:32-34): ~7 us of latency each on the step's critical chain for a few thousand flops.  One workgroup
// does them back to back through the LDS, each stage with the arithmetic of the kernel it replaces (same operation order: bit-identical
// results; tests/test_gpu_fused_glue.py):
//   logits = pooled @ w^T + bias                          (linear_head_fwd_kernel)
//   kind 0: BCE-with-logits against `target`, grad = (sigmoid - target) gscale / n          (bce_logits_kernel)
//   kind 1: message MSE + bitwise error against messages, grad = (logits - messages) gscale  (message_loss_kernel)
//   dw, db (+)=, gvec = (grad @ w) inv_hw                  (linear_head_bwd_kernel)
//   dgamma, dbeta (+)=, coef of the pooled ConvBNRelu from (gvec, N+, S+)                    (bn_bwd_finalize_kernel, pooled rows)
constexpr int PH_MAX_BC = 2048, PH_MAX_OI = 4096, PH_MAX_BO = 1024;
__global__ __launch_bounds__(256) void pooled_head_kernel(const float* __restrict__ out3, int B, int CP, int I, int O,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int kind, float target,
                                                          const float* __restrict__ messages, float gscale, const float* __restrict__ gscale_dev,
                                                          float* __restrict__ logits, float* __restrict__ loss_out, float* __restrict__ dw,
                                                          float* __restrict__ db, int accumulate, float* __restrict__ gvec, float inv_hw, int C,
                                                          double count, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ coef) {
    __shared__ float s_pool[PH_MAX_BC], s_w[PH_MAX_OI], s_logit[PH_MAX_BO], s_g[PH_MAX_BO], s_gv[PH_MAX_BC], s_np[PH_MAX_BC], s_ys[PH_MAX_BC], s_msg[PH_MAX_BO];
    __shared__ float s_red[2][4];
    const int tid = threadIdx.x;
    const float* pooled = out3;                       // plane 0 [B][CP]
    const float* npos = out3 + (size_t)B * CP;        // plane 1
    const float* ysum = out3 + (size_t)2 * B * CP;    // plane 2
    if (gscale_dev) gscale *= gscale_dev[0];
    // every global operand of every stage is fetched here, all loads in flight together (the stages then run out of the LDS: a stage that
    // loaded inside its loop paid one L2 round trip per trip -- 16 us for the whole kernel instead of 6)
    for (int i = tid; i < B * I; i += 256) s_pool[i] = pooled[(size_t)(i / I) * CP + (i % I)];
    for (int i = tid; i < O * I; i += 256) s_w[i] = w[i];
    for (int i = tid; i < B * CP; i += 256) { s_np[i] = npos[i]; s_ys[i] = ysum[i]; }
    if (kind == 1)
        for (int i = tid; i < B * O; i += 256) s_msg[i] = messages[i];
    __syncthreads();
    for (int idx = tid; idx < B * O; idx += 256) {
        const int b = idx / O, o = idx - b * O;
        float acc = bias ? bias[o] : 0.f;
        for (int i = 0; i < I; ++i) acc = fmaf(s_pool[b * I + i], s_w[o * I + i], acc);
        logits[idx] = acc;
        s_logit[idx] = acc;
    }
    __syncthreads();
    const int n = B * O;
    float a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float v = s_logit[i];
        if (kind == 0) {
            a1 += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
            s_g[i] = (1.f / (1.f + expf(-v)) - target) * gscale / (float)n;
        } else {
            const float df = v - s_msg[i];
            a1 += df * df;
            a2 += fabsf(fminf(fmaxf(rintf(v), 0.f), 1.f) - s_msg[i]);
            s_g[i] = df * gscale;
        }
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = a1; s_red[1][tid >> 6] = a2; }
    __syncthreads();
    if (tid == 0) {
        loss_out[0] = (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) / (float)n;
        if (kind == 1) loss_out[1] = (s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) / (float)n;
    }
    for (int idx = tid; idx < O * I; idx += 256) {
        const int o = idx / I, i = idx - o * I;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(s_g[b * O + o], s_pool[b * I + i], acc);
        dw[idx] = (accumulate ? dw[idx] : 0.f) + acc;
    }
    for (int o = tid; o < O; o += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += s_g[b * O + o];
        db[o] = (accumulate ? db[o] : 0.f) + acc;
    }
    for (int idx = tid; idx < B * CP; idx += 256) {
        const int b = idx / CP, i = idx - b * CP;
        float acc = 0.f;
        if (i < I)
            for (int o = 0; o < O; ++o) acc = fmaf(s_g[b * O + o], s_w[o * I + i], acc);
        const float gv = acc * inv_hw;
        gvec[idx] = gv;
        s_gv[idx] = gv;
    }
    __syncthreads();
    // the pooled layer's BatchNorm-backward finalisation: wm_bn_bwd_finalize_block's sums in ITS order (32 row slices, then the slices in order)
    for (int c = tid; c < CP; c += 256) {
        double t1 = 0.0, t2 = 0.0;
        for (int sl = 0; sl < 32; ++sl) {
            double p1 = 0.0, p2 = 0.0;
            for (int p = sl; p < B; p += 32) {
                const float gv = s_gv[p * CP + c];
                p1 += (double)(gv * s_np[p * CP + c]);
                p2 += (double)(gv * s_ys[p * CP + c]);
            }
            t1 = sl == 0 ? p1 : t1 + p1;
            t2 = sl == 0 ? p2 : t2 + p2;
        }
        if (c < C) {
            t2 = (t2 - (double)mean[c] * t1) * (double)invstd[c];
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)t1;
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)t2;
            coef[c] = gamma[c] * invstd[c];
            coef[CP + c] = (float)(t1 / count);
            coef[2 * CP + c] = (float)(t2 / count);
        } else {
            coef[c] = 0.f; coef[CP + c] = 0.f; coef[2 * CP + c] = 0.f;
        }
    }
}
}  // namespace

extern "C" int wm_pooled_head_supported(int B, int CP, int I, int O) {
    return (B > 0 && CP > 0 && I > 0 && O > 0 && I <= CP && B * CP <= PH_MAX_BC && O * I <= PH_MAX_OI && B * O <= PH_MAX_BO) ? 1 : 0;
}

extern "C" int wm_pooled_head(const float* out3, int B, int CP, int I, int O, const float* w, const float* bias, int kind, float target,
                              const float* messages, float gscale, const float* gscale_dev, float* logits, float* loss_out, float* dw, float* db,
                              int accumulate, float* gvec, float inv_hw, int C, double count, const float* gamma, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, float* coef, void* stream) {
    WM_REQUIRE(out3 && w && logits && loss_out && dw && db && gvec && gamma && mean && invstd && coef, WM_E_BADARG, "wm_pooled_head: null pointer");
    WM_REQUIRE(kind == 0 || (kind == 1 && messages), WM_E_BADARG, "wm_pooled_head: kind 0 (BCE against a constant label) or 1 (message loss, messages non-NULL)");
    WM_REQUIRE(wm_pooled_head_supported(B, CP, I, O) && C > 0 && C <= CP && count > 0, WM_E_SHAPE, "wm_pooled_head: B=%d CP=%d I=%d O=%d beyond the one-workgroup form", B, CP, I, O);
    hipLaunchKernelGGL(pooled_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out3, B, CP, I, O, w, bias, kind, target, messages, gscale,
                       gscale_dev, logits, loss_out, dw, db, accumulate, gvec, inv_hw, C, count, gamma, mean, invstd, dgamma, dbeta, coef);
    WM_LAUNCH_CHECK("wm_pooled_head");
    return WM_OK;
}

extern "C" int wm_linear_head_fwd(const float* pooled, int ldp, const float* w, const float* bias, float* out, int B, int I,
                                  int O, void* stream) {
    WM_REQUIRE(pooled && w && out && B > 0 && I > 0 && O > 0 && ldp >= I, WM_E_BADARG, "wm_linear_head_fwd: bad arguments");
    hipLaunchKernelGGL(linear_head_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pooled, ldp, w, bias, out, B, I, O);
    WM_LAUNCH_CHECK("wm_linear_head_fwd");
    return WM_OK;
}

extern "C" int wm_linear_head_bwd(const float* pooled, int ldp, const float* w, const float* g_out, float* dw, float* db,
                                  int accumulate, float* gvec, int CP, float inv_hw, int B, int I, int O, void* stream) {
    WM_REQUIRE(pooled && w && g_out && dw && db && gvec && B > 0 && I > 0 && O > 0 && ldp >= I && CP >= I, WM_E_BADARG,
               "wm_linear_head_bwd: bad arguments");
    const int work = O * I > B * CP ? O * I : B * CP;
    const int blocks = wm_cdiv(work, 256) > 16 ? 16 : wm_cdiv(work, 256);
    hipLaunchKernelGGL(linear_head_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pooled, ldp, w, g_out, dw, db,
                       accumulate, gvec, CP, inv_hw, B, I, O);
    WM_LAUNCH_CHECK("wm_linear_head_bwd");
    return WM_OK;
}
