// Elementwise kernels of the tamper-localisation branch (models/IRNcrop_model.py:337-416 of the reference), f32 NCHW planes:
//   clamp_with_grad + Quantization                 IRNcrop_model.py:320-322,344-345,372-373 ; models/modules/Quantization.py:7-14
//   tamper splice fwd*(1-mask) + prev*mask         IRNcrop_model.py:348
//   PSNR(postprocess(input), postprocess(forward)) IRNcrop_model.py:379-380,660-664 ; metrics.py:30-46
//   PSNR-gated forward weight 1.0 / 0.8 at 33 dB   IRNcrop_model.py:382-388
//   BCEWithLogitsLoss(predicted mask, gt mask)     IRNcrop_model.py:378,391-393 (applied to the UNet's sigmoid outputs)
//   clip_grad_norm_ over a group of flat buffers   IRNcrop_model.py:410-412
// All HBM-bound streaming passes; the reductions are two-stage and deterministic (per-workgroup partials in a fixed order).
#include "wm_common.h"

namespace {

__device__ __forceinline__ float clamp_quant(float v) {
    // torch.clamp(x,0,1) then (x*255).round()/255 -- rintf = round half to even like torch.round, true f32 division
    const float c = fminf(fmaxf(v, 0.f), 1.f);
    return rintf(c * 255.f) / 255.f;
}

inline int grid_for(size_t n, int cap = 2048) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

__global__ __launch_bounds__(256) void clamp_quant_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = clamp_quant(x[i]);
}

// tampered = Q(clamp(enc))*(1-mask) + prev*mask ; partial sums of (int(real*255) - int(Q*255))^2 for the PSNR
// enc, real, prev, out: [B,C,HW]; mask: [B,1,HW].  One workgroup walks whole planes slices so the mask index is a plain offset.
__global__ __launch_bounds__(256) void splice_kernel(const float* __restrict__ enc, const float* __restrict__ real,
                                                     const float* __restrict__ prev, const float* __restrict__ mask,
                                                     float* __restrict__ fwd_q, float* __restrict__ out, double* __restrict__ partials,
                                                     int C, size_t HW, size_t n) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = i / HW, px = i - plane * HW;
        const size_t b = plane / (size_t)C;
        const float q = clamp_quant(enc[i]);
        if (fwd_q) fwd_q[i] = q;
        if (out) {
            const float m = mask[b * HW + px];
            out[i] = q * (1.f - m) + prev[i] * m;
        }
        if (real) {
            const int a = (int)(real[i] * 255.0f), f = (int)(q * 255.0f);   // postprocess(): (img * 255.0).int() truncates
            const float d = (float)a - (float)f;
            acc += (double)(d * d);
        }
    }
    __shared__ double s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = s[0];
}

// metrics.py:30-46 with max_val 255: psnr = 20*log(255)/log(10) - 10*log(mse)/log(10); mse == 0 -> 0.
// out[0] = psnr, out[1] = psnr < threshold ? w_below : w_above   (IRNcrop_model.py:382-388)
__global__ void psnr_gate_kernel(const double* __restrict__ partials, int nparts, double n, float threshold, float w_below,
                                 float w_above, float* __restrict__ out) {
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += partials[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(s[0] / n);
        const float base10 = logf(10.0f);
        const float maxv = 20.f * logf(255.0f) / base10;
        const float psnr = mse == 0.f ? 0.f : maxv - 10.f * logf(mse) / base10;
        out[0] = psnr;
        out[1] = psnr < threshold ? w_below : w_above;
    }
}

// nn.BCEWithLogitsLoss (mean) of a tensor against a tensor target: partial sums of the per-element loss (already / n) and
// grad = gscale * (sigmoid(p) - t) / n
// chain != 0: p is itself a sigmoid output s(z) (the UNet head, UNet.py:65) and grad is taken wrt z: multiplied by p*(1-p)
__global__ __launch_bounds__(256) void bce_target_kernel(const float* __restrict__ p, const float* __restrict__ t, size_t n, float gscale,
                                                         float* __restrict__ partials, float* __restrict__ grad, int chain,
                                                         const float* __restrict__ gscale_dev) {
    if (gscale_dev) gscale *= gscale_dev[0];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = p[i], tt = t[i];
        acc += fmaxf(v, 0.f) - v * tt + log1pf(expf(-fabsf(v)));
        if (grad) {
            const float gv = (1.f / (1.f + expf(-v)) - tt) * gscale * inv;
            grad[i] = chain ? gv * v * (1.f - v) : gv;
        }
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1] + s[2] + s[3]) * inv;
}

__global__ void sum_partials_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += (double)partials[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)s[0];
}

// a += g * (1 - mask): the splice's backward onto the encoder gradient (STE clamp and Quantization pass g unchanged)
__global__ __launch_bounds__(256) void masked_axpy_kernel(float* __restrict__ a, const float* __restrict__ g, const float* __restrict__ mask,
                                                          int C, size_t HW, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = i / HW, px = i - plane * HW;
        a[i] += g[i] * (1.f - mask[(plane / (size_t)C) * HW + px]);
    }
}

__global__ __launch_bounds__(256) void mask_threshold_kernel(const float* __restrict__ p, float thr, uint8_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = p[i] > thr ? 1 : 0;
}

// clip_grad_norm_(params, max_norm): total = sqrt(sum of all squares); coef = min(1, max_norm / (total + 1e-6))
struct ClipGroups { const float* parts[4]; int n[4]; };
__global__ void clip_coef_kernel(ClipGroups g, int ngroups, float max_norm, float* __restrict__ out) {
    __shared__ double s[256];
    double a = 0.0;
    for (int k = 0; k < ngroups; ++k)
        for (int i = threadIdx.x; i < g.n[k]; i += 256) a += (double)g.parts[k][i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = sqrtf((float)s[0]);
        const float coef = max_norm / (total + 1e-6f);
        out[0] = coef < 1.f ? coef : 1.f;
        out[1] = total;
    }
}

__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, size_t n, const float* __restrict__ s) {
    const float f = s[0];
    if (f == 1.f) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= f;
}

// grad = (gscale * gate[0]) * (a - b) with the gate a device scalar; partial sums of (a-b)^2 as wm_mse_fwd_bwd
__global__ __launch_bounds__(256) void mse_gated_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ grad_a,
                                                        float gscale, const float* __restrict__ gate, float* __restrict__ partials, size_t n,
                                                        const float* __restrict__ gscale_dev) {
    const float gs = gscale * gate[0] * (gscale_dev ? gscale_dev[0] : 1.f);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        acc += d * d;
        grad_a[i] = gs * d;
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

}  // namespace

extern "C" int wm_clamp_quant_fwd(const float* x, float* y, size_t n, void* stream) {
    WM_REQUIRE(x && y && n > 0, WM_E_BADARG, "wm_clamp_quant_fwd: bad arguments");
    hipLaunchKernelGGL(clamp_quant_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    WM_LAUNCH_CHECK("wm_clamp_quant_fwd");
    return WM_OK;
}

extern "C" int wm_splice_nparts(size_t n) { return grid_for(n, 1024); }

extern "C" int wm_splice_fwd(const float* enc, const float* real, const float* prev, const float* mask, float* fwd_q, float* tampered,
                             double* psnr_partials, int B, int C, size_t HW, void* stream) {
    WM_REQUIRE(enc && B > 0 && C > 0 && HW > 0, WM_E_BADARG, "wm_splice_fwd: bad arguments");
    WM_REQUIRE((tampered == nullptr) || (prev && mask), WM_E_BADARG, "wm_splice_fwd: the splice needs prev and mask");
    WM_REQUIRE((real == nullptr) == (psnr_partials == nullptr), WM_E_BADARG, "wm_splice_fwd: real and psnr_partials go together");
    const size_t n = (size_t)B * C * HW;
    hipLaunchKernelGGL(splice_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, enc, real, prev, mask, fwd_q, tampered,
                       psnr_partials, C, HW, n);
    WM_LAUNCH_CHECK("wm_splice_fwd");
    return WM_OK;
}

extern "C" int wm_psnr_gate(const double* psnr_partials, int nparts, double n, float threshold, float w_below, float w_above, float* out2,
                            void* stream) {
    WM_REQUIRE(psnr_partials && out2 && nparts > 0 && n > 0, WM_E_BADARG, "wm_psnr_gate: bad arguments");
    hipLaunchKernelGGL(psnr_gate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, psnr_partials, nparts, n, threshold, w_below, w_above, out2);
    WM_LAUNCH_CHECK("wm_psnr_gate");
    return WM_OK;
}

extern "C" int wm_bce_logits_target(const float* p, const float* target, size_t n, float gscale, const float* gscale_dev, float* partials,
                                    int nparts, float* loss_out, float* grad_out, int chain_sigmoid, void* stream) {
    WM_REQUIRE(p && target && partials && loss_out && n > 0 && nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_bce_logits_target: bad arguments");
    hipLaunchKernelGGL(bce_target_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, p, target, n, gscale, partials, grad_out, chain_sigmoid, gscale_dev);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nparts, loss_out);
    WM_LAUNCH_CHECK("wm_bce_logits_target");
    return WM_OK;
}

extern "C" int wm_masked_axpy(float* a, const float* g, const float* mask, int B, int C, size_t HW, void* stream) {
    WM_REQUIRE(a && g && mask && B > 0 && C > 0 && HW > 0, WM_E_BADARG, "wm_masked_axpy: bad arguments");
    const size_t n = (size_t)B * C * HW;
    hipLaunchKernelGGL(masked_axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, g, mask, C, HW, n);
    WM_LAUNCH_CHECK("wm_masked_axpy");
    return WM_OK;
}

extern "C" int wm_mask_threshold(const float* p, float threshold, uint8_t* out, size_t n, void* stream) {
    WM_REQUIRE(p && out && n > 0, WM_E_BADARG, "wm_mask_threshold: bad arguments");
    hipLaunchKernelGGL(mask_threshold_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, threshold, out, n);
    WM_LAUNCH_CHECK("wm_mask_threshold");
    return WM_OK;
}

extern "C" int wm_clip_coef(const float* const* partials, const int* nparts, int ngroups, float max_norm, float* out2, void* stream) {
    WM_REQUIRE(partials && nparts && out2 && ngroups >= 1 && ngroups <= 4 && max_norm > 0.f, WM_E_BADARG, "wm_clip_coef: bad arguments (1..4 groups)");
    ClipGroups g;
    for (int k = 0; k < 4; ++k) { g.parts[k] = k < ngroups ? partials[k] : nullptr; g.n[k] = k < ngroups ? nparts[k] : 0; }
    for (int k = 0; k < ngroups; ++k) WM_REQUIRE(g.parts[k] && g.n[k] > 0, WM_E_BADARG, "wm_clip_coef: null group");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g, ngroups, max_norm, out2);
    WM_LAUNCH_CHECK("wm_clip_coef");
    return WM_OK;
}

extern "C" int wm_scale_dev(float* x, size_t n, const float* scale_dev, void* stream) {
    WM_REQUIRE(x && scale_dev && n > 0, WM_E_BADARG, "wm_scale_dev: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, scale_dev);
    WM_LAUNCH_CHECK("wm_scale_dev");
    return WM_OK;
}

extern "C" int wm_mse_fwd_bwd_gated(const float* a, const float* b, float* grad_a, float gscale, const float* gate_dev, const float* gscale_dev,
                                    float* loss_partials, int nparts, size_t n, void* stream) {
    WM_REQUIRE(a && b && grad_a && gate_dev && n > 0, WM_E_BADARG, "wm_mse_fwd_bwd_gated: bad arguments");
    WM_REQUIRE(nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_mse_fwd_bwd_gated: nparts must be in 1..2048");
    hipLaunchKernelGGL(mse_gated_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, a, b, grad_a, gscale, gate_dev, loss_partials, n, gscale_dev);
    WM_LAUNCH_CHECK("wm_mse_fwd_bwd_gated");
    return WM_OK;
}
