// Loss / optimiser streaming kernels (f32, flat buffers):
//   nn.MSELoss forward+backward           hidden_models/hidden.py:37,90,97
//   torch.optim.Adam / AdamW step         hidden_models/hidden.py:24-25, models/IRNrhi_model.py:270-272
//   sum of squares (clip_grad_norm_)      models/IRNrhi_model.py:459-460
// All parameters of a network live in ONE flat f32 buffer (grads, exp_avg, exp_avg_sq likewise), so an
// optimiser step is one launch and the gradient all-reduce is one RCCL bucket.
#include "wm_common.h"

namespace {

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ grad_a, float gscale, float* __restrict__ partials,
                                                  size_t n, const float* __restrict__ gscale_dev) {
    if (gscale_dev) gscale *= gscale_dev[0];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        acc += d * d;
        if (grad_a) grad_a[i] = gscale * d;
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// The gradient wrt the encoded image at the point where hidden.py:85-101's three terms meet, in ONE pass (round 4; it was three launches:
// the NHWC -> NCHW conversion of the discriminator's input gradient, wm_mse_fwd_bwd and wm_axpy):
//   out[b][c][q] = float(g[b][q][c0 + c]) + gscale * (a - b),   partials[block] = sum (a - b)^2
// g: the NHWC input gradient the discriminator's first layer wrote (16-bit or f32, pixel stride ld); a = encoded, b = cover: f32 NCHW planes.
// The two roundings of the separate kernels are kept (product, then sum: no contraction into an fma), so the values are the same.
template <typename T>
__global__ __launch_bounds__(256) void image_grad_mse_kernel(const T* __restrict__ g, int ld, int c0, const float* __restrict__ a,
                                                            const float* __restrict__ b, float* __restrict__ out, float gscale,
                                                            const float* __restrict__ gscale_dev, float* __restrict__ partials, int B,
                                                            int C, size_t hw) {
    if (gscale_dev) gscale *= gscale_dev[0];
    float acc = 0.f;
    const size_t total = (size_t)B * hw;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t bi = p / hw, q = p - bi * hw;
        const T* gp = g + p * ld + c0;
        for (int c = 0; c < C; ++c) {
            const size_t i = (bi * C + c) * hw + q;
            const float d = a[i] - b[i];
            acc += d * d;
            float m = gscale * d;
            asm volatile("" : "+v"(m));          // (keeps the product's own rounding: hipcc would contract product + sum into one fma)
            out[i] = to_f32(gp[c]) + m;
        }
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// nn.BCEWithLogitsLoss (mean) of a small logit vector against a constant label, value and gradient in one launch
// (hidden_models/hidden.py:68-97: three of them per step on [B,1] tensors)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, float target, int n, float gscale,
                                                         float* __restrict__ loss_out, float* __restrict__ grad_out,
                                                         const float* __restrict__ gscale_dev) {
    if (gscale_dev) gscale *= gscale_dev[0];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        acc += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));   // the numerically stable form ATen uses
        if (grad_out) grad_out[i] = (1.f / (1.f + expf(-v)) - target) * gscale / (float)n;
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (s[0] + s[1] + s[2] + s[3]) / (float)n;
}

// decoder message loss (hidden.py:96-99,109-111): out[0] = mean (d-m)^2, out[1] = sum |clip(round(d),0,1) - m| / n,
// grad = (d-m) * gscale
__global__ __launch_bounds__(256) void message_loss_kernel(const float* __restrict__ d, const float* __restrict__ m, int n,
                                                           float gscale, float* __restrict__ out, float* __restrict__ grad,
                                                           const float* __restrict__ gscale_dev) {
    if (gscale_dev) gscale *= gscale_dev[0];
    float a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float df = d[i] - m[i];
        a1 += df * df;
        a2 += fabsf(fminf(fmaxf(rintf(d[i]), 0.f), 1.f) - m[i]);   // torch.round = round half to even
        if (grad) grad[i] = df * gscale;
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    __shared__ float s[2][4];
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = a1; s[1][threadIdx.x >> 6] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (s[0][0] + s[0][1] + s[0][2] + s[0][3]) / (float)n;
        out[1] = (s[1][0] + s[1][1] + s[1][2] + s[1][3]) / (float)n;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ partials, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i] * x[i];
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += s * b[i];
}

// torch.optim.Adam (single-tensor, non-amsgrad, non-capturable) arithmetic:
//   g += wd*p (coupled)  |  p *= 1 - lr*wd (decoupled)
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m/denom
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, int decoupled, float step_size, float bc2_sqrt,
                                                   float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gg = g[i] * grad_scale;
        float pp = p[i];
        if (wd != 0.f) {
            if (decoupled) pp *= 1.f - lr * wd;
            else gg += wd * pp;
        }
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pp - step_size * (mm / denom);
    }
}

// the same step with its two step-count-dependent constants read from DEVICE memory (hyper = {lr / (1 - b1^t), sqrt(1 - b2^t)}, the values
// wm_adam_hyper computes on the host): a step captured into a hipGraph replays with the constants of the CURRENT step count -- the host
// refreshes the two floats before each replay -- and is bit-identical to wm_adam_step
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                       float wd, int decoupled, const float* __restrict__ hyper, float grad_scale) {
    const float step_size = hyper[0], bc2_sqrt = hyper[1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gg = g[i] * grad_scale;
        float pp = p[i];
        if (wd != 0.f) {
            if (decoupled) pp *= 1.f - lr * wd;
            else gg += wd * pp;
        }
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pp - step_size * (mm / denom);
    }
}

// ---- torch.cuda.amp.GradScaler (models/IRNcrop_model.py:143,407-416) kept on the device.  state (f32[WM_AMP_STATE]):
//   [0] scale  [1] growth tracker  [2] growth_factor  [3] backoff_factor  [4] growth_interval
//   [8 + k] found_inf of optimiser k (k < 4)   [12 + k] step count of optimiser k (torch's `step`: not advanced by a skipped step)
// found_inf[k] = !isfinite(sum of squares of optimiser k's gradients) -- the rows wm_sumsq wrote (shared with clip_grad_norm_)
struct AmpGroups { const float* parts[4]; int n[4]; };
__global__ void amp_found_inf_kernel(AmpGroups g, int ngroups, float* __restrict__ state, int k) {
    __shared__ float s[256];
    float a = 0.f;
    for (int q = 0; q < ngroups; ++q)
        for (int i = threadIdx.x; i < g.n[q]; i += 256) a += g.parts[q][i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) state[8 + k] = isfinite(s[0]) ? 0.f : 1.f;
}

// GradScaler.update(): any found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and at growth_interval scale *= growth.
// Also advances the step count of every optimiser that did step, and clears the flags for the next iteration.
__global__ void amp_update_kernel(float* __restrict__ state, int nopt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool inf = false;
    for (int k = 0; k < nopt; ++k) {
        if (state[8 + k] != 0.f) inf = true;
        else state[12 + k] += 1.f;
        state[8 + k] = 0.f;
    }
    if (inf) { state[0] *= state[3]; state[1] = 0.f; }
    else {
        state[1] += 1.f;
        if (state[1] >= state[4]) { state[0] *= state[2]; state[1] = 0.f; }
    }
}

// Adam / AdamW under the scaler: the gradients in g are still multiplied by state[0]; skipped entirely when found_inf[k]
__global__ __launch_bounds__(256) void adam_amp_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                       int decoupled, float grad_scale, const float* __restrict__ state, int k) {
    if (state[8 + k] != 0.f) return;
    const float t = state[12 + k] + 1.f;
    const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    const float gs = grad_scale / state[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gg = g[i] * gs;
        float pp = p[i];
        if (wd != 0.f) {
            if (decoupled) pp *= 1.f - lr * wd;
            else gg += wd * pp;
        }
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pp - step_size * (mm / denom);
    }
}

inline int grid_for(size_t n, int cap = 2048) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int wm_mse_fwd_bwd(const float* a, const float* b, float* grad_a, float gscale, const float* gscale_dev, float* loss_partials,
                              int nparts, size_t n, void* stream) {
    WM_REQUIRE(a && b && n > 0, WM_E_BADARG, "wm_mse_fwd_bwd: bad arguments");
    WM_REQUIRE(nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_mse_fwd_bwd: nparts must be in 1..2048");
    hipLaunchKernelGGL(mse_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, a, b, grad_a, gscale, loss_partials, n, gscale_dev);
    WM_LAUNCH_CHECK("wm_mse_fwd_bwd");
    return WM_OK;
}

extern "C" int wm_image_grad_mse(const void* g, int ld, int c0, const float* a, const float* b, float* out, float gscale, const float* gscale_dev,
                                 float* loss_partials, int nparts, int B, int C, int H, int W, int dtype, void* stream) {
    WM_REQUIRE(g && a && b && out && loss_partials, WM_E_BADARG, "wm_image_grad_mse: null pointer");
    WM_REQUIRE(B > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && c0 >= 0 && ld >= c0 + C && nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_image_grad_mse: bad shape");
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_image_grad_mse",
        hipLaunchKernelGGL((image_grad_mse_kernel<T>), dim3(nparts), dim3(256), 0, s, (const T*)g, ld, c0, a, b, out, gscale, gscale_dev, loss_partials, B, C, hw));
    WM_LAUNCH_CHECK("wm_image_grad_mse");
    return WM_OK;
}

extern "C" int wm_bce_logits(const float* logits, float target, int n, float gscale, const float* gscale_dev, float* loss_out,
                             float* grad_out, void* stream) {
    WM_REQUIRE(logits && loss_out && n > 0, WM_E_BADARG, "wm_bce_logits: bad arguments");
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, n, gscale, loss_out, grad_out, gscale_dev);
    WM_LAUNCH_CHECK("wm_bce_logits");
    return WM_OK;
}

extern "C" int wm_message_loss(const float* decoded, const float* messages, int n, float gscale, const float* gscale_dev, float* out2,
                               float* grad_out, void* stream) {
    WM_REQUIRE(decoded && messages && out2 && n > 0, WM_E_BADARG, "wm_message_loss: bad arguments");
    hipLaunchKernelGGL(message_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, decoded, messages, n, gscale, out2, grad_out, gscale_dev);
    WM_LAUNCH_CHECK("wm_message_loss");
    return WM_OK;
}

// the seven logged scalars of a HiDDeN step (hidden.py:105-113) in one launch:
// out = [w_adv*adv + w_enc*enc + w_dec*dec, enc, dec, bit error, adv, d_cover, d_encoded], enc = sum(enc_partials) / n_img
__global__ __launch_bounds__(256) void hidden_metrics_kernel(const float* __restrict__ enc_partials, int nparts, double n_img,
                                                             const float* __restrict__ msg2, const float* __restrict__ adv,
                                                             const float* __restrict__ d_cover, const float* __restrict__ d_enc,
                                                             float w_adv, float w_enc, float w_dec, float* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += (double)enc_partials[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float enc = (float)(red[0] / n_img), dec = msg2[0], av = adv[0];
        out[0] = w_adv * av + w_enc * enc + w_dec * dec;
        out[1] = enc; out[2] = dec; out[3] = msg2[1]; out[4] = av; out[5] = d_cover[0]; out[6] = d_enc[0];
    }
}

extern "C" int wm_hidden_metrics(const float* enc_partials, int nparts, double n_img, const float* msg2, const float* adv,
                                 const float* d_cover, const float* d_enc, float w_adv, float w_enc, float w_dec, float* out7,
                                 void* stream) {
    WM_REQUIRE(enc_partials && msg2 && adv && d_cover && d_enc && out7 && nparts > 0 && n_img > 0, WM_E_BADARG, "wm_hidden_metrics: bad arguments");
    hipLaunchKernelGGL(hidden_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, enc_partials, nparts, n_img, msg2, adv, d_cover,
                       d_enc, w_adv, w_enc, w_dec, out7);
    WM_LAUNCH_CHECK("wm_hidden_metrics");
    return WM_OK;
}

extern "C" int wm_sumsq(const float* x, size_t n, float* partials, int nparts, void* stream) {
    WM_REQUIRE(x && partials && n > 0 && nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_sumsq: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, partials, n);
    WM_LAUNCH_CHECK("wm_sumsq");
    return WM_OK;
}

extern "C" int wm_axpy(float* a, const float* b, float s, size_t n, void* stream) {
    WM_REQUIRE(a && b && n > 0, WM_E_BADARG, "wm_axpy: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, s, n);
    WM_LAUNCH_CHECK("wm_axpy");
    return WM_OK;
}

extern "C" int wm_adam_hyper(float lr, float beta1, float beta2, int step, float* out2) {
    WM_REQUIRE(out2 && step >= 1, WM_E_BADARG, "wm_adam_hyper: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out2[0] = (float)((double)lr / bc1);
    out2[1] = (float)sqrt(bc2);
    return WM_OK;
}

extern "C" int wm_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int decoupled, int step, float grad_scale, void* stream) {
    WM_REQUIRE(p && g && m && v && n > 0 && step >= 1, WM_E_BADARG, "wm_adam_step: bad arguments");
    float hy[2];
    wm_adam_hyper(lr, beta1, beta2, step, hy);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, decoupled, hy[0], hy[1], grad_scale);
    WM_LAUNCH_CHECK("wm_adam_step");
    return WM_OK;
}

extern "C" int wm_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int decoupled, const float* hyper_dev, float grad_scale, void* stream) {
    WM_REQUIRE(p && g && m && v && hyper_dev && n > 0, WM_E_BADARG, "wm_adam_step_dev: bad arguments");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, decoupled, hyper_dev, grad_scale);
    WM_LAUNCH_CHECK("wm_adam_step_dev");
    return WM_OK;
}


extern "C" int wm_amp_found_inf(const float* const* sumsq_partials, const int* nparts, int ngroups, float* state, int k, void* stream) {
    WM_REQUIRE(sumsq_partials && nparts && state && ngroups >= 1 && ngroups <= 4 && k >= 0 && k < 4, WM_E_BADARG, "wm_amp_found_inf: bad arguments");
    AmpGroups g;
    for (int q = 0; q < 4; ++q) { g.parts[q] = q < ngroups ? sumsq_partials[q] : nullptr; g.n[q] = q < ngroups ? nparts[q] : 0; }
    for (int q = 0; q < ngroups; ++q) WM_REQUIRE(g.parts[q] && g.n[q] > 0, WM_E_BADARG, "wm_amp_found_inf: null group");
    hipLaunchKernelGGL(amp_found_inf_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g, ngroups, state, k);
    WM_LAUNCH_CHECK("wm_amp_found_inf");
    return WM_OK;
}

extern "C" int wm_amp_update(float* state, int noptimizers, void* stream) {
    WM_REQUIRE(state && noptimizers >= 1 && noptimizers <= 4, WM_E_BADARG, "wm_amp_update: bad arguments");
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, noptimizers);
    WM_LAUNCH_CHECK("wm_amp_update");
    return WM_OK;
}

extern "C" int wm_adam_step_amp(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int decoupled, float grad_scale, const float* amp_state, int k, void* stream) {
    WM_REQUIRE(p && g && m && v && amp_state && n > 0 && k >= 0 && k < 4, WM_E_BADARG, "wm_adam_step_amp: bad arguments");
    hipLaunchKernelGGL(adam_amp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       decoupled, grad_scale, amp_state, k);
    WM_LAUNCH_CHECK("wm_adam_step_amp");
    return WM_OK;
}
