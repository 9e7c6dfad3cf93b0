// Losses and the value clamp of the literal IRNrhi step (models/IRNrhi_model.py:425-560 of the reference), each one streaming pass
// that yields the mean loss (device scalar) AND the gradient for an upstream gradient of 1:
//   nn.SmoothL1Loss()      :148,476,481   wm_smooth_l1
//   nn.BCELoss()           :147,492-493,505   wm_bce_prob   (log clamped at -100 and the gradient's denominator at 1e-12, as torch does)
//   nn.CrossEntropyLoss()  :156,454,485   wm_cross_entropy  (int64 class labels)
//   torch.clamp(x, 0, 1)   :430,472       wm_clamp01_fwd / _bwd  (gradient passes where 0 <= x <= 1)
//   PSNR of postprocess()ed images  :527, metrics.py:30-46   wm_psnr255_partials (+ wm_psnr_gate of localise.hip for the final value)
// f32, any layout (flat element-wise) except the logits [B][ld].
#include "wm_common.h"

namespace {

inline int grid_for(size_t n, int cap = 2048) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

__device__ __forceinline__ void block_partial(float acc, float* partials) {
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ void sum_partials_scaled_kernel(const float* __restrict__ partials, int nparts, double scale, float* __restrict__ out) {
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += (double)partials[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(s[0] * scale);
}

__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float beta, float* __restrict__ partials,
                                                        float* __restrict__ grad) {
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i], ad = fabsf(d);
        acc += ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
        if (grad) grad[i] = (ad < beta ? d / beta : (d > 0.f ? 1.f : -1.f)) * inv;
    }
    block_partial(acc, partials);
}

__global__ __launch_bounds__(256) void bce_prob_kernel(const float* __restrict__ p, float target, size_t n, float* __restrict__ partials, float* __restrict__ grad) {
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = p[i];
        acc -= target * fmaxf(logf(v), -100.f) + (1.f - target) * fmaxf(logf(1.f - v), -100.f);
        if (grad) grad[i] = (v - target) / fmaxf((1.f - v) * v, 1e-12f) * inv;
    }
    block_partial(acc, partials);
}

// one workgroup; thread per sample row
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int B, int K, int ld,
                                                            float* __restrict__ loss, float* __restrict__ grad) {
    __shared__ double s[256];
    double acc = 0.0;
    const float inv = 1.f / (float)B;
    for (int r = threadIdx.x; r < B; r += 256) {
        const float* z = logits + (size_t)r * ld;
        float m = z[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(z[k] - m);
        const float lse = m + logf(se);
        const int y = (int)labels[r];
        acc += (double)(lse - z[y]);
        if (grad)
            for (int k = 0; k < K; ++k) grad[(size_t)r * ld + k] = (expf(z[k] - lse) - (k == y ? 1.f : 0.f)) * inv;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(s[0] / (double)B);
}

__global__ __launch_bounds__(256) void clamp01_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}
__global__ __launch_bounds__(256) void clamp01_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        gx[i] = (v >= 0.f && v <= 1.f) ? g[i] : 0.f;
    }
}

// partial sums of (int(255 a) - int(255 b))^2 with a, b clamped to [0,1] first: postprocess() (IRNrhi_model.py:867-871; its inputs are already clamped) then metrics.py:30-46
__global__ __launch_bounds__(256) void psnr255_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* __restrict__ partials) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ia = (int)(fminf(fmaxf(a[i] * 255.f, 0.f), 255.f)), ib = (int)(fminf(fmaxf(b[i] * 255.f, 0.f), 255.f));
        const double d = (double)(ia - ib);
        acc += d * d;
    }
    __shared__ double s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = s[0];
}

}  // namespace

// *loss_out = mean SmoothL1(a - b; beta); grad_out (may be NULL) = d loss / d a; partials: f32 scratch [nparts <= 2048]
extern "C" int wm_smooth_l1(const float* a, const float* b, size_t n, float beta, float* partials, int nparts, float* loss_out, float* grad_out, void* stream) {
    WM_REQUIRE(a && b && partials && loss_out && n > 0 && nparts > 0 && nparts <= 2048 && beta > 0.f, WM_E_BADARG, "wm_smooth_l1: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(nparts), dim3(256), 0, s, a, b, n, beta, partials, grad_out);
    hipLaunchKernelGGL(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, s, partials, nparts, 1.0 / (double)n, loss_out);
    WM_LAUNCH_CHECK("wm_smooth_l1");
    return WM_OK;
}
// nn.BCELoss()(p, full_like(p, target)): *loss_out = mean; grad_out (may be NULL) = d loss / d p
extern "C" int wm_bce_prob(const float* p, float target, size_t n, float* partials, int nparts, float* loss_out, float* grad_out, void* stream) {
    WM_REQUIRE(p && partials && loss_out && n > 0 && nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_bce_prob: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_prob_kernel, dim3(nparts), dim3(256), 0, s, p, target, n, partials, grad_out);
    hipLaunchKernelGGL(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, s, partials, nparts, 1.0 / (double)n, loss_out);
    WM_LAUNCH_CHECK("wm_bce_prob");
    return WM_OK;
}
// nn.CrossEntropyLoss()(logits [B][ld] (K classes), labels int64 [B]); grad_out [B][ld] (may be NULL; columns >= K untouched)
extern "C" int wm_cross_entropy(const float* logits, const long long* labels, int B, int K, int ld, float* loss_out, float* grad_out, void* stream) {
    WM_REQUIRE(logits && labels && loss_out && B > 0 && K > 0 && ld >= K, WM_E_BADARG, "wm_cross_entropy: bad arguments");
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, B, K, ld, loss_out, grad_out);
    WM_LAUNCH_CHECK("wm_cross_entropy");
    return WM_OK;
}
extern "C" int wm_clamp01_fwd(const float* x, float* y, size_t n, void* stream) {
    WM_REQUIRE(x && y && n > 0, WM_E_BADARG, "wm_clamp01_fwd: bad arguments");
    hipLaunchKernelGGL(clamp01_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    WM_LAUNCH_CHECK("wm_clamp01_fwd");
    return WM_OK;
}
extern "C" int wm_clamp01_bwd(const float* x, const float* g, float* gx, size_t n, void* stream) {
    WM_REQUIRE(x && g && gx && n > 0, WM_E_BADARG, "wm_clamp01_bwd: bad arguments");
    hipLaunchKernelGGL(clamp01_bwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, g, gx, n);
    WM_LAUNCH_CHECK("wm_clamp01_bwd");
    return WM_OK;
}
// partials [nparts <= 2048] doubles; feed them to wm_psnr_gate(partials, nparts, n, ...) for 20 log10(255 / sqrt(mse))
extern "C" int wm_psnr255_partials(const float* a, const float* b, size_t n, double* partials, int nparts, void* stream) {
    WM_REQUIRE(a && b && partials && n > 0 && nparts > 0 && nparts <= 2048, WM_E_BADARG, "wm_psnr255_partials: bad arguments");
    hipLaunchKernelGGL(psnr255_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, a, b, n, partials);
    WM_LAUNCH_CHECK("wm_psnr255_partials");
    return WM_OK;
}
