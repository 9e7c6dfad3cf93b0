// Layout conversions between the reference's NCHW f32 image planes (attack / loss boundary) and the
// NHWC bf16|f32 activation layout of the conv path; message broadcast (hidden_models/encoder.py:34-37).
// Pure data movement, HBM-bound: a thread owns one pixel, plane reads are coalesced across the wave,
// NHWC writes are 16-byte vectors when the destination slice is aligned.
#include "wm_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C,
                                                           size_t hw, int ld, int c0, int zero_tail) {
    const size_t total = (size_t)B * hw;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t b = p / hw, q = p - b * hw;
        T* o = y + p * ld + c0;
        for (int c = 0; c < C; ++c) o[c] = from_f32<T>(x[(b * C + c) * hw + q]);
        for (int c = C; c < C + zero_tail; ++c) o[c] = from_f32<T>(0.f);
    }
}

// specialisation used for image inputs: C = 3 -> 16-channel zero padded pixel (32 B bf16 / 64 B f32), c0 = 0
template <typename T>
__global__ __launch_bounds__(256) void nchw3_to_nhwc16_kernel(const float* __restrict__ x, T* __restrict__ y, int B,
                                                              size_t hw, int ld) {
    constexpr int VE = vec16<T>::N;
    const size_t total = (size_t)B * hw;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t b = p / hw, q = p - b * hw;
        vec16<T> v;
#pragma unroll
        for (int e = 0; e < VE; ++e) v.set(e, 0.f);
        vec16<T> z = v;
        v.set(0, x[(b * 3 + 0) * hw + q]);
        v.set(1, x[(b * 3 + 1) * hw + q]);
        v.set(2, x[(b * 3 + 2) * hw + q]);
        vec16<T>* o = reinterpret_cast<vec16<T>*>(y + p * ld);
        o[0] = v;
#pragma unroll
        for (int k = 1; k < 16 / VE; ++k) o[k] = z;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C,
                                                           size_t hw, int ld, int c0) {
    const size_t total = (size_t)B * hw;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t b = p / hw, q = p - b * hw;
        const T* i = x + p * ld + c0;
        for (int c = 0; c < C; ++c) y[(b * C + c) * hw + q] = to_f32(i[c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void broadcast_kernel(const float* __restrict__ v, T* __restrict__ y, int B, int L,
                                                        size_t hw, int ld, int c0) {
    const size_t total = (size_t)B * hw;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t b = p / hw;
        T* o = y + p * ld + c0;
        for (int c = 0; c < L; ++c) o[c] = from_f32<T>(v[b * L + c]);
    }
}

// encoder concat tail (hidden_models/encoder.py:34-40): [message(L) | image(3) | zero pad] written as whole
// 16-byte vectors at channel offset c0 (c0 and the tail width are multiples of the vector width)
template <typename T>
__global__ __launch_bounds__(256) void concat_tail_kernel(const float* __restrict__ msg, const float* __restrict__ img,
                                                          T* __restrict__ y, int B, int L, size_t hw, int ld, int c0,
                                                          int tail) {
    constexpr int VE = vec16<T>::N;
    const int nv = tail / VE;
    const size_t total = (size_t)B * hw * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % nv);
        const size_t p = i / nv;
        const size_t b = p / hw, q = p - b * hw;
        vec16<T> o;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int c = v * VE + e;
            float f = 0.f;
            if (c < L) f = msg[b * L + c];
            else if (c < L + 3) f = img[(b * 3 + (c - L)) * hw + q];
            o.set(e, f);
        }
        *reinterpret_cast<vec16<T>*>(y + p * ld + c0 + v * VE) = o;
    }
}

// whole encoder concat row in one pass (hidden_models/encoder.py:34-40): [relu(scale*x+shift) (C feature channels) | message(L) |
// image(3) | zero pad], every thread one 16-byte vector, consecutive threads consecutive vectors of the pixel: the ld-channel
// rows are written as full contiguous lines (the two-kernel form wrote them as 128 + 96 byte pieces)
template <typename T>
__global__ __launch_bounds__(256) void concat_full_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ msg,
                                                          const float* __restrict__ img, T* __restrict__ y, int B, int C,
                                                          int L, size_t hw, int ld) {
    // 64 pixels per iteration through an LDS image of their output rows: feature vectors are read, and the finished rows
    // written, as whole contiguous runs; tail vectors are built slot by slot (a wave works on ONE slot: no divergence,
    // coalesced image-plane reads)
    constexpr int VE = vec16<T>::N, PX = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* rows = reinterpret_cast<T*>(smem_raw);                           // [PX][ld]
    float* sSc = reinterpret_cast<float*>(smem_raw + (size_t)PX * ld * sizeof(T));   // [C] scale, [C] shift
    float* sSh = sSc + C;
    for (int c = threadIdx.x; c < C; c += 256) { sSc[c] = scale[c]; sSh[c] = shift[c]; }
    __syncthreads();
    const int nv = ld / VE, nvc = C / VE, nvt = nv - nvc;
    const size_t npix = (size_t)B * hw;
    for (size_t p0 = (size_t)blockIdx.x * PX; p0 < npix; p0 += (size_t)gridDim.x * PX) {
        const int npx = (int)((npix - p0) < (size_t)PX ? (npix - p0) : (size_t)PX);
        for (int idx = threadIdx.x; idx < npx * nvc; idx += 256) {
            const int px = idx / nvc, v = idx - px * nvc;
            vec16<T> o = *reinterpret_cast<const vec16<T>*>(x + (p0 + px) * ldx + v * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) o.set(e, fmaxf(sSc[v * VE + e] * o.get(e) + sSh[v * VE + e], 0.f));
            *reinterpret_cast<vec16<T>*>(rows + px * ld + v * VE) = o;
        }
        for (int idx = threadIdx.x; idx < PX * nvt; idx += 256) {
            const int vt = idx / PX, px = idx - vt * PX;
            if (px < npx) {
                const size_t p = p0 + px, b = p / hw, q = p - b * hw;
                vec16<T> o;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int c = vt * VE + e;
                    float f = 0.f;
                    if (c < L) f = msg[b * L + c];
                    else if (c < L + 3) f = img[(b * 3 + (c - L)) * hw + q];
                    o.set(e, f);
                }
                *reinterpret_cast<vec16<T>*>(rows + px * ld + C + vt * VE) = o;
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < npx * nv; idx += 256)
            *reinterpret_cast<vec16<T>*>(y + p0 * ld + (size_t)idx * VE) = *reinterpret_cast<const vec16<T>*>(rows + (size_t)idx * VE);
        __syncthreads();
    }
}

inline int grid_for(size_t n) {
    const size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// decoded frames, uint8 [N][H][W][C] (what an image decoder hands out) -> float planes [N][C][H][W] * scale: the first step of the clip
// loader's device path (data/Dataloader.py: scale = 1/255); a thread converts one pixel (C <= 4 bytes read as one unit when C == 4)
__global__ void u8_hwc_to_planes_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, size_t npix, size_t hw, int C, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, p = i - n * hw;
        for (int c = 0; c < C; ++c) y[(n * C + c) * hw + p] = (float)x[i * C + c] * scale;
    }
}

}  // namespace

extern "C" int wm_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int ld, int c0, int zero_tail,
                               int dtype, void* stream) {
    WM_REQUIRE(x && y, WM_E_BADARG, "wm_nchw_to_nhwc: null pointer");
    WM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && c0 >= 0 && zero_tail >= 0 && ld >= c0 + C + zero_tail, WM_E_BADARG,
               "wm_nchw_to_nhwc: bad shape (ld=%d c0=%d C=%d tail=%d)", ld, c0, C, zero_tail);
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    const int esz = dtype != WM_F32 ? 2 : 4;
    const bool fast = C == 3 && c0 == 0 && zero_tail == 13 && (ld * esz) % 16 == 0 && ((uintptr_t)y & 15) == 0;
    WM_DISPATCH_DTYPE(dtype, "wm_nchw_to_nhwc",
        if (fast) hipLaunchKernelGGL((nchw3_to_nhwc16_kernel<T>), dim3(grid_for(B * hw)), dim3(256), 0, s, x, (T*)y, B, hw, ld);
        else hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(B * hw)), dim3(256), 0, s, x, (T*)y, B, C, hw, ld, c0, zero_tail));
    WM_LAUNCH_CHECK("wm_nchw_to_nhwc");
    return WM_OK;
}

extern "C" int wm_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int ld, int c0, int dtype,
                               void* stream) {
    WM_REQUIRE(x && y, WM_E_BADARG, "wm_nhwc_to_nchw: null pointer");
    WM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && c0 >= 0 && ld >= c0 + C, WM_E_BADARG, "wm_nhwc_to_nchw: bad shape");
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_nhwc_to_nchw",
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(B * hw)), dim3(256), 0, s, (const T*)x, y, B, C, hw, ld, c0));
    WM_LAUNCH_CHECK("wm_nhwc_to_nchw");
    return WM_OK;
}

extern "C" int wm_broadcast_to_nhwc(const float* v, void* y, int B, int L, int H, int W, int ld, int c0, int dtype,
                                    void* stream) {
    WM_REQUIRE(v && y, WM_E_BADARG, "wm_broadcast_to_nhwc: null pointer");
    WM_REQUIRE(B > 0 && L > 0 && H > 0 && W > 0 && c0 >= 0 && ld >= c0 + L, WM_E_BADARG, "wm_broadcast_to_nhwc: bad shape");
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_broadcast_to_nhwc",
        hipLaunchKernelGGL((broadcast_kernel<T>), dim3(grid_for(B * hw)), dim3(256), 0, s, v, (T*)y, B, L, hw, ld, c0));
    WM_LAUNCH_CHECK("wm_broadcast_to_nhwc");
    return WM_OK;
}

extern "C" int wm_concat_tail(const float* msg, const float* img, void* y, int B, int L, int H, int W, int ld, int c0,
                              int tail, int dtype, void* stream) {
    WM_REQUIRE(msg && img && y, WM_E_BADARG, "wm_concat_tail: null pointer");
    const int ve = dtype != WM_F32 ? 8 : 4;
    WM_REQUIRE(B > 0 && L > 0 && H > 0 && W > 0 && tail >= L + 3 && ld >= c0 + tail, WM_E_BADARG, "wm_concat_tail: bad shape");
    WM_REQUIRE(c0 % ve == 0 && tail % ve == 0 && ld % ve == 0, WM_E_SHAPE, "wm_concat_tail: c0=%d tail=%d ld=%d must be multiples of %d", c0, tail, ld, ve);
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_concat_tail",
        hipLaunchKernelGGL((concat_tail_kernel<T>), dim3(grid_for(B * hw * (tail / ve))), dim3(256), 0, s, msg, img, (T*)y, B, L, hw, ld, c0, tail));
    WM_LAUNCH_CHECK("wm_concat_tail");
    return WM_OK;
}

extern "C" int wm_concat_full(const void* x, int ldx, const float* scale, const float* shift, const float* msg, const float* img,
                              void* y, int B, int C, int L, int H, int W, int ld, int dtype, void* stream) {
    WM_REQUIRE(x && scale && shift && msg && img && y, WM_E_BADARG, "wm_concat_full: null pointer");
    const int ve = dtype != WM_F32 ? 8 : 4;
    WM_REQUIRE(B > 0 && C > 0 && L > 0 && H > 0 && W > 0 && ld >= C + L + 3 && ldx >= C, WM_E_BADARG, "wm_concat_full: bad shape");
    WM_REQUIRE(C % ve == 0 && ld % ve == 0 && ldx % ve == 0, WM_E_SHAPE, "wm_concat_full: C=%d ld=%d ldx=%d must be multiples of %d", C, ld, ldx, ve);
    const size_t hw = (size_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)64 * ld * (dtype != WM_F32 ? 2 : 4) + (size_t)2 * C * 4;
    WM_REQUIRE(lds <= 64 * 1024, WM_E_SHAPE, "wm_concat_full: row of %d channels is too wide", ld);
    const size_t nb = ((size_t)B * hw + 63) / 64;
    const int grid = (int)(nb > 2048 ? 2048 : nb);
    WM_DISPATCH_DTYPE(dtype, "wm_concat_full",
        hipLaunchKernelGGL((concat_full_kernel<T>), dim3(grid), dim3(256), lds, s, (const T*)x, ldx, scale, shift,
                           msg, img, (T*)y, B, C, L, hw, ld));
    WM_LAUNCH_CHECK("wm_concat_full");
    return WM_OK;
}

extern "C" int wm_u8_hwc_to_planes(const unsigned char* x, float* y, int N, int H, int W, int C, float scale, void* stream) {
    WM_REQUIRE(x && y, WM_E_BADARG, "wm_u8_hwc_to_planes: null pointer");
    WM_REQUIRE(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, WM_E_BADARG, "wm_u8_hwc_to_planes: bad shape (1 <= C <= 4)");
    const size_t npix = (size_t)N * H * W;
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    hipLaunchKernelGGL(u8_hwc_to_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, npix, (size_t)H * W, C, scale);
    WM_LAUNCH_CHECK("wm_u8_hwc_to_planes");
    return WM_OK;
}
