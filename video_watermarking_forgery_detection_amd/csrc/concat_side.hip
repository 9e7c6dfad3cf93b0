// The encoder's after-concat layer without the concat (hidden_models/encoder.py:25,34-41).
//
// The reference builds cat([message expanded to HxW (L ch), features (64 ch), image (3 ch)]) = 97 channels and runs ConvBNRelu(97 -> 64)
// on it.  A 3x3 convolution is linear in its input channels, so
//     conv(cat) = conv64(features) + conv3(image) + conv_L(message planes) + bias
// and a message plane is CONSTANT over the image: its contribution at a pixel is the sum of W[:, l, tap] * m[b, l] over the taps that
// fall inside the image (zero padding), i.e. one of 9 vectors per sample -- interior, 4 edges, 4 corners.  This file computes the
// "side" term  P[b,h,w,:] = conv3(image)[b,h,w,:] + bias + mbias[b, class(h,w), :]  in ONE store-bound pass (the 27 image taps are
// a single K = 32 MFMA step), which conv3x3_ws_kernel<..., ADDIN> then adds to conv64(features) before the BatchNorm statistics:
// the 97(112)-channel tensor, its 79 us build and the 232 us streamed-filter conv are gone.
// Backward: the feature part is an ordinary 64 -> 64 layer; dW of the image channels comes from the image-fed weight-gradient kernel;
// dW of message channel l at tap t is sum_b m[b,l] * S[b,t,:] with S[b,t,:] = the sum of dy over the pixels for which tap t is inside
// the image -- per-sample sums of dy over the 9 border classes (dy_total / border kernels below).
#include "wm_common.h"

// compiled twice (build.py): the 16-bit activation dtype of P / dy is bf16, or f16 with -DWM_H16_F16
#ifdef WM_H16_F16
typedef f16_t hx_t;
#define WM_HSYM(name) name##_f16
#else
typedef bf16_t hx_t;
#define WM_HSYM(name) name##_bf16
#endif
typedef h16<hx_t> HX;
typedef HX::x8 hx8;

namespace {

constexpr int TS = 16, HS = 18;   // 16x16-pixel tiles, 18x18 halo
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// tap (kh, kw) reads pixel (h + kh - 1, w + kw - 1); row class rc: 0 = first row, 2 = last row, 1 = neither
__device__ __forceinline__ bool tap_valid(int k, int c) { return !(k == 0 && c == 0) && !(k == 2 && c == 2); }
__device__ __forceinline__ int cls_of(int h, int w, int H, int W) {
    return (h == 0 ? 0 : (h == H - 1 ? 2 : 1)) * 3 + (w == 0 ? 0 : (w == W - 1 ? 2 : 1));
}

#ifndef WM_H16_F16
// mbias[b][cls][co] = bias[co] + sum_l m[b][l] * (sum over the taps valid in cls of w[co][c0 + l][tap])   (w: [Cout][Cin][3][3] f32)
// one wave per output channel: its L x 9 message weights are ONE contiguous run in w (coalesced), folded per class in LDS
constexpr int MAXL = 64;
__global__ __launch_bounds__(64) void msg_bias_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ msg,
                                                      float* __restrict__ mbias, int B, int Cin, int c0, int L) {
    __shared__ float sw[MAXL * 9];
    __shared__ float sT[9][MAXL];
    const int co = blockIdx.x, lane = threadIdx.x;
    const float* wl = w + ((size_t)co * Cin + c0) * 9;
    for (int i = lane; i < L * 9; i += 64) sw[i] = wl[i];
    __syncthreads();
    for (int i = lane; i < 9 * L; i += 64) {
        const int cls = i / L, l = i - cls * L, rc = cls / 3, cc = cls % 3;
        float t = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            if (tap_valid(tap / 3, rc) && tap_valid(tap % 3, cc)) t += sw[l * 9 + tap];
        sT[cls][l] = t;
    }
    __syncthreads();
    const float b0 = bias ? bias[co] : 0.f;
    for (int i = lane; i < B * 9; i += 64) {
        const int b = i / 9, cls = i - b * 9;
        float acc = b0;
        for (int l = 0; l < L; ++l) acc = __builtin_fmaf(sT[cls][l], msg[(size_t)b * L + l], acc);
        mbias[((size_t)b * 9 + cls) * 64 + co] = acc;
    }
}

// dW of the message channels: dw[co][c0 + l][tap] (+)= sum_b m[b][l] * S[b][tap][co]
__global__ __launch_bounds__(64) void msg_wgrad_kernel(const float* __restrict__ S, const float* __restrict__ msg, float* __restrict__ dw, int B,
                                                       int Cin, int c0, int L, int accumulate) {
    const int l = blockIdx.x / 9, tap = blockIdx.x % 9, co = threadIdx.x;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = __builtin_fmaf(msg[(size_t)b * L + l], S[((size_t)b * 9 + tap) * 64 + co], acc);
    float* o = dw + ((size_t)co * Cin + c0 + l) * 9 + tap;
    *o = (accumulate ? *o : 0.f) + acc;
}
#endif

struct SideArgs {
    const float* img;      // [B][3][H][W] f32 planes (the reference's layout)
    const hx_t* wside;     // [64][32]: row = output channel, k = tap*3 + c (27 used), packed by side_pack_kernel
    const float* mbias;    // [B][9][64]
    hx_t* P;               // [B][H][W][64]
    int B, H, W, tilesX, tilesY, stripsX;
};
constexpr int XT = 4;   // 16x16 tiles per workgroup (a 16 x 64 strip): the filter fragments and the launch cost are shared

// wside[co][k] <- w[co][c0 + c][tap], k = 3*tap + c
__global__ void side_pack_kernel(const float* __restrict__ w, hx_t* __restrict__ wside, int Cin, int c0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 32) return;
    const int co = i / 32, k = i % 32;
    float v = 0.f;
    if (k < 27) v = w[((size_t)co * Cin + c0 + (k % 3)) * 9 + k / 3];
    wside[i] = (hx_t)v;
}

// one workgroup = one 16x16 tile; wave = 4 tile rows; per row one v_mfma_f32_16x16x32 per 16 output channels with the FILTER as the A
// operand (accumulator rows = channels): A row 4q'+i of fragment nf holds channel 16q' + 4nf + i, so lane (p, q) ends up with the 16
// adjacent channels [16q, 16q+16) of pixel p -- two 16-byte stores, no transpose (the layout trick of conv3x3_ws.hip)
__global__ __launch_bounds__(256) void concat_side_kernel(SideArgs a) {
    __shared__ float sImg[3][HS][XT * TS + 3];
    constexpr int SW = XT * TS + 2;   // halo columns of the strip
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int sx = t % a.stripsX; t /= a.stripsX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int b = t, y0 = ty * TS, xs = sx * XT * TS;
    const size_t plane = (size_t)a.H * a.W;
    for (int i = tid; i < 3 * HS * SW; i += 256) {
        const int c = i / (HS * SW), r = (i / SW) % HS, col = i % SW;
        const int gy = y0 - 1 + r, gx = xs - 1 + col;
        float v = 0.f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.img[((size_t)b * 3 + c) * plane + (size_t)gy * a.W + gx];
        sImg[c][r][col] = v;
    }
    const int p = lane & 15, q = lane >> 4;
    hx8 afr[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        const int co = 16 * (p >> 2) + 4 * nf + (p & 3);          // A row p of fragment nf
        afr[nf] = *reinterpret_cast<const hx8*>(a.wside + co * 32 + 8 * q);
    }
    __syncthreads();
    for (int xt = 0; xt < XT; ++xt) {
        const int x0 = xs + xt * TS;
        if (x0 >= a.W) break;   // (workgroup-uniform)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = wave * 4 + rr;
            const int gy = y0 + row, gx = x0 + p;
            // B operand: lane (p, q) = pixel p, k = 8q .. 8q+7
            unsigned bw[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 8 * q + 2 * e2 + u;
                    const int tap = k / 3, c = k - 3 * tap;
                    v[u] = k < 27 ? sImg[c][row + tap / 3][xt * TS + p + tap % 3] : 0.f;
                }
                bw[e2] = h16_pack<hx_t>(v[0], v[1]);
            }
            const hx8 bfr = __builtin_bit_cast(hx8, u32x4{bw[0], bw[1], bw[2], bw[3]});
            // every lane takes part in the MFMAs of its wave; only the stores are predicated
            const bool inb = gy < a.H && gx < a.W;
            const int cy = gy < a.H ? gy : a.H - 1, cx = gx < a.W ? gx : a.W - 1;
            const float* mb = a.mbias + ((size_t)b * 9 + cls_of(cy, cx, a.H, a.W)) * 64 + 16 * q;
            unsigned pk[8];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const f32x4 init = *reinterpret_cast<const f32x4*>(mb + 4 * nf);
                const f32x4 acc = HX::mfma16(afr[nf], bfr, init);
                pk[2 * nf] = h16_pack<hx_t>(acc[0], acc[1]);
                pk[2 * nf + 1] = h16_pack<hx_t>(acc[2], acc[3]);
            }
            if (inb) {
                hx_t* o = a.P + (((size_t)b * a.H + gy) * a.W + gx) * 64 + 16 * q;
                *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
        }
    }
}

// per-sample column sums of dy [B][HW][64]: partial[b][slice][64] (f32), deterministic two-stage
constexpr int DSL = 64;   // slices per sample
__global__ __launch_bounds__(256) void dy_total_kernel(const hx_t* __restrict__ dy, float* __restrict__ partial, size_t HW) {
    const int b = blockIdx.x / DSL, sl = blockIdx.x % DSL;
    const int vec = threadIdx.x & 7, pl = threadIdx.x >> 3;   // 8 vectors of 8 channels per pixel, 32 pixel lanes
    const size_t per = (HW + DSL - 1) / DSL, p0 = (size_t)sl * per, p1 = p0 + per < HW ? p0 + per : HW;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const hx_t* base = dy + (size_t)b * HW * 64 + vec * 8;
    size_t px = p0 + pl;
    for (; px + 96 < p1; px += 128) {   // four independent 16-byte loads in flight per thread
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(base + (px + 32 * u) * 64);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += HX::lo(w[u][e]); acc[2 * e + 1] += HX::hi(w[u][e]); }
    }
    for (; px < p1; px += 32) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(base + px * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += HX::lo(w[e]); acc[2 * e + 1] += HX::hi(w[e]); }
    }
    __shared__ float s[32][65];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[pl][vec * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += s[k][threadIdx.x];
        partial[((size_t)b * DSL + sl) * 64 + threadIdx.x] = t;
    }
}

// sums of dy over the border pixels of sample b by class: R[b][cls][64] for the 8 border classes (R[b][4] is not written here).
// grid = (4 edges, B): edge 0 / 1 = first / last row (corners included, split by column class), 2 / 3 = first / last column of the
// rows between.  16-byte loads, 32 pixel lanes.
__global__ __launch_bounds__(256) void dy_border_kernel(const hx_t* __restrict__ dy, float* __restrict__ R, int H, int W) {
    const int edge = blockIdx.x, b = blockIdx.y;
    const int vec = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const hx_t* base = dy + (size_t)b * H * W * 64 + vec * 8;
    float acc[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
    auto add = [&](int h, int w, int slot) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((size_t)h * W + w) * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[slot][2 * e] += HX::lo(v[e]); acc[slot][2 * e + 1] += HX::hi(v[e]); }
    };
    if (edge < 2) {
        const int h = edge == 0 ? 0 : H - 1;
        for (int w = pl; w < W; w += 32) add(h, w, w == 0 ? 0 : (w == W - 1 ? 2 : 1));
    } else {
        const int w = edge == 2 ? 0 : W - 1;
        for (int h = 1 + pl; h < H - 1; h += 32) add(h, w, 1);
    }
    __shared__ float s[3][32][65];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[c][pl][vec * 8 + e] = acc[c][e];
    __syncthreads();
    if (threadIdx.x < 192) {
        const int c = threadIdx.x / 64, co = threadIdx.x % 64;
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += s[c][k][co];
        // rows: classes (rc, 0..2) with rc = 0 / 2; columns: class (1, 0) / (1, 2) from slot 1 only
        if (edge < 2) R[((size_t)b * 9 + (edge == 0 ? 0 : 6) + c) * 64 + co] = t;
        else if (c == 1) R[((size_t)b * 9 + (edge == 2 ? 3 : 5)) * 64 + co] = t;
    }
}

// S[b][tap][co] = sum of dy over the pixels of sample b for which tap is inside the image
//             = sum over the classes in which the tap is valid of R[b][cls][co];  R[interior] = total - the 8 border classes
__global__ __launch_bounds__(64) void dy_S_kernel(const float* __restrict__ partial, const float* __restrict__ R, float* __restrict__ S) {
    const int b = blockIdx.x, co = threadIdx.x;
    float tot = 0.f;
    for (int sl = 0; sl < DSL; ++sl) tot += partial[((size_t)b * DSL + sl) * 64 + co];
    float rc[9], border = 0.f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        rc[c] = c == 4 ? 0.f : R[((size_t)b * 9 + c) * 64 + co];
        border += rc[c];
    }
    rc[4] = tot - border;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < 9; ++c)
            if (tap_valid(tap / 3, c / 3) && tap_valid(tap % 3, c % 3)) sacc += rc[c];
        S[((size_t)b * 9 + tap) * 64 + co] = sacc;
    }
}

}  // namespace

int WM_HSYM(wm_concat_side_fwd_impl)(const float* img, const float* w, const float* bias, const float* msg, void* wside, float* mbias, void* P,
                                     int B, int H, int W, int Cin, int c_msg, int L, int c_img, hipStream_t s);
int WM_HSYM(wm_concat_side_bwd_impl)(const void* dy, const float* msg, float* partial, float* S, float* dw, int accumulate, int B, int H, int W,
                                     int Cin, int c_msg, int L, hipStream_t s);

#ifndef WM_H16_F16
void wm_launch_msg_bias(const float* w, const float* bias, const float* msg, float* mbias, int B, int Cin, int c0, int L, hipStream_t s) {
    hipLaunchKernelGGL(msg_bias_kernel, dim3(64), dim3(64), 0, s, w, bias, msg, mbias, B, Cin, c0, L);
}
void wm_launch_msg_wgrad(const float* S, const float* msg, float* dw, int B, int Cin, int c0, int L, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(msg_wgrad_kernel, dim3(L * 9), dim3(64), 0, s, S, msg, dw, B, Cin, c0, L, accumulate);
}
#else
void wm_launch_msg_bias(const float* w, const float* bias, const float* msg, float* mbias, int B, int Cin, int c0, int L, hipStream_t s);
void wm_launch_msg_wgrad(const float* S, const float* msg, float* dw, int B, int Cin, int c0, int L, int accumulate, hipStream_t s);
#endif

int WM_HSYM(wm_concat_side_fwd_impl)(const float* img, const float* w, const float* bias, const float* msg, void* wside, float* mbias, void* P,
                                     int B, int H, int W, int Cin, int c_msg, int L, int c_img, hipStream_t s) {
    wm_launch_msg_bias(w, bias, msg, mbias, B, Cin, c_msg, L, s);
    hipLaunchKernelGGL(side_pack_kernel, dim3(8), dim3(256), 0, s, w, (hx_t*)wside, Cin, c_img);
    SideArgs a;
    a.img = img; a.wside = (const hx_t*)wside; a.mbias = mbias; a.P = (hx_t*)P; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TS); a.tilesY = wm_cdiv(H, TS); a.stripsX = wm_cdiv(a.tilesX, XT);
    hipLaunchKernelGGL(concat_side_kernel, dim3((unsigned)(B * a.stripsX * a.tilesY)), dim3(256), 0, s, a);
    return WM_OK;
}

int WM_HSYM(wm_concat_side_bwd_impl)(const void* dy, const float* msg, float* partial, float* S, float* dw, int accumulate, int B, int H, int W,
                                     int Cin, int c_msg, int L, hipStream_t s) {
    float* R = partial + (size_t)B * DSL * 64;   // [B][9][64] behind the slice rows
    hipLaunchKernelGGL(dy_total_kernel, dim3(B * DSL), dim3(256), 0, s, (const hx_t*)dy, partial, (size_t)H * W);
    hipLaunchKernelGGL(dy_border_kernel, dim3(4, B), dim3(256), 0, s, (const hx_t*)dy, R, H, W);
    hipLaunchKernelGGL(dy_S_kernel, dim3(B), dim3(64), 0, s, partial, R, S);
    wm_launch_msg_wgrad(S, msg, dw, B, Cin, c_msg, L, accumulate, s);
    return WM_OK;
}

#ifndef WM_H16_F16
int wm_concat_side_fwd_impl_f16(const float* img, const float* w, const float* bias, const float* msg, void* wside, float* mbias, void* P, int B,
                                int H, int W, int Cin, int c_msg, int L, int c_img, hipStream_t s);
int wm_concat_side_bwd_impl_f16(const void* dy, const float* msg, float* partial, float* S, float* dw, int accumulate, int B, int H, int W,
                                int Cin, int c_msg, int L, hipStream_t s);

extern "C" int wm_concat_side_partial_rows(void) { return DSL + 9; }   // rows of 64 floats per sample in `partial`

// P[B,H,W,64] = conv3x3 of the image channels [c_img, c_img+3) of w [64][Cin][3][3] over img [B,3,H,W] (f32 NCHW planes, zero padded)
//               + bias + the message term of channels [c_msg, c_msg+L) for messages msg [B][L]
// mbias: f32 [B][9][64] scratch (kept for nothing else); wside: 4 KB scratch for the packed 27-tap filter
extern "C" int wm_concat_side_fwd(const float* img, const float* w, const float* bias, const float* msg, void* wside, float* mbias, void* P,
                                  int B, int H, int W, int Cin, int c_msg, int L, int c_img, int dtype, void* stream) {
    WM_REQUIRE(img && w && msg && wside && mbias && P, WM_E_BADARG, "wm_concat_side_fwd: null pointer");
    WM_REQUIRE(B > 0 && H >= 2 && W >= 2 && L > 0 && L <= 64 && c_msg >= 0 && c_img >= 0 && c_msg + L <= Cin && c_img + 3 <= Cin, WM_E_BADARG,
               "wm_concat_side_fwd: bad shape (H, W >= 2; message length <= 64; the channel ranges must lie inside Cin=%d)", Cin);
    WM_REQUIRE(((uintptr_t)P & 15) == 0 && ((uintptr_t)wside & 15) == 0, WM_E_SHAPE, "wm_concat_side_fwd: P / wside must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == WM_BF16) rc = wm_concat_side_fwd_impl_bf16(img, w, bias, msg, wside, mbias, P, B, H, W, Cin, c_msg, L, c_img, s);
    else if (dtype == WM_F16) rc = wm_concat_side_fwd_impl_f16(img, w, bias, msg, wside, mbias, P, B, H, W, Cin, c_msg, L, c_img, s);
    else { wm_set_error("wm_concat_side_fwd: dtype must be WM_BF16 or WM_F16 (got %d)", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_concat_side_fwd");
    return rc;
}

// dw[co][c_msg + l][tap] (+)= sum_b msg[b][l] * (sum of dy[b] over the pixels for which the tap lies inside the image)
// dy: [B,H,W,64] dense, 16-bit; partial: f32 [B][64 + 9][64] scratch; S: f32 [B][9][64] scratch
extern "C" int wm_concat_side_msg_wgrad(const void* dy, const float* msg, float* partial, float* S, float* dw, int accumulate, int B, int H,
                                        int W, int Cin, int c_msg, int L, int dtype, void* stream) {
    WM_REQUIRE(dy && msg && partial && S && dw, WM_E_BADARG, "wm_concat_side_msg_wgrad: null pointer");
    WM_REQUIRE(B > 0 && H >= 2 && W >= 2 && L > 0 && c_msg >= 0 && c_msg + L <= Cin, WM_E_BADARG, "wm_concat_side_msg_wgrad: bad shape");
    WM_REQUIRE(((uintptr_t)dy & 15) == 0, WM_E_SHAPE, "wm_concat_side_msg_wgrad: dy must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == WM_BF16) rc = wm_concat_side_bwd_impl_bf16(dy, msg, partial, S, dw, accumulate, B, H, W, Cin, c_msg, L, s);
    else if (dtype == WM_F16) rc = wm_concat_side_bwd_impl_f16(dy, msg, partial, S, dw, accumulate, B, H, W, Cin, c_msg, L, s);
    else { wm_set_error("wm_concat_side_msg_wgrad: dtype must be WM_BF16 or WM_F16 (got %d)", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_concat_side_msg_wgrad");
    return rc;
}
#endif
