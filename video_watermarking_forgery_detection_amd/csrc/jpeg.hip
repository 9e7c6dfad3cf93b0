// Fused block-JPEG attack kernels (forward + backward) for gfx950.
//
// Reference semantics: noise_layers/jpeg.py:52-306 (Jpeg / JpegSS / JpegMask) -- see
// include/wm_hip.h.  One kernel per direction does the whole chain
//   x255 -> zero-pad to /8 -> RGB->YUV -> (4:2:0 by replication) -> 8x8 DCT -> quantise ->
//   round / round_ss / mask -> de-quantise -> IDCT -> YUV->RGB -> crop -> /255
// with a single read and a single write of the image: 24 B/px algorithmic traffic, HBM-bound.
//
// Work decomposition (wave64-native): one wavefront owns 8 horizontally adjacent 8x8 blocks
// (a 64 px x 8 row strip) of all three channels.  Lane l = 8*r + blk holds row r of block blk
// (8 px x 3 channels in registers), so a wave's global loads cover 256 contiguous bytes per
// image row.  The row pass of the separable DCT is register-only; the column pass needs the
// 8x8 transpose between the 8 lanes of a block, done through a wave-private LDS tile
// (row stride 68 floats: conflict-free for both the b128 row access and the strided
// column access).  No workgroup barrier is needed: every exchange is inside one wave.
#include "wm_common.h"

namespace {

// orthonormal DCT-II basis, jpeg.py:117-121: C[i][j] = sqrt(2/8) cos(pi i (2j+1)/16), row 0 = sqrt(1/8)
__device__ constexpr float kC[8][8] = {
    {3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f},
    {4.903926253e-01f, 4.157347977e-01f, 2.777851224e-01f, 9.754516184e-02f, -9.754516184e-02f, -2.777851224e-01f, -4.157347977e-01f, -4.903926253e-01f},
    {4.619397521e-01f, 1.913417131e-01f, -1.913417131e-01f, -4.619397521e-01f, -4.619397521e-01f, -1.913417131e-01f, 1.913417131e-01f, 4.619397521e-01f},
    {4.157347977e-01f, -9.754516184e-02f, -4.903926253e-01f, -2.777851224e-01f, 2.777851224e-01f, 4.903926253e-01f, 9.754516184e-02f, -4.157347977e-01f},
    {3.535533845e-01f, -3.535533845e-01f, -3.535533845e-01f, 3.535533845e-01f, 3.535533845e-01f, -3.535533845e-01f, -3.535533845e-01f, 3.535533845e-01f},
    {2.777851224e-01f, -4.903926253e-01f, 9.754516184e-02f, 4.157347977e-01f, -4.157347977e-01f, -9.754516184e-02f, 4.903926253e-01f, -2.777851224e-01f},
    {1.913417131e-01f, -4.619397521e-01f, 4.619397521e-01f, -1.913417131e-01f, -1.913417131e-01f, 4.619397521e-01f, -4.619397521e-01f, 1.913417131e-01f},
    {9.754516184e-02f, -2.777851224e-01f, 4.157347977e-01f, -4.903926253e-01f, 4.903926253e-01f, -4.157347977e-01f, 2.777851224e-01f, -9.754516184e-02f}};

struct JpegTables {
    float t[128];  // lum[64], chroma[64], row-major [u][v]
};

constexpr int LDS_BLK = 68;               // floats per 8x8 block image in LDS (64 + 4 pad)
constexpr int LDS_WAVE = 3 * 8 * LDS_BLK;  // 3 channels x 8 blocks per wave

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// out[v] = sum_j C[v][j] in[j]   (X * C^T along the register axis)
__device__ __forceinline__ void dct8(const float (&in)[8], float (&out)[8]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += kC[v][j] * in[j];
        out[v] = a;
    }
}
// out[j] = sum_v C[v][j] in[v]   (X * C along the register axis)
__device__ __forceinline__ void idct8(const float (&in)[8], float (&out)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) a += kC[v][j] * in[v];
        out[j] = a;
    }
}

// 8x8 transpose between the 8 lanes (r = 0..7) that share `blk`; three channels at once.
// in: lane r holds row r.  out: lane r holds column r (element k = row k).
__device__ __forceinline__ void transpose3(float (&v)[3][8], float* lds, int r, int blk) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* p = lds + (c * 8 + blk) * LDS_BLK + r * 8;
        *reinterpret_cast<float4*>(p) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[c][4], v[c][5], v[c][6], v[c][7]);
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = lds + (c * 8 + blk) * LDS_BLK + r;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = p[k * 8];
    }
    wave_lds_sync();
}

// forward 2-D DCT of three channels.  in: lane r holds pixel row r.  out: lane r holds
// coefficient column v = r, element u = vertical frequency:  F[u][v] = (C X C^T)[u][v].
__device__ __forceinline__ void dct2d3(float (&v)[3][8], float* lds, int r, int blk) {
    float t[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dct8(v[c], t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = t[k];
    }
    transpose3(v, lds, r, blk);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dct8(v[c], t);  // F[u] = sum_k C[u][k] col[k]
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = t[k];
    }
}
// inverse: in column layout (lane r = column v), out pixel rows:  X = C^T F C.
__device__ __forceinline__ void idct2d3(float (&v)[3][8], float* lds, int r, int blk) {
    float t[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        idct8(v[c], t);  // S[k] = sum_u C[u][k] F[u]
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = t[k];
    }
    transpose3(v, lds, r, blk);  // symmetric exchange: lane r now holds row r, element = column v
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        idct8(v[c], t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = t[k];
    }
}

struct Task {
    int b, y, x0;
    bool valid;
};

__device__ __forceinline__ Task wave_task(int H, int W, int B, int r, int blk) {
    const int Hb = (H + 7) >> 3;
    const int Ws = (W + 63) >> 6;  // 64-px strips per block-row
    const long total = (long)B * Hb * Ws;
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    Task t;
    t.valid = wid < total;
    const long w = t.valid ? wid : 0;
    const int strip = (int)(w % Ws);
    const int br = (int)((w / Ws) % Hb);
    t.b = (int)(w / ((long)Ws * Hb));
    t.y = br * 8 + r;
    t.x0 = strip * 64 + blk * 8;
    return t;
}

__device__ __forceinline__ void load_rows(const float* __restrict__ x, const Task& t, int H, int W,
                                          float (&v)[3][8]) {
    const size_t plane = (size_t)H * W;
    const bool rowok = t.valid && t.y < H;
    const bool vec = rowok && ((W & 3) == 0) && (t.x0 + 8 <= W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = x + ((size_t)t.b * 3 + c) * plane + (size_t)t.y * W + t.x0;
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(p);
            const float4 b = *reinterpret_cast<const float4*>(p + 4);
            v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
            v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = (rowok && t.x0 + j < W) ? p[j] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_rows(float* __restrict__ y, const Task& t, int H, int W,
                                           const float (&v)[3][8]) {
    const size_t plane = (size_t)H * W;
    const bool rowok = t.valid && t.y < H;
    if (!rowok) return;
    const bool vec = ((W & 3) == 0) && (t.x0 + 8 <= W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* p = y + ((size_t)t.b * 3 + c) * plane + (size_t)t.y * W + t.x0;
        if (vec) {
            *reinterpret_cast<float4*>(p) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(v[c][4], v[c][5], v[c][6], v[c][7]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (t.x0 + j < W) p[j] = v[c][j];
        }
    }
}

// jpeg.py:147-155
__device__ __forceinline__ void rgb2yuv(float (&v)[3][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float R = v[0][j], G = v[1][j], B = v[2][j];
        v[0][j] = 0.299f * R + 0.587f * G + 0.114f * B;
        v[1][j] = -0.1687f * R - 0.3313f * G + 0.5f * B;
        v[2][j] = 0.5f * R - 0.4187f * G - 0.0813f * B;
    }
}
// jpeg.py:157-163
__device__ __forceinline__ void yuv2rgb(float (&v)[3][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float Y = v[0][j], U = v[1][j], V = v[2][j];
        v[0][j] = Y + 1.40198758f * V;
        v[1][j] = Y - 0.344113281f * U - 0.714103821f * V;
        v[2][j] = Y + 1.77197812f * U;
    }
}
// jpeg.py:202-211: odd rows of U,V <- even row above, then odd columns <- even column left
__device__ __forceinline__ void subsample420(float (&v)[3][8], int r) {
#pragma unroll
    for (int c = 1; c < 3; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float up = __shfl_up(v[c][j], 8, 64);
            if (r & 1) v[c][j] = up;
        }
#pragma unroll
        for (int j = 1; j < 8; j += 2) v[c][j] = v[c][j - 1];
    }
}
// transpose of subsample420 (gradient): even rows/cols collect their 2x2 group, odd get zero
__device__ __forceinline__ void subsample420_T(float (&v)[3][8], int r) {
#pragma unroll
    for (int c = 1; c < 3; ++c) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            v[c][j] += v[c][j + 1];
            v[c][j + 1] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dn = __shfl_down(v[c][j], 8, 64);
            v[c][j] = (r & 1) ? 0.f : v[c][j] + dn;
        }
    }
}

__device__ __forceinline__ void load_tables(const JpegTables& tb, float* s_tbl) {
    if (threadIdx.x < 128) s_tbl[threadIdx.x] = tb.t[threadIdx.x];
    __syncthreads();
}

// act16 (round 4): the attacked image a second time as [B][H][W][16] pixels of the activation dtype (channels 0..2, zero tail): the tensor
// the decoder's image-fed first layer reads, which wm_nchw_to_nhwc would otherwise make from y in a launch of its own.  A lane holds 8
// adjacent pixels of one row: 8 x 32 (16-bit) or 8 x 64 (f32) contiguous bytes
template <typename T>
__device__ __forceinline__ void store_act16_t(T* __restrict__ a, const Task& t, int H, int W, const float (&v)[3][8], float* lds, int r, int blk) {
    constexpr int VE = vec16<T>::N;
    if constexpr (VE == 8) {
        // 16-bit activations: a pixel is 32 bytes.  Through the wave's LDS tile ([channel][row][64 pixels], stride 68) so that one store
        // instruction writes 32 ADJACENT pixels = 1 KB contiguous (lane l: pixel l >> 1, 16-byte half l & 1; the second half is the zero
        // tail) -- written straight from the registers (a lane holds 8 pixels of one row) every store instruction would touch 64 different
        // 256-byte segments with 16 bytes each: 40 us instead of 15 for the whole kernel at B=16, 256x256
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) lds[(c * 8 + r) * LDS_BLK + blk * 8 + j] = v[c][j];
        if (!t.valid) return;
        const int lane = threadIdx.x & 63, y0 = t.y - r, xs = t.x0 - blk * 8;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = (lane >> 1) + 32 * h;
                vec16<T> o;
#pragma unroll
                for (int e = 0; e < VE; ++e) o.set(e, 0.f);
                if ((lane & 1) == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) o.set(c, lds[(c * 8 + rr) * LDS_BLK + px]);
                }
                if (y0 + rr < H && xs + px < W)
                    *reinterpret_cast<vec16<T>*>(a + (((size_t)t.b * H + y0 + rr) * W + xs + px) * 16 + (lane & 1) * VE) = o;
            }
    } else {
        if (!(t.valid && t.y < H)) return;
        T* p = a + (((size_t)t.b * H + t.y) * W + t.x0) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (t.x0 + j < W) {
#pragma unroll
                for (int k = 0; k < 16 / VE; ++k) {
                    vec16<T> o;
#pragma unroll
                    for (int e = 0; e < VE; ++e) o.set(e, (k == 0 && e < 3) ? v[e][j] : 0.f);
                    *reinterpret_cast<vec16<T>*>(p + j * 16 + k * VE) = o;
                }
            }
        }
    }
}
__device__ __forceinline__ void store_act16(void* a, int dtype, const Task& t, int H, int W, const float (&v)[3][8], float* lds, int r, int blk) {
    if (dtype == WM_BF16) store_act16_t(reinterpret_cast<bf16_t*>(a), t, H, W, v, lds, r, blk);
    else if (dtype == WM_F16) store_act16_t(reinterpret_cast<f16_t*>(a), t, H, W, v, lds, r, blk);
    else store_act16_t(reinterpret_cast<float*>(a), t, H, W, v, lds, r, blk);
}

template <int MODE, int SUB>
__global__ __launch_bounds__(256) void jpeg_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int B, int H, int W, JpegTables tb, void* __restrict__ act16, int act_dtype) {
    __shared__ __attribute__((aligned(16))) float s_lds[4 * LDS_WAVE];
    __shared__ float s_tbl[128];
    if (MODE != WM_JPEG_MASK) load_tables(tb, s_tbl);
    const int lane = threadIdx.x & 63;
    const int r = lane >> 3, blk = lane & 7;
    float* lds = s_lds + (threadIdx.x >> 6) * LDS_WAVE;
    const Task t = wave_task(H, W, B, r, blk);

    float v[3][8];
    load_rows(x, t, H, W, v);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] *= 255.f;
    rgb2yuv(v);
    if (SUB == 2) subsample420(v, r);
    dct2d3(v, lds, r, blk);
    // lane r holds F[u][v=r] for u = 0..7
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == WM_JPEG_MASK) {
                const int keep = (c == 0) ? 5 : 3;  // jpeg.py:288-291
                v[c][u] = (u < keep && r < keep) ? v[c][u] : 0.f;
            } else {
                const float tq = s_tbl[(c == 0 ? 0 : 64) + u * 8 + r];
                float q = v[c][u] / tq;
                if (MODE == WM_JPEG_ROUND) q = rintf(q);                     // torch.round: half to even
                else q = (fabsf(q) < 0.5f) ? q * q * q : q;                  // round_ss, jpeg.py:255-257
                v[c][u] = q * tq;
            }
        }
    }
    idct2d3(v, lds, r, blk);
    yuv2rgb(v);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = v[c][j] / 255.f;
    store_rows(y, t, H, W, v);
    if (act16) store_act16(act16, act_dtype, t, H, W, v, lds, r, blk);
}

template <int MODE, int SUB>
__global__ __launch_bounds__(256) void jpeg_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gx, int B, int H, int W,
                                                        JpegTables tb) {
    __shared__ __attribute__((aligned(16))) float s_lds[4 * LDS_WAVE];
    __shared__ float s_tbl[128];
    if (MODE == WM_JPEG_SS) load_tables(tb, s_tbl);
    const int lane = threadIdx.x & 63;
    const int r = lane >> 3, blk = lane & 7;
    float* lds = s_lds + (threadIdx.x >> 6) * LDS_WAVE;
    const Task t = wave_task(H, W, B, r, blk);

    float g[3][8];
    if (MODE == WM_JPEG_ROUND) {
        // d round / dq == 0 everywhere torch.round is differentiated
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) g[c][j] = 0.f;
        store_rows(gx, t, H, W, g);
        return;
    }
    load_rows(gy, t, H, W, g);
    // y = rgb/255 ; rgb = M * yuv  -> d yuv = M^T d rgb
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float dR = g[0][j] / 255.f, dG = g[1][j] / 255.f, dB = g[2][j] / 255.f;
        g[0][j] = dR + dG + dB;
        g[1][j] = -0.344113281f * dG + 1.77197812f * dB;
        g[2][j] = 1.40198758f * dR - 0.714103821f * dG;
    }
    // rec = C^T D C  ->  dD = C dRec C^T  (a forward DCT of the incoming gradient)
    dct2d3(g, lds, r, blk);
    if (MODE == WM_JPEG_MASK) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int keep = (c == 0) ? 5 : 3;
#pragma unroll
            for (int u = 0; u < 8; ++u) g[c][u] = (u < keep && r < keep) ? g[c][u] : 0.f;
        }
    } else {
        // recompute the pre-rounding coefficients q = F/t from x; D = round_ss(q)*t so
        // dD/dF = round_ss'(q) = 3 q^2 (|q| < 0.5) else 1
        float v[3][8];
        load_rows(x, t, H, W, v);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] *= 255.f;
        rgb2yuv(v);
        if (SUB == 2) subsample420(v, r);
        dct2d3(v, lds, r, blk);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float q = v[c][u] / s_tbl[(c == 0 ? 0 : 64) + u * 8 + r];
                g[c][u] *= (fabsf(q) < 0.5f) ? 3.f * q * q : 1.f;
            }
        }
    }
    // F = C X C^T  ->  dX = C^T dF C
    idct2d3(g, lds, r, blk);
    if (SUB == 2) subsample420_T(g, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float dY = g[0][j], dU = g[1][j], dV = g[2][j];
        g[0][j] = (0.299f * dY - 0.1687f * dU + 0.5f * dV) * 255.f;
        g[1][j] = (0.587f * dY - 0.3313f * dU - 0.4187f * dV) * 255.f;
        g[2][j] = (0.114f * dY + 0.5f * dU - 0.0813f * dV) * 255.f;
    }
    store_rows(gx, t, H, W, g);
}


// =====================================================================================================
// DiffJPEG (utils/JPEG.py:115-540 of the reference): JFIF colour matrices with +128 offsets, 2x2 average
// chroma subsampling, 8x8 DCT on Y (full res) and Cb/Cr (half res), transposed tables * factor, selectable
// rounding, nearest chroma up-sampling, clamp to [0,255].  H, W multiples of 16.
//
// One wave owns a 64 x 16 pixel strip = 4 MCUs.  The (row r, block blk) lane layout of the block-JPEG
// kernel is reused with the three "channels" of dct2d3 re-purposed as
//     v[0] = Y row r of block blk in the upper 8 rows,  v[1] = same in the lower 8 rows,
//     v[2] = row r of a half-resolution chroma block: Cb of MCU blk (blk < 4) or Cr of MCU blk-4
// so a single dct2d3 / idct2d3 call transforms all six blocks of every MCU.
struct DiffTables { float t[128]; };  // y_table*factor [64], c_table*factor [64], row-major [u][v]

template <int RND> __device__ __forceinline__ float dj_round(float q) {
    if (RND == 0) return rintf(q);
    if (RND == 1) return (fabsf(q) < 0.5f) ? q * q * q : q;       // round_only_at_0, JPEG.py:482-484
    const float r = rintf(q);                                      // diff_round, JPEG.py:472-479
    return r + (q - r) * (q - r) * (q - r);
}
template <int RND> __device__ __forceinline__ float dj_round_grad(float q) {
    if (RND == 0) return 0.f;
    if (RND == 1) return (fabsf(q) < 0.5f) ? 3.f * q * q : 1.f;
    const float d = q - rintf(q);
    return 3.f * d * d;
}

struct DjTask { int b, y0, x0; bool valid; };
__device__ __forceinline__ DjTask dj_task(int B, int H, int W) {
    const int Hs = H >> 4, Ws = (W + 63) >> 6;
    const long total = (long)B * Hs * Ws;
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    DjTask t;
    t.valid = wid < total;
    const long w = t.valid ? wid : 0;
    t.x0 = (int)(w % Ws) * 64;
    t.y0 = (int)((w / Ws) % Hs) * 16;
    t.b = (int)(w / ((long)Ws * Hs));
    return t;
}

// JPEG.py:121-130 on x*255
__device__ __forceinline__ void dj_ycc(float R, float G, float B, float& Y, float& Cb, float& Cr) {
    Y = 0.299f * R + 0.587f * G + 0.114f * B;
    Cb = -0.168736f * R - 0.331264f * G + 0.5f * B + 128.f;
    Cr = 0.5f * R - 0.418688f * G - 0.081312f * B + 128.f;
}

// loads the strip into v[0], v[1] (Y - 128) and v[2] (avg-pooled chroma - 128) for lane (r, blk)
__device__ __forceinline__ void dj_load(const float* __restrict__ x, const DjTask& t, int H, int W, int r, int blk,
                                        float (&v)[3][8]) {
    const size_t plane = (size_t)H * W;
    const float* base = x + (size_t)t.b * 3 * plane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int yy = t.y0 + 8 * s + r, xx = t.x0 + 8 * blk;
        const bool ok = t.valid && xx < W;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float Y = 128.f, Cb, Cr;
            if (ok) {
                const size_t o = (size_t)yy * W + xx + j;
                dj_ycc(base[o] * 255.f, base[plane + o] * 255.f, base[2 * plane + o] * 255.f, Y, Cb, Cr);
            }
            v[s][j] = Y - 128.f;
        }
    }
    const int m = blk & 3;
    const bool cr_lane = blk >= 4;
    const int xx = t.x0 + 16 * m, yy = t.y0 + 2 * r;
    const bool ok = t.valid && xx < W;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
        if (ok) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t o = (size_t)(yy + dy) * W + xx + 2 * j + dx;
                    float Y, Cb, Cr;
                    dj_ycc(base[o] * 255.f, base[plane + o] * 255.f, base[2 * plane + o] * 255.f, Y, Cb, Cr);
                    acc += cr_lane ? Cr : Cb;
                }
            acc *= 0.25f;
        } else acc = 128.f;
        v[2][j] = acc - 128.f;
    }
}

// chroma exchange: lanes publish their chroma row, every lane picks the (nearest-upsampled) Cb/Cr of its two Y rows
__device__ __forceinline__ void dj_chroma_to_y(const float (&c)[8], float* lds, int r, int blk, float (&cb)[2][8],
                                               float (&cr)[2][8]) {
    float* p = lds + blk * 64 + r * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = c[j];
    wave_lds_sync();
    const int m = blk >> 1, half = (blk & 1) * 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int crow = 4 * s + (r >> 1);
        const float* pb = lds + m * 64 + crow * 8 + half;
        const float* pr = lds + (m + 4) * 64 + crow * 8 + half;
#pragma unroll
        for (int j = 0; j < 8; ++j) { cb[s][j] = pb[j >> 1]; cr[s][j] = pr[j >> 1]; }
    }
    wave_lds_sync();
}

// forward core shared by both kernels: leaves q (pre-rounding coefficients, column layout) in `q` and the
// unclamped 0..255 RGB of the lane's two rows in rgb[s][c][j]
template <int RND>
__device__ __forceinline__ void dj_forward(const float* __restrict__ x, const DjTask& t, int H, int W, int r, int blk,
                                           float* lds, const float* s_tbl, float (&q)[3][8], float (&rgb)[2][3][8]) {
    float v[3][8];
    dj_load(x, t, H, W, r, blk, v);
    dct2d3(v, lds, r, blk);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float tq = s_tbl[(c == 2 ? 64 : 0) + u * 8 + r];
            q[c][u] = v[c][u] / tq;
            v[c][u] = dj_round<RND>(q[c][u]) * tq;
        }
    idct2d3(v, lds, r, blk);
    float cbv[2][8], crv[2][8];
    float chroma[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) chroma[j] = v[2][j] + 128.f;
    dj_chroma_to_y(chroma, lds, r, blk, cbv, crv);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float Y = v[s][j] + 128.f, Cb = cbv[s][j] - 128.f, Cr = crv[s][j] - 128.f;  // JPEG.py:419-426
            rgb[s][0][j] = Y + 1.402f * Cr;
            rgb[s][1][j] = Y - 0.344136f * Cb - 0.714136f * Cr;
            rgb[s][2][j] = Y + 1.772f * Cb;
        }
}

template <int RND>
__global__ __launch_bounds__(256) void diffjpeg_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                            int W, DiffTables tb) {
    __shared__ __attribute__((aligned(16))) float s_lds[4 * LDS_WAVE];
    __shared__ float s_tbl[128];
    if (threadIdx.x < 128) s_tbl[threadIdx.x] = tb.t[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane >> 3, blk = lane & 7;
    float* lds = s_lds + (threadIdx.x >> 6) * LDS_WAVE;
    const DjTask t = dj_task(B, H, W);
    float q[3][8], rgb[2][3][8];
    dj_forward<RND>(x, t, H, W, r, blk, lds, s_tbl, q, rgb);
    const size_t plane = (size_t)H * W;
    const int xx = t.x0 + 8 * blk;
    if (t.valid && xx < W) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float* p = y + ((size_t)t.b * 3 + c) * plane + (size_t)(t.y0 + 8 * s + r) * W + xx;
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = fminf(255.f, fmaxf(0.f, rgb[s][c][j])) / 255.f;  // JPEG.py:467-469
            }
    }
}

template <int RND>
__global__ __launch_bounds__(256) void diffjpeg_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                            float* __restrict__ gx, int B, int H, int W, DiffTables tb) {
    __shared__ __attribute__((aligned(16))) float s_lds[4 * LDS_WAVE];
    __shared__ float s_tbl[128];
    if (threadIdx.x < 128) s_tbl[threadIdx.x] = tb.t[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane >> 3, blk = lane & 7;
    float* lds = s_lds + (threadIdx.x >> 6) * LDS_WAVE;
    const DjTask t = dj_task(B, H, W);
    const size_t plane = (size_t)H * W;
    const int xx = t.x0 + 8 * blk;
    const bool ok = t.valid && xx < W;
    float q[3][8], rgb[2][3][8];
    dj_forward<RND>(x, t, H, W, r, blk, lds, s_tbl, q, rgb);
    // ---- d rgb (clamp mask, /255) -> dY per pixel, dCb/dCr per pixel
    float g[3][8];
    float dcb[2][8], dcr[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float gv = 0.f;
                if (ok) gv = gy[((size_t)t.b * 3 + c) * plane + (size_t)(t.y0 + 8 * s + r) * W + xx + j];
                const float o = rgb[s][c][j];
                d[c] = (o >= 0.f && o <= 255.f) ? gv / 255.f : 0.f;   // torch.min/max pass the gradient inside the range
            }
            g[s][j] = d[0] + d[1] + d[2];
            dcb[s][j] = -0.344136f * d[1] + 1.772f * d[2];
            dcr[s][j] = 1.402f * d[0] - 0.714136f * d[1];
        }
    // ---- nearest up-sampling^T: every chroma sample collects its 2x2 pixels.  Layout [kind][s][r][blk][4]
    {
        float* pb = lds + ((0 * 2 + 0) * 8 + r) * 32 + blk * 4;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pb[((0 * 2 + s) * 8) * 32 + k] = dcb[s][2 * k] + dcb[s][2 * k + 1];
                pb[((1 * 2 + s) * 8) * 32 + k] = dcr[s][2 * k] + dcr[s][2 * k + 1];
            }
        wave_lds_sync();
        const int kind = blk >> 2, m = blk & 3, s = r >> 2, r0 = 2 * (r & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int yb = 2 * m + (j >> 2), k = j & 3;
            const float* pq = lds + (((kind * 2 + s) * 8 + r0) * 8 + yb) * 4 + k;
            g[2][j] = pq[0] + pq[32];  // rows r0 and r0+1
        }
        wave_lds_sync();
    }
    // ---- IDCT^T = DCT, rounding derivative, DCT^T = IDCT
    dct2d3(g, lds, r, blk);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < 8; ++u) g[c][u] *= dj_round_grad<RND>(q[c][u]);
    idct2d3(g, lds, r, blk);
    // ---- avg-pool^T (x 1/4) + chroma exchange, colour transform^T, x255
    float chroma[8], gcb[2][8], gcr[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) chroma[j] = 0.25f * g[2][j];
    dj_chroma_to_y(chroma, lds, r, blk, gcb, gcr);
    if (ok) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float* p = gx + (size_t)t.b * 3 * plane + (size_t)(t.y0 + 8 * s + r) * W + xx;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dY = g[s][j], dCb = gcb[s][j], dCr = gcr[s][j];
                p[j] = (0.299f * dY - 0.168736f * dCb + 0.5f * dCr) * 255.f;
                p[plane + j] = (0.587f * dY - 0.331264f * dCb - 0.418688f * dCr) * 255.f;
                p[2 * plane + j] = (0.114f * dY + 0.5f * dCb - 0.081312f * dCr) * 255.f;
            }
        }
    }
}

int check_args(const char* name, const void* a, const void* b, int B, int H, int W, int mode, const float* tables,
               int subsample) {
    WM_REQUIRE(a && b, WM_E_BADARG, "%s: null tensor pointer", name);
    WM_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, WM_E_SHAPE, "%s: tensors must be 16-byte aligned", name);
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "%s: bad shape B=%d H=%d W=%d", name, B, H, W);
    WM_REQUIRE(mode >= 0 && mode <= 2, WM_E_BADARG, "%s: mode %d not in {0 round,1 ss,2 mask}", name, mode);
    WM_REQUIRE(subsample == 0 || subsample == 2, WM_E_BADARG, "%s: subsample must be 0 or 2 (got %d)", name, subsample);
    WM_REQUIRE(mode == WM_JPEG_MASK || tables, WM_E_BADARG, "%s: quantisation tables required", name);
    return WM_OK;
}

inline unsigned jpeg_grid(int B, int H, int W) {
    const long waves = (long)B * ((H + 7) / 8) * ((W + 63) / 64);
    return (unsigned)((waves + 3) / 4);
}

}  // namespace

extern "C" int wm_jpeg_fwd(const float* x, float* y, int B, int H, int W, int mode, const float* tables,
                           int subsample, void* stream) {
    return wm_jpeg_fwd_act(x, y, nullptr, WM_F32, B, H, W, mode, tables, subsample, stream);
}

extern "C" int wm_jpeg_fwd_act(const float* x, float* y, void* act16, int act_dtype, int B, int H, int W, int mode, const float* tables,
                               int subsample, void* stream) {
    int rc = check_args("wm_jpeg_fwd", x, y, B, H, W, mode, tables, subsample);
    if (rc) return rc;
    WM_REQUIRE(!act16 || (((uintptr_t)act16 & 15) == 0 && (act_dtype == WM_F32 || act_dtype == WM_BF16 || act_dtype == WM_F16)), WM_E_BADARG,
               "wm_jpeg_fwd_act: act16 must be 16-byte aligned, its dtype f32 / bf16 / f16");
    JpegTables tb;
    for (int i = 0; i < 128; ++i) tb.t[i] = tables ? tables[i] : 1.f;
    const dim3 grid(jpeg_grid(B, H, W)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(M, S) hipLaunchKernelGGL((jpeg_fwd_kernel<M, S>), grid, block, 0, s, x, y, B, H, W, tb, act16, act_dtype)
    if (subsample == 0) {
        if (mode == 0) L(0, 0); else if (mode == 1) L(1, 0); else L(2, 0);
    } else {
        if (mode == 0) L(0, 2); else if (mode == 1) L(1, 2); else L(2, 2);
    }
#undef L
    WM_LAUNCH_CHECK("wm_jpeg_fwd");
    return WM_OK;
}

extern "C" int wm_jpeg_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int mode,
                           const float* tables, int subsample, void* stream) {
    int rc = check_args("wm_jpeg_bwd", gy, gx, B, H, W, mode, tables, subsample);
    if (rc) return rc;
    WM_REQUIRE(mode != WM_JPEG_SS || x, WM_E_BADARG, "wm_jpeg_bwd: x required for round_ss");
    JpegTables tb;
    for (int i = 0; i < 128; ++i) tb.t[i] = tables ? tables[i] : 1.f;
    const dim3 grid(jpeg_grid(B, H, W)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(M, S) hipLaunchKernelGGL((jpeg_bwd_kernel<M, S>), grid, block, 0, s, x, gy, gx, B, H, W, tb)
    if (subsample == 0) {
        if (mode == 0) L(0, 0); else if (mode == 1) L(1, 0); else L(2, 0);
    } else {
        if (mode == 0) L(0, 2); else if (mode == 1) L(1, 2); else L(2, 2);
    }
#undef L
    WM_LAUNCH_CHECK("wm_jpeg_bwd");
    return WM_OK;
}


static int dj_check(const char* name, const void* a, const void* b, int B, int H, int W, int rounding, float factor) {
    WM_REQUIRE(a && b, WM_E_BADARG, "%s: null tensor pointer", name);
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "%s: bad shape", name);
    WM_REQUIRE(H % 16 == 0 && W % 16 == 0, WM_E_SHAPE, "%s: H, W must be multiples of 16 (got %dx%d), like the reference's block split", name, H, W);
    WM_REQUIRE(rounding >= 0 && rounding <= 2, WM_E_BADARG, "%s: rounding must be 0 (round), 1 (round_only_at_0) or 2 (diff_round)", name);
    WM_REQUIRE(factor > 0.f, WM_E_BADARG, "%s: factor must be positive", name);
    return WM_OK;
}

static void dj_tables(float factor, DiffTables& tb) {
    // utils/JPEG.py:96-108: the standard tables, TRANSPOSED; chroma = 99 with a transposed 4x4 corner
    static const float lum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                  14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const float c4[16] = {17, 18, 24, 47, 18, 21, 26, 66, 24, 26, 56, 99, 47, 66, 99, 99};
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            tb.t[u * 8 + v] = lum[v * 8 + u] * factor;
            tb.t[64 + u * 8 + v] = ((u < 4 && v < 4) ? c4[v * 4 + u] : 99.f) * factor;
        }
}

extern "C" int wm_diffjpeg_fwd(const float* x, float* y, int B, int H, int W, int rounding, float factor, void* stream) {
    int rc = dj_check("wm_diffjpeg_fwd", x, y, B, H, W, rounding, factor);
    if (rc) return rc;
    DiffTables tb;
    dj_tables(factor, tb);
    const long waves = (long)B * (H / 16) * ((W + 63) / 64);
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (rounding == 0) hipLaunchKernelGGL(diffjpeg_fwd_kernel<0>, grid, block, 0, s, x, y, B, H, W, tb);
    else if (rounding == 1) hipLaunchKernelGGL(diffjpeg_fwd_kernel<1>, grid, block, 0, s, x, y, B, H, W, tb);
    else hipLaunchKernelGGL(diffjpeg_fwd_kernel<2>, grid, block, 0, s, x, y, B, H, W, tb);
    WM_LAUNCH_CHECK("wm_diffjpeg_fwd");
    return WM_OK;
}

extern "C" int wm_diffjpeg_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int rounding, float factor,
                               void* stream) {
    int rc = dj_check("wm_diffjpeg_bwd", gy, gx, B, H, W, rounding, factor);
    if (rc) return rc;
    WM_REQUIRE(x, WM_E_BADARG, "wm_diffjpeg_bwd: x required (coefficients and the clamp mask are recomputed)");
    DiffTables tb;
    dj_tables(factor, tb);
    const long waves = (long)B * (H / 16) * ((W + 63) / 64);
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (rounding == 0) hipLaunchKernelGGL(diffjpeg_bwd_kernel<0>, grid, block, 0, s, x, gy, gx, B, H, W, tb);
    else if (rounding == 1) hipLaunchKernelGGL(diffjpeg_bwd_kernel<1>, grid, block, 0, s, x, gy, gx, B, H, W, tb);
    else hipLaunchKernelGGL(diffjpeg_bwd_kernel<2>, grid, block, 0, s, x, gy, gx, B, H, W, tb);
    WM_LAUNCH_CHECK("wm_diffjpeg_bwd");
    return WM_OK;
}
