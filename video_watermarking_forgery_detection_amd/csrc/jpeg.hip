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

template <int MODE, int SUB>
__global__ __launch_bounds__(256) void jpeg_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int B, int H, int W, JpegTables tb) {
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
    int rc = check_args("wm_jpeg_fwd", x, y, B, H, W, mode, tables, subsample);
    if (rc) return rc;
    JpegTables tb;
    for (int i = 0; i < 128; ++i) tb.t[i] = tables ? tables[i] : 1.f;
    const dim3 grid(jpeg_grid(B, H, W)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(M, S) hipLaunchKernelGGL((jpeg_fwd_kernel<M, S>), grid, block, 0, s, x, y, B, H, W, tb)
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
