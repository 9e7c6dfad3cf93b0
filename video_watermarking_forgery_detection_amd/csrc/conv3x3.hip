// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on MFMA (gfx950), forward + dgrad.
//
// Reference op: nn.Conv2d(cin, cout, 3, 1, padding=1) in hidden_models/conv_bn_relu.py:11 and
// network/UNet.py:67-97 (followed there by BatchNorm2d + ReLU, which this kernel helps fuse:
// the *consumer* applies relu(scale*x+shift) of the producer while staging its input tile, and
// the *producer* emits the per-channel sum / sum-of-squares of its f32 accumulators).
//
// GEMM view: M = B*H*W output pixels, N = Cout, K = 9*Cin.  NHWC activations.
//   workgroup  = 256 threads (4 waves) -> 16x16 output pixels x BN output channels
//   wave       = 4 tile rows (64 pixels) x BN channels = 2 x (BN/32) MFMA 32x32 accumulators
//   K loop     = Cin chunks of CK channels; per chunk the 18x18xCK input halo tile and the
//                [9][BN][CK] weight slab are staged into LDS once and reused by all 9 taps
//                (each input element is fetched from HBM once per Cout tile, +27% halo).
//   bf16 path  : v_mfma_f32_32x32x16_bf16, CK = 32, f32 accumulate
//   f32 path   : v_mfma_f32_32x32x2_f32 (exact f32 FMA chain), CK = 16 -- the parity path
// LDS rows (one pixel's CK channels / one filter row) are padded by 16 B to an 80-byte stride:
// 16 consecutive pixels then hit 16 distinct 16-byte bank slots for ds_read_b128.
// The epilogue restages the accumulators through LDS so the global stores are 16 B per lane
// and contiguous over a pixel's channels (128 B for 64 bf16 channels).
#include <stdlib.h>
#include <type_traits>
#include "wm_common.h"

// streamed-filter wave-specialised kernel for bf16 Cin > 64 (conv3x3_stream.hip)
int wm_conv3x3_stream_supported(int Cin, int CoutP);
int wm_conv3x3_stream_nparts(int B, int H, int W);
#define WM_DECL_STREAM(sfx)                                                                                                        \
    int wm_launch_conv3x3_stream##sfx(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale, \
                                      const float* in_shift, void* y, int ldy, float* stat, int B, int H, int W, int Cin, int CoutP, hipStream_t s)
WM_DECL_STREAM(_bf16);
WM_DECL_STREAM(_f16);
// persistent wave-specialised kernel for bf16 Cin in {64,32,16}, Cout in {64,32} (conv3x3_ws.hip)
#define WM_DECL_WS(sfx)                                                                                                                     \
    int wm_launch_conv3x3_ws##sfx(const void* x, int ldx, int Cin, int CoutP, const void* wp, const float* bias, int nbias,                  \
                                  const float* in_scale, const float* in_shift, void* y, float* stat, int B, int H, int W, int wgs,          \
                                  int tiles_per_wg, hipStream_t s, int reverse, const float* bw_stats4 = nullptr, int bw_ld = 0,            \
                                  const float* bw_coef = nullptr, const float* bw_gvec = nullptr, const void* ry = nullptr,                 \
                                  const float* r_scale = nullptr, const float* r_shift = nullptr, const void* ay = nullptr, void* dy_out = nullptr, \
                                  const void* addend = nullptr, int act = 0, int ldy = 0)
WM_DECL_WS(_bf16);
WM_DECL_WS(_f16);
// the two compilations of conv3x3_ws.hip, by activation dtype (WM_BF16 / WM_F16)
template <typename... A> static inline int wm_launch_conv3x3_ws(int dtype, A... args) {
    return dtype == WM_F16 ? wm_launch_conv3x3_ws_f16(args...) : wm_launch_conv3x3_ws_bf16(args...);
}

namespace {

constexpr int TH = 16, TW = 16;
constexpr int HH = TH + 2, HW = TW + 2;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int CK = 32;   // channels per K chunk
    static constexpr int VE = 8;    // elements per 16-byte vector
    static constexpr int PS = 40;   // LDS row stride in elements (80 B)
};
template <> struct Cfg<f16_t> : Cfg<bf16_t> {};
template <> struct Cfg<float> {
    static constexpr int CK = 16;
    static constexpr int VE = 4;
    static constexpr int PS = 20;   // 80 B
};

template <typename T>
struct ConvArgs {
    const T* x;
    int ldx;
    const T* wp;          // [9][CoutP][Cin]
    const float* bias;    // [nbias] or null
    int nbias;
    const float* in_scale;
    const float* in_shift;
    T* y;
    int ldy;
    float* stat;          // [gridDim.x][2][CoutP] or null
    int B, H, W, Cin, CoutP;
    int tilesX, tilesY;
};

template <typename T> __device__ __forceinline__ void zero_vec(vec16<T>& v) {
#pragma unroll
    for (int i = 0; i < vec16<T>::N; ++i) v.set(i, 0.f);
}

template <typename T, int BN, bool XFORM>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs<T> a) {
    constexpr int CK = Cfg<T>::CK, VE = Cfg<T>::VE, PS = Cfg<T>::PS;
    constexpr int NF = BN / 32;
    constexpr int A_ELEMS = HH * HW * PS;
    constexpr int B_ELEMS = 9 * BN * PS;
    constexpr int OPS = BN + 16 / (int)sizeof(T);          // output staging row stride (elements)
    constexpr int MAIN_BYTES = (A_ELEMS + B_ELEMS) * (int)sizeof(T);
    constexpr int OUT_BYTES = TH * TW * OPS * (int)sizeof(T);
    constexpr int LDS_BYTES = MAIN_BYTES > OUT_BYTES ? MAIN_BYTES : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + 4 * 2 * BN * 4];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + A_ELEMS;
    T* sOut = reinterpret_cast<T*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + LDS_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile assignment: workgroups b and b+8 share an XCD (round-robin dispatch); XCD x takes the
    // contiguous tiles [x*G/8, (x+1)*G/8) so that neighbouring tiles share their halo in one L2
    const int G = gridDim.x;
    int t = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int txi = t % a.tilesX; t /= a.tilesX;
    const int tyi = t % a.tilesY; t /= a.tilesY;
    const int b = t;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[2][NF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int vec = tid & 3;  // this thread's 16-byte group inside a CK chunk (stride 256 keeps it fixed)
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        const int kvalid = min(CK, a.Cin - c0);
        const int cbase = c0 + vec * VE;
        const bool cok = vec * VE < kvalid;
        float sc[VE], sh[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) { sc[i] = 0.f; sh[i] = 0.f; }
        if (XFORM && cok) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { sc[i] = a.in_scale[cbase + i]; sh[i] = a.in_shift[cbase + i]; }
        }
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the input halo tile: 18x18 pixels x CK channels
        const int cload = cok ? cbase : 0;  // loads are unconditional (clamped address); validity is applied afterwards
#pragma unroll
        for (int it = 0; it < (HH * HW * 4 + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int pix = min(i >> 2, HH * HW - 1);
            const int py = pix / HW, px = pix - py * HW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
            const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            vec16<T> v = *reinterpret_cast<const vec16<T>*>(a.x + ((size_t)(b * a.H + gyc) * a.W + gxc) * a.ldx + cload);
            const bool inb = cok && gy == gyc && gx == gxc;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float f = v.get(e);
                if (XFORM) f = fmaxf(sc[e] * f + sh[e], 0.f);
                v.set(e, inb ? f : 0.f);  // zero padding is applied AFTER the fused BN+ReLU, as the reference pads the activated map
            }
            if (i < HH * HW * 4) *reinterpret_cast<vec16<T>*>(sA + pix * PS + vec * VE) = v;
        }
        // ---- stage the weight slab [9][BN][CK]
#pragma unroll
        for (int it = 0; it < (9 * BN * 4 + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int row = min(i >> 2, 9 * BN - 1);  // tap*BN + n
            const int tap = row / BN, n = row - tap * BN;
            vec16<T> v = *reinterpret_cast<const vec16<T>*>(a.wp + ((size_t)tap * a.CoutP + n0 + n) * a.Cin + cload);
            if (!cok) zero_vec(v);
            if (i < 9 * BN * 4) *reinterpret_cast<vec16<T>*>(sB + row * PS + vec * VE) = v;
        }
        __syncthreads();
        // ---- 9 taps x K steps of MFMA
        if constexpr (sizeof(T) == 2) {
            const int ksteps = (kvalid + 15) >> 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                for (int ks = 0; ks < ksteps; ++ks) {
                    typedef typename h16<T>::x8 hx8;
                    hx8 af[2], bfr[NF];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const int py = wave * 4 + mf * 2 + (r >> 4), px = r & 15;
                        af[mf] = *reinterpret_cast<const hx8*>(sA + ((py + kh) * HW + px + kw) * PS + ks * 16 + h * 8);
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        bfr[nf] = *reinterpret_cast<const hx8*>(sB + (tap * BN + nf * 32 + r) * PS + ks * 16 + h * 8);
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = h16<T>::mfma32(af[mf], bfr[nf], acc[mf][nf]);
                }
            }
        } else {
            const int ksteps = kvalid >> 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                for (int ks = 0; ks < ksteps; ++ks) {
                    float af[2], bfr[NF];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const int py = wave * 4 + mf * 2 + (r >> 4), px = r & 15;
                        af[mf] = sA[((py + kh) * HW + px + kw) * PS + ks * 2 + h];
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) bfr[nf] = sB[(tap * BN + nf * 32 + r) * PS + ks * 2 + h];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mf], bfr[nf], acc[mf][nf], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // all waves done with sA/sB before the output tile overwrites them

    // ---- epilogue: bias, batch statistics, restage through LDS, coalesced store
    // accumulator element i of lane (r,h): pixel row-in-fragment = (i&3) + 8*(i>>2) + 4*h, column n = r
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const float bv = (a.bias && n0 + nf * 32 + r < a.nbias) ? a.bias[n0 + nf * 32 + r] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int prow = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int py = wave * 4 + mf * 2 + (prow >> 4), px = prow & 15;
                const float v = acc[mf][nf][i] + bv;
                const bool ok = (ty0 + py < a.H) && (tx0 + px < a.W);
                if (ok) { s1 += v; s2 += v * v; }
                sOut[(py * TW + px) * OPS + nf * 32 + r] = from_f32<T>(v);
            }
        }
        if (a.stat) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                sRed[(wave * 2 + 0) * BN + nf * 32 + r] = s1;
                sRed[(wave * 2 + 1) * BN + nf * 32 + r] = s2;
            }
        }
    }
    __syncthreads();
    if (a.stat && tid < 2 * BN) {
        const int which = tid / BN, n = tid - which * BN;
        const float s = sRed[(0 * 2 + which) * BN + n] + sRed[(1 * 2 + which) * BN + n] +
                        sRed[(2 * 2 + which) * BN + n] + sRed[(3 * 2 + which) * BN + n];
        a.stat[((size_t)blockIdx.x * 2 + which) * a.CoutP + n0 + n] = s;
    }
    constexpr int VPP = BN / VE;  // 16-byte vectors per pixel
    for (int i = tid; i < TH * TW * VPP; i += 256) {
        const int pix = i / VPP, vv = i - pix * VPP;
        const int py = pix / TW, px = pix - py * TW;
        const int gy = ty0 + py, gx = tx0 + px;
        if (gy < a.H && gx < a.W) {
            const vec16<T> v = *reinterpret_cast<const vec16<T>*>(sOut + pix * OPS + vv * VE);
            *reinterpret_cast<vec16<T>*>(a.y + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldy + n0 + vv * VE) = v;
        }
    }
}

#ifndef WM_MAX_WGS
#define WM_MAX_WGS 256           // (tools/build_variant.sh -DWM_MAX_WGS=128: half-chip grids, the two-chains-side-by-side experiment of round 4)
#endif
constexpr int WS_MAX_WGS = WM_MAX_WGS;  // one persistent workgroup per CU
inline bool is16(int dtype) { return dtype == WM_BF16 || dtype == WM_F16; }   // the two 16-bit activation dtypes share every MFMA kernel
inline bool use_ws(int Cin, int CoutP, int dtype) {
    static const bool off = WM_ENV_FLAG("WM_NO_WS");  // diagnostic knob (debug build): force the generic kernel
    return !off && is16(dtype) && (((Cin == 64 || Cin == 32 || Cin == 16) && CoutP == 64) || (Cin == 64 && CoutP == 32));
}  // + ldy == CoutP (a dense output tensor)
inline bool use_stream(int Cin, int CoutP, int dtype) {
    static const bool off = WM_ENV_FLAG("WM_NO_STREAM");  // diagnostic knob (debug build): force the generic kernel
    return !off && is16(dtype) && !use_ws(Cin, CoutP, dtype) && wm_conv3x3_stream_supported(Cin, CoutP);
}
inline int ws_tiles_per_wg(int ntiles) { return (ntiles + WS_MAX_WGS - 1) / WS_MAX_WGS; }
inline int ws_wgs(int ntiles) { const int per = ws_tiles_per_wg(ntiles); return (ntiles + per - 1) / per; }

// ---- weight packing: PyTorch [Cout][Cin][3][3] f32 -> [9][RowsP][ColsP] T
//   transpose == 0: rows = Cout, cols = packed Cin (perm applied), tap = kh*3+kw        (forward)
//   transpose == 1: rows = packed Cin, cols = Cout, tap = (2-kh)*3+(2-kw)               (dgrad)
struct PermArg { int p[128]; };
template <typename T>
__global__ void pack_w3x3_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int CoutP,
                                 int CinP, int transpose, PermArg perm, int has_perm) {
    const int rowsP = transpose ? CinP : CoutP, colsP = transpose ? CoutP : CinP;
    const size_t total = (size_t)9 * rowsP * colsP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % colsP);
        const int row = (int)((i / colsP) % rowsP);
        const int tap = (int)(i / ((size_t)colsP * rowsP));
        const int co = transpose ? col : row;
        const int cip = transpose ? row : col;  // packed input-channel index
        int kh = tap / 3, kw = tap % 3;
        if (transpose) { kh = 2 - kh; kw = 2 - kw; }
        float v = 0.f;
        if (co < Cout) {
            // find the reference input channel stored at packed position cip
            int ci = -1;
            if (!has_perm) ci = cip < Cin ? cip : -1;
            else {
                for (int q = 0; q < Cin; ++q) if (perm.p[q] == cip) { ci = q; break; }
            }
            if (ci >= 0) v = w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw];
        }
        wp[i] = from_f32<T>(v);
    }
}

// batched form: one launch packs every conv of a network (blockIdx.y = job).  Jobs live in device memory.
struct PackJob {
    const float* w; void* wp; const int* perm;   // perm: device int[Cin] or null
    int Cout, Cin, CoutP, CinP, transpose, pad;
};
template <typename T>
__global__ void pack_w3x3_batch_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int rowsP = j.transpose ? j.CinP : j.CoutP, colsP = j.transpose ? j.CoutP : j.CinP;
    const size_t total = (size_t)9 * rowsP * colsP;
    T* wp = (T*)j.wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % colsP);
        const int row = (int)((i / colsP) % rowsP);
        const int tap = (int)(i / ((size_t)colsP * rowsP));
        const int co = j.transpose ? col : row;
        const int cip = j.transpose ? row : col;
        int kh = tap / 3, kw = tap % 3;
        if (j.transpose) { kh = 2 - kh; kw = 2 - kw; }
        float v = 0.f;
        if (co < j.Cout) {
            int ci = -1;
            if (!j.perm) ci = cip < j.Cin ? cip : -1;
            else {
                for (int q = 0; q < j.Cin; ++q) if (j.perm[q] == cip) { ci = q; break; }
            }
            if (ci >= 0) v = j.w[(((size_t)co * j.Cin + ci) * 3 + kh) * 3 + kw];
        }
        wp[i] = from_f32<T>(v);
    }
}

template <typename T>
int launch_conv(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                const float* in_shift, void* y, int ldy, float* stat, int B, int H, int W, int Cin, int CoutP,
                hipStream_t s, int reverse) {
    ConvArgs<T> a;
    a.x = (const T*)x; a.ldx = ldx; a.wp = (const T*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale; a.in_shift = in_shift;
    a.y = (T*)y; a.ldy = ldy; a.stat = stat; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.CoutP = CoutP;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH);
    const bool xf = in_scale != nullptr;
    if constexpr (sizeof(T) == 2) {
        if (use_ws(Cin, CoutP, wm_dtype<T>::id)) {
            const int ntiles = B * a.tilesX * a.tilesY;
            const int per = ws_tiles_per_wg(ntiles);
            return wm_launch_conv3x3_ws(wm_dtype<T>::id, x, ldx, Cin, CoutP, wp, bias, nbias, in_scale, in_shift, y, stat, B, H, W, ws_wgs(ntiles), per, s, reverse);
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (use_stream(Cin, CoutP, wm_dtype<T>::id))
            return (wm_dtype<T>::id == WM_F16 ? wm_launch_conv3x3_stream_f16 : wm_launch_conv3x3_stream_bf16)(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat, B, H, W, Cin, CoutP, s);
    }
    const int BN = (CoutP % 64 == 0) ? 64 : 32;
    dim3 grid((unsigned)(B * a.tilesX * a.tilesY), (unsigned)(CoutP / BN)), block(256);
    if (BN == 64) {
        if (xf) hipLaunchKernelGGL((conv3x3_kernel<T, 64, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 64, false>), grid, block, 0, s, a);
    } else {
        if (xf) hipLaunchKernelGGL((conv3x3_kernel<T, 32, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 32, false>), grid, block, 0, s, a);
    }
    return WM_OK;
}

}  // namespace

extern "C" int wm_conv3x3_nparts(int B, int H, int W, int Cin, int CoutP, int dtype) {
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    if (use_stream(Cin, CoutP, dtype)) return wm_conv3x3_stream_nparts(B, H, W);
    return use_ws(Cin, CoutP, dtype) ? ws_wgs(ntiles) : ntiles;
}

extern "C" int wm_conv3x3_fwd(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                              const float* in_shift, void* y, int ldy, float* stat_partials, int B, int H, int W,
                              int Cin, int CoutP, int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(x && wp && y, WM_E_BADARG, "wm_conv3x3_fwd: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && CoutP > 0, WM_E_BADARG, "wm_conv3x3_fwd: bad shape");
    WM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), WM_E_BADARG, "wm_conv3x3_fwd: in_scale/in_shift must come together");
    WM_REQUIRE(dtype == WM_F32 || is16(dtype), WM_E_BADARG, "wm_conv3x3_fwd: unsupported dtype %d", dtype);
    const int esz = is16(dtype) ? 2 : 4;
    const int cmul = is16(dtype) ? 16 : 4;
    WM_REQUIRE(Cin % cmul == 0, WM_E_SHAPE, "wm_conv3x3_fwd: Cin=%d must be a multiple of %d (pad the channels)", Cin, cmul);
    WM_REQUIRE(CoutP % 32 == 0, WM_E_SHAPE, "wm_conv3x3_fwd: CoutP=%d must be a multiple of 32", CoutP);
    WM_REQUIRE(ldx >= Cin && ldy >= CoutP && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, WM_E_SHAPE,
               "wm_conv3x3_fwd: pixel strides ldx=%d ldy=%d must cover the channels and be 16-byte multiples", ldx, ldy);
    WM_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y) & 15) == 0, WM_E_SHAPE, "wm_conv3x3_fwd: pointers must be 16-byte aligned");
    WM_REQUIRE(!use_ws(Cin, CoutP, dtype) || ldy == CoutP, WM_E_SHAPE,
               "wm_conv3x3_fwd: the persistent bf16 path writes a dense output (ldy must equal CoutP=%d, got %d)", CoutP, ldy);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WM_BF16) launch_conv<bf16_t>(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat_partials, B, H, W, Cin, CoutP, s, sweep_reverse ? 1 : 0);
    else if (dtype == WM_F16) launch_conv<f16_t>(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat_partials, B, H, W, Cin, CoutP, s, sweep_reverse ? 1 : 0);
    else launch_conv<float>(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat_partials, B, H, W, Cin, CoutP, s, 0);
    WM_LAUNCH_CHECK("wm_conv3x3_fwd");
    return WM_OK;
}

// forward 64 -> 64 conv (16-bit dtypes, dense tensors) whose epilogue adds a second tensor before the BatchNorm statistics:
// y = conv3x3(relu(in_scale*x + in_shift), wp) + addend
extern "C" int wm_conv3x3_fwd_addin(const void* x, const void* wp, const float* in_scale, const float* in_shift, const void* addend, void* y,
                                    float* stat_partials, int B, int H, int W, int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(x && wp && in_scale && in_shift && addend && y && stat_partials, WM_E_BADARG, "wm_conv3x3_fwd_addin: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && is16(dtype), WM_E_BADARG, "wm_conv3x3_fwd_addin: bad shape / dtype (WM_BF16 or WM_F16)");
    WM_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)addend) & 15) == 0, WM_E_SHAPE, "wm_conv3x3_fwd_addin: pointers must be 16-byte aligned");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    const int rc = wm_launch_conv3x3_ws(dtype, x, 64, 64, 64, wp, nullptr, 0, in_scale, in_shift, y, stat_partials, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, sweep_reverse ? 1 : 0, nullptr, 0, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, addend);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_fwd_addin: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_fwd_addin");
    return WM_OK;
}

// input gradient of a ConvBNRelu whose output was globally pooled, with the BatchNorm-backward apply pass fused: the
// kernel reads the layer's raw conv output y and forms dy from (gvec[b], y, stats4, coef) while staging each tile
extern "C" int wm_conv3x3_dgrad_gvfused(const void* y, int ldy, int CoutY, const void* wpt, const float* gvec, const float* stats4,
                                        const float* coef, void* dx, int B, int H, int W, int CinP, int dtype, void* stream) {
    WM_REQUIRE(y && wpt && gvec && stats4 && coef && dx, WM_E_BADARG, "wm_conv3x3_dgrad_gvfused: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_dgrad_gvfused: bad shape");
    WM_REQUIRE(is16(dtype) && (CoutY == 64 || CoutY == 32) && CinP == 64 && use_ws(CoutY, CinP, dtype), WM_E_SHAPE,
               "wm_conv3x3_dgrad_gvfused: unsupported shape CoutY=%d CinP=%d dtype=%d", CoutY, CinP, dtype);
    WM_REQUIRE(ldy >= CoutY && ldy % 8 == 0 && (((uintptr_t)y | (uintptr_t)wpt | (uintptr_t)dx) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_dgrad_gvfused: bad stride / alignment");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    const int rc = wm_launch_conv3x3_ws(dtype, y, ldy, CoutY, CinP, wpt, nullptr, 0, nullptr, nullptr, dx, nullptr, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, 0, stats4, CoutY, coef, gvec);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_dgrad_gvfused: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_dgrad_gvfused");
    return WM_OK;
}

// input gradient whose epilogue also reduces the BatchNorm-backward sums of the layer it feeds.  src: dy [B,H,W,lds] of this
// layer, or (gvec != NULL) this layer's raw output y with the apply pass fused as in wm_conv3x3_dgrad_gvfused.
WM_KNOB_ON(g_bwdst, "WM_NO_BWDST");
WM_KNOB_SETTER(wm_debug_bwdst, g_bwdst)   // A/B knob (tools/ab_step.py, debug build only)
extern "C" int wm_conv3x3_dgrad_bwdstats_supported(int CoutY, int CinP, int dtype) {
    return (g_bwdst && is16(dtype) && (CoutY == 64 || CoutY == 32) && CinP == 64 && use_ws(CoutY, CinP, dtype)) ? 1 : 0;
}
extern "C" int wm_conv3x3_dgrad_bwdstats(const void* src, int lds, int CoutY, const void* wpt, const float* gvec, const float* stats4,
                                         const float* coef, const void* ry, const float* r_scale, const float* r_shift, void* dx,
                                         float* partials, int B, int H, int W, int CinP, int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(src && wpt && ry && r_scale && r_shift && dx && partials, WM_E_BADARG, "wm_conv3x3_dgrad_bwdstats: null pointer");
    WM_REQUIRE((gvec == nullptr) == (stats4 == nullptr) && (gvec == nullptr) == (coef == nullptr), WM_E_BADARG,
               "wm_conv3x3_dgrad_bwdstats: gvec, stats4 and coef come together");
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_dgrad_bwdstats: bad shape");
    WM_REQUIRE(wm_conv3x3_dgrad_bwdstats_supported(CoutY, CinP, dtype), WM_E_SHAPE,
               "wm_conv3x3_dgrad_bwdstats: unsupported shape CoutY=%d CinP=%d dtype=%d", CoutY, CinP, dtype);
    WM_REQUIRE(lds >= CoutY && lds % 8 == 0 && (((uintptr_t)src | (uintptr_t)wpt | (uintptr_t)dx | (uintptr_t)ry) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_dgrad_bwdstats: bad stride / alignment");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    const int rc = wm_launch_conv3x3_ws(dtype, src, lds, CoutY, CinP, wpt, nullptr, 0, nullptr, nullptr, dx, partials, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, sweep_reverse ? 1 : 0, stats4, CoutY, coef, gvec, ry, r_scale, r_shift);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_dgrad_bwdstats: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_dgrad_bwdstats");
    return WM_OK;
}

// input gradient of an ordinary 64 -> 64 ConvBNRelu with the BatchNorm-backward APPLY pass fused: reads g and the layer's raw
// output y, writes dy (for the weight gradient that follows) and dx; optionally reduces the feeding layer's sums as above.
WM_KNOB_ON(g_applyfuse, "WM_NO_APPLY_FUSE");
WM_KNOB_SETTER(wm_debug_apply_fuse, g_applyfuse)   // A/B knob (tools/ab_step.py, debug build only)
extern "C" int wm_conv3x3_dgrad_applyfused_supported(int CoutY, int CinP, int dtype) {
    return (g_applyfuse && is16(dtype) && CoutY == 64 && (CinP == 64 || CinP == 32) && use_ws(CoutY, CinP, dtype)) ? 1 : 0;
}
extern "C" int wm_conv3x3_dgrad_applyfused(const void* g, const void* y, const float* stats4, const float* coef, const void* wpt,
                                           void* dy_out, void* dx, const void* ry, const float* r_scale, const float* r_shift,
                                           float* partials, int B, int H, int W, int CinP, int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(g && y && stats4 && coef && wpt && dx, WM_E_BADARG, "wm_conv3x3_dgrad_applyfused: null pointer");   // (dy_out may be NULL)
    WM_REQUIRE((ry == nullptr) == (partials == nullptr) && (ry == nullptr) == (r_scale == nullptr) && (ry == nullptr) == (r_shift == nullptr),
               WM_E_BADARG, "wm_conv3x3_dgrad_applyfused: ry, r_scale, r_shift and partials come together");
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_dgrad_applyfused: bad shape");
    WM_REQUIRE(wm_conv3x3_dgrad_applyfused_supported(64, CinP, dtype) && (CinP == 64 || !ry), WM_E_SHAPE,
               "wm_conv3x3_dgrad_applyfused: unsupported CinP=%d dtype=%d (the feeding layer's sums need CinP = 64)", CinP, dtype);
    WM_REQUIRE((((uintptr_t)g | (uintptr_t)y | (uintptr_t)wpt | (uintptr_t)dy_out | (uintptr_t)dx | (uintptr_t)ry) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_dgrad_applyfused: pointers must be 16-byte aligned");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    const int rc = wm_launch_conv3x3_ws(dtype, g, 64, 64, CinP, wpt, nullptr, 0, nullptr, nullptr, dx, partials, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, sweep_reverse ? 1 : 0, stats4, 64, coef, nullptr, ry, r_scale, r_shift, y, dy_out);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_dgrad_applyfused: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_dgrad_applyfused");
    return WM_OK;
}

// conv + bias + ELU as one launch, and its backward's input-gradient half with the ELU derivative, gz and the bias gradient's partial sums
// formed while the tiles are staged (rows f1 / f2: the coupling subnets, models/invertible_net.py:326-366)
extern "C" int wm_conv3x3_fwd_elu_supported(int Cin, int CoutP, int dtype) {
    return (is16(dtype) && CoutP == 64 && (Cin == 64 || Cin == 32 || Cin == 16) && use_ws(Cin, CoutP, dtype)) ? 1 : 0;
}
extern "C" int wm_conv3x3_fwd_elu(const void* x, int ldx, const void* wp, const float* bias, int nbias, void* out, int B, int H, int W, int Cin,
                                  int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(x && wp && out, WM_E_BADARG, "wm_conv3x3_fwd_elu: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_fwd_elu: bad shape");
    WM_REQUIRE(wm_conv3x3_fwd_elu_supported(Cin, 64, dtype), WM_E_SHAPE, "wm_conv3x3_fwd_elu: unsupported Cin=%d dtype=%d (16-bit dtypes, Cin in {16, 32, 64}, 64 output channels)", Cin, dtype);
    WM_REQUIRE(ldx >= Cin && (ldx * 2) % 16 == 0 && nbias >= 0 && nbias <= 64, WM_E_SHAPE, "wm_conv3x3_fwd_elu: bad ldx=%d / nbias=%d", ldx, nbias);
    WM_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)out) & 15) == 0, WM_E_SHAPE, "wm_conv3x3_fwd_elu: pointers must be 16-byte aligned");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    const int rc = wm_launch_conv3x3_ws(dtype, x, ldx, Cin, 64, wp, bias, nbias, nullptr, nullptr, out, nullptr, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, sweep_reverse ? 1 : 0, nullptr, 0, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_fwd_elu: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_fwd_elu");
    return WM_OK;
}
extern "C" int wm_conv3x3_dgrad_elufused_supported(int CinP, int dtype) {   // (CinP = 16: the 32-channel consumers, half of them stored)
    return (is16(dtype) && (CinP == 64 || CinP == 32 || CinP == 16) && use_ws(64, CinP == 16 ? 32 : CinP, dtype)) ? 1 : 0;
}
extern "C" int wm_conv3x3_dgrad_elufused_nparts(int B, int H, int W) { return ws_wgs(B * wm_cdiv(H, TH) * wm_cdiv(W, TW)); }
extern "C" int wm_conv3x3_dgrad_elufused(const void* g, const void* out, const void* wpt, void* dx, void* gz_out, float* bias_partials, int B,
                                         int H, int W, int CinP, int dtype, int sweep_reverse, void* stream) {
    WM_REQUIRE(g && out && wpt && dx && bias_partials, WM_E_BADARG, "wm_conv3x3_dgrad_elufused: null pointer");   // (gz_out may be NULL: no weight gradient wanted)
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_dgrad_elufused: bad shape");
    WM_REQUIRE(wm_conv3x3_dgrad_elufused_supported(CinP, dtype), WM_E_SHAPE, "wm_conv3x3_dgrad_elufused: unsupported CinP=%d dtype=%d", CinP, dtype);
    WM_REQUIRE((((uintptr_t)g | (uintptr_t)out | (uintptr_t)wpt | (uintptr_t)dx | (uintptr_t)gz_out) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_dgrad_elufused: pointers must be 16-byte aligned");
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    // CinP = 16: wpt is packed with 32 rows (the upper 16 zero), dx keeps its 16-channel stride
    const int rc = wm_launch_conv3x3_ws(dtype, g, 64, 64, CinP == 16 ? 32 : CinP, wpt, nullptr, 0, nullptr, nullptr, dx, bias_partials, B, H, W, ws_wgs(ntiles),
                                        ws_tiles_per_wg(ntiles), (hipStream_t)stream, sweep_reverse ? 1 : 0, nullptr, 0, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, out, gz_out, nullptr, 1, CinP);
    WM_REQUIRE(rc == WM_OK, WM_E_SHAPE, "wm_conv3x3_dgrad_elufused: no kernel for this shape");
    WM_LAUNCH_CHECK("wm_conv3x3_dgrad_elufused");
    return WM_OK;
}

extern "C" int wm_pack_w3x3(const float* w, void* wp, int Cout, int Cin, int CoutP, int CinP, const int* perm,
                            int transpose, int dtype, void* stream) {
    WM_REQUIRE(w && wp, WM_E_BADARG, "wm_pack_w3x3: null pointer");
    WM_REQUIRE(Cout > 0 && Cin > 0 && CoutP >= Cout && (perm || CinP >= Cin), WM_E_BADARG, "wm_pack_w3x3: bad shape");
    WM_REQUIRE(!perm || Cin <= 128, WM_E_SHAPE, "wm_pack_w3x3: channel permutation supports Cin <= 128");
    PermArg pa;
    for (int i = 0; i < 128; ++i) pa.p[i] = -1;
    if (perm)
        for (int i = 0; i < Cin; ++i) {
            // a packed position >= CinP means "this input channel is dropped" (dgrad of a channel slice)
            WM_REQUIRE(perm[i] >= 0, WM_E_BADARG, "wm_pack_w3x3: perm[%d]=%d out of range", i, perm[i]);
            pa.p[i] = perm[i];
        }
    const size_t total = (size_t)9 * CoutP * CinP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WM_BF16)
        hipLaunchKernelGGL(pack_w3x3_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)wp, Cout, Cin, CoutP, CinP, transpose, pa, perm ? 1 : 0);
    else if (dtype == WM_F16)
        hipLaunchKernelGGL(pack_w3x3_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, w, (f16_t*)wp, Cout, Cin, CoutP, CinP, transpose, pa, perm ? 1 : 0);
    else if (dtype == WM_F32)
        hipLaunchKernelGGL(pack_w3x3_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)wp, Cout, Cin, CoutP, CinP, transpose, pa, perm ? 1 : 0);
    else { wm_set_error("wm_pack_w3x3: unsupported dtype %d", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_pack_w3x3");
    return WM_OK;
}

extern "C" int wm_pack_w3x3_batch(const void* jobs_dev, int njobs, size_t max_elems, int dtype, void* stream) {
    WM_REQUIRE(jobs_dev && njobs > 0 && max_elems > 0, WM_E_BADARG, "wm_pack_w3x3_batch: bad arguments");
    static_assert(sizeof(PackJob) == 48, "PackJob layout is part of the ABI (wm_hip.h)");
    int blocks = (int)((max_elems + 255) / 256);   // per job (grid.y = jobs); small jobs leave their surplus blocks idle
    if (blocks > 512) blocks = 512;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks, (unsigned)njobs);
    if (dtype == WM_BF16) hipLaunchKernelGGL(pack_w3x3_batch_kernel<bf16_t>, grid, dim3(256), 0, s, (const PackJob*)jobs_dev);
    else if (dtype == WM_F16) hipLaunchKernelGGL(pack_w3x3_batch_kernel<f16_t>, grid, dim3(256), 0, s, (const PackJob*)jobs_dev);
    else if (dtype == WM_F32) hipLaunchKernelGGL(pack_w3x3_batch_kernel<float>, grid, dim3(256), 0, s, (const PackJob*)jobs_dev);
    else { wm_set_error("wm_pack_w3x3_batch: unsupported dtype %d", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_pack_w3x3_batch");
    return WM_OK;
}
