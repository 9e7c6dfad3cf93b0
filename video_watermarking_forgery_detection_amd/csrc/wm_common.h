// Internal helpers shared by the gfx950 kernels of libwm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wm_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (thread-local text, no exceptions across the ABI)
void wm_set_error(const char* fmt, ...);
#define WM_REQUIRE(cond, code, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            wm_set_error(__VA_ARGS__);              \
            return (code);                          \
        }                                           \
    } while (0)
#define WM_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            wm_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return WM_E_HIP;                                                         \
        }                                                                            \
    } while (0)

// ---- A/B knobs.  The release library (lib/libwm_hip.so) holds NO mutable global state and reads no environment variable: every
// knob below is a compile-time constant there.  The -DWM_DEBUG build (lib/libwm_hip_dbg.so: tools/ab_step.py and the tests that
// compare a fused kernel with its unfused form) turns them into process-global switches with wm_debug_* setters.
#ifdef WM_DEBUG
#include <stdlib.h>
#define WM_KNOB_ON(var, envname) static int var = getenv(envname) ? 0 : 1          /* default on; the variable turns it off */
#define WM_KNOB_INT(var, envname, dflt) static int var = getenv(envname) ? atoi(getenv(envname)) : (dflt)
#define WM_KNOB_SETTER(fn, var) extern "C" void fn(int v) { var = v; }
#define WM_ENV_FLAG(envname) (getenv(envname) != nullptr)
#else
#define WM_KNOB_ON(var, envname) static constexpr int var = 1
#define WM_KNOB_INT(var, envname, dflt) static constexpr int var = (dflt)
#define WM_KNOB_SETTER(fn, var)
#define WM_ENV_FLAG(envname) false
#endif

static inline int wm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- BatchNorm+ReLU backward apply, folded (bf16 path).  With g one value per (sample, channel) -- a globally pooled layer:
//   dy = ca*(g*[z>0] - c1 - (y-mean)*invstd*c2),  z = scale*y + shift
//      = (z > 0 ? k3 + ca*g : k3) - k2*y,         k2 = ca*invstd*c2,  k3 = k2*mean - ca*c1
// two fused multiply-adds, a compare and a select per element.  The stand-alone apply pass (bn.hip) and the two kernels that
// fuse it (conv3x3_ws.hip, wgrad_ws.hip) all evaluate exactly these expressions, so their results are bit-identical.
__device__ __forceinline__ void wm_bn_fold(float mean, float invstd, float ca, float c1, float c2, float& k2, float& k3) {
    k2 = ca * (invstd * c2);
    k3 = __builtin_fmaf(k2, mean, -(ca * c1));
}
// g a tensor element:  dy = ca*(g*[z>0]) + (k3 - k2*y)
__device__ __forceinline__ float wm_bn_fold_dyg(float y, float g, float scale, float shift, float ca, float k2, float k3) {
    const float z = __builtin_fmaf(scale, y, shift);
    return __builtin_fmaf(ca, z > 0.f ? g : 0.f, __builtin_fmaf(-k2, y, k3));
}
__device__ __forceinline__ float wm_bn_fold_g(float ca, float g, float k3) { return __builtin_fmaf(ca, g, k3); }
__device__ __forceinline__ float wm_bn_fold_dy(float y, float scale, float shift, float k2, float k3, float k3g) {
    const float z = __builtin_fmaf(scale, y, shift);
    return __builtin_fmaf(-k2, y, z > 0.f ? k3g : k3);
}

// ---- BatchNorm-backward finalisation of 8 channels (one 256-thread workgroup: 8 channels x 32 row slices, double accumulate, fixed
// order).  partials [nparts][2][CP] -> dgamma, dbeta, coef[3][CP].  mean != nullptr: the second row holds sum(gz*y) instead of
// sum(gz*xhat) and xhat = (y-mean)*invstd is applied here.  Runs as bn_bwd_finalize_kernel (bn.hip) or as extra workgroups of the
// weight-gradient slab reduction (wgrad.hip).  sh: 2 x 32 x 8 doubles of LDS.
__device__ __forceinline__ void wm_bn_bwd_finalize_block(const WmBnBwdFin& j, int blk, double* sh, const float* pgv = nullptr,
                                                         const float* pn = nullptr, const float* ps = nullptr) {
    double (*s1)[8] = reinterpret_cast<double (*)[8]>(sh);
    double (*s2)[8] = reinterpret_cast<double (*)[8]>(sh + 32 * 8);
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blk * 8 + cl, CP = j.CP;
    double a1 = 0.0, a2 = 0.0;
    if (c < CP) {
#pragma unroll 8
        for (int p = sl; p < j.nparts; p += 32) {
            if (pgv) {   // a globally pooled layer: row p is sample p's (gv*N+, gv*S+), formed here from the pooled statistics
                const float gv = pgv[(size_t)p * CP + c];
                a1 += (double)(gv * pn[(size_t)p * CP + c]);
                a2 += (double)(gv * ps[(size_t)p * CP + c]);
            } else {
                a1 += (double)j.partials[((size_t)p * 2 + 0) * CP + c];
                a2 += (double)j.partials[((size_t)p * 2 + 1) * CP + c];
            }
        }
    }
    s1[sl][cl] = a1; s2[sl][cl] = a2;
    __syncthreads();
    if (sl == 0 && c < CP) {
        for (int k = 1; k < 32; ++k) { a1 += s1[k][cl]; a2 += s2[k][cl]; }
        if (j.mean && c < j.C) a2 = (a2 - (double)j.mean[c] * a1) * (double)j.invstd[c];
        if (c < j.C) {
            if (j.dbeta) j.dbeta[c] = (j.accumulate ? j.dbeta[c] : 0.f) + (float)a1;
            if (j.dgamma) j.dgamma[c] = (j.accumulate ? j.dgamma[c] : 0.f) + (float)a2;
            j.coef[c] = j.gamma[c] * j.invstd[c];
            j.coef[CP + c] = (float)(a1 / j.count);
            j.coef[2 * CP + c] = (float)(a2 / j.count);
        } else {
            j.coef[c] = 0.f; j.coef[CP + c] = 0.f; j.coef[2 * CP + c] = 0.f;
        }
    }
}

// ---- dtype helpers
template <typename T> struct wm_dtype;
template <> struct wm_dtype<float> { static constexpr int id = WM_F32; };
template <> struct wm_dtype<bf16_t> { static constexpr int id = WM_BF16; };
template <> struct wm_dtype<f16_t> { static constexpr int id = WM_F16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct vec16;
template <> struct vec16<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ float get(int i) const { return ((const float*)&v)[i]; }
    __device__ __forceinline__ void set(int i, float f) { ((float*)&v)[i] = f; }
};
template <> struct vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16_t)f; }
};

template <> struct vec16<f16_t> {
    static constexpr int N = 8;
    f16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (f16_t)f; }
};

// ---- the 16-bit activation type of the MFMA kernels (bf16 = production default, f16 = the reference's autocast dtype, config C5):
// everything that depends on the bit layout goes through these helpers, so one kernel source serves both
// pack2(a, b): the pair is converted as a VECTOR (round to nearest even, as the scalar casts).  Written as {(H)a, (H)b}, a pair whose halves are
// then used as 16-bit integers (the packed ReLU: v_pk_max_i16) or selected between is converted one value at a time and joined with a
// v_perm_b32 -- three instructions instead of one, beside MFMAs that leave the SIMD's vector port ~2 free slots each
typedef float f32x2p __attribute__((ext_vector_type(2)));
template <typename H> struct h16;
template <> struct h16<bf16_t> {
    typedef bf16x8 x8;
    typedef bf16_t x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }             // low half of a packed pair
    static __device__ __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static __device__ __forceinline__ x2 pack2(float a, float b) { return __builtin_convertvector(f32x2p{a, b}, x2); }   // ONE v_cvt_pk_bf16_f32 (below)
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct h16<f16_t> {
    typedef f16x8 x8;
    typedef f16_t x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16_t, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ x2 pack2(float a, float b) { return __builtin_convertvector(f32x2p{a, b}, x2); }   // ONE v_cvt_pk_f16_f32
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// two f32 -> one packed pair of H
template <typename H> __device__ __forceinline__ unsigned h16_pack(float a, float b) {
    return __builtin_bit_cast(unsigned, h16<H>::pack2(a, b));
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define WM_DISPATCH_DTYPE(dtype, NAME, ...)                                    \
    do {                                                                       \
        if ((dtype) == WM_F32) {                                               \
            using T = float;                                                   \
            __VA_ARGS__;                                                       \
        } else if ((dtype) == WM_BF16) {                                       \
            using T = bf16_t;                                                  \
            __VA_ARGS__;                                                       \
        } else if ((dtype) == WM_F16) {                                        \
            using T = f16_t;                                                   \
            __VA_ARGS__;                                                       \
        } else {                                                               \
            wm_set_error("%s: unsupported dtype %d", NAME, (int)(dtype));      \
            return WM_E_BADARG;                                                \
        }                                                                      \
    } while (0)
