// ConvTranspose2d(kernel 2, stride 2) of network/UNet.py:14-38 on MFMA (bf16, Cin % 64 == 0, Cout % 16 == 0).
// k2s2 means every output pixel (2h+i, 2w+j) has exactly ONE tap (i, j): the layer is four 1x1 GEMMs over the INPUT
// pixels, Y4[p][(ij, co)] = sum_ci A[p][ci] * W[ci][co][ij], followed by a pixel shuffle.  Three kernels:
//   upconv_mfma_kernel<0>  forward : A = relu(scale*x+shift) fused while staging, columns (ij, co), shuffle + bias on store
//   upconv_mfma_kernel<1>  dgrad   : dA[p][ci] = sum_(ij,co) G4[p][(ij,co)] * W[ci][co][ij], G4 gathered (un-shuffled) from gy
//   upconv_wgrad_kernel    dW      : dW[ci][(ij,co)] = sum_p A[p][ci] * G4[p][(ij,co)] (split over pixel ranges) + column sums
// All use v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (accumulator rows = output channels), so a lane owns
// 16 adjacent channels of one pixel and stores 2 x 16 bytes (see conv3x3_ws.hip).  The f32 / odd-shape path stays in unet.hip.
#include "wm_common.h"

// This file is compiled twice (build.py): plain for bf16 (production), and with -DWM_H16_F16 for the f16 twin of every kernel in
// it (the reference's autocast dtype, BASELINE config C5).  Everything that depends on the 16-bit layout goes through h16<> (wm_common.h).
#ifdef WM_H16_F16
typedef f16_t hx_t;
#define WM_HSYM(name) name##_f16
#else
typedef bf16_t hx_t;
#define WM_HSYM(name) name##_bf16
#endif
typedef h16<hx_t> HX;
typedef HX::x8 hx8;
typedef HX::x2 hx2;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int PT = 128;   // pixels per workgroup tile
constexpr int NT = 64;    // output columns per workgroup tile
constexpr int KC = 64;    // K chunk

__device__ __forceinline__ int swz128(int row, int slot) { return slot ^ ((row >> 1) & 7); }   // 16-byte slots of a 128-byte row

struct UpArgs {
    const hx_t* in; int ldin;              // MODE 0: x [Min][ldin];  MODE 1: gy [B,2H,2W,ldin]
    const float* scale; const float* shift;  // MODE 0 only (may be null)
    const hx_t* w;                         // MODE 0: wf [(ij,co)][Cin];  MODE 1: wb [Cin][(ij,co)]
    const float* bias;                       // MODE 0
    hx_t* out; int ldout; int c0;          // MODE 0: y [B,2H,2W,ldout] at channel c0;  MODE 1: gx [Min][ldout] (c0 = gy's channel offset)
    int B, H, W, Cin, Cout;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void upconv_mfma_kernel(UpArgs a) {
    __shared__ __attribute__((aligned(16))) hx_t sIn[2][PT * KC];
    __shared__ __attribute__((aligned(16))) hx_t sW[2][NT * KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const size_t Min = (size_t)a.B * a.H * a.W;
    const size_t p0 = (size_t)blockIdx.x * PT;
    const int n0 = blockIdx.y * NT;
    const int K = MODE == 0 ? a.Cin : 4 * a.Cout;
    const int nchunks = K / KC;

    // ---- staging geometry: 4 input vectors + 2 weight vectors per thread and chunk
    int ipx[4];          // pixel row of the tile
    size_t ibase[4];     // element offset of that pixel's row start (MODE 0) / of its (b, 2h, 2w) corner (MODE 1)
    bool iok[4];
    const int ivec = tid & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ipx[k] = (tid >> 3) + 32 * k;
        const size_t P = p0 + ipx[k];
        iok[k] = P < Min;
        const size_t Pc = iok[k] ? P : 0;
        if (MODE == 0) ibase[k] = Pc * a.ldin;
        else {
            const int w_ = (int)(Pc % a.W);
            const int h_ = (int)((Pc / a.W) % a.H);
            const size_t b = Pc / ((size_t)a.W * a.H);
            ibase[k] = ((b * 2 * a.H + 2 * h_) * 2 * a.W + 2 * w_) * (size_t)a.ldin + a.c0;
        }
    }
    const int wrow0 = tid >> 3, wvec = tid & 7;   // weight rows wrow0, wrow0 + 32
    hx8 rin[4], rw[2];
    auto load_chunk = [&](int c) {
        const int k0 = c * KC;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t off;
            if (MODE == 0) off = ibase[k] + k0 + ivec * 8;
            else {
                const int kk = k0 + ivec * 8;                           // an 8-channel vector lies inside one tap (Cout % 16 == 0)
                const int ij = kk / a.Cout, co = kk - ij * a.Cout;
                off = ibase[k] + ((size_t)(ij >> 1) * 2 * a.W + (ij & 1)) * a.ldin + co;
            }
            rin[k] = *reinterpret_cast<const hx8*>(a.in + off);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            rw[k] = *reinterpret_cast<const hx8*>(a.w + (size_t)(n0 + wrow0 + 32 * k) * K + k0 + wvec * 8);
    };
    auto put_chunk = [&](int c, int buf) {
        float sc[8], sh[8];
        const bool xf = MODE == 0 && a.scale != nullptr;
        if (xf) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = a.scale[c * KC + ivec * 8 + e]; sh[e] = a.shift[c * KC + ivec * 8 + e]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4 w = __builtin_bit_cast(u32x4, rin[k]);
            if (xf) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const float f0 = __builtin_fmaf(HX::lo(w[pq]), sc[2 * pq], sh[2 * pq]);
                    const float f1 = __builtin_fmaf(HX::hi(w[pq]), sc[2 * pq + 1], sh[2 * pq + 1]);
                    const hx2 pk = HX::pack2(f0, f1);
                    const i16x2 z = {0, 0};
                    w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                }
            }
            const unsigned keep = iok[k] ? 0xffffffffu : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] &= keep;
            *reinterpret_cast<u32x4*>(&sIn[buf][ipx[k] * KC + swz128(ipx[k], ivec) * 8]) = w;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = wrow0 + 32 * k;   // column of the tile -> accumulator-row permutation (a lane owns 16 adjacent columns)
            const int lrow = ((n >> 2) & 3) * 16 + 4 * (n >> 4) + (n & 3);
            *reinterpret_cast<hx8*>(&sW[buf][lrow * KC + swz128(lrow, wvec) * 8]) = rw[k];
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[ml][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_chunk(0);
    put_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            hx8 pix[2], fil[4];
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) {
                const int row = wave * 32 + ml * 16 + p;
                pix[ml] = *reinterpret_cast<const hx8*>(&sIn[buf][row * KC + swz128(row, ks * 4 + q) * 8]);
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int row = nf * 16 + p;
                fil[nf] = *reinterpret_cast<const hx8*>(&sW[buf][row * KC + swz128(row, ks * 4 + q) * 8]);
            }
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    acc[ml][nf] = HX::mfma16(fil[nf], pix[ml], acc[ml][nf]);
        }
        if (c + 1 < nchunks) put_chunk(c + 1, buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: lane (p, q) holds columns n0 + 16q .. +15 of pixel row wave*32 + ml*16 + p
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = 0.f;
    int ij = 0, co0 = 0;   // the lane's 16 columns n0 + 16q .. lie inside one tap (Cout % 16 == 0)
    if (MODE == 0) {
        ij = (n0 + 16 * q) / a.Cout; co0 = (n0 + 16 * q) - ij * a.Cout;
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bv[j] = a.bias[co0 + j];
        }
    }
#pragma unroll
    for (int ml = 0; ml < 2; ++ml) {
        const size_t P = p0 + wave * 32 + ml * 16 + p;
        if (P >= Min) continue;
        unsigned pk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int nf = j >> 1, i0 = 2 * (j & 1);
            const hx2 p2 = HX::pack2((acc[ml][nf][i0] + bv[2 * j]), (acc[ml][nf][i0 + 1] + bv[2 * j + 1]));
            pk[j] = __builtin_bit_cast(unsigned, p2);
        }
        hx_t* o;
        if (MODE == 0) {
            const int w_ = (int)(P % a.W);
            const int h_ = (int)((P / a.W) % a.H);
            const size_t b = P / ((size_t)a.W * a.H);
            o = a.out + ((b * 2 * a.H + 2 * h_ + (ij >> 1)) * 2 * a.W + 2 * w_ + (ij & 1)) * (size_t)a.ldout + a.c0 + co0;
        } else {
            o = a.out + P * a.ldout + n0 + 16 * q;
        }
        *reinterpret_cast<u32x4*>(o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(o + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
}

// ---------------------------------------------------------------------------------------------------------------- dW
// partial[split][ci][(ij,co)] = sum over the split's pixels of A[p][ci] * G4[p][(ij,co)];  bias_partial[split][(ij,co)] = sum G4.
// Workgroup = 64 ci x 64 columns; wave w owns ci fragment w (16 ci) x 4 column fragments; both operands reach the MFMA
// through transposing reads (K = pixels).  64-pixel chunks, double-buffered in LDS, rows of 128 B with the byte-offset
// swizzle of wgrad_ws.hip (bits 5-6 keyed on pixel bits 1 and 3): conflict-free ds_read_b64_tr_b16.
struct UpWgArgs {
    const hx_t* x; int ldx; const float* scale; const float* shift;
    const hx_t* gy; int ldgy; int c0;
    float* partial; float* bias_partial;
    int B, H, W, Cin, Cout, nsplit;
};

__device__ __forceinline__ int swzt(int pix) { return (((pix >> 1) & 1) << 5) | (((pix >> 3) & 1) << 6); }
__device__ __forceinline__ hx8 tr_frag2(const char* p0, const char* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(hx8, v);
}

__global__ __launch_bounds__(256, 2) void upconv_wgrad_kernel(UpWgArgs a) {
    constexpr int PC = 64;   // pixels per chunk
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][PC * 128];
    __shared__ __attribute__((aligned(16))) unsigned char sG[2][PC * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 64, n0 = blockIdx.y * 64, split = blockIdx.z;
    const int N = 4 * a.Cout;
    const size_t Min = (size_t)a.B * a.H * a.W;
    const size_t per = ((Min + a.nsplit - 1) / a.nsplit + PC - 1) / PC * PC;
    const size_t pbeg = (size_t)split * per, pend = pbeg + per < Min ? pbeg + per : Min;
    const int nchunks = pbeg < pend ? (int)((pend - pbeg + PC - 1) / PC) : 0;

    const int vec = tid & 7, prow = tid >> 3;   // staging: pixel rows prow, prow + 32; 16-byte vector `vec`
    const int ij = (n0 + vec * 8) / a.Cout, co0 = (n0 + vec * 8) - ij * a.Cout;   // this thread's 8 columns lie inside one tap
    float sc[8], sh[8];
    const bool xf = a.scale != nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = xf ? a.scale[ci0 + vec * 8 + e] : 1.f; sh[e] = xf ? a.shift[ci0 + vec * 8 + e] : 0.f; }
    hx8 ra[2], rg[2];
    bool rok[2];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const size_t P = pbeg + (size_t)c * PC + prow + 32 * k;
            rok[k] = P < pend;
            const size_t Pc = rok[k] ? P : pbeg;
            ra[k] = *reinterpret_cast<const hx8*>(a.x + Pc * a.ldx + ci0 + vec * 8);
            const int w_ = (int)(Pc % a.W);
            const int h_ = (int)((Pc / a.W) % a.H);
            const size_t b = Pc / ((size_t)a.W * a.H);
            rg[k] = *reinterpret_cast<const hx8*>(a.gy + ((b * 2 * a.H + 2 * h_ + (ij >> 1)) * 2 * a.W + 2 * w_ + (ij & 1)) * (size_t)a.ldgy +
                                                     a.c0 + co0);
        }
    };
    auto put_chunk = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pix = prow + 32 * k;
            u32x4 w = __builtin_bit_cast(u32x4, ra[k]);
            if (xf) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const float f0 = __builtin_fmaf(HX::lo(w[pq]), sc[2 * pq], sh[2 * pq]);
                    const float f1 = __builtin_fmaf(HX::hi(w[pq]), sc[2 * pq + 1], sh[2 * pq + 1]);
                    const hx2 pk = HX::pack2(f0, f1);
                    const i16x2 z = {0, 0};
                    w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                }
            }
            u32x4 g = __builtin_bit_cast(u32x4, rg[k]);
            const unsigned keep = rok[k] ? 0xffffffffu : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { w[e] &= keep; g[e] &= keep; }
            const int off = pix * 128 + ((vec * 16) ^ swzt(pix));
            *reinterpret_cast<u32x4*>(sA[buf] + off) = w;
            *reinterpret_cast<u32x4*>(sG[buf] + off) = g;
        }
    };
    // transposing-read geometry (see wgrad_ws.hip): lane = (r = lane & 15, kq = lane >> 4); the 16-lane group kq reads pixel
    // rows 8kq + qq (+4), lane 4qq + pp of the group supplies the address of row qq, channels 4pp .. 4pp+3
    const int r = lane & 15, kq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    int aoffs[2], goffs[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pix = 8 * kq + qq + 4 * s2;
        aoffs[s2] = pix * 128 + (((wave * 16 + 4 * pp) * 2) ^ swzt(pix));
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) goffs[s2][nf] = pix * 128 + (((nf * 16 + 4 * pp) * 2) ^ swzt(pix));
    }
    f32x4 acc[4], accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    hx8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (hx_t)1.0f;
    if (nchunks > 0) {
        load_chunk(0);
        put_chunk(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const char* cA = reinterpret_cast<const char*>(sA[buf]);
        const char* cG = reinterpret_cast<const char*>(sG[buf]);
#pragma unroll
        for (int ks = 0; ks < PC / 32; ++ks) {
            const hx8 af = tr_frag2(cA + ks * 32 * 128 + aoffs[0], cA + ks * 32 * 128 + aoffs[1]);
            hx8 gf[4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) gf[nf] = tr_frag2(cG + ks * 32 * 128 + goffs[0][nf], cG + ks * 32 * 128 + goffs[1][nf]);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[nf] = HX::mfma16(af, gf[nf], acc[nf]);
            // column sums of G4 (the bias gradient): ones x G for this wave's own column fragment
            accb = HX::mfma16(ones, gf[wave], accb);
        }
        if (c + 1 < nchunks) put_chunk(buf ^ 1);
        __syncthreads();
    }
    // D layout: lane (col = r -> column, rows 4 kq + i -> ci of the wave's fragment)
    float* prow_out = a.partial + (size_t)split * a.Cin * N;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            prow_out[(size_t)(ci0 + wave * 16 + 4 * kq + i) * N + n0 + nf * 16 + r] = acc[nf][i];
    if (blockIdx.x == 0 && kq == 0) a.bias_partial[(size_t)split * N + n0 + wave * 16 + r] = accb[0];
}

// dw[ci][co][ij] (+)= sum_split partial[split][ci][(ij,co)];  dbias[co] (+)= sum_split sum_ij bias_partial[split][(ij,co)]
__global__ __launch_bounds__(256) void upconv_dw_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias_partial,
                                                               int nsplit, int Cin, int Cout, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int accumulate) {
    const int N = 4 * Cout;
    const size_t total = (size_t)Cin * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total + Cout; i += (size_t)gridDim.x * 256) {
        if (i < total) {
            const int ci = (int)(i / N), n = (int)(i - (size_t)ci * N);
            const int ij = n / Cout, co = n - ij * Cout;
            float s = 0.f;
            for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * total + i];
            float* o = dw + ((size_t)ci * Cout + co) * 4 + ij;
            *o = (accumulate ? *o : 0.f) + s;
        } else {
            const int co = (int)(i - total);
            float s = 0.f;
            for (int k = 0; k < nsplit; ++k)
                for (int t = 0; t < 4; ++t) s += bias_partial[(size_t)k * N + t * Cout + co];
            dbias[co] = (accumulate ? dbias[co] : 0.f) + s;
        }
    }
}

// w [Cin][Cout][2][2] f32 -> wf [(ij,co)][Cin] bf16 (forward operand), wb [Cin][(ij,co)] bf16 (dgrad operand)
__global__ __launch_bounds__(256) void upconv_pack_kernel(const float* __restrict__ w, hx_t* __restrict__ wf, hx_t* __restrict__ wb,
                                                          int Cin, int Cout) {
    const size_t total = (size_t)Cin * Cout * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ijx = (int)(i & 3);
        const int co = (int)((i >> 2) % Cout);
        const int ci = (int)(i / ((size_t)4 * Cout));
        const hx_t v = (hx_t)w[i];
        wf[((size_t)ijx * Cout + co) * Cin + ci] = v;
        wb[(size_t)ci * 4 * Cout + (size_t)ijx * Cout + co] = v;
    }
}

}  // namespace

#ifndef WM_H16_F16
extern "C" int wm_upconv2x2_mfma_supported(int Cin, int Cout, int dtype) {
    return ((dtype == WM_BF16 || dtype == WM_F16) && Cin % 64 == 0 && Cout % 16 == 0) ? 1 : 0;
}
#else
extern "C" int wm_upconv2x2_mfma_supported(int Cin, int Cout, int dtype);
#endif

int WM_HSYM(wm_upconv2x2_pack)(const float* w, void* wf, void* wb, int Cin, int Cout, void* stream) {
    WM_REQUIRE(w && wf && wb && Cin > 0 && Cout > 0, WM_E_BADARG, "wm_upconv2x2_pack: bad arguments");
    const size_t total = (size_t)Cin * Cout * 4;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(upconv_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (hx_t*)wf, (hx_t*)wb, Cin, Cout);
    WM_LAUNCH_CHECK("wm_upconv2x2_pack");
    return WM_OK;
}

int WM_HSYM(wm_upconv2x2_fwd_mfma)(const void* x, int ldx, const float* scale, const float* shift, const void* wf,
                                     const float* bias, void* y, int ldy, int c0, int B, int H, int W, int Cin, int Cout,
                                     void* stream) {
    WM_REQUIRE(x && wf && y, WM_E_BADARG, "wm_upconv2x2_fwd_mfma: null pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_upconv2x2_fwd_mfma: scale/shift must come together");
    WM_REQUIRE(wm_upconv2x2_mfma_supported(Cin, Cout, wm_dtype<hx_t>::id), WM_E_SHAPE, "wm_upconv2x2_fwd_mfma: Cin=%d, Cout=%d must be multiples of 64 / 16", Cin, Cout);
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && ldx >= Cin && ldy >= c0 + Cout && ldx % 8 == 0 && ldy % 8 == 0 && c0 % 8 == 0, WM_E_SHAPE,
               "wm_upconv2x2_fwd_mfma: bad strides (ldx=%d ldy=%d c0=%d)", ldx, ldy, c0);
    UpArgs a;
    a.in = (const hx_t*)x; a.ldin = ldx; a.scale = scale; a.shift = shift; a.w = (const hx_t*)wf; a.bias = bias;
    a.out = (hx_t*)y; a.ldout = ldy; a.c0 = c0; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const size_t Min = (size_t)B * H * W;
    const dim3 grid((unsigned)((Min + PT - 1) / PT), (unsigned)(4 * Cout / NT));
    hipLaunchKernelGGL((upconv_mfma_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, a);
    WM_LAUNCH_CHECK("wm_upconv2x2_fwd_mfma");
    return WM_OK;
}

int WM_HSYM(wm_upconv2x2_dgrad_mfma)(const void* gy, int ldgy, int c0, const void* wb, void* gx, int ldgx, int B, int H, int W,
                                       int Cin, int Cout, void* stream) {
    WM_REQUIRE(gy && wb && gx, WM_E_BADARG, "wm_upconv2x2_dgrad_mfma: null pointer");
    WM_REQUIRE(wm_upconv2x2_mfma_supported(Cin, Cout, wm_dtype<hx_t>::id), WM_E_SHAPE, "wm_upconv2x2_dgrad_mfma: Cin=%d, Cout=%d must be multiples of 64 / 16", Cin, Cout);
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && ldgx >= Cin && ldgy >= c0 + Cout && ldgx % 8 == 0 && ldgy % 8 == 0 && c0 % 8 == 0, WM_E_SHAPE,
               "wm_upconv2x2_dgrad_mfma: bad strides (ldgx=%d ldgy=%d c0=%d)", ldgx, ldgy, c0);
    UpArgs a;
    a.in = (const hx_t*)gy; a.ldin = ldgy; a.scale = nullptr; a.shift = nullptr; a.w = (const hx_t*)wb; a.bias = nullptr;
    a.out = (hx_t*)gx; a.ldout = ldgx; a.c0 = c0; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const size_t Min = (size_t)B * H * W;
    const dim3 grid((unsigned)((Min + PT - 1) / PT), (unsigned)(Cin / NT));
    hipLaunchKernelGGL((upconv_mfma_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    WM_LAUNCH_CHECK("wm_upconv2x2_dgrad_mfma");
    return WM_OK;
}

#ifndef WM_H16_F16
extern "C" int wm_upconv2x2_wgrad_nsplit(int B, int H, int W, int Cin, int Cout) {
    const size_t Min = (size_t)B * H * W;
    const int blocks = (Cin / 64) * (4 * Cout / 64);
    size_t ns = blocks > 0 ? (size_t)((1024 + blocks - 1) / blocks) : 1;
    const size_t cap = (Min + 255) / 256;
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return (int)ns;
}
#else
extern "C" int wm_upconv2x2_wgrad_nsplit(int B, int H, int W, int Cin, int Cout);
#endif

int WM_HSYM(wm_upconv2x2_wgrad_mfma)(const void* x, int ldx, const float* scale, const float* shift, const void* gy, int ldgy,
                                       int c0, float* partial, float* bias_partial, float* dw, float* dbias, int accumulate, int B,
                                       int H, int W, int Cin, int Cout, void* stream) {
    WM_REQUIRE(x && gy && partial && bias_partial && dw && dbias, WM_E_BADARG, "wm_upconv2x2_wgrad_mfma: null pointer");
    WM_REQUIRE((scale == nullptr) == (shift == nullptr), WM_E_BADARG, "wm_upconv2x2_wgrad_mfma: scale/shift must come together");
    WM_REQUIRE(wm_upconv2x2_mfma_supported(Cin, Cout, wm_dtype<hx_t>::id), WM_E_SHAPE, "wm_upconv2x2_wgrad_mfma: Cin=%d, Cout=%d must be multiples of 64 / 16", Cin, Cout);
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && ldx >= Cin && ldgy >= c0 + Cout && ldx % 8 == 0 && ldgy % 8 == 0 && c0 % 8 == 0, WM_E_SHAPE,
               "wm_upconv2x2_wgrad_mfma: bad strides (ldx=%d ldgy=%d c0=%d)", ldx, ldgy, c0);
    UpWgArgs a;
    a.x = (const hx_t*)x; a.ldx = ldx; a.scale = scale; a.shift = shift; a.gy = (const hx_t*)gy; a.ldgy = ldgy; a.c0 = c0;
    a.partial = partial; a.bias_partial = bias_partial; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.nsplit = wm_upconv2x2_wgrad_nsplit(B, H, W, Cin, Cout);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(Cin / 64), (unsigned)(4 * Cout / 64), (unsigned)a.nsplit);
    hipLaunchKernelGGL(upconv_wgrad_kernel, grid, dim3(256), 0, s, a);
    WM_LAUNCH_CHECK("wm_upconv2x2_wgrad_mfma");
    const size_t total = (size_t)Cin * 4 * Cout + Cout;
    const int rgrid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(upconv_dw_reduce_kernel, dim3(rgrid), dim3(256), 0, s, partial, bias_partial, a.nsplit, Cin, Cout, dw, dbias, accumulate);
    WM_LAUNCH_CHECK("wm_upconv2x2_wgrad_mfma(reduce)");
    return WM_OK;
}


#ifndef WM_H16_F16
// ---- the C ABI: one entry point per operation, dispatching on the 16-bit activation dtype to the two compilations of this file
int wm_upconv2x2_pack_f16(const float* w, void* wf, void* wb, int Cin, int Cout, void* stream);
int wm_upconv2x2_fwd_mfma_f16(const void* x, int ldx, const float* scale, const float* shift, const void* wf, const float* bias, void* y,
                              int ldy, int c0, int B, int H, int W, int Cin, int Cout, void* stream);
int wm_upconv2x2_dgrad_mfma_f16(const void* gy, int ldgy, int c0, const void* wb, void* gx, int ldgx, int B, int H, int W, int Cin, int Cout,
                                void* stream);
int wm_upconv2x2_wgrad_mfma_f16(const void* x, int ldx, const float* scale, const float* shift, const void* gy, int ldgy, int c0,
                                float* partial, float* bias_partial, float* dw, float* dbias, int accumulate, int B, int H, int W, int Cin,
                                int Cout, void* stream);
#define WM_H16_DISPATCH(name, ...)                                                                        \
    do {                                                                                                  \
        if (dtype == WM_BF16) return name##_bf16(__VA_ARGS__);                                            \
        if (dtype == WM_F16) return name##_f16(__VA_ARGS__);                                              \
        wm_set_error(#name ": dtype must be WM_BF16 or WM_F16 (got %d)", dtype);                         \
        return WM_E_BADARG;                                                                               \
    } while (0)
extern "C" int wm_upconv2x2_pack(const float* w, void* wf, void* wb, int Cin, int Cout, int dtype, void* stream) {
    WM_H16_DISPATCH(wm_upconv2x2_pack, w, wf, wb, Cin, Cout, stream);
}
extern "C" int wm_upconv2x2_fwd_mfma(const void* x, int ldx, const float* scale, const float* shift, const void* wf, const float* bias,
                                     void* y, int ldy, int c0, int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    WM_H16_DISPATCH(wm_upconv2x2_fwd_mfma, x, ldx, scale, shift, wf, bias, y, ldy, c0, B, H, W, Cin, Cout, stream);
}
extern "C" int wm_upconv2x2_dgrad_mfma(const void* gy, int ldgy, int c0, const void* wb, void* gx, int ldgx, int B, int H, int W, int Cin,
                                       int Cout, int dtype, void* stream) {
    WM_H16_DISPATCH(wm_upconv2x2_dgrad_mfma, gy, ldgy, c0, wb, gx, ldgx, B, H, W, Cin, Cout, stream);
}
extern "C" int wm_upconv2x2_wgrad_mfma(const void* x, int ldx, const float* scale, const float* shift, const void* gy, int ldgy, int c0,
                                       float* partial, float* bias_partial, float* dw, float* dbias, int accumulate, int B, int H, int W,
                                       int Cin, int Cout, int dtype, void* stream) {
    WM_H16_DISPATCH(wm_upconv2x2_wgrad_mfma, x, ldx, scale, shift, gy, ldgy, c0, partial, bias_partial, dw, dbias, accumulate, B, H, W, Cin,
                    Cout, stream);
}
#endif
