// General convolution family for the networks either side of the HiDDeN path (models/networks.py:631-749
// Discriminator, models/conditional_jpeg_generator.py:185-374 FBCNN, :697-826 QF_predictor): any kernel size up to 5x5, stride 1 or 2,
// zero padding, bias; forward, input gradient (also = ConvTranspose2d forward) and weight gradient, NHWC, f32 / bf16 / f16.
//
// These layers are outside the benchmarked step, so the kernels are DIRECT implicit GEMMs on MFMA without an LDS stage: a wave owns 16
// output pixels x 64 output channels; per filter tap and 32-channel chunk every lane loads its 16 bytes of the pixel fragment and of the
// filter fragments straight from global memory (the filter is L1/L2-resident: every wave reads the same rows).  The filter is the A
// operand, so accumulator rows are channels and a lane ends with 16 adjacent channels of one pixel (the store pattern of
// conv3x3_ws.hip).  f32 runs the same loop on v_mfma_f32_16x16x4_f32 (an exact f32 FMA chain: the parity path).
//   forward : out[b,oy,ox,n] = bias[n] + sum_{ky,kx,k} in[b, oy*s - p + ky, ox*s - p + kx, k] * w[tap][n][k]
//   dgrad   : din[b,iy,ix,k] = sum_{ky,kx,n} dout[b,(iy + p - ky)/s, (ix + p - kx)/s, n] * wT[tap][k][n]   (taps with a non-integer or
//             out-of-range source contribute nothing) -- the same kernel with the source geometry inverted
//   wgrad   : dw[tap][n][k] = sum_{b,oy,ox} dout[b,oy,ox,n] * in[b, oy*s - p + ky, ox*s - p + kx, k], pixels as the MFMA K dimension
#include <algorithm>
#include "wm_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GArgs {
    const void* in; const void* w; const float* bias; void* out;
    int B, IH, IW, KC;     // input tensor [B,IH,IW,KC] (KC = its channel stride)
    int OH, OW, NC;        // output tensor [B,OH,OW,NC]
    int KH, KW, stride, pad;
    int dgrad;             // 0: forward geometry, 1: transposed geometry (in = dout [B,IH,IW,KC] of the forward conv whose INPUT is `out`)
};

// source pixel of output pixel (oy, ox) for tap (ky, kx); returns false if it contributes nothing
__device__ __forceinline__ bool src_of(const GArgs& a, int oy, int ox, int ky, int kx, int& sy, int& sx) {
    if (!a.dgrad) {
        sy = oy * a.stride - a.pad + ky;
        sx = ox * a.stride - a.pad + kx;
        return sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW;
    }
    const int ty = oy + a.pad - ky, tx = ox + a.pad - kx;
    if (ty < 0 || tx < 0) return false;
    if (a.stride == 2) {
        if ((ty | tx) & 1) return false;
        sy = ty >> 1; sx = tx >> 1;
    } else { sy = ty; sx = tx; }
    return sy < a.IH && sx < a.IW;
}

template <typename T> struct GOp;   // one MFMA K-step of T
template <> struct GOp<float> {
    static constexpr int KSTEP = 4;
    typedef float frag;
    static __device__ __forceinline__ frag load(const float* p, int q, int krem) { return q < krem ? p[q] : 0.f; }
    static __device__ __forceinline__ frag zero() { return 0.f; }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <typename H> struct GOp16 {
    static constexpr int KSTEP = 32;
    typedef typename h16<H>::x8 frag;
    // 8 consecutive channels starting at 8q; channels at or beyond krem (a multiple of 8) read as zero
    static __device__ __forceinline__ frag load(const H* p, int q, int krem) {
        if (8 * q >= krem) return zero();
        return *reinterpret_cast<const frag*>(p + 8 * q);
    }
    static __device__ __forceinline__ frag zero() { return __builtin_bit_cast(frag, u32x4{0u, 0u, 0u, 0u}); }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return h16<H>::mfma16(a, b, c); }
};
template <> struct GOp<bf16_t> : GOp16<bf16_t> {};
template <> struct GOp<f16_t> : GOp16<f16_t> {};

// grid.x = 64-pixel blocks of the flattened output, grid.y = 64-channel blocks; 256 threads = 4 waves x 16 pixels
template <typename T>
__global__ __launch_bounds__(256) void gconv_kernel(GArgs a) {
    typedef GOp<T> Op;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, q = lane >> 4;
    const size_t npix = (size_t)a.B * a.OH * a.OW;
    const size_t lin = ((size_t)blockIdx.x * 4 + wave) * 16 + p;
    const bool pok = lin < npix;
    const size_t pc = pok ? lin : npix - 1;
    int ox, oy, b;
    if (a.dgrad && a.stride == 2 && !((a.OH | a.OW) & 1)) {
        // stride-2 input gradient: an output pixel's parity class (oy & 1, ox & 1) decides which taps reach it (4 of 16 for a 4x4 filter,
        // 1 of 4 for 2x2).  Pixels are walked class by class, so a wave's 16 pixels share the class and the other taps are skipped whole
        const int hw = a.OW >> 1, hh = a.OH >> 1;
        const size_t per = npix >> 2, cls = pc / per, idx = pc - cls * per;
        const int j = (int)(idx % hw), i = (int)((idx / hw) % hh);
        b = (int)(idx / ((size_t)hw * hh));
        oy = 2 * i + (int)(cls >> 1); ox = 2 * j + (int)(cls & 1);
    } else {
        ox = (int)(pc % a.OW); oy = (int)((pc / a.OW) % a.OH); b = (int)(pc / ((size_t)a.OW * a.OH));
    }
    const size_t pix = ((size_t)b * a.OH + oy) * a.OW + ox;
    const int n0 = blockIdx.y * 64;
    const T* in = (const T*)a.in;
    const T* w = (const T*)a.w;
    // A rows: row r of fragment f <-> channel n0 + 16 (r >> 2) + 4 f + (r & 3); this lane provides row p
    int arow[4];
    bool aok[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        arow[f] = n0 + 16 * (p >> 2) + 4 * f + (p & 3);
        aok[f] = arow[f] < a.NC;
    }
    const int taps = a.KH * a.KW;
    if (a.NC - n0 <= 16) {
        // narrow output (<= 16 channels left in this block: the subnets' 72 -> 8 convolutions, input gradients towards 8 / 3 channels, the 5x5
        // Bayar layer): ONE filter fragment whose row r is channel n0 + r -- a quarter of the MFMAs and filter loads of the 64-channel form
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
        const int row = n0 + p;
        const bool rok = row < a.NC;
        for (int tap = 0; tap < taps; ++tap) {
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            int sy = 0, sx = 0;
            const bool ok = src_of(a, oy, ox, ky, kx, sy, sx) && pok;
            if (__ballot(ok) == 0) continue;
            const T* px = in + (((size_t)b * a.IH + (ok ? sy : 0)) * a.IW + (ok ? sx : 0)) * a.KC;
            const T* wt = w + ((size_t)tap * a.NC + (rok ? row : 0)) * a.KC;
            for (int c0 = 0; c0 < a.KC; c0 += Op::KSTEP) {
                const int krem = a.KC - c0;
                const typename Op::frag bf = ok ? Op::load(px + c0, q, krem) : Op::zero();
                const typename Op::frag af = rok ? Op::load(wt + c0, q, krem) : Op::zero();
                acc1 = Op::mma(af, bf, acc1);
            }
        }
        // lane (p, q): channels n0 + 4 q + i of pixel p
        const int cb = n0 + 4 * q;
        if (pok && cb < a.NC) {
            T* o = (T*)a.out + pix * a.NC + cb;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = from_f32<T>(acc1[i] + (a.bias ? a.bias[cb + i] : 0.f));
        }
        return;
    }
    f32x4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        int sy = 0, sx = 0;
        const bool ok = src_of(a, oy, ox, ky, kx, sy, sx) && pok;
        if (__ballot(ok) == 0) continue;   // no lane of the wave has a source pixel under this tap (wave-uniform)
        const T* px = in + (((size_t)b * a.IH + (ok ? sy : 0)) * a.IW + (ok ? sx : 0)) * a.KC;
        const T* wt = w + (size_t)tap * a.NC * a.KC;
        for (int c0 = 0; c0 < a.KC; c0 += Op::KSTEP) {
            const int krem = a.KC - c0;
            const typename Op::frag bf = ok ? Op::load(px + c0, q, krem) : Op::zero();
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const typename Op::frag af = aok[f] ? Op::load(wt + (size_t)arow[f] * a.KC + c0, q, krem) : Op::zero();
                acc[f] = Op::mma(af, bf, acc[f]);
            }
        }
    }
    // lane (p, q): channels n0 + 16 q + 4 f + i of pixel p
    const int cb = n0 + 16 * q;
    if (pok && cb < a.NC) {
        T* o = (T*)a.out + pix * a.NC + cb;
        float v[16];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * f + i] = acc[f][i] + (a.bias ? a.bias[cb + 4 * f + i] : 0.f);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = from_f32<T>(v[e]);
    }
}

// ---- weight gradient.  job = (tap, 16-row block of n, 16-column block of k); grid.y = pixel splits; the 4 waves of a workgroup take
// interleaved 32-pixel (f32: 4-pixel) steps of the split and are summed through LDS.  partial: [nsplit][taps][NC][KC] f32.
struct GWArgs {
    const void* dout; const void* in; float* partial;
    int B, IH, IW, KC, OH, OW, NC, KH, KW, stride, pad, nsplit;
};
template <typename T> struct GLd { static __device__ __forceinline__ float f(const T* p) { return to_f32(*p); } };

template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(GWArgs a) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int KS = F32 ? 4 : 32, PER = F32 ? 1 : 8;   // pixels per MFMA step, per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nb = a.NC / 16, kb = (a.KC + 15) / 16;
    int job = blockIdx.x;
    const int kblk = job % kb; job /= kb;
    const int nblk = job % nb; job /= nb;
    const int tap = job, ky = tap / a.KW, kx = tap - ky * a.KW;
    const size_t npix = (size_t)a.B * a.OH * a.OW;
    const size_t per = (npix + a.nsplit - 1) / a.nsplit, p0 = (size_t)blockIdx.y * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* dout = (const T*)a.dout;
    const T* in = (const T*)a.in;
    const int n = nblk * 16 + r, k = kblk * 16 + r;
    const bool kok = k < a.KC;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t s0 = p0 + (size_t)wave * KS; s0 < p1; s0 += 4 * KS) {
        float av[PER], bv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const size_t pix = s0 + (size_t)PER * q + e;
            av[e] = 0.f; bv[e] = 0.f;
            if (pix < p1) {
                av[e] = to_f32(dout[pix * a.NC + n]);
                const int ox = (int)(pix % a.OW), oy = (int)((pix / a.OW) % a.OH), b = (int)(pix / ((size_t)a.OW * a.OH));
                const int sy = oy * a.stride - a.pad + ky, sx = ox * a.stride - a.pad + kx;
                if (kok && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW) bv[e] = to_f32(in[(((size_t)b * a.IH + sy) * a.IW + sx) * a.KC + k]);
            }
        }
        if constexpr (F32) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
        } else {
            const unsigned a0 = h16_pack<T>(av[0], av[1]), a1 = h16_pack<T>(av[2], av[3]), a2 = h16_pack<T>(av[4], av[5]), a3 = h16_pack<T>(av[6], av[7]);
            const unsigned b0 = h16_pack<T>(bv[0], bv[1]), b1 = h16_pack<T>(bv[2], bv[3]), b2 = h16_pack<T>(bv[4], bv[5]), b3 = h16_pack<T>(bv[6], bv[7]);
            typedef typename h16<T>::x8 fr;
            acc = h16<T>::mfma16(__builtin_bit_cast(fr, u32x4{a0, a1, a2, a3}), __builtin_bit_cast(fr, u32x4{b0, b1, b2, b3}), acc);
        }
    }
    // D[row = n index 4q + i][col = k index r]
    __shared__ float red[4][16][17];
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * q + i][r] = acc[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            const float v = (red[0][row][r] + red[1][row][r]) + (red[2][row][r] + red[3][row][r]);
            const int kk = kblk * 16 + r;
            if (kk < a.KC) a.partial[(((size_t)blockIdx.y * a.KH * a.KW + tap) * a.NC + nblk * 16 + row) * a.KC + kk] = v;
        }
    }
}

// ---- weight gradient, 16-bit dtypes: a workgroup owns a 64 n x 64 k block for a group of <= 8 filter taps and a split of the pixels.  Per
// chunk of 64 output pixels it stages dout [64 px][64 n] once and, per tap, the gathered input rows [64 px][64 k] (zero where the tap
// falls outside the image) into the LDS as 128-byte pixel rows (16-byte vector loads, wgrad_ws.hip's bank swizzle); the four waves each
// keep a 32 x 32 sub-block x taps in registers and read both operands through the transposing ds_read_b64_tr_b16 (pixels are the MFMA's
// K dimension).  Every input row is read once per (tap, n-block), not once per 16 x 16 block as in the scalar kernel below.
constexpr int WG_TAPS = 8;
__device__ __forceinline__ int gswz16(int col) { return (((col >> 1) & 1) << 5) | (((col >> 3) & 1) << 6); }
template <typename H>
__device__ __forceinline__ typename h16<H>::x8 g_tr_frag(const char* p0, const char* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename h16<H>::x8, v);
}
template <typename H>
__global__ __launch_bounds__(256) void gconv_wgrad16_kernel(GWArgs a, int tapgroups) {
    typedef typename h16<H>::x8 frag;
    __shared__ __attribute__((aligned(16))) unsigned char sm[3 * 64 * 128];   // dout chunk | two input chunks (double-buffered over the taps)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = a.KH * a.KW;
    const int kb = (a.KC + 63) / 64;
    int job = blockIdx.y;
    const int tg = job % tapgroups; job /= tapgroups;
    const int kblk = job % kb, nblk = job / kb;
    const int tap0 = tg * WG_TAPS, ntap = min(WG_TAPS, taps - tap0);
    const size_t npix = (size_t)a.B * a.OH * a.OW;
    const size_t chunks = (npix + 63) / 64, cper = (chunks + a.nsplit - 1) / a.nsplit;
    const size_t c0 = (size_t)blockIdx.x * cper, c1 = c0 + cper < chunks ? c0 + cper : chunks;
    const H* dout = (const H*)a.dout;
    const H* in = (const H*)a.in;
    // staging: thread -> (pixel of the chunk, 16-byte vector), two rounds of 32 pixels
    const int vec = tid & 7, prow = tid >> 3;
    // fragment addressing (wgrad_ws.hip): lane (r, kq), (q2, p2); a K-step = 32 pixels = rows 2ks, 2ks+1 of 16 pixels
    const int mi = wave >> 1, ni = wave & 1;
    const int r = lane & 15, kq = lane >> 4, q2 = (lane >> 2) & 3, p2 = lane & 3;
    const int colb = 8 * (kq & 1) + q2;
    int aof[2][2], bof[2][2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int col = colb + 4 * sx;
            aof[sx][f] = ((kq >> 1) * 16 + col) * 128 + (((mi * 32 + f * 16 + 4 * p2) * 2) ^ gswz16(col));
            bof[sx][f] = ((kq >> 1) * 16 + col) * 128 + (((ni * 32 + f * 16 + 4 * p2) * 2) ^ gswz16(col));
        }
    f32x4 acc[WG_TAPS][2][2];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    for (size_t ch = c0; ch < c1; ++ch) {
        // ---- stage the dout chunk and the first tap's input chunk
        int oys[2], oxs[2], bs[2]; bool pok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t pix = ch * 64 + prow + 32 * h;
            pok[h] = pix < npix;
            const size_t pc = pok[h] ? pix : 0;
            oxs[h] = (int)(pc % a.OW); oys[h] = (int)((pc / a.OW) % a.OH); bs[h] = (int)(pc / ((size_t)a.OW * a.OH));
            const int n = nblk * 64 + vec * 8;
            u32x4v v = {0u, 0u, 0u, 0u};
            if (pok[h] && n < a.NC) v = *reinterpret_cast<const u32x4v*>(dout + pc * a.NC + n);
            const int pl = prow + 32 * h;
            *reinterpret_cast<u32x4v*>(sm + pl * 128 + ((vec * 16) ^ gswz16(pl & 15))) = v;
        }
        auto stage_x = [&](int tap, unsigned char* buf) {
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sy = oys[h] * a.stride - a.pad + ky, sx = oxs[h] * a.stride - a.pad + kx;
                const int k = kblk * 64 + vec * 8;
                u32x4v v = {0u, 0u, 0u, 0u};
                if (pok[h] && k < a.KC && sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW)
                    v = *reinterpret_cast<const u32x4v*>(in + (((size_t)bs[h] * a.IH + sy) * a.IW + sx) * a.KC + k);
                const int pl = prow + 32 * h;
                *reinterpret_cast<u32x4v*>(buf + pl * 128 + ((vec * 16) ^ gswz16(pl & 15))) = v;
            }
        };
        stage_x(tap0, sm + 64 * 128);
        __syncthreads();
        for (int t = 0; t < ntap; ++t) {
            const char* sd = reinterpret_cast<const char*>(sm);
            const char* sxc = reinterpret_cast<const char*>(sm + (1 + (t & 1)) * 64 * 128);
            if (t + 1 < ntap) stage_x(tap0 + t + 1, sm + (1 + ((t + 1) & 1)) * 64 * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag af[2], bf[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    af[f] = g_tr_frag<H>(sd + 2 * ks * 16 * 128 + aof[0][f], sd + 2 * ks * 16 * 128 + aof[1][f]);
                    bf[f] = g_tr_frag<H>(sxc + 2 * ks * 16 * 128 + bof[0][f], sxc + 2 * ks * 16 * 128 + bof[1][f]);
                }
#pragma unroll
                for (int tt = 0; tt < WG_TAPS; ++tt)
                    if (tt == t) {   // (t is a run-time index into register-resident accumulators: a wave-uniform select)
#pragma unroll
                        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                            for (int fj = 0; fj < 2; ++fj) acc[tt][fi][fj] = h16<H>::mfma16(af[fi], bf[fj], acc[tt][fi][fj]);
                    }
            }
            __syncthreads();
        }
    }
    // acc[t][fi][fj][i] = dW[tap0 + t][n = 64 nblk + 32 mi + 16 fi + 4 kq + i][k = 64 kblk + 32 ni + 16 fj + r]
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
        if (t < ntap) {
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = nblk * 64 + mi * 32 + fi * 16 + 4 * kq + i, k = kblk * 64 + ni * 32 + fj * 16 + r;
                        if (n < a.NC && k < a.KC) a.partial[(((size_t)blockIdx.x * taps + tap0 + t) * a.NC + n) * a.KC + k] = acc[t][fi][fj][i];
                    }
        }
}

// ---- weight gradient of a THIN layer (input channel stride 16: the 3-channel image planes of the 5x5 Bayar layer and of the first 4x4
// stride-2 discriminator layer), 16-bit dtypes.  A 64 x 64 block per tap as above wastes 15/16 of every staged row and stages the input once
// per tap.  Here a workgroup owns one 16-channel block of dout and a split of the chunks (64 consecutive output pixels of one output row);
// per chunk it stages dout [64 px][16 n] and ONCE the input window [KH rows][63 stride + KW px][16 k] (32 bytes per pixel), and every tap's
// B fragment is a shifted (stride-2: strided) transposing read of that window.  Wave w accumulates the taps w, w + 4, ... in registers.
constexpr int THIN_MAXT = 7;   // taps per wave: 25 / 4 rounded up
template <typename H>
__global__ __launch_bounds__(256) void gconv_wgrad_thin_kernel(GWArgs a, int chunks_x, int winw) {
    typedef typename h16<H>::x8 frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];   // dout chunk 2 KB | input window KH * winw * 32 B
    unsigned char* sd = tsm;
    unsigned char* sx = tsm + 64 * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = a.KH * a.KW, nblk = blockIdx.y;
    const size_t chunks = (size_t)a.B * a.OH * chunks_x, cper = (chunks + a.nsplit - 1) / a.nsplit;
    const size_t c0 = (size_t)blockIdx.x * cper, c1 = c0 + cper < chunks ? c0 + cper : chunks;
    const H* dout = (const H*)a.dout;
    const H* in = (const H*)a.in;
    const int r = lane & 15, kq = lane >> 4, q2 = (lane >> 2) & 3, p2 = lane & 3;
    f32x4 acc[THIN_MAXT];
#pragma unroll
    for (int t = 0; t < THIN_MAXT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const int nwin = a.KH * winw * 2;   // 16-byte vectors of the window
    for (size_t ch = c0; ch < c1; ++ch) {
        const int xs = (int)(ch % chunks_x), oy = (int)((ch / chunks_x) % a.OH), b = (int)(ch / ((size_t)chunks_x * a.OH));
        const int ox0 = xs * 64;
        if (tid < 128) {
            const int px = tid >> 1, h = tid & 1;
            u32x4v v = {0u, 0u, 0u, 0u};
            if (ox0 + px < a.OW) v = *reinterpret_cast<const u32x4v*>(dout + (((size_t)b * a.OH + oy) * a.OW + ox0 + px) * a.NC + nblk * 16 + 8 * h);
            *reinterpret_cast<u32x4v*>(sd + px * 32 + 16 * h) = v;
        }
        const int iy0 = oy * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
        for (int i = tid; i < nwin; i += 256) {
            const int h = i & 1, wp = (i >> 1) % winw, wr = (i >> 1) / winw;
            const int sy = iy0 + wr, sxx = ix0 + wp;
            u32x4v v = {0u, 0u, 0u, 0u};
            if (sy >= 0 && sy < a.IH && sxx >= 0 && sxx < a.IW) v = *reinterpret_cast<const u32x4v*>(in + (((size_t)b * a.IH + sy) * a.IW + sxx) * 16 + 8 * h);
            *reinterpret_cast<u32x4v*>(sx + (wr * winw + wp) * 32 + 16 * h) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pxl = 32 * ks + 8 * kq + q2;   // this lane's pixel of the K-step (and pxl + 4)
            const char* ap = reinterpret_cast<const char*>(sd) + pxl * 32 + 8 * p2;
            const frag af = g_tr_frag<H>(ap, ap + 4 * 32);
#pragma unroll
            for (int t = 0; t < THIN_MAXT; ++t) {
                const int tap = wave + 4 * t;
                if (tap < taps) {
                    const int ky = tap / a.KW, kx = tap - ky * a.KW;
                    const char* bp = reinterpret_cast<const char*>(sx) + ((ky * winw + pxl * a.stride + kx) * 32) + 8 * p2;
                    const frag bf = g_tr_frag<H>(bp, bp + 4 * a.stride * 32);
                    acc[t] = h16<H>::mfma16(af, bf, acc[t]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < THIN_MAXT; ++t) {
        const int tap = wave + 4 * t;
        if (tap < taps) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a.partial[(((size_t)blockIdx.x * taps + tap) * a.NC + nblk * 16 + 4 * kq + i) * 16 + r] = acc[t][i];
        }
    }
}

// dw[co][ci][ky][kx] (+)= sum over splits of partial[split][tap][co][ci]   (co < Cout, ci < Cin: the real extents)
// SL split lanes per output (1, 4 or 16, the host's choice by nsplit): thread (o, sl) sums the splits sl, sl + SL, ... of output o, the SL
// partial sums meet in the LDS in a fixed order
__global__ __launch_bounds__(256) void gconv_wreduce_kernel(const float* __restrict__ partial, int nsplit, int taps, int NC, int KC, float* __restrict__ dw,
                                                            int Cout, int Cin, int accumulate, int SL) {
    __shared__ float red[256];
    const size_t total = (size_t)Cout * Cin * taps;
    const int per = 256 / SL, o = threadIdx.x % per, sl = threadIdx.x / per;
    for (size_t base = (size_t)blockIdx.x * per; base < total; base += (size_t)gridDim.x * per) {   // (uniform trip count: barriers inside)
        const size_t i = base + o;
        float s = 0.f;
        if (i < total) {
            const int tap = (int)(i % taps), ci = (int)((i / taps) % Cin), co = (int)(i / ((size_t)taps * Cin));
            const float* pp = partial + ((size_t)tap * NC + co) * KC + ci;
            const size_t step = (size_t)taps * NC * KC;
            for (int sp = sl; sp < nsplit; sp += SL) s += pp[(size_t)sp * step];
        }
        if (SL > 1) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (sl == 0) {
                for (int k = 1; k < SL; ++k) s += red[k * per + o];
            }
            __syncthreads();
        }
        if (sl == 0 && i < total) dw[i] = (accumulate ? dw[i] : 0.f) + s;
    }
}

// pack w [Cout][Cin][KH][KW] f32 -> [taps][RP][CP] T with rows = Cout, cols = Cin (transpose == 0) or rows = Cin, cols = Cout (== 1,
// the dgrad operand); padding rows / columns are zero
template <typename T>
__global__ void gconv_pack_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int taps, int RP, int CP, int transpose) {
    const size_t total = (size_t)taps * RP * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CP), r = (int)((i / CP) % RP), tap = (int)(i / ((size_t)CP * RP));
        const int co = transpose ? c : r, ci = transpose ? r : c;
        wp[i] = from_f32<T>((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f);
    }
}

// column sums of x [npix][C] (the bias gradient), two deterministic stages: grid (64-channel blocks, pixel splits) -> part [nsplit][C], then
// out[c] (+)= sum over the splits in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void gcolsum_kernel(const T* __restrict__ x, size_t npix, int C, float* __restrict__ part) {
    // a thread owns one 16-byte vector (VE channels) of the block's cw <= 64 channels and every PL-th pixel of the split; a narrow block
    // (C = 32: 4 vectors) spreads its threads over more pixel lanes instead of idling; four loads in flight per thread
    constexpr int VE = 16 / sizeof(T), MAXPL = 256 / (16 / VE);
    const int cw = min(64, C - (int)blockIdx.x * 64), nv = cw / VE, PL = 256 / nv;
    const int v = threadIdx.x % nv, pl = threadIdx.x / nv;
    const int c0 = blockIdx.x * 64 + v * VE;
    const size_t per = (npix + gridDim.y - 1) / gridDim.y, p0 = (size_t)blockIdx.y * per, p1 = p0 + per < npix ? p0 + per : npix;
    float acc[4][VE];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[u][e] = 0.f;
    if (pl < PL) {
        size_t p = p0 + pl;
        for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {
            vec16<T> t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec16<T>*>(x + (p + (size_t)u * PL) * C + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[u][e] += t[u].get(e);
        }
        for (; p < p1; p += PL) {
            const vec16<T> t = *reinterpret_cast<const vec16<T>*>(x + p * C + c0);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[0][e] += t.get(e);
        }
    }
    __shared__ float s[MAXPL][65];
    if (pl < PL) {
#pragma unroll
        for (int e = 0; e < VE; ++e) s[pl][v * VE + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    }
    __syncthreads();
    if ((int)threadIdx.x < cw) {
        float t = 0.f;
        for (int k = 0; k < PL; ++k) t += s[k][threadIdx.x];
        part[(size_t)blockIdx.y * C + blockIdx.x * 64 + threadIdx.x] = t;
    }
}
// out[c] (+)= sum over the splits: 64 channels x 16 split lanes per workgroup, fixed order
__global__ __launch_bounds__(1024) void gcolsum_reduce_kernel(const float* __restrict__ part, int nsplit, int C, float* __restrict__ out, int Creal, int accumulate) {
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, sub = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Creal) {
        int k = sub;
        for (; k + 48 < nsplit; k += 64) {   // four loads in flight
            const float r0 = part[(size_t)k * C + c], r1 = part[(size_t)(k + 16) * C + c], r2 = part[(size_t)(k + 32) * C + c], r3 = part[(size_t)(k + 48) * C + c];
            s += r0; s += r1; s += r2; s += r3;
        }
        for (; k < nsplit; k += 16) s += part[(size_t)k * C + c];
    }
    __shared__ float sh[16][64];
    sh[sub][cl] = s;
    __syncthreads();
    if (sub == 0 && c < Creal) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += sh[k][cl];
        out[c] = (accumulate ? out[c] : 0.f) + r;
    }
}
inline int colsum_nsplit(size_t npix) {   // >= 256 pixels per split, at most 512 splits (2 workgroups per CU; the reduce reads them all from one)
    const size_t n = (npix + 255) / 256;
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

inline int grid1(size_t n, int cap = 4096) {
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

int check_geo(const char* name, int B, int IH, int IW, int KC, int OH, int OW, int NC, int KH, int KW, int stride, int pad, int dtype) {
    WM_REQUIRE(B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && KC > 0 && NC > 0, WM_E_BADARG, "%s: bad shape", name);
    WM_REQUIRE(KH >= 1 && KH <= 5 && KW >= 1 && KW <= 5 && (stride == 1 || stride == 2) && pad >= 0 && pad <= 4, WM_E_SHAPE,
               "%s: kernel %dx%d stride %d pad %d unsupported (<= 5x5, stride 1 or 2)", name, KH, KW, stride, pad);
    WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16 || dtype == WM_F16, WM_E_BADARG, "%s: unsupported dtype %d", name, dtype);
    const int cm = dtype == WM_F32 ? 4 : 16;
    WM_REQUIRE(KC % cm == 0 && NC % 16 == 0, WM_E_SHAPE, "%s: channel strides KC=%d NC=%d must be multiples of %d / 16 (pad the channels)", name, KC, NC, cm);
    return WM_OK;
}

}  // namespace

extern "C" int wm_gconv_pack(const float* w, void* wp, int Cout, int Cin, int KH, int KW, int RP, int CP, int transpose, int dtype, void* stream) {
    WM_REQUIRE(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, WM_E_BADARG, "wm_gconv_pack: bad arguments");
    WM_REQUIRE(RP >= (transpose ? Cin : Cout) && CP >= (transpose ? Cout : Cin), WM_E_BADARG, "wm_gconv_pack: padded extents smaller than the weight");
    const size_t total = (size_t)KH * KW * RP * CP;
    hipStream_t s = (hipStream_t)stream;
    WM_DISPATCH_DTYPE(dtype, "wm_gconv_pack",
        hipLaunchKernelGGL(gconv_pack_kernel<T>, dim3(grid1(total)), dim3(256), 0, s, w, (T*)wp, Cout, Cin, KH * KW, RP, CP, transpose));
    WM_LAUNCH_CHECK("wm_gconv_pack");
    return WM_OK;
}

// forward (dgrad == 0): in [B,IH,IW,KC], w [KH*KW][NC][KC] (wm_gconv_pack transpose 0), out [B,OH,OW,NC], OH = (IH + 2 pad - KH)/stride + 1.
// input gradient / ConvTranspose2d forward (dgrad == 1): in = dout [B,IH,IW,KC] on the conv's OUTPUT grid, w [KH*KW][NC][KC] (pack with
// transpose 1: rows = the conv's input channels), out = din [B,OH,OW,NC] on the conv's INPUT grid.
// nn.Linear on a handful of rows (the QF embedding MLPs, the predictor heads: [B <= 64][K] x [N][K]^T): the direct kernel above walks K
// serially with five dependent global loads per 4-wide MFMA step (38 us for 24 x 512 x 512).  Here a wave owns one output column n, its
// lanes stride K (coalesced filter row), the rows' partial dots are reduced across the wave; out[p][n] = bias[n] + sum_k in[p][k] w[n][k].
template <typename T, int PMAX>
__global__ __launch_bounds__(256) void glinear_small_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                                                            int P, int KC, int NC) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= NC) return;
    float acc[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) acc[p] = 0.f;
    for (int k = lane; k < KC; k += 64) {
        const float wv = to_f32(w[(size_t)n * KC + k]);
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (p < P) acc[p] = __builtin_fmaf(to_f32(in[(size_t)p * KC + k]), wv, acc[p]);
    }
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        if (p < P) {      // (wave-uniform)
            const float t = wave_sum(acc[p]);
            if (lane == 0) out[(size_t)p * NC + n] = from_f32<T>(t + bv);
        }
    }
}

extern "C" int wm_gconv_fwd(const void* in, const void* w, const float* bias, void* out, int B, int IH, int IW, int KC, int OH, int OW, int NC,
                            int KH, int KW, int stride, int pad, int dgrad, int dtype, void* stream) {
    WM_REQUIRE(in && w && out, WM_E_BADARG, "wm_gconv_fwd: null pointer");
    int rc = check_geo("wm_gconv_fwd", B, IH, IW, KC, OH, OW, NC, KH, KW, stride, pad, dtype);
    if (rc) return rc;
    WM_REQUIRE((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) == 0, WM_E_SHAPE, "wm_gconv_fwd: pointers must be 16-byte aligned");
    GArgs a;
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.B = B; a.IH = IH; a.IW = IW; a.KC = KC; a.OH = OH; a.OW = OW; a.NC = NC;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dgrad = dgrad ? 1 : 0;
    const size_t npix = (size_t)B * OH * OW;
    const dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((NC + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && npix <= 32 && IH == OH && IW == OW) {   // a Linear layer on a few rows (either direction)
        WM_DISPATCH_DTYPE(dtype, "wm_gconv_fwd",
            hipLaunchKernelGGL((glinear_small_kernel<T, 32>), dim3((unsigned)((NC + 3) / 4)), dim3(256), 0, s, (const T*)in, (const T*)w, bias, (T*)out, (int)npix, KC, NC));
        WM_LAUNCH_CHECK("wm_gconv_fwd(linear)");
        return WM_OK;
    }
    WM_DISPATCH_DTYPE(dtype, "wm_gconv_fwd", hipLaunchKernelGGL(gconv_kernel<T>, grid, dim3(256), 0, s, a));
    WM_LAUNCH_CHECK("wm_gconv_fwd");
    return WM_OK;
}

// pixel splits of the 16-bit kernel: enough workgroups to fill the chip, at most 64 slabs
static int wgrad16_nsplit(int B, int OH, int OW, int KC, int NC, int KH, int KW) {
    // >= 2048 workgroups in flight (24 KB of LDS each: 6 per CU hide the per-tap load -> LDS -> barrier latency), at least 4 chunks each;
    // the 5x5 3 -> 3 Bayar layer has 4 jobs and 24,576 chunks: at the former cap of 64 splits its weight gradient took 2.5 ms
    const size_t chunks = ((size_t)B * OH * OW + 63) / 64;
    const long jobs = (long)((NC + 63) / 64) * ((KC + 63) / 64) * ((KH * KW + WG_TAPS - 1) / WG_TAPS);
    const long target = jobs < 16 ? 2048 : 1024;   // (layers with many jobs keep the slab traffic of the splits down)
    long ns = (target + jobs - 1) / jobs;
    if (ns > 1024) ns = 1024;
    if ((size_t)ns > (chunks + 3) / 4) ns = (long)((chunks + 3) / 4);
    return (int)(ns < 1 ? 1 : ns);
}
static bool thin_ok(int KC, int NC, int KH, int KW, int stride) { return KC == 16 && NC % 16 == 0 && KH * KW <= 4 * THIN_MAXT && (stride == 1 || stride == 2); }
static int thin_chunks_x(int OW) { return (OW + 63) / 64; }
static int thin_nsplit(int B, int OH, int OW) {
    const size_t chunks = (size_t)B * OH * thin_chunks_x(OW);
    size_t ns = (chunks + 7) / 8;   // >= 8 chunks per workgroup
    if (ns > 1024) ns = 1024;
    return (int)(ns < 1 ? 1 : ns);
}
extern "C" int wm_gconv_wgrad_nsplit(int B, int OH, int OW, int KC, int NC, int KH, int KW) {
    const size_t npix = (size_t)B * OH * OW;
    const long jobs = (long)KH * KW * (NC / 16) * ((KC + 15) / 16);
    long ns = jobs > 0 ? (4096 + jobs - 1) / jobs : 1;
    const long cap = (long)((npix + 255) / 256);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return (int)ns;
}

extern "C" size_t wm_gconv_wgrad_scratch_floats(int B, int OH, int OW, int KC, int NC, int KH, int KW) {
    const int ns = std::max(std::max(wm_gconv_wgrad_nsplit(B, OH, OW, KC, NC, KH, KW), wgrad16_nsplit(B, OH, OW, KC, NC, KH, KW)), thin_nsplit(B, OH, OW));   // any kernel
    const size_t a = (size_t)ns * KH * KW * NC * KC, b = (size_t)colsum_nsplit((size_t)B * OH * OW) * NC;
    return a > b ? a : b;
}
extern "C" size_t wm_gcolsum_scratch_floats(size_t npix, int C) { return (size_t)colsum_nsplit(npix) * C; }

// dw [Cout][Cin][KH][KW] f32 (+)= the weight gradient; dbias [Cout] (+)= column sums of dout (may be NULL).
// partial: f32 scratch of wm_gconv_wgrad_scratch_floats(..) floats.
extern "C" int wm_gconv_wgrad(const void* dout, const void* in, float* partial, float* dw, float* dbias, int accumulate, int B, int IH, int IW,
                              int KC, int OH, int OW, int NC, int KH, int KW, int stride, int pad, int Cout, int Cin, int dtype, void* stream) {
    WM_REQUIRE(dout && in && partial && dw, WM_E_BADARG, "wm_gconv_wgrad: null pointer");
    int rc = check_geo("wm_gconv_wgrad", B, IH, IW, KC, OH, OW, NC, KH, KW, stride, pad, dtype);
    if (rc) return rc;
    WM_REQUIRE(Cout > 0 && Cout <= NC && Cin > 0 && Cin <= KC, WM_E_BADARG, "wm_gconv_wgrad: Cout / Cin exceed the tensors' channel strides");
    GWArgs a;
    a.dout = dout; a.in = in; a.partial = partial; a.B = B; a.IH = IH; a.IW = IW; a.KC = KC; a.OH = OH; a.OW = OW; a.NC = NC;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WM_F32) {
        a.nsplit = wm_gconv_wgrad_nsplit(B, OH, OW, KC, NC, KH, KW);
        const int jobs = KH * KW * (NC / 16) * ((KC + 15) / 16);
        hipLaunchKernelGGL(gconv_wgrad_kernel<float>, dim3((unsigned)jobs, (unsigned)a.nsplit), dim3(256), 0, s, a);
    } else if (thin_ok(KC, NC, KH, KW, stride)) {
        a.nsplit = thin_nsplit(B, OH, OW);
        const int winw = 63 * stride + KW;
        const size_t lds = 64 * 32 + (size_t)KH * winw * 32;
        const dim3 grid((unsigned)a.nsplit, (unsigned)(NC / 16));
        if (dtype == WM_BF16) hipLaunchKernelGGL(gconv_wgrad_thin_kernel<bf16_t>, grid, dim3(256), lds, s, a, thin_chunks_x(OW), winw);
        else hipLaunchKernelGGL(gconv_wgrad_thin_kernel<f16_t>, grid, dim3(256), lds, s, a, thin_chunks_x(OW), winw);
    } else {
        a.nsplit = wgrad16_nsplit(B, OH, OW, KC, NC, KH, KW);
        const int tapgroups = (KH * KW + WG_TAPS - 1) / WG_TAPS;
        const dim3 grid((unsigned)a.nsplit, (unsigned)(((NC + 63) / 64) * ((KC + 63) / 64) * tapgroups));
        if (dtype == WM_BF16) hipLaunchKernelGGL(gconv_wgrad16_kernel<bf16_t>, grid, dim3(256), 0, s, a, tapgroups);
        else hipLaunchKernelGGL(gconv_wgrad16_kernel<f16_t>, grid, dim3(256), 0, s, a, tapgroups);
    }
    WM_LAUNCH_CHECK("wm_gconv_wgrad");
    {
        const int SL = a.nsplit >= 64 ? 16 : a.nsplit >= 8 ? 4 : 1;
        const size_t outs = (size_t)Cout * Cin * KH * KW, per = 256 / SL;
        const size_t g = (outs + per - 1) / per;
        hipLaunchKernelGGL(gconv_wreduce_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, s, partial, a.nsplit, KH * KW, NC, KC, dw, Cout, Cin,
                           accumulate, SL);
    }
    WM_LAUNCH_CHECK("wm_gconv_wgrad(reduce)");
    if (dbias) {   // (the slab partials have been consumed by the reduce above: the scratch is free again)
        const size_t npix = (size_t)B * OH * OW;
        const int ns = colsum_nsplit(npix);
        WM_DISPATCH_DTYPE(dtype, "wm_gconv_wgrad(bias)",
            hipLaunchKernelGGL(gcolsum_kernel<T>, dim3((unsigned)((NC + 63) / 64), (unsigned)ns), dim3(256), 0, s, (const T*)dout, npix, NC, partial));
        hipLaunchKernelGGL(gcolsum_reduce_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0, s, partial, ns, NC, dbias, Cout, accumulate);
        WM_LAUNCH_CHECK("wm_gconv_wgrad(bias)");
    }
    return WM_OK;
}

// out [Creal] f32 (+)= column sums of x [npix][C] (the bias gradient of a ConvTranspose2d / Linear whose dout is not a wgrad operand);
// scratch: wm_gcolsum_scratch_floats(npix, C) floats
extern "C" int wm_gcolsum(const void* x, size_t npix, int C, float* out, int Creal, int accumulate, float* scratch, int dtype, void* stream) {
    WM_REQUIRE(x && out && scratch && npix > 0 && C > 0 && Creal > 0 && Creal <= C, WM_E_BADARG, "wm_gcolsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int ns = colsum_nsplit(npix);
    WM_DISPATCH_DTYPE(dtype, "wm_gcolsum",
        hipLaunchKernelGGL(gcolsum_kernel<T>, dim3((unsigned)((C + 63) / 64), (unsigned)ns), dim3(256), 0, s, (const T*)x, npix, C, scratch));
    hipLaunchKernelGGL(gcolsum_reduce_kernel, dim3((unsigned)((Creal + 63) / 64)), dim3(1024), 0, s, scratch, ns, C, out, Creal, accumulate);
    WM_LAUNCH_CHECK("wm_gcolsum");
    return WM_OK;
}
