// Weight gradient of the 3x3 / stride 1 / pad 1 convolution on MFMA (gfx950).
//
//   dW[co][ci][kh][kw] = sum_{b,h,w} a[b, h+kh-1, w+kw-1, ci] * dy[b, h, w, co]
//
// (what autograd computes for nn.Conv2d in hidden_models/conv_bn_relu.py:11 and
// network/UNet.py:67-97), with a = x or the fused relu(in_scale*x+in_shift) of the producer.
//
// GEMM view per tap: M = Cin, N = Cout, K = B*H*W pixels -- the reduction runs over pixels, the
// slow index of both NHWC operands, so both MFMA operands need an [k = pixel][row/col = channel]
// -> k-contiguous transpose.  On gfx950 that is free: ds_read_b64_tr_b16 reads a 4 pixel x 16
// channel LDS block and hands each lane 4 consecutive pixels of one channel, which is exactly
// the v_mfma_f32_32x32x16_bf16 operand layout (two reads per 8-deep fragment).
//
//   workgroup  = 256 threads, loops over 16x16-pixel tiles (grid-stride), one (64 ci x 64 co)
//                channel block per blockIdx.y; stages the 18x18x64 input halo tile and the
//                16x16x64 dy tile in LDS per tile
//   wave (mi,ni) owns the 32 ci x 32 co block for all 9 taps: 9 accumulators (144 VGPRs) that
//                live across the whole tile loop; per K step (16 pixels of a tile row) it reads
//                the dy fragment once and 9 shifted x fragments
//   epilogue   = f32 slab [9][CinP][CoutP] per workgroup; a second kernel sums the slabs in a
//                fixed order (deterministic, no float atomics) into the PyTorch-layout gradient.
//   f32 path   : v_mfma_f32_32x32x2_f32 on 8x16 tiles (parity path; plain ds_read_b32 operands).
#include <stdlib.h>
#include <type_traits>
#include "wm_common.h"

// wave-specialised bf16 kernel (wgrad_ws.hip)
#define WM_DECL_WGWS(sfx)                                                                                                              \
    void wm_launch_wgrad_ws##sfx(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const void* dy, int lddy,  \
                                 int CoutY, float* ws, int B, int H, int W, int nslabs, hipStream_t s, int reverse,                       \
                                 const void* yb = nullptr, int ldyb = 0, const float* bstats4 = nullptr, int bstats_ld = 0,               \
                                 const float* bcoef = nullptr, const float* gvec = nullptr)
WM_DECL_WGWS(_bf16);
WM_DECL_WGWS(_f16);
// the two compilations of wgrad_ws.hip, by activation dtype (WM_BF16 / WM_F16)
template <typename... A> static inline void wm_launch_wgrad_ws(int dtype, A... args) {
    if (dtype == WM_F16) wm_launch_wgrad_ws_f16(args...);
    else wm_launch_wgrad_ws_bf16(args...);
}
static inline bool is16(int dtype) { return dtype == WM_BF16 || dtype == WM_F16; }
// the same with the two GEMMs on different waves (bwd_ws8.hip: whole-tile shapes, premasked gradient), compiled twice
#define WM_DECL_BWDWS8(sfx)                                                                                                            \
    int wm_launch_bwd_ws8##sfx(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt,     \
                                const void* xr, const float* in_scale, const float* in_shift, void* dx, float* stat, float* ws, int B, \
                                int H, int W, int nwg, int reverse, hipStream_t s, int premasked, const float* gvec, int gv_ld, int stamps)
WM_DECL_BWDWS8(_bf16);
WM_DECL_BWDWS8(_f16);
// fused input + weight gradient of an image-fed first layer (bwd_ws16.hip), compiled twice
void wm_launch_bwd_ws16_bf16(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt, const void* x,
                             void* dx, float* ws, int B, int H, int W, int nwg, int reverse, hipStream_t s, int premasked);
void wm_launch_bwd_ws16_f16(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt, const void* x,
                            void* dx, float* ws, int B, int H, int W, int nwg, int reverse, hipStream_t s, int premasked);
// fused input + weight gradient of a 64 -> 64 body layer (bwd_ws.hip), compiled twice like the above
#define WM_DECL_BWDWS(sfx)                                                                                                             \
    void wm_launch_bwd_ws##sfx(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt,      \
                               const void* xr, const float* in_scale, const float* in_shift, void* dx, float* stat, float* ws, int B,  \
                               int H, int W, int nwg, int reverse, hipStream_t s, int dbg, int premasked, const float* gvec, int gv_ld); \
    int wm_bwd_ws_gvec_max_batch##sfx()
WM_DECL_BWDWS(_bf16);
WM_DECL_BWDWS(_f16);


namespace {

constexpr int TW = 16;
constexpr int HW_ = TW + 2;
constexpr int CB = 64;  // channel block (both ci and co)

template <typename T> struct WCfg;
template <> struct WCfg<bf16_t> { static constexpr int TH = 16, VE = 8, PS = 72; };   // 144-byte pixel rows
template <> struct WCfg<f16_t> : WCfg<bf16_t> {};
template <> struct WCfg<float>  { static constexpr int TH = 8,  VE = 4, PS = 68; };   // 272-byte pixel rows

template <typename T>
struct WgArgs {
    const T* x; int ldx; int CinX;
    const float* in_scale; const float* in_shift;
    const T* dy; int lddy; int CoutY;
    float* ws;           // [gridDim.x][9][CinP][CoutP]
    int B, H, W;
    int tilesX, tilesY, ntiles;
    int ciBlocks, coBlocks;
};

template <typename T>
__device__ __forceinline__ typename h16<T>::x8 tr_frag(const T* p0, const T* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename h16<T>::x8, v);
}

template <typename T, bool XFORM>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgArgs<T> a) {
    constexpr int TH = WCfg<T>::TH, VE = WCfg<T>::VE, PS = WCfg<T>::PS;
    constexpr int HH = TH + 2;
    constexpr int VPP = CB / VE;
    constexpr int X_ELEMS = HH * HW_ * PS;
    constexpr int D_ELEMS = TH * TW * PS;
    __shared__ __attribute__((aligned(16))) T smem[X_ELEMS + D_ELEMS];
    T* sX = smem;
    T* sD = smem + X_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = wave >> 1, ni = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int cc = blockIdx.y / a.coBlocks, oc = blockIdx.y % a.coBlocks;
    const int ci0 = cc * CB, co0 = oc * CB;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int vv = tid % VPP;
    const int cx = ci0 + vv * VE;      // this thread's x channel group (fixed: 256 % VPP == 0)
    const int cd = co0 + vv * VE;
    const bool cxok = cx < a.CinX, cdok = cd < a.CoutY;
    float sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
    if (XFORM && cxok) {
#pragma unroll
        for (int e = 0; e < VE; ++e) { sc[e] = a.in_scale[cx + e]; sh[e] = a.in_shift[cx + e]; }
    }

    // ---- software pipeline over this workgroup's contiguous run of tiles (same scheme as the persistent conv
    // kernel, measured there with in-kernel phase stamps): the x halo tile and the dy tile travel two tiles ahead
    // in registers; their global loads are issued from INSIDE the MFMA loop, one every few MFMAs, so the
    // per-CU memory queue never backs up and the HBM time hides under the matrix pipe; the fused BN+ReLU of the
    // next tile's registers is done in the same loop.  Loads are never under a per-lane branch (clamped, always
    // valid address; validity travels as bits).
    constexpr int NXV = (HH * HW_ * VPP + 255) / 256;
    constexpr int NDV = TH * TW * VPP / 256;
    const int cxl = cxok ? cx : 0, cdl = cdok ? cd : 0;
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = tile;
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };
    auto load_x = [&](const TileGeo& g, int it, vec16<T>& dst, unsigned& okbits) {
        const int i = tid + it * 256;
        const int pix = min(i / VPP, HH * HW_ - 1);
        const int py = pix / HW_, px = pix - py * HW_;
        const int gy = g.ty0 - 1 + py, gx = g.tx0 - 1 + px;
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        dst = *reinterpret_cast<const vec16<T>*>(a.x + ((size_t)(g.b * a.H + gyc) * a.W + gxc) * a.ldx + cxl);
        const unsigned okb = (cxok && gy == gyc && gx == gxc) ? 1u : 0u;
        okbits |= okb << it;
    };
    auto load_d = [&](const TileGeo& g, int it, vec16<T>& dst, unsigned& okbits) {
        const int i = tid + it * 256;
        const int pix = i / VPP;
        const int py = pix / TW, px = pix - py * TW;
        const int gy = g.ty0 + py, gx = g.tx0 + px;
        const int gyc = min(gy, a.H - 1), gxc = min(gx, a.W - 1);
        dst = *reinterpret_cast<const vec16<T>*>(a.dy + ((size_t)(g.b * a.H + gyc) * a.W + gxc) * a.lddy + cdl);
        const unsigned okb = (cdok && gy == gyc && gx == gxc) ? 1u : 0u;
        okbits |= okb << it;
    };
    auto xform_x = [&](vec16<T>& v, bool ok) {   // fused BN+ReLU of the producer, zero padding after it
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float f = v.get(e);
            if (XFORM) f = fmaxf(sc[e] * f + sh[e], 0.f);
            v.set(e, ok ? f : 0.f);
        }
    };
    auto mask_d = [&](vec16<T>& v, bool ok) {
#pragma unroll
        for (int e = 0; e < VE; ++e) v.set(e, ok ? v.get(e) : 0.f);
    };

    const int per = (a.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    // XCD-aware run assignment (workgroups b and b+8 share an XCD): XCD x walks consecutive runs
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int t_begin = run * per, t_end = min(a.ntiles, t_begin + per);

    // one register set (a second one spills next to the 144 accumulator registers): the loads of tile+1 are
    // issued in the first 19 of the 32 slots of the MFMA loop of tile, so the last of them still has ~40 % of
    // the loop (>3,000 cycles) to land before the LDS write that consumes it
    vec16<T> nxtx[NXV], nxtd[NDV];
    unsigned nokx = 0, nokd = 0;
    if (t_begin < t_end) {
        const TileGeo g0 = geo(t_begin);
#pragma unroll
        for (int it = 0; it < NXV; ++it) load_x(g0, it, nxtx[it], nokx);
#pragma unroll
        for (int it = 0; it < NDV; ++it) load_d(g0, it, nxtd[it], nokd);
    }

    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // previous tile consumed
#pragma unroll
        for (int it = 0; it < NXV; ++it) {
            const int i = tid + it * 256;
            xform_x(nxtx[it], (nokx >> it) & 1u);
            if (i < HH * HW_ * VPP) *reinterpret_cast<vec16<T>*>(sX + (i / VPP) * PS + vv * VE) = nxtx[it];
        }
#pragma unroll
        for (int it = 0; it < NDV; ++it) {
            mask_d(nxtd[it], (nokd >> it) & 1u);
            *reinterpret_cast<vec16<T>*>(sD + ((tid + it * 256) / VPP) * PS + vv * VE) = nxtd[it];
        }
        __syncthreads();
        nokx = 0; nokd = 0;
        const bool have1 = tile + 1 < t_end;
        const TileGeo g1 = geo(have1 ? tile + 1 : tile);
        // slot s of the MFMA loop: global load s of tile+1 (x vectors first, then dy)
        auto slot = [&](int sl, auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            if (sl < NXV) {
                if (STEADY || have1) load_x(g1, sl, nxtx[sl], nokx);
            } else if (sl < NXV + NDV) {
                if (STEADY || have1) load_d(g1, sl - NXV, nxtd[sl - NXV], nokd);
            }
        };

        auto mfma_loop = [&](auto steady_tag) {
            if constexpr (sizeof(T) == 2) {
                // transposing-read lane geometry: 16-lane group g = lane>>4; lane i = 4q+p of the group
                const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
                const int chan = 16 * (g & 1) + 4 * p;  // channel offset inside the 32-channel block
                const int pk = 8 * (g >> 1) + q;        // pixel offset inside the 16-pixel K step (+4 for the 2nd read)
#pragma unroll
                for (int kr = 0; kr < TH; ++kr) {
                    const T* pd = sD + (kr * TW + pk) * PS + ni * 32 + chan;
                    const auto bfrag = tr_frag(pd, pd + 4 * PS);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int kh = tap / 3, kw = tap % 3;
                        const T* px_ = sX + ((kr + kh) * HW_ + pk + kw) * PS + mi * 32 + chan;
                        const auto afrag = tr_frag(px_, px_ + 4 * PS);
                        if (tap == 2) slot(2 * kr, steady_tag);
                        if (tap == 6) slot(2 * kr + 1, steady_tag);
                        acc[tap] = h16<T>::mfma32(afrag, bfrag, acc[tap]);
                    }
                }
            } else {
                // f32 parity path: the loads of the next tile go out as one burst ahead of the loop (a fully
                // unrolled loop with in-loop slots spills next to 144 accumulator registers)
#pragma unroll
                for (int sl = 0; sl < NXV + NDV; ++sl) slot(sl, steady_tag);
                for (int kr = 0; kr < TH; ++kr) {
                    for (int kp = 0; kp < TW / 2; ++kp) {
                        const int px = 2 * kp + h;
                        const float bfrag = sD[(kr * TW + px) * PS + ni * 32 + r];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) {
                            const int kh = tap / 3, kw = tap % 3;
                            const float afrag = sX[((kr + kh) * HW_ + px + kw) * PS + mi * 32 + r];
                            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag, acc[tap], 0, 0, 0);
                        }
                    }
                }
            }
        };
        if (have1) mfma_loop(std::true_type{});
        else mfma_loop(std::false_type{});
    }
    // slab write: acc[tap][i] -> row (ci) = (i&3)+8*(i>>2)+4h, col (co) = r
    const int CinP = a.ciBlocks * CB, CoutP = a.coBlocks * CB;
    float* slab = a.ws + (size_t)blockIdx.x * 9 * CinP * CoutP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            slab[((size_t)tap * CinP + ci0 + mi * 32 + row) * CoutP + co0 + ni * 32 + r] = acc[tap][i];
        }
}

// dw[co][ci][kh][kw] (+)= sum_slab ws[slab][tap][perm(ci)][co]
// A workgroup owns 32 quads of 4 consecutive co and splits the slabs 8 ways: every thread streams nslabs/8 slabs with
// 16-byte loads, 8 of them in flight; the 8 partial sums meet in LDS (fixed order: deterministic).
constexpr int RSPLIT = 8, RQUADS = 256 / RSPLIT;
// Workgroups >= nred carry a rider: the BatchNorm-backward finalisation of another layer (WmBnBwdFin, 8 channels each) -- in a
// backward sweep the sums of layer l-1 are ready when layer l's weight gradient runs, and a launch of its own would cost 5 us.
// A second rider (the LAST workgroup when `cs.partials` is set): the column sums of per-workgroup partial rows -- the bias gradient of a
// conv + ELU layer whose input-gradient kernel left them (wm_conv3x3_dgrad_elufused) -- folded in a fixed order.
struct ColsumRider { const float* partials; int nparts, C; float* out; int accumulate; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nslabs, int CinP, int CoutP,
                                                           float* __restrict__ dw, int Cin, int Cout,
                                                           const int* __restrict__ perm, int accumulate, int nred, WmBnBwdFin fin, ColsumRider cs) {
    __shared__ float4 red[RSPLIT][RQUADS];
    static_assert(sizeof(float4) * RSPLIT * RQUADS >= sizeof(double) * 2 * 32 * 8, "the rider's LDS fits in the reduction's");
    if (cs.partials && blockIdx.x == gridDim.x - 1) {
        // thread (cq, grp): channels 4 cq .. 4 cq + 3 of the rows grp, grp + 16, ...: 16-byte loads, four in flight; the 16 groups are then
        // folded in a fixed order (C a multiple of 4: the rows are those of a 16-bit activation)
        float4* r4 = &red[0][0];
        for (int c0 = 0; c0 < cs.C; c0 += 64) {
            const int cq = threadIdx.x & 15, grp = threadIdx.x >> 4, c = c0 + 4 * cq;
            float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            if (c < cs.C) {
                const float* base = cs.partials + c;
                int k = grp;
                for (; k + 48 < cs.nparts; k += 64) {
                    const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)k * cs.C);
                    const float4 v1 = *reinterpret_cast<const float4*>(base + (size_t)(k + 16) * cs.C);
                    const float4 v2 = *reinterpret_cast<const float4*>(base + (size_t)(k + 32) * cs.C);
                    const float4 v3 = *reinterpret_cast<const float4*>(base + (size_t)(k + 48) * cs.C);
                    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
                }
                for (; k < cs.nparts; k += 16) {
                    const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)k * cs.C);
                    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                }
            }
            r4[threadIdx.x] = float4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
            __syncthreads();
            if (threadIdx.x < 64 && c0 + (int)threadIdx.x < cs.C) {
                const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) sum += reinterpret_cast<const float*>(&r4[g * 16 + q])[e];
                float* o = cs.out + c0 + threadIdx.x;
                *o = (cs.accumulate ? *o : 0.f) + sum;
            }
            __syncthreads();
        }
        return;
    }
    if ((int)blockIdx.x >= nred) {
        wm_bn_bwd_finalize_block(fin, (int)blockIdx.x - nred, reinterpret_cast<double*>(&red[0][0]));
        return;
    }
    const size_t slab_elems = (size_t)9 * CinP * CoutP;
    const size_t nquads = slab_elems / 4;
    const int ql = threadIdx.x % RQUADS, sp = threadIdx.x / RQUADS;
    for (size_t q0 = (size_t)blockIdx.x * RQUADS; q0 < nquads; q0 += (size_t)nred * RQUADS) {
        const size_t qd = q0 + ql;
        float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (qd < nquads) {
            const float* base = ws + qd * 4;
            int k = sp;
            for (; k + 3 * RSPLIT < nslabs; k += 4 * RSPLIT) {
                const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)k * slab_elems);
                const float4 v1 = *reinterpret_cast<const float4*>(base + (size_t)(k + RSPLIT) * slab_elems);
                const float4 v2 = *reinterpret_cast<const float4*>(base + (size_t)(k + 2 * RSPLIT) * slab_elems);
                const float4 v3 = *reinterpret_cast<const float4*>(base + (size_t)(k + 3 * RSPLIT) * slab_elems);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; k < nslabs; k += RSPLIT) {
                const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)k * slab_elems);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            }
        }
        red[sp][ql] = float4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                             (a0.w + a1.w) + (a2.w + a3.w)};
        __syncthreads();
        if (threadIdx.x < 4 * RQUADS) {
            const int qq = threadIdx.x >> 2, e = threadIdx.x & 3;
            const size_t i = (q0 + qq) * 4 + e;
            if (i < slab_elems) {
                double sum = 0.0;
#pragma unroll
                for (int t = 0; t < RSPLIT; ++t) sum += (double)reinterpret_cast<const float*>(&red[t][qq])[e];
                const int co = (int)(i % CoutP);
                const int cip = (int)((i / CoutP) % CinP);
                const int tap = (int)(i / ((size_t)CoutP * CinP));
                int ci = -1;
                if (co < Cout) {
                    if (!perm) ci = cip < Cin ? cip : -1;
                    else
                        for (int k2 = 0; k2 < Cin; ++k2)
                            if (perm[k2] == cip) { ci = k2; break; }
                }
                if (ci >= 0) {
                    float* o = dw + (((size_t)co * Cin + ci) * 9 + tap);
                    *o = (accumulate ? *o : 0.f) + (float)sum;
                }
            }
        }
        __syncthreads();
    }
}

inline int nslabs_for(int B, int H, int W);
static bool fin_rider_ok(const WmBnBwdFin* fin) {
    return !fin || (fin->partials && fin->gamma && fin->invstd && fin->coef && fin->nparts > 0 && fin->nparts <= 256 && fin->C > 0 && fin->CP >= fin->C);
}
// the slab reduction (+ an optional rider)
static int launch_wgrad_reduce(float* ws, int nslabs, int CinP, int CoutP, float* dw, int Cin, int Cout, const int* perm, int accumulate,
                               const WmBnBwdFin* fin, hipStream_t s, ColsumRider cs = ColsumRider{nullptr, 0, 0, nullptr, 0}) {
    const size_t slab_elems = (size_t)9 * CinP * CoutP;
    const size_t rb = (slab_elems / 4 + RQUADS - 1) / RQUADS;
    const int blocks = (int)(rb > 2048 ? 2048 : rb);
    WmBnBwdFin f = {};
    int extra = 0;
    if (fin) {
        if (!fin_rider_ok(fin)) return WM_E_BADARG;
        f = *fin;
        extra = wm_cdiv(fin->CP, 8);
    }
    if (cs.partials && (!cs.out || cs.nparts <= 0 || cs.C <= 0 || cs.C % 4 != 0 || (((uintptr_t)cs.partials) & 15) != 0)) return WM_E_BADARG;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks + extra + (cs.partials ? 1 : 0)), dim3(256), 0, s, ws, nslabs, CinP, CoutP, dw, Cin, Cout, perm,
                       accumulate, blocks, f, cs);
    return WM_OK;
}

// wide layers run (input blocks x output blocks) workgroups per slab: fewer slabs fill the chip as well, and the slab traffic (9 x CinP x
// CoutP floats each, written once and read once by the reduction) shrinks with them -- 151 MB per 128 x 128 layer at 256 slabs
inline int nslabs_ch(int B, int H, int W, int CinX, int CoutY);
#ifndef WM_MAX_WGS
#define WM_MAX_WGS 256           // as conv3x3.hip
#endif
inline int nslabs_for(int B, int H, int W) {
    const long n = (long)B * wm_cdiv(H, 16) * wm_cdiv(W, 16);
    return (int)(n < WM_MAX_WGS ? n : WM_MAX_WGS);
}

inline int nslabs_ch(int B, int H, int W, int CinX, int CoutY) {
    const int n = nslabs_for(B, H, W), cb = wm_cdiv(CinX, CB) * wm_cdiv(CoutY, CB);
    if (cb <= 1) return n;
    const int want = wm_cdiv(512, cb);
    return n < want ? n : (want < 1 ? 1 : want);
}

template <typename T>
void launch_wgrad(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const void* dy, int lddy,
                  int CoutY, float* ws, int B, int H, int W, hipStream_t s) {
    WgArgs<T> a;
    a.x = (const T*)x; a.ldx = ldx; a.CinX = CinX; a.in_scale = in_scale; a.in_shift = in_shift;
    a.dy = (const T*)dy; a.lddy = lddy; a.CoutY = CoutY; a.ws = ws; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, WCfg<T>::TH); a.ntiles = B * a.tilesX * a.tilesY;
    a.ciBlocks = wm_cdiv(CinX, CB); a.coBlocks = wm_cdiv(CoutY, CB);
    dim3 grid((unsigned)nslabs_ch(B, H, W, CinX, CoutY), (unsigned)(a.ciBlocks * a.coBlocks)), block(256);
    if (in_scale) hipLaunchKernelGGL((wgrad_kernel<T, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<T, false>), grid, block, 0, s, a);
}

}  // namespace

extern "C" int wm_conv3x3_wgrad_nslabs(int B, int H, int W) { return nslabs_for(B, H, W); }

extern "C" size_t wm_conv3x3_wgrad_ws_bytes(int B, int H, int W, int CinX, int CoutY) {
    return (size_t)nslabs_for(B, H, W) * 9 * (wm_cdiv(CinX, CB) * CB) * (wm_cdiv(CoutY, CB) * CB) * sizeof(float);
}

// weight gradient with the BatchNorm-backward APPLY pass fused (bf16, image-fed first layers: CinX <= 16): the layer's dy is
// never materialised -- it is formed from g (gradient wrt the ReLU output), y (raw conv output), the BatchNorm constants
// stats4 = [scale | shift | mean | invstd] (4 rows of CP) and coef = wm_bn_bwd_finalize's [3][CP] while the tile is staged
extern "C" int wm_conv3x3_wgrad_bnfused_supported(int CinX, int CoutY, int dtype) {
    static const bool off = WM_ENV_FLAG("WM_NO_WGRAD_FUSE");
    return (!off && is16(dtype) && CinX <= 16 && CoutY % 64 == 0) ? 1 : 0;
}

extern "C" int wm_conv3x3_wgrad_bnfused(const void* x, int ldx, int CinX, const void* g, int ldg, const void* y, int ldy, int CoutY,
                                        const float* stats4, const float* coef, float* ws, float* dw, int accumulate, int B, int H,
                                        int W, int Cin, int Cout, int dtype, void* stream) {
    WM_REQUIRE(x && g && y && stats4 && coef && ws && dw, WM_E_BADARG, "wm_conv3x3_wgrad_bnfused: null pointer");
    WM_REQUIRE(wm_conv3x3_wgrad_bnfused_supported(CinX, CoutY, dtype), WM_E_SHAPE, "wm_conv3x3_wgrad_bnfused: unsupported shape CinX=%d CoutY=%d dtype=%d", CinX, CoutY, dtype);
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && CinX >= Cin && CoutY >= Cout && ldx >= CinX && ldg >= CoutY && ldy >= CoutY &&
               ldx % 8 == 0 && ldg % 8 == 0 && ldy % 8 == 0, WM_E_SHAPE, "wm_conv3x3_wgrad_bnfused: bad shape / strides");
    hipStream_t s = (hipStream_t)stream;
    wm_launch_wgrad_ws(dtype, x, ldx, CinX, nullptr, nullptr, g, ldg, CoutY, ws, B, H, W, nslabs_for(B, H, W), s, 0, y, ldy, stats4, CoutY, coef);
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_bnfused");
    const int CinP = wm_cdiv(CinX, CB) * CB, CoutP = wm_cdiv(CoutY, CB) * CB;
    launch_wgrad_reduce(ws, nslabs_for(B, H, W), CinP, CoutP, dw, Cin, Cout, nullptr, accumulate, nullptr, s);
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_bnfused(reduce)");
    return WM_OK;
}

WM_KNOB_ON(g_fin_rider, "WM_NO_FIN_RIDER");
WM_KNOB_SETTER(wm_debug_fin_rider, g_fin_rider)   // A/B knob (tools/ab_step.py, debug build only)
extern "C" int wm_fin_rider_enabled(void) { return g_fin_rider; }

WM_KNOB_ON(g_gv_fuse, "WM_NO_GV_FUSE");
WM_KNOB_SETTER(wm_debug_gv_fuse, g_gv_fuse)   // A/B knob (tools/ab_step.py, debug build only)

extern "C" int wm_conv3x3_gvfused_supported(int CinX, int CoutY, int dtype) {
    return (g_gv_fuse && is16(dtype) && CinX == 64 && (CoutY == 64 || CoutY == 32)) ? 1 : 0;
}

extern "C" int wm_conv3x3_wgrad_gvfused_fin(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const float* gvec,
                                            const void* y, int ldy, int CoutY, const float* stats4, const float* coef, float* ws, float* dw,
                                            int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, const WmBnBwdFin* fin,
                                            void* stream) {
    WM_REQUIRE(x && in_scale && in_shift && gvec && y && stats4 && coef && ws && dw, WM_E_BADARG, "wm_conv3x3_wgrad_gvfused: null pointer");
    WM_REQUIRE(fin_rider_ok(fin), WM_E_BADARG, "wm_conv3x3_wgrad_gvfused: bad finalisation rider (null pointer, or more than 256 partial rows)");
    WM_REQUIRE(wm_conv3x3_gvfused_supported(CinX, CoutY, dtype), WM_E_SHAPE, "wm_conv3x3_wgrad_gvfused: unsupported shape CinX=%d CoutY=%d dtype=%d", CinX, CoutY, dtype);
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && CinX >= Cin && CoutY >= Cout && ldx >= CinX && ldy >= CoutY &&
               ldx % 8 == 0 && ldy % 8 == 0, WM_E_SHAPE, "wm_conv3x3_wgrad_gvfused: bad shape / strides");
    hipStream_t s = (hipStream_t)stream;
    wm_launch_wgrad_ws(dtype, x, ldx, CinX, in_scale, in_shift, nullptr, 0, CoutY, ws, B, H, W, nslabs_for(B, H, W), s, 0, y, ldy, stats4, CoutY, coef, gvec);
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_gvfused");
    const int CinP = wm_cdiv(CinX, CB) * CB, CoutP = wm_cdiv(CoutY, CB) * CB;
    WM_REQUIRE(launch_wgrad_reduce(ws, nslabs_for(B, H, W), CinP, CoutP, dw, Cin, Cout, nullptr, accumulate, fin, s) == WM_OK, WM_E_BADARG,
               "wm_conv3x3_wgrad_gvfused: bad finalisation rider");
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_gvfused(reduce)");
    return WM_OK;
}
extern "C" int wm_conv3x3_wgrad_gvfused(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const float* gvec,
                                        const void* y, int ldy, int CoutY, const float* stats4, const float* coef, float* ws, float* dw,
                                        int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    return wm_conv3x3_wgrad_gvfused_fin(x, ldx, CinX, in_scale, in_shift, gvec, y, ldy, CoutY, stats4, coef, ws, dw, accumulate, B, H, W, Cin, Cout,
                                        dtype, nullptr, stream);
}

extern "C" int wm_conv3x3_wgrad_fin(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift,
                                    const void* dy, int lddy, int CoutY, float* ws, float* dw, int accumulate, int B, int H,
                                    int W, int Cin, int Cout, const int* perm_dev, int dtype, const WmBnBwdFin* fin, int sweep_reverse,
                                    void* stream) {
    WM_REQUIRE(x && dy && ws && dw, WM_E_BADARG, "wm_conv3x3_wgrad: null pointer");
    WM_REQUIRE(fin_rider_ok(fin), WM_E_BADARG, "wm_conv3x3_wgrad: bad finalisation rider (null pointer, or more than 256 partial rows)");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && CinX > 0 && CoutY >= Cout, WM_E_BADARG, "wm_conv3x3_wgrad: bad shape");
    WM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), WM_E_BADARG, "wm_conv3x3_wgrad: in_scale/in_shift must come together");
    WM_REQUIRE(dtype == WM_F32 || is16(dtype), WM_E_BADARG, "wm_conv3x3_wgrad: unsupported dtype %d", dtype);
    const int ve = is16(dtype) ? 8 : 4, esz = is16(dtype) ? 2 : 4;
    WM_REQUIRE(CinX % ve == 0 && CoutY % ve == 0, WM_E_SHAPE, "wm_conv3x3_wgrad: channel counts %d/%d must be multiples of %d", CinX, CoutY, ve);
    WM_REQUIRE(ldx >= CinX && lddy >= CoutY && (ldx * esz) % 16 == 0 && (lddy * esz) % 16 == 0, WM_E_SHAPE,
               "wm_conv3x3_wgrad: bad pixel strides ldx=%d lddy=%d", ldx, lddy);
    WM_REQUIRE(perm_dev || CinX >= Cin, WM_E_BADARG, "wm_conv3x3_wgrad: x has fewer channels than the weight");
    hipStream_t s = (hipStream_t)stream;
    static const bool v1 = WM_ENV_FLAG("WM_WGRAD_V1");  // diagnostic knob (debug build): single-role kernel
    const int nsl = nslabs_ch(B, H, W, CinX, CoutY);
    if (is16(dtype) && !v1) wm_launch_wgrad_ws(dtype, x, ldx, CinX, in_scale, in_shift, dy, lddy, CoutY, ws, B, H, W, nsl, s, sweep_reverse ? 1 : 0);
    else if (dtype == WM_BF16) launch_wgrad<bf16_t>(x, ldx, CinX, in_scale, in_shift, dy, lddy, CoutY, ws, B, H, W, s);
    else if (dtype == WM_F16) launch_wgrad<f16_t>(x, ldx, CinX, in_scale, in_shift, dy, lddy, CoutY, ws, B, H, W, s);
    else launch_wgrad<float>(x, ldx, CinX, in_scale, in_shift, dy, lddy, CoutY, ws, B, H, W, s);
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad");
    const int CinP = wm_cdiv(CinX, CB) * CB, CoutP = wm_cdiv(CoutY, CB) * CB;
    WM_REQUIRE(launch_wgrad_reduce(ws, nsl, CinP, CoutP, dw, Cin, Cout, perm_dev, accumulate, fin, s) == WM_OK, WM_E_BADARG,
               "wm_conv3x3_wgrad: bad finalisation rider");
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad(reduce)");
    return WM_OK;
}
// the weight gradient of a conv + ELU layer (16-bit dtypes): dw as wm_conv3x3_wgrad from (x, gz), and in the SAME reduction launch the bias
// gradient db [Cout] (+)= the column sums of the partial rows wm_conv3x3_dgrad_elufused left (bias_partials f32 [nparts][CoutY])
extern "C" int wm_conv3x3_wgrad_bias(const void* x, int ldx, int CinX, const void* gz, int ldgz, int CoutY, float* ws, float* dw, int accumulate,
                                     int B, int H, int W, int Cin, int Cout, int dtype, const float* bias_partials, int nparts, float* db,
                                     int db_accumulate, void* stream) {
    WM_REQUIRE(x && gz && ws && dw && bias_partials && db, WM_E_BADARG, "wm_conv3x3_wgrad_bias: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && CinX >= Cin && CoutY >= Cout && nparts > 0, WM_E_BADARG, "wm_conv3x3_wgrad_bias: bad shape");
    WM_REQUIRE(is16(dtype), WM_E_BADARG, "wm_conv3x3_wgrad_bias: 16-bit dtypes only (got %d)", dtype);
    WM_REQUIRE(CinX % 8 == 0 && CoutY % 8 == 0 && ldx >= CinX && ldgz >= CoutY && (ldx * 2) % 16 == 0 && (ldgz * 2) % 16 == 0, WM_E_SHAPE,
               "wm_conv3x3_wgrad_bias: channel counts %d/%d must be multiples of 8 and the pixel strides cover them", CinX, CoutY);
    hipStream_t s = (hipStream_t)stream;
    const int nsl = nslabs_ch(B, H, W, CinX, CoutY);
    wm_launch_wgrad_ws(dtype, x, ldx, CinX, nullptr, nullptr, gz, ldgz, CoutY, ws, B, H, W, nsl, s, 0);
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_bias");
    const int CinP = wm_cdiv(CinX, CB) * CB, CoutP = wm_cdiv(CoutY, CB) * CB;
    // (the rows hold CoutY sums each; the first Cout of them are the bias gradient)
    WM_REQUIRE(launch_wgrad_reduce(ws, nsl, CinP, CoutP, dw, Cin, Cout, nullptr, accumulate, nullptr, s,
                                   ColsumRider{bias_partials, nparts, CoutY, db, db_accumulate}) == WM_OK, WM_E_BADARG, "wm_conv3x3_wgrad_bias: bad rider");
    WM_LAUNCH_CHECK("wm_conv3x3_wgrad_bias(reduce)");
    return WM_OK;
}
extern "C" int wm_conv3x3_wgrad(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift,
                                const void* dy, int lddy, int CoutY, float* ws, float* dw, int accumulate, int B, int H,
                                int W, int Cin, int Cout, const int* perm_dev, int dtype, void* stream) {
    return wm_conv3x3_wgrad_fin(x, ldx, CinX, in_scale, in_shift, dy, lddy, CoutY, ws, dw, accumulate, B, H, W, Cin, Cout, perm_dev, dtype, nullptr,
                                0, stream);
}

// ---- fused backward of a 64 -> 64 ConvBNRelu body layer (bwd_ws.hip): dx, the feeding layer's BatchNorm-backward partial rows and
// the weight gradient from one pass over (g, y, the feeding layer's raw output); then the slab reduction (+ an optional rider).
WM_KNOB_ON(g_bwdfuse, "WM_NO_BWD_FUSE");
WM_KNOB_SETTER(wm_debug_bwd_fuse, g_bwdfuse)   // A/B knob (tools/ab_step.py, debug build only)
WM_KNOB_INT(g_bwd_dbg, "WM_BWD_DBG", 0);
WM_KNOB_SETTER(wm_debug_bwd_variant, g_bwd_dbg)   // phase ablations of bwd_ws.hip (debug build only; results are then meaningless)
WM_KNOB_ON(g_bwd_split, "WM_NO_BWD_SPLIT");
WM_KNOB_SETTER(wm_debug_bwd_split, g_bwd_split)   // A/B knob (debug build only): 0 = every form on bwd_ws.hip
extern "C" int wm_conv3x3_bwd_fused_supported(int dtype) { return (g_bwdfuse && is16(dtype)) ? 1 : 0; }
// ... and the tensor fits the kernel's 32-bit element offsets / 24-bit row and column counts
// (byte offsets are 32-bit, and the last MB of the range stays free so that a "negative" (wrapped) halo offset lies outside the buffer
// descriptor's range: tensors up to 2^31 - 2^19 elements)
extern "C" int wm_conv3x3_bwd_fused_supported_shape(int B, int H, int W, int dtype) {
    if (!(wm_conv3x3_bwd_fused_supported(dtype) && B > 0 && H > 0 && W > 0 && (long long)B * H * W * 64 <= (1LL << 31) - (1LL << 19) && (long long)B * H < (1 << 23) &&
          W < (1 << 23)))
        return 0;
    // the tile index is divided by mulhi with ceil(2^32 / d): exact while index * d < 2^32
    const long long tx = wm_cdiv(W, 16), ty = wm_cdiv(H, 8);
    return ((long long)B * tx * ty * 2 * tx < (1LL << 32) && (long long)B * ty * ty < (1LL << 32)) ? 1 : 0;
}
extern "C" int wm_conv3x3_bwd_fused_nwg(int B, int H, int W) {
    const long n = (long)B * wm_cdiv(H, 8) * wm_cdiv(W, 16);
    return (int)(n < WM_MAX_WGS ? n : WM_MAX_WGS);
}
extern "C" int wm_conv3x3_bwd_fused_gvec_max_batch(void) { return wm_bwd_ws_gvec_max_batch_bf16(); }
// whole-tile shapes with a premasked tensor gradient or a per-sample gradient (all 13 launches of the step): the role-split 8-wave form
static bool bwd_split(int H, int W, int g_premasked, bool has_gvec) {
    return g_bwd_split && (has_gvec || g_premasked) && H % 8 == 0 && W % 16 == 0 && (g_bwd_dbg == 0 || g_bwd_dbg == (1 << 21));
}
extern "C" int wm_conv3x3_bwd_fused_kernel(int H, int W, int g_premasked, int has_gvec) { return bwd_split(H, W, g_premasked, has_gvec != 0) ? 8 : 1; }
extern "C" int wm_conv3x3_bwd_fused(const void* g, const float* gvec, const void* y, const float* stats4, const float* coef, const void* wpt,
                                    const void* xr, const float* in_scale, const float* in_shift, void* dx, float* partials, float* ws, int B,
                                    int H, int W, int dtype, int g_premasked, int sweep_reverse, void* stream) {
    WM_REQUIRE((g != nullptr) != (gvec != nullptr), WM_E_BADARG, "wm_conv3x3_bwd_fused: exactly one of g (a tensor) and gvec (one row per sample) is given");
    WM_REQUIRE(!gvec || B <= wm_conv3x3_bwd_fused_gvec_max_batch(), WM_E_SHAPE, "wm_conv3x3_bwd_fused: gvec form holds at most %d samples' rows in the LDS", wm_conv3x3_bwd_fused_gvec_max_batch());
    WM_REQUIRE(y && stats4 && coef && wpt && xr && in_scale && in_shift && dx && partials && ws, WM_E_BADARG, "wm_conv3x3_bwd_fused: null pointer");
    WM_REQUIRE(wm_conv3x3_bwd_fused_supported(dtype), WM_E_SHAPE, "wm_conv3x3_bwd_fused: 16-bit activations only (dtype %d)", dtype);
    WM_REQUIRE(B > 0 && H > 0 && W > 0, WM_E_BADARG, "wm_conv3x3_bwd_fused: bad shape");
    WM_REQUIRE((long long)B * H * W * 64 <= (1LL << 31) - (1LL << 19) && (long long)B * H < (1 << 23) && W < (1 << 23), WM_E_SHAPE,
               "wm_conv3x3_bwd_fused: tensors of 2^31 elements or more (32-bit element offsets, 24-bit row / column counts)");
    WM_REQUIRE((((uintptr_t)g | (uintptr_t)y | (uintptr_t)wpt | (uintptr_t)xr | (uintptr_t)dx | (uintptr_t)ws) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_bwd_fused: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int nwg = wm_conv3x3_bwd_fused_nwg(B, H, W);
    const bool split = bwd_split(H, W, g_premasked, gvec != nullptr);
    const int stamps = g_bwd_dbg == (1 << 21);   // debug build: tools/phase_bwd8.py
    if (split) {
        const int rc8 = dtype == WM_F16
            ? wm_launch_bwd_ws8_f16(g, y, stats4, 64, coef, wpt, xr, in_scale, in_shift, dx, partials, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_premasked, gvec, 64, stamps)
            : wm_launch_bwd_ws8_bf16(g, y, stats4, 64, coef, wpt, xr, in_scale, in_shift, dx, partials, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_premasked, gvec, 64, stamps);
        WM_REQUIRE(rc8 == WM_OK, rc8, "wm_conv3x3_bwd_fused: the role-split kernel takes a premasked tensor gradient or a per-sample gradient only");
    } else if (dtype == WM_F16) wm_launch_bwd_ws_f16(g, y, stats4, 64, coef, wpt, xr, in_scale, in_shift, dx, partials, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_bwd_dbg, g_premasked, gvec, 64);
    else wm_launch_bwd_ws_bf16(g, y, stats4, 64, coef, wpt, xr, in_scale, in_shift, dx, partials, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_bwd_dbg, g_premasked, gvec, 64);
    WM_LAUNCH_CHECK("wm_conv3x3_bwd_fused");
    return WM_OK;
}
// ---- the image-fed first layer in one pass (bwd_ws16.hip): input gradient wrt the 16-channel image tensor + weight gradient
WM_KNOB_ON(g_bwdfuse16, "WM_NO_BWD_FUSE16");
WM_KNOB_SETTER(wm_debug_bwd_fuse16, g_bwdfuse16)   // A/B knob (debug build only)
extern "C" int wm_conv3x3_bwd_fused16_supported(int B, int H, int W, int dtype) {
    return (g_bwdfuse16 && is16(dtype) && B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 16 == 0 &&
            (long long)B * H * W * 64 <= (1LL << 31) - (1LL << 19)) ? 1 : 0;
}
extern "C" int wm_conv3x3_bwd_fused16_nwg(int B, int H, int W) {
    const long n = (long)B * (H / 8) * (W / 16);
    return (int)(n < WM_MAX_WGS ? n : WM_MAX_WGS);   // one workgroup per CU (round 4: the kernel's LDS request no longer lets two share one, bwd_ws16.hip)
}
extern "C" int wm_conv3x3_bwd_fused16(const void* g, const void* y, const float* stats4, const float* coef, const void* wpt, const void* x,
                                      void* dx, float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout, int dtype,
                                      int g_premasked, int sweep_reverse, void* stream) {
    WM_REQUIRE(g && y && stats4 && coef && wpt && x && dx && ws && dw, WM_E_BADARG, "wm_conv3x3_bwd_fused16: null pointer");
    WM_REQUIRE(wm_conv3x3_bwd_fused16_supported(B, H, W, dtype), WM_E_SHAPE,
               "wm_conv3x3_bwd_fused16: 16-bit activations, H %% 8 == 0, W %% 16 == 0, fewer than 2^31 elements (B=%d H=%d W=%d dtype=%d)", B, H, W, dtype);
    WM_REQUIRE(Cin > 0 && Cin <= 16 && Cout > 0 && Cout <= 64, WM_E_SHAPE, "wm_conv3x3_bwd_fused16: Cin <= 16, Cout <= 64 (got %d, %d)", Cin, Cout);
    WM_REQUIRE((((uintptr_t)g | (uintptr_t)y | (uintptr_t)wpt | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)ws) & 15) == 0, WM_E_SHAPE,
               "wm_conv3x3_bwd_fused16: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int nwg = wm_conv3x3_bwd_fused16_nwg(B, H, W);
    if (dtype == WM_F16) wm_launch_bwd_ws16_f16(g, y, stats4, 64, coef, wpt, x, dx, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_premasked);
    else wm_launch_bwd_ws16_bf16(g, y, stats4, 64, coef, wpt, x, dx, ws, B, H, W, nwg, sweep_reverse ? 1 : 0, s, g_premasked);
    WM_LAUNCH_CHECK("wm_conv3x3_bwd_fused16");
    WM_REQUIRE(launch_wgrad_reduce(ws, nwg, 16, 64, dw, Cin, Cout, nullptr, accumulate, nullptr, s) == WM_OK, WM_E_BADARG, "wm_conv3x3_bwd_fused16: reduction");
    WM_LAUNCH_CHECK("wm_conv3x3_bwd_fused16(reduce)");
    return WM_OK;
}

// the slab reduction of wm_conv3x3_bwd_fused's workspace (its own call so that the kernel above can be timed alone): dw (+)= sum of the
// nwg slabs; `fin`: an optional BatchNorm-backward finalisation riding on the launch, as in wm_conv3x3_wgrad_fin
extern "C" int wm_conv3x3_bwd_fused_reduce(float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout, const WmBnBwdFin* fin,
                                           void* stream) {
    WM_REQUIRE(ws && dw && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 64 && Cout > 0 && Cout <= 64, WM_E_BADARG, "wm_conv3x3_bwd_fused_reduce: bad arguments");
    WM_REQUIRE(fin_rider_ok(fin), WM_E_BADARG, "wm_conv3x3_bwd_fused_reduce: bad finalisation rider (null pointer, or more than 256 partial rows)");
    WM_REQUIRE(launch_wgrad_reduce(ws, wm_conv3x3_bwd_fused_nwg(B, H, W), 64, 64, dw, Cin, Cout, nullptr, accumulate, fin, (hipStream_t)stream) == WM_OK,
               WM_E_BADARG, "wm_conv3x3_bwd_fused_reduce: bad finalisation rider");
    WM_LAUNCH_CHECK("wm_conv3x3_bwd_fused_reduce");
    return WM_OK;
}
