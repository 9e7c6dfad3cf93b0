// Wave-specialised persistent 3x3 convolution for bf16, Cout = 64, Cin = 64 (body layers, dgrad) or 16 (image-fed
// first layers) (gfx950).
//
// Why: phase stamps of its single-role predecessor show that one wave per SIMD cannot overlap its own phases -- the MFMA loop
// (4,742 cycles / tile), the HBM traffic of a tile (78 KB per CU = ~7,400 cycles at the ~10.7 B/clk a CU gets)
// and the VALU work (BN+ReLU transform, bias / statistics / pack) simply add up (10-12k cycles / tile), whatever
// the instruction order.  Here the two kinds of work live in different waves of one 512-thread workgroup, two
// waves per SIMD, so the hardware issues them concurrently:
//   waves 0-3  CONSUMERS: 36 x (4 ds_read_b128 + 4 v_mfma_f32_32x32x16_bf16) on halo tile X[t&1] and the resident
//              filter (the filter is the A operand, so accumulator rows are channels and a lane owns 16 adjacent
//              channels of one pixel), then the epilogue from registers: bias + BatchNorm partial sums as packed
//              f32 pairs, bf16 pack, 8 dwordx4 stores per lane -- no transpose, no LDS staging;
//   waves 4-7  PRODUCERS: global loads of the halo of tile t+2 (registers, two tiles ahead), fused BN+ReLU +
//              zero padding of tile t+1, 16-byte LDS writes into X[(t+1)&1].
// One workgroup barrier per tile hands X[(t+1)&1] to the consumers and X[t&1] back to the producers.
// LDS: filter 73,728 B + 2 x 41,472 B halo tiles + 2 KB = 158,720 B: rows are 128 B (no padding), the 16-byte
// column index is XOR-swizzled with (row >> 1) & 7, which makes 16 consecutive rows hit 16 distinct bank slots
// for ds_read_b128 and for the producers' ds_write_b128.
#include <stdlib.h>
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

constexpr int TH = 16, TW = 16, HH = 18, HW = 18;
constexpr int C64 = 64;
constexpr int NPIX = HH * HW;                  // 324 halo pixels

struct WsArgs {
    const hx_t* x; int ldx;
    const hx_t* wp;            // [9][64][64]
    const float* bias; int nbias;
    const float* in_scale; const float* in_shift;
    hx_t* y;                   // dense [B,H,W,COUT]
    int ldy;                   // NARROW (the 32-channel consumers only): the output's pixel stride when it is below COUT -- channels >= ldy are not stored
    float* stat;                 // [gridDim.x][2][64] or null
    int B, H, W, tilesX, tilesY, ntiles, tiles_per_wg;
    int dbg;      // STAMPS build only: 1 = skip the MFMA loop, 2 = skip the stores, 4 = skip the halo loads
    int xcd_map;
    const float* bw_stats4; int bw_ld; const float* bw_coef; const float* bw_gvec;   // BNBWD: [scale|shift|mean|invstd][bw_ld], coef [3][bw_ld], gvec [B][bw_ld]
    const hx_t* ry; const float* r_scale; const float* r_shift;   // BWDST: raw output [B,H,W,COUT] and scale / shift of the layer whose output gradient this kernel writes
    const hx_t* ay; hx_t* dy_out;   // BNBWD == 2: x is g [B,H,W,64]; ay the layer's raw output, dy_out where dy is written (both dense)
    int reverse;  // walk the workgroup's run of tiles backwards (Infinity Cache reuse of the previous kernel's tail)
    unsigned mX, mY;   // ceil(2^32 / tilesX), ceil(2^32 / tilesY) (0 for 1): the tile index is divided by mulhi, not by ~40 scalar instructions
};

inline void ws_magic(WsArgs& a) {
    auto m = [](int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
    a.mX = m(a.tilesX); a.mY = m(a.tilesY);
}

// 16-byte column swizzles.  Filter rows: key = (row >> 1) & 7.  Halo pixels: key = (halo column >> 1) & 7 -- it does
// not depend on the halo ROW, so for a consumer lane the swizzled address of tap (kh, kw) is a per-(kw, k-step)
// register plus a compile-time (kh, M-fragment) offset; 16 consecutive pixels of a row (from any kw) and the
// pixels of the row below all land on distinct 16-byte bank slots.
// CIN = 16 / 32 (image-fed first layers, the dgrad of the 30-channel decoder layer) are stored unswizzled: their
// 9 / 18-step MFMA loops are a small part of a store-bound kernel.
template <int CIN> __device__ __forceinline__ int swz(int row, int slot) { return CIN == 64 ? slot ^ ((row >> 1) & 7) : slot; }
template <int CIN> __device__ __forceinline__ int swz_px(int px, int slot) { return CIN == 64 ? slot ^ ((px >> 1) & 7) : slot; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// STAMPS: diagnostic build only (tools/phase_ws.py) -- per-role cycle totals of the phases of the tile loop
// M16: consumers use v_mfma_f32_16x16x32_bf16 (needs CIN % 32 == 0) instead of 32x32x16: same FLOPs per cycle, but the
// chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back item 7); A/B: tools/ab_step.py
// BNBWD (dgrad of a layer whose output was globally pooled): the input tensor is that layer's raw conv output y, and the
// producers turn it into dy = ca * (gvec[b]*[scale*y+shift > 0] - c1 - (y-mean)*invstd * c2) while staging -- the
// BatchNorm-backward APPLY pass, fused (the folded form of wm_common.h: bit-identical to bn_bwd_kernel<bf16, APPLY, GVEC>)
// BNBWD == 2 (dgrad of an ordinary layer, gradient g a tensor): the producers read g AND the layer's raw output y, form dy the
// same way (wm_bn_fold_dyg), publish it to the LDS tile and write each tile's own 16x16 pixels of it to `dy_out` for the weight
// gradient that follows: the stand-alone apply pass (402 MB, HBM-bound) is gone for 268 MB inside this kernel.  Two input
// streams leave no registers for a second tile in flight, so this mode loads one tile ahead instead of two.
// BWDST (dgrad whose output is the gradient g wrt a ConvBNRelu's ReLU output): the consumers also reduce that layer's
// BatchNorm-backward sums from the tile they hold -- per channel sum(gz) and sum(gz*y), gz = g*[scale*y+shift > 0], g rounded
// to bf16 as stored, y = that layer's raw conv output read at the tile's pixels -- into the partial rows `stat`: the separate
// reduce pass over (g, y) disappears (its y read moves here, its g read is gone)
// ADDIN (forward, M16): the consumers add a second tensor `a.ry` [B,H,W,COUT] (same 16-bit dtype) to the accumulators before the
// statistics and the pack -- y = conv(x) + addend.  The encoder's after-concat layer (hidden_models/encoder.py:25,40) runs as
// conv64(features) + [conv(image) + message bias] this way: the 97-channel concat tensor is never built (csrc/concat_side.hip)
// ACT == 1 (rows f1 / f2: the coupling subnets' conv + ELU, models/invertible_net.py:326-366): the consumers write elu(conv + bias) -- the
// pre-activation is never stored; its backward needs only the output (elu'(z) = out > 0 ? 1 : out + 1)
// BNBWD == 3 (the backward of such a layer): the two-stream form of BNBWD == 2 with the ELU derivative instead of the BatchNorm fold --
// x is g (gradient wrt the ELU output), ay the layer's OUTPUT; the producers form gz = g * (out > 0 ? 1 : out + 1), publish it, write each
// tile's own pixels to dy_out for the weight gradient and sum them per channel into `stat` [gridDim.x][COUT... 64] (the bias gradient's partials)
// ELU of an f32 accumulator, cheap enough for the MFMA waves' epilogue (expm1f costs ~25 instructions): exp - 1 away from zero, the cubic
// Taylor polynomial near it (|z| < 1/64: its error z^4 / 24 < 3e-9 |z|; the hardware exp's cancellation would show in f16 there)
__device__ __forceinline__ float ws_elu(float z) {
    const float e = __expf(z) - 1.f;
    const float p = z * __builtin_fmaf(z, __builtin_fmaf(z, 0.16666667f, 0.5f), 1.f);
    const float neg = __builtin_fabsf(z) < 0.015625f ? p : e;
    return z > 0.f ? z : neg;
}

template <int CIN, int COUT, bool XFORM, bool STATS, bool M16 = false, bool STAMPS = false, int BNBWD = 0, bool BWDST = false, bool PIN = true,
          bool ADDIN = false, bool WHOLE = false, int ACT = 0, bool NARROW = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(WsArgs a, unsigned long long* __restrict__ stamps = nullptr) {
    static_assert(!NARROW || (M16 && COUT == 32 && !BWDST && !ADDIN && !STATS), "NARROW: an input gradient towards a 16-channel (stride) tensor");
    static_assert(ACT == 0 || (!STATS && !BWDST && BNBWD == 0 && !ADDIN), "ACT: plain forward form");
    static_assert(BNBWD != 3 || (!STATS && !BWDST && M16), "BNBWD == 3: no other reduction in the same launch");
    static_assert(!ADDIN || (M16 && !BWDST && BNBWD == 0), "ADDIN: forward form of the 16x16x32 consumers");
    static_assert(CIN == 64 || CIN == 32 || CIN == 16, "input channels");
    static_assert(!M16 || CIN % 32 == 0, "the 16x16x32 MFMA consumes 32 input channels per step");
    static_assert(COUT == 64 || (COUT == 32 && M16), "32 output channels (image dgrads, the 30-channel layer) only with the 16x16x32 consumers");
    constexpr int VPP = CIN / 8;                       // 16-byte vectors per pixel
    constexpr int KS = CIN / 16;                       // MFMA k-steps per filter tap
    constexpr int NSTEP = 9 * KS;
    constexpr int XVP = (NPIX * VPP + 255) / 256;      // halo vectors per PRODUCER thread (256 producer threads)
    unsigned long long ph[4] = {0, 0, 0, 0}, tlast = 0, t_start = 0, rt_start = 0;
    auto now = [&]() {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    auto stamp = [&](int which) {
        if (STAMPS) {
            const unsigned long long t = now();
            if (which >= 0) ph[which] += t - tlast;
            tlast = t;
        }
    };
    if (STAMPS) {
        t_start = now();
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_start)::"memory");
    }
    constexpr int SW_BYTES = 9 * COUT * CIN * 2, SX_BYTES = NPIX * CIN * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * SX_BYTES + 4 * 2 * C64 * 4 + C64 * 4];
    hx_t* sW = reinterpret_cast<hx_t*>(smem);
    hx_t* sX0 = reinterpret_cast<hx_t*>(smem + SW_BYTES);  // two halo tiles back to back
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + 2 * SX_BYTES);
    float* sBias = sRed + 4 * 2 * C64;   // the accumulators start from the bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    if (tid < COUT) sBias[tid] = (a.bias && tid < a.nbias) ? a.bias[tid] : 0.f;

    // ---- filter -> LDS (all 512 threads).  The filter is the A operand of the MFMA (D rows = output channels, D
    // columns = pixels), so a lane of the accumulator tile holds ONE pixel and, in its 16 registers, the MFMA rows
    // (i&3) + 8(i>>2) + 4h.  Output channel c of a 32-channel fragment is therefore stored at filter row
    // rho(c) = (i&3) + 8(i>>2) + 4hh with hh = c>>4, i = c&15: register i of lane half h is channel 16h + i, and a
    // lane owns 16 ADJACENT channels of its pixel -- two 16-byte stores, no transpose.
    // The loads are issued here; the LDS stores (commit_filter) come after the producers have issued their first tile loads,
    // so the filter's trip from L2 and the first halo tiles' trip from HBM overlap at the start of every launch.
    constexpr int NV = 9 * COUT * VPP, WV = (NV + 511) / 512;
    hx8 wv[WV];
#pragma unroll
    for (int k = 0; k < WV; ++k) {
        const int i = min(tid + 512 * k, NV - 1);
        wv[k] = *reinterpret_cast<const hx8*>(a.wp + (size_t)(i / VPP) * CIN + (i % VPP) * 8);
    }
    auto commit_filter = [&]() {
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 512 * k;
            const int row = i / VPP, tap = row / COUT, n = row % COUT;
            const int c = n & 31, ci = c & 15;
            const int rho = (ci & 3) + 8 * (ci >> 2) + 4 * (c >> 4);
            // M16: accumulator row 4q + i of channel fragment nf is channel (COUT/4) q + 4nf + i: a lane again owns COUT/4
            // adjacent channels
            constexpr int CPL = COUT / 4, NFR = COUT / 16;
            const int lrow = M16 ? tap * COUT + ((n >> 2) % NFR) * 16 + 4 * (n / CPL) + (n & 3) : tap * COUT + (n >> 5) * 32 + rho;
            if (i < NV) *reinterpret_cast<hx8*>(sW + lrow * CIN + swz<CIN>(lrow, i % VPP) * 8) = wv[k];
        }
    };
    const bool early_filter = (a.dbg & 128) != 0;   // A/B knob (variant 10): the filter committed before anything else is issued
    if (early_filter) commit_filter();

    // XCD-aware run assignment: workgroups b and b+8 share an XCD (round-robin dispatch), so give XCD x the
    // consecutive runs [x*G/8, (x+1)*G/8): vertically adjacent tile rows then meet in ONE L2 at about the same time
    // and the halo rows they share are fetched from HBM once
    const int G = gridDim.x;
    const int run = (a.xcd_map && (G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int t_begin = run * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    struct TileGeo { int b, ty0, tx0; };
    // (exact while index * divisor < 2^32: the launchers check)
    auto fdiv = [](int t, int d, unsigned m) { return d == 1 ? t : (int)__umulhi((unsigned)t, m); };
    auto geo = [&](int tile) {
        TileGeo g;
        const int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;
        const int q1 = fdiv(t, a.tilesX, a.mX), txi = t - q1 * a.tilesX;
        const int q2 = fdiv(q1, a.tilesY, a.mY), tyi = q1 - q2 * a.tilesY;
        g.b = q2; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };

    if (producer) {
        // ================================================================== PRODUCER waves
        if (a.dbg & 8) __builtin_amdgcn_s_setprio(3);   // A/B knob (variant 3): raising the producers measured 1 % slower
        const int ptid = tid - 256;
        const int vec = ptid % VPP;
        f32x2 sc2[4], sh2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc2[e] = f32x2{1.f, 1.f}; sh2[e] = f32x2{0.f, 0.f}; }
        if (XFORM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc2[e] = f32x2{a.in_scale[vec * 8 + 2 * e], a.in_scale[vec * 8 + 2 * e + 1]};
                sh2[e] = f32x2{a.in_shift[vec * 8 + 2 * e], a.in_shift[vec * 8 + 2 * e + 1]};
            }
        }
        static_assert(!(BNBWD && XFORM), "one input transform at a time");
        static_assert(BNBWD < 2 || CIN == 64, "the tensor-gradient forms are built for 64-channel layers");
        float kca[8];   // BNBWD == 2: ca of the 8 channels
        float kb[4][8], k3g[8];   // BNBWD: scale, shift, k2, k3 of this thread's 8 channels (wm_bn_fold); k3g = k3 + ca*gvec[sample being published]
        int gvb = -1;
#pragma unroll
        for (int e = 0; e < 8; ++e) k3g[e] = 0.f;
        if (BNBWD == 1 || BNBWD == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = vec * 8 + e;
                kb[0][e] = a.bw_stats4[c]; kb[1][e] = a.bw_stats4[a.bw_ld + c];
                wm_bn_fold(a.bw_stats4[2 * a.bw_ld + c], a.bw_stats4[3 * a.bw_ld + c], a.bw_coef[c], a.bw_coef[a.bw_ld + c],
                           a.bw_coef[2 * a.bw_ld + c], kb[2][e], kb[3][e]);
                kca[e] = a.bw_coef[c];
            }
        }
        auto load_gvec = [&](int b) {   // wave-uniform: a run of tiles rarely crosses a sample
            if (BNBWD == 1 && b != gvb) {
                gvb = b;
#pragma unroll
                for (int e = 0; e < 8; ++e) k3g[e] = wm_bn_fold_g(a.bw_coef[vec * 8 + e], a.bw_gvec[(size_t)b * a.bw_ld + vec * 8 + e], kb[3][e]);
            }
        };
        // tile-invariant per-vector geometry: halo pixel (py, px), its offset inside an interior tile, its LDS slot
        // EDGE (CIN = 64): the vector slots are ordered so that k < KMAIN covers halo columns 2..17 (18 x 16 pixels x 8 vectors
        // = 9 x 256 exactly) and k >= KMAIN the columns 0..1, which a tile shares with its left neighbour: when the previous
        // tile of the run IS that neighbour, those 36 pixels are copied LDS -> LDS from its columns 16..17 (already transformed)
        // instead of being loaded again: 11 % fewer bytes through the CU's memory path, which is what bounds this kernel
        constexpr bool EDGE = CIN == 64;
        constexpr int KMAIN = EDGE ? 9 : XVP;
        int hpy[XVP], hpx[XVP], rel[XVP], lds[XVP];
#pragma unroll
        for (int k = 0; k < XVP; ++k) {
            int py, px;
            if (EDGE) {   // a backward sweep (tiles right to left) shares its RIGHT two columns with the tile before it
                if (k < KMAIN) { const int m = (ptid + 256 * k) >> 3; py = m >> 4; px = (a.reverse ? 0 : 2) + (m & 15); }
                else { const int e = min((ptid + 256 * (k - KMAIN)) >> 3, 35); py = e >> 1; px = (a.reverse ? 16 : 0) + (e & 1); }
            } else {
                const int pix = min((ptid + 256 * k) / VPP, NPIX - 1);
                py = pix / HW; px = pix - py * HW;
            }
            hpy[k] = py; hpx[k] = px;
            rel[k] = (py * a.W + px) * a.ldx + vec * 8;
            lds[k] = (py * HW + px) * CIN + swz_px<CIN>(px, vec) * 8;
        }
        const bool last_live = EDGE ? (((ptid + 256 * (XVP - 1 - KMAIN)) >> 3) < 36) : (((ptid + 256 * (XVP - 1)) / VPP) < NPIX);
        // does tile T start with the two columns its predecessor in the run ended with?
        auto reuse_of = [&](int T) {
            if (!(EDGE && BNBWD < 2 && !(a.dbg & 64) && T > t_begin)) return false;
            const int tt = a.reverse ? t_begin + (t_end - 1 - T) : T;
            const int col = tt - fdiv(tt, a.tilesX, a.mX) * a.tilesX;
            return a.reverse ? col != a.tilesX - 1 : col != 0;
        };
        // loads: always a valid address, never under a per-lane branch; the branches on `interior` are wave-uniform
        auto is_interior = [&](const TileGeo& g) {
            return g.ty0 >= 1 && g.ty0 + TH + 1 <= a.H && g.tx0 >= 1 && g.tx0 + TW + 1 <= a.W;
        };
        auto load_interior = [&](const hx_t* xt, int k, hx8& d) { d = *reinterpret_cast<const hx8*>(xt + rel[k]); };
        auto load_border = [&](const TileGeo& g, const hx_t* xb, int k, hx8& d, unsigned& okbits) {
            const int gy = g.ty0 - 1 + hpy[k], gx = g.tx0 - 1 + hpx[k];
            const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            d = *reinterpret_cast<const hx8*>(xb + (gyc * a.W + gxc) * a.ldx + vec * 8);
            okbits |= ((gy == gyc && gx == gxc) ? 1u : 0u) << k;
        };
        auto tile_ptr = [&](const TileGeo& g) { return a.x + ((size_t)(g.b * a.H + g.ty0 - 1) * a.W + (g.tx0 - 1)) * a.ldx; };
        auto image_ptr = [&](const TileGeo& g) { return a.x + (size_t)g.b * a.H * a.W * a.ldx; };
        auto load_tile = [&](const TileGeo& g, hx8 (&d)[XVP], unsigned& okbits, bool reuse) {
            if (STAMPS && (a.dbg & 4)) { okbits = 0xffffffffu; return; }
            if (is_interior(g)) {
                const hx_t* xt = tile_ptr(g);
#pragma unroll
                for (int k = 0; k < KMAIN; ++k) load_interior(xt, k, d[k]);
                if (!reuse) {
#pragma unroll
                    for (int k = KMAIN; k < XVP; ++k) load_interior(xt, k, d[k]);
                }
                okbits = 0xffffffffu;
            } else {
                const hx_t* xb = image_ptr(g);
                okbits = 0;
#pragma unroll
                for (int k = 0; k < KMAIN; ++k) load_border(g, xb, k, d[k], okbits);
                if (!reuse) {
#pragma unroll
                    for (int k = KMAIN; k < XVP; ++k) load_border(g, xb, k, d[k], okbits);
                }
            }
        };
        // edge vector k of the tile being published <- columns 16..17 of the tile in the other buffer
        auto copy_edge = [&](hx_t* sX, const hx_t* sPrev, int k) {
            const int px = hpx[k] + (a.reverse ? -16 : 16);
            const u32x4 w = *reinterpret_cast<const u32x4*>(sPrev + (hpy[k] * HW + px) * CIN + swz_px<CIN>(px, vec) * 8);
            if (k + 1 < XVP || last_live) *reinterpret_cast<u32x4*>(sX + lds[k]) = w;
        };
        // fused BN + ReLU (ReLU on the packed bf16 pair as a signed 16-bit max), zero padding AFTER the activation
        auto put_one = [&](hx_t* sX, int k, const hx8& d, unsigned okbits) {
            u32x4 w = __builtin_bit_cast(u32x4, d);
            if (XFORM) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    // scalar v_fma_f32 on purpose: packed f32 VALU beside the partner wave's MFMAs costs far more than two scalar ops
                    const float f0 = __builtin_fmaf(HX::lo(w[pq]), sc2[pq][0], sh2[pq][0]);
                    const float f1 = __builtin_fmaf(HX::hi(w[pq]), sc2[pq][1], sh2[pq][1]);
                    const hx2 pk = HX::pack2(f0, f1);
                    const i16x2 z = {0, 0};
                    w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                }
            }
            if (BNBWD == 1) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const float d0 = wm_bn_fold_dy(HX::lo(w[pq]), kb[0][2 * pq], kb[1][2 * pq], kb[2][2 * pq], kb[3][2 * pq], k3g[2 * pq]);
                    const float d1 = wm_bn_fold_dy(HX::hi(w[pq]), kb[0][2 * pq + 1], kb[1][2 * pq + 1], kb[2][2 * pq + 1],
                                                   kb[3][2 * pq + 1], k3g[2 * pq + 1]);
                    const hx2 pk = HX::pack2(d0, d1);
                    w[pq] = __builtin_bit_cast(unsigned, pk);
                }
            }
            const unsigned keep = 0u - ((okbits >> k) & 1u);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] &= keep;
            if (k + 1 < XVP || last_live) *reinterpret_cast<u32x4*>(sX + lds[k]) = w;
        };
        if constexpr (BNBWD >= 2) {
            // vectors whose halo pixel is one of the tile's own 16x16 pixels: those are written out as dy
            unsigned wmask = 0;
#pragma unroll
            for (int k = 0; k < XVP; ++k)
                if (hpy[k] >= 1 && hpy[k] <= TH && hpx[k] >= 1 && hpx[k] <= TW && (k + 1 < XVP || last_live)) wmask |= 1u << k;
            hx8 dG[XVP], dY[XVP];
            unsigned ok = 0;
            float bsum[8];   // BNBWD == 3: this thread's part of the bias gradient (its 8 channels over the own pixels it stages)
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
            auto load_both = [&](const TileGeo& g) {
                if (is_interior(g)) {
                    const hx_t* gt = tile_ptr(g);
                    const hx_t* yt = a.ay + (gt - a.x);
#pragma unroll
                    for (int k = 0; k < XVP; ++k) { load_interior(gt, k, dG[k]); load_interior(yt, k, dY[k]); }
                    ok = 0xffffffffu;
                } else {
                    const hx_t* gb = image_ptr(g);
                    const hx_t* yb = a.ay + (gb - a.x);
                    unsigned unused = 0;
                    ok = 0;
#pragma unroll
                    for (int k = 0; k < XVP; ++k) { load_border(g, gb, k, dG[k], ok); load_border(g, yb, k, dY[k], unused); }
                }
            };
            auto put_both = [&](hx_t* sX, const TileGeo& g) {
                const long tofs = ((long)(g.b * a.H + g.ty0 - 1) * a.W + (g.tx0 - 1)) * CIN;   // + rel[k] (ldx == CIN): the pixel's offset
#pragma unroll
                for (int k = 0; k < XVP; ++k) {
                    u32x4 w = __builtin_bit_cast(u32x4, dG[k]);
                    const u32x4 wy = __builtin_bit_cast(u32x4, dY[k]);
                    const unsigned inimg = (ok >> k) & 1u, keep = 0u - inimg;
                    const float own = (((wmask >> k) & 1u) && inimg) ? 1.f : 0.f;
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        float d0, d1;
                        if (BNBWD == 3) {
                            const float o0 = HX::lo(wy[pq]), o1 = HX::hi(wy[pq]);
                            d0 = HX::lo(w[pq]) * (o0 > 0.f ? 1.f : o0 + 1.f);
                            d1 = HX::hi(w[pq]) * (o1 > 0.f ? 1.f : o1 + 1.f);
                            bsum[2 * pq] = __builtin_fmaf(d0, own, bsum[2 * pq]);
                            bsum[2 * pq + 1] = __builtin_fmaf(d1, own, bsum[2 * pq + 1]);
                        } else {
                            d0 = wm_bn_fold_dyg(HX::lo(wy[pq]), HX::lo(w[pq]), kb[0][2 * pq],
                                                kb[1][2 * pq], kca[2 * pq], kb[2][2 * pq], kb[3][2 * pq]);
                            d1 = wm_bn_fold_dyg(HX::hi(wy[pq]), HX::hi(w[pq]),
                                                kb[0][2 * pq + 1], kb[1][2 * pq + 1], kca[2 * pq + 1], kb[2][2 * pq + 1], kb[3][2 * pq + 1]);
                        }
                        const hx2 pk = HX::pack2(d0, d1);
                        w[pq] = __builtin_bit_cast(unsigned, pk);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] &= keep;
                    if (k + 1 < XVP || last_live) *reinterpret_cast<u32x4*>(sX + lds[k]) = w;
                    if (((wmask >> k) & 1u) && inimg && a.dy_out) *reinterpret_cast<u32x4*>(a.dy_out + tofs + rel[k]) = w;   // (dy_out NULL: nobody wants dy -- an input gradient without its weight gradient)
                }
            };
            if (t_begin < t_end) load_both(geo(t_begin));
            if (!early_filter) commit_filter();
            if (t_begin < t_end) put_both(sX0, geo(t_begin));
            __syncthreads();  // filter + first halo tile visible
            stamp(-1);
            for (int tile = t_begin; tile < t_end; ++tile) {
                if (tile + 1 < t_end) {
                    const TileGeo g1 = geo(tile + 1);
                    load_both(g1);
                    put_both(sX0 + (((tile - t_begin) & 1) ^ 1) * (NPIX * CIN), g1);
                }
                stamp(0);
                stamp(1);
                __syncthreads();
                stamp(2);
            }
            if (BNBWD == 3) {   // lanes l, l + 8, ... of a wave stage the same 8 channels: fold them, one row of 64 sums per producer wave
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = bsum[e];
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) u += __shfl_xor(u, o, 64);
                    if (lane < 8) sRed[(wave - 4) * C64 + vec * 8 + e] = u;
                }
            }
            if (STATS || BWDST || BNBWD == 3) __syncthreads();
            return;
        }
        hx8 dA[XVP], dB[XVP];
        unsigned okA = 0, okB = 0;
        if (t_begin < t_end) load_tile(geo(t_begin), dA, okA, false);
        if (t_begin + 1 < t_end) load_tile(geo(t_begin + 1), dB, okB, reuse_of(t_begin + 1));
        if (!early_filter) commit_filter();
        if (t_begin < t_end) {
            load_gvec(geo(t_begin).b);
#pragma unroll
            for (int k = 0; k < XVP; ++k) put_one(sX0, k, dA[k], okA);
        }
        __syncthreads();  // filter + first halo tile visible
        stamp(-1);
        // iteration `tile`: `cur` holds tile+1 (loaded one iteration ago) and is published (transform + LDS write) while
        // the loads of tile+2 go into `nxt`; one load, one vector of VALU work, alternating, so the memory queue is fed
        // at an even pace and never holds the whole burst
        auto iter = [&](int tile, hx8 (&nxt)[XVP], unsigned& oknxt, const hx8 (&cur)[XVP], unsigned okcur) {
            hx_t* sXn = sX0 + (((tile - t_begin) & 1) ^ 1) * (NPIX * CIN);
            const hx_t* sXc = sX0 + ((tile - t_begin) & 1) * (NPIX * CIN);   // the tile the consumers are on: left neighbour of tile+1
            const bool have_next = tile + 2 < t_end && !(STAMPS && (a.dbg & 4));
            const bool reuse_cur = reuse_of(tile + 1), reuse_nxt = reuse_of(tile + 2);
            if (BNBWD == 1 && tile + 1 < t_end) load_gvec(geo(tile + 1).b);
            if (have_next) {
                const TileGeo g2 = geo(tile + 2);
                if (is_interior(g2)) {
                    const hx_t* xt = tile_ptr(g2);
                    oknxt = 0xffffffffu;
#pragma unroll
                    for (int k = 0; k < KMAIN; ++k) {
                        load_interior(xt, k, nxt[k]);
                        put_one(sXn, k, cur[k], okcur);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!reuse_nxt) {
#pragma unroll
                        for (int k = KMAIN; k < XVP; ++k) load_interior(xt, k, nxt[k]);
                    }
                } else {
                    const hx_t* xb = image_ptr(g2);
                    oknxt = 0;
#pragma unroll
                    for (int k = 0; k < KMAIN; ++k) {
                        load_border(g2, xb, k, nxt[k], oknxt);
                        put_one(sXn, k, cur[k], okcur);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!reuse_nxt) {
#pragma unroll
                        for (int k = KMAIN; k < XVP; ++k) load_border(g2, xb, k, nxt[k], oknxt);
                    }
                }
            } else if (tile + 1 < t_end) {
#pragma unroll
                for (int k = 0; k < KMAIN; ++k) put_one(sXn, k, cur[k], okcur);
            }
            if (tile + 1 < t_end) {   // the edge columns of tile+1: copied from the neighbour or published from the loaded data
                if (reuse_cur) {
#pragma unroll
                    for (int k = KMAIN; k < XVP; ++k) copy_edge(sXn, sXc, k);
                } else {
#pragma unroll
                    for (int k = KMAIN; k < XVP; ++k) put_one(sXn, k, cur[k], okcur);
                }
            }
            stamp(0);  // loads of tile+2 interleaved with transform + LDS writes of tile+1
            if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(1);
            __syncthreads();
            stamp(2);  // barrier
        };
        for (int tile = t_begin; tile < t_end; tile += 2) {
            iter(tile, dA, okA, dB, okB);
            if (tile + 1 < t_end) iter(tile + 1, dB, okB, dA, okA);
        }
        if (STATS || BWDST) __syncthreads();
        if (STAMPS && stamps && tid == 256) {
            unsigned long long rt_end;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
            unsigned long long* o = stamps + (size_t)blockIdx.x * 16 + 8;
            o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3];
            o[4] = now() - t_start; o[5] = rt_end - rt_start;
        }
        return;
    }

    // ====================================================================== CONSUMER waves
    // lane (r, h): pixel r of a 32-pixel fragment (tile row wave*4 + 2*mf + (r>>4), column r&15); accumulator
    // register i of channel fragment nf = output channel nf*32 + 16h + i.
    // Software pipeline at HALF-tile granularity, no extra registers: a tile is computed as two passes over K, pixel
    // fragment mf = 0 then mf = 1 (2 MFMAs per step; the filter fragments are read twice -- the LDS has the room), and
    // while one half accumulates, the finished other half is drained (BatchNorm sums, bf16 pack, stores) in the
    // shadow of the MFMAs, a few instructions per step, so the matrix pipe never waits for an epilogue.
    if (!early_filter) commit_filter();
    if constexpr (M16) {
        // lane (p, q): pixel column p of the tile row wave*4 + mf; accumulator [mf][nf] register i = channel 16q + 4nf + i
        constexpr int KS2 = CIN / 32, NSTEP2 = 9 * KS2;
        constexpr int CPL = COUT / 4;          // adjacent channels per lane (16 or 8)
        constexpr int NFR = COUT / 16;         // 16-channel filter fragments (4 or 2)
        constexpr int NPAIR = CPL / 2;         // channel pairs per lane and tile row
        constexpr int NDR = 2 * (NPAIR + 1);   // drain micro-steps per half
        const int p = lane & 15, q = lane >> 4;
        float s1[CPL], s2[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
        int aoff[3][KS2], boff[KS2];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                aoff[kw][ks] = ((wave * 4 * HW + p + kw) * CIN + swz_px<CIN>(p + kw, ks * 4 + q) * 8) * 2;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) boff[ks] = (p * CIN + swz<CIN>(p, ks * 4 + q) * 8) * 2;

        f32x4 acc[4][NFR];   // [pixel fragment mf = tile row][channel fragment nf]
        static_assert(!BWDST || (!STATS && COUT == 64), "BWDST: 64-channel dgrads");
        struct Drain { hx_t* yp; float mk; bool inb; };   // mk: STATS: 1 / 0 inside / outside the image; BWDST: the ReLU threshold 0 / +inf (nothing passes outside)
        float rsc[CPL], rsh[CPL];
        if (BWDST) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) { rsc[c] = a.r_scale[CPL * q + c]; rsh[c] = a.r_shift[CPL * q + c]; }
        }
        unsigned ryv[2][NPAIR];   // BWDST: the layer's y at the two pixels of the half being drained, this lane's channels (ADDIN: the addend)
        auto load_ry = [&](const Drain (&d)[2]) {
            if (BWDST || ADDIN) {
#pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const hx_t* src = a.ry + (d[ml].inb ? (d[ml].yp - a.y) : (ptrdiff_t)(CPL * q));
#pragma unroll
                    for (int v = 0; v < NPAIR / 4; ++v) {
                        const u32x4 t = *reinterpret_cast<const u32x4*>(src + 8 * v);
                        ryv[ml][4 * v] = t[0]; ryv[ml][4 * v + 1] = t[1]; ryv[ml][4 * v + 2] = t[2]; ryv[ml][4 * v + 3] = t[3];
                    }
                }
            }
        };
        auto drain_of = [&](const TileGeo& g, int mf) {
            Drain d;
            const int gy = g.ty0 + wave * 4 + mf, gx = g.tx0 + p;
            // WHOLE (the launcher: H and W multiples of the 16 x 16 tile): every pixel of every tile is inside the image -- the mask is the
            // constant 1 and folds out of the statistics (one multiply per element less), the stores lose their predicate
            d.inb = WHOLE ? true : (gy < a.H && gx < a.W);
            if (NARROW) d.inb = d.inb && CPL * q < a.ldy;      // (lanes whose channels lie beyond the output's stride store nothing)
            d.mk = BWDST ? (d.inb ? 0.f : __builtin_inff()) : (d.inb ? 1.f : 0.f);
            d.yp = a.y + (((size_t)g.b * a.H + gy) * a.W + gx) * (NARROW ? a.ldy : COUT) + CPL * q;
            return d;
        };
        unsigned pk[NPAIR];
        // micro-step m of draining the half {2*dh, 2*dh+1}: m = (NPAIR+1) ml + j; j < NPAIR: channel pair j, j == NPAIR: the stores
        auto drain_step = [&](int m, int dh, const Drain (&d)[2]) {
            const int ml = m / (NPAIR + 1), j = m - ml * (NPAIR + 1), mf = 2 * dh + ml;
            if (j < NPAIR) {
                const int nf = j >> 1, i0 = 2 * (j & 1);
                float v0 = acc[mf][nf][i0], v1 = acc[mf][nf][i0 + 1];
                if (ADDIN) { v0 += HX::lo(ryv[ml][j]); v1 += HX::hi(ryv[ml][j]); }
                if (ACT == 1) { v0 = ws_elu(v0); v1 = ws_elu(v1); }
                if (STATS) {   // scalar f32 on purpose (packed f32 VALU is slow beside MFMAs)
                    const float t0 = v0 * d[ml].mk, t1 = v1 * d[ml].mk;
                    s1[2 * j] += t0; s1[2 * j + 1] += t1;
                    s2[2 * j] = __builtin_fmaf(t0, v0, s2[2 * j]);
                    s2[2 * j + 1] = __builtin_fmaf(t1, v1, s2[2 * j + 1]);
                    // pin: without it instruction selection gathers every one of these sums AFTER the pass's last MFMA (they
                    // are pure arithmetic, the sched_barriers do not hold them), out of the matrix pipe's shadow
                    // (WHOLE: the sched_group_barrier pairing in pass() does it instead -- the asm's tied operands cost 4 v_mov per channel pair;
                    // measured both ways on both forms: A/B table in DESIGN section 9)
                    if (PIN && !WHOLE) asm volatile("" : "+v"(s1[2 * j]), "+v"(s1[2 * j + 1]), "+v"(s2[2 * j]), "+v"(s2[2 * j + 1]));
                }
                float y0 = 0.f, y1 = 0.f;
                if (BWDST) {   // The gradient LEAVES multiplied by the ReLU mask of the layer it belongs to (gz, not g: every consumer applies
                    // that mask anyway -- twice is the identity -- and bwd_ws.hip's PREMASKED staging skips it): the select sits in front
                    // of the pack instead of behind it, the same number of instructions
                    y0 = HX::lo(ryv[ml][j]); y1 = HX::hi(ryv[ml][j]);
                    const float z0 = __builtin_fmaf(rsc[2 * j], y0, rsh[2 * j]), z1 = __builtin_fmaf(rsc[2 * j + 1], y1, rsh[2 * j + 1]);
                    v0 = z0 > d[ml].mk ? v0 : 0.f; v1 = z1 > d[ml].mk ? v1 : 0.f;   // (outside the image the threshold is +inf: zeros for the sums)
                }
                const hx2 p2 = HX::pack2(v0, v1);
                pk[j] = __builtin_bit_cast(unsigned, p2);
                if (BWDST) {   // scalar f32, as above
                    const float gz0 = HX::lo(pk[j]), gz1 = HX::hi(pk[j]);
                    s1[2 * j] += gz0; s1[2 * j + 1] += gz1;
                    s2[2 * j] = __builtin_fmaf(gz0, y0, s2[2 * j]);
                    s2[2 * j + 1] = __builtin_fmaf(gz1, y1, s2[2 * j + 1]);
                    // (no pin here: measured 0.4 % slower on the step than the compiler's placement after the pass)
                }
            } else if (d[ml].inb && !(STAMPS && (a.dbg & 2))) {
#pragma unroll
                for (int v = 0; v < NPAIR / 4; ++v)
                    *reinterpret_cast<u32x4*>(d[ml].yp + 8 * v) = u32x4{pk[4 * v], pk[4 * v + 1], pk[4 * v + 2], pk[4 * v + 3]};
            }
        };
        // one pass over K for the tile rows {2*half, 2*half+1}; optionally drains half dh on the way
        auto pass = [&](const hx_t* sX, int half, bool drain, int dh, const Drain (&d)[2]) {
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int nf = 0; nf < NFR; ++nf) acc[2 * half + ml][nf] = *reinterpret_cast<const f32x4*>(sBias + CPL * q + 4 * nf);
            if (STAMPS && (a.dbg & 1)) {
                if (drain) {
                    load_ry(d);
#pragma unroll
                    for (int m = 0; m < NDR; ++m) drain_step(m, dh, d);
                }
                return;
            }
            // BWDST: the y values of the half being drained are requested here and first used DS0 K-steps later
            constexpr int DS0 = (BWDST || ADDIN) ? 6 : 0;
            if (drain) load_ry(d);
            constexpr int PF = 2;   // a deeper ring measured the same
            hx8 pix[PF][2], fil[PF][NFR];
            auto load_frags = [&](int sidx, int buf) {
                const int tap = sidx / KS2, ks = sidx % KS2;
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
                    pix[buf][ml] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sX) + aoff[kw][ks] + (2 * half + ml + kh) * (HW * CIN * 2));
#pragma unroll
                for (int nf = 0; nf < NFR; ++nf)
                    fil[buf][nf] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * COUT + nf * 16) * (CIN * 2));
            };
#pragma unroll
            for (int i = 0; i < PF - 1; ++i) load_frags(i, i);
#pragma unroll
            for (int sidx = 0; sidx < NSTEP2; ++sidx) {
                const int cb = sidx % PF;
                if (sidx + PF - 1 < NSTEP2) load_frags(sidx + PF - 1, (sidx + PF - 1) % PF);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                    for (int nf = 0; nf < NFR; ++nf)
                        acc[2 * half + ml][nf] = HX::mfma16(fil[cb][nf], pix[cb][ml], acc[2 * half + ml][nf]);
                if (drain) {
#pragma unroll
                    for (int m = max(sidx - DS0, 0) * NDR / (NSTEP2 - DS0); m < max(sidx + 1 - DS0, 0) * NDR / (NSTEP2 - DS0); ++m) drain_step(m, dh, d);
                    if (PIN && STATS && WHOLE) {   // one drain instruction behind each of the step's MFMAs
#pragma unroll
                        for (int i = 0; i < 2 * NFR; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __syncthreads();  // filter + first halo tile visible
        stamp(-1);
        if (t_begin < t_end) {
            Drain dnone[2];
            dnone[0].yp = a.y; dnone[0].mk = 0.f; dnone[0].inb = false; dnone[1] = dnone[0];
            pass(sX0, 0, false, 0, dnone);
            stamp(0);
            for (int tile = t_begin; tile < t_end; ++tile) {
                const TileGeo g = geo(tile);
                const hx_t* sX = sX0 + ((tile - t_begin) & 1) * (NPIX * CIN);
                const Drain d0[2] = {drain_of(g, 0), drain_of(g, 1)};
                pass(sX, 1, true, 0, d0);
                stamp(1);
                __syncthreads();  // X[t&1] is free for the producers, X[(t+1)&1] is ready
                stamp(2);
                const Drain d1[2] = {drain_of(g, 2), drain_of(g, 3)};
                if (tile + 1 < t_end) {
                    pass(sX0 + (((tile - t_begin) & 1) ^ 1) * (NPIX * CIN), 0, true, 1, d1);
                } else {
                    load_ry(d1);
#pragma unroll
                    for (int m = 0; m < NDR; ++m) drain_step(m, 1, d1);
                }
                stamp(0);
            }
        }
        if (STAMPS && stamps && tid == 0) {
            unsigned long long rt_end;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
            unsigned long long* o = stamps + (size_t)blockIdx.x * 16;
            o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3];
            o[4] = now() - t_start; o[5] = rt_end - rt_start;
        }
        if (BNBWD == 3) {
            __syncthreads();  // matched by the producers' final barrier: their four rows of channel sums are in sRed
            if (tid < C64) a.stat[(size_t)blockIdx.x * C64 + tid] = sRed[tid] + sRed[C64 + tid] + sRed[2 * C64 + tid] + sRed[3 * C64 + tid];
        }
        if (STATS || BWDST) {
            // (the lane's channel base is re-derived from an opaque copy of the thread index: the compiler otherwise keeps the pre-loop
            // value alive across the whole tile loop -- in the BWDST forms, which sit at the 256-register limit, by spilling it)
            int tid_post = threadIdx.x;
            asm volatile("" : "+v"(tid_post));
            const int q_post = (tid_post & 63) >> 4, wave_post = tid_post >> 6;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                float u1 = s1[c], u2 = s2[c];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
                if ((tid_post & 15) == 0) {
                    sRed[(wave_post * 2 + 0) * COUT + CPL * q_post + c] = u1;
                    sRed[(wave_post * 2 + 1) * COUT + CPL * q_post + c] = u2;
                }
            }
            __syncthreads();  // matched by the producers' final barrier
            if (tid < 2 * COUT) {
                const int which = tid / COUT, n = tid - which * COUT;
                a.stat[((size_t)blockIdx.x * 2 + which) * COUT + n] =
                    sRed[(0 * 2 + which) * COUT + n] + sRed[(1 * 2 + which) * COUT + n] + sRed[(2 * 2 + which) * COUT + n] +
                    sRed[(3 * 2 + which) * COUT + n];
            }
        }
        return;
    }
    const int r = lane & 31, h = lane >> 5;
    f32x2 s1[2][8], s2[2][8];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[nf][j] = f32x2{0.f, 0.f}; s2[nf][j] = f32x2{0.f, 0.f}; }
    // swizzled LDS byte offsets of this lane's fragments (see swz_px): 12 + 4 registers, everything else immediate
    int aoff[3][KS], boff[KS];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int px = (r & 15) + kw;
            aoff[kw][ks] = (((wave * 4 + (r >> 4)) * HW + px) * CIN + swz_px<CIN>(px, ks * 2 + h) * 8) * 2;
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) boff[ks] = (r * CIN + swz<CIN>(r, ks * 2 + h) * 8) * 2;

    f32x16 acc[2][2];   // [pixel fragment mf][channel fragment nf]
    struct Drain { hx_t* yp; float mk; bool inb; };
    auto drain_of = [&](const TileGeo& g, int mf) {
        Drain d;
        const int gy = g.ty0 + wave * 4 + mf * 2 + (r >> 4), gx = g.tx0 + (r & 15);
        d.inb = WHOLE ? true : (gy < a.H && gx < a.W);
        d.mk = d.inb ? 1.f : 0.f;
        d.yp = a.y + (((size_t)g.b * a.H + gy) * a.W + gx) * C64 + 16 * h;
        return d;
    };
    unsigned pk[8];
    // micro-step m = 0..17 of draining half mf: m = 9 nf + j; j < 8: pair j (statistics + pack), j == 8: the two stores
    auto drain_step = [&](int m, int mf, const Drain& d) {
        const int nf = m / 9, j = m - nf * 9;
        if (j < 8) {
            float v0 = acc[mf][nf][2 * j], v1 = acc[mf][nf][2 * j + 1];
            if (ACT == 1) { v0 = ws_elu(v0); v1 = ws_elu(v1); }
            if (STATS) {   // scalar f32 on purpose (packed f32 VALU is slow beside MFMAs)
                const float t0 = v0 * d.mk, t1 = v1 * d.mk;
                s1[nf][j][0] += t0; s1[nf][j][1] += t1;
                s2[nf][j][0] = __builtin_fmaf(t0, v0, s2[nf][j][0]);
                s2[nf][j][1] = __builtin_fmaf(t1, v1, s2[nf][j][1]);
            }
            const hx2 p2 = HX::pack2(v0, v1);
            pk[j] = __builtin_bit_cast(unsigned, p2);
        } else if (d.inb && !(STAMPS && (a.dbg & 2))) {
            *reinterpret_cast<u32x4*>(d.yp + nf * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
            *reinterpret_cast<u32x4*>(d.yp + nf * 32 + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
    };
    // one pass over K for pixel fragment mf of the halo tile sX; optionally drains half dmf of `d` on the way
    auto pass = [&](const hx_t* sX, int mf, bool drain, int dmf, const Drain& d) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[mf][nf] = *reinterpret_cast<const f32x16*>(sBias + nf * 32 + 16 * h);
        if (STAMPS && (a.dbg & 1)) {
            if (drain) {
#pragma unroll
                for (int m = 0; m < 18; ++m) drain_step(m, dmf, d);
            }
            return;
        }
        constexpr int PF = STATS ? 2 : 3;   // fragment ring: fetched PF-1 steps ahead of use (the statistics take the registers)
        hx8 af[PF], bfr[PF][2];
        auto load_frags = [&](int sidx, int buf) {
            const int tap = sidx / KS, ks = sidx % KS;
            const int kh = tap / 3, kw = tap - kh * 3;
            af[buf] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sX) + aoff[kw][ks] + (mf * 2 + kh) * (HW * CIN * 2));
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                bfr[buf][nf] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * C64 + nf * 32) * (CIN * 2));
        };
#pragma unroll
        for (int i = 0; i < PF - 1; ++i) load_frags(i, i);
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; ++sidx) {
            const int cb = sidx % PF;
            if (sidx + PF - 1 < NSTEP) load_frags(sidx + PF - 1, (sidx + PF - 1) % PF);
            __builtin_amdgcn_sched_barrier(0);  // reads of later steps stay ahead of the MFMAs of step s
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                acc[mf][nf] = HX::mfma32(bfr[cb][nf], af[cb], acc[mf][nf]);
            if (drain) {   // the 18 drain micro-steps, spread evenly over the pass
#pragma unroll
                for (int m = sidx * 18 / NSTEP; m < (sidx + 1) * 18 / NSTEP; ++m) drain_step(m, dmf, d);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();  // filter + first halo tile visible
    stamp(-1);

    if (t_begin < t_end) {
        Drain dnone; dnone.yp = a.y; dnone.mk = 0.f; dnone.inb = false;
        pass(sX0, 0, false, 0, dnone);                       // prologue: first half of the first tile
        stamp(0);
        for (int tile = t_begin; tile < t_end; ++tile) {
            const TileGeo g = geo(tile);
            const hx_t* sX = sX0 + ((tile - t_begin) & 1) * (NPIX * CIN);
            pass(sX, 1, true, 0, drain_of(g, 0));            // second half; drain the first
            stamp(1);
            __syncthreads();  // X[t&1] is free for the producers, X[(t+1)&1] is ready
            stamp(2);
            const Drain d1 = drain_of(g, 1);
            if (tile + 1 < t_end) {
                pass(sX0 + (((tile - t_begin) & 1) ^ 1) * (NPIX * CIN), 0, true, 1, d1);   // next tile's first half; drain the second
            } else {
#pragma unroll
                for (int m = 0; m < 18; ++m) drain_step(m, 1, d1);
            }
            stamp(0);
        }
    }
    if (STAMPS && stamps && tid == 0) {
        unsigned long long rt_end;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
        unsigned long long* o = stamps + (size_t)blockIdx.x * 16;
        o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3];
        o[4] = now() - t_start; o[5] = rt_end - rt_start;
    }
    if (STATS) {
        // per-lane partial sums -> per-wave sums over the 32 pixel lanes of each half (xor 1..16 stays inside a half)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float u1 = s1[nf][j][e], u2 = s2[nf][j][e];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
                    if (r == 0) {
                        const int c = nf * 32 + 16 * h + 2 * j + e;
                        sRed[(wave * 2 + 0) * C64 + c] = u1;
                        sRed[(wave * 2 + 1) * C64 + c] = u2;
                    }
                }
        __syncthreads();  // matched by the producers' final barrier
        if (tid < 2 * C64) {
            const int which = tid / C64, n = tid - which * C64;
            a.stat[((size_t)blockIdx.x * 2 + which) * C64 + n] =
                sRed[(0 * 2 + which) * C64 + n] + sRed[(1 * 2 + which) * C64 + n] + sRed[(2 * 2 + which) * C64 + n] +
                sRed[(3 * 2 + which) * C64 + n];
        }
    }
}

}  // namespace

#if defined(WM_DEBUG) && !defined(WM_H16_F16)
// diagnostic entry of the debug build (tools/phase_ws.py): stamps [wgs][16] u64 = consumer {mfma, epilogue, barrier, -, cycles, realtime} at +0,
// producer {load issue, transform + LDS write, barrier, -, cycles, realtime} at +8
extern "C" int wm_debug_conv3x3_ws64_phases(const void* x, const void* wp, const float* in_scale, const float* in_shift,
                                            void* y, float* stat, int B, int H, int W, unsigned long long* stamps, int dbg, void* stream) {
    WsArgs a;
    a.dbg = dbg; a.xcd_map = 1;
    a.x = (const hx_t*)x; a.ldx = 64; a.wp = (const hx_t*)wp; a.bias = nullptr; a.nbias = 0; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (hx_t*)y; a.stat = stat; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY; ws_magic(a);
    const int wgs = a.ntiles < 256 ? a.ntiles : 256;
    a.ldy = 64; a.tiles_per_wg = wm_cdiv(a.ntiles, wgs); a.reverse = 0; a.bw_stats4 = nullptr; a.bw_ld = 0; a.bw_coef = nullptr; a.bw_gvec = nullptr; a.ry = nullptr; a.r_scale = nullptr; a.r_shift = nullptr; a.ay = nullptr; a.dy_out = nullptr;
    const dim3 grid((unsigned)wm_cdiv(a.ntiles, a.tiles_per_wg)), block(512);
    if (in_scale && stat) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, true, true, false, true>), grid, block, 0, (hipStream_t)stream, a, stamps);
    else hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, false, true>), grid, block, 0, (hipStream_t)stream, a, stamps);
    return (int)grid.x;
}
#endif

// `sweep_reverse` of the conv / dgrad / wgrad entry points: sweep the tiles backwards.  A kernel that starts where its input's
// producer stopped finds the freshest part of that tensor in the Infinity Cache (256 MB; a layer's tensor is 134 MB): the host
// alternates the direction along a chain of layers (engine.py).  Results do not depend on it except for the summation order
// inside the per-workgroup statistics rows.  (A per-call argument: the library keeps no state between calls.)
#ifndef WM_H16_F16
WM_KNOB_INT(g_ws_reverse, "WM_WS_REVERSE", -1);   // debug build: -1 follow the caller (default), 0 / 1 force forward / backward sweeps
WM_KNOB_SETTER(wm_debug_ws_direction, g_ws_reverse)
int wm_sweep_dir(int reverse) { return (g_ws_reverse == 0 || g_ws_reverse == 1) ? g_ws_reverse : (reverse ? 1 : 0); }
#else
int wm_sweep_dir(int reverse);
#endif
// debug build A/B knobs -- 1: 32x32x16 MFMA consumers, 2: no XCD-aware run assignment, 3: producers at s_setprio 3, 8: no halo-edge reuse, 9: statistics sums not pinned, 10: filter committed to LDS before the first tile loads are issued, 11: the masked form for whole-tile shapes too
#ifndef WM_H16_F16
WM_KNOB_INT(g_ws_variant, "WM_WS_VARIANT", 0);
WM_KNOB_SETTER(wm_debug_ws_variant, g_ws_variant)
#else
static constexpr int g_ws_variant = 0;
#endif

// launcher used by conv3x3.hip
int WM_HSYM(wm_launch_conv3x3_ws)(const void* x, int ldx, int Cin, int CoutP, const void* wp, const float* bias, int nbias, const float* in_scale,
                           const float* in_shift, void* y, float* stat, int B, int H, int W, int wgs, int tiles_per_wg,
                           hipStream_t s, int reverse, const float* bw_stats4 = nullptr, int bw_ld = 0, const float* bw_coef = nullptr,
                           const float* bw_gvec = nullptr, const void* ry = nullptr, const float* r_scale = nullptr,
                           const float* r_shift = nullptr, const void* ay = nullptr, void* dy_out = nullptr, const void* addend = nullptr,
                           int act = 0, int ldy = 0) {
    WsArgs a;
    a.ldy = ldy > 0 ? ldy : CoutP;
    a.dbg = g_ws_variant == 3 ? 8 : (g_ws_variant == 8 ? 64 : (g_ws_variant == 10 ? 128 : 0)); a.xcd_map = g_ws_variant != 2;
    a.x = (const hx_t*)x; a.ldx = ldx; a.wp = (const hx_t*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (hx_t*)y; a.stat = stat; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY; ws_magic(a); a.tiles_per_wg = tiles_per_wg;
    if ((long long)a.ntiles * (a.tilesX > a.tilesY ? a.tilesX : a.tilesY) >= (1LL << 32)) return WM_E_SHAPE;   // (the mulhi tile geometry's range)
    a.reverse = wm_sweep_dir(reverse);
    a.bw_stats4 = bw_stats4; a.bw_ld = bw_ld; a.bw_coef = bw_coef; a.bw_gvec = bw_gvec;
    a.ry = (const hx_t*)ry; a.r_scale = r_scale; a.r_shift = r_shift;
    a.ay = (const hx_t*)ay; a.dy_out = (hx_t*)dy_out;
    const dim3 grid((unsigned)wgs), block(512);
    if (addend) {   // forward 64 -> 64 with a second tensor added before the statistics (the encoder's after-concat layer)
        if (Cin != 64 || CoutP != 64 || !in_scale || !stat || ay || ry || bw_stats4) return WM_E_SHAPE;
        a.ry = (const hx_t*)addend;
        hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, true, true, true, false, 0, false, true, true>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (act && ay) {   // backward of a conv + ELU layer: x = g, ay = the layer's output, dy_out = gz for the weight gradient, stat = the bias gradient's partial rows
        if (Cin != 64 || (CoutP != 64 && CoutP != 32) || ldx != 64 || in_scale || bw_stats4 || ry || !stat || addend) return WM_E_SHAPE;
        if (CoutP == 32 && a.ldy == 16)   // towards a 16-channel (stride) tensor: the 32-channel consumers, channels 16..31 (zero filter rows) not stored
            hipLaunchKernelGGL((conv3x3_ws_kernel<64, 32, false, false, true, false, 3, false, true, false, false, 0, true>), grid, block, 0, s, a, nullptr);
        else if (a.ldy != CoutP) return WM_E_SHAPE;
        else if (CoutP == 32) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 32, false, false, true, false, 3, false>), grid, block, 0, s, a, nullptr);
        else hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 3, false>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (act) {   // forward conv + bias + ELU (the pre-activation is not stored)
        if (CoutP != 64 || in_scale || stat || bw_stats4 || ry || addend) return WM_E_SHAPE;
        if (Cin == 64) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 0, false, true, false, false, 1>), grid, block, 0, s, a, nullptr);
        else if (Cin == 32) hipLaunchKernelGGL((conv3x3_ws_kernel<32, 64, false, false, true, false, 0, false, true, false, false, 1>), grid, block, 0, s, a, nullptr);
        else if (Cin == 16) hipLaunchKernelGGL((conv3x3_ws_kernel<16, 64, false, false, false, false, 0, false, true, false, false, 1>), grid, block, 0, s, a, nullptr);
        else return WM_E_SHAPE;
        return WM_OK;
    }
    if (ay) {   // dgrad with the BatchNorm-backward apply (tensor gradient) fused; x = g, dy written out for the weight gradient
        if (Cin != 64 || (CoutP != 64 && CoutP != 32) || ldx != 64 || in_scale || !bw_stats4 || !bw_coef || bw_gvec ||
            (ry != nullptr) != (stat != nullptr) || (ry && CoutP != 64))
            return WM_E_SHAPE;
        if (CoutP == 32) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 32, false, false, true, false, 2, false>), grid, block, 0, s, a, nullptr);   // image-fed layer: dx has 3 (-> 32) channels
        else if (ry) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 2, true>), grid, block, 0, s, a, nullptr);
        else hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 2, false>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (ry) {   // dgrad that also reduces the BatchNorm-backward sums of the layer it feeds
        if (CoutP != 64 || in_scale || !stat || (Cin != 64 && Cin != 32)) return WM_E_SHAPE;
        if (bw_stats4) {
            if (Cin == 64) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 1, true>), grid, block, 0, s, a, nullptr);
            else hipLaunchKernelGGL((conv3x3_ws_kernel<32, 64, false, false, true, false, 1, true>), grid, block, 0, s, a, nullptr);
        } else {
            if (Cin == 64) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 0, true>), grid, block, 0, s, a, nullptr);
            else hipLaunchKernelGGL((conv3x3_ws_kernel<32, 64, false, false, true, false, 0, true>), grid, block, 0, s, a, nullptr);
        }
        return WM_OK;
    }
    if (bw_stats4) {   // dgrad with the BatchNorm-backward apply (per-sample gradient rows) fused: 64 or 32 -> 64
        if ((Cin != 64 && Cin != 32) || CoutP != 64 || in_scale || stat) return WM_E_SHAPE;
        if (Cin == 64) hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, false, false, true, false, 1>), grid, block, 0, s, a, nullptr);
        else hipLaunchKernelGGL((conv3x3_ws_kernel<32, 64, false, false, true, false, 1>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    const bool xf = in_scale != nullptr, st = stat != nullptr;
#define WM_WS_LAUNCH2(CIN_, COUT_, M16_)                                                                                     \
    do {                                                                                                                     \
        if (xf && st) hipLaunchKernelGGL((conv3x3_ws_kernel<CIN_, COUT_, true, true, M16_>), grid, block, 0, s, a, nullptr);   \
        else if (xf) hipLaunchKernelGGL((conv3x3_ws_kernel<CIN_, COUT_, true, false, M16_>), grid, block, 0, s, a, nullptr);   \
        else if (st) hipLaunchKernelGGL((conv3x3_ws_kernel<CIN_, COUT_, false, true, M16_>), grid, block, 0, s, a, nullptr);   \
        else hipLaunchKernelGGL((conv3x3_ws_kernel<CIN_, COUT_, false, false, M16_>), grid, block, 0, s, a, nullptr);          \
    } while (0)
#define WM_WS_LAUNCH(CIN_, M16_) WM_WS_LAUNCH2(CIN_, 64, M16_)
    // 16x16x32 consumers by default where Cin allows (-4.5 % on the 64->64 conv in the training step, tools/ab_step.py)
    // whole 16 x 16 tiles (every shape of the benchmarked step): the form without the inside-the-image mask (-1.0 % step time; variant 11 = off)
    if (Cin == 64 && CoutP == 64 && xf && st && g_ws_variant == 0 && H % TH == 0 && W % TW == 0) {
        hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, true, true, true, false, 0, false, true, false, true>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (Cin == 16 && CoutP == 64 && !xf && st && g_ws_variant == 0 && H % TH == 0 && W % TW == 0) {   // the image-fed first layers' forward
        hipLaunchKernelGGL((conv3x3_ws_kernel<16, 64, false, true, false, false, 0, false, true, false, true>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (g_ws_variant == 9 && Cin == 64 && CoutP == 64 && xf && st) {   // knob 9: the statistics sums left to the compiler's placement
        hipLaunchKernelGGL((conv3x3_ws_kernel<64, 64, true, true, true, false, 0, false, false>), grid, block, 0, s, a, nullptr);
        return WM_OK;
    }
    if (CoutP == 32) WM_WS_LAUNCH2(64, 32, true);
    else if (Cin == 64 && g_ws_variant != 1) WM_WS_LAUNCH(64, true);
    else if (Cin == 64) WM_WS_LAUNCH(64, false);
    else if (Cin == 32) WM_WS_LAUNCH(32, true);
    else WM_WS_LAUNCH(16, false);
#undef WM_WS_LAUNCH
#undef WM_WS_LAUNCH2
    return WM_OK;
}
