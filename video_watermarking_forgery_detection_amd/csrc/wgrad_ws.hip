// Wave-specialised weight-gradient kernels (bf16) -- same role split as conv3x3_ws.hip:
//   waves 0-3  CONSUMERS: per 16x16-pixel tile 8 K-steps of 32 pixels x (dy fragments + 9 shifted x fragments, all through
//              ds_read_b64_tr_b16, v_mfma_f32_16x16x32_bf16); the tap accumulators of the wave's block live across the
//              whole run of tiles;
//   waves 4-7  PRODUCERS: global loads of the x halo tile and the dy tile two tiles ahead (registers), fused
//              BN+ReLU of x + zero padding, 16-byte LDS writes into the other buffer.
// One workgroup barrier per tile.  LDS (CI = 64): 2 x (x halo 41,472 B + dy 32,768 B) = 148,480 B.
// (A 32x32x16 form with 64-byte-half swizzle preceded this one: 1.6 % slower on the step; a paced / prioritised
// producer variant of it 1.4 % slower still -- tools/ab_step.py.)
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

int wm_sweep_dir(int reverse);   // conv3x3_ws.hip

namespace {

constexpr int TH = 16, TW = 16, HH = 18, HW = 18, CB = 64;
constexpr int NPIX = HH * HW;
constexpr int DVP = TH * TW * 8 / 256;        // dy vectors per producer thread (8)
constexpr int D_BYTES = TH * TW * CB * 2;
// Dynamic LDS requested on top of the 16-channel form's 86,272 static bytes, for NOTHING but the allocation's size: with it a workgroup
// takes 137,472 B of the CU's 163,840, which leaves no room for a workgroup of the JPEG kernels (26,624 B; csrc/jpeg.hip).  Round 4
// (tools/concurrent_kernels.py, tools/train_sanity_modes.py): while a workgroup of THIS kernel or of bwd_ws16.hip -- the two kernels with
// transposing LDS reads that are small enough to share a CU with them -- is resident beside a JPEG workgroup, that workgroup's results come
// out wrong in a quarter-wave of one register (8x8 blocks with two wrong pixels, 22-30 of 30 launches); nothing written out of bounds
// (LDS guard kernel, patterned arenas), not cured by workgroup barriers in the victim.  The cause is not understood; the two-chain schedule
// of the training step makes the pairing possible, so the pairing is made impossible.  This form runs one workgroup per CU anyway.
constexpr int WM_LDS_PAD16 = 137472 - (2 * (NPIX * 16 * 2 + D_BYTES) + 32);

struct WsWgArgs {
    const hx_t* x; int ldx; int CinX;
    const float* in_scale; const float* in_shift;
    const hx_t* dy; int lddy; int CoutY;   // DYF: the gradient wrt the layer's ReLU output (g), not dy
    // DYF (fused BatchNorm-backward apply): the layer's raw conv output and its BatchNorm constants
    const hx_t* yb; int ldyb; const float* bscale; const float* bshift; const float* bmean; const float* binvstd; const float* bcoef;
    const float* gvec; int ldgv;   // DYF == 2: g is one row per sample (the layer's output was globally pooled); dy/lddy unused
    float* ws;           // [gridDim.x][9][CinP][CoutP]
    int B, H, W, tilesX, tilesY, ntiles, ciBlocks, coBlocks;
    int reverse;
};

__device__ __forceinline__ hx8 tr_frag(const char* p0, const char* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(hx8, v);
}

// ----------------------------------------------------------------------------------------------------------------
// 16x16x32 form (v_mfma_f32_16x16x32_bf16): the chip holds a higher clock on this shape (see conv3x3_ws.hip), and a
// 16-row M lets the image-fed first layers (CinX = 16) run with CI = 16 instead of padding their input to 64 channels.
//   D[ci 16 x co 16] += A[ci x 32 pixels] * B[32 pixels x co]; a K-step is two tile rows; lane (r = lane&15, kq = lane>>4)
//   owns pixels 8kq .. 8kq+7 of the step = tile row (kq>>1), columns 8(kq&1) .. +7, fetched by two transposing reads.
// LDS: pixel rows of CI*2 bytes.  CI = 64 and the dy tile: byte offset XOR (((col>>1)&1) << 5 | ((col>>3)&1) << 6) -- the
// 8 pixels x 32 B one half-wave touches ({c..c+3} and {c+8..c+11}, any tap shift) fill a 256-byte bank row exactly once.
// CI = 16 (32-byte pixels): halo column c is stored at slot c ^ (((c>>3)&1) << 2), which does the same.
__device__ __forceinline__ int swz16(int col) { return (((col >> 1) & 1) << 5) | (((col >> 3) & 1) << 6); }
__device__ __forceinline__ int slot16(int col) { return col ^ (((col >> 3) & 1) << 2); }

// DYF: dy is not read but formed on the fly from (g, y) -- the apply pass of the BatchNorm backward, fused for layers whose dy
// has no other consumer (no input gradient wanted): dy = ca * (g*[scale*y+shift > 0] - c1 - (y-mean)*invstd * c2) in the folded
// form of wm_common.h, bit-identical to bn_bwd_kernel<bf16, APPLY>
// DYF == 2: the same with g[b, pixel, c] = gvec[b][c] (a globally pooled layer): the producers read y only
template <int CI, bool XFORM, int DYF = 0>
__global__ __launch_bounds__(512, 2) void wgrad_ws16_kernel(WsWgArgs a) {
    static_assert(CI == 64 || CI == 16, "input-channel block");
    constexpr int VPX = CI / 8;                                   // 16-byte vectors per x pixel
    constexpr int XV = (NPIX * VPX + 255) / 256;                  // x halo vectors per producer thread
    constexpr int XB = NPIX * CI * 2;                             // x halo tile bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (XB + D_BYTES)];
    __shared__ __attribute__((aligned(16))) float sK[DYF == 2 ? CB * 8 : 8];   // DYF == 2: scale, shift, k2, k3, ca of each of the block's 64 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int cc = blockIdx.y / a.coBlocks, oc = blockIdx.y % a.coBlocks;
    const int ci0 = cc * CI, co0 = oc * CB;
    if (DYF == 2) {   // (the producers have no registers to spare for the constants: they fetch them channel by channel while transforming)
        if (tid < CB) {
            const int c = co0 + tid;
            const bool okc = c < a.CoutY;
            const int cl = okc ? c : 0;
            float k2, k3;
            wm_bn_fold(a.bmean[cl], a.binvstd[cl], a.bcoef[cl], a.bcoef[a.ldgv + cl], a.bcoef[2 * a.ldgv + cl], k2, k3);
            const float v[8] = {a.bscale[cl], a.bshift[cl], k2, k3, a.bcoef[cl], 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) sK[tid * 8 + i] = okc ? v[i] : 0.f;
        }
        __syncthreads();
    }
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int t_begin = (int)(((long)run * a.ntiles) / G);
    const int t_end = (int)(((long)(run + 1) * a.ntiles) / G);
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;   // (wm_conv3x3_sweep_hint: start where the producer of dy stopped)
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef short i16x2 __attribute__((ext_vector_type(2)));

    if (producer) {
        // ================================================================== PRODUCER waves
        const int ptid = tid - 256;
        const int vx = ptid % VPX, vd = ptid & 7;
        const int cx = ci0 + vx * 8, cd = co0 + vd * 8;
        const bool cxok = cx < a.CinX, cdok = cd < a.CoutY;
        const int cxl = cxok ? cx : 0, cdl = cdok ? cd : 0;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (XFORM && cxok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale[cx + e]; sh[e] = a.in_shift[cx + e]; }
        }
        auto load_x = [&](const TileGeo& g, int k, hx8& dst, unsigned& okbits) {
            const int pix = min((ptid + 256 * k) / VPX, NPIX - 1);
            const int py = pix / HW, px = pix - py * HW;
            const int gy = g.ty0 - 1 + py, gx = g.tx0 - 1 + px;
            const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            dst = *reinterpret_cast<const hx8*>(a.x + ((size_t)(g.b * a.H + gyc) * a.W + gxc) * a.ldx + cxl);
            okbits |= ((cxok && gy == gyc && gx == gxc) ? 1u : 0u) << k;
        };
        float bsc[8], bsh[8], bca[8], bk2[8], bk3[8];   // DYF == 1: scale, shift, ca and the folded k2, k3 (wm_bn_fold) of this thread's 8 channels
        if (DYF == 1) {
            const int CP = a.ldgv;   // row pitch of the constants (= the layer's physical channel count)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bsc[e] = a.bscale[cdl + e]; bsh[e] = a.bshift[cdl + e]; bca[e] = a.bcoef[cdl + e];
                wm_bn_fold(a.bmean[cdl + e], a.binvstd[cdl + e], bca[e], a.bcoef[CP + cdl + e], a.bcoef[2 * CP + cdl + e], bk2[e], bk3[e]);
            }
        }
        auto load_d = [&](const TileGeo& g, int k, hx8& dst, hx8& ydst, unsigned& okbits) {
            const int pix = (ptid + 256 * k) >> 3;
            const int gy = g.ty0 + (pix >> 4), gx = g.tx0 + (pix & 15);
            const int gyc = min(gy, a.H - 1), gxc = min(gx, a.W - 1);
            const size_t pofs = (size_t)(g.b * a.H + gyc) * a.W + gxc;
            if (DYF == 2) dst = *reinterpret_cast<const hx8*>(a.yb + pofs * a.ldyb + cdl);
            else dst = *reinterpret_cast<const hx8*>(a.dy + pofs * a.lddy + cdl);
            if (DYF == 1) ydst = *reinterpret_cast<const hx8*>(a.yb + pofs * a.ldyb + cdl);
            okbits |= ((cdok && gy == gyc && gx == gxc) ? 1u : 0u) << k;
        };
        auto put_x = [&](unsigned char* base, int k, const hx8& src, bool ok) {
            const int pix = (ptid + 256 * k) / VPX;
            const int py = pix / HW, px = pix - py * HW;
            u32x4 w = __builtin_bit_cast(u32x4, src);
            if (XFORM) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const float f0 = __builtin_fmaf(HX::lo(w[pq]), sc[2 * pq], sh[2 * pq]);
                    const float f1 = __builtin_fmaf(HX::hi(w[pq]), sc[2 * pq + 1], sh[2 * pq + 1]);
                    const hx2 pk = HX::pack2(f0, f1);
                    const i16x2 z = {0, 0};
                    w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                }
            }
            const unsigned keep = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] &= keep;
            const int off = CI == 64 ? (py * HW + px) * 128 + ((vx * 16) ^ swz16(px)) : (py * HW + slot16(px)) * 32 + vx * 16;
            if (pix < NPIX) *reinterpret_cast<u32x4*>(base + off) = w;
        };
        auto put_d = [&](unsigned char* base, int k, const hx8& src, const hx8& ysrc, bool ok) {
            const int pix = (ptid + 256 * k) >> 3;
            u32x4 w = __builtin_bit_cast(u32x4, src);
            if (DYF == 1) {
                const u32x4 yw = __builtin_bit_cast(u32x4, ysrc);
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    float dd[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const int e = 2 * pq + hlf;
                        const float gg = hlf ? HX::hi(w[pq]) : HX::lo(w[pq]);
                        const float yy = hlf ? HX::hi(yw[pq]) : HX::lo(yw[pq]);
                        dd[hlf] = wm_bn_fold_dyg(yy, gg, bsc[e], bsh[e], bca[e], bk2[e], bk3[e]);
                    }
                    const hx2 pk = HX::pack2(dd[0], dd[1]);
                    w[pq] = __builtin_bit_cast(unsigned, pk);
                }
            }
            const unsigned keep = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] &= keep;
            *reinterpret_cast<u32x4*>(base + XB + pix * 128 + ((vd * 16) ^ swz16(pix & 15))) = w;
        };
        // DYF == 2: the 8 staged y vectors of a tile -> dy, in place, one channel pair at a time (its constants come from the
        // LDS, its gradient from the sample's gvec row: a dozen registers live instead of 64)
        auto gv_apply = [&](hx8 (&d)[DVP], int b) {
            if (DYF != 2) return;
            const float* gv = a.gvec + (size_t)b * a.ldgv + cdl;
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {   // a channel pair = one dword of each staged vector
                const f32x4 ka = *reinterpret_cast<const f32x4*>(sK + (vd * 8 + 2 * pq) * 8);       // scale, shift, k2, k3 (wm_bn_fold)
                const f32x4 kb = *reinterpret_cast<const f32x4*>(sK + (vd * 8 + 2 * pq + 1) * 8);
                const float k3ga = wm_bn_fold_g(sK[(vd * 8 + 2 * pq) * 8 + 4], gv[2 * pq], ka[3]);
                const float k3gb = wm_bn_fold_g(sK[(vd * 8 + 2 * pq + 1) * 8 + 4], gv[2 * pq + 1], kb[3]);
#pragma unroll
                for (int k = 0; k < DVP; ++k) {
                    u32x4 w = __builtin_bit_cast(u32x4, d[k]);
                    const float da = wm_bn_fold_dy(HX::lo(w[pq]), ka[0], ka[1], ka[2], ka[3], k3ga);
                    const float db = wm_bn_fold_dy(HX::hi(w[pq]), kb[0], kb[1], kb[2], kb[3], k3gb);
                    const hx2 pk = HX::pack2(da, db);
                    w[pq] = __builtin_bit_cast(unsigned, pk);
                    d[k] = __builtin_bit_cast(hx8, w);
                }
            }
        };
        hx8 x0[XV], x1[XV], d0[DVP], d1[DVP], y0[DVP], y1[DVP];
        unsigned okx0 = 0, okx1 = 0, okd0 = 0, okd1 = 0;
        if (t_begin < t_end) {
            const TileGeo g0 = geo(t_begin);
#pragma unroll
            for (int k = 0; k < XV; ++k) load_x(g0, k, x0[k], okx0);
#pragma unroll
            for (int k = 0; k < DVP; ++k) load_d(g0, k, d0[k], y0[k], okd0);
        }
        if (t_begin + 1 < t_end) {
            const TileGeo g1 = geo(t_begin + 1);
#pragma unroll
            for (int k = 0; k < XV; ++k) load_x(g1, k, x1[k], okx1);
#pragma unroll
            for (int k = 0; k < DVP; ++k) load_d(g1, k, d1[k], y1[k], okd1);
        }
        if (t_begin < t_end) {
            gv_apply(d0, geo(t_begin).b);
#pragma unroll
            for (int k = 0; k < XV; ++k) put_x(smem, k, x0[k], (okx0 >> k) & 1u);
#pragma unroll
            for (int k = 0; k < DVP; ++k) put_d(smem, k, d0[k], y0[k], (okd0 >> k) & 1u);
        }
        __syncthreads();
        for (int tile = t_begin; tile < t_end; ++tile) {
            unsigned char* nb = smem + ((((tile - t_begin) & 1) ^ 1) * (XB + D_BYTES));
#pragma unroll
            for (int k = 0; k < XV; ++k) x0[k] = x1[k];
#pragma unroll
            for (int k = 0; k < DVP; ++k) { d0[k] = d1[k]; if (DYF == 1) y0[k] = y1[k]; }
            okx0 = okx1; okd0 = okd1; okx1 = 0; okd1 = 0;
            if (tile + 2 < t_end) {
                const TileGeo g2 = geo(tile + 2);
#pragma unroll
                for (int k = 0; k < XV; ++k) load_x(g2, k, x1[k], okx1);
#pragma unroll
                for (int k = 0; k < DVP; ++k) load_d(g2, k, d1[k], y1[k], okd1);
            }
            if (tile + 1 < t_end) {
#pragma unroll
                for (int k = 0; k < XV; ++k) put_x(nb, k, x0[k], (okx0 >> k) & 1u);
                gv_apply(d0, geo(tile + 1).b);
#pragma unroll
                for (int k = 0; k < DVP; ++k) put_d(nb, k, d0[k], y0[k], (okd0 >> k) & 1u);
            }
            __syncthreads();
        }
        return;
    }

    // ====================================================================== CONSUMER waves
    // CI = 64: wave (mi, ni) owns the 32 ci x 32 co block = 2 x 2 fragments x 9 taps; CI = 16: wave w owns co fragment w
    constexpr int FI = CI == 64 ? 2 : 1, FJ = CI == 64 ? 2 : 1;
    const int mi = CI == 64 ? (wave >> 1) : 0, ni = CI == 64 ? (wave & 1) : 0;
    const int r = lane & 15, kq = lane >> 4;
    const int q = (lane >> 2) & 3, p = lane & 3;
    f32x4 acc[9][FI][FJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane part of the fragment addresses (tile row (kq>>1) of the K-step, column 8(kq&1) + q [+4]); the K-step / tap rows
    // are compile-time offsets
    const int colb = 8 * (kq & 1) + q;
    int xo[3][2][FI], dof[2][FJ];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                const int col = colb + kw + 4 * s2;
                if (CI == 64) xo[kw][s2][fi] = ((kq >> 1) * HW + col) * 128 + (((mi * 32 + fi * 16 + 4 * p) * 2) ^ swz16(col));
                else xo[kw][s2][fi] = ((kq >> 1) * HW + slot16(col)) * 32 + 4 * p * 2;
            }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int fj = 0; fj < FJ; ++fj) {
            const int col = colb + 4 * s2;
            const int cb = CI == 64 ? (ni * 32 + fj * 16 + 4 * p) * 2 : (wave * 16 + 4 * p) * 2;
            dof[s2][fj] = ((kq >> 1) * TW + col) * 128 + (cb ^ swz16(col));
        }
    constexpr int XROW = HW * CI * 2;     // bytes per halo row
    __syncthreads();

    for (int tile = t_begin; tile < t_end; ++tile) {
        const char* sXc = reinterpret_cast<const char*>(smem + (((tile - t_begin) & 1) * (XB + D_BYTES)));
        const char* sDc = sXc + XB;
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            hx8 bfrag[FJ];
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj)
                bfrag[fj] = tr_frag(sDc + 2 * ks * TW * 128 + dof[0][fj], sDc + 2 * ks * TW * 128 + dof[1][fj]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                hx8 afrag[FI];
#pragma unroll
                for (int fi = 0; fi < FI; ++fi)
                    afrag[fi] = tr_frag(sXc + (2 * ks + kh) * XROW + xo[kw][0][fi], sXc + (2 * ks + kh) * XROW + xo[kw][1][fi]);
#pragma unroll
                for (int fi = 0; fi < FI; ++fi)
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj)
                        acc[tap][fi][fj] = HX::mfma16(afrag[fi], bfrag[fj], acc[tap][fi][fj]);
            }
        }
        __syncthreads();
    }
    // slab write: acc[tap][fi][fj][i] -> ci row 4*kq + i of fragment fi, co column r of fragment fj
    const int CinP = a.ciBlocks * CB, CoutP = a.coBlocks * CB;
    float* slab = a.ws + (size_t)blockIdx.x * 9 * CinP * CoutP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fi = 0; fi < FI; ++fi)
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ci = ci0 + (CI == 64 ? mi * 32 + fi * 16 : 0) + 4 * kq + i;
                    const int co = co0 + (CI == 64 ? ni * 32 + fj * 16 : wave * 16) + r;
                    slab[((size_t)tap * CinP + ci) * CoutP + co] = acc[tap][fi][fj][i];
                }
}

}  // namespace

void WM_HSYM(wm_launch_wgrad_ws)(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const void* dy,
                        int lddy, int CoutY, float* ws, int B, int H, int W, int nslabs, hipStream_t s, int reverse, const void* yb = nullptr,
                        int ldyb = 0, const float* bstats4 = nullptr, int bstats_ld = 0, const float* bcoef = nullptr,
                        const float* gvec = nullptr) {
    WsWgArgs a;
    a.x = (const hx_t*)x; a.ldx = ldx; a.CinX = CinX; a.in_scale = in_scale; a.in_shift = in_shift;
    a.dy = (const hx_t*)dy; a.lddy = lddy; a.CoutY = CoutY; a.ws = ws; a.B = B; a.H = H; a.W = W;
    a.yb = (const hx_t*)yb; a.ldyb = ldyb;
    a.bscale = bstats4; a.bshift = bstats4 ? bstats4 + bstats_ld : nullptr; a.bmean = bstats4 ? bstats4 + 2 * bstats_ld : nullptr;
    a.binvstd = bstats4 ? bstats4 + 3 * bstats_ld : nullptr; a.bcoef = bcoef; a.gvec = gvec; a.ldgv = bstats_ld;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY;
    a.reverse = wm_sweep_dir(reverse);
    a.ciBlocks = wm_cdiv(CinX, CB); a.coBlocks = wm_cdiv(CoutY, CB);
    const dim3 block(512);
    if (CinX <= 16) {   // one 16-channel input block; the slab keeps its 64-row pitch (rows >= 16 are never read back)
        a.ciBlocks = 1;
        const dim3 grid((unsigned)nslabs, (unsigned)a.coBlocks);
        if (yb) {       // fused BatchNorm-backward apply (image-fed first layers whose input needs no gradient)
            if (in_scale) hipLaunchKernelGGL((wgrad_ws16_kernel<16, true, 1>), grid, block, WM_LDS_PAD16, s, a);
            else hipLaunchKernelGGL((wgrad_ws16_kernel<16, false, 1>), grid, block, WM_LDS_PAD16, s, a);
        } else {
            if (in_scale) hipLaunchKernelGGL((wgrad_ws16_kernel<16, true>), grid, block, WM_LDS_PAD16, s, a);
            else hipLaunchKernelGGL((wgrad_ws16_kernel<16, false>), grid, block, WM_LDS_PAD16, s, a);
        }
    } else {
        const dim3 grid((unsigned)nslabs, (unsigned)(a.ciBlocks * a.coBlocks));
        if (gvec) {     // fused BatchNorm-backward apply of a globally pooled layer (its input is always an activated tensor)
            hipLaunchKernelGGL((wgrad_ws16_kernel<64, true, 2>), grid, block, 0, s, a);
        } else if (in_scale) hipLaunchKernelGGL((wgrad_ws16_kernel<64, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_ws16_kernel<64, false>), grid, block, 0, s, a);
    }
}
