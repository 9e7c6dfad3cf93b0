// Wave-specialised persistent 3x3 convolution for bf16, Cin = Cout = 64 (gfx950).
//
// Why: tools/phase_c64.py shows that a single wave per SIMD cannot overlap its own phases -- the MFMA loop
// (4,742 cycles / tile), the HBM traffic of a tile (78 KB per CU = ~7,400 cycles at the ~10.7 B/clk a CU gets)
// and the VALU work (BN+ReLU transform, bias / statistics / pack) simply add up (10-12k cycles / tile), whatever
// the instruction order.  Here the two kinds of work live in different waves of one 512-thread workgroup, two
// waves per SIMD, so the hardware issues them concurrently:
//   waves 0-3  CONSUMERS: 36 x (4 ds_read_b128 + 4 v_mfma_f32_32x32x16_bf16) on halo tile X[t&1] and the resident
//              filter, then the epilogue from registers: bias, BatchNorm partial sums, bf16 pack, a 4x4 quad
//              transpose with DPP (a lane ends up with 8 adjacent channels of a pixel) and 8 dwordx4 stores per
//              lane -- 1 KB contiguous per store instruction, no LDS staging;
//   waves 4-7  PRODUCERS: global loads of the halo of tile t+2 (registers, two tiles ahead), fused BN+ReLU +
//              zero padding of tile t+1, 16-byte LDS writes into X[(t+1)&1].
// One workgroup barrier per tile hands X[(t+1)&1] to the consumers and X[t&1] back to the producers.
// LDS: filter 73,728 B + 2 x 41,472 B halo tiles + 2 KB = 158,720 B: rows are 128 B (no padding), the 16-byte
// column index is XOR-swizzled with (row >> 1) & 7, which makes 16 consecutive rows hit 16 distinct bank slots
// for ds_read_b128 and for the producers' ds_write_b128.
#include <stdlib.h>
#include "wm_common.h"

namespace {

constexpr int TH = 16, TW = 16, HH = 18, HW = 18;
constexpr int C64 = 64;
constexpr int NPIX = HH * HW;                  // 324 halo pixels
constexpr int XVP = (NPIX * 8 + 255) / 256;    // halo vectors per PRODUCER thread (256 producer threads): 11

struct WsArgs {
    const bf16_t* x; int ldx;
    const bf16_t* wp;            // [9][64][64]
    const float* bias; int nbias;
    const float* in_scale; const float* in_shift;
    bf16_t* y;                   // dense [B,H,W,64]
    float* stat;                 // [gridDim.x][2][64] or null
    int B, H, W, tilesX, tilesY, ntiles, tiles_per_wg;
};

// 16-byte column swizzles.  Filter rows: key = (row >> 1) & 7.  Halo pixels: key = (halo column >> 1) & 7 -- it does
// not depend on the halo ROW, so for a consumer lane the swizzled address of tap (kh, kw) is a per-(kw, k-step)
// register plus a compile-time (kh, M-fragment) offset; 16 consecutive pixels of a row (from any kw) and the
// pixels of the row below all land on distinct 16-byte bank slots.
__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_px(int px, int slot) { return slot ^ ((px >> 1) & 7); }

template <bool XFORM, bool STATS>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64_kernel(WsArgs a) {
    constexpr int SW_BYTES = 9 * C64 * C64 * 2, SX_BYTES = NPIX * C64 * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * SX_BYTES + 4 * 2 * C64 * 4];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sX0 = reinterpret_cast<bf16_t*>(smem + SW_BYTES);  // two halo tiles back to back
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + 2 * SX_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;

    // ---- filter -> LDS (all 512 threads): output channel n at row (n&1)*32 + (n>>1), swizzled 16-byte columns
    {
        constexpr int WV = 9 * C64 * 8 / 512;  // 9 vectors per thread
        bf16x8 wv[WV];
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 512 * k;
            wv[k] = *reinterpret_cast<const bf16x8*>(a.wp + (size_t)(i >> 3) * C64 + (i & 7) * 8);
        }
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 512 * k;
            const int row = i >> 3, tap = row >> 6, n = row & 63;
            const int lrow = tap * C64 + (n & 1) * 32 + (n >> 1);
            *reinterpret_cast<bf16x8*>(sW + lrow * C64 + swz(lrow, i & 7) * 8) = wv[k];
        }
    }

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = tile;
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };

    if (producer) {
        // ================================================================== PRODUCER waves
        const int ptid = tid - 256;
        const int vec = ptid & 7;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (XFORM) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale[vec * 8 + e]; sh[e] = a.in_shift[vec * 8 + e]; }
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        auto load_one = [&](const TileGeo& g, int k, bf16x8& dst, unsigned& okbits) {
            // clamped, always valid address (never a load under a per-lane branch); validity as a bit
            const int pix = min((ptid + 256 * k) >> 3, NPIX - 1);
            const int py = pix / HW, px = pix - py * HW;
            const int gy = g.ty0 - 1 + py, gx = g.tx0 - 1 + px;
            const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            dst = *reinterpret_cast<const bf16x8*>(a.x + ((size_t)(g.b * a.H + gyc) * a.W + gxc) * a.ldx + vec * 8);
            const unsigned okb = (gy == gyc && gx == gxc) ? 1u : 0u;
            okbits |= okb << k;
        };
        auto put_one = [&](bf16_t* sX, int k, const bf16x8& src, bool ok) {
            const int pix = (ptid + 256 * k) >> 3;
            u32x4 w = __builtin_bit_cast(u32x4, src);
            if (XFORM) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    f32x2 f = {__builtin_bit_cast(float, w[pq] << 16), __builtin_bit_cast(float, w[pq] & 0xffff0000u)};
                    const f32x2 s2 = {sc[2 * pq], sc[2 * pq + 1]}, h2 = {sh[2 * pq], sh[2 * pq + 1]};
                    f = f * s2 + h2;
                    const bf16x2 pk = {(bf16_t)fmaxf(f[0], 0.f), (bf16_t)fmaxf(f[1], 0.f)};
                    w[pq] = __builtin_bit_cast(unsigned, pk);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = ok ? w[q] : 0u;   // zero padding AFTER the activation
            if (pix < NPIX) *reinterpret_cast<u32x4*>(sX + pix * C64 + swz_px(pix % HW, vec) * 8) = w;
        };
        bf16x8 d0[XVP], d1[XVP];
        unsigned ok0 = 0, ok1 = 0;
        if (t_begin < t_end) {
            const TileGeo g0 = geo(t_begin);
#pragma unroll
            for (int k = 0; k < XVP; ++k) load_one(g0, k, d0[k], ok0);
        }
        if (t_begin + 1 < t_end) {
            const TileGeo g1 = geo(t_begin + 1);
#pragma unroll
            for (int k = 0; k < XVP; ++k) load_one(g1, k, d1[k], ok1);
        }
        if (t_begin < t_end) {
#pragma unroll
            for (int k = 0; k < XVP; ++k) put_one(sX0, k, d0[k], (ok0 >> k) & 1u);
        }
        __syncthreads();  // filter + first halo tile visible
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int nb = ((tile - t_begin) & 1) ^ 1;
            // d1 holds tile+1 (loaded one iteration ago); fetch tile+2 into d0, then publish tile+1
#pragma unroll
            for (int k = 0; k < XVP; ++k) d0[k] = d1[k];
            ok0 = ok1;
            ok1 = 0;
            if (tile + 2 < t_end) {
                const TileGeo g2 = geo(tile + 2);
#pragma unroll
                for (int k = 0; k < XVP; ++k) load_one(g2, k, d1[k], ok1);
            }
            if (tile + 1 < t_end) {
#pragma unroll
                for (int k = 0; k < XVP; ++k) put_one(sX0 + nb * (NPIX * C64), k, d0[k], (ok0 >> k) & 1u);
            }
            __syncthreads();
        }
        if (STATS) __syncthreads();
        return;
    }

    // ====================================================================== CONSUMER waves
    const int r = lane & 31, h = lane >> 5;
    float bv[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) bv[nf] = (a.bias && 2 * r + nf < a.nbias) ? a.bias[2 * r + nf] : 0.f;
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
    const int q = r & 3, m = r >> 2;
    // swizzled LDS byte offsets of this lane's fragments (see swz_px): 12 + 4 registers, everything else immediate
    int aoff[3][4], boff[4];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int px = (r & 15) + kw;
            aoff[kw][ks] = (((wave * 4 + (r >> 4)) * HW + px) * C64 + swz_px(px, ks * 2 + h) * 8) * 2;
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[ks] = (r * C64 + swz(r, ks * 2 + h) * 8) * 2;
    __syncthreads();  // filter + first halo tile visible

    for (int tile = t_begin; tile < t_end; ++tile) {
        const TileGeo g = geo(tile);
        const bf16_t* sX = sX0 + ((tile - t_begin) & 1) * (NPIX * C64);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        bf16x8 af[2][2], bfr[2][2];
        auto load_frags = [&](int sidx, int buf) {
            const int tap = sidx >> 2, ks = sidx & 3;
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
                af[buf][mf] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(sX) + aoff[kw][ks] + (mf * 2 + kh) * (HW * C64 * 2));
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                bfr[buf][nf] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * C64 + nf * 32) * (C64 * 2));
        };
        load_frags(0, 0);
#pragma unroll
        for (int sidx = 0; sidx < 36; ++sidx) {
            const int cb = sidx & 1;
            if (sidx + 1 < 36) load_frags(sidx + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);  // reads of step s+1 stay ahead of the MFMAs of step s
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][mf], bfr[cb][nf], acc[mf][nf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue from registers.  lane (r,h) holds channels (2r, 2r+1) of pixel rows (i&3)+8(i>>2)+4h
        const bool full_tile = g.ty0 + TH <= a.H && g.tx0 + TW <= a.W;
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            unsigned pk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v0 = acc[mf][0][i] + bv[0], v1 = acc[mf][1][i] + bv[1];
                if (STATS) {
                    float mk = 1.f;
                    if (!full_tile) {
                        const int prow = (i & 3) + 8 * (i >> 2) + 4 * h;
                        mk = ((g.ty0 + wave * 4 + mf * 2 + (prow >> 4) < a.H) & (g.tx0 + (prow & 15) < a.W)) ? 1.f : 0.f;
                    }
                    st1[0] += mk * v0; st2[0] += mk * v0 * v0;
                    st1[1] += mk * v1; st2[1] += mk * v1 * v1;
                }
                const bf16x2 p2 = {(bf16_t)v0, (bf16_t)v1};
                pk[i] = __builtin_bit_cast(unsigned, p2);
            }
            // 4x4 transpose inside each quad of lanes (DPP quad_perm): afterwards lane q of the quad holds, for
            // pixel element i = 4g+q, the pairs of lanes 0..3 = channels 8m .. 8m+7: one 16-byte vector
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                unsigned a0 = pk[4 * gq], a1 = pk[4 * gq + 1], a2 = pk[4 * gq + 2], a3 = pk[4 * gq + 3];
                {   // lane bit 0 <-> register bit 0
                    const unsigned s01 = (q & 1) ? a0 : a1, s23 = (q & 1) ? a2 : a3;
                    const unsigned r01 = __builtin_amdgcn_mov_dpp(s01, 0xB1, 0xf, 0xf, true);
                    const unsigned r23 = __builtin_amdgcn_mov_dpp(s23, 0xB1, 0xf, 0xf, true);
                    if (q & 1) { a0 = r01; a2 = r23; } else { a1 = r01; a3 = r23; }
                }
                {   // lane bit 1 <-> register bit 1
                    const unsigned s02 = (q & 2) ? a0 : a2, s13 = (q & 2) ? a1 : a3;
                    const unsigned r02 = __builtin_amdgcn_mov_dpp(s02, 0x4E, 0xf, 0xf, true);
                    const unsigned r13 = __builtin_amdgcn_mov_dpp(s13, 0x4E, 0xf, 0xf, true);
                    if (q & 2) { a0 = r02; a1 = r13; } else { a2 = r02; a3 = r13; }
                }
                const int prow = q + 8 * gq + 4 * h;                 // element i = 4*gq + q
                const int gy = g.ty0 + wave * 4 + mf * 2 + (prow >> 4), gx = g.tx0 + (prow & 15);
                const u32x4 v = {a0, a1, a2, a3};
                if (full_tile || (gy < a.H && gx < a.W))
                    *reinterpret_cast<u32x4*>(a.y + (((size_t)g.b * a.H + gy) * a.W + gx) * C64 + 8 * m) = v;
            }
        }
        __syncthreads();  // X[t&1] is free for the producers, X[(t+1)&1] is ready
    }
    if (STATS) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            const float s1 = st1[nf] + __shfl_xor(st1[nf], 32, 64);
            const float s2 = st2[nf] + __shfl_xor(st2[nf], 32, 64);
            if (h == 0) {
                sRed[(wave * 2 + 0) * C64 + 2 * r + nf] = s1;
                sRed[(wave * 2 + 1) * C64 + 2 * r + nf] = s2;
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

// launcher used by conv3x3.hip
int wm_launch_conv3x3_ws64(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                           const float* in_shift, void* y, float* stat, int B, int H, int W, int wgs, int tiles_per_wg,
                           hipStream_t s) {
    WsArgs a;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.wp = (const bf16_t*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (bf16_t*)y; a.stat = stat; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY; a.tiles_per_wg = tiles_per_wg;
    const dim3 grid((unsigned)wgs), block(512);
    const bool xf = in_scale != nullptr, st = stat != nullptr;
    if (xf && st) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, true>), grid, block, 0, s, a);
    else if (xf) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, false>), grid, block, 0, s, a);
    else if (st) hipLaunchKernelGGL((conv3x3_ws64_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_ws64_kernel<false, false>), grid, block, 0, s, a);
    return WM_OK;
}
