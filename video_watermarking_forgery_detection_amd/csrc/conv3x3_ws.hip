// Wave-specialised persistent 3x3 convolution for bf16, Cin = Cout = 64 (gfx950).
//
// Why: tools/phase_c64.py shows that a single wave per SIMD cannot overlap its own phases -- the MFMA loop
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
    int dbg;      // STAMPS build only: 1 = skip the MFMA loop, 2 = skip the stores, 4 = skip the halo loads
    int reverse;  // walk the workgroup's run of tiles backwards (Infinity Cache reuse of the previous kernel's tail)
};

// 16-byte column swizzles.  Filter rows: key = (row >> 1) & 7.  Halo pixels: key = (halo column >> 1) & 7 -- it does
// not depend on the halo ROW, so for a consumer lane the swizzled address of tap (kh, kw) is a per-(kw, k-step)
// register plus a compile-time (kh, M-fragment) offset; 16 consecutive pixels of a row (from any kw) and the
// pixels of the row below all land on distinct 16-byte bank slots.
__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_px(int px, int slot) { return slot ^ ((px >> 1) & 7); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// STAMPS: diagnostic build only (tools/phase_ws.py) -- per-role cycle totals of the phases of the tile loop
template <bool XFORM, bool STATS, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64_kernel(WsArgs a, unsigned long long* __restrict__ stamps = nullptr) {
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
    constexpr int SW_BYTES = 9 * C64 * C64 * 2, SX_BYTES = NPIX * C64 * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * SX_BYTES + 4 * 2 * C64 * 4 + C64 * 4];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sX0 = reinterpret_cast<bf16_t*>(smem + SW_BYTES);  // two halo tiles back to back
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + 2 * SX_BYTES);
    float* sBias = sRed + 4 * 2 * C64;   // the accumulators start from the bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    if (tid < C64) sBias[tid] = (a.bias && tid < a.nbias) ? a.bias[tid] : 0.f;

    // ---- filter -> LDS (all 512 threads).  The filter is the A operand of the MFMA (D rows = output channels, D
    // columns = pixels), so a lane of the accumulator tile holds ONE pixel and, in its 16 registers, the MFMA rows
    // (i&3) + 8(i>>2) + 4h.  Output channel c of a 32-channel fragment is therefore stored at filter row
    // rho(c) = (i&3) + 8(i>>2) + 4hh with hh = c>>4, i = c&15: register i of lane half h is channel 16h + i, and a
    // lane owns 16 ADJACENT channels of its pixel -- two 16-byte stores, no transpose.
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
            const int c = n & 31, ci = c & 15;
            const int rho = (ci & 3) + 8 * (ci >> 2) + 4 * (c >> 4);
            const int lrow = tap * C64 + (n >> 5) * 32 + rho;
            *reinterpret_cast<bf16x8*>(sW + lrow * C64 + swz(lrow, i & 7) * 8) = wv[k];
        }
    }

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };

    if (producer) {
        // ================================================================== PRODUCER waves
        const int ptid = tid - 256;
        const int vec = ptid & 7;
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
        // tile-invariant per-vector geometry: halo pixel (py, px), its offset inside an interior tile, its LDS slot
        int hpy[XVP], hpx[XVP], rel[XVP], lds[XVP];
#pragma unroll
        for (int k = 0; k < XVP; ++k) {
            const int pix = min((ptid + 256 * k) >> 3, NPIX - 1);
            hpy[k] = pix / HW; hpx[k] = pix - hpy[k] * HW;
            rel[k] = (hpy[k] * a.W + hpx[k]) * a.ldx + vec * 8;
            lds[k] = pix * C64 + swz_px(hpx[k], vec) * 8;
        }
        const bool last_live = ((ptid + 256 * (XVP - 1)) >> 3) < NPIX;
        // loads: always a valid address, never under a per-lane branch; the branch on `interior` is wave-uniform
        auto load_tile = [&](const TileGeo& g, bf16x8 (&d)[XVP], unsigned& okbits) {
            if (STAMPS && (a.dbg & 4)) { okbits = 0xffffffffu; return; }
            const bool interior = g.ty0 >= 1 && g.ty0 + TH + 1 <= a.H && g.tx0 >= 1 && g.tx0 + TW + 1 <= a.W;
            if (interior) {
                const bf16_t* xt = a.x + ((size_t)(g.b * a.H + g.ty0 - 1) * a.W + (g.tx0 - 1)) * a.ldx;
#pragma unroll
                for (int k = 0; k < XVP; ++k) d[k] = *reinterpret_cast<const bf16x8*>(xt + rel[k]);
                okbits = 0xffffffffu;
            } else {
                const bf16_t* xb = a.x + (size_t)g.b * a.H * a.W * a.ldx;
                okbits = 0x80000000u;  // bit 31: "this tile needs masking"
#pragma unroll
                for (int k = 0; k < XVP; ++k) {
                    const int gy = g.ty0 - 1 + hpy[k], gx = g.tx0 - 1 + hpx[k];
                    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
                    d[k] = *reinterpret_cast<const bf16x8*>(xb + (gyc * a.W + gxc) * a.ldx + vec * 8);
                    okbits |= ((gy == gyc && gx == gxc) ? 1u : 0u) << k;
                }
            }
        };
        // fused BN + ReLU (ReLU on the packed bf16 pair as a signed 16-bit max), zero padding AFTER the activation
        auto put_tile = [&](bf16_t* sX, const bf16x8 (&d)[XVP], unsigned okbits) {
            const bool masked = okbits != 0xffffffffu;   // wave-uniform (per tile)
#pragma unroll
            for (int k = 0; k < XVP; ++k) {
                u32x4 w = __builtin_bit_cast(u32x4, d[k]);
                if (XFORM) {
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        f32x2 f = {__builtin_bit_cast(float, w[pq] << 16), __builtin_bit_cast(float, w[pq] & 0xffff0000u)};
                        f = f * sc2[pq] + sh2[pq];
                        const bf16x2 pk = {(bf16_t)f[0], (bf16_t)f[1]};
                        const i16x2 z = {0, 0};
                        w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                    }
                }
                if (masked) {
                    const unsigned keep = ((okbits >> k) & 1u) ? 0xffffffffu : 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] &= keep;
                }
                if (k + 1 < XVP || last_live) *reinterpret_cast<u32x4*>(sX + lds[k]) = w;
            }
        };
        bf16x8 dA[XVP], dB[XVP];
        unsigned okA = 0, okB = 0;
        if (t_begin < t_end) load_tile(geo(t_begin), dA, okA);
        if (t_begin + 1 < t_end) load_tile(geo(t_begin + 1), dB, okB);
        if (t_begin < t_end) put_tile(sX0, dA, okA);
        __syncthreads();  // filter + first halo tile visible
        stamp(-1);
        // iteration `tile`: `cur` holds tile+1 (loaded one iteration ago): fetch tile+2 into `nxt`, then publish tile+1
        auto iter = [&](int tile, bf16x8 (&nxt)[XVP], unsigned& oknxt, const bf16x8 (&cur)[XVP], unsigned okcur) {
            const int nb = ((tile - t_begin) & 1) ^ 1;
            if (tile + 2 < t_end) load_tile(geo(tile + 2), nxt, oknxt);
            stamp(0);  // load issue
            if (tile + 1 < t_end) put_tile(sX0 + nb * (NPIX * C64), cur, okcur);
            if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(1);  // wait for the tile loaded one iteration ago + transform + LDS writes
            __syncthreads();
            stamp(2);  // barrier
        };
        for (int tile = t_begin; tile < t_end; tile += 2) {
            iter(tile, dA, okA, dB, okB);
            if (tile + 1 < t_end) iter(tile + 1, dB, okB, dA, okA);
        }
        if (STATS) __syncthreads();
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
    // register i of channel fragment nf = output channel nf*32 + 16h + i
    const int r = lane & 31, h = lane >> 5;
    f32x2 s1[2][8], s2[2][8];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[nf][j] = f32x2{0.f, 0.f}; s2[nf][j] = f32x2{0.f, 0.f}; }
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
    stamp(-1);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const TileGeo g = geo(tile);
        const bf16_t* sX = sX0 + ((tile - t_begin) & 1) * (NPIX * C64);
        f32x16 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[0][j] = *reinterpret_cast<const f32x16*>(sBias + j * 32 + 16 * h);
            acc[1][j] = acc[0][j];
        }
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
        if (!(STAMPS && (a.dbg & 1))) {
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
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cb][nf], af[cb][mf], acc[mf][nf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        stamp(0);  // MFMA loop
        // ---- epilogue from registers (bias already inside): BatchNorm partial sums (packed f32 pairs), bf16 pack, 2 x 16-byte stores
        const bool full_tile = g.ty0 + TH <= a.H && g.tx0 + TW <= a.W;   // wave-uniform
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const int gy = g.ty0 + wave * 4 + mf * 2 + (r >> 4), gx = g.tx0 + (r & 15);
            const bool inb = full_tile || (gy < a.H && gx < a.W);
            bf16_t* yp = a.y + (((size_t)g.b * a.H + gy) * a.W + gx) * C64 + 16 * h;
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                unsigned pk[8];
                f32x2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f32x2{acc[mf][nf][2 * j], acc[mf][nf][2 * j + 1]};
                if (STATS) {
                    if (full_tile) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s1[nf][j] += v[j]; s2[nf][j] = __builtin_elementwise_fma(v[j], v[j], s2[nf][j]); }
                    } else {
                        const float mk = inb ? 1.f : 0.f;
                        const f32x2 mk2 = {mk, mk};
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const f32x2 t = v[j] * mk2; s1[nf][j] += t; s2[nf][j] = __builtin_elementwise_fma(t, v[j], s2[nf][j]); }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16x2 p2 = {(bf16_t)v[j][0], (bf16_t)v[j][1]};
                    pk[j] = __builtin_bit_cast(unsigned, p2);
                }
                if (inb && !(STAMPS && (a.dbg & 2))) {
                    *reinterpret_cast<u32x4*>(yp + nf * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    *reinterpret_cast<u32x4*>(yp + nf * 32 + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
        stamp(1);  // epilogue issue
        __syncthreads();  // X[t&1] is free for the producers, X[(t+1)&1] is ready
        stamp(2);  // barrier
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

// diagnostic entry (tools/phase_ws.py): stamps [wgs][16] u64 = consumer {mfma, epilogue, barrier, -, cycles, realtime} at +0,
// producer {load issue, transform + LDS write, barrier, -, cycles, realtime} at +8
extern "C" int wm_debug_conv3x3_ws64_phases(const void* x, const void* wp, const float* in_scale, const float* in_shift,
                                            void* y, float* stat, int B, int H, int W, unsigned long long* stamps, int dbg, void* stream) {
    WsArgs a;
    a.dbg = dbg;
    a.x = (const bf16_t*)x; a.ldx = 64; a.wp = (const bf16_t*)wp; a.bias = nullptr; a.nbias = 0; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (bf16_t*)y; a.stat = stat; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY;
    const int wgs = a.ntiles < 256 ? a.ntiles : 256;
    a.tiles_per_wg = wm_cdiv(a.ntiles, wgs); a.reverse = 0;
    const dim3 grid((unsigned)wm_cdiv(a.ntiles, a.tiles_per_wg)), block(512);
    if (in_scale && stat) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, true, true>), grid, block, 0, (hipStream_t)stream, a, stamps);
    else hipLaunchKernelGGL((conv3x3_ws64_kernel<false, false, true>), grid, block, 0, (hipStream_t)stream, a, stamps);
    return (int)grid.x;
}

static int g_ws_reverse = 0;
extern "C" void wm_debug_ws_direction(int reverse) { g_ws_reverse = reverse; }

// launcher used by conv3x3.hip
int wm_launch_conv3x3_ws64(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                           const float* in_shift, void* y, float* stat, int B, int H, int W, int wgs, int tiles_per_wg,
                           hipStream_t s) {
    WsArgs a;
    a.dbg = 0;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.wp = (const bf16_t*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (bf16_t*)y; a.stat = stat; a.B = B; a.H = H; a.W = W;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY; a.tiles_per_wg = tiles_per_wg;
    a.reverse = g_ws_reverse;
    const dim3 grid((unsigned)wgs), block(512);
    const bool xf = in_scale != nullptr, st = stat != nullptr;
    if (xf && st) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, true>), grid, block, 0, s, a);
    else if (xf) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, false>), grid, block, 0, s, a);
    else if (st) hipLaunchKernelGGL((conv3x3_ws64_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_ws64_kernel<false, false>), grid, block, 0, s, a);
    return WM_OK;
}
