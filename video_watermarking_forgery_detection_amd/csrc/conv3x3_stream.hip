// 3x3 convolution, bf16, for the layers the LDS-resident-filter kernel (conv3x3_ws.hip) cannot hold: Cin > 64 (UNet levels
// 3-5, the 112-channel encoder concat) -- any Cin % 16 == 0, CoutP % 64 == 0.
// Same wave-specialised scheme, but the FILTER is streamed with the input: the K loop runs over 32-channel chunks, and a
// chunk = the 18x18x32 halo slice of the input (20.7 KB) + the [9][64][32] filter slab (36.9 KB), double-buffered in LDS
// (115 KB).  4 producer waves fetch chunk c+2 into registers and publish chunk c+1 (fused BN+ReLU + zero padding of the
// input slice); 4 consumer waves run 9 taps x 16 v_mfma_f32_16x16x32_bf16 per chunk on [4 tile rows][4 channel fragments]
// accumulators that live across the whole K loop, then drain (BatchNorm sums, pack, 2 x 16-byte stores per pixel).
// A workgroup walks a run of (pixel tile, 64-channel output tile) items; the chunk stream continues across items, so the
// next item's first chunks load while the current one drains.  Filter slab traffic is served by L2 (every workgroup of an
// output tile reads the same slabs).
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

constexpr int TH = 16, TW = 16, HH = 18, HW = 18, NPIX = HH * HW;
constexpr int CK = 32;                       // channels per chunk
constexpr int NT = 64;                       // output channels per item
constexpr int XB = NPIX * CK * 2;            // 20,736 B
constexpr int WB = 9 * NT * CK * 2;          // 36,864 B
constexpr int XV = (NPIX * 4 + 255) / 256;   // halo vectors per producer thread (6)
constexpr int WV = 9 * NT * 4 / 256;         // filter vectors per producer thread (9)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

struct StArgs {
    const hx_t* x; int ldx;
    const hx_t* wp;            // [9][CoutP][Cin]
    const float* bias; int nbias;
    const float* in_scale; const float* in_shift;
    hx_t* y; int ldy;
    float* stat;                 // [4 * ntiles][2][CoutP] or null: one partial row per (pixel tile, consumer wave)
    int B, H, W, Cin, CoutP, tilesX, tilesY, ntiles, nct, nitems, items_per_wg;
};

// 64-byte rows (32 channels): the 16-byte slot is XOR-ed with (-(row >> 2)) & 3.  ds_read_b128 serves a wave in four groups
// of 16 lanes ({0-3,12-15,20-27}, ...): with lane = (row p, slot q) of a 16x16x32 fragment each group then touches all 16
// slots of a 256-byte bank row exactly once (filter fragments, and pixel fragments at tap shift 0; shifted pixel fragments
// see 2-way conflicts on a quarter of their lanes -- the LDS pipe is half idle in this kernel)
__device__ __forceinline__ int swz64(int row, int slot) { return slot ^ ((0 - (row >> 2)) & 3); }

template <bool XFORM, bool STATS>
__global__ __launch_bounds__(512, 2) void conv3x3_stream_kernel(StArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (XB + WB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int nch = (a.Cin + CK - 1) / CK;               // chunks per item
    // XCD-aware run assignment (see conv3x3_ws.hip); items are ordered (output tile, pixel tile): a run shares its filter slabs
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int i_begin = run * a.items_per_wg;
    const int i_end = min(a.nitems, i_begin + a.items_per_wg);
    const int nstream = (i_end - i_begin) * nch;         // chunks this workgroup streams
    struct Item { int b, ty0, tx0, n0; };
    auto item_of = [&](int it) {
        Item g;
        const int ct = it / a.ntiles;
        int t = it - ct * a.ntiles;
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW; g.n0 = ct * NT;
        return g;
    };

    if (producer) {
        // ================================================================== PRODUCER waves
        const int ptid = tid - 256;
        const int vec = ptid & 3;                        // 16-byte vector (8 channels) inside the chunk
        hx8 xA[XV], xB_[XV], wA[WV], wB_[WV];
        unsigned okA = 0, okB = 0;
        auto load_chunk = [&](int s, hx8 (&xd)[XV], hx8 (&wd)[WV], unsigned& okbits) {
            const int it = i_begin + s / nch, c = s - (s / nch) * nch;
            const Item g = item_of(it);
            const int cb = c * CK + vec * 8;
            const bool cok = cb < a.Cin;
            const int cl = cok ? cb : 0;
            okbits = 0;
            const hx_t* xb = a.x + (size_t)g.b * a.H * a.W * a.ldx + cl;
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                const int pix = min((ptid + 256 * k) >> 2, NPIX - 1);
                const int py = pix / HW, px = pix - py * HW;
                const int gy = g.ty0 - 1 + py, gx = g.tx0 - 1 + px;
                const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
                xd[k] = *reinterpret_cast<const hx8*>(xb + (size_t)(gyc * a.W + gxc) * a.ldx);
                okbits |= ((cok && gy == gyc && gx == gxc) ? 1u : 0u) << k;
            }
            const hx_t* wb = a.wp + (size_t)g.n0 * a.Cin + cl;
#pragma unroll
            for (int k = 0; k < WV; ++k) {
                const int row = (ptid + 256 * k) >> 2;   // tap * 64 + n
                const int tap = row >> 6, n = row & 63;
                wd[k] = *reinterpret_cast<const hx8*>(wb + ((size_t)tap * a.CoutP + n) * a.Cin);
            }
            if (cok) okbits |= 0x80000000u;              // the filter vectors of this thread are real channels
        };
        auto put_chunk = [&](int s, const hx8 (&xd)[XV], const hx8 (&wd)[WV], unsigned okbits) {
            unsigned char* bx = smem + (s & 1) * (XB + WB);
            unsigned char* bw = bx + XB;
            float sc[8], sh[8];
            if (XFORM) {
                const int c = s - (s / nch) * nch;
                const int cb = min(c * CK + vec * 8, a.Cin - 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale[cb + e]; sh[e] = a.in_shift[cb + e]; }
            }
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                const int pix = (ptid + 256 * k) >> 2;
                u32x4 w = __builtin_bit_cast(u32x4, xd[k]);
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
                const unsigned keep = 0u - ((okbits >> k) & 1u);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] &= keep;
                if (pix < NPIX) *reinterpret_cast<u32x4*>(bx + pix * 64 + swz64(pix % HW, vec) * 16) = w;
            }
            const unsigned keepw = 0u - (okbits >> 31);
#pragma unroll
            for (int k = 0; k < WV; ++k) {
                const int row = (ptid + 256 * k) >> 2;   // tap * 64 + n
                const int tap = row >> 6, n = row & 63;
                // accumulator-row permutation: channel n -> fragment (n>>2)&3, row 4*(n>>4) + (n&3): a lane owns 16 adjacent channels
                const int lrow = tap * 64 + ((n >> 2) & 3) * 16 + 4 * (n >> 4) + (n & 3);
                u32x4 w = __builtin_bit_cast(u32x4, wd[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] &= keepw;
                *reinterpret_cast<u32x4*>(bw + lrow * 64 + swz64(lrow, vec) * 16) = w;
            }
        };
        if (nstream > 0) load_chunk(0, xA, wA, okA);
        if (nstream > 1) load_chunk(1, xB_, wB_, okB);
        if (nstream > 0) put_chunk(0, xA, wA, okA);
        __syncthreads();
        // iteration s: the consumers compute chunk s; fetch chunk s+2, publish chunk s+1
        auto iter = [&](int s, hx8 (&xn)[XV], hx8 (&wn)[WV], unsigned& okn, const hx8 (&xc)[XV], const hx8 (&wc)[WV], unsigned okc) {
            if (s + 2 < nstream) load_chunk(s + 2, xn, wn, okn);
            if (s + 1 < nstream) put_chunk(s + 1, xc, wc, okc);
            __syncthreads();
        };
        for (int s = 0; s < nstream; s += 2) {
            iter(s, xA, wA, okA, xB_, wB_, okB);
            if (s + 1 < nstream) iter(s + 1, xB_, wB_, okB, xA, wA, okA);
        }
        return;
    }

    // ====================================================================== CONSUMER waves: tile rows 4w .. 4w+3, all 64 channels
    const int p = lane & 15, q = lane >> 4;
    float s1[16], s2[16];
    f32x4 acc[4][4];
    int aoff[3], boff;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) aoff[kw] = (wave * 4 * HW + p + kw) * 64 + swz64(p + kw, q) * 16;
    boff = p * 64 + swz64(p, q) * 16;
    __syncthreads();   // first chunk visible
    int s = 0;
    for (int it = i_begin; it < i_end; ++it) {
        const Item g = item_of(it);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c, ++s) {
            const char* bx = reinterpret_cast<const char*>(smem + (s & 1) * (XB + WB));
            const char* bw = bx + XB;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                hx8 pix[4], fil[4];
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) pix[mf] = *reinterpret_cast<const hx8*>(bx + aoff[kw] + (mf + kh) * (HW * 64));
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) fil[nf] = *reinterpret_cast<const hx8*>(bw + boff + (tap * 64 + nf * 16) * 64);
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
                        acc[mf][nf] = HX::mfma16(fil[nf], pix[mf], acc[mf][nf]);
            }
            __syncthreads();   // chunk s consumed, chunk s+1 published
        }
        // ---- drain: lane (p, q) holds channels n0 + 16q .. +15 of pixel (row 4w + mf, column p)
        if (STATS) {
#pragma unroll
            for (int cidx = 0; cidx < 16; ++cidx) { s1[cidx] = 0.f; s2[cidx] = 0.f; }
        }
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) bv[j] = (a.bias && g.n0 + 16 * q + j < a.nbias) ? a.bias[g.n0 + 16 * q + j] : 0.f;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int gy = g.ty0 + wave * 4 + mf, gx = g.tx0 + p;
            const bool inb = gy < a.H && gx < a.W;
            const float mk = inb ? 1.f : 0.f;
            unsigned pk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nf = j >> 1, i0 = 2 * (j & 1);
                const float v0 = acc[mf][nf][i0] + bv[2 * j], v1 = acc[mf][nf][i0 + 1] + bv[2 * j + 1];
                if (STATS) {
                    const float t0 = v0 * mk, t1 = v1 * mk;
                    s1[2 * j] += t0; s1[2 * j + 1] += t1;
                    s2[2 * j] = __builtin_fmaf(t0, v0, s2[2 * j]);
                    s2[2 * j + 1] = __builtin_fmaf(t1, v1, s2[2 * j + 1]);
                }
                const hx2 p2 = HX::pack2(v0, v1);
                pk[j] = __builtin_bit_cast(unsigned, p2);
            }
            if (inb) {
                hx_t* yp = a.y + (((size_t)g.b * a.H + gy) * a.W + gx) * a.ldy + g.n0 + 16 * q;
                *reinterpret_cast<u32x4*>(yp) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(yp + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
        }
        if (STATS) {
            // one partial row per (pixel tile, consumer wave): [4*tile + wave][2][CoutP], columns n0 .. n0+63 -- no rendezvous
            const int tile = it - (it / a.ntiles) * a.ntiles;
            float* row = a.stat + ((size_t)(4 * tile + wave) * 2) * a.CoutP + g.n0 + 16 * q;
#pragma unroll
            for (int cidx = 0; cidx < 16; ++cidx) {
                float u1 = s1[cidx], u2 = s2[cidx];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
                if (p == 0) { row[cidx] = u1; row[a.CoutP + cidx] = u2; }
            }
        }
    }
}

}  // namespace

// launcher used by conv3x3.hip
#ifndef WM_H16_F16
int wm_conv3x3_stream_supported(int Cin, int CoutP) { return (Cin % 16 == 0 && Cin >= 32 && CoutP % 64 == 0) ? 1 : 0; }
int wm_conv3x3_stream_nparts(int B, int H, int W) { return 4 * B * wm_cdiv(H, TH) * wm_cdiv(W, TW); }
#endif

int WM_HSYM(wm_launch_conv3x3_stream)(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                             const float* in_shift, void* y, int ldy, float* stat, int B, int H, int W, int Cin, int CoutP,
                             hipStream_t s) {
    StArgs a;
    a.x = (const hx_t*)x; a.ldx = ldx; a.wp = (const hx_t*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (hx_t*)y; a.ldy = ldy; a.stat = stat; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.CoutP = CoutP;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY;
    a.nct = CoutP / NT; a.nitems = a.ntiles * a.nct;
    const int wgs_max = 256;
    a.items_per_wg = wm_cdiv(a.nitems, wgs_max);
    const dim3 grid((unsigned)wm_cdiv(a.nitems, a.items_per_wg)), block(512);
    const bool xf = in_scale != nullptr, st = stat != nullptr;
    if (xf && st) hipLaunchKernelGGL((conv3x3_stream_kernel<true, true>), grid, block, 0, s, a);
    else if (xf) hipLaunchKernelGGL((conv3x3_stream_kernel<true, false>), grid, block, 0, s, a);
    else if (st) hipLaunchKernelGGL((conv3x3_stream_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_stream_kernel<false, false>), grid, block, 0, s, a);
    return WM_OK;
}
