// Fused backward of a 64 -> 64 ConvBNRelu body layer (gfx950, bf16 / f16): ONE persistent kernel computes, per 8x16-pixel tile,
//   dy   = BatchNorm-backward apply of (g, y) of layer L                      (formed while staging, never written to memory)
//   dx   = conv3x3(dy, W^T)                     the input gradient            -> written (gradient wrt layer L-1's ReLU output)
//   sums = layer L-1's BatchNorm-backward partial sums of dx                  -> partial rows (as conv3x3_ws.hip's BWDST)
//   dW  += sum_pixels dy (x) a,  a = ReLU(BN(y_{L-1}))     the weight gradient -> one slab per workgroup (as wgrad_ws.hip)
// from ONE staged dy halo tile and ONE staged tile of y_{L-1}.  The two-kernel form (conv3x3_ws.hip BNBWD = 2 + wgrad_ws.hip) moves
// 671 + 268 MB per layer at B = 16, 256x256 -- it writes dy (134 MB) only for the weight gradient to read it back together with y_{L-1}
// (which the input gradient's epilogue had read already); this kernel moves 537 MB: g, y, y_{L-1} read once, dx written once.
//
// Why not a mode of the wave-specialised kernels: the weight gradient keeps 9 x 64 x 64 f32 accumulators per workgroup (144 registers per
// lane over 4 waves) across the whole run of tiles; next to the input gradient's accumulators and epilogue state that does not fit the
// 256 registers of a two-waves-per-SIMD kernel.  Here a workgroup is FOUR waves, one per SIMD, with the whole 512-register budget each;
// every wave stages 1/4 of the next tile (global loads issued before the MFMA phases, transform + LDS writes after them), computes two
// tile rows of the input gradient and a 32 x 32 block x 9 taps of the weight gradient.  LDS: filter 73,728 B + 2 x (dy halo 10x18 px
// 23,040 B + a tile 8x16 px 16,384 B) = 152,576 B + 2 KB of sums: the 8x16 tile is what lets both operand tiles be double-buffered.
//
// LDS layouts: 128-byte pixel rows.  The dy halo is read BOTH by the input gradient's ds_read_b128 (16 consecutive pixels, one 16-byte
// slot each) and by the weight gradient's transposing ds_read_b64_tr_b16 (pixels {c..c+3, c+8..c+11}, 32 bytes each); the 16-byte slot
// index is XORed with fsw(px) = bit2(px) | bit1(px) << 1 | bit3(px) << 2 -- a bit permutation of (px >> 1) & 7, so 16 consecutive
// pixels hit 16 distinct slots, and its upper two bits are wgrad_ws.hip's swz16, so the transposing reads fill a bank row exactly once.
#include <type_traits>
#include "wm_common.h"

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

constexpr int TH = 8, TW = 16, HH = 10, HW = 18, NPX = HH * HW, C = 64;
constexpr int SW_BYTES = 9 * C * C * 2, SDY_BYTES = NPX * 128, SA_BYTES = TH * TW * 128, BUF_BYTES = SDY_BYTES + SA_BYTES;
constexpr int XV = (NPX * 8 + 255) / 256;        // dy halo vectors per thread (6; the last one partially live)
constexpr int AV = TH * TW * 8 / 256;            // a-tile vectors per thread (4)
constexpr int GV_MAXB = 24;                      // GVEC: samples whose k3g rows fit the LDS left over (24 x 256 B)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct BwdArgs {
    const hx_t* g; const hx_t* y;                          // layer L: gradient wrt its ReLU output, its raw conv output [B,H,W,64]
    const float* gvec; int gv_ld;                          // GVEC: the gradient is one row per sample [B][gv_ld] (a globally pooled layer); g unused
    const float* stats4; int st_ld; const float* coef;     // [scale | shift | mean | invstd][st_ld], wm_bn_bwd_finalize's coef [3][st_ld]
    const hx_t* wpt;                                       // [9][64][64] filter packed for the input gradient (rows = input channels)
    const hx_t* xr; const float* in_scale; const float* in_shift;   // layer L-1: raw conv output, its BatchNorm scale / shift
    hx_t* dx;                                              // [B,H,W,64]
    float* stat;                                           // [gridDim.x][2][64]
    float* ws;                                             // [gridDim.x][9][64][64]
    int B, H, W, tilesX, tilesY, ntiles, reverse;
    unsigned mX, mY, m2X;                                  // ceil(2^32 / d) of tilesX, tilesY, 2 tilesX (0 for d = 1): tile index / d = mulhi
};

__device__ __forceinline__ int fsw(int px) { return ((px >> 2) & 1) | (((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2); }
__device__ __forceinline__ int swz16(int col) { return (((col >> 1) & 1) << 5) | (((col >> 3) & 1) << 6); }
__device__ __forceinline__ int swzw(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ hx8 tr_frag(const char* p0, const char* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(hx8, v);
}

// DBG (debug build only, tools/ab_step.py wm_debug_bwd_variant): phase ablations -- 1 skip the input-gradient MFMAs, 2 the weight-gradient
// MFMAs, 4 the epilogue, 8 the staging of the next tile, 16 stage the same tile again and again (results are then meaningless; compile-
// time so the real kernel is untouched), 512 no dx stores, 1024 / 2048 half the filter / dy fragment reads of the input- / weight-gradient loop,
// 4096 phase stamps (with 512: the sums land where dx would)
// PREMASKED: g arrives already multiplied by its layer's ReLU mask (this kernel's own dx is written that way, see the epilogue), so the
// staging's compare + select + the z fma disappear; masking twice is the identity, so results do not depend on the flag
// GVEC: the layer's output was globally pooled, so its gradient is one value per (sample, channel): the staging reads y only and takes
// k3 + ca * gvec[b][c] (wm_bn_fold_g) from an LDS table [B][64] built at the start (bit-identical to conv3x3_ws.hip's BNBWD = 1 form)
// ALIGNED: H % 8 == 0 and W % 16 == 0 (every tile is whole).  The staged operands are then fetched with BUFFER loads: a slot's byte offset
// is a per-thread constant (its halo pixel, relative to the tile's halo origin) plus one wave-uniform tile base -- one v_add per slot instead
// of ~24 instructions of clamped addressing per load pair (one wave per SIMD pays an issue slot for every one of them) --, a halo pixel
// left or right of the image reads its neighbour in memory (a valid address; the value is zeroed when it is published), one above the first
// or below the last image falls outside the buffer descriptor's range and reads zeros; which slots lie outside the image is a wave-uniform
// choice among four per-thread bit masks (top / bottom / left / right edge), and the a tile needs no masking at all.
template <int DBG, bool PREMASKED, bool GVEC = false, bool ALIGNED = false>
__global__ __launch_bounds__(256, 1) void bwd_ws_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * BUF_BYTES + 4 * 2 * C * 4 + 2 * C * 4 + (C * 8 + 32) * 4 + (GVEC ? GV_MAXB * C * 4 : 0)];
    hx_t* sW = reinterpret_cast<hx_t*>(smem);
    unsigned char* sBuf = smem + SW_BYTES;
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + 2 * BUF_BYTES);
    float* sTab = sRed + 4 * 2 * C;   // in_scale | in_shift of the feeding layer (the epilogue's mask)
    float* sG = sTab + 2 * C + C * 8 + 32;   // GVEC: [B][64] k3 + ca * gvec[b][c]
    float* sK = sTab + 2 * C;         // per channel: scale, shift, ca, k2, k3 (wm_bn_fold) of layer L; in_scale, in_shift of layer L-1; 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < C) {
        sTab[tid] = a.in_scale[tid]; sTab[C + tid] = a.in_shift[tid];
        float k2, k3;
        wm_bn_fold(a.stats4[2 * a.st_ld + tid], a.stats4[3 * a.st_ld + tid], a.coef[tid], a.coef[a.st_ld + tid], a.coef[2 * a.st_ld + tid], k2, k3);
        const float v[8] = {a.stats4[tid], a.stats4[a.st_ld + tid], a.coef[tid], k2, k3, a.in_scale[tid], a.in_shift[tid], 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) sK[tid * 8 + (tid >> 3) * 4 + i] = v[i];   // each 8-channel block shifted by 16 B: the 8 blocks a wave reads together sit in 8 different bank groups
    }

    // ---- filter -> LDS, laid out for the 16x16x32 consumers of conv3x3_ws.hip (a lane ends with 16 adjacent channels of its pixel)
    {
        // (in two batches of 9 loads in flight: a `load, wait, store` loop runs its 18 trips one L2 round trip after the other)
        constexpr int NV = 9 * C * 8;
        static_assert(NV == 18 * 256, "18 vectors per thread");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            hx8 wv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = tid + 256 * (9 * half + k);
                wv[k] = *reinterpret_cast<const hx8*>(a.wpt + (size_t)(i >> 3) * C + (i & 7) * 8);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = tid + 256 * (9 * half + k);
                const int row = i >> 3, tap = row / C, n = row % C;
                const int lrow = tap * C + ((n >> 2) & 3) * 16 + 4 * (n >> 4) + (n & 3);
                *reinterpret_cast<hx8*>(sW + lrow * C + swzw(lrow, i & 7) * 8) = wv[k];
            }
        }
    }
    // ---- run of tiles (XCD-aware: workgroups b and b+8 share an L2, give each XCD consecutive runs)
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int t_begin = (int)(((long)run * a.ntiles) / G), t_end = (int)(((long)(run + 1) * a.ntiles) / G);
    struct TileGeo { int b, ty0, tx0; };
    // tile order: pairs of tile rows walked column by column (upper tile, lower tile, next column ...), so the two halo rows a tile
    // shares with the one below it and the two halo columns it shares with its right neighbour are re-read within 1 and 2 tiles -- from
    // this XCD's L2, not from the Infinity Cache 16 tiles later (tilesY odd: plain row-major order)
    // (the divisions are mulhi by the host's ceil(2^32 / d): exact while tile * d < 2^32, and 3 scalar instructions instead of ~40 --
    // one wave per SIMD pays the issue slot of every instruction, scalar ones included)
    auto fdiv = [](int t, int d, unsigned m) { return d == 1 ? t : (int)__umulhi((unsigned)t, m); };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;
        if (a.tilesY & 1) {
            const int q1 = fdiv(t, a.tilesX, a.mX), txi = t - q1 * a.tilesX;
            const int q2 = fdiv(q1, a.tilesY, a.mY), tyi = q1 - q2 * a.tilesY;
            g.b = q2; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        } else {
            const int pr = fdiv(t, 2 * a.tilesX, a.m2X), rem = t - pr * 2 * a.tilesX;     // pr: pair of tile rows, over all images
            const int row = 2 * pr + (rem & 1);
            g.b = fdiv(row, a.tilesY, a.mY); g.ty0 = (row - g.b * a.tilesY) * TH; g.tx0 = (rem >> 1) * TW;
        }
        return g;
    };

    // ================================================================== staging role (every thread: 8 channels `vec` of its pixels)
    const int vec = tid & 7, slot = tid >> 3;
    int hlds[XV];
    const int alds0 = SDY_BYTES + slot * 128 + ((vec << 4) ^ swz16(slot & 15));   // a-tile pixel slot + 32k: same column, row + 2k
#pragma unroll
    for (int k = 0; k < XV; ++k) {
        const int hp = min(slot + 32 * k, NPX - 1);
        const int py = hp / HW, px = hp - py * HW;
        hlds[k] = hp * 128 + ((vec ^ fsw(px)) << 4);
    }
    const bool last_live = slot + 32 * (XV - 1) < NPX;
    hx8 dG[XV], dY[XV], dA[AV];
    unsigned okh = 0, oka = 0;
    // ALIGNED: byte offsets of this thread's slots relative to the tile's halo origin (halo) / first pixel (a tile); edge masks: bit
    // k + 6 e of `edge` = slot k lies in the halo's top (e = 0) / bottom (1) row, left (2) / right (3) column
    unsigned hofs[XV], aofs[AV], edge = 0;
    __amdgpu_buffer_rsrc_t rsG, rsY, rsX, rsD;
    unsigned eofs = 0;   // this lane's epilogue pixel (tile row 2 wave, column lane & 15), channels 16 (lane >> 4): byte offset from the tile's first pixel
    if constexpr (ALIGNED) {
        eofs = (unsigned)(((wave * 2 * a.W + (lane & 15)) * C + 16 * (lane >> 4)) * 2);
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int hp = min(slot + 32 * k, NPX - 1), py = hp / HW, px = hp - py * HW;
            hofs[k] = (unsigned)(((py * a.W + px) * C + vec * 8) * 2);
            edge |= (py == 0 ? 1u : 0u) << k | (py == HH - 1 ? 1u : 0u) << (k + 6) | (px == 0 ? 1u : 0u) << (k + 12) | (px == HW - 1 ? 1u : 0u) << (k + 18);
        }
#pragma unroll
        for (int k = 0; k < AV; ++k) {
            const int ip = slot + 32 * k;
            aofs[k] = (unsigned)((((ip >> 4) * a.W + (ip & 15)) * C + vec * 8) * 2);
        }
        const unsigned nbytes = (unsigned)a.B * (unsigned)a.H * (unsigned)a.W * (unsigned)(C * 2);
        rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(GVEC ? a.y : a.g), 0, nbytes, 0x00020000);
        rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.y), 0, nbytes, 0x00020000);
        rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.xr), 0, nbytes, 0x00020000);
        rsD = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, nbytes, 0x00020000);
    }
    // wave-uniform byte offset of the tile's halo origin (pixel (ty0 - 1, tx0 - 1); "negative" = wraps beyond the descriptor's range)
    auto halo_base = [&](const TileGeo& t) { return ((unsigned)(t.b * a.H + t.ty0) * (unsigned)a.W + (unsigned)t.tx0) * (unsigned)(C * 2) - (unsigned)((a.W + 1) * C * 2); };
    auto inside_bits = [&](const TileGeo& t) {   // ALIGNED: bit k = slot k's halo pixel of tile t lies inside the image
        const unsigned sel = (t.ty0 == 0 ? 0x3fu : 0u) | (t.ty0 + TH == a.H ? 0x3fu << 6 : 0u) | (t.tx0 == 0 ? 0x3fu << 12 : 0u) | (t.tx0 + TW == a.W ? 0x3fu << 18 : 0u);
        const unsigned e = edge & sel;
        return ~(e | (e >> 6) | (e >> 12) | (e >> 18)) & 0x3fu;
    };
    // Staged operands travel in registers for a whole tile: slot k (a 16-byte vector of g + y, or of the feeding layer's y) is requested
    // during tile t-1 right after the slot's previous content was published, and is transformed + written to the LDS during tile t
    // (for tile t+1): one register set, a prefetch distance of one tile, so the wave never waits on memory it has just asked for.
    // branch-free addressing (a branch would split the tile body's one basic block): clamp with v_med3, 24-bit multiplies, 32-bit
    // element offsets (the host checks B*H*W*64 < 2^31)
    auto load_dy_slot = [&](const TileGeo& t, int k) {
        if constexpr (ALIGNED) {   // (okh: inside_bits(t), computed once per tile by the caller)
            const unsigned o = halo_base(t) + hofs[k];
            if constexpr (!GVEC) dG[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsG, o, 0, 0));
            dY[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsY, o, 0, 0));
            return;
        }
        const int hp = min(slot + 32 * k, NPX - 1), py = (hp * 3641) >> 16, px = hp - py * HW;   // / 18 for hp < 200
        const int gy = t.ty0 - 1 + py, gx = t.tx0 - 1 + px;
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        const unsigned o = ((unsigned)__mul24(t.b * a.H + gyc, a.W) + (unsigned)gxc) * C + vec * 8;
        if constexpr (!GVEC) dG[k] = *reinterpret_cast<const hx8*>(a.g + o);
        dY[k] = *reinterpret_cast<const hx8*>(a.y + o);
        okh = (okh & ~(1u << k)) | (((gy == gyc && gx == gxc) ? 1u : 0u) << k);
    };
    auto load_a_slot = [&](const TileGeo& t, int k) {
        if constexpr (ALIGNED) {   // every pixel of a whole tile lies inside the image
            dA[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, halo_base(t) + (unsigned)((a.W + 1) * C * 2) + aofs[k], 0, 0));
            return;
        }
        const int ip = slot + 32 * k;
        const int gy = t.ty0 + (ip >> 4), gx = t.tx0 + (ip & 15);
        const int gyc = min(gy, a.H - 1), gxc = min(gx, a.W - 1);
        const unsigned o = ((unsigned)__mul24(t.b * a.H + gyc, a.W) + (unsigned)gxc) * C + vec * 8;
        dA[k] = *reinterpret_cast<const hx8*>(a.xr + o);
        oka = (oka & ~(1u << k)) | (((gy == gyc && gx == gxc) ? 1u : 0u) << k);
    };
    auto load_halo = [&](const TileGeo& t) {
#pragma unroll
        for (int k = 0; k < XV; ++k) load_dy_slot(t, k);
        if constexpr (ALIGNED) okh = inside_bits(t);
    };
    auto load_atile = [&](const TileGeo& t) {
#pragma unroll
        for (int k = 0; k < AV; ++k) load_a_slot(t, k);
    };
    // transform + LDS writes, one channel pair at a time (its constants come from the LDS table: a dozen registers live, not 56)
    auto publish_tile = [&](unsigned char* buf, int bsample) {
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const float* kp = sK + (vec * 8 + 2 * pq) * 8 + vec * 4;
            f32x2 kg = {0.f, 0.f};
            if constexpr (GVEC) kg = *reinterpret_cast<const f32x2*>(sG + bsample * C + vec * 8 + 2 * pq);
            const f32x4 ka = *reinterpret_cast<const f32x4*>(kp), ka2 = *reinterpret_cast<const f32x4*>(kp + 4);
            const f32x4 kb = *reinterpret_cast<const f32x4*>(kp + 8), kb2 = *reinterpret_cast<const f32x4*>(kp + 12);
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
                const u32x4 wy = __builtin_bit_cast(u32x4, dY[k]);
                float d0, d1;
                if constexpr (GVEC) {
                    d0 = wm_bn_fold_dy(HX::lo(wy[pq]), ka[0], ka[1], ka[3], ka2[0], kg[0]);
                    d1 = wm_bn_fold_dy(HX::hi(wy[pq]), kb[0], kb[1], kb[3], kb2[0], kg[1]);
                } else if constexpr (PREMASKED) {
                    d0 = __builtin_fmaf(ka[2], HX::lo(w[pq]), __builtin_fmaf(-ka[3], HX::lo(wy[pq]), ka2[0]));
                    d1 = __builtin_fmaf(kb[2], HX::hi(w[pq]), __builtin_fmaf(-kb[3], HX::hi(wy[pq]), kb2[0]));
                } else {
                    d0 = wm_bn_fold_dyg(HX::lo(wy[pq]), HX::lo(w[pq]), ka[0], ka[1], ka[2], ka[3], ka2[0]);
                    d1 = wm_bn_fold_dyg(HX::hi(wy[pq]), HX::hi(w[pq]), kb[0], kb[1], kb[2], kb[3], kb2[0]);
                }
                const hx2 pk = HX::pack2(d0, d1);
                w[pq] = __builtin_bit_cast(unsigned, pk);
                if constexpr (GVEC) dY[k] = __builtin_bit_cast(hx8, w); else dG[k] = __builtin_bit_cast(hx8, w);
            }
#pragma unroll
            for (int k = 0; k < AV; ++k) {
                u32x4 w = __builtin_bit_cast(u32x4, dA[k]);
                const float f0 = __builtin_fmaf(HX::lo(w[pq]), ka2[1], ka2[2]);
                const float f1 = __builtin_fmaf(HX::hi(w[pq]), kb2[1], kb2[2]);
                const hx2 pk = HX::pack2(f0, f1);
                const i16x2 z = {0, 0};
                w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                dA[k] = __builtin_bit_cast(hx8, w);
            }
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
            const unsigned keep = 0u - ((okh >> k) & 1u);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] &= keep;
            if (k + 1 < XV || last_live) *reinterpret_cast<u32x4*>(buf + hlds[k]) = w;
        }
#pragma unroll
        for (int k = 0; k < AV; ++k) {
            u32x4 w = __builtin_bit_cast(u32x4, dA[k]);
            if constexpr (!ALIGNED) {
                const unsigned keep = 0u - ((oka >> k) & 1u);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] &= keep;
            }
            *reinterpret_cast<u32x4*>(buf + (alds0 + k * 32 * 128)) = w;
        }
    };

    // ================================================================== input-gradient role: tile rows 2*wave, 2*wave + 1
    // lane (p, q): pixel column p; accumulator [ml][nf] register i = channel 16q + 4nf + i
    const int p = lane & 15, q = lane >> 4;
    int aoff[3][2], boff[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[kw][ks] = ((wave * 2 * HW + p + kw) * 128) + (((ks * 4 + q) ^ fsw(p + kw)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (p * C + swzw(p, ks * 4 + q) * 8) * 2;
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }

    // ================================================================== weight-gradient role: wave (mi, ni) owns the 32 co x 32 ci block
    // D[co 16 x ci 16] += A[co x 32 pixels] (dy halo, shifted by the tap) * B[32 pixels x ci] (a tile); a K-step is two tile rows
    const int mi = wave >> 1, ni = wave & 1;
    const int r = lane & 15, kq = lane >> 4, q2 = (lane >> 2) & 3, p2 = lane & 3;
    const int colb = 8 * (kq & 1) + q2;
    int xo[3][2][2], dof[2][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {
                const int col = colb + kw + 4 * sx;
                const int sl = mi * 4 + fi * 2 + (p2 >> 1);
                xo[kw][sx][fi] = ((kq >> 1) * HW + col) * 128 + ((sl ^ fsw(col)) << 4) + (p2 & 1) * 8;
            }
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
            const int col = colb + 4 * sx;
            dof[sx][fj] = SDY_BYTES + ((kq >> 1) * TW + col) * 128 + (((ni * 32 + fj * 16 + 4 * p2) * 2) ^ swz16(col));
        }
    f32x4 wacc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) wacc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();   // the constant table
    if constexpr (GVEC) {
        for (int i = tid; i < a.B * C; i += 256) {
            const int b = i / C, c = i - b * C;
            const float* kc = sK + c * 8 + (c >> 3) * 4;
            sG[i] = wm_bn_fold_g(kc[2], a.gvec[(size_t)b * a.gv_ld + c], kc[4]);
        }
        __syncthreads();
    }
    if (t_begin < t_end) {
        load_halo(geo(t_begin));
        load_atile(geo(t_begin));
        publish_tile(sBuf, geo(t_begin).b);
    }
    __syncthreads();   // filter + first tile visible

    // staging micro-steps (50 per tile), issued between the MFMAs of the input-gradient loop.  Two groups of 5 slots (3 dy-halo vectors +
    // 2 a-tile vectors each); per group: 4 channel pairs x 5 transform units (the pair's constants are fetched from the LDS table at the
    // first), then 5 publish units: zero padding, the LDS write, and the request of the slot's content two tiles ahead.
    f32x4 pka, pka2, pkb, pkb2;
    f32x2 pkg = {0.f, 0.f};   // GVEC: k3 + ca * gvec of the channel pair, for the sample of the tile being published
    int bpub = 0;
    auto pub_consts = [&](int pq) {
        if constexpr (GVEC) pkg = *reinterpret_cast<const f32x2*>(sG + bpub * C + vec * 8 + 2 * pq);
        const float* kp = sK + (vec * 8 + 2 * pq) * 8 + vec * 4;
        pka = *reinterpret_cast<const f32x4*>(kp); pka2 = *reinterpret_cast<const f32x4*>(kp + 4);
        pkb = *reinterpret_cast<const f32x4*>(kp + 8); pkb2 = *reinterpret_cast<const f32x4*>(kp + 12);
    };
    // half-step h of the input-gradient loop (36 of them, 4 MFMAs each) carries the units [hs_first(h), hs_first(h) + hs_count(h)): a channel
    // pair owns 4 half-steps -- its constants are requested at the top of the first (no unit there: the read has 4 MFMAs to arrive), its 5
    // transform units run 2-2-1 in the other three --, a group's 5 publish units 3-2 in the last two.  ONE wave per SIMD issues in order:
    // a unit's VALU instructions hide behind the MFMAs only when they sit BETWEEN them (the matrix pipe is busy 16 cycles per MFMA, the
    // issue port 4), hence the sched_group_barrier pipelines in the loop (tools/phase_bwd.py: 6.8k -> cycles per tile for this loop)
    auto hs_first = [](int h) { const int g = h / 18, r = h - 18 * g, m = r & 3; return 25 * g + (r < 16 ? (r / 4) * 5 + (m == 0 ? 0 : m == 1 ? 0 : m == 2 ? 2 : 4) : 20 + (r == 16 ? 0 : 3)); };
    auto hs_count = [](int h) { const int r = h % 18, m = r & 3; return r < 16 ? (m == 0 ? 0 : m == 3 ? 1 : 2) : (r == 16 ? 3 : 2); };
    auto pub_unit = [&](int u, unsigned char* buf, bool refill, const TileGeo& g2, unsigned okn) {
        const int grp = u / 25, v = u - grp * 25;
        const int j = v < 20 ? v % 5 : v - 20;
        const bool is_dy = j < 3;
        const int k = is_dy ? grp * 3 + j : grp * 2 + (j - 3);
        if (v < 20) {
            if constexpr ((DBG & 32) != 0) return;   // DBG 32: no transform arithmetic (raw operands are published)
            const int pq = v / 5;
            if (is_dy) {
                u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
                const u32x4 wy = __builtin_bit_cast(u32x4, dY[k]);
                float d0, d1;
                if constexpr (GVEC) {
                    d0 = wm_bn_fold_dy(HX::lo(wy[pq]), pka[0], pka[1], pka[3], pka2[0], pkg[0]);
                    d1 = wm_bn_fold_dy(HX::hi(wy[pq]), pkb[0], pkb[1], pkb[3], pkb2[0], pkg[1]);
                } else if constexpr (PREMASKED || (DBG & 256) != 0) {   // (DBG 256: timing probe of the same on unmasked data)
                    d0 = __builtin_fmaf(pka[2], HX::lo(w[pq]), __builtin_fmaf(-pka[3], HX::lo(wy[pq]), pka2[0]));
                    d1 = __builtin_fmaf(pkb[2], HX::hi(w[pq]), __builtin_fmaf(-pkb[3], HX::hi(wy[pq]), pkb2[0]));
                } else {
                    d0 = wm_bn_fold_dyg(HX::lo(wy[pq]), HX::lo(w[pq]), pka[0], pka[1], pka[2], pka[3], pka2[0]);
                    d1 = wm_bn_fold_dyg(HX::hi(wy[pq]), HX::hi(w[pq]), pkb[0], pkb[1], pkb[2], pkb[3], pkb2[0]);
                }
                const hx2 pk = HX::pack2(d0, d1);
                w[pq] = __builtin_bit_cast(unsigned, pk);
                if constexpr (GVEC) dY[k] = __builtin_bit_cast(hx8, w); else dG[k] = __builtin_bit_cast(hx8, w);
            } else {
                u32x4 w = __builtin_bit_cast(u32x4, dA[k]);
                const float f0 = __builtin_fmaf(HX::lo(w[pq]), pka2[1], pka2[2]);
                const float f1 = __builtin_fmaf(HX::hi(w[pq]), pkb2[1], pkb2[2]);
                const hx2 pk = HX::pack2(f0, f1);
                const i16x2 z = {0, 0};
                w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
                dA[k] = __builtin_bit_cast(hx8, w);
            }
        } else if (is_dy) {
            u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
            const unsigned keep = 0u - ((okh >> k) & 1u);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) w[q4] &= keep;
            if (k + 1 < XV || last_live) *reinterpret_cast<u32x4*>(buf + hlds[k]) = w;
            if (refill && !(DBG & 64)) {   // DBG 64: no refill loads
                load_dy_slot(g2, k);
                if constexpr (ALIGNED) okh = (okh & ~(1u << k)) | (okn & (1u << k));   // the slot now holds tile + 2's pixel (one v_bfi)
            }
        } else {
            u32x4 w = __builtin_bit_cast(u32x4, dA[k]);
            if constexpr (!ALIGNED) {
                const unsigned keep = 0u - ((oka >> k) & 1u);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) w[q4] &= keep;
            }
            *reinterpret_cast<u32x4*>(buf + (alds0 + k * 32 * 128)) = w;
            if (refill && !(DBG & 64)) load_a_slot(g2, k);
        }
    };
    constexpr int NUNIT = 50;
    static_assert(XV == 6 && AV == 4, "the staging schedule is written for 6 + 4 slots");
    // the second tile of the run is requested before the loop; from then on every slot is re-requested as soon as it has been published
    if (t_begin + 1 < t_end && !(DBG & 8)) { const TileGeo g1 = geo(t_begin + 1); load_halo(g1); load_atile(g1); }

    // one tile; STAGE: tile + 1 exists (its operands are in the registers: publish them); REFILL: tile + 2 exists (request it).  Compile-
    // time, so the body is straight-line code the scheduler can interleave; the last two tiles of a run use the reduced bodies
    // DBG 4096: s_memtime stamps at the phase boundaries, summed per workgroup and written over the start of dx (tools/phase_bwd.py)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t_run0 = 0, r_run0 = 0;
    if constexpr ((DBG & 4096) != 0) { t_run0 = __builtin_amdgcn_s_memtime(); r_run0 = __builtin_amdgcn_s_memrealtime(); }
    auto tile_body = [&](int tile, auto stage_c, auto refill_c) {
        unsigned long long ts[7];
        if constexpr ((DBG & 4096) != 0) ts[0] = __builtin_amdgcn_s_memtime();
        constexpr bool stage = decltype(stage_c)::value && !(DBG & 8);
        constexpr bool refill = decltype(refill_c)::value && !(DBG & 8);
        const unsigned char* cur = sBuf + ((tile - t_begin) & 1) * BUF_BYTES;
        unsigned char* nxt = sBuf + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
        // the tile's scalar bookkeeping and the request of its epilogue operand (the feeding layer's y at this lane's two output pixels:
        // requested early, used last) run AFTER the weight-gradient loop's first fragment reads have been issued, in the shadow of their
        // LDS latency (one wave per SIMD: nobody else hides it)
        TileGeo g, g2;
        unsigned okn = 0;
        unsigned ryv[2][8];
        bool inb[2] = {true, true};
        hx_t* outp[2] = {nullptr, nullptr};
        unsigned eo[2] = {0, 0};   // ALIGNED: byte offset of this lane's 16 channels of output pixel (row 2 wave + ml, column p) -- for xr and for dx
        auto head_work = [&]() __attribute__((always_inline)) {
            g = geo(tile);
            if constexpr (GVEC && decltype(stage_c)::value) bpub = geo(tile + 1).b;
            g2 = (DBG & 16) ? geo(t_begin) : geo(refill ? tile + 2 : tile);   // DBG 16: every refill re-reads the run's first tile (L2 hits)
            if constexpr (ALIGNED && refill) okn = inside_bits(g2);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) {
                u32x4 t0, t1;
                if constexpr (ALIGNED) {
                    eo[ml] = halo_base(g) + (unsigned)((a.W + 1) * C * 2) + eofs + (unsigned)(ml * a.W * C * 2);
                    t0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, eo[ml], 0, 0);
                    t1 = __builtin_amdgcn_raw_buffer_load_b128(rsX, eo[ml] + 16u, 0, 0);
                } else {
                    const int gy = g.ty0 + wave * 2 + ml, gx = g.tx0 + p;
                    inb[ml] = gy < a.H && gx < a.W;
                    const size_t o = inb[ml] ? (((size_t)g.b * a.H + gy) * a.W + gx) * C + 16 * q : (size_t)(16 * q);
                    outp[ml] = a.dx + o;
                    t0 = *reinterpret_cast<const u32x4*>(a.xr + o); t1 = *reinterpret_cast<const u32x4*>(a.xr + o + 8);
                }
                ryv[ml][0] = t0[0]; ryv[ml][1] = t0[1]; ryv[ml][2] = t0[2]; ryv[ml][3] = t0[3];
                ryv[ml][4] = t1[0]; ryv[ml][5] = t1[1]; ryv[ml][6] = t1[2]; ryv[ml][7] = t1[3];
            }
        };
        if constexpr ((DBG & 2) != 0) { head_work(); if constexpr ((DBG & 4096) != 0) ts[1] = __builtin_amdgcn_s_memtime(); }
        // ---------------- weight gradient: 4 K-steps x 9 taps x (2 x 2 fragments); the dy fragments of the next tap are requested while
        // this tap's four MFMAs run (fenced: an unfenced schedule hoists dozens of fragment reads and spills)
        if constexpr (!(DBG & 2)) {
            const char* curc = reinterpret_cast<const char*>(cur);
            // one wave per SIMD has nobody to hide its LDS latency behind: the dy fragments are requested WR - 1 taps ahead (a ring of WR),
            // the a fragments of the next K-step half a K-step ahead
            constexpr int WR = 4;
            hx8 afrag[WR][2], bfrag[2][2];
            auto load_a = [&](int st, int buf) {   // st = ks * 9 + tap
                const int ks = st / 9, tap = st - ks * 9, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
                    afrag[buf][fi] = tr_frag(curc + (2 * ks + kh) * (HW * 128) + xo[kw][0][fi], curc + (2 * ks + kh) * (HW * 128) + xo[kw][1][fi]);
            };
            auto load_b = [&](int ks, int buf) {
#pragma unroll
                for (int fj = 0; fj < 2; ++fj) bfrag[buf][fj] = tr_frag(curc + 2 * ks * TW * 128 + dof[0][fj], curc + 2 * ks * TW * 128 + dof[1][fj]);
            };
            load_b(0, 0);
#pragma unroll
            for (int i = 0; i < WR - 1; ++i) load_a(i, i);
            head_work();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((DBG & 4096) != 0) { ts[1] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int st = 0; st < 36; ++st) {
                const int ks = st / 9, tap = st - ks * 9;
                if (st + WR - 1 < 36 && !((DBG & 2048) && ((st + WR - 1) & 1))) load_a(st + WR - 1, (st + WR - 1) % WR);   // DBG 2048: half the dy fragment reads
                if (tap == 4 && ks + 1 < TH / 2) load_b(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int fj = 0; fj < 2; ++fj) wacc[tap][fi][fj] = HX::mfma16(afrag[st % WR][fi], bfrag[ks & 1][fj], wacc[tap][fi][fj]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // gfx9 counts loads and stores in ONE counter and only loads return in order: with the previous tile's dx stores possibly pending,
        // every wait on a staged slot would become vmcnt(0) -- also for the slots just re-requested.  So ONE full wait here, where it is
        // free (everything outstanding was issued at least a weight-gradient phase ago and is needed now), and none after the refills
        if constexpr ((DBG & 4096) != 0) { ts[2] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((DBG & 4096) != 0) { ts[3] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        // ---------------- input gradient: 18 K-steps x (2 pixel + 4 filter fragments, 8 MFMAs), the next tile's staging in their shadow
        f32x4 acc[2][4];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[ml][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the epilogue's per-channel constants (the feeding layer's scale | shift of this lane's 16 channels): the first half is requested at the
        // top of the LAST K-step, whose fragment ring has a free half, so that the epilogue does not open with an exposed LDS round trip
        f32x4 rsv[4], rhv[4];
        auto load_rsrh = [&](int j0, int j1) {
#pragma unroll
            for (int jj = j0; jj < j1; ++jj) {
                rsv[jj] = *reinterpret_cast<const f32x4*>(sTab + 16 * q + 4 * jj);
                rhv[jj] = *reinterpret_cast<const f32x4*>(sTab + C + 16 * q + 4 * jj);
            }
        };
        if constexpr ((DBG & 1) != 0) load_rsrh(0, 2);
        if constexpr (!(DBG & 1)) {
            // fragment schedule: every fragment of K-step sidx + 1 is requested at the top of K-step sidx (8 MFMAs + the staging units ahead
            // of its use: one wave per SIMD has to cover the LDS latency itself); 48 registers of operands
            hx8 pix[2][2], filA[2][2], filB[2][2];
            auto load_pix = [&](int sidx, int buf) {
                const int tap = sidx >> 1, ks = sidx & 1;
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
                    pix[buf][ml] = *reinterpret_cast<const hx8*>(cur + aoff[kw][ks] + (ml + kh) * (HW * 128));
            };
            auto load_fil = [&](int sidx, int half, hx8 (&f)[2]) {
                const int tap = sidx >> 1, ks = sidx & 1;
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    f[n] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * C + (2 * half + n) * 16) * (C * 2));
            };
            // the region's schedule: MFMA, up to IV VALU instructions, MFMA, ... (whatever is left follows the fourth group)
            auto interleave4 = [] {
                constexpr int IV = (DBG & 8192) ? 3 : 5;
#pragma unroll
                for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, IV, 0); }
            };
            load_pix(0, 0);
            load_fil(0, 0, filA[0]);
            load_fil(0, 1, filB[0]);
#pragma unroll
            for (int sidx = 0; sidx < 18; ++sidx) {
                const int cb = sidx & 1;
                constexpr bool inter = stage && !(DBG & 128);
                if constexpr (inter) { if ((2 * sidx) % 18 < 16 && ((2 * sidx) % 18) % 4 == 0) pub_consts(((2 * sidx) % 18) / 4); }
                if (sidx == 17 && (PREMASKED || GVEC)) load_rsrh(0, 2);   // (all four vectors here cost the unmasked-g variant 2 spilled registers; its f16 twin has no room for any)
                if (sidx + 1 < 18) {
                    load_pix(sidx + 1, cb ^ 1);
                    if (!((DBG & 1024) && (sidx & 1))) { load_fil(sidx + 1, 0, filA[cb ^ 1]); load_fil(sidx + 1, 1, filB[cb ^ 1]); }   // DBG 1024: half the filter fragment reads
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[ml][n] = HX::mfma16(filA[cb][n], pix[cb][ml], acc[ml][n]);
                if constexpr (inter) {
#pragma unroll
                    for (int u = hs_first(2 * sidx); u < hs_first(2 * sidx) + hs_count(2 * sidx); ++u) pub_unit(u, nxt, refill, g2, okn);
                    interleave4();
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (inter) { if ((2 * sidx + 1) % 18 < 16 && ((2 * sidx + 1) % 18) % 4 == 0) pub_consts(((2 * sidx + 1) % 18) / 4); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[ml][2 + n] = HX::mfma16(filB[cb][n], pix[cb][ml], acc[ml][2 + n]);
                if constexpr (inter) {
#pragma unroll
                    for (int u = hs_first(2 * sidx + 1); u < hs_first(2 * sidx + 1) + hs_count(2 * sidx + 1); ++u) pub_unit(u, nxt, refill, g2, okn);
                    interleave4();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (stage && (DBG & 128) != 0) {   // DBG 128: the staging as one block after the MFMAs instead of between them
#pragma unroll
                for (int u = 0; u < NUNIT; ++u) { if (u % 25 < 20 && u % 5 == 0) pub_consts((u % 25) / 5); pub_unit(u, nxt, refill, g2, okn); __builtin_amdgcn_sched_barrier(0); }
            }
        } else if constexpr (stage) {
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) { if (u % 25 < 20 && u % 5 == 0) pub_consts((u % 25) / 5); pub_unit(u, nxt, refill, g2, okn); }
        }
        if constexpr ((DBG & 4096) != 0) { __builtin_amdgcn_sched_barrier(0); ts[4] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        // ---------------- input-gradient epilogue: layer L-1's BatchNorm-backward sums (gz = dx * [z > 0], dx rounded as stored), pack, store
        // (one wave per SIMD pays an issue slot for every instruction: 15 VALU instructions per channel pair; branch-free, so both pixel rows
        // schedule as one block)
        if constexpr (!(PREMASKED || GVEC) && !(DBG & 1)) load_rsrh(0, 2);
        load_rsrh(2, 4);
        if constexpr (!(DBG & 4))
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            unsigned pk[8];
            // the ReLU mask is applied IN FRONT of the pack (a select per value, the same count as selecting halves of the packed word
            // behind it, but without the two ANDs that prepared those halves); outside the image the threshold is +inf: nothing passes,
            // the sums take zeros (whole-tile shapes: the threshold is the constant 0)
            const float thr = (ALIGNED || inb[ml]) ? 0.f : __builtin_inff();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nf = j >> 1, i0 = 2 * (j & 1), jj = j >> 1, jh = j & 1;
                const float y0 = HX::lo(ryv[ml][j]), y1 = HX::hi(ryv[ml][j]);
                const float z0 = __builtin_fmaf(rsv[jj][2 * jh], y0, rhv[jj][2 * jh]), z1 = __builtin_fmaf(rsv[jj][2 * jh + 1], y1, rhv[jj][2 * jh + 1]);
                const hx2 p2v = HX::pack2((z0 > thr ? acc[ml][nf][i0] : 0.f), (z1 > thr ? acc[ml][nf][i0 + 1] : 0.f));
                pk[j] = __builtin_bit_cast(unsigned, p2v);   // dx leaves masked: gz, not g
                const float gz0 = HX::lo(pk[j]), gz1 = HX::hi(pk[j]);
                s1[2 * j] += gz0; s1[2 * j + 1] += gz1;
                s2[2 * j] = __builtin_fmaf(gz0, y0, s2[2 * j]);
                s2[2 * j + 1] = __builtin_fmaf(gz1, y1, s2[2 * j + 1]);
            }
            if constexpr (ALIGNED) {
                if (!(DBG & 512)) {   // DBG 512: no dx stores
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, rsD, eo[ml], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[4], pk[5], pk[6], pk[7]}, rsD, eo[ml] + 16u, 0, 0);
                }
            } else if (inb[ml] && !(DBG & 512)) {
                *reinterpret_cast<u32x4*>(outp[ml]) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(outp[ml] + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
        }
        if constexpr ((DBG & 4096) != 0) { __builtin_amdgcn_sched_barrier(0); ts[5] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        __syncthreads();
        if constexpr ((DBG & 4096) != 0) {
            ts[6] = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 6; ++i) ph[i] += ts[i + 1] - ts[i];
        }
    };
    {
        typedef std::integral_constant<bool, true> yes;
        typedef std::integral_constant<bool, false> no;
        int tile = t_begin;
        for (; tile + 2 < t_end; ++tile) tile_body(tile, yes{}, yes{});
        if (tile + 1 < t_end) { tile_body(tile, yes{}, no{}); ++tile; }
        if (tile < t_end) tile_body(tile, no{}, no{});
    }

    if constexpr ((DBG & 4096) != 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dx) + (size_t)blockIdx.x * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = ph[i];
            o[6] = t1 - t_run0; o[7] = r1 - r_run0;
        }
    }
    // ---- weight-gradient slab: wacc[tap][fi][fj][i] = sum_q dy[q + (kh-1, kw-1)][co] * a[q][ci], co = 32mi + 16fi + 4kq + i,
    // ci = 32ni + 16fj + r: that is dW of filter tap 8 - tap; slab layout [tap][ci][co] (wgrad.hip's reduction)
    float* slab = a.ws + (size_t)blockIdx.x * 9 * C * C;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const int ci = ni * 32 + fj * 16 + r, co = mi * 32 + fi * 16 + 4 * kq;
                *reinterpret_cast<f32x4*>(slab + ((size_t)(8 - tap) * C + ci) * C + co) = wacc[tap][fi][fj];
            }
    // ---- BatchNorm-backward partial rows of layer L-1
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float u1 = s1[c], u2 = s2[c];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
        if (p == 0) {
            sRed[(wave * 2 + 0) * C + 16 * q + c] = u1;
            sRed[(wave * 2 + 1) * C + 16 * q + c] = u2;
        }
    }
    __syncthreads();
    if (tid < 2 * C) {
        const int which = tid / C, n = tid - which * C;
        a.stat[((size_t)blockIdx.x * 2 + which) * C + n] =
            (sRed[(0 * 2 + which) * C + n] + sRed[(1 * 2 + which) * C + n]) + (sRed[(2 * 2 + which) * C + n] + sRed[(3 * 2 + which) * C + n]);
    }
}

}  // namespace

int WM_HSYM(wm_bwd_ws_gvec_max_batch)() { return GV_MAXB; }

// nwg workgroups (= slabs = partial rows), each a run of 8x16-pixel tiles
void WM_HSYM(wm_launch_bwd_ws)(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt, const void* xr,
                               const float* in_scale, const float* in_shift, void* dx, float* stat, float* ws, int B, int H, int W, int nwg,
                               int reverse, hipStream_t s, int dbg, int premasked, const float* gvec, int gv_ld) {
    BwdArgs a;
    a.g = (const hx_t*)g; a.y = (const hx_t*)y; a.stats4 = stats4; a.st_ld = st_ld; a.coef = coef; a.wpt = (const hx_t*)wpt;
    a.gvec = gvec; a.gv_ld = gv_ld;
    a.xr = (const hx_t*)xr; a.in_scale = in_scale; a.in_shift = in_shift; a.dx = (hx_t*)dx; a.stat = stat; a.ws = ws;
    a.B = B; a.H = H; a.W = W; a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH); a.ntiles = B * a.tilesX * a.tilesY;
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
    a.mX = magic(a.tilesX); a.mY = magic(a.tilesY); a.m2X = magic(2 * a.tilesX);
    a.reverse = wm_sweep_dir(reverse);
    const bool aligned = H % TH == 0 && W % TW == 0 && !(dbg & (1 << 20));   // (debug bit 20: the general addressing on a whole-tile shape, for A/B)
#ifdef WM_DEBUG
    if (!gvec && aligned)   // phase stamps of the whole-tile (buffer-addressed) form: tools/phase_bwd.py
    switch (dbg) {
        case 4608: hipLaunchKernelGGL((bwd_ws_kernel<4608, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4864: hipLaunchKernelGGL((bwd_ws_kernel<4864, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4616: hipLaunchKernelGGL((bwd_ws_kernel<4616, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        default: break;
    }
    if (!gvec)   // (the ablation variants exist for the tensor-gradient form only; a gvec launch has no g to read)
    switch (dbg & ~(1 << 20)) {
        case 1: hipLaunchKernelGGL((bwd_ws_kernel<1, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 2: hipLaunchKernelGGL((bwd_ws_kernel<2, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 3: hipLaunchKernelGGL((bwd_ws_kernel<3, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4: hipLaunchKernelGGL((bwd_ws_kernel<4, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 8: hipLaunchKernelGGL((bwd_ws_kernel<8, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 32: hipLaunchKernelGGL((bwd_ws_kernel<32, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 64: hipLaunchKernelGGL((bwd_ws_kernel<64, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 512: hipLaunchKernelGGL((bwd_ws_kernel<512, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 1024: hipLaunchKernelGGL((bwd_ws_kernel<1024, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 2048: hipLaunchKernelGGL((bwd_ws_kernel<2048, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 3072: hipLaunchKernelGGL((bwd_ws_kernel<3072, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4608: hipLaunchKernelGGL((bwd_ws_kernel<4608, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4616: hipLaunchKernelGGL((bwd_ws_kernel<4616, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4640: hipLaunchKernelGGL((bwd_ws_kernel<4640, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4672: hipLaunchKernelGGL((bwd_ws_kernel<4672, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 4864: hipLaunchKernelGGL((bwd_ws_kernel<4864, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        case 15: hipLaunchKernelGGL((bwd_ws_kernel<15, false>), dim3((unsigned)nwg), dim3(256), 0, s, a); return;
        default: break;
    }
#endif
    (void)dbg;
    if (aligned) {
        if (gvec) hipLaunchKernelGGL((bwd_ws_kernel<0, false, true, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        else if (premasked) hipLaunchKernelGGL((bwd_ws_kernel<0, true, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((bwd_ws_kernel<0, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        return;
    }
    if (gvec) hipLaunchKernelGGL((bwd_ws_kernel<0, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    else if (premasked) hipLaunchKernelGGL((bwd_ws_kernel<0, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bwd_ws_kernel<0, false>), dim3((unsigned)nwg), dim3(256), 0, s, a);
}
