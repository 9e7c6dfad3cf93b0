// Fused backward of an IMAGE-FED first ConvBNRelu layer, 3 (stored as 16) -> 64 channels (gfx950, bf16 / f16): the decoder's and the
// discriminator's conv1 (hidden_models/decoder.py:16, discriminator.py:13), whose input is an image that needs a gradient.  ONE kernel per
// 8x16-pixel tile computes
//   dy  = BatchNorm-backward apply of (g, y)                    (formed while staging, never written to memory)
//   dx  = conv3x3(dy, W^T)          [B,H,W,16]                  the gradient wrt the image (3 real channels)
//   dW += sum_pixels dy (x) x       one slab [9][16][64] per workgroup (wgrad.hip's reduction finishes it)
// from ONE staged dy halo tile.  The two-kernel form (conv3x3_ws.hip <64,32,..,BNBWD = 2> + wgrad_ws.hip <16,..,0>) reads (g, y) to form
// dy, writes dy (134 MB at B = 16, 256x256) and reads it back for the weight gradient: 470 + 167 MB; this kernel moves 268 + 33 + 33 MB.
//
// It is csrc/bwd_ws.hip's scheme (same LDS layout of the dy halo: 128-byte pixel rows, 16-byte slots XORed with fsw(px), read both by the
// input gradient's ds_read_b128 and by the weight gradient's transposing ds_read_b64_tr_b16) with two differences that follow from the
// shapes.  (1) Both GEMMs are tiny -- 36 + 36 MFMAs per wave and tile against 144 + 144 -- so the kernel is bound by the staging and by
// memory, not by the matrix pipe: instead of hand-interleaving the staging with the MFMAs, a workgroup needs so little LDS (75 KB) and so
// few registers that TWO of them share a CU and fill each other's stalls.  (2) The image tile is stored TRANSPOSED in the LDS
// ([channel][pixel], 8 two-byte writes per thread and tile) so that the weight gradient's B operand is a plain 16-byte read.
// Whole-tile shapes only (H % 8 == 0, W % 16 == 0: buffer addressing with per-thread constant offsets, as bwd_ws.hip's ALIGNED form);
// other shapes keep the two-kernel form.
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

constexpr int TH = 8, TW = 16, HH = 10, HW = 18, NPX = HH * HW, C = 64, CX = 16;
constexpr int SW_BYTES = 9 * CX * C * 2;          // filter for the input gradient [9][16 rows = image channel][64 dy channels]
constexpr int SDY_BYTES = NPX * 128;              // dy halo
constexpr int XROW = TH * TW * 2 + 16;            // transposed image tile: [16 channels][128 pixels] + 16 B pad per row (bank spread)
constexpr int SX_BYTES = CX * XROW;
constexpr int BUF_BYTES = SDY_BYTES + SX_BYTES;
constexpr int XV = (NPX * 8 + 255) / 256;         // dy halo vectors per thread (6; the last one partially live)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Bwd16Args {
    const hx_t* g; const hx_t* y;                 // layer 1: gradient wrt its ReLU output, its raw conv output [B,H,W,64]
    const float* stats4; int st_ld; const float* coef;
    const hx_t* wpt;                              // [9][16][64] filter packed for the input gradient (wm_pack_w3x3, transposed)
    const hx_t* x;                                // the layer's input: image [B,H,W,16] (3 real channels)
    hx_t* dx;                                     // [B,H,W,16]
    float* ws;                                    // [gridDim.x][9][16][64]
    int B, H, W, tilesX, tilesY, ntiles, reverse;
};

__device__ __forceinline__ int fsw(int px) { return ((px >> 2) & 1) | (((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2); }
__device__ __forceinline__ int swzw(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ hx8 tr_frag(const char* p0, const char* p1) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(hx8, v);
}

// PREMASKED: g arrives already multiplied by the layer's ReLU mask (what every gradient-producing kernel of this library writes)
template <bool PREMASKED>
__global__ __launch_bounds__(256, 2) void bwd_ws16_kernel(Bwd16Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * BUF_BYTES + (C * 8 + 32) * 4];
    hx_t* sW = reinterpret_cast<hx_t*>(smem);
    unsigned char* sBuf = smem + SW_BYTES;
    float* sK = reinterpret_cast<float*>(smem + SW_BYTES + 2 * BUF_BYTES);   // per channel: scale, shift, ca, k2, k3 (wm_bn_fold), 0, 0, 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < C) {
        float k2, k3;
        wm_bn_fold(a.stats4[2 * a.st_ld + tid], a.stats4[3 * a.st_ld + tid], a.coef[tid], a.coef[a.st_ld + tid], a.coef[2 * a.st_ld + tid], k2, k3);
        const float v[8] = {a.stats4[tid], a.stats4[a.st_ld + tid], a.coef[tid], k2, k3, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) sK[tid * 8 + (tid >> 3) * 4 + i] = v[i];   // (8-channel blocks shifted by 16 B: bwd_ws.hip's bank spread)
    }
    // ---- filter -> LDS: row = tap * 16 + image channel, 64 dy channels = 8 sixteen-byte slots, slot XORed with swzw(row)
    // (all loads in flight, then the stores: a `load, wait, store` loop runs its trips one L2 round trip after the other)
    {
        constexpr int NV = 9 * CX * 8, WV = (NV + 255) / 256;
        hx8 wv[WV];
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = min(tid + 256 * k, NV - 1);
            wv[k] = *reinterpret_cast<const hx8*>(a.wpt + (size_t)(i >> 3) * C + (i & 7) * 8);
        }
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 256 * k, row = i >> 3;
            if (i < NV) *reinterpret_cast<hx8*>(sW + row * C + swzw(row, i & 7) * 8) = wv[k];
        }
    }
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware: consecutive runs per XCD
    const int t_begin = (int)(((long)run * a.ntiles) / G), t_end = (int)(((long)(run + 1) * a.ntiles) / G);
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {   // pairs of tile rows walked column by column (bwd_ws.hip's order; tilesY odd: row-major)
        TileGeo g;
        const int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;
        if (a.tilesY & 1) {
            const int q1 = t / a.tilesX;
            g.tx0 = (t - q1 * a.tilesX) * TW; g.b = q1 / a.tilesY; g.ty0 = (q1 - g.b * a.tilesY) * TH;
        } else {
            const int pr = t / (2 * a.tilesX), rem = t - pr * 2 * a.tilesX, row = 2 * pr + (rem & 1);
            g.b = row / a.tilesY; g.ty0 = (row - g.b * a.tilesY) * TH; g.tx0 = (rem >> 1) * TW;
        }
        return g;
    };

    // ================================================================== staging role
    const int vec = tid & 7, slot = tid >> 3;
    unsigned hofs[XV], edge = 0;
    int hlds[XV];
#pragma unroll
    for (int k = 0; k < XV; ++k) {
        const int hp = min(slot + 32 * k, NPX - 1), py = hp / HW, px = hp - py * HW;
        hlds[k] = hp * 128 + ((vec ^ fsw(px)) << 4);
        hofs[k] = (unsigned)(((py * a.W + px) * C + vec * 8) * 2);
        edge |= (py == 0 ? 1u : 0u) << k | (py == HH - 1 ? 1u : 0u) << (k + 6) | (px == 0 ? 1u : 0u) << (k + 12) | (px == HW - 1 ? 1u : 0u) << (k + 18);
    }
    const bool last_live = slot + 32 * (XV - 1) < NPX;
    // image tile: thread = (pixel tid >> 1, 8-channel half tid & 1): one 16-byte load, eight 2-byte transposed writes
    const int xpx = tid >> 1, xvec = tid & 1;
    const unsigned xofs = (unsigned)((((xpx >> 4) * a.W + (xpx & 15)) * CX + xvec * 8) * 2);
    const unsigned nbytes64 = (unsigned)a.B * (unsigned)a.H * (unsigned)a.W * (unsigned)(C * 2);
    const unsigned nbytes16 = (unsigned)a.B * (unsigned)a.H * (unsigned)a.W * (unsigned)(CX * 2);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.g), 0, nbytes64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.y), 0, nbytes64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.x), 0, nbytes16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, nbytes16, 0x00020000);
    auto pix_index = [&](const TileGeo& t) { return (unsigned)(t.b * a.H + t.ty0) * (unsigned)a.W + (unsigned)t.tx0; };   // the tile's first pixel
    auto inside_bits = [&](const TileGeo& t) {
        const unsigned sel = (t.ty0 == 0 ? 0x3fu : 0u) | (t.ty0 + TH == a.H ? 0x3fu << 6 : 0u) | (t.tx0 == 0 ? 0x3fu << 12 : 0u) | (t.tx0 + TW == a.W ? 0x3fu << 18 : 0u);
        const unsigned e = edge & sel;
        return ~(e | (e >> 6) | (e >> 12) | (e >> 18)) & 0x3fu;
    };
    hx8 dG[XV], dY[XV], dX;
    unsigned okh = 0;
    auto load_tile = [&](const TileGeo& t) {
        const unsigned hb = pix_index(t) * (unsigned)(C * 2) - (unsigned)((a.W + 1) * C * 2);   // halo origin; "negative" wraps beyond the descriptor's range: zeros
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            dG[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsG, hb + hofs[k], 0, 0));
            dY[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsY, hb + hofs[k], 0, 0));
        }
        dX = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, pix_index(t) * (unsigned)(CX * 2) + xofs, 0, 0));
        okh = inside_bits(t);
    };
    auto publish_tile = [&](unsigned char* buf) {
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const float* kp = sK + (vec * 8 + 2 * pq) * 8 + vec * 4;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(kp), kb = *reinterpret_cast<const f32x4*>(kp + 8);
            const float k3a = kp[4], k3b = kp[12];
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                u32x4 w = __builtin_bit_cast(u32x4, dG[k]);
                const u32x4 wy = __builtin_bit_cast(u32x4, dY[k]);
                float d0, d1;
                if constexpr (PREMASKED) {
                    d0 = __builtin_fmaf(ka[2], HX::lo(w[pq]), __builtin_fmaf(-ka[3], HX::lo(wy[pq]), k3a));
                    d1 = __builtin_fmaf(kb[2], HX::hi(w[pq]), __builtin_fmaf(-kb[3], HX::hi(wy[pq]), k3b));
                } else {
                    d0 = wm_bn_fold_dyg(HX::lo(wy[pq]), HX::lo(w[pq]), ka[0], ka[1], ka[2], ka[3], k3a);
                    d1 = wm_bn_fold_dyg(HX::hi(wy[pq]), HX::hi(w[pq]), kb[0], kb[1], kb[2], kb[3], k3b);
                }
                const hx2 pk = HX::pack2(d0, d1);
                w[pq] = __builtin_bit_cast(unsigned, pk);
                dG[k] = __builtin_bit_cast(hx8, w);
            }
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            u32x4 w = __builtin_bit_cast(u32x4, dG[k]);
            const unsigned keep = 0u - ((okh >> k) & 1u);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] &= keep;
            if (k + 1 < XV || last_live) *reinterpret_cast<u32x4*>(buf + hlds[k]) = w;
        }
        {   // the image tile, transposed: channel 8 xvec + e of pixel xpx -> row (8 xvec + e), column xpx
            const u32x4 w = __builtin_bit_cast(u32x4, dX);
            unsigned char* xt = buf + SDY_BYTES + (8 * xvec) * XROW + xpx * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<unsigned short*>(xt + (2 * e) * XROW) = (unsigned short)(w[e] & 0xffffu);
                *reinterpret_cast<unsigned short*>(xt + (2 * e + 1) * XROW) = (unsigned short)(w[e] >> 16);
            }
        }
    };

    // ================================================================== input-gradient role: tile rows 2*wave, 2*wave + 1, 16 image channels
    // A = filter fragment (rows = image channel p, K = 32 dy channels), B = dy pixels: D row 4q + i = image channel, column = pixel p
    const int p = lane & 15, q = lane >> 4;
    int aoff[3][2], boff[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[kw][ks] = ((wave * 2 * HW + p + kw) * 128) + (((ks * 4 + q) ^ fsw(p + kw)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (p * C + swzw(p, ks * 4 + q) * 8) * 2;
    const unsigned eofs = (unsigned)(((wave * 2 * a.W + p) * CX + 4 * q) * 2);   // this lane's 4 channels of output pixel (row 2 wave, column p)

    // ================================================================== weight-gradient role: wave w owns dy channels [16w, 16w + 16) x 16 image channels x 9 taps
    // D[co 16 x ci 16] += A[co x 32 pixels] (dy halo shifted by the tap, transposing reads as bwd_ws.hip) * B[32 pixels x ci] (transposed image tile)
    const int r = lane & 15, kq = lane >> 4, q2 = (lane >> 2) & 3, p2 = lane & 3;
    const int colb = 8 * (kq & 1) + q2;
    int xo[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int col = colb + kw + 4 * sx;
            const int sl = wave * 2 + (p2 >> 1);
            xo[kw][sx] = ((kq >> 1) * HW + col) * 128 + ((sl ^ fsw(col)) << 4) + (p2 & 1) * 8;
        }
    const int xb = SDY_BYTES + r * XROW + kq * 16;   // B: image channel r, pixels 8 kq .. 8 kq + 7 of the K-step
    f32x4 wacc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();   // the constant table
    if (t_begin < t_end) {
        load_tile(geo(t_begin));
        publish_tile(sBuf);
    }
    __syncthreads();   // filter + first tile visible

    for (int tile = t_begin; tile < t_end; ++tile) {
        const TileGeo g = geo(tile);
        const unsigned char* cur = sBuf + ((tile - t_begin) & 1) * BUF_BYTES;
        unsigned char* nxt = sBuf + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
        const bool more = tile + 1 < t_end;
        if (more) load_tile(geo(tile + 1));   // in flight during this tile's MFMAs (and the partner workgroup's work)
        const char* curc = reinterpret_cast<const char*>(cur);
        // ---------------- weight gradient: 4 K-steps (two tile rows each) x 9 taps
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            const hx8 bfr = *reinterpret_cast<const hx8*>(curc + xb + ks * 64);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const hx8 afr = tr_frag(curc + (2 * ks + kh) * (HW * 128) + xo[kw][0], curc + (2 * ks + kh) * (HW * 128) + xo[kw][1]);
                wacc[tap] = HX::mfma16(afr, bfr, wacc[tap]);
            }
        }
        // ---------------- input gradient: 9 taps x 2 K-steps of 32 dy channels
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int sidx = 0; sidx < 18; ++sidx) {
            const int tap = sidx >> 1, ks = sidx & 1, kh = tap / 3, kw = tap - kh * 3;
            const hx8 fil = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * CX) * (C * 2));
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) {
                const hx8 pix = *reinterpret_cast<const hx8*>(cur + aoff[kw][ks] + (ml + kh) * (HW * 128));
                acc[ml] = HX::mfma16(fil, pix, acc[ml]);
            }
        }
        // ---------------- epilogue: the image needs no mask and feeds no BatchNorm: pack and store 4 channels (8 bytes) per lane and row
        const unsigned eb = pix_index(g) * (unsigned)(CX * 2) + eofs;
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            const hx2 lo = HX::pack2(acc[ml][0], acc[ml][1]), hi = HX::pack2(acc[ml][2], acc[ml][3]);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)}, rsD,
                                                  eb + (unsigned)(ml * a.W * CX * 2), 0, 0);
        }
        if (more) publish_tile(nxt);
        __syncthreads();
    }
    // ---- weight-gradient slab [tap][ci][co] (wgrad.hip's reduction): wacc[tap][i] = sum_q dy[q + (kh-1, kw-1)][co] * x[q][ci] = dW of tap 8 - tap,
    // co = 16 wave + 4 kq + i, ci = r
    float* slab = a.ws + (size_t)blockIdx.x * 9 * CX * C;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        *reinterpret_cast<f32x4*>(slab + ((size_t)(8 - tap) * CX + r) * C + 16 * wave + 4 * kq) = wacc[tap];
}

}  // namespace

// nwg workgroups (= slabs), each a run of 8x16-pixel tiles
void WM_HSYM(wm_launch_bwd_ws16)(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt, const void* x,
                                 void* dx, float* ws, int B, int H, int W, int nwg, int reverse, hipStream_t s, int premasked) {
    Bwd16Args a;
    a.g = (const hx_t*)g; a.y = (const hx_t*)y; a.stats4 = stats4; a.st_ld = st_ld; a.coef = coef; a.wpt = (const hx_t*)wpt;
    a.x = (const hx_t*)x; a.dx = (hx_t*)dx; a.ws = ws;
    a.B = B; a.H = H; a.W = W; a.tilesX = W / TW; a.tilesY = H / TH; a.ntiles = B * a.tilesX * a.tilesY;
    a.reverse = wm_sweep_dir(reverse);
    // (dynamic LDS for the allocation's size alone: see wgrad_ws.hip WM_LDS_PAD16 -- a workgroup of the JPEG kernels must not fit beside this
    //  kernel on a CU.  It costs this kernel its second workgroup per CU.)
    constexpr int PAD = 137472 - (SW_BYTES + 2 * BUF_BYTES + (C * 8 + 32) * 4);
    if (premasked) hipLaunchKernelGGL((bwd_ws16_kernel<true>), dim3((unsigned)nwg), dim3(256), PAD, s, a);
    else hipLaunchKernelGGL((bwd_ws16_kernel<false>), dim3((unsigned)nwg), dim3(256), PAD, s, a);
}
