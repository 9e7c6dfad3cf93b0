// The one-pass backward of a 64 -> 64 ConvBNRelu body layer (see csrc/bwd_ws.hip for what it computes and for the LDS layouts) with the
// two GEMMs on DIFFERENT waves: a 512-thread workgroup = 4 "D" waves (input gradient of two tile rows each + staging of the dy halo + the
// epilogue with the feeding layer's BatchNorm sums) and 4 "W" waves (weight gradient, a 32 x 32 block x 9 taps each, + staging of the
// feeding layer's activated tile), one of each per SIMD.
//
// Why (round 3, rocprofv3 SQ counters of bwd_ws.hip alone, profiles/r03_bwd_sq_counters.txt): with ONE wave per SIMD the wave is parked on
// s_waitcnt / the barrier for 20 % of its cycles and stalled at issue (the next MFMA waits for the matrix pipe) for another 26 %; only 53 %
// of the time something issues, and the matrix pipe is busy 36 %.  A single wave issues in order, so nothing fills those gaps -- moving
// instructions between the two MFMA loops or into each other's shadow moved the cycles with them.  A second wave on the
// SIMD does fill them.  Round 2 ruled a partner wave out from a microbenchmark in which the MFMA wave issued back to back (a partner gets
// ~1.6 VALU slots per MFMA there); this kernel is nowhere near back-to-back, and both roles carry MFMAs.
//
// Registers decide the split: two waves per SIMD have 256 registers each.  The weight gradient's 144 accumulators + its fragment ring leave
// a W wave room for the 4 a-tile slots only; a D wave holds the input gradient's accumulators, 48 registers of fragments, the 6 dy-halo
// slots (g and y: 48), the epilogue's sums and operand.  The roles run their own tile loops (separate code paths: a variable of one role is
// not live in the other's loop) and meet at the one barrier per tile.  Whole-tile shapes only (buffer addressing as bwd_ws.hip's ALIGNED
// form); everything else stays on bwd_ws.hip.
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
constexpr int XV = (NPX * 8 + 255) / 256;        // dy halo vectors per D thread (6; the last one partially live)
constexpr int AV = TH * TW * 8 / 256;            // a-tile vectors per W thread (4)
constexpr int GV_MAXB = 24;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Bwd8Args {
    const hx_t* g; const hx_t* y;
    const float* gvec; int gv_ld;
    const float* stats4; int st_ld; const float* coef;
    const hx_t* wpt;
    const hx_t* xr; const float* in_scale; const float* in_shift;
    hx_t* dx;
    float* stat;                                           // [gridDim.x][2][64]
    float* ws;                                             // [gridDim.x][9][64][64]
    int B, H, W, tilesX, tilesY, ntiles, reverse;
    int tq, trem;                                          // ntiles / gridDim.x, ntiles % gridDim.x
    unsigned mX, mY, m2X;
    int stamps;                                            // debug build: per-wave phase cycle sums instead of the partial rows
};

// debug build (tools/phase_bwd8.py): s_memtime sums per wave -- [0] the MFMA loop (+ the units between its MFMAs), [1] W: staging of the
// a tile / D: the wait at the tile's barrier, [2] W: the wait at the barrier / D: the epilogue; written over the workgroup's partial rows
#ifdef WM_DEBUG
#define WM_STAMP(i) if (a.stamps) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
#else
#define WM_STAMP(i)
#endif

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

template <bool PREMASKED, bool GVEC>
__global__ __launch_bounds__(512, 1) void bwd_ws8_kernel(Bwd8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + 2 * BUF_BYTES + 4 * 2 * C * 4 + 2 * C * 4 + (C * 8 + 32) * 4 + (GVEC ? GV_MAXB * C * 4 : 0)];
    hx_t* sW = reinterpret_cast<hx_t*>(smem);
    unsigned char* sBuf = smem + SW_BYTES;
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + 2 * BUF_BYTES);
    float* sTab = sRed + 4 * 2 * C;          // in_scale | in_shift of the feeding layer
    float* sK = sTab + 2 * C;                // per channel: scale, shift, ca, k2, k3 of layer L; in_scale, in_shift of layer L-1; 0
    float* sG = sK + C * 8 + 32;             // GVEC: [B][64] k3 + ca * gvec[b][c]
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wrole = tid >= 256;           // waves 4..7: the weight gradient
    const int rt = tid & 255, wave = rt >> 6;   // index inside the role
#ifdef WM_DEBUG
    long long tacc[4] = {0, 0, 0, 0}, tprev = a.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;   // [3]: everything outside the tile loop
#endif
    // ---- prologue (round 4): EVERY global load of the prologue is issued before anything waits -- the constants (wave 0), the filter
    // (9 vectors per thread; round 3's loop `load, wait, write` ran its 9 trips one L2 round trip after the other: ~4 us of every launch)
    // and, further down, the role's first tile; the LDS commits follow in issue order (vmcnt retires in order), one barrier for all
    float cst[9];
    if (tid < C) {
        cst[0] = a.stats4[tid]; cst[1] = a.stats4[a.st_ld + tid]; cst[2] = a.stats4[2 * a.st_ld + tid]; cst[3] = a.stats4[3 * a.st_ld + tid];
        cst[4] = a.coef[tid]; cst[5] = a.coef[a.st_ld + tid]; cst[6] = a.coef[2 * a.st_ld + tid];
        cst[7] = a.in_scale[tid]; cst[8] = a.in_shift[tid];
    }
    hx8 wv[9];   // filter rows, permuted for the 16x16x32 consumers when committed (bwd_ws.hip)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = tid + 512 * k;
        wv[k] = *reinterpret_cast<const hx8*>(a.wpt + (size_t)(i >> 3) * C + (i & 7) * 8);
    }
    auto commit_prologue = [&]() {
        if (tid < C) {
            sTab[tid] = cst[7]; sTab[C + tid] = cst[8];
            float k2, k3;
            wm_bn_fold(cst[2], cst[3], cst[4], cst[5], cst[6], k2, k3);
            const float v[8] = {cst[0], cst[1], cst[4], k2, k3, cst[7], cst[8], 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) sK[tid * 8 + (tid >> 3) * 4 + i] = v[i];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = tid + 512 * k;
            const int row = i >> 3, tap = row / C, n = row % C;
            const int lrow = tap * C + ((n >> 2) & 3) * 16 + 4 * (n >> 4) + (n & 3);
            *reinterpret_cast<hx8*>(sW + lrow * C + swzw(lrow, i & 7) * 8) = wv[k];
        }
    };
    const int G = gridDim.x;
    const int run = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // run r takes tiles [r q + min(r, rem), ...): q = ntiles / G, rem = ntiles % G from the host (no 64-bit division in the prologue)
    const int t_begin = run * a.tq + min(run, a.trem), t_end = t_begin + a.tq + (run < a.trem ? 1 : 0);
    struct TileGeo { int b, ty0, tx0; };
    auto fdiv = [](int t, int d, unsigned m) { return d == 1 ? t : (int)__umulhi((unsigned)t, m); };
    auto geo = [&](int tile) {
        TileGeo g;
        const int t = a.reverse ? t_begin + (t_end - 1 - tile) : tile;
        if (a.tilesY & 1) {
            const int q1 = fdiv(t, a.tilesX, a.mX), txi = t - q1 * a.tilesX;
            const int q2 = fdiv(q1, a.tilesY, a.mY), tyi = q1 - q2 * a.tilesY;
            g.b = q2; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        } else {
            const int pr = fdiv(t, 2 * a.tilesX, a.m2X), rem = t - pr * 2 * a.tilesX;
            const int row = 2 * pr + (rem & 1);
            g.b = fdiv(row, a.tilesY, a.mY); g.ty0 = (row - g.b * a.tilesY) * TH; g.tx0 = (rem >> 1) * TW;
        }
        return g;
    };
    const unsigned nbytes = (unsigned)a.B * (unsigned)a.H * (unsigned)a.W * (unsigned)(C * 2);
    // byte offset of the tile's halo origin (pixel (ty0 - 1, tx0 - 1)); "negative" wraps beyond the descriptors' range: zeros
    auto halo_base = [&](const TileGeo& t) { return ((unsigned)(t.b * a.H + t.ty0) * (unsigned)a.W + (unsigned)t.tx0) * (unsigned)(C * 2) - (unsigned)((a.W + 1) * C * 2); };
    const int vec = rt & 7, slot = rt >> 3;
    typedef std::integral_constant<bool, true> yes;
    typedef std::integral_constant<bool, false> no;

    // after the role's first-tile loads are in flight: commit constants + filter, barrier; GVEC: the per-sample table needs the constants
    auto finish_prologue = [&]() {
        commit_prologue();
        __syncthreads();   // the constant table + the filter
        if constexpr (GVEC) {
            for (int i = tid; i < a.B * C; i += 512) {
                const int b = i / C, c = i - b * C;
                const float* kc = sK + c * 8 + (c >> 3) * 4;
                sG[i] = wm_bn_fold_g(kc[2], a.gvec[(size_t)b * a.gv_ld + c], kc[4]);
            }
            __syncthreads();
        }
    };

    if (wrole) {
        // ============================================================================================================ W role
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.xr), 0, nbytes, 0x00020000);
        const int alds0 = SDY_BYTES + slot * 128 + ((vec << 4) ^ swz16(slot & 15));   // a-tile slot k: + 32 k * 128 (two rows below)
        const unsigned aofs0 = (unsigned)((((slot >> 4) * a.W + (slot & 15)) * C + vec * 8) * 2);
        hx8 dA[AV];
        auto load_a_slot = [&](const TileGeo& t, int k) {
            dA[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, halo_base(t) + (unsigned)(((2 * k + 1) * a.W + 1) * C * 2) + aofs0, 0, 0));
        };
        f32x4 kin = {0.f, 0.f, 0.f, 0.f};   // in_scale, in_shift of the channel pair being transformed: {s0, s1, h0, h1}
        auto a_consts = [&](int pq) {
            const f32x2 s = *reinterpret_cast<const f32x2*>(sTab + vec * 8 + 2 * pq), h = *reinterpret_cast<const f32x2*>(sTab + C + vec * 8 + 2 * pq);
            kin = f32x4{s[0], s[1], h[0], h[1]};
        };
        auto a_transform = [&](int k, int pq) {
            u32x4 w = __builtin_bit_cast(u32x4, dA[k]);
            const float f0 = __builtin_fmaf(HX::lo(w[pq]), kin[0], kin[2]);
            const float f1 = __builtin_fmaf(HX::hi(w[pq]), kin[1], kin[3]);
            const hx2 pk = HX::pack2(f0, f1);
            const i16x2 z = {0, 0};
            w[pq] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), z));
            dA[k] = __builtin_bit_cast(hx8, w);
        };
        auto a_publish = [&](unsigned char* buf, int k) { *reinterpret_cast<u32x4*>(buf + (alds0 + k * 32 * 128)) = __builtin_bit_cast(u32x4, dA[k]); };
        // weight-gradient fragments: wave (mi, ni) owns the 32 co x 32 ci block
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

        if (t_begin < t_end) {   // first tile: load ...
            const TileGeo g0 = geo(t_begin);
#pragma unroll
            for (int k = 0; k < AV; ++k) load_a_slot(g0, k);
        }
        finish_prologue();
        if (t_begin < t_end) {   // ... transform, publish
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                a_consts(pq);
#pragma unroll
                for (int k = 0; k < AV; ++k) a_transform(k, pq);
            }
#pragma unroll
            for (int k = 0; k < AV; ++k) a_publish(sBuf, k);
        }
        __syncthreads();   // first tile visible
        if (t_begin + 1 < t_end) {
            const TileGeo g1 = geo(t_begin + 1);
#pragma unroll
            for (int k = 0; k < AV; ++k) load_a_slot(g1, k);
        }
        // one tile: 36 steps of 4 MFMAs, then the staging of the next tile's a tile
        auto w_tile = [&](int tile, auto stage_c, auto refill_c) __attribute__((always_inline)) {
            constexpr bool stage = decltype(stage_c)::value, refill = decltype(refill_c)::value;
            const char* curc = reinterpret_cast<const char*>(sBuf + ((tile - t_begin) & 1) * BUF_BYTES);
            unsigned char* nxt = sBuf + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
            constexpr int WR = 4;
            hx8 afrag[WR][2], bfrag[2][2];
            auto load_a = [&](int st, int buf) {
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
#pragma unroll
            for (int st = 0; st < 36; ++st) {
                const int ks = st / 9, tap = st - ks * 9;
                if (st + WR - 1 < 36) load_a(st + WR - 1, (st + WR - 1) % WR);
                if (tap == 4 && ks + 1 < TH / 2) load_b(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int fj = 0; fj < 2; ++fj) wacc[tap][fi][fj] = HX::mfma16(afrag[st % WR][fi], bfrag[ks & 1][fj], wacc[tap][fi][fj]);
                __builtin_amdgcn_sched_barrier(0);
            }
            WM_STAMP(0)
            // the a tile of tile + 1 AFTER the loop: a W wave is done long before its D partner (144 MFMAs against 144 + the dy staging + the
            // epilogue), the fragment registers are free now, and nothing of it sits on the tile's critical path
            if constexpr (stage) {
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    a_consts(pq);
#pragma unroll
                    for (int k = 0; k < AV; ++k) a_transform(k, pq);
                }
#pragma unroll
                for (int k = 0; k < AV; ++k) a_publish(nxt, k);
                if constexpr (refill) {
                    const TileGeo g2 = geo(tile + 2);
#pragma unroll
                    for (int k = 0; k < AV; ++k) load_a_slot(g2, k);
                }
            }
            WM_STAMP(1)
            __syncthreads();
            WM_STAMP(2)
        };
        WM_STAMP(3)
        {
            int tile = t_begin;
            for (; tile + 2 < t_end; ++tile) w_tile(tile, yes{}, yes{});
            if (tile + 1 < t_end) { w_tile(tile, yes{}, no{}); ++tile; }
            if (tile < t_end) w_tile(tile, no{}, no{});
        }
        // weight-gradient slab [tap][ci][co] (wgrad.hip's reduction): wacc[tap] is dW of filter tap 8 - tap
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
#ifdef WM_DEBUG
        if (a.stamps) {
            __builtin_amdgcn_s_waitcnt(0);
            WM_STAMP(3)
            if (lane == 0) {
                long long* o = reinterpret_cast<long long*>(a.stat + (size_t)blockIdx.x * 2 * C) + (tid >> 6) * 4;
                o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3];
            }
        }
#endif
        __syncthreads();   // (the D role's final barrier: its partial sums)
        return;
    }

    // ================================================================================================================ D role
    // (s_setprio for either role measured nothing: D at 2 152.4-152.9 us against 152.7-153.1, W at 2 155.6-155.9 -- the SIMD's vector issue
    // port is what both roles compete for, not the arbitration order: tools/phase_bwd8.py)
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(GVEC ? a.y : a.g), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.y), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<hx_t*>(a.xr), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, nbytes, 0x00020000);
    // Slots: k < 5 cover the 10 x 16 pixels of the halo tile's columns 2..17, two rows per slot (slot k = slot 0's pixel, 2 k rows further
    // down: ONE LDS offset + an immediate, ONE global offset + a wave-uniform addend instead of six of each); k == 5 the two columns 0..1,
    // live in the first 20 slot rows.  (The split is also what a reuse of the shared columns / rows needs: tools/patches.)
    int hlds0, hlds5;
    unsigned hofs0, hofs5, edge = 0;
    {
        const int px0 = (slot & 15) + 2, py0 = slot >> 4;
        hlds0 = (py0 * HW + px0) * 128 + ((vec ^ fsw(px0)) << 4);
        hofs0 = (unsigned)(((py0 * a.W + px0) * C + vec * 8) * 2);
        const int e = min(slot, 19), erow = e >> 1, ecol = e & 1;
        hlds5 = (erow * HW + ecol) * 128 + ((vec ^ fsw(ecol)) << 4);
        hofs5 = (unsigned)(((erow * a.W + ecol) * C + vec * 8) * 2);
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int py = k < XV - 1 ? py0 + 2 * k : erow, px = k < XV - 1 ? px0 : ecol;
            edge |= (py == 0 ? 1u : 0u) << k | (py == HH - 1 ? 1u : 0u) << (k + 6) | (px == 0 ? 1u : 0u) << (k + 12) | (px == HW - 1 ? 1u : 0u) << (k + 18);
        }
    }
    auto hl = [&](int k) { return k < XV - 1 ? hlds0 + k * (2 * HW * 128) : hlds5; };
    const bool last_live = slot < 20;
    auto inside_bits = [&](const TileGeo& t) {
        const unsigned sel = (t.ty0 == 0 ? 0x3fu : 0u) | (t.ty0 + TH == a.H ? 0x3fu << 6 : 0u) | (t.tx0 == 0 ? 0x3fu << 12 : 0u) | (t.tx0 + TW == a.W ? 0x3fu << 18 : 0u);
        const unsigned e = edge & sel;
        return ~(e | (e >> 6) | (e >> 12) | (e >> 18)) & 0x3fu;
    };
    hx8 dG[XV], dY[XV];
    unsigned okh = 0;
    auto load_dy_slot = [&](const TileGeo& t, int k) {
        const unsigned o = (k < XV - 1 ? hofs0 : hofs5) + (halo_base(t) + (unsigned)(k < XV - 1 ? k * 2 * a.W * C * 2 : 0));
        if constexpr (!GVEC) dG[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsG, o, 0, 0));
        dY[k] = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rsY, o, 0, 0));
    };
    f32x4 pka, pkb;              // the channel pair's {scale, shift, ca, k2} of both channels
    float pk3a = 0.f, pk3b = 0.f;
    f32x2 pkg = {0.f, 0.f};     // GVEC: k3 + ca * gvec of the pair, for the sample of the tile being published
    int bpub = 0;
    auto d_consts = [&](int pq) {
        if constexpr (GVEC) pkg = *reinterpret_cast<const f32x2*>(sG + bpub * C + vec * 8 + 2 * pq);
        const float* kp = sK + (vec * 8 + 2 * pq) * 8 + vec * 4;
        pka = *reinterpret_cast<const f32x4*>(kp); pkb = *reinterpret_cast<const f32x4*>(kp + 8);
        pk3a = kp[4]; pk3b = kp[12];
    };
    auto d_transform = [&](int k, int pq) {
        u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
        const u32x4 wy = __builtin_bit_cast(u32x4, dY[k]);
        float d0, d1;
        if constexpr (GVEC) {
            d0 = wm_bn_fold_dy(HX::lo(wy[pq]), pka[0], pka[1], pka[3], pk3a, pkg[0]);
            d1 = wm_bn_fold_dy(HX::hi(wy[pq]), pkb[0], pkb[1], pkb[3], pk3b, pkg[1]);
        } else if constexpr (PREMASKED) {
            d0 = __builtin_fmaf(pka[2], HX::lo(w[pq]), __builtin_fmaf(-pka[3], HX::lo(wy[pq]), pk3a));
            d1 = __builtin_fmaf(pkb[2], HX::hi(w[pq]), __builtin_fmaf(-pkb[3], HX::hi(wy[pq]), pk3b));
        } else {
            d0 = wm_bn_fold_dyg(HX::lo(wy[pq]), HX::lo(w[pq]), pka[0], pka[1], pka[2], pka[3], pk3a);
            d1 = wm_bn_fold_dyg(HX::hi(wy[pq]), HX::hi(w[pq]), pkb[0], pkb[1], pkb[2], pkb[3], pk3b);
        }
        const hx2 pk = HX::pack2(d0, d1);
        w[pq] = __builtin_bit_cast(unsigned, pk);
        if constexpr (GVEC) dY[k] = __builtin_bit_cast(hx8, w); else dG[k] = __builtin_bit_cast(hx8, w);
    };
    auto d_publish = [&](unsigned char* buf, int k) {
        u32x4 w = __builtin_bit_cast(u32x4, GVEC ? dY[k] : dG[k]);
        const unsigned keep = 0u - ((okh >> k) & 1u);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) w[q4] &= keep;
        if (k + 1 < XV || last_live) *reinterpret_cast<u32x4*>(buf + hl(k)) = w;
    };
    // input-gradient fragments: tile rows 2*wave, 2*wave + 1; lane (p, q): pixel column p, 16 channels [16q, 16q + 16)
    const int p = lane & 15, q = lane >> 4;
    int aoff[3][2], boff[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[kw][ks] = ((wave * 2 * HW + p + kw) * 128) + (((ks * 4 + q) ^ fsw(p + kw)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (p * C + swzw(p, ks * 4 + q) * 8) * 2;
    const unsigned eofs = (unsigned)(((wave * 2 * a.W + p) * C + 16 * q) * 2);
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }

    if (t_begin < t_end) {
        const TileGeo g0 = geo(t_begin);
#pragma unroll
        for (int k = 0; k < XV; ++k) load_dy_slot(g0, k);
        okh = inside_bits(g0);
        if constexpr (GVEC) bpub = g0.b;
    }
    finish_prologue();
    if (t_begin < t_end) {
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            d_consts(pq);
#pragma unroll
            for (int k = 0; k < XV; ++k) d_transform(k, pq);
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) d_publish(sBuf, k);
    }
    __syncthreads();   // first tile visible
    if (t_begin + 1 < t_end) {
        const TileGeo g1 = geo(t_begin + 1);
#pragma unroll
        for (int k = 0; k < XV; ++k) load_dy_slot(g1, k);
        okh = inside_bits(g1);
    }
    // one tile: 18 K-steps = 36 half-steps of 4 MFMAs; staging units (the dy halo of tile + 1) between them: half-step h < 32: channel pair
    // h / 8 -- constants at h % 8 == 0, slot k's transform at h % 8 == 1 + k (k < 6); half-steps 32..35: publish + refill slots {0,1},{2,3},{4},{5}
    auto d_tile = [&](int tile, auto stage_c, auto refill_c) __attribute__((always_inline)) {
        constexpr bool stage = decltype(stage_c)::value, refill = decltype(refill_c)::value;
        const unsigned char* cur = sBuf + ((tile - t_begin) & 1) * BUF_BYTES;
        unsigned char* nxt = sBuf + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
        hx8 pix[2][2], filA[2][2], filB[2][2];
        auto load_pix = [&](int sidx, int buf) {
            const int tap = sidx >> 1, ks = sidx & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) pix[buf][ml] = *reinterpret_cast<const hx8*>(cur + aoff[kw][ks] + (ml + kh) * (HW * 128));
        };
        auto load_fil = [&](int sidx, int half, hx8 (&f)[2]) {
            const int tap = sidx >> 1, ks = sidx & 1;
#pragma unroll
            for (int n = 0; n < 2; ++n)
                f[n] = *reinterpret_cast<const hx8*>(reinterpret_cast<const char*>(sW) + boff[ks] + (tap * C + (2 * half + n) * 16) * (C * 2));
        };
        load_pix(0, 0);
        load_fil(0, 0, filA[0]);
        load_fil(0, 1, filB[0]);
        // the tile's bookkeeping and the request of its epilogue operand, in the shadow of the first fragment reads
        const TileGeo g = geo(tile);
        const TileGeo g2 = geo(refill ? tile + 2 : tile);
        unsigned okn = 0;
        if constexpr (refill) okn = inside_bits(g2);
        if constexpr (GVEC && stage) bpub = geo(tile + 1).b;
        unsigned ryv[2][8], eo[2];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            eo[ml] = halo_base(g) + (unsigned)((a.W + 1) * C * 2) + eofs + (unsigned)(ml * a.W * C * 2);
            const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, eo[ml], 0, 0), t1 = __builtin_amdgcn_raw_buffer_load_b128(rsX, eo[ml] + 16u, 0, 0);
            ryv[ml][0] = t0[0]; ryv[ml][1] = t0[1]; ryv[ml][2] = t0[2]; ryv[ml][3] = t0[3];
            ryv[ml][4] = t1[0]; ryv[ml][5] = t1[1]; ryv[ml][6] = t1[2]; ryv[ml][7] = t1[3];
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[ml][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto units = [&](int h) __attribute__((always_inline)) {   // after the 4 MFMAs of half-step h
            if constexpr (stage) {
                if (h < 32) { if ((h & 7) >= 1 && (h & 7) <= XV) d_transform((h & 7) - 1, h >> 3); }
                else {
                    const int k0 = h == 32 ? 0 : h == 33 ? 2 : h == 34 ? 4 : 5, nk = h < 34 ? 2 : 1;
#pragma unroll
                    for (int k = k0; k < k0 + nk; ++k) {
                        d_publish(nxt, k);
                        if constexpr (refill) { load_dy_slot(g2, k); okh = (okh & ~(1u << k)) | (okn & (1u << k)); }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
            }
        };
#pragma unroll
        for (int sidx = 0; sidx < 18; ++sidx) {
            const int cb = sidx & 1;
            if constexpr (stage) { if (2 * sidx < 32 && ((2 * sidx) & 7) == 0) d_consts((2 * sidx) >> 3); }
            if (sidx + 1 < 18) { load_pix(sidx + 1, cb ^ 1); load_fil(sidx + 1, 0, filA[cb ^ 1]); load_fil(sidx + 1, 1, filB[cb ^ 1]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[ml][n] = HX::mfma16(filA[cb][n], pix[cb][ml], acc[ml][n]);
            units(2 * sidx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[ml][2 + n] = HX::mfma16(filB[cb][n], pix[cb][ml], acc[ml][2 + n]);
            units(2 * sidx + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        WM_STAMP(0)
        // The tile's barrier sits HERE, in front of the epilogue: every read of this tile's LDS buffer and every staging write into the next
        // one is done, and the epilogue touches neither (accumulators, its operand registers, the constant table) -- so the W waves start
        // the next tile's weight-gradient MFMAs while this role still masks, packs and sums: VALU work beside the partner's matrix work
        __syncthreads();
        WM_STAMP(1)
        // epilogue: the feeding layer's BatchNorm-backward sums (gz = dx * [z > 0], dx rounded as stored), pack, store
        f32x4 rsv[4], rhv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            rsv[jj] = *reinterpret_cast<const f32x4*>(sTab + 16 * q + 4 * jj);
            rhv[jj] = *reinterpret_cast<const f32x4*>(sTab + C + 16 * q + 4 * jj);
        }
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            unsigned pk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nf = j >> 1, i0 = 2 * (j & 1), jj = j >> 1, jh = j & 1;
                const float y0 = HX::lo(ryv[ml][j]), y1 = HX::hi(ryv[ml][j]);
                const float z0 = __builtin_fmaf(rsv[jj][2 * jh], y0, rhv[jj][2 * jh]), z1 = __builtin_fmaf(rsv[jj][2 * jh + 1], y1, rhv[jj][2 * jh + 1]);
                const hx2 p2v = HX::pack2((z0 > 0.f ? acc[ml][nf][i0] : 0.f), (z1 > 0.f ? acc[ml][nf][i0 + 1] : 0.f));
                pk[j] = __builtin_bit_cast(unsigned, p2v);   // dx leaves masked: gz, not g
                const float gz0 = HX::lo(pk[j]), gz1 = HX::hi(pk[j]);
                s1[2 * j] += gz0; s1[2 * j + 1] += gz1;
                s2[2 * j] = __builtin_fmaf(gz0, y0, s2[2 * j]);
                s2[2 * j + 1] = __builtin_fmaf(gz1, y1, s2[2 * j + 1]);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0], pk[1], pk[2], pk[3]}, rsD, eo[ml], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[4], pk[5], pk[6], pk[7]}, rsD, eo[ml] + 16u, 0, 0);
        }
        WM_STAMP(2)
    };
    WM_STAMP(3)
    {
        int tile = t_begin;
        for (; tile + 2 < t_end; ++tile) d_tile(tile, yes{}, yes{});
        if (tile + 1 < t_end) { d_tile(tile, yes{}, no{}); ++tile; }
        if (tile < t_end) d_tile(tile, no{}, no{});
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
#ifdef WM_DEBUG
    if (a.stamps) {
        WM_STAMP(3)
        if (lane == 0) {
            long long* o = reinterpret_cast<long long*>(a.stat + (size_t)blockIdx.x * 2 * C) + (tid >> 6) * 4;
            o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3];
        }
        return;
    }
#endif
    if (tid < 2 * C) {
        const int which = tid / C, n = tid - which * C;
        a.stat[((size_t)blockIdx.x * 2 + which) * C + n] =
            (sRed[(0 * 2 + which) * C + n] + sRed[(1 * 2 + which) * C + n]) + (sRed[(2 * 2 + which) * C + n] + sRed[(3 * 2 + which) * C + n]);
    }
}

}  // namespace

int WM_HSYM(wm_launch_bwd_ws8)(const void* g, const void* y, const float* stats4, int st_ld, const float* coef, const void* wpt, const void* xr,
                                const float* in_scale, const float* in_shift, void* dx, float* stat, float* ws, int B, int H, int W, int nwg,
                                int reverse, hipStream_t s, int premasked, const float* gvec, int gv_ld, int stamps) {
    Bwd8Args a;
    a.stamps = stamps;
    a.g = (const hx_t*)g; a.y = (const hx_t*)y; a.stats4 = stats4; a.st_ld = st_ld; a.coef = coef; a.wpt = (const hx_t*)wpt;
    a.gvec = gvec; a.gv_ld = gv_ld;
    a.xr = (const hx_t*)xr; a.in_scale = in_scale; a.in_shift = in_shift; a.dx = (hx_t*)dx; a.stat = stat; a.ws = ws;
    a.B = B; a.H = H; a.W = W; a.tilesX = W / TW; a.tilesY = H / TH; a.ntiles = B * a.tilesX * a.tilesY;
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
    a.mX = magic(a.tilesX); a.mY = magic(a.tilesY); a.m2X = magic(2 * a.tilesX);
    a.tq = a.ntiles / nwg; a.trem = a.ntiles % nwg;
    a.reverse = wm_sweep_dir(reverse);
    // (the unmasked-gradient form is not instantiated: it needs ~15 registers more than a two-waves-per-SIMD kernel has and stays on
    // bwd_ws.hip -- wgrad.hip dispatches.  Both instantiated forms: 250 registers, no scratch.)  An unmasked tensor gradient handed to
    // this launcher would be staged WITHOUT its ReLU mask: refused here, whatever the caller's dispatch condition says
    if (!gvec && !premasked) return WM_E_SHAPE;
    if (gvec) hipLaunchKernelGGL((bwd_ws8_kernel<false, true>), dim3((unsigned)nwg), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((bwd_ws8_kernel<true, false>), dim3((unsigned)nwg), dim3(512), 0, s, a);
    return WM_OK;
}
