// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on MFMA (gfx950), forward + dgrad.
//
// Reference op: nn.Conv2d(cin, cout, 3, 1, padding=1) in hidden_models/conv_bn_relu.py:11 and
// network/UNet.py:67-97 (followed there by BatchNorm2d + ReLU, which this kernel helps fuse:
// the *consumer* applies relu(scale*x+shift) of the producer while staging its input tile, and
// the *producer* emits the per-channel sum / sum-of-squares of its f32 accumulators).
//
// GEMM view: M = B*H*W output pixels, N = Cout, K = 9*Cin.  NHWC activations.
//   workgroup  = 256 threads (4 waves) -> 16x16 output pixels x BN output channels
//   wave       = 4 tile rows (64 pixels) x BN channels = 2 x (BN/32) MFMA 32x32 accumulators
//   K loop     = Cin chunks of CK channels; per chunk the 18x18xCK input halo tile and the
//                [9][BN][CK] weight slab are staged into LDS once and reused by all 9 taps
//                (each input element is fetched from HBM once per Cout tile, +27% halo).
//   bf16 path  : v_mfma_f32_32x32x16_bf16, CK = 32, f32 accumulate
//   f32 path   : v_mfma_f32_32x32x2_f32 (exact f32 FMA chain), CK = 16 -- the parity path
// LDS rows (one pixel's CK channels / one filter row) are padded by 16 B to an 80-byte stride:
// 16 consecutive pixels then hit 16 distinct 16-byte bank slots for ds_read_b128.
// The epilogue restages the accumulators through LDS so the global stores are 16 B per lane
// and contiguous over a pixel's channels (128 B for 64 bf16 channels).
#include <stdlib.h>
#include <type_traits>
#include "wm_common.h"

// wave-specialised {64,16}->64 kernel (conv3x3_ws.hip)
int wm_launch_conv3x3_ws(const void* x, int ldx, int Cin, int CoutP, const void* wp, const float* bias, int nbias, const float* in_scale,
                         const float* in_shift, void* y, float* stat, int B, int H, int W, int wgs, int tiles_per_wg,
                         hipStream_t s);

namespace {

constexpr int TH = 16, TW = 16;
constexpr int HH = TH + 2, HW = TW + 2;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int CK = 32;   // channels per K chunk
    static constexpr int VE = 8;    // elements per 16-byte vector
    static constexpr int PS = 40;   // LDS row stride in elements (80 B)
};
template <> struct Cfg<float> {
    static constexpr int CK = 16;
    static constexpr int VE = 4;
    static constexpr int PS = 20;   // 80 B
};

template <typename T>
struct ConvArgs {
    const T* x;
    int ldx;
    const T* wp;          // [9][CoutP][Cin]
    const float* bias;    // [nbias] or null
    int nbias;
    const float* in_scale;
    const float* in_shift;
    T* y;
    int ldy;
    float* stat;          // [gridDim.x][2][CoutP] or null
    int B, H, W, Cin, CoutP;
    int tilesX, tilesY;
};

template <typename T> __device__ __forceinline__ void zero_vec(vec16<T>& v) {
#pragma unroll
    for (int i = 0; i < vec16<T>::N; ++i) v.set(i, 0.f);
}

template <typename T, int BN, bool XFORM>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs<T> a) {
    constexpr int CK = Cfg<T>::CK, VE = Cfg<T>::VE, PS = Cfg<T>::PS;
    constexpr int NF = BN / 32;
    constexpr int A_ELEMS = HH * HW * PS;
    constexpr int B_ELEMS = 9 * BN * PS;
    constexpr int OPS = BN + 16 / (int)sizeof(T);          // output staging row stride (elements)
    constexpr int MAIN_BYTES = (A_ELEMS + B_ELEMS) * (int)sizeof(T);
    constexpr int OUT_BYTES = TH * TW * OPS * (int)sizeof(T);
    constexpr int LDS_BYTES = MAIN_BYTES > OUT_BYTES ? MAIN_BYTES : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + 4 * 2 * BN * 4];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + A_ELEMS;
    T* sOut = reinterpret_cast<T*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + LDS_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile assignment: workgroups b and b+8 share an XCD (round-robin dispatch); XCD x takes the
    // contiguous tiles [x*G/8, (x+1)*G/8) so that neighbouring tiles share their halo in one L2
    const int G = gridDim.x;
    int t = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int txi = t % a.tilesX; t /= a.tilesX;
    const int tyi = t % a.tilesY; t /= a.tilesY;
    const int b = t;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[2][NF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int vec = tid & 3;  // this thread's 16-byte group inside a CK chunk (stride 256 keeps it fixed)
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        const int kvalid = min(CK, a.Cin - c0);
        const int cbase = c0 + vec * VE;
        const bool cok = vec * VE < kvalid;
        float sc[VE], sh[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) { sc[i] = 0.f; sh[i] = 0.f; }
        if (XFORM && cok) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { sc[i] = a.in_scale[cbase + i]; sh[i] = a.in_shift[cbase + i]; }
        }
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the input halo tile: 18x18 pixels x CK channels
        const int cload = cok ? cbase : 0;  // loads are unconditional (clamped address); validity is applied afterwards
#pragma unroll
        for (int it = 0; it < (HH * HW * 4 + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int pix = min(i >> 2, HH * HW - 1);
            const int py = pix / HW, px = pix - py * HW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
            const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            vec16<T> v = *reinterpret_cast<const vec16<T>*>(a.x + ((size_t)(b * a.H + gyc) * a.W + gxc) * a.ldx + cload);
            const bool inb = cok && gy == gyc && gx == gxc;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float f = v.get(e);
                if (XFORM) f = fmaxf(sc[e] * f + sh[e], 0.f);
                v.set(e, inb ? f : 0.f);  // zero padding is applied AFTER the fused BN+ReLU, as the reference pads the activated map
            }
            if (i < HH * HW * 4) *reinterpret_cast<vec16<T>*>(sA + pix * PS + vec * VE) = v;
        }
        // ---- stage the weight slab [9][BN][CK]
#pragma unroll
        for (int it = 0; it < (9 * BN * 4 + 255) / 256; ++it) {
            const int i = tid + it * 256;
            const int row = min(i >> 2, 9 * BN - 1);  // tap*BN + n
            const int tap = row / BN, n = row - tap * BN;
            vec16<T> v = *reinterpret_cast<const vec16<T>*>(a.wp + ((size_t)tap * a.CoutP + n0 + n) * a.Cin + cload);
            if (!cok) zero_vec(v);
            if (i < 9 * BN * 4) *reinterpret_cast<vec16<T>*>(sB + row * PS + vec * VE) = v;
        }
        __syncthreads();
        // ---- 9 taps x K steps of MFMA
        if constexpr (sizeof(T) == 2) {
            const int ksteps = (kvalid + 15) >> 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                for (int ks = 0; ks < ksteps; ++ks) {
                    bf16x8 af[2], bfr[NF];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const int py = wave * 4 + mf * 2 + (r >> 4), px = r & 15;
                        af[mf] = *reinterpret_cast<const bf16x8*>(sA + ((py + kh) * HW + px + kw) * PS + ks * 16 + h * 8);
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        bfr[nf] = *reinterpret_cast<const bf16x8*>(sB + (tap * BN + nf * 32 + r) * PS + ks * 16 + h * 8);
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mf], bfr[nf], acc[mf][nf], 0, 0, 0);
                }
            }
        } else {
            const int ksteps = kvalid >> 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                for (int ks = 0; ks < ksteps; ++ks) {
                    float af[2], bfr[NF];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const int py = wave * 4 + mf * 2 + (r >> 4), px = r & 15;
                        af[mf] = sA[((py + kh) * HW + px + kw) * PS + ks * 2 + h];
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) bfr[nf] = sB[(tap * BN + nf * 32 + r) * PS + ks * 2 + h];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mf], bfr[nf], acc[mf][nf], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // all waves done with sA/sB before the output tile overwrites them

    // ---- epilogue: bias, batch statistics, restage through LDS, coalesced store
    // accumulator element i of lane (r,h): pixel row-in-fragment = (i&3) + 8*(i>>2) + 4*h, column n = r
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const float bv = (a.bias && n0 + nf * 32 + r < a.nbias) ? a.bias[n0 + nf * 32 + r] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int prow = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int py = wave * 4 + mf * 2 + (prow >> 4), px = prow & 15;
                const float v = acc[mf][nf][i] + bv;
                const bool ok = (ty0 + py < a.H) && (tx0 + px < a.W);
                if (ok) { s1 += v; s2 += v * v; }
                sOut[(py * TW + px) * OPS + nf * 32 + r] = from_f32<T>(v);
            }
        }
        if (a.stat) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                sRed[(wave * 2 + 0) * BN + nf * 32 + r] = s1;
                sRed[(wave * 2 + 1) * BN + nf * 32 + r] = s2;
            }
        }
    }
    __syncthreads();
    if (a.stat && tid < 2 * BN) {
        const int which = tid / BN, n = tid - which * BN;
        const float s = sRed[(0 * 2 + which) * BN + n] + sRed[(1 * 2 + which) * BN + n] +
                        sRed[(2 * 2 + which) * BN + n] + sRed[(3 * 2 + which) * BN + n];
        a.stat[((size_t)blockIdx.x * 2 + which) * a.CoutP + n0 + n] = s;
    }
    constexpr int VPP = BN / VE;  // 16-byte vectors per pixel
    for (int i = tid; i < TH * TW * VPP; i += 256) {
        const int pix = i / VPP, vv = i - pix * VPP;
        const int py = pix / TW, px = pix - py * TW;
        const int gy = ty0 + py, gx = tx0 + px;
        if (gy < a.H && gx < a.W) {
            const vec16<T> v = *reinterpret_cast<const vec16<T>*>(sOut + pix * OPS + vv * VE);
            *reinterpret_cast<vec16<T>*>(a.y + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldy + n0 + vv * VE) = v;
        }
    }
}

// =====================================================================================================
// Specialisation for the layers that dominate the step: bf16, Cin = Cout = 64 (all body layers of the
// HiDDeN encoder / decoder / discriminator, forward and dgrad).  Differences from the generic kernel:
//   * persistent workgroups (one per CU): the whole [9][64][64] filter (83 KB with row padding) is staged
//     into LDS ONCE and stays resident while the workgroup walks a contiguous run of 16x16 tiles
//     (one image row of tiles at 256x256, so consecutive tiles share their halo columns in L1/L2);
//   * all 64 input channels are one K chunk: 36 (tap, k-step) iterations of 4 ds_read_b128 + 4 MFMA 32x32x16,
//     software-pipelined by hand (fragments of step s+1 are read before the MFMAs of step s): the loop runs
//     at 97 % of the MFMA issue rate (4742 cycles for 144 MFMAs per wave, tools/phase_c64.py);
//   * HBM traffic is issued FROM INSIDE that loop, one vector-memory instruction every 2-3 steps: the 11
//     halo loads of tile t+2 and the 8 output stores of tile t-1.  A CU can only move ~10-12 B/clk, so a wave
//     that issues its 19 KB in one burst blocks for ~1,500 cycles per 20 instructions while the matrix pipe
//     idles (measured: 7,600 of 13,800 cycles per tile); interleaved, the queue never fills and the memory
//     time hides under the 4.7k-cycle MFMA loop.  Loads run two tiles ahead so their data has a full loop to
//     arrive before the register copy that consumes it;
//   * outputs: channel n sits at filter row (n&1)*32+(n>>1), so a lane owns channels (2r, 2r+1) of its pixels;
//     the tile is restaged through LDS (aliasing the halo tile) as 32-bit pairs and leaves as 16-byte stores
//     (store ISSUE is the cost: 8 dwordx4 instead of 32 dword per lane);
//   * BatchNorm partial sums are carried in registers across the workgroup's tiles: one partial row per
//     workgroup (256 rows instead of one per tile).
// LDS: filter 82,944 B + halo tile 46,656 B + output staging tile 32,768 B = 162,368 B (of 163,840);
// three workgroup barriers per tile.
constexpr int C64 = 64;
constexpr int PS64 = 72;  // 144-byte rows: 16 consecutive pixels/filters hit 16 distinct 16-byte bank slots
constexpr int XV = (HH * HW * 8 + 255) / 256;  // halo vectors per thread (11)

template <bool XFORM, bool STATS, bool STAMPS = false>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(ConvArgs<bf16_t> a, const bf16_t* __restrict__ xin,
                                                           bf16_t* __restrict__ yout, int ntiles, int tiles_per_wg,
                                                           unsigned long long* __restrict__ stamps = nullptr) {
    // STAMPS: diagnostic build only (tools/phase_c64.py): per-wave cycle totals of the phases of the tile loop
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int which) {
        if (STAMPS) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (which >= 0) ph[which] += tnow - tprev;
            tprev = tnow;
        }
    };
    constexpr int SW_BYTES = 9 * C64 * PS64 * 2, SX_BYTES = HH * HW * PS64 * 2, SO_BYTES = TH * TW * C64 * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_BYTES + SX_BYTES + SO_BYTES];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sX = reinterpret_cast<bf16_t*>(smem + SW_BYTES);
    bf16_t* sOut = reinterpret_cast<bf16_t*>(smem + SW_BYTES + SX_BYTES);  // [256 px][64 ch], dense
    float* sRed = reinterpret_cast<float*>(smem + SW_BYTES + SX_BYTES);     // aliases sOut after the last tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int vec = tid & 7;  // this thread's 8-channel group (fixed: 256 % 8 == 0)

    // ---- filter [9][64 co][64 ci] -> LDS, all loads in flight at once; output channel n -> row (n&1)*32+(n>>1)
    {
        constexpr int WV = 9 * C64 * 8 / 256;  // 18 vectors per thread
        bf16x8 wv[WV];
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 256 * k;
            wv[k] = *reinterpret_cast<const bf16x8*>(a.wp + (size_t)(i >> 3) * C64 + (i & 7) * 8);
        }
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int i = tid + 256 * k;
            const int row = i >> 3, tap = row >> 6, n = row & 63;
            *reinterpret_cast<bf16x8*>(sW + (tap * C64 + (n & 1) * 32 + (n >> 1)) * PS64 + (i & 7) * 8) = wv[k];
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (XFORM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale[vec * 8 + e]; sh[e] = a.in_shift[vec * 8 + e]; }
    }
    float bv[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) bv[nf] = (a.bias && 2 * r + nf < a.nbias) ? a.bias[2 * r + nf] : 0.f;
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};

    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    if (t_begin >= t_end) return;

    // ---- halo loads: (tile-uniform base) + (per-lane byte offset).  voff[] is the lane's offset for a halo
    // that lies inside the image; the address is always clamped into the image (branch-free: a load under a
    // branch gets duplicated and serialised by the compiler) and validity travels as a bit.
    int voff[XV];
#pragma unroll
    for (int k = 0; k < XV; ++k) {
        const int pix = min((tid + 256 * k) >> 3, HH * HW - 1);
        const int py = pix / HW, px = pix - py * HW;
        voff[k] = ((py * a.W + px) * a.ldx + vec * 8) * 2;
    }
    struct TileGeo { int b, ty0, tx0; };
    auto geo = [&](int tile) {
        TileGeo g;
        int t = tile;
        const int txi = t % a.tilesX; t /= a.tilesX;
        const int tyi = t % a.tilesY; t /= a.tilesY;
        g.b = t; g.ty0 = tyi * TH; g.tx0 = txi * TW;
        return g;
    };
    auto load_one = [&](const TileGeo& g, int k, bf16x8& dst, unsigned& okbits) {
        const int pix = min((tid + 256 * k) >> 3, HH * HW - 1);
        const int py = pix / HW, px = pix - py * HW;
        const int gy = g.ty0 - 1 + py, gx = g.tx0 - 1 + px;
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        const int off = voff[k] + (((gyc - gy) * a.W + (gxc - gx)) * a.ldx) * 2;
        const char* tb = reinterpret_cast<const char*>(xin) + (((long)g.b * a.H + g.ty0 - 1) * a.W + g.tx0 - 1) * a.ldx * 2;
        dst = *reinterpret_cast<const bf16x8*>(tb + off);
        const unsigned okb = (gy == gyc && gx == gxc && ((tid + 256 * k) >> 3) < HH * HW) ? 1u : 0u;
        okbits |= okb << k;
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // fused BN+ReLU of one halo vector (packed f32 math on bf16 pairs) + zero padding AFTER the activation
    auto transform = [&](bf16x8& v, bool ok) {
        u32x4 w = __builtin_bit_cast(u32x4, v);
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
        for (int q = 0; q < 4; ++q) w[q] = ok ? w[q] : 0u;
        v = __builtin_bit_cast(bf16x8, w);
    };

    bf16x8 cur[XV], nxt[XV];      // halo data (already transformed) of the tile to write next / raw data of the one after
    unsigned cur_ok = 0, nxt_ok = 0;
    {
        const TileGeo g0 = geo(t_begin);
#pragma unroll
        for (int k = 0; k < XV; ++k) load_one(g0, k, cur[k], cur_ok);
        if (t_begin + 1 < t_end) {
            const TileGeo g1 = geo(t_begin + 1);
#pragma unroll
            for (int k = 0; k < XV; ++k) load_one(g1, k, nxt[k], nxt_ok);
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) transform(cur[k], (cur_ok >> k) & 1u);
    }

    f32x16 accp[2][2];            // accumulators of the previous tile, packed during the current MFMA loop
    TileGeo gp = geo(t_begin);
    bool have_p = false;
    bf16x8 outv[8];               // output of the tile before that, as 16-byte vectors, stored during the loop
    bf16_t* out_ptr[8];
    unsigned out_ok = 0;
    bool have_out = false;

    // bias, BatchNorm partial sums, bf16 pair -> output staging tile, for accumulator element (mf, i)
    auto pack_item = [&](const f32x16 (&ac)[2][2], const TileGeo& gq, int mf, int i, bool masked) {
        const int prow = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int py = wave * 4 + mf * 2 + (prow >> 4), px = prow & 15;
        const float v0 = ac[mf][0][i] + bv[0], v1 = ac[mf][1][i] + bv[1];
        if (STATS) {
            float m = 1.f;
            if (masked) m = ((gq.ty0 + py < a.H) & (gq.tx0 + px < a.W)) ? 1.f : 0.f;
            st1[0] += m * v0; st2[0] += m * v0 * v0;
            st1[1] += m * v1; st2[1] += m * v1 * v1;
        }
        const bf16x2 pk = {(bf16_t)v0, (bf16_t)v1};
        *reinterpret_cast<bf16x2*>(sOut + (py * TW + px) * C64 + 2 * r) = pk;
    };
    auto fetch_out = [&](const TileGeo& gq) {   // staging tile -> 16-byte vectors + their global addresses
        out_ok = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = tid + 256 * k;
            const int pix = i >> 3, vv = i & 7;
            const int gy = gq.ty0 + (pix >> 4), gx = gq.tx0 + (pix & 15);
            outv[k] = *reinterpret_cast<const bf16x8*>(sOut + pix * C64 + vv * 8);
            out_ptr[k] = yout + (((size_t)gq.b * a.H + min(gy, a.H - 1)) * a.W + min(gx, a.W - 1)) * C64 + vv * 8;
            const unsigned okb = (gy < a.H && gx < a.W) ? 1u : 0u;
            out_ok |= okb << k;
        }
    };

    for (int tile = t_begin; tile < t_end; ++tile) {
        const TileGeo g = geo(tile);
        stamp(-1);
        __syncthreads();  // A: all MFMA reads of the previous halo tile are done
        stamp(0);
#pragma unroll
        for (int k = 0; k < XV; ++k)
            if (((tid + 256 * k) >> 3) < HH * HW)
                *reinterpret_cast<bf16x8*>(sX + ((tid + 256 * k) >> 3) * PS64 + vec * 8) = cur[k];
        stamp(1);
        __syncthreads();  // B
        stamp(2);
        // rotate the prefetch registers: nxt (loaded during the previous MFMA loop, long since arrived) -> cur
#pragma unroll
        for (int k = 0; k < XV; ++k) cur[k] = nxt[k];
        cur_ok = nxt_ok;
        nxt_ok = 0;
        const bool have1 = tile + 1 < t_end, have2 = tile + 2 < t_end;
        const TileGeo g2 = geo(have2 ? tile + 2 : tile);
        stamp(3);

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        const int apix0 = (wave * 4 + (r >> 4)) * HW + (r & 15);
        bf16x8 af[2][2], bfr[2][2];
        auto load_frags = [&](int sidx, int buf) {
            const int tap = sidx >> 2, ks = sidx & 3;
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
                af[buf][mf] = *reinterpret_cast<const bf16x8*>(sX + (apix0 + (mf * 2 + kh) * HW + kw) * PS64 + ks * 16 + h * 8);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                bfr[buf][nf] = *reinterpret_cast<const bf16x8*>(sW + (tap * C64 + nf * 32 + r) * PS64 + ks * 16 + h * 8);
        };
        // 36 (tap, k-step) iterations.  Besides its 4 ds_read_b128 + 4 MFMA each iteration carries a slice of the
        // other work of the pipeline:  halo load of tile+2 (11x) | output store of tile-2 (8x) | BN+ReLU
        // transform of the tile+1 halo registers (11x) | bias/statistics/pack of one accumulator element of
        // tile-1 (32x).  The steady-state instance (all four streams active, every tile full) is branch-free:
        // ~90 scalar branches per tile otherwise break the instruction stream.
        const bool all_full = (a.H % TH == 0) && (a.W % TW == 0);
        auto mfma_loop = [&](auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            load_frags(0, 0);
#pragma unroll
            for (int sidx = 0; sidx < 36; ++sidx) {
                const int cb = sidx & 1;
                if (sidx + 1 < 36) load_frags(sidx + 1, cb ^ 1);
                if (sidx % 3 == 1 && sidx / 3 < XV) {
                    if (STEADY || have2) load_one(g2, sidx / 3, nxt[sidx / 3], nxt_ok);
                }
                if (sidx % 4 == 2 && sidx / 4 < 8) {
                    if (STEADY) *reinterpret_cast<bf16x8*>(out_ptr[sidx / 4]) = outv[sidx / 4];
                    else if (have_out && ((out_ok >> (sidx / 4)) & 1u)) *reinterpret_cast<bf16x8*>(out_ptr[sidx / 4]) = outv[sidx / 4];
                }
                if (sidx % 3 == 2 && sidx / 3 < XV) {
                    if (STEADY || have1) transform(cur[sidx / 3], (cur_ok >> (sidx / 3)) & 1u);
                }
                if (sidx < 32) {
                    if (STEADY) pack_item(accp, gp, sidx >> 4, sidx & 15, false);
                    else if (have_p) pack_item(accp, gp, sidx >> 4, sidx & 15, true);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads of step s+1 and this slice ahead of the MFMAs of step s
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf)
                        acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][mf], bfr[cb][nf], acc[mf][nf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (all_full && have_p && have_out && have2) mfma_loop(std::true_type{});
        else mfma_loop(std::false_type{});
        stamp(4);
        __syncthreads();  // C: the staging tile of the previous tile is complete
        if (have_p) { fetch_out(gp); have_out = true; } else have_out = false;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) accp[mf][nf] = acc[mf][nf];
        gp = g;
        have_p = true;
        stamp(5);
    }
    // ---- drain: stores of the tile before last, then pack / stage / store the last tile
    if (have_out) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((out_ok >> k) & 1u) *reinterpret_cast<bf16x8*>(out_ptr[k]) = outv[k];
    }
    __syncthreads();  // every thread has fetched its vectors of the staging tile
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int i = 0; i < 16; ++i) pack_item(accp, gp, mf, i, true);
    __syncthreads();
    fetch_out(gp);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((out_ok >> k) & 1u) *reinterpret_cast<bf16x8*>(out_ptr[k]) = outv[k];
    if (STAMPS && stamps && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) stamps[((size_t)blockIdx.x * 4 + wave) * 6 + k] = ph[k];
    }
    if (STATS) {
        __syncthreads();  // sRed aliases the staging tile
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            float s1 = st1[nf] + __shfl_xor(st1[nf], 32, 64);
            float s2 = st2[nf] + __shfl_xor(st2[nf], 32, 64);
            if (h == 0) {
                sRed[(wave * 2 + 0) * C64 + 2 * r + nf] = s1;
                sRed[(wave * 2 + 1) * C64 + 2 * r + nf] = s2;
            }
        }
        __syncthreads();
        if (tid < 2 * C64) {
            const int which = tid / C64, n = tid - which * C64;
            a.stat[((size_t)blockIdx.x * 2 + which) * a.CoutP + n] =
                sRed[(0 * 2 + which) * C64 + n] + sRed[(1 * 2 + which) * C64 + n] + sRed[(2 * 2 + which) * C64 + n] +
                sRed[(3 * 2 + which) * C64 + n];
        }
    }
}

constexpr int C64_MAX_WGS = 256;  // one persistent workgroup per CU
inline bool use_c64(int Cin, int CoutP, int dtype) {
    static const bool off = getenv("WM_NO_C64") != nullptr;  // diagnostic knob: force the generic kernel
    return !off && dtype == WM_BF16 && (((Cin == 64 || Cin == 32 || Cin == 16) && CoutP == 64) || (Cin == 64 && CoutP == 32));
}  // + ldy == 64 (always true for a 64-channel output tensor)
inline int c64_tiles_per_wg(int ntiles) { return (ntiles + C64_MAX_WGS - 1) / C64_MAX_WGS; }
inline int c64_wgs(int ntiles) { const int per = c64_tiles_per_wg(ntiles); return (ntiles + per - 1) / per; }

// ---- weight packing: PyTorch [Cout][Cin][3][3] f32 -> [9][RowsP][ColsP] T
//   transpose == 0: rows = Cout, cols = packed Cin (perm applied), tap = kh*3+kw        (forward)
//   transpose == 1: rows = packed Cin, cols = Cout, tap = (2-kh)*3+(2-kw)               (dgrad)
struct PermArg { int p[128]; };
template <typename T>
__global__ void pack_w3x3_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int CoutP,
                                 int CinP, int transpose, PermArg perm, int has_perm) {
    const int rowsP = transpose ? CinP : CoutP, colsP = transpose ? CoutP : CinP;
    const size_t total = (size_t)9 * rowsP * colsP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % colsP);
        const int row = (int)((i / colsP) % rowsP);
        const int tap = (int)(i / ((size_t)colsP * rowsP));
        const int co = transpose ? col : row;
        const int cip = transpose ? row : col;  // packed input-channel index
        int kh = tap / 3, kw = tap % 3;
        if (transpose) { kh = 2 - kh; kw = 2 - kw; }
        float v = 0.f;
        if (co < Cout) {
            // find the reference input channel stored at packed position cip
            int ci = -1;
            if (!has_perm) ci = cip < Cin ? cip : -1;
            else {
                for (int q = 0; q < Cin; ++q) if (perm.p[q] == cip) { ci = q; break; }
            }
            if (ci >= 0) v = w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw];
        }
        wp[i] = from_f32<T>(v);
    }
}

// batched form: one launch packs every conv of a network (blockIdx.y = job).  Jobs live in device memory.
struct PackJob {
    const float* w; void* wp; const int* perm;   // perm: device int[Cin] or null
    int Cout, Cin, CoutP, CinP, transpose, pad;
};
template <typename T>
__global__ void pack_w3x3_batch_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int rowsP = j.transpose ? j.CinP : j.CoutP, colsP = j.transpose ? j.CoutP : j.CinP;
    const size_t total = (size_t)9 * rowsP * colsP;
    T* wp = (T*)j.wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % colsP);
        const int row = (int)((i / colsP) % rowsP);
        const int tap = (int)(i / ((size_t)colsP * rowsP));
        const int co = j.transpose ? col : row;
        const int cip = j.transpose ? row : col;
        int kh = tap / 3, kw = tap % 3;
        if (j.transpose) { kh = 2 - kh; kw = 2 - kw; }
        float v = 0.f;
        if (co < j.Cout) {
            int ci = -1;
            if (!j.perm) ci = cip < j.Cin ? cip : -1;
            else {
                for (int q = 0; q < j.Cin; ++q) if (j.perm[q] == cip) { ci = q; break; }
            }
            if (ci >= 0) v = j.w[(((size_t)co * j.Cin + ci) * 3 + kh) * 3 + kw];
        }
        wp[i] = from_f32<T>(v);
    }
}

template <typename T>
int launch_conv(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                const float* in_shift, void* y, int ldy, float* stat, int B, int H, int W, int Cin, int CoutP,
                hipStream_t s) {
    ConvArgs<T> a;
    a.x = (const T*)x; a.ldx = ldx; a.wp = (const T*)wp; a.bias = bias; a.nbias = nbias; a.in_scale = in_scale; a.in_shift = in_shift;
    a.y = (T*)y; a.ldy = ldy; a.stat = stat; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.CoutP = CoutP;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH);
    const bool xf = in_scale != nullptr;
    if constexpr (sizeof(T) == 2) {
        if (use_c64(Cin, CoutP, WM_BF16)) {
            const int ntiles = B * a.tilesX * a.tilesY;
            const int per = c64_tiles_per_wg(ntiles);
            dim3 grid((unsigned)c64_wgs(ntiles)), block(256);
            static const bool v1 = getenv("WM_C64_V1") != nullptr;  // diagnostic knob: single-role persistent kernel
            if (!v1 || Cin != 64 || CoutP != 64) return wm_launch_conv3x3_ws(x, ldx, Cin, CoutP, wp, bias, nbias, in_scale, in_shift, y, stat, B, H, W, c64_wgs(ntiles), per, s);
            unsigned long long* nost = nullptr;
            const bool st = stat != nullptr;
            if (xf && st) hipLaunchKernelGGL((conv3x3_c64_kernel<true, true>), grid, block, 0, s, a, a.x, a.y, ntiles, per, nost);
            else if (xf) hipLaunchKernelGGL((conv3x3_c64_kernel<true, false>), grid, block, 0, s, a, a.x, a.y, ntiles, per, nost);
            else if (st) hipLaunchKernelGGL((conv3x3_c64_kernel<false, true>), grid, block, 0, s, a, a.x, a.y, ntiles, per, nost);
            else hipLaunchKernelGGL((conv3x3_c64_kernel<false, false>), grid, block, 0, s, a, a.x, a.y, ntiles, per, nost);
            return WM_OK;
        }
    }
    const int BN = (CoutP % 64 == 0) ? 64 : 32;
    dim3 grid((unsigned)(B * a.tilesX * a.tilesY), (unsigned)(CoutP / BN)), block(256);
    if (BN == 64) {
        if (xf) hipLaunchKernelGGL((conv3x3_kernel<T, 64, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 64, false>), grid, block, 0, s, a);
    } else {
        if (xf) hipLaunchKernelGGL((conv3x3_kernel<T, 32, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 32, false>), grid, block, 0, s, a);
    }
    return WM_OK;
}

}  // namespace

extern "C" int wm_conv3x3_nparts(int B, int H, int W, int Cin, int CoutP, int dtype) {
    const int ntiles = B * wm_cdiv(H, TH) * wm_cdiv(W, TW);
    return use_c64(Cin, CoutP, dtype) ? c64_wgs(ntiles) : ntiles;
}

extern "C" int wm_conv3x3_fwd(const void* x, int ldx, const void* wp, const float* bias, int nbias, const float* in_scale,
                              const float* in_shift, void* y, int ldy, float* stat_partials, int B, int H, int W,
                              int Cin, int CoutP, int dtype, void* stream) {
    WM_REQUIRE(x && wp && y, WM_E_BADARG, "wm_conv3x3_fwd: null pointer");
    WM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && CoutP > 0, WM_E_BADARG, "wm_conv3x3_fwd: bad shape");
    WM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), WM_E_BADARG, "wm_conv3x3_fwd: in_scale/in_shift must come together");
    WM_REQUIRE(dtype == WM_F32 || dtype == WM_BF16, WM_E_BADARG, "wm_conv3x3_fwd: unsupported dtype %d", dtype);
    const int esz = dtype == WM_BF16 ? 2 : 4;
    const int cmul = dtype == WM_BF16 ? 16 : 4;
    WM_REQUIRE(Cin % cmul == 0, WM_E_SHAPE, "wm_conv3x3_fwd: Cin=%d must be a multiple of %d (pad the channels)", Cin, cmul);
    WM_REQUIRE(CoutP % 32 == 0, WM_E_SHAPE, "wm_conv3x3_fwd: CoutP=%d must be a multiple of 32", CoutP);
    WM_REQUIRE(ldx >= Cin && ldy >= CoutP && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, WM_E_SHAPE,
               "wm_conv3x3_fwd: pixel strides ldx=%d ldy=%d must cover the channels and be 16-byte multiples", ldx, ldy);
    WM_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y) & 15) == 0, WM_E_SHAPE, "wm_conv3x3_fwd: pointers must be 16-byte aligned");
    WM_REQUIRE(!use_c64(Cin, CoutP, dtype) || ldy == CoutP, WM_E_SHAPE,
               "wm_conv3x3_fwd: the persistent bf16 path writes a dense output (ldy must equal CoutP=%d, got %d)", CoutP, ldy);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WM_BF16) launch_conv<bf16_t>(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat_partials, B, H, W, Cin, CoutP, s);
    else launch_conv<float>(x, ldx, wp, bias, nbias, in_scale, in_shift, y, ldy, stat_partials, B, H, W, Cin, CoutP, s);
    WM_LAUNCH_CHECK("wm_conv3x3_fwd");
    return WM_OK;
}

extern "C" int wm_pack_w3x3(const float* w, void* wp, int Cout, int Cin, int CoutP, int CinP, const int* perm,
                            int transpose, int dtype, void* stream) {
    WM_REQUIRE(w && wp, WM_E_BADARG, "wm_pack_w3x3: null pointer");
    WM_REQUIRE(Cout > 0 && Cin > 0 && CoutP >= Cout && (perm || CinP >= Cin), WM_E_BADARG, "wm_pack_w3x3: bad shape");
    WM_REQUIRE(!perm || Cin <= 128, WM_E_SHAPE, "wm_pack_w3x3: channel permutation supports Cin <= 128");
    PermArg pa;
    for (int i = 0; i < 128; ++i) pa.p[i] = -1;
    if (perm)
        for (int i = 0; i < Cin; ++i) {
            // a packed position >= CinP means "this input channel is dropped" (dgrad of a channel slice)
            WM_REQUIRE(perm[i] >= 0, WM_E_BADARG, "wm_pack_w3x3: perm[%d]=%d out of range", i, perm[i]);
            pa.p[i] = perm[i];
        }
    const size_t total = (size_t)9 * CoutP * CinP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WM_BF16)
        hipLaunchKernelGGL(pack_w3x3_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)wp, Cout, Cin, CoutP, CinP, transpose, pa, perm ? 1 : 0);
    else if (dtype == WM_F32)
        hipLaunchKernelGGL(pack_w3x3_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)wp, Cout, Cin, CoutP, CinP, transpose, pa, perm ? 1 : 0);
    else { wm_set_error("wm_pack_w3x3: unsupported dtype %d", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_pack_w3x3");
    return WM_OK;
}

extern "C" int wm_pack_w3x3_batch(const void* jobs_dev, int njobs, size_t max_elems, int dtype, void* stream) {
    WM_REQUIRE(jobs_dev && njobs > 0 && max_elems > 0, WM_E_BADARG, "wm_pack_w3x3_batch: bad arguments");
    static_assert(sizeof(PackJob) == 48, "PackJob layout is part of the ABI (wm_hip.h)");
    int blocks = (int)((max_elems + 255) / 256);
    if (blocks > 64) blocks = 64;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks, (unsigned)njobs);
    if (dtype == WM_BF16) hipLaunchKernelGGL(pack_w3x3_batch_kernel<bf16_t>, grid, dim3(256), 0, s, (const PackJob*)jobs_dev);
    else if (dtype == WM_F32) hipLaunchKernelGGL(pack_w3x3_batch_kernel<float>, grid, dim3(256), 0, s, (const PackJob*)jobs_dev);
    else { wm_set_error("wm_pack_w3x3_batch: unsupported dtype %d", dtype); return WM_E_BADARG; }
    WM_LAUNCH_CHECK("wm_pack_w3x3_batch");
    return WM_OK;
}

// diagnostic only (not part of the public header): phase cycle totals of the persistent 64-channel kernel
extern "C" int wm_debug_conv3x3_c64_phases(const void* x, const void* wp, const float* in_scale, const float* in_shift,
                                           void* y, int B, int H, int W, unsigned long long* stamps, void* stream) {
    ConvArgs<bf16_t> a;
    a.x = (const bf16_t*)x; a.ldx = 64; a.wp = (const bf16_t*)wp; a.bias = nullptr; a.nbias = 0; a.in_scale = in_scale;
    a.in_shift = in_shift; a.y = (bf16_t*)y; a.ldy = 64; a.stat = nullptr; a.B = B; a.H = H; a.W = W; a.Cin = 64; a.CoutP = 64;
    a.tilesX = wm_cdiv(W, TW); a.tilesY = wm_cdiv(H, TH);
    const int ntiles = B * a.tilesX * a.tilesY, per = c64_tiles_per_wg(ntiles);
    dim3 grid((unsigned)c64_wgs(ntiles)), block(256);
    if (in_scale) hipLaunchKernelGGL((conv3x3_c64_kernel<true, false, true>), grid, block, 0, (hipStream_t)stream, a, a.x, a.y, ntiles, per, stamps);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<false, false, true>), grid, block, 0, (hipStream_t)stream, a, a.x, a.y, ntiles, per, stamps);
    WM_LAUNCH_CHECK("wm_debug_conv3x3_c64_phases");
    return c64_wgs(ntiles);
}
