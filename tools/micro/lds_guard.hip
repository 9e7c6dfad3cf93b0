// Does a kernel of libwm_hip.so write into the LDS of ANOTHER workgroup that shares its CU?  A guard kernel fills its 26 KB of LDS with a
// pattern and re-reads it for a few microseconds while wm_conv3x3_bwd_fused16 (bwd_ws16: 75 KB of LDS, two workgroups per CU) runs on a second
// stream.   build: hipcc --offload-arch=gfx950 -O2 tools/micro/lds_guard.hip -o tools/micro/lds_guard -L<lib dir> -lwm_hip -Wl,-rpath,<lib dir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int wm_conv3x3_bwd_fused16(const void* g, const void* y, const float* stats4, const float* coef, const void* wpt, const void* x, void* dx,
                                      float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, int premasked,
                                      int sweep_reverse, void* stream);
extern "C" int wm_conv3x3_bwd_fused16_nwg(int B, int H, int W);
extern "C" const char* wm_last_error_string();

constexpr int NW = 6656;   // dwords of LDS per guard workgroup (26,624 B: the JPEG kernels' footprint)
__global__ __launch_bounds__(256) void lds_guard_kernel(unsigned* errors, unsigned* first, int rounds) {
    __shared__ unsigned buf[NW];
    for (int i = threadIdx.x; i < NW; i += 256) buf[i] = 0x9e3779b9u * (unsigned)(i + 1) ^ (blockIdx.x << 16);
    __syncthreads();
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < NW; i += 256) {
            const unsigned v = ((volatile unsigned*)buf)[i];
            const unsigned want = 0x9e3779b9u * (unsigned)(i + 1) ^ (blockIdx.x << 16);
            if (v != want) {
                if (!bad && atomicAdd(&errors[1], 1u) < 8) { const unsigned k = atomicAdd(&errors[2], 1u); if (k < 8) { first[4 * k] = blockIdx.x; first[4 * k + 1] = i; first[4 * k + 2] = v; first[4 * k + 3] = want; } }
                ++bad;
                ((volatile unsigned*)buf)[i] = want;   // repair, count the next hit separately
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (bad) atomicAdd(&errors[0], bad);
}

// the JPEG kernels' wave-private 8x8 transposes (csrc/jpeg.hip transpose3: six ds_write_b128, a wait, 24 strided reads, no workgroup barrier) on
// known values: lane (r, blk) holds row r of block blk; afterwards it must hold column r
constexpr int LDS_BLK = 68, LDS_WAVE = 3 * 8 * LDS_BLK;
__device__ __forceinline__ float pat(int c, int blk, int row, int col, int it, int wg) { return (float)(((c * 8 + blk) * 8 + row) * 8 + col) + 4096.f * (float)((it + wg) & 63); }
__global__ __launch_bounds__(256) void transpose_guard_kernel(unsigned* errors, unsigned* first, int rounds) {
    __shared__ __attribute__((aligned(16))) float s_lds[4 * LDS_WAVE];
    const int lane = threadIdx.x & 63, r = lane >> 3, blk = lane & 7;
    float* lds = s_lds + (threadIdx.x >> 6) * LDS_WAVE;
    unsigned bad = 0;
    for (int it = 0; it < rounds; ++it) {
        float v[3][8];
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 8; ++k) v[c][k] = pat(c, blk, r, k, it, blockIdx.x);
        for (int c = 0; c < 3; ++c) {
            float* p = lds + (c * 8 + blk) * LDS_BLK + r * 8;
            *reinterpret_cast<float4*>(p) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(v[c][4], v[c][5], v[c][6], v[c][7]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int c = 0; c < 3; ++c) {
            const float* p = lds + (c * 8 + blk) * LDS_BLK + r;
            for (int k = 0; k < 8; ++k) v[c][k] = p[k * 8];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 8; ++k) {
                const float want = pat(c, blk, k, r, it, blockIdx.x);
                if (v[c][k] != want) {
                    if (!bad) { const unsigned n = atomicAdd(&errors[2], 1u); if (n < 8) { first[4 * n] = blockIdx.x; first[4 * n + 1] = (unsigned)(threadIdx.x * 1000 + c * 100 + k); first[4 * n + 2] = __float_as_uint(v[c][k]); first[4 * n + 3] = __float_as_uint(want); } }
                    ++bad;
                }
            }
    }
    if (bad) atomicAdd(&errors[0], bad);
}

int main(int argc, char** argv) {
    const int B = 16, H = 256, W = 256;
    const size_t n64 = (size_t)B * H * W * 64, n16 = (size_t)B * H * W * 16;
    unsigned short *g, *y, *x, *dx, *wpt; float *stats, *coef, *ws, *dw; unsigned *err, *first;
    hipMalloc(&g, n64 * 2); hipMalloc(&y, n64 * 2); hipMalloc(&x, n16 * 2); hipMalloc(&dx, n16 * 2); hipMalloc(&wpt, 9 * 16 * 64 * 2);
    hipMalloc(&stats, 4 * 64 * 4); hipMalloc(&coef, 3 * 64 * 4); hipMalloc(&dw, 64 * 3 * 9 * 4);
    const int nwg = wm_conv3x3_bwd_fused16_nwg(B, H, W);
    hipMalloc(&ws, (size_t)nwg * 9 * 16 * 64 * 4); hipMalloc(&err, 16); hipMalloc(&first, 32 * 4);
    std::vector<unsigned short> h(n64); for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));   // bf16 values near 0.01..0.03
    hipMemcpy(g, h.data(), n64 * 2, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), n64 * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), n16 * 2, hipMemcpyHostToDevice); hipMemcpy(wpt, h.data(), 9 * 16 * 64 * 2, hipMemcpyHostToDevice);
    std::vector<float> f(4 * 64, 1.0f); hipMemcpy(stats, f.data(), 4 * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(coef, f.data(), 3 * 64 * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(err, 0, 16); hipMemset(first, 0, 128);
        hipDeviceSynchronize();
        for (int it = 0; it < 20; ++it) {
            if (mode == 1)
                for (int k = 0; k < 3; ++k)
                    if (wm_conv3x3_bwd_fused16(g, y, stats, coef, wpt, x, dx, ws, dw, 0, B, H, W, 3, 64, 1 /* bf16 */, 1, 0, s1)) { printf("bwd_fused16 failed: %s\n", wm_last_error_string()); return 1; }
            if (argc > 1) hipLaunchKernelGGL(transpose_guard_kernel, dim3(4096), dim3(256), 0, s2, err, first, 60);
            else hipLaunchKernelGGL(lds_guard_kernel, dim3(4096), dim3(256), 0, s2, err, first, 40);
            hipDeviceSynchronize();
        }
        unsigned e[4], fi[32]; hipMemcpy(e, err, 16, hipMemcpyDeviceToHost); hipMemcpy(fi, first, 128, hipMemcpyDeviceToHost);
        printf("%s: %u corrupted LDS dwords seen by the guard kernel\n", mode ? "beside bwd_ws16" : "alone", e[0]);
        for (unsigned k = 0; k < e[2] && k < 8; ++k) printf("   workgroup %u dword %u (byte %u): read 0x%08x, wrote 0x%08x\n", fi[4 * k], fi[4 * k + 1], 4 * fi[4 * k + 1], fi[4 * k + 2], fi[4 * k + 3]);
    }
    return 0;
}
