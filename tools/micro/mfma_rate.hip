// What one wave per SIMD can issue next to its MFMAs (gfx950): cycles per v_mfma_f32_16x16x32_bf16 in a loop of independent MFMAs with
// NV independent VALU instructions and NL ds_read_b128 behind each.  One workgroup of 256 threads per CU (as bwd_ws.hip runs).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate ; run: tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int DEP>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, int iters, float seed) {
    __shared__ u32x4 lds[1024];
    lds[threadIdx.x] = u32x4{1, 2, 3, 4};
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    u32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = u32x4{0, 0, 0, 0};
    const unsigned laddr = (threadIdx.x & 255) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[DEP ? 0 : m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[(m * NV + v) & 7]) : "v"(seed));
#pragma unroll
            for (int q = 0; q < NL; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(l[(m * NL + q) & 3]) : "v"(laddr));
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
    for (int i = 0; i < 4; ++i) s += (float)l[i][0];
    if (s == 123.456f) out[1] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NL, int PK>
__global__ __launch_bounds__(256, 1) void k32(unsigned long long* out, int iters, float seed) {
    __shared__ u32x4 lds[1024];
    lds[threadIdx.x] = u32x4{1, 2, 3, 4};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = seed;
    s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed + i, seed};
    const f32x2 sd = {seed, seed};
    u32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = u32x4{0, 0, 0, 0};
    const unsigned laddr = (threadIdx.x & 255) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(x[(m * NV + v) & 7]) : "v"(sd));
                else asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[(m * NV + v) & 7][0]) : "v"(seed));
            }
#pragma unroll
            for (int q = 0; q < NL; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(l[(m * NL + q) & 3]) : "v"(laddr));
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    for (int i = 0; i < 4; ++i) s += (float)l[i][0];
    if (s == 123.456f) out[1] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int NV, int NL, int PK>
void run32(const char* what) {
    const int nwg = 256, iters = 2000;
    unsigned long long* d;
    hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL((k32<NV, NL, PK>), dim3(nwg), dim3(256), 0, 0, d, iters, 0.f);
    hipLaunchKernelGGL((k32<NV, NL, PK>), dim3(nwg), dim3(256), 0, 0, d, iters, 0.f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-44s %6.1f cycles per MFMA 32x32x16 (= two 16x16x32)\n", what, s / nwg / iters / 4);
    hipFree(d);
}

// two waves per SIMD: waves 0-3 issue MFMAs (+ NL ds_read_b128 each), waves 4-7 a stream of independent VALU instructions (+ one
// ds_write_b128 per 16) until the MFMA waves are done; reports cycles per MFMA and the VALU waves' instructions per cycle
template <int NL>
__global__ __launch_bounds__(512, 1) void k2(unsigned long long* out, int iters, float seed) {
    __shared__ u32x4 lds[2048];
    __shared__ volatile int done;
    lds[threadIdx.x] = u32x4{1, 2, 3, 4};
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
        s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
        u32x4 l[4];
        for (int i = 0; i < 4; ++i) l[i] = u32x4{0, 0, 0, 0};
        const unsigned laddr = (threadIdx.x & 255) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
                for (int q = 0; q < NL; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(l[(m * NL + q) & 3]) : "v"(laddr));
            }
            if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        for (int i = 0; i < 4; ++i) s += (float)l[i][0];
        if (s == 123.456f) out[1] = 1;
        if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; done = 1; }
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i;
        unsigned long long n = 0;
        const unsigned waddr = 16384 + (threadIdx.x & 255) * 16;
        while (!done) {
#pragma unroll
            for (int v = 0; v < 64; ++v) {
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[v & 7]) : "v"(seed));
                if ((v & 15) == 15) asm volatile("ds_write_b128 %0, %1" : : "v"(waddr), "v"(u32x4{1, 2, 3, 4}) : "memory");
            }
            n += 64;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i];
        if (s == 123.456f) out[1] = 1;
        if (threadIdx.x == 256) { out[blockIdx.x * 4 + 1] = t1 - t0; out[blockIdx.x * 4 + 2] = n; }
    }
}
template <int NL>
void run2(const char* what) {
    const int nwg = 256, iters = 2000;
    unsigned long long* d;
    hipMalloc(&d, nwg * 32);
    hipLaunchKernelGGL((k2<NL>), dim3(nwg), dim3(512), 0, 0, d, iters, 0.f);
    hipLaunchKernelGGL((k2<NL>), dim3(nwg), dim3(512), 0, 0, d, iters, 0.f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), d, nwg * 32, hipMemcpyDeviceToHost);
    double s = 0, v = 0, tv = 0;
    for (int i = 0; i < nwg; ++i) { s += (double)h[4 * i]; tv += (double)h[4 * i + 1]; v += (double)h[4 * i + 2]; }
    printf("%-44s %6.1f cycles per MFMA; the VALU wave beside it: %.2f VALU per MFMA (%.1f cycles per VALU)\n", what, s / nwg / iters / 8, v / nwg / (iters * 8.0), tv / v);
    hipFree(d);
}

template <int NV, int NL, int DEP>
void run(const char* what) {
    const int nwg = 256, iters = 2000;
    unsigned long long* d;
    hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL((k<NV, NL, DEP>), dim3(nwg), dim3(256), 0, 0, d, iters, 0.f);
    hipLaunchKernelGGL((k<NV, NL, DEP>), dim3(nwg), dim3(256), 0, 0, d, iters, 0.f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-44s %6.1f cycles per MFMA\n", what, s / nwg / iters / 8);
    hipFree(d);
}

int main() {
    run<0, 0, 0>("MFMA only (8 independent accumulators)");
    run<0, 0, 1>("MFMA only (one accumulator: dependent)");
    run<1, 0, 0>("+1 VALU each");
    run<2, 0, 0>("+2 VALU each");
    run<3, 0, 0>("+3 VALU each");
    run<4, 0, 0>("+4 VALU each");
    run<6, 0, 0>("+6 VALU each");
    run<8, 0, 0>("+8 VALU each");
    run<0, 1, 0>("+1 ds_read_b128 each");
    run<0, 2, 0>("+2 ds_read_b128 each");
    run<3, 1, 0>("+3 VALU +1 ds_read_b128 each");
    run2<0>("2 waves/SIMD: MFMA wave | VALU wave");
    run2<1>("2 waves/SIMD: MFMA +1 ds_read_b128 | VALU wave");
    run32<0, 0, 0>("32x32x16 only");
    run32<2, 0, 0>("32x32x16 +2 VALU each");
    run32<4, 0, 0>("32x32x16 +4 VALU each");
    run32<6, 0, 0>("32x32x16 +6 VALU each");
    run32<8, 0, 0>("32x32x16 +8 VALU each");
    run32<12, 0, 0>("32x32x16 +12 VALU each");
    run32<4, 0, 1>("32x32x16 +4 v_pk_fma_f32 each");
    run32<8, 0, 1>("32x32x16 +8 v_pk_fma_f32 each");
    run32<0, 1, 0>("32x32x16 +1 ds_read_b128 each");
    run32<0, 2, 0>("32x32x16 +2 ds_read_b128 each");
    run32<0, 3, 0>("32x32x16 +3 ds_read_b128 each");
    run32<6, 2, 0>("32x32x16 +6 VALU +2 ds_read_b128 each");
    return 0;
}
