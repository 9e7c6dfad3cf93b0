// Would the forward conv's consumer role run faster as TWO waves per SIMD (gfx950)?  A model of conv3x3_ws.hip's tile loop on one CU:
// a SIMD has to get through S "steps" of [R ds_read_b128 for the next step | 8 v_mfma_f32_16x16x32_bf16 with V VALU spread between them]
// (R = 6, V = 12: the production kernel's step with its drain; reads counted with lgkmcnt so that the next step's are in flight), next to a
// producer wave per SIMD that issues PV VALU + 1 ds_write_b128 per consumer step x PSCALE (the BN + ReLU staging).  NW = 1: one consumer wave
// per SIMD does all S steps; NW = 2: two consumer waves per SIMD do S / 2 steps each (12 waves per workgroup).  Prints cycles per MFMA of
// the SIMD (S * 8 MFMAs / the slowest consumer wave's cycles).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/consumer_split.hip -o tools/micro/consumer_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NW, int R, int V, int PV>
__global__ __launch_bounds__(256 * (NW + 1), 1) void k(unsigned long long* out, int steps, float seed) {
    __shared__ u32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = u32x4{1, 2, 3, 4};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 * NW) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
        s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i;
        u32x4 l[2][R > 0 ? R : 1];
        const unsigned laddr = (threadIdx.x & 1023) * 16;
#pragma unroll
        for (int q = 0; q < R; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[0][q]) : "v"(laddr), "n"(q * 4096));
        for (int it = 0; it < steps / NW; it += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < R; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[h ^ 1][q]) : "v"(laddr), "n"(q * 4096));
                if (R) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R));
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    // the fragments just waited for feed the MFMAs (a data dependence on the loaded registers, as in the real loop)
                    if (R) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(l[h][m % R]), "v"(l[h][(m + 1) % R]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
                    for (int v = (m * V) / 8; v < ((m + 1) * V) / 8; ++v) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[v & 7]) : "v"(seed));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
        if (s == 123.456f) out[1] = 1;
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i;
        const unsigned waddr = 32768 + (threadIdx.x & 255) * 16;
        for (int it = 0; it < steps; ++it) {
#pragma unroll
            for (int v = 0; v < PV; ++v) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[v & 7]) : "v"(seed));
            if (PV && (it & 3) == 3) asm volatile("ds_write_b128 %0, %1" : : "v"(waddr), "v"(u32x4{1, 2, 3, 4}) : "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i];
        if (s == 123.456f) out[1] = 1;
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    }
}

template <int NW, int R, int V, int PV>
void run(const char* what) {
    const int nwg = 256, steps = 4000;
    unsigned long long* d;
    hipMalloc(&d, nwg * 16 * 8);
    hipMemset(d, 0, nwg * 16 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NW, R, V, PV>), dim3(nwg), dim3(256 * (NW + 1)), 0, 0, d, steps, 0.f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 16);
    hipMemcpy(h.data(), d, nwg * 16 * 8, hipMemcpyDeviceToHost);
    double c = 0, p = 0;
    for (int i = 0; i < nwg; ++i) {
        unsigned long long mc = 0, mp = 0;
        for (int w = 0; w < 4 * NW; ++w) mc = h[i * 16 + w] > mc ? h[i * 16 + w] : mc;
        for (int w = 4 * NW; w < 4 * NW + 4; ++w) mp = h[i * 16 + w] > mp ? h[i * 16 + w] : mp;
        c += (double)mc; p += (double)mp;
    }
    printf("%-78s %6.1f cycles per MFMA of the SIMD   (producer wave done after %5.1f per MFMA)\n", what, c / nwg / (steps * 8.0), p / nwg / (steps * 8.0));
    hipFree(d);
}

int main() {
    run<1, 0, 0, 0>("1 consumer wave/SIMD: MFMAs only");
    run<2, 0, 0, 0>("2 consumer waves/SIMD: MFMAs only");
    run<1, 6, 0, 0>("1 consumer wave/SIMD: 6 reads per 8 MFMAs");
    run<2, 6, 0, 0>("2 consumer waves/SIMD: 6 reads per 8 MFMAs");
    run<1, 6, 12, 0>("1 consumer wave/SIMD: 6 reads + 12 VALU per 8 MFMAs");
    run<2, 6, 12, 0>("2 consumer waves/SIMD: 6 reads + 12 VALU per 8 MFMAs");
    run<1, 6, 12, 11>("1 consumer wave/SIMD: 6 reads + 12 VALU per 8 MFMAs | producer 11 VALU per step");
    run<2, 6, 12, 11>("2 consumer waves/SIMD: 6 reads + 12 VALU per 8 MFMAs | producer 11 VALU per step");
    run<1, 6, 4, 11>("1 consumer wave/SIMD: 6 reads + 4 VALU per 8 MFMAs | producer 11 VALU per step");
    run<2, 6, 4, 11>("2 consumer waves/SIMD: 6 reads + 4 VALU per 8 MFMAs | producer 11 VALU per step");
    run<1, 3, 12, 11>("1 consumer wave/SIMD: 3 reads + 12 VALU per 8 MFMAs | producer 11 VALU per step");
    run<2, 3, 12, 11>("2 consumer waves/SIMD: 3 reads + 12 VALU per 8 MFMAs | producer 11 VALU per step");
    run<1, 6, 12, 23>("1 consumer wave/SIMD: 6 reads + 12 VALU per 8 MFMAs | producer 23 VALU per step");
    run<1, 6, 0, 23>("1 consumer wave/SIMD: 6 reads per 8 MFMAs | producer 23 VALU per step (the drain moved over)");
    run<1, 3, 0, 23>("1 consumer wave/SIMD: 3 reads per 8 MFMAs | producer 23 VALU per step");
    return 0;
}
