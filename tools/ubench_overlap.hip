// Dev microbenchmark: how much VALU work of one wave issues while ANOTHER wave of the same SIMD runs MFMAs,
// as a function of (a) whether the MFMAs form one dependent accumulator chain or rotate over independent
// accumulators and (b) the VALU instruction kind.  One workgroup of 512 threads on one CU: waves 0-3 (one per SIMD)
// run MFMAs, waves 4-7 (their SIMD partners) run VALU.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench tools/ubench_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE_MFMA, int MODE_VALU>   // MFMA: 0 none, 1 one chain, 2 two chains, 4 four chains; VALU: 0 none, 1 add, 2 exp, 3 mix
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    long long t0 = __builtin_readcyclecounter();
    float res = 0.f;
    if (mf) {
        if constexpr (MODE_MFMA > 0) {
            half8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
            f32x16 acc[4];
            for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    constexpr int NC = MODE_MFMA;
                    acc[u % NC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NC], 0, 0, 0);
                }
            }
            for (int c = 0; c < 4; ++c) res += acc[c][0];
        }
    } else {
        if constexpr (MODE_VALU > 0) {
            float v[8];
            for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.01f + e;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (MODE_VALU == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[e]));
                        if (MODE_VALU == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
                        if (MODE_VALU == 3) {
                            if (e < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
                            else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[e]));
                        }
                    }
                }
            }
            for (int e = 0; e < 8; ++e) res += v[e];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int M, int V>
void run(const char* name, float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((k<M, V>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<M, V>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on this part: report relative numbers
    printf("%-34s mfma-wave ticks %8lld  valu-wave ticks %8lld\n", name, h[0], h[4]);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 20000;   // MFMA wave: 16*iters MFMAs (32 cyc each = 10.2 M cycles); VALU wave: 64*iters instrs
    run<1, 0>("mfma 1 chain, no valu", out, cyc, iters);
    run<4, 0>("mfma 4 chains, no valu", out, cyc, iters);
    run<0, 1>("no mfma, add", out, cyc, iters);
    run<0, 2>("no mfma, exp", out, cyc, iters);
    run<0, 3>("no mfma, mix(2 exp+6 add)", out, cyc, iters);
    run<1, 1>("mfma 1 chain + add", out, cyc, iters);
    run<2, 1>("mfma 2 chains + add", out, cyc, iters);
    run<4, 1>("mfma 4 chains + add", out, cyc, iters);
    run<1, 2>("mfma 1 chain + exp", out, cyc, iters);
    run<4, 2>("mfma 4 chains + exp", out, cyc, iters);
    run<1, 3>("mfma 1 chain + mix", out, cyc, iters);
    run<4, 3>("mfma 4 chains + mix", out, cyc, iters);
    return 0;
}
