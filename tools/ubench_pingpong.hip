// Dev microbenchmark: cost of the s_barrier-separated ping-pong (one wave of a SIMD issues 8 MFMAs while its
// partner issues ~60 VALU, then they swap).  One workgroup of 8 waves on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool BARRIER, int NEXP, int NADD>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.01f + e;
    long long t0 = __builtin_readcyclecounter();
    if (grp == 1 && BARRIER) asm volatile("s_barrier" ::: "memory");
    for (int i = 0; i < iters; ++i) {
        // M segment
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
        // V segment
#pragma unroll
        for (int u = 0; u < NEXP; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
#pragma unroll
        for (int u = 0; u < NADD; ++u) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[u & 7]));
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
    }
    if (grp == 0 && BARRIER) asm volatile("s_barrier" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float res = acc0[0] + acc1[0];
    for (int e = 0; e < 8; ++e) res += v[e];
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = res;
}

template <bool B, int NE, int NA>
void run(const char* name, float* out, long long* cyc, int iters) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<B, NE, NA>), dim3(1), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s cycles/iteration (2 phases): wave0 %.1f wave4 %.1f\n", name, (double)h[0] / iters, (double)h[4] / iters);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 20000;
    run<false, 0, 0>("8 MFMA only, no barrier", out, cyc, iters);
    run<true, 0, 0>("8 MFMA only, barrier", out, cyc, iters);
    run<false, 16, 44>("8 MFMA + 16 exp + 44 add, no barrier", out, cyc, iters);
    run<true, 16, 44>("8 MFMA + 16 exp + 44 add, barrier", out, cyc, iters);
    run<true, 16, 28>("8 MFMA + 16 exp + 28 add, barrier", out, cyc, iters);
    run<true, 8, 22>("8 MFMA + 8 exp + 22 add, barrier", out, cyc, iters);
    return 0;
}
