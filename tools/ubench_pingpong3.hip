// Dev microbenchmark 3: which VALU instruction classes overlap with a partner wave's MFMAs?
// Ping-pong phases as in ubench_pingpong.hip; the V segment is N copies of one instruction class.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CLS, int N, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[8];
    f32x2 w[8];
    unsigned pk[8];
    for (int e = 0; e < 8; ++e) { v[e] = threadIdx.x * 0.01f + e; w[e] = (f32x2){v[e], v[e] + 1.f}; pk[e] = e; }
    long long t0 = __builtin_readcyclecounter();
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    for (int i = 0; i < iters; ++i) {
        if (MFMA) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int e = u & 7;
            if (CLS == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[e]));
            if (CLS == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
            if (CLS == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
            if (CLS == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w[e]));
            if (CLS == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[e]) : "v"(v[e]), "v"(v[(e + 1) & 7]));
            if (CLS == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w[e]));
            if (CLS == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_cbranch_vccz 1f\n\ts_nop 0\n1:" ::"v"(v[e]), "v"(v[(e + 1) & 7]) : "vcc");
            if (CLS == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[e]), "+v"(v[(e + 1) & 7]));
            if (CLS == 8) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[e]));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float res = acc0[0] + acc1[0];
    for (int e = 0; e < 8; ++e) res += v[e] + w[e][0] + w[e][1] + (float)pk[e];
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = res;
}

template <int CLS, int N>
void run(const char* name, float* out, long long* cyc, int iters) {
    double r[2];
    for (int m = 0; m < 2; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL((k<CLS, N, false>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL((k<CLS, N, true>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
            (void)hipDeviceSynchronize();
        }
        long long h[8];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        r[m] = (double)h[0] / iters / 2;
    }
    printf("%-28s x%3d: cycles/phase alone %6.1f   beside 8 MFMAs (256 cyc) %6.1f\n", name, N, r[0], r[1]);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 10000;
    run<0, 64>("v_add_f32", out, cyc, iters);
    run<8, 64>("v_fma_f32", out, cyc, iters);
    run<1, 32>("v_exp_f32", out, cyc, iters);
    run<2, 64>("v_max3_f32", out, cyc, iters);
    run<3, 32>("v_pk_add_f32", out, cyc, iters);
    run<5, 32>("v_pk_mul_f32", out, cyc, iters);
    run<4, 64>("v_cvt_pk_f16_f32", out, cyc, iters);
    run<6, 32>("v_cmp + s_cbranch_vccz", out, cyc, iters);
    run<7, 32>("v_permlane32_swap", out, cyc, iters);
    return 0;
}
