// Dev probe: issue cost of v_exp_f16 against v_exp_f32 in the gaps of v_mfma_f32_32x32x16_f16 -- the regime of the d <= 32 set-attention
// kernels, where softmax's exponentials, not the matrix pipe, set the pace (csrc/attention.hip: P is rounded to fp16 for the PV product anyway).
// One or two waves per SIMD, every CU busy; a loop body = 1 MFMA + E transcendentals on independent registers; prints shader cycles per body (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_exp tools/ubench_exp.hip && tools/ubench_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int E>          // KIND 4 / 5: as 3 / 0 with the MFMA accumulators in AGPRs; KIND 0: v_exp_f32; 1: v_exp_f16; 2: v_exp_f32 + v_cvt_pk (fp16 pack of two results); 3: v_add_f32 (a 4-cycle reference)
__global__ void body(float* out, unsigned long long* cyc, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = -0.01f * (float)(threadIdx.x + i);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND >= 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));     // accumulators in AGPRs
            else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));     // (asm: the compiler may not move it across the fillers)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (KIND == 0 || KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x[e]));
                else if constexpr (KIND == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(x[e]));
                else if constexpr (KIND == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[e])); if (e & 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[e - 1]) : "v"(x[e - 1]), "v"(x[e])); }
                else asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[e]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int E>
static double run(int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((body<KIND, E>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    hipFree(out); hipFree(cyc);
    // s_memtime ticks at 100 MHz on gfx950?  No: shader clock (MI355X_MICROARCH.md: "tick = shader cycle")
    return (double)h[h.size() / 2] / (iters * 8.0);
}

template <int KIND>
static void sweep(const char* name) {
    for (int w = 1; w <= 2; ++w) {
        printf("%-34s %d wave(s)/SIMD, cycles per [MFMA 32x32x16 + E ops] of ONE wave, E = 0..8:", name, w);
        printf(" %6.1f", run<KIND, 0>(w)); printf(" %6.1f", run<KIND, 1>(w)); printf(" %6.1f", run<KIND, 2>(w)); printf(" %6.1f", run<KIND, 3>(w));
        printf(" %6.1f", run<KIND, 4>(w)); printf(" %6.1f", run<KIND, 5>(w)); printf(" %6.1f", run<KIND, 6>(w)); printf(" %6.1f", run<KIND, 7>(w));
        printf(" %6.1f\n", run<KIND, 8>(w));
    }
}

int main() {
    sweep<3>("v_add_f32 (4-cycle reference)");
    sweep<0>("v_exp_f32");
    sweep<1>("v_exp_f16");
    sweep<2>("v_exp_f32 + v_cvt_pk_f16_f32 per 2");
    sweep<4>("v_add_f32, MFMA acc in AGPRs");
    sweep<5>("v_exp_f32, MFMA acc in AGPRs");
    return 0;
}
