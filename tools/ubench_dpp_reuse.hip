// Dev probe for the next VAE step: the k3 halo kernels read one 1-KB voxel fragment (16 voxels x 32 channels, MFMA B operand) from LDS per TWO MFMAs (C_out 32 per wave),
// which keeps the LDS port as busy as the matrix pipe.  The three dx taps of a (dz, dy) row use the SAME voxels shifted by one lane: can the shifted fragments be made
// from the first one with DPP moves (VALU, idle in those kernels) instead of two more LDS reads?
//   variant 0: per step 3 x [ds_read_b128 + 2 MFMA 16x16x32]            (what the kernels do)
//   variant 1: per step 1 x ds_read_b128, 2 x [4 v_mov_dpp row_shl:1 + 4 v_mov_dpp fix-up of lane 15 from the neighbour fragment] , 6 MFMA
//   variant 2: per step 1 x ds_read_b128, 6 MFMA                            (the floor: no shifts at all)
// Two waves per SIMD, every CU busy, conflict-free reads; prints shader cycles per step and the matrix pipe's share (6 x 16 = 96 cycles of MFMA per wave and step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dpp_reuse tools/ubench_dpp_reuse.hip && tools/ubench_dpp_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 shl1(u32x4 v, u32x4 next) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // lane l <- lane l + 1 inside its row of 16 (row_shl:1; lane 15 keeps the old value), then lane 15 <- lane 0 of the neighbour fragment (row_shr:15, lanes 15 only)
        unsigned a = __builtin_amdgcn_update_dpp(v[i], v[i], 0x101, 0xf, 0xf, false);
        r[i] = __builtin_amdgcn_update_dpp(a, next[i], 0x11f, 0xf, 0x8, false);
    }
    return r;
}

template <int VARIANT>
__global__ __launch_bounds__(512) void body(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 w0, w1;
    for (int i = 0; i < 8; ++i) { w0[i] = (_Float16)(0.01f * (lane + i)); w1[i] = (_Float16)(0.02f * (lane ^ i)); }
    f32x4 acc[6] = {};
    const char* base = smem + wave * 8192 + lane * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const char* p = base + ((it * 2 + u) & 1) * 4096;
            if constexpr (VARIANT == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const half8 xb = *(const half8*)(p + t * 1024);
                    acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[2 * t], 0, 0, 0);
                    acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xb, acc[2 * t + 1], 0, 0, 0);
                }
            } else {
                const u32x4 x0 = *(const u32x4*)p;
                const u32x4 nb = *(const u32x4*)(p + 1024);                       // (the neighbour fragment: in a kernel it is the next voxel group's, already loaded)
                u32x4 x1 = x0, x2 = x0;
                if constexpr (VARIANT == 1) { x1 = shl1(x0, nb); x2 = shl1(x1, nb); }
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, __builtin_bit_cast(half8, x0), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, __builtin_bit_cast(half8, x0), acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, __builtin_bit_cast(half8, x1), acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, __builtin_bit_cast(half8, x1), acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, __builtin_bit_cast(half8, x2), acc[4], 0, 0, 0);
                acc[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, __builtin_bit_cast(half8, x2), acc[5], 0, 0, 0);
                if constexpr (VARIANT == 2) acc[0][0] += __builtin_bit_cast(float, nb[0]);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int VARIANT>
static double run() {
    const int threads = 512, blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((body<VARIANT>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)h[h.size() / 2] / (iters * 2.0);
}

int main() {
    const char* names[3] = {"3 x [LDS read + 2 MFMA]", "1 LDS read (+ neighbour), 2 x 8 DPP moves, 6 MFMA", "1 LDS read (+ neighbour), 6 MFMA (no shifts: floor)"};
    const double c[3] = {run<0>(), run<1>(), run<2>()};
    for (int v = 0; v < 3; ++v)
        printf("%-56s %7.1f cycles per step of one wave (2 waves per SIMD) -> matrix pipe %4.1f %% busy\n", names[v], c[v], 100.0 * 2 * 96.0 / c[v]);
    return 0;
}
