// Dev experiment: what does a CU's 256x256 fp16 output tile (128 KB) cost to store, by access pattern?
// 256 workgroups x 512 threads (one per CU), each stores `tiles` tiles of 256 rows x 512 B into a [M][C] fp16 matrix
// (row stride C * 2 B), every lane 16 B per instruction.  Pattern = bytes of one row covered by one wave instruction:
//   64: 16 rows x 64 B (the GEMM epilogue today), 128: 8 rows x 128 B, 256: 4 x 256, 512: 2 x 512 (needs an LDS transpose)
// build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int RB>   // row bytes per instruction
__global__ __launch_bounds__(512) void k(char* out, int C, int tiles_n, int tiles, int busy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 v = {1.f * lane, 2.f, 3.f, 4.f};
    constexpr int LPR = RB / 16, RPI = 64 / LPR;            // lanes per row, rows per instruction
    // a wave owns 128 KB / 8 = 16 KB of the tile = 16 instructions
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        char* base = out + ((size_t)tm * 256) * C * 2 + (size_t)tn * 512;
        // wave w covers a (rows x RBW) patch: keep each wave's 16 KB as a rectangle RB wide (or 128 B wide for RB < 128)
        constexpr int W = RB < 128 ? 128 : RB;               // patch width in bytes
        constexpr int PR = 16384 / W;                        // patch rows
        constexpr int PPR = 512 / W;                         // patches per tile row
        const int pr0 = (wave / PPR) * PR, pc0 = (wave % PPR) * W;
        if (busy) { float a = v[0]; for (int i = 0; i < busy; ++i) a = a * 1.0001f + 0.5f; v[1] = a; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // instruction i: rows and column offset inside the patch
            int r, cb;
            if (RB >= 128) { r = i * RPI + lane / LPR; cb = (lane % LPR) * 16; }
            else { r = (i / 2) * RPI + lane / LPR; cb = (i & 1) * 64 + (lane % LPR) * 16; }
            *(f4*)(base + (size_t)(pr0 + r) * C * 2 + pc0 + cb) = v;
        }
    }
}

int main(int argc, char** argv) {
    const int M = 131072, C = argc > 1 ? atoi(argv[1]) : 2048;
    char* out; hipMalloc(&out, (size_t)M * C * 2);
    const int tiles_n = C / 256, total = (M / 256) * tiles_n, per = total / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 128, 64, 32, 8}) for (int pat : {64, 256}) {
        const int busy = 0; const int per = 16;
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (pat == 64) k<64><<<grid, 512>>>(out, C, tiles_n, per, busy);
            if (pat == 128) k<128><<<grid, 512>>>(out, C, tiles_n, per, busy);
            if (pat == 256) k<256><<<grid, 512>>>(out, C, tiles_n, per, busy);
            if (pat == 512) k<512><<<grid, 512>>>(out, C, tiles_n, per, busy);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("C=%d grid=%d row-bytes/instr %3d: %8.1f us total, %6.2f us per tile, %.2f TB/s\n", C, grid, pat, best * 1e3,
               best * 1e3 / per, (double)grid * per * 131072 / best / 1e9);
    }
    return 0;
}
