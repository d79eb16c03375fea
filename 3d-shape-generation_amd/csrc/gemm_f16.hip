// K1: fp16-in / fp32-accumulate MFMA GEMM for gfx950 with fused epilogues.
//
//   out[m][c] = act( sum_k A[m][k] * W[c][k] + bias[c] + shape_bias[m / rps][c] )
//
// Replaces Conv1d(k=1)+BatchNorm1d(eval)+ReLU (reference networks.py:46-48) and the
// Linear layers; BN is folded into W/bias on the host.  Both operands are K-major, so
// activation rows and weight rows are staged identically: global -> LDS with
// global_load_lds (16 B/lane), XOR-swizzled through the *source* address (the LDS image
// written by an LDS-DMA is lane-linear), read back with ds_read_b128 conflict-free, and
// fed to v_mfma_f32_16x16x32_f16.  Double-buffered, one barrier per 64-deep K tile.
//
// Tile: BM x BN x 64, 256 threads = 4 waves as 2(M) x 2(N).
#include <mutex>
#include "common.h"
#include <type_traits>

namespace pcd {

enum { EPI_F16 = 0, EPI_F32 = 1, EPI_COLMAX = 2, EPI_RESID = 3 };

struct GemmParams {
    const half_t* a1; int64_t lda1; int k1;
    const half_t* a2; int64_t lda2; int k2;
    const half_t* w; int64_t ldw;
    const float* bias; const float* shape_bias; int rows_per_shape;
    int relu; int m; int c;
    half_t* out16; float* out32; int64_t ldo;
    const half_t* resid; int64_t ldr;
    float* colmax; int cm_rps;
    int tiles_m; int tiles_n;
    int patch_pn, patch_xn;   // XCD patch mapping (0 = linear tile order)
    // split-K (EPI_F32 only): `splits` independent products over consecutive k1-deep slices of the reduction;
    // slice s reads A and W at column offset s*k1 and writes the fp32 slab out32 + s*split_out
    int splits; int64_t split_out;
    int xp_depth;             // gemm_xp_kernel: K tiles of the next tile requested before the epilogue's stores (1 or 2)
    int krep;                 // 2 = hi / lo weights (pcd_gemm_f16_hilo): the sources are walked TWICE, first against columns [0, K) of W (the fp16
                              // weights), then against [K, 2K) (fp16 of the rounding residuals): ~22-bit weights on the fp16 matrix cores; 0 / 1 = once
};

constexpr int BK = 64;          // K granularity required by the API (k1, k2 multiples of 64)

// LDS-DMA from inline asm (common.h): with the builtin form pending, the compiler treats it as a FLAT access and drains vmcnt and
// lgkmcnt in front of the fragment reads that follow; the kernel owns its vmcnt waits and barriers anyway (+0.7-1 % on the step,
// A/B on one box; +25 % on the implicit-GEMM convolution, where it was found)
__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) { lds_dma16(g, lds_wave_base); }

// XOR swizzle of the 16-B chunk index inside a staged row (applied to the LDS-DMA source address and
// again on the fragment reads): makes the four 16-lane groups of a ds_read_b128 hit 16 distinct
// 16-B bank slots for the 16x16x32 operand map (lane -> row lane&15, chunk lane>>4).
//   128-B rows (K tile 64): chunk ^ ((row>>1) & 7)
//    64-B rows (K tile 32): chunk ^ ((-(row>>2)) & 3)
template <int BKT>
__device__ __forceinline__ int swz(int row, int chunk) {
    return BKT == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((-(row >> 2)) & 3));
}

// Stage ROWS x BKT halfs (row-major, ld elements per row) into a lane-linear LDS image.  NT threads;
// one wave-instruction covers 1 KB = 1024 / (2*BKT) rows.
template <int ROWS, int NT, int BKT>
__device__ __forceinline__ void stage_rows(const half_t* __restrict__ src, int64_t ld, int row0, int row_limit,
                                           int kofs, char* lds_tile, int wave, int lane) {
    constexpr int CPR = BKT / 8;                 // 16-B chunks per row
    constexpr int RPI = 64 / CPR;                // rows per wave-instruction
    constexpr int RPR = (NT / 64) * RPI;         // rows per round (all waves)
    constexpr int ROUNDS = ROWS / RPR;
    static_assert(ROWS % RPR == 0, "tile rows must be a multiple of the rows staged per round");
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int row = r * RPR + wave * RPI + lane / CPR;
        const int logical = swz<BKT>(row, lane % CPR);
        int grow = row0 + row;
        grow = grow < row_limit ? grow : row_limit - 1;   // clamp: rows past the edge are masked in the epilogue
        const half_t* g = src + (int64_t)grow * ld + kofs + logical * 8;
        glds16(g, lds_tile + (r * RPR + wave * RPI) * (BKT * 2));
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same with a wave-uniform run-time count (0 .. 31): a scalar branch tree in front of the immediate forms
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
    switch (n) {
#define PCD_W(k) case k: wait_vmcnt<k>(); break;
        PCD_W(0) PCD_W(1) PCD_W(2) PCD_W(3) PCD_W(4) PCD_W(5) PCD_W(6) PCD_W(7) PCD_W(8) PCD_W(9) PCD_W(10) PCD_W(11) PCD_W(12) PCD_W(13) PCD_W(14) PCD_W(15)
        PCD_W(16) PCD_W(17) PCD_W(18) PCD_W(19) PCD_W(20) PCD_W(21) PCD_W(22) PCD_W(23) PCD_W(24) PCD_W(25) PCD_W(26) PCD_W(27) PCD_W(28) PCD_W(29) PCD_W(30)
#undef PCD_W
        default: wait_vmcnt<0>(); break;          // (never reached with the counts of this file; waiting for everything is always safe)
    }
}

// BM x BN x 64 tile, WGM x WGN waves (each wave (BM/WGM) x (BN/WGN)), STAGES LDS buffers.
// STAGES == 2: one tile prefetched, plain barrier.  STAGES >= 3: STAGES-1 tiles prefetched, the
// LDS-DMA of the younger ones stays in flight across the (raw) barrier behind a counted vmcnt.
template <int BM, int BN, int WGM, int WGN, int STAGES, int EPI, int BKT = 64>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_f16_kernel(GemmParams p) {
    constexpr int ROWB = BKT * 2;                // bytes per staged row
    constexpr int KS = BKT / 32;                 // 32-deep MFMA steps per K tile
    // cross-tile prefetch (next tile's first K tile requested before this tile's epilogue): only where the
    // registers allow it; the 256x256 store variant would spill (128 accumulators + epilogue temporaries)
    constexpr bool XPREF = (STAGES == 2) && (BM * BN <= 128 * 128);
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    constexpr int LOADS_PER_TILE = (BM + BN) / ((NT / 64) * (512 / BKT));   // global_load_lds per thread per K tile
    // Store epilogues compute the transposed product (weights as the MFMA A operand): an accumulator then
    // holds 4 CONSECUTIVE CHANNELS of one point, i.e. an 8-byte piece of an output row, and goes straight to
    // global memory (no LDS staging, no barriers).  Column-max / fp32 epilogues keep channels on the lanes.
    constexpr bool SWAP = (EPI == EPI_F16 || EPI == EPI_RESID);
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KB LDS");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;

    // Persistent blocks: block b walks tiles b, b + gridDim.x, ... (n fastest, so neighbouring blocks share
    // the activation row panel).  The first K tile of the NEXT output tile is requested before the epilogue
    // of the current one, so its load latency and the store tail overlap.
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int ntiles = tiles_mn * p.splits;
    const int nk1 = p.k1 / BKT, nks = (p.k1 + p.k2) / BKT, nk = p.krep == 2 ? 2 * nks : nks;

    f32x4 acc[MI][NI];

    auto stage = [&](int m0, int n0, int kt, int buf, int split = 0) {
        char* base = smem + buf * STAGE_BYTES;
        const int64_t koff = (int64_t)split * p.k1;          // split-K: slice `split` of the reduction (k2 == 0 then)
        const int ka = kt >= nks ? kt - nks : kt;            // hi / lo weights: the second pass re-reads the sources against W's lo columns
        if (ka < nk1) stage_rows<BM, NT, BKT>(p.a1 + koff, p.lda1, m0, p.m, ka * BKT, base, wave, lane);
        else          stage_rows<BM, NT, BKT>(p.a2, p.lda2, m0, p.m, (ka - nk1) * BKT, base, wave, lane);
        stage_rows<BN, NT, BKT>(p.w + koff, p.ldw, n0, p.c, kt * BKT, base + BM * ROWB, wave, lane);
    };

    // per-lane fragment read offsets (bytes) inside a staged tile
    const int ra = wm * WM + (lane & 15);
    const int rb = wn * WN + (lane & 15);
    const int q = lane >> 4;
    int offa[KS], offb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        offa[ks] = ra * ROWB + (swz<BKT>(ra, ks * 4 + q) << 4);
        offb[ks] = BM * ROWB + rb * ROWB + (swz<BKT>(rb, ks * 4 + q) << 4);
    }

    auto compute = [&](const char* base) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + i * 16 * ROWB);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const half8*)(base + offb[ks] + j * 16 * ROWB);
            __builtin_amdgcn_s_setprio(1);   // keeps the MFMA cluster together (+2.8 % on the dominant GEMM)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // tile index -> (row panel, column tile).  Default: n fastest.  XCD-aware variant (T1; speed only): a
    // persistent grid of 256 blocks is dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2);
    // within each round of 256 tiles XCD x owns a pm x pn patch (pm*pn = 32), so its L2 streams pm + pn
    // operand panels instead of up to 32 + 1.
    auto tile_coords = [&](int tile, int& tm, int& tn, int& split) {
        split = 0;
        if (p.splits > 1) { split = tile / tiles_mn; tile -= split * tiles_mn; }
        if (gridDim.x == 256 && p.patch_pn > 0) {
            const int pn = p.patch_pn, pm = 32 / pn, xn = p.patch_xn, xm = 8 / xn;
            const int round = tile >> 8, b = tile & 255;
            const int xcd = b & 7, slot = b >> 3;
            const int sbn = p.tiles_n / (xn * pn);                   // super-blocks per row of super-blocks
            const int sb_m = round / sbn, sb_n = round - sb_m * sbn;
            tm = (sb_m * xm + xcd / xn) * pm + slot / pn;
            tn = (sb_n * xn + xcd % xn) * pn + slot % pn;
        } else {
            tm = tile / p.tiles_n;
            tn = tile - tm * p.tiles_n;
        }
    };

    int it = 0;                      // K tiles consumed so far by this block (LDS ring position)
    bool counted_wait = false;       // previous epilogue issued exactly MI*NI/2 stores after the prefetch
    if constexpr (XPREF) {
        int tm0, tn0, sp0;
        tile_coords(blockIdx.x, tm0, tn0, sp0);
        stage(tm0 * BM, tn0 * BN, 0, 0, sp0);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int tm, tn, sp;
    tile_coords(tile, tm, tn, sp);
    const int m0 = tm * BM, n0 = tn * BN;
    const int next = tile + (int)gridDim.x;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (STAGES == 2) {
        if constexpr (!XPREF) {
            __builtin_amdgcn_s_barrier();      // every wave is done reading the previous tile's LDS stages
            stage(m0, n0, 0, it & 1, sp);
        }
        for (int kt = 0; kt < nk; ++kt) {
            // the loads of this K tile are OLDER than the previous tile's epilogue stores: a counted wait
            // retires the loads and leaves the stores draining (vmcnt counts loads and stores in issue order)
            if (kt == 0 && counted_wait) wait_vmcnt<MI * NI / 2>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // raw: __syncthreads() would add its own vmcnt(0) while an LDS-DMA is pending
            if (kt + 1 < nk) stage(m0, n0, kt + 1, (it + 1) & 1, sp);
            else if (XPREF && next < ntiles) {
                int tm2, tn2, sp2;
                tile_coords(next, tm2, tn2, sp2);
                stage(tm2 * BM, tn2 * BN, 0, (it + 1) & 1, sp2);
            }
            compute(smem + (it & 1) * STAGE_BYTES);
            ++it;
        }
    } else {
        // prologue: STAGES-1 tiles in flight (no cross-tile prefetch in the deep-ring variant)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) stage(m0, n0, t, t, sp);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // tiles issued so far: min(kt + STAGES - 1, nk); tile kt must have landed, younger ones may fly
            const int younger = min(kt + STAGES - 1, nk) - (kt + 1);
            if (younger >= STAGES - 2) wait_vmcnt<(STAGES - 2) * LOADS_PER_TILE>();
            else if (STAGES > 3 && younger == 1) wait_vmcnt<LOADS_PER_TILE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();        // every wave's share of tile kt landed; buffer of tile kt-1 is free
            if (kt + STAGES - 1 < nk) {
                int nb = buf + STAGES - 1;
                nb = nb >= STAGES ? nb - STAGES : nb;
                stage(m0, n0, kt + STAGES - 1, nb, sp);
            }
            compute(smem + buf * STAGE_BYTES);
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
    }

    // ------------------------------------------------------------------ epilogue
    if constexpr (SWAP) {
        // accumulator element (i, j, r): point row = wm*WM + i*16 + (lane&15), channel = wn*WN + j*16 + 4*(lane>>4) + r
        const int q4 = (lane >> 4) * 4, pr = lane & 15;
        const bool one_shape = p.shape_bias != nullptr && (p.rows_per_shape % BM == 0);
        const float* sb_rows = (p.shape_bias != nullptr && !one_shape) ? p.shape_bias : nullptr;   // generic: per row
        // per-channel constants loaded once (bias + the tile's shape bias when the tile lies in one shape)
        f32x4 bv[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int col = n0 + wn * WN + j * 16 + q4;
            col = col < p.c ? col : 0;
            bv[j] = p.bias != nullptr ? *(const f32x4*)(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (one_shape) bv[j] += *(const f32x4*)(p.shape_bias + (int64_t)(m0 / p.rows_per_shape) * p.c + col);
        }
        const float lo = p.relu ? 0.f : -65504.f;      // ReLU and the fp16 saturation are one v_med3
        const bool full = (m0 + BM <= p.m) && (n0 + BN <= p.c) && sb_rows == nullptr;
        // 16-byte stores: two neighbouring 16-channel tiles exchange halves between lane rows (16-lane
        // groups) with v_permlane16_swap, so lane group q ends up with 8 consecutive channels of tile
        // j + (q&1) at channel offset 8*(q>>1).  Half the store instructions of the 8-byte form: the store
        // tail of a tile is issue-bound, not bandwidth-bound (cdna guide T21).
        const int qq = lane >> 4;
        static_assert(NI % 2 == 0, "store epilogue pairs column tiles");
        // two instances: whole tiles without per-row shape bias run straight-line code (no per-lane predicates, no
        // exec-masked branches around the stores); edge tiles and per-row biases take the predicated form
        auto store_rows = [&](auto full_c) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = m0 + wm * WM + i * 16 + pr;
                const bool row_ok = FULL || row < p.m;
                half_t* orow = p.out16 + (int64_t)(row_ok ? row : 0) * p.ldo;
                const float* sbrow = (!FULL && sb_rows != nullptr && row_ok) ? sb_rows + (int64_t)(row / p.rows_per_shape) * p.c : nullptr;
#pragma unroll
                for (int j = 0; j < NI; j += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = n0 + wn * WN + (j + t) * 16 + q4;
                        f32x4 v = acc[i][j + t] + bv[j + t];
                        if (!FULL && sbrow != nullptr && col < p.c) v += *(const f32x4*)(sbrow + col);
                        half2_ lo2, hi2;
                        lo2[0] = (half_t)__builtin_amdgcn_fmed3f(v[0], lo, 65504.f);
                        lo2[1] = (half_t)__builtin_amdgcn_fmed3f(v[1], lo, 65504.f);
                        hi2[0] = (half_t)__builtin_amdgcn_fmed3f(v[2], lo, 65504.f);
                        hi2[1] = (half_t)__builtin_amdgcn_fmed3f(v[3], lo, 65504.f);
                        pk[t][0] = __builtin_bit_cast(unsigned, lo2);
                        pk[t][1] = __builtin_bit_cast(unsigned, hi2);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    // lane group qq now holds: s0[0], s1[0] = channels +0..3 ; s0[1], s1[1] = channels +4..7
                    const int col = n0 + wn * WN + (j + (qq & 1)) * 16 + (qq >> 1) * 8;
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    if (FULL || (row_ok && col < p.c)) {
                        if constexpr (EPI == EPI_RESID) {
                            half8 ov = __builtin_bit_cast(half8, o);
                            const half8 rs = *(const half8*)(p.resid + (int64_t)row * p.ldr + col);
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = to_half_sat((float)ov[e] + (float)rs[e]);
                            o = __builtin_bit_cast(u32x4, ov);
                        }
                        *(u32x4*)(orow + col) = o;
                    }
                }
            }
        };
        if (full) store_rows(std::true_type{});
        else store_rows(std::false_type{});
        counted_wait = XPREF && (EPI == EPI_F16) && full;
        continue;
    }
    const int colq = lane & 15;
    // when a whole tile lies inside one shape the per-shape bias is just another per-column bias
    const bool tile_one_shape = p.shape_bias != nullptr && (p.rows_per_shape % BM == 0);
    const float* sb_generic = tile_one_shape ? nullptr : p.shape_bias;
    float bcol[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WN + j * 16 + colq;
        float v = (p.bias != nullptr && col < p.c) ? p.bias[col] : 0.f;
        if (tile_one_shape && col < p.c) v += p.shape_bias[(int64_t)(m0 / p.rows_per_shape) * p.c + col];
        bcol[j] = v;
    }

    if constexpr (EPI == EPI_COLMAX) {
        // values are post-ReLU (>= 0): float bits order like unsigned ints, colmax pre-zeroed
        unsigned* cm = reinterpret_cast<unsigned*>(p.colmax);
        const int wrow0 = m0 + wm * WM;
        const bool fast = (p.cm_rps % WM == 0) && (wrow0 + WM <= p.m) && sb_generic == nullptr;
        if (fast) {
            const int shape = wrow0 / p.cm_rps;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = n0 + wn * WN + j * 16 + colq;
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r] + bcol[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (q == 0 && col < p.c) atomicMax(cm + (int64_t)shape * p.c + col, __float_as_uint(mx));
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wm * WM + i * 16 + q * 4 + r;
                        const int col = n0 + wn * WN + j * 16 + colq;
                        if (row < p.m && col < p.c) {
                            float v = acc[i][j][r] + bcol[j];
                            if (sb_generic != nullptr)
                                v += sb_generic[(int64_t)(row / p.rows_per_shape) * p.c + col];
                            v = fmaxf(v, 0.f);
                            atomicMax(cm + (int64_t)(row / p.cm_rps) * p.c + col, __float_as_uint(v));
                        }
                    }
        }
    } else if constexpr (EPI == EPI_F32) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * WM + i * 16 + q * 4 + r;
                    const int col = n0 + wn * WN + j * 16 + colq;
                    if (row < p.m && col < p.c) {
                        float v = acc[i][j][r] + bcol[j];
                        if (sb_generic != nullptr)
                            v += sb_generic[(int64_t)(row / p.rows_per_shape) * p.c + col];
                        if (p.relu) v = fmaxf(v, 0.f);
                        p.out32[(int64_t)sp * p.split_out + (int64_t)row * p.ldo + col] = v;
                    }
                }
    }
    }   // tile loop
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 store-epilogue tile, whole tiles only, with the NEXT tile's first two K tiles requested BEFORE the epilogue's
// stores.  Why: a round of tiles ends in a 32 MB store burst that the chip drains in ~6 us (5.1-5.5 TB/s whatever the access
// pattern), and a load issued behind a wave's stores waits for that drain: in the generic kernel the first K tile of every tile
// does (timeline in profiles/r01_c_gemm_tile_sweep.txt: 7-12 % "wait for the first K tile" + the epilogue).  Here both LDS
// stages are refilled for the next tile while the accumulators are still being converted, the stores go out behind those loads,
// and the next tile computes two K tiles (3.2 us at K tile 64) under the drain.  What makes that affordable in registers: the
// LDS-DMA takes an SGPR base (tile and K-tile dependent, scalar) + a 32-bit VGPR offset (row-in-tile x ld + swizzled chunk,
// tile-INVARIANT), so staging for another tile costs no vector registers and no address arithmetic (the generic kernel keeps
// sixteen 64-bit pointers and spills 59 registers with a cross-tile prefetch).  Measured (tools/bench_gemm_xp.py, profiles/r03_k): one K tile
// ahead beats the generic kernel on every store shape (+3 ... +15 %) and on the column-max GEMM (+4 %); two K tiles ahead (the second
// one behind an extra barrier) is better only at K = 1024, C = 512 and worse on the wide layers; converting the tile to fp16 and
// storing it in pieces during the next tile's first four K tiles (64 more live registers, A fragments in two halves) ran 12-30 % SLOWER;
// non-temporal output stores change nothing (forward 3.67-3.71 v. 3.63-3.65 ms).
// Requires m % 256 == 0, c % 256 == 0, one K
// source layout (k2 == 0 or lda2 == lda1), a per-shape bias only with rows_per_shape % 256 == 0; launch() falls back otherwise.
__device__ __forceinline__ void glds16_s(unsigned voff, const half_t* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// 64 lanes x 4 bytes: lane l's dword lands at lds_addr + 4 l
__device__ __forceinline__ void glds4_s(unsigned voff, const float* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// bias / per-shape bias arrive by LDS-DMA as well (both pointers non-null: the host substitutes a zero row): a compiler-visible
// vector load in the epilogue would make the compiler wait for vmcnt(0), i.e. for the prefetched K tiles it knows nothing about
// EPI_F16 / EPI_RESID: fp16 store epilogue (16 stores per thread, + 16 residual loads); EPI_COLMAX: max over the rows of a shape
// (4 atomics per wave; cm_rps % 128 == 0)
template <int EPI>
__global__ __launch_bounds__(512) void gemm_xp_kernel(GemmParams p) {
    // vector-memory operations of one epilogue, per wave (they count in vmcnt): 16 stores (+ 16 residual loads), or 4 atomics
    constexpr int NST = EPI == EPI_F16 ? 16 : (EPI == EPI_RESID ? 32 : 4);
    constexpr int BM = 256, BN = 256, BKT = 64, ROWB = 128, KS = 2, WGN = 4;
    constexpr int WM = 128, WN = 64, MI = 8, NI = 4;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_lds[2][2][BN];      // [tile parity][bias | per-shape bias][column of the tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int nk1 = p.k1 / BKT, nk = (p.k1 + p.k2) / BKT;

    // staging: round r covers rows r * 64 + wave * 8 + lane / 8 of the 256-row operand panel, 16-byte chunk swz(row, lane % 8)
    unsigned voa[4], vow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r * 64 + wave * 8 + (lane >> 3);
        const int chunk = swz<BKT>(row, lane & 7);
        voa[r] = (unsigned)(row * (int)p.lda1 * 2 + chunk * 16);
        vow[r] = (unsigned)(row * (int)p.ldw * 2 + chunk * 16);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto stage = [&](int m0, int n0, int kt, int buf) __attribute__((always_inline)) {
        const half_t* ab = kt < nk1 ? p.a1 + (int64_t)m0 * p.lda1 + kt * BKT : p.a2 + (int64_t)m0 * p.lda2 + (kt - nk1) * BKT;
        const half_t* wb = p.w + (int64_t)n0 * p.ldw + kt * BKT;
        const unsigned la = lds0 + buf * STAGE_BYTES + wave * 8 * ROWB;
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16_s(voa[r], ab, la + r * 64 * ROWB);
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16_s(vow[r], wb, la + BM * ROWB + r * 64 * ROWB);
    };
    // the tile's bias columns of this wave's column quarter (waves with the same wn write the same 256 bytes: benign) -> LDS, 2 pieces
    const unsigned lds_b = (unsigned)(size_t)&bias_lds[0][0][0];
    auto stage_bias = [&](int m0, int n0, int par) __attribute__((always_inline)) {
        const unsigned dst = lds_b + par * (2 * BN * 4) + wn * WN * 4;
        glds4_s((unsigned)(lane * 4), p.bias + n0 + wn * WN, dst);
        glds4_s((unsigned)(lane * 4), p.shape_bias + (int64_t)(m0 / p.rows_per_shape) * p.c + n0 + wn * WN, dst + BN * 4);
    };
    auto tile_coords = [&](int tile, int& tm, int& tn) {
        if (p.patch_pn > 0) {
            const int pn = p.patch_pn, pm = 32 / pn, xn = p.patch_xn, xm = 8 / xn;
            const int round = tile >> 8, b = tile & 255;
            const int xcd = b & 7, slot = b >> 3;
            const int sbn = p.tiles_n / (xn * pn);
            const int sb_m = round / sbn, sb_n = round - sb_m * sbn;
            tm = (sb_m * xm + xcd / xn) * pm + slot / pn;
            tn = (sb_n * xn + xcd % xn) * pn + slot % pn;
        } else {
            tm = tile / p.tiles_n;
            tn = tile - tm * p.tiles_n;
        }
    };

    // fragment read offsets (as in gemm_f16_kernel)
    const int ra = wm * WM + (lane & 15), rb = wn * WN + (lane & 15), q = lane >> 4;
    int offa[KS], offb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        offa[ks] = ra * ROWB + (swz<BKT>(ra, ks * 4 + q) << 4);
        offb[ks] = BM * ROWB + rb * ROWB + (swz<BKT>(rb, ks * 4 + q) << 4);
    }
    f32x4 acc[MI][NI];
    auto compute = [&](const char* base) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + i * 16 * ROWB);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const half8*)(base + offb[ks] + j * 16 * ROWB);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = EPI != EPI_COLMAX ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    int tile = blockIdx.x;
    if (tile >= tiles_mn) return;
    int tm, tn;
    tile_coords(tile, tm, tn);
    int m0 = tm * BM, n0 = tn * BN;
    stage(m0, n0, 0, 0);
    stage_bias(m0, n0, 0);
    stage(m0, n0, 1, 1);                                   // nk >= 2 (checked by the host)
    int it = 0;                                            // K tiles consumed by this block: tile `it` lives in stage it & 1
    int par = 0;                                           // parity of this block's tile count: which bias slot the tile uses
    bool behind_stores = false;                            // the staged K tiles are followed by the previous tile's 16 stores
    bool pre1 = true;                                      // K tile 1 of the current tile was staged ahead (prologue / before the stores)
    const bool deep = p.xp_depth >= 2;
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < tiles_mn;
        int m1 = 0, n1 = 0;
        if (has_next) { int tm1, tn1; tile_coords(next, tm1, tn1); m1 = tm1 * BM; n1 = tn1 * BN; }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            // vmcnt counts loads and stores in issue order: K tile kt's loads are older than K tile kt + 1's (8 per thread) and than the
            // previous tile's stores (16 per thread) when those were issued behind the prefetch
            // (issue order around a tile boundary: K tile 0 x 8, bias x 2, [K tile 1 x 8,] the epilogue's NST stores / atomics)
            if (kt == 0) { if (!behind_stores) wait_vmcnt<10>(); else if (pre1) wait_vmcnt<10 + NST>(); else wait_vmcnt<2 + NST>(); }
            else if (kt == 1 && behind_stores && pre1) wait_vmcnt<NST>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            // stage it & 1 is about to be read; the other one held K tile kt - 1 and is free now (every wave passed the barrier), except at
            // kt == 0, where it already holds K tile 1
            if (kt >= 1 || !pre1) {
                if (kt + 1 < nk) stage(m0, n0, kt + 1, (it + 1) & 1);
                else if (has_next) { stage(m1, n1, 0, (it + 1) & 1); stage_bias(m1, n1, par ^ 1); }
            }
            compute(smem + (it & 1) * STAGE_BYTES);
            ++it;
        }
        if (has_next && deep) {
            __builtin_amdgcn_s_barrier();                  // every wave is done with the last K tile's stage: it takes the next tile's K tile 1
            stage(m1, n1, 1, (it + 1) & 1);
        }
        if constexpr (EPI == EPI_COLMAX) {
            // values are post-ReLU (>= 0): float bits order like unsigned ints, colmax pre-zeroed; the wave's 128 rows lie in one shape
            unsigned* cm = reinterpret_cast<unsigned*>(p.colmax);
            const int shape = (m0 + wm * WM) / p.cm_rps, colq = lane & 15;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * WN + j * 16 + colq;
                const float b = bias_lds[par][0][cl] + bias_lds[par][1][cl];
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r] + b);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (q == 0) atomicMax(cm + (int64_t)shape * p.c + n0 + cl, __float_as_uint(mx));
            }
        } else {
        // ---- store epilogue (whole tile): accumulator element (i, j, r) = point row wm*WM + i*16 + (lane & 15), channel wn*WN + j*16 + 4 (lane >> 4) + r
        {
            const int q4 = (lane >> 4) * 4, pr = lane & 15, qq = lane >> 4;
            f32x4 bv[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bv[j] = *(const f32x4*)&bias_lds[par][0][wn * WN + j * 16 + q4] + *(const f32x4*)&bias_lds[par][1][wn * WN + j * 16 + q4];
            const float lo = p.relu ? 0.f : -65504.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                half_t* orow = p.out16 + (int64_t)(m0 + wm * WM + i * 16 + pr) * p.ldo;
#pragma unroll
                for (int j = 0; j < NI; j += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 v = acc[i][j + t] + bv[j + t];
                        half2_ lo2, hi2;
                        lo2[0] = (half_t)__builtin_amdgcn_fmed3f(v[0], lo, 65504.f);
                        lo2[1] = (half_t)__builtin_amdgcn_fmed3f(v[1], lo, 65504.f);
                        hi2[0] = (half_t)__builtin_amdgcn_fmed3f(v[2], lo, 65504.f);
                        hi2[1] = (half_t)__builtin_amdgcn_fmed3f(v[3], lo, 65504.f);
                        pk[t][0] = __builtin_bit_cast(unsigned, lo2);
                        pk[t][1] = __builtin_bit_cast(unsigned, hi2);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    const int col = n0 + wn * WN + (j + (qq & 1)) * 16 + (qq >> 1) * 8;
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    if constexpr (EPI == EPI_RESID) {
                        // (a compiler-visible load: the compiler's own wait for it also waits for the older prefetch, which is at most early)
                        half8 ov = __builtin_bit_cast(half8, o);
                        const half8 rs = *(const half8*)(p.resid + (int64_t)(m0 + wm * WM + i * 16 + pr) * p.ldr + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = to_half_sat((float)ov[e] + (float)rs[e]);
                        o = __builtin_bit_cast(u32x4, ov);
                    }
                    *(u32x4*)(orow + col) = o;
                }
            }
        }
        }
        if (!has_next) break;
        behind_stores = true;
        pre1 = deep;
        par ^= 1;
        tile = next; m0 = m1; n0 = n1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_xw_kernel: gemm_xp_kernel's tile, waves and tile walk with the WEIGHT fragments straight from global memory.
// In the kernels above a K tile costs the LDS port 8 waves x 24 KB of fragment reads + 64 KB of LDS-DMA fills = 256 KB = 2048 cycles at
// 128 B per cycle -- exactly the 2048 cycles the K tile's 2 x 64 MFMAs per SIMD take: the LDS port is as busy as the matrix pipe, and any
// hiccup of either stalls both (the four-wave assembly kernel of the yardstick moves 192 KB per K tile: profiles/r04_l).  Here only the
// activation panel goes through LDS (32-KB stages: 8 x 16 KB of reads + 32 KB of fills = 160 KB per K tile, 61 % of the port); a wave's
// eight weight fragments of a K tile are ONE contiguous 8 KB of a fragment-order copy of W (pcd_gemm_pack_wfrag: [256-column tile][K tile]
// [wave column][k step][16-column block][lane][8]) and come by eight coalesced 1-KB global loads into the registers the MFMAs read --
// 64 KB per K tile and CU through L1 = 32 B per cycle, half its rate; the two waves that share a wave column ask for the same lines at
// the same time.  A k step's four registers-of-four are reloaded with the NEXT K tile's fragments as soon as its MFMAs are issued (asm
// loads, counted vmcnt: top of a K tile vmcnt(4) = the activation pieces and the first k step's weights have landed; in front of the
// second k step vmcnt(8)).  Column-max epilogue only (the store epilogue's registers leave no room for weights in flight across it).
__device__ __forceinline__ half8 gload16_v(const half_t* g) {
    half8 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g) : "memory");
    return v;
}

// SPLIT: the 32 LDS-DMA pieces of a K tile's activation panel are requested by ONE wave of each SIMD (waves 0-3 for even K tiles, 4-7 for odd ones: 8 pieces each)
// instead of 4 pieces by every wave -- while one wave of a SIMD issues pieces (60-180 cycles each) its partner already issues MFMAs (csrc/wideffn.hip: +11 % there)
// WLATE (a diagnostic, pcd_gemm_set_config(33)): k step 0's weight registers are reloaded at the END of the K tile (lead ~0.05 K tile instead of ~0.5): if the K loop
// does not slow down, the weights are not what its top wait waits for
template <bool MID = false, bool SPLIT = false, bool WLATE = false>
__global__ __launch_bounds__(512) void gemm_xw_kernel(GemmParams p, const half_t* __restrict__ wfrag) {
    constexpr int NST = 4;                                 // the epilogue's atomics per wave (they count in vmcnt)
    constexpr int BM = 256, BN = 256, BKT = 64, ROWB = 128, WGN = 4;
    constexpr int WM = 128, WN = 64, MI = 8, NI = 4;
    constexpr int STAGE_BYTES = BM * ROWB;                 // activation panel only
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_lds[2][2][BN];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int nk1 = p.k1 / BKT, nk = (p.k1 + p.k2) / BKT;

    // staging: a piece = 8 rows x 128 bytes; by default wave w moves rows r * 64 + w * 8 + lane / 8 (r < 4) of the 256-row panel; SPLIT: the waves of the issuing
    // group (w' = wave & 3) move rows r * 32 + w' * 8 + lane / 8 (r < 8).  The swizzle does not see the r term, so a lane's offsets differ by a scalar only.
    constexpr int NPIECE = SPLIT ? 8 : 4, RSTEP = SPLIT ? 32 : 64;
    const int srow = (SPLIT ? (wave & 3) : wave) * 8 + (lane >> 3);
    const unsigned voa = (unsigned)(srow * (int)p.lda1 * 2 + swz<BKT>(srow, lane & 7) * 16);
    const unsigned lds0 = (unsigned)(size_t)smem;
    // itn = the workgroup's running index of the K tile being requested (SPLIT: group itn & 1 requests it)
    auto stage = [&](int m0, int kt, int buf, int itn) __attribute__((always_inline)) {
        if (SPLIT && (wave >> 2) != (itn & 1)) return;
        const half_t* ab = kt < nk1 ? p.a1 + (int64_t)m0 * p.lda1 + kt * BKT : p.a2 + (int64_t)m0 * p.lda2 + (kt - nk1) * BKT;
        const unsigned la = lds0 + buf * STAGE_BYTES + (SPLIT ? (wave & 3) : wave) * 8 * ROWB;
        // (the base is wave-uniform; say so: the divergence analysis loses it through the tile walk below and would hand the asm a VGPR pair)
        const uint64_t av = (uint64_t)ab;
        const half_t* abu = (const half_t*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(av >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((int)av));
#pragma unroll
        for (int r = 0; r < NPIECE; ++r) glds16_s(voa, abu + (int64_t)(r * RSTEP) * p.lda1, la + r * RSTEP * ROWB);
    };
    const unsigned lds_b = (unsigned)(size_t)&bias_lds[0][0][0];
    auto stage_bias = [&](int m0, int n0, int par) __attribute__((always_inline)) {
        const unsigned dst = lds_b + par * (2 * BN * 4) + wn * WN * 4;
        glds4_s((unsigned)(lane * 4), p.bias + n0 + wn * WN, dst);
        glds4_s((unsigned)(lane * 4), p.shape_bias + (int64_t)(m0 / p.rows_per_shape) * p.c + n0 + wn * WN, dst + BN * 4);
    };
    auto tile_coords = [&](int tile, int& tm, int& tn) {
        if (p.patch_pn > 0) {
            const int pn = p.patch_pn, pm = 32 / pn, xn = p.patch_xn, xm = 8 / xn;
            const int round = tile >> 8, b = tile & 255;
            const int xcd = b & 7, slot = b >> 3;
            const int sbn = p.tiles_n / (xn * pn);
            const int sb_m = round / sbn, sb_n = round - sb_m * sbn;
            tm = (sb_m * xm + xcd / xn) * pm + slot / pn;
            tn = (sb_n * xn + xcd % xn) * pn + slot % pn;
        } else {
            tm = tile / p.tiles_n;
            tn = tile - tm * p.tiles_n;
        }
    };
    // this wave's weight fragments of K tile kt of column tile tn: 8 KB at wfrag + ((tn * nk + kt) * 4 + wn) * 4096 halfs: [ks][j][lane][8]
    auto wbase = [&](int tn_, int kt) __attribute__((always_inline)) {
        return wfrag + ((int64_t)(tn_ * nk + kt) * 4 + wn) * 4096 + lane * 8;
    };

    const int ra = wm * WM + (lane & 15), q = lane >> 4;
    int offa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offa[ks] = ra * ROWB + (swz<BKT>(ra, ks * 4 + q) << 4);
    f32x4 acc[MI][NI];
    half8 wq[2][NI];                                       // the current K tile's weight fragments, per k step

    int tile = blockIdx.x;
    if (tile >= tiles_mn) return;
    // diagnostic (tools/bench_gemm_clock.py; p.out32 is unused by this epilogue): shader-clock and 100-MHz stamps around the workgroup's whole tile walk
    unsigned long long stamp_c0 = 0, stamp_r0 = 0;
    if (p.out32 != nullptr) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
    int tm, tn;
    tile_coords(tile, tm, tn);
    int m0 = tm * BM, n0 = tn * BN;
    // prologue: K tile 0's activation pieces (4), bias rows (2), weight fragments of both k steps (8)
    stage(m0, 0, 0, 0);
    stage_bias(m0, n0, 0);
    {
        const half_t* wb = wbase(tn, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NI; ++j) wq[ks][j] = gload16_v(wb + (ks * 4 + j) * 512);
    }
    int it = 0;
    int par = 0;
    bool behind_atomics = false;
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < tiles_mn;
        int m1 = m0, n1 = n0, tn1 = tn;                    // without a next tile the look-ahead re-requests this tile's first K tile (harmless)
        if (has_next) { int tm1; tile_coords(next, tm1, tn1); m1 = tm1 * BM; n1 = tn1 * BN; }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            // issue order per K tile: [A pieces of K tile + 1: 4] [weights of K tile + 1, k step 0: 4, behind k step 0's MFMAs] [... k step 1: 4, behind
            // k step 1's MFMAs].  Here the youngest four loads are this K tile's k-step-1 weights (+ a tile's NST atomics and 2 bias pieces when the
            // previous output tile's epilogue came in between)
            if (kt == 0 && behind_atomics) wait_vmcnt<4 + NST>(); else wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();
            const bool last = kt + 1 == nk;
            const int mN = last ? m1 : m0, tnN = last ? tn1 : tn, ktN = last ? 0 : kt + 1;
            // (MID: the next K tile's activation pieces behind k step 0's MFMAs instead of right behind the barrier, where both waves of a SIMD would be issuing
            // LDS-DMA pieces -- 60-180 cycles of issue each -- with the matrix pipe idle; same issue ORDER, so every counted wait keeps its count)
            if constexpr (!MID) {
                stage(mN, ktN, (it + 1) & 1, it + 1);
                if (last) stage_bias(m1, n1, par ^ 1);
            }
            const half_t* wb = wbase(tnN, ktN);
            const char* base = smem + (it & 1) * STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + i * 16 * ROWB);
                if (ks == 1) {                                                         // k step 1's weights: behind them the A pieces (4; SPLIT: 8 or none) [+ bias 2] and k step 0's reload (4)
                    if constexpr (SPLIT && WLATE) {                                      // (k step 0's reload has not been issued yet)
                        const bool issuer = (wave >> 2) == ((it + 1) & 1);
                        if (issuer) { if (last) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
                        else { if (last) wait_vmcnt<2>(); else wait_vmcnt<0>(); }
                    } else if constexpr (SPLIT) {
                        const bool issuer = (wave >> 2) == ((it + 1) & 1);
                        if (issuer) { if (last) wait_vmcnt<14>(); else wait_vmcnt<12>(); }
                        else { if (last) wait_vmcnt<6>(); else wait_vmcnt<4>(); }
                    } else {
                        if (last) wait_vmcnt<10>(); else wait_vmcnt<8>();
                    }
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], wq[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MID) {
                    if (ks == 0) {
                        stage(mN, ktN, (it + 1) & 1, it + 1);
                        if (last) stage_bias(m1, n1, par ^ 1);
                    }
                }
                // this k step's registers take the next K tile's fragments (the MFMAs that read them have been issued)
                if constexpr (WLATE) {
                    if (ks == 1) {
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int j = 0; j < NI; ++j) wq[k2][j] = gload16_v(wb + (k2 * 4 + j) * 512);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NI; ++j) wq[ks][j] = gload16_v(wb + (ks * 4 + j) * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ++it;
        }
        {
            unsigned* cm = reinterpret_cast<unsigned*>(p.colmax);
            const int shape = (m0 + wm * WM) / p.cm_rps, colq = lane & 15;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * WN + j * 16 + colq;
                const float b = bias_lds[par][0][cl] + bias_lds[par][1][cl];
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r] + b);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (q == 0) atomicMax(cm + (int64_t)shape * p.c + n0 + cl, __float_as_uint(mx));
            }
        }
        if (!has_next) break;
        behind_atomics = true;
        par ^= 1;
        tile = next; m0 = m1; n0 = n1; tn = tn1;
    }
    wait_vmcnt<0>();                                       // the look-ahead's loads: nothing may land after the wave has gone
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(wq[ks][j]));
    if (p.out32 != nullptr && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.out32) + blockIdx.x * 2;
        o[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        o[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_xs_kernel: the STORE GEMM on gemm_xw_kernel's structure, with most of the output tile leaving through LDS during the next tile's K loop.
// Why (profiles/r04_l, r05_a): a round of 256 x 256 output tiles ends in a 32 MB store burst (16 x 1 KB per wave) during which no workgroup
// computes -- gemm_xp_kernel hides two K tiles of the next tile under it and still pays ~4 us per tile at K = 1024 (514 us against 399 us for
// the same K loop without stores); dripping the converted tile from REGISTERS needs 64 registers nobody has.  With the weights out of LDS
// (fragment-order copy, straight into the MFMA operand registers as in gemm_xw_kernel) the activation ring is 64 KB and 88 KB of LDS are
// free: the epilogue stores ND = 5 (6) of a wave's sixteen 1-KB chunks at once and parks NDRIP = 11 (10) in the wave's own LDS slots; the
// next tile's K tiles read R = 1 (2) chunks back per K tile (ds_read_b128, 4 transient registers) and store them from there -- one 1-KB store
// per wave every 1.5 us instead of sixteen at once.  The waves keep their vmcnt waits counted: the drip stores sit at a fixed place of the
// K tile's issue order ([activation pieces x 4 (+ bias x 2)] [weights of k step 0 x 4] [drip x R] [weights of k step 1 x 4]) and each wait names
// the operations younger than the ones it needs.  A store must have completed one K tile after its issue (vmcnt retires in order).
// Output addresses: SGPR base (tile, chunk) + one tile-invariant 32-bit lane offset (global_store_dwordx4 v, v[4], s[2]): no per-row pointers.
// Same fp16 products summed in fp32 in the same order as gemm_xp_kernel<EPI_F16>: bitwise the same output.
// Requires whole tiles (m, c % 256 == 0), one A layout, bias + per-shape bias rows (zero rows stand in), nk >= 6.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gstore16_s(unsigned voff, u32x4_t data, const half_t* sbase) {
    // (s_nop: a VALU write of the data registers needs one wait state behind a > 8-byte store; the compiler does not look into the asm)
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 0" ::"v"(voff), "v"(data), "s"(sbase) : "memory");
}

__device__ __forceinline__ const half_t* uniform_ptr(const half_t* q) {
    const uint64_t v = (uint64_t)q;
    return (const half_t*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

template <int R, bool ABL = false, bool MID = false, bool SPLIT = true>
__global__ __launch_bounds__(512) void gemm_xs_kernel(GemmParams p, const half_t* __restrict__ wfrag) {
    constexpr int NCH = 16;                                // 1-KB output chunks of a wave's 128 x 64 tile: chunk c = rows i = c / 2 (16 each), column pair c % 2
    constexpr int NDRIP = R == 1 ? 11 : 10;                // chunks parked in LDS, R of them stored per K tile of the next tile
    constexpr int ND = NCH - NDRIP;                        // chunks stored straight from the epilogue (they count in vmcnt)
    constexpr int BM = 256, BN = 256, BKT = 64, ROWB = 128, WGN = 4;
    constexpr int WM = 128, WN = 64, MI = 8, NI = 4;
    constexpr int STAGE_BYTES = BM * ROWB;                 // activation panel only
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_lds[2][2][BN];
    __shared__ __attribute__((aligned(16))) char drip[8 * NDRIP * 1024];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int nk1 = p.k1 / BKT, nk = (p.k1 + p.k2) / BKT;
    // Scalar registers are this kernel's scarce resource (the compiler moves scalar values it cannot hold into VECTOR registers, and there are none to
    // spare next to 128 accumulators + 64 operand registers + the drip): leading dimensions as 32-bit values (the host checks the ranges), per-tile bases
    // formed once, and the K loop's address arithmetic kept dependent on an opaque copy of its counter so that it is not hoisted into more live values.
    const int lda = (int)p.lda1, ldo = (int)p.ldo;
    const int abl = ABL ? p.xp_depth : 0;                  // timing ablations (a separate instance: the product kernel carries none of it) (pcd_gemm_set_config(16 + bits); outputs are then WRONG): 1 no direct stores, 2 no drip
                                                           // stores, 4 no LDS parking, 8 no epilogue arithmetic

    // staging: a piece = 8 rows x 128 bytes; 16-byte chunk swz(row, lane % 8), and the swizzle does not see the r term below.  SPLIT (default): the 32 pieces of a K
    // tile are requested by ONE wave of each SIMD -- waves 0-3 for even K tiles of the workgroup's run, 4-7 for odd ones, rows r * 32 + (wave & 3) * 8 + lane / 8, r < 8 --
    // so that its SIMD partner issues MFMAs meanwhile (a piece holds its wave's issue for 60-180 cycles; +1.5-2 % on the K loop, profiles/r05_d); otherwise every
    // wave requests rows r * 64 + wave * 8 + lane / 8, r < 4.
    constexpr int NPIECE = SPLIT ? 8 : 4, RSTEP = SPLIT ? 32 : 64;
    const int srow = (SPLIT ? (wave & 3) : wave) * 8 + (lane >> 3);
    const unsigned voa = (unsigned)(srow * lda * 2 + swz<BKT>(srow, lane & 7) * 16);
    const unsigned lds0 = (unsigned)(size_t)smem + (SPLIT ? (wave & 3) : wave) * 8 * ROWB;
    // K tile kt of the panel whose row-0 bases are (r1, r2): r1 = a1 + m0 * lda, r2 = a2 + m0 * lda - k1 (so that r2 + kt * 64 is K tile kt - nk1 of a2);
    // itn = the workgroup's running index of the K tile being requested
    auto stage = [&](const half_t* r1, const half_t* r2, int kt, int buf, int itn) __attribute__((always_inline)) {
        if (SPLIT && (wave >> 2) != (itn & 1)) return;
        const half_t* ab = uniform_ptr((kt < nk1 ? r1 : r2) + kt * BKT);
        const unsigned la = lds0 + buf * STAGE_BYTES;
#pragma unroll
        for (int r = 0; r < NPIECE; ++r) glds16_s(voa, ab + (int64_t)(r * RSTEP) * lda, la + r * RSTEP * ROWB);
    };
    const unsigned lds_b = (unsigned)(size_t)&bias_lds[0][0][0] + wn * WN * 4;
    auto stage_bias = [&](int m0, int n0, int par) __attribute__((always_inline)) {
        const unsigned dst = lds_b + par * (2 * BN * 4);
        glds4_s((unsigned)(lane * 4), p.bias + n0 + wn * WN, dst);
        glds4_s((unsigned)(lane * 4), p.shape_bias + (int64_t)(m0 / p.rows_per_shape) * p.c + n0 + wn * WN, dst + BN * 4);
    };
    auto tile_coords = [&](int tile, int& tm, int& tn) {
        if (p.patch_pn > 0) {
            const int pn = p.patch_pn, pm = 32 / pn, xn = p.patch_xn, xm = 8 / xn;
            const int round = tile >> 8, b = tile & 255;
            const int xcd = b & 7, slot = b >> 3;
            const int sbn = p.tiles_n / (xn * pn);
            const int sb_m = round / sbn, sb_n = round - sb_m * sbn;
            tm = (sb_m * xm + xcd / xn) * pm + slot / pn;
            tn = (sb_n * xn + xcd % xn) * pn + slot % pn;
        } else {
            tm = tile / p.tiles_n;
            tn = tile - tm * p.tiles_n;
        }
        // (wave-uniform by construction; say so -- the divergence analysis loses it through the tile walk and would keep every tile-derived base in VGPRs)
        tm = __builtin_amdgcn_readfirstlane(tm);
        tn = __builtin_amdgcn_readfirstlane(tn);
    };
    // this wave's weight fragments of K tile kt of column tile tn: 8 KB at wfrag + ((tn * nk + kt) * 4 + wn) * 4096 halfs, [ks][j][lane][8]: a scalar base per
    // k step + the lane's 16-byte slot + an immediate (no per-lane 64-bit pointers)
    const unsigned vlane16 = (unsigned)(lane * 16);
    auto wload = [&](const half_t* wks, auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        half8 v;
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(vlane16), "s"(wks), "n"(J * 1024) : "memory");
        return v;
    };
    auto wload4 = [&](half8 (&dst)[NI], const half_t* wks) __attribute__((always_inline)) {
        dst[0] = wload(wks, std::integral_constant<int, 0>{});
        dst[1] = wload(wks, std::integral_constant<int, 1>{});
        dst[2] = wload(wks, std::integral_constant<int, 2>{});
        dst[3] = wload(wks, std::integral_constant<int, 3>{});
    };

    const int ra = wm * WM + (lane & 15), q = lane >> 4;
    int offa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offa[ks] = ra * ROWB + (swz<BKT>(ra, ks * 4 + q) << 4);
    // output: accumulator element (i, j, r) = point row wm*WM + i*16 + (lane & 15), channel wn*WN + j*16 + 4 (lane >> 4) + r; after the half swap a lane
    // holds 8 consecutive channels: chunk (i, jp) goes to row i*16 + (lane & 15), column (2 jp + (q & 1)) * 16 + (q >> 1) * 8 of the wave's tile
    const unsigned vout = (unsigned)((((wm * WM + (lane & 15)) * ldo) + wn * WN + (q & 1) * 16 + (q >> 1) * 8) * 2);
    char* const my_drip = drip + (wave * NDRIP * 64 + lane) * 16;
    f32x4 acc[MI][NI];
    half8 wq[2][NI];

    int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    if (tile >= tiles_mn) return;
    int tm, tn;
    tile_coords(tile, tm, tn);
    int m0 = tm * BM, n0 = tn * BN;
    const half_t* r1 = p.a1 + (int64_t)m0 * lda;
    const half_t* r2 = p.a2 + (int64_t)m0 * lda - p.k1;
    const half_t* wt = wfrag + ((int64_t)tn * nk * 4 + wn) * 4096;       // this wave's fragments of the tile's K tile 0
    stage(r1, r2, 0, 0, 0);
    stage_bias(m0, n0, 0);
    wload4(wq[0], uniform_ptr(wt));
    wload4(wq[1], uniform_ptr(wt + 2048));
    int it = 0;
    int par = 0;
    bool behind = false;                                   // this tile follows an epilogue: ND stores behind its K tile 0's operands, NDRIP chunks waiting in LDS
    const half_t* prev_out = p.out16;                      // output tile of the tile whose chunks are being dripped
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < tiles_mn;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int dnext = behind ? 0 : NDRIP;                    // next parked chunk to store
        bool prev_drip = false;                            // the previous K tile issued R drip stores
        int m1 = m0, n1 = n0;
        for (int kt = 0; kt < nk; ++kt) {
            // issue order of a K tile: [A x 4 (+ bias x 2)] [w0 x 4] [drip x R] [w1 x 4].  Top: the activation pieces and k step 0's weights of this K tile have
            // landed; younger than those: [the previous K tile's drip x R] w1 x 4 [the epilogue's ND stores in front of a tile's K tile 0]
            if (kt == 0) { if (behind) wait_vmcnt<4 + ND>(); else wait_vmcnt<4>(); }
            else if (prev_drip) wait_vmcnt<4 + R>();
            else wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();
            int ktv = kt;
            asm volatile("" : "+s"(ktv));                  // opaque: what follows is recomputed per K tile, not carried in registers across the loop
            const bool last = ktv + 1 == nk;
            const half_t* wnext;                           // this wave's fragments of the next K tile (of the next output tile behind the last one)
            if (!last) {
                wnext = wt + (int64_t)(ktv + 1) * 16384;
            } else {
                int tm1 = tm, tn1 = tn;                    // without a next tile the look-ahead re-requests this tile's first K tile (harmless)
                if (has_next) tile_coords(next, tm1, tn1);
                m1 = tm1 * BM; n1 = tn1 * BN;
                tm = tm1; tn = tn1;
                r1 = p.a1 + (int64_t)m1 * lda;
                r2 = p.a2 + (int64_t)m1 * lda - p.k1;
                wt = wfrag + ((int64_t)tn1 * nk * 4 + wn) * 4096;
                wnext = wt;
            }
            wnext = uniform_ptr(wnext);
            // the next K tile's activation pieces (+ the next tile's bias rows): at the top of the K tile, or (MID) behind k step 0's MFMAs -- an LDS-DMA piece
            // holds its wave's issue for 60-180 cycles, and right behind the barrier BOTH waves of a SIMD would be issuing pieces with the matrix pipe idle
            auto request_next = [&]() __attribute__((always_inline)) {
                stage(r1, r2, last ? 0 : ktv + 1, (it + 1) & 1, it + 1);
                if (last) stage_bias(m1, n1, par ^ 1);
            };
            if constexpr (!MID) request_next();
            const char* base = smem + (it & 1) * STAGE_BYTES;
            const bool dripping = dnext < NDRIP;
            u32x4_t dv;                                    // (R = 2: the second chunk is read once the first has been stored -- the same four registers)
            if (dripping) dv = *(const u32x4_t*)(my_drip + dnext * 1024);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + i * 16 * ROWB);
                if (ks == 1) {
                    // k step 1's weights of this K tile; younger: [the epilogue's ND stores, at kt == 0 behind one] A x 4 (SPLIT: x 8 on the requesting waves, none on
                    // the others) [+ bias x 2] w0 x 4 [drip x R]
                    if constexpr (SPLIT) {
                        const bool issuer = (wave >> 2) == ((it + 1) & 1);
                        wait_vmcnt_rt(4 + (issuer ? 8 : 0) + (last ? 2 : 0) + (dripping ? R : 0) + (kt == 0 && behind ? ND : 0));
                    } else {
                        if (kt == 0 && behind) { if (dripping) wait_vmcnt<8 + ND + R>(); else wait_vmcnt<8 + ND>(); }
                        else if (last) { if (dripping) wait_vmcnt<10 + R>(); else wait_vmcnt<10>(); }
                        else { if (dripping) wait_vmcnt<8 + R>(); else wait_vmcnt<8>(); }
                    }
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks][j], af[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MID) { if (ks == 0) request_next(); }
                wload4(wq[ks], wnext + ks * 2048);         // this k step's registers take the next K tile's fragments (the MFMAs that read them have been issued)
                if (ks == 0 && dripping) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int c = ND + dnext + r;
                        if (r > 0) dv = *(const u32x4_t*)(my_drip + (dnext + r) * 1024);
                        if (abl & 2) { asm volatile("global_load_dword %0, %1, %2" : "=v"(dv[0]) : "v"(vlane16), "s"(wnext) : "memory"); continue; }   // (same vmcnt count)
                        gstore16_s(vout, dv, uniform_ptr(prev_out + (int64_t)((c >> 1) * 16) * ldo + (c & 1) * 32));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            prev_drip = dripping;
            if (dripping) dnext += R;
            ++it;
        }
        // ---- epilogue: bias + ReLU + fp16, half swap -> sixteen 16-byte-per-lane chunks; ND stored now, NDRIP parked for the next tile's K loop
        const half_t* tile_out = uniform_ptr(p.out16 + (int64_t)m0 * ldo + n0);
        if (abl & 8) {
            // (keep the accumulators alive and the vmcnt count: ND dummy loads)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
#pragma unroll
            for (int c = 0; c < (has_next ? ND : NCH); ++c) { unsigned junk; asm volatile("global_load_dword %0, %1, %2" : "=v"(junk) : "v"(vlane16), "s"(tile_out) : "memory"); }
        } else {
            const int q4 = q * 4;
            f32x4 bv[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bv[j] = *(const f32x4*)&bias_lds[par][0][wn * WN + j * 16 + q4] + *(const f32x4*)&bias_lds[par][1][wn * WN + j * 16 + q4];
            const float lo = p.relu ? 0.f : -65504.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; j += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 v = acc[i][j + t] + bv[j + t];
                        half2_ lo2, hi2;
                        lo2[0] = (half_t)__builtin_amdgcn_fmed3f(v[0], lo, 65504.f);
                        lo2[1] = (half_t)__builtin_amdgcn_fmed3f(v[1], lo, 65504.f);
                        hi2[0] = (half_t)__builtin_amdgcn_fmed3f(v[2], lo, 65504.f);
                        hi2[1] = (half_t)__builtin_amdgcn_fmed3f(v[3], lo, 65504.f);
                        pk[t][0] = __builtin_bit_cast(unsigned, lo2);
                        pk[t][1] = __builtin_bit_cast(unsigned, hi2);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    const u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
                    const int c = i * 2 + (j >> 1);
                    if (c < ND || !has_next) {
                        if (abl & 1) { unsigned junk; asm volatile("global_load_dword %0, %1, %2" : "=v"(junk) : "v"(vlane16), "s"(tile_out), "v"(o) : "memory"); }
                        else gstore16_s(vout, o, tile_out + (int64_t)(i * 16) * ldo + (j >> 1) * 32);
                    } else if (!(abl & 4)) *(u32x4_t*)(my_drip + (c - ND) * 1024) = o;
                    else asm volatile("" ::"v"(o));
                }
            }
        }
        if (!has_next) break;
        behind = true;
        prev_out = tile_out;
        par ^= 1;
        tile = next; m0 = m1; n0 = n1;
    }
    wait_vmcnt<0>();                                       // the look-ahead's loads: nothing may land after the wave has gone
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(wq[ks][j]));
}

// fragment-order copy of W [c][ldw] for gemm_xw_kernel: one thread per 16-byte piece
__global__ __launch_bounds__(256) void gemm_pack_wfrag_kernel(const half_t* __restrict__ w, int64_t ldw, int k, int c, half_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = k / 64;
    const int64_t total = (int64_t)(c / 256) * nk * 4 * 2 * 4 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63); int64_t r = idx >> 6;
    const int j = (int)(r & 3); r >>= 2;
    const int ks = (int)(r & 1); r >>= 1;
    const int wn = (int)(r & 3); r >>= 2;
    const int kt = (int)(r % nk); const int tn = (int)(r / nk);
    const int col = tn * 256 + wn * 64 + j * 16 + (lane & 15);
    *(half8*)(out + idx * 8) = *(const half8*)(w + (int64_t)col * ldw + kt * 64 + ks * 32 + (lane >> 4) * 8);
}

static int g_split = 1;         // pcd_gemm_set_config(14) / (15): the fragment-order kernels' LDS-DMA pieces requested by every wave / by one wave per SIMD (default)
static int g_xs_mid = 0;        // pcd_gemm_set_config(12) / (13): gemm_xs_kernel / gemm_xw_kernel request the next K tile's activation pieces at the top of a K tile / behind k step 0's MFMAs
static void* g_wfrag_stamps = nullptr;
static int g_xw = 1;            // tuning hook (pcd_gemm_set_config(8) / (9)): callers that hold a fragment-order weight copy use gemm_xw_kernel: off / on
static int g_xs_abl = 0;        // timing ablations of gemm_xs_kernel (pcd_gemm_set_config(16 + bits)): see the kernel; outputs are wrong while set
static int g_xs = 1;            // tuning hook (pcd_gemm_set_config(10) / (11)): pcd_gemm_f16_wfrag uses gemm_xs_kernel (output dripped through LDS): off / on
static int g_xp = 1;            // tuning hook (pcd_gemm_set_config(5) / (6) / (7)): the cross-tile prefetching store kernel off / 2 K tiles ahead / 1

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// a row of zeros standing in for a missing bias / per-shape bias (the counted-vmcnt kernels stage a fixed number of bias pieces per tile): one row per DEVICE
// (a process may drive several), created under a lock; never inside a stream capture (nullptr then, and the caller takes another kernel)
static float* zero_row(int cols, hipStream_t s) {
    constexpr int kZeroCols = 16384, kMaxDev = 64;
    static float* zeros_of[kMaxDev] = {};
    static std::mutex zeros_mu;
    int dev = -1;
    if (cols > kZeroCols || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    std::lock_guard<std::mutex> lock(zeros_mu);
    if (zeros_of[dev] == nullptr) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { cap = hipStreamCaptureStatusActive; (void)hipGetLastError(); }
        float* z = nullptr;
        if (cap == hipStreamCaptureStatusNone && hipMalloc(&z, kZeroCols * sizeof(float)) == hipSuccess) {
            if (hipMemset(z, 0, kZeroCols * sizeof(float)) == hipSuccess) zeros_of[dev] = z; else (void)hipFree(z);
        }
    }
    return zeros_of[dev];
}

template <int BM, int BN, int WGM, int WGN, int STAGES, int EPI, int BKT = 64>
static int launch(const GemmParams& p0, hipStream_t s, int blocks_per_cu) {
    GemmParams p = p0;
    p.tiles_m = (int)ceil_div(p.m, BM);
    p.tiles_n = (int)ceil_div(p.c, BN);
    if (p.splits < 1) p.splits = 1;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * p.splits;
    if (tiles <= 0 || tiles > 0x7fffffff) { set_error("gemm: grid out of range"); return PCD_ERR_ARG; }
    // persistent grid: as many blocks as are resident at once, each walking tiles b, b + grid, ...
    const int64_t resident = (int64_t)num_cus() * blocks_per_cu;
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    p.patch_pn = p.patch_xn = 0;
    if (grid == 256 && tiles % 256 == 0 && p.splits == 1) {
        const int pn = p.tiles_n >= 8 ? 8 : p.tiles_n;               // 8, 4, 2 or 1 column tiles per patch
        const int xn = p.tiles_n >= 16 ? 2 : 1;
        if ((pn & (pn - 1)) == 0 && p.tiles_n % (xn * pn) == 0 && p.tiles_m % ((8 / xn) * (32 / pn)) == 0) {
            p.patch_pn = pn; p.patch_xn = xn;
        }
    }
    if constexpr (BM == 256 && BN == 256 && (EPI == EPI_F16 || EPI == EPI_COLMAX || EPI == EPI_RESID) && STAGES == 2 && BKT == 64) {
        // whole tiles, one A layout, per-shape bias only if a tile lies in one shape: the variant that refills both LDS stages for
        // the next tile before the epilogue's stores (gemm_xp_kernel)
        const bool ok = g_xp && p.krep != 2 && p.m % 256 == 0 && p.c % 256 == 0 && p.splits == 1 && (p.k1 + p.k2) >= 128 &&
                        (p.k2 == 0 || p.lda2 == p.lda1) && (p.shape_bias == nullptr || p.rows_per_shape % 256 == 0) &&
                        (EPI != EPI_COLMAX || p.cm_rps % 128 == 0) &&
                        (int64_t)255 * p.lda1 * 2 + 128 < 0x7fffffffLL && (int64_t)255 * p.ldw * 2 + 128 < 0x7fffffffLL;
        if (ok && (p.bias == nullptr || p.shape_bias == nullptr)) {
            // a zero row stands in for a missing bias (fixed number of LDS-DMA pieces per tile: the waits are counted)
            float* zeros = zero_row(p.c, s);
            if (zeros != nullptr) {
                if (p.bias == nullptr) p.bias = zeros;
                if (p.shape_bias == nullptr) { p.shape_bias = zeros; p.rows_per_shape = p.m; }       // shape 0 for every tile
            }
        }
        if (ok && p.bias != nullptr && p.shape_bias != nullptr) {
            p.xp_depth = EPI == EPI_F16 ? g_xp : 1;
            hipLaunchKernelGGL(gemm_xp_kernel<EPI>, dim3(grid), dim3(512), 0, s, p);
            PCD_CHECK_LAUNCH();
            return PCD_OK;
        }
    }
    hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WGM, WGN, STAGES, EPI, BKT>), dim3(grid), dim3(64 * WGM * WGN), 0, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

static int g_force_cfg = -1;   // tuning hook (pcd_gemm_set_config): -1 = heuristic

template <int EPI>
static int dispatch(const GemmParams& p, hipStream_t s) {
    int cfg = g_force_cfg;
    if (cfg < 0) {
        // narrow outputs (C <= 64): 128x64 tile so no MFMA work is spent on masked columns.
        // large problems: 256x256 tile, 8 waves (2x4, 128x64 per wave): half the L2->LDS bytes per FLOP
        // of the 128x128 tile; measured 1223 vs 1074 TFLOP/s on 131072x2048x4096 and faster for every
        // layer with K, C >= 256 (tools/bench_gemm.py, profiles/r01_c_gemm_tile_sweep.txt).
        if (p.c <= 64) cfg = 0;
        else if (p.m >= 16384 && p.c >= 256 && (p.k1 + p.k2) >= 256) cfg = 3;
        else cfg = 1;
    }
    GemmParams q = p;
    switch (cfg) {
        case 0: return launch<128, 64, 2, 2, 2, EPI>(q, s, 3);      // 48 KB LDS
        case 1: return launch<128, 128, 2, 2, 2, EPI>(q, s, 2);     // 64 KB LDS
        case 2: return launch<256, 128, 4, 2, 3, EPI>(q, s, 1);     // 144 KB LDS
        case 3: return launch<256, 256, 2, 4, 2, EPI>(q, s, 1);     // 128 KB LDS
        case 4: return launch<128, 128, 2, 2, 3, EPI>(q, s, 1);     // 96 KB LDS
        // measured and rejected: K tile 32 with 256x128 / 128x256 tiles (48 KB LDS, two blocks per CU, so one
        // block's store tail hides behind the other's MFMAs): 733-753 TFLOP/s on the dominant GEMM vs 1195-1260
        // for the 256x256 tile, and slower on every mid-size layer (profiles/r01_c_gemm_tile_sweep.txt).
        default: set_error("gemm: unknown config %d", cfg); return PCD_ERR_ARG;
    }
}

static int fill(const pcd_gemm_desc_t* d, GemmParams& p) {
    PCD_CHECK_ARG(d != nullptr);
    PCD_CHECK_ARG(d->a1 != nullptr && d->w != nullptr);
    PCD_CHECK_ARG(d->m > 0 && d->c > 0);
    PCD_CHECK_ARG(d->k1 > 0 && d->k1 % BK == 0);
    PCD_CHECK_ARG(d->k2 >= 0 && d->k2 % BK == 0);
    PCD_CHECK_ARG(d->k2 == 0 || d->a2 != nullptr);
    PCD_CHECK_ARG(d->lda1 >= d->k1 && (d->k2 == 0 || d->lda2 >= d->k2));
    PCD_CHECK_ARG(d->ldw >= d->k1 + d->k2);
    PCD_CHECK_ARG(d->lda1 % 8 == 0 && d->ldw % 8 == 0 && (d->k2 == 0 || d->lda2 % 8 == 0));
    PCD_CHECK_ARG(d->shape_bias == nullptr || d->rows_per_shape > 0);
    p = GemmParams{};
    p.a1 = (const half_t*)d->a1; p.lda1 = d->lda1; p.k1 = d->k1;
    p.a2 = (const half_t*)d->a2; p.lda2 = d->lda2; p.k2 = d->k2;
    p.w = (const half_t*)d->w; p.ldw = d->ldw;
    p.bias = d->bias; p.shape_bias = d->shape_bias;
    p.rows_per_shape = d->rows_per_shape > 0 ? d->rows_per_shape : 1;
    p.relu = d->relu; p.m = d->m; p.c = d->c;
    p.cm_rps = 1;
    return PCD_OK;
}

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_gemm_f16(const pcd_gemm_desc_t* d, void* out, int64_t ldo, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(out != nullptr && ldo >= d->c && ldo % 8 == 0 && d->c % 8 == 0);
    p.out16 = (half_t*)out; p.ldo = ldo;
    return dispatch<EPI_F16>(p, (hipStream_t)stream);
}

extern "C" int pcd_gemm_f16_hilo(const pcd_gemm_desc_t* d, void* out, int64_t ldo, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(out != nullptr && ldo >= d->c && ldo % 8 == 0 && d->c % 8 == 0);
    PCD_CHECK_ARG(d->ldw >= 2 * (int64_t)(d->k1 + d->k2));
    p.out16 = (half_t*)out; p.ldo = ldo; p.krep = 2;
    return dispatch<EPI_F16>(p, (hipStream_t)stream);
}

extern "C" int pcd_gemm_f16_out32(const pcd_gemm_desc_t* d, float* out, int64_t ldo, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(out != nullptr && ldo >= d->c);
    p.out32 = out; p.ldo = ldo;
    return dispatch<EPI_F32>(p, (hipStream_t)stream);
}

// split-K: slabs[s][m][c] = A[:, s*K/S : (s+1)*K/S] W[:, same]^T ; pcd_sum_slabs_f32 adds them in a fixed order
extern "C" int pcd_gemm_f16_splitk(const pcd_gemm_desc_t* d, int splits, float* slabs, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(slabs != nullptr && splits >= 1 && d->k2 == 0 && d->bias == nullptr && d->shape_bias == nullptr && d->relu == 0);
    PCD_CHECK_ARG(d->k1 % splits == 0 && (d->k1 / splits) % BK == 0);
    p.k1 = d->k1 / splits;
    p.splits = splits;
    p.out32 = slabs; p.ldo = d->c; p.split_out = (int64_t)d->m * d->c;
    return dispatch<EPI_F32>(p, (hipStream_t)stream);
}

namespace pcd {
__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslabs, int64_t rows, int cols, float* __restrict__ out,
                                 int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    float s = 0.f;
    for (int k = 0; k < nslabs; ++k) s += slabs[(int64_t)k * rows * cols + i];
    const int64_t r = i / cols;
    out[r * ldo + (i - r * cols)] = s;
}
}  // namespace pcd

extern "C" int pcd_sum_slabs_f32(const float* slabs, int nslabs, int64_t rows, int cols, float* out, int64_t ldo, void* stream) {
    PCD_CHECK_ARG(slabs && out && nslabs >= 1 && rows > 0 && cols > 0 && ldo >= cols);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)ceil_div(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, slabs, nslabs,
                       rows, cols, out, ldo);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_gemm_f16_residual(const pcd_gemm_desc_t* d, const void* resid, int64_t ldr,
                                     void* out, int64_t ldo, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(out != nullptr && resid != nullptr);
    PCD_CHECK_ARG(ldo >= d->c && ldo % 8 == 0 && ldr >= d->c && ldr % 8 == 0 && d->c % 8 == 0);
    p.out16 = (half_t*)out; p.ldo = ldo; p.resid = (const half_t*)resid; p.ldr = ldr;
    return dispatch<EPI_RESID>(p, (hipStream_t)stream);
}

extern "C" int pcd_gemm_f16_colmax(const pcd_gemm_desc_t* d, float* colmax, int rows_per_shape, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(colmax != nullptr && rows_per_shape > 0 && d->relu == 1);
    p.colmax = colmax; p.cm_rps = rows_per_shape;
    return dispatch<EPI_COLMAX>(p, (hipStream_t)stream);
}


extern "C" int pcd_gemm_pack_wfrag(const void* w, int64_t ldw, int k, int c, void* wfrag, void* stream) {
    PCD_CHECK_ARG(w && wfrag && k > 0 && k % 64 == 0 && c > 0 && c % 256 == 0 && ldw >= k && ldw % 8 == 0);
    const int64_t total = (int64_t)(c / 256) * (k / 64) * 4 * 2 * 4 * 64;
    hipLaunchKernelGGL(gemm_pack_wfrag_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)w, ldw, k, c,
                       (half_t*)wfrag);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_gemm_f16_colmax_wfrag(const pcd_gemm_desc_t* d, const void* wfrag, float* colmax, int rows_per_shape, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(wfrag != nullptr && colmax != nullptr && rows_per_shape > 0 && d->relu == 1);
    PCD_CHECK_ARG(p.m % 256 == 0 && p.c % 256 == 0 && (p.k1 + p.k2) >= 128 && (p.k2 == 0 || p.lda2 == p.lda1) && rows_per_shape % 128 == 0 &&
                  p.bias != nullptr && (p.shape_bias == nullptr || p.rows_per_shape % 256 == 0));
    p.colmax = colmax; p.cm_rps = rows_per_shape;
    p.tiles_m = p.m / 256; p.tiles_n = p.c / 256;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    PCD_CHECK_ARG(tiles >= 256 && tiles % 256 == 0 && p.shape_bias == nullptr);
    float* zeros = zero_row(p.c, (hipStream_t)stream);
    if (zeros == nullptr) { set_error("pcd_gemm_f16_colmax_wfrag: no zero row (stream capture before the first eager call, or c > 16384)"); return PCD_ERR_ARG; }
    p.shape_bias = zeros; p.rows_per_shape = p.m;
    p.patch_pn = p.patch_xn = 0;
    const int pn = p.tiles_n >= 8 ? 8 : p.tiles_n, xn = p.tiles_n >= 16 ? 2 : 1;
    if ((pn & (pn - 1)) == 0 && p.tiles_n % (xn * pn) == 0 && p.tiles_m % ((8 / xn) * (32 / pn)) == 0) { p.patch_pn = pn; p.patch_xn = xn; }
    p.out32 = (float*)g_wfrag_stamps;
    if (g_xs_mid == 2) hipLaunchKernelGGL((gemm_xw_kernel<false, true, true>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    else if (g_xs_mid) hipLaunchKernelGGL((gemm_xw_kernel<true, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    else if (g_split) hipLaunchKernelGGL((gemm_xw_kernel<false, true>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    else hipLaunchKernelGGL((gemm_xw_kernel<false, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// Store GEMM with a fragment-order weight copy (pcd_gemm_pack_wfrag of the SAME weights as d->w): gemm_xs_kernel where its launch shape holds
// (whole 256 x 256 tiles, a multiple of 256 of them, 6 or more K tiles, one A layout), pcd_gemm_f16's dispatch otherwise -- same bits either way.
extern "C" int pcd_gemm_f16_wfrag(const pcd_gemm_desc_t* d, const void* wfrag, void* out, int64_t ldo, void* stream) {
    GemmParams p;
    int rc = fill(d, p);
    if (rc) return rc;
    PCD_CHECK_ARG(out != nullptr && ldo >= d->c && ldo % 8 == 0 && d->c % 8 == 0);
    p.out16 = (half_t*)out; p.ldo = ldo;
    const int nk = (p.k1 + p.k2) / 64;
    const int64_t tiles = (int64_t)(p.m / 256) * (p.c / 256);
    bool ok = g_xs && wfrag != nullptr && p.m % 256 == 0 && p.c % 256 == 0 && nk >= 6 && (p.k2 == 0 || p.lda2 == p.lda1) &&
              (p.shape_bias == nullptr || p.rows_per_shape % 256 == 0) && tiles >= 256 && tiles % 256 == 0 && num_cus() == 256 &&
              (int64_t)255 * p.lda1 * 2 + 128 < 0x7fffffffLL && ((int64_t)255 * p.ldo + 256) * 2 < 0x7fffffffLL;
    if (ok && (p.bias == nullptr || p.shape_bias == nullptr)) {
        float* zeros = zero_row(p.c, (hipStream_t)stream);
        if (zeros == nullptr) ok = false;
        else {
            if (p.bias == nullptr) p.bias = zeros;
            if (p.shape_bias == nullptr) { p.shape_bias = zeros; p.rows_per_shape = p.m; }
        }
    }
    if (!ok) return dispatch<EPI_F16>(p, (hipStream_t)stream);
    p.tiles_m = p.m / 256; p.tiles_n = p.c / 256;
    p.patch_pn = p.patch_xn = 0;
    const int pn = p.tiles_n >= 8 ? 8 : p.tiles_n, xn = p.tiles_n >= 16 ? 2 : 1;
    if ((pn & (pn - 1)) == 0 && p.tiles_n % (xn * pn) == 0 && p.tiles_m % ((8 / xn) * (32 / pn)) == 0) { p.patch_pn = pn; p.patch_xn = xn; }
    p.xp_depth = g_xs_abl;
    if (g_xs_abl) {
        if (nk >= 12) hipLaunchKernelGGL((gemm_xs_kernel<1, true, false, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
        else hipLaunchKernelGGL((gemm_xs_kernel<2, true, false, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    } else if (g_xs_mid) {
        if (nk >= 12) hipLaunchKernelGGL((gemm_xs_kernel<1, false, true, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
        else hipLaunchKernelGGL((gemm_xs_kernel<2, false, true, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    } else if (g_split == 0) {
        if (nk >= 12) hipLaunchKernelGGL((gemm_xs_kernel<1, false, false, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
        else hipLaunchKernelGGL((gemm_xs_kernel<2, false, false, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    } else if (nk >= 12) hipLaunchKernelGGL((gemm_xs_kernel<1, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    else hipLaunchKernelGGL((gemm_xs_kernel<2, false>), dim3(256), dim3(512), 0, (hipStream_t)stream, p, (const half_t*)wfrag);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// diagnostic: the next pcd_gemm_f16_colmax_wfrag launches write (shader cycles, 100-MHz ticks) of each workgroup's tile walk to stamps[256][2] (u64); NULL = off
extern "C" int pcd_gemm_wfrag_stamps(void* stamps) { g_wfrag_stamps = stamps; return PCD_OK; }
extern "C" int pcd_gemm_wfrag_enabled(void) { return g_xw; }
extern "C" int pcd_gemm_store_wfrag_enabled(void) { return g_xs; }

extern "C" int pcd_gemm_set_config(int cfg) {
    if (cfg == 33) { g_xs_mid = 2; return PCD_OK; }                                 // diagnostic: gemm_xw_kernel reloads k step 0's weights late (see the kernel)
    PCD_CHECK_ARG(cfg >= -1 && cfg <= 31);
    if (cfg >= 16) { g_xs_abl = cfg - 16; return PCD_OK; }                          // timing ablations of gemm_xs_kernel (dev tools only)
    if (cfg == 14 || cfg == 15) { g_split = cfg - 14; return PCD_OK; }              // gemm_xw / gemm_xs: every wave requests 4 activation pieces / one wave per SIMD 8 (default)
    PCD_CHECK_ARG(cfg <= 13);
    if (cfg >= 12) { g_xs_mid = cfg - 12; return PCD_OK; }
    if (cfg >= 10) { g_xs = cfg - 10; return PCD_OK; }                              // A/B switch of gemm_xs_kernel (pcd_gemm_f16_wfrag)
    if (cfg >= 8) { g_xw = cfg - 8; return PCD_OK; }                                // A/B switch of gemm_xw_kernel (pcd_gemm_wfrag_enabled)
    if (cfg >= 5) { g_xp = cfg == 5 ? 0 : (cfg == 6 ? 2 : 1); return PCD_OK; }      // A/B switch of gemm_xp_kernel; the tile choice is left as it is
    g_force_cfg = cfg;
    return PCD_OK;
}
