// a11 as ONE launch: whole timesteps of the latent sampler (SimpleLatentUNetPointNet.forward, reference
// networks.py:1051-1086, + the DDIM loop body diffusion.py:637-645) inside one persistent kernel.
//
// Why.  At B <= 32 the step is 12 dependent Linear(+GroupNorm+ReLU) layers over 38 MB of fp16 weights; as one launch
// per layer (+ a finishing launch for the five big ones) every dependent launch costs ~3.8 us (boundary + kernel
// fill + two or three dependent memory round trips for weights that do not depend on anything), 19 launches = 72.8 us
// per step against a 4.8 us weight-streaming roofline.  Here:
//   * 256 workgroups (one per CU) stay resident for all the steps of a call; the chip's LDS (256 x 160 KB = 41.9 MB)
//     holds ALL 38.2 MB of weights: every workgroup stages its share once per call by LDS-DMA and keeps it, so
//     no weight byte is on any step's critical path;
//   * a layer = units (32 batch rows x 32 columns x a K slice, v_mfma_f32_32x32x16_f16, 4 waves split the slice) that
//     publish fp32 partial tiles (the first slice's carries the bias), + finish units (one GroupNorm group x 8 rows: slab sum in
//     slice order, GroupNorm, ReLU) that publish fp16 activations; layers whose 32-column unit holds whole GroupNorm groups over the
//     full K (enc1-3, dec1, the output head) finish inside the unit;
//   * there are NO flags, counters, atomics or fences between layers: the exchanged data carries its own validity.
//     Activations are post-ReLU fp16 (sign bit clear), partial sums are finite fp32, the state is finite fp16; the buffers are poisoned
//     with 0xFF bytes (sign bit set / NaN) and a consumer simply loads its operands with sc1 (L1-bypassing) loads
//     until no element is poison: one memory round trip per layer boundary.  THREE buffer sets rotate by step; at the
//     end of its step s a workgroup re-poisons its own output regions of set (s - 1) % 3 (every reader of that set's
//     contents has finished: step s's first layer needed the complete state published by step s - 1) for step s + 2; a reader of
//     step s + 2 has by then finished its own dec4 unit of step s + 1, which needed every workgroup's global_feat.3
//     tile of step s + 1, whose stores wait (s_waitcnt vmcnt(0)) for all earlier stores of their wave;
//   * activations are stored in MFMA-fragment order ([64-k chunk][j][lane][8 halfs]), so a consumer's operand loads are
//     1-KB coalesced instructions straight into the registers the MFMA reads: no LDS staging of activations; stores
//     are whole 128-byte lines per 8 lanes (finish units transpose through LDS for that);
//   * work is assigned by (XCD, arrival rank on that XCD) read from HW_REG_XCC_ID, not by blockIdx: the seven small
//     layers that chain the end of a step to the start of the next (dec2, dec1, output.0, output.2, update + enc1,
//     enc2, enc3's input) all sit on XCD 0 and exchange through that XCD's L2 with PLAIN stores (tools/ubench_handoff:
//     0.40 us per hand-off against 0.92 us for write-through stores across XCDs); everything else uses sc1 stores;
//   * the eight workgroups that own output.2's tiles keep their 32 columns of the latent state z (fp32, registers: 4 values per
//     thread) and apply the DDIM update to them in the tile's epilogue (same fp32 operation order as pcd_ddim_update, FMA
//     contraction off): what they publish is the NEXT step's enc1 operand, z rounded to fp16 in fragment order (poison = fp16 NaN,
//     the state is signed), so the update is 4 divisions per thread instead of a separate 32-per-thread stage between two layers.
// Every wait is bounded (s_memrealtime) and reports through a status word; the grid is exactly the number of CUs and
// the host refuses to launch on a device with fewer (pcd_latent_persist_supported).  Timeline tool:
// tools/trace_latent_persist.py (instrumented build).  Measured, B = 32: 47 us / step against 74 us for the per-layer launches.
#include <new>
#include <vector>
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace pcd {

constexpr int LP_WGS = 256, LP_THREADS = 256, LP_LAYERS = 12, LP_MAX_UNITS = 8;
constexpr int LP_SCHED = 16;                           // two streams: entries of a workgroup's merged unit order (one period)
constexpr int LP_LDS_BYTES = 160 * 1024;
constexpr int LP_DYN_LDS = LP_LDS_BYTES - 256;         // dynamic part: the kernel's static words (abort flag, wait history) take the rest
constexpr int LP_RED_BYTES = 4 * 32 * 33 * 4;          // cross-wave reduction scratch
constexpr int LP_CTRL_BYTES = 256;
constexpr unsigned LP_TIMEOUT_TICKS = 20000000u;       // 0.2 s of the 100 MHz s_memrealtime clock

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct LpLayer {
    const half_t* w; int ldw;          // fp16 [C][K1+K2]
    const float* bias; const float* gamma; const float* beta;
    int k1, k2;                        // widths of the two input sources (K2 = 0: one source)
    int src1_off, src2_off;            // byte offsets (inside a set) of the sources' fp16 fragment-major buffers (src2: -1 = none)
    int c, gsz, mode;                  // mode 0: GroupNorm(gsz)+ReLU -> fp16; 1: ReLU -> fp16; 2: identity = eps -> DDIM update -> next state, signed fp16
    int slabs;                         // 0: the unit finishes the layer itself; S >= 1: S K-slices of fp32 partial tiles + finish units
    int out_off;                       // byte offset (inside a set) of this layer's output, fp16 fragment-major
    int slab_off;                      // byte offset (inside a set) of the fp32 slabs [S][32][C]
    int slab_local;                    // 1: every GroupNorm group's partial tiles are produced and finished on ONE XCD: plain stores for them too
    int out_local;                     // 1: producers and consumers of this layer's outputs all sit on XCD 0: plain stores (the XCD's L2 is
                                       //    their meeting point: ~0.4 us per hand-off instead of ~0.9 through the fabric); 0: sc1 write-through
};

enum { LP_END = 0, LP_GEMM = 1, LP_FINISH = 2 };

struct LpUnit {
    int kind, layer;
    int col0, ct;                      // gemm: first output column, number of 32-column tiles (1 or 2)
    int chunk0, nchunks;               // gemm: K slice in 64-k chunks of the concatenated input
    int slice;                         // gemm: slab index
    int lds_w;                         // gemm: byte offset of the unit's weight images in LDS
    int group, row0;                   // finish: GroupNorm group, first row
    int head;                          // gemm on the latent state (enc1): operand = the previous step's z16, bias = the step's time row
    int lds_p;                         // byte offset in LDS of the unit's constants (its columns only): gemm that finishes its layer
                                       // bias | gamma | beta, slice-0 gemm of a layer with partial tiles bias, finish gamma | beta; or -1
    int pad[4];
};

struct LpArgs {
    const LpLayer* layers;             // [LP_LAYERS]
    const LpUnit* units;               // [LP_WGS][LP_MAX_UNITS]
    char* ws;                          // ctrl | stream 0: sets 0, 1, 2 | stream 1: sets 0, 1, 2
    int set_bytes;
    float* z;                          // fp32 [batch][256], in/out (DDIM mode) or in (forward mode)
    float* x0;                         // fp32 [batch][256] or null
    float* eps_out;                    // forward mode: fp32 [batch][256]
    int batch;
    const float* tb_table; int tb_elems;      // [T][128] (or one row in forward mode)
    const float* rates; int rate_width, rate_stride, T;     // (4, T, R) fp32
    int* counter;                      // device step counter (pcd_step_select semantics), or null: k = 0
    int nsteps;
    int forward_only;
    const int* sched;                  // two streams (batch > 32): [LP_WGS][LP_SCHED] merged unit order of one period, else null
    int sleep;                         // s_sleep argument between polls
    int predict;                       // sleep through most of the wait the previous step measured before probing
    int fault_wg, fault_step;          // fault injection (tests): workgroup `fault_wg` leaves at the start of step `fault_step` (-1: off)
    unsigned* trace;                   // diagnostic build only: [wg][step][unit][8] s_memrealtime stamps (enter, operands in, stored, epilogue computed, operands in of waves 0..3)
    int trace_steps;
};

// ------------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ int lp_swz(int row, int ch) { return ch ^ ((row >> 1) & 7); }

// LDS-DMA of rows [row0, row0 + 32) x 128 bytes at halfs column k0 of w[rows][ld] into a swizzled 4-KiB image
// (instruction i of 4 covers rows 8 i .. 8 i + 7); same image format as csrc/skinny.hip
__device__ __forceinline__ void lp_stage(const half_t* base, int64_t ld, int row0, int k0, char* img, int i, int lane) {
    const int row = 8 * i + (lane >> 3), ch = lane & 7;
    lds_dma16(base + (int64_t)(row0 + row) * ld + k0 + lp_swz(row, ch) * 8, img + i * 1024);
}

__device__ __forceinline__ half8 lp_wfrag(const char* img, int row, int hh, int j) {
    return *(const half8*)(img + row * 128 + (lp_swz(row, 4 * hh + j) << 4));
}

__device__ __forceinline__ bool lp_poison16(u32x4 v) { return ((v.x | v.y | v.z | v.w) & 0x80008000u) != 0u; }
// signed fp16 data (the latent state): poison = NaN halves.  (h & 0x7fff) > 0x7c00 <=> bit 15 of (h & 0x7fff) + 0x03ff, per half, no carry
__device__ __forceinline__ bool lp_nan16(u32x4 v) {
    const unsigned m = 0x7fff7fffu, a = 0x03ff03ffu;
    return ((((v.x & m) + a) | ((v.y & m) + a) | ((v.z & m) + a) | ((v.w & m) + a)) & 0x80008000u) != 0u;
}
__device__ __forceinline__ bool lp_nan(unsigned b) { return (b & 0x7fffffffu) > 0x7f800000u; }
__device__ __forceinline__ bool lp_nan4(u32x4 v) { return (int)lp_nan(v.x) | (int)lp_nan(v.y) | (int)lp_nan(v.z) | (int)lp_nan(v.w); }

// between two polls: the back-off, and a compiler barrier so that the next pass's loads are issued again
__device__ __forceinline__ void lp_sleep(int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned lp_now() { return (unsigned)__builtin_amdgcn_s_memrealtime(); }

// Steps are periodic: a wait that took `prev` ticks in the previous step will take about as long in this one.  Waits of 8 us and
// more sleep through a fraction of it (shift: 1 = half, 2 = three quarters) before the second polling pass: hundreds of waves wait for most
// of a step, and each pass is a set of fabric reads that competes with the stores everybody is waiting for.  Short waits poll at once
// (they are the ones close to the critical path; the sleep loop's granularity is ~0.5 us).
__device__ __forceinline__ void lp_presleep(unsigned t_enter, unsigned prev, int shift) {
    if (prev < 800u) return;
    const unsigned until = t_enter + prev - (prev >> shift);
    while ((int)(until - lp_now()) > 0) __builtin_amdgcn_s_sleep(8);
    asm volatile("" ::: "memory");
}

// fragment-major byte offset of (row, col) in an fp16 activation buffer: [chunk = col / 64][j][hh * 32 + row][e]
__device__ __forceinline__ int lp_frag_off(int row, int col) {
    const int within = col & 63;
    return (col >> 6) * 4096 + ((within & 31) >> 3) * 1024 + (((within >> 5) << 5) + row) * 16 + (within & 7) * 2;
}

__device__ __forceinline__ unsigned lp_pack_relu_f16(float a, float b) {
    // post-ReLU values: saturate, round to fp16, sign bit cleared (the validity bit of the exchange)
    const half_t ha = to_half_sat(fmaxf(a, 0.f)), hb = to_half_sat(fmaxf(b, 0.f));
    return ((unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16)) & 0x7fff7fffu;
}

// Sum over the W consecutive lanes a lane belongs to (W = 2 .. 32, aligned groups), result in every lane: DPP quad permutes, half-row and
// row mirrors (VALU, a few cycles each) and one v_permlane16_swap for the last step, instead of a chain of ds_bpermute round trips through
// the LDS crossbar (~60 ns each, two dependent chains of up to five per GroupNorm: 0.3-0.5 us on every phase's critical path).  Both
// partners of a pair add the same two numbers, so all lanes of a group end with identical bits.
template <int CTRL>
__device__ __forceinline__ float lp_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int W>
__device__ __forceinline__ float lp_group_sum(float v) {
    if constexpr (W >= 2) v += lp_dpp<0xB1>(v);              // quad_perm [1,0,3,2]
    if constexpr (W >= 4) v += lp_dpp<0x4E>(v);              // quad_perm [2,3,0,1]
    if constexpr (W >= 8) v += lp_dpp<0x141>(v);             // row_half_mirror: lane i <-> 7 - i of each 8
    if constexpr (W >= 16) v += lp_dpp<0x140>(v);            // row_mirror: lane i <-> 15 - i of each 16
    if constexpr (W >= 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // rows (16 lanes) 0 <-> 1, 2 <-> 3
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}

struct LpCtx {
    __amdgpu_buffer_rsrc_t rs;         // the whole workspace
    unsigned* ctrl;                    // ctrl[0]: status (0 ok; else code << 16 | workgroup)
    char* smem;
    int* lds_flag;                     // [0]: abort seen by some wave of this workgroup
    int tid, lane, wave, r, hh;
    int sleep;
    unsigned* tr;                      // where this unit's stamps go (diagnostic build), else null
    unsigned* hist;                    // LDS word of (unit, wave): how long this wait took in the previous step (100 MHz ticks)
    int predict;                       // 1: sleep through most of the predicted wait before probing
};

__device__ __forceinline__ u32x4 lp_ld16(const LpCtx& c, int off) { return __builtin_amdgcn_raw_buffer_load_b128(c.rs, off, 0, 16); }
// stores: sc1 (write-through to the fabric: visible to every XCD) or plain (stays in this XCD's L2: visible to the sc1 loads of the same XCD)
__device__ __forceinline__ void lp_st16(const LpCtx& c, u32x4 v, int off, int local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, c.rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b128(v, c.rs, off, 0, 16);
}
__device__ __forceinline__ void lp_st8(const LpCtx& c, u32x2 v, int off, int local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b64(v, c.rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b64(v, c.rs, off, 0, 16);
}
__device__ __forceinline__ void lp_st4(const LpCtx& c, unsigned v, int off, int local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b32(v, c.rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b32(v, c.rs, off, 0, 16);
}

// called every 64 unsuccessful polls: true = give up (somebody aborted, or this wait exceeded the limit)
__device__ __forceinline__ bool lp_give_up(const LpCtx& c, unsigned t0, unsigned code) {
    const unsigned st = __hip_atomic_load(c.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st != 0u) return true;
    if (lp_now() - t0 > LP_TIMEOUT_TICKS) {
        if (c.lane == 0) {
            unsigned expected = 0u;
            __hip_atomic_compare_exchange_strong(c.ctrl, &expected, (code << 16) | (unsigned)blockIdx.x, __ATOMIC_RELAXED,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    return false;
}

// --------------------------------------------------------------------------------------------- gemm unit
// Operand fetch of one wave: its chunks (wave, wave + 4, ...) of the unit's K slice, 4 x 16 B per lane and chunk, polled
// until no fp16 carries the poison bit.  Returns false when the wait was abandoned.
template <int MAXCH>
__device__ __forceinline__ bool lp_fetch(const LpCtx& c, const LpLayer& L, const LpUnit& U, int set_off, u32x4 (&pc)[MAXCH][4]) {
    const bool sgn = U.head != 0;                        // enc1 reads the signed state: NaN poison instead of the sign bit
    int off[MAXCH];
    unsigned need = 0u;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cl = c.wave + 4 * i;                   // chunk of the slice
        const int kc = U.chunk0 + cl;                    // chunk of the concatenated K
        const int n1 = L.k1 >> 6;
        off[i] = set_off + (kc < n1 ? L.src1_off + kc * 4096 : L.src2_off + (kc - n1) * 4096) + c.lane * 16;
        if (cl < U.nchunks) need |= 1u << i;
    }
    const unsigned t0 = lp_now();
    for (unsigned spin = 1;; ++spin) {
        // all the missing chunks at once (when the producers are done this is the only pass) ...
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
            if (need & (1u << i)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pc[i][j] = lp_ld16(c, off[i] + j * 1024);
            }
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
            if (need & (1u << i)) {
                const bool bad = sgn ? (bool)((int)lp_nan16(pc[i][0]) | (int)lp_nan16(pc[i][1]) | (int)lp_nan16(pc[i][2]) | (int)lp_nan16(pc[i][3]))
                                     : (bool)((int)lp_poison16(pc[i][0]) | (int)lp_poison16(pc[i][1]) | (int)lp_poison16(pc[i][2]) | (int)lp_poison16(pc[i][3]));
                if (__builtin_amdgcn_ballot_w64(bad) == 0ull) need &= ~(1u << i);
            }
        if (need == 0u) {
            if (c.hist != nullptr && c.lane == 0) *c.hist = lp_now() - t0;
            return true;
        }
        if (spin == 1u && c.predict && c.hist != nullptr) lp_presleep(t0, __builtin_amdgcn_readfirstlane(*c.hist), c.predict);
        // ... otherwise again, after the back-off.  (Waiting on a single 1-KiB probe piece and fetching after it -- 1/16 of the polling
        // traffic -- was measured 1.4 us/step slower: the fetch after the probe is one more dependent round trip on every edge.)
        lp_sleep(c.sleep);
        if ((spin & 63u) == 0u && lp_give_up(c, t0, 1u)) return false;
    }
}

#pragma clang fp contract(off)
// DDIM update of one element, the operation order of pcd_ddim_update (diffusion.py:283-287)
__device__ __forceinline__ void lp_ddim(float z, float e, float n, float s, float n2, float s2, float& x0, float& zn) {
    const float ne = n * e;
    x0 = (z - ne) / s;
    const float a = s2 * x0;
    const float b = n2 * e;
    zn = a + b;
}
#pragma clang fp contract(fast)

struct LpStep {
    int k;                 // row of the step tables
    int set_off;           // byte offset of this step's buffer set in the workspace
    int other_off;         // ... of the other set (previous step's data; to be poisoned for the next step)
    int r0, nb;            // the rows of the batch this stream carries: global row r0 + local row, local rows < nb exist
};

// epilogue stores of a gemm unit (or its poison when `poison`): thread = (row = tid >> 3, q = tid & 7), 4 columns per tile
template <int CT>
__device__ __forceinline__ void lp_gemm_store(const LpCtx& c, const LpLayer& L, const LpUnit& U, int set_off, const float (&v)[CT][4],
                                              bool poison) {
    const int row = c.tid >> 3, q = c.tid & 7;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = U.col0 + 32 * t + 4 * q;
        if (L.slabs > 0) {                                  // fp32 partial tile -> slab [slice][row][col]
            const int off = set_off + L.slab_off + ((U.slice * 32 + row) * L.c + col) * 4;
            u32x4 o = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (!poison) o = (u32x4){__float_as_uint(v[t][0]), __float_as_uint(v[t][1]), __float_as_uint(v[t][2]), __float_as_uint(v[t][3])};
            lp_st16(c, o, off, L.slab_local);
        } else if (L.mode == 2) {                           // the next step's state, signed fp16 (saturating, like pcd_skinny_fused_f32in), fragment-major
            const int off = set_off + L.out_off + lp_frag_off(row, col);
            u32x2 o = {0xffffffffu, 0xffffffffu};           // fp16 NaN halves
            if (!poison) {
                const half_t h0 = to_half_sat(v[t][0]), h1 = to_half_sat(v[t][1]), h2 = to_half_sat(v[t][2]), h3 = to_half_sat(v[t][3]);
                o = (u32x2){(unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16),
                            (unsigned)__builtin_bit_cast(unsigned short, h2) | ((unsigned)__builtin_bit_cast(unsigned short, h3) << 16)};
            }
            lp_st8(c, o, off, L.out_local);
        } else {                                            // fp16 activation, fragment-major, 4 columns = 8 bytes
            const int off = set_off + L.out_off + lp_frag_off(row, col);
            u32x2 o = {0xffffffffu, 0xffffffffu};
            if (!poison) o = (u32x2){lp_pack_relu_f16(v[t][0], v[t][1]), lp_pack_relu_f16(v[t][2], v[t][3])};
            lp_st8(c, o, off, L.out_local);
        }
    }
}

__device__ __forceinline__ void lp_poison_all(const LpCtx& c, const LpArgs& A, const LpUnit* units, int other_off);

template <int CT, int MAXCH>
__device__ __forceinline__ bool lp_run_gemm(const LpCtx& c, const LpLayer& L, const LpUnit& U, const LpStep& S, const LpArgs& A,
                                            bool& poisoned, const LpUnit* units, const float (&tbv)[4], float (&zs)[4], const float (&rates)[4],
                                            bool last_step) {
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bool ok = true;
    const char* wimg = c.smem + U.lds_w;
    {
        u32x4 pc[MAXCH][4];
        // enc1's operand is the state the previous step's output.2 units published (in that step's set)
        ok = lp_fetch<MAXCH>(c, L, U, U.head ? S.other_off : S.set_off, pc);
        if (ok) {
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int cl = c.wave + 4 * i;
                if (cl < U.nchunks) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const half8 a = __builtin_bit_cast(half8, pc[i][j]);
#pragma unroll
                        for (int t = 0; t < CT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, lp_wfrag(wimg + (t * U.nchunks + cl) * 4096, c.r, c.hh, j), acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!ok) c.lds_flag[0] = 1;
    if (c.tr != nullptr && c.lane == 0) { c.tr[4 + c.wave] = lp_now(); if (c.wave == 0) c.tr[1] = lp_now(); }
    // cross-wave reduction through LDS, one 32-column tile at a time
    float (*red)[32][33] = (float (*)[32][33])c.smem;
    const int row = c.tid >> 3, q = c.tid & 7;
    float v[CT][4];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[c.wave][(e & 3) + 8 * (e >> 2) + 4 * c.hh][c.r] = acc[t][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[t][i] = (red[0][row][4 * q + i] + red[1][row][4 * q + i]) + (red[2][row][4 * q + i] + red[3][row][4 * q + i]);
    }
    if (c.lds_flag[0]) return false;                        // uniform: written before the barriers above
    if (L.slabs > 0 && U.slice == 0) {
        // partial tile of the first K slice: + bias here, so that the finish unit keeps only gamma | beta (its sum (s0 + b) + s1 + ...
        // has the order of csrc/skinny.hip's finish)
        const float* pst = (const float*)(c.smem + U.lds_p);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[t][i] += pst[32 * t + 4 * q + i];
    }
    if (L.slabs == 0) {
        // the unit holds whole GroupNorm groups over the full K: bias, GroupNorm, ReLU here; the per-column constants were
        // copied to LDS when the kernel started (a global load here is a memory round trip on every step's critical path)
        const float* pst = (const float*)(c.smem + U.lds_p);
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float b = U.head ? tbv[i] : pst[32 * t + 4 * q + i];
                v[t][i] += b;
                s1 += v[t][i];
            }
        if (L.mode == 0) {
            const bool g8 = L.gsz >= 32;                       // threads of this row that share a group: 8, or 4 (groups of 16)
            s1 = g8 ? lp_group_sum<8>(s1) : lp_group_sum<4>(s1);
            const float mean = s1 / (float)L.gsz;
            float s2 = 0.f;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[t][i] - mean; s2 += d * d; }
            s2 = g8 ? lp_group_sum<8>(s2) : lp_group_sum<4>(s2);
            const float rstd = rsqrtf(s2 / (float)L.gsz + 1e-5f);
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[t][i] = (v[t][i] - mean) * rstd * pst[64 + 32 * t + 4 * q + i] + pst[128 + 32 * t + 4 * q + i];
                }
        }
    }
    if (L.slabs == 0 && L.mode == 2) {
        // output.2: v = eps of this thread's 4 columns.  Forward mode hands eps back; the sampler applies the DDIM update to the
        // thread's piece of the state and publishes the new state as the next step's enc1 operand
        if (A.forward_only) {
            if (row < S.nb) *(float4*)(A.eps_out + (int64_t)(S.r0 + row) * L.c + U.col0 + 4 * q) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            return true;
        }
        float x0v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float zn;
            lp_ddim(zs[i], v[0][i], rates[0], rates[1], rates[2], rates[3], x0v[i], zn);
            zs[i] = row < S.nb ? zn : 0.f;
            v[0][i] = zs[i];
        }
        if (last_step && row < S.nb) {
            *(float4*)(A.z + (int64_t)(S.r0 + row) * 256 + U.col0 + 4 * q) = make_float4(zs[0], zs[1], zs[2], zs[3]);
            if (A.x0 != nullptr) *(float4*)(A.x0 + (int64_t)(S.r0 + row) * 256 + U.col0 + 4 * q) = make_float4(x0v[0], x0v[1], x0v[2], x0v[3]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's poison stores are complete before its data stores leave
    if (c.tr != nullptr && c.tid == 0) c.tr[3] = lp_now();
    lp_gemm_store<CT>(c, L, U, S.set_off, v, false);
    return true;
}

// --------------------------------------------------------------------------------------------- finish unit
// One GroupNorm group x (256 / TPR) rows: slabs added in slice order + bias, two-pass statistics over the TPR threads of
// a row, affine, ReLU, fp16 fragment-major store.  CPT columns per thread.
// Output of a finish unit: ROWS = 256 / TPR rows x gsz columns of fp16, fragment-major.  The rows of one (chunk, j, hh) piece are 16 bytes
// apart, so the values go through LDS ([row][gsz] fp16 in the reduction scratch) and leave with consecutive lanes on consecutive rows: 8
// lanes = one full 128-byte line per store instruction (a thread's own CPT columns of 2 rows per wave would be 32-byte fragments of 32
// lines: 4x the fabric writes, and the write-through acknowledgement of the big layers' finish phases took 3-5 us).  Poison: same addresses.
// column (inside the group) of element i of thread q of a row: 4-column pieces dealt round-robin to the TPR threads (piece p of thread q =
// columns 4 (p TPR + q) ..): a wave-wide 16-byte access then covers consecutive addresses, in the slabs and in LDS (16 columns per thread in
// one run made every LDS read of the per-column constants a 16-way bank conflict and every slab load touch 32 lines for a quarter each)
template <int CPT, int TPR>
__device__ __forceinline__ int lp_fin_col(int q, int i) { return CPT >= 4 ? (((i >> 2) * TPR + q) << 2) + (i & 3) : q * CPT + i; }

template <int CPT, int TPR>
__device__ __forceinline__ void lp_finish_store(const LpCtx& c, const LpLayer& L, const LpUnit& U, int set_off, const float (&x)[CPT], bool poison) {
    constexpr int ROWS = LP_THREADS / TPR;
    const int gsz = TPR * CPT;
    unsigned* stage = (unsigned*)c.smem;                       // [ROWS][gsz / 2] packed pairs
    if (!poison) {
        const int rr = c.tid / TPR, q = c.tid % TPR;
        __syncthreads();                                       // the scratch may still be read by a previous unit's epilogue
        if constexpr (CPT >= 4) {
#pragma unroll
            for (int p = 0; p < CPT / 4; ++p)
                *(u32x2*)(stage + rr * (gsz / 2) + (lp_fin_col<CPT, TPR>(q, 4 * p) >> 1)) =
                    (u32x2){lp_pack_relu_f16(x[4 * p], x[4 * p + 1]), lp_pack_relu_f16(x[4 * p + 2], x[4 * p + 3])};
        } else {
#pragma unroll
            for (int i = 0; i < CPT / 2; ++i) stage[rr * (gsz / 2) + q * (CPT / 2) + i] = lp_pack_relu_f16(x[2 * i], x[2 * i + 1]);
        }
        __syncthreads();
    }
    const int npieces = ROWS * (gsz / 8);                      // 16-byte pieces: 8 columns of one row
    for (int id = c.tid; id < npieces; id += LP_THREADS) {
        const int rr = id % ROWS, cg = id / ROWS;
        u32x4 v = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (!poison) v = *(const u32x4*)(stage + rr * (gsz / 2) + cg * 4);
        lp_st16(c, v, set_off + L.out_off + lp_frag_off(U.row0 + rr, U.group * gsz + cg * 8), L.out_local);
    }
}

template <int CPT, int TPR, int SMAX>
__device__ __forceinline__ bool lp_run_finish(const LpCtx& c, const LpLayer& L, const LpUnit& U, const LpStep& S, const LpArgs& A, bool& poisoned,
                                              const LpUnit* units) {
    const int row = U.row0 + c.tid / TPR, q = c.tid % TPR;
    constexpr int V = CPT >= 4 ? CPT / 4 : 1;               // 16-byte loads per slab (CPT = 2: one 8-byte load)
    u32x4 raw[SMAX][V];
    const int base = S.set_off + L.slab_off + (row * L.c + U.group * L.gsz + lp_fin_col<CPT, TPR>(q, 0)) * 4;
    constexpr int PSTEP = TPR * 16;                         // bytes between a thread's consecutive 4-column pieces
    const int sstride = 32 * L.c * 4;
    const unsigned t0 = lp_now();
    bool ok = true;
    for (unsigned spin = 1;; ++spin) {
        bool bad = false;
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < L.slabs) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    if constexpr (CPT >= 4) raw[s][i] = lp_ld16(c, base + s * sstride + i * PSTEP);
                    else {
                        const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(c.rs, base + s * sstride, 0, 16);
                        raw[s][i] = (u32x4){t2.x, t2.y, 0u, 0u};
                    }
                }
            }
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < L.slabs) {
#pragma unroll
                for (int i = 0; i < V; ++i) bad |= lp_nan4(raw[s][i]);
            }
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
            if (c.hist != nullptr && c.lane == 0) *c.hist = lp_now() - t0;
            break;
        }
        if ((spin & 63u) == 0u && lp_give_up(c, t0, 2u)) { ok = false; break; }
        if (spin == 1u && c.predict && c.hist != nullptr) lp_presleep(t0, __builtin_amdgcn_readfirstlane(*c.hist), c.predict);
        lp_sleep(c.sleep);
    }
    if (!ok) c.lds_flag[0] = 1;
    if (c.tr != nullptr && c.lane == 0) { c.tr[4 + c.wave] = lp_now(); if (c.wave == 0) c.tr[1] = lp_now(); }
    __syncthreads();
    if (c.lds_flag[0]) return false;
    // slab 0 already carries the bias (added by the slice-0 gemm unit): x = ((s0 + b) + s1) + ..., the order of csrc/skinny.hip's finish
    const float* pst = (const float*)(c.smem + U.lds_p);    // gamma | beta of this group's columns
    float x[CPT];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        float v = __uint_as_float(raw[0][CPT >= 4 ? i / 4 : 0][i & 3]);
#pragma unroll
        for (int s = 1; s < SMAX; ++s)
            if (s < L.slabs) v += __uint_as_float(raw[s][CPT >= 4 ? i / 4 : 0][i & 3]);
        x[i] = v;
        s1 += v;
    }
    s1 = lp_group_sum<TPR>(s1);
    const float mean = s1 / (float)L.gsz;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) { const float d = x[i] - mean; s2 += d * d; }
    s2 = lp_group_sum<TPR>(s2);
    const float rstd = rsqrtf(s2 / (float)L.gsz + 1e-5f);
    if constexpr (CPT >= 4) {
#pragma unroll
        for (int p = 0; p < CPT / 4; ++p) {
            const float4 ga = *(const float4*)(pst + lp_fin_col<CPT, TPR>(q, 4 * p));
            const float4 be = *(const float4*)(pst + L.gsz + lp_fin_col<CPT, TPR>(q, 4 * p));
            x[4 * p] = (x[4 * p] - mean) * rstd * ga.x + be.x;
            x[4 * p + 1] = (x[4 * p + 1] - mean) * rstd * ga.y + be.y;
            x[4 * p + 2] = (x[4 * p + 2] - mean) * rstd * ga.z + be.z;
            x[4 * p + 3] = (x[4 * p + 3] - mean) * rstd * ga.w + be.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPT; ++i) x[i] = (x[i] - mean) * rstd * pst[q * CPT + i] + pst[L.gsz + q * CPT + i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c.tr != nullptr && c.tid == 0) c.tr[3] = lp_now();
    lp_finish_store<CPT, TPR>(c, L, U, S.set_off, x, false);
    return true;
}

__device__ __forceinline__ bool lp_dispatch_finish(const LpCtx& c, const LpLayer& L, const LpUnit& U, const LpStep& S, const LpArgs& A, bool& poisoned,
                                                   const LpUnit* units) {
    switch (L.gsz) {
        case 512: return lp_run_finish<16, 32, 2>(c, L, U, S, A, poisoned, units);
        case 256: return lp_run_finish<8, 32, 2>(c, L, U, S, A, poisoned, units);
        case 128: return lp_run_finish<4, 32, 8>(c, L, U, S, A, poisoned, units);
        case 64: return lp_run_finish<2, 32, 4>(c, L, U, S, A, poisoned, units);
        default: return lp_run_finish<2, 16, 2>(c, L, U, S, A, poisoned, units);       // gsz 32
    }
}

// poison every output region of this workgroup in the buffer set at `other_off` (same addresses and store widths as the data stores)
__device__ __forceinline__ void lp_poison_all(const LpCtx& c, const LpArgs& A, const LpUnit* units, int other_off) {
    const float zero[2][4] = {};
    const float zero16[16] = {};
    for (int p = 0; p < LP_MAX_UNITS; ++p) {
        const LpUnit P = units[p];
        if (P.kind == LP_END) break;
        const LpLayer& L = A.layers[P.layer];
        if (P.kind == LP_GEMM) {
            if (P.ct == 2) lp_gemm_store<2>(c, L, P, other_off, zero, true);
            else lp_gemm_store<1>(c, L, P, other_off, *(const float (*)[1][4])zero, true);
        } else {
            switch (L.gsz) {
                case 512: lp_finish_store<16, 32>(c, L, P, other_off, zero16, true); break;
                case 256: lp_finish_store<8, 32>(c, L, P, other_off, *(const float (*)[8])zero16, true); break;
                case 128: lp_finish_store<4, 32>(c, L, P, other_off, *(const float (*)[4])zero16, true); break;
                case 64: lp_finish_store<2, 32>(c, L, P, other_off, *(const float (*)[2])zero16, true); break;
                default: lp_finish_store<2, 16>(c, L, P, other_off, *(const float (*)[2])zero16, true); break;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------- the kernel
// NS = 1: a batch of up to 32 rows as one stream of steps.  NS = 2 (batches of 33 .. 64): rows 0 .. 31 and the rest as TWO independent
// streams (GroupNorm is per row: they never meet), each with its own three buffer sets, that every workgroup interleaves in a FIXED order:
// a step is 18 dependent exchanges of ~2 us during which a workgroup works ~20 % of the time, so stream 1 runs about half a step behind
// stream 0 and its units fill stream 0's waits.  The order is the same total order on every workgroup -- key = the time a phase becomes
// runnable in a step + the lag for stream 1 (host: lp_upload_sched); a unit depends only on units of its own stream with smaller keys --
// so blocking waits in that order cannot deadlock (the unfinished unit with the smallest key is always runnable), and every wait stays the
// per-wave spin of the single-stream loop.  51-53 us per step of the pair (26 per 32 rows) against 2 x 35 for one stream after the other.
// (Trying both streams' next units in turn, or one probe load per stream in flight, were both slower than running the streams one after
// the other: a poll that involves the whole workgroup delays the other stream's detection by its round trip: 71 and 103 us per step.)
template <bool TRACE, int NS>
__global__ __launch_bounds__(LP_THREADS, 1) void latent_persist_kernel(LpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lp_smem[];
    __shared__ int lp_flag[4];
    __shared__ unsigned lp_hist[(LP_MAX_UNITS + 1) * 4];
    LpCtx c;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(A.ws, 0, LP_CTRL_BYTES + NS * 3 * A.set_bytes, 0x00020000);
    c.ctrl = (unsigned*)A.ws;
    c.smem = lp_smem;
    c.lds_flag = lp_flag;
    c.tid = threadIdx.x;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.r = c.lane & 31;
    c.hh = c.lane >> 5;
    c.sleep = A.sleep;
    c.tr = nullptr;
    c.hist = nullptr;
    c.predict = A.predict;
    if (c.tid < (LP_MAX_UNITS + 1) * 4) lp_hist[c.tid] = 0u;
    // Work is assigned by (XCD, arrival rank on that XCD), not by blockIdx: the 7 small layers that chain the end of one step to the
    // start of the next all run on XCD 0 and hand their outputs over through that XCD's L2.  256 workgroups on 256 CUs = 32 per XCD
    // whatever the dispatch order (one per CU: each needs the whole LDS); a rank of 32 or more reports and aborts.
    if (c.tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned rank = __hip_atomic_fetch_add(c.ctrl + 4 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lp_flag[1] = rank < 32u ? (int)(xcc * 32u + rank) : -1;
        lp_flag[0] = 0;
        if (rank >= 32u) {
            unsigned expected = 0u;
            __hip_atomic_compare_exchange_strong(c.ctrl, &expected, (4u << 16) | (unsigned)blockIdx.x, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int vwg = lp_flag[1];
    if (vwg < 0) return;
    const LpUnit* units = A.units + (int64_t)vwg * LP_MAX_UNITS;

    // ---- stage this workgroup's weights once: every 4-KiB image = 4 LDS-DMA instructions, dealt round-robin to the waves
    bool is_head = false;                                   // an enc1 unit: its bias is the step's time row
    int zu = -1;                                            // this workgroup's output.2 unit (it keeps 32 columns of the state), or -1
    {
        int task = 0;
        for (int u = 0; u < LP_MAX_UNITS; ++u) {
            const LpUnit U = units[u];
            if (U.kind == LP_END) break;
            if (U.kind != LP_GEMM) continue;
            is_head |= U.head != 0;
            const LpLayer& L = A.layers[U.layer];
            if (L.mode == 2) zu = u;
            for (int t = 0; t < U.ct; ++t)
                for (int cl = 0; cl < U.nchunks; ++cl)
                    for (int i = 0; i < 4; ++i, ++task)
                        if ((task & 3) == c.wave)
                            lp_stage(L.w, L.ldw, U.col0 + 32 * t, (U.chunk0 + cl) * 64, lp_smem + U.lds_w + (t * U.nchunks + cl) * 4096, i, c.lane);
        }
    }
    // per-column constants of this workgroup's units -> LDS: gemm units that finish their layer [bias | gamma | beta] x 64 columns,
    // slice-0 gemm units of a layer with partial tiles [bias] x their columns, finish units [gamma | beta] x gsz columns
    for (int u = 0; u < LP_MAX_UNITS; ++u) {
        const LpUnit U = units[u];
        if (U.kind == LP_END) break;
        if (U.lds_p < 0) continue;
        const LpLayer& L = A.layers[U.layer];
        float* pst = (float*)(lp_smem + U.lds_p);
        if (U.kind == LP_GEMM) {
            for (int i = c.tid; i < 32 * U.ct; i += LP_THREADS) {
                pst[i] = L.bias[U.col0 + i];
                if (L.slabs == 0 && L.mode == 0) { pst[64 + i] = L.gamma[U.col0 + i]; pst[128 + i] = L.beta[U.col0 + i]; }
            }
        } else {
            for (int i = c.tid; i < L.gsz; i += LP_THREADS) {
                pst[i] = L.gamma[U.group * L.gsz + i];
                pst[L.gsz + i] = L.beta[U.group * L.gsz + i];
            }
        }
    }
    const int k_base = A.counter != nullptr ? A.counter[0] : 0;
    // rows of the batch per stream: stream s carries global rows sr0[s] .. sr0[s] + snb[s] - 1 as its local rows 0 ..
    int sr0[NS], snb[NS];
    sr0[0] = 0; snb[0] = NS == 1 ? A.batch : 32;
    if constexpr (NS == 2) { sr0[1] = 32; snb[1] = A.batch - 32; }
    // the latent state: the thread (row = tid >> 3, q = tid & 7) of an output.2 unit keeps columns col0 + 4 q .. + 3 of its row (rows past
    // the stream's are zero) and publishes the initial state the way every later step's epilogue does: fp16, fragment order, into the set
    // "before" step 0 (the stream's set 2; the host poisoned all the sets)
    float zs[NS][4];
    const int zrow = c.tid >> 3;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        zs[st][0] = zs[st][1] = zs[st][2] = zs[st][3] = 0.f;
        if (zu >= 0) {
            const LpUnit U = units[zu];
            const LpLayer& L = A.layers[U.layer];
            if (zrow < snb[st]) {
                const float4 v = *(const float4*)(A.z + (int64_t)(sr0[st] + zrow) * 256 + U.col0 + 4 * (c.tid & 7));
                zs[st][0] = v.x; zs[st][1] = v.y; zs[st][2] = v.z; zs[st][3] = v.w;
            }
            const float v1[1][4] = {{zs[st][0], zs[st][1], zs[st][2], zs[st][3]}};
            lp_gemm_store<1>(c, L, U, LP_CTRL_BYTES + (st * 3 + 2) * A.set_bytes, v1, false);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int total = A.forward_only ? 1 : A.nsteps;
    // one step's constants of a stream: its buffer sets (three per stream, rotating by step: the previous step's state -- enc1's operand --
    // is in set (step + 2) % 3, which is also the set re-poisoned at the end of the step, for step + 2), the row of the step tables
    auto step_of = [&](int st, int step) {
        LpStep S;
        const int k = k_base + step;
        S.k = k < A.T ? k : A.T - 1;
        S.set_off = LP_CTRL_BYTES + (st * 3 + step % 3) * A.set_bytes;
        S.other_off = LP_CTRL_BYTES + (st * 3 + (step + 2) % 3) * A.set_bytes;
        S.r0 = sr0[st]; S.nb = snb[st];
        return S;
    };
    // per-step constants, requested at the start of a step and used behind the waits: enc1's bias is the hoisted time row of the step;
    // the update's four rates of this thread's row
    auto step_consts = [&](const LpStep& S, float (&tbv)[4], float (&rates)[4]) __attribute__((always_inline)) {
        tbv[0] = tbv[1] = tbv[2] = tbv[3] = 0.f;
        if (is_head) {
            const float4 t4 = *(const float4*)(A.tb_table + (int64_t)S.k * A.tb_elems + units[0].col0 + 4 * (c.tid & 7));
            tbv[0] = t4.x; tbv[1] = t4.y; tbv[2] = t4.z; tbv[3] = t4.w;
        }
        rates[0] = 0.f; rates[1] = 1.f; rates[2] = 0.f; rates[3] = 0.f;
        if (zu >= 0 && !A.forward_only && zrow < S.nb) {
            const int rb = (S.r0 + zrow) * A.rate_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) rates[i] = A.rates[((int64_t)i * A.T + S.k) * A.rate_width + rb];
        }
    };
    bool poisoned = false;
    // one unit of (stream, step), blocking
    auto run_unit = [&](const LpStep& S, int u, int step, bool traced, const float (&tbv)[4], float (&zst)[4], const float (&rates)[4]) __attribute__((always_inline)) {
        const LpUnit U = units[u];
        const LpLayer& L = A.layers[U.layer];
        if (TRACE && traced && step < A.trace_steps) {
            c.tr = A.trace + (((int64_t)vwg * A.trace_steps + step) * LP_MAX_UNITS + u) * 8;
            if (c.tid == 0) c.tr[0] = lp_now();
        } else c.tr = nullptr;
        bool ok;
        if (U.kind == LP_GEMM) {
            if (U.ct == 2) ok = lp_run_gemm<2, 1>(c, L, U, S, A, poisoned, units, tbv, zst, rates, step == total - 1);
            else ok = lp_run_gemm<1, 4>(c, L, U, S, A, poisoned, units, tbv, zst, rates, step == total - 1);
        } else {
            ok = lp_dispatch_finish(c, L, U, S, A, poisoned, units);
        }
        if (ok && TRACE && c.tr != nullptr && c.tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c.tr[2] = lp_now(); }
        return ok;
    };
    if constexpr (NS == 1) {
        for (int step = 0; step < total; ++step) {
            if (step == A.fault_step && vwg == A.fault_wg) return;      // injected fault: this workgroup's publishes of the step never happen
            const LpStep S = step_of(0, step);
            float tbv[4], rates[4];
            step_consts(S, tbv, rates);
            for (int u = 0; u < LP_MAX_UNITS; ++u) {
                if (units[u].kind == LP_END) break;
                c.hist = lp_hist + u * 4 + c.wave;
                if (!run_unit(S, u, step, true, tbv, zs[0], rates)) return;
            }
            // End of this workgroup's step s.  Its units consumed step-s data, so every reader of the previous step's set is done (step
            // s's first layer needed the complete state published by step s - 1): re-poison this workgroup's regions of that set, which step s + 2
            // will use.  Here, behind the last unit, the stores are off every critical path (issued right behind a unit they delay the
            // next unit's operand loads: vmcnt retires in order).  Why a reader of step s + 2 cannot see step s - 1's values: before it
            // polls, it has finished its own dec4 unit of step s + 1, which needed every workgroup's gf3 output of step s + 1, and every
            // data store waits (s_waitcnt vmcnt(0)) for all earlier stores of its wave, these included.
            lp_poison_all(c, A, units, S.other_off);
        }
    } else {
        // two streams in the fixed merged order of this workgroup (A.sched: per period, entries sorted by key; an entry = unit index |
        // stream << 8 | (its step is the period's - 1) << 9 | (last unit of the stream's step on this workgroup) << 10; -1 ends the list).
        // Period t: stream 0 runs its step t, stream 1 the late part of its step t - 1 and then the early part of its step t.
        const int* sched = A.sched + (int64_t)vwg * LP_SCHED;
        float tbv_[NS][4], rates_[NS][4];
        c.hist = nullptr;                                    // (no wait prediction: a wait's length depends on the other stream too)
        for (int t = 0; t <= total; ++t) {
            if (t == A.fault_step && vwg == A.fault_wg) return;
            for (int e = 0; e < LP_SCHED; ++e) {
                const int ent = sched[e];
                if (ent < 0) break;
                const int u = ent & 255, st = (ent >> 8) & 1, step = t - ((ent >> 9) & 1);
                if (step < 0 || step >= total) continue;
                const LpStep S = step_of(st, step);
                bool ok;
                if (st == 0) {
                    if (u == 0) step_consts(S, tbv_[0], rates_[0]);
                    ok = run_unit(S, u, step, true, tbv_[0], zs[0], rates_[0]);
                } else {
                    if (u == 0) step_consts(S, tbv_[NS - 1], rates_[NS - 1]);
                    ok = run_unit(S, u, step, false, tbv_[NS - 1], zs[NS - 1], rates_[NS - 1]);
                }
                if (!ok) return;
                // end of a stream's step on this workgroup: the ordering argument of the single-stream loop holds inside a stream
                if ((ent >> 10) & 1) lp_poison_all(c, A, units, S.other_off);
            }
        }
    }
    if (vwg == 0 && c.tid == 0 && A.counter != nullptr && !A.forward_only) {
        const int last = k_base + A.nsteps - 1;
        A.counter[1] = last < A.T ? last : A.T - 1;
        A.counter[0] = k_base + A.nsteps;
    }
}

}  // namespace pcd

using namespace pcd;

// ------------------------------------------------------------------------------------------------ host side
namespace {

const int kK1[LP_LAYERS] = {256, 128, 256, 512, 1024, 2048, 4096, 1024, 512, 256, 128, 128};
const int kK2[LP_LAYERS] = {0, 0, 0, 0, 0, 0, 1024, 512, 256, 128, 0, 0};
const int kC[LP_LAYERS] = {128, 256, 512, 1024, 2048, 4096, 1024, 512, 256, 128, 128, 256};
// K-slices of fp32 partial tiles per layer (0: the unit finishes the layer itself)
const int kSlabs[LP_LAYERS] = {0, 0, 0, 1, 2, 2, 8, 4, 2, 0, 0, 0};
// input activations: index of the producing layer (-1: the latent state), second source = the skip
const int kSrc1[LP_LAYERS] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
const int kSrc2[LP_LAYERS] = {-1, -1, -1, -1, -1, -1, 3, 2, 1, 0, -1, -1};

struct LpPlan {
    LpLayer layers[LP_LAYERS];
    std::vector<LpUnit> units;         // [LP_WGS * LP_MAX_UNITS]
    int set_bytes;
};

int lp_align(int v, int a = 256) { return (v + a - 1) / a * a; }

// the static work assignment described at the top of the file; returns false if it does not fit (never, for this network)
bool lp_build_plan(const pcd_latent_desc_t& d, LpPlan& plan) {
    int off = 0;
    int out_off[LP_LAYERS], slab_off[LP_LAYERS];
    for (int l = 0; l < LP_LAYERS; ++l) {
        out_off[l] = off;
        off += lp_align(32 * kC[l] * 2);                  // fp16, fragment-major (the last layer's output is the next state)
    }
    for (int l = 0; l < LP_LAYERS; ++l) {
        slab_off[l] = off;
        off += lp_align(kSlabs[l] * 32 * kC[l] * 4);
    }
    plan.set_bytes = lp_align(off, 4096);
    for (int l = 0; l < LP_LAYERS; ++l) {
        LpLayer& L = plan.layers[l];
        L.w = (const half_t*)d.lin[l].w; L.ldw = d.lin[l].k;
        L.bias = d.lin[l].b; L.gamma = d.gn_gamma[l]; L.beta = d.gn_beta[l];
        L.k1 = kK1[l]; L.k2 = kK2[l];
        L.src1_off = out_off[kSrc1[l] < 0 ? LP_LAYERS - 1 : kSrc1[l]];      // enc1 reads the state output.2 published (previous step's set)
        L.src2_off = kSrc2[l] < 0 ? -1 : out_off[kSrc2[l]];
        L.c = kC[l]; L.gsz = kC[l] / 8; L.mode = l < 10 ? 0 : (l == 10 ? 1 : 2);
        L.slabs = kSlabs[l]; L.out_off = out_off[l]; L.slab_off = slab_off[l];
        L.out_local = (l == 0 || l == 1 || l == 8 || l == 9 || l == 10 || l == 11) ? 1 : 0;      // producers and consumers all on XCD 0
        L.slab_local = kSlabs[l] > 0 ? 1 : 0;          // finish units are placed on the XCD that produces their group (checked below)
    }
    struct Item { int wg, phase; LpUnit u; int wbytes; };
    std::vector<Item> items;
    auto gemm = [&](int wg, int layer, int col0, int ct, int chunk0, int nch, int slice, int head) {
        LpUnit u{};
        u.kind = LP_GEMM; u.layer = layer; u.col0 = col0; u.ct = ct; u.chunk0 = chunk0; u.nchunks = nch; u.slice = slice; u.head = head;
        items.push_back({wg, 2 * layer, u, ct * nch * 4096});
    };
    auto finish = [&](int wg, int layer, int group, int row0) {
        LpUnit u{};
        u.kind = LP_FINISH; u.layer = layer; u.group = group; u.row0 = row0;
        items.push_back({wg, 2 * layer + 1, u, 0});
    };
    // w = virtual workgroup id = XCD * 32 + rank.  XCD 0 (w < 32) holds the small layers whose outputs stay in its L2.
    for (int w = 0; w < LP_WGS; ++w) {
        gemm(w, 5, 32 * (w >> 1), 1, 16 * (w & 1), 16, w & 1, 0);              // global_feat.3: 128 tiles x 2 slices, 64 KB
        gemm(w, 6, 32 * (w >> 3), 1, 10 * (w & 7), 10, w & 7, 0);              // dec4: 32 tiles x 8 slices, 40 KB
        if (w < 8) gemm(w, 2, 64 * w, 2, 0, 4, 0, 0);                          // enc3: 8 units of 2 tiles (one group), 32 KB
        else if (w < 12) gemm(w, 0, 32 * (w - 8), 1, 0, 4, 0, 1);              // enc1: 4 tiles, 16 KB + the state
        else if (w < 28) {
            const int u = w - 12;
            gemm(w, 8, 32 * (u >> 1), 1, 6 * (u & 1), 6, u & 1, 0);            // dec2: 8 tiles x 2 slices, 24 KB
            if (u < 8) gemm(w, 1, 32 * u, 1, 0, 2, 0, 0);                      // enc2: 8 tiles, 8 KB
            else if (u < 12) gemm(w, 10, 32 * (u - 8), 1, 0, 2, 0, 0);         // output.0: 4 tiles, 8 KB
            else gemm(w, 11, 32 * (u - 12), 1, 0, 2, 0, 0);                    // output.2: tiles 0..3, 8 KB
        } else if (w < 32) {
            gemm(w, 9, 32 * (w - 28), 1, 0, 6, 0, 0);                          // dec1: 4 tiles, 24 KB
            gemm(w, 11, 32 * (w - 28 + 4), 1, 0, 2, 0, 0);                     // output.2: tiles 4..7, 8 KB
        } else if (w < 160) { const int u = w - 32; gemm(w, 4, 32 * (u >> 1), 1, 8 * (u & 1), 8, u & 1, 0); }   // global_feat.0: 64 tiles x 2, 32 KB
        else if (w < 192) gemm(w, 3, 32 * (w - 160), 1, 0, 8, 0, 0);           // enc4: 32 tiles, 32 KB
        else { const int u = w - 192; gemm(w, 7, 32 * (u >> 2), 1, 6 * (u & 3), 6, u & 3, 0); }                  // dec3: 16 tiles x 4, 24 KB
    }
    // Finish units go to the XCD whose workgroups produce their group's partial tiles (all K slices of all its column tiles: true for every
    // layer of this assignment, verified here), so that hand-over too stays in one L2; inside the XCD, to the workgroup with the most LDS left
    // for the unit's constants.
    {
        std::vector<int> used(LP_WGS, LP_RED_BYTES);
        for (const Item& it : items)
            used[it.wg] += it.wbytes +
                           (kSlabs[it.u.layer] == 0 ? 768 : (it.u.slice == 0 ? 128 * it.u.ct : 0));
        const int fin_layers[6] = {5, 4, 3, 6, 7, 8};         // largest constants first
        for (int f = 0; f < 6; ++f) {
            const int l = fin_layers[f], gsz = kC[l] / 8;
            const int rows = gsz >= 64 ? 8 : 16, per_group = 32 / rows;
            std::vector<int> mine(LP_WGS, 0);                 // finish units of THIS layer already on a workgroup: they would run one after
                                                              // the other, and the second one's consumers wait a whole unit longer
            for (int g = 0; g < 8; ++g) {
                int xcd = -1;
                for (const Item& it : items)
                    if (it.u.kind == LP_GEMM && it.u.layer == l && it.u.col0 / gsz == g) {
                        if (it.u.col0 / gsz != (it.u.col0 + 32 * it.u.ct - 1) / gsz) return false;
                        if (xcd >= 0 && xcd != it.wg / 32) return false;
                        xcd = it.wg / 32;
                    }
                if (xcd < 0) return false;
                for (int rb = 0; rb < per_group; ++rb) {
                    int best = -1;
                    for (int r = 0; r < 32; ++r) {
                        const int w = xcd * 32 + r;
                        if (used[w] + 2 * gsz * 4 > LP_DYN_LDS) continue;
                        if (best < 0 || mine[w] < mine[best] || (mine[w] == mine[best] && used[w] < used[best])) best = w;
                    }
                    if (best < 0) return false;
                    used[best] += 2 * gsz * 4;
                    mine[best]++;
                    finish(best, l, g, rb * rows);
                }
            }
        }
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.wg != b.wg ? a.wg < b.wg : a.phase < b.phase; });
    plan.units.assign((size_t)LP_WGS * LP_MAX_UNITS, LpUnit{});
    std::vector<int> count(LP_WGS, 0), lds(LP_WGS, LP_RED_BYTES);
    for (const Item& it : items) {
        if (count[it.wg] >= LP_MAX_UNITS - 1) return false;                  // the last slot stays LP_END
        LpUnit u = it.u;
        if (u.kind == LP_GEMM) {
            u.lds_w = lds[it.wg];
            lds[it.wg] += it.wbytes;
            const LpLayer& L = plan.layers[u.layer];
            if (L.slabs == 0 && L.mode == 0 && !((u.ct == 1 && (L.gsz == 16 || L.gsz == 32)) || L.gsz == 32 * u.ct)) return false;
            if ((u.nchunks + 3) / 4 > (u.ct == 2 ? 1 : 4)) return false;
            u.lds_p = -1;
            if (L.slabs == 0) { u.lds_p = lds[it.wg]; lds[it.wg] += 3 * 64 * 4; }
            else if (u.slice == 0) { u.lds_p = lds[it.wg]; lds[it.wg] += 128 * u.ct; }      // bias of its columns
        } else {
            u.lds_p = lds[it.wg];
            lds[it.wg] += 2 * plan.layers[u.layer].gsz * 4;                                  // gamma | beta
        }
        if (lds[it.wg] > LP_DYN_LDS) return false;                 // the launch's dynamic LDS (64 bytes left to the static flag word)
        plan.units[(size_t)it.wg * LP_MAX_UNITS + count[it.wg]++] = u;
    }
    // at most one output.2 unit per workgroup (it keeps its piece of the state in registers)
    for (int w = 0; w < LP_WGS; ++w) {
        int n = 0;
        for (int i = 0; i < LP_MAX_UNITS; ++i) {
            const LpUnit& u = plan.units[(size_t)w * LP_MAX_UNITS + i];
            if (u.kind == LP_GEMM && u.layer == LP_LAYERS - 1) { ++n; if (u.ct != 1) return false; }
        }
        if (n > 1) return false;
    }
    return true;
}

}  // namespace

// Host-only self check of the static work assignment (no HIP calls; tests/test_abi_cpu.py): every (layer, 32-column tile,
// 64-k chunk) is computed by exactly one gemm unit, every (layer with slabs, group, row) by exactly one finish unit, no
// workgroup's LDS plan overflows.  Returns the bytes of one buffer set, or -1.
extern "C" int pcd_latent_persist_plan_check(void) {
    pcd_latent_desc_t d{};
    for (int i = 0; i < LP_LAYERS; ++i) { d.lin[i].k = kK1[i] + kK2[i]; d.lin[i].c = kC[i]; }
    LpPlan plan;
    if (!lp_build_plan(d, plan)) return -1;
    std::vector<std::vector<int>> cover(LP_LAYERS), fin(LP_LAYERS);
    for (int l = 0; l < LP_LAYERS; ++l) {
        cover[l].assign((size_t)(kC[l] / 32) * ((kK1[l] + kK2[l]) / 64), 0);
        fin[l].assign(8 * 32, 0);
    }
    for (int w = 0; w < LP_WGS; ++w) {
        int last_phase = -1;
        bool ended = false;
        for (int i = 0; i < LP_MAX_UNITS; ++i) {
            const LpUnit& u = plan.units[(size_t)w * LP_MAX_UNITS + i];
            if (u.kind == LP_END) { ended = true; continue; }
            if (ended) return -1;
            const int phase = 2 * u.layer + (u.kind == LP_FINISH ? 1 : 0);
            if (phase <= last_phase) return -1;                 // at most one unit of a phase per workgroup (two would run one after the other)
            last_phase = phase;
            // partial tiles stored plain: the finish unit of a group sits on the XCD of that group's gemm units
            if (u.kind == LP_FINISH && plan.layers[u.layer].slab_local) {
                const int gsz = kC[u.layer] / 8;
                for (int w2 = 0; w2 < LP_WGS; ++w2)
                    for (int i2 = 0; i2 < LP_MAX_UNITS; ++i2) {
                        const LpUnit& v = plan.units[(size_t)w2 * LP_MAX_UNITS + i2];
                        if (v.kind == LP_GEMM && v.layer == u.layer && v.col0 / gsz == u.group && w2 / 32 != w / 32) return -1;
                    }
            }
            // a layer whose outputs are stored plain (XCD-local) must be produced AND consumed on XCD 0 only (virtual ids < 32)
            if (plan.layers[u.layer].out_local && w >= 32) return -1;
            if (u.kind == LP_GEMM) {
                const int s1 = kSrc1[u.layer], s2 = kSrc2[u.layer];
                const int n1 = kK1[u.layer] / 64;
                for (int c = u.chunk0; c < u.chunk0 + u.nchunks; ++c) {
                    int src = c < n1 ? s1 : s2;
                    if (c < n1 && src < 0) src = LP_LAYERS - 1;       // enc1 reads the state published by output.2
                    if (src >= 0 && plan.layers[src].out_local && w >= 32) return -1;
                }
            }
            const int nk = (kK1[u.layer] + kK2[u.layer]) / 64;
            if (u.kind == LP_GEMM) {
                if (u.col0 % 32 || u.chunk0 + u.nchunks > nk || (kSlabs[u.layer] == 0 && u.nchunks != nk)) return -1;
                if (kSlabs[u.layer] > 0 && (u.slice < 0 || u.slice >= kSlabs[u.layer])) return -1;
                for (int t = 0; t < u.ct; ++t)
                    for (int c = 0; c < u.nchunks; ++c) cover[u.layer][(size_t)(u.col0 / 32 + t) * nk + u.chunk0 + c]++;
            } else {
                const int gsz = kC[u.layer] / 8, rows = gsz >= 64 ? 8 : 16;
                if (kSlabs[u.layer] == 0 || u.group < 0 || u.group >= 8 || u.row0 % rows) return -1;
                for (int r = 0; r < rows; ++r) fin[u.layer][u.group * 32 + u.row0 + r]++;
            }
        }
        if (!ended) return -1;
    }
    for (int l = 0; l < LP_LAYERS; ++l) {
        for (int v : cover[l]) if (v != 1) return -1;
        for (int v : fin[l]) if (v != (kSlabs[l] > 0 ? 1 : 0)) return -1;
        // slices of a slabbed layer partition K in order: slice s covers chunks [s * nk / S, (s + 1) * nk / S)
    }
    return plan.set_bytes;
}

// host-only: phase ids (2 * layer + (finish ? 1 : 0), -1 = none) of every workgroup's unit list, [256][8] ints (tools/trace_latent_persist.py)
extern "C" int pcd_latent_persist_plan_dump(int* phases_host) {
    PCD_CHECK_ARG(phases_host != nullptr);
    pcd_latent_desc_t d{};
    for (int i = 0; i < LP_LAYERS; ++i) { d.lin[i].k = kK1[i] + kK2[i]; d.lin[i].c = kC[i]; }
    LpPlan plan;
    PCD_CHECK_ARG(lp_build_plan(d, plan));
    for (size_t i = 0; i < plan.units.size(); ++i) {
        const LpUnit& u = plan.units[i];
        phases_host[i] = u.kind == LP_END ? -1 : 2 * u.layer + (u.kind == LP_FINISH ? 1 : 0);
    }
    return PCD_OK;
}

struct pcd_latent_persist {
    LpPlan plan;
    LpLayer* d_layers = nullptr;
    LpUnit* d_units = nullptr;
    int* d_sched = nullptr;            // [LP_WGS][LP_SCHED]: the merged unit order of the two interleaved streams (batch > 32)
    int sleep = 0;                     // back-off between polling passes (s_sleep count): 0 measured best (37.1 v. 37.3 / 37.5 / 37.9 us at 1 / 2 / 4)
    int predict = 1;
    unsigned* trace = nullptr;         // diagnostic: device buffer [256][trace_steps][8][8] u32
    int trace_steps = 0;
    int fault_wg = -1, fault_step = -1;        // pcd_latent_persist_inject_fault
};

// Merged order of the two streams per workgroup (one period = one step of stream 0).  Key of a unit = when its phase becomes runnable in
// a step, in 1/100 us (single-stream timeline, profiles/r03_c), + the lag for stream 1 (units whose key passes the period belong to the
// PREVIOUS step of stream 1 and come first); ties go to stream 0: the same total order on every workgroup.  Lag 15 us: the best of a sweep
// (10 ... 40 us: 55.8, 51.2, 52.7, 52.9, 56.6, 54.9, 60.8 us per step of the pair).
static hipError_t lp_upload_sched(pcd_latent_persist* h) {
    static const int kWhen[2 * LP_LAYERS] = {190, 190, 400, 400, 620, 620, 960, 1150, 1390, 1580, 1900, 2080, 2450, 2680, 2920, 3110, 3310, 3470,
                                             3610, 3610, 3800, 3800, 3960, 3960};
    const int period = 3970, lag = 1500;
    std::vector<int> sched((size_t)LP_WGS * LP_SCHED, -1);
    for (int w = 0; w < LP_WGS; ++w) {
        struct Ent { int key, st, u, prev; };
        std::vector<Ent> ents;
        int nunits = 0;
        for (int i = 0; i < LP_MAX_UNITS; ++i) {
            const LpUnit& u = h->plan.units[(size_t)w * LP_MAX_UNITS + i];
            if (u.kind == LP_END) break;
            const int when = kWhen[2 * u.layer + (u.kind == LP_FINISH ? 1 : 0)];
            ents.push_back({when, 0, i, 0});
            const int k1 = when + lag;
            if (k1 >= period) ents.push_back({k1 - period, 1, i, 1}); else ents.push_back({k1, 1, i, 0});
            ++nunits;
        }
        if ((int)ents.size() > LP_SCHED) return hipErrorInvalidValue;
        std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key != b.key ? a.key < b.key : a.st < b.st; });
        for (size_t i = 0; i < ents.size(); ++i)
            sched[(size_t)w * LP_SCHED + i] = ents[i].u | ents[i].st << 8 | ents[i].prev << 9 | (ents[i].u == nunits - 1 ? 1 << 10 : 0);
    }
    return hipMemcpy(h->d_sched, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice);
}

extern "C" int pcd_latent_persist_supported(int batch) {
    if (batch <= 0 || batch > 64) return 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    // a CU mask takes CUs away from this process: the 256 co-resident workgroups would never all be scheduled
    for (const char* name : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"})
        if (const char* v = getenv(name); v != nullptr && v[0] != '\0') return 0;
    // one workgroup per CU, all resident for the whole call: exactly the 256 CUs of an MI355X, all of its LDS
    return (prop.multiProcessorCount == LP_WGS && (int)prop.maxSharedMemoryPerMultiProcessor >= LP_LDS_BYTES) ? 1 : 0;
}

extern "C" int pcd_latent_persist_create(const pcd_latent_desc_t* desc, pcd_latent_persist_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    for (int i = 0; i < LP_LAYERS; ++i)
        PCD_CHECK_ARG(desc->lin[i].w && desc->lin[i].b && desc->lin[i].k == kK1[i] + kK2[i] && desc->lin[i].c == kC[i] &&
                      (i >= 10 || (desc->gn_gamma[i] && desc->gn_beta[i])));
    pcd_latent_persist* h = new (std::nothrow) pcd_latent_persist;
    PCD_CHECK_ARG(h != nullptr);
    if (!lp_build_plan(*desc, h->plan)) {
        delete h;
        set_error("pcd_latent_persist_create: the static work assignment does not fit");
        return PCD_ERR_ARG;
    }
    hipError_t e = hipMalloc(&h->d_layers, sizeof(LpLayer) * LP_LAYERS);
    if (e == hipSuccess) e = hipMalloc(&h->d_units, sizeof(LpUnit) * h->plan.units.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_layers, h->plan.layers, sizeof(LpLayer) * LP_LAYERS, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_units, h->plan.units.data(), sizeof(LpUnit) * h->plan.units.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->d_sched, (size_t)LP_WGS * LP_SCHED * sizeof(int));
    if (e == hipSuccess) e = lp_upload_sched(h);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)latent_persist_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LP_DYN_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)latent_persist_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LP_DYN_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)latent_persist_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LP_DYN_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)latent_persist_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LP_DYN_LDS);
    if (e != hipSuccess) {
        set_error("pcd_latent_persist_create: %s", hipGetErrorString(e));
        if (h->d_layers) (void)hipFree(h->d_layers);
        if (h->d_units) (void)hipFree(h->d_units);
        if (h->d_sched) (void)hipFree(h->d_sched);
        delete h;
        return PCD_ERR_HIP;
    }
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_latent_persist_destroy(pcd_latent_persist_t* h) {
    if (!h) return;
    if (h->d_layers) (void)hipFree(h->d_layers);
    if (h->d_units) (void)hipFree(h->d_units);
    if (h->d_sched) (void)hipFree(h->d_sched);
    delete h;
}

extern "C" size_t pcd_latent_persist_workspace_bytes(const pcd_latent_persist_t* h) {
    return h ? (size_t)LP_CTRL_BYTES + 2 * 3 * (size_t)h->plan.set_bytes : 0;          // two streams x three sets
}

// diagnostic: the next launches run the instrumented kernel and leave s_memrealtime stamps (100 MHz) of the first
// `steps` steps in `buf` ([256 workgroups][steps][8 units][8]: enter, operands in, stored, epilogue computed, operands in of waves 0..3); buf = NULL switches it off
extern "C" int pcd_latent_persist_trace(pcd_latent_persist_t* h, unsigned* buf, int steps) {
    PCD_CHECK_ARG(h != nullptr && (buf == nullptr || steps > 0));
    h->trace = buf;
    h->trace_steps = buf ? steps : 0;
    return PCD_OK;
}

extern "C" int pcd_latent_persist_config(pcd_latent_persist_t* h, int poll_sleep, int predict_waits) {
    PCD_CHECK_ARG(h != nullptr && poll_sleep >= 0 && poll_sleep <= 127);
    h->sleep = poll_sleep;
    h->predict = predict_waits < 0 ? 0 : (predict_waits > 3 ? 3 : predict_waits);
    return PCD_OK;
}

extern "C" int pcd_latent_persist_inject_fault(pcd_latent_persist_t* h, int workgroup, int step) {
    PCD_CHECK_ARG(h != nullptr && workgroup < LP_WGS);
    h->fault_wg = workgroup;
    h->fault_step = workgroup < 0 ? -1 : step;
    return PCD_OK;
}

static int lp_launch(pcd_latent_persist_t* h, LpArgs& a, void* workspace, size_t workspace_bytes, void* stream) {
    const size_t need = pcd_latent_persist_workspace_bytes(h);
    if (workspace_bytes < need) {
        set_error("pcd_latent_persist: workspace %zu < required %zu", workspace_bytes, need);
        return PCD_ERR_WORKSPACE;
    }
    PCD_CHECK_ARG(((uintptr_t)workspace & 255) == 0);
    if (!pcd_latent_persist_supported(a.batch)) {
        set_error("pcd_latent_persist: needs batch <= 64 and a device with exactly %d CUs of %d KB LDS", LP_WGS, LP_LDS_BYTES / 1024);
        return PCD_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    a.layers = h->d_layers;
    a.units = h->d_units;
    a.ws = (char*)workspace;
    a.set_bytes = h->plan.set_bytes;
    a.sleep = h->sleep;
    a.predict = h->predict;
    a.trace = h->trace;
    a.trace_steps = h->trace_steps;
    a.fault_wg = h->fault_wg;
    a.fault_step = h->fault_step;
    // a batch of more than 32 rows: two interleaved streams (rows 0 .. 31 | the rest)
    const int ns = a.batch > 32 ? 2 : 1;
    a.sched = ns == 2 ? h->d_sched : nullptr;
    // status word cleared, every set poisoned (a stream's set 2 receives its initial state before step 0 and is re-poisoned at the end of step 0)
    PCD_CHECK_HIP(hipMemsetAsync(workspace, 0, LP_CTRL_BYTES, s));
    PCD_CHECK_HIP(hipMemsetAsync((char*)workspace + LP_CTRL_BYTES, 0xff, (size_t)ns * 3 * (size_t)h->plan.set_bytes, s));
    // dynamic LDS = everything but the static flag words (rounded)
    if (ns == 2) {
        if (a.trace != nullptr) hipLaunchKernelGGL((latent_persist_kernel<true, 2>), dim3(LP_WGS), dim3(LP_THREADS), LP_DYN_LDS, s, a);
        else hipLaunchKernelGGL((latent_persist_kernel<false, 2>), dim3(LP_WGS), dim3(LP_THREADS), LP_DYN_LDS, s, a);
    } else {
        if (a.trace != nullptr) hipLaunchKernelGGL((latent_persist_kernel<true, 1>), dim3(LP_WGS), dim3(LP_THREADS), LP_DYN_LDS, s, a);
        else hipLaunchKernelGGL((latent_persist_kernel<false, 1>), dim3(LP_WGS), dim3(LP_THREADS), LP_DYN_LDS, s, a);
    }
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_latent_persist_forward(pcd_latent_persist_t* h, const float* z, int batch, const float* tbias, float* eps,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(h && z && tbias && eps && workspace && batch > 0);
    LpArgs a{};
    a.z = (float*)z; a.x0 = nullptr; a.eps_out = eps; a.batch = batch;
    a.tb_table = tbias; a.tb_elems = 128;
    a.rates = nullptr; a.rate_width = 1; a.rate_stride = 0; a.T = 1;
    a.counter = nullptr; a.nsteps = 1; a.forward_only = 1;
    return lp_launch(h, a, workspace, workspace_bytes, stream);
}

extern "C" int pcd_latent_persist_ddim(pcd_latent_persist_t* h, float* z, float* x0, int batch, const float* tb_table, int tb_elems,
                                       const float* rate_tables, int rate_width, int n_steps_table, int* counter, int nsteps,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(h && z && tb_table && rate_tables && counter && workspace && batch > 0 && nsteps > 0 && n_steps_table > 0);
    PCD_CHECK_ARG(tb_elems == 128 && (rate_width == 1 || rate_width == batch));
    LpArgs a{};
    a.z = z; a.x0 = x0; a.eps_out = nullptr; a.batch = batch;
    a.tb_table = tb_table; a.tb_elems = tb_elems;
    a.rates = rate_tables; a.rate_width = rate_width; a.rate_stride = rate_width == 1 ? 0 : 1; a.T = n_steps_table;
    a.counter = counter; a.nsteps = nsteps; a.forward_only = 0;
    return lp_launch(h, a, workspace, workspace_bytes, stream);
}

extern "C" int pcd_latent_persist_status(const void* workspace, unsigned* status_host, void* stream) {
    PCD_CHECK_ARG(workspace && status_host);
    // the copy is ordered behind the launch on the launch's own stream and only that stream is waited for: other streams of the
    // process (another model, a copy engine) keep running
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemcpyAsync(status_host, workspace, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    PCD_CHECK_HIP(hipStreamSynchronize(s));
    return PCD_OK;
}
