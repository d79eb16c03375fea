// The feed-forward half of SetAttentionBlock at C = 256 (reference networks.py:62-68, 82) as ONE launch:
//     y = x1 + W2 relu(W1 LN2(x1) + b1) + b2            x1, y fp16 [M][256], hidden width 1024
// Before: LN2 + Linear(256, 1024) + ReLU (pw_wide_chain_kernel<true>, 126 us at B = 64, N = 2048) wrote a 268-MB hidden tensor that Linear(1024, 256) + residual
// (gemm_xp_kernel<RESID>, 110 us, HBM-bound at 3.8 TB/s) read back.  The register-resident chain of widechain.hip cannot hold this layer pair: a wave that owns 32
// points for ALL channels needs the 256-wide input (64 registers), a hidden slab's accumulators and fragments, and the 256-wide output accumulators (128) at once --
// ~400 registers (DESIGN, round 3).  Here TWO waves share 32 points and split the CHANNELS of both products:
//   * a workgroup = 8 waves = 4 pairs = 128 points; both waves of a pair hold the normalised input as B fragments (16 k-steps x 4 registers = 64);
//   * the hidden layer is walked in 8 slabs of 128 channels.  Phase A: wave h of a pair computes hidden channels [64 h, 64 h + 64) of the slab
//     (2 accumulator blocks of 32 x 32 = 32 registers, K = 256); bias + ReLU + fp16, regrouped with v_permlane32_swap into 4 B fragments (16 registers) -- its half
//     of the slab as the next product's K -- and written to a 4-KB LDS slot; the partner's half is read from the partner's slot behind the next stage barrier
//     (16 more registers).  Phase B: wave h accumulates OUTPUT channels [128 h, 128 h + 128) (4 blocks = 64 registers, alive over all slabs) over the slab's 128 k;
//   * only weights stream: 32 fragment-order stage images of 32 KB per 128-point tile (per slab: W1's 128 rows x K 256 as two images, W2's 256 rows x the slab's
//     128 k as two images) through the 3-deep LDS-DMA ring of widechain.hip, one barrier per image, 16 MFMAs (32x32x16) per wave and image;
//   * epilogue: + b2 + x1 (re-read, 16 bytes per lane and piece), fp16, 16-byte stores.
// Registers: 64 (input) + 64 (output accumulators) + 32 + 16 + 16 + weight fragments: two waves per SIMD.  LDS: 96 KB ring + 32 KB exchange + 7 KB constants.
// 128 FLOP per filled byte (a 128-point tile re-streams the 1 MB of weights); the 268-MB hidden tensor never exists.  rows % 128 == 0.
#include "common.h"

namespace pcd {

constexpr int WF_WAVES = 8, WF_THREADS = 64 * WF_WAVES, WF_TILE = 128;
constexpr int WF_STAGE = 32768, WF_RING = 3, WF_SLABS = 8, WF_STAGES_PER_TILE = 4 * WF_SLABS;
constexpr int WF_EXCH = WF_WAVES * 4096;
constexpr int WF_NCONST = 1024 + 256 + 256 + 256;               // b1 | b2 | gamma | beta (fp32)
constexpr size_t WF_IMG_BYTES = (size_t)WF_STAGES_PER_TILE * WF_STAGE;
constexpr size_t WF_LDS = (size_t)WF_RING * WF_STAGE + WF_EXCH + WF_NCONST * sizeof(float);

struct WfParams {
    const half_t* x;                   // [M][256]
    const char* wpacked;               // 32 stage images, then the constants
    half_t* y;                         // [M][256]
    int64_t m;
    const float* post_e;               // optional: y leaves as fp16(y + post_e[row / rps][c]) (the additive time embedding behind the block, networks.py:688)
    int64_t estride; int rps;
};

__device__ __forceinline__ void wf_dma(const char* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// two accumulator groups of one 32-channel block (g = 2 gp, 2 gp + 1) -> the 8 consecutive channels 16 gp + 8 hh .. + 7 of the block a lane holds as a B fragment /
// output piece (widechain.hip's regroup: accumulator register 4 g + e = channel 8 g + 4 hh + e of the block, point = lane & 31)
__device__ __forceinline__ half8 wf_regroup(const f32x16& acc, int gp, const float* bias_blk, int hh, float lo) {
    unsigned p[2], q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g0 = 2 * gp, g1 = 2 * gp + 1;
        const float a0 = acc[4 * g0 + 2 * h] + bias_blk[8 * g0 + 4 * hh + 2 * h], a1 = acc[4 * g0 + 2 * h + 1] + bias_blk[8 * g0 + 4 * hh + 2 * h + 1];
        const float b0 = acc[4 * g1 + 2 * h] + bias_blk[8 * g1 + 4 * hh + 2 * h], b1 = acc[4 * g1 + 2 * h + 1] + bias_blk[8 * g1 + 4 * hh + 2 * h + 1];
        half2_ pa, pb;
        pa.x = (half_t)__builtin_amdgcn_fmed3f(a0, lo, 65504.f); pa.y = (half_t)__builtin_amdgcn_fmed3f(a1, lo, 65504.f);
        pb.x = (half_t)__builtin_amdgcn_fmed3f(b0, lo, 65504.f); pb.y = (half_t)__builtin_amdgcn_fmed3f(b1, lo, 65504.f);
        p[h] = __builtin_bit_cast(unsigned, pa);
        q[h] = __builtin_bit_cast(unsigned, pb);
    }
    unsigned f[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const auto r = __builtin_amdgcn_permlane32_swap(p[h], q[h], false, false);
        f[h] = r[0];
        f[2 + h] = r[1];
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, (u4){f[0], f[1], f[2], f[3]});
}

// SPLIT: the 32 LDS-DMA pieces of an image are requested by ONE wave of each SIMD (waves 0-3 for even images, 4-7 for odd ones, 8 pieces each) instead of 4 pieces by
// every wave: a piece holds its wave's issue for 60-180 cycles, and with every wave requesting behind the barrier no wave of a SIMD issues MFMAs meanwhile
// ABL: timing ablations for tools/bench_wide_ffn.py (pcd_wide_ffn_config(16 + bits); OUTPUTS ARE WRONG while set): 1 = no image requests in the loop (the ring keeps the
// first three images), 2 = no fragment reads (every MFMA takes the fragments of the first one), 4 = no waits / barriers in the loop, 8 = no MFMAs
template <bool SPLIT, int ABL = 0>
__global__ __launch_bounds__(WF_THREADS, 2) void wide_ffn_kernel(WfParams p) {
    extern __shared__ __attribute__((aligned(16))) char wf_smem[];          // [WF_RING][WF_STAGE] | exchange [8 waves][4 KB] | b1 | b2 | gamma | beta
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, h = wave & 1;
    const int pnt = lane & 31, hh = lane >> 5;
    char* const exch = wf_smem + WF_RING * WF_STAGE;
    float* const cst = (float*)(exch + WF_EXCH);
    {
        const float* src = (const float*)(p.wpacked + WF_IMG_BYTES);
        for (int i = threadIdx.x; i < WF_NCONST; i += WF_THREADS) cst[i] = src[i];
    }
    const float* const b1 = cst, * const b2 = cst + 1024, * const gam = cst + 1280, * const bet = cst + 1536;
    const unsigned lds0 = (unsigned)(size_t)wf_smem;
    const int64_t ntiles = p.m / WF_TILE;
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total_stages = my_tiles * WF_STAGES_PER_TILE;
    // image n of this workgroup's run = stage image n % 32; wave w moves pieces 4 w .. 4 w + 3 of its 32 (SPLIT: the waves of group n & 1 move 8 w' .. 8 w' + 7)
    auto issue = [&](int n) __attribute__((always_inline)) {
        if ((ABL & 1) && n >= WF_RING) return;
        if (n < total_stages) {
            if constexpr (SPLIT) {
                if ((wave >> 2) == (n & 1)) {
                    const int w4 = wave & 3;
                    const char* src = p.wpacked + (size_t)(n % WF_STAGES_PER_TILE) * WF_STAGE + (size_t)(8 * w4) * 1024 + lane * 16;
                    const unsigned dst = lds0 + (n % WF_RING) * WF_STAGE + (8 * w4) * 1024;
#pragma unroll
                    for (int i = 0; i < 8; ++i) wf_dma(src + i * 1024, dst + i * 1024);
                }
            } else {
                const char* src = p.wpacked + (size_t)(n % WF_STAGES_PER_TILE) * WF_STAGE + (size_t)(4 * wave) * 1024 + lane * 16;
                const unsigned dst = lds0 + (n % WF_RING) * WF_STAGE + (4 * wave) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf_dma(src + i * 1024, dst + i * 1024);
            }
        }
    };
    issue(0);
    issue(1);
    __syncthreads();                                           // the constants are read before the first stage barrier
    int n = 0;                                                 // next image to consume
    // image n has landed (all but this wave's 4 youngest LDS-DMA pieces; SPLIT: every piece of the group that requested it), every wave is done with image n - 1
    // (its LDS reads included): its slot takes image n + 2
    auto acquire = [&]() __attribute__((always_inline)) -> const char* {
        if constexpr ((ABL & 4) != 0) {
            issue(n + 2);
            const char* img0 = wf_smem + (n % WF_RING) * WF_STAGE + lane * 16;
            ++n;
            return img0;
        }
        if constexpr (SPLIT) {
            if ((wave >> 2) == (n & 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (n + 1 < total_stages) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();
        issue(n + 2);
        const char* img = wf_smem + (n % WF_RING) * WF_STAGE + lane * 16;
        ++n;
        return img;
    };
    char* const my_slot = exch + wave * 4096 + lane * 16;
    const char* const partner_slot = exch + (wave ^ 1) * 4096 + lane * 16;

    // the pair's 32 points x 256 channels as B fragments; the rows of the NEXT tile are requested before a tile's epilogue (the registers are free by then), so
    // their latency sits under the epilogue's arithmetic, loads and stores instead of in front of the next tile's first MFMA
    half8 xin[16];
    auto request_rows = [&](int ti) __attribute__((always_inline)) {
        const int64_t pt = (blockIdx.x + (int64_t)ti * gridDim.x) * WF_TILE + pair * 32 + pnt;
        const half_t* row = p.x + pt * 256 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 16; ++s) xin[s] = *(const half8*)(row + 16 * s);
    };
    if (my_tiles > 0) request_rows(0);
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int64_t tile = blockIdx.x + (int64_t)ti * gridDim.x;
        const int64_t pt = tile * WF_TILE + pair * 32 + pnt;
        // ---- LayerNorm on the fragments (widechain.hip's LN prologue: two passes, fp32 statistics)
        {
            float sum = 0.f, sq = 0.f;
            half2_ one2; one2.x = one2.y = (half_t)1.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_ v; v.x = xin[s][2 * e]; v.y = xin[s][2 * e + 1];
                    sum = __builtin_amdgcn_fdot2(v, one2, sum, false);
                }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.f / 256.f);
            const half_t mh = (half_t)__builtin_amdgcn_fmed3f(mean, -65504.f, 65504.f);
            half2_ mh2; mh2.x = mh2.y = mh;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_ v; v.x = xin[s][2 * e]; v.y = xin[s][2 * e + 1];
                    const half2_ d = v - mh2;
                    sq = __builtin_amdgcn_fdot2(d, d, sq, false);
                }
            sq += __shfl_xor(sq, 32);
            const float shift = mean - (float)mh;
            const float rstd = rsqrtf(fmaxf(sq - 256.f * shift * shift, 0.f) * (1.f / 256.f) + 1e-5f);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const f32x4 g0 = *(const f32x4*)&gam[16 * s + 8 * hh], g1 = *(const f32x4*)&gam[16 * s + 8 * hh + 4];
                const f32x4 c0 = *(const f32x4*)&bet[16 * s + 8 * hh], c1 = *(const f32x4*)&bet[16 * s + 8 * hh + 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xin[s][e] = (half_t)__builtin_amdgcn_fmed3f(((float)xin[s][e] - mean) * rstd * g0[e] + c0[e], -65504.f, 65504.f);
                    xin[s][4 + e] = (half_t)__builtin_amdgcn_fmed3f(((float)xin[s][4 + e] - mean) * rstd * g1[e] + c1[e], -65504.f, 65504.f);
                }
            }
        }
        f32x16 accY[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) accY[b][e] = 0.f;
#pragma unroll 1
        for (int slab = 0; slab < WF_SLABS; ++slab) {
            // ---- phase A: hidden channels slab * 128 + 64 h .. + 63 of the pair's points, K = 256 in two images ([4 blocks][8 k-steps] fragments each)
            f32x16 accH[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) accH[b][e] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* img = acquire() + (size_t)(2 * h) * 8 * 1024;
#pragma unroll
                for (int qq = 0; qq < 8; qq += 2) {
                    half8 af[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int b = 0; b < 2; ++b) af[u][b] = *(const half8*)(img + ((ABL & 2) ? 0 : (b * 8 + qq + u) * 1024));
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if constexpr ((ABL & 8) != 0) accH[b][0] += (float)af[u][b][0];
                            else accH[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][b], xin[8 * j + qq + u], accH[b], 0, 0, 0);
                        }
                }
            }
            // bias + ReLU + fp16 -> this wave's half of the slab as B fragments (4 k-steps of 16 hidden channels), shared with the partner through LDS
            half8 hown[4], hoth[4];
            {
                const float* bb = b1 + slab * 128 + 64 * h;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        hown[2 * b + gp] = wf_regroup(accH[b], gp, bb + 32 * b, hh, 0.f);
                        *(half8*)(my_slot + (2 * b + gp) * 1024) = hown[2 * b + gp];
                    }
            }
            // ---- phase B: output channels 128 h .. + 127 += W2[., slab's 128 k] . hidden; image j carries the k of half j ([8 blocks][4 k-steps] fragments)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* img = acquire() + (size_t)(4 * h) * 4 * 1024;
                if (j == 0) {
                    // behind the barrier of `acquire` every wave's slot is written; the partner's four fragments
#pragma unroll
                    for (int i = 0; i < 4; ++i) hoth[i] = *(const half8*)(partner_slot + i * 1024);
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    half8 af[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) af[b] = *(const half8*)(img + ((ABL & 2) ? 0 : (b * 4 + qq) * 1024));
                    const half8 hf = (j == h) ? hown[qq] : hoth[qq];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if constexpr ((ABL & 8) != 0) accY[b][0] += (float)af[b][0] * (float)hf[0];
                        else accY[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b], hf, accY[b], 0, 0, 0);
                    }
                }
            }
        }
        if (ti + 1 < my_tiles) request_rows(ti + 1);
        // ---- epilogue: y = x1 + (accY + b2) rounded like the two-launch form (the GEMM's fp16 result, then the fp16 residual add)
        {
            const half_t* xrow = p.x + pt * 256 + 128 * h + 8 * hh;
            half_t* yrow = p.y + pt * 256 + 128 * h + 8 * hh;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    half8 v = wf_regroup(accY[b], gp, b2 + 128 * h + 32 * b, hh, -65504.f);
                    const half8 r = *(const half8*)(xrow + 32 * b + 16 * gp);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = to_half_sat((float)v[e] + (float)r[e]);
                    if (p.post_e != nullptr) {                 // the rounding points of pcd_add_shape_bias_strided_f16 behind the block: bitwise the two launches
                        const float* er = p.post_e + (pt / p.rps) * p.estride + 128 * h + 8 * hh + 32 * b + 16 * gp;
                        const f32x4 e0 = *(const f32x4*)er, e1 = *(const f32x4*)(er + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = to_half_sat((float)v[e] + e0[e]); v[4 + e] = to_half_sat((float)v[4 + e] + e1[e]); }
                    }
                    *(half8*)(yrow + 32 * b + 16 * gp) = v;
                }
        }
    }
}

// W [rows][ldw] fp16 -> one stage image of nblk x nq fragments: fragment (blk, qq) = rows row0 + 32 blk .. + 31, columns k0 + 16 qq .. + 15
__global__ __launch_bounds__(256) void wf_pack_kernel(const half_t* __restrict__ w, int64_t ldw, int row0, int nblk, int k0, int nq, char* __restrict__ img) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nblk * nq * 64) return;
    const int lane = id & 63, fr = id >> 6, blk = fr / nq, qq = fr - blk * nq;
    *(half8*)(img + (size_t)id * 16) = *(const half8*)(w + (int64_t)(row0 + 32 * blk + (lane & 31)) * ldw + k0 + 16 * qq + 8 * (lane >> 5));
}

}  // namespace pcd

using namespace pcd;

static int g_wf_split = 1;          // pcd_wide_ffn_config: who requests the weight images (0: every wave 4 pieces; 1, default: one wave per SIMD 8 pieces, alternating
                                    // groups: 187 v. 210 us at B = 64, N = 2048, same bits)
static int g_wf_abl = 0;            // timing ablations (pcd_wide_ffn_config(16 + bits)); outputs are wrong while set
extern "C" int pcd_wide_ffn_config(int split) {
    if (split >= 16 && split < 32) { g_wf_abl = split - 16; return PCD_OK; }
    g_wf_split = split ? 1 : 0;
    return PCD_OK;
}

extern "C" size_t pcd_wide_ffn_packed_bytes(void) { return WF_IMG_BYTES + (size_t)WF_NCONST * sizeof(float); }

extern "C" int pcd_wide_ffn_supported(int dim, int64_t rows) { return dim == 256 && rows > 0 && rows % WF_TILE == 0 ? 1 : 0; }

// w1 [1024][256], b1 [1024], w2 [256][1024], b2 [256] (ff.0 / ff.2 of the block), LayerNorm affine of ln2
extern "C" int pcd_wide_ffn_pack(const void* w1, const float* b1, const void* w2, const float* b2, const float* ln_g, const float* ln_b, void* packed,
                                 void* stream) {
    PCD_CHECK_ARG(w1 && b1 && w2 && b2 && ln_g && ln_b && packed);
    hipStream_t s = (hipStream_t)stream;
    char* img = (char*)packed;
    for (int slab = 0; slab < WF_SLABS; ++slab)
        for (int j = 0; j < 2; ++j) {
            hipLaunchKernelGGL(wf_pack_kernel, dim3(8), dim3(256), 0, s, (const half_t*)w1, (int64_t)256, slab * 128, 4, 128 * j, 8,
                               img + (size_t)(4 * slab + j) * WF_STAGE);
            hipLaunchKernelGGL(wf_pack_kernel, dim3(8), dim3(256), 0, s, (const half_t*)w2, (int64_t)1024, 0, 8, slab * 128 + 64 * j, 4,
                               img + (size_t)(4 * slab + 2 + j) * WF_STAGE);
        }
    PCD_CHECK_LAUNCH();
    float* f = (float*)(img + WF_IMG_BYTES);
    PCD_CHECK_HIP(hipMemcpyAsync(f, b1, 1024 * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(f + 1024, b2, 256 * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(f + 1280, ln_g, 256 * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(f + 1536, ln_b, 256 * 4, hipMemcpyDeviceToDevice, s));
    return PCD_OK;
}

extern "C" int pcd_wide_ffn_f16(const void* packed, const void* x, int64_t rows, void* y, void* stream) {
    return pcd_wide_ffn_bias_f16(packed, x, rows, 1, nullptr, 0, y, stream);
}

// the same with y + post_e[row / rows_per_shape][.] (fp32 rows of 256, e_stride floats apart, 16-byte aligned) on the way out
extern "C" int pcd_wide_ffn_bias_f16(const void* packed, const void* x, int64_t rows, int rows_per_shape, const float* post_e, int64_t e_stride, void* y, void* stream) {
    PCD_CHECK_ARG(packed && x && y && rows > 0 && rows % WF_TILE == 0);
    PCD_CHECK_ARG(post_e == nullptr || (rows_per_shape > 0 && e_stride >= 0 && e_stride % 4 == 0 && ((uintptr_t)post_e & 15) == 0));
    static PcdLdsOnce once, once_split;
    PCD_CHECK_HIP(pcd_allow_lds(once, (const void*)wide_ffn_kernel<false>, (int)WF_LDS));
    PCD_CHECK_HIP(pcd_allow_lds(once_split, (const void*)wide_ffn_kernel<true>, (int)WF_LDS));
    WfParams p{};
    p.x = (const half_t*)x; p.wpacked = (const char*)packed; p.y = (half_t*)y; p.m = rows;
    p.post_e = post_e; p.estride = e_stride; p.rps = rows_per_shape > 0 ? rows_per_shape : 1;
    const int64_t tiles = rows / WF_TILE;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
#define PCD_WF_ABL(A)                                                                                                  \
    if (g_wf_abl == A) {                                                                                                \
        static PcdLdsOnce once_a;                                                                                       \
        PCD_CHECK_HIP(pcd_allow_lds(once_a, (const void*)wide_ffn_kernel<true, A>, (int)WF_LDS));                       \
        hipLaunchKernelGGL((wide_ffn_kernel<true, A>), dim3(grid), dim3(WF_THREADS), WF_LDS, (hipStream_t)stream, p);   \
        PCD_CHECK_LAUNCH();                                                                                             \
        return PCD_OK;                                                                                                  \
    }
    PCD_WF_ABL(1) PCD_WF_ABL(2) PCD_WF_ABL(3) PCD_WF_ABL(4) PCD_WF_ABL(5) PCD_WF_ABL(8) PCD_WF_ABL(9) PCD_WF_ABL(10) PCD_WF_ABL(7) PCD_WF_ABL(11)
#undef PCD_WF_ABL
    if (g_wf_split) hipLaunchKernelGGL(wide_ffn_kernel<true>, dim3(grid), dim3(WF_THREADS), WF_LDS, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wide_ffn_kernel<false>, dim3(grid), dim3(WF_THREADS), WF_LDS, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
