// The tail of SetAttentionBlock.forward (reference networks.py:78-83) behind the attention kernel, for C = 128 and C = 64, as ONE launch:
//     x1 = x + out_proj(a) ;  y = x1 + W2 relu(W1 LN2(x1) + b1) + b2          (a = the heads' outputs, x = the block's input)
// As four launches (out_proj GEMM + residual, LayerNorm, W1 GEMM + ReLU, W2 GEMM + residual) the tail costs 145 / 70 us per block at
// B = 64, N = 2048 for 39 / 9.7 GFLOP: K = 64 .. 512 GEMMs are latency / HBM bound, and the 4C-wide hidden tensor (134 / 67 MB) is written and
// read back.  Here a wave owns 32 points for the whole tail and nothing leaves its REGISTERS between the layers (the scheme of
// pw_wide_chain_kernel, csrc/widechain.hip):
//   * every product is transposed, D[channel][point] = W[channel][k] . act[point][k] on v_mfma_f32_32x32x16_f16 (weights = A operand,
//     activations = B operand); an accumulator group holds 4 consecutive channels of one point, and one v_permlane32_swap per register turns two
//     groups of the two lane halves into the 8 consecutive k of the next product's B fragment;
//   * x1 stays in fp32 in the accumulators: the residual x is added there, the LayerNorm statistics of a point are a per-lane sum over its
//     C / 2 channels + one exchange with lane ^ 32 (two passes: mean, then squared deviations), and the same registers then take
//     b2 and are the C operand of the W2 products -- y accumulates onto x1;
//   * the FFN streams over its 4C hidden channels in chunks of 64: h = relu(W1[chunk] . LN + b1) (two accumulator tiles) -> fp16 B fragments
//     -> y += W2[:, chunk] . h.  The hidden tensor never exists;
//   * only the weights move: packed once per block (pcd_sab_tail_pack) into stage images in fragment order -- [W_out] and, per chunk,
//     [W1 chunk | W2 chunk] -- they arrive by LDS-DMA in a three-stage ring shared by the workgroup's eight waves (256 points), one barrier
//     per stage, 262 FLOP per byte of LDS fill.
// sab_head_kernel below does the block's first half in the same form (LN1 + in_proj).  Rows must be a multiple of 256 (otherwise the caller keeps the
// separate launches).  Arithmetic: fp16 operands, fp32 accumulation; x1 and the
// LayerNorm in fp32 (the four-launch form rounds x1 and LN2(x1) to fp16 in between: this form is the closer one to the reference's fp32).
#include "common.h"

namespace pcd {

constexpr int ST_WAVES = 8, ST_THREADS = 64 * ST_WAVES, ST_TILE = 32 * ST_WAVES, ST_RING = 3;

template <int C>
struct StCfg {
    static constexpr int NT = C / 32;                  // 32-channel tiles of a C-wide output
    static constexpr int KS = C / 16;                  // 16-deep k steps of a C-wide input
    static constexpr int NCH = C / 16;                 // 64-wide chunks of the 4C hidden channels
    static constexpr int STAGE = C == 128 ? 32768 : 16384;      // bytes: [W1 chunk: 2 tiles x KS pieces | W2 chunk: NT tiles x 4 pieces] of 1 KB
    static constexpr int NSTG = 1 + NCH;               // stage images per tile of points: W_out, then the chunks
    static constexpr int PPW = STAGE / 1024 / ST_WAVES;         // LDS-DMA pieces per wave and stage
    static constexpr int W2OFF = 2 * KS * 1024;        // the W2 part of a chunk image
    // fp32 parameters behind the images: b_out [C] | ln2 gamma [C] | ln2 beta [C] | b_ff2 [C] | b_ff1 [4C]
    static constexpr int BO = 0, GA = C, BE = 2 * C, B2 = 3 * C, B1 = 4 * C, NPAR = 8 * C;
    // the head (LN1 + in_proj, sab_head_kernel) behind the tail's image: three stage images of W_in (one C-wide output pass each), then b_in [3C] | ln1 gamma [C] |
    // ln1 beta [C]
    static constexpr int HSTAGE = NT * KS * 1024;      // one pass of W_in: C x C fp16
    static constexpr int HPPW = HSTAGE / 1024 / ST_WAVES;
    static constexpr int HB = 0, HGA = 3 * C, HBE = 4 * C, HNPAR = 5 * C;
    static constexpr size_t TAIL_BYTES = (size_t)NSTG * STAGE + NPAR * sizeof(float);
    static constexpr size_t HEAD_BYTES = (size_t)3 * HSTAGE + HNPAR * sizeof(float);
    static_assert(C == 64 || C == 128, "the tail kernel is built for C = 64 and C = 128");
    static_assert((2 * KS + 4 * NT) * 1024 == STAGE && NT * KS * 1024 <= STAGE, "stage image layout");
};

struct SabTailParams {
    const half_t* a;          // [M][C] attention output (heads concatenated)
    const half_t* x;          // [M][C] the block's input (residual)
    const char* packed;       // NSTG stage images, then NPAR floats
    half_t* y;                // [M][C]
    int64_t m;
    // the attention U-Net's additive per-level time embeddings (networks.py:669-698) inside the launch: x is read as fp16(x + pre_e[shape]) (the block
    // runs on x + emb), y leaves as fp16(y + post_e[shape]) (the skip tensor is block(x) + emb); each exactly as pcd_add_shape_bias_strided_f16 rounds it
    const float* pre_e; const float* post_e; int64_t estride; int rps;
    int split;                // 1: one wave per SIMD requests a stage's LDS-DMA pieces (pcd_sab_tail_config bit 1)
};

__device__ __forceinline__ void st_dma(const char* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// v0 / v1: the four values (already biased / normalised / clamped) of accumulator groups 2 gp and 2 gp + 1 of one 32-channel tile.  Lane half 0
// ends with the tile's channels 16 gp .. + 7, lane half 1 with 16 gp + 8 .. + 15: the next product's B fragment / one 16-byte output piece.
__device__ __forceinline__ half8 st_pack_swap(const float (&v0)[4], const float (&v1)[4]) {
    unsigned f[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        half2_ pa, pb;
        pa.x = (half_t)v0[2 * h]; pa.y = (half_t)v0[2 * h + 1];
        pb.x = (half_t)v1[2 * h]; pb.y = (half_t)v1[2 * h + 1];
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
        f[h] = r[0];
        f[2 + h] = r[1];
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, (u4){f[0], f[1], f[2], f[3]});
}

template <int C>
__global__ __launch_bounds__(ST_THREADS, 2) void sab_tail_kernel(SabTailParams p) {
    using K = StCfg<C>;
    constexpr int NT = K::NT, KS = K::KS;
    extern __shared__ __attribute__((aligned(16))) char st_smem[];          // [ST_RING][STAGE] | parameters
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pnt = lane & 31, hh = lane >> 5;
    float* par = (float*)(st_smem + ST_RING * K::STAGE);
    {
        const float* src = (const float*)(p.packed + (size_t)K::NSTG * K::STAGE);
        for (int i = threadIdx.x; i < K::NPAR; i += ST_THREADS) par[i] = src[i];
    }
    const unsigned lds0 = (unsigned)(size_t)st_smem;
    const int64_t ntiles = p.m / ST_TILE;
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total_stages = my_tiles * K::NSTG;
    // stage n of this workgroup's run = image n % NSTG; wave w moves pieces PPW w .. PPW w + PPW - 1
    // (p.split: the pieces of stage n are requested by ONE wave of each SIMD -- waves 0-3 for even stages, 4-7 for odd ones, 2 PPW pieces each -- so that its SIMD partner
    // issues MFMAs meanwhile: round 5, as csrc/wideffn.hip)
    const bool split = p.split != 0;
    auto issue = [&](int n) __attribute__((always_inline)) {
        if (n < total_stages) {
            if (split) {
                if ((wave >> 2) == (n & 1)) {
                    const int w4 = wave & 3;
                    const char* src = p.packed + (size_t)(n % K::NSTG) * K::STAGE + (size_t)(2 * K::PPW * w4) * 1024 + lane * 16;
                    const unsigned dst = lds0 + (n % ST_RING) * K::STAGE + (2 * K::PPW * w4) * 1024;
#pragma unroll
                    for (int i = 0; i < 2 * K::PPW; ++i) st_dma(src + i * 1024, dst + i * 1024);
                }
            } else {
                const char* src = p.packed + (size_t)(n % K::NSTG) * K::STAGE + (size_t)(K::PPW * wave) * 1024 + lane * 16;
                const unsigned dst = lds0 + (n % ST_RING) * K::STAGE + (K::PPW * wave) * 1024;
#pragma unroll
                for (int i = 0; i < K::PPW; ++i) st_dma(src + i * 1024, dst + i * 1024);
            }
        }
    };
    issue(0);
    issue(1);
    int n = 0;                                                 // next stage to consume
    // stage n has landed (all but this wave's PPW youngest pieces), this wave's reads of stage n - 1 have RETURNED (the refill of its slot is
    // issued right behind the barrier: tools/check_barrier_reads.py) and every wave is past them: the slot takes stage n + 2
    auto acquire = [&]() __attribute__((always_inline)) -> const char* {
        if (split) {
            if ((wave >> 2) == (n & 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (n + 1 < total_stages) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K::PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        issue(n + 2);
        return st_smem + (n % ST_RING) * K::STAGE + lane * 16;
    };
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int64_t tile = blockIdx.x + (int64_t)ti * gridDim.x;
        const int64_t pt = tile * ST_TILE + wave * 32 + pnt;
        // B fragments of a: lane (point, half) takes the 8 channels 16 s + 8 half of every k step; the residual in accumulator layout
        half8 bf[KS];
        {
            const half_t* row = p.a + pt * C + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) bf[s] = *(const half8*)(row + 16 * s);
        }
        half4 xr[NT][4];
        {
            const half_t* row = p.x + pt * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) xr[t][g] = *(const half4*)(row + 32 * t + 8 * g);
        }
        if (p.pre_e != nullptr) {
            const float* er = p.pre_e + (pt / p.rps) * p.estride + 4 * hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 ev = *(const f32x4*)(er + 32 * t + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[t][g][e] = to_half_sat((float)xr[t][g][e] + ev[e]);
                }
        }
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // ---- out_proj
        {
            const char* img = acquire();
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                half8 af[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) af[t] = *(const half8*)(img + (t * KS + q) * 1024);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[q], acc[t], 0, 0, 0);
            }
            ++n;
        }
        // ---- x1 = x + out_proj(a) + b_out (fp32), its LayerNorm statistics over the point's C channels (this lane's C / 2 + lane ^ 32's)
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bo = *(const f32x4*)&par[K::BO + 32 * t + 8 * g + 4 * hh];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[t][4 * g + e] += bo[e] + (float)xr[t][g][e];
                    sum += acc[t][4 * g + e];
                }
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.f / C);
        float ssq = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float dv = acc[t][e] - mean; ssq += dv * dv; }
        ssq += __shfl_xor(ssq, 32);
        const float rstd = rsqrtf(ssq * (1.f / C) + 1e-5f);
        // ---- LN2(x1) -> B fragments (k step 2 t + gp); then the accumulators take b_ff2 and go on as y
        half8 lnf[KS];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = 2 * gp + u;
                    const f32x4 ga = *(const f32x4*)&par[K::GA + 32 * t + 8 * g + 4 * hh];
                    const f32x4 be = *(const f32x4*)&par[K::BE + 32 * t + 8 * g + 4 * hh];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = __builtin_amdgcn_fmed3f((acc[t][4 * g + e] - mean) * rstd * ga[e] + be[e], -65504.f, 65504.f);
                }
                lnf[2 * t + gp] = st_pack_swap(v[0], v[1]);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b2 = *(const f32x4*)&par[K::B2 + 32 * t + 8 * g + 4 * hh];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * g + e] += b2[e];
            }
        // ---- the FFN, 64 hidden channels at a time
#pragma unroll 1
        for (int ch = 0; ch < K::NCH; ++ch) {
            const char* img = acquire();
            f32x16 h[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) h[u][e] = 0.f;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                half8 af[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) af[u] = *(const half8*)(img + (u * KS + q) * 1024);
#pragma unroll
                for (int u = 0; u < 2; ++u) h[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u], lnf[q], h[u], 0, 0, 0);
            }
            half8 hf[4];
            const float* b1 = par + K::B1 + 64 * ch + 4 * hh;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[2][4];
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const int g = 2 * gp + w;
                        const f32x4 bb = *(const f32x4*)&b1[32 * u + 8 * g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[w][e] = __builtin_amdgcn_fmed3f(h[u][4 * g + e] + bb[e], 0.f, 65504.f);
                    }
                    hf[2 * u + gp] = st_pack_swap(v[0], v[1]);
                }
            const char* img2 = img + K::W2OFF;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                half8 af[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) af[t] = *(const half8*)(img2 + (t * 4 + q) * 1024);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], hf[q], acc[t], 0, 0, 0);
            }
            ++n;
        }
        // ---- y rows leave as 16-byte pieces (8 consecutive channels per lane)
        half_t* orow = p.y + pt * C + 8 * hh;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[2][4];
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[w][e] = __builtin_amdgcn_fmed3f(acc[t][4 * (2 * gp + w) + e], -65504.f, 65504.f);
                if (p.post_e != nullptr) {
                    const float* er = p.post_e + (pt / p.rps) * p.estride + 32 * t + 4 * hh;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const f32x4 ev = *(const f32x4*)(er + 8 * (2 * gp + w));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[w][e] = __builtin_amdgcn_fmed3f((float)(half_t)v[w][e] + ev[e], -65504.f, 65504.f);
                    }
                }
                *(half8*)(orow + 32 * t + 16 * gp) = st_pack_swap(v[0], v[1]);
            }
    }
}

// The head of the block for C <= 128: qkv = in_proj(LN1(x)) (reference networks.py:81) in one launch of the same form -- the B fragments of x are normalised
// as they are loaded (statistics by v_dot2_f32_f16 over the lane's C / 2 channels + lane ^ 32, fp32; result fp16 like pcd_layernorm_f16's), then three
// C-wide output passes over them, one stage image of W_in each, bias, no activation, 16-byte stores into the [rows][3C] qkv tensor.
struct SabHeadParams {
    const half_t* x;          // [M][C] the block's input
    const char* packed;       // the head part of the block's image: 3 stage images, then HNPAR floats
    half_t* qkv;              // [M][3C]
    int64_t m;
    const float* pre_e; int64_t estride; int rps;      // x is read as fp16(x + pre_e[shape]) (see SabTailParams)
    int split;                // as SabTailParams
};

template <int C>
__global__ __launch_bounds__(ST_THREADS, 2) void sab_head_kernel(SabHeadParams p) {
    using K = StCfg<C>;
    constexpr int NT = K::NT, KS = K::KS;
    extern __shared__ __attribute__((aligned(16))) char st_smem[];          // [ST_RING][HSTAGE] | parameters
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pnt = lane & 31, hh = lane >> 5;
    float* par = (float*)(st_smem + ST_RING * K::HSTAGE);
    {
        const float* src = (const float*)(p.packed + (size_t)3 * K::HSTAGE);
        for (int i = threadIdx.x; i < K::HNPAR; i += ST_THREADS) par[i] = src[i];
    }
    const unsigned lds0 = (unsigned)(size_t)st_smem;
    const int64_t ntiles = p.m / ST_TILE;
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total_stages = my_tiles * 3;
    const bool split = p.split != 0;
    auto issue = [&](int n) __attribute__((always_inline)) {
        if (n < total_stages) {
            if (split) {
                if ((wave >> 2) == (n & 1)) {
                    const int w4 = wave & 3;
                    const char* src = p.packed + (size_t)(n % 3) * K::HSTAGE + (size_t)(2 * K::HPPW * w4) * 1024 + lane * 16;
                    const unsigned dst = lds0 + (n % ST_RING) * K::HSTAGE + (2 * K::HPPW * w4) * 1024;
#pragma unroll
                    for (int i = 0; i < 2 * K::HPPW; ++i) st_dma(src + i * 1024, dst + i * 1024);
                }
            } else {
                const char* src = p.packed + (size_t)(n % 3) * K::HSTAGE + (size_t)(K::HPPW * wave) * 1024 + lane * 16;
                const unsigned dst = lds0 + (n % ST_RING) * K::HSTAGE + (K::HPPW * wave) * 1024;
#pragma unroll
                for (int i = 0; i < K::HPPW; ++i) st_dma(src + i * 1024, dst + i * 1024);
            }
        }
    };
    issue(0);
    issue(1);
    __syncthreads();                                           // the LayerNorm affine is read before the first stage barrier
    int n = 0;
    auto acquire = [&]() __attribute__((always_inline)) -> const char* {
        if (split) {
            if ((wave >> 2) == (n & 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (n + 1 < total_stages) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K::HPPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        issue(n + 2);
        return st_smem + (n % ST_RING) * K::HSTAGE + lane * 16;
    };
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int64_t tile = blockIdx.x + (int64_t)ti * gridDim.x;
        const int64_t pt = tile * ST_TILE + wave * 32 + pnt;
        half8 bf[KS];
        {
            const half_t* row = p.x + pt * C + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) bf[s] = *(const half8*)(row + 16 * s);
        }
        if (p.pre_e != nullptr) {
            const float* er = p.pre_e + (pt / p.rps) * p.estride + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 e0 = *(const f32x4*)(er + 16 * s), e1 = *(const f32x4*)(er + 16 * s + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf[s][e] = to_half_sat((float)bf[s][e] + e0[e]);
                    bf[s][4 + e] = to_half_sat((float)bf[s][4 + e] + e1[e]);
                }
            }
        }
        {
            // variance about the mean in a second pass over the packed fp16 pairs (sum(x^2) - C mean^2 cancels when |mean| >> std): subtract
            // mh = fp16(mean) in packed fp16 and remove the shift exactly, sum (x - mh)^2 = sum (x - mean)^2 + C (mean - mh)^2  (widechain.hip)
            float sum = 0.f, sq = 0.f;
            half2_ one2; one2.x = one2.y = (half_t)1.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_ v; v.x = bf[s][2 * e]; v.y = bf[s][2 * e + 1];
                    sum = __builtin_amdgcn_fdot2(v, one2, sum, false);
                }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.f / C);
            const half_t mh = (half_t)__builtin_amdgcn_fmed3f(mean, -65504.f, 65504.f);
            half2_ mh2; mh2.x = mh2.y = mh;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_ v; v.x = bf[s][2 * e]; v.y = bf[s][2 * e + 1];
                    const half2_ d = v - mh2;
                    sq = __builtin_amdgcn_fdot2(d, d, sq, false);
                }
            sq += __shfl_xor(sq, 32);
            const float shift = mean - (float)mh;
            const float rstd = rsqrtf(fmaxf(sq - C * shift * shift, 0.f) * (1.f / C) + 1e-5f);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 g0 = *(const f32x4*)&par[K::HGA + 16 * s + 8 * hh], g1 = *(const f32x4*)&par[K::HGA + 16 * s + 8 * hh + 4];
                const f32x4 b0 = *(const f32x4*)&par[K::HBE + 16 * s + 8 * hh], b1 = *(const f32x4*)&par[K::HBE + 16 * s + 8 * hh + 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf[s][e] = (half_t)__builtin_amdgcn_fmed3f(((float)bf[s][e] - mean) * rstd * g0[e] + b0[e], -65504.f, 65504.f);
                    bf[s][4 + e] = (half_t)__builtin_amdgcn_fmed3f(((float)bf[s][4 + e] - mean) * rstd * g1[e] + b1[e], -65504.f, 65504.f);
                }
            }
        }
        half_t* orow = p.qkv + pt * (3 * C) + 8 * hh;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const char* img = acquire();
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                half8 af[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) af[t] = *(const half8*)(img + (t * KS + q) * 1024);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[q], acc[t], 0, 0, 0);
            }
            ++n;
            const float* bb = par + K::HB + pass * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[2][4];
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const int g = 2 * gp + w;
                        const f32x4 b4 = *(const f32x4*)&bb[32 * t + 8 * g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[w][e] = __builtin_amdgcn_fmed3f(acc[t][4 * g + e] + b4[e], -65504.f, 65504.f);
                    }
                    *(half8*)(orow + pass * C + 32 * t + 16 * gp) = st_pack_swap(v[0], v[1]);
                }
        }
    }
}

// W [.][ldw] fp16 -> fragment-order pieces: piece (t * nq + q) * 64 + lane = the 8 halfs W[c0 + 32 t + (lane & 31)][k0 + 16 q + 8 (lane >> 5) .. + 7]
__global__ __launch_bounds__(256) void st_pack_kernel(const half_t* __restrict__ w, int64_t ldw, int c0, int k0, int ntile, int nq, char* __restrict__ img) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ntile * nq * 64) return;
    const int lane = id & 63, tq = id >> 6, t = tq / nq, q = tq - t * nq;
    const int ch = c0 + 32 * t + (lane & 31);
    *(half8*)(img + (size_t)id * 16) = *(const half8*)(w + (int64_t)ch * ldw + k0 + 16 * q + 8 * (lane >> 5));
}

static int g_sab_split = 1;     // pcd_sab_tail_config + 2 switches it off: one wave per SIMD requests a stage's LDS-DMA pieces
static int g_sab_tail = 1;      // pcd_sab_tail_config: 0 = pcd_sab_forward keeps the four launches even where the descriptor carries a packed tail

template <int C>
static int tail_pack(const pcd_sab_desc_t& d, char* packed, hipStream_t s) {
    using K = StCfg<C>;
    PCD_CHECK_HIP(hipMemsetAsync(packed, 0, (size_t)K::NSTG * K::STAGE, s));
    auto pack = [&](const void* w, int ldw, int c0, int k0, int ntile, int nq, char* img) {
        const int n = ntile * nq * 64;
        hipLaunchKernelGGL(st_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const half_t*)w, (int64_t)ldw, c0, k0, ntile, nq, img);
    };
    pack(d.w_out, C, 0, 0, K::NT, K::KS, packed);
    for (int ch = 0; ch < K::NCH; ++ch) {
        char* img = packed + (size_t)(1 + ch) * K::STAGE;
        pack(d.w_ff1, C, 64 * ch, 0, 2, K::KS, img);
        pack(d.w_ff2, 4 * C, 0, 64 * ch, K::NT, 4, img + K::W2OFF);
    }
    PCD_CHECK_LAUNCH();
    float* par = (float*)(packed + (size_t)K::NSTG * K::STAGE);
    PCD_CHECK_HIP(hipMemcpyAsync(par + K::BO, d.b_out, C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(par + K::GA, d.ln2_g, C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(par + K::BE, d.ln2_b, C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(par + K::B2, d.b_ff2, C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(par + K::B1, d.b_ff1, 4 * C * 4, hipMemcpyDeviceToDevice, s));
    // the head's part: W_in as three passes, b_in, LN1 affine
    char* head = packed + K::TAIL_BYTES;
    for (int i = 0; i < 3; ++i) pack(d.w_in, C, i * C, 0, K::NT, K::KS, head + (size_t)i * K::HSTAGE);
    PCD_CHECK_LAUNCH();
    float* hpar = (float*)(head + (size_t)3 * K::HSTAGE);
    PCD_CHECK_HIP(hipMemcpyAsync(hpar + K::HB, d.b_in, 3 * C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(hpar + K::HGA, d.ln1_g, C * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(hpar + K::HBE, d.ln1_b, C * 4, hipMemcpyDeviceToDevice, s));
    return PCD_OK;
}

template <int C>
static int head_launch(const void* packed, const void* x, int64_t m, void* qkv, const float* pre_e, int64_t estride, int rps, hipStream_t s) {
    using K = StCfg<C>;
    SabHeadParams p{};
    p.x = (const half_t*)x; p.packed = (const char*)packed + K::TAIL_BYTES; p.qkv = (half_t*)qkv; p.m = m;
    p.pre_e = pre_e; p.estride = estride; p.rps = rps > 0 ? rps : 1;
    p.split = g_sab_split;
    static PcdLdsOnce once;
    const size_t lds = (size_t)ST_RING * K::HSTAGE + K::HNPAR * sizeof(float);
    PCD_CHECK_HIP(pcd_allow_lds(once, (const void*)sab_head_kernel<C>, (int)lds));
    const int64_t tiles = m / ST_TILE;
    const int64_t cap = C == 64 ? 768 : 256;                   // C = 64: 74 registers, 26 KB of LDS: three workgroups per CU (HBM-bound: 67 MB per launch at cfg2)
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(sab_head_kernel<C>, dim3(grid), dim3(ST_THREADS), lds, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

template <int C>
static int tail_launch(const void* packed, const void* a, const void* x, int64_t m, void* y, const float* pre_e, const float* post_e, int64_t estride, int rps,
                       hipStream_t s) {
    using K = StCfg<C>;
    SabTailParams p{};
    p.a = (const half_t*)a; p.x = (const half_t*)x; p.packed = (const char*)packed; p.y = (half_t*)y; p.m = m;
    p.pre_e = pre_e; p.post_e = post_e; p.estride = estride; p.rps = rps > 0 ? rps : 1;
    p.split = g_sab_split;
    static PcdLdsOnce once;
    const size_t lds = (size_t)ST_RING * K::STAGE + K::NPAR * sizeof(float);
    PCD_CHECK_HIP(pcd_allow_lds(once, (const void*)sab_tail_kernel<C>, (int)lds));
    const int64_t tiles = m / ST_TILE;
    // C = 64: 128 registers and 50 KB of LDS let two workgroups share a CU, and the kernel is closer to its HBM bound than to the matrix pipe's
    const int64_t cap = C == 64 ? 512 : 256;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(sab_tail_kernel<C>, dim3(grid), dim3(ST_THREADS), lds, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

}  // namespace pcd

using namespace pcd;

extern "C" size_t pcd_sab_tail_packed_bytes(int dim) {
    if (dim == 128) return StCfg<128>::TAIL_BYTES + StCfg<128>::HEAD_BYTES;
    if (dim == 64) return StCfg<64>::TAIL_BYTES + StCfg<64>::HEAD_BYTES;
    return 0;
}

extern "C" int pcd_sab_tail_supported(int dim, int64_t rows) {
    return (dim == 64 || dim == 128) && rows > 0 && rows % ST_TILE == 0 && rows * dim <= 0x7fffffff ? 1 : 0;
}

extern "C" int pcd_sab_tail_config(int fused) {
    PCD_CHECK_ARG(fused >= 0 && fused <= 3);
    g_sab_tail = fused & 1;
    g_sab_split = (fused & 2) ? 0 : 1;
    return PCD_OK;
}

extern "C" int pcd_sab_tail_enabled(void) { return g_sab_tail; }

extern "C" int pcd_sab_tail_pack(const pcd_sab_desc_t* d, void* packed, void* stream) {
    PCD_CHECK_ARG(d && packed && (d->dim == 64 || d->dim == 128));
    PCD_CHECK_ARG(d->w_out && d->b_out && d->ln2_g && d->ln2_b && d->w_ff1 && d->b_ff1 && d->w_ff2 && d->b_ff2);
    PCD_CHECK_ARG(d->w_in && d->b_in && d->ln1_g && d->ln1_b);
    return d->dim == 128 ? tail_pack<128>(*d, (char*)packed, (hipStream_t)stream) : tail_pack<64>(*d, (char*)packed, (hipStream_t)stream);
}

extern "C" int pcd_sab_tail_bias_f16(int dim, const void* packed, const void* a, const void* x, int64_t rows, int rows_per_shape, const float* pre_e,
                                     const float* post_e, int64_t e_stride, void* y, void* stream) {
    PCD_CHECK_ARG(packed && a && x && y && y != a && y != x);
    PCD_CHECK_ARG(pcd_sab_tail_supported(dim, rows));
    PCD_CHECK_ARG((pre_e == nullptr && post_e == nullptr) || (rows_per_shape > 0 && e_stride >= 0 && e_stride % 4 == 0));
    PCD_CHECK_ARG((((uintptr_t)pre_e | (uintptr_t)post_e) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    return dim == 128 ? tail_launch<128>(packed, a, x, rows, y, pre_e, post_e, e_stride, rows_per_shape, s)
                      : tail_launch<64>(packed, a, x, rows, y, pre_e, post_e, e_stride, rows_per_shape, s);
}

extern "C" int pcd_sab_tail_f16(int dim, const void* packed, const void* a, const void* x, int64_t rows, void* y, void* stream) {
    return pcd_sab_tail_bias_f16(dim, packed, a, x, rows, 0, nullptr, nullptr, 0, y, stream);
}

extern "C" int pcd_sab_head_bias_f16(int dim, const void* packed, const void* x, int64_t rows, int rows_per_shape, const float* pre_e, int64_t e_stride,
                                     void* qkv, void* stream) {
    PCD_CHECK_ARG(packed && x && qkv && qkv != x);
    PCD_CHECK_ARG(pcd_sab_tail_supported(dim, rows));
    PCD_CHECK_ARG(pre_e == nullptr || (rows_per_shape > 0 && e_stride >= 0 && e_stride % 4 == 0 && ((uintptr_t)pre_e & 15) == 0));
    hipStream_t s = (hipStream_t)stream;
    return dim == 128 ? head_launch<128>(packed, x, rows, qkv, pre_e, e_stride, rows_per_shape, s)
                      : head_launch<64>(packed, x, rows, qkv, pre_e, e_stride, rows_per_shape, s);
}

extern "C" int pcd_sab_head_f16(int dim, const void* packed, const void* x, int64_t rows, void* qkv, void* stream) {
    return pcd_sab_head_bias_f16(dim, packed, x, rows, 0, nullptr, 0, qkv, stream);
}
