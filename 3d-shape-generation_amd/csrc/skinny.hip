// K8: weight-streaming GEMM for few rows (M <= 256: the latent denoiser's (B, C) vectors).
//
// The dense tile GEMM leaves most of the chip idle here (C/128 blocks, each streaming a long K);
// this kernel is shaped for HBM bandwidth instead: a block owns 32 output columns and a K slice of
// KS, so a layer is (C/32) x (K/KS) blocks that each stream 32 x KS x 2 bytes of weights exactly once.
// No LDS staging: every lane loads 64 contiguous bytes of its weight row per 64-deep chunk (two
// lanes = one 128-B line) and the same 64 bytes of its activation row; the four 16-k MFMA steps of a
// chunk use a permuted k order that is identical for both operands, so no shuffles are needed.
// The four waves split the block's K slice and reduce through LDS; split-K partial sums go to fp32
// slabs [S][M][C] that the finishing kernel (bias + GroupNorm + ReLU) adds in a fixed order, so the
// result is bitwise reproducible (no float atomics).
#include "common.h"

namespace pcd {

struct SkinnyParams {
    const half_t* a1; int k1;
    const half_t* a2; int k2;
    const half_t* w; int ldw;
    int m, c, ks;
    float* slabs;          // [S][M][C]
};

template <int MT>   // number of 32-row tiles (M <= 32*MT)
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyParams p) {
    __shared__ float red[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int slice = blockIdx.y;
    const int kbeg = slice * p.ks;
    const int chunks = p.ks / 64;                    // 64-deep chunks in this block's slice
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    int wn = n0 + r;
    wn = wn < p.c ? wn : p.c - 1;
    const half_t* wrow = p.w + (int64_t)wn * p.ldw;
    for (int ch = wave; ch < chunks; ch += 4) {
        const int k = kbeg + ch * 64 + 32 * hh;      // this lane's 32 consecutive k
        half8 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const half8*)(wrow + k + 8 * j);
        const half_t* src = k < p.k1 ? p.a1 : p.a2;
        const int lda = k < p.k1 ? p.k1 : p.k2;
        const int ka = k < p.k1 ? k : k - p.k1;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            int row = t * 32 + r;
            row = row < p.m ? row : p.m - 1;
            const half_t* arow = src + (int64_t)row * lda + ka;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half8 af = *(const half8*)(arow + 8 * j);
                // D[row = x-row][col = weight-row]: A = activations, B = weights
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf[j], acc[t], 0, 0, 0);
            }
        }
    }
    // cross-wave reduction, one 32x32 tile at a time; accumulator register e of lane (r, hh) is
    // row (e&3) + 8*(e>>2) + 4*hh (x row), column r (weight row)
    float* out = p.slabs + (int64_t)slice * p.m * p.c;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hh][r] = acc[t][e];
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 32; i += 256) {
            const int row = i >> 5, col = i & 31;
            const float v = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
            const int gr = t * 32 + row, gc = n0 + col;
            if (gr < p.m && gc < p.c) out[(int64_t)gr * p.c + gc] = v;
        }
    }
}

// M <= 32 form with the operands staged through LDS by LDS-DMA.  The register-fragment loads of the kernel above are
// "fragment shaped" (each load instruction touches 32 rows x 2 pieces of 16 bytes): a workgroup then streams its weights
// at only ~10-13 GB/s (measured: 32 KB per workgroup in ~5 us, the same rate in every skinny kernel).  Here every
// 64-deep chunk of the weight slice and of the activation slice is one [32 rows][128 B] image filled by four 1-KiB
// LDS-DMA instructions in full 128-byte lines (8 rows each), ALL chunks of the slice requested at kernel entry, with the
// XOR swizzle of the attention K tile on the source address so that the ds_read_b128 fragment reads are conflict free.
// Same products, same summation order as skinny_gemm_kernel<1>: bitwise identical slabs.
__device__ __forceinline__ int sk_swz(int row, int ch) { return ch ^ ((row >> 1) & 7); }

// stage rows [0, 32) x 128 bytes at byte column `kbyte` of a row-major fp16 matrix (row stride `ld` halfs, rows clamped
// to `row_limit - 1`) into the 4-KiB image at `img`: instruction i of 4 covers rows 8 i .. 8 i + 7
__device__ __forceinline__ void sk_stage_chunk(const half_t* base, int64_t ld, int row0, int row_limit, int k0, char* img,
                                               int i, int lane) {
    const int row = 8 * i + (lane >> 3), ch = lane & 7;
    int gr = row0 + row;
    gr = gr < row_limit ? gr : row_limit - 1;
    lds_dma16(base + (int64_t)gr * ld + k0 + sk_swz(row, ch) * 8, img + i * 1024);
}

__device__ __forceinline__ half8 sk_frag(const char* img, int row, int hh, int j) {
    return *(const half8*)(img + row * 128 + (sk_swz(row, 4 * hh + j) << 4));
}

__global__ __launch_bounds__(256) void skinny_gemm_dma_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];     // [chunks][W image 4 KiB | A image 4 KiB]; reused for the reduction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int slice = blockIdx.y;
    const int kbeg = slice * p.ks;
    const int chunks = p.ks / 64;
    // every chunk of the slice goes out now: 8 LDS-DMA instructions per chunk, dealt round-robin to the four waves
    for (int t = wave; t < chunks * 8; t += 4) {
        const int ch = t >> 3, i = t & 7;
        const int k = kbeg + ch * 64;
        char* img = sk_smem + ch * 8192;
        if (i < 4) {
            sk_stage_chunk(p.w, p.ldw, n0, p.c, k, img, i, lane);
        } else {
            const half_t* src = k < p.k1 ? p.a1 : p.a2;
            const int lda = k < p.k1 ? p.k1 : p.k2;
            sk_stage_chunk(src, lda, 0, p.m, k < p.k1 ? k : k - p.k1, img + 4096, i - 4, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int ch = wave; ch < chunks; ch += 4) {
        const char* img = sk_smem + ch * 8192;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sk_frag(img + 4096, r, hh, j), sk_frag(img, r, hh, j), acc, 0, 0, 0);
    }
    __syncthreads();                                    // the images are dead: the reduction reuses their space
    float (*red)[32][33] = (float (*)[32][33])sk_smem;
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hh][r] = acc[e];
    __syncthreads();
    float* out = p.slabs + (int64_t)slice * p.m * p.c;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int row = i >> 5, col = i & 31;
        const float v = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
        const int gc = n0 + col;
        if (row < p.m && gc < p.c) out[(int64_t)row * p.c + gc] = v;
    }
}

// out = act( sum_s slabs[s] + bias [+ per-row bias] ), act: 0 = GroupNorm(groups)+affine+ReLU -> fp16,
// 1 = ReLU -> fp16, 2 = identity -> fp32.  One block per (row, channel group): GroupNorm statistics
// are per (row, group), so the groups of a row are independent (modes 1/2 just use the same split).
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float* __restrict__ slabs, int nslabs, int m, int c,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ row_bias, int mode, int groups,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, half_t* __restrict__ out16,
                                                             float* __restrict__ out32) {
    extern __shared__ float buf[];   // [c / groups]
    __shared__ float red[8];
    const int row = blockIdx.x, g = blockIdx.y;
    const int gsz = c / groups, c0 = g * gsz;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // GroupNorm affine of this thread's (first two) channels requested up front: their latency then overlaps the
    // slab sums and the two block reductions instead of following them
    float gpre[2] = {1.f, 1.f}, bpre[2] = {0.f, 0.f};
    if (mode == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * (int)blockDim.x;
            if (i < gsz) { gpre[j] = gamma[c0 + i]; bpre[j] = beta[c0 + i]; }
        }
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < gsz; i += blockDim.x) {
        const int ch = c0 + i;
        float v = bias != nullptr ? bias[ch] : 0.f;
        if (row_bias != nullptr) v += row_bias[(int64_t)row * c + ch];
        // slabs added in slab order; the loads of four slabs are in flight together (a one-by-one loop pays a
        // memory latency per slab: 10 slabs = most of this kernel's 6 us)
        const float* sp = slabs + (int64_t)row * c + ch;
        const int64_t sstride = (int64_t)m * c;
        for (int sl = 0; sl < nslabs; sl += 4) {
            const float a0 = sp[(int64_t)sl * sstride];
            const float a1 = sl + 1 < nslabs ? sp[(int64_t)(sl + 1) * sstride] : 0.f;
            const float a2 = sl + 2 < nslabs ? sp[(int64_t)(sl + 2) * sstride] : 0.f;
            const float a3 = sl + 3 < nslabs ? sp[(int64_t)(sl + 3) * sstride] : 0.f;
            v += a0;
            if (sl + 1 < nslabs) v += a1;
            if (sl + 2 < nslabs) v += a2;
            if (sl + 3 < nslabs) v += a3;
        }
        buf[i] = v;
        s += v;
    }
    if (mode == 2) {
        for (int i = threadIdx.x; i < gsz; i += blockDim.x) out32[(int64_t)row * c + c0 + i] = buf[i];
        return;
    }
    if (mode == 1) {
        for (int i = threadIdx.x; i < gsz; i += blockDim.x)
            out16[(int64_t)row * c + c0 + i] = to_half_sat(fmaxf(buf[i], 0.f));
        return;
    }
    // two-pass mean / biased variance over the group (fixed reduction order: deterministic)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)gsz;
    float v = 0.f;
    for (int i = threadIdx.x; i < gsz; i += blockDim.x) { const float d = buf[i] - mean; v += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[4 + wave] = v;
    __syncthreads();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)gsz + 1e-5f);
    for (int i = threadIdx.x, j = 0; i < gsz; i += blockDim.x, ++j) {
        const int ch = c0 + i;
        const float ga = j < 2 ? gpre[j] : gamma[ch], be = j < 2 ? bpre[j] : beta[ch];
        out16[(int64_t)row * c + ch] = to_half_sat(fmaxf((buf[i] - mean) * rstd * ga + be, 0.f));
    }
}

// Small layers in ONE launch: Linear (+bias, + per-row bias) + GroupNorm + ReLU for weights of <= ~1.5 MB.  The
// split-K + finish pair above pays two dependent launches (~3.7 us each inside a HIP graph), which is most of such
// a layer's time in the latent denoiser step.  A workgroup owns BC = 32 * CT consecutive columns -- whole
// GroupNorm groups -- and 32 rows over the FULL K; its 16 waves are CT column tiles x (16 / CT) K splits that are
// summed through LDS in a fixed order; GroupNorm statistics are per (row, group), so nothing crosses workgroups.
struct SkinnyFusedParams {
    const half_t* a1; int k1;
    const half_t* a2; int k2;
    const half_t* w; int ldw;
    int m, c;
    const float* bias; const float* row_bias;
    int mode, gsz;                                   // 0: GroupNorm(gsz channels)+ReLU -> fp16; 1: ReLU -> fp16; 2: fp32
    const float* gamma; const float* beta;
    half_t* out16; float* out32;
    const float* a1_f32;                             // if set: the first source in fp32 (rounded to fp16 on load)
};

template <int CT>
__global__ __launch_bounds__(1024) void skinny_fused_kernel(SkinnyFusedParams p) {
    // 16 waves = CT column tiles x KSPLIT K splits: every wave has at most a few 64-deep chunks, all of whose loads
    // are in flight together, so the layer costs about one memory latency
    constexpr int BC = 32 * CT, KSPLIT = 16 / CT;
    __shared__ float zt[KSPLIT][32][BC + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ct = wave % CT, ksi = wave / CT;
    const int n0 = blockIdx.x * BC + ct * 32;
    const int row0 = blockIdx.y * 32;
    const int chunks = (p.k1 + p.k2) / 64;
    // the finish phase's per-column constants are requested now, so their latency overlaps the K loop's
    const int frow = threadIdx.x >> 5, part = threadIdx.x & 31;
    const int grow = row0 + frow;
    const int c0 = blockIdx.x * BC + part * CT;
    float cb[CT], cg[CT], cbt[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ch = c0 + i < p.c ? c0 + i : p.c - 1;
        float x = p.bias != nullptr ? p.bias[ch] : 0.f;
        if (p.row_bias != nullptr) x += p.row_bias[(int64_t)(grow < p.m ? grow : p.m - 1) * p.c + ch];
        cb[i] = x;
        cg[i] = p.mode == 0 ? p.gamma[ch] : 1.f;
        cbt[i] = p.mode == 0 ? p.beta[ch] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    int wn = n0 + r;
    wn = wn < p.c ? wn : p.c - 1;
    const half_t* wrow = p.w + (int64_t)wn * p.ldw;
    int row = row0 + r;
    row = row < p.m ? row : p.m - 1;
#pragma unroll 3
    for (int ch = ksi; ch < chunks; ch += KSPLIT) {
        const int k = ch * 64 + 32 * hh;             // this lane's 32 consecutive k (same permuted order for both)
        const half_t* src = k < p.k1 ? p.a1 : p.a2;
        const int lda = k < p.k1 ? p.k1 : p.k2;
        const int ka = k < p.k1 ? k : k - p.k1;
        const half_t* arow = src + (int64_t)row * lda + ka;
        half8 wf[4], af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const half8*)(wrow + k + 8 * j);
        if (p.a1_f32 != nullptr && k < p.k1) {
            // same conversion as the separate fp32 -> fp16 pass this replaces (saturating, round to nearest even)
            const float* frow = p.a1_f32 + (int64_t)row * p.k1 + k;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 lo = *(const float4*)(frow + 8 * j), hi = *(const float4*)(frow + 8 * j + 4);
                af[j] = (half8){to_half_sat(lo.x), to_half_sat(lo.y), to_half_sat(lo.z), to_half_sat(lo.w),
                                to_half_sat(hi.x), to_half_sat(hi.y), to_half_sat(hi.z), to_half_sat(hi.w)};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *(const half8*)(arow + 8 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], wf[j], acc, 0, 0, 0);
    }
    // accumulator e of lane (r, hh) is row (e&3) + 8*(e>>2) + 4*hh, column r; one LDS slab per K split
#pragma unroll
    for (int e = 0; e < 16; ++e) zt[ksi][(e & 3) + 8 * (e >> 2) + 4 * hh][ct * 32 + r] = acc[e];
    __syncthreads();
    // finish: 32 threads per row, CT consecutive columns each, K splits added in split order (deterministic);
    // a GroupNorm group is gsz / CT neighbouring threads of the row
    float v[CT];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        float x = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < KSPLIT; ++sidx) x += zt[sidx][frow][part * CT + i];
        x += cb[i];
        v[i] = x;
        s1 += x;
    }
    if (p.mode == 0) {
        const int tpg = p.gsz / CT;                  // threads per group: 16 (gsz 16) or 32
        for (int o = 1; o < tpg; o <<= 1) s1 += __shfl_xor(s1, o);
        const float mean = s1 / (float)p.gsz;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CT; ++i) { const float dlt = v[i] - mean; s2 += dlt * dlt; }
        for (int o = 1; o < tpg; o <<= 1) s2 += __shfl_xor(s2, o);
        const float rstd = rsqrtf(s2 / (float)p.gsz + 1e-5f);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ch = c0 + i;
            if (ch < p.c) v[i] = fmaxf((v[i] - mean) * rstd * cg[i] + cbt[i], 0.f);
        }
    } else if (p.mode == 1) {
#pragma unroll
        for (int i = 0; i < CT; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (grow < p.m) {
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ch = c0 + i;
            if (ch < p.c) {
                if (p.mode == 2) p.out32[(int64_t)grow * p.c + ch] = v[i];
                else p.out16[(int64_t)grow * p.c + ch] = to_half_sat(v[i]);
            }
        }
    }
}

}  // namespace pcd

using namespace pcd;

static int g_skinny_dma = 1;          // tuning / testing hook (pcd_skinny_config): 0 = register-fragment loads everywhere

extern "C" int pcd_skinny_config(int use_lds_dma) {
    g_skinny_dma = use_lds_dma ? 1 : 0;
    return PCD_OK;
}

// K slice per block: aim for >= ~256 blocks per layer, slices of at least 256 (4 waves x one 64-chunk)
static int pick_ks(int k, int c) {
    int ks = 512;
    while (ks > 256 && (int64_t)(c / 32) * (k / ks) < 256) ks >>= 1;
    while (k % ks != 0) ks >>= 1;      // k is a multiple of 64
    return ks;
}

extern "C" int pcd_skinny_slabs(int k, int c) {
    if (k <= 0 || c <= 0 || k % 64 != 0) return 0;
    return k / pick_ks(k, c);
}

extern "C" int pcd_skinny_gemm_f16(const void* a1, int k1, const void* a2, int k2, const void* w, int64_t ldw, int m,
                                   int c, float* slabs, void* stream) {
    PCD_CHECK_ARG(a1 && w && slabs && m > 0 && m <= 256 && c > 0);
    PCD_CHECK_ARG(k1 > 0 && k1 % 64 == 0 && k2 >= 0 && k2 % 64 == 0 && (k2 == 0 || a2 != nullptr));
    PCD_CHECK_ARG(ldw >= k1 + k2 && ldw % 8 == 0);
    SkinnyParams p{};
    p.a1 = (const half_t*)a1; p.k1 = k1; p.a2 = (const half_t*)a2; p.k2 = k2;
    p.w = (const half_t*)w; p.ldw = (int)ldw; p.m = m; p.c = c;
    p.ks = pick_ks(k1 + k2, c);
    // a 64-chunk must not straddle the two sources: k1 is a multiple of 64 (checked above)
    p.slabs = slabs;
    dim3 grid((unsigned)ceil_div(c, 32), (unsigned)((k1 + k2) / p.ks));
    hipStream_t s = (hipStream_t)stream;
    const int mt = (int)ceil_div(m, 32);
    const size_t dma_lds = (size_t)(p.ks / 64) * 8192;                 // >= the 4 x 32 x 33 floats of the reduction
    if (mt <= 1 && g_skinny_dma && dma_lds >= sizeof(float) * 4 * 32 * 33 && dma_lds <= 65536)
        hipLaunchKernelGGL(skinny_gemm_dma_kernel, grid, dim3(256), dma_lds, s, p);
    else if (mt <= 1) hipLaunchKernelGGL((skinny_gemm_kernel<1>), grid, dim3(256), 0, s, p);
    else if (mt <= 2) hipLaunchKernelGGL((skinny_gemm_kernel<2>), grid, dim3(256), 0, s, p);
    else if (mt <= 4) hipLaunchKernelGGL((skinny_gemm_kernel<4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((skinny_gemm_kernel<8>), grid, dim3(256), 0, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_skinny_finish(const float* slabs, int nslabs, int m, int c, const float* bias,
                                 const float* row_bias, int mode, int groups, const float* gamma, const float* beta,
                                 void* out16, float* out32, void* stream) {
    PCD_CHECK_ARG(slabs && nslabs > 0 && m > 0 && c > 0 && mode >= 0 && mode <= 2);
    PCD_CHECK_ARG(mode == 2 ? out32 != nullptr : out16 != nullptr);
    PCD_CHECK_ARG(mode != 0 || (groups > 0 && c % groups == 0 && gamma && beta));
    const int split = mode == 0 ? groups : (c % 8 == 0 ? 8 : 1);
    hipLaunchKernelGGL(skinny_finish_kernel, dim3(m, split), dim3(256), (size_t)(c / split) * sizeof(float),
                       (hipStream_t)stream, slabs, nslabs, m, c, bias, row_bias, mode, split, gamma, beta,
                       (half_t*)out16, out32);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// One-launch Linear (+bias) + GroupNorm(groups) + ReLU (mode 0) / ReLU (1) / fp32 identity (2) for small layers:
// whole GroupNorm groups per workgroup, full K.  Requires a group size of 16, 32, 64 or 128 channels for mode 0.
extern "C" int pcd_skinny_fused_supported(int k, int c, int mode, int groups) {
    if (k <= 0 || c <= 0 || k % 64 != 0 || mode < 0 || mode > 2) return 0;
    if ((int64_t)k * c > 768 * 256) return 0;   // above ~400 KB of weights the split-K pair streams faster (8 CUs here)
    if (mode == 0) {
        if (groups <= 0 || c % groups != 0) return 0;
        const int gsz = c / groups;
        return gsz == 16 || gsz == 32 || gsz == 64 || gsz == 128;
    }
    return c % 32 == 0;
}

static int skinny_fused_launch(const void* a1, const float* a1_f32, int k1, const void* a2, int k2, const void* w,
                               int64_t ldw, int m, int c, const float* bias, const float* row_bias, int mode, int groups,
                               const float* gamma, const float* beta, void* out16, float* out32, void* stream) {
    PCD_CHECK_ARG((a1 || a1_f32) && w && m > 0 && m <= 256 && c > 0);
    PCD_CHECK_ARG(k1 > 0 && k1 % 64 == 0 && k2 >= 0 && k2 % 64 == 0 && (k2 == 0 || a2 != nullptr));
    PCD_CHECK_ARG(ldw >= k1 + k2 && ldw % 8 == 0);
    PCD_CHECK_ARG(pcd_skinny_fused_supported(k1 + k2, c, mode, groups));
    PCD_CHECK_ARG(mode == 2 ? out32 != nullptr : out16 != nullptr);
    PCD_CHECK_ARG(mode != 0 || (gamma && beta));
    SkinnyFusedParams p{};
    p.a1 = (const half_t*)a1; p.k1 = k1; p.a2 = (const half_t*)a2; p.k2 = k2;
    p.w = (const half_t*)w; p.ldw = (int)ldw; p.m = m; p.c = c;
    p.bias = bias; p.row_bias = row_bias; p.mode = mode; p.gsz = mode == 0 ? c / groups : 0;
    p.gamma = gamma; p.beta = beta; p.out16 = (half_t*)out16; p.out32 = out32;
    p.a1_f32 = a1_f32;
    if (a1_f32 != nullptr) p.a1 = (const half_t*)a1_f32;             // never dereferenced as fp16 (k < k1 takes the fp32 branch)
    const int bc = mode == 0 ? (p.gsz < 32 ? 32 : p.gsz) : 32;       // whole groups per workgroup
    const dim3 grid((unsigned)ceil_div(c, bc), (unsigned)ceil_div(m, 32));
    hipStream_t s = (hipStream_t)stream;
    if (bc == 32) hipLaunchKernelGGL((skinny_fused_kernel<1>), grid, dim3(1024), 0, s, p);
    else if (bc == 64) hipLaunchKernelGGL((skinny_fused_kernel<2>), grid, dim3(1024), 0, s, p);
    else hipLaunchKernelGGL((skinny_fused_kernel<4>), grid, dim3(1024), 0, s, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_skinny_fused(const void* a1, int k1, const void* a2, int k2, const void* w, int64_t ldw, int m, int c,
                                const float* bias, const float* row_bias, int mode, int groups, const float* gamma,
                                const float* beta, void* out16, float* out32, void* stream) {
    PCD_CHECK_ARG(a1 != nullptr);
    return skinny_fused_launch(a1, nullptr, k1, a2, k2, w, ldw, m, c, bias, row_bias, mode, groups, gamma, beta, out16,
                               out32, stream);
}

// the same with an fp32 activation matrix [m][k] (rounded to fp16 on load: the latent state z)
extern "C" int pcd_skinny_fused_f32in(const float* a, int k, const void* w, int64_t ldw, int m, int c, const float* bias,
                                      const float* row_bias, int mode, int groups, const float* gamma, const float* beta,
                                      void* out16, float* out32, void* stream) {
    PCD_CHECK_ARG(a != nullptr);
    return skinny_fused_launch(nullptr, a, k, nullptr, 0, w, ldw, m, c, bias, row_bias, mode, groups, gamma, beta, out16,
                               out32, stream);
}
