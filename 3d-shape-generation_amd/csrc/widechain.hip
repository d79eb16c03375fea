// Chains of 256-channel pointwise layers of UNetPointNetLarge (reference networks.py:16-49, 779-818) as one launch each:
//   E3: x2 [M][256] -> enc3.conv1 256->256 -> conv2 256->256 -> conv3 256->512 -> x3
//   D2: [dec3 out [M][256] | x2 [M][256]] -> dec2.conv1 512->256 -> conv2 256->256 -> conv3 256->128
// As separate GEMMs these layers are HBM / latency bound (K = 256: four K tiles per output tile; 33-57 us per layer against a
// 22 us traffic floor) and write + re-read a 67 MB intermediate each.  Here a wave owns 32 points for the whole chain and the
// activations never leave its REGISTERS between layers:
//   * every layer is the transposed product D[channel][point] = W[channel][k] . act[point][k] on v_mfma_f32_32x32x16_f16 (weights
//     = A operand, activations = B operand); the B fragments of all 16 k-steps of a 256-wide input are 64 registers per lane;
//   * an accumulator group holds 4 consecutive channels of one point; after bias + ReLU + fp16 rounding, one v_permlane32_swap per
//     register turns two such groups of the two lane halves into the 8 consecutive k of the NEXT layer's B fragment: no LDS, no
//     barrier, no memory traffic between layers;
//   * only the weights stream: packed once into 32-KB stage images ([8 channel tiles][4 k-steps][64 lanes][16 B] = 256 channels x
//     64 k in fragment order), they arrive by LDS-DMA into a 3-deep ring, one barrier per 64-deep K tile for the 8 waves
//     (256 points) of a workgroup; a 256 x 256 tile of work per 32 KB of LDS fill = 256 FLOP per filled byte, twice the GEMM's.
// Output rows leave as 16-byte pieces (8 consecutive channels per lane after the same swap).  M must be a multiple of 256.
#include "common.h"

namespace pcd {

constexpr int WC_WAVES = 8, WC_THREADS = 64 * WC_WAVES, WC_TILE = 32 * WC_WAVES;
constexpr int WC_STAGE = 32768, WC_RING = 3;
constexpr int WC_MAXSEG = 8;

// one 256-channel output pass: 4 K tiles of the current B fragments (accumulating onto the previous segment when `cont`)
struct WcSeg {
    int src;          // 0: the registers hold the previous layer's output; 1 / 2: load the B fragments from in1 / in2
    int cont;         // 1: keep accumulating (second K half of a two-source layer), 0: start from zero
    int finish;       // 0: more K to come; 1: epilogue -> next layer's fragments (registers); 2: epilogue -> global at channel offset `coff`
    int coff;         // finish 2: first output channel of this pass
    int cvalid;       // finish 2: channels of this pass that exist (256, or 128 for a padded last layer)
    int bias_off;     // offset into the bias array
    int keep_b;       // 1: the B fragments are needed by the next segment too (second output pass of a 512-channel layer)
    int linear;       // 1: no ReLU in this pass's epilogue (a plain Linear: attention in_proj); 0: bias + ReLU
};

struct WcParams {
    const half_t* in1; const half_t* in2;     // [M][256]
    const char* wpacked;                       // stage images, segment after segment, 4 per segment
    const float* bias;                         // fp32, indexed by bias_off + channel
    const float* ln;                           // optional LayerNorm over the 256 input channels of in1 (gamma [256] | beta [256], eps 1e-5) applied to the B
                                               // fragments as they are loaded: LN + Linear(256, 256 P) [+ ReLU] in one launch (pcd_pw_wide_ln_linear)
    half_t* out; int ldo;                      // [M][ldo]
    int64_t m;
    int nseg;
    WcSeg seg[WC_MAXSEG];
};

__device__ __forceinline__ void wc_dma(const char* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// two accumulator groups (same channel tile, g = 2 gp and 2 gp + 1) -> the 8 consecutive channels a lane needs as the next B fragment /
// as one 16-byte output piece: lane half 0 ends with channels 16 s .. + 7, lane half 1 with 16 s + 8 .. + 15 (s = 2 t + gp)
__device__ __forceinline__ half8 wc_regroup(const f32x16& acc, int gp, const float* bias_t, int hh, float lo = 0.f) {
    // group g holds channels 8 g + 4 hh + e of the 32-channel tile
    unsigned p[2], q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g0 = 2 * gp, g1 = 2 * gp + 1;
        const float a0 = acc[4 * g0 + 2 * h] + bias_t[8 * g0 + 4 * hh + 2 * h], a1 = acc[4 * g0 + 2 * h + 1] + bias_t[8 * g0 + 4 * hh + 2 * h + 1];
        const float b0 = acc[4 * g1 + 2 * h] + bias_t[8 * g1 + 4 * hh + 2 * h], b1 = acc[4 * g1 + 2 * h + 1] + bias_t[8 * g1 + 4 * hh + 2 * h + 1];
        half2_ pa, pb;
        pa.x = (half_t)__builtin_amdgcn_fmed3f(a0, lo, 65504.f); pa.y = (half_t)__builtin_amdgcn_fmed3f(a1, lo, 65504.f);
        pb.x = (half_t)__builtin_amdgcn_fmed3f(b0, lo, 65504.f); pb.y = (half_t)__builtin_amdgcn_fmed3f(b1, lo, 65504.f);
        p[h] = __builtin_bit_cast(unsigned, pa);
        q[h] = __builtin_bit_cast(unsigned, pb);
    }
    // registers P = (half 0: X0, half 1: X1), Q = (half 0: Y0, half 1: Y1)  ->  (X0, Y0) and (X1, Y1): swap P's upper lanes with Q's lower lanes
    unsigned f[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const auto r = __builtin_amdgcn_permlane32_swap(p[h], q[h], false, false);
        f[h] = r[0];            // half 0: X0[h], half 1: Y0[h]
        f[2 + h] = r[1];        // half 0: X1[h], half 1: Y1[h]
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, (u4){f[0], f[1], f[2], f[3]});
}

// LN = false: the chains of UNetPointNetLarge; LN = true: LayerNorm + Linear (pcd_pw_wide_ln_linear) -- its own instantiation so that the chains keep their
// register allocation (246 registers, no spill)
// SPLIT: an image's 32 LDS-DMA pieces are requested by ONE wave of each SIMD (waves 0-3 for even images of the workgroup's run, 4-7 for odd ones, 8 pieces each) instead of
// 4 pieces by every wave, so that the requesting wave's SIMD partner issues MFMAs meanwhile (round 5, as csrc/wideffn.hip)
template <bool LN, bool SPLIT>
__global__ __launch_bounds__(WC_THREADS, 2) void pw_wide_chain_kernel(WcParams p) {
    extern __shared__ __attribute__((aligned(16))) char wc_smem[];          // [WC_RING][WC_STAGE] | bias copy
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pnt = lane & 31, hh = lane >> 5;
    float* bias_lds = (float*)(wc_smem + WC_RING * WC_STAGE);
    const int nbias = p.nseg * 256;
    for (int i = threadIdx.x; i < nbias; i += WC_THREADS) bias_lds[i] = p.bias[i];
    float* ln_lds = bias_lds + WC_MAXSEG * 256;                // (present when p.ln: the host sizes the allocation)
    if constexpr (LN)
        for (int i = threadIdx.x; i < 512; i += WC_THREADS) ln_lds[i] = p.ln[i];
    const unsigned lds0 = (unsigned)(size_t)wc_smem;
    const int64_t ntiles = p.m / WC_TILE;
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int nstage_seq = p.nseg * 4;                         // stages per tile of points
    const int total_stages = my_tiles * nstage_seq;
    // stage n of this workgroup's run = image (n % nstage_seq); wave w moves pieces 4 w .. 4 w + 3 of its 32
    auto issue = [&](int n) __attribute__((always_inline)) {
        if (n < total_stages) {
            if constexpr (SPLIT) {
                if ((wave >> 2) == (n & 1)) {
                    const int w4 = wave & 3;
                    const char* src = p.wpacked + (size_t)(n % nstage_seq) * WC_STAGE + (size_t)(8 * w4) * 1024 + lane * 16;
                    const unsigned dst = lds0 + (n % WC_RING) * WC_STAGE + (8 * w4) * 1024;
#pragma unroll
                    for (int i = 0; i < 8; ++i) wc_dma(src + i * 1024, dst + i * 1024);
                }
            } else {
                const char* src = p.wpacked + (size_t)(n % nstage_seq) * WC_STAGE + (size_t)(4 * wave) * 1024 + lane * 16;
                const unsigned dst = lds0 + (n % WC_RING) * WC_STAGE + (4 * wave) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) wc_dma(src + i * 1024, dst + i * 1024);
            }
        }
    };
    issue(0);
    issue(1);
    if constexpr (LN) __syncthreads();                         // ln_lds is read before the first stage barrier
    int n = 0;                                                 // next stage to consume
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int64_t tile = blockIdx.x + (int64_t)ti * gridDim.x;
        const int64_t pt = tile * WC_TILE + wave * 32 + pnt;
        half8 bf[16];
        f32x16 acc[8];
#pragma unroll 1
        for (int sg = 0; sg < p.nseg; ++sg) {
            const WcSeg S = p.seg[sg];
            if (S.src != 0) {
                // B fragments straight from the [point][256] rows: lane (point, half) takes the 8 channels 16 s + 8 half of every k-step
                const half_t* row = (S.src == 1 ? p.in1 : p.in2) + pt * 256 + 8 * hh;
#pragma unroll
                for (int s = 0; s < 16; ++s) bf[s] = *(const half8*)(row + 16 * s);
                if constexpr (LN) {
                    // LayerNorm of the point's 256 channels (this lane holds 128 of them, lane ^ 32 the others): fp32 statistics, result in fp16
                    // like pcd_layernorm_f16's.  (The first barrier below orders these reads of ln_lds behind its fill; at the first tile of a
                    // workgroup the fill is ordered by the __syncthreads() in front of the tile loop.)
                    // Statistics straight from the packed fp16 pairs by v_dot2_f32_f16 with fp32 accumulation (a pass over converted values would keep
                    // 128 more registers alive and spill).  Two passes, the variance about the mean: sum(x^2) - 256 mean^2 cancels when |mean| >> std
                    // (post-ReLU rows near the fp16 range).  The second pass subtracts mh = fp16(mean) in packed fp16 (exact or 1 ulp of a small
                    // difference) and removes the shift exactly: sum (x - mh)^2 = sum (x - mean)^2 + 256 (mean - mh)^2 because sum (x - mean) = 0.
                    float sum = 0.f, sq = 0.f;
                    half2_ one2; one2.x = one2.y = (half_t)1.f;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            half2_ v; v.x = bf[s][2 * e]; v.y = bf[s][2 * e + 1];
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
                            half2_ v; v.x = bf[s][2 * e]; v.y = bf[s][2 * e + 1];
                            const half2_ d = v - mh2;
                            sq = __builtin_amdgcn_fdot2(d, d, sq, false);
                        }
                    sq += __shfl_xor(sq, 32);
                    const float shift = mean - (float)mh;
                    const float ssq = fmaxf(sq - 256.f * shift * shift, 0.f);
                    const float rstd = rsqrtf(ssq * (1.f / 256.f) + 1e-5f);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const f32x4 g0 = *(const f32x4*)&ln_lds[16 * s + 8 * hh], g1 = *(const f32x4*)&ln_lds[16 * s + 8 * hh + 4];
                        const f32x4 b0 = *(const f32x4*)&ln_lds[256 + 16 * s + 8 * hh], b1 = *(const f32x4*)&ln_lds[256 + 16 * s + 8 * hh + 4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bf[s][e] = (half_t)__builtin_amdgcn_fmed3f(((float)bf[s][e] - mean) * rstd * g0[e] + b0[e], -65504.f, 65504.f);
                            bf[s][4 + e] = (half_t)__builtin_amdgcn_fmed3f(((float)bf[s][4 + e] - mean) * rstd * g1[e] + b1[e], -65504.f, 65504.f);
                        }
                    }
                }
            }
            if (!S.cont) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                // stage n has landed (all but this wave's 4 youngest LDS-DMA pieces), every wave is done with stage n - 1: its
                // slot takes stage n + 2
                if constexpr (SPLIT) {
                    // the group that requested stage n waits for all its pieces; the other group's pieces (stage n + 1) stay in flight; the barrier publishes
                    if ((wave >> 2) == (n & 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    if (n + 1 < total_stages) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");     // (lgkmcnt: tools/check_barrier_reads.py)
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __syncthreads();
                issue(n + 2);
                const char* img = wc_smem + (n % WC_RING) * WC_STAGE + lane * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half8 af[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) af[t] = *(const half8*)(img + (t * 4 + q) * 1024);
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[4 * kt + q], acc[t], 0, 0, 0);
                }
                ++n;
            }
            if (S.finish == 0) continue;
            const float* bseg = bias_lds + S.bias_off;
            const float lo = LN && S.linear ? -65504.f : 0.f;
            if (S.finish == 1) {
                // -> the next layer's B fragments: k-step s = 2 t + gp of the new input
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) bf[2 * t + gp] = wc_regroup(acc[t], gp, bseg + 32 * t, hh, lo);
            } else {
                half_t* orow = p.out + pt * p.ldo + S.coff + 8 * hh;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const half8 v = wc_regroup(acc[t], gp, bseg + 32 * t, hh, lo);
                        if (32 * t + 16 * gp < S.cvalid) *(half8*)(orow + 32 * t + 16 * gp) = v;
                    }
            }
        }
    }
}

// the largest dynamic LDS either launch form asks for (ring + bias rows of WC_MAXSEG passes + LayerNorm affine): set once, for both
constexpr size_t WC_LDS_MAX = (size_t)WC_RING * WC_STAGE + (size_t)(WC_MAXSEG * 256 + 512) * sizeof(float);
static hipError_t wc_allow_lds() {
    static PcdLdsOnce once[4];
    hipError_t e = pcd_allow_lds(once[0], (const void*)pw_wide_chain_kernel<false, false>, (int)WC_LDS_MAX);
    if (e == hipSuccess) e = pcd_allow_lds(once[1], (const void*)pw_wide_chain_kernel<true, false>, (int)WC_LDS_MAX);
    if (e == hipSuccess) e = pcd_allow_lds(once[2], (const void*)pw_wide_chain_kernel<false, true>, (int)WC_LDS_MAX);
    if (e == hipSuccess) e = pcd_allow_lds(once[3], (const void*)pw_wide_chain_kernel<true, true>, (int)WC_LDS_MAX);
    return e;
}

// W [C][ldw] fp16 (columns k0 .. k0 + 63 of channels c0 .. c0 + 255, rows >= c_limit read as zero) -> one stage image
__global__ __launch_bounds__(256) void wc_pack_kernel(const half_t* __restrict__ w, int64_t ldw, int c0, int c_limit, int k0, char* __restrict__ img) {
    // piece id = (t * 4 + q) * 64 + lane: 8 halfs W[c0 + 32 t + (lane & 31)][k0 + 16 q + 8 (lane >> 5) .. + 7]
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 8 * 4 * 64) return;
    const int lane = id & 63, tq = id >> 6, t = tq >> 2, q = tq & 3;
    const int ch = c0 + 32 * t + (lane & 31);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
    if (ch < c_limit) v = *(const half8*)(w + (int64_t)ch * ldw + k0 + 16 * q + 8 * (lane >> 5));
    *(half8*)(img + (size_t)id * 16) = v;
}

}  // namespace pcd

using namespace pcd;

static int g_wc_split = 1;          // pcd_pw_wide_config: which waves request the weight images (0: every wave 4 pieces; 1, default: one wave per SIMD 8 in the chains;
                                    // 2: in the LN + Linear launches too)
extern "C" int pcd_pw_wide_config(int split) { g_wc_split = split < 0 ? 0 : (split > 2 ? 2 : split); return PCD_OK; }

extern "C" size_t pcd_pw_wide_packed_bytes(int chain) {
    return (chain == 0 || chain == 1) ? (size_t)16 * WC_STAGE + (size_t)4 * 256 * sizeof(float) : 0;
}

// chain 0 (E3): w[0..2] = enc3.conv1 [256][256], conv2 [256][256], conv3 [512][256]; chain 1 (D2): dec2.conv1 [256][512], conv2 [256][256],
// conv3 [128][256].  Writes the stage images followed by the per-pass bias rows (4 x 256 fp32) into `packed`.
extern "C" int pcd_pw_wide_pack(int chain, const void* const* w, const float* const* b, void* packed, void* stream) {
    PCD_CHECK_ARG((chain == 0 || chain == 1) && w && b && packed && w[0] && w[1] && w[2] && b[0] && b[1] && b[2]);
    hipStream_t s = (hipStream_t)stream;
    char* img = (char*)packed;
    float* bias = (float*)(img + (size_t)16 * WC_STAGE);
    // passes in execution order: (weights, ldw, first channel, channel limit, first k, bias source, bias channels)
    struct Pass { int layer; int ldw; int c0; int climit; int k0; };
    const Pass e3[4] = {{0, 256, 0, 256, 0}, {1, 256, 0, 256, 0}, {2, 256, 0, 512, 0}, {2, 256, 256, 512, 0}};
    const Pass d2[4] = {{0, 512, 0, 256, 0}, {0, 512, 0, 256, 256}, {1, 256, 0, 256, 0}, {2, 256, 0, 128, 0}};
    const Pass* ps = chain == 0 ? e3 : d2;
    for (int i = 0; i < 4; ++i)
        for (int kt = 0; kt < 4; ++kt)
            hipLaunchKernelGGL(wc_pack_kernel, dim3(8), dim3(256), 0, s, (const half_t*)w[ps[i].layer], (int64_t)ps[i].ldw, ps[i].c0, ps[i].climit,
                               ps[i].k0 + 64 * kt, img + (size_t)(4 * i + kt) * WC_STAGE);
    PCD_CHECK_LAUNCH();
    PCD_CHECK_HIP(hipMemsetAsync(bias, 0, 4 * 256 * sizeof(float), s));
    if (chain == 0) {
        PCD_CHECK_HIP(hipMemcpyAsync(bias, b[0], 256 * 4, hipMemcpyDeviceToDevice, s));
        PCD_CHECK_HIP(hipMemcpyAsync(bias + 256, b[1], 256 * 4, hipMemcpyDeviceToDevice, s));
        PCD_CHECK_HIP(hipMemcpyAsync(bias + 512, b[2], 512 * 4, hipMemcpyDeviceToDevice, s));
    } else {
        PCD_CHECK_HIP(hipMemcpyAsync(bias + 256, b[0], 256 * 4, hipMemcpyDeviceToDevice, s));      // pass 1 finishes dec2.conv1 (pass 0 has no epilogue)
        PCD_CHECK_HIP(hipMemcpyAsync(bias + 512, b[1], 256 * 4, hipMemcpyDeviceToDevice, s));
        PCD_CHECK_HIP(hipMemcpyAsync(bias + 768, b[2], 128 * 4, hipMemcpyDeviceToDevice, s));
    }
    return PCD_OK;
}

extern "C" int pcd_pw_wide_chain(int chain, const void* in1, const void* in2, int64_t m, const void* packed, void* out, void* stream) {
    PCD_CHECK_ARG((chain == 0 || chain == 1) && in1 && packed && out && m > 0 && m % WC_TILE == 0);
    PCD_CHECK_ARG(chain == 0 || in2 != nullptr);
    WcParams p{};
    p.in1 = (const half_t*)in1; p.in2 = (const half_t*)in2; p.m = m;
    p.wpacked = (const char*)packed;
    p.bias = (const float*)((const char*)packed + (size_t)16 * WC_STAGE);
    p.out = (half_t*)out;
    p.nseg = 4;
    if (chain == 0) {
        p.ldo = 512;
        p.seg[0] = WcSeg{1, 0, 1, 0, 256, 0, 0, 0};
        p.seg[1] = WcSeg{0, 0, 1, 0, 256, 256, 0, 0};
        p.seg[2] = WcSeg{0, 0, 2, 0, 256, 512, 1, 0};
        p.seg[3] = WcSeg{0, 0, 2, 256, 256, 768, 0, 0};
    } else {
        p.ldo = 128;
        p.seg[0] = WcSeg{1, 0, 0, 0, 256, 0, 0, 0};
        p.seg[1] = WcSeg{2, 1, 1, 0, 256, 256, 0, 0};
        p.seg[2] = WcSeg{0, 0, 1, 0, 256, 512, 0, 0};
        p.seg[3] = WcSeg{0, 0, 2, 0, 128, 768, 0, 0};
    }
    const size_t lds = (size_t)WC_RING * WC_STAGE + 4 * 256 * sizeof(float);
    PCD_CHECK_HIP(wc_allow_lds());
    const int64_t tiles = m / WC_TILE;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    if (g_wc_split) hipLaunchKernelGGL((pw_wide_chain_kernel<false, true>), dim3(grid), dim3(WC_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((pw_wide_chain_kernel<false, false>), dim3(grid), dim3(WC_THREADS), lds, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// ---- LayerNorm(256) + Linear(256, 256 P) [+ ReLU] as one launch of the same kernel (P <= 4 output passes over B fragments that are normalised as they are
// loaded): the in_proj and the first FFN layer of the C = 256 attention blocks (reference networks.py:61-66, 81-82) without the LayerNorm launch and its
// 67 + 67 MB.  packed = P x 4 stage images | bias [P][256] fp32 | gamma [256] | beta [256].
extern "C" size_t pcd_pw_wide_ln_linear_packed_bytes(int passes) {
    return (passes >= 1 && passes <= 4) ? (size_t)passes * 4 * WC_STAGE + (size_t)(passes * 256 + 512) * sizeof(float) : 0;
}

extern "C" int pcd_pw_wide_ln_linear_pack(const void* w, const float* b, int passes, const float* ln_g, const float* ln_b, void* packed, void* stream) {
    PCD_CHECK_ARG(w && b && ln_g && ln_b && packed && passes >= 1 && passes <= 4);
    hipStream_t s = (hipStream_t)stream;
    char* img = (char*)packed;
    for (int i = 0; i < passes; ++i)
        for (int kt = 0; kt < 4; ++kt)
            hipLaunchKernelGGL(wc_pack_kernel, dim3(8), dim3(256), 0, s, (const half_t*)w, (int64_t)256, 256 * i, 256 * passes, 64 * kt,
                               img + (size_t)(4 * i + kt) * WC_STAGE);
    PCD_CHECK_LAUNCH();
    float* f = (float*)(img + (size_t)passes * 4 * WC_STAGE);
    PCD_CHECK_HIP(hipMemcpyAsync(f, b, (size_t)passes * 256 * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(f + passes * 256, ln_g, 256 * 4, hipMemcpyDeviceToDevice, s));
    PCD_CHECK_HIP(hipMemcpyAsync(f + passes * 256 + 256, ln_b, 256 * 4, hipMemcpyDeviceToDevice, s));
    return PCD_OK;
}

extern "C" int pcd_pw_wide_ln_linear_supported(int dim, int64_t rows) { return dim == 256 && rows > 0 && rows % WC_TILE == 0 ? 1 : 0; }

extern "C" int pcd_pw_wide_ln_linear(const void* packed, int passes, int relu, const void* x, int64_t m, void* out, void* stream) {
    PCD_CHECK_ARG(packed && x && out && x != out && passes >= 1 && passes <= 4 && m > 0 && m % WC_TILE == 0);
    WcParams p{};
    p.in1 = (const half_t*)x; p.in2 = nullptr; p.m = m;
    p.wpacked = (const char*)packed;
    p.bias = (const float*)((const char*)packed + (size_t)passes * 4 * WC_STAGE);
    p.ln = p.bias + passes * 256;
    p.out = (half_t*)out; p.ldo = 256 * passes;
    p.nseg = passes;
    for (int i = 0; i < passes; ++i) p.seg[i] = WcSeg{i == 0 ? 1 : 0, 0, 2, 256 * i, 256, 256 * i, i + 1 < passes ? 1 : 0, relu ? 0 : 1};
    const size_t lds = WC_LDS_MAX;
    PCD_CHECK_HIP(wc_allow_lds());
    const int64_t tiles = m / WC_TILE;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    // (the LN + Linear launches store a 256-wide output piece per pass: a requesting wave's vmcnt(0) would wait for those stores at every stage -- measured neutral to
    // slightly slower in the attention U-Net's forward, so they keep "every wave requests" unless pcd_pw_wide_config(2) asks for the split form)
    if (g_wc_split == 2) hipLaunchKernelGGL((pw_wide_chain_kernel<true, true>), dim3(grid), dim3(WC_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((pw_wide_chain_kernel<true, false>), dim3(grid), dim3(WC_THREADS), lds, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
