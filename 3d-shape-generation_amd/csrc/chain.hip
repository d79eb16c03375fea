// Chains of narrow pointwise layers of UNetPointNetLarge (reference networks.py:16-49, 779-818) in one launch each.
//
// The 64..128-channel layers at the two ends of the U-net are HBM / launch bound as separate GEMMs: each reads and writes
// a [B*N][64..128] fp16 tensor (17-33 MB) for 2-4 GFLOP of work and costs 12-20 us.  Here a wave owns 32 points and
// carries them through the whole chain with the intermediate activations in its own LDS buffers; the chain's weights
// (24-64 KB, BatchNorm folded) sit in LDS for the lifetime of the persistent workgroup:
//   E1: xyz (fp32) -> enc1.conv1 (K = 3, per-shape time bias, VALU) -> enc1.conv2 64->64 -> enc1.conv3 64->128 -> x1
//   E2: x1 -> enc2.conv1 128->128 -> enc2.conv2 128->128 -> (enc2.conv3 128->256 stays a GEMM: its 67 MB output bounds it)
//   D1: dec1.conv1's output -> dec1.conv2 128->128 -> dec1.conv3 128->64 -> output.0 64->64 -> output.3 64->3 (fp32 eps)
// Every layer is the transposed product D[channel][point] = W[channel][k] . act[point][k] on v_mfma_f32_32x32x16_f16
// (weights as the A operand): a lane then holds 4 consecutive channels of one point per accumulator group, which is an
// 8-byte piece of the next layer's [point][channel] LDS image (or of the output row).  Waves never synchronise after the
// weights are in place.  fp32 accumulation, bias + ReLU + fp16 rounding per layer exactly as the GEMM epilogue does.
//
// Round 4: hi / lo weights (E1 and D1).  These 64..128-channel layers are the direct route from the input coordinates to the
// predicted noise, and the fp16 rounding of THEIR weights is what the 1000-step DDPM trajectory of the fp16 path deviates by
// (tools/attribute_fp16_layers.py, profiles/r04_f: 1.4e-3 of cloud error from layers 0, 1, 22-25 against 2.6e-5 from the fourteen big
// layers in between).  With LO the weight image carries, next to the fp16 weights, the fp16 of their rounding residuals
// (W = hi + lo to ~22 bits), and every layer runs its K loop twice: first against hi, then against lo (the order
// pcd_gemm_f16_hilo uses, so the per-layer launches stay bit-identical).  < 2 % of the step's FLOPs.
#include "common.h"

namespace pcd {

constexpr int PW_MAXL = 3;
constexpr int PW_WAVES = 8, PW_THREADS = 64 * PW_WAVES, PW_TILE = 32 * PW_WAVES;      // 2 waves per SIMD, 256 points per workgroup tile
struct PwChainParams {
    int64_t m;                       // points (B * N)
    int rows_per_shape;
    // E1 input
    const float* xyz; const float* w_xyz; const float* tbias; int tb_stride;      // [m][3], [64][3], [n_t][64], row stride
    // E2 / D1 input
    const half_t* in16;              // [m][128]
    const half_t* w[PW_MAXL]; const float* b[PW_MAXL];                             // fp16 [C][K] (LO: [C][2 K] = hi | lo), fp32 [C]
    const float* head_w; const float* head_b;                                      // D1: fp32 [3][64], [3]
    half_t* out16; float* out32;
    float* zero; int64_t zero_n;     // E1 only (internal): floats to clear while the chain runs -- the pooled-maximum buffer the column-max GEMM later
                                     // in the same forward accumulates into (one launch less per step; multiple of 4, 16-byte aligned)
};

template <int K> struct PwImg { static constexpr int STR = K + 8; };              // halfs per LDS row (16-byte pad)

// one layer for this wave's 32 points: act_in [32][K + 8] (LDS) x W image [C][K + 8] (LDS) -> act_out [32][C + 8] (LDS) or
// global rows [pt0 + point][C]
template <int K, int C, bool TO_GLOBAL, bool RELU = true, bool LO = false>
__device__ __forceinline__ void pw_layer(const half_t* act_in, const half_t* wimg, const float* __restrict__ bias,
                                         half_t* act_out, half_t* __restrict__ gout, int64_t pt0, int64_t m, int lane,
                                         const half_t* wimg_lo = nullptr) {
    constexpr int NT = C / 32;
    const int pnt = lane & 31, hh = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll 2
    for (int s = 0; s < K / 16; ++s) {
        const half8 bfrag = *(const half8*)(act_in + pnt * PwImg<K>::STR + 16 * s + 8 * hh);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const half8 afrag = *(const half8*)(wimg + (32 * t + pnt) * PwImg<K>::STR + 16 * s + 8 * hh);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag, acc[t], 0, 0, 0);
        }
    }
    if constexpr (LO) {           // second pass of the K loop against the residual weights (after ALL the hi products, like pcd_gemm_f16_hilo)
#pragma unroll 2
        for (int s = 0; s < K / 16; ++s) {
            const half8 bfrag = *(const half8*)(act_in + pnt * PwImg<K>::STR + 16 * s + 8 * hh);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const half8 afrag = *(const half8*)(wimg_lo + (32 * t + pnt) * PwImg<K>::STR + 16 * s + 8 * hh);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag, acc[t], 0, 0, 0);
            }
        }
    }
    // accumulator register 4 g + e of tile t: channel 32 t + 8 g + 4 hh + e of point pnt
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * t + 8 * g + 4 * hh;
            const f32x4 bv = *(const f32x4*)(bias + c0);
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)__builtin_amdgcn_fmed3f(acc[t][4 * g + e] + bv[e], RELU ? 0.f : -65504.f, 65504.f);   // (ReLU +) fp16 saturation
            *(half4*)(act_out + pnt * PwImg<C>::STR + c0) = o;
        }
    if (TO_GLOBAL) {
        // rows leave through the wave's LDS image as whole 16-byte chunks: 2 C / 16 lanes cover one point's row, so a
        // store instruction writes complete cache lines (the accumulator layout itself would write 16 B of 32 different rows)
        constexpr int CPR = C / 8;                    // chunks per row
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = i * 64 + lane, row = c / CPR, ch = c - row * CPR;
            const half8 v = *(const half8*)(act_out + row * PwImg<C>::STR + ch * 8);
            if (pt0 + row < m) *(half8*)(gout + (pt0 + row) * C + ch * 8) = v;
        }
    }
}

// weights [C][K] fp16 (global, row stride ldw) -> LDS image [C][K + 8], all NTHR threads; every load of the layer is in
// flight before the first LDS store
template <int K, int C, int NTHR = PW_THREADS>
__device__ __forceinline__ void pw_load_w(const half_t* __restrict__ w, int64_t ldw, half_t* img) {
    constexpr int CH = C * (K / 8), IT = (CH + NTHR - 1) / NTHR;
    half8 r[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int i = j * NTHR + threadIdx.x, row = i / (K / 8), ch = i - row * (K / 8);
        if (CH % NTHR == 0 || i < CH) r[j] = *(const half8*)(w + row * ldw + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int i = j * NTHR + threadIdx.x, row = i / (K / 8), ch = i - row * (K / 8);
        if (CH % NTHR == 0 || i < CH) *(half8*)(img + row * PwImg<K>::STR + ch * 8) = r[j];
    }
}

template <int CHAIN, bool LO = false, int WAVES = PW_WAVES>
__global__ __launch_bounds__(64 * WAVES) void pw_chain_kernel(PwChainParams p) {
    // layer shapes of the chain
    constexpr int K0 = CHAIN == 0 ? 64 : 128, C0 = CHAIN == 0 ? 64 : 128;
    constexpr int K1 = C0, C1 = CHAIN == 0 ? 128 : (CHAIN == 1 ? 128 : 64);
    constexpr int K2 = 64, C2 = 64;                                              // D1 only: output.0
    constexpr int W0 = C0 * PwImg<K0>::STR, W1 = C1 * PwImg<K1>::STR, W2 = CHAIN == 2 ? C2 * PwImg<K2>::STR : 0;
    constexpr int WALL = W0 + W1 + W2;
    constexpr int ACT = 32 * PwImg<128>::STR;
    constexpr int NTHR = 64 * WAVES, TILE = 32 * WAVES;
    __shared__ __attribute__((aligned(16))) half_t wimg[WALL * (LO ? 2 : 1)];   // LO: the residual images behind the weight images
    __shared__ __attribute__((aligned(16))) half_t act[WAVES][ACT];         // one buffer per wave, rewritten in place layer by layer
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pnt = lane & 31, hh = lane >> 5;
    constexpr int LDM = LO ? 2 : 1;                                          // global rows are [hi K | lo K] when LO
    pw_load_w<K0, C0, NTHR>(p.w[0], LDM * K0, wimg);
    pw_load_w<K1, C1, NTHR>(p.w[1], LDM * K1, wimg + W0);
    if (CHAIN == 2) pw_load_w<K2, C2, NTHR>(p.w[2], LDM * K2, wimg + W0 + W1);
    if constexpr (LO) {
        pw_load_w<K0, C0, NTHR>(p.w[0] + K0, 2 * K0, wimg + WALL);
        pw_load_w<K1, C1, NTHR>(p.w[1] + K1, 2 * K1, wimg + WALL + W0);
        if (CHAIN == 2) pw_load_w<K2, C2, NTHR>(p.w[2] + K2, 2 * K2, wimg + WALL + W0 + W1);
    }
    __syncthreads();
    if (CHAIN == 0 && p.zero != nullptr) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        for (int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x; i < p.zero_n / 4; i += (int64_t)gridDim.x * NTHR) ((f32x4*)p.zero)[i] = z4;
    }
    half_t* buf = act[wave];
    const int64_t ntiles = (p.m + TILE - 1) / TILE;
    // E2 / D1: the next tile's [32 points][128] fp16 rows travel in registers while this tile computes (one wave per
    // SIMD: nothing else hides the load latency).  16 chunks of 16 B per point, 8 per lane.
    half8 pre[8];
    auto fetch = [&](int64_t tile) __attribute__((always_inline)) {
        const int64_t pt0 = tile * TILE + wave * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane;
            int64_t pt = pt0 + (c >> 4);
            pt = pt < p.m ? pt : p.m - 1;
            pre[i] = *(const half8*)(p.in16 + pt * 128 + (c & 15) * 8);
        }
    };
    if (CHAIN != 0 && (int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pt0 = tile * TILE + wave * 32;
        // ---- input -> buf
        if (CHAIN == 0) {
            // enc1.conv1: K = 3 half + per-shape time bias + ReLU (the arithmetic of enc1_xyz_kernel); lane = (point, 32 channels)
            int64_t pt = pt0 + pnt;
            pt = pt < p.m ? pt : p.m - 1;
            const float px = p.xyz[pt * 3 + 0], py = p.xyz[pt * 3 + 1], pz = p.xyz[pt * 3 + 2];
            const float* tb = p.tbias + (int64_t)(pt / p.rows_per_shape) * p.tb_stride * 64 + 32 * hh;
            const float* wr = p.w_xyz + (int64_t)(32 * hh) * 3;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c8 * 8 + e;
                    float v = tb[c];
                    v = fmaf(wr[c * 3 + 0], px, v);
                    v = fmaf(wr[c * 3 + 1], py, v);
                    v = fmaf(wr[c * 3 + 2], pz, v);
                    o[e] = to_half_sat(fmaxf(v, 0.f));
                }
                *(half8*)(buf + pnt * PwImg<64>::STR + 32 * hh + c8 * 8) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = i * 64 + lane;
                *(half8*)(buf + (c >> 4) * PwImg<128>::STR + (c & 15) * 8) = pre[i];
            }
            if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        }
        // ---- layers
        const half_t* wlo = wimg + WALL;
        pw_layer<K0, C0, false, true, LO>(buf, wimg, p.b[0], buf, nullptr, pt0, p.m, lane, wlo);
        if (CHAIN != 2) {
            pw_layer<K1, C1, true, true, LO>(buf, wimg + W0, p.b[1], buf, p.out16, pt0, p.m, lane, wlo + W0);
        } else {
            pw_layer<K1, C1, false, true, LO>(buf, wimg + W0, p.b[1], buf, nullptr, pt0, p.m, lane, wlo + W0);
            pw_layer<K2, C2, false, true, LO>(buf, wimg + W0 + W1, p.b[2], buf, nullptr, pt0, p.m, lane, wlo + W0 + W1);
            // output.3: 64 -> 3, fp32 (the arithmetic of head3_kernel): lanes 0..31, one point each
            if (hh == 0 && pt0 + pnt < p.m) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const half8 v = *(const half8*)(buf + pnt * PwImg<64>::STR + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        a0 = fmaf(p.head_w[c8 * 8 + e], f, a0);
                        a1 = fmaf(p.head_w[64 + c8 * 8 + e], f, a1);
                        a2 = fmaf(p.head_w[128 + c8 * 8 + e], f, a2);
                    }
                }
                float* o = p.out32 + (pt0 + pnt) * 3;
                o[0] = a0 + p.head_b[0];
                o[1] = a1 + p.head_b[1];
                o[2] = a2 + p.head_b[2];
            }
        }
    }
}

// One pointwise layer out[m][C] = act(in[m][K] . W^T + bias): the 1x1x1 shortcut convolutions of ResidualBlock3D
// (reference networks.py:485-490; NDHWC rows) on the same scheme.  The implicit-GEMM convolution kernel spends a
// prologue of index arithmetic and an LDS epilogue per 128-row workgroup on what is one K tile of work.
template <int K, int C, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void pw_1x1_kernel(const half_t* __restrict__ in, int64_t m,
                                                          const half_t* __restrict__ w, int64_t ldw,
                                                          const float* __restrict__ bias, int relu,
                                                          half_t* __restrict__ out) {
    constexpr int KC = K > C ? K : C, ACT = 32 * (KC + 8), NCH = K / 16;      // NCH: 16-byte input chunks per lane and tile
    __shared__ __attribute__((aligned(16))) half_t wimg[C * PwImg<K>::STR];
    __shared__ __attribute__((aligned(16))) half_t act[WAVES][ACT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    pw_load_w<K, C, 64 * WAVES>(w, ldw, wimg);
    __syncthreads();
    half_t* buf = act[wave];
    constexpr int TILE = 32 * WAVES, CPR = K / 8;
    const int64_t ntiles = (m + TILE - 1) / TILE;
    half8 pre[NCH];
    auto fetch = [&](int64_t tile) __attribute__((always_inline)) {
        const int64_t pt0 = tile * TILE + wave * 32;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            int64_t pt = pt0 + c / CPR;
            pt = pt < m ? pt : m - 1;
            pre[i] = *(const half8*)(in + pt * K + (c % CPR) * 8);
        }
    };
    if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pt0 = tile * TILE + wave * 32;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            *(half8*)(buf + (c / CPR) * PwImg<K>::STR + (c % CPR) * 8) = pre[i];
        }
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        if (relu) pw_layer<K, C, true, true>(buf, wimg, bias, buf, out, pt0, m, lane);
        else pw_layer<K, C, true, false>(buf, wimg, bias, buf, out, pt0, m, lane);
    }
}

}  // namespace pcd

using namespace pcd;

static inline unsigned pw_grid(int64_t m) {
    const int64_t tiles = (m + PW_TILE - 1) / PW_TILE;
    return (unsigned)(tiles < 256 ? tiles : 256);        // persistent: the weight image is loaded once per workgroup
}

namespace pcd {
// pcd_pw_chain_enc1[_hilo] + an optional buffer to clear in the same launch (csrc/unet.hip: the pooled maxima)
int pw_chain_enc1_impl(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias, int tbias_shape_stride,
                       const void* w_conv2, const float* b_conv2, const void* w_conv3, const float* b_conv3, void* x1, bool hilo,
                       float* zero, int64_t zero_n, void* stream) {
    PCD_CHECK_ARG(x && w_xyz && tbias && w_conv2 && b_conv2 && w_conv3 && b_conv3 && x1);
    PCD_CHECK_ARG(m > 0 && rows_per_shape > 0 && tbias_shape_stride >= 0);
    PCD_CHECK_ARG(zero == nullptr || (zero_n % 4 == 0 && ((uintptr_t)zero & 15) == 0));
    PwChainParams p{};
    p.m = m; p.rows_per_shape = rows_per_shape; p.xyz = x; p.w_xyz = w_xyz; p.tbias = tbias; p.tb_stride = tbias_shape_stride;
    p.w[0] = (const half_t*)w_conv2; p.b[0] = b_conv2; p.w[1] = (const half_t*)w_conv3; p.b[1] = b_conv3;
    p.out16 = (half_t*)x1; p.zero = zero; p.zero_n = zero_n;
    if (hilo) hipLaunchKernelGGL((pw_chain_kernel<0, true, PW_WAVES>), dim3(pw_grid(m)), dim3(PW_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((pw_chain_kernel<0>), dim3(pw_grid(m)), dim3(PW_THREADS), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
}  // namespace pcd

extern "C" int pcd_pw_chain_enc1(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias,
                                 int tbias_shape_stride, const void* w_conv2, const float* b_conv2, const void* w_conv3,
                                 const float* b_conv3, void* x1, void* stream) {
    return pcd::pw_chain_enc1_impl(x, m, rows_per_shape, w_xyz, tbias, tbias_shape_stride, w_conv2, b_conv2, w_conv3, b_conv3, x1, false,
                                   nullptr, 0, stream);
}

extern "C" int pcd_pw_chain_enc1_hilo(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias,
                                      int tbias_shape_stride, const void* w_conv2, const float* b_conv2, const void* w_conv3,
                                      const float* b_conv3, void* x1, void* stream) {
    return pcd::pw_chain_enc1_impl(x, m, rows_per_shape, w_xyz, tbias, tbias_shape_stride, w_conv2, b_conv2, w_conv3, b_conv3, x1, true,
                                   nullptr, 0, stream);
}

extern "C" int pcd_pw_chain_128(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b,
                                const float* b_b, void* out, void* stream) {
    PCD_CHECK_ARG(in && w_a && b_a && w_b && b_b && out && m > 0);
    PwChainParams p{};
    p.m = m; p.rows_per_shape = 1; p.in16 = (const half_t*)in;
    p.w[0] = (const half_t*)w_a; p.b[0] = b_a; p.w[1] = (const half_t*)w_b; p.b[1] = b_b;
    p.out16 = (half_t*)out;
    hipLaunchKernelGGL((pw_chain_kernel<1>), dim3(pw_grid(m)), dim3(PW_THREADS), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_pw_chain_tail(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b,
                                 const float* b_b, const void* w_c, const float* b_c, const float* head_w,
                                 const float* head_b, float* eps, void* stream) {
    PCD_CHECK_ARG(in && w_a && b_a && w_b && b_b && w_c && b_c && head_w && head_b && eps && m > 0);
    PwChainParams p{};
    p.m = m; p.rows_per_shape = 1; p.in16 = (const half_t*)in;
    p.w[0] = (const half_t*)w_a; p.b[0] = b_a; p.w[1] = (const half_t*)w_b; p.b[1] = b_b; p.w[2] = (const half_t*)w_c; p.b[2] = b_c;
    p.head_w = head_w; p.head_b = head_b; p.out32 = eps;
    hipLaunchKernelGGL((pw_chain_kernel<2>), dim3(pw_grid(m)), dim3(PW_THREADS), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// D1 with hi / lo weights: two weight images (2 x 61 KB) leave room for four waves' activation buffers, not eight
extern "C" int pcd_pw_chain_tail_hilo(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b,
                                      const float* b_b, const void* w_c, const float* b_c, const float* head_w,
                                      const float* head_b, float* eps, void* stream) {
    PCD_CHECK_ARG(in && w_a && b_a && w_b && b_b && w_c && b_c && head_w && head_b && eps && m > 0);
    PwChainParams p{};
    p.m = m; p.rows_per_shape = 1; p.in16 = (const half_t*)in;
    p.w[0] = (const half_t*)w_a; p.b[0] = b_a; p.w[1] = (const half_t*)w_b; p.b[1] = b_b; p.w[2] = (const half_t*)w_c; p.b[2] = b_c;
    p.head_w = head_w; p.head_b = head_b; p.out32 = eps;
    constexpr int WV = 4;
    const int64_t tiles = (m + 32 * WV - 1) / (32 * WV);
    hipLaunchKernelGGL((pw_chain_kernel<2, true, WV>), dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(64 * WV), 0, (hipStream_t)stream, p);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv1x1_supported(int k, int c) {
    return (k == 32 && c == 64) || (k == 64 && c == 128) || (k == 128 && c == 256);
}

extern "C" int pcd_conv1x1_f16(const void* in, int64_t m, int k, const void* w, int64_t ldw, const float* bias, int relu,
                               int c, void* out, void* stream) {
    PCD_CHECK_ARG(in && w && bias && out && m > 0 && ldw >= k && ldw % 8 == 0);
    PCD_CHECK_ARG(pcd_conv1x1_supported(k, c));
    hipStream_t s = (hipStream_t)stream;
    const half_t* in16 = (const half_t*)in; const half_t* w16 = (const half_t*)w; half_t* o16 = (half_t*)out;
    auto grid = [&](int tile) { const int64_t t = (m + tile - 1) / tile; return dim3((unsigned)(t < 512 ? t : 512)); };
    if (k == 32) hipLaunchKernelGGL((pw_1x1_kernel<32, 64, 8>), grid(256), dim3(512), 0, s, in16, m, w16, ldw, bias, relu, o16);
    else if (k == 64) hipLaunchKernelGGL((pw_1x1_kernel<64, 128, 8>), grid(256), dim3(512), 0, s, in16, m, w16, ldw, bias, relu, o16);
    else hipLaunchKernelGGL((pw_1x1_kernel<128, 256, 4>), grid(128), dim3(256), 0, s, in16, m, w16, ldw, bias, relu, o16);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
