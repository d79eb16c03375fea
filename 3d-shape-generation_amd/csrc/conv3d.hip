// K9: 3-D convolutions of VAE3DLarge (reference networks.py:2225-2264, 471-504) as implicit GEMM
// on MFMA.  Activations are NDHWC fp16, so the Cin channels of one input voxel are contiguous
// and an im2col row is a list of (tap, voxel) segments: the A-tile loader below gathers those
// segments with the same 16-B global_load_lds + source-side XOR swizzle as the dense GEMM
// (csrc/gemm_f16.hip); out-of-range taps (padding, K padding) read from a zero page.
// Conv3d (any k/stride/pad) and each output-parity class of ConvTranspose3d(k4,s2,p1) are the
// same kernel with different tap tables and output row maps.  Eval-mode BatchNorm3d is folded
// into W/bias on the host; epilogue = bias (+ residual) (+ ReLU).
//
// Also here: the two degenerate layers that are not GEMM shaped -- the first conv (Cin = 1,
// K = 27) and the last conv (Cout = 1) with its sigmoid -- as direct VALU kernels.
#include "common.h"

namespace pcd {

constexpr int CBK = 64;
constexpr int CROWB = CBK * 2;

struct ConvParams {
    const half_t* in; int D, H, W, Cin, cin_shift;
    int Do, Ho, Wo, stride;           // row space: m = ((b*Do+oz)*Ho+oy)*Wo+ox ; input coord = o*stride + d
    int ntaps, kpad;
    const int* taps;                  // device [ntaps] packed (dz & 0xff) | (dy & 0xff) << 8 | (dx & 0xff) << 16
    const half_t* w;                  // [Cout][kpad]
    const float* bias;
    const half_t* resid;              // optional, indexed like out
    half_t* out; int Cout;
    int OD, OH, OW, os, pz, py, px;   // output voxel = (oz*os+pz, oy*os+py, ox*os+px) in an OD x OH x OW grid
    int relu;
    int M;
    const half_t* zero;               // >= 128 B of zeros
    int tiles_n;
};

__device__ __forceinline__ void cglds16(const half_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(ConvParams p) {
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int STAGE_BYTES = (BM + BN) * CROWB;
    constexpr int OUT_LD = BN * 2 + 16;
    constexpr int LDS_BYTES = (2 * STAGE_BYTES > BM * OUT_LD) ? 2 * STAGE_BYTES : BM * OUT_LD;
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.kpad / CBK;

    // the rows this thread stages (fixed for the whole K loop): decode the output voxel once
    const int srow = wave * 8 + (lane >> 3);                 // + r*32
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);       // logical 16-B chunk (same for every r)
    int rb[AR], rz[AR], ry[AR], rx[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        int m = m0 + r * 32 + srow;
        m = m < p.M ? m : p.M - 1;
        const int ox = m % p.Wo; int t = m / p.Wo;
        const int oy = t % p.Ho; t /= p.Ho;
        const int oz = t % p.Do;
        rb[r] = t / p.Do;
        rz[r] = oz * p.stride; ry[r] = oy * p.stride; rx[r] = ox * p.stride;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const int kidx = kt * CBK + lchunk * 8;
        const int tap = kidx >> p.cin_shift, c = kidx & (p.Cin - 1);
        const bool tap_ok = tap < p.ntaps;
        int dz = 0, dy = 0, dx = 0;
        if (tap_ok) {
            const int pk = p.taps[tap];
            dz = (int)(signed char)(pk & 0xff); dy = (int)(signed char)((pk >> 8) & 0xff);
            dx = (int)(signed char)((pk >> 16) & 0xff);
        }
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iz = rz[r] + dz, iy = ry[r] + dy, ix = rx[r] + dx;
            const bool ok = tap_ok && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H &&
                            (unsigned)ix < (unsigned)p.W;
            const half_t* g = ok ? p.in + ((((int64_t)rb[r] * p.D + iz) * p.H + iy) * p.W + ix) * p.Cin + c
                                 : p.zero + (lane & 7) * 8;
            cglds16(g, base + (r * 32 + wave * 8) * CROWB);
        }
#pragma unroll
        for (int r = 0; r < BR; ++r) {
            int n = n0 + r * 32 + srow;
            n = n < p.Cout ? n : p.Cout - 1;
            cglds16(p.w + (int64_t)n * p.kpad + kidx, base + BM * CROWB + (r * 32 + wave * 8) * CROWB);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ra = wm * WM + (lane & 15), rbw = wn * WN + (lane & 15);
    const int swa = (ra >> 1) & 7, swb = (rbw >> 1) & 7, q = lane >> 4;
    int offa[2], offb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        offa[ks] = ra * CROWB + (((ks * 4 + q) ^ swa) << 4);
        offb[ks] = BM * CROWB + rbw * CROWB + (((ks * 4 + q) ^ swb) << 4);
    }

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const half8*)(base + offa[ks] + i * 16 * CROWB);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const half8*)(base + offb[ks] + j * 16 * CROWB);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: bias -> LDS (fp16) -> row-contiguous 16-B stores (+ residual, ReLU) through the row map
    const int colq = lane & 15;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int lcol = wn * WN + j * 16 + colq;
        const float bc = (p.bias != nullptr && n0 + lcol < p.Cout) ? p.bias[n0 + lcol] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = wm * WM + i * 16 + q * 4 + r;
                float v = acc[i][j][r] + bc;
                if (p.relu && p.resid == nullptr) v = fmaxf(v, 0.f);
                *(half_t*)(smem + lrow * OUT_LD + lcol * 2) = to_half_sat(v);
            }
    }
    __syncthreads();
    constexpr int CPR = BN / 8, TOTAL = BM * CPR;
#pragma unroll
    for (int it = 0; it < TOTAL / 256; ++it) {
        const int idx = it * 256 + tid;
        const int lrow = idx / CPR, ch = idx - lrow * CPR;
        const int m = m0 + lrow, col = n0 + ch * 8;
        if (m < p.M && col < p.Cout) {
            const int ox = m % p.Wo; int t = m / p.Wo;
            const int oy = t % p.Ho; t /= p.Ho;
            const int oz = t % p.Do; const int b = t / p.Do;
            const int64_t orow = (((int64_t)b * p.OD + oz * p.os + p.pz) * p.OH + oy * p.os + p.py) * p.OW +
                                 ox * p.os + p.px;
            half8 v = *(const half8*)(smem + lrow * OUT_LD + ch * 16);
            if (p.resid != nullptr) {
                const half8 rs = *(const half8*)(p.resid + orow * p.Cout + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[e] + (float)rs[e];
                    if (p.relu) f = fmaxf(f, 0.f);
                    v[e] = to_half_sat(f);
                }
            }
            *(half8*)(p.out + orow * p.Cout + col) = v;
        }
    }
}

// first layer: x fp32 [B][D][H][W] (Cin = 1), k3 s1 p1 -> fp16 NDHWC [..][cout], ReLU.
// thread = (voxel, 8-channel chunk); w fp32 [cout][27], b fp32 [cout]
__global__ __launch_bounds__(256) void conv3d_first_kernel(const float* __restrict__ x, int B, int D, int H, int W,
                                                            int stride, const float* __restrict__ w,
                                                            const float* __restrict__ b, int cout,
                                                            half_t* __restrict__ out) {
    extern __shared__ float ws[];   // [cout][27] + [cout]
    for (int i = threadIdx.x; i < cout * 28; i += blockDim.x) ws[i] = i < cout * 27 ? w[i] : b[i - cout * 27];
    __syncthreads();
    const int chunks = cout / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;   // k3, pad 1
    const int64_t nvox = (int64_t)B * Do * Ho * Wo;
    if (idx >= nvox * chunks) return;
    const int64_t vox = idx / chunks;
    const int ch = (int)(idx - vox * chunks);
    int xx = (int)(vox % Wo); int64_t t = vox / Wo;
    int yy = (int)(t % Ho); t /= Ho;
    int zz = (int)(t % Do); const int bb = (int)(t / Do);
    xx *= stride; yy *= stride; zz *= stride;
    float tap[27];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                const bool ok = (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                tap[(kz * 3 + ky) * 3 + kx] = ok ? x[(((int64_t)bb * D + iz) * H + iy) * W + ix] : 0.f;
            }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 8 + e;
        float a = ws[cout * 27 + c];
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(ws[c * 27 + k], tap[k], a);
        o[e] = to_half_sat(fmaxf(a, 0.f));
    }
    *(half8*)(out + vox * cout + ch * 8) = o;
}

// last layer: fp16 NDHWC [..][CIN] -> fp32 [B][D][H][W], k3 s1 p1, Cout = 1, sigmoid.
// w fp32 [27][CIN], one thread per output voxel.
template <int CIN>
__global__ __launch_bounds__(256) void conv3d_last_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                           const float* __restrict__ w, float bias,
                                                           float* __restrict__ out) {
    __shared__ float ws[27 * CIN];
    for (int i = threadIdx.x; i < 27 * CIN; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= (int64_t)B * D * H * W) return;
    const int xx = (int)(vox % W); int64_t t = vox / W;
    const int yy = (int)(t % H); t /= H;
    const int zz = (int)(t % D); const int bb = (int)(t / D);
    float a = bias;
    for (int kz = 0; kz < 3; ++kz)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                if ((unsigned)iz >= (unsigned)D || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const half8* src = (const half8*)(in + ((((int64_t)bb * D + iz) * H + iy) * W + ix) * CIN);
                const float* wk = ws + ((kz * 3 + ky) * 3 + kx) * CIN;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8) {
                    const half8 v = src[c8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
                }
            }
    out[vox] = 1.f / (1.f + expf(-a));
}

// VAE3D's last layer (networks.py:2018-2019): ConvTranspose3d(CIN, 1, k3, s2, p1, output_padding 1) + Sigmoid.
// o = 2 i - 1 + k per dimension: even o takes (k=1, i=o/2); odd o takes (k=0, i=(o+1)/2) and (k=2, i=(o-1)/2).
// in fp16 NDHWC [B][D][H][W][CIN]; w fp32 [27][CIN] (tap-major); out fp32 [B][2D][2H][2W].
template <int CIN>
__global__ __launch_bounds__(256) void convT3d_last_kernel(const half_t* __restrict__ in, int B, int D, int H, int W,
                                                            const float* __restrict__ w, float bias,
                                                            float* __restrict__ out) {
    __shared__ float ws[27 * CIN];
    for (int i = threadIdx.x; i < 27 * CIN; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= (int64_t)B * OD * OH * OW) return;
    const int ox = (int)(vox % OW); int64_t t = vox / OW;
    const int oy = (int)(t % OH); t /= OH;
    const int oz = (int)(t % OD); const int bb = (int)(t / OD);
    float a = bias;
    for (int kz = (oz & 1) ? 0 : 1; kz < 3; kz += 2) {
        const int iz = (oz + 1 - kz) >> 1;
        if ((unsigned)iz >= (unsigned)D) continue;
        for (int ky = (oy & 1) ? 0 : 1; ky < 3; ky += 2) {
            const int iy = (oy + 1 - ky) >> 1;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = (ox & 1) ? 0 : 1; kx < 3; kx += 2) {
                const int ix = (ox + 1 - kx) >> 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                const half8* src = (const half8*)(in + ((((int64_t)bb * D + iz) * H + iy) * W + ix) * CIN);
                const float* wk = ws + ((kz * 3 + ky) * 3 + kx) * CIN;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8) {
                    const half8 v = src[c8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = fmaf(wk[c8 * 8 + e], (float)v[e], a);
                }
            }
        }
    }
    out[vox] = 1.f / (1.f + expf(-a));
}

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_conv3d_f16(const pcd_conv3d_desc_t* d, void* stream) {
    PCD_CHECK_ARG(d != nullptr);
    PCD_CHECK_ARG(d->in && d->w && d->out && d->taps && d->zero_page);
    PCD_CHECK_ARG(d->batch > 0 && d->in_d > 0 && d->in_h > 0 && d->in_w > 0);
    PCD_CHECK_ARG(d->cin >= 8 && (d->cin & (d->cin - 1)) == 0);
    PCD_CHECK_ARG(d->cout > 0 && d->cout % 8 == 0);
    PCD_CHECK_ARG(d->ntaps > 0 && d->kpad % CBK == 0 && d->kpad >= d->ntaps * d->cin);
    PCD_CHECK_ARG(d->rows_d > 0 && d->rows_h > 0 && d->rows_w > 0 && d->stride > 0);
    PCD_CHECK_ARG(d->out_scale > 0 && d->out_d > 0 && d->out_h > 0 && d->out_w > 0);
    ConvParams p{};
    p.in = (const half_t*)d->in; p.D = d->in_d; p.H = d->in_h; p.W = d->in_w; p.Cin = d->cin;
    p.cin_shift = __builtin_ctz((unsigned)d->cin);
    p.Do = d->rows_d; p.Ho = d->rows_h; p.Wo = d->rows_w; p.stride = d->stride;
    p.ntaps = d->ntaps; p.kpad = d->kpad; p.taps = d->taps;
    p.w = (const half_t*)d->w; p.bias = d->bias; p.resid = (const half_t*)d->resid;
    p.out = (half_t*)d->out; p.Cout = d->cout;
    p.OD = d->out_d; p.OH = d->out_h; p.OW = d->out_w; p.os = d->out_scale;
    p.pz = d->out_off_z; p.py = d->out_off_y; p.px = d->out_off_x;
    p.relu = d->relu;
    const int64_t m = (int64_t)d->batch * d->rows_d * d->rows_h * d->rows_w;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    p.M = (int)m;
    p.zero = (const half_t*)d->zero_page;
    hipStream_t s = (hipStream_t)stream;
    if (d->cout <= 64) {
        p.tiles_n = (int)ceil_div(d->cout, 64);
        hipLaunchKernelGGL((conv3d_igemm_kernel<128, 64>), dim3((unsigned)(ceil_div(m, 128) * p.tiles_n)), dim3(256), 0, s, p);
    } else {
        p.tiles_n = (int)ceil_div(d->cout, 128);
        hipLaunchKernelGGL((conv3d_igemm_kernel<128, 128>), dim3((unsigned)(ceil_div(m, 128) * p.tiles_n)), dim3(256), 0, s, p);
    }
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv3d_first(const float* x, int batch, int d, int h, int w, int stride, const float* wgt,
                                const float* bias, int cout, void* out, void* stream) {
    PCD_CHECK_ARG(x && wgt && bias && out && batch > 0 && d > 0 && h > 0 && w > 0 && cout > 0 && cout % 8 == 0);
    PCD_CHECK_ARG(stride == 1 || stride == 2);
    const int64_t ovox = (int64_t)((d - 1) / stride + 1) * ((h - 1) / stride + 1) * ((w - 1) / stride + 1);
    const int64_t total = (int64_t)batch * ovox * (cout / 8);
    hipLaunchKernelGGL(conv3d_first_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256),
                       (size_t)cout * 28 * sizeof(float), (hipStream_t)stream, x, batch, d, h, w, stride, wgt, bias,
                       cout, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_conv3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                                       float bias, float* out, void* stream) {
    PCD_CHECK_ARG(in && wgt && out && batch > 0 && d > 0 && h > 0 && w > 0);
    PCD_CHECK_ARG(cin == 32);
    const int64_t total = (int64_t)batch * d * h * w;
    hipLaunchKernelGGL((conv3d_last_kernel<32>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)in, batch, d, h, w, wgt, bias, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_convt3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                                        float bias, float* out, void* stream) {
    PCD_CHECK_ARG(in && wgt && out && batch > 0 && d > 0 && h > 0 && w > 0);
    PCD_CHECK_ARG(cin == 32);
    const int64_t total = (int64_t)batch * d * h * w * 8;
    hipLaunchKernelGGL((convT3d_last_kernel<32>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)in, batch, d, h, w, wgt, bias, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
