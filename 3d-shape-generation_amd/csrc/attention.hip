// K6/K7: set attention over the N points of each shape (reference networks.py:51-83,
// nn.MultiheadAttention explicit bmm/softmax/bmm path) as a flash-style fused kernel:
// the N x N score matrix is never written.
//
// Formulation (everything K-major for v_mfma_f32_32x32x16_f16):
//   S^T[key][query] = sum_d K[key][d] Q[query][d]       A = K rows,  B = Q rows
//   O^T[dd][query]  = sum_key Vt[dd][key] P^T[key][query]  A = Vt rows, B = P^T straight from
//                                                           the S^T accumulators (no LDS trip)
// With queries on the MFMA lane index, the softmax row reduction is 15 in-lane max/adds plus
// one exchange between lane l and l+32, and the O^T rescale is a per-lane scalar.
// V is transposed once per call into Vt[b][h][dd][n] (keys contiguous) by a small kernel.
#include "common.h"

namespace pcd {

// ---------------------------------------------------------------- LayerNorm
// one wave per row; fp32 statistics, biased variance, eps inside the sqrt (torch semantics)
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, int64_t rows, int c,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         half_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const half_t* xr = x + row * c;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += (float)xr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)c;
    float v = 0.f;
    for (int i = lane; i < c; i += 64) { const float d = (float)xr[i] - mean; v += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = rsqrtf(v / (float)c + 1e-5f);
    half_t* orow = out + row * c;
    for (int i = lane; i < c; i += 64) orow[i] = to_half_sat(((float)xr[i] - mean) * rstd * gamma[i] + beta[i]);
}

// ---------------------------------------------------------- V -> Vt transpose
// qkv [B*N][3C]; Vt [B][H][d][Npad] with Npad = N rounded up to 64 (pad keys are zero).
__global__ __launch_bounds__(256) void v_transpose_kernel(const half_t* __restrict__ qkv, int n, int npad, int c,
                                                           half_t* __restrict__ vt) {
    extern __shared__ __attribute__((aligned(16))) half_t tile[];   // [64 keys][c + 2]
    const int b = blockIdx.y, k0 = blockIdx.x * 64;
    const int ldt = c + 2;
    for (int i = threadIdx.x; i < 64 * c; i += blockDim.x) {
        const int kr = i / c, cc = i - kr * c;
        const int key = k0 + kr;
        tile[kr * ldt + cc] = key < n ? qkv[((int64_t)b * n + key) * (3 * c) + 2 * c + cc] : (half_t)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * c; i += blockDim.x) {
        const int cc = i >> 6, kr = i & 63;
        vt[((int64_t)b * c + cc) * npad + k0 + kr] = tile[kr * ldt + cc];
    }
}

// ------------------------------------------------------------ flash attention
// block = 4 waves; each wave owns 32 queries of one (shape, head); the block shares K / Vt
// tiles of KT keys through LDS.
constexpr int KT = 64;

template <int D>
__global__ __launch_bounds__(256) void set_attention_kernel(const half_t* __restrict__ qkv,
                                                             const half_t* __restrict__ vt, int n, int npad, int c,
                                                             int heads, float scale_log2e, half_t* __restrict__ out) {
    constexpr int DP = (D < 32) ? 32 : D;        // O^T rows padded to the 32-row MFMA tile
    constexpr int KSTEPS = D / 16;               // MFMA k-steps of the S^T product
    constexpr int OT = DP / 32;                  // 32-row O^T tiles
    constexpr int KLD = D + 8;                   // halfs per staged K row (pad breaks the power-of-2 stride)
    constexpr int VLD = KT + 8;                  // halfs per staged Vt row
    __shared__ __attribute__((aligned(16))) half_t ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) half_t vs[DP * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qr = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, head = bh - b * heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int64_t row_base = (int64_t)b * n;
    const int ld = 3 * c;

    // Q fragments (B operand): lane (query qr, half hh) holds Q[query][16*s + 8*hh + j]
    half8 qf[KSTEPS];
    {
        int qi = q0 + qr;
        qi = qi < n ? qi : n - 1;
        const half_t* qp = qkv + (row_base + qi) * ld + head * D;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *(const half8*)(qp + 16 * s + 8 * hh);
    }

    f32x16 oacc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (D < 32) {  // zero the padded Vt rows once
        for (int i = tid; i < (DP - D) * VLD; i += 256) vs[D * VLD + i] = (half_t)0.f;
    }

    const half_t* kbase = qkv + row_base * ld + c + head * D;
    const half_t* vbase = vt + ((int64_t)b * c + head * D) * npad;

    for (int k0 = 0; k0 < n; k0 += KT) {
        __syncthreads();
        // stage K tile [KT][D] and Vt tile [D][KT] (16-B pieces)
        for (int i = tid; i < KT * (D / 8); i += 256) {
            const int kr = i / (D / 8), ch = i - kr * (D / 8);
            int key = k0 + kr;
            key = key < n ? key : n - 1;
            *(half8*)(ks + kr * KLD + ch * 8) = *(const half8*)(kbase + (int64_t)key * ld + ch * 8);
        }
        for (int i = tid; i < D * (KT / 8); i += 256) {
            const int dr = i / (KT / 8), ch = i - dr * (KT / 8);
            *(half8*)(vs + dr * VLD + ch * 8) = *(const half8*)(vbase + (int64_t)dr * npad + k0 + ch * 8);
        }
        __syncthreads();

#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            // S^T tile: 32 keys x 32 queries
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const half8 kf = *(const half8*)(ks + (sub * 32 + qr) * KLD + 16 * s + 8 * hh);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
            }
            // register r of lane (qr, hh) is key (r&3) + 8*(r>>2) + 4*hh of this sub-tile
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                sacc[r] = key < n ? sacc[r] * scale_log2e : -INFINITY;
                mx = fmaxf(mx, sacc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = exp2f(sacc[r] - m_new);
                psum += sacc[r];
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
            // P^T as B operand: k-step s2 takes registers 8*s2 .. 8*s2+7; element j of lane half hh
            // is key 16*s2 + 8*(j>>2) + 4*hh + (j&3), so the Vt fragment gathers the same keys.
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[8 * s2 + j];
#pragma unroll
                for (int t = 0; t < OT; ++t) {
                    const half_t* vrow = vs + (t * 32 + qr) * VLD + sub * 32 + 16 * s2 + 4 * hh;
                    const half4 lo = *(const half4*)(vrow);
                    const half4 hi = *(const half4*)(vrow + 8);
                    half8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[t], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: lane (query qr, half hh) holds O^T rows dd = t*32 + (r&3) + 8*(r>>2) + 4*hh
    const int qi = q0 + qr;
    if (qi < n) {
        const float inv = 1.f / l_run;
        half_t* orow = out + (row_base + qi) * c + head * D;
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = t * 32 + 8 * g + 4 * hh;
                if (dd < D) {
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_half_sat(oacc[t][4 * g + e] * inv);
                    *(half4*)(orow + dd) = o;
                }
            }
    }
}

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_layernorm_f16(const void* x, int64_t rows, int c, const float* gamma, const float* beta,
                                 void* out, void* stream) {
    PCD_CHECK_ARG(x && gamma && beta && out && rows > 0 && c > 0);
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, rows, c, gamma, beta, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" size_t pcd_set_attention_workspace_bytes(int batch, int n_points, int c) {
    if (batch <= 0 || n_points <= 0 || c <= 0) return 0;
    const size_t npad = (size_t)ceil_div(n_points, KT) * KT;
    return (size_t)batch * c * npad * sizeof(half_t);
}

extern "C" int pcd_set_attention_f16(const void* qkv, int batch, int n_points, int c, int heads, void* out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(qkv && out && workspace && batch > 0 && n_points > 0 && heads > 0 && c % heads == 0);
    const int d = c / heads;
    PCD_CHECK_ARG(d == 16 || d == 32 || d == 64);
    PCD_CHECK_ARG(c % 8 == 0);
    const size_t need = pcd_set_attention_workspace_bytes(batch, n_points, c);
    if (workspace_bytes < need) {
        set_error("pcd_set_attention_f16: workspace %zu < required %zu", workspace_bytes, need);
        return PCD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int npad = (int)(ceil_div(n_points, KT) * KT);
    half_t* vt = (half_t*)workspace;
    hipLaunchKernelGGL(v_transpose_kernel, dim3(npad / 64, batch), dim3(256), (size_t)64 * (c + 2) * sizeof(half_t), s,
                       (const half_t*)qkv, n_points, npad, c, vt);
    PCD_CHECK_LAUNCH();
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)d);
    dim3 grid((unsigned)ceil_div(n_points, 128), (unsigned)(batch * heads));
    if (d == 16)
        hipLaunchKernelGGL((set_attention_kernel<16>), grid, dim3(256), 0, s, (const half_t*)qkv, vt, n_points, npad, c,
                           heads, scale_log2e, (half_t*)out);
    else if (d == 32)
        hipLaunchKernelGGL((set_attention_kernel<32>), grid, dim3(256), 0, s, (const half_t*)qkv, vt, n_points, npad, c,
                           heads, scale_log2e, (half_t*)out);
    else
        hipLaunchKernelGGL((set_attention_kernel<64>), grid, dim3(256), 0, s, (const half_t*)qkv, vt, n_points, npad, c,
                           heads, scale_log2e, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
