// Elementwise / small kernels of the sampler path (all HBM- or latency-bound):
//   time embedding + time_mlp + hoisted enc1 time bias (K3), enc1 xyz half (K=3),
//   output head (C=3), add/remove noise and DDIM/DDPM updates (K4), Philox normals (K5),
//   fp32<->fp16 converts.
#include "common.h"

namespace pcd {

// ------------------------------------------------------------------ time path
// One block per time value.  fp32 throughout (this path defines per-step constants).
__global__ __launch_bounds__(256) void time_embed_kernel(
    const float* __restrict__ t, const float* __restrict__ freqs, int time_dim, int dim,
    const float* __restrict__ w0, const float* __restrict__ b0,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ temb,
    const float* __restrict__ e1w_t, const float* __restrict__ e1b, int c1, float* __restrict__ tbias) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* emb = sm;                 // [time_dim]
    float* hid = sm + time_dim;      // [dim]
    float* out = hid + dim;          // [dim]
    const int i = blockIdx.x;
    const float tv = t[i];
    const int half = time_dim / 2;
    for (int j = threadIdx.x; j < time_dim; j += blockDim.x) {
        float v = 0.f;
        if (j < half) v = sinf(tv * freqs[j]);
        else if (j < 2 * half) v = cosf(tv * freqs[j - half]);
        emb[j] = v;  // odd time_dim: trailing zero pad (networks.py:836-837)
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        const float* wr = w0 + (int64_t)c * time_dim;
        float a = 0.f;
        for (int k = 0; k < time_dim; ++k) a = fmaf(wr[k], emb[k], a);
        a += b0[c];
        hid[c] = a / (1.f + expf(-a));  // SiLU
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        const float* wr = w2 + (int64_t)c * dim;
        float a = 0.f;
        for (int k = 0; k < dim; ++k) a = fmaf(wr[k], hid[k], a);
        a += b2[c];
        out[c] = a;
        if (temb != nullptr) temb[(int64_t)i * dim + c] = a;
    }
    if (e1w_t == nullptr) return;
    __syncthreads();
    for (int c = threadIdx.x; c < c1; c += blockDim.x) {
        const float* wr = e1w_t + (int64_t)c * dim;
        float a = 0.f;
        for (int k = 0; k < dim; ++k) a = fmaf(wr[k], out[k], a);
        tbias[(int64_t)i * c1 + c] = a + e1b[c];
    }
}

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int rows, int k,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          int c, float* __restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * c) return;
    const int r = (int)(idx / c), cc = (int)(idx - (int64_t)r * c);
    const float* xr = x + (int64_t)r * k;
    const float* wr = w + (int64_t)cc * k;
    float a = 0.f;
    for (int j = 0; j < k; ++j) a = fmaf(wr[j], xr[j], a);
    y[idx] = a + (b != nullptr ? b[cc] : 0.f);
}

// ------------------------------------------------------ enc1.conv1, xyz half
// thread = (point, 8-channel chunk): 16-B coalesced fp16 stores, x broadcast within 8 lanes
__global__ __launch_bounds__(256) void enc1_xyz_kernel(const float* __restrict__ x, int64_t m, int rows_per_shape,
                                                        const float* __restrict__ w, int c1,
                                                        const float* __restrict__ tbias, int tb_stride,
                                                        half_t* __restrict__ out) {
    const int chunks = c1 / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * chunks) return;
    const int64_t pt = idx / chunks;
    const int ch = (int)(idx - pt * chunks);
    const float px = x[pt * 3 + 0], py = x[pt * 3 + 1], pz = x[pt * 3 + 2];
    const float* tb = tbias + (int64_t)(pt / rows_per_shape) * tb_stride * c1 + ch * 8;
    const float* wr = w + (int64_t)ch * 8 * 3;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = tb[e];
        v = fmaf(wr[e * 3 + 0], px, v);
        v = fmaf(wr[e * 3 + 1], py, v);
        v = fmaf(wr[e * 3 + 2], pz, v);
        o[e] = to_half_sat(fmaxf(v, 0.f));
    }
    *(half8*)(out + pt * c1 + ch * 8) = o;
}

// ---------------------------------------------------------------- output head
// eps[m][0..2] = W3 . h[m] + b3; one thread per point, h row read as 16-B pieces
template <int K>
__global__ __launch_bounds__(256) void head3_kernel(const half_t* __restrict__ h, int64_t m,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ eps) {
    __shared__ float ws[3 * K];
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const half8* row = (const half8*)(h + pt * K);
#pragma unroll
    for (int cidx = 0; cidx < K / 8; ++cidx) {
        const half8 v = row[cidx];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            a0 = fmaf(ws[cidx * 8 + e], f, a0);
            a1 = fmaf(ws[K + cidx * 8 + e], f, a1);
            a2 = fmaf(ws[2 * K + cidx * 8 + e], f, a2);
        }
    }
    eps[pt * 3 + 0] = a0 + b[0];
    eps[pt * 3 + 1] = a1 + b[1];
    eps[pt * 3 + 2] = a2 + b[2];
}

// ------------------------------------------------------------ diffusion updates
// Expression order follows the reference's torch-CPU ops (no FMA contraction) so that,
// given the same eps, results are bit-identical to diffusion.py:151,167,255,287.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                         const float* __restrict__ n, const float* __restrict__ s,
                                                         int stride, int64_t total, int64_t per_shape,
                                                         float* __restrict__ xt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = (i / per_shape) * stride;
    const float a = s[b] * x0[i];
    const float c = n[b] * noise[i];
    xt[i] = a + c;
}

__global__ __launch_bounds__(256) void remove_noise_kernel(const float* __restrict__ xt, const float* __restrict__ eps,
                                                            const float* __restrict__ n, const float* __restrict__ s,
                                                            int stride, int64_t total, int64_t per_shape,
                                                            float* __restrict__ x0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = (i / per_shape) * stride;
    const float ne = n[b] * eps[i];
    x0[i] = (xt[i] - ne) / s[b];
}

__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                           const float* __restrict__ n, const float* __restrict__ s,
                                                           const float* __restrict__ n2, const float* __restrict__ s2,
                                                           int stride, int64_t total, int64_t per_shape,
                                                           float* __restrict__ x0o, float* __restrict__ xn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = (i / per_shape) * stride;
    const float e = eps[i];
    const float ne = n[b] * e;
    const float x0 = (x[i] - ne) / s[b];
    if (x0o != nullptr) x0o[i] = x0;
    if (xn != nullptr) {
        const float a = s2[b] * x0;
        const float c = n2[b] * e;
        xn[i] = a + c;
    }
}

// one element of the DDPM update (diffusion.py:246-255), shared by ddpm_update_kernel and the Philox-fused form below
__device__ __forceinline__ void ddpm_elem(float xv, float ev, float zv, float nb, float sb, float cb, float s2b, bool next, float& x0, float& xn) {
    const float ne = nb * ev;
    x0 = (xv - ne) / sb;
    xn = 0.f;
    if (next) {
        const float a = s2b * x0;
        const float cn = cb * nb;          // (coefficient * noise_rates) * noise, diffusion.py:255
        const float c = cn * zv;
        xn = a + c;
    }
}

__global__ __launch_bounds__(256) void ddpm_update_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                           const float* __restrict__ z, const float* __restrict__ n,
                                                           const float* __restrict__ s, const float* __restrict__ coef,
                                                           const float* __restrict__ s2, int stride, int64_t total,
                                                           int64_t per_shape, float* __restrict__ x0o,
                                                           float* __restrict__ xn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = (i / per_shape) * stride;
    const float ne = n[b] * eps[i];
    const float x0 = (x[i] - ne) / s[b];
    if (x0o != nullptr) x0o[i] = x0;
    if (xn != nullptr) {
        const float a = s2[b] * x0;
        const float cn = coef[b] * n[b];   // (coefficient * noise_rates) * noise, diffusion.py:255
        const float c = cn * z[i];
        xn[i] = a + c;
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------- Philox
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 4 outputs per thread
    if (q * 4 >= n) return;
    const uint64_t ctr = (uint64_t)q + offset;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        v[2 * h] = r * cs;
        v[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q * 4 + e < n) out[q * 4 + e] = v[e];
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ s, half_t* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = to_half_sat(s[i]);
}
__global__ __launch_bounds__(256) void f16_to_f32_kernel(const half_t* __restrict__ s, float* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}

// x[m][c] += e[m / rows_per_shape][c]   (additive per-level time embeddings, networks.py:669-698)
__global__ __launch_bounds__(256) void add_shape_bias_kernel(const half_t* __restrict__ x, int64_t m, int c,
                                                              int rows_per_shape, const float* __restrict__ e,
                                                              int64_t e_stride, half_t* __restrict__ out) {
    const int chunks = c / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * chunks) return;
    const int64_t row = idx / chunks;
    const int ch = (int)(idx - row * chunks);
    const float* er = e + (row / rows_per_shape) * e_stride + ch * 8;
    half8 v = *(const half8*)(x + row * c + ch * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = to_half_sat((float)v[i] + er[i]);
    *(half8*)(out + row * c + ch * 8) = v;
}

// Tail of UNetAttentionPointExperimental (networks.py:647-650,700-702): dec1 = PointNetLayer(128,3,3)
// on cat[a (ka ch) | b (kb ch)] followed by output = Conv1d(3,3).  BN folded; fp32 math per point.
// w1 [3][ka+kb], w2/w3/w4 [3][3], biases [3].
__global__ __launch_bounds__(256) void tail3_kernel(const half_t* __restrict__ a, int ka, const half_t* __restrict__ b,
                                                     int kb, int64_t m, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w234,
                                                     const float* __restrict__ b234, float* __restrict__ out) {
    extern __shared__ float ws[];   // w1 [3][ka+kb]
    const int k = ka + kb;
    for (int i = threadIdx.x; i < 3 * k; i += blockDim.x) ws[i] = w1[i];
    __syncthreads();
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    float h[3] = {b1[0], b1[1], b1[2]};
    for (int cidx = 0; cidx < ka / 8; ++cidx) {
        const half8 v = *(const half8*)(a + pt * ka + cidx * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
#pragma unroll
            for (int j = 0; j < 3; ++j) h[j] = fmaf(ws[j * k + cidx * 8 + e], f, h[j]);
        }
    }
    for (int cidx = 0; cidx < kb / 8; ++cidx) {
        const half8 v = *(const half8*)(b + pt * kb + cidx * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
#pragma unroll
            for (int j = 0; j < 3; ++j) h[j] = fmaf(ws[j * k + ka + cidx * 8 + e], f, h[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) h[j] = fmaxf(h[j], 0.f);
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        float o[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = b234[layer * 3 + j];
#pragma unroll
            for (int i = 0; i < 3; ++i) acc = fmaf(w234[layer * 9 + j * 3 + i], h[i], acc);
            o[j] = layer < 2 ? fmaxf(acc, 0.f) : acc;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) h[j] = o[j];
    }
    out[pt * 3 + 0] = h[0]; out[pt * 3 + 1] = h[1]; out[pt * 3 + 2] = h[2];
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                       const float* __restrict__ eps, float* __restrict__ z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sd = expf(0.5f * lv[i]);
    const float e = eps[i] * sd;
    z[i] = mu[i] + e;
}
#pragma clang fp contract(fast)

// Device-side step selection so that ONE captured HIP graph can be replayed for every timestep:
// k = counter[0]; copy row k of the time-bias table and column k of the four rate tables to fixed
// buffers; counter[1] = k (for consumers later in the step); counter[0] = k + 1.
__global__ __launch_bounds__(256) void step_select_kernel(int* __restrict__ counter, int n_steps,
                                                           const float* __restrict__ tb_table, int tb_elems,
                                                           float* __restrict__ tb_cur,
                                                           const float* __restrict__ rate_tables, int width,
                                                           float* __restrict__ rates_cur) {
    int k = counter[0];
    k = k < n_steps ? k : n_steps - 1;
    for (int i = threadIdx.x; i < tb_elems; i += blockDim.x) tb_cur[i] = tb_table[(int64_t)k * tb_elems + i];
    for (int i = threadIdx.x; i < 4 * width; i += blockDim.x) {
        const int tbl = i / width, j = i - tbl * width;
        rates_cur[i] = rate_tables[((int64_t)tbl * n_steps + k) * width + j];
    }
    __syncthreads();
    if (threadIdx.x == 0) { counter[1] = k; counter[0] = k + 1; }
}

__global__ __launch_bounds__(256) void randn_step_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                          uint64_t base, uint64_t stride,
                                                          const int* __restrict__ counter) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    const uint64_t ctr = (uint64_t)q + base + stride * (uint64_t)counter[1];
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        v[2 * h] = r * cs;
        v[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q * 4 + e < n) out[q * 4 + e] = v[e];
}

static inline unsigned nblk(int64_t n) { return (unsigned)ceil_div(n, 256); }

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_time_embed(const float* t, int n_t, const float* freqs, int time_dim, int dim,
                              const float* w0, const float* b0, const float* w2, const float* b2,
                              float* temb, const float* e1w_t, const float* e1b, int c1, float* tbias,
                              void* stream) {
    PCD_CHECK_ARG(t && freqs && w0 && b0 && w2 && b2);
    PCD_CHECK_ARG(n_t > 0 && time_dim >= 2 && dim > 0);
    PCD_CHECK_ARG(temb != nullptr || e1w_t != nullptr);
    PCD_CHECK_ARG(e1w_t == nullptr || (e1b != nullptr && tbias != nullptr && c1 > 0));
    const size_t sm = (size_t)(time_dim + 2 * dim) * sizeof(float);
    hipLaunchKernelGGL(time_embed_kernel, dim3(n_t), dim3(256), sm, (hipStream_t)stream, t, freqs, time_dim, dim,
                       w0, b0, w2, b2, temb, e1w_t, e1b, c1, tbias);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_linear_f32(const float* x, int rows, int k, const float* w, const float* b, int c,
                              float* y, void* stream) {
    PCD_CHECK_ARG(x && w && y && rows > 0 && k > 0 && c > 0);
    hipLaunchKernelGGL(linear_f32_kernel, dim3(nblk((int64_t)rows * c)), dim3(256), 0, (hipStream_t)stream,
                       x, rows, k, w, b, c, y);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_enc1_xyz(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, int c1,
                            const float* tbias, int tbias_shape_stride, void* out, void* stream) {
    PCD_CHECK_ARG(x && w_xyz && tbias && out);
    PCD_CHECK_ARG(m > 0 && rows_per_shape > 0 && c1 > 0 && c1 % 8 == 0);
    PCD_CHECK_ARG(tbias_shape_stride >= 0);
    hipLaunchKernelGGL(enc1_xyz_kernel, dim3(nblk(m * (c1 / 8))), dim3(256), 0, (hipStream_t)stream,
                       x, m, rows_per_shape, w_xyz, c1, tbias, tbias_shape_stride, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_head3(const void* h, int64_t m, int k, const float* w, const float* b, float* eps, void* stream) {
    PCD_CHECK_ARG(h && w && b && eps && m > 0);
    PCD_CHECK_ARG(k == 64);
    hipLaunchKernelGGL((head3_kernel<64>), dim3(nblk(m)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)h, m, w, b, eps);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_add_noise(const float* x0, const float* noise, const float* n, const float* s, int stride,
                             int64_t total, int64_t per_shape, float* x_t, void* stream) {
    PCD_CHECK_ARG(x0 && noise && n && s && x_t && total > 0 && per_shape > 0 && (stride == 0 || stride == 1));
    hipLaunchKernelGGL(add_noise_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream,
                       x0, noise, n, s, stride, total, per_shape, x_t);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_remove_noise(const float* x_t, const float* eps, const float* n, const float* s, int stride,
                                int64_t total, int64_t per_shape, float* x0, void* stream) {
    PCD_CHECK_ARG(x_t && eps && n && s && x0 && total > 0 && per_shape > 0 && (stride == 0 || stride == 1));
    hipLaunchKernelGGL(remove_noise_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream,
                       x_t, eps, n, s, stride, total, per_shape, x0);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_ddim_update(const float* x, const float* eps, const float* n, const float* s,
                               const float* n2, const float* s2, int stride, int64_t total, int64_t per_shape,
                               float* x0, float* x_next, void* stream) {
    PCD_CHECK_ARG(x && eps && n && s && total > 0 && per_shape > 0 && (stride == 0 || stride == 1));
    PCD_CHECK_ARG(x0 != nullptr || x_next != nullptr);
    PCD_CHECK_ARG(x_next == nullptr || (n2 != nullptr && s2 != nullptr));
    hipLaunchKernelGGL(ddim_update_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream,
                       x, eps, n, s, n2, s2, stride, total, per_shape, x0, x_next);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_ddpm_update(const float* x, const float* eps, const float* z, const float* n, const float* s,
                               const float* coef, const float* s2, int stride, int64_t total, int64_t per_shape,
                               float* x0, float* x_next, void* stream) {
    PCD_CHECK_ARG(x && eps && n && s && total > 0 && per_shape > 0 && (stride == 0 || stride == 1));
    PCD_CHECK_ARG(x0 != nullptr || x_next != nullptr);
    PCD_CHECK_ARG(x_next == nullptr || (z != nullptr && coef != nullptr && s2 != nullptr));
    hipLaunchKernelGGL(ddpm_update_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream,
                       x, eps, z, n, s, coef, s2, stride, total, per_shape, x0, x_next);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    PCD_CHECK_ARG(out && n > 0);
    hipLaunchKernelGGL(randn_kernel, dim3(nblk(ceil_div(n, 4))), dim3(256), 0, (hipStream_t)stream,
                       out, n, seed, offset);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_fill_zero(void* p, size_t bytes, void* stream) {
    PCD_CHECK_ARG(p != nullptr);
    if (bytes == 0) return PCD_OK;
    PCD_CHECK_HIP(hipMemsetAsync(p, 0, bytes, (hipStream_t)stream));
    return PCD_OK;
}

extern "C" int pcd_f32_to_f16(const float* src, void* dst, int64_t n, void* stream) {
    PCD_CHECK_ARG(src && dst && n > 0);
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, src, (half_t*)dst, n);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_f16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    PCD_CHECK_ARG(src && dst && n > 0);
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)src, dst, n);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_add_shape_bias_strided_f16(const void* x, int64_t m, int c, int rows_per_shape, const float* e,
                                              int64_t e_stride, void* out, void* stream) {
    PCD_CHECK_ARG(x && e && out && m > 0 && c > 0 && c % 8 == 0 && rows_per_shape > 0 && e_stride >= 0);
    hipLaunchKernelGGL(add_shape_bias_kernel, dim3(nblk(m * (c / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, m, c, rows_per_shape, e, e_stride, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_add_shape_bias_f16(const void* x, int64_t m, int c, int rows_per_shape, const float* e, void* out,
                                      void* stream) {
    return pcd_add_shape_bias_strided_f16(x, m, c, rows_per_shape, e, c, out, stream);
}

extern "C" int pcd_tail3(const void* a, int ka, const void* b, int kb, int64_t m, const float* w1, const float* b1,
                         const float* w234, const float* b234, float* out, void* stream) {
    PCD_CHECK_ARG(a && b && w1 && b1 && w234 && b234 && out && m > 0);
    PCD_CHECK_ARG(ka > 0 && kb > 0 && ka % 8 == 0 && kb % 8 == 0 && ka + kb <= 4096);
    hipLaunchKernelGGL(tail3_kernel, dim3(nblk(m)), dim3(256), (size_t)3 * (ka + kb) * sizeof(float),
                       (hipStream_t)stream, (const half_t*)a, ka, (const half_t*)b, kb, m, w1, b1, w234, b234, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, int64_t n,
                                  void* stream) {
    PCD_CHECK_ARG(mu && logvar && eps && z && n > 0);
    hipLaunchKernelGGL(reparam_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, n);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// DDPM update with its normal draw generated in place: thread = 4 consecutive elements = one Philox counter, the counter layout and
// the Box-Muller arithmetic of randn_step_kernel, the update arithmetic of ddpm_update_kernel (ddpm_elem): bitwise the two launches
__global__ __launch_bounds__(256) void ddpm_update_philox_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                  const float* __restrict__ n, const float* __restrict__ s,
                                                                  const float* __restrict__ coef, const float* __restrict__ s2,
                                                                  int stride, int64_t total, int64_t per_shape,
                                                                  float* __restrict__ x0o, float* __restrict__ xn, uint64_t seed,
                                                                  uint64_t base, uint64_t pstride, const int* __restrict__ counter) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= total) return;
    const uint64_t ctr = (uint64_t)q + base + pstride * (uint64_t)counter[1];
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        v[2 * h] = r * cs;
        v[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = q * 4 + e;
        if (i >= total) break;
        const int64_t b = (i / per_shape) * stride;
        float x0, xv;
        ddpm_elem(x[i], eps[i], v[e], n[b], s[b], coef[b], s2[b], xn != nullptr, x0, xv);
        if (x0o != nullptr) x0o[i] = x0;
        if (xn != nullptr) xn[i] = xv;
    }
}

extern "C" int pcd_step_select(int* counter, int n_steps, const float* tb_table, int tb_elems, float* tb_cur,
                               const float* rate_tables, int width, float* rates_cur, void* stream) {
    PCD_CHECK_ARG(counter && tb_table && tb_cur && rate_tables && rates_cur && n_steps > 0 && tb_elems > 0 && width > 0);
    hipLaunchKernelGGL(step_select_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counter, n_steps, tb_table,
                       tb_elems, tb_cur, rate_tables, width, rates_cur);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_ddpm_update_philox(const float* x, const float* eps, const float* n, const float* s, const float* coef,
                                      const float* s2, int stride, int64_t total, int64_t per_shape, float* x0, float* x_next,
                                      uint64_t seed, uint64_t base_offset, uint64_t per_step_stride, const int* counter, void* stream) {
    PCD_CHECK_ARG(x && eps && n && s && coef && s2 && counter && total > 0 && per_shape > 0 && (stride == 0 || stride == 1));
    PCD_CHECK_ARG(x0 != nullptr || x_next != nullptr);
    hipLaunchKernelGGL(ddpm_update_philox_kernel, dim3(nblk(ceil_div(total, 4))), dim3(256), 0, (hipStream_t)stream, x, eps, n, s, coef, s2,
                       stride, total, per_shape, x0, x_next, seed, base_offset, per_step_stride, counter);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_randn_step(float* out, int64_t n, uint64_t seed, uint64_t base_offset, uint64_t per_step_stride,
                              const int* counter, void* stream) {
    PCD_CHECK_ARG(out && counter && n > 0);
    hipLaunchKernelGGL(randn_step_kernel, dim3(nblk(ceil_div(n, 4))), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                       base_offset, per_step_stride, counter);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
