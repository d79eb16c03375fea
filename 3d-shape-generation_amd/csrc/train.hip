// Training-step kernels of the point denoiser (reference diffusion.py:70-86,170-186:
// add_noise -> model in train() mode -> L1 loss -> AdamW).  The dense contractions (forward, backward-data,
// backward-weight) all go through the fp16 MFMA GEMM of gemm_f16.hip; this file holds what surrounds them:
// BatchNorm1d with batch statistics (networks.py:31-48) forward and backward, the max-pool with its argmax,
// the K=3 / C=3 edge layers, column reductions, transposes, the L1 loss and the AdamW update.
// All of it is HBM-bound elementwise / reduction work: 16-byte loads, fp32 accumulation, one pass per tensor.
//
// Layout: activations and their gradients are point-major fp16 [M][C] like everywhere else in the library;
// the pre-BatchNorm conv outputs z are kept in fp32 (x_hat = (z - mean) * rstd cancels: fp16 z costs ~1e-3
// per layer when |mean| >> std); statistics, parameter gradients and optimizer state are fp32.  Gradients carry the caller's loss scale.
#include "common.h"

namespace pcd {

constexpr int CR_COLS = 64;     // columns per block: 8 threads x 8 halfs
constexpr int CR_LANES = 32;    // row lanes per block

// sum the 32 row lanes of acc[v][0..8) per column and add the block's partial to out[v][col] atomically
template <int NV>
__device__ __forceinline__ void cr_finish(float (&acc)[NV][8], float* const (&out)[NV], int col0, int c) {
    __shared__ float red[NV][CR_LANES][CR_COLS + 1];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[v][rl][cg * 8 + e] = acc[v][e];
    __syncthreads();
    if (threadIdx.x < CR_COLS) {
        const int col = col0 + threadIdx.x;
        if (col < c) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float s = 0.f;
                for (int r = 0; r < CR_LANES; ++r) s += red[v][r][threadIdx.x];
                atomicAdd(out[v] + col, s);
            }
        }
    }
}

// rows [row0, row1) of this block; returns false if the block has nothing to do
__device__ __forceinline__ bool cr_range(int64_t rows_per_group, int rows_per_block, int64_t& row0, int64_t& row1) {
    const int64_t g0 = (int64_t)blockIdx.z * rows_per_group;
    row0 = g0 + (int64_t)blockIdx.y * rows_per_block;
    row1 = row0 + rows_per_block;
    if (row1 > g0 + rows_per_group) row1 = g0 + rows_per_group;
    return row0 < row1;
}

__device__ __forceinline__ half8 ld8(const half_t* p, int col, int c) {
    if (col + 8 <= c) return *(const half8*)(p + col);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col + e < c ? p[col + e] : (half_t)0.f;
    return v;
}
struct f8 { float v[8]; };
__device__ __forceinline__ f8 ldf8(const half_t* p, int col, int c) {
    const half8 h = ld8(p, col, c);
    f8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = (float)h[e];
    return r;
}
__device__ __forceinline__ f8 ldf8(const float* p, int col, int c) {   // col is a multiple of 8: 32-byte aligned rows of c % 8 == 0
    f8 r;
    if (col + 8 <= c) {
        const f32x4 a = *(const f32x4*)(p + col), b = *(const f32x4*)(p + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { r.v[e] = a[e]; r.v[4 + e] = b[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[e] = col + e < c ? p[col + e] : 0.f;
    }
    return r;
}
__device__ __forceinline__ void st8(half_t* p, int col, int c, const half8& v) {
    if (col + 8 <= c) { *(half8*)(p + col) = v; return; }
#pragma unroll
    for (int e = 0; e < 8; ++e) if (col + e < c) p[col + e] = v[e];
}

// out[g][col] += sum over the rows of group g of x[row][col]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t rows_per_group, int c,
                                                      float* __restrict__ out, int rpb) {
    int64_t row0, row1;
    if (!cr_range(rows_per_group, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col0 = blockIdx.x * CR_COLS, col = col0 + cg * 8;
    float acc[1][8] = {};
    if (col < c)
        for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
            const f8 v = ldf8(x + r * c, col, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[0][e] += v.v[e];
        }
    float* const outs[1] = {out + (int64_t)blockIdx.z * c};
    cr_finish<1>(acc, outs, col0, c);
}

// One pass over z for the batch statistics: sums of (z - shift) and (z - shift)^2 with shift[col] = z[0][col].  The
// shift keeps E[d^2] - E[d]^2 free of cancellation (|mean - shift| is a few sigma at most), so a second read of z
// for sum (z - mean)^2 is not needed.
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int64_t m, int c, float* __restrict__ s1,
                                                        float* __restrict__ s2, int rpb) {
    int64_t row0, row1;
    if (!cr_range(m, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col0 = blockIdx.x * CR_COLS, col = col0 + cg * 8;
    float acc[2][8] = {};
    if (col < c) {
        const f8 sh = ldf8(x, col, c);
        for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
            const f8 v = ldf8(x + r * c, col, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v.v[e] - sh.v[e]; acc[0][e] += d; acc[1][e] += d * d; }
        }
    }
    float* const outs[2] = {s1, s2};
    cr_finish<2>(acc, outs, col0, c);
}

__global__ void bn_finalize_kernel(const float* x, const float* s1, const float* s2, int64_t m, int c, float momentum,
                                   float* mean, float* var, float* running_mean, float* running_var) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float d1 = s1[i] / (float)m;
    const float mu = x[i] + d1;                                        // shift = z[0][i]
    const float v = fmaxf(s2[i] / (float)m - d1 * d1, 0.f);
    mean[i] = mu;
    var[i] = v;
    if (running_mean != nullptr) {   // torch: running_var tracks the UNBIASED variance (m/(m-1))
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mu;
        const float unb = m > 1 ? v * (float)m / (float)(m - 1) : v;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * unb;
    }
}

// a = act(gamma * (z - mean) * rstd + beta).  Column strips: a thread keeps the constants of its 8 columns in
// registers and walks rows (the first version reloaded 4 x 8 parameters per 8 elements and ran at ~1/5 of HBM speed).
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int64_t m, int c,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int relu, half_t* __restrict__ out, int rpb) {
    int64_t row0, row1;
    if (!cr_range(m, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col = blockIdx.x * CR_COLS + cg * 8;
    if (col >= c) return;
    float mu[8], sc[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cc = col + e < c ? col + e : c - 1;
        mu[e] = mean[cc]; sc[e] = rsqrtf(var[cc] + eps) * gamma[cc]; be[e] = beta[cc];
    }
    for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
        const f8 v = ldf8(z + r * c, col, c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (v.v[e] - mu[e]) * sc[e] + be[e];
            if (relu) y = fmaxf(y, 0.f);
            o[e] = to_half_sat(y);
        }
        st8(out + r * c, col, c, o);
    }
}

// g = da * [bn(z) > 0];  dbeta[col] += sum g ;  dgamma[col] += sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const half_t* __restrict__ da, const float* __restrict__ z,
                                                             int64_t m, int c, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int relu,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma, int rpb) {
    int64_t row0, row1;
    if (!cr_range(m, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col0 = blockIdx.x * CR_COLS, col = col0 + cg * 8;
    float acc[2][8] = {};
    if (col < c) {
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cc = col + e < c ? col + e : c - 1;
            mu[e] = mean[cc]; rs[e] = rsqrtf(var[cc] + eps); ga[e] = gamma[cc]; be[e] = beta[cc];
        }
        for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
            const f8 zv = ldf8(z + r * c, col, c);
            const half8 gv = ld8(da + r * c, col, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (zv.v[e] - mu[e]) * rs[e];
                const float g = (!relu || xh * ga[e] + be[e] > 0.f) ? (float)gv[e] : 0.f;
                acc[0][e] += g;
                acc[1][e] += g * xh;
            }
        }
    }
    float* const outs[2] = {dbeta, dgamma};
    cr_finish<2>(acc, outs, col0, c);
}

// dz = gamma * rstd * (g - dbeta/m - xhat * dgamma/m), column strips like bn_apply_kernel
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const half_t* __restrict__ da, const float* __restrict__ z,
                                                            int64_t m, int c, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int relu,
                                                            const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                            half_t* __restrict__ dz, int rpb) {
    int64_t row0, row1;
    if (!cr_range(m, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col = blockIdx.x * CR_COLS + cg * 8;
    if (col >= c) return;
    const float inv_m = 1.f / (float)m;
    float mu[8], rs[8], ga[8], be[8], kb[8], kg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cc = col + e < c ? col + e : c - 1;
        mu[e] = mean[cc]; rs[e] = rsqrtf(var[cc] + eps); ga[e] = gamma[cc]; be[e] = beta[cc];
        kb[e] = dbeta[cc] * inv_m; kg[e] = dgamma[cc] * inv_m;
    }
    for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
        const f8 zv = ldf8(z + r * c, col, c);
        const half8 gv = ld8(da + r * c, col, c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = (zv.v[e] - mu[e]) * rs[e];
            const float g = (!relu || xh * ga[e] + be[e] > 0.f) ? (float)gv[e] : 0.f;
            o[e] = to_half_sat(ga[e] * rs[e] * (g - kb[e] - xh * kg[e]));
        }
        st8(dz + r * c, col, c, o);
    }
}

// dst[col][row] = src[row][col]; 64 x 64 tiles through LDS, 16-byte global accesses on both sides (the 2-byte version
// ran at ~40 % of HBM speed and was 1.9 ms of the 12.7 ms training step)
__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ src, int64_t rows, int cols,
                                                         half_t* __restrict__ dst) {
    __shared__ half_t tile[64][72];                       // +8 halfs: 16-byte aligned rows, conflict-light column reads
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;          // 8 x 8-half chunks per row, 32 rows per pass
    const bool vec = (cols % 8 == 0) && (rows % 8 == 0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = pass * 32 + ty;
        const int64_t r = r0 + i;
        const int cc = c0 + tx * 8;
        half8 v;
        if (vec && r < rows && cc + 8 <= cols) v = *(const half8*)(src + r * cols + cc);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (r < rows && cc + e < cols) ? src[r * cols + cc + e] : (half_t)0.f;
        }
        *(half8*)(&tile[i][tx * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = pass * 32 + ty;                     // output row = source column c0 + i
        const int cc = c0 + i;
        const int64_t r = r0 + tx * 8;
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[tx * 8 + e][i];
        if (cc < cols) {
            if (vec && r + 8 <= rows) *(half8*)(dst + (int64_t)cc * rows + r) = v;
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (r + e < rows) dst[(int64_t)cc * rows + r + e] = v[e];
            }
        }
    }
}

// per shape and column: max over the n_points rows and the FIRST row index attaining it (torch.max semantics)
__global__ __launch_bounds__(256) void colmax_argmax_kernel(const half_t* __restrict__ a, int n_points, int c,
                                                             float* __restrict__ mx, int* __restrict__ arg) {
    __shared__ float smax[4][64];
    __shared__ int sarg[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float best = -INFINITY;
    int bi = 0;
    if (col < c) {
        const half_t* p = a + (int64_t)b * n_points * c + col;
        for (int n = part; n < n_points; n += 4) {
            const float v = (float)p[(int64_t)n * c];
            if (v > best) { best = v; bi = n; }
        }
    }
    smax[part][threadIdx.x & 63] = best;
    sarg[part][threadIdx.x & 63] = bi;
    __syncthreads();
    if (part == 0 && col < c) {
        for (int q = 1; q < 4; ++q) {
            const float v = smax[q][threadIdx.x];
            const int i = sarg[q][threadIdx.x];
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        mx[(int64_t)b * c + col] = best;
        arg[(int64_t)b * c + col] = bi;
    }
}

// da[b*n + arg[b][col]][col] = dg[b][col]   (da zero-filled by the caller side of the ABI)
__global__ void maxpool_bwd_kernel(const float* __restrict__ dg, const int* __restrict__ arg, int n_points, int c,
                                   int batch, half_t* __restrict__ da) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)batch * c) return;
    const int b = (int)(i / c), col = (int)(i - (int64_t)b * c);
    da[((int64_t)b * n_points + arg[i]) * c + col] = to_half_sat(dg[i]);
}

// z0[m][c] = sum_j x[m][j] w[c][j] + tbias[m / n_points][c]      (enc1.conv1 before its BatchNorm)
__global__ void enc1_linear_kernel(const float* __restrict__ x, int64_t m, int n_points, const float* __restrict__ w,
                                   int c, const float* __restrict__ tbias, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * c) return;
    const int64_t row = i / c;
    const int col = (int)(i - row * c);
    const float* xr = x + row * 3;
    out[i] = xr[0] * w[col * 3] + xr[1] * w[col * 3 + 1] + xr[2] * w[col * 3 + 2] + tbias[(row / n_points) * c + col];
}

// out[j][k] += sum_m vec[m][j] * mat[m][k]  (j < 3)  and  vsum[j] += sum_m vec[m][j]
__global__ __launch_bounds__(256) void vec3_outer_kernel(const half_t* __restrict__ mat, const float* __restrict__ vec,
                                                          int64_t m, int k, float* __restrict__ out, float* __restrict__ vsum, int rpb) {
    int64_t row0, row1;
    if (!cr_range(m, rpb, row0, row1)) return;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col0 = blockIdx.x * CR_COLS, col = col0 + cg * 8;
    float acc[3][8] = {};
    float vs[3] = {0.f, 0.f, 0.f};
    for (int64_t r = row0 + rl; r < row1; r += CR_LANES) {
        const float v0 = vec[r * 3], v1 = vec[r * 3 + 1], v2 = vec[r * 3 + 2];
        if (blockIdx.x == 0 && cg == 0) { vs[0] += v0; vs[1] += v1; vs[2] += v2; }
        if (col < k) {
            const half8 mv = ld8(mat + r * k, col, k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)mv[e];
                acc[0][e] += v0 * f; acc[1][e] += v1 * f; acc[2][e] += v2 * f;
            }
        }
    }
    float* const outs[3] = {out, out + k, out + 2 * k};
    cr_finish<3>(acc, outs, col0, k);
    if (vsum != nullptr && blockIdx.x == 0 && cg == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) atomicAdd(vsum + j, vs[j]);
    }
}

// out[m][k] = sum_j vec[m][j] * w[j][k]
__global__ void vec3_expand_kernel(const float* __restrict__ vec, const float* __restrict__ w, int64_t m, int k,
                                   half_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * k) return;
    const int64_t row = i / k;
    const int col = (int)(i - row * k);
    out[i] = to_half_sat(vec[row * 3] * w[col] + vec[row * 3 + 1] * w[k + col] + vec[row * 3 + 2] * w[2 * k + col]);
}

// loss_sum += sum |pred - target| ; dpred = scale * sign(pred - target) / n
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                  float scale, float* __restrict__ loss_sum, float* __restrict__ dpred) {
    __shared__ float red[256];
    float s = 0.f;
    const float gs = scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        s += fabsf(d);
        dpred[i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0]);
}

// C[i][j] (+)= sum_k opA(i,k) * opB(k,j): small fp32 products (time MLP, per-shape bias paths); one thread per output
__global__ void matmul_f32_kernel(const float* __restrict__ a, int64_t lda, int ta, const float* __restrict__ b, int64_t ldb,
                                  int tb, int mm, int nn, int kk, const float* __restrict__ bias, int accumulate,
                                  float* __restrict__ c, int64_t ldc) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)mm * nn) return;
    const int i = (int)(idx / nn), j = (int)(idx - (int64_t)i * nn);
    float s = bias != nullptr ? bias[j] : 0.f;
    for (int k = 0; k < kk; ++k) {
        const float av = ta ? a[(int64_t)k * lda + i] : a[(int64_t)i * lda + k];
        const float bv = tb ? b[(int64_t)j * ldb + k] : b[(int64_t)k * ldb + j];
        s += av * bv;
    }
    float* o = c + (int64_t)i * ldc + j;
    *o = accumulate ? *o + s : s;
}

// The same product for few rows (m <= 32: one row per shape of the batch), A not transposed.  The generic kernel
// re-reads B once per output row and, for B stored [n][k], walks it uncoalesced (697 us for the 16 x 1024 x 4096
// per-shape bias of dec4.conv1); here B is read once.
// B stored [k][n]: a thread owns output column j and the k range of its block row (64 deep), keeps all m partial sums
// and adds them atomically (c zeroed by the host side when not accumulating): n/256 x k/64 blocks instead of n/256.
__global__ __launch_bounds__(256) void matmul_f32_fewrows_nn_kernel(const float* __restrict__ a, int64_t lda,
                                                                     const float* __restrict__ b, int64_t ldb, int mm, int nn,
                                                                     int kk, const float* __restrict__ bias,
                                                                     float* __restrict__ c, int64_t ldc) {
    __shared__ float as[32][64];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = blockIdx.y * 64, k1 = min(kk, k0 + 64);
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
        const int i = idx >> 6, k = k0 + (idx & 63);
        as[i][idx & 63] = (i < mm && k < k1) ? a[(int64_t)i * lda + k] : 0.f;
    }
    __syncthreads();
    if (j >= nn) return;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float bv = b[(int64_t)k * ldb + j];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] += as[i][k - k0] * bv;      // LDS broadcast reads
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < mm) atomicAdd(c + (int64_t)i * ldc + j, acc[i] + ((bias != nullptr && blockIdx.y == 0) ? bias[j] : 0.f));
}
// B stored [n][k]: one wave per output column j; lanes stride over k (both operands read contiguously)
__global__ __launch_bounds__(256) void matmul_f32_fewrows_nt_kernel(const float* __restrict__ a, int64_t lda,
                                                                     const float* __restrict__ b, int64_t ldb, int mm, int nn,
                                                                     int kk, const float* __restrict__ bias, int accumulate,
                                                                     float* __restrict__ c, int64_t ldc) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nn) return;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const float* brow = b + (int64_t)j * ldb;
    for (int k = lane; k < kk; k += 64) {
        const float bv = brow[k];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < mm) acc[i] += a[(int64_t)i * lda + k] * bv;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && i < mm) {
            float* op = c + (int64_t)i * ldc + j;
            v += bias != nullptr ? bias[j] : 0.f;
            *op = accumulate ? *op + v : v;
        }
    }
}

__global__ void silu_kernel(const float* x, int64_t n, float* y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] / (1.f + expf(-x[i]));
}
__global__ void silu_bwd_kernel(const float* x, const float* dy, int64_t n, float* dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.f / (1.f + expf(-x[i]));
    dx[i] = dy[i] * s * (1.f + x[i] * (1.f - s));
}

// ---- latent denoiser training (networks.py:977-1049 layers on (B, C) rows, all fp32: B is the batch, 16-32 rows)
// y = act(gamma * (x - mean) * rstd + beta) per (row, group of c/groups channels); one wave per (row, group)
__global__ __launch_bounds__(256) void gn_fwd_f32_kernel(const float* __restrict__ x, int rows, int c, int groups,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, int relu, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rows * groups) return;
    const int row = item / groups, g = item - row * groups, gsz = c / groups;
    const float* xg = x + (int64_t)row * c + g * gsz;
    float s = 0.f;
    for (int i = lane; i < gsz; i += 64) s += xg[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)gsz;
    float v = 0.f;
    for (int i = lane; i < gsz; i += 64) { const float d = xg[i] - mu; v += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rs = rsqrtf(v / (float)gsz + eps);
    if (lane == 0) { mean[item] = mu; rstd[item] = rs; }
    for (int i = lane; i < gsz; i += 64) {
        const int ch = g * gsz + i;
        float o = (xg[i] - mu) * rs * gamma[ch] + beta[ch];
        if (relu) o = fmaxf(o, 0.f);
        y[(int64_t)row * c + ch] = o;
    }
}
// dx of the above; g = dy * [y > 0]
__global__ __launch_bounds__(256) void gn_bwd_dx_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, int rows,
                                                             int c, int groups, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int relu, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rows * groups) return;
    const int row = item / groups, g = item - row * groups, gsz = c / groups;
    const int64_t base = (int64_t)row * c + g * gsz;
    const float mu = mean[item], rs = rstd[item];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < gsz; i += 64) {
        const int ch = g * gsz + i;
        const float xh = (x[base + i] - mu) * rs;
        const float gg = (!relu || xh * gamma[ch] + beta[ch] > 0.f) ? dy[base + i] * gamma[ch] : 0.f;
        s1 += gg; s2 += gg * xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= (float)gsz; s2 /= (float)gsz;
    for (int i = lane; i < gsz; i += 64) {
        const int ch = g * gsz + i;
        const float xh = (x[base + i] - mu) * rs;
        const float gg = (!relu || xh * gamma[ch] + beta[ch] > 0.f) ? dy[base + i] * gamma[ch] : 0.f;
        dx[base + i] = rs * (gg - s1 - xh * s2);
    }
}
// dgamma[ch] = sum_rows g * xhat, dbeta[ch] = sum_rows g ; one thread per channel (few rows)
__global__ void gn_bwd_affine_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, int rows, int c, int groups,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const int g = ch / (c / groups);
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float xh = (x[(int64_t)r * c + ch] - mean[r * groups + g]) * rstd[r * groups + g];
        const float gg = (!relu || xh * gamma[ch] + beta[ch] > 0.f) ? dy[(int64_t)r * c + ch] : 0.f;
        a += gg * xh; b += gg;
    }
    dgamma[ch] = a; dbeta[ch] = b;
}
// y = x * mask * scale (Dropout forward with a given keep mask, and its backward); y = max(x, 0); dx = dy * [x > 0]
__global__ void mask_scale_kernel(const float* x, const float* mask, float scale, int64_t n, float* y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * mask[i] * scale;
}
__global__ void relu_f32_kernel(const float* x, int64_t n, float* y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_f32_kernel(const float* x, const float* dy, int64_t n, float* dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
}

// ---- voxel VAE training (networks.py:2225-2264): every Conv3d / ConvTranspose3d becomes "gather rows, then the GEMM".
// Geometry of one layer: input grid (di,hi,wi), output grid (do,ho,wo), cubic kernel k, stride s, padding p.
// transposed = 0 (Conv3d): output o reads input o*s - p + t.  transposed = 1 (ConvTranspose3d): output o reads input
// (o + p - t)/s where that is an integer.  Channels-last fp16 rows [b*D*H*W][C].
struct ConvGeom { int di, hi, wi, d_o, ho, wo, k, s, p, transposed; };

__device__ __forceinline__ bool conv_src(const ConvGeom& g, int o, int t, int in_extent, int& i) {
    if (!g.transposed) { i = o * g.s - g.p + t; return i >= 0 && i < in_extent; }
    const int num = o + g.p - t;
    if (num < 0 || num % g.s != 0) return false;
    i = num / g.s;
    return i < in_extent;
}
// the inverse relation: which output position reads input i at tap t
__device__ __forceinline__ bool conv_dst(const ConvGeom& g, int i, int t, int out_extent, int& o) {
    if (!g.transposed) {
        const int num = i + g.p - t;
        if (num < 0 || num % g.s != 0) return false;
        o = num / g.s;
        return o < out_extent;
    }
    o = i * g.s - g.p + t;
    return o >= 0 && o < out_extent;
}

// col[r][t*cin + c] = x[src(r, t)][c] or 0; columns >= k^3*cin (padding up to kp) are zeroed.  One thread per (row, tap).
__global__ __launch_bounds__(256) void im2col_kernel(const half_t* __restrict__ x, int batch, int cin, ConvGeom g, int kp,
                                                      half_t* __restrict__ col) {
    const int taps = g.k * g.k * g.k;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)batch * g.d_o * g.ho * g.wo;
    if (idx >= rows * (taps + 1)) return;
    const int64_t r = idx / (taps + 1);
    const int t = (int)(idx - r * (taps + 1));
    half_t* dst = col + r * kp;
    if (t == taps) {                                      // zero the padding columns of this row
        for (int c = taps * cin; c < kp; ++c) dst[c] = (half_t)0.f;
        return;
    }
    int ow = (int)(r % g.wo), oh = (int)((r / g.wo) % g.ho), od = (int)((r / ((int64_t)g.wo * g.ho)) % g.d_o);
    const int b = (int)(r / ((int64_t)g.wo * g.ho * g.d_o));
    const int tz = t / (g.k * g.k), ty = (t / g.k) % g.k, tx = t % g.k;
    int iz, iy, ix;
    const bool ok = conv_src(g, od, tz, g.di, iz) && conv_src(g, oh, ty, g.hi, iy) && conv_src(g, ow, tx, g.wi, ix);
    dst += (int64_t)t * cin;
    if (!ok) {
        for (int c = 0; c < cin; ++c) dst[c] = (half_t)0.f;
        return;
    }
    const half_t* src = x + ((((int64_t)b * g.di + iz) * g.hi + iy) * g.wi + ix) * cin;
    if (cin % 8 == 0) {
        for (int c = 0; c < cin; c += 8) *(half8*)(dst + c) = *(const half8*)(src + c);
    } else {
        for (int c = 0; c < cin; ++c) dst[c] = src[c];
    }
}

// dx[i][c] = sum over taps of dcol[dst(i, t)][t*cin + c]   (fp32 accumulation, fixed tap order: deterministic)
__global__ __launch_bounds__(256) void col2im_kernel(const half_t* __restrict__ dcol, int batch, int cin, ConvGeom g, int kp,
                                                      half_t* __restrict__ dx) {
    const int cchunks = (cin + 7) / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = (int64_t)batch * g.di * g.hi * g.wi;
    if (idx >= rows * cchunks) return;
    const int64_t i = idx / cchunks;
    const int c0 = (int)(idx - i * cchunks) * 8;
    const int iw = (int)(i % g.wi), ih = (int)((i / g.wi) % g.hi), id = (int)((i / ((int64_t)g.wi * g.hi)) % g.di);
    const int b = (int)(i / ((int64_t)g.wi * g.hi * g.di));
    float acc[8] = {};
    for (int tz = 0; tz < g.k; ++tz) {
        int od;
        if (!conv_dst(g, id, tz, g.d_o, od)) continue;
        for (int ty = 0; ty < g.k; ++ty) {
            int oh;
            if (!conv_dst(g, ih, ty, g.ho, oh)) continue;
            for (int tx = 0; tx < g.k; ++tx) {
                int ow;
                if (!conv_dst(g, iw, tx, g.wo, ow)) continue;
                const int t = (tz * g.k + ty) * g.k + tx;
                const half_t* src = dcol + ((((int64_t)b * g.d_o + od) * g.ho + oh) * g.wo + ow) * kp + (int64_t)t * cin + c0;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (c0 + e < cin) acc[e] += (float)src[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (c0 + e < cin) dx[i * cin + c0 + e] = to_half_sat(acc[e]);
}

// out = relu(a + b) (ResidualBlock3D tail, networks.py:502-504) and the shared mask of its backward: d = dout * [out > 0]
__global__ void add_relu_f16_kernel(const half_t* a, const half_t* b, int64_t n, half_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_half_sat(fmaxf((float)a[i] + (float)b[i], 0.f));
}
__global__ void relu_mask_f16_kernel(const half_t* dout, const half_t* out, int64_t n, half_t* d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)out[i] > 0.f ? dout[i] : (half_t)0.f;
}
__global__ void add_f16_kernel(const half_t* a, const half_t* b, int64_t n, half_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_half_sat((float)a[i] + (float)b[i]);
}
// out[r][c] = act(x[r][c] + bias[c])   (bias + ReLU after a ConvTranspose3d computed as product-then-col2im)
__global__ void bias_act_f16_kernel(const half_t* x, const float* bias, int64_t rows, int c, int relu, half_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c) return;
    float v = (float)x[i] + bias[i % c];
    if (relu) v = fmaxf(v, 0.f);
    out[i] = to_half_sat(v);
}
// recon = sigmoid(logit) ; loss_sum += BCE(recon, x) with torch's log clamp at -100 ; dlogit = scale * (recon - x) / n
__global__ __launch_bounds__(256) void sigmoid_bce_kernel(const half_t* __restrict__ logit, int64_t ld, const float* __restrict__ target,
                                                           int64_t n, float scale, float* __restrict__ loss_sum,
                                                           float* __restrict__ recon, half_t* __restrict__ dlogit) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = (float)logit[i * ld];
        const float r = 1.f / (1.f + expf(-z));
        const float x = target[i];
        s -= x * fmaxf(logf(r), -100.f) + (1.f - x) * fmaxf(logf(1.f - r), -100.f);
        recon[i] = r;
        dlogit[i * ld] = to_half_sat(scale * (r - x) / (float)n);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0]);
}

// VAE latent head backward (networks.py:2323-2325, 2389-2396): z = mu + eps * exp(logvar / 2), KL = -0.5 mean(1 + lv - mu^2 - e^lv)
//   dmu = dz + kl_scale * mu / n ; dlogvar = dz * eps * 0.5 * exp(lv / 2) + kl_scale * 0.5 * (e^lv - 1) / n ; kl_sum += 1 + lv - mu^2 - e^lv
__global__ void vae_latent_bwd_kernel(const float* mu, const float* lv, const float* eps, const float* dz, int64_t n, float kl_scale,
                                      float* dmu, float* dlv, float* kl_sum) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = expf(lv[i]);
    dmu[i] = dz[i] + kl_scale * mu[i] / (float)n;
    dlv[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]) + kl_scale * 0.5f * (e - 1.f) / (float)n;
    atomicAdd(kl_sum, 1.f + lv[i] - mu[i] * mu[i] - e);
}

// torch.optim.AdamW (decoupled weight decay), one flat buffer
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                             float inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * inv_scale;
    float w = p[i];
    w -= lr * wd * w;
    const float m = b1 * m1[i] + (1.f - b1) * gr;
    const float v = b2 * m2[i] + (1.f - b2) * gr * gr;
    m1[i] = m; m2[i] = v;
    w -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + eps);
    p[i] = w;
}

// column-strip launch geometry: enough row chunks for ~2048 blocks in total, chunks a multiple of the 32 row lanes
struct ColGrid { dim3 grid; int rpb; };
static inline ColGrid cr_grid(int c, int64_t rows_per_group, int groups) {
    const int64_t cb = ceil_div(c, CR_COLS);
    int64_t chunks = 2048 / (cb * groups);
    if (chunks < 1) chunks = 1;
    int64_t rpb = ceil_div(ceil_div(rows_per_group, chunks), CR_LANES) * CR_LANES;
    if (rpb < 4 * CR_LANES) rpb = 4 * CR_LANES;
    return {dim3((unsigned)cb, (unsigned)ceil_div(rows_per_group, rpb), (unsigned)groups), (int)rpb};
}
static inline unsigned nblk256(int64_t n) { return (unsigned)ceil_div(n, 256); }

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_colsum_f16(const void* x, int64_t rows_per_group, int groups, int c, float* out, void* stream) {
    PCD_CHECK_ARG(x && out && rows_per_group > 0 && groups > 0 && groups <= 65535 && c > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)groups * c, s));
    const ColGrid g = cr_grid(c, rows_per_group, groups);
    hipLaunchKernelGGL(colsum_kernel<half_t>, g.grid, dim3(256), 0, s, (const half_t*)x, rows_per_group, c, out, g.rpb);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_bn_batch_stats(const float* z, int64_t m, int c, float momentum, float* mean, float* var,
                                  float* running_mean, float* running_var, float* scratch, void* stream) {
    PCD_CHECK_ARG(z && mean && var && scratch && m > 0 && c > 0 && c % 8 == 0);
    PCD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    hipStream_t s = (hipStream_t)stream;
    float *s1 = scratch, *s2 = scratch + c;
    PCD_CHECK_HIP(hipMemsetAsync(scratch, 0, sizeof(float) * 2 * (size_t)c, s));
    const ColGrid g = cr_grid(c, m, 1);
    hipLaunchKernelGGL(colstats_kernel, g.grid, dim3(256), 0, s, z, m, c, s1, s2, g.rpb);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk256(c)), dim3(256), 0, s, z, s1, s2, m, c, momentum, mean, var, running_mean,
                       running_var);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_bn_apply_f16(const float* z, int64_t m, int c, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, int relu, void* out, void* stream) {
    PCD_CHECK_ARG(z && mean && var && gamma && beta && out && m > 0 && c > 0);
    const ColGrid g = cr_grid(c, m, 1);
    hipLaunchKernelGGL(bn_apply_kernel, g.grid, dim3(256), 0, (hipStream_t)stream, z, m, c, mean, var, gamma, beta, eps, relu,
                       (half_t*)out, g.rpb);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_bn_backward_f16(const void* da, const float* z, int64_t m, int c, const float* mean, const float* var,
                                   const float* gamma, const float* beta, float eps, int relu, float* dgamma, float* dbeta,
                                   void* dz, void* stream) {
    PCD_CHECK_ARG(da && z && mean && var && gamma && beta && dgamma && dbeta && dz && m > 0 && c > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * (size_t)c, s));
    PCD_CHECK_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * (size_t)c, s));
    const ColGrid g = cr_grid(c, m, 1);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, g.grid, dim3(256), 0, s, (const half_t*)da, z, m, c, mean, var, gamma, beta, eps, relu,
                       dbeta, dgamma, g.rpb);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, g.grid, dim3(256), 0, s, (const half_t*)da, z, m, c, mean, var, gamma, beta, eps, relu,
                       dbeta, dgamma, (half_t*)dz, g.rpb);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_transpose_f16(const void* src, int64_t rows, int cols, void* dst, void* stream) {
    PCD_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ceil_div(rows, 64) <= 65535);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)src, rows, cols, (half_t*)dst);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_colmax_argmax_f16(const void* a, int batch, int n_points, int c, float* mx, int* arg, void* stream) {
    PCD_CHECK_ARG(a && mx && arg && batch > 0 && batch <= 65535 && n_points > 0 && c > 0);
    hipLaunchKernelGGL(colmax_argmax_kernel, dim3((unsigned)ceil_div(c, 64), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)a, n_points, c, mx, arg);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_maxpool_backward_f16(const float* dg, const int* arg, int batch, int n_points, int c, void* da, void* stream) {
    PCD_CHECK_ARG(dg && arg && da && batch > 0 && n_points > 0 && c > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(da, 0, sizeof(half_t) * (size_t)batch * n_points * c, s));
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk256((int64_t)batch * c)), dim3(256), 0, s, dg, arg, n_points, c, batch, (half_t*)da);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_enc1_linear(const float* x, int64_t m, int n_points, const float* w_xyz, int c1, const float* tbias,
                               float* out, void* stream) {
    PCD_CHECK_ARG(x && w_xyz && tbias && out && m > 0 && n_points > 0 && c1 > 0);
    hipLaunchKernelGGL(enc1_linear_kernel, dim3(nblk256(m * c1)), dim3(256), 0, (hipStream_t)stream, x, m, n_points, w_xyz, c1, tbias, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_vec3_outer(const void* mat, const float* vec, int64_t m, int k, float* out, float* vsum, void* stream) {
    PCD_CHECK_ARG(mat && vec && out && m > 0 && k > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * 3 * (size_t)k, s));
    if (vsum != nullptr) PCD_CHECK_HIP(hipMemsetAsync(vsum, 0, sizeof(float) * 3, s));
    const ColGrid g = cr_grid(k, m, 1);
    hipLaunchKernelGGL(vec3_outer_kernel, g.grid, dim3(256), 0, s, (const half_t*)mat, vec, m, k, out, vsum, g.rpb);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_vec3_expand_f16(const float* vec, const float* w, int64_t m, int k, void* out, void* stream) {
    PCD_CHECK_ARG(vec && w && out && m > 0 && k > 0);
    hipLaunchKernelGGL(vec3_expand_kernel, dim3(nblk256(m * k)), dim3(256), 0, (hipStream_t)stream, vec, w, m, k, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_l1_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss_sum, float* dpred,
                           void* stream) {
    PCD_CHECK_ARG(pred && target && loss_sum && dpred && n > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), s));
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024);
    hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(256), 0, s, pred, target, n, grad_scale, loss_sum, dpred);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_matmul_f32(const float* a, int64_t lda, int trans_a, const float* b, int64_t ldb, int trans_b, int m, int n,
                              int k, const float* bias, int accumulate, float* c, int64_t ldc, void* stream) {
    PCD_CHECK_ARG(a && b && c && m > 0 && n > 0 && k > 0);
    if (!trans_a && m <= 32 && !trans_b) {
        if (!accumulate) PCD_CHECK_HIP(hipMemset2DAsync(c, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)n, (size_t)m, (hipStream_t)stream));
        hipLaunchKernelGGL(matmul_f32_fewrows_nn_kernel, dim3(nblk256(n), (unsigned)ceil_div(k, 64)), dim3(256), 0, (hipStream_t)stream, a,
                           lda, b, ldb, m, n, k, bias, c, ldc);
    }
    else if (!trans_a && m <= 32 && trans_b)
        hipLaunchKernelGGL(matmul_f32_fewrows_nt_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                           m, n, k, bias, accumulate, c, ldc);
    else
        hipLaunchKernelGGL(matmul_f32_kernel, dim3(nblk256((int64_t)m * n)), dim3(256), 0, (hipStream_t)stream, a, lda, trans_a, b, ldb,
                           trans_b, m, n, k, bias, accumulate, c, ldc);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_silu_f32(const float* x, int64_t n, float* y, void* stream) {
    PCD_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(silu_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_silu_backward_f32(const float* x, const float* dy, int64_t n, float* dx, void* stream) {
    PCD_CHECK_ARG(x && dy && dx && n > 0);
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, x, dy, n, dx);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    PCD_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1 && grad_scale > 0.f);
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2, 1.f / grad_scale);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_groupnorm_f32(const float* x, int rows, int c, int groups, const float* gamma, const float* beta, float eps,
                                 int relu, float* y, float* mean, float* rstd, void* stream) {
    PCD_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && c > 0 && groups > 0 && c % groups == 0);
    hipLaunchKernelGGL(gn_fwd_f32_kernel, dim3((unsigned)ceil_div((int64_t)rows * groups, 4)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       c, groups, gamma, beta, eps, relu, y, mean, rstd);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_groupnorm_backward_f32(const float* dy, const float* x, int rows, int c, int groups, const float* gamma,
                                          const float* beta, const float* mean, const float* rstd, int relu, float* dx,
                                          float* dgamma, float* dbeta, void* stream) {
    PCD_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && rows > 0 && c > 0 && groups > 0 && c % groups == 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_dx_f32_kernel, dim3((unsigned)ceil_div((int64_t)rows * groups, 4)), dim3(256), 0, s, dy, x, rows, c, groups,
                       gamma, beta, mean, rstd, relu, dx);
    hipLaunchKernelGGL(gn_bwd_affine_f32_kernel, dim3(nblk256(c)), dim3(256), 0, s, dy, x, rows, c, groups, gamma, beta, mean, rstd, relu,
                       dgamma, dbeta);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_mask_scale_f32(const float* x, const float* mask, float scale, int64_t n, float* y, void* stream) {
    PCD_CHECK_ARG(x && mask && y && n > 0);
    hipLaunchKernelGGL(mask_scale_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, n, y);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_relu_f32(const float* x, int64_t n, float* y, void* stream) {
    PCD_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(relu_f32_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_relu_backward_f32(const float* x, const float* dy, int64_t n, float* dx, void* stream) {
    PCD_CHECK_ARG(x && dy && dx && n > 0);
    hipLaunchKernelGGL(relu_bwd_f32_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, x, dy, n, dx);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_im2col_f16(const void* x, int batch, int cin, int di, int hi, int wi, int d_o, int ho, int wo, int k, int stride,
                              int pad, int transposed, int kp, void* col, void* stream) {
    PCD_CHECK_ARG(x && col && batch > 0 && cin > 0 && di > 0 && hi > 0 && wi > 0 && d_o > 0 && ho > 0 && wo > 0 && k > 0 && stride > 0);
    PCD_CHECK_ARG(pad >= 0 && kp >= k * k * k * cin && kp % 8 == 0);
    const ConvGeom g{di, hi, wi, d_o, ho, wo, k, stride, pad, transposed};
    const int64_t items = (int64_t)batch * d_o * ho * wo * (k * k * k + 1);
    hipLaunchKernelGGL(im2col_kernel, dim3(nblk256(items)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, batch, cin, g, kp,
                       (half_t*)col);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_col2im_f16(const void* dcol, int batch, int cin, int di, int hi, int wi, int d_o, int ho, int wo, int k, int stride,
                              int pad, int transposed, int kp, void* dx, void* stream) {
    PCD_CHECK_ARG(dcol && dx && batch > 0 && cin > 0 && di > 0 && hi > 0 && wi > 0 && d_o > 0 && ho > 0 && wo > 0 && k > 0 && stride > 0);
    PCD_CHECK_ARG(pad >= 0 && kp >= k * k * k * cin);
    const ConvGeom g{di, hi, wi, d_o, ho, wo, k, stride, pad, transposed};
    const int64_t items = (int64_t)batch * di * hi * wi * ceil_div(cin, 8);
    hipLaunchKernelGGL(col2im_kernel, dim3(nblk256(items)), dim3(256), 0, (hipStream_t)stream, (const half_t*)dcol, batch, cin, g, kp,
                       (half_t*)dx);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_add_relu_f16(const void* a, const void* b, int64_t n, int relu, void* out, void* stream) {
    PCD_CHECK_ARG(a && b && out && n > 0);
    if (relu) hipLaunchKernelGGL(add_relu_f16_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, n, (half_t*)out);
    else hipLaunchKernelGGL(add_f16_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, n, (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_relu_mask_f16(const void* dout, const void* out, int64_t n, void* d, void* stream) {
    PCD_CHECK_ARG(dout && out && d && n > 0);
    hipLaunchKernelGGL(relu_mask_f16_kernel, dim3(nblk256(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)dout, (const half_t*)out, n,
                       (half_t*)d);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_sigmoid_bce(const void* logit, int64_t ld, const float* target, int64_t n, float grad_scale, float* loss_sum,
                               float* recon, void* dlogit, void* stream) {
    PCD_CHECK_ARG(logit && target && loss_sum && recon && dlogit && n > 0 && ld >= 1);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), s));
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 2048 ? ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(blocks), dim3(256), 0, s, (const half_t*)logit, ld, target, n, grad_scale, loss_sum, recon,
                       (half_t*)dlogit);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_vae_latent_backward(const float* mu, const float* logvar, const float* eps, const float* dz, int64_t n,
                                       float kl_scale, float* dmu, float* dlogvar, float* kl_sum, void* stream) {
    PCD_CHECK_ARG(mu && logvar && eps && dz && dmu && dlogvar && kl_sum && n > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(kl_sum, 0, sizeof(float), s));
    hipLaunchKernelGGL(vae_latent_bwd_kernel, dim3(nblk256(n)), dim3(256), 0, s, mu, logvar, eps, dz, n, kl_scale, dmu, dlogvar, kl_sum);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_bias_act_f16(const void* x, const float* bias, int64_t rows, int c, int relu, void* out, void* stream) {
    PCD_CHECK_ARG(x && bias && out && rows > 0 && c > 0);
    hipLaunchKernelGGL(bias_act_f16_kernel, dim3(nblk256(rows * c)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, bias, rows, c, relu,
                       (half_t*)out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
