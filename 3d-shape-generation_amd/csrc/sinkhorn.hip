// K12: log-domain Sinkhorn EMD (reference metrics.py:94-158) without ever storing the
// B x n x m cost matrix: every pass recomputes C_ij = |x_i - y_j| / Cmax from the 3-float
// points (12 bytes per point from LDS), so a pass is VALU/exp-bound, not HBM-bound.
//   alpha_i = eps * (log(mu + 1e-10) - logsumexp_j(-C_ij / eps + beta_j))     (and the mirror for beta)
// One thread per row, the other cloud + its dual streamed through LDS tiles, online logsumexp.
#include "common.h"

namespace pcd {

constexpr int SK_TILE = 512;

__global__ __launch_bounds__(256) void pair_max_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        int n, int m, unsigned* __restrict__ out_bits) {
    __shared__ float ty[SK_TILE * 3];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xb = x + (int64_t)b * n * 3;
    const float* yb = y + (int64_t)b * m * 3;
    const bool live = i < n;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = xb[i * 3]; py = xb[i * 3 + 1]; pz = xb[i * 3 + 2]; }
    float best = 0.f;
    for (int j0 = 0; j0 < m; j0 += SK_TILE) {
        const int cnt = min(SK_TILE, m - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += blockDim.x) ty[t] = yb[(int64_t)j0 * 3 + t];
        __syncthreads();
        if (live)
            for (int j = 0; j < cnt; ++j) {
                const float dx = px - ty[j * 3], dy = py - ty[j * 3 + 1], dz = pz - ty[j * 3 + 2];
                best = fmaxf(best, dx * dx + dy * dy + dz * dz);
            }
    }
    best = sqrtf(best);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(best));
}

// dual update for the rows of `p` against the cloud `q` with dual `dq`:
//   dp_new[i] = eps * (logmarg - lse_j(-lambda * |p_i - q_j| / cmax + dq[j])),  err = max |dp_new - dp_old|
__global__ __launch_bounds__(256) void sinkhorn_dual_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                             int np_, int nq, const float* __restrict__ cmax,
                                                             float lambda, float eps, float logmarg,
                                                             const float* __restrict__ dq, float* __restrict__ dp,
                                                             unsigned* __restrict__ err_bits) {
    __shared__ float tq[SK_TILE * 4];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* pb = p + (int64_t)b * np_ * 3;
    const float* qb = q + (int64_t)b * nq * 3;
    const float* dqb = dq + (int64_t)b * nq;
    const bool live = i < np_;
    const float scale = lambda / cmax[0];
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = pb[i * 3]; py = pb[i * 3 + 1]; pz = pb[i * 3 + 2]; }
    float mrun = -INFINITY, srun = 0.f;
    for (int j0 = 0; j0 < nq; j0 += SK_TILE) {
        const int cnt = min(SK_TILE, nq - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            tq[t * 4] = qb[(int64_t)(j0 + t) * 3]; tq[t * 4 + 1] = qb[(int64_t)(j0 + t) * 3 + 1];
            tq[t * 4 + 2] = qb[(int64_t)(j0 + t) * 3 + 2]; tq[t * 4 + 3] = dqb[j0 + t];
        }
        __syncthreads();
        if (live)
            for (int j = 0; j < cnt; ++j) {
                const float dx = px - tq[j * 4], dy = py - tq[j * 4 + 1], dz = pz - tq[j * 4 + 2];
                const float v = tq[j * 4 + 3] - scale * sqrtf(dx * dx + dy * dy + dz * dz);
                if (v > mrun) { srun = srun * __expf(mrun - v) + 1.f; mrun = v; }
                else srun += __expf(v - mrun);
            }
    }
    float e = 0.f;
    if (live) {
        const float nv = eps * (logmarg - (mrun + logf(srun)));
        float* dst = dp + (int64_t)b * np_ + i;
        e = fabsf(nv - *dst);
        *dst = nv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = fmaxf(e, __shfl_xor(e, o));
    if ((threadIdx.x & 63) == 0) atomicMax(err_bits, __float_as_uint(e));
}

// row_cost[b][i] = sum_j exp(-lambda C_ij + alpha_i + beta_j) * C_ij
__global__ __launch_bounds__(256) void sinkhorn_cost_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             int n, int m, const float* __restrict__ cmax, float lambda,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ row_cost) {
    __shared__ float tq[SK_TILE * 4];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xb = x + (int64_t)b * n * 3;
    const float* yb = y + (int64_t)b * m * 3;
    const bool live = i < n;
    const float inv = 1.f / cmax[0];
    float px = 0.f, py = 0.f, pz = 0.f, a = 0.f;
    if (live) { px = xb[i * 3]; py = xb[i * 3 + 1]; pz = xb[i * 3 + 2]; a = alpha[(int64_t)b * n + i]; }
    float acc = 0.f;
    for (int j0 = 0; j0 < m; j0 += SK_TILE) {
        const int cnt = min(SK_TILE, m - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            tq[t * 4] = yb[(int64_t)(j0 + t) * 3]; tq[t * 4 + 1] = yb[(int64_t)(j0 + t) * 3 + 1];
            tq[t * 4 + 2] = yb[(int64_t)(j0 + t) * 3 + 2]; tq[t * 4 + 3] = beta[(int64_t)b * m + j0 + t];
        }
        __syncthreads();
        if (live)
            for (int j = 0; j < cnt; ++j) {
                const float dx = px - tq[j * 4], dy = py - tq[j * 4 + 1], dz = pz - tq[j * 4 + 2];
                const float c = sqrtf(dx * dx + dy * dy + dz * dz) * inv;
                acc += __expf(-lambda * c + a + tq[j * 4 + 3]) * c;
            }
    }
    if (live) row_cost[(int64_t)b * n + i] = acc;
}

// out[b] = sum_i row_cost[b][i], fixed order (one block per batch entry)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ double ws[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)v[(int64_t)blockIdx.x * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(ws[0] + ws[1] + ws[2] + ws[3]);
}

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_pairwise_max_dist(const float* x, const float* y, int batch, int n, int m, float* out_max,
                                     void* stream) {
    PCD_CHECK_ARG(x && y && out_max && batch > 0 && n > 0 && m > 0);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(out_max, 0, sizeof(float), s));
    hipLaunchKernelGGL(pair_max_kernel, dim3((unsigned)ceil_div(n, 256), batch), dim3(256), 0, s, x, y, n, m,
                       (unsigned*)out_max);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_sinkhorn_dual_update(const float* p, const float* q, int batch, int np_, int nq, const float* cmax,
                                        float epsilon, float log_marginal, const float* dual_q, float* dual_p,
                                        float* err_max, void* stream) {
    PCD_CHECK_ARG(p && q && cmax && dual_q && dual_p && err_max && batch > 0 && np_ > 0 && nq > 0 && epsilon > 0.f);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(err_max, 0, sizeof(float), s));
    hipLaunchKernelGGL(sinkhorn_dual_kernel, dim3((unsigned)ceil_div(np_, 256), batch), dim3(256), 0, s, p, q, np_, nq,
                       cmax, 1.f / epsilon, epsilon, log_marginal, dual_q, dual_p, (unsigned*)err_max);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_sinkhorn_cost(const float* x, const float* y, int batch, int n, int m, const float* cmax,
                                 float epsilon, const float* alpha, const float* beta, float* row_scratch,
                                 float* cost, void* stream) {
    PCD_CHECK_ARG(x && y && cmax && alpha && beta && row_scratch && cost && batch > 0 && n > 0 && m > 0 && epsilon > 0.f);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3((unsigned)ceil_div(n, 256), batch), dim3(256), 0, s, x, y, n, m, cmax,
                       1.f / epsilon, alpha, beta, row_scratch);
    PCD_CHECK_LAUNCH();
    hipLaunchKernelGGL(row_sum_kernel, dim3(batch), dim3(256), 0, s, row_scratch, n, cost);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
