// K10-K12 for a BATCH of independent cloud pairs: what the reference's evaluation loop computes one pair at a time
// (test_point_ddpm.py:85-92 -> metrics.compute_metrics, metrics.py:160-183): normalize_to_cube of both clouds, Chamfer
// distance, Sinkhorn EMD (metrics.py:94-158) and the voxel-occupancy BCE, for P ragged pairs in one enqueue with no
// host synchronisation.  Pair p: a[p][0..na[p]) and b[p][0..nb[p]) inside padded [P][NA][3] / [P][NB][3] arrays.
//   * Chamfer: thread = query, targets streamed through LDS; the target range is split over blockIdx.y so that a
//     single pair still spreads over the chip (per-query minima meet in an atomicMin on the float bits of d^2 >= 0,
//     which is order independent), then one block per pair adds sqrt(min) in a fixed order.
//   * Sinkhorn: the log-domain dual updates of sinkhorn.hip with per-pair sizes, per-pair cost normalisation and the
//     reference's `if err < thresh: break` kept ON THE DEVICE: iteration k+1 of a pair is skipped when both of its
//     iteration-k errors are below the threshold (a skipped iteration leaves its error slots at zero, so the skip is
//     sticky), and the host enqueues all max_iter iterations without reading anything back.
//   * voxel BCE: both (un-normalised, as metrics.py:181 passes them) clouds are rasterised to 32^3 bit sets; F.binary_cross_entropy of two binary grids with torch's
//     log clamp at -100 is exactly 100 * (number of differing voxels) / 32768.
#include "common.h"

namespace pcd {

#pragma clang fp contract(off)

struct PMaxOp { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct PMinOp { __device__ float operator()(float a, float b) const { return fminf(a, b); } };
template <typename Op>
__device__ __forceinline__ float pblock_reduce(float v, Op op, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = op(r, scratch[w]);
    return r;
}

// grid (P, 2): cloud 0 = a, 1 = b of pair blockIdx.x; same arithmetic as normalize_kernel (metrics.py:7-21)
__global__ __launch_bounds__(256) void pair_normalize_kernel(const float* __restrict__ a, const int* __restrict__ na, int NA,
                                                              const float* __restrict__ b, const int* __restrict__ nb, int NB,
                                                              float* __restrict__ an, float* __restrict__ bn) {
    __shared__ float scratch[4];
    const int p = blockIdx.x, which = blockIdx.y;
    const int n = which ? nb[p] : na[p];
    const float* src = which ? b + (int64_t)p * NB * 3 : a + (int64_t)p * NA * 3;
    float* dst = which ? bn + (int64_t)p * NB * 3 : an + (int64_t)p * NA * 3;
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mn[3] = {INFINITY, INFINITY, INFINITY};
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = src[i * 3 + k];
            mx[k] = fmaxf(mx[k], v);
            mn[k] = fminf(mn[k], v);
        }
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float hi = pblock_reduce(mx[k], PMaxOp(), scratch);
        const float lo = pblock_reduce(mn[k], PMinOp(), scratch);
        c[k] = (hi + lo) / 2.f;
    }
    float am = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; ++k) am = fmaxf(am, fabsf(src[i * 3 + k] - c[k]));
    const float scale = pblock_reduce(am, PMaxOp(), scratch);
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[i * 3 + k] = (src[i * 3 + k] - c[k]) / scale;
}

constexpr int PT = 512;     // targets per LDS tile
constexpr int CQ = 4;       // queries per thread in the Chamfer kernel: one LDS read of a target serves CQ distances

// grid (query blocks of 256 * CQ, target splits, 2 * P); mins[(2p + dir)][q] = min_j |q - t_j|^2 as float bits (init
// +inf).  Targets sit in LDS as float4 (one ds_read_b128, broadcast to the wave); every thread keeps CQ queries in
// registers, so the kernel issues 1 LDS read + 7 CQ VALU operations per CQ distances instead of 3 + 7 per distance.
__global__ __launch_bounds__(256) void pair_chamfer_min_kernel(const float* __restrict__ an, const int* __restrict__ na, int NA,
                                                                const float* __restrict__ bn, const int* __restrict__ nb, int NB,
                                                                int NQ, unsigned* __restrict__ mins) {
    __shared__ float4 tile[PT];
    const int p = blockIdx.z >> 1, dir = blockIdx.z & 1;
    const float* q = dir == 0 ? an + (int64_t)p * NA * 3 : bn + (int64_t)p * NB * 3;
    const float* r = dir == 0 ? bn + (int64_t)p * NB * 3 : an + (int64_t)p * NA * 3;
    const int nq = dir == 0 ? na[p] : nb[p], nr = dir == 0 ? nb[p] : na[p];
    const int q0 = blockIdx.x * (256 * CQ);
    if (q0 >= nq) return;
    const int per = (nr + gridDim.y - 1) / gridDim.y;
    const int r_lo = blockIdx.y * per, r_hi = min(nr, r_lo + per);
    if (r_lo >= r_hi) return;
    float qx[CQ], qy[CQ], qz[CQ], best[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
        int qi = q0 + c * 256 + threadIdx.x;
        qi = qi < nq ? qi : nq - 1;                          // clamped duplicates are not stored
        qx[c] = q[qi * 3]; qy[c] = q[qi * 3 + 1]; qz[c] = q[qi * 3 + 2];
        best[c] = INFINITY;
    }
    for (int r0 = r_lo; r0 < r_hi; r0 += PT) {
        const int cnt = min(PT, r_hi - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float* t = r + (int64_t)(r0 + i) * 3;
            tile[i] = make_float4(t[0], t[1], t[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {
            const float4 t = tile[j];
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
                const float dx = qx[c] - t.x, dy = qy[c] - t.y, dz = qz[c] - t.z;
                best[c] = fminf(best[c], dx * dx + dy * dy + dz * dz);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
        const int qi = q0 + c * 256 + threadIdx.x;
        if (qi < nq) atomicMin(mins + (int64_t)blockIdx.z * NQ + qi, __float_as_uint(best[c]));
    }
}

// grid (P): cd[p] = sum_i sqrt(min_a[i]) / na + sum_j sqrt(min_b[j]) / nb, fixed summation order (metrics.py:41-46)
__global__ __launch_bounds__(256) void pair_chamfer_sum_kernel(const unsigned* __restrict__ mins, const int* __restrict__ na,
                                                                const int* __restrict__ nb, int NQ, float* __restrict__ rows) {
    __shared__ float w[4];
    const int p = blockIdx.x;
    float tot[2];
    for (int dir = 0; dir < 2; ++dir) {
        const int n = dir == 0 ? na[p] : nb[p];
        const unsigned* m = mins + (int64_t)(2 * p + dir) * NQ;
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += sqrtf(__uint_as_float(m[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
        __syncthreads();
        tot[dir] = ((w[0] + w[1]) + w[2]) + w[3];
    }
    if (threadIdx.x == 0) rows[p * 3 + 0] = tot[0] / (float)na[p] + tot[1] / (float)nb[p];
}

#pragma clang fp contract(fast)

// ---- Sinkhorn, ragged.  grid (row blocks, P).
__global__ __launch_bounds__(256) void pair_cmax_kernel(const float* __restrict__ an, const int* __restrict__ na, int NA,
                                                         const float* __restrict__ bn, const int* __restrict__ nb, int NB,
                                                         unsigned* __restrict__ cmax_bits) {
    __shared__ float ty[PT * 3];
    const int p = blockIdx.y, n = na[p], m = nb[p];
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xb = an + (int64_t)p * NA * 3;
    const float* yb = bn + (int64_t)p * NB * 3;
    const bool live = i < n;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = xb[i * 3]; py = xb[i * 3 + 1]; pz = xb[i * 3 + 2]; }
    float best = 0.f;
    for (int j0 = 0; j0 < m; j0 += PT) {
        const int cnt = min(PT, m - j0);
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
    if ((threadIdx.x & 63) == 0) atomicMax(cmax_bits + p, __float_as_uint(best));
}

// one half-iteration for every pair: rows of `pc` (count np[p]) against `qc` (count nq[p]).
// err layout: err[parity][p][2] (alpha, beta) as float bits; `slot` = 0 (alpha update) or 1 (beta update).
// A pair whose two errors of the previous iteration are both below `thresh` is converged and skipped (metrics.py:147-150).
__global__ __launch_bounds__(256) void pair_sinkhorn_dual_kernel(const float* __restrict__ pc, const int* __restrict__ np_, int NP,
                                                                  const float* __restrict__ qc, const int* __restrict__ nq_, int NQc,
                                                                  const unsigned* __restrict__ cmax_bits, float lambda, float eps,
                                                                  const float* __restrict__ logmarg, const float* __restrict__ dq,
                                                                  float* __restrict__ dp, unsigned* __restrict__ err, int P,
                                                                  int iter, int slot, float thresh) {
    __shared__ float tq[PT * 4];
    const int p = blockIdx.y, n = np_[p], m = nq_[p];
    const unsigned* eprev = err + (int64_t)((iter + 1) & 1) * P * 2 + p * 2;
    unsigned* ecur = err + (int64_t)(iter & 1) * P * 2 + p * 2;
    if (iter > 0 && __uint_as_float(eprev[0]) < thresh && __uint_as_float(eprev[1]) < thresh) return;   // converged
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* pb = pc + (int64_t)p * NP * 3;
    const float* qb = qc + (int64_t)p * NQc * 3;
    const float* dqb = dq + (int64_t)p * NQc;
    const bool live = i < n;
    const float scale = lambda / __uint_as_float(cmax_bits[p]);
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = pb[i * 3]; py = pb[i * 3 + 1]; pz = pb[i * 3 + 2]; }
    float mrun = -INFINITY, srun = 0.f;
    for (int j0 = 0; j0 < m; j0 += PT) {
        const int cnt = min(PT, m - j0);
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
        const float nv = eps * (logmarg[p] - (mrun + logf(srun)));
        float* dst = dp + (int64_t)p * NP + i;
        e = fabsf(nv - *dst);
        *dst = nv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = fmaxf(e, __shfl_xor(e, o));
    if ((threadIdx.x & 63) == 0) atomicMax(ecur + slot, __float_as_uint(e));
}

// clears the error slots the NEXT iteration will write (grid 1)
__global__ void pair_sinkhorn_clear_kernel(unsigned* __restrict__ err, int P, int next_iter) {
    unsigned* e = err + (int64_t)(next_iter & 1) * P * 2;
    for (int i = threadIdx.x; i < 2 * P; i += blockDim.x) e[i] = 0u;
}

__global__ __launch_bounds__(256) void pair_sinkhorn_cost_kernel(const float* __restrict__ an, const int* __restrict__ na, int NA,
                                                                  const float* __restrict__ bn, const int* __restrict__ nb, int NB,
                                                                  const unsigned* __restrict__ cmax_bits, float lambda,
                                                                  const float* __restrict__ alpha, const float* __restrict__ beta,
                                                                  float* __restrict__ row_cost) {
    __shared__ float tq[PT * 4];
    const int p = blockIdx.y, n = na[p], m = nb[p];
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xb = an + (int64_t)p * NA * 3;
    const float* yb = bn + (int64_t)p * NB * 3;
    const bool live = i < n;
    const float inv = 1.f / __uint_as_float(cmax_bits[p]);
    float px = 0.f, py = 0.f, pz = 0.f, a = 0.f;
    if (live) { px = xb[i * 3]; py = xb[i * 3 + 1]; pz = xb[i * 3 + 2]; a = alpha[(int64_t)p * NA + i]; }
    float acc = 0.f;
    for (int j0 = 0; j0 < m; j0 += PT) {
        const int cnt = min(PT, m - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            tq[t * 4] = yb[(int64_t)(j0 + t) * 3]; tq[t * 4 + 1] = yb[(int64_t)(j0 + t) * 3 + 1];
            tq[t * 4 + 2] = yb[(int64_t)(j0 + t) * 3 + 2]; tq[t * 4 + 3] = beta[(int64_t)p * NB + j0 + t];
        }
        __syncthreads();
        if (live)
            for (int j = 0; j < cnt; ++j) {
                const float dx = px - tq[j * 4], dy = py - tq[j * 4 + 1], dz = pz - tq[j * 4 + 2];
                const float c = sqrtf(dx * dx + dy * dy + dz * dz) * inv;
                acc += __expf(-lambda * c + a + tq[j * 4 + 3]) * c;
            }
    }
    if (live) row_cost[(int64_t)p * NA + i] = acc;
}

// rows[p][1] = sum_i row_cost[p][i], fixed order, double accumulation (as row_sum_kernel of sinkhorn.hip)
__global__ __launch_bounds__(256) void pair_row_sum_kernel(const float* __restrict__ v, const int* __restrict__ na, int NA,
                                                            float* __restrict__ rows) {
    __shared__ double ws[4];
    const int p = blockIdx.x, n = na[p];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)v[(int64_t)p * NA + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rows[p * 3 + 1] = (float)(ws[0] + ws[1] + ws[2] + ws[3]);
}

#pragma clang fp contract(off)
// ---- voxel BCE: 32^3 occupancy as 1024-word bit sets (index order [x][y][z], utils.py:488-509)
__global__ __launch_bounds__(256) void pair_voxelize_bits_kernel(const float* __restrict__ an, const int* __restrict__ na, int NA,
                                                                  const float* __restrict__ bn, const int* __restrict__ nb, int NB,
                                                                  unsigned* __restrict__ bits /* [P][2][1024], zeroed */) {
    const int p = blockIdx.y, which = blockIdx.z;
    const int n = which ? nb[p] : na[p];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* pt = (which ? bn + (int64_t)p * NB * 3 : an + (int64_t)p * NA * 3) + (int64_t)i * 3;
    int idx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = ((pt[k] + 1.f) * 31.f) / 2.f;                        // utils.py:501
        long long t = (v == v) ? (long long)v : 0;                            // .long(): truncate toward zero
        t = t < 0 ? 0 : (t > 31 ? 31 : t);
        idx[k] = (int)t;
    }
    const int lin = (idx[0] * 32 + idx[1]) * 32 + idx[2];
    atomicOr(bits + ((int64_t)p * 2 + which) * 1024 + (lin >> 5), 1u << (lin & 31));
}

__global__ __launch_bounds__(256) void pair_bce_kernel(const unsigned* __restrict__ bits, float* __restrict__ rows) {
    __shared__ int ws[4];
    const int p = blockIdx.x;
    const unsigned* ga = bits + (int64_t)p * 2 * 1024;
    const unsigned* gb = ga + 1024;
    int diff = 0;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) diff += __popc(ga[i] ^ gb[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = diff;
    __syncthreads();
    if (threadIdx.x == 0) rows[p * 3 + 2] = (100.f * (float)(ws[0] + ws[1] + ws[2] + ws[3])) / 32768.f;   // metrics.py:181
}
#pragma clang fp contract(fast)

// a pair with an empty cloud has no metrics (the reference's compute_metrics would raise on it): its row is NaN, explicitly --
// otherwise the Chamfer mean is 0/0 but the Sinkhorn scale divides by a zero cost maximum and the BCE of an empty grid is a number
__global__ void pair_empty_rows_kernel(const int* __restrict__ na, const int* __restrict__ nb, int P, float* __restrict__ rows) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P && (na[p] <= 0 || nb[p] <= 0)) rows[p * 3 + 0] = rows[p * 3 + 1] = rows[p * 3 + 2] = __builtin_nanf("");
}

static inline size_t pm_align(size_t v) { return (v + 255) / 256 * 256; }
struct PmWs { size_t an, bn, mins, alpha, beta, rowc, cmax, err, bits, total; };
static PmWs pm_carve(int P, int NA, int NB) {
    PmWs w{};
    const int NQ = NA > NB ? NA : NB;
    size_t o = 0;
    w.an = o; o += pm_align((size_t)P * NA * 3 * 4);
    w.bn = o; o += pm_align((size_t)P * NB * 3 * 4);
    w.mins = o; o += pm_align((size_t)P * 2 * NQ * 4);
    w.alpha = o; o += pm_align((size_t)P * NA * 4);
    w.beta = o; o += pm_align((size_t)P * NB * 4);
    w.rowc = o; o += pm_align((size_t)P * NA * 4);
    w.cmax = o; o += pm_align((size_t)P * 4);
    w.err = o; o += pm_align((size_t)P * 4 * 4);
    w.bits = o; o += pm_align((size_t)P * 2 * 1024 * 4);
    w.total = o;
    return w;
}

}  // namespace pcd

using namespace pcd;

extern "C" size_t pcd_pair_metrics_workspace_bytes(int pairs, int na_max, int nb_max) {
    if (pairs <= 0 || na_max <= 0 || nb_max <= 0) return 0;
    return pm_carve(pairs, na_max, nb_max).total;
}

extern "C" int pcd_pair_metrics(const float* a, const int* na, int na_max, const float* b, const int* nb, int nb_max, int pairs,
                                int with_sinkhorn, float epsilon, float thresh, int max_iter, const float* log_mu,
                                const float* log_nu, float* rows, void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(a && na && b && nb && rows && workspace && pairs > 0 && na_max > 0 && nb_max > 0);
    PCD_CHECK_ARG(!with_sinkhorn || (log_mu && log_nu && epsilon > 0.f && max_iter > 0));
    const int P = pairs, NA = na_max, NB = nb_max, NQ = NA > NB ? NA : NB;
    const PmWs w = pm_carve(P, NA, NB);
    if (workspace_bytes < w.total) {
        set_error("pcd_pair_metrics: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    float *an = (float*)(ws + w.an), *bn = (float*)(ws + w.bn), *alpha = (float*)(ws + w.alpha), *beta = (float*)(ws + w.beta);
    float* rowc = (float*)(ws + w.rowc);
    unsigned *mins = (unsigned*)(ws + w.mins), *cmax = (unsigned*)(ws + w.cmax), *err = (unsigned*)(ws + w.err);
    unsigned* bits = (unsigned*)(ws + w.bits);
    hipStream_t s = (hipStream_t)stream;
    PCD_CHECK_HIP(hipMemsetAsync(mins, 0x7f, (size_t)P * 2 * NQ * 4, s));            // 0x7f7f7f7f = 3.39e38 > any d^2
    PCD_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)P * 2 * 1024 * 4, s));
    PCD_CHECK_HIP(hipMemsetAsync(rows, 0, (size_t)P * 3 * 4, s));
    hipLaunchKernelGGL(pair_normalize_kernel, dim3(P, 2), dim3(256), 0, s, a, na, NA, b, nb, NB, an, bn);
    // Chamfer: split the targets so that ~1024 blocks exist even for one pair
    const int qblocks = (int)ceil_div(NQ, 256);
    const int cqblocks = (int)ceil_div(NQ, 256 * CQ);
    int tsplit = (int)ceil_div(1024, (int64_t)cqblocks * 2 * P);
    const int max_split = (int)ceil_div(NQ, 128);
    tsplit = tsplit < 1 ? 1 : (tsplit > max_split ? max_split : tsplit);
    hipLaunchKernelGGL(pair_chamfer_min_kernel, dim3(cqblocks, tsplit, 2 * P), dim3(256), 0, s, an, na, NA, bn, nb, NB, NQ, mins);
    hipLaunchKernelGGL(pair_chamfer_sum_kernel, dim3(P), dim3(256), 0, s, mins, na, nb, NQ, rows);
    hipLaunchKernelGGL(pair_voxelize_bits_kernel, dim3(qblocks, P, 2), dim3(256), 0, s, a, na, NA, b, nb, NB, bits);   // the RAW clouds (metrics.py:181)
    hipLaunchKernelGGL(pair_bce_kernel, dim3(P), dim3(256), 0, s, bits, rows);
    PCD_CHECK_LAUNCH();
    if (with_sinkhorn) {
        PCD_CHECK_HIP(hipMemsetAsync(cmax, 0, (size_t)P * 4, s));
        PCD_CHECK_HIP(hipMemsetAsync(err, 0, (size_t)P * 4 * 4, s));
        PCD_CHECK_HIP(hipMemsetAsync(alpha, 0, (size_t)P * NA * 4, s));
        PCD_CHECK_HIP(hipMemsetAsync(beta, 0, (size_t)P * NB * 4, s));
        const int ablocks = (int)ceil_div(NA, 256), bblocks = (int)ceil_div(NB, 256);
        hipLaunchKernelGGL(pair_cmax_kernel, dim3(ablocks, P), dim3(256), 0, s, an, na, NA, bn, nb, NB, cmax);
        const float lambda = 1.f / epsilon;
        for (int it = 0; it < max_iter; ++it) {
            hipLaunchKernelGGL(pair_sinkhorn_dual_kernel, dim3(ablocks, P), dim3(256), 0, s, an, na, NA, bn, nb, NB, cmax, lambda,
                               epsilon, log_mu, beta, alpha, err, P, it, 0, thresh);
            hipLaunchKernelGGL(pair_sinkhorn_dual_kernel, dim3(bblocks, P), dim3(256), 0, s, bn, nb, NB, an, na, NA, cmax, lambda,
                               epsilon, log_nu, alpha, beta, err, P, it, 1, thresh);
            hipLaunchKernelGGL(pair_sinkhorn_clear_kernel, dim3(1), dim3(256), 0, s, err, P, it + 1);
        }
        hipLaunchKernelGGL(pair_sinkhorn_cost_kernel, dim3(ablocks, P), dim3(256), 0, s, an, na, NA, bn, nb, NB, cmax, lambda, alpha,
                           beta, rowc);
        hipLaunchKernelGGL(pair_row_sum_kernel, dim3(P), dim3(256), 0, s, rowc, na, NA, rows);
        PCD_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(pair_empty_rows_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, s, na, nb, P, rows);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
