// K10/K11: evaluation kernels -- normalize_to_cube, Chamfer, voxelize, voxel->points.
// All HBM/latency-bound integer-and-compare work; no MFMA (reference metrics.py:7-47,
// utils.py:488-539).  fp32 expressions keep the reference's operation order (contract off)
// so normalisation, voxel indices and compacted coordinates are bit-exact.
#include "common.h"

namespace pcd {

#pragma clang fp contract(off)

template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T* scratch /* [blockDim/64] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = op(r, scratch[w]);
    return r;
}

struct MaxOp { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct MinOp { __device__ float operator()(float a, float b) const { return fminf(a, b); } };

// one block per cloud
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ pts, int n, float* __restrict__ out) {
    __shared__ float scratch[4];
    const float* p = pts + (int64_t)blockIdx.x * n * 3;
    float* o = out + (int64_t)blockIdx.x * n * 3;
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mn[3] = {INFINITY, INFINITY, INFINITY};
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[i * 3 + a];
            mx[a] = fmaxf(mx[a], v);
            mn[a] = fminf(mn[a], v);
        }
    float c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float hi = block_reduce(mx[a], MaxOp(), scratch);
        const float lo = block_reduce(mn[a], MinOp(), scratch);
        c[a] = (hi + lo) / 2.f;   // metrics.py:17
    }
    float am = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) am = fmaxf(am, fabsf(p[i * 3 + a] - c[a]));
    const float scale = block_reduce(am, MaxOp(), scratch);   // metrics.py:19
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) o[i * 3 + a] = (p[i * 3 + a] - c[a]) / scale;
}

// grid (B, 2): direction 0 = queries x against refs y, 1 = queries y against refs x.
// thread per query, refs streamed through LDS in tiles; deterministic block-tree sum.
constexpr int CH_TILE = 1024;
__global__ __launch_bounds__(1024) void chamfer_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        int n1, int n2, float* __restrict__ sums) {
    __shared__ float tile[CH_TILE * 3];
    __shared__ float wsum[16];
    const int b = blockIdx.x, dir = blockIdx.y;
    const float* q = dir == 0 ? x + (int64_t)b * n1 * 3 : y + (int64_t)b * n2 * 3;
    const float* r = dir == 0 ? y + (int64_t)b * n2 * 3 : x + (int64_t)b * n1 * 3;
    const int nq = dir == 0 ? n1 : n2, nr = dir == 0 ? n2 : n1;
    float acc = 0.f;
    for (int q0 = 0; q0 < nq; q0 += blockDim.x) {
        const int qi = q0 + threadIdx.x;
        const bool live = qi < nq;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (live) { qx = q[qi * 3]; qy = q[qi * 3 + 1]; qz = q[qi * 3 + 2]; }
        float best = INFINITY;
        for (int r0 = 0; r0 < nr; r0 += CH_TILE) {
            const int cnt = min(CH_TILE, nr - r0);
            __syncthreads();
            for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) tile[i] = r[(int64_t)r0 * 3 + i];
            __syncthreads();
            if (live) {
#pragma unroll 4
                for (int j = 0; j < cnt; ++j) {
                    const float dx = qx - tile[j * 3], dy = qy - tile[j * 3 + 1], dz = qz - tile[j * 3 + 2];
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    best = fminf(best, d2);
                }
            }
        }
        if (live) acc += sqrtf(best);
    }
    // deterministic reduction: wave shuffle tree then fixed-order sum over waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
        sums[b * 2 + dir] = t;
    }
}

__global__ __launch_bounds__(256) void voxelize_kernel(const float* __restrict__ pts, int64_t total_pts, int n, int res,
                                                        float* __restrict__ vox) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_pts) return;
    const int b = (int)(i / n);
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = ((pts[i * 3 + a] + 1.f) * (float)(res - 1)) / 2.f;   // utils.py:501
        long long t = (v == v) ? (long long)v : 0;                            // .long(): truncate toward zero
        t = t < 0 ? 0 : (t > res - 1 ? res - 1 : t);
        idx[a] = (int)t;
    }
    vox[(((int64_t)b * res + idx[0]) * res + idx[1]) * res + idx[2]] = 1.f;   // utils.py:507, index order [x][y][z]
}

// one block per grid: ordered stream compaction (wave ballot + block scan of wave counts)
__global__ __launch_bounds__(1024) void voxels_to_points_kernel(const float* __restrict__ vox, int d, int h, int w,
                                                                 float thr, int32_t* __restrict__ counts,
                                                                 float* __restrict__ points) {
    __shared__ int wcount[16];
    __shared__ int base;
    const int64_t nvox = (int64_t)d * h * w;
    const float* v = vox + (int64_t)blockIdx.x * nvox;
    float* out = points + (int64_t)blockIdx.x * nvox * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const float fw = (float)(w - 1), fh = (float)(h - 1), fd = (float)(d - 1);
    for (int64_t i0 = 0; i0 < nvox; i0 += blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const bool on = i < nvox && v[i] > thr;
        const unsigned long long m = __ballot(on);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wave; ++k) off += wcount[k];
        if (on) {
            const int ix = (int)(i % w), iy = (int)((i / w) % h), iz = (int)(i / ((int64_t)w * h));
            float* o = out + (int64_t)(off + before) * 3;
            o[0] = (2.f * (float)ix) / fw - 1.f;   // utils.py:533
            o[1] = (2.f * (float)iy) / fh - 1.f;
            o[2] = (2.f * (float)iz) / fd - 1.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int k = 0; k < nw; ++k) t += wcount[k];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = base;
}

// mean binary cross entropy with torch's log clamp at -100 (F.binary_cross_entropy, metrics.py:181);
// single block, double accumulation in a fixed order => deterministic
__global__ __launch_bounds__(1024) void bce_mean_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                         int64_t n, float* __restrict__ out) {
    __shared__ double wsum[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float xv = x[i], tv = t[i];
        const float l1 = fmaxf(logf(xv), -100.f), l0 = fmaxf(logf(1.f - xv), -100.f);
        acc += (double)(-(tv * l1 + (1.f - tv) * l0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        out[0] = (float)(tot / (double)n);
    }
}
#pragma clang fp contract(fast)

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_normalize_to_cube(const float* pts, int batch, int n, float* out, void* stream) {
    PCD_CHECK_ARG(pts && out && batch > 0 && n > 0);
    hipLaunchKernelGGL(normalize_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, pts, n, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_chamfer_sums(const float* x, const float* y, int batch, int n1, int n2, float* sums, void* stream) {
    PCD_CHECK_ARG(x && y && sums && batch > 0 && n1 > 0 && n2 > 0);
    hipLaunchKernelGGL(chamfer_kernel, dim3(batch, 2), dim3(1024), 0, (hipStream_t)stream, x, y, n1, n2, sums);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_voxelize(const float* pts, int batch, int n, int res, float* vox, void* stream) {
    PCD_CHECK_ARG(pts && vox && batch > 0 && n > 0 && res > 1);
    const int64_t total = (int64_t)batch * n;
    hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       pts, total, n, res, vox);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_voxels_to_points(const float* vox, int batch, int d, int h, int w, float threshold,
                                    int32_t* counts, float* points, void* stream) {
    PCD_CHECK_ARG(vox && counts && points && batch > 0 && d > 1 && h > 1 && w > 1);
    hipLaunchKernelGGL(voxels_to_points_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream,
                       vox, d, h, w, threshold, counts, points);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_binary_bce_mean(const float* x, const float* target, int64_t n, float* out, void* stream) {
    PCD_CHECK_ARG(x && target && out && n > 0);
    hipLaunchKernelGGL(bce_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, target, n, out);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}
