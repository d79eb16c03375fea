// fp32 PARITY MODE of a7, UNetPointNetLarge.forward (reference networks.py:779-818; SURVEY.md 8(c): "HIP fp32 parity
// mode: eps rel-L2 <= 1e-4 per forward, 50-step DDIM cloud max-abs <= 1e-3").
//
// The reference computes in fp32 throughout.  The product path (csrc/unet.hip) rounds operands to fp16; this file is
// the same network -- same weight folding (packing.py), same hoisting of the time / global-feature channels, same
// fused max over N -- with fp32 weights, fp32 activations and fp32 products on `v_mfma_f32_32x32x2_f32`, so that
// (a) north_star's "within a stated fp32 tolerance" has a path that is held to one, and (b) a long-horizon difference
// between the fp16 path and the reference can be bisected layer by layer (same taps as pcd_unet_tap, in fp32).
// Speed is not a goal here (the fp32 matrix peak is 1/16 of the fp16 one): one generic tiled GEMM, one launch per layer.
//
// Selected by `UNetPointNetLarge.set_precision("fp32")` / PCD_PARITY=fp32 on the Python side; the lin[] execution order
// and the descriptor are those of csrc/unet.hip (every `w` pointer fp32 here, `wg` included).
#include <string.h>
#include <new>
#include "common.h"

using namespace pcd;

namespace {

constexpr int TM = 128, TC = 128, TK = 16, LDT = TM + 4;    // LDS images are k-major: [k][row], row pitch 132 floats

struct GemmF32 {
    const float* a1; int64_t lda1; int k1;
    const float* a2; int64_t lda2; int k2;
    const float* w; int64_t ldw;
    const float* bias; const float* shape_bias; int rows_per_shape;
    int relu; int m; int c;
    float* out; int64_t ldo;          // store epilogue
    float* pooled;                    // column-max epilogue: [shape][c], zero-initialised (values are post-ReLU, >= 0)
    int round16;                      // diagnostic: round the stored activation to fp16 and back (where does ACTIVATION rounding matter?)
};

// out[m][c] = act([A1 | A2][m][k] . W[c][k]^T + bias[c] (or shape_bias[row / rows_per_shape][c]))
// 256 threads = 4 waves as 2 x 2; a wave owns 64 rows x 64 columns = 2 x 2 MFMA tiles of 32 x 32 (k = 2 per instruction).
template <int COLMAX>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32 p) {
    __shared__ float As[TK * LDT];
    __shared__ float Ws[TK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * TM;
    const int col0 = blockIdx.y * TC;
    const int kt = p.k1 + p.k2;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging: thread -> (row = tid / 4 (+ 64), 4 consecutive k at (tid & 3) * 4)
    const int sr = tid >> 2, sk = (tid & 3) * 4;
    for (int k0 = 0; k0 < kt; k0 += TK) {
        const bool second = k0 >= p.k1;
        const float* ab = second ? p.a2 : p.a1;
        const int64_t lda = second ? p.lda2 : p.lda1;
        const int ka = (second ? k0 - p.k1 : k0) + sk;
        f32x4 av[2], wv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t r = row0 + sr + h * 64;
            av[h] = r < p.m ? *(const f32x4*)(ab + r * lda + ka) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int cc = col0 + sr + h * 64;
            wv[h] = cc < p.c ? *(const f32x4*)(p.w + (int64_t)cc * p.ldw + k0 + sk) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();            // the previous K tile's fragment reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[(sk + e) * LDT + sr + h * 64] = av[h][e];
                Ws[(sk + e) * LDT + sr + h * 64] = wv[h][e];
            }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const int kl = kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kl * LDT + wr * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Ws[kl * LDT + wc * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // accumulator element e of lane l: row = (e / 4) * 8 + (l / 32) * 4 + (e % 4), column = l % 32
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + wc * 64 + j * 32 + (lane & 31);
        const bool col_ok = col < p.c;
        const float bcol = (col_ok && p.bias) ? p.bias[col] : 0.f;
        if (COLMAX) {
            // whole wave tile inside one shape and inside m: reduce its 64 rows in registers, one atomic per column
            const int64_t wrow0 = row0 + wr * 64;
            const bool whole = wrow0 + 63 < p.m && (wrow0 / p.rows_per_shape) == ((wrow0 + 63) / p.rows_per_shape);
            if (whole) {
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, acc[i][j][e] + bcol);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (col_ok && lane < 32)
                    atomicMax((unsigned*)(p.pooled + (wrow0 / p.rows_per_shape) * p.c + col), __float_as_uint(mx));
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t r = wrow0 + i * 32 + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
                        if (r < p.m && col_ok)
                            atomicMax((unsigned*)(p.pooled + (r / p.rows_per_shape) * p.c + col),
                                      __float_as_uint(fmaxf(acc[i][j][e] + bcol, 0.f)));
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t r = row0 + wr * 64 + i * 32 + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
                    if (r < p.m && col_ok) {
                        float v = acc[i][j][e];
                        v += p.shape_bias ? p.shape_bias[(r / p.rows_per_shape) * p.c + col] : bcol;
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (p.round16) v = (float)to_half_sat(v);
                        p.out[r * p.ldo + col] = v;
                    }
                }
        }
    }
}

// enc1.conv1, xyz half + per-shape time bias + ReLU: thread = (point, channel)
__global__ __launch_bounds__(256) void enc1_xyz_f32_kernel(const float* __restrict__ x, int64_t m, int rows_per_shape,
                                                            const float* __restrict__ w, int c1,
                                                            const float* __restrict__ tbias, int tb_stride,
                                                            float* __restrict__ out, int round16) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * c1) return;
    const int64_t pt = idx / c1;
    const int ch = (int)(idx - pt * c1);
    float v = tbias[(pt / rows_per_shape) * tb_stride * c1 + ch];
    v = fmaf(w[ch * 3 + 0], x[pt * 3 + 0], v);
    v = fmaf(w[ch * 3 + 1], x[pt * 3 + 1], v);
    v = fmaf(w[ch * 3 + 2], x[pt * 3 + 2], v);
    v = fmaxf(v, 0.f);
    out[idx] = round16 ? (float)to_half_sat(v) : v;
}

// output.3: eps[m][0..2] = W3 . h[m] + b3, one thread per point
__global__ __launch_bounds__(256) void head3_f32_kernel(const float* __restrict__ h, int64_t m, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ eps) {
    __shared__ float ws[3 * 64];
    for (int i = threadIdx.x; i < 3 * 64; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const f32x4* row = (const f32x4*)(h + pt * 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const f32x4 v = row[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0 = fmaf(ws[q * 4 + e], v[e], a0);
            a1 = fmaf(ws[64 + q * 4 + e], v[e], a1);
            a2 = fmaf(ws[128 + q * 4 + e], v[e], a2);
        }
    }
    eps[pt * 3 + 0] = a0 + b[0];
    eps[pt * 3 + 1] = a1 + b[1];
    eps[pt * 3 + 2] = a2 + b[2];
}

int launch_gemm(const GemmF32& g, bool colmax, hipStream_t s) {
    PCD_CHECK_ARG(g.a1 && g.w && g.m > 0 && g.c > 0);
    PCD_CHECK_ARG(g.k1 > 0 && g.k1 % TK == 0 && g.k2 >= 0 && g.k2 % TK == 0 && (g.k2 == 0 || g.a2));
    PCD_CHECK_ARG(g.lda1 % 4 == 0 && g.lda2 % 4 == 0 && g.ldw % 4 == 0);
    PCD_CHECK_ARG((g.shape_bias == nullptr && !colmax) || g.rows_per_shape > 0);
    const dim3 grid((unsigned)ceil_div(g.m, TM), (unsigned)ceil_div(g.c, TC));
    if (colmax) {
        PCD_CHECK_ARG(g.pooled && g.bias);
        hipLaunchKernelGGL((gemm_f32_kernel<1>), grid, dim3(256), 0, s, g);
    } else {
        PCD_CHECK_ARG(g.out && g.ldo >= g.c);
        hipLaunchKernelGGL((gemm_f32_kernel<0>), grid, dim3(256), 0, s, g);
    }
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

struct Ws32 {
    size_t x1, x2, x3, x4, s0, s1, pooled, gbias, d4, d3, d2, d1, total;
};

size_t up(size_t v) { return (v + 255) / 256 * 256; }

Ws32 carve32(int64_t batch, int64_t n) {
    const size_t m = (size_t)batch * (size_t)n;
    Ws32 w{};
    size_t o = 0;
    w.x1 = o; o += up(m * 128 * 4);
    w.x2 = o; o += up(m * 256 * 4);
    w.x3 = o; o += up(m * 512 * 4);
    w.x4 = o; o += up(m * 1024 * 4);
    w.s0 = o; o += up(m * 2048 * 4);
    w.s1 = o; o += up(m * 1024 * 4);
    w.pooled = o; o += up((size_t)batch * 4096 * 4);
    w.gbias = o; o += up((size_t)batch * 1024 * 4);
    // decoder-block outputs are kept (not ping-ponged away) so that every tap of a forward can be read afterwards
    w.d4 = o; o += up(m * 512 * 4);
    w.d3 = o; o += up(m * 256 * 4);
    w.d2 = o; o += up(m * 128 * 4);
    w.d1 = o; o += up(m * 64 * 4);
    w.total = o;
    return w;
}

}  // namespace

struct pcd_unet_f32 {
    pcd_unet_desc_t d;
    unsigned round_mask = 0;          // pcd_unet_f32_round_activations: bit i = round lin[i]'s output to fp16 (bit 26: enc1.conv1's)
};

extern "C" int pcd_gemm_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2, const float* w,
                            int64_t ldw, const float* bias, const float* shape_bias, int rows_per_shape, int relu, int m,
                            int c, float* out, int64_t ldo, void* stream) {
    GemmF32 g{a1, lda1, k1, a2, lda2, k2, w, ldw, bias, shape_bias, rows_per_shape, relu, m, c, out, ldo, nullptr};
    return launch_gemm(g, false, (hipStream_t)stream);
}

extern "C" int pcd_gemm_f32_colmax(const float* a, int64_t lda, int k, const float* w, int64_t ldw, const float* bias, int m,
                                   int c, float* pooled, int rows_per_shape, void* stream) {
    GemmF32 g{a, lda, k, nullptr, 0, 0, w, ldw, bias, nullptr, rows_per_shape, 1, m, c, nullptr, 0, pooled};
    return launch_gemm(g, true, (hipStream_t)stream);
}

extern "C" int pcd_unet_f32_create(const pcd_unet_desc_t* desc, pcd_unet_f32_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    PCD_CHECK_ARG(desc->e1w_xyz && desc->wg && desc->head_w && desc->head_b);
    PCD_CHECK_ARG(desc->wg_k == 4096 && desc->wg_c == 1024);
    for (int i = 0; i < PCD_UNET_NLIN; ++i) PCD_CHECK_ARG(desc->lin[i].w && desc->lin[i].b && desc->lin[i].k % TK == 0);
    pcd_unet_f32* h = new (std::nothrow) pcd_unet_f32;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_unet_f32_destroy(pcd_unet_f32_t* h) { delete h; }

extern "C" int pcd_unet_f32_round_activations(pcd_unet_f32_t* h, unsigned mask) {
    PCD_CHECK_ARG(h != nullptr);
    h->round_mask = mask;
    return PCD_OK;
}

extern "C" size_t pcd_unet_f32_workspace_bytes(int batch, int n_points) {
    if (batch <= 0 || n_points <= 0) return 0;
    return carve32(batch, n_points).total;
}

extern "C" int pcd_unet_f32_forward(pcd_unet_f32_t* h, const float* x, int batch, int n_points, const float* tbias,
                                    int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    PCD_CHECK_ARG(h && x && tbias && eps && workspace);
    PCD_CHECK_ARG(batch > 0 && n_points > 0 && tbias_shape_stride >= 0);
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const Ws32 w = carve32(batch, n_points);
    if (workspace_bytes < w.total) {
        set_error("pcd_unet_f32_forward: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    float *x1 = F(w.x1), *x2 = F(w.x2), *x3 = F(w.x3), *x4 = F(w.x4), *s0 = F(w.s0), *s1 = F(w.s1);
    float *pooled = F(w.pooled), *gbias = F(w.gbias), *d4 = F(w.d4), *d3 = F(w.d3), *d2 = F(w.d2), *d1 = F(w.d1);
    hipStream_t s = (hipStream_t)stream;
    const pcd_unet_desc_t& d = h->d;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
    auto lin = [&](int idx, const float* a1, const float* a2, int k2, const float* shape_bias, float* out) {
        const pcd_linear_desc_t& L = d.lin[idx];
        GemmF32 g{a1, L.k - k2, L.k - k2, a2, k2, k2, (const float*)L.w, L.k, shape_bias ? nullptr : L.b, shape_bias,
                  n_points, 1, (int)m, L.c, out, L.c, nullptr, (int)((h->round_mask >> idx) & 1u)};
        return launch_gemm(g, false, s);
    };
    hipLaunchKernelGGL(enc1_xyz_f32_kernel, dim3((unsigned)ceil_div(m * 64, 256)), dim3(256), 0, s, x, m, n_points,
                       d.e1w_xyz, 64, tbias, tbias_shape_stride, s0, (int)((h->round_mask >> 26) & 1u));
    PCD_CHECK_LAUNCH();
    RUN(lin(0, s0, nullptr, 0, nullptr, s1));
    RUN(lin(1, s1, nullptr, 0, nullptr, x1));
    RUN(lin(2, x1, nullptr, 0, nullptr, s0));
    RUN(lin(3, s0, nullptr, 0, nullptr, s1));
    RUN(lin(4, s1, nullptr, 0, nullptr, x2));
    RUN(lin(5, x2, nullptr, 0, nullptr, s0));
    RUN(lin(6, s0, nullptr, 0, nullptr, s1));
    RUN(lin(7, s1, nullptr, 0, nullptr, x3));
    RUN(lin(8, x3, nullptr, 0, nullptr, s0));
    RUN(lin(9, s0, nullptr, 0, nullptr, s1));
    RUN(lin(10, s1, nullptr, 0, nullptr, x4));
    RUN(lin(11, x4, nullptr, 0, nullptr, s0));
    PCD_CHECK_HIP(hipMemsetAsync(pooled, 0, (size_t)batch * 4096 * sizeof(float), s));
    {   // global_feat.3 + max over the N points of each shape
        GemmF32 g{s0, 2048, 2048, nullptr, 0, 0, (const float*)d.lin[12].w, 2048, d.lin[12].b, nullptr, n_points, 1, (int)m,
                  4096, nullptr, 0, pooled};
        RUN(launch_gemm(g, true, s));
    }
    {   // hoisted global half of dec4.conv1: per-shape bias [B][1024] = pooled . Wg^T + folded bias
        GemmF32 g{pooled, 4096, 4096, nullptr, 0, 0, (const float*)d.wg, 4096, d.lin[13].b, nullptr, 0, 0, batch, 1024, gbias,
                  1024, nullptr};
        RUN(launch_gemm(g, false, s));
    }
    RUN(lin(13, x4, nullptr, 0, gbias, s1));
    RUN(lin(14, s1, nullptr, 0, nullptr, s0));
    RUN(lin(15, s0, nullptr, 0, nullptr, d4));
    RUN(lin(16, d4, x3, 512, nullptr, s0));
    RUN(lin(17, s0, nullptr, 0, nullptr, s1));
    RUN(lin(18, s1, nullptr, 0, nullptr, d3));
    RUN(lin(19, d3, x2, 256, nullptr, s0));
    RUN(lin(20, s0, nullptr, 0, nullptr, s1));
    RUN(lin(21, s1, nullptr, 0, nullptr, d2));
    RUN(lin(22, d2, x1, 128, nullptr, s0));
    RUN(lin(23, s0, nullptr, 0, nullptr, s1));
    RUN(lin(24, s1, nullptr, 0, nullptr, d1));
    RUN(lin(25, d1, nullptr, 0, nullptr, s0));
    hipLaunchKernelGGL(head3_f32_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s, s0, m, d.head_w, d.head_b, eps);
    PCD_CHECK_LAUNCH();
#undef RUN
    return PCD_OK;
}

extern "C" int pcd_unet_f32_tap(pcd_unet_f32_t* h, const char* name, int batch, int n_points, const void* workspace,
                                void* dst, size_t dst_bytes, void* stream) {
    PCD_CHECK_ARG(h && name && workspace && dst && batch > 0 && n_points > 0);
    const Ws32 w = carve32(batch, n_points);
    const size_t m = (size_t)batch * n_points;
    const struct { const char* n; size_t off, bytes; } taps[] = {
        {"x1", w.x1, m * 128 * 4}, {"x2", w.x2, m * 256 * 4}, {"x3", w.x3, m * 512 * 4}, {"x4", w.x4, m * 1024 * 4},
        {"pooled", w.pooled, (size_t)batch * 4096 * 4}, {"gbias", w.gbias, (size_t)batch * 1024 * 4},
        {"d4", w.d4, m * 512 * 4}, {"d3", w.d3, m * 256 * 4}, {"d2", w.d2, m * 128 * 4}, {"d1", w.d1, m * 64 * 4}};
    for (const auto& t : taps) {
        if (strcmp(name, t.n)) continue;
        PCD_CHECK_ARG(dst_bytes >= t.bytes);
        PCD_CHECK_HIP(hipMemcpyAsync(dst, (const char*)workspace + t.off, t.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return PCD_OK;
    }
    set_error("pcd_unet_f32_tap: unknown tap '%s'", name);
    return PCD_ERR_ARG;
}
