// fp32 PARITY MODE of a9 / a10: SetAttentionBlock.forward (reference networks.py:70-83) and UNetAttentionPointExperimental.forward
// (networks.py:652-704) with fp32 weights, fp32 activations and fp32 arithmetic throughout -- the reference's arithmetic type.
//
// The product path (csrc/attention.hip, csrc/attn_unet.hip) runs fp16 operands on the matrix cores and is held to 3e-3 per block / 5e-3 per
// network; this file is the same sequencing (same BatchNorm folding, same time-bias rows, same layer order: csrc/attn_unet.hip) on
// pcd_gemm_f32 (csrc/unet_f32.hip), an fp32 LayerNorm and a plain fp32 softmax(Q K^T / sqrt d) V kernel, held to 1e-4.  It exists so that
// the attention backbone, too, has a path that is compared with the reference at its own precision; speed is not a goal (VALU attention).
// Selected from Python by `set_precision("fp32")` / PCD_PARITY=fp32 on SetAttentionBlock / UNetAttentionPointExperimental; every weight
// pointer of the descriptors is fp32 here.
#include <string.h>
#include <new>
#include "common.h"

using namespace pcd;

namespace {

size_t up(size_t v) { return (v + 255) / 256 * 256; }

// LayerNorm over C (eps 1e-5, biased variance, affine): one wave per row, C <= 256
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int64_t rows, int c, const float* __restrict__ g,
                                                             const float* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * c;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ch = lane + 64 * i; v[i] = ch < c ? xr[ch] : 0.f; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ch = lane + 64 * i; if (ch < c) { const float d = v[i] - mean; q += d * d; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)c + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ch = lane + 64 * i; if (ch < c) out[row * c + ch] = (v[i] - mean) * rstd * g[ch] + b[ch]; }
}

// softmax(q k^T / sqrt(d)) v per (shape, head): qkv fp32 [B*N][3C] (q | k | v, head h = columns h*d .. of each third), out fp32 [B*N][C].
// Workgroup = 64 queries x 4 key quarters: thread (query, part) walks the keys j % 4 == part of every 64-key tile (K / V tile in LDS) with its own
// running (max, sum, o[D]); the four parts of a query are merged through LDS at the end.  q is scaled BEFORE the products, as nn.MultiheadAttention does.
template <int D>
__global__ __launch_bounds__(256) void set_attention_f32_kernel(const float* __restrict__ qkv, int n, int c, int heads, float* __restrict__ out) {
    __shared__ float ks[64][D + 1];
    __shared__ float vs[64][D + 1];
    __shared__ float red[4][64][D + 2];
    const int tid = threadIdx.x, ql = tid & 63, part = tid >> 6;
    const int b = blockIdx.y / heads, h = blockIdx.y - b * heads;
    const int q0 = blockIdx.x * 64;
    const int64_t base = (int64_t)b * n;
    const int qi = q0 + ql;
    float q[D], o[D];
    const float scale = 1.f / sqrtf((float)D);
#pragma unroll
    for (int e = 0; e < D; ++e) { q[e] = qi < n ? qkv[(base + qi) * 3 * c + h * D + e] * scale : 0.f; o[e] = 0.f; }
    float mx = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < n; k0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * D; i += 256) {
            const int r = i / D, e = i - r * D;
            const bool ok = k0 + r < n;
            ks[r][e] = ok ? qkv[(base + k0 + r) * 3 * c + c + h * D + e] : 0.f;
            vs[r][e] = ok ? qkv[(base + k0 + r) * 3 * c + 2 * c + h * D + e] : 0.f;
        }
        __syncthreads();
        for (int j = part; j < 64 && k0 + j < n; j += 4) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < D; ++e) s = fmaf(q[e], ks[j][e], s);
            const float mn = fmaxf(mx, s);
            const float corr = expf(mx - mn), p = expf(s - mn);       // first key: exp(-inf) = 0
            l = l * corr + p;
#pragma unroll
            for (int e = 0; e < D; ++e) o[e] = fmaf(p, vs[j][e], o[e] * corr);
            mx = mn;
        }
    }
    red[part][ql][D] = mx;
    red[part][ql][D + 1] = l;
#pragma unroll
    for (int e = 0; e < D; ++e) red[part][ql][e] = o[e];
    __syncthreads();
    if (part == 0 && qi < n) {
        float m = red[0][ql][D];
#pragma unroll
        for (int p = 1; p < 4; ++p) m = fmaxf(m, red[p][ql][D]);
        float ls = 0.f, w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { w[p] = red[p][ql][D + 1] > 0.f ? expf(red[p][ql][D] - m) : 0.f; ls += w[p] * red[p][ql][D + 1]; }
#pragma unroll
        for (int e = 0; e < D; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) acc += w[p] * red[p][ql][e];
            out[(base + qi) * c + h * D + e] = acc / ls;
        }
    }
}

__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

// x[m][c] + e[(m / rows_per_shape) * e_stride + c]
__global__ __launch_bounds__(256) void add_shape_bias_f32_kernel(const float* __restrict__ x, int64_t m, int c, int rows_per_shape,
                                                                  const float* __restrict__ e, int64_t e_stride, float* __restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * c) return;
    const int64_t row = i / c;
    o[i] = x[i] + e[(row / rows_per_shape) * e_stride + (i - row * c)];
}

// enc1.conv1 (K = 3) + the per-shape time-bias row + ReLU (the arithmetic of pcd_enc1_xyz, fp32 out)
__global__ __launch_bounds__(256) void au_enc1_f32_kernel(const float* __restrict__ x, int64_t m, int rows_per_shape, const float* __restrict__ w,
                                                          const float* __restrict__ tbias, int64_t tb_stride, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * 64) return;
    const int64_t pt = idx >> 6;
    const int ch = (int)(idx & 63);
    float v = tbias[(pt / rows_per_shape) * tb_stride + ch];
    v = fmaf(w[ch * 3 + 0], x[pt * 3 + 0], v);
    v = fmaf(w[ch * 3 + 1], x[pt * 3 + 1], v);
    v = fmaf(w[ch * 3 + 2], x[pt * 3 + 2], v);
    out[idx] = fmaxf(v, 0.f);
}

// dec1 = PointNetLayer(128, 3, 3) on cat[a | b] + output Conv1d(3, 3) (the arithmetic of pcd_tail3, fp32 inputs)
__global__ __launch_bounds__(256) void au_tail3_f32_kernel(const float* __restrict__ a, int ka, const float* __restrict__ b, int kb, int64_t m,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w234, const float* __restrict__ b234,
                                                           float* __restrict__ out) {
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= m) return;
    const int k = ka + kb;
    float h[3] = {b1[0], b1[1], b1[2]};
    for (int i = 0; i < ka; ++i) {
        const float f = a[pt * ka + i];
        for (int j = 0; j < 3; ++j) h[j] = fmaf(w1[j * k + i], f, h[j]);
    }
    for (int i = 0; i < kb; ++i) {
        const float f = b[pt * kb + i];
        for (int j = 0; j < 3; ++j) h[j] = fmaf(w1[j * k + ka + i], f, h[j]);
    }
    for (int j = 0; j < 3; ++j) h[j] = fmaxf(h[j], 0.f);
    for (int layer = 0; layer < 3; ++layer) {
        float o[3];
        for (int j = 0; j < 3; ++j) {
            float acc = b234[layer * 3 + j];
            for (int i = 0; i < 3; ++i) acc = fmaf(w234[layer * 9 + j * 3 + i], h[i], acc);
            o[j] = layer < 2 ? fmaxf(acc, 0.f) : acc;
        }
        for (int j = 0; j < 3; ++j) h[j] = o[j];
    }
    out[pt * 3 + 0] = h[0]; out[pt * 3 + 1] = h[1]; out[pt * 3 + 2] = h[2];
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

struct SabWs { size_t t1, t2, qkv, ffh, total; };
SabWs sab_carve(int64_t rows, int dim) {
    SabWs w{};
    size_t o = 0;
    w.t1 = o; o += up((size_t)rows * dim * 4);
    w.t2 = o; o += up((size_t)rows * dim * 4);
    w.qkv = o; o += up((size_t)rows * 3 * dim * 4);
    w.ffh = o; o += up((size_t)rows * 4 * dim * 4);
    w.total = o;
    return w;
}

int lin(const float* a1, int k1, const float* a2, int k2, const void* w, const float* b, int relu, int64_t m, int c, float* out, hipStream_t s) {
    return pcd_gemm_f32(a1, k1, k1, a2, k2, k2, (const float*)w, k1 + k2, b, nullptr, 0, relu, (int)m, c, out, c, s);
}

int attention(const float* qkv, int batch, int n, int c, int heads, float* out, hipStream_t s) {
    const int d = c / heads;
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)(batch * heads));
    if (d == 16) hipLaunchKernelGGL((set_attention_f32_kernel<16>), grid, dim3(256), 0, s, qkv, n, c, heads, out);
    else if (d == 32) hipLaunchKernelGGL((set_attention_f32_kernel<32>), grid, dim3(256), 0, s, qkv, n, c, heads, out);
    else if (d == 64) hipLaunchKernelGGL((set_attention_f32_kernel<64>), grid, dim3(256), 0, s, qkv, n, c, heads, out);
    else { set_error("set attention (fp32): head width %d not in {16, 32, 64}", d); return PCD_ERR_ARG; }
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

// x += MHA(LN1 x); x += W2 relu(W1 LN2 x)   (networks.py:80-83), y != x
int sab_run(const pcd_sab_desc_t& d, const float* x, int batch, int n, int heads, float* y, char* ws, hipStream_t s) {
    const int64_t m = (int64_t)batch * n;
    const int C = d.dim;
    const SabWs w = sab_carve(m, C);
    float *t1 = (float*)(ws + w.t1), *t2 = (float*)(ws + w.t2), *qkv = (float*)(ws + w.qkv), *ffh = (float*)(ws + w.ffh);
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, x, m, C, d.ln1_g, d.ln1_b, t1);
    RUN(lin(t1, C, nullptr, 0, d.w_in, d.b_in, 0, m, 3 * C, qkv, s));
    RUN(attention(qkv, batch, n, C, heads, t2, s));
    RUN(lin(t2, C, nullptr, 0, d.w_out, d.b_out, 0, m, C, t1, s));
    hipLaunchKernelGGL(add_f32_kernel, dim3(nblk(m * C)), dim3(256), 0, s, x, t1, t1, m * C);                 // t1 = x + out_proj(.)
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, t1, m, C, d.ln2_g, d.ln2_b, t2);
    RUN(lin(t2, C, nullptr, 0, d.w_ff1, d.b_ff1, 1, m, 4 * C, ffh, s));
    RUN(lin(ffh, 4 * C, nullptr, 0, d.w_ff2, d.b_ff2, 0, m, C, t2, s));
    hipLaunchKernelGGL(add_f32_kernel, dim3(nblk(m * C)), dim3(256), 0, s, t1, t2, y, m * C);
    PCD_CHECK_LAUNCH();
#undef RUN
    return PCD_OK;
}

bool sab_ok(const pcd_sab_desc_t& d) {
    return (d.dim == 64 || d.dim == 128 || d.dim == 256) && d.w_in && d.b_in && d.w_out && d.b_out && d.ln1_g && d.ln1_b && d.ln2_g &&
           d.ln2_b && d.w_ff1 && d.b_ff1 && d.w_ff2 && d.b_ff2;
}

struct AuWs { size_t x1, x2, x3, p0, p1, p2, sab, total; };
AuWs au_carve(int64_t batch, int64_t n) {
    const size_t m = (size_t)batch * (size_t)n;
    AuWs w{};
    size_t o = 0;
    w.x1 = o; o += up(m * 64 * 4);
    w.x2 = o; o += up(m * 128 * 4);
    w.x3 = o; o += up(m * 256 * 4);
    w.p0 = o; o += up(m * 256 * 4);
    w.p1 = o; o += up(m * 256 * 4);
    w.p2 = o; o += up(m * 256 * 4);
    w.sab = o; o += sab_carve((int64_t)m, 256).total;
    w.total = o;
    return w;
}

}  // namespace

struct pcd_attn_unet_f32 {
    pcd_attn_unet_desc_t d;
};

extern "C" size_t pcd_sab_f32_workspace_bytes(int64_t rows, int dim) { return rows > 0 && dim > 0 ? sab_carve(rows, dim).total : 0; }

extern "C" int pcd_sab_f32_forward(const pcd_sab_desc_t* d, const float* x, int batch, int n_points, int heads, float* y, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(d && x && y && workspace && x != y && batch > 0 && n_points > 0 && heads > 0);
    PCD_CHECK_ARG(sab_ok(*d) && d->dim % heads == 0);
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const size_t need = sab_carve(m, d->dim).total;
    if (workspace_bytes < need) {
        set_error("pcd_sab_f32_forward: workspace %zu < required %zu", workspace_bytes, need);
        return PCD_ERR_WORKSPACE;
    }
    return sab_run(*d, x, batch, n_points, heads, y, (char*)workspace, (hipStream_t)stream);
}

extern "C" int pcd_attn_unet_f32_create(const pcd_attn_unet_desc_t* desc, pcd_attn_unet_f32_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr && desc->heads > 0 && desc->e1w && desc->t_w1 && desc->t_b1 && desc->t_w234 && desc->t_b234);
    for (int i = 0; i < PCD_ATTN_UNET_NLIN; ++i) PCD_CHECK_ARG(desc->lin[i].w && desc->lin[i].b && desc->lin[i].k % 16 == 0);
    for (int i = 0; i < PCD_ATTN_UNET_NSAB; ++i) PCD_CHECK_ARG(sab_ok(desc->sab[i]) && desc->sab[i].dim % desc->heads == 0);
    pcd_attn_unet_f32* h = new (std::nothrow) pcd_attn_unet_f32;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_attn_unet_f32_destroy(pcd_attn_unet_f32_t* h) { delete h; }

extern "C" size_t pcd_attn_unet_f32_workspace_bytes(int batch, int n_points) {
    return batch > 0 && n_points > 0 ? au_carve(batch, n_points).total : 0;
}

// tbias: the 704-float rows of pcd_attn_unet_time_bias (the time path is fp32 in both modes)
extern "C" int pcd_attn_unet_f32_forward(pcd_attn_unet_f32_t* h, const float* x, int batch, int n_points, const float* tbias,
                                         int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(h && x && tbias && eps && workspace);
    PCD_CHECK_ARG(batch > 0 && n_points > 0 && (tbias_shape_stride == 0 || tbias_shape_stride == 1));
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const AuWs w = au_carve(batch, n_points);
    if (workspace_bytes < w.total) {
        set_error("pcd_attn_unet_f32_forward: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    float *x1 = F(w.x1), *x2 = F(w.x2), *x3 = F(w.x3), *p0 = F(w.p0), *p1 = F(w.p1), *p2 = F(w.p2);
    char* sws = ws + w.sab;
    hipStream_t s = (hipStream_t)stream;
    const pcd_attn_unet_desc_t& d = h->d;
    const int H = d.heads, N = n_points;
    const int64_t estr = (int64_t)tbias_shape_stride * PCD_ATTN_UNET_TB;
    const int rps = tbias_shape_stride ? N : (int)m;
    const float *tb_e1 = tbias, *tb_e2 = tbias + 64, *tb_e3 = tbias + 128, *tb_d3 = tbias + 256, *tb_d2 = tbias + 512, *tb_d1 = tbias + 640;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
#define LIN(i, a1, a2, k2, out) RUN(lin(a1, d.lin[i].k - (k2), a2, k2, d.lin[i].w, d.lin[i].b, 1, m, d.lin[i].c, out, s))
#define EMB(src, c, tb, dst) hipLaunchKernelGGL(add_shape_bias_f32_kernel, dim3(nblk(m * (c))), dim3(256), 0, s, src, m, c, rps, tb, estr, dst)
    hipLaunchKernelGGL(au_enc1_f32_kernel, dim3(nblk(m * 64)), dim3(256), 0, s, x, m, rps, d.e1w, tb_e1, estr, p0);
    LIN(0, p0, nullptr, 0, p1);
    LIN(1, p1, nullptr, 0, p0);
    RUN(sab_run(d.sab[0], p0, batch, N, H, p1, sws, s));                                  // att1
    EMB(p1, 64, tb_e2, x1);                                                               // x1 + emb2
    LIN(2, x1, nullptr, 0, p0); LIN(3, p0, nullptr, 0, p1); LIN(4, p1, nullptr, 0, p0);   // enc2
    RUN(sab_run(d.sab[1], p0, batch, N, H, p1, sws, s));                                  // att2
    EMB(p1, 128, tb_e3, x2);                                                              // x2 + emb3
    LIN(5, x2, nullptr, 0, p0); LIN(6, p0, nullptr, 0, p1); LIN(7, p1, nullptr, 0, p0);   // enc3
    RUN(sab_run(d.sab[2], p0, batch, N, H, x3, sws, s));                                  // att3 -> x3
    RUN(sab_run(d.sab[3], x3, batch, N, H, p0, sws, s));                                  // bottleneck
    EMB(p0, 256, tb_d3, p1);
    RUN(sab_run(d.sab[4], p1, batch, N, H, p0, sws, s));                                  // att_dec3
    LIN(8, p0, x3, 256, p1); LIN(9, p1, nullptr, 0, p2); LIN(10, p2, nullptr, 0, p1);     // dec3 on cat[xb | x3]
    EMB(p1, 128, tb_d2, p0);
    RUN(sab_run(d.sab[5], p0, batch, N, H, p1, sws, s));                                  // att_dec2
    LIN(11, p1, x2, 128, p0); LIN(12, p0, nullptr, 0, p2); LIN(13, p2, nullptr, 0, p0);   // dec2 on cat[. | x2]
    EMB(p0, 64, tb_d1, p1);
    RUN(sab_run(d.sab[6], p1, batch, N, H, p0, sws, s));                                  // att_dec1
    hipLaunchKernelGGL(au_tail3_f32_kernel, dim3(nblk(m)), dim3(256), 0, s, p0, 64, x1, 64, m, d.t_w1, d.t_b1, d.t_w234, d.t_b234, eps);
    PCD_CHECK_LAUNCH();
#undef EMB
#undef LIN
#undef RUN
    return PCD_OK;
}

extern "C" int pcd_attn_unet_f32_tap(pcd_attn_unet_f32_t* h, const char* name, int batch, int n_points, const void* workspace, void* dst,
                                     size_t dst_bytes, void* stream) {
    PCD_CHECK_ARG(h && name && workspace && dst && batch > 0 && n_points > 0);
    const AuWs w = au_carve(batch, n_points);
    const size_t m = (size_t)batch * n_points;
    size_t off = 0, bytes = 0;
    if (!strcmp(name, "x1")) { off = w.x1; bytes = m * 64 * 4; }
    else if (!strcmp(name, "x2")) { off = w.x2; bytes = m * 128 * 4; }
    else if (!strcmp(name, "x3")) { off = w.x3; bytes = m * 256 * 4; }
    else { set_error("pcd_attn_unet_f32_tap: unknown tap '%s'", name); return PCD_ERR_ARG; }
    PCD_CHECK_ARG(dst_bytes >= bytes);
    PCD_CHECK_HIP(hipMemcpyAsync(dst, (const char*)workspace + off, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PCD_OK;
}
