// a9/a10: SetAttentionBlock.forward (reference networks.py:70-83) and UNetAttentionPointExperimental.forward
// (networks.py:652-704) as single enqueues behind handles, like pcd_unet_forward.
//
// Host-side sequencing only: every launch goes through the layer-level entry points of this library
// (LayerNorm, fp16 MFMA GEMM with residual / dual-source K, set attention, per-shape bias adds).
// Layout: activations fp16 [B*N][C] (point-major, the reference's two transposes around the MHA disappear).
#include <new>
#include "common.h"

struct pcd_attn_unet {
    pcd_attn_unet_desc_t d;
};

namespace pcd {

static inline size_t au_align(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// scratch of one block: two C-wide temporaries, the 3C-wide qkv, the 4C-wide FFN hidden
struct SabWs { size_t t1, t2, qkv, ffh, total; };
static SabWs sab_carve(int64_t rows, int dim) {
    SabWs w{};
    size_t o = 0;
    w.t1 = o; o += au_align((size_t)rows * dim * 2);
    w.t2 = o; o += au_align((size_t)rows * dim * 2);
    w.qkv = o; o += au_align((size_t)rows * 3 * dim * 2);
    w.ffh = o; o += au_align((size_t)rows * 4 * dim * 2);
    w.total = o;
    return w;
}

static int gemm(const void* a1, int k1, const void* a2, int k2, const void* w, const float* b, int relu, int64_t m, int c,
                const void* resid, void* out, hipStream_t s) {
    pcd_gemm_desc_t g{};
    g.a1 = a1; g.k1 = k1; g.lda1 = k1;
    g.a2 = a2; g.k2 = k2; g.lda2 = k2;
    g.w = w; g.ldw = k1 + k2; g.bias = b; g.relu = relu; g.m = (int)m; g.c = c;
    return resid ? pcd_gemm_f16_residual(&g, resid, c, out, c, s) : pcd_gemm_f16(&g, out, c, s);
}

// pre_e / post_e (optional, only where sab_fuses(): the C <= 128 head / tail launches): the block runs on fp16(x + pre_e[shape]) and y leaves as
// fp16(block + post_e[shape]) -- the additive time embeddings around the attention U-Net's blocks without their own launches
static bool sab_fuses(const pcd_sab_desc_t& d, int64_t m) {
    return d.tail_packed != nullptr && pcd_sab_tail_enabled() && pcd_sab_tail_supported(d.dim, m);
}

static int sab_run(const pcd_sab_desc_t& d, const void* x, int batch, int n, int heads, void* y, char* ws, hipStream_t s,
                   const float* pre_e = nullptr, const float* post_e = nullptr, int64_t estride = 0, int rps = 1) {
    const int64_t m = (int64_t)batch * n;
    const int C = d.dim;
    const SabWs w = sab_carve(m, C);
    void *t1 = ws + w.t1, *t2 = ws + w.t2, *qkv = ws + w.qkv, *ffh = ws + w.ffh;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
    // C = 256 with packed images: LayerNorm + Linear as one launch of the wide-chain kernel (widechain.hip), B fragments normalised as they are loaded
    const bool lnlin = pcd_sab_tail_enabled() && pcd_pw_wide_ln_linear_supported(C, m);
    const bool fused128 = sab_fuses(d, m);
    const bool wide_ffn = d.ffn_packed != nullptr && pcd_sab_tail_enabled() && pcd_wide_ffn_supported(C, m);
    if (!fused128 && (pre_e != nullptr || (post_e != nullptr && !wide_ffn))) { set_error("sab_run: embeddings need the fused head / tail launches"); return PCD_ERR_ARG; }
    if (lnlin && d.ln_in_packed != nullptr) {
        RUN(pcd_pw_wide_ln_linear(d.ln_in_packed, 3, 0, x, m, qkv, s));                  // LN1 + in_proj C -> 3C
    } else if (fused128) {
        RUN(pcd_sab_head_bias_f16(C, d.tail_packed, x, m, rps, pre_e, estride, qkv, s)); // C <= 128: the same as one register-resident launch (sab_tail.hip)
    } else {
        RUN(pcd_layernorm_f16(x, m, C, d.ln1_g, d.ln1_b, t1, s));                        // LN1 (q = k = v source)
        RUN(gemm(t1, C, nullptr, 0, d.w_in, d.b_in, 0, m, 3 * C, nullptr, qkv, s));      // in_proj C -> 3C
    }
    RUN(pcd_set_attention_f16(qkv, batch, n, C, heads, t2, nullptr, 0, s));          // softmax(QK^T/sqrt d) V
    if (fused128)
        return pcd_sab_tail_bias_f16(C, d.tail_packed, t2, x, m, rps, pre_e, post_e, estride, y, s);                   // C <= 128: the rest of the block as one launch (sab_tail.hip)
    RUN(gemm(t2, C, nullptr, 0, d.w_out, d.b_out, 0, m, C, x, t1, s));               // x + out_proj(.)
    if (wide_ffn)                                                                        // C = 256: LN2 + FFN + residual (+ the embedding behind the block) as one launch (wideffn.hip)
        return pcd_wide_ffn_bias_f16(d.ffn_packed, t1, m, rps, post_e, estride, y, s);
    if (lnlin && d.ln_ff1_packed != nullptr) {
        RUN(pcd_pw_wide_ln_linear(d.ln_ff1_packed, 4, 1, t1, m, ffh, s));                // LN2 + Linear(C,4C) + ReLU
    } else {
        RUN(pcd_layernorm_f16(t1, m, C, d.ln2_g, d.ln2_b, t2, s));
        RUN(gemm(t2, C, nullptr, 0, d.w_ff1, d.b_ff1, 1, m, 4 * C, nullptr, ffh, s));    // Linear(C,4C) + ReLU
    }
    RUN(gemm(ffh, 4 * C, nullptr, 0, d.w_ff2, d.b_ff2, 0, m, C, t1, y, s));          // + Linear(4C,C)
#undef RUN
    return PCD_OK;
}

static bool sab_ok(const pcd_sab_desc_t& d) {
    return (d.dim == 64 || d.dim == 128 || d.dim == 256) && d.w_in && d.b_in && d.w_out && d.b_out && d.ln1_g && d.ln1_b &&
           d.ln2_g && d.ln2_b && d.w_ff1 && d.b_ff1 && d.w_ff2 && d.b_ff2;
}

// ---- time path: one block per t row.  temb = time_mlp(sinusoid(t)) arrives in `temb` (pcd_time_embed);
// this kernel applies the six emb* Linear layers and folds emb1 through enc1.conv1.
__global__ __launch_bounds__(256) void attn_time_bias_kernel(const float* __restrict__ temb, int dim, pcd_attn_unet_desc_t d,
                                                             float* __restrict__ tbias) {
    extern __shared__ float sm[];        // temb [dim] | emb1 [4]
    float* e1 = sm + dim;
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) sm[i] = temb[(int64_t)row * dim + i];
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* w = d.emb_w[0] + threadIdx.x * dim;
        float acc = d.emb_b[0][threadIdx.x];
        for (int k = 0; k < dim; ++k) acc += w[k] * sm[k];
        e1[threadIdx.x] = acc;
    }
    __syncthreads();
    float* out = tbias + (int64_t)row * PCD_ATTN_UNET_TB;
    // [0, 64): enc1.conv1(x + emb1) = W x + (W emb1 + b)
    for (int c = threadIdx.x; c < 64; c += blockDim.x)
        out[c] = d.e1b[c] + d.e1w[c * 3 + 0] * e1[0] + d.e1w[c * 3 + 1] * e1[1] + d.e1w[c * 3 + 2] * e1[2];
    const int widths[5] = {64, 128, 256, 128, 64};
    int off = 64;
    for (int j = 0; j < 5; ++j) {
        const float* w = d.emb_w[j + 1];
        const float* b = d.emb_b[j + 1];
        for (int c = threadIdx.x; c < widths[j]; c += blockDim.x) {
            const float* wr = w + (int64_t)c * dim;
            float acc = b[c];
            for (int k = 0; k < dim; ++k) acc += wr[k] * sm[k];
            out[off + c] = acc;
        }
        off += widths[j];
    }
}

struct AuWs { size_t x1, x2, x3, p0, p1, p2, sab, total; };
static AuWs au_carve(int64_t batch, int64_t n) {
    const size_t m = (size_t)batch * (size_t)n;
    AuWs w{};
    size_t o = 0;
    w.x1 = o; o += au_align(m * 64 * 2);
    w.x2 = o; o += au_align(m * 128 * 2);
    w.x3 = o; o += au_align(m * 256 * 2);
    w.p0 = o; o += au_align(m * 256 * 2);
    w.p1 = o; o += au_align(m * 256 * 2);
    w.p2 = o; o += au_align(m * 256 * 2);
    w.sab = o; o += sab_carve((int64_t)m, 256).total;
    w.total = o;
    return w;
}

}  // namespace pcd

using namespace pcd;

extern "C" size_t pcd_sab_workspace_bytes(int64_t rows, int dim) {
    if (rows <= 0 || dim <= 0) return 0;
    return sab_carve(rows, dim).total;
}

extern "C" int pcd_sab_forward(const pcd_sab_desc_t* d, const void* x, int batch, int n_points, int heads, void* y,
                               void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(d && x && y && workspace && x != y && batch > 0 && n_points > 0 && heads > 0);
    PCD_CHECK_ARG(sab_ok(*d) && d->dim % heads == 0);
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const size_t need = sab_carve(m, d->dim).total;
    if (workspace_bytes < need) {
        set_error("pcd_sab_forward: workspace %zu < required %zu", workspace_bytes, need);
        return PCD_ERR_WORKSPACE;
    }
    return sab_run(*d, x, batch, n_points, heads, y, (char*)workspace, (hipStream_t)stream);
}

// lin[] order and shapes (K -> C), BatchNorm folded by the host packer
static const int kAuK[PCD_ATTN_UNET_NLIN] = {64, 64, 64, 128, 128, 128, 256, 256, 512, 128, 128, 256, 64, 64};
static const int kAuC[PCD_ATTN_UNET_NLIN] = {64, 64, 128, 128, 128, 256, 256, 256, 128, 128, 128, 64, 64, 64};
static const int kSabC[PCD_ATTN_UNET_NSAB] = {64, 128, 256, 256, 256, 128, 64};

extern "C" int pcd_attn_unet_create(const pcd_attn_unet_desc_t* desc, pcd_attn_unet_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    PCD_CHECK_ARG(desc->dim > 0 && desc->time_dim >= 2 && desc->heads > 0);
    PCD_CHECK_ARG(desc->freqs && desc->tw0 && desc->tb0 && desc->tw2 && desc->tb2 && desc->e1w && desc->e1b);
    PCD_CHECK_ARG(desc->t_w1 && desc->t_b1 && desc->t_w234 && desc->t_b234);
    for (int i = 0; i < PCD_ATTN_UNET_NEMB; ++i) PCD_CHECK_ARG(desc->emb_w[i] && desc->emb_b[i]);
    for (int i = 0; i < PCD_ATTN_UNET_NLIN; ++i) {
        if (!desc->lin[i].w || !desc->lin[i].b || desc->lin[i].k != kAuK[i] || desc->lin[i].c != kAuC[i]) {
            set_error("pcd_attn_unet_create: layer %d expects %d->%d, got %d->%d (or null pointers)", i, kAuK[i], kAuC[i],
                      desc->lin[i].k, desc->lin[i].c);
            return PCD_ERR_ARG;
        }
    }
    for (int i = 0; i < PCD_ATTN_UNET_NSAB; ++i) {
        if (!sab_ok(desc->sab[i]) || desc->sab[i].dim != kSabC[i] || kSabC[i] % desc->heads) {
            set_error("pcd_attn_unet_create: attention block %d expects dim %d divisible by %d heads", i, kSabC[i], desc->heads);
            return PCD_ERR_ARG;
        }
    }
    pcd_attn_unet* h = new (std::nothrow) pcd_attn_unet;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_attn_unet_destroy(pcd_attn_unet_t* h) { delete h; }

extern "C" size_t pcd_attn_unet_workspace_bytes(int batch, int n_points) {
    if (batch <= 0 || n_points <= 0) return 0;
    return au_carve(batch, n_points).total;
}

extern "C" int pcd_attn_unet_time_bias(pcd_attn_unet_t* h, const float* t, int n_t, float* scratch, float* tbias,
                                       void* stream) {
    PCD_CHECK_ARG(h && t && scratch && tbias && n_t > 0);
    const pcd_attn_unet_desc_t& d = h->d;
    int rc = pcd_time_embed(t, n_t, d.freqs, d.time_dim, d.dim, d.tw0, d.tb0, d.tw2, d.tb2, scratch, nullptr, nullptr, 0,
                            nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_time_bias_kernel, dim3(n_t), dim3(256), (size_t)(d.dim + 4) * sizeof(float), (hipStream_t)stream,
                       (const float*)scratch, d.dim, d, tbias);
    PCD_CHECK_LAUNCH();
    return PCD_OK;
}

extern "C" int pcd_attn_unet_forward(pcd_attn_unet_t* h, const float* x, int batch, int n_points, const float* tbias,
                                     int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    PCD_CHECK_ARG(h && x && tbias && eps && workspace);
    PCD_CHECK_ARG(batch > 0 && n_points > 0 && (tbias_shape_stride == 0 || tbias_shape_stride == 1));
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const AuWs w = au_carve(batch, n_points);
    if (workspace_bytes < w.total) {
        set_error("pcd_attn_unet_forward: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    void *x1 = ws + w.x1, *x2 = ws + w.x2, *x3 = ws + w.x3, *p0 = ws + w.p0, *p1 = ws + w.p1, *p2 = ws + w.p2;
    char* sws = ws + w.sab;
    hipStream_t s = (hipStream_t)stream;
    const pcd_attn_unet_desc_t& d = h->d;
    const int H = d.heads, N = n_points;
    const int64_t estr = (int64_t)tbias_shape_stride * PCD_ATTN_UNET_TB;     // floats between the shapes' time-bias rows
    // one shape = N rows; with a shared row every point reads row 0
    const int rps = tbias_shape_stride ? N : (int)m;
    const float *tb_e1 = tbias, *tb_e2 = tbias + 64, *tb_e3 = tbias + 128, *tb_d3 = tbias + 256, *tb_d2 = tbias + 512,
                *tb_d1 = tbias + 640;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
#define LIN(i, a1, a2, k2, out) RUN(gemm(a1, d.lin[i].k - (k2), a2, k2, d.lin[i].w, d.lin[i].b, 1, m, d.lin[i].c, nullptr, out, s))
    // enc1: K=3 conv + time bias (row stride in units of 64 floats), then conv2, conv3
    RUN(pcd_enc1_xyz(x, m, rps, d.e1w, 64, tb_e1, tbias_shape_stride * (PCD_ATTN_UNET_TB / 64), p0, s));
    LIN(0, p0, nullptr, 0, p1);
    LIN(1, p1, nullptr, 0, p0);
    if (sab_fuses(d.sab[0], m)) {
        RUN(sab_run(d.sab[0], p0, batch, N, H, x1, sws, s, nullptr, tb_e2, estr, rps));   // att1, + emb2 on the way out
    } else {
        RUN(sab_run(d.sab[0], p0, batch, N, H, p1, sws, s));                              // att1
        RUN(pcd_add_shape_bias_strided_f16(p1, m, 64, rps, tb_e2, estr, x1, s));          // x1 + emb2
    }
    // enc2: 64 -> 128 -> 128 -> 128; the two 128 -> 128 layers as one launch with the intermediate in LDS (pcd_pw_chain_128, chain.hip: bit-identical
    // to the two GEMM launches), like enc2.conv1-2 of the point U-Net
    const bool chain = pcd_sab_tail_enabled() != 0;
    LIN(2, x1, nullptr, 0, p0);
    if (chain) { RUN(pcd_pw_chain_128(p0, m, d.lin[3].w, d.lin[3].b, d.lin[4].w, d.lin[4].b, p2, s)); }
    else { LIN(3, p0, nullptr, 0, p1); LIN(4, p1, nullptr, 0, p2); }
    if (sab_fuses(d.sab[1], m)) {
        RUN(sab_run(d.sab[1], p2, batch, N, H, x2, sws, s, nullptr, tb_e3, estr, rps));   // att2, + emb3 on the way out
    } else {
        RUN(sab_run(d.sab[1], p2, batch, N, H, p1, sws, s));                              // att2
        RUN(pcd_add_shape_bias_strided_f16(p1, m, 128, rps, tb_e3, estr, x2, s));         // x2 + emb3
    }
    LIN(5, x2, nullptr, 0, p0); LIN(6, p0, nullptr, 0, p1); LIN(7, p1, nullptr, 0, p0);   // enc3
    RUN(sab_run(d.sab[2], p0, batch, N, H, x3, sws, s));                                  // att3 -> x3
    if (d.sab[3].ffn_packed != nullptr && pcd_sab_tail_enabled() && pcd_wide_ffn_supported(256, m)) {
        RUN(sab_run(d.sab[3], x3, batch, N, H, p1, sws, s, nullptr, tb_d3, estr, rps));   // bottleneck, + emb_dec3 on the way out of its feed-forward launch
    } else {
        RUN(sab_run(d.sab[3], x3, batch, N, H, p0, sws, s));                              // bottleneck
        RUN(pcd_add_shape_bias_strided_f16(p0, m, 256, rps, tb_d3, estr, p1, s));
    }
    RUN(sab_run(d.sab[4], p1, batch, N, H, p0, sws, s));                                  // att_dec3
    LIN(8, p0, x3, 256, p1);                                                              // dec3 on cat[xb | x3]: 512 -> 128 -> 128 -> 128
    if (chain) { RUN(pcd_pw_chain_128(p1, m, d.lin[9].w, d.lin[9].b, d.lin[10].w, d.lin[10].b, p2, s)); }
    else { LIN(9, p1, nullptr, 0, p0); LIN(10, p0, nullptr, 0, p2); }
    if (sab_fuses(d.sab[5], m)) {
        RUN(sab_run(d.sab[5], p2, batch, N, H, p1, sws, s, tb_d2, nullptr, estr, rps));   // att_dec2 on (dec3 + emb_dec2), the sum formed as x is read
    } else {
        RUN(pcd_add_shape_bias_strided_f16(p2, m, 128, rps, tb_d2, estr, p0, s));
        RUN(sab_run(d.sab[5], p0, batch, N, H, p1, sws, s));                              // att_dec2
    }
    LIN(11, p1, x2, 128, p0); LIN(12, p0, nullptr, 0, p2); LIN(13, p2, nullptr, 0, p0);   // dec2 on cat[. | x2]
    const void* last = p0;
    if (sab_fuses(d.sab[6], m)) {
        RUN(sab_run(d.sab[6], p0, batch, N, H, p2, sws, s, tb_d1, nullptr, estr, rps));   // att_dec1 on (dec2 + emb_dec1)
        last = p2;
    } else {
        RUN(pcd_add_shape_bias_strided_f16(p0, m, 64, rps, tb_d1, estr, p1, s));
        RUN(sab_run(d.sab[6], p1, batch, N, H, p0, sws, s));                              // att_dec1
    }
    RUN(pcd_tail3(last, 64, x1, 64, m, d.t_w1, d.t_b1, d.t_w234, d.t_b234, eps, s));     // dec1 on cat[. | x1] + output
#undef LIN
#undef RUN
    return PCD_OK;
}

// parity taps: the three skip tensors of the last forward (x1 = att1 + emb2 [m][64], x2 = att2 + emb3 [m][128], x3 = att3 [m][256], fp16)
extern "C" int pcd_attn_unet_tap(pcd_attn_unet_t* h, const char* name, int batch, int n_points, const void* workspace, void* dst,
                                 size_t dst_bytes, void* stream) {
    PCD_CHECK_ARG(h && name && workspace && dst && batch > 0 && n_points > 0);
    const AuWs w = au_carve(batch, n_points);
    const size_t m = (size_t)batch * n_points;
    size_t off = 0, bytes = 0;
    if (name[0] == 'x' && name[1] == '1' && !name[2]) { off = w.x1; bytes = m * 64 * 2; }
    else if (name[0] == 'x' && name[1] == '2' && !name[2]) { off = w.x2; bytes = m * 128 * 2; }
    else if (name[0] == 'x' && name[1] == '3' && !name[2]) { off = w.x3; bytes = m * 256 * 2; }
    else { set_error("pcd_attn_unet_tap: unknown tap '%s'", name); return PCD_ERR_ARG; }
    PCD_CHECK_ARG(dst_bytes >= bytes);
    PCD_CHECK_HIP(hipMemcpyAsync(dst, (const char*)workspace + off, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PCD_OK;
}
