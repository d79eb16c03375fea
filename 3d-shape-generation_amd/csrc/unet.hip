// a7: UNetPointNetLarge.forward (reference networks.py:779-818) as one enqueue.
//
// Host-side sequencing only: every launch goes through the public layer-level entry
// points of this library.  Algebra applied by the host packer:
//   (i)  time channels of enc1.conv1 are constant over the N points of a shape
//        -> per-shape bias `tbias` (pcd_time_embed), the K=3 xyz half runs in pcd_enc1_xyz;
//   (ii) refine_k (bare conv) followed by the skip half of dec_k.conv1 is one matrix
//        -> folded into lin[13,16,19,22] weights at load;
//   (iii) the 4096 global-feature channels of dec4.conv1 are constant over N
//        -> per-shape bias from a [B x 4096] x [4096 x 1024] product (`wg`);
//   (iv) max over N is fused in the epilogue of global_feat.3 (pcd_gemm_f16_colmax),
//        the (B,4096,N) tensor and its repeat/cat are never materialised.
//
// lin[] execution order (K -> C):
//   0 enc1.conv2 64->64      1 enc1.conv3 64->128 (x1)
//   2 enc2.conv1 128->128    3 enc2.conv2 128->128   4 enc2.conv3 128->256 (x2)
//   5 enc3.conv1 256->256    6 enc3.conv2 256->256   7 enc3.conv3 256->512 (x3)
//   8 enc4.conv1 512->512    9 enc4.conv2 512->512  10 enc4.conv3 512->1024 (x4)
//  11 global_feat.0 1024->2048                      12 global_feat.3 2048->4096 (+max)
//  13 dec4.conv1 skip half x4:1024->1024 (bias -> per-shape, holds the folded bias)
//  14 dec4.conv2 1024->1024 15 dec4.conv3 1024->512
//  16 dec3.conv1 [512|x3 512]->512   17 dec3.conv2   18 dec3.conv3 512->256
//  19 dec2.conv1 [256|x2 256]->256   20 dec2.conv2   21 dec2.conv3 256->128
//  22 dec1.conv1 [128|x1 128]->128   23 dec1.conv2   24 dec1.conv3 128->64
//  25 output.0 64->64 (+BN+ReLU)      head: output.3 64->3 (fp32)
#include <string.h>
#include <new>
#include "common.h"

struct pcd_unet {
    pcd_unet_desc_t d;
    // optional HIP-event timing of the dominant kernel (global_feat.3 + max), see pcd_unet_profile
    static constexpr int kMaxEv = 4096;
    int prof_on = 0;
    int prof_n = 0;
    hipEvent_t ev0[kMaxEv];
    hipEvent_t ev1[kMaxEv];
    int ev_created = 0;
    // packed stage images + biases of the two 256-channel chains (csrc/widechain.hip): enc3 and dec2
    void* wide[2] = {nullptr, nullptr};
    // fragment-order copy of global_feat.3's weights for gemm_xw_kernel (csrc/gemm_f16.hip), made at create; null: the LDS-staged kernel runs
    void* gf3_frag = nullptr;
    // the same for the store GEMMs with K >= 512 and C >= 256 (enc4.conv1-3, global_feat.0, dec4.conv1-3, dec3.conv1-3: gemm_xs_kernel); null: pcd_gemm_f16
    void* lin_frag[PCD_UNET_NLIN] = {};
    // parity-test capture of the decoder blocks' outputs (pcd_unet_capture): dec4 [M][512], dec3 [M][256], dec2 [M][128], dec1 [M][64]
    void* dec_tap[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace pcd {

// csrc/chain.hip: the enc1 chain with an optional buffer cleared in the same launch
int pw_chain_enc1_impl(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias, int tbias_shape_stride,
                       const void* w_conv2, const float* b_conv2, const void* w_conv3, const float* b_conv3, void* x1, bool hilo,
                       float* zero, int64_t zero_n, void* stream);

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct UnetWs {
    size_t x1, x2, x3, x4, s0, s1, pooled, pooled16, gbias, total;
};

static UnetWs carve(int64_t batch, int64_t n) {
    const size_t m = (size_t)batch * (size_t)n;
    UnetWs w{};
    size_t o = 0;
    w.x1 = o; o += align_up(m * 128 * 2);
    w.x2 = o; o += align_up(m * 256 * 2);
    w.x3 = o; o += align_up(m * 512 * 2);
    w.x4 = o; o += align_up(m * 1024 * 2);
    w.s0 = o; o += align_up(m * 2048 * 2);
    w.s1 = o; o += align_up(m * 1024 * 2);
    w.pooled = o; o += align_up((size_t)batch * 4096 * 4);
    w.pooled16 = o; o += align_up((size_t)batch * 4096 * 2);
    w.gbias = o; o += align_up((size_t)batch * 1024 * 4);
    w.total = o;
    return w;
}

}  // namespace pcd

using namespace pcd;

static const int kLinK[PCD_UNET_NLIN] = {64, 64, 128, 128, 128, 256, 256, 256, 512, 512, 512, 1024, 2048,
                                         1024, 1024, 1024, 1024, 512, 512, 512, 256, 256, 256, 128, 128, 64};
static const int kLinC[PCD_UNET_NLIN] = {64, 128, 128, 128, 256, 256, 256, 512, 512, 512, 1024, 2048, 4096,
                                         1024, 1024, 512, 512, 512, 256, 256, 256, 128, 128, 128, 64, 64};

static int g_unet_chains = 3;        // tuning / testing hook (pcd_unet_config): bit 0 = the narrow chains (csrc/chain.hip), bit 1 = the
                                     // 256-channel chains enc3 / dec2 (csrc/widechain.hip); 0 = one GEMM launch per layer

extern "C" int pcd_unet_config(int use_chains) {
    g_unet_chains = use_chains & 3;
    return PCD_OK;
}

extern "C" int pcd_unet_create(const pcd_unet_desc_t* desc, pcd_unet_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    PCD_CHECK_ARG(desc->freqs && desc->tw0 && desc->tb0 && desc->tw2 && desc->tb2);
    PCD_CHECK_ARG(desc->e1w_xyz && desc->e1w_t && desc->e1b && desc->wg && desc->head_w && desc->head_b);
    PCD_CHECK_ARG(desc->wg_k == 4096 && desc->wg_c == 1024);
    if (desc->hilo_mask & ~PCD_UNET_HILO_ALLOWED) {
        set_error("pcd_unet_create: hilo_mask 0x%x names layers outside 0x%x", desc->hilo_mask, PCD_UNET_HILO_ALLOWED);
        return PCD_ERR_ARG;
    }
    for (int i = 0; i < PCD_UNET_NLIN; ++i) {
        if (desc->lin[i].w == nullptr || desc->lin[i].b == nullptr || desc->lin[i].k != kLinK[i] ||
            desc->lin[i].c != kLinC[i]) {
            set_error("pcd_unet_create: layer %d expects %d->%d, got %d->%d (or null pointers)", i, kLinK[i],
                      kLinC[i], desc->lin[i].k, desc->lin[i].c);
            return PCD_ERR_ARG;
        }
    }
    pcd_unet* h = new (std::nothrow) pcd_unet;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    // the two 256-channel chains read their weights as packed stage images: built once, here (the descriptor's weights are final)
    const int first[2] = {5, 19};
    for (int c = 0; c < 2; ++c) {
        const void* w[3] = {desc->lin[first[c]].w, desc->lin[first[c] + 1].w, desc->lin[first[c] + 2].w};
        const float* b[3] = {desc->lin[first[c]].b, desc->lin[first[c] + 1].b, desc->lin[first[c] + 2].b};
        hipError_t e = hipMalloc(&h->wide[c], pcd_pw_wide_packed_bytes(c));
        int rc = e == hipSuccess ? pcd_pw_wide_pack(c, w, b, h->wide[c], nullptr) : PCD_ERR_HIP;
        if (rc == PCD_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = PCD_ERR_HIP;
        if (rc != PCD_OK) {
            if (e != hipSuccess) set_error("pcd_unet_create: %s", hipGetErrorString(e));
            for (int k = 0; k < 2; ++k) if (h->wide[k]) (void)hipFree(h->wide[k]);
            delete h;
            return rc;
        }
    }
    // global_feat.3 (2048 -> 4096): a fragment-order copy of its weights (16.8 MB) for the kernel that reads them straight from global memory;
    // a failed allocation only means the LDS-staged kernel keeps running
    hipPointerAttribute_t at;
    int cur = -1;
    const bool here = hipGetDevice(&cur) == hipSuccess && hipPointerGetAttributes(&at, desc->lin[12].w) == hipSuccess && at.device == cur;
    if (!here) (void)hipGetLastError();
    if (here && hipMalloc(&h->gf3_frag, (size_t)4096 * 2048 * 2) == hipSuccess) {
        if (pcd_gemm_pack_wfrag(desc->lin[12].w, 2048, 2048, 4096, h->gf3_frag, nullptr) != PCD_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(h->gf3_frag);
            h->gf3_frag = nullptr;
        }
    } else {
        (void)hipGetLastError();
        h->gf3_frag = nullptr;
    }
    // the ten store GEMMs that run as 256 x 256 tiles (K >= 512): fragment-order copies (29 MB in all) for gemm_xs_kernel; hi / lo layers keep pcd_gemm_f16_hilo
    for (int i = 8; here && i <= 18; ++i) {
        if (i == 12 || ((desc->hilo_mask >> i) & 1u) || kLinK[i] < 384 || kLinC[i] % 256 != 0) continue;
        if (hipMalloc(&h->lin_frag[i], (size_t)kLinK[i] * kLinC[i] * 2) != hipSuccess ||
            pcd_gemm_pack_wfrag(desc->lin[i].w, kLinK[i], kLinK[i], kLinC[i], h->lin_frag[i], nullptr) != PCD_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipGetLastError();
            if (h->lin_frag[i]) (void)hipFree(h->lin_frag[i]);
            h->lin_frag[i] = nullptr;
        }
    }
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_unet_destroy(pcd_unet_t* h) {
    if (h == nullptr) return;
    if (h->gf3_frag) (void)hipFree(h->gf3_frag);
    for (int i = 0; i < PCD_UNET_NLIN; ++i) if (h->lin_frag[i]) (void)hipFree(h->lin_frag[i]);
    for (int k = 0; k < 2; ++k) if (h->wide[k]) (void)hipFree(h->wide[k]);
    for (int i = 0; i < h->ev_created; ++i) { (void)hipEventDestroy(h->ev0[i]); (void)hipEventDestroy(h->ev1[i]); }
    delete h;
}

extern "C" int pcd_unet_profile(pcd_unet_t* h, int enable) {
    PCD_CHECK_ARG(h != nullptr);
    h->prof_on = enable ? 1 : 0;
    h->prof_n = 0;
    return PCD_OK;
}

extern "C" int pcd_unet_profile_read(pcd_unet_t* h, double* total_ms, int* launches) {
    PCD_CHECK_ARG(h && total_ms && launches);
    double tot = 0.0;
    for (int i = 0; i < h->prof_n; ++i) {
        PCD_CHECK_HIP(hipEventSynchronize(h->ev1[i]));
        float ms = 0.f;
        PCD_CHECK_HIP(hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = h->prof_n;
    return PCD_OK;
}

extern "C" int pcd_unet_capture(pcd_unet_t* h, void* d4, void* d3, void* d2, void* d1) {
    PCD_CHECK_ARG(h != nullptr);
    h->dec_tap[0] = d4; h->dec_tap[1] = d3; h->dec_tap[2] = d2; h->dec_tap[3] = d1;
    return PCD_OK;
}

extern "C" size_t pcd_unet_workspace_bytes(int batch, int n_points) {
    if (batch <= 0 || n_points <= 0) return 0;
    return carve(batch, n_points).total;
}

static int run_lin(const pcd_unet_desc_t& d, int idx, int64_t m, const void* a1, const void* a2, int k2,
                   const float* shape_bias, int rps, void* out, hipStream_t s, const void* wfrag = nullptr) {
    pcd_gemm_desc_t g{};
    const pcd_linear_desc_t& L = d.lin[idx];
    g.a1 = a1; g.k1 = L.k - k2; g.lda1 = g.k1;
    g.a2 = a2; g.k2 = k2; g.lda2 = k2;
    const bool hilo = (d.hilo_mask >> idx) & 1u;          // [c][2 k]: the fp16 weights | the fp16 of their rounding residuals
    g.w = L.w; g.ldw = hilo ? 2 * L.k : L.k;
    g.bias = shape_bias ? nullptr : L.b;
    g.shape_bias = shape_bias; g.rows_per_shape = rps;
    g.relu = 1; g.m = (int)m; g.c = L.c;
    if (hilo) return pcd_gemm_f16_hilo(&g, out, L.c, s);
    return wfrag != nullptr ? pcd_gemm_f16_wfrag(&g, wfrag, out, L.c, s) : pcd_gemm_f16(&g, out, L.c, s);
}

extern "C" int pcd_unet_forward(pcd_unet_t* h, const float* x, int batch, int n_points, const float* tbias,
                                int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes,
                                void* stream) {
    PCD_CHECK_ARG(h && x && tbias && eps && workspace);
    PCD_CHECK_ARG(batch > 0 && n_points > 0);
    const int64_t m = (int64_t)batch * n_points;
    PCD_CHECK_ARG(m <= 0x7fffffff);
    const UnetWs w = carve(batch, n_points);
    if (workspace_bytes < w.total) {
        set_error("pcd_unet_forward: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    void *x1 = ws + w.x1, *x2 = ws + w.x2, *x3 = ws + w.x3, *x4 = ws + w.x4, *s0 = ws + w.s0, *s1 = ws + w.s1;
    float* pooled = (float*)(ws + w.pooled);
    void* pooled16 = ws + w.pooled16;
    float* gbias = (float*)(ws + w.gbias);
    hipStream_t s = (hipStream_t)stream;
    const pcd_unet_desc_t& d = h->d;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
    const bool chains = (g_unet_chains & 1) != 0;
    bool pooled_cleared = false;
    const bool wide = (g_unet_chains & 2) != 0 && m % 256 == 0;     // enc3 / dec2 as register-resident chains (whole 256-point tiles only)
    if (chains) {
        // enc1 (xyz -> 64 -> 64 -> 128) and enc2.conv1-2 (128 -> 128 -> 128): one launch each, intermediates in LDS
        const unsigned e1 = d.hilo_mask & 3u;              // lin 0, 1 travel together (one launch)
        if (e1 == 3u || e1 == 0u) {
            // the same launch clears the pooled maxima that global_feat.3's column-max epilogue accumulates into further down
            RUN(pw_chain_enc1_impl(x, m, n_points, d.e1w_xyz, tbias, tbias_shape_stride, d.lin[0].w, d.lin[0].b, d.lin[1].w, d.lin[1].b, x1,
                                   e1 == 3u, pooled, (int64_t)batch * 4096, s));
            pooled_cleared = true;
        } else {
            RUN(pcd_enc1_xyz(x, m, n_points, d.e1w_xyz, 64, tbias, tbias_shape_stride, s0, s));
            RUN(run_lin(d, 0, m, s0, nullptr, 0, nullptr, 0, s1, s));
            RUN(run_lin(d, 1, m, s1, nullptr, 0, nullptr, 0, x1, s));
        }
        RUN(pcd_pw_chain_128(x1, m, d.lin[2].w, d.lin[2].b, d.lin[3].w, d.lin[3].b, s1, s));
    } else {
        RUN(pcd_enc1_xyz(x, m, n_points, d.e1w_xyz, 64, tbias, tbias_shape_stride, s0, s));
        RUN(run_lin(d, 0, m, s0, nullptr, 0, nullptr, 0, s1, s));
        RUN(run_lin(d, 1, m, s1, nullptr, 0, nullptr, 0, x1, s));
        RUN(run_lin(d, 2, m, x1, nullptr, 0, nullptr, 0, s0, s));
        RUN(run_lin(d, 3, m, s0, nullptr, 0, nullptr, 0, s1, s));
    }
    RUN(run_lin(d, 4, m, s1, nullptr, 0, nullptr, 0, x2, s));
    if (wide) {
        RUN(pcd_pw_wide_chain(0, x2, nullptr, m, h->wide[0], x3, s));
    } else {
        RUN(run_lin(d, 5, m, x2, nullptr, 0, nullptr, 0, s0, s));
        RUN(run_lin(d, 6, m, s0, nullptr, 0, nullptr, 0, s1, s));
        RUN(run_lin(d, 7, m, s1, nullptr, 0, nullptr, 0, x3, s));
    }
    RUN(run_lin(d, 8, m, x3, nullptr, 0, nullptr, 0, s0, s, h->lin_frag[8]));
    RUN(run_lin(d, 9, m, s0, nullptr, 0, nullptr, 0, s1, s, h->lin_frag[9]));
    RUN(run_lin(d, 10, m, s1, nullptr, 0, nullptr, 0, x4, s, h->lin_frag[10]));
    RUN(run_lin(d, 11, m, x4, nullptr, 0, nullptr, 0, s0, s, h->lin_frag[11]));
    {   // global_feat.3 + max over the N points of each shape
        if (!pooled_cleared) RUN(pcd_fill_zero(pooled, (size_t)batch * 4096 * sizeof(float), s));
        pcd_gemm_desc_t g{};
        g.a1 = s0; g.k1 = 2048; g.lda1 = 2048; g.w = d.lin[12].w; g.ldw = 2048; g.bias = d.lin[12].b;
        g.relu = 1; g.m = (int)m; g.c = 4096;
        const bool prof = h->prof_on && h->prof_n < pcd_unet::kMaxEv;
        if (prof) {
            if (h->prof_n >= h->ev_created) {
                PCD_CHECK_HIP(hipEventCreate(&h->ev0[h->ev_created]));
                PCD_CHECK_HIP(hipEventCreate(&h->ev1[h->ev_created]));
                ++h->ev_created;
            }
            PCD_CHECK_HIP(hipEventRecord(h->ev0[h->prof_n], s));
        }
        // whole 256 x 256 tiles, a multiple of 256 of them, shapes of whole 128-row wave tiles: the weights-from-global kernel; else the LDS-staged one
        const int64_t tiles = (m / 256) * 16;
        if (h->gf3_frag != nullptr && pcd_gemm_wfrag_enabled() && m % 256 == 0 && tiles >= 256 && tiles % 256 == 0 && n_points % 128 == 0)
            RUN(pcd_gemm_f16_colmax_wfrag(&g, h->gf3_frag, pooled, n_points, s));
        else
            RUN(pcd_gemm_f16_colmax(&g, pooled, n_points, s));
        if (prof) { PCD_CHECK_HIP(hipEventRecord(h->ev1[h->prof_n], s)); ++h->prof_n; }
    }
    {   // hoisted global half of dec4.conv1: per-shape bias [B][1024] = pooled . Wg^T + folded bias
        RUN(pcd_f32_to_f16(pooled, pooled16, (int64_t)batch * 4096, s));
        if (batch <= 256) {
            // few rows: weight-streaming split-K kernel (csrc/skinny.hip), slabs live in the free s1 buffer
            RUN(pcd_skinny_gemm_f16(pooled16, 4096, nullptr, 0, d.wg, 4096, batch, 1024, (float*)s1, s));
            RUN(pcd_skinny_finish((const float*)s1, pcd_skinny_slabs(4096, 1024), batch, 1024, d.lin[13].b, nullptr, 2, 8,
                                  nullptr, nullptr, nullptr, gbias, s));
        } else {
            pcd_gemm_desc_t g{};
            g.a1 = pooled16; g.k1 = 4096; g.lda1 = 4096; g.w = d.wg; g.ldw = 4096; g.bias = d.lin[13].b;
            g.relu = 0; g.m = batch; g.c = 1024;
            RUN(pcd_gemm_f16_out32(&g, gbias, 1024, s));
        }
    }
    RUN(run_lin(d, 13, m, x4, nullptr, 0, gbias, n_points, s1, s, h->lin_frag[13]));
    RUN(run_lin(d, 14, m, s1, nullptr, 0, nullptr, 0, s0, s, h->lin_frag[14]));
    RUN(run_lin(d, 15, m, s0, nullptr, 0, nullptr, 0, s1, s, h->lin_frag[15]));
#define TAP(i, buf, ch) do { if (h->dec_tap[i]) PCD_CHECK_HIP(hipMemcpyAsync(h->dec_tap[i], buf, (size_t)m * (ch) * 2, \
                                                                             hipMemcpyDeviceToDevice, s)); } while (0)
    TAP(0, s1, 512);
    RUN(run_lin(d, 16, m, s1, x3, 512, nullptr, 0, s0, s, h->lin_frag[16]));
    RUN(run_lin(d, 17, m, s0, nullptr, 0, nullptr, 0, s1, s, h->lin_frag[17]));
    RUN(run_lin(d, 18, m, s1, nullptr, 0, nullptr, 0, s0, s, h->lin_frag[18]));
    TAP(1, s0, 256);
    if (wide) {
        RUN(pcd_pw_wide_chain(1, s0, x2, m, h->wide[1], s1, s));
    } else {
        RUN(run_lin(d, 19, m, s0, x2, 256, nullptr, 0, s1, s));
        RUN(run_lin(d, 20, m, s1, nullptr, 0, nullptr, 0, s0, s));
        RUN(run_lin(d, 21, m, s0, nullptr, 0, nullptr, 0, s1, s));
    }
    TAP(2, s1, 128);
    RUN(run_lin(d, 22, m, s1, x1, 128, nullptr, 0, s0, s));
    const unsigned tl = (d.hilo_mask >> 23) & 7u;         // lin 23, 24, 25 travel together (one launch)
    if (chains && h->dec_tap[3] == nullptr && (tl == 0u || tl == 7u)) {
        // dec1.conv2 -> conv3 -> output.0 -> output.3 (128 -> 128 -> 64 -> 64 -> 3): one launch
        if (tl == 7u)
            RUN(pcd_pw_chain_tail_hilo(s0, m, d.lin[23].w, d.lin[23].b, d.lin[24].w, d.lin[24].b, d.lin[25].w, d.lin[25].b, d.head_w,
                                       d.head_b, eps, s));
        else
            RUN(pcd_pw_chain_tail(s0, m, d.lin[23].w, d.lin[23].b, d.lin[24].w, d.lin[24].b, d.lin[25].w, d.lin[25].b, d.head_w,
                                  d.head_b, eps, s));
    } else {
        RUN(run_lin(d, 23, m, s0, nullptr, 0, nullptr, 0, s1, s));
        RUN(run_lin(d, 24, m, s1, nullptr, 0, nullptr, 0, s0, s));
        TAP(3, s0, 64);       // dec1's output exists only inside the chained tail: a capture runs the tail as per-layer launches (same bits)
        RUN(run_lin(d, 25, m, s0, nullptr, 0, nullptr, 0, s1, s));
        RUN(pcd_head3(s1, m, 64, d.head_w, d.head_b, eps, s));
    }
#undef TAP
#undef RUN
    return PCD_OK;
}

extern "C" int pcd_unet_tap(pcd_unet_t* h, const char* name, int batch, int n_points, const void* workspace,
                            void* dst, size_t dst_bytes, void* stream) {
    PCD_CHECK_ARG(h && name && workspace && dst && batch > 0 && n_points > 0);
    const UnetWs w = carve(batch, n_points);
    const size_t m = (size_t)batch * n_points;
    size_t off = 0, bytes = 0;
    if (!strcmp(name, "x1")) { off = w.x1; bytes = m * 128 * 2; }
    else if (!strcmp(name, "x2")) { off = w.x2; bytes = m * 256 * 2; }
    else if (!strcmp(name, "x3")) { off = w.x3; bytes = m * 512 * 2; }
    else if (!strcmp(name, "x4")) { off = w.x4; bytes = m * 1024 * 2; }
    else if (!strcmp(name, "pooled")) { off = w.pooled; bytes = (size_t)batch * 4096 * 4; }
    else if (!strcmp(name, "gbias")) { off = w.gbias; bytes = (size_t)batch * 1024 * 4; }
    else { set_error("pcd_unet_tap: unknown tap '%s'", name); return PCD_ERR_ARG; }
    PCD_CHECK_ARG(dst_bytes >= bytes);
    PCD_CHECK_HIP(hipMemcpyAsync(dst, (const char*)workspace + off, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PCD_OK;
}
