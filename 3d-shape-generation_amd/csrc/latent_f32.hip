// fp32 PARITY MODE of a11, SimpleLatentUNetPointNet.forward (reference networks.py:1051-1086): the network of
// csrc/latent.hip -- same folding (packing.py: refine_k into the skip half of dec_k, the time half of enc1 hoisted
// into a per-t bias), same lin[] order -- with fp32 weights, fp32 activations, fp32 products (pcd_gemm_f32,
// csrc/unet_f32.hip) and the fp32 GroupNorm of csrc/train.hip.  The reference computes in fp32; this path is held to
// eps rel-L2 <= 1e-4 per forward and is the bisecting tool for the 1000-step latent loops (the product paths --
// per-layer fp16 launches and the persistent kernel -- are held to 5e-3 over 1000 steps).  One launch per layer
// + one per GroupNorm; speed is not a goal.
#include <new>
#include "common.h"

using namespace pcd;

struct pcd_latent_f32 {
    pcd_latent_desc_t d;
};

namespace {

const int kK[PCD_LATENT_NLIN] = {256, 128, 256, 512, 1024, 2048, 5120, 1536, 768, 384, 128, 128};
const int kC[PCD_LATENT_NLIN] = {128, 256, 512, 1024, 2048, 4096, 1024, 512, 256, 128, 128, 256};

size_t up(size_t v) { return (v + 255) / 256 * 256; }

struct Ws { size_t z1, z2, z3, z4, g0, g1, da, db, pre, stats, total; };

Ws carve(int64_t b) {
    Ws w{};
    size_t o = 0;
    w.z1 = o; o += up(b * 128 * 4);
    w.z2 = o; o += up(b * 256 * 4);
    w.z3 = o; o += up(b * 512 * 4);
    w.z4 = o; o += up(b * 1024 * 4);
    w.g0 = o; o += up(b * 2048 * 4);
    w.g1 = o; o += up(b * 4096 * 4);
    w.da = o; o += up(b * 1024 * 4);
    w.db = o; o += up(b * 1024 * 4);
    w.pre = o; o += up(b * 4096 * 4);          // a layer's pre-norm output
    w.stats = o; o += up(b * 8 * 4) * 2;       // (mean, rstd) per (row, group): written by the GroupNorm kernel, not used further
    w.total = o;
    return w;
}

}  // namespace

extern "C" int pcd_latent_f32_create(const pcd_latent_desc_t* desc, pcd_latent_f32_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    for (int i = 0; i < PCD_LATENT_NLIN; ++i)
        PCD_CHECK_ARG(desc->lin[i].w && desc->lin[i].b && desc->lin[i].k == kK[i] && desc->lin[i].c == kC[i] &&
                      (i >= 10 || (desc->gn_gamma[i] && desc->gn_beta[i])));
    pcd_latent_f32* h = new (std::nothrow) pcd_latent_f32;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_latent_f32_destroy(pcd_latent_f32_t* h) { delete h; }

extern "C" size_t pcd_latent_f32_workspace_bytes(int batch) { return batch > 0 ? carve(batch).total : 0; }

extern "C" int pcd_latent_f32_forward(pcd_latent_f32_t* h, const float* z, int batch, const float* tbias, int tbias_shape_stride,
                                      float* eps, void* workspace, size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(h && z && tbias && eps && workspace && batch > 0);
    PCD_CHECK_ARG(tbias_shape_stride == 0 || tbias_shape_stride == 1);
    const Ws w = carve(batch);
    if (workspace_bytes < w.total) {
        set_error("pcd_latent_f32_forward: workspace %zu < required %zu", workspace_bytes, w.total);
        return PCD_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    float *z1 = F(w.z1), *z2 = F(w.z2), *z3 = F(w.z3), *z4 = F(w.z4), *g0 = F(w.g0), *g1 = F(w.g1), *da = F(w.da), *db = F(w.db);
    float *pre = F(w.pre), *mean = F(w.stats), *rstd = F(w.stats + up((size_t)batch * 8 * 4));
    const pcd_latent_desc_t& d = h->d;
    int rc;
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
    // Linear (bias, or one bias row per sample) [+ GroupNorm(8) + ReLU | + ReLU | identity]
    auto lin = [&](int idx, const float* a1, const float* a2, int k2, const float* bias, const float* row_bias, int mode, float* out) -> int {
        const pcd_linear_desc_t& L = d.lin[idx];
        float* dst = mode == 0 ? pre : out;
        int r = pcd_gemm_f32(a1, L.k - k2, L.k - k2, a2, k2, k2, (const float*)L.w, L.k, row_bias ? nullptr : bias, row_bias, row_bias ? 1 : 0,
                             mode == 1 ? 1 : 0, batch, L.c, dst, L.c, stream);
        if (r || mode != 0) return r;
        return pcd_groupnorm_f32(pre, batch, L.c, 8, d.gn_gamma[idx], d.gn_beta[idx], 1e-5f, 1, out, mean, rstd, stream);
    };
    RUN(lin(0, z, nullptr, 0, tbias_shape_stride ? nullptr : tbias, tbias_shape_stride ? tbias : nullptr, 0, z1));
    RUN(lin(1, z1, nullptr, 0, d.lin[1].b, nullptr, 0, z2));
    RUN(lin(2, z2, nullptr, 0, d.lin[2].b, nullptr, 0, z3));
    RUN(lin(3, z3, nullptr, 0, d.lin[3].b, nullptr, 0, z4));
    RUN(lin(4, z4, nullptr, 0, d.lin[4].b, nullptr, 0, g0));
    RUN(lin(5, g0, nullptr, 0, d.lin[5].b, nullptr, 0, g1));
    RUN(lin(6, g1, z4, 1024, d.lin[6].b, nullptr, 0, da));
    RUN(lin(7, da, z3, 512, d.lin[7].b, nullptr, 0, db));
    RUN(lin(8, db, z2, 256, d.lin[8].b, nullptr, 0, da));
    RUN(lin(9, da, z1, 128, d.lin[9].b, nullptr, 0, db));
    RUN(lin(10, db, nullptr, 0, d.lin[10].b, nullptr, 1, da));
    RUN(lin(11, da, nullptr, 0, d.lin[11].b, nullptr, 2, eps));
#undef RUN
    return PCD_OK;
}
