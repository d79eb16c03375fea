// a12: VAE3DLarge.encode / decode (reference networks.py:2299-2310, 2327-2339; nets :2225-2264, ResidualBlock3D
// :471-504) as single enqueues behind a handle.
//
// Host-side sequencing only: every launch goes through the layer-level entry points of this library (direct first /
// last layers, LDS-halo k3 convolution, implicit-GEMM convolution with the ConvTranspose3d parity classes in one
// launch and split-K for the small-grid layers, fp16 GEMMs for the Linear layers).  Activations NDHWC fp16, eval-mode
// BatchNorm3d folded into the weights by the host packer.
#include <new>
#include "common.h"

struct pcd_vae {
    pcd_vae_desc_t d;
    // fragment-order copies of the k3 layers with C_in = 64 (pcd_conv3d_pack_wfrag), made at create: encoder.5.conv1, decoder.8.conv1 / conv2
    void* wf_enc2c1 = nullptr;
    void* wf_enc2c2 = nullptr;          // (with its fused projection shortcut's columns as "tap 27")
    void* wf_dec9 = nullptr;
    void* wf_dec11c1 = nullptr;
    void* wf_dec11c2 = nullptr;
    void* wf_enc5c1 = nullptr;
    void* wf_enc5c2 = nullptr;          // (128 -> 128 with its 64-channel shortcut)
    void* wf_dec5c1 = nullptr;
    void* wf_dec5c2 = nullptr;
    void* wf_dec8c1 = nullptr;
    void* wf_dec8c2 = nullptr;
    void* wf_last = nullptr;            // decoder.12's weights as MFMA A operands (pcd_conv3d_last_pack)
};

namespace pcd {

static inline size_t vae_align(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// three rotating activation buffers (the largest activation is 32^3 x 64 channels), the split-K scratch of the
// convolution launches, and the small GEMM operands
struct VaeWs { size_t a, b, c, scratch, scratch_bytes, small, total; };
static constexpr size_t kConvScratchPerSample = (size_t)8 << 20;     // >= the split-K slabs of any layer (checked per launch)
static VaeWs vae_carve(int batch) {
    VaeWs w{};
    const size_t act = vae_align((size_t)batch * 32768 * 64 * 2);
    size_t o = 0;
    w.a = o; o += act;
    w.b = o; o += act;
    w.c = o; o += act;
    w.scratch = o;
    w.scratch_bytes = vae_align(kConvScratchPerSample * (size_t)(batch < 4 ? 4 : batch));
    o += w.scratch_bytes;
    w.small = o; o += vae_align((size_t)batch * 1024 * 4);
    w.total = o;
    return w;
}

// pcd_vae_config: bit mask of the round-4 kernels in use (A/B, tests, bisecting): 2 = decoder.6 from one LDS halo, 4 = encoder.3 from sub-grid halos,
// 8 = k3 layers with weights in registers; 0 = none (implicit-GEMM / LDS-ring forms), 1 = all
static int g_convt_halo = 14;
enum { kCfgConvT = 2, kCfgK4S2 = 4, kCfgWreg = 8 };

struct Runner {
    const pcd_vae_desc_t& d;
    int batch;
    char* scratch;
    size_t scratch_bytes;
    hipStream_t s;

    void fill(pcd_conv3d_desc_t& c, const pcd_vae_conv_t& L, const void* in, int din, int stride, const int* taps, int ntaps,
              int dout, int relu, const void* resid, void* out, const void* in2 = nullptr, int cin2 = 0) const {
        c = pcd_conv3d_desc_t{};
        c.in = in; c.batch = batch; c.in_d = c.in_h = c.in_w = din; c.cin = L.cin;
        c.rows_d = c.rows_h = c.rows_w = dout; c.stride = stride;
        c.taps = taps; c.ntaps = ntaps; c.kpad = L.kpad;
        c.w = L.w; c.bias = L.b; c.resid = resid; c.relu = relu;
        c.out = out; c.cout = L.cout;
        c.out_d = c.out_h = c.out_w = dout; c.out_scale = 1;
        c.zero_page = d.zero_page;
        c.in2 = in2; c.cin2 = in2 ? cin2 : 0;
    }
    int launch(const pcd_conv3d_desc_t* descs, int n) const {
        size_t need = pcd_conv3d_workspace_bytes(descs, n);
        if (need > scratch_bytes) need = 0;                           // too small: the launch simply runs unsplit
        return pcd_conv3d_f16_multi(descs, n, need ? scratch : nullptr, need, s);
    }
    // Conv3d(k, stride, pad) [+ residual] [+ ReLU]
    int conv(const pcd_vae_conv_t& L, const void* in, int din, int stride, const int* taps, int dout, int relu,
             const void* resid, void* out, const void* in2 = nullptr, int cin2 = 0, const void* wfrag = nullptr) const {
        pcd_conv3d_desc_t c;
        fill(c, L, in, din, stride, taps, L.k * L.k * L.k, dout, relu, resid, out, in2, cin2);
        // k3 layers with C_in = 64: weights in registers (fragment-order copy made at create), 256-row workgroups, no barrier in the tap loop
        if (wfrag != nullptr && (g_convt_halo & kCfgWreg) && pcd_conv3d_k3s1_wreg_supported(&c)) return pcd_conv3d_k3s1_wreg_f16(&c, wfrag, s);
        if (L.k == 3 && stride == 1 && pcd_conv3d_k3s1_supported(&c)) return pcd_conv3d_k3s1_f16(&c, s);   // LDS-resident halo
        // encoder.3 (k4 s2, 64 -> 64, 32^3 -> 16^3): the eight input-parity classes from LDS-resident sub-grid halos
        if ((g_convt_halo & kCfgK4S2) && L.k == 4 && stride == 2 && resid == nullptr && in2 == nullptr && dout * 2 == din &&
            pcd_conv3d_k4s2_halo_supported(batch, din, din, din, L.cin, L.cout, L.kpad))
            return pcd_conv3d_k4s2_halo_f16(in, batch, din, din, din, L.cin, L.w, L.kpad, L.b, relu, L.cout, out, s);
        return launch(&c, 1);
    }
    // ResidualBlock3D: relu(bn2(conv2(relu(bn1(conv1 x)))) + (downsample(x) | x)).  x <- the block's output; h, r: scratch buffers.
    int res(const pcd_vae_res_t& R, void*& x, int dim, void*& h, void*& r, const void* wf1 = nullptr, const void* wf2 = nullptr) const {
        int rc = conv(R.c1, x, dim, 1, d.taps3, dim, 1, nullptr, h, nullptr, 0, wf1);
        if (rc) return rc;
        const void* resid = x;
        void* out = x;                                                // in place: conv2 reads h and the residual row it overwrites
        if (R.has_ds && R.fused_ds) {
            // projection shortcut inside conv2's launch: its weights are K columns behind the 27 taps, x the second source.  NOT in
            // place: rows of x (ds.cin channels) and rows of the output (c2.cout channels) have different strides
            rc = conv(R.c2, h, dim, 1, d.taps3, dim, 1, nullptr, r, x, R.ds.cin, wf2);
            void* t = x; x = r; r = t;
            return rc;
        }
        if (R.has_ds) {
            // 1x1x1 shortcut + BN: a pointwise layer over the NDHWC rows (weights in LDS) where the shape has one
            if (pcd_conv1x1_supported(R.ds.cin, R.ds.cout))
                rc = pcd_conv1x1_f16(x, (int64_t)batch * dim * dim * dim, R.ds.cin, R.ds.w, R.ds.kpad, R.ds.b, 0, R.ds.cout, r, s);
            else
                rc = conv(R.ds, x, dim, 1, d.taps1, dim, 0, nullptr, r);
            if (rc) return rc;
            resid = r;
        }
        return conv(R.c2, h, dim, 1, d.taps3, dim, 1, resid, out, nullptr, 0, wf2);
    }
    // ConvTranspose3d(k4, s2, p1) + ReLU: the 8 output-parity classes (2x2x2 taps each) in one launch
    int convT(const pcd_vae_convT_t& T, const void* in, int din, void* out) const {
        // decoder.6 (128 -> 64, 16^3 -> 32^3): all eight classes from one LDS-resident input halo
        if ((g_convt_halo & kCfgConvT) && pcd_convt3d_k4s2_halo_supported(batch, din, din, din, T.cin, T.cout))
            return pcd_convt3d_k4s2_halo_f16(in, batch, din, din, din, T.cin, T.w, T.b, T.cout, out, s);
        pcd_conv3d_desc_t c[8];
        for (int k = 0; k < 8; ++k) {
            c[k] = pcd_conv3d_desc_t{};
            c[k].in = in; c[k].batch = batch; c[k].in_d = c[k].in_h = c[k].in_w = din; c[k].cin = T.cin;
            c[k].rows_d = c[k].rows_h = c[k].rows_w = din; c[k].stride = 1;
            c[k].taps = T.taps[k]; c[k].ntaps = 8; c[k].kpad = 8 * T.cin;
            c[k].w = T.w[k]; c[k].bias = T.b; c[k].resid = nullptr; c[k].relu = 1;
            c[k].out = out; c[k].cout = T.cout;
            c[k].out_d = c[k].out_h = c[k].out_w = 2 * din; c[k].out_scale = 2;
            c[k].out_off_z = (k >> 2) & 1; c[k].out_off_y = (k >> 1) & 1; c[k].out_off_x = k & 1;
            c[k].zero_page = d.zero_page;
        }
        return launch(c, 8);
    }
};

static bool conv_ok(const pcd_vae_conv_t& L) { return L.w && L.b && L.cin > 0 && L.cout > 0 && L.kpad > 0 && L.k > 0; }
static bool res_ok(const pcd_vae_res_t& R) {
    if (!conv_ok(R.c1) || !conv_ok(R.c2)) return false;
    if (R.has_ds && R.fused_ds)
        return R.ds.cin >= 32 && R.ds.cout == R.c2.cout && R.c2.kpad >= 27 * R.c2.cin + R.ds.cin &&
               (R.ds.cin == 32 || R.c2.kpad == 27 * R.c2.cin + R.ds.cin);
    return !R.has_ds || conv_ok(R.ds);
}

}  // namespace pcd

using namespace pcd;

extern "C" int pcd_vae_config(int convt_halo) {
    PCD_CHECK_ARG(convt_halo >= 0 && convt_halo <= 15);
    g_convt_halo = convt_halo == 1 ? 14 : (convt_halo & 14);
    return PCD_OK;
}

extern "C" int pcd_vae_create(const pcd_vae_desc_t* desc, pcd_vae_t** out) {
    PCD_CHECK_ARG(desc != nullptr && out != nullptr);
    PCD_CHECK_ARG(desc->latent_dim > 0 && desc->latent_dim % 64 == 0);
    PCD_CHECK_ARG(desc->enc0_w && desc->enc0_b && desc->fc_w && desc->fc_b && desc->din_w && desc->din_b && desc->last_w);
    PCD_CHECK_ARG(desc->taps3 && desc->taps4s2 && desc->taps4p0 && desc->taps1 && desc->zero_page);
    for (int i = 0; i < 4; ++i) PCD_CHECK_ARG(res_ok(desc->enc_res[i]) && res_ok(desc->dec_res[i]));
    for (int i = 0; i < 3; ++i) {
        PCD_CHECK_ARG(conv_ok(desc->enc_down[i]) && desc->dec_up[i].b && desc->dec_up[i].cin > 0 && desc->dec_up[i].cout > 0);
        for (int k = 0; k < 8; ++k) PCD_CHECK_ARG(desc->dec_up[i].w[k] && desc->dec_up[i].taps[k]);
    }
    PCD_CHECK_ARG(conv_ok(desc->enc_last) && conv_ok(desc->dec_conv9));
    // the channel plan of networks.py:2225-2264
    PCD_CHECK_ARG(desc->enc_res[0].c1.cin == 32 && desc->enc_res[3].c2.cout == 512 && desc->enc_last.cout == 512);
    PCD_CHECK_ARG(desc->dec_up[0].cin == 512 && desc->dec_conv9.cin == 64 && desc->dec_conv9.cout == 32);
    pcd_vae* h = new (std::nothrow) pcd_vae;
    PCD_CHECK_ARG(h != nullptr);
    h->d = *desc;
    // fragment-order weight copies (one-time device work on the null stream, finished before the handle is returned); a failed allocation only
    // means those layers keep the LDS-ring kernel
    struct { const pcd_vae_conv_t* L; void** dst; } packs[] = {{&h->d.enc_res[1].c1, &h->wf_enc5c1}, {&h->d.dec_res[2].c1, &h->wf_dec8c1},
                                                                {&h->d.dec_res[2].c2, &h->wf_dec8c2}, {&h->d.enc_res[0].c2, &h->wf_enc2c2},
                                                                {&h->d.enc_res[0].c1, &h->wf_enc2c1}, {&h->d.dec_res[3].c1, &h->wf_dec11c1},
                                                                {&h->d.dec_res[3].c2, &h->wf_dec11c2}, {&h->d.dec_conv9, &h->wf_dec9},
                                                                {&h->d.enc_res[1].c2, &h->wf_enc5c2}, {&h->d.dec_res[1].c1, &h->wf_dec5c1},
                                                                {&h->d.dec_res[1].c2, &h->wf_dec5c2}};
    for (auto& pk : packs) {
        const pcd_vae_conv_t& L = *pk.L;
        const size_t bytes = L.k == 3 && L.kpad >= 27 * L.cin ? pcd_conv3d_wfrag_bytes(L.cin, L.cout) : 0;
        if (bytes == 0) continue;
        // a fused projection shortcut's columns sit behind the 27 taps: up to kpad - 27 cin of them (the packer pads K to a multiple of 64 with zeros)
        const int extra = L.kpad - 27 * L.cin, cin2 = L.cin >= 64 ? (extra >= 64 ? 64 : (extra >= 32 ? 32 : 0)) : 0;
        // (the copy is allocated on the CURRENT device: only if that is where the weights live -- one process per GPU sets it so)
        hipPointerAttribute_t at;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || hipPointerGetAttributes(&at, L.w) != hipSuccess || at.device != cur) { (void)hipGetLastError(); continue; }
        void* buf = nullptr;
        if (hipMalloc(&buf, bytes) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (pcd_conv3d_pack_wfrag(L.w, L.kpad, L.cin, L.cout, cin2, buf, nullptr) != PCD_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(buf);
            continue;
        }
        *pk.dst = buf;
    }
    {   // decoder.12's weights as MFMA A operands (27 KB); a failure leaves the VALU form
        hipPointerAttribute_t at;
        int cur = -1;
        void* buf = nullptr;
        if (hipGetDevice(&cur) == hipSuccess && hipPointerGetAttributes(&at, desc->last_w) == hipSuccess && at.device == cur &&
            hipMalloc(&buf, pcd_conv3d_last_packed_bytes()) == hipSuccess) {
            if (pcd_conv3d_last_pack(desc->last_w, buf, nullptr) == PCD_OK && hipStreamSynchronize(nullptr) == hipSuccess) h->wf_last = buf;
            else { (void)hipGetLastError(); (void)hipFree(buf); }
        } else {
            (void)hipGetLastError();
        }
    }
    *out = h;
    return PCD_OK;
}

extern "C" void pcd_vae_destroy(pcd_vae_t* h) {
    if (h == nullptr) return;
    if (h->wf_last) (void)hipFree(h->wf_last);
    for (void* b : {h->wf_enc2c1, h->wf_enc2c2, h->wf_enc5c1, h->wf_enc5c2, h->wf_dec5c1, h->wf_dec5c2, h->wf_dec8c1, h->wf_dec8c2, h->wf_dec9, h->wf_dec11c1, h->wf_dec11c2})
        if (b != nullptr) (void)hipFree(b);
    delete h;
}

extern "C" size_t pcd_vae_workspace_bytes(int batch) {
    if (batch <= 0) return 0;
    return vae_carve(batch).total;
}

#define VAE_PROLOGUE()                                                                        \
    const VaeWs w = vae_carve(batch);                                                         \
    if (workspace_bytes < w.total) {                                                          \
        set_error("%s: workspace %zu < required %zu", __func__, workspace_bytes, w.total);    \
        return PCD_ERR_WORKSPACE;                                                             \
    }                                                                                         \
    char* ws = (char*)workspace;                                                              \
    void *x = ws + w.a, *hb = ws + w.b, *r = ws + w.c, *t_;  /* x: current activation; hb, r: the other two */ \
    const pcd_vae_desc_t& d = h->d;                                                           \
    hipStream_t s = (hipStream_t)stream;                                                      \
    const Runner R{d, batch, ws + w.scratch, w.scratch_bytes, s};                             \
    int rc
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)
#define SWAP(a, b) do { t_ = a; a = b; b = t_; } while (0)

extern "C" int pcd_vae_encode(pcd_vae_t* h, const float* vox, int batch, float* mu_logvar, void* workspace,
                              size_t workspace_bytes, void* stream) {
    PCD_CHECK_ARG(h && vox && mu_logvar && workspace && batch > 0);
    VAE_PROLOGUE();
    // encoder.0/1: Conv3d(1, 32, k3, p1) + ReLU straight from the fp32 occupancy grid
    // a residual block leaves its output in x, whichever buffer that is
    RUN(pcd_conv3d_first(vox, batch, 32, 32, 32, 1, d.enc0_w, d.enc0_b, 32, x, s));
    RUN(R.res(d.enc_res[0], x, 32, hb, r, h->wf_enc2c1, h->wf_enc2c2));           // encoder.2   32 -> 64 @ 32^3
    RUN(R.conv(d.enc_down[0], x, 32, 2, d.taps4s2, 16, 1, nullptr, hb));          // encoder.3/4 k4 s2 -> 16^3
    SWAP(x, hb);
    RUN(R.res(d.enc_res[1], x, 16, hb, r, h->wf_enc5c1, h->wf_enc5c2));           // encoder.5   64 -> 128
    RUN(R.conv(d.enc_down[1], x, 16, 2, d.taps4s2, 8, 1, nullptr, hb));           // encoder.6/7 -> 8^3
    SWAP(x, hb);
    RUN(R.res(d.enc_res[2], x, 8, hb, r));                                        // encoder.8   128 -> 256
    RUN(R.conv(d.enc_down[2], x, 8, 2, d.taps4s2, 4, 1, nullptr, hb));            // encoder.9/10 -> 4^3
    SWAP(x, hb);
    RUN(R.res(d.enc_res[3], x, 4, hb, r));                                        // encoder.11  256 -> 512
    void* const B = x; void* const A = hb;                                       // below: B = encoder.11's output, A = scratch
    (void)r;
    // encoder.12/13: k4 p0 on a 4^3 grid = one row of 32768 inputs per sample -> (B, 512).  With <= 256 samples that is the
    // weight-streaming GEMM of the latent denoiser (33.5 MB of weights against B rows), not a 128-row convolution tile
    const int k_last = d.enc_last.cin * 64;
    const size_t slab_bytes = (size_t)pcd_skinny_slabs(k_last, d.enc_last.cout) * batch * d.enc_last.cout * sizeof(float);
    if (batch <= 256 && d.enc_last.kpad == k_last && slab_bytes <= w.scratch_bytes) {
        float* slabs = (float*)(ws + w.scratch);
        RUN(pcd_skinny_gemm_f16(B, k_last, nullptr, 0, d.enc_last.w, d.enc_last.kpad, batch, d.enc_last.cout, slabs, s));
        RUN(pcd_skinny_finish(slabs, pcd_skinny_slabs(k_last, d.enc_last.cout), batch, d.enc_last.cout, d.enc_last.b, nullptr,
                              1, 0, nullptr, nullptr, A, nullptr, s));
    } else {
        RUN(R.conv(d.enc_last, B, 4, 1, d.taps4p0, 1, 1, nullptr, A));
    }
    // [fc_mu ; fc_logvar]: 512 -> 2 * latent, fp32 out
    const int c_fc = 2 * d.latent_dim, s_fc = pcd_skinny_slabs(512, c_fc);
    if (batch <= 256 && (size_t)s_fc * batch * c_fc * sizeof(float) <= w.scratch_bytes) {
        float* slabs = (float*)(ws + w.scratch);
        RUN(pcd_skinny_gemm_f16(A, 512, nullptr, 0, d.fc_w, 512, batch, c_fc, slabs, s));
        RUN(pcd_skinny_finish(slabs, s_fc, batch, c_fc, d.fc_b, nullptr, 2, 0, nullptr, nullptr, nullptr, mu_logvar, s));
    } else {
        pcd_gemm_desc_t g{};
        g.a1 = A; g.k1 = 512; g.lda1 = 512; g.w = d.fc_w; g.ldw = 512; g.bias = d.fc_b; g.relu = 0; g.m = batch;
        g.c = c_fc;
        RUN(pcd_gemm_f16_out32(&g, mu_logvar, c_fc, s));
    }
    return PCD_OK;
}

extern "C" int pcd_vae_decode(pcd_vae_t* h, const float* z, int batch, float* out, void* workspace, size_t workspace_bytes,
                              void* stream) {
    PCD_CHECK_ARG(h && z && out && workspace && batch > 0);
    VAE_PROLOGUE();
    void* z16 = ws + w.small;
    RUN(pcd_f32_to_f16(z, z16, (int64_t)batch * d.latent_dim, s));
    pcd_gemm_desc_t g{};                                                         // decoder_input, columns already in NDHWC order
    g.a1 = z16; g.k1 = d.latent_dim; g.lda1 = d.latent_dim; g.w = d.din_w; g.ldw = d.latent_dim; g.bias = d.din_b;
    g.relu = 0; g.m = batch; g.c = 512 * 64;
    RUN(pcd_gemm_f16(&g, x, 512 * 64, s));                                       // (B, 4,4,4, 512)
    RUN(R.convT(d.dec_up[0], x, 4, hb));                                          // decoder.0/1  512 -> 256 @ 8^3
    SWAP(x, hb);
    RUN(R.res(d.dec_res[0], x, 8, hb, r));                                        // decoder.2
    RUN(R.convT(d.dec_up[1], x, 8, hb));                                          // decoder.3/4  256 -> 128 @ 16^3
    SWAP(x, hb);
    RUN(R.res(d.dec_res[1], x, 16, hb, r, h->wf_dec5c1, h->wf_dec5c2));           // decoder.5
    RUN(R.convT(d.dec_up[2], x, 16, hb));                                         // decoder.6/7  128 -> 64 @ 32^3
    SWAP(x, hb);
    RUN(R.res(d.dec_res[2], x, 32, hb, r, h->wf_dec8c1, h->wf_dec8c2));           // decoder.8
    RUN(R.conv(d.dec_conv9, x, 32, 1, d.taps3, 32, 1, nullptr, hb, nullptr, 0, h->wf_dec9));   // decoder.9/10  64 -> 32
    SWAP(x, hb);
    RUN(R.res(d.dec_res[3], x, 32, hb, r, h->wf_dec11c1, h->wf_dec11c2));         // decoder.11
    RUN(pcd_conv3d_last_sigmoid_packed(x, batch, 32, 32, 32, 32, d.last_w, h->wf_last, d.last_b, out, s));   // decoder.12/13
    return PCD_OK;
}
#undef SWAP
#undef RUN
#undef VAE_PROLOGUE
