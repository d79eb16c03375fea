/*
 * pcd_hip.h -- C ABI of the MI355X (gfx950) point-cloud diffusion sampler library.
 *
 * The reference (dhillon24/3d-shape-generation) has no FFI/plugin layer: its
 * boundary is the Python object API of diffusion.py / networks.py / metrics.py
 * (SURVEY.md section 8(b)).  This header is the C-ABI shared-library boundary
 * that sits UNDER a re-implementation of that Python API; each entry point
 * cites the reference code it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller owns all memory; nothing is retained except by *_create handles,
 *    which keep the pointers given in their descriptor (they must outlive the handle);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued, never synchronised;
 *  - return value: 0 = ok, negative = error, message via pcd_last_error() (thread local);
 *  - activations are point-major: row m = b*N + n holds the C channels of point n
 *    of shape b, fp16 ("f16") unless stated; weights are [C_out][K] fp16 row-major
 *    (the reference's Conv1d (C_out, C_in, 1) / Linear (out, in) layout, K contiguous).
 */
#ifndef PCD_HIP_H
#define PCD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCD_OK 0
#define PCD_ERR_ARG (-1)
#define PCD_ERR_HIP (-2)
#define PCD_ERR_WORKSPACE (-3)

/* The ABI version this header describes: bumped whenever a signature or a descriptor struct changes (2: pcd_latent_persist_status takes the
 * launch stream; the descriptor structs pcd_unet_desc_t / pcd_sab_desc_t / pcd_conv3d_desc_t / the VAE residual descriptor as they stand since round 4).
 * A binding compares pcd_abi_version() of the library it loaded with the PCD_ABI_VERSION it was written against and refuses a mismatch. */
#define PCD_ABI_VERSION 2

const char* pcd_last_error(void);
/* the PCD_ABI_VERSION the library was built from */
int pcd_abi_version(void);
/* 0 if a gfx950 device is usable by this process, negative otherwise */
int pcd_device_check(void);

/* ------------------------------------------------------------------ GEMM (K1)
 * out[m][c] = act( sum_k A[m][k] * W[c][k] + bias[c] + shape_bias[m / rows_per_shape][c] )
 * Replaces every Conv1d(k=1)+BatchNorm1d(eval)+ReLU of networks.py:46-48 (BN folded
 * into W,bias by the host) and the Linear layers of networks.py:64-66.
 * A may be the K-concatenation of two sources [A1 (K1 cols) | A2 (K2 cols)]
 * (the torch.cat skip inputs of networks.py:811-814) without materialising it.
 * K1, K2 multiples of 64; C multiple of 8.
 */
typedef struct {
    const void* a1; int64_t lda1; int k1;      /* fp16 [M][lda1] */
    const void* a2; int64_t lda2; int k2;      /* optional second source, NULL/0 */
    const void* w;  int64_t ldw;               /* fp16 [C][ldw], ldw >= k1+k2 */
    const float* bias;                         /* fp32 [C] or NULL */
    const float* shape_bias; int rows_per_shape; /* fp32 [ceil(M/rows_per_shape)][C] or NULL */
    int relu;                                  /* 1: max(.,0) */
    int m, c;
} pcd_gemm_desc_t;

/* epilogue: store fp16 out[m][ldo] */
int pcd_gemm_f16(const pcd_gemm_desc_t* d, void* out, int64_t ldo, void* stream);
/* the same with hi / lo weights: w is [c][2 (k1 + k2)] (ldw >= that): columns [0, K) the fp16 weights, [K, 2K) the fp16 of their
 * rounding residuals; the sources are walked twice (all hi products, then all lo products, fp32 accumulation): ~22-bit weights */
int pcd_gemm_f16_hilo(const pcd_gemm_desc_t* d, void* out, int64_t ldo, void* stream);
/* epilogue: store fp32 out[m][ldo] */
int pcd_gemm_f16_out32(const pcd_gemm_desc_t* d, float* out, int64_t ldo, void* stream);
/* split-K form for short-and-wide products with a long reduction (the backward-weight products dW = dz^T a of the
 * training step, reduction over the B*N points): slabs[s][m][c] = A[:, s*K/S:(s+1)*K/S] W[:, same]^T, fp32, all S
 * slices in ONE launch (S * tiles of work instead of a handful).  k2 = 0, no bias/relu; K/S a multiple of 64.
 * pcd_sum_slabs_f32 then adds the S slabs in a fixed order into out[rows][ldo] (deterministic, no float atomics). */
int pcd_gemm_f16_splitk(const pcd_gemm_desc_t* d, int splits, float* slabs, void* stream);
int pcd_sum_slabs_f32(const float* slabs, int nslabs, int64_t rows, int cols, float* out, int64_t ldo, void* stream);
/* epilogue: residual add, out[m][c] = resid[m][c] + (A W^T + bias), fp16 in/out (networks.py:81-82) */
int pcd_gemm_f16_residual(const pcd_gemm_desc_t* d, const void* resid, int64_t ldr,
                          void* out, int64_t ldo, void* stream);
/* epilogue: per-shape column max, colmax[s][c] = max over the rows of shape s (post-ReLU,
 * so values >= 0); replaces torch.max(global_feat, 2) of networks.py:807 without ever
 * writing the (B,4096,N) tensor.  colmax fp32 [n_shapes][C] must be zeroed by the caller
 * (pcd_fill_zero) before the call; requires d->relu == 1. */
int pcd_gemm_f16_colmax(const pcd_gemm_desc_t* d, float* colmax, int rows_per_shape, void* stream);

/* the column-max GEMM with the weights in MFMA-fragment order, read straight from global memory into the MFMA operand registers (csrc/gemm_f16.hip:
 * gemm_xw_kernel; only the activation panel goes through LDS): wfrag = pcd_gemm_pack_wfrag's copy of w ([c][ldw], k = k1 + k2 columns used, c % 256 == 0,
 * same byte count as the k x c weights).  Whole 256 x 256 tiles, at least 256 of them and a multiple of 256; bias required; no per-shape bias. */
int pcd_gemm_pack_wfrag(const void* w, int64_t ldw, int k, int c, void* wfrag, void* stream);
int pcd_gemm_f16_colmax_wfrag(const pcd_gemm_desc_t* d, const void* wfrag, float* colmax, int rows_per_shape, void* stream);
/* 1 unless pcd_gemm_set_config(8) switched the fragment-order path off (9: on): callers that hold a copy ask before they use it */
int pcd_gemm_wfrag_enabled(void);
/* pcd_gemm_f16 (bias [+ per-shape bias] [+ ReLU], fp16 store) for a caller that holds wfrag = pcd_gemm_pack_wfrag's copy of d->w: where the launch shape
 * allows (whole 256 x 256 tiles, a multiple of 256 of them, k1 + k2 >= 384, one A layout) gemm_xs_kernel runs -- weights straight from global memory,
 * and most of each output tile leaves through LDS DURING the next tile's K loop instead of as one store burst behind it; otherwise pcd_gemm_f16's
 * kernels run.  Bitwise the same output either way.  pcd_gemm_set_config(10) / (11): that kernel off / on (default on); pcd_gemm_store_wfrag_enabled reads it. */
int pcd_gemm_f16_wfrag(const pcd_gemm_desc_t* d, const void* wfrag, void* out, int64_t ldo, void* stream);
int pcd_gemm_store_wfrag_enabled(void);
/* diagnostic (dev tools): the following pcd_gemm_f16_colmax_wfrag launches write, per workgroup, (shader-clock cycles, 100-MHz ticks) of its whole tile walk to
 * stamps[256][2] (uint64, device memory); NULL switches it off.  In-kernel clock = cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS item 6). */
int pcd_gemm_wfrag_stamps(void* stamps);
/* tuning/benchmark hook: force a tile configuration for every following GEMM launch of this
 * process (-1 = shape heuristic; 0: 128x64, 1: 128x128, 2: 256x128 3-stage, 3: 256x256, 4: 128x128 3-stage).
 * 5 / 7 / 6 leave the tile choice alone and switch the 256x256 store / column-max kernel that requests the next tile's first
 * K tile(s) before the epilogue's stores (whole tiles only): off / one K tile ahead (default) / two (store epilogue only).
 * 8 / 9: pcd_gemm_wfrag_enabled() off / on (default on).  10 / 11: pcd_gemm_store_wfrag_enabled() off / on (default on).
 * 12 / 13: the fragment-order kernels request a K tile's LDS-DMA pieces at its top (default) / behind its first 32 MFMAs (measured: 14 % slower; kept as the
 * experiment that shows the K loop waits on those pieces).  16 + bits: timing ablations of gemm_xs_kernel.
 * TEST / BENCHMARK ONLY: process-global, read by every handle at every launch -- not for use while another thread is launching. */
int pcd_gemm_set_config(int cfg);

int pcd_fill_zero(void* p, size_t bytes, void* stream);
int pcd_f32_to_f16(const float* src, void* dst, int64_t n, void* stream);
int pcd_f16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* ------------------------------------------------- time embedding + enc1 (K3)
 * For each of n_t time values: sinusoidal embedding (networks.py:820-838, freqs
 * passed in, computed on the host with the reference's torch ops) -> time_mlp
 * Linear/SiLU/Linear (networks.py:737-741) -> temb[n_t][dim].
 * If e1w_t != NULL also the hoisted time half of enc1.conv1:
 * tbias[i][c] = sum_j e1w_t[c][j]*temb[i][j] + e1b[c]   (BN already folded), c < c1.
 * All fp32.
 */
int pcd_time_embed(const float* t, int n_t, const float* freqs, int time_dim, int dim,
                   const float* w0, const float* b0, const float* w2, const float* b2,
                   float* temb,
                   const float* e1w_t, const float* e1b, int c1, float* tbias, void* stream);

/* small fp32 linear: y[r][c] = sum_k x[r][k] w[c][k] + b[c]  (emb* layers networks.py:613-618) */
int pcd_linear_f32(const float* x, int rows, int k, const float* w, const float* b, int c,
                   float* y, void* stream);

/* enc1.conv1 xyz half (K=3) + per-shape time bias + ReLU -> fp16 [M][c1]
 * x fp32 [M][3]; w_xyz fp32 [c1][3]; tbias fp32 [n_t][c1] with row index
 * (m / rows_per_shape) * tbias_shape_stride (stride 0 = same t for all shapes; rows are c1 floats, so a
 * stride > 1 walks a wider table whose rows are a multiple of c1 apart). */
int pcd_enc1_xyz(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, int c1,
                 const float* tbias, int tbias_shape_stride, void* out, void* stream);

/* ---------------------------------------------- elementwise diffusion ops (K4)
 * All fp32, bit-exact with the reference's torch-CPU expression order.
 * rates are per shape: r[b * stride], stride 0 or 1. `per_shape` = elements per shape.
 */
/* diffusion.py:151  x_t = s*x0 + n*noise */
int pcd_add_noise(const float* x0, const float* noise, const float* n, const float* s, int stride,
                  int64_t total, int64_t per_shape, float* x_t, void* stream);
/* diffusion.py:167  x0 = (x_t - n*eps)/s */
int pcd_remove_noise(const float* x_t, const float* eps, const float* n, const float* s, int stride,
                     int64_t total, int64_t per_shape, float* x0, void* stream);
/* diffusion.py:283-287 (DDIM):  x0 = (x-n*eps)/s ; x_next = s2*x0 + n2*eps.  x_next may be NULL. */
int pcd_ddim_update(const float* x, const float* eps, const float* n, const float* s,
                    const float* n2, const float* s2, int stride,
                    int64_t total, int64_t per_shape, float* x0, float* x_next, void* stream);
/* diffusion.py:246-255 (DDPM):  x0 as above ; x_next = s2*x0 + coef*n*z */
int pcd_ddpm_update(const float* x, const float* eps, const float* z, const float* n, const float* s,
                    const float* coef, const float* s2, int stride,
                    int64_t total, int64_t per_shape, float* x0, float* x_next, void* stream);
/* pcd_randn_step + pcd_ddpm_update in ONE launch: the step's normal draw z is generated in place (same Philox counters:
 * base_offset + per_step_stride * counter[1] + element / 4, same Box-Muller arithmetic) and never stored; bitwise the two launches. */
int pcd_ddpm_update_philox(const float* x, const float* eps, const float* n, const float* s, const float* coef, const float* s2,
                           int stride, int64_t total, int64_t per_shape, float* x0, float* x_next, uint64_t seed,
                           uint64_t base_offset, uint64_t per_step_stride, const int* counter, void* stream);
/* standard normal fill (Philox4x32-10 + Box-Muller), for perf runs (diffusion.py:239,254,275) */
int pcd_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* HIP-graph support: select the per-step constants ON THE DEVICE so one captured step can be
 * replayed for every timestep.  k = counter[0] (clamped to n_steps-1); tb_cur[0..tb_elems) =
 * tb_table[k][:]; rates_cur[t*width + j] = rate_tables[t][k][j] for the 4 tables t (n, s, a, b of
 * the sampler); counter[1] = k; counter[0] = k+1.  counter: device int32[2]. */
int pcd_step_select(int* counter, int n_steps, const float* tb_table, int tb_elems, float* tb_cur,
                    const float* rate_tables, int width, float* rates_cur, void* stream);
/* pcd_randn whose Philox offset is base_offset + per_step_stride * counter[1] (read on the device) */
int pcd_randn_step(float* out, int64_t n, uint64_t seed, uint64_t base_offset, uint64_t per_step_stride,
                   const int* counter, void* stream);

/* output head: eps[m][j] = sum_k h[m][k]*w[j][k] + b[j], j<3  (networks.py:770 `output.3`),
 * h fp16 [M][k], w fp32 [3][k]; eps fp32 [M][3]. */
int pcd_head3(const void* h, int64_t m, int k, const float* w, const float* b, float* eps, void* stream);

/* ------------------------------------------------ whole point denoiser (a7)
 * UNetPointNetLarge.forward (networks.py:779-818) as one enqueue of ~30 kernels.
 * The descriptor holds device pointers to host-folded weights (BN folded, refine_k folded
 * into dec_k.conv1, time/global halves hoisted: SURVEY.md A.3).
 */
#define PCD_UNET_NLIN 26
typedef struct {
    const void* w; const float* b; int k; int c;
} pcd_linear_desc_t;

typedef struct {
    int time_dim, dim;
    const float* freqs;                       /* [time_dim/2] */
    const float *tw0, *tb0, *tw2, *tb2;       /* time_mlp fp32 */
    const float* e1w_xyz;                     /* [64][3]  */
    const float* e1w_t;                       /* [64][dim] */
    const float* e1b;                         /* [64] */
    pcd_linear_desc_t lin[PCD_UNET_NLIN];     /* execution order, see csrc/unet.hip */
    const void* wg; int wg_k, wg_c;           /* dec4.conv1 global-feature half fp16 [1024][4096] */
    const float* head_w; const float* head_b; /* output.3 fp32 [3][64], [3] */
    unsigned hilo_mask;                       /* bit i: lin[i].w is [c][2 k] = the fp16 weights | the fp16 of their rounding residuals
                                                 (hi / lo weights, ~22 bits): allowed for the narrow layers PCD_UNET_HILO_ALLOWED */
} pcd_unet_desc_t;
/* layers that may carry hi / lo weights: enc1.conv2, enc1.conv3 (0, 1), enc2.conv3 (4), dec1.conv1 .. output.0 (22 .. 25) */
#define PCD_UNET_HILO_ALLOWED 0x3C00013u

/* Chains of the narrow pointwise layers of UNetPointNetLarge in one launch each (csrc/chain.hip): a wave carries 32
 * points through the chain with the intermediates in LDS, the chain's weights (fp16 [C][K], BatchNorm folded) resident
 * in LDS.  Every layer = Conv1d(k=1) + BN + ReLU with fp32 accumulation and one fp16 rounding, like pcd_gemm_f16.
 *   enc1:  x fp32 [m][3] -> conv1 (K = 3, + per-shape time bias, as pcd_enc1_xyz) -> conv2 64->64 -> conv3 64->128 -> x1
 *   128:   in fp16 [m][128] -> 128->128 -> 128->128 -> out fp16 [m][128]                 (enc2.conv1, enc2.conv2)
 *   tail:  in fp16 [m][128] -> 128->128 -> 128->64 -> 64->64 -> 64->3 fp32 (no ReLU)     (dec1.conv2, conv3, output.0, output.3) */
int pcd_pw_chain_enc1(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias,
                      int tbias_shape_stride, const void* w_conv2, const float* b_conv2, const void* w_conv3,
                      const float* b_conv3, void* x1, void* stream);
int pcd_pw_chain_128(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b, const float* b_b,
                     void* out, void* stream);
int pcd_pw_chain_tail(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b, const float* b_b,
                      const void* w_c, const float* b_c, const float* head_w, const float* head_b, float* eps, void* stream);
/* enc1 and tail chains with hi / lo weights (every w_* is [C][2 K] = hi | lo; see pcd_gemm_f16_hilo): the same arithmetic with the K
 * loop run against hi, then against lo.  These narrow layers are the direct route from the coordinates to the predicted noise; the
 * fp16 rounding of their weights is what the fp16 path's 1000-step DDPM trajectory deviates by. */
int pcd_pw_chain_enc1_hilo(const float* x, int64_t m, int rows_per_shape, const float* w_xyz, const float* tbias,
                           int tbias_shape_stride, const void* w_conv2, const float* b_conv2, const void* w_conv3,
                           const float* b_conv3, void* x1, void* stream);
int pcd_pw_chain_tail_hilo(const void* in, int64_t m, const void* w_a, const float* b_a, const void* w_b, const float* b_b,
                           const void* w_c, const float* b_c, const float* head_w, const float* head_b, float* eps, void* stream);
/* One pointwise layer out[m][c] = act(in[m][k] . W^T + bias) with the weights resident in LDS: the 1x1x1 shortcut
 * convolutions of ResidualBlock3D (reference networks.py:485-490) on NDHWC rows.  in fp16 [m][k]; w fp16 [c][ldw] (k used);
 * bias fp32 [c]; relu 0/1; out fp16 [m][c].  Shapes: (k, c) = (32, 64), (64, 128), (128, 256) -- pcd_conv1x1_supported(). */
int pcd_conv1x1_supported(int k, int c);
int pcd_conv1x1_f16(const void* in, int64_t m, int k, const void* w, int64_t ldw, const float* bias, int relu, int c,
                    void* out, void* stream);
/* testing / tuning hook for pcd_unet_forward: 0 = one GEMM launch per layer, 1 (default) = the chained kernels above */
int pcd_unet_config(int use_chains);
/* The 256-channel chains of the same network as one launch each (csrc/widechain.hip): chain 0 = enc3.conv1-3 (x2 [M][256] -> x3 [M][512]),
 * chain 1 = dec2.conv1-3 ([in1 [M][256] | in2 [M][256]] -> [M][128]).  A wave carries 32 points through the chain with the activations
 * in registers (the layer-to-layer hand-over is a v_permlane32_swap per register); only the weights stream, as 32-KB stage images that
 * pcd_pw_wide_pack builds once from the three layers' [C][K] fp16 weights and fp32 biases (BatchNorm folded, as for pcd_gemm_f16) into a
 * caller-owned buffer of pcd_pw_wide_packed_bytes(chain) bytes.  M must be a multiple of 256.  pcd_unet_config bit 1 switches
 * pcd_unet_forward between these chains and one GEMM launch per layer (bit 0: the narrow chains above). */
size_t pcd_pw_wide_packed_bytes(int chain);
int pcd_pw_wide_pack(int chain, const void* const* w, const float* const* b, void* packed, void* stream);
int pcd_pw_wide_chain(int chain, const void* in1, const void* in2, int64_t m, const void* packed, void* out, void* stream);
/* A/B hook (TEST / BENCHMARK ONLY, process-global): which waves request the weight images of the wide-chain launches (1, default: one wave per SIMD,
 * alternating groups per image; 0: every wave; 2: the split form in the LN + Linear launches too).  Same bits either way. */
int pcd_pw_wide_config(int split);
/* LayerNorm(256) + Linear(256, 256 passes) [+ ReLU] as one launch of the wide-chain kernel (csrc/widechain.hip): the B fragments are normalised as they
 * are loaded (two-pass fp32 statistics, eps 1e-5, fp16 result as pcd_layernorm_f16's), so the LayerNorm launch and its tensor disappear: attention in_proj
 * (passes 3, relu 0) and ff.0 (passes 4, relu 1) of the C = 256 SetAttentionBlocks (networks.py:61-66, 81-82).  w fp16 [256 passes][256], b fp32;
 * packed: pcd_pw_wide_ln_linear_packed_bytes(passes) bytes of device memory; rows % 256 == 0; out fp16 [rows][256 passes], out != x. */
size_t pcd_pw_wide_ln_linear_packed_bytes(int passes);
int pcd_pw_wide_ln_linear_pack(const void* w, const float* b, int passes, const float* ln_g, const float* ln_b, void* packed, void* stream);
int pcd_pw_wide_ln_linear_supported(int dim, int64_t rows);
int pcd_pw_wide_ln_linear(const void* packed, int passes, int relu, const void* x, int64_t rows, void* out, void* stream);

typedef struct pcd_unet pcd_unet_t;
int pcd_unet_create(const pcd_unet_desc_t* desc, pcd_unet_t** out);
void pcd_unet_destroy(pcd_unet_t* h);
size_t pcd_unet_workspace_bytes(int batch, int n_points);
/* eps = model(x, t):  x fp32 [B][N][3]; tbias fp32 [n_t][64] from pcd_time_embed with
 * tbias_shape_stride 0 (one t for the batch) or 1 (one per shape); eps fp32 [B][N][3]. */
int pcd_unet_forward(pcd_unet_t* h, const float* x, int batch, int n_points,
                     const float* tbias, int tbias_shape_stride,
                     float* eps, void* workspace, size_t workspace_bytes, void* stream);
/* HIP-event timing of the dominant kernel (global_feat.3 GEMM + max) on the launch stream:
 * enable, run forwards, then read the summed duration and launch count (bench.py roofline). */
int pcd_unet_profile(pcd_unet_t* h, int enable);
int pcd_unet_profile_read(pcd_unet_t* h, double* total_ms, int* launches);
/* optional taps for parity tests: copies of x1..x4 (fp16), pooled / gbias (fp32) out of the last forward's workspace */
int pcd_unet_tap(pcd_unet_t* h, const char* name, int batch, int n_points,
                 const void* workspace, void* dst, size_t dst_bytes, void* stream);
/* The decoder blocks' outputs (reference networks.py:811-814: dec4 [M][512], dec3 [M][256], dec2 [M][128], dec1 [M][64], fp16)
 * live in ping-pong buffers / inside the chained tail; while capture buffers are set, every forward copies them out
 * (dec1: the tail runs as per-layer launches, bit-identical).  Null pointers switch the capture off. */
int pcd_unet_capture(pcd_unet_t* h, void* d4, void* d3, void* d2, void* d1);

/* ------------------------------------------------ fp32 parity mode of the point denoiser (csrc/unet_f32.hip)
 * SURVEY 8(c) "HIP fp32 parity mode": the network of pcd_unet_forward with fp32 weights, fp32 activations and fp32
 * products (v_mfma_f32_32x32x2_f32); same descriptor layout and lin[] order, but EVERY weight pointer (lin[i].w, wg) is
 * fp32 [C][K].  Held to eps rel-L2 <= 1e-4 per forward against the reference's fp32 arithmetic (networks.py:779-818).
 * pcd_gemm_f32: out[m][c] = act([A1 | A2][m][k1 + k2] . W[c][k]^T + bias[c] (or shape_bias[row / rows_per_shape][c]));
 * k1, k2 multiples of 16.  pcd_gemm_f32_colmax: pooled[row / rows_per_shape][c] = max(pooled, relu(A . W^T + bias)), the
 * `torch.max(x, 2)` of networks.py:807 fused (pooled zero-initialised by the caller).
 * Taps of the last forward (all fp32): x1..x4, pooled, gbias, d4..d1. */
int pcd_gemm_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2, const float* w, int64_t ldw,
                 const float* bias, const float* shape_bias, int rows_per_shape, int relu, int m, int c, float* out,
                 int64_t ldo, void* stream);
int pcd_gemm_f32_colmax(const float* a, int64_t lda, int k, const float* w, int64_t ldw, const float* bias, int m, int c,
                        float* pooled, int rows_per_shape, void* stream);
typedef struct pcd_unet_f32 pcd_unet_f32_t;
int pcd_unet_f32_create(const pcd_unet_desc_t* desc, pcd_unet_f32_t** out);
void pcd_unet_f32_destroy(pcd_unet_f32_t* h);
size_t pcd_unet_f32_workspace_bytes(int batch, int n_points);
int pcd_unet_f32_forward(pcd_unet_f32_t* h, const float* x, int batch, int n_points, const float* tbias,
                         int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes, void* stream);
/* diagnostic: bit i of `mask` makes lin[i]'s output (bit 26: enc1.conv1's) round to fp16 and back before it is stored, everything else
 * staying fp32 -- locates where ACTIVATION rounding of the fp16 product path matters (tools/attribute_fp16_layers.py); 0 = off */
int pcd_unet_f32_round_activations(pcd_unet_f32_t* h, unsigned mask);
int pcd_unet_f32_tap(pcd_unet_f32_t* h, const char* name, int batch, int n_points, const void* workspace, void* dst,
                     size_t dst_bytes, void* stream);

/* ------------------------------------------------ latent denoiser (a11, K8)
 * GroupNorm(groups, C, eps 1e-5, biased var) + affine + ReLU on rows of x fp32 [rows][c] -> fp16
 * (the Linear+GroupNorm(8)+ReLU stages of networks.py:984-1036; 2-D input = per-sample groups). */
int pcd_groupnorm_relu_f16(const float* x, int rows, int c, int groups, const float* gamma,
                           const float* beta, void* out, void* stream);
/* Weight-streaming GEMM for M <= 256 rows (the latent vectors): split-K partial sums
 * slabs[s][m][c] = sum over the s-th K slice of [A1|A2][m][k] * W[c][k]; pcd_skinny_slabs gives S.
 * pcd_skinny_finish adds the slabs in a fixed order (+bias, + optional per-row bias) and applies
 * mode 0: GroupNorm(groups)+affine+ReLU -> fp16, 1: ReLU -> fp16, 2: identity -> fp32. */
int pcd_skinny_slabs(int k, int c);
int pcd_skinny_gemm_f16(const void* a1, int k1, const void* a2, int k2, const void* w, int64_t ldw,
                        int m, int c, float* slabs, void* stream);
int pcd_skinny_finish(const float* slabs, int nslabs, int m, int c, const float* bias, const float* row_bias,
                      int mode, int groups, const float* gamma, const float* beta,
                      void* out16, float* out32, void* stream);
/* testing / tuning hook: use_lds_dma = 0 makes the skinny kernels load their MFMA fragments straight from global memory
 * (the first form) instead of staging full 128-byte lines through LDS by LDS-DMA; 1 restores the default.  Same bits. */
int pcd_skinny_config(int use_lds_dma);
/* The same layer in ONE launch for small weights (k * c <= 768 * 256): a workgroup owns whole GroupNorm groups
 * over the full K, so Linear + bias + GroupNorm + ReLU need no second kernel (enc1-3, dec1-2 and the output head of
 * SimpleLatentUNetPointNet, networks.py:984-1049).  pcd_skinny_fused_supported() tells (1/0) whether a shape
 * qualifies (mode 0: group size 16/32/64/128). */
int pcd_skinny_fused_supported(int k, int c, int mode, int groups);
int pcd_skinny_fused(const void* a1, int k1, const void* a2, int k2, const void* w, int64_t ldw, int m, int c,
                     const float* bias, const float* row_bias, int mode, int groups, const float* gamma,
                     const float* beta, void* out16, float* out32, void* stream);
/* ... with the activations given in fp32 [m][k] and rounded to fp16 on load (enc1 reads the latent state z directly;
 * saves the separate conversion launch). */
int pcd_skinny_fused_f32in(const float* a, int k, const void* w, int64_t ldw, int m, int c, const float* bias,
                           const float* row_bias, int mode, int groups, const float* gamma, const float* beta,
                           void* out16, float* out32, void* stream);
/* SimpleLatentUNetPointNet.forward (networks.py:1051-1086), latent_dim=256, dim=512, time_dim=256.
 * lin[] order documented in csrc/latent.hip; refine_k folded into dec_k, enc1's time half hoisted
 * into tbias [n_t][128] (pcd_time_embed with c1=128). */
#define PCD_LATENT_NLIN 12
typedef struct {
    pcd_linear_desc_t lin[PCD_LATENT_NLIN];
    const float* gn_gamma[PCD_LATENT_NLIN];   /* entries 0..9 (GroupNorm layers), rest NULL */
    const float* gn_beta[PCD_LATENT_NLIN];
} pcd_latent_desc_t;
typedef struct pcd_latent pcd_latent_t;
int pcd_latent_create(const pcd_latent_desc_t* desc, pcd_latent_t** out);
void pcd_latent_destroy(pcd_latent_t* h);
size_t pcd_latent_workspace_bytes(int batch);
/* eps = model(z, t): z fp32 [B][256] -> eps fp32 [B][256]; tbias stride 0 (one t) or 1 (per sample) */
int pcd_latent_forward(pcd_latent_t* h, const float* z, int batch, const float* tbias,
                       int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes,
                       void* stream);

/* fp32 parity mode of the latent denoiser (csrc/latent_f32.hip): pcd_latent_forward's network with fp32 weights (EVERY
 * lin[i].w of the descriptor is fp32 [C][K] here), fp32 activations and products, one launch per layer and GroupNorm;
 * held to eps rel-L2 <= 1e-4 against the reference's fp32 arithmetic (networks.py:1051-1086). */
typedef struct pcd_latent_f32 pcd_latent_f32_t;
int pcd_latent_f32_create(const pcd_latent_desc_t* desc, pcd_latent_f32_t** out);
void pcd_latent_f32_destroy(pcd_latent_f32_t* h);
size_t pcd_latent_f32_workspace_bytes(int batch);
int pcd_latent_f32_forward(pcd_latent_f32_t* h, const float* z, int batch, const float* tbias, int tbias_shape_stride,
                           float* eps, void* workspace, size_t workspace_bytes, void* stream);

/* The same network, and whole DDIM timesteps of the latent sampler (the loop body of LatentDiffusion.sample / sample3,
 * diffusion.py:637-645, 691-700), as ONE persistent launch for batch <= 64 (csrc/latent_persist.hip; 33 .. 64 rows run as two interleaved streams of <= 32): 256 workgroups, one
 * per CU, keep all 38 MB of fp16 weights in LDS for the whole call and exchange activations through self-validating
 * buffers (no flags or atomics between layers).  pcd_latent_persist_supported(batch) tells (1/0) whether the current
 * device can run it (exactly 256 CUs with 160 KB LDS, batch <= 64); callers fall back to pcd_latent_forward otherwise.
 * The handle keeps the descriptor's pointers.  Workspace: pcd_latent_persist_workspace_bytes(h), 256-byte aligned, owned
 * by the caller, private to one in-flight call.
 *  - pcd_latent_persist_forward: eps = model(z, t) with tbias = the hoisted time row [128] (one t for the batch).
 *  - pcd_latent_persist_ddim: `nsteps` consecutive DDIM steps starting at table row counter[0] (pcd_step_select
 *    semantics: rows clamp at n_steps_table - 1; on return counter[0] += nsteps, counter[1] = last row used): z is
 *    updated in place, x0 (may be NULL) receives the last step's x0.  rate_tables is (4, n_steps_table, rate_width)
 *    fp32 = (n, s, n_next, s_next) like pcd_ddim_update's operands, rate_width 1 or batch.  The update is bitwise
 *    pcd_ddim_update's.
 *  - every in-kernel wait is bounded (0.2 s); pcd_latent_persist_status waits for the launch stream and copies the status word to the
 *    host: 0 = ok, otherwise (wait kind << 16 | workgroup) of the first wait that gave up (outputs
 *    are then undefined).
 *  - pcd_latent_persist_config: tuning hooks: poll back-off (s_sleep count between polls); predict_waits = 1 (default): a wait
 *    sleeps through 7/8 of the time the same wait took in the previous step before it starts probing. */
typedef struct pcd_latent_persist pcd_latent_persist_t;
int pcd_latent_persist_supported(int batch);
int pcd_latent_persist_create(const pcd_latent_desc_t* desc, pcd_latent_persist_t** out);
void pcd_latent_persist_destroy(pcd_latent_persist_t* h);
size_t pcd_latent_persist_workspace_bytes(const pcd_latent_persist_t* h);
int pcd_latent_persist_config(pcd_latent_persist_t* h, int poll_sleep, int predict_waits);
/* diagnostic: instrumented kernel for the next launches; buf [256][steps][8 units][8] u32 of 100 MHz stamps, NULL = off */
int pcd_latent_persist_trace(pcd_latent_persist_t* h, unsigned* buf, int steps);
int pcd_latent_persist_forward(pcd_latent_persist_t* h, const float* z, int batch, const float* tbias, float* eps,
                               void* workspace, size_t workspace_bytes, void* stream);
int pcd_latent_persist_ddim(pcd_latent_persist_t* h, float* z, float* x0, int batch, const float* tb_table, int tb_elems,
                            const float* rate_tables, int rate_width, int n_steps_table, int* counter, int nsteps,
                            void* workspace, size_t workspace_bytes, void* stream);
/* waits for `stream` (the stream the launch was enqueued on; nothing else is synchronised), then reads the status word of the last launch:
 * 0 = every wait was met; else (wait kind << 16) | workgroup.
 * A non-zero status means the launch was ABANDONED (its outputs are undefined): the caller re-runs from its saved input on
 * the per-layer path (shapegen_amd.diffusion.LatentDiffusion._run does). */
int pcd_latent_persist_status(const void* workspace, unsigned* status_host, void* stream);
/* fault injection for the recovery tests: in the following launches workgroup `workgroup` (role index 0..255) leaves at the
 * start of step `step`, as a workgroup that never became resident would; workgroup < 0 switches it off. */
int pcd_latent_persist_inject_fault(pcd_latent_persist_t* h, int workgroup, int step);
/* host-only self check of the kernel's static work assignment (no device needed): bytes of one buffer set, or -1 */
int pcd_latent_persist_plan_check(void);
/* host-only: phase id (2 * layer + finish, -1 = none) of every workgroup's units, [256][8] ints */
int pcd_latent_persist_plan_dump(int* phases_host);

/* ------------------------------------------------------ 3-D convolution (a12, K9)
 * Implicit-GEMM Conv3d on NDHWC fp16 activations (replaces nn.Conv3d / nn.ConvTranspose3d +
 * BatchNorm3d(eval) + ReLU / residual add of networks.py:2225-2264, 471-504; BN folded on the host).
 * Rows m enumerate (b, oz, oy, ox) over rows_d x rows_h x rows_w; tap t reads input voxel
 * (oz*stride+dz_t, oy*stride+dy_t, ox*stride+dx_t) (zero outside the grid) and multiplies
 * w[cout][t*cin .. t*cin+cin); K = ntaps*cin zero-padded to kpad (multiple of 64).
 * Output voxel = (oz*out_scale+out_off_z, ...) of an out_d x out_h x out_w grid, so one
 * output-parity class of ConvTranspose3d(k4,s2,p1) is a 2x2x2-tap call with out_scale 2.
 * out = relu?( conv + bias (+ resid) ).  taps: device int32 [ntaps], bytes (dz, dy, dx, 0) signed.
 */
typedef struct {
    const void* in; int batch, in_d, in_h, in_w, cin;      /* fp16 [B][D][H][W][cin], cin power of two >= 8 */
    int rows_d, rows_h, rows_w, stride;
    const int* taps; int ntaps, kpad;
    const void* w; const float* bias;                      /* fp16 [cout][kpad], fp32 [cout] */
    const void* resid; int relu;                           /* optional fp16 residual indexed like out */
    void* out; int cout;                                   /* fp16 [B][out_d][out_h][out_w][cout], cout % 8 == 0 */
    int out_d, out_h, out_w, out_scale, out_off_z, out_off_y, out_off_x;
    const void* zero_page;                                 /* >= 128 bytes of zeros */
    /* optional SECOND SOURCE (NULL = none): fp16 [B][rows_d][rows_h][rows_w][cin2] on the row grid; K columns
     * [ntaps*cin, ntaps*cin + cin2) of w multiply row m of in2 -- a 1x1x1 convolution of another tensor summed into the same
     * accumulators, which is how ResidualBlock3D's projection shortcut (networks.py:485-490, 500-503) rides in conv2's launch:
     * relu(bn2(conv2 h) + downsample(x)) = relu([gather(h) | x] . [W2 | Wds]^T + b2 + bds), one fp16 rounding, no shortcut
     * tensor.  Needs stride 1, out_scale 1, rows == in dims, resid NULL, ntaps*cin % 64 == 0; the implicit GEMM takes cin2 a
     * power of two >= 64 with kpad == ntaps*cin + cin2, the k3s1 halo kernel cin 64, cout % 64 == 0 with cin2 == 32. */
    const void* in2; int cin2;
} pcd_conv3d_desc_t;
int pcd_conv3d_f16(const pcd_conv3d_desc_t* d, void* stream);
/* n (1..8) problems in ONE launch that share everything except taps, w and out_off_* -- the 2x2x2
 * output-parity classes of ConvTranspose3d(k4, s2, p1) (networks.py:2243,2248,2253) -- and, when the
 * launch would leave the chip mostly idle (few output rows, long K: encoder.9-12, decoder.0), split-K
 * over blockIdx.z with fp32 partial slabs in `workspace`, summed in split order by a finish kernel that
 * applies the same bias/residual/ReLU epilogue.  pcd_conv3d_workspace_bytes() returns the scratch that
 * split needs (0 = the launch is not split); with a NULL/short workspace the launch runs unsplit. */
size_t pcd_conv3d_workspace_bytes(const pcd_conv3d_desc_t* descs, int n);
int pcd_conv3d_f16_multi(const pcd_conv3d_desc_t* descs, int n, void* workspace, size_t workspace_bytes,
                         void* stream);
/* testing / tuning hook for pcd_conv3d_k3s1_f16 at C_in = 32, C_out <= 32: 0 = 128-row workgroups, 1 (default) = 256-row workgroups
 * (4 x 8 x 8 voxels, eight waves) where the grid has at least 512 of them, 2 = wherever H % 8 == 0 (tests).  Same bits.
 * pcd_conv3d_last_sigmoid_packed: default = per-tap partial products on the matrix pipe (a wave per four 8 x 8 output slices, every input slice read once; the fp32
 * weights as fp16 hi + lo + lo2 parts); + 16384: a wave per output slice; + 24: 8 x 8 x 8 output blocks with one MFMA per tap and 16 voxels (from the packed copy); + 16: the same blocks on the VALU (fp32 weights); + 8:
 * 4 x 4 x 8 blocks on the VALU.  The five forms agree to 1e-6.  + 64 / + 32 / + 96: split-K aims at 384 / 768 / 1024 workgroups
 * instead of 512 (all measured slower on VAE3DLarge); + 512: pcd_conv3d_first on 4 x 4 x 8 tiles where 8 x 8 x 8 would fit (same bits); + 128 / + 256: timing
 * ablations of the last layer's kernel, + 1024 x bits: of the 128 x 128 implicit GEMM (1 / 2 operand staging, 4 fragment reads, 8 MFMAs) (OUTPUTS WRONG).  TEST / BENCHMARK ONLY: process-global. */
int pcd_conv3d_config(int tall_halo_tiles);
/* Conv3d(k3, stride 1, pad 1) (+ folded BN, residual, ReLU) with the input halo of a 4x4x8 output block held in
 * LDS and reused by all 27 taps -- the 32^3 layers of VAE3DLarge (encoder.2, decoder.8-11; networks.py:2227,
 * 2256-2259), whose gather traffic bounds the generic kernel.  Same descriptor; the tap table is implied
 * (regular 3x3x3, (kz, ky, kx) order, offsets -1..1) and d->taps is not read.  Supported: cin 32 or 64, spatial
 * dims multiples of (4, 4, 8), out_scale 1; pcd_conv3d_k3s1_supported() tells (1/0), unsupported -> PCD_ERR_ARG. */
int pcd_conv3d_k3s1_supported(const pcd_conv3d_desc_t* d);
int pcd_conv3d_k3s1_f16(const pcd_conv3d_desc_t* d, void* stream);
/* Conv3d(k3, stride 1, pad 1), C_in = 128, 64 or 32, with the weights in registers (csrc/conv3d.hip: conv3d_halo_wreg_kernel): the same function and descriptor as
 * pcd_conv3d_k3s1_f16 (bias, residual, ReLU; d->w / d->kpad / d->taps are not read), the weights given in MFMA-fragment order -- wfrag, made once per
 * layer by pcd_conv3d_pack_wfrag from the [cout][kpad] matrix (pcd_conv3d_wfrag_bytes(cin, cout) bytes).  A workgroup owns 4 x 8 x 8 output voxels with their
 * halo in LDS; weight fragments are coalesced 1-KB global loads straight into MFMA operand registers; no barrier in the tap loop.  Supported: cin 128 (eight
 * waves over a 153.6-KB halo, cout % 128 == 0), 64 or 32 (cout 32 or a multiple of 64), dims multiples of (4, 8, 8), at cin >= 64 a second source of cin2 = 32 or
 * 64 channels (pack with the same cin2) (its weight columns are "tap 27" of wfrag, its rows come straight
 * from global memory); pcd_conv3d_k3s1_wreg_supported() tells.  Same sums in the same k order as
 * pcd_conv3d_k3s1_f16 per accumulator (bitwise the same outputs). */
size_t pcd_conv3d_wfrag_bytes(int cin, int cout);
int pcd_conv3d_pack_wfrag(const void* w, int kpad, int cin, int cout, int cin2, void* wfrag, void* stream);
int pcd_conv3d_k3s1_wreg_supported(const pcd_conv3d_desc_t* d);
int pcd_conv3d_k3s1_wreg_f16(const pcd_conv3d_desc_t* d, const void* wfrag, void* stream);
/* Conv3d(k4, s2, p1) (+ folded BN) (+ ReLU) with LDS-resident input (encoder.3/4, networks.py:2229-2230; csrc/conv3d.hip: conv3d_k4s2_halo_kernel):
 * in fp16 [B][d][h][w][cin], w fp16 [cout][kpad] in the tap-major layout of pcd_conv3d_f16 (k = ((kz * 4 + ky) * 4 + kx) * cin + c), out fp16
 * [B][d/2][h/2][w/2][cout].  The kernel walks the eight input-parity classes (each a 2 x 2 x 2 stride-1 convolution of one sub-sampled grid) of a
 * 4 x 4 x 8 output block with that class's sub-grid halo in LDS.  Supported: cin 64, cout 64, d, h, w multiples of (8, 8, 16) and <= 64;
 * pcd_conv3d_k4s2_halo_supported() tells (1 / 0), unsupported -> PCD_ERR_ARG.  Same sums as pcd_conv3d_f16 in another order (exact on integers). */
int pcd_conv3d_k4s2_halo_supported(int batch, int d, int h, int w, int cin, int cout, int kpad);
int pcd_conv3d_k4s2_halo_f16(const void* in, int batch, int d, int h, int w, int cin, const void* wgt, int kpad, const float* bias, int relu,
                             int cout, void* out, void* stream);
/* ConvTranspose3d(k4, s2, p1) + ReLU (decoder.6/7, networks.py:2253-2254) with the 6 x 6 x 10 input halo of a 4 x 4 x 8 block of INPUT
 * voxels in LDS and all eight output-parity classes computed from it (csrc/conv3d.hip: convT3d_halo_kernel): in fp16 [B][d][h][w][cin],
 * out fp16 [B][2d][2h][2w][cout].  w8[4 pz + 2 py + px] = that class's fp16 [cout][8 * cin] matrix, K column (tz * 2 + tz... tap t) * cin + c
 * with t = (tz * 2 + ty) * 2 + tx, tap t_axis of parity p_axis reading input offset p - t (kernel index 1 + 2 t - p ... i.e. o = 2 i - 1 + k:
 * parity 0 -> k = 1, 3 at offsets 0, -1; parity 1 -> k = 0, 2 at offsets +1, 0) -- the layout pcd_vae_convT_t carries.  Supported:
 * cin 128, cout 64, dims multiples of (4, 4, 8); pcd_convt3d_k4s2_halo_supported() tells (1 / 0), unsupported -> PCD_ERR_ARG.
 * Same sums as the eight pcd_conv3d_f16 class launches in another order (fp32 accumulation; exact on integer operands). */
int pcd_convt3d_k4s2_halo_supported(int batch, int d, int h, int w, int cin, int cout);
int pcd_convt3d_k4s2_halo_f16(const void* in, int batch, int d, int h, int w, int cin, const void* const* w8, const float* bias,
                              int cout, void* out, void* stream);
/* encoder.0: Conv3d(1, cout, k3, stride 1|2, p1) (+ folded BN) + ReLU straight from the fp32 occupancy
 * grid x [B][D][H][W]; w fp32 [cout][27], out fp16 NDHWC (VAE3DLarge networks.py:2226, VAE3D :1999). */
int pcd_conv3d_first(const float* x, int batch, int d, int h, int w, int stride, const float* wgt,
                     const float* bias, int cout, void* out, void* stream);
/* decoder.12 + decoder.13: Conv3d(32, 1, k3, p1) + Sigmoid -> fp32 [B][D][H][W]; w fp32 [27][cin]. */
int pcd_conv3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                            float bias, float* out, void* stream);
/* the same layer on the matrix pipe: wfrag = pcd_conv3d_last_pack's copy of wgt (pcd_conv3d_last_packed_bytes() bytes: the fp32 weights as fp16 hi / lo / lo2 rows of
 * the MFMA A operand of every tap); d, h, w multiples of 8, otherwise (or wfrag NULL, or pcd_conv3d_config + 8 / + 16) pcd_conv3d_last_sigmoid's kernels run from wgt */
size_t pcd_conv3d_last_packed_bytes(void);
int pcd_conv3d_last_pack(const float* wgt, void* wfrag, void* stream);
int pcd_conv3d_last_sigmoid_packed(const void* in, int batch, int d, int h, int w, int cin, const float* wgt, const void* wfrag, float bias, float* out,
                                   void* stream);
/* VAE3D's last layer (networks.py:2018-2019): ConvTranspose3d(32, 1, k3, s2, p1, output_padding 1) + Sigmoid;
 * in fp16 NDHWC [B][d][h][w][32], w fp32 [27][32] tap-major, out fp32 [B][2d][2h][2w]. */
int pcd_convt3d_last_sigmoid(const void* in, int batch, int d, int h, int w, int cin, const float* wgt,
                             float bias, float* out, void* stream);
/* z = mu + eps * exp(0.5 * logvar)  (networks.py:2323-2325), fp32 */
int pcd_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream);

/* ---- whole VAE3DLarge.encode / decode (networks.py:2299-2310, 2327-2339) behind a handle ----
 * The host packer folds eval-mode BatchNorm3d into the conv weights and lays them out as the layer-level entry
 * points above expect: Conv3d [cout][k^3 * cin] tap-major zero-padded to kpad; each ConvTranspose3d(k4,s2,p1) as 8
 * output-parity classes [cout][8 * cin] with their 2x2x2 tap tables (class index = 4 pz + 2 py + px). */
typedef struct { const void* w; const float* b; int kpad, cin, cout, k; } pcd_vae_conv_t;
/* ResidualBlock3D.  fused_ds (with has_ds): c2 is [conv2 | downsample] -- kpad >= 27*c2.cin + ds.cin, the shortcut's folded weights in the
 * K columns behind the 27 taps, c2.b = b2 + b_ds -- and the block runs as two launches (pcd_conv3d_desc_t.in2); ds.cin / ds.cout describe
 * the shortcut, ds.w / ds.b are not read. */
typedef struct { pcd_vae_conv_t c1, c2, ds; int has_ds; int fused_ds; } pcd_vae_res_t;
typedef struct { const void* w[8]; const int* taps[8]; const float* b; int cin, cout; } pcd_vae_convT_t;
typedef struct {
    int latent_dim;
    const float* enc0_w; const float* enc0_b;          /* encoder.0 fp32 [32][27], [32] */
    pcd_vae_res_t enc_res[4];                          /* encoder.2, 5, 8, 11 */
    pcd_vae_conv_t enc_down[3];                        /* encoder.3, 6, 9   (k4, s2, p1) */
    pcd_vae_conv_t enc_last;                           /* encoder.12        (k4, p0) */
    const void* fc_w; const float* fc_b;               /* [fc_mu ; fc_logvar] fp16 [2*latent][512], fp32 [2*latent] */
    const void* din_w; const float* din_b;             /* decoder_input fp16 [512*64][latent], rows permuted to (z,y,x,c) */
    pcd_vae_convT_t dec_up[3];                         /* decoder.0, 3, 6 */
    pcd_vae_res_t dec_res[4];                          /* decoder.2, 5, 8, 11 */
    pcd_vae_conv_t dec_conv9;                          /* decoder.9 */
    const float* last_w; float last_b;                 /* decoder.12 fp32 [27][32] tap-major, bias */
    const int* taps3; const int* taps4s2; const int* taps4p0; const int* taps1;   /* regular tap tables (k3 p1, k4 p1, k4 p0, k1) */
    const void* zero_page;                             /* >= 128 bytes of zeros */
} pcd_vae_desc_t;
typedef struct pcd_vae pcd_vae_t;
int pcd_vae_create(const pcd_vae_desc_t* desc, pcd_vae_t** out);
/* testing / tuning hook, a bit mask of the round-4 kernels in use: 2 = decoder.6 through pcd_convt3d_k4s2_halo_f16, 4 = encoder.3 through
 * pcd_conv3d_k4s2_halo_f16, 8 = the k3 layers through pcd_conv3d_k3s1_wreg_f16; 1 (default) = all of them, 0 = none (implicit GEMM / LDS-ring forms) */
int pcd_vae_config(int convt_halo);
void pcd_vae_destroy(pcd_vae_t* h);
size_t pcd_vae_workspace_bytes(int batch);
/* vox fp32 [B][1][32][32][32] in [0,1] -> mu_logvar fp32 [B][2*latent] = [mu | logvar] */
int pcd_vae_encode(pcd_vae_t* h, const float* vox, int batch, float* mu_logvar, void* workspace, size_t workspace_bytes,
                   void* stream);
/* z fp32 [B][latent] -> occupancy probabilities fp32 [B][1][32][32][32] */
int pcd_vae_decode(pcd_vae_t* h, const float* z, int batch, float* out, void* workspace, size_t workspace_bytes,
                   void* stream);

/* -------------------------------------------------------- set attention (K6/K7)
 * LayerNorm over C (eps 1e-5, biased var; networks.py:62,68) fp16 in -> fp16 out. */
int pcd_layernorm_f16(const void* x, int64_t rows, int c, const float* gamma, const float* beta,
                      void* out, void* stream);
/* softmax(Q K^T) V per (shape, head), flash style, never materialising N x N
 * (replaces the bmm/softmax/bmm path of nn.MultiheadAttention, networks.py:61,81).
 * qkv fp16 [B*N][3C] = [q | k | v] as produced by in_proj; q is scaled by 1/sqrt(d)
 * inside.  out fp16 [B*N][C] (heads concatenated, ready for out_proj).
 * V is transposed on the fly (ds_read_b64_tr_b16), so the workspace query returns 0 and
 * workspace may be NULL; the parameters are kept for ABI stability. */
size_t pcd_set_attention_workspace_bytes(int batch, int n_points, int c);
int pcd_set_attention_f16(const void* qkv, int batch, int n_points, int c, int heads,
                          void* out, void* workspace, size_t workspace_bytes, void* stream);
/* testing / tuning hook.  Dispatch: N % 256 == 0 selects the software-pipelined kernels (two 32-query blocks per wave: set_attention_sp_kernel at d = 64,
 * set_attention_spn_kernel at d = 32 / 16); every other length runs the one-block kernel with the same max-free softmax (set_attention_om_kernel).
 *   0 default; 1 = always the round-1 kernel (running max per tile); 2 = always the one-block max-free kernel;
 *   3 / 4 = d = 32 / 16 on the one-block kernel / on the pipelined kernel (default);
 *   5 / 6 = the d = 64 kernel as workgroups of four (default) / eight waves (same bits, measured equal);
 *   16 + bits = timing ablations of the pipelined kernels (1 no K/V restaging, 2 no rare-path test, 4 no waits / barriers): OUTPUTS ARE WRONG while set; 16 clears.
 * TEST / BENCHMARK ONLY: process-global. */
int pcd_set_attention_config(int force_generic);
/* name of the kernel the last pcd_set_attention_f16 call of this process launched (measurement reports quote it) */
const char* pcd_set_attention_last_kernel(void);

/* x[m][c] + e[m / rows_per_shape][c] -> out (fp16 in/out, e fp32): the additive per-level time
 * embeddings of UNetAttentionPointExperimental (networks.py:669-698). */
int pcd_add_shape_bias_f16(const void* x, int64_t m, int c, int rows_per_shape, const float* e,
                           void* out, void* stream);
/* the same with the per-shape rows of e `e_stride` floats apart (0: one row shared by every shape) */
int pcd_add_shape_bias_strided_f16(const void* x, int64_t m, int c, int rows_per_shape, const float* e,
                                   int64_t e_stride, void* out, void* stream);
/* tail of UNetAttentionPointExperimental (networks.py:647-650,700-702): dec1 = PointNetLayer(128,3,3)
 * on cat[a | b] then Conv1d(3,3); BN folded.  w1 fp32 [3][ka+kb], w234 fp32 [3][3][3], b234 [3][3];
 * out fp32 [M][3]. */
int pcd_tail3(const void* a, int ka, const void* b, int kb, int64_t m, const float* w1, const float* b1,
              const float* w234, const float* b234, float* out, void* stream);

/* ---- whole SetAttentionBlock (networks.py:51-83) and UNetAttentionPointExperimental (networks.py:597-722) ----
 * Weights fp16 [C_out][C_in] (nn.Linear / in_proj layout), biases and LayerNorm affine fp32. */
typedef struct {
    int dim;                                         /* C: 64, 128 or 256 */
    const void* w_in;  const float* b_in;            /* attention.in_proj  [3C][C] */
    const void* w_out; const float* b_out;           /* attention.out_proj [C][C] */
    const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
    const void* w_ff1; const float* b_ff1;           /* ff.0 [4C][C] */
    const void* w_ff2; const float* b_ff2;           /* ff.2 [C][4C] */
    const void* ln_in_packed; const void* ln_ff1_packed;   /* optional (dim 256): pcd_pw_wide_ln_linear_pack images of (ln1, in_proj: 3 passes) and
                                                      * (ln2, ff.0: 4 passes): pcd_sab_forward then runs LN + Linear as one launch each when rows % 256 == 0 */
    const void* ffn_packed;                          /* optional (dim 256): pcd_wide_ffn_pack's image of ln2, ff.0, ff.2: LN2 + Linear + ReLU + Linear + residual as ONE launch
                                                      * (csrc/wideffn.hip; the 4C-wide hidden tensor is never written); rows % 128 == 0 */
    const void* tail_packed;                         /* optional (dim 64 / 128): pcd_sab_tail_pack's image of w_out .. b_ff2; pcd_sab_forward then runs
                                                      * out_proj + residual + LN2 + FFN + residual as ONE launch when rows % 256 == 0.  NULL: four launches.
                                                      * Ignored by the fp32 parity entry points */
} pcd_sab_desc_t;
/* The feed-forward half of the block at dim 256 as one launch (csrc/wideffn.hip; reference networks.py:62-68, 82):  y = x1 + W2 relu(W1 LN2(x1) + b1) + b2,
 * x1, y fp16 [rows][256] (y may not alias x1), rows % 128 == 0.  Two waves share 32 points and split the channels of both products, the 1024-wide hidden row lives in
 * registers / LDS 128 channels at a time, only the weights stream (pcd_wide_ffn_pack: 32 fragment-order stage images + b1 | b2 | gamma | beta,
 * pcd_wide_ffn_packed_bytes() bytes).  w1 [1024][256], w2 [256][1024] fp16; biases and the LayerNorm affine fp32. */
size_t pcd_wide_ffn_packed_bytes(void);
int pcd_wide_ffn_supported(int dim, int64_t rows);
int pcd_wide_ffn_pack(const void* w1, const float* b1, const void* w2, const float* b2, const float* ln_g, const float* ln_b, void* packed, void* stream);
int pcd_wide_ffn_f16(const void* packed, const void* x, int64_t rows, void* y, void* stream);
/* the same with the additive per-shape rows of networks.py:688 on the way out: y = fp16(y + post_e[row / rows_per_shape][c]) (fp32 rows of 256, e_stride floats
 * apart, 16-byte aligned; NULL = none) -- the rounding points of pcd_add_shape_bias_strided_f16 behind the block, bitwise the two launches */
int pcd_wide_ffn_bias_f16(const void* packed, const void* x, int64_t rows, int rows_per_shape, const float* post_e, int64_t e_stride, void* y, void* stream);
/* A/B hook (TEST / BENCHMARK ONLY, process-global): which waves request the weight images, see csrc/wideffn.hip; same bits either way */
int pcd_wide_ffn_config(int split);   /* 0 / 1 request form; 16 + bits = timing ablations of the kernel (OUTPUTS WRONG while set; 16 clears): TEST / BENCHMARK ONLY */
/* bytes of scratch one block needs for `rows` = B*N points */
size_t pcd_sab_workspace_bytes(int64_t rows, int dim);
/* the block's tail behind the attention kernel as one launch (csrc/sab_tail.hip; reference networks.py:78-83, the second half of SetAttentionBlock.forward):
 *   y = x1 + W2 relu(W1 LN2(x1) + b1) + b2,  x1 = x + W_out a + b_out      a = the heads' outputs [rows][dim], x = the block's input, all fp16
 * pcd_sab_tail_pack writes the fragment-order stage images of w_out / w_ff1 / w_ff2 (and w_in, for pcd_sab_head_f16) and the fp32 biases / LayerNorm affine of `d` into `packed`
 * (pcd_sab_tail_packed_bytes(dim) bytes of device memory; 0 = dim not supported); pcd_sab_tail_supported: dim 64 or 128 and rows % 256 == 0.
 * pcd_sab_tail_config(0) makes pcd_sab_forward / pcd_attn_unet_forward keep the four launches (A/B, tests); + 2: the fused launches with every wave requesting its
 * share of a stage's LDS-DMA pieces instead of one wave per SIMD (A/B); pcd_sab_tail_enabled reads bit 0 back.  TEST / BENCHMARK ONLY: process-global. */
size_t pcd_sab_tail_packed_bytes(int dim);
int pcd_sab_tail_supported(int dim, int64_t rows);
int pcd_sab_tail_pack(const pcd_sab_desc_t* d, void* packed, void* stream);
int pcd_sab_tail_f16(int dim, const void* packed, const void* a, const void* x, int64_t rows, void* y, void* stream);
/* the block's head in the same form, from the same image: qkv [rows][3 dim] = in_proj(LN1(x)) (networks.py:81), one launch instead of LayerNorm + GEMM */
int pcd_sab_head_f16(int dim, const void* packed, const void* x, int64_t rows, void* qkv, void* stream);
/* both with the attention U-Net's additive per-level time embeddings (networks.py:669-698) folded in: x is read as fp16(x + pre_e[row / rows_per_shape]) and
 * y leaves as fp16(y + post_e[row / rows_per_shape]), each rounded exactly as pcd_add_shape_bias_strided_f16 rounds it; pre_e / post_e fp32, rows e_stride
 * floats apart (0: one row for every shape; a multiple of 4), 16-byte aligned, either may be NULL */
int pcd_sab_tail_bias_f16(int dim, const void* packed, const void* a, const void* x, int64_t rows, int rows_per_shape, const float* pre_e,
                          const float* post_e, int64_t e_stride, void* y, void* stream);
int pcd_sab_head_bias_f16(int dim, const void* packed, const void* x, int64_t rows, int rows_per_shape, const float* pre_e, int64_t e_stride,
                          void* qkv, void* stream);
int pcd_sab_tail_config(int fused);
int pcd_sab_tail_enabled(void);
/* y = x + MHA(LN1 x); y = y + W2 relu(W1 LN2 y)   x, y fp16 [B*N][C], y must not alias x */
int pcd_sab_forward(const pcd_sab_desc_t* d, const void* x, int batch, int n_points, int heads, void* y,
                    void* workspace, size_t workspace_bytes, void* stream);

#define PCD_ATTN_UNET_NLIN 14       /* enc1.conv2,3  enc2.conv1-3  enc3.conv1-3  dec3.conv1-3  dec2.conv1-3 (BN folded) */
#define PCD_ATTN_UNET_NSAB 7        /* att1 att2 att3 bottleneck att_dec3 att_dec2 att_dec1 */
#define PCD_ATTN_UNET_NEMB 6        /* emb1 emb2 emb3 emb_dec3 emb_dec2 emb_dec1 */
#define PCD_ATTN_UNET_TB 704        /* floats of one time-bias row: [enc1 bias 64 | emb2 64 | emb3 128 | emb_dec3 256 | emb_dec2 128 | emb_dec1 64] */
typedef struct {
    int dim, time_dim, heads;
    const float* freqs;                              /* sinusoidal frequencies [time_dim/2] */
    const float* tw0; const float* tb0; const float* tw2; const float* tb2;    /* time_mlp */
    const float* emb_w[PCD_ATTN_UNET_NEMB]; const float* emb_b[PCD_ATTN_UNET_NEMB];   /* [c][dim], c = 3,64,128,256,128,64 */
    const float* e1w; const float* e1b;              /* enc1.conv1 + bn1 folded, fp32 [64][3], [64] */
    pcd_linear_desc_t lin[PCD_ATTN_UNET_NLIN];
    pcd_sab_desc_t sab[PCD_ATTN_UNET_NSAB];
    const float* t_w1; const float* t_b1; const float* t_w234; const float* t_b234;   /* dec1 + output, see pcd_tail3 */
} pcd_attn_unet_desc_t;
typedef struct pcd_attn_unet pcd_attn_unet_t;
int pcd_attn_unet_create(const pcd_attn_unet_desc_t* desc, pcd_attn_unet_t** out);
void pcd_attn_unet_destroy(pcd_attn_unet_t* h);
size_t pcd_attn_unet_workspace_bytes(int batch, int n_points);
/* time path for n_t values of t: tbias fp32 [n_t][PCD_ATTN_UNET_TB] (time_mlp, the six emb* layers, and emb1 pushed
 * through enc1.conv1: conv(x + e) = W x + (W e + b), networks.py:664-672); scratch fp32 [n_t][dim] */
int pcd_attn_unet_time_bias(pcd_attn_unet_t* h, const float* t, int n_t, float* scratch, float* tbias, void* stream);
/* eps = model(x, t) given the time-bias rows: tbias row (b * tbias_shape_stride), stride 0 = one t for every shape */
int pcd_attn_unet_forward(pcd_attn_unet_t* h, const float* x, int batch, int n_points, const float* tbias,
                          int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes, void* stream);
/* parity taps of the last forward: "x1" [B*N][64], "x2" [B*N][128], "x3" [B*N][256] fp16 (the skip tensors, networks.py:663-672) */
int pcd_attn_unet_tap(pcd_attn_unet_t* h, const char* name, int batch, int n_points, const void* workspace, void* dst,
                      size_t dst_bytes, void* stream);

/* ------------------------------------------------ fp32 parity mode of the set-attention block and the attention U-Net (csrc/attn_f32.hip)
 * pcd_sab_forward / pcd_attn_unet_forward with fp32 weights (EVERY weight pointer of the descriptors is fp32 here), fp32 activations and
 * fp32 arithmetic (pcd_gemm_f32, fp32 LayerNorm, a plain fp32 softmax(q k^T / sqrt d) v kernel): the reference's arithmetic type
 * (networks.py:51-83, 597-722), held to 1e-4.  x / y / eps fp32; tbias = the 704-float rows of pcd_attn_unet_time_bias (fp32 in both modes).
 * Taps of the last forward: x1 / x2 / x3, fp32. */
size_t pcd_sab_f32_workspace_bytes(int64_t rows, int dim);
int pcd_sab_f32_forward(const pcd_sab_desc_t* d, const float* x, int batch, int n_points, int heads, float* y, void* workspace,
                        size_t workspace_bytes, void* stream);
typedef struct pcd_attn_unet_f32 pcd_attn_unet_f32_t;
int pcd_attn_unet_f32_create(const pcd_attn_unet_desc_t* desc, pcd_attn_unet_f32_t** out);
void pcd_attn_unet_f32_destroy(pcd_attn_unet_f32_t* h);
size_t pcd_attn_unet_f32_workspace_bytes(int batch, int n_points);
int pcd_attn_unet_f32_forward(pcd_attn_unet_f32_t* h, const float* x, int batch, int n_points, const float* tbias,
                              int tbias_shape_stride, float* eps, void* workspace, size_t workspace_bytes, void* stream);
int pcd_attn_unet_f32_tap(pcd_attn_unet_f32_t* h, const char* name, int batch, int n_points, const void* workspace, void* dst,
                          size_t dst_bytes, void* stream);

/* ------------------------------------------------------------- metrics (K10-K12)
 * normalize_to_cube (metrics.py:7-21) for B clouds of N points, fp32 in/out. */
int pcd_normalize_to_cube(const float* pts, int batch, int n, float* out, void* stream);
/* Chamfer sums (metrics.py:41-46) on already-normalised clouds:
 * sums[b][0] = sum_i min_j |x_i - y_j|, sums[b][1] = sum_j min_i |x_i - y_j| (unsquared L2),
 * direct differences in fp32 (no matmul cancellation).  sums fp32 [B][2]. */
int pcd_chamfer_sums(const float* x, const float* y, int batch, int n1, int n2, float* sums, void* stream);
/* voxelize (utils.py:488-509): occupancy fp32 [B][R][R][R] indexed [x][y][z]; caller zeroes out. */
int pcd_voxelize(const float* pts, int batch, int n, int res, float* vox, void* stream);
/* voxel_tensor_to_point_clouds (utils.py:511-539): for each grid (D,H,W) emit the points with
 * v > threshold in row-major (z,y,x) order as [x,y,z] mapped to [-1,1].
 * counts int32 [B]; points fp32 [B][D*H*W][3] (first counts[b] rows valid). */
int pcd_voxels_to_points(const float* vox, int batch, int d, int h, int w, float threshold,
                         int32_t* counts, float* points, void* stream);
/* mean BCE(x, target) with torch's log clamp at -100 (metrics.py:181); out fp32 [1]. */
int pcd_binary_bce_mean(const float* x, const float* target, int64_t n, float* out, void* stream);

/* ---- Sinkhorn EMD (metrics.py:94-158, `earth_mover_distance_gpu`), cost matrix never stored ----
 * out_max[0] = max over (b,i,j) of |x_i - y_j|   (the global C.max() of metrics.py:123) */
int pcd_pairwise_max_dist(const float* x, const float* y, int batch, int n, int m, float* out_max, void* stream);
/* one half-iteration (metrics.py:141 / :144): for the rows of p against cloud q with dual dual_q,
 * dual_p[i] <- epsilon * (log_marginal - logsumexp_j(-|p_i - q_j| / (cmax*epsilon) + dual_q[j]));
 * err_max[0] = max_i |new - old| (metrics.py:147-148).  duals fp32 [B][n]. */
int pcd_sinkhorn_dual_update(const float* p, const float* q, int batch, int np, int nq, const float* cmax,
                             float epsilon, float log_marginal, const float* dual_q, float* dual_p,
                             float* err_max, void* stream);
/* cost[b] = sum_ij exp(-C/eps + alpha_i + beta_j) * C_ij (metrics.py:153-156); row_scratch fp32 [B][n] */
int pcd_sinkhorn_cost(const float* x, const float* y, int batch, int n, int m, const float* cmax,
                      float epsilon, const float* alpha, const float* beta, float* row_scratch,
                      float* cost, void* stream);

/* ---- a batch of independent cloud pairs in one enqueue (the evaluation loop of test_point_ddpm.py:85-92) ----
 * Pair p = a[p][0..na[p]) vs b[p][0..nb[p]) inside padded fp32 arrays [P][na_max][3] / [P][nb_max][3]; na, nb device
 * int32 [P]; a pair with a count of 0 gets a NaN row (the reference's compute_metrics would raise on an empty cloud).  rows fp32 [P][3] = (Chamfer with scaling 1 (metrics.py:23-47), Sinkhorn EMD
 * (metrics.py:94-158, only when with_sinkhorn, else 0), voxel BCE (metrics.py:177-181)), each computed exactly as
 * compute_metrics(a_p, b_p) would for that pair alone: per-pair normalize_to_cube, per-pair cost normalisation, per-pair
 * convergence test (kept on the device: all max_iter iterations are enqueued, converged pairs skip theirs; no host
 * synchronisation).  log_mu[p] = log(1/na[p] + 1e-10), log_nu likewise, fp32 device arrays from the host's torch ops. */
size_t pcd_pair_metrics_workspace_bytes(int pairs, int na_max, int nb_max);
int pcd_pair_metrics(const float* a, const int* na, int na_max, const float* b, const int* nb, int nb_max, int pairs,
                     int with_sinkhorn, float epsilon, float thresh, int max_iter, const float* log_mu,
                     const float* log_nu, float* rows, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------ training step of the point denoiser
 * diffusion.py:70-86,170-186 (add_noise -> model in train() mode -> F.l1_loss -> AdamW, diffusion.py:60).
 * The dense products (forward, backward-data, backward-weight) are pcd_gemm_f16* calls; these entry points
 * are the BatchNorm1d-with-batch-statistics forward/backward (networks.py:31-48), the max-pool with its
 * argmax (networks.py:807), the K=3 / C=3 edge layers, reductions, transposes, loss and optimizer.
 * Activations / activation gradients fp16 [M][C]; statistics, parameter gradients, optimizer state fp32.
 * Gradients carry the caller's loss scale; pcd_adamw_step divides it out. */
/* out[g][c] = sum over the rows_per_group rows of group g of x[row][c]   (bias gradients, per-shape sums) */
int pcd_colsum_f16(const void* x, int64_t rows_per_group, int groups, int c, float* out, void* stream);
/* BatchNorm1d training statistics over the M rows of the fp32 conv output z [M][c] (c % 8 == 0): mean[c], biased
 * var[c] (one pass, sums shifted by row 0 so that E[d^2] - E[d]^2 does not cancel); if running_* are given they are updated as torch does (momentum, UNBIASED variance).
 * scratch: fp32 [2*c]. */
int pcd_bn_batch_stats(const float* z, int64_t m, int c, float momentum, float* mean, float* var,
                       float* running_mean, float* running_var, float* scratch, void* stream);
/* a = act(gamma * (z - mean) / sqrt(var + eps) + beta) -> fp16, act = ReLU if relu; z fp32 */
int pcd_bn_apply_f16(const float* z, int64_t m, int c, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, int relu, void* out, void* stream);
/* backward of pcd_bn_apply_f16 with batch statistics: da -> dz, dgamma[c], dbeta[c] */
int pcd_bn_backward_f16(const void* da, const float* z, int64_t m, int c, const float* mean, const float* var,
                        const float* gamma, const float* beta, float eps, int relu, float* dgamma, float* dbeta,
                        void* dz, void* stream);
/* dst[c][r] = src[r][c]  (operands of the backward-weight product dW = dz^T a) */
int pcd_transpose_f16(const void* src, int64_t rows, int cols, void* dst, void* stream);
/* torch.max(x, 2) of networks.py:807 on a [B*N][C]: per shape and channel the max and the FIRST index attaining it */
int pcd_colmax_argmax_f16(const void* a, int batch, int n_points, int c, float* mx, int* arg, void* stream);
/* its backward: da = 0 except da[b*N + arg[b][c]][c] = dg[b][c] */
int pcd_maxpool_backward_f16(const float* dg, const int* arg, int batch, int n_points, int c, void* da, void* stream);
/* enc1.conv1 before its BatchNorm: z[m][c] = sum_j x[m][j] w_xyz[c][j] + tbias[m / n_points][c]; x fp32 [M][3], z fp32 */
int pcd_enc1_linear(const float* x, int64_t m, int n_points, const float* w_xyz, int c1, const float* tbias,
                    float* out, void* stream);
/* out[j][k] = sum_m vec[m][j] * mat[m][k], j < 3 (dW of the 64->3 head and of enc1's xyz columns); vsum[j] = sum_m vec[m][j] or NULL */
int pcd_vec3_outer(const void* mat, const float* vec, int64_t m, int k, float* out, float* vsum, void* stream);
/* out[m][k] = sum_j vec[m][j] * w[j][k]  (backward-data of the 64->3 head) */
int pcd_vec3_expand_f16(const float* vec, const float* w, int64_t m, int k, void* out, void* stream);
/* F.l1_loss(noise, pred) of diffusion.py:182: loss_sum[0] = sum |pred - target| (divide by n on the host),
 * dpred = grad_scale * sign(pred - target) / n */
int pcd_l1_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss_sum, float* dpred,
                void* stream);
/* small fp32 product C[m][n] (+)= op(A)[m][k] op(B)[k][n] + bias[n]: time_mlp and the per-shape bias paths */
int pcd_matmul_f32(const float* a, int64_t lda, int trans_a, const float* b, int64_t ldb, int trans_b, int m, int n,
                   int k, const float* bias, int accumulate, float* c, int64_t ldc, void* stream);
int pcd_silu_f32(const float* x, int64_t n, float* y, void* stream);
int pcd_silu_backward_f32(const float* x, const float* dy, int64_t n, float* dx, void* stream);
/* latent denoiser training (networks.py:977-1049: Linear + GroupNorm(8) + ReLU on (B, C) rows; all fp32, B = batch):
 * GroupNorm forward keeping (mean, rstd) [rows][groups], and its backward (dx, dgamma, dbeta), ReLU folded in if relu */
int pcd_groupnorm_f32(const float* x, int rows, int c, int groups, const float* gamma, const float* beta, float eps,
                      int relu, float* y, float* mean, float* rstd, void* stream);
int pcd_groupnorm_backward_f32(const float* dy, const float* x, int rows, int c, int groups, const float* gamma,
                               const float* beta, const float* mean, const float* rstd, int relu, float* dx,
                               float* dgamma, float* dbeta, void* stream);
/* y = x * mask * scale: nn.Dropout forward with a given keep mask (scale = 1/(1-p)) and, applied to dy, its backward */
int pcd_mask_scale_f32(const float* x, const float* mask, float scale, int64_t n, float* y, void* stream);
int pcd_relu_f32(const float* x, int64_t n, float* y, void* stream);
int pcd_relu_backward_f32(const float* x, const float* dy, int64_t n, float* dx, void* stream);
/* voxel VAE training (networks.py:2225-2264, 471-504): a Conv3d / ConvTranspose3d layer as "gather rows, then the GEMM".
 * Channels-last fp16 rows [b*D*H*W][C]; cubic kernel k, stride, pad; transposed = 1 for ConvTranspose3d indexing.
 * col[r][t*cin + c] = x[source of output r at tap t][c] or 0, row length kp >= k^3*cin (extra columns zeroed);
 * pcd_col2im_f16 is its adjoint as a gather: dx[i][c] = sum_t dcol[output that reads i at tap t][t*cin + c]. */
int pcd_im2col_f16(const void* x, int batch, int cin, int di, int hi, int wi, int d_o, int ho, int wo, int k, int stride,
                   int pad, int transposed, int kp, void* col, void* stream);
int pcd_col2im_f16(const void* dcol, int batch, int cin, int di, int hi, int wi, int d_o, int ho, int wo, int k, int stride,
                   int pad, int transposed, int kp, void* dx, void* stream);
/* out[r][c] = act(x[r][c] + bias[c]): a ConvTranspose3d is run as product-then-col2im (the adjoint of a Conv3d), its
 * bias and ReLU applied afterwards */
int pcd_bias_act_f16(const void* x, const float* bias, int64_t rows, int c, int relu, void* out, void* stream);
/* out = a + b, with ReLU if relu (ResidualBlock3D tail); d = dout * [out > 0] */
int pcd_add_relu_f16(const void* a, const void* b, int64_t n, int relu, void* out, void* stream);
int pcd_relu_mask_f16(const void* dout, const void* out, int64_t n, void* d, void* stream);
/* recon = sigmoid(logit[i*ld]); loss_sum[0] = sum BCE(recon, target) with torch's -100 log clamp (networks.py:2387);
 * dlogit[i*ld] = grad_scale * (recon - target) / n  (BCE and Sigmoid backward fused) */
int pcd_sigmoid_bce(const void* logit, int64_t ld, const float* target, int64_t n, float grad_scale, float* loss_sum,
                    float* recon, void* dlogit, void* stream);
/* backward of z = mu + eps * exp(logvar / 2) and of kl_scale * KL (networks.py:2323-2325, 2389-2396) over n = B * latent
 * elements: dmu, dlogvar; kl_sum[0] = sum(1 + logvar - mu^2 - exp(logvar)) (KL = -0.5 * kl_sum / n) */
int pcd_vae_latent_backward(const float* mu, const float* logvar, const float* eps, const float* dz, int64_t n,
                            float kl_scale, float* dmu, float* dlogvar, float* kl_sum, void* stream);
/* torch.optim.AdamW step on one flat fp32 buffer (diffusion.py:60: lr, weight_decay 1e-5); grads are divided by grad_scale */
int pcd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCD_HIP_H */
