"""ctypes binding of the gfx950 shared library (C ABI: include/pcd_hip.h).

The product path has NO fallback: if `libpcd_hip.so` is missing or a call returns an
error, a RuntimeError is raised.  Build the library with `python __graft_entry__.py build`
(or `make -C 3d-shape-generation_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcd_hip.so")
ABI_VERSION = 2          # include/pcd_hip.h PCD_ABI_VERSION these prototypes and ctypes structures are written against
CSRC = os.path.join(_HERE, "csrc")

vp = C.c_void_p
i32, i64, u64, f32, sz = C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t

PCD_UNET_NLIN = 26


class GemmDesc(C.Structure):
    _fields_ = [("a1", vp), ("lda1", i64), ("k1", i32),
                ("a2", vp), ("lda2", i64), ("k2", i32),
                ("w", vp), ("ldw", i64),
                ("bias", vp), ("shape_bias", vp), ("rows_per_shape", i32),
                ("relu", i32), ("m", i32), ("c", i32)]


class LinearDesc(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("k", i32), ("c", i32)]


class UnetDesc(C.Structure):
    _fields_ = [("time_dim", i32), ("dim", i32), ("freqs", vp),
                ("tw0", vp), ("tb0", vp), ("tw2", vp), ("tb2", vp),
                ("e1w_xyz", vp), ("e1w_t", vp), ("e1b", vp),
                ("lin", LinearDesc * PCD_UNET_NLIN),
                ("wg", vp), ("wg_k", i32), ("wg_c", i32),
                ("head_w", vp), ("head_b", vp), ("hilo_mask", C.c_uint)]


PCD_UNET_HILO_ALLOWED = 0x3C00013        # lin 0, 1, 4, 22 .. 25 (include/pcd_hip.h)
PCD_LATENT_NLIN = 12


class LatentDesc(C.Structure):
    _fields_ = [("lin", LinearDesc * PCD_LATENT_NLIN),
                ("gn_gamma", vp * PCD_LATENT_NLIN), ("gn_beta", vp * PCD_LATENT_NLIN)]


class SabDesc(C.Structure):
    _fields_ = [("dim", i32), ("w_in", vp), ("b_in", vp), ("w_out", vp), ("b_out", vp),
                ("ln1_g", vp), ("ln1_b", vp), ("ln2_g", vp), ("ln2_b", vp),
                ("w_ff1", vp), ("b_ff1", vp), ("w_ff2", vp), ("b_ff2", vp),
                ("ln_in_packed", vp), ("ln_ff1_packed", vp), ("ffn_packed", vp), ("tail_packed", vp)]


PCD_ATTN_UNET_NLIN, PCD_ATTN_UNET_NSAB, PCD_ATTN_UNET_NEMB, PCD_ATTN_UNET_TB = 14, 7, 6, 704


class AttnUnetDesc(C.Structure):
    _fields_ = [("dim", i32), ("time_dim", i32), ("heads", i32), ("freqs", vp),
                ("tw0", vp), ("tb0", vp), ("tw2", vp), ("tb2", vp),
                ("emb_w", vp * PCD_ATTN_UNET_NEMB), ("emb_b", vp * PCD_ATTN_UNET_NEMB),
                ("e1w", vp), ("e1b", vp),
                ("lin", LinearDesc * PCD_ATTN_UNET_NLIN), ("sab", SabDesc * PCD_ATTN_UNET_NSAB),
                ("t_w1", vp), ("t_b1", vp), ("t_w234", vp), ("t_b234", vp)]


class Conv3dDesc(C.Structure):
    _fields_ = [("inp", vp), ("batch", i32), ("in_d", i32), ("in_h", i32), ("in_w", i32), ("cin", i32),
                ("rows_d", i32), ("rows_h", i32), ("rows_w", i32), ("stride", i32),
                ("taps", vp), ("ntaps", i32), ("kpad", i32),
                ("w", vp), ("bias", vp), ("resid", vp), ("relu", i32),
                ("out", vp), ("cout", i32),
                ("out_d", i32), ("out_h", i32), ("out_w", i32), ("out_scale", i32),
                ("out_off_z", i32), ("out_off_y", i32), ("out_off_x", i32),
                ("zero_page", vp), ("in2", vp), ("cin2", i32)]


class VaeConv(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("kpad", i32), ("cin", i32), ("cout", i32), ("k", i32)]


class VaeRes(C.Structure):
    _fields_ = [("c1", VaeConv), ("c2", VaeConv), ("ds", VaeConv), ("has_ds", i32), ("fused_ds", i32)]


class VaeConvT(C.Structure):
    _fields_ = [("w", vp * 8), ("taps", vp * 8), ("b", vp), ("cin", i32), ("cout", i32)]


class VaeDesc(C.Structure):
    _fields_ = [("latent_dim", i32), ("enc0_w", vp), ("enc0_b", vp),
                ("enc_res", VaeRes * 4), ("enc_down", VaeConv * 3), ("enc_last", VaeConv),
                ("fc_w", vp), ("fc_b", vp), ("din_w", vp), ("din_b", vp),
                ("dec_up", VaeConvT * 3), ("dec_res", VaeRes * 4), ("dec_conv9", VaeConv),
                ("last_w", vp), ("last_b", f32),
                ("taps3", vp), ("taps4s2", vp), ("taps4p0", vp), ("taps1", vp), ("zero_page", vp)]


# name -> (restype, argtypes).  Kept in the order of include/pcd_hip.h.
_SIGS = {
    "pcd_last_error": (C.c_char_p, []),
    "pcd_abi_version": (i32, []),
    "pcd_device_check": (i32, []),
    "pcd_gemm_f16": (i32, [C.POINTER(GemmDesc), vp, i64, vp]),
    "pcd_gemm_f16_hilo": (i32, [C.POINTER(GemmDesc), vp, i64, vp]),
    "pcd_gemm_f16_out32": (i32, [C.POINTER(GemmDesc), vp, i64, vp]),
    "pcd_gemm_f16_splitk": (i32, [C.POINTER(GemmDesc), i32, vp, vp]),
    "pcd_sum_slabs_f32": (i32, [vp, i32, i64, i32, vp, i64, vp]),
    "pcd_gemm_f16_residual": (i32, [C.POINTER(GemmDesc), vp, i64, vp, i64, vp]),
    "pcd_gemm_f16_colmax": (i32, [C.POINTER(GemmDesc), vp, i32, vp]),
    "pcd_gemm_pack_wfrag": (i32, [vp, i64, i32, i32, vp, vp]),
    "pcd_gemm_f16_colmax_wfrag": (i32, [C.POINTER(GemmDesc), vp, vp, i32, vp]),
    "pcd_gemm_wfrag_enabled": (i32, []),
    "pcd_gemm_f16_wfrag": (i32, [C.POINTER(GemmDesc), vp, vp, i64, vp]),
    "pcd_gemm_store_wfrag_enabled": (i32, []),
    "pcd_gemm_wfrag_stamps": (i32, [vp]),
    "pcd_gemm_set_config": (i32, [i32]),
    "pcd_fill_zero": (i32, [vp, sz, vp]),
    "pcd_f32_to_f16": (i32, [vp, vp, i64, vp]),
    "pcd_f16_to_f32": (i32, [vp, vp, i64, vp]),
    "pcd_time_embed": (i32, [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "pcd_linear_f32": (i32, [vp, i32, i32, vp, vp, i32, vp, vp]),
    "pcd_enc1_xyz": (i32, [vp, i64, i32, vp, i32, vp, i32, vp, vp]),
    "pcd_add_noise": (i32, [vp, vp, vp, vp, i32, i64, i64, vp, vp]),
    "pcd_remove_noise": (i32, [vp, vp, vp, vp, i32, i64, i64, vp, vp]),
    "pcd_ddim_update": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, vp, vp]),
    "pcd_ddpm_update": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, vp, vp]),
    "pcd_ddpm_update_philox": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, vp, u64, u64, u64, vp, vp]),
    "pcd_randn": (i32, [vp, i64, u64, u64, vp]),
    "pcd_step_select": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, vp]),
    "pcd_randn_step": (i32, [vp, i64, u64, u64, u64, vp, vp]),
    "pcd_head3": (i32, [vp, i64, i32, vp, vp, vp, vp]),
    "pcd_pw_chain_enc1": (i32, [vp, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "pcd_pw_chain_enc1_hilo": (i32, [vp, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "pcd_pw_chain_tail_hilo": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "pcd_pw_chain_128": (i32, [vp, i64, vp, vp, vp, vp, vp, vp]),
    "pcd_pw_chain_tail": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "pcd_unet_config": (i32, [i32]),
    "pcd_pw_wide_packed_bytes": (sz, [i32]),
    "pcd_pw_wide_pack": (i32, [i32, vp, vp, vp, vp]),
    "pcd_pw_wide_chain": (i32, [i32, vp, vp, i64, vp, vp, vp]),
    "pcd_conv1x1_supported": (i32, [i32, i32]),
    "pcd_conv1x1_f16": (i32, [vp, i64, i32, vp, i64, vp, i32, i32, vp, vp]),
    "pcd_unet_create": (i32, [C.POINTER(UnetDesc), C.POINTER(vp)]),
    "pcd_unet_destroy": (None, [vp]),
    "pcd_unet_workspace_bytes": (sz, [i32, i32]),
    "pcd_unet_forward": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_unet_profile": (i32, [vp, i32]),
    "pcd_unet_profile_read": (i32, [vp, C.POINTER(C.c_double), C.POINTER(i32)]),
    "pcd_unet_tap": (i32, [vp, C.c_char_p, i32, i32, vp, vp, sz, vp]),
    "pcd_unet_capture": (i32, [vp, vp, vp, vp, vp]),
    "pcd_gemm_f32": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, vp, vp, i32, i32, i32, i32, vp, i64, vp]),
    "pcd_gemm_f32_colmax": (i32, [vp, i64, i32, vp, i64, vp, i32, i32, vp, i32, vp]),
    "pcd_unet_f32_create": (i32, [C.POINTER(UnetDesc), C.POINTER(vp)]),
    "pcd_unet_f32_destroy": (None, [vp]),
    "pcd_unet_f32_workspace_bytes": (sz, [i32, i32]),
    "pcd_unet_f32_forward": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_unet_f32_tap": (i32, [vp, C.c_char_p, i32, i32, vp, vp, sz, vp]),
    "pcd_unet_f32_round_activations": (i32, [vp, C.c_uint]),
    "pcd_groupnorm_relu_f16": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "pcd_skinny_slabs": (i32, [i32, i32]),
    "pcd_skinny_gemm_f16": (i32, [vp, i32, vp, i32, vp, i64, i32, i32, vp, vp]),
    "pcd_skinny_finish": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "pcd_skinny_config": (i32, [i32]),
    "pcd_skinny_fused_supported": (i32, [i32, i32, i32, i32]),
    "pcd_skinny_fused": (i32, [vp, i32, vp, i32, vp, i64, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "pcd_skinny_fused_f32in": (i32, [vp, i32, vp, i64, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "pcd_latent_create": (i32, [C.POINTER(LatentDesc), C.POINTER(vp)]),
    "pcd_latent_destroy": (None, [vp]),
    "pcd_latent_workspace_bytes": (sz, [i32]),
    "pcd_latent_forward": (i32, [vp, vp, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_latent_f32_create": (i32, [C.POINTER(LatentDesc), C.POINTER(vp)]),
    "pcd_latent_f32_destroy": (None, [vp]),
    "pcd_latent_f32_workspace_bytes": (sz, [i32]),
    "pcd_latent_f32_forward": (i32, [vp, vp, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_latent_persist_supported": (i32, [i32]),
    "pcd_latent_persist_create": (i32, [C.POINTER(LatentDesc), C.POINTER(vp)]),
    "pcd_latent_persist_destroy": (None, [vp]),
    "pcd_latent_persist_workspace_bytes": (sz, [vp]),
    "pcd_latent_persist_config": (i32, [vp, i32, i32]),
    "pcd_latent_persist_trace": (i32, [vp, vp, i32]),
    "pcd_latent_persist_forward": (i32, [vp, vp, i32, vp, vp, vp, sz, vp]),
    "pcd_latent_persist_ddim": (i32, [vp, vp, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp, sz, vp]),
    "pcd_latent_persist_status": (i32, [vp, C.POINTER(C.c_uint), vp]),
    "pcd_latent_persist_inject_fault": (i32, [vp, i32, i32]),
    "pcd_latent_persist_plan_check": (i32, []),
    "pcd_latent_persist_plan_dump": (i32, [vp]),
    "pcd_conv3d_f16": (i32, [C.POINTER(Conv3dDesc), vp]),
    "pcd_conv3d_k3s1_supported": (i32, [C.POINTER(Conv3dDesc)]),
    "pcd_conv3d_k3s1_f16": (i32, [C.POINTER(Conv3dDesc), vp]),
    "pcd_conv3d_workspace_bytes": (sz, [C.POINTER(Conv3dDesc), i32]),
    "pcd_conv3d_f16_multi": (i32, [C.POINTER(Conv3dDesc), i32, vp, sz, vp]),
    "pcd_conv3d_wfrag_bytes": (sz, [i32, i32]),
    "pcd_conv3d_pack_wfrag": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "pcd_conv3d_k3s1_wreg_supported": (i32, [C.POINTER(Conv3dDesc)]),
    "pcd_conv3d_k3s1_wreg_f16": (i32, [C.POINTER(Conv3dDesc), vp, vp]),
    "pcd_conv3d_k4s2_halo_supported": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "pcd_conv3d_k4s2_halo_f16": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp]),
    "pcd_convt3d_k4s2_halo_supported": (i32, [i32, i32, i32, i32, i32, i32]),
    "pcd_convt3d_k4s2_halo_f16": (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(vp), vp, i32, vp, vp]),
    "pcd_conv3d_first": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    "pcd_convt3d_last_sigmoid": (i32, [vp, i32, i32, i32, i32, i32, vp, f32, vp, vp]),
    "pcd_conv3d_config": (i32, [i32]),
    "pcd_conv3d_last_sigmoid": (i32, [vp, i32, i32, i32, i32, i32, vp, f32, vp, vp]),
    "pcd_conv3d_last_packed_bytes": (sz, []),
    "pcd_conv3d_last_pack": (i32, [vp, vp, vp]),
    "pcd_conv3d_last_sigmoid_packed": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp]),
    "pcd_reparameterize": (i32, [vp, vp, vp, vp, i64, vp]),
    "pcd_vae_config": (i32, [i32]),
    "pcd_vae_create": (i32, [C.POINTER(VaeDesc), C.POINTER(vp)]),
    "pcd_vae_destroy": (None, [vp]),
    "pcd_vae_workspace_bytes": (sz, [i32]),
    "pcd_vae_encode": (i32, [vp, vp, i32, vp, vp, sz, vp]),
    "pcd_vae_decode": (i32, [vp, vp, i32, vp, vp, sz, vp]),
    "pcd_layernorm_f16": (i32, [vp, i64, i32, vp, vp, vp, vp]),
    "pcd_set_attention_workspace_bytes": (sz, [i32, i32, i32]),
    "pcd_set_attention_f16": (i32, [vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "pcd_set_attention_config": (i32, [i32]),
    "pcd_set_attention_last_kernel": (C.c_char_p, []),
    "pcd_add_shape_bias_f16": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "pcd_add_shape_bias_strided_f16": (i32, [vp, i64, i32, i32, vp, i64, vp, vp]),
    "pcd_tail3": (i32, [vp, i32, vp, i32, i64, vp, vp, vp, vp, vp, vp]),
    "pcd_sab_workspace_bytes": (sz, [i64, i32]),
    "pcd_sab_forward": (i32, [C.POINTER(SabDesc), vp, i32, i32, i32, vp, vp, sz, vp]),
    "pcd_pw_wide_ln_linear_packed_bytes": (sz, [i32]),
    "pcd_pw_wide_ln_linear_pack": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "pcd_pw_wide_ln_linear_supported": (i32, [i32, i64]),
    "pcd_pw_wide_ln_linear": (i32, [vp, i32, i32, vp, i64, vp, vp]),
    "pcd_wide_ffn_packed_bytes": (sz, []),
    "pcd_wide_ffn_supported": (i32, [i32, i64]),
    "pcd_wide_ffn_pack": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "pcd_wide_ffn_f16": (i32, [vp, vp, i64, vp, vp]),
    "pcd_wide_ffn_bias_f16": (i32, [vp, vp, i64, i32, vp, i64, vp, vp]),
    "pcd_wide_ffn_config": (i32, [i32]),
    "pcd_pw_wide_config": (i32, [i32]),
    "pcd_sab_tail_packed_bytes": (sz, [i32]),
    "pcd_sab_tail_supported": (i32, [i32, i64]),
    "pcd_sab_tail_pack": (i32, [C.POINTER(SabDesc), vp, vp]),
    "pcd_sab_tail_f16": (i32, [i32, vp, vp, vp, i64, vp, vp]),
    "pcd_sab_head_f16": (i32, [i32, vp, vp, i64, vp, vp]),
    "pcd_sab_tail_bias_f16": (i32, [i32, vp, vp, vp, i64, i32, vp, vp, i64, vp, vp]),
    "pcd_sab_head_bias_f16": (i32, [i32, vp, vp, i64, i32, vp, i64, vp, vp]),
    "pcd_sab_tail_config": (i32, [i32]),
    "pcd_sab_tail_enabled": (i32, []),
    "pcd_attn_unet_create": (i32, [C.POINTER(AttnUnetDesc), C.POINTER(vp)]),
    "pcd_attn_unet_destroy": (None, [vp]),
    "pcd_attn_unet_workspace_bytes": (sz, [i32, i32]),
    "pcd_attn_unet_time_bias": (i32, [vp, vp, i32, vp, vp, vp]),
    "pcd_attn_unet_forward": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_attn_unet_tap": (i32, [vp, C.c_char_p, i32, i32, vp, vp, sz, vp]),
    "pcd_sab_f32_workspace_bytes": (sz, [i64, i32]),
    "pcd_sab_f32_forward": (i32, [C.POINTER(SabDesc), vp, i32, i32, i32, vp, vp, sz, vp]),
    "pcd_attn_unet_f32_create": (i32, [C.POINTER(AttnUnetDesc), C.POINTER(vp)]),
    "pcd_attn_unet_f32_destroy": (None, [vp]),
    "pcd_attn_unet_f32_workspace_bytes": (sz, [i32, i32]),
    "pcd_attn_unet_f32_forward": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, sz, vp]),
    "pcd_attn_unet_f32_tap": (i32, [vp, C.c_char_p, i32, i32, vp, vp, sz, vp]),
    "pcd_normalize_to_cube": (i32, [vp, i32, i32, vp, vp]),
    "pcd_chamfer_sums": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "pcd_voxelize": (i32, [vp, i32, i32, i32, vp, vp]),
    "pcd_voxels_to_points": (i32, [vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "pcd_binary_bce_mean": (i32, [vp, vp, i64, vp, vp]),
    "pcd_pairwise_max_dist": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "pcd_sinkhorn_dual_update": (i32, [vp, vp, i32, i32, i32, vp, f32, f32, vp, vp, vp, vp]),
    "pcd_sinkhorn_cost": (i32, [vp, vp, i32, i32, i32, vp, f32, vp, vp, vp, vp, vp]),
    "pcd_pair_metrics_workspace_bytes": (sz, [i32, i32, i32]),
    "pcd_pair_metrics": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp, sz, vp]),
    "pcd_colsum_f16": (i32, [vp, i64, i32, i32, vp, vp]),
    "pcd_bn_batch_stats": (i32, [vp, i64, i32, f32, vp, vp, vp, vp, vp, vp]),
    "pcd_bn_apply_f16": (i32, [vp, i64, i32, vp, vp, vp, vp, f32, i32, vp, vp]),
    "pcd_bn_backward_f16": (i32, [vp, vp, i64, i32, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp]),
    "pcd_transpose_f16": (i32, [vp, i64, i32, vp, vp]),
    "pcd_colmax_argmax_f16": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "pcd_maxpool_backward_f16": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "pcd_enc1_linear": (i32, [vp, i64, i32, vp, i32, vp, vp, vp]),
    "pcd_vec3_outer": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "pcd_vec3_expand_f16": (i32, [vp, vp, i64, i32, vp, vp]),
    "pcd_l1_loss": (i32, [vp, vp, i64, f32, vp, vp, vp]),
    "pcd_matmul_f32": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, vp, i32, vp, i64, vp]),
    "pcd_silu_f32": (i32, [vp, i64, vp, vp]),
    "pcd_silu_backward_f32": (i32, [vp, vp, i64, vp, vp]),
    "pcd_groupnorm_f32": (i32, [vp, i32, i32, i32, vp, vp, f32, i32, vp, vp, vp, vp]),
    "pcd_groupnorm_backward_f32": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "pcd_mask_scale_f32": (i32, [vp, vp, f32, i64, vp, vp]),
    "pcd_relu_f32": (i32, [vp, i64, vp, vp]),
    "pcd_relu_backward_f32": (i32, [vp, vp, i64, vp, vp]),
    "pcd_im2col_f16": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "pcd_col2im_f16": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "pcd_bias_act_f16": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "pcd_add_relu_f16": (i32, [vp, vp, i64, i32, vp, vp]),
    "pcd_relu_mask_f16": (i32, [vp, vp, i64, vp, vp]),
    "pcd_sigmoid_bce": (i32, [vp, i64, vp, i64, f32, vp, vp, vp, vp]),
    "pcd_vae_latent_backward": (i32, [vp, vp, vp, vp, i64, f32, vp, vp, vp, vp]),
    "pcd_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
}

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libpcd_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout)
    return LIB_PATH


def load() -> C.CDLL:
    """Load the library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
            "Run `python __graft_entry__.py build`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    got = lib.pcd_abi_version()
    if got != ABI_VERSION:       # a stale libpcd_hip.so next to newer bindings (or the reverse): argument lists / struct layouts differ
        raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, these bindings are written against {ABI_VERSION} "
                           "(include/pcd_hip.h PCD_ABI_VERSION): rebuild with `python __graft_entry__.py build`")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().pcd_last_error()
        raise RuntimeError(f"pcd_hip {what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("an MI355X (gfx950) device is required: this framework has no CPU path")
    check(load().pcd_device_check(), "device_check")
