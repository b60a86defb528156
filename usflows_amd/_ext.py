"""ctypes binding of ``libusflows_hip.so`` (C ABI in ``include/usflows_hip.h``).

The library is built in-tree by ``usflows_amd/csrc/Makefile`` (``__graft_entry__.build()``)
for gfx950.  There is NO fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised -- the product path never silently degrades to eager PyTorch.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct
import threading
from typing import Optional

import torch  # noqa: F401  -- must be imported first: the .so binds to torch's HIP runtime instance

_HERE = os.path.dirname(os.path.abspath(__file__))
from .config import config  # noqa: E402

LIB_PATH = config.lib_path     # (USFLOWS_AMD_LIB: A/B builds)

USF_ABI_VERSION = 35
USF_MAX_HIDDEN = 4

ACT_NONE, ACT_LEAKY_RELU, ACT_GATE = 0, 1, 2
BASE_LAPLACE, BASE_NORMAL, BASE_LPNORM1, BASE_LPNORM2, BASE_LPNORMINF, BASE_ROWSUM = 0, 1, 2, 3, 4, 5
NORM_LOGNORMAL, NORM_GAMMA, NORM_RAW_PARAMS = 0, 1, 0x100
RADIAL_MAX_K = 64
OP_LINEAR, OP_COUPLING, OP_PACK_PLANES, OP_GEMM_PLANES, OP_COUPLING_PLANES, OP_GATED_NORM, OP_CALL = 1, 2, 5, 6, 7, 9, 10

_fp = C.c_void_p  # device pointers travel as integers


class LinearDesc(C.Structure):
    _fields_ = [
        ("A", _fp), ("lda", C.c_int64),
        ("W", _fp), ("ldw", C.c_int64),
        ("bias", _fp), ("pre_div", _fp), ("pre_sub", _fp),
        ("residual", _fp), ("ldr", C.c_int64),
        ("addend", _fp), ("ldadd", C.c_int64),
        ("post_mul", _fp),
        ("C", _fp), ("ldc", C.c_int64),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("res_sign", C.c_float), ("slope", C.c_float),
        ("act", C.c_int32), ("reserved", C.c_int32),
        ("W_split", _fp), ("ldw_split", C.c_int64), ("split_plane_stride", C.c_int64),
        ("A_planes_out", _fp), ("ldp_out", C.c_int64), ("planes_out_stride", C.c_int64),
    ]


class CouplingDesc(C.Structure):
    _fields_ = [
        ("z", _fp), ("ldz", C.c_int64),
        ("out", _fp), ("ldo", C.c_int64),
        ("M", C.c_int64),
        ("off_pass", C.c_int64), ("n_pass", C.c_int64),
        ("off_trans", C.c_int64), ("n_trans", C.c_int64),
        ("n_hidden", C.c_int32), ("hidden", C.c_int32 * USF_MAX_HIDDEN),
        ("W_in", _fp), ("ldw_in", C.c_int64), ("b_in", _fp),
        ("W_hid", _fp * USF_MAX_HIDDEN), ("b_hid", _fp * USF_MAX_HIDDEN), ("ldw_hid", C.c_int64 * USF_MAX_HIDDEN),
        ("W_out", _fp), ("ldw_out", C.c_int64), ("b_out", _fp),
        ("context", _fp), ("W_ctx", _fp), ("b_ctx", _fp),
        ("post_sub", _fp),
        ("sign", C.c_float), ("slope", C.c_float),
        ("act", C.c_int32), ("reserved", C.c_int32),
        ("split_in", _fp), ("split_in_ld", C.c_int64), ("split_in_plane", C.c_int64),
        ("split_hid", _fp * USF_MAX_HIDDEN), ("split_hid_ld", C.c_int64), ("split_hid_plane", C.c_int64),
        ("split_out", _fp), ("split_out_ld", C.c_int64), ("split_out_plane", C.c_int64),
        ("hidden_out", _fp * USF_MAX_HIDDEN), ("ld_hidden_out", C.c_int64),
        ("gate", _fp * USF_MAX_HIDDEN), ("ld_gate", C.c_int64),
    ]


class PackPlanesDesc(C.Structure):
    _fields_ = [("src", _fp), ("ld", C.c_int64), ("M", C.c_int64), ("nkb", C.c_int64), ("idx", _fp),
                ("pre_div", _fp), ("pre_sub", _fp), ("planes", _fp), ("format", C.c_int32), ("reserved", C.c_int32), ("range_flag", _fp),
                ("src_cols", C.c_int64), ("row_weight", _fp), ("loc", _fp), ("scale", _fp), ("grad_base", C.c_int32),
                ("reserved2", C.c_int32)]


class GemmPlanesDesc(C.Structure):
    _fields_ = [("A", _fp), ("a_nkb", C.c_int64), ("a_kb0", C.c_int64), ("nk", C.c_int64),
                ("W_planes", _fp), ("ldw", C.c_int64), ("w_plane_stride", C.c_int64), ("w_rows", C.c_int64),
                ("bias", _fp), ("post_mul", _fp), ("residual", _fp),
                ("C_planes", _fp), ("c_nkb", C.c_int64), ("c_kb0", C.c_int64), ("c_kbn", C.c_int64),
                ("C_f32", _fp), ("ldc", C.c_int64), ("N", C.c_int64), ("M", C.c_int64),
                ("res_sign", C.c_float), ("slope", C.c_float), ("act", C.c_int32), ("format", C.c_int32), ("range_flag", _fp),
                ("base_tab", _fp), ("base_tab_stride", C.c_int64), ("base_part", _fp), ("base", C.c_int32), ("reserved", C.c_int32)]


class CouplingPlanesDesc(C.Structure):
    _fields_ = [("z", _fp), ("z_nkb", C.c_int64), ("M", C.c_int64),
                ("kb_p0", C.c_int64), ("nk_p", C.c_int64), ("kb_t0", C.c_int64), ("nk_t", C.c_int64),
                ("n_hidden", C.c_int32), ("hidden_padded", C.c_int32),
                ("W_in", _fp), ("ldw_in", C.c_int64), ("w_in_plane", C.c_int64), ("b_in", _fp),
                ("W_hid", _fp * 2), ("b_hid", _fp * 2), ("ldw_hid", C.c_int64), ("w_hid_plane", C.c_int64),
                ("W_out", _fp), ("ldw_out", C.c_int64), ("w_out_plane", C.c_int64), ("b_out", _fp),
                ("sign", C.c_float), ("slope", C.c_float), ("act", C.c_int32), ("format", C.c_int32),
                ("range_flag", _fp), ("hidden_out", _fp * 2), ("gate", _fp * 2)]


class MtChunk(C.Structure):
    """usf_mt_chunk: one block's share of one parameter tensor (SophiaG multi-tensor kernels)"""
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("h", _fp), ("n", C.c_int32), ("reserved", C.c_int32)]


class GatedNormDesc(C.Structure):
    """usf_gated_norm_desc: row pass of the vector ConvNet conditioner (gate, layer norm, activation)"""
    _fields_ = [("skip", _fp), ("ld_skip", C.c_int64), ("vg", _fp), ("ld_vg", C.c_int64), ("gate_off", C.c_int64),
                ("gamma", _fp), ("beta", _fp), ("out", _fp), ("ld_out", C.c_int64), ("out_act", _fp), ("ld_act", C.c_int64),
                ("M", C.c_int64), ("C", C.c_int64), ("c_pad", C.c_int64), ("eps", C.c_float), ("slope", C.c_float),
                ("act", C.c_int32), ("reserved", C.c_int32)]


class GatedNormBwdDesc(C.Structure):
    """usf_gated_norm_bwd_desc: the backward twin of the row pass"""
    _fields_ = [("skip", _fp), ("ld_skip", C.c_int64), ("vg", _fp), ("ld_vg", C.c_int64), ("gate_off", C.c_int64),
                ("gamma", _fp), ("dy", _fp), ("ld_dy", C.c_int64), ("d_skip", _fp), ("ld_d_skip", C.c_int64),
                ("d_vg", _fp), ("ld_d_vg", C.c_int64), ("dy_xh", _fp), ("ld_dy_xh", C.c_int64),
                ("M", C.c_int64), ("C", C.c_int64), ("c_pad", C.c_int64), ("eps", C.c_float), ("reserved", C.c_float)]


class CallDesc(C.Structure):
    """usf_call_desc: one entry-point call inside an op list, arguments as 64-bit words"""
    _fields_ = [("fn", C.c_int32), ("n_args", C.c_int32), ("a", C.c_uint64 * 20)]


class _OpUnion(C.Union):
    _fields_ = [("linear", LinearDesc), ("coupling", CouplingDesc), ("pack_planes", PackPlanesDesc),
                ("gemm_planes", GemmPlanesDesc), ("coupling_planes", CouplingPlanesDesc), ("gated_norm", GatedNormDesc),
                ("call", CallDesc)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("u", _OpUnion)]


class LuPrepDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("D", C.c_int64),
        ("L_raw", C.POINTER(C.c_void_p)), ("U_raw", C.POINTER(C.c_void_p)),
        ("tri", _fp), ("tri_inv", _fp), ("work", _fp), ("M", _fp), ("Minv", _fp), ("ladj", _fp),
    ]


class PackJob(C.Structure):
    _fields_ = [
        ("src", _fp), ("out_idx", _fp), ("in_idx", _fp), ("W", _fp), ("planes", _fp),
        ("ld_src", C.c_int64), ("n_out", C.c_int64), ("n_in", C.c_int64), ("ldw", C.c_int64),
        ("ld_planes", C.c_int64), ("plane_stride", C.c_int64),
        ("src_is_f32", C.c_int32), ("transpose", C.c_int32),
    ]


class PsumJob(C.Structure):
    """usf_psum_job: one deferred sum of per-wave partial slots (usf_conv_wgrad_deferred_f32 / usf_partial_sum_jobs_f32)"""
    _fields_ = [("part", _fp), ("out", _fp), ("out2", _fp),
                ("nparts", C.c_int32), ("n", C.c_int32), ("mode", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("CIT", C.c_int32), ("T", C.c_int32), ("ntile", C.c_int32), ("first_block", C.c_int32), ("per", C.c_int32),
                ("rows", C.c_int32), ("vec4", C.c_int32)]


class WgradJob(C.Structure):
    """usf_wgrad_job: one queued weight-gradient launch (usf_conv_wgrad_plan_f32 / usf_conv_wgrad_jobs_f32)"""
    _fields_ = [("args", C.c_ubyte * 192), ("CIT", C.c_int32), ("COT", C.c_int32), ("T", C.c_int32), ("blocks", C.c_int32),
                ("lds_bytes", C.c_int32), ("first_block", C.c_int32)]


class WReduceJob(C.Structure):
    """usf_wreduce_job: one queued reduction of usf_wgrad_blocked_plan_f32 (usf_wgrad_reduce_jobs_f32)"""
    _fields_ = [("part", _fp), ("out", _fp), ("cs_part", _fp), ("cs_out", _fp), ("rows", C.c_int64), ("cols", C.c_int64),
                ("ldo", C.c_int64), ("alpha", C.c_float), ("beta", C.c_float), ("cs_alpha", C.c_float), ("cs_beta", C.c_float),
                ("first_block", C.c_int32), ("blocks", C.c_int32), ("sched", C.c_ubyte * 64)]


class GradJob(C.Structure):
    _fields_ = [
        ("Y", _fp), ("A", _fp), ("G", _fp),
        ("ldy", C.c_int64), ("lda", C.c_int64), ("ldg", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("first_block", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float),
    ]


# every symbol include/usflows_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "usf_abi_version": (C.c_int, []),
    "usf_set_tuning": (C.c_int, [C.c_char_p, C.c_int64]),
    "usf_get_tuning": (C.c_int64, [C.c_char_p, C.c_int64]),
    "usf_sizeof_desc": (C.c_int, [C.c_int32]),
    "usf_last_error": (C.c_char_p, []),
    "usf_build_info": (C.c_char_p, []),
    "usf_linear_f32": (C.c_int, [C.POINTER(LinearDesc), C.c_void_p]),
    "usf_linear_variant": (C.c_int, [C.POINTER(LinearDesc)]),
    "usf_pack_planes_f32": (C.c_int, [C.POINTER(PackPlanesDesc), C.c_void_p]),
    "usf_gemm_planes_bf16x3": (C.c_int, [C.POINTER(GemmPlanesDesc), C.c_void_p]),
    "usf_gemm_planes_variant": (C.c_int, [C.POINTER(GemmPlanesDesc)]),
    "usf_coupling_planes": (C.c_int, [C.POINTER(CouplingPlanesDesc), C.c_void_p]),
    "usf_coupling_additive_f32": (C.c_int, [C.POINTER(CouplingDesc), C.c_void_p]),
    "usf_coupling_variant": (C.c_int, [C.POINTER(CouplingDesc)]),
    "usf_coupling_max_width": (C.c_int, []),
    "usf_coupling_padded_width": (C.c_int, [C.c_int]),
    "usf_base_logprob_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, C.c_float,
                                       _fp, _fp, _fp, C.c_void_p]),
    "usf_base_tables_f32": (C.c_int, [C.c_int32, _fp, _fp, C.c_int64, _fp, C.c_int64, C.c_void_p]),
    "usf_base_sample_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, C.c_uint64,
                                      C.c_uint64, C.c_int64, C.c_void_p]),
    "usf_radial_sample_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, C.c_uint64,
                                        C.c_uint64, C.c_int64, C.c_void_p]),
    "usf_radial_logprob_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _fp, C.c_int32, C.c_int32, _fp, _fp, _fp,
                                         C.c_double, C.c_float, _fp, _fp, _fp, _fp, C.c_void_p]),
    "usf_radial_logprob_grad_workspace": (C.c_int64, [C.c_int64, C.c_int64]),
    "usf_radial_logprob_grad_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, _fp, C.c_int32, C.c_int32,
                                              _fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_void_p]),
    "usf_variates_from_bits_f32": (C.c_int, [_fp, C.c_int64, _fp, _fp, _fp, C.c_void_p]),
    "usf_scale_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int32, C.c_void_p]),
    "usf_affine_coupling_apply_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64,
                                                C.c_float, C.c_int32, _fp, C.c_void_p]),
    "usf_channel_affine_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp, C.c_void_p]),
    "usf_layernorm_channels_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, C.c_float, C.c_int32,
                                             C.c_float, C.c_void_p]),
    "usf_gated_residual_f32": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_gated_norm_rows_f32": (C.c_int, [C.POINTER(GatedNormDesc), C.c_void_p]),
    "usf_gated_norm_rows_bwd_f32": (C.c_int, [C.POINTER(GatedNormBwdDesc), C.c_void_p]),
    "usf_pointwise_conv_supported": (C.c_int, [C.c_int64, C.c_int64, C.c_int32]),
    "usf_pointwise_conv_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, C.c_int32, C.c_float,
                                         C.c_int32, C.c_float, _fp, _fp, _fp, C.c_float, C.c_void_p]),
    "usf_conv2d_weight_elems": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "usf_conv2d_same_fits": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "usf_conv2d_same_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp,
                                      C.c_int32, C.c_float, C.c_int32, C.c_float, _fp, C.c_int64, C.c_void_p]),
    "usf_conv2d_same_res_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp,
                                          C.c_int32, C.c_float, _fp, _fp, C.c_float, C.c_void_p]),
    "usf_masked_residual_f32": (C.c_int, [_fp, _fp, _fp, C.c_float, _fp, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_conv2d_same_gate_f32": (C.c_int, [_fp, _fp] + [C.c_int64] * 6 + [C.c_void_p, _fp, C.c_float, _fp, _fp, C.c_void_p]),
    "usf_conv_wgrad_workspace": (C.c_int64, [C.c_int64] * 6),
    "usf_conv_wgrad_f32": (C.c_int, [_fp, _fp] + [C.c_int64] * 6 + [_fp, _fp, C.c_int32, C.c_float, _fp, _fp, _fp, C.c_int64,
                                     C.c_void_p]),
    "usf_conv_wgrad_deferred_f32": (C.c_int, [_fp, _fp] + [C.c_int64] * 6 + [_fp, _fp, C.c_int32, C.c_float, _fp, _fp, _fp, C.c_int64,
                                              C.POINTER(PsumJob), C.c_void_p]),
    "usf_conv_wgrad_plan_f32": (C.c_int, [_fp, _fp] + [C.c_int64] * 6 + [_fp, _fp, C.c_int32, C.c_float, _fp, _fp, _fp, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "usf_conv_wgrad_jobs_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),      # (job: two entries)
    "usf_partial_sum_jobs_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_void_p]),
    "usf_layernorm_channels_bwd_workspace": (C.c_int64, [C.c_int64] * 3),
    "usf_layernorm_channels_bwd_f32": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_float, C.c_int32, C.c_float,
                                                 _fp, _fp, C.c_int64, C.c_void_p]),
    "usf_gated_residual_bwd_f32": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_conv2d_weight_planes_batch_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "usf_gated_tail_supported": (C.c_int, [C.c_int64]),
    "usf_gated_tail_workspace": (C.c_int64, [C.c_int64] * 3),
    "usf_gated_tail_f32": (C.c_int, [_fp, _fp, _fp] + [C.c_int64] * 3 + [_fp, _fp, C.c_int32, C.c_float, C.c_int32, C.c_float, _fp, _fp,
                                     C.c_float, C.c_void_p]),
    "usf_gated_tail_bwd_f32": (C.c_int, [_fp] * 6 + [C.c_int64] * 3 + [_fp, _fp, C.c_int32, C.c_float, C.c_int32, C.c_float, _fp, _fp,
                                         C.c_float, _fp, _fp, C.c_int64, C.c_void_p, C.c_void_p]),
    "usf_conv2d_weight_planes_f32": (C.c_int, [_fp, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "usf_gather_cols_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_void_p]),
    "usf_run_ops": (C.c_int, [C.POINTER(Op), C.c_int32, C.c_void_p]),
    "usf_lu_prepare_f64": (C.c_int, [C.POINTER(LuPrepDesc), C.c_void_p]),
    "usf_gemm_f64": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, _fp, C.c_int64, C.c_int64, C.c_int32,
                               _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                               C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    "usf_householder_f64": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, _fp, C.c_void_p]),
    "usf_lu_grad_finish_f64": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, _fp, _fp, C.c_void_p]),
    "usf_pack_weight_f32": (C.c_int, [_fp, C.c_int32, C.c_int64, C.c_int32, _fp, C.c_int64, _fp, C.c_int64,
                                      _fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_wgrad_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int64,
                                C.c_float, C.c_float, C.c_int32, _fp, C.c_int64, C.c_void_p]),
    "usf_wgrad_workspace_floats": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "usf_wgrad_bias_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int64,
                                     C.c_float, C.c_float, C.c_int32, _fp, C.c_float, C.c_float, _fp, C.c_int64, _fp]),
    "usf_wgrad_bias_ok": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "usf_wgrad_planes_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int64, C.c_int64, _fp, C.c_int64, C.c_float, C.c_float, _fp, C.c_float, C.c_float,
                                       _fp, C.c_int64, _fp]),
    "usf_wgrad_blocked_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp,
                                        C.c_int64, C.c_float, C.c_float, _fp, C.c_float, C.c_float, _fp, C.c_int64, _fp]),
    "usf_wgrad_blocked_plan_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp,
                                             C.c_int64, C.c_float, C.c_float, _fp, C.c_float, C.c_float, _fp, C.c_int64, C.c_void_p, _fp]),
    "usf_wgrad_reduce_jobs_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "usf_base_param_grad_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, _fp, _fp, C.c_int64, C.c_void_p]),
    "usf_mfma_probe": (C.c_int, [_fp, _fp, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_void_p]),
    "usf_set_clock_buffer": (C.c_int, [_fp]),
    "usf_wgrad_planes_colsum_ok": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "usf_wgrad_planes_workspace_floats": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "usf_wgrad_planes_ok": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "usf_split_planes_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_int64, C.c_int64, _fp]),
    "usf_wgrad_variant": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "usf_sophiag_step_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                       C.c_void_p]),
    "usf_sophiag_hessian_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "usf_colsum_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, _fp, C.c_float, C.c_float, _fp, C.c_int64,
                                 C.c_void_p]),
    "usf_act_grad_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "usf_base_logprob_grad_f32": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, _fp,
                                            C.c_int64, C.c_void_p]),
    "usf_pack_weights_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_pack_weights_t_f32": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "usf_grad_jobs_f32": (C.c_int, [_fp, _fp, C.c_int64, C.c_void_p]),
    "usf_affine_prep_f32": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp,
                                      C.c_void_p]),
    "usf_affine_prep_bwd_f32": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int32,
                                          C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p]),
    "usf_matvec_f64": (C.c_int, [_fp, C.c_int64, C.c_int64, _fp, C.c_int64, _fp, C.c_double, _fp, _fp, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def lib_exists() -> bool:
    return os.path.isfile(LIB_PATH)


def load() -> C.CDLL:
    """Load the HIP library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not lib_exists():
        raise RuntimeError(
            f"usflows_amd: HIP extension not built ({LIB_PATH} missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C usflows_amd/csrc`. "
            "There is no CPU/eager fallback for the device path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.usf_abi_version() != USF_ABI_VERSION:
        raise RuntimeError(f"usflows_amd: ABI mismatch: library {lib.usf_abi_version()} != binding {USF_ABI_VERSION}")
    for kind, st in ((OP_LINEAR, LinearDesc), (OP_COUPLING, CouplingDesc), (0, Op), (3, LuPrepDesc), (4, PackJob),
                     (OP_PACK_PLANES, PackPlanesDesc), (OP_GEMM_PLANES, GemmPlanesDesc),
                     (OP_COUPLING_PLANES, CouplingPlanesDesc), (8, MtChunk), (OP_GATED_NORM, GatedNormDesc), (OP_CALL, CallDesc),
                     (11, GradJob), (12, PsumJob)):
        if lib.usf_sizeof_desc(kind) != C.sizeof(st):
            raise RuntimeError(f"usflows_amd: struct layout mismatch for {st.__name__}: "
                               f"C {lib.usf_sizeof_desc(kind)} vs ctypes {C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    cl = getattr(_tls, "calls", None)
    if cl is not None and what.startswith("usf_") and what not in CALL_FNS:
        # a library call with no USF_OP_CALL form ran inside a pass that is being recorded as an op list: a replay of the
        # list would skip it -- the pass is not replayable (Flow._layer_loop_listed keeps the eager loop)
        cl.bad = cl.bad or what
    if rc != 0:
        msg = load().usf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"usflows_amd HIP call failed ({what}, rc={rc}): {msg}")


# ---- launch tapes ---------------------------------------------------------------------------------
# A training step re-issues the SAME ~2000 launches with the same device pointers every iteration (persistent
# buffers, parameters updated in place).  A Tape records each C call once -- (function, fully converted ctypes
# arguments) -- and `replay` re-issues them without going through the Python wrappers again (~3 us per launch
# instead of ~40 us).  `host_op` records parameter-sized torch work that has to be redone on replay.
TAPES_ENABLED = True


class Tape:
    __slots__ = ("entries", "stream", "dx")

    def __init__(self):
        self.entries = []
        self.stream = None
        self.dx = None          # training.TrainPath: (buffer, row stride) the recorded backward leaves the input's gradient in


class _ThreadState(threading.local):
    """per-thread recording / batching state: two threads driving flows at once must not interleave their tapes"""

    def __init__(self):
        self.rec = []        # recording stack; the innermost tape receives the launches, None = recording suspended
        self.jobs = []       # stack of open job batches, innermost receives


_tls = _ThreadState()


class record:
    def __init__(self, tape: Optional[Tape]):
        self.tape = tape

    def __enter__(self):
        _tls.rec.append(self.tape)
        return self.tape

    def __exit__(self, *exc):
        _tls.rec.pop()
        return False


# bench.py --mode train: {entry-point name: [(start_event, end_event, args), ...]} -- every launch (and replay) of the
# named entry points is bracketed by HIP events on torch's current stream.  None (the default): no events, no cost.
launch_timing = None


def _timed_call(fn, args, name):
    lt = launch_timing
    if lt is not None and name in lt:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        lt[name].append((e0, e1, args))
        return rc
    return fn(*args)


# ---- a layer loop's calls as ONE op list (USF_OP_CALL; flows.py: image-shaped flows) ------------------------------------
CALL_FNS = {"usf_scale_f32": 1, "usf_channel_affine_f32": 2, "usf_layernorm_channels_f32": 3, "usf_gated_residual_f32": 4,
            "usf_masked_residual_f32": 5, "usf_pointwise_conv_f32": 6, "usf_conv2d_same_f32": 7, "usf_conv2d_same_res_f32": 8,
            "usf_base_logprob_f32": 9, "usf_radial_logprob_f32": 10, "usf_gated_tail_f32": 11}


class CallList:
    """the entry-point calls of one pass, recorded while they run (``with recording_calls(cl)``): (function id, argument
    words, which of them are pointers).  ``bad`` names the first call that has no USF_OP_CALL form."""

    def __init__(self):
        self.calls = []
        self.bad = None

    def add(self, name, args):
        fid = CALL_FNS.get(name)
        if fid is None:
            self.bad = self.bad or name
            return
        types = SYMBOLS[name][1][:-1]                       # (the stream is not part of the record)
        words, is_ptr = [], []
        for v, t in zip(args[:-1], types):
            if t is C.c_float:
                words.append(struct.unpack("<I", struct.pack("<f", float(v)))[0])
                is_ptr.append(False)
            elif t is C.c_double:
                words.append(struct.unpack("<Q", struct.pack("<d", float(v)))[0])
                is_ptr.append(False)
            elif t in (C.c_void_p, _fp):
                words.append(int(v) if v else 0)
                is_ptr.append(True)
            else:
                words.append(int(v) & 0xFFFFFFFFFFFFFFFF)
                is_ptr.append(False)
        self.calls.append((fid, words, is_ptr))

    def ops(self):
        """the list as a ctypes array of usf_op (kind USF_OP_CALL)"""
        arr = (Op * max(1, len(self.calls)))()
        for i, (fid, words, _) in enumerate(self.calls):
            arr[i].kind = OP_CALL
            arr[i].u.call.fn = fid
            arr[i].u.call.n_args = len(words)
            for j, w in enumerate(words):
                arr[i].u.call.a[j] = w
        return arr


class recording_calls:
    def __init__(self, cl):
        self.cl, self.prev = cl, None

    def __enter__(self):
        self.prev, _tls.calls = getattr(_tls, "calls", None), self.cl
        return self.cl

    def __exit__(self, *exc):
        _tls.calls = self.prev
        return False


def _launch(name: str, args: tuple, keep=None) -> None:
    fn = getattr(load(), name)
    rec = _tls.rec
    if rec and rec[-1] is not None:
        rec[-1].entries.append((fn, args, name, keep))
    cl = getattr(_tls, "calls", None)
    if cl is not None:
        cl.add(name, args)
    rc = _timed_call(fn, args, name)
    if rc != 0:
        check(rc, name)


def _direct(name: str, *args) -> None:
    """call entry point `name` now (never taped); bracketed by HIP events when launch_timing asks for this name"""
    cl = getattr(_tls, "calls", None)
    if cl is not None:
        cl.add(name, args)
    check(_timed_call(getattr(load(), name), args, name), name)


def host_op(fn) -> None:
    """run fn() now and, when recording, again on every replay (its own launches are not taped separately)"""
    rec = _tls.rec
    if rec and rec[-1] is not None:
        rec[-1].entries.append(fn)
    with record(None):
        fn()


def replay(tape: Tape) -> None:
    with record(None):
        for e in tape.entries:
            if e.__class__ is tuple:
                rc = e[0](*e[1]) if launch_timing is None else _timed_call(e[0], e[1], e[2])
                if rc != 0:
                    check(rc, e[2])
            else:
                e()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---- thin typed wrappers (each enqueues on torch's current stream) ---------------------------
def linear(A, W, C_out, *, M, N, K, lda, ldw, ldc, bias=None, pre_div=None, pre_sub=None, residual=None,
           ldr=0, post_mul=None, res_sign=1.0, act=ACT_NONE, slope=0.0, a_off=0, c_off=0, r_off=0,
           addend=None, ldadd=0, W_split=None, planes_out=None):
    """usf_linear_f32 on raw tensors; *_off are element offsets into A/C/residual.  planes_out: a ``row_planes`` buffer
    [3, ceil32(M), ld] that receives the bf16 planes of the input A (usf_linear_desc::A_planes_out)."""
    d = LinearDesc()
    d.A = A.data_ptr() + 4 * a_off
    d.lda = lda
    d.W = W.data_ptr()
    d.ldw = ldw
    d.bias = ptr(bias)
    d.pre_div = ptr(pre_div)
    d.pre_sub = ptr(pre_sub)
    d.residual = None if residual is None else residual.data_ptr() + 4 * r_off
    d.ldr = ldr
    d.post_mul = ptr(post_mul)
    d.addend = ptr(addend)
    d.ldadd = ldadd
    if W_split is not None:            # [3, N, ceil32(K)] bf16 planes (see include/usflows_hip.h)
        d.W_split = W_split.data_ptr()
        d.ldw_split = W_split.shape[2]
        d.split_plane_stride = W_split.shape[1] * W_split.shape[2]
    d.C = C_out.data_ptr() + 4 * c_off
    d.ldc = ldc
    d.M, d.N, d.K = M, N, K
    d.res_sign, d.slope, d.act = res_sign, slope, act
    if planes_out is not None:
        d.A_planes_out, d.ldp_out, d.planes_out_stride = planes_out.data_ptr(), planes_out.shape[2], planes_out.shape[1] * planes_out.shape[2]
    _launch("usf_linear_f32", (C.byref(d), current_stream(A.device)), d)


def planes_bytes(M: int, nkb: int, fmt: int = 0) -> int:
    """size of a planes buffer of M rows and nkb 32-feature blocks (include/usflows_hip.h)"""
    return (-(-M // 16)) * nkb * (2048 if fmt == 1 else 3072)


PLANES_BF16X3, PLANES_F16X2 = 0, 1


def pack_planes(src, planes, *, M, nkb, idx, ld=None, pre_div=None, pre_sub=None, fmt=PLANES_BF16X3, range_flag=None, src_cols=0,
                grad=None):
    """usf_pack_planes_f32.  src_cols > 0: a promise that every idx[l] < src_cols (rows are then read coalesced and gathered out
    of LDS).  grad = (base id, row_weight [M], loc [src_cols], scale [src_cols]): the planes receive the gradient of the base
    density at src = z, weighted per row (the head of the training backward pass; Laplace / Normal)."""
    d = PackPlanesDesc()
    d.format, d.range_flag = fmt, ptr(range_flag)
    d.src, d.ld, d.M, d.nkb = src.data_ptr(), (src.stride(0) if ld is None else ld), M, nkb
    d.idx, d.pre_div, d.pre_sub, d.planes = idx.data_ptr(), ptr(pre_div), ptr(pre_sub), planes.data_ptr()
    d.src_cols = src_cols
    keep = (d, src, planes, idx, pre_div, pre_sub)
    if grad is not None:
        base, w, loc, scale = grad
        d.grad_base, d.row_weight, d.loc, d.scale = 1 + base, w.data_ptr(), loc.data_ptr(), scale.data_ptr()
        keep = keep + (w, loc, scale)
    _launch("usf_pack_planes_f32", (C.byref(d), current_stream(src.device)), keep)


def gemm_planes(A, W_planes, *, M, a_nkb, nk, a_kb0=0, bias=None, post_mul=None, residual=None, C_planes=None, c_nkb=0,
                c_kb0=0, c_kbn=0, C_f32=None, ldc=0, N=0, res_sign=1.0, act=ACT_NONE, slope=0.0, fmt=PLANES_BF16X3,
                range_flag=None):
    """usf_gemm_planes_bf16x3; W_planes: [3, w_rows, ldw] bf16 or [2, w_rows, ldw] fp16 (K axis in slot order)"""
    d = GemmPlanesDesc()
    d.format, d.range_flag = fmt, ptr(range_flag)
    d.A, d.a_nkb, d.a_kb0, d.nk = A.data_ptr(), a_nkb, a_kb0, nk
    d.W_planes, d.ldw, d.w_plane_stride, d.w_rows = (W_planes.data_ptr(), W_planes.shape[2],
                                                      W_planes.shape[1] * W_planes.shape[2], W_planes.shape[1])
    d.bias, d.post_mul, d.residual = ptr(bias), ptr(post_mul), ptr(residual)
    d.C_planes, d.c_nkb, d.c_kb0, d.c_kbn = ptr(C_planes), c_nkb, c_kb0, c_kbn
    d.C_f32, d.ldc, d.N, d.M = ptr(C_f32), ldc, N, M
    d.res_sign, d.slope, d.act = res_sign, slope, act
    _launch("usf_gemm_planes_bf16x3", (C.byref(d), current_stream(A.device)),
            (d, A, W_planes, bias, post_mul, residual, C_planes, C_f32))


def base_logprob(z, ldz, M, D, base, loc, scale, logdet_const, out, sum_out=None, logdet_dev=None):
    """logdet_dev: optional fp64 device scalar added to logdet_const by the kernel (no host read-back of the constant)"""
    _launch("usf_base_logprob_f32", (z.data_ptr(), ldz, M, D, base, ptr(loc), ptr(scale), float(logdet_const),
                                     ptr(logdet_dev), out.data_ptr(), ptr(sum_out), current_stream(z.device)),
            keep=logdet_dev)


def base_tables(base, loc, scale, D, tab, stride):
    """loc | 1 / scale | constant of a Laplace / Normal base, [3, stride] fp32: the tables of the last GEMM's fused tail"""
    _launch("usf_base_tables_f32", (base, loc.data_ptr(), scale.data_ptr(), D, tab.data_ptr(), stride, current_stream(tab.device)))


def radial_logprob(z, ldz, M, D, p_id, loc, norm, K, par_a, par_b, logits, logdv_const, logdet_const, out, r_out=None, sum_out=None,
                   logdet_dev=None):
    """usf_radial_logprob_f32: RadialDistribution.log_prob of the rows of z in one launch (include/usflows_hip.h); z None:
    the radii are given in r_out"""
    _launch("usf_radial_logprob_f32", (ptr(z), ldz, M, D, int(p_id), ptr(loc), int(norm), int(K), par_a.data_ptr(),
                                       par_b.data_ptr(), ptr(logits), float(logdv_const), float(logdet_const), ptr(logdet_dev),
                                       out.data_ptr(), ptr(r_out), ptr(sum_out), current_stream(out.device)),
            keep=(z, loc, par_a, par_b, logits, logdet_dev, out, r_out, sum_out))


def radial_logprob_grad(z, ldz, r, g_lp, M, D, p_id, loc, norm, K, par_a, par_b, logits, g, ldg, d_loc=None, d_a=None, d_b=None,
                        d_logits=None):
    """usf_radial_logprob_grad_f32 (the workspace is allocated here and kept alive by the tape / the caller's stream order)"""
    lib = load()
    ws_n = max(8, lib.usf_radial_logprob_grad_workspace(max(M, 1), D))
    ws = torch.empty((ws_n + 7) // 8, dtype=torch.float64, device=r.device)
    _launch("usf_radial_logprob_grad_f32", (ptr(z), ldz, r.data_ptr(), g_lp.data_ptr(), M, D, int(p_id), ptr(loc),
                                            int(norm), int(K), par_a.data_ptr(), par_b.data_ptr(), ptr(logits), g.data_ptr(), ldg,
                                            ptr(d_loc), ptr(d_a), ptr(d_b), ptr(d_logits), ws.data_ptr(), ws.numel() * 8,
                                            current_stream(r.device)),
            keep=(z, r, g_lp, loc, par_a, par_b, logits, g, d_loc, d_a, d_b, d_logits, ws))


def base_sample(z, ldz, M, D, base, loc, scale, seed, offset, row_offset=0):
    check(load().usf_base_sample_f32(z.data_ptr(), ldz, M, D, base, ptr(loc), ptr(scale), seed, offset,
                                     row_offset, current_stream(z.device)), "usf_base_sample_f32")


def radial_sample(z, ldz, M, D, base, loc, r, seed, offset, row_offset=0):
    check(load().usf_radial_sample_f32(z.data_ptr(), ldz, M, D, base, loc.data_ptr(), r.data_ptr(), seed, offset,
                                       row_offset, current_stream(z.device)), "usf_radial_sample_f32")


def variates_from_bits(bits, u=None, laplace=None, exponential=None):
    """the head kernels' word -> variate maps on caller-supplied int32/uint32 words (include/usflows_hip.h)"""
    check(load().usf_variates_from_bits_f32(bits.data_ptr(), bits.numel(), ptr(u), ptr(laplace), ptr(exponential),
                                            current_stream(bits.device)), "usf_variates_from_bits_f32")


def scale(x, ldx, y, ldy, M, D, s, divide):
    _direct("usf_scale_f32", x.data_ptr(), ldx, y.data_ptr(), ldy, M, D, s.data_ptr(), int(divide), current_stream(x.device))


def affine_coupling_apply(z, ldz, t, ldt, s, lds, M, n, bound, inverse, logdet=None, z_off=0, t_off=0, s_off=0):
    """usf_affine_coupling_apply_f32 (element offsets *_off into the fp32 tensors)"""
    check(load().usf_affine_coupling_apply_f32(z.data_ptr() + 4 * z_off, ldz, t.data_ptr() + 4 * t_off, ldt,
                                               s.data_ptr() + 4 * s_off, lds, M, n, float(bound), int(inverse),
                                               ptr(logdet), current_stream(z.device)), "usf_affine_coupling_apply_f32")


def channel_affine(x, y, W, *, pre_sub=None, bias=None):
    """usf_channel_affine_f32 on a contiguous [B, C, *spatial] fp32 tensor (1 x 1 convolution over the channel axis)"""
    B, Cc = x.shape[0], x.shape[1]
    P = math.prod(x.shape[2:])            # (from the shape, not from numel: an empty batch still has pixels)
    _direct("usf_channel_affine_f32", x.data_ptr(), y.data_ptr(), B, Cc, P, W.data_ptr(), ptr(pre_sub), ptr(bias),
                                        current_stream(x.device))


def layernorm_channels(x, gamma, beta, eps, act=ACT_NONE, slope=0.0):
    """usf_layernorm_channels_f32 on a contiguous [B, C, *spatial] fp32 tensor -> new tensor"""
    B, Cc = x.shape[0], x.shape[1]
    P = math.prod(x.shape[2:])            # (from the shape, not from numel: an empty batch still has pixels)
    y = torch.empty_like(x)
    _direct("usf_layernorm_channels_f32", x.data_ptr(), y.data_ptr(), B, Cc, P, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            int(act), float(slope), current_stream(x.device))
    return y


def conv2d_weight_planes_pair(weight: torch.Tensor):
    """(planes, planes_t) of a device fp32 Conv2d weight -- the convolution's own planes and those of its data-gradient
    convolution -- from ONE usf_conv2d_weight_planes_f32 launch (transposed = 2); the same bits as two separate calls"""
    cout, cin, k, _ = weight.shape
    lib = load()
    w = weight.detach().contiguous()
    shapes = []
    for rows, cols in ((cout, cin), (cin, cout)):
        cp, coutp = (cols + 7) // 8 * 8, (rows + 15) // 16 * 16
        shapes.append((3, coutp, (k * k * cp + 31) // 32 * 32))
    n_f, n_t = math.prod(shapes[0]), math.prod(shapes[1])
    assert n_f == lib.usf_conv2d_weight_elems(cin, cout, k) and n_t == lib.usf_conv2d_weight_elems(cout, cin, k)
    buf = torch.empty(n_f + n_t, dtype=torch.bfloat16, device=weight.device)
    check(lib.usf_conv2d_weight_planes_f32(w.data_ptr(), buf.data_ptr(), cin, cout, k, 2, current_stream(weight.device)),
          "usf_conv2d_weight_planes_f32")
    return buf[:n_f].view(shapes[0]), buf[n_f:].view(shapes[1])


class WPlanesJob(C.Structure):
    """usf_wplanes_job"""
    _fields_ = [("w", C.c_void_p), ("out_off", C.c_int64), ("cin", C.c_int32), ("cout", C.c_int32), ("ks", C.c_int32),
                ("first_block", C.c_int32)]


class WeightPlanesBatch:
    """the plane pairs of MANY device fp32 Conv2d weights from ONE launch (usf_conv2d_weight_planes_batch_f32): the job table is
    built once for a set of weights (it names their addresses) and stays on the device; every ``run()`` splits the weights'
    CURRENT values into a fresh buffer and returns [(planes, planes_t)] -- the same bits as conv2d_weight_planes_pair"""

    def __init__(self, weights):
        self.key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
        self.device = weights[0].device
        jobs, block_job, self.views, off, first = [], [], [], 0, 0
        for i, w in enumerate(weights):
            cout, cin, k, _ = w.shape
            shapes = []
            for rows, cols in ((cout, cin), (cin, cout)):
                cp, coutp = (cols + 7) // 8 * 8, (rows + 15) // 16 * 16
                shapes.append((3, coutp, (k * k * cp + 31) // 32 * 32))
            n_f, n_t = math.prod(shapes[0]), math.prod(shapes[1])
            nb = (max(n_f, n_t) // 3 + 255) // 256
            jobs.append(WPlanesJob(w.data_ptr(), off, cin, cout, k, first))
            block_job.extend([i] * nb)
            self.views.append((off, n_f, shapes[0], n_t, shapes[1]))
            off += n_f + n_t
            first += nb
        self.total, self.n_blocks = off, first
        raw = bytearray(bytes((WPlanesJob * len(jobs))(*jobs)))
        raw += b"\0" * ((-len(raw)) % 16)
        self.off_blocks = len(raw)
        raw += struct.pack(f"<{len(block_job)}i", *block_job)
        self.host = torch.frombuffer(raw, dtype=torch.uint8)
        self.table = None

    def run(self):
        """None: inside a stream capture without a table buffer (the caller splits weight by weight)"""
        table = self.table
        if table is None:
            table = _device_table(self.host, self.device)
            if table is None:
                return None
            if not torch.cuda.is_current_stream_capturing():
                self.table = table
        buf = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
        _launch("usf_conv2d_weight_planes_batch_f32", (table.data_ptr(), table.data_ptr() + self.off_blocks, self.n_blocks, buf.data_ptr(),
                                                       current_stream(self.device)), (table, buf))
        return [(buf[o: o + n_f].view(sf), buf[o + n_f: o + n_f + n_t].view(st)) for o, n_f, sf, n_t, st in self.views]


def conv2d_weight_planes(weight: torch.Tensor, gate_channels: int = 0, transposed: bool = False) -> torch.Tensor:
    """bf16x3 planes [3, coutp, kp] of a Conv2d weight [cout, cin, k, k] in the K order usf_conv2d_same_f32 reads
    (tap-major, channel-minor, channels padded to a multiple of 8; include/usflows_hip.h).  gate_channels = C > 0: the
    weight has 2C rows (C values, C gates) and is packed for the gated mode (rows interleaved in tiles of 16).
    transposed: planes of the convolution that computes the data gradient (flipped taps, channels swapped).  A device
    weight without gate packing is split by usf_conv2d_weight_planes_f32 (one launch); the torch formulation below gives the
    same bits (round-to-nearest-even residual split)"""
    cout, cin, k, _ = weight.shape
    if weight.is_cuda and not gate_channels and weight.dtype == torch.float32:
        lib = load()
        w = weight.detach().contiguous()
        rows, cols = (cin, cout) if transposed else (cout, cin)
        cp, coutp = (cols + 7) // 8 * 8, (rows + 15) // 16 * 16
        planes = torch.empty(3, coutp, (k * k * cp + 31) // 32 * 32, dtype=torch.bfloat16, device=weight.device)
        assert planes.numel() == lib.usf_conv2d_weight_elems(cols, rows, k)
        check(lib.usf_conv2d_weight_planes_f32(w.data_ptr(), planes.data_ptr(), cin, cout, k, int(transposed),
                                               current_stream(weight.device)), "usf_conv2d_weight_planes_f32")
        return planes
    if transposed:
        weight = weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        cout, cin = cin, cout
    w32 = weight.detach().to(torch.float32)
    if gate_channels:
        w32 = w32[conv2d_gate_row_order(gate_channels, weight.device)] * \
            (conv2d_gate_row_order(gate_channels, weight.device, valid=True)).view(-1, 1, 1, 1)
        cout = w32.shape[0]
    cp, coutp = (cin + 7) // 8 * 8, (cout + 15) // 16 * 16
    kp = (k * k * cp + 31) // 32 * 32
    w = torch.zeros(coutp, k * k, cp, dtype=torch.float32, device=weight.device)
    w[:cout, :, :cin] = w32.reshape(cout, cin, k * k).permute(0, 2, 1)
    flat = torch.zeros(coutp, kp, dtype=torch.float32, device=weight.device)
    flat[:, : k * k * cp] = w.reshape(coutp, k * k * cp)
    h = flat.to(torch.bfloat16)
    r = flat - h.float()
    m = r.to(torch.bfloat16)
    lo = (r - m.float()).to(torch.bfloat16)
    planes = torch.stack([h, m, lo]).contiguous()
    assert planes.numel() == load().usf_conv2d_weight_elems(cin, cout, k)
    return planes


def conv2d_gate_row_order(C: int, device, valid: bool = False) -> torch.Tensor:
    """gated mode of usf_conv2d_same_f32: packed row 32 t + r <- original row (16 t + r | C + 16 t + r - 16), clamped;
    valid=True: 1.0 where the packed row is a real channel, 0.0 where it is padding"""
    nt = (C + 15) // 16
    t = torch.arange(nt, device=device).view(-1, 1)
    r = torch.arange(16, device=device).view(1, -1)
    ch = 16 * t + r                                         # [nt, 16]
    ok = (ch < C)
    val = ch.clamp(max=C - 1)
    rows = torch.stack([val, val + C], dim=1).reshape(-1)   # per tile: 16 value rows, then 16 gate rows
    if valid:
        return torch.stack([ok, ok], dim=1).reshape(-1).to(torch.float32)
    return rows


def conv2d_same(x, planes, cout, ks, bias=None, in_mul=None, in_act=ACT_NONE, in_slope=0.0, out_act=ACT_NONE, out_slope=0.0,
                gate_x=None):
    """usf_conv2d_same_f32 on a contiguous [B, cin, H, W] fp32 tensor -> new [B, cout, H, W] tensor; gate_x [B, C, H, W]:
    gated mode (planes / bias packed with gate_channels = C, cout = 32 * ceil(C / 16)) -> new [B, C, H, W] tensor"""
    B, cin, H, W = x.shape
    gc = 0 if gate_x is None else gate_x.shape[1]
    y = torch.empty(B, gc if gc else cout, H, W, dtype=torch.float32, device=x.device)
    _direct("usf_conv2d_same_f32", x.data_ptr(), y.data_ptr(), B, cin, cout, H, W, ks, planes.data_ptr(), ptr(bias),
                                     ptr(in_mul), int(in_act), float(in_slope), int(out_act), float(out_slope),
                                     ptr(gate_x), gc, current_stream(x.device))
    return y


def conv2d_same_res(x, planes, cout, ks, res_x, res_mul, res_sign, bias=None, in_mul=None, in_act=ACT_NONE, in_slope=0.0):
    """usf_conv2d_same_res_f32: the conditioner's last convolution with MaskedCoupling's residual in its output stream ->
    res_x + res_sign * (res_mul * conv(x)) as a new tensor, or None when the fused form does not serve the shape"""
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    fn = load().usf_conv2d_same_res_f32
    args = (x.data_ptr(), y.data_ptr(), B, cin, cout, H, W, ks, planes.data_ptr(), ptr(bias), ptr(in_mul), int(in_act),
            float(in_slope), res_x.data_ptr(), res_mul.data_ptr(), float(res_sign), current_stream(x.device))
    rc = _timed_call(fn, args, "usf_conv2d_same_res_f32")
    if rc == 1:
        return None
    cl = getattr(_tls, "calls", None)
    if cl is not None and rc == 0:
        cl.add("usf_conv2d_same_res_f32", args)
    check(rc, "usf_conv2d_same_res_f32")
    return y


def conv2d_same_gate(x, planes, cout, ks, gate_h, gate_slope, gate_mul=None, gate_add=None):
    """usf_conv2d_same_gate_f32: conv(x) * (gate_h > 0 ? 1 : gate_slope) * gate_mul, or gate_add + conv(x) * (...), as a new
    tensor (a data-gradient convolution with the input nonlinearity's / mask's factors, or the other branch's gradient, in its
    output stream), or None when not served"""
    B, cin, H, W = x.shape
    y = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    args = (x.data_ptr(), y.data_ptr(), B, cin, cout, H, W, ks, planes.data_ptr(), gate_h.data_ptr(), float(gate_slope), ptr(gate_mul),
            ptr(gate_add), current_stream(x.device))
    rc = _timed_call(load().usf_conv2d_same_gate_f32, args, "usf_conv2d_same_gate_f32")
    if rc == 1:
        return None
    check(rc, "usf_conv2d_same_gate_f32")
    return y


def pointwise_conv_supported(cin: int, cout: int, gated: bool = False) -> bool:
    return bool(load().usf_pointwise_conv_supported(int(cin), int(cout), int(bool(gated))))


def pointwise_conv(x, W, bias=None, in_act=ACT_NONE, in_slope=0.0, out_act=ACT_NONE, out_slope=0.0, gate_x=None, ln=None):
    """usf_pointwise_conv_f32: 1 x 1 convolution of a contiguous [B, cin, *spatial] fp32 tensor with W [cout, cin] on the vector
    ALUs -> new [B, cout, *spatial] tensor; gate_x [B, cout / 2, *spatial]: gated mode -> new tensor shaped like gate_x;
    ln = (gamma [C], beta [C], eps): out_act and a layer norm over the channels joined to the gated pass"""
    B, cin = x.shape[0], x.shape[1]
    cout = W.shape[0]
    P = math.prod(x.shape[2:])
    y = torch.empty_like(gate_x) if gate_x is not None else torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _direct("usf_pointwise_conv_f32", x.data_ptr(), y.data_ptr(), B, cin, cout, P, W.data_ptr(), ptr(bias), int(in_act),
                                        float(in_slope), int(out_act), float(out_slope), ptr(gate_x),
                                        None if ln is None else ln[0].data_ptr(), None if ln is None else ln[1].data_ptr(),
                                        0.0 if ln is None else float(ln[2]), current_stream(x.device))
    return y


def gated_residual(x, vg):
    """x + vg[:, :C] * sigmoid(vg[:, C:]) for contiguous x [B, C, *spatial], vg [B, 2C, *spatial] -> new tensor"""
    B = x.shape[0]
    CP = math.prod(x.shape[1:])
    y = torch.empty_like(x)
    _direct("usf_gated_residual_f32", x.data_ptr(), vg.data_ptr(), y.data_ptr(), B, CP, current_stream(x.device))
    return y


def gated_norm_rows(skip, *, M, C_cols, c_pad=None, ld_skip=None, vg=None, ld_vg=0, gate_off=0, gamma=None, beta=None, eps=1e-5,
                    out=None, ld_out=0, out_act=None, ld_act=0, act=ACT_NONE, slope=0.0):
    """usf_gated_norm_rows_f32 on raw [M, ld] fp32 buffers (see include/usflows_hip.h)"""
    d = GatedNormDesc(skip=skip.data_ptr(), ld_skip=ld_skip if ld_skip is not None else C_cols, vg=ptr(vg), ld_vg=ld_vg,
                      gate_off=gate_off, gamma=ptr(gamma), beta=ptr(beta), out=ptr(out), ld_out=ld_out, out_act=ptr(out_act),
                      ld_act=ld_act, M=M, C=C_cols, c_pad=c_pad if c_pad is not None else C_cols, eps=eps, slope=slope, act=act)
    _launch("usf_gated_norm_rows_f32", (C.byref(d), current_stream(skip.device)), (d, skip, vg, gamma, beta, out, out_act))


def gated_norm_rows_bwd(skip, dy, d_skip, *, M, C_cols, c_pad, ld_skip, ld_dy, ld_d_skip, vg=None, ld_vg=0, gate_off=0, d_vg=None,
                        ld_d_vg=0, gamma=None, eps=1e-5, dy_xh=None, ld_dy_xh=0):
    """usf_gated_norm_rows_bwd_f32 on raw [M, ld] fp32 buffers (see include/usflows_hip.h)"""
    d = GatedNormBwdDesc(skip=skip.data_ptr(), ld_skip=ld_skip, vg=ptr(vg), ld_vg=ld_vg, gate_off=gate_off, gamma=ptr(gamma),
                         dy=dy.data_ptr(), ld_dy=ld_dy, d_skip=d_skip.data_ptr(), ld_d_skip=ld_d_skip, d_vg=ptr(d_vg), ld_d_vg=ld_d_vg,
                         dy_xh=ptr(dy_xh), ld_dy_xh=ld_dy_xh, M=M, C=C_cols, c_pad=c_pad, eps=eps)
    _launch("usf_gated_norm_rows_bwd_f32", (C.byref(d), current_stream(skip.device)), (d, skip, vg, gamma, dy, d_skip, d_vg, dy_xh))


def add_rows(x, t, ones) -> None:
    """x += t for two [M, ld] fp32 buffers of equal row stride (usf_masked_residual_f32 with a mask of ones [ld]; taped)"""
    M, ld = x.shape[0], x.shape[1]
    _launch("usf_masked_residual_f32", (x.data_ptr(), t.data_ptr(), ones.data_ptr(), 1.0, x.data_ptr(), M, ld, current_stream(x.device)),
            (x, t, ones))


def masked_residual(x, t, one_minus_mask, sign):
    """x + sign * one_minus_mask * t (mask [C * P] fp32, broadcast over the batch) -> new tensor; x None: taken as zeros"""
    B = t.shape[0]
    CP = math.prod(t.shape[1:])
    y = torch.empty_like(t)
    _direct("usf_masked_residual_f32", ptr(x), t.data_ptr(), one_minus_mask.data_ptr(), float(sign), y.data_ptr(), B,
                                         CP, current_stream(t.device))
    return y


# ---- deferred sums of a backward pass (small batches): usf_conv_wgrad_deferred_f32 + ONE usf_partial_sum_jobs_f32 ------------
# A weight gradient ends with a sum over per-wave partial slots -- at the reference's training batch a launch of a few
# microseconds on the chain of dependent launches that bounds the step (live MNIST configuration: ~165 of ~1250 launches).  A
# caller inside an autograd backward pass may ask ``conv_wgrad(..., defer=True)``: the sums of the pass are queued and leave as
# ONE launch when the pass ends (an autograd engine callback).  dW / db are handed out UNWRITTEN until then, so the caller
# vouches that nothing reads them before the pass ends (a parameter without a ``.grad`` takes the tensor as it is).
# Inside a stream capture (Flow.fit's captured step) the job table cannot be uploaded (a host-to-device copy is not capturable)
# and must not live in memory allocated inside the capture (recycled between the graph's kernels): ``capture_tables`` hands out
# slices of a buffer allocated BEFORE the capture and uploads their contents after it, before the first replay.
PSUM_DEFER_MAX_ROWS = 1024        # (launch-bound batches; above, the sums are a small part of kernels that take real time)
n_jobs_flushed = [0]             # (introspection for tests / tools: sums that left through usf_partial_sum_jobs_f32)
n_wgrad_jobs_flushed = [0]       # (... and weight-gradient launches that left through usf_conv_wgrad_jobs_f32)


class _PsumState:
    """process-wide (under a lock), NOT thread-local: the autograd engine runs a device's backward nodes on a worker thread of
    its own, while the end-of-pass callback runs on the thread that called backward()"""

    def __init__(self):
        self.lock = threading.Lock()
        self.capture_lock = threading.RLock()    # held for the whole of a capture_tables block (re-entrant: nested blocks of one thread)
        self.jobs = {}           # autograd graph task id -> [(PsumJob, tensors to keep alive until the launch)]
        self.arena = None        # capture_tables: [buffer, bytes used, [(device slice, host tensor)]]
        self.scope = 0           # > 0: inside deferred_sums_scope (a backward pass whose caller vouches for its parameters)
        self.owners = {}         # autograd graph task id -> ids of the Parameters that already have a queued gradient


_psum = _PsumState()


class deferred_sums_scope:
    """``with deferred_sums_scope(): loss.backward()`` -- only inside such a scope may a weight gradient's last sum be queued
    until the backward pass ends.  A queued gradient is handed to autograd UNWRITTEN, so whoever opens the scope vouches that
    nothing reads a parameter's gradient before the pass ends: no hooks on the parameters' gradient accumulators (torch's
    DistributedDataParallel registers such hooks from C++, invisible to Python -- hence a scope, not a per-parameter check) and
    no optimiser / user code between the node and the end of the pass.  Flow.fit opens it around its own steps' backward
    passes; a ``loss.backward()`` anywhere else never defers.  What the scope cannot know is checked per call: a parameter that
    already holds a ``.grad`` or has tensor hooks (image_training._takeable), and a SECOND producer for the same parameter in
    one pass (``conv_wgrad(owners=...)``: autograd adds the two gradients as soon as the second arrives -- the queue is flushed
    first and that call stays undeferred)."""

    def __enter__(self):
        with _psum.lock:
            _psum.scope += 1
        return self

    def __exit__(self, *exc):
        with _psum.lock:
            _psum.scope -= 1
        return False


class capture_tables:
    """``with capture_tables(device, nbytes): <stream capture>`` -- device tables needed by launches inside the capture"""

    def __init__(self, device, nbytes: int = 8 << 20):
        self.device, self.nbytes, self.prev = device, nbytes, None

    def __enter__(self):
        # ONE capture with tables at a time per process: the buffer is read by the autograd engine's worker threads (not the
        # thread that captures), so it cannot be thread-local; a second thread that wants to capture waits here until the first
        # capture has ended (its tables would otherwise land in the wrong buffer)
        _psum.capture_lock.acquire()
        buf = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
        with _psum.lock:
            self.prev = _psum.arena
            _psum.arena = [buf, 0, []]
        return self

    def __exit__(self, exc_type, *exc):
        with _psum.lock:
            arena, _psum.arena = _psum.arena, self.prev
        _psum.capture_lock.release()
        self.pending = arena[2] if exc_type is None else []
        self.keep = arena[0]
        return False

    def upload(self):
        """after the capture has ended: the tables' contents (nothing of the graph has run yet)"""
        for dev, host in self.pending:
            dev.copy_(host)
        self.pending = []


def _device_table(host: torch.Tensor, device) -> Optional[torch.Tensor]:
    """a uint8 host tensor on the device: an upload -- or, inside a capture, a slice of the capture's table buffer whose upload is
    pending (None: capturing without such a buffer)"""
    if torch.cuda.is_current_stream_capturing():
        n = (host.numel() + 15) // 16 * 16
        with _psum.lock:
            ar = _psum.arena
            if ar is None or ar[0].device != torch.device(device) or ar[1] + n > ar[0].numel():
                return None
            dev = ar[0][ar[1]: ar[1] + host.numel()]
            ar[1] += n
            ar[2].append((dev, host))
        return dev
    return host.to(device)


def psum_defer_ok(x) -> bool:
    """may a weight gradient of this input queue its last sum?  (small batch, inside a backward pass, and -- inside a capture --
    a table buffer at hand)"""
    return (_psum.scope > 0 and config.psum_jobs and 0 < x.shape[0] <= PSUM_DEFER_MAX_ROWS
            and torch._C._current_graph_task_id() >= 0
            and (not torch.cuda.is_current_stream_capturing() or _psum.arena is not None))


def flush_partial_sums(task: int, final: bool = True) -> None:
    """issue the sums backward pass `task` queued on the current stream (the pass's end-of-pass callback): ONE launch for all
    first rounds, ONE for all final rounds"""
    with _psum.lock:
        jobs = _psum.jobs.pop(task, [])
        if final:          # (a flush in the middle of the pass keeps the record of which parameters already have a gradient)
            _psum.owners.pop(task, None)
    if not jobs:
        return
    device = jobs[0][2][0].device
    n_jobs_flushed[0] += len(jobs)
    # the queued weight-gradient launches first: one per tile shape, all its jobs' blocks side by side
    groups = {}
    for j in jobs:
        w = j[4]
        if w is not None:
            groups.setdefault((w.CIT, w.COT, w.T), []).append((w, j[2]))
    for (cit, cot, t_), members in groups.items():
        block_job, first = [], 0
        for i, (w, _keep) in enumerate(members):
            w.first_block = first
            block_job.extend([i] * w.blocks)
            first += w.blocks
        raw = bytearray(bytes((WgradJob * len(members))(*[w for w, _ in members])))
        raw += b"\0" * ((-len(raw)) % 16)
        off = len(raw)
        raw += struct.pack(f"<{len(block_job)}i", *block_job)
        table = _device_table(torch.frombuffer(raw, dtype=torch.uint8), device)
        if table is None:
            raise RuntimeError("usflows_amd: deferred weight gradients inside a stream capture without a capture_tables buffer")
        n_wgrad_jobs_flushed[0] += len(members)
        _launch("usf_conv_wgrad_jobs_f32", (table.data_ptr(), table.data_ptr() + off, first, cit, cot, t_,
                                            max(w.lds_bytes for w, _ in members), current_stream(device)),
                (table, [k for _, k in members]))
    for stage in (0, 1):
        part = [(j[stage], j[2]) for j in jobs if j[stage] is not None]
        if not part:
            continue
        block_job, first = [], 0
        for i, (j, _keep) in enumerate(part):
            nb = ((j.n + 255) // 256 if j.vec4 else (j.n + 63) // 64) * j.rows
            j.first_block = first
            block_job.extend([i] * nb)
            first += nb
        arr = (PsumJob * len(part))(*[j for j, _ in part])
        raw = bytearray(bytes(arr))
        raw += b"\0" * ((-len(raw)) % 16)
        off = len(raw)
        raw += struct.pack(f"<{len(block_job)}i", *block_job)
        host = torch.frombuffer(raw, dtype=torch.uint8)
        table = _device_table(host, device)
        if table is None:
            raise RuntimeError("usflows_amd: deferred partial sums inside a stream capture without a capture_tables buffer")
        _launch("usf_partial_sum_jobs_f32", (table.data_ptr(), table.data_ptr() + off, first, current_stream(device)),
                (table, [k for _, k in part]))


def _psum_may_defer(x, owners) -> bool:
    """may the producer of gradients for the Parameters `owners` (ids) queue its last sum in the current backward pass?  A second
    producer for the same parameter flushes the queue first and is not deferred (autograd adds the two gradients when the second
    one arrives: the first must be complete by then)"""
    if not (psum_defer_ok(x) and owners):
        return False
    task = torch._C._current_graph_task_id()
    with _psum.lock:
        if task not in _psum.owners:
            for old in [t for t in _psum.owners if t < task - 256]:      # (passes that never met their end-of-pass callback)
                del _psum.owners[old]
        seen = _psum.owners.setdefault(task, set())
        dup = any(o in seen for o in owners)
        seen.update(owners)
    if dup:
        flush_partial_sums(task, final=False)      # the earlier producer's sums run now, in stream order before autograd's add
        return False
    # (table bytes this pass has queued so far: inside a capture they must fit the pre-capture buffer)
    with _psum.lock:
        queued = _psum.jobs.get(task)
        used = sum(q_[3] for q_ in queued) if queued else 0
    if torch.cuda.is_current_stream_capturing():
        with _psum.lock:
            ar = _psum.arena
        if ar is None or used > ar[0].numel() // 2 - (1 << 16):
            return False
    return True


def _psum_queue(job2, keep, wjob=None) -> None:
    """queue the (first round | none, last round) jobs an entry point returned; `keep` = the tensors holding the partial slots
    (and, with a queued weight-gradient launch `wjob`, its operands)"""
    if job2[1].nparts <= 0:
        return
    task = torch._C._current_graph_task_id()
    with _psum.lock:
        q = _psum.jobs.get(task)
        if q is None:
            # (queues of passes that died of an exception never met their callback: graph task ids only grow, so what is
            # far behind the current one is dropped -- NOT everything else: another thread's pass may be queueing too)
            for old in [t for t in _psum.jobs if t < task - 256]:
                del _psum.jobs[old]
            q = _psum.jobs[task] = []
            torch.autograd.Variable._execution_engine.queue_callback(lambda t=task: flush_partial_sums(t, final=True))
        # (only the partial slots are kept alive: an extra reference to the outputs would stop the autograd engine from
        # TAKING them as the parameter's gradient -- it would copy them on the spot, before the sum has run)
        j0 = PsumJob.from_buffer_copy(job2[0]) if job2[0].nparts > 0 else None
        nbytes = 2 * C.sizeof(PsumJob) + 4 * sum(((j_.n + 63) // 64) * j_.rows for j_ in job2 if j_.nparts > 0) + 64
        if wjob is not None:
            nbytes += C.sizeof(WgradJob) + 4 * wjob.blocks + 32
        q.append((j0, PsumJob.from_buffer_copy(job2[1]), tuple(keep), nbytes, wjob))


def conv_wgrad(x, dy, ks, in_mul=None, pre_sub=None, in_act=ACT_NONE, in_slope=0.0, want_bias=True, defer=False, owners=()):
    """usf_conv_wgrad_f32: (dW [cout, cin, ks, ks], db [cout] | None) of a stride-1 "same" convolution from its input x
    [B, cin, H, W] (with the forward's input transforms) and the output gradient dy [B, cout, H, W]; None when the shape is
    not served.  defer (see above): the final sum may be queued -- dW / db are then complete when the backward pass ends.
    owners: ids of the Parameters the gradients are for -- a second call for the same parameter inside one backward pass
    flushes the queue and is not deferred (autograd sums the two gradients when the second one arrives)"""
    B, cin, H, W = x.shape
    cout = dy.shape[1]
    lib = load()
    ws_n = lib.usf_conv_wgrad_workspace(B, cin, cout, H, W, ks)
    if ws_n <= 0:
        return None
    ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    dW = torch.empty(cout, cin, ks, ks, dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    if defer and _psum_may_defer(x, owners):
        job2 = (PsumJob * 2)()
        if config.wgrad_jobs:
            # the weight-gradient launch itself is queued too (one launch per tile shape when the pass ends); the direct kernel-1
            # form is launched here (wjob.blocks == 0) and only its sums wait
            wjob = WgradJob()
            rc = lib.usf_conv_wgrad_plan_f32(x.data_ptr(), dy.data_ptr(), B, cin, cout, H, W, ks, ptr(in_mul), ptr(pre_sub), int(in_act),
                                             float(in_slope), dW.data_ptr(), ptr(db), ws.data_ptr(), ws_n, C.addressof(job2),
                                             C.addressof(wjob), current_stream(x.device))
            if rc == 1:
                return None
            check(rc, "usf_conv_wgrad_plan_f32")
            if wjob.blocks > 0:
                _psum_queue(job2, (ws, x, dy, in_mul, pre_sub), wjob)
            else:
                _psum_queue(job2, (ws,))
            return dW, db
        rc = lib.usf_conv_wgrad_deferred_f32(x.data_ptr(), dy.data_ptr(), B, cin, cout, H, W, ks, ptr(in_mul), ptr(pre_sub), int(in_act),
                                             float(in_slope), dW.data_ptr(), ptr(db), ws.data_ptr(), ws_n, job2, current_stream(x.device))
        if rc == 1:
            return None
        check(rc, "usf_conv_wgrad_deferred_f32")
        _psum_queue(job2, (ws,))
        return dW, db
    args = (x.data_ptr(), dy.data_ptr(), B, cin, cout, H, W, ks, ptr(in_mul), ptr(pre_sub), int(in_act), float(in_slope),
            dW.data_ptr(), ptr(db), ws.data_ptr(), ws_n, current_stream(x.device))
    rc = _timed_call(lib.usf_conv_wgrad_f32, args, "usf_conv_wgrad_f32")
    if rc == 1:
        return None
    check(rc, "usf_conv_wgrad_f32")
    return dW, db


def layernorm_channels_bwd(x, dy, gamma, eps, act=ACT_NONE, slope=0.0):
    """usf_layernorm_channels_bwd_f32 -> (dx, dgamma [C], dbeta [C])"""
    B, Cc = x.shape[0], x.shape[1]
    P = math.prod(x.shape[2:])
    lib = load()
    ws_n = lib.usf_layernorm_channels_bwd_workspace(B, Cc, P)
    ws = torch.empty(max(1, ws_n), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgb = torch.empty(2 * Cc, dtype=torch.float32, device=x.device)
    _direct("usf_layernorm_channels_bwd_f32", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, Cc, P, gamma.data_ptr(), float(eps),
            int(act), float(slope), dgb.data_ptr(), ws.data_ptr(), ws_n, current_stream(x.device))
    return dx, dgb[:Cc], dgb[Cc:]


def gated_residual_bwd(dy, vg):
    """usf_gated_residual_bwd_f32 -> d(vg) [B, 2C, *spatial] (the gradient with respect to x is dy)"""
    B = dy.shape[0]
    CP = math.prod(dy.shape[1:])
    dvg = torch.empty_like(vg)
    _direct("usf_gated_residual_bwd_f32", dy.data_ptr(), vg.data_ptr(), dvg.data_ptr(), B, CP, current_stream(dy.device))
    return dvg


def gated_tail_supported(C_: int) -> bool:
    return bool(load().usf_gated_tail_supported(int(C_)))


def gated_tail(h, x, W, bias, in_act=ACT_NONE, in_slope=0.0, post_act=ACT_NONE, post_slope=0.0, ln=None):
    """usf_gated_tail_f32: GatedConv's 1 x 1 convolution + gate + skip [+ nonlinearity + LayerNormChannels, ln = (gamma, beta, eps)]
    of contiguous [B, C, *spatial] tensors h (the 3 x 3 convolution's output) and x (the layer's input) in one launch"""
    B, Cc = x.shape[0], x.shape[1]
    P = math.prod(x.shape[2:])
    y = torch.empty_like(x)
    g, bt, eps = ln if ln is not None else (None, None, 0.0)
    _direct("usf_gated_tail_f32", h.data_ptr(), x.data_ptr(), y.data_ptr(), B, Cc, P, W.data_ptr(), ptr(bias), int(in_act),
            float(in_slope), int(post_act), float(post_slope), ptr(g), ptr(bt), float(eps), current_stream(x.device))
    return y


def gated_tail_bwd(h, x, dy, W, bias, in_act=ACT_NONE, in_slope=0.0, post_act=ACT_NONE, post_slope=0.0, ln=None, defer=False,
                   owners=(), want_dvg=False):
    """usf_gated_tail_bwd_f32 -> (dx, dh, dW [2C, C], dbias [2C], dgamma [C] | None, dbeta [C] | None, dvg | None); the parameter
    gradients are views of one buffer; defer / owners as in conv_wgrad (their last sum may be queued until the backward pass
    ends); want_dvg: also d[val, gate] [B, 2C, *spatial] (tests)"""
    B, Cc = x.shape[0], x.shape[1]
    P = math.prod(x.shape[2:])
    lib = load()
    dx, dh = torch.empty_like(x), torch.empty_like(x)
    dvg = torch.empty((B, 2 * Cc) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device) if want_dvg else None
    g, bt, eps = ln if ln is not None else (None, None, 0.0)
    ws_n = lib.usf_gated_tail_workspace(B, Cc, P)
    ws = torch.empty(max(1, ws_n), dtype=torch.float32, device=x.device)
    nW = 2 * Cc * Cc
    dpar = torch.empty(nW + 2 * Cc + (2 * Cc if ln is not None else 0), dtype=torch.float32, device=x.device)
    job2 = (PsumJob * 2)() if (defer and _psum_may_defer(x, owners)) else None
    args = (h.data_ptr(), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), dh.data_ptr(), ptr(dvg), B, Cc, P, W.data_ptr(), ptr(bias),
            int(in_act), float(in_slope), int(post_act), float(post_slope), ptr(g), ptr(bt), float(eps), dpar.data_ptr(), ws.data_ptr(),
            ws_n, C.cast(job2, C.c_void_p) if job2 is not None else None, current_stream(x.device))
    check(_timed_call(lib.usf_gated_tail_bwd_f32, args, "usf_gated_tail_bwd_f32"), "usf_gated_tail_bwd_f32")
    if job2 is not None:
        _psum_queue(job2, (ws,))
    dW, db = dpar[:nW].view(2 * Cc, Cc), dpar[nW: nW + 2 * Cc]
    if ln is None:
        return dx, dh, dW, db, None, None, dvg
    return dx, dh, dW, db, dpar[nW + 2 * Cc: nW + 3 * Cc], dpar[nW + 3 * Cc:], dvg


AFFINE_PREP_MAX_C = 64


def affine_prep(Lr, Ur, bias, vk=None, w0=None):
    """usf_affine_prep_f32 on stacked parameters -> (M, Minv [n,C,C], b, c [n,C], ladj [n], save [n,7,C,C])"""
    n, Cc = int(Lr.shape[0]), int(Lr.shape[1])
    nvs = 0 if vk is None else int(vk.shape[1])
    dev = Lr.device
    # (M | Minv and b | c share a buffer each: image_training.compose_runs gathers rows of either kind with one index_select)
    MM = torch.empty(2 * n, Cc, Cc, dtype=torch.float32, device=dev)
    bc = torch.empty(2 * n, Cc, dtype=torch.float32, device=dev)
    M, Minv, b, c = MM[:n], MM[n:], bc[:n], bc[n:]
    ladj = torch.empty(n, dtype=torch.float32, device=dev)
    save = torch.empty(n, 7, Cc, Cc, dtype=torch.float32, device=dev)
    _direct("usf_affine_prep_f32", Lr.data_ptr(), Ur.data_ptr(), bias.data_ptr(), ptr(vk), ptr(w0), n, Cc, nvs, M.data_ptr(),
            Minv.data_ptr(), b.data_ptr(), c.data_ptr(), ladj.data_ptr(), save.data_ptr(), current_stream(dev))
    return M, Minv, b, c, ladj, save


def affine_prep_bwd(save, bias, vk, w0, Minv, b, dM, dMinv, db, dc, dladj):
    """usf_affine_prep_bwd_f32 -> (dL_raw, dU_raw [n,C,C], dbias [n,C], dvk [n,nvs,C] | None)"""
    n, Cc = int(save.shape[0]), int(save.shape[2])
    nvs = 0 if vk is None else int(vk.shape[1])
    dev = save.device
    dLr = torch.empty(n, Cc, Cc, dtype=torch.float32, device=dev)
    dUr = torch.empty_like(dLr)
    dbias = torch.empty(n, Cc, dtype=torch.float32, device=dev)
    dvk = torch.empty(n, nvs, Cc, dtype=torch.float32, device=dev) if nvs else None
    _direct("usf_affine_prep_bwd_f32", save.data_ptr(), bias.data_ptr(), ptr(vk), ptr(w0), Minv.data_ptr(), b.data_ptr(),
            dM.data_ptr(), dMinv.data_ptr(), db.data_ptr(), dc.data_ptr(), dladj.data_ptr(), n, Cc, nvs, dLr.data_ptr(),
            dUr.data_ptr(), dbias.data_ptr(), ptr(dvk), current_stream(dev))
    return dLr, dUr, dbias, dvk


def gather_cols(src, lds, dst, ldd, M, n, idx):
    check(load().usf_gather_cols_f32(src.data_ptr(), lds, dst.data_ptr(), ldd, M, n, idx.data_ptr(),
                                     current_stream(src.device)), "usf_gather_cols_f32")


def coupling_op(op, device):
    """one usf_coupling_additive_f32 launch from an Op built by the engine (its descriptor; taped like every launch)"""
    _launch("usf_coupling_additive_f32", (C.byref(op.u.coupling), current_stream(device)), op)


def mfma_probe(device, iters: int = 400, repeats: int = 5) -> dict:
    """usf_mfma_probe timed with HIP events on torch's current stream: what the part sustains on the bf16 matrix cores for
    the planes GEMM's instruction mix (register-only loop).  Returns TFLOP/s of bf16 MFMA work and the fp32-equivalent (/ 6);
    the median of `repeats` launches after one warm-up."""
    src = torch.rand(1024, device=device) + 0.5
    sink = torch.zeros(1, device=device)
    flops = C.c_double(0.0)
    times = []
    for i in range(repeats + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(load().usf_mfma_probe(src.data_ptr(), sink.data_ptr(), iters, 0, C.byref(flops), current_stream(device)), "usf_mfma_probe")
        e1.record()
        e1.synchronize()
        if i:
            times.append(e0.elapsed_time(e1))
    times.sort()
    ms = times[len(times) // 2]
    tf = flops.value / (ms * 1e-3) / 1e12
    return dict(ms=ms, tflops_bf16=tf, tflops_f32_equiv=tf / 6.0, iters=iters, launches=repeats)


class clock_meter:
    """``with clock_meter(device) as m: ...launches...`` then ``m.mhz()``: the shader clock under the planes GEMM / the MFMA probe
    launched inside the block (usf_set_clock_buffer: block lifetimes in shader cycles over the same in 100 MHz ticks)"""

    def __init__(self, device):
        self.buf = torch.zeros(2, dtype=torch.int64, device=device)

    def __enter__(self):
        check(load().usf_set_clock_buffer(self.buf.data_ptr()), "usf_set_clock_buffer")
        return self

    def __exit__(self, *exc):
        check(load().usf_set_clock_buffer(0), "usf_set_clock_buffer")
        return False

    def mhz(self):
        c, t = (int(v) for v in self.buf.cpu().tolist())
        return 100.0 * c / t if t > 0 else None


def coupling_planes_op(op, device):
    """one usf_coupling_planes launch from an Op built by the engine (taped like every launch)"""
    _launch("usf_coupling_planes", (C.byref(op.u.coupling_planes), current_stream(device)), op)


def run_ops(ops_array, n, device=None):
    check(load().usf_run_ops(ops_array, n, current_stream(device)), "usf_run_ops")


# ---- parameter prep (usf_prep.hip) -------------------------------------------------
def lu_prepare(L_raws, U_raws, want_M=True, want_Minv=True, keep_factors=False):
    """Batched LUTransform prep (transforms.py:1271-1320) for a list of [D,D] fp32 device parameters.

    Returns dict(M [n,D,D] f64 | None, Minv | None, ladj [n] f64, and with keep_factors: tri / tri_inv
    [2n,D,D] f64 = (L_i, U_i^T) / (L_i^-1, (U_i^-1)^T))."""
    n = len(L_raws)
    D = int(L_raws[0].shape[0])
    dev = L_raws[0].device
    for t in list(L_raws) + list(U_raws):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (D, D) or t.device != dev:
            raise ValueError("lu_prepare: parameters must be contiguous [D,D] fp32 tensors on one device")
    f64 = lambda *shape: torch.empty(*shape, dtype=torch.float64, device=dev)
    tri, tri_inv, work = f64(2 * n, D, D), f64(2 * n, D, D), f64(2 * n, D, D)
    out = dict(M=f64(n, D, D) if want_M else None, Minv=f64(n, D, D) if want_Minv else None, ladj=f64(n))
    d = LuPrepDesc()
    d.n, d.D = n, D
    Lp = (C.c_void_p * n)(*[t.data_ptr() for t in L_raws])
    Up = (C.c_void_p * n)(*[t.data_ptr() for t in U_raws])
    d.L_raw, d.U_raw = Lp, Up
    d.tri, d.tri_inv, d.work = tri.data_ptr(), tri_inv.data_ptr(), work.data_ptr()
    d.M, d.Minv, d.ladj = ptr(out["M"]), ptr(out["Minv"]), out["ladj"].data_ptr()
    _launch("usf_lu_prepare_f64", (C.byref(d), current_stream(dev)), (d, Lp, Up, tri, tri_inv, work, out))
    if keep_factors:
        out["tri"], out["tri_inv"] = tri, tri_inv
    return out


def gemm_f64(A, B, Cout, *, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=1, strideA=0, strideB=0,
             strideC=0, alpha=1.0, beta=0.0, tri=0, a_off=0, b_off=0, c_off=0):
    """usf_gemm_f64 on fp64 device tensors (element offsets *_off)."""
    _launch("usf_gemm_f64", (A.data_ptr() + 8 * a_off, lda, strideA, int(transA), B.data_ptr() + 8 * b_off, ldb,
                             strideB, int(transB), Cout.data_ptr() + 8 * c_off, ldc, strideC, M, N, K, batch,
                             float(alpha), float(beta), tri, current_stream(A.device)), (A, B, Cout))


def matmul_f64(A: torch.Tensor, B: torch.Tensor, transA=False, transB=False, tri=0) -> torch.Tensor:
    """op(A) @ op(B) for contiguous 2-D fp64 device tensors."""
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    out = torch.empty(M, N, dtype=torch.float64, device=A.device)
    gemm_f64(A, B, out, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, transA=transA, transB=transB, tri=tri)
    return out


def lu_grad_finish(dL, dU, TL, TU, c, tri, n, D, out_L, out_U):
    """tril(dL + TL, -1) and triu(dU + TU) + diag(c / diag U) of n blocks, fp64 -> the fp32 gradient arena, one launch"""
    _launch("usf_lu_grad_finish_f64", (dL.data_ptr(), dU.data_ptr(), ptr(TL), ptr(TU), c.data_ptr(), tri.data_ptr(), n, D,
                                       out_L.data_ptr(), out_U.data_ptr(), current_stream(dL.device)), (dL, dU, TL, TU, c, tri, out_L, out_U))


def householder(w_0, vk, out=None):
    D = int(w_0.shape[0])
    if out is None:
        out = torch.empty(D, D, dtype=torch.float64, device=w_0.device)
    _launch("usf_householder_f64", (w_0.data_ptr(), vk.data_ptr(), int(vk.shape[0]), D, out.data_ptr(),
                                    current_stream(w_0.device)), (w_0, vk, out))
    return out


TILED_TRANSPOSE = True      # transposed images of a batch through usf_pack_weights_t_f32 (False: the per-element kernel)


class batch_jobs:
    """Defer the usf_pack_weight_f32 calls made inside the block and issue them as ONE usf_pack_weights_f32 launch
    per size class at exit (or at an explicit ``flush``).  Only for calls whose sources are ready when the batch is
    flushed and whose outputs nobody reads before that."""

    def __init__(self, device, defer_grads: bool = False):
        self.device = device
        self.jobs = []
        # defer_grads: usf_wgrad_f32 / usf_colsum_f32 calls of at most GRAD_JOB_MAX_ROWS rows are queued as well and leave
        # as ONE usf_grad_jobs_f32 launch ahead of the pack jobs (which may read their outputs).  The caller guarantees
        # that their operands stay untouched until the flush.
        self.defer_grads = bool(defer_grads)
        self.grad_jobs = []

    def __enter__(self):
        _tls.jobs.append(self)
        return self

    def __exit__(self, exc_type, *exc):
        _tls.jobs.pop()
        if exc_type is None:
            self.flush()
        return False

    def flush(self):
        self._flush_grads()
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        # size classes: transposed matrices (read through LDS tiles), other matrices, single rows (vectors) -- keeps the
        # grid of empty blocks small
        kind = lambda j: 2 if j.n_out <= 1 else (0 if (j.transpose & 1) and TILED_TRANSPOSE else 1)    # noqa: E731
        for cls, entry in ((0, "usf_pack_weights_t_f32"), (1, "usf_pack_weights_f32"), (2, "usf_pack_weights_f32")):
            part = [j for j in jobs if kind(j[0]) == cls]
            if not part:
                continue
            arr = (PackJob * len(part))(*[j[0] for j in part])
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            max_rows = max(j[0].n_out for j in part)
            max_cols = max(max(j[0].n_in, j[0].ld_planes if j[0].planes else 0) for j in part)
            _launch(entry, (table.data_ptr(), len(part), max_rows, max_cols, current_stream(self.device)),
                    (table, [j[1] for j in part]))


    def _flush_grads(self):
        gj, self.grad_jobs = self.grad_jobs, []
        if not gj:
            return
        block_job, first = [], 0
        for i, (j, _keep) in enumerate(gj):
            nb = ((j.N + 127) // 128) * ((j.K + 127) // 128) if j.A else (j.N + 63) // 64
            j.first_block = first
            block_job.extend([i] * nb)
            first += nb
        arr = (GradJob * len(gj))(*[j[0] for j in gj])
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        bj = torch.tensor(block_job, dtype=torch.int32).to(self.device)
        _launch("usf_grad_jobs_f32", (table.data_ptr(), bj.data_ptr(), first, current_stream(self.device)),
                (table, bj, [j[1] for j in gj]))


GRAD_JOB_MAX_ROWS = 256      # usf_grad_jobs_f32 runs one row range per job (usf_wgrad_f32 splits the batch above this)


def _defer_grad_job(M: int):
    """the open batch that takes a weight / bias gradient of M rows as a queued job, or None"""
    if _tls.jobs and _tls.jobs[-1].defer_grads and 0 < M <= GRAD_JOB_MAX_ROWS:
        return _tls.jobs[-1]
    return None


def flush_jobs() -> None:
    """issue what the innermost open batch has queued so far (a later job is about to read an earlier one's output)"""
    if _tls.jobs:
        _tls.jobs[-1].flush()


def pack_weight(src, out_idx, n_out, in_idx, n_in, *, W=None, ldw=0, planes=None, transpose=False, ld_src=None):
    """usf_pack_weight_f32: src fp64/fp32 2-D (or 1-D = one row) device tensor; W [n_out, ldw] fp32 and/or planes
    [3, n_out, ldp] bf16 (preallocated).  Inside a ``batch_jobs`` block the call is queued, not launched."""
    transpose = int(transpose) | (2 if (planes is not None and planes.dtype == torch.float16) else 0)   # bit 1: fp16x2 planes
    if src.dtype not in (torch.float32, torch.float64):
        raise ValueError("pack_weight: source must be fp32 or fp64")
    if ld_src is None:
        ld_src = src.shape[-1]
    ldp = planes.shape[2] if planes is not None else 0
    ps = planes.shape[1] * planes.shape[2] if planes is not None else 0
    if _tls.jobs and n_out > 0 and n_in > 0:
        j = PackJob(src.data_ptr(), ptr(out_idx), ptr(in_idx), ptr(W), ptr(planes), ld_src, n_out, n_in, ldw, ldp, ps,
                    int(src.dtype == torch.float32), int(transpose))
        _tls.jobs[-1].jobs.append((j, (src, out_idx, in_idx, W, planes)))
        return
    _launch("usf_pack_weight_f32", (src.data_ptr(), int(src.dtype == torch.float32), ld_src, int(transpose),
                                    ptr(out_idx), n_out, ptr(in_idx), n_in, ptr(W), ldw, ptr(planes), ldp, ps,
                                    current_stream(src.device)), (src, out_idx, in_idx, W, planes))


def matvec_f64(src, b, *, idx=None, n_out=None, alpha=1.0, out32=None, out64=None, ld_src=None):
    if n_out is None:
        n_out = src.shape[0]
    _launch("usf_matvec_f64", (src.data_ptr(), ld_src or src.shape[1], b.shape[0], ptr(idx), n_out, b.data_ptr(),
                               float(alpha), ptr(out32), ptr(out64), current_stream(src.device)), (src, b, idx, out32, out64))


# ---- training backward (usf_train.hip) ---------------------------------------------
_wg_ws = {}


def _workspace(device, floats: int) -> torch.Tensor:
    """grow-only fp32 scratch per device for the split reductions (wgrad / colsum partials)"""
    key = str(device)
    ws = _wg_ws.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
        _wg_ws[key] = ws
    return ws


def wgrad_bias_ok(M, N, K, ldy, lda, mode) -> bool:
    return bool(load().usf_wgrad_bias_ok(M, N, K, ldy, lda, mode))


def wgrad(Y, A, G, *, M, N, K, ldy, lda, ldg, y_off=0, a_off=0, g_off=0, alpha=1.0, beta=0.0, mode=0, defer=True, colsum=None,
          cs_alpha=1.0, cs_beta=0.0):
    """G[n,k] = alpha * sum_m Y[m,n] A[m,k] + beta * G (element offsets *_off into the fp32 tensors).  Inside a
    ``batch_jobs(defer_grads=True)`` block a small-batch call is queued (same arithmetic: see usf_grad_jobs_f32).
    colsum [N] (only where ``wgrad_bias_ok``): cs_alpha * sum_m Y[m,n] + cs_beta * colsum from the same pass (usf_wgrad_bias_f32)"""
    lib = load()
    if colsum is not None:
        ws = _workspace(Y.device, lib.usf_wgrad_workspace_floats(M, N, K))
        _launch("usf_wgrad_bias_f32", (Y.data_ptr() + 4 * y_off, ldy, A.data_ptr() + 4 * a_off, lda, M, N, K,
                                       G.data_ptr() + 4 * g_off, ldg, float(alpha), float(beta), int(mode), colsum.data_ptr(),
                                       float(cs_alpha), float(cs_beta), ws.data_ptr(), ws.numel(), current_stream(Y.device)),
                (Y, A, G, ws, colsum))
        return
    bj = _defer_grad_job(M) if defer else None
    if bj is not None and ldy % 4 == 0 and lda % 4 == 0 and (Y.data_ptr() + 4 * y_off) % 16 == 0 \
            and (A.data_ptr() + 4 * a_off) % 16 == 0:
        bj.grad_jobs.append((GradJob(Y.data_ptr() + 4 * y_off, A.data_ptr() + 4 * a_off, G.data_ptr() + 4 * g_off, ldy, lda,
                                     ldg, M, N, K, 0, float(alpha), float(beta)), (Y, A, G)))
        return
    need = lib.usf_wgrad_workspace_floats(M, N, K)
    ws = _workspace(Y.device, need)
    _launch("usf_wgrad_f32", (Y.data_ptr() + 4 * y_off, ldy, A.data_ptr() + 4 * a_off, lda, M, N, K,
                              G.data_ptr() + 4 * g_off, ldg, float(alpha), float(beta), int(mode), ws.data_ptr(),
                              ws.numel(), current_stream(Y.device)), (Y, A, G, ws))


def row_planes(M: int, cols: int, device) -> torch.Tensor:
    """[3, ceil32(M), ceil32(cols)] bf16: the row-major operand planes of usf_wgrad_planes_f32 (include/usflows_hip.h)"""
    return torch.zeros(3, -(-M // 32) * 32, -(-cols // 32) * 32, dtype=torch.bfloat16, device=device)


def split_planes(X, planes, *, M, N, ldx, x_off=0):
    """usf_split_planes_f32: planes[p, m, c] = plane p of X[m, c] (zeros beyond M / N)"""
    _launch("usf_split_planes_f32", (X.data_ptr() + 4 * x_off, ldx, M, N, planes.data_ptr(), planes.shape[2],
                                     planes.shape[1] * planes.shape[2], current_stream(X.device)), (X, planes))


def wgrad_planes_ok(M: int, N: int, K: int) -> bool:
    return bool(load().usf_wgrad_planes_ok(M, N, K))


def wgrad_planes(Yp, Ap, G, *, M, N, K, ldg, y_off=0, a_off=0, g_off=0, alpha=1.0, beta=0.0, colsum=None, cs_alpha=1.0,
                 cs_beta=0.0):
    """usf_wgrad_planes_f32: G[n,k] = alpha * sum_m Y[m, y_off+n] A[m, a_off+k] + beta * G from ``row_planes`` operands;
    colsum [N] (optional, K >= 64): cs_alpha * sum_m Y[m, y_off+n] + cs_beta * colsum -- the bias gradient from the same pass"""
    lib = load()
    need = lib.usf_wgrad_planes_workspace_floats(M, N, K)
    ws = _workspace(Yp.device, need)
    _launch("usf_wgrad_planes_f32", (Yp.data_ptr(), Yp.shape[2], Yp.shape[1] * Yp.shape[2], y_off, Ap.data_ptr(), Ap.shape[2],
                                     Ap.shape[1] * Ap.shape[2], a_off, M, N, K, G.data_ptr() + 4 * g_off, ldg, float(alpha),
                                     float(beta), ptr(colsum), float(cs_alpha), float(cs_beta), ws.data_ptr(), ws.numel(),
                                     current_stream(Yp.device)), (Yp, Ap, G, ws, colsum))


def wgrad_blocked(Yp, y_nkb, y_kb0, Ap, a_nkb, a_kb0, G, *, M, N, K, ldg, g_off=0, alpha=1.0, beta=0.0, colsum=None,
                  cs_alpha=1.0, cs_beta=0.0, queue=None, ws=None):
    """usf_wgrad_blocked_f32: G[n,k] = alpha * sum_m Y[m, 32 y_kb0 + n] A[m, 32 a_kb0 + k] + beta * G with both operands
    planes buffers of the planes pipeline (uint8 tensors; logical positions); colsum as for ``wgrad_planes``.
    queue (a list) + ws (a workspace of this call's own, >= wgrad_blocked_workspace floats): the multiply kernel is launched, the
    reduction is appended to the queue -- G / colsum are complete after ``wgrad_reduce_flush(queue, device)``"""
    lib = load()
    if queue is not None:
        job = WReduceJob()
        _launch("usf_wgrad_blocked_plan_f32", (Yp.data_ptr(), y_nkb, y_kb0, Ap.data_ptr(), a_nkb, a_kb0, M, N, K,
                                               G.data_ptr() + 4 * g_off, ldg, float(alpha), float(beta), ptr(colsum), float(cs_alpha),
                                               float(cs_beta), ws.data_ptr(), ws.numel(), C.addressof(job), current_stream(Yp.device)),
                (Yp, Ap, G, ws, colsum, job))
        queue.append((job, (G, ws, colsum)))
        return
    need = lib.usf_wgrad_planes_workspace_floats(M, N, K)
    ws = _workspace(Yp.device, need)
    _launch("usf_wgrad_blocked_f32", (Yp.data_ptr(), y_nkb, y_kb0, Ap.data_ptr(), a_nkb, a_kb0, M, N, K,
                                      G.data_ptr() + 4 * g_off, ldg, float(alpha), float(beta), ptr(colsum), float(cs_alpha),
                                      float(cs_beta), ws.data_ptr(), ws.numel(), current_stream(Yp.device)),
            (Yp, Ap, G, ws, colsum))


def wgrad_blocked_workspace(M: int, N: int, K: int) -> int:
    return int(load().usf_wgrad_planes_workspace_floats(M, N, K))


def wgrad_reduce_flush(queue, device) -> None:
    """ONE usf_wgrad_reduce_jobs_f32 launch for the queued reductions (taped like every launch: the job table is a device tensor
    built once -- the queued pointers are plan-owned buffers, the same on every replay)"""
    if not queue:
        return
    block_job, first = [], 0
    for i, (j, _keep) in enumerate(queue):
        j.first_block = first
        block_job.extend([i] * j.blocks)
        first += j.blocks
    raw = bytearray(bytes((WReduceJob * len(queue))(*[j for j, _ in queue])))
    raw += b"\0" * ((-len(raw)) % 16)
    off = len(raw)
    raw += struct.pack(f"<{len(block_job)}i", *block_job)
    table = torch.frombuffer(raw, dtype=torch.uint8).to(device)
    _launch("usf_wgrad_reduce_jobs_f32", (table.data_ptr(), table.data_ptr() + off, first, current_stream(device)),
            (table, [k for _, k in queue]))
    del queue[:]


def colsum(Y, out, *, M, N, ldy, y_off=0, alpha=1.0, beta=0.0):
    bj = _defer_grad_job(M)
    if bj is not None:
        bj.grad_jobs.append((GradJob(Y.data_ptr() + 4 * y_off, None, out.data_ptr(), ldy, 0, 0, M, N, 0, 0, float(alpha),
                                     float(beta)), (Y, out)))
        return
    ws = _workspace(Y.device, (M // 256 + M // 65536 + 4) * N)
    _launch("usf_colsum_f32", (Y.data_ptr() + 4 * y_off, ldy, M, N, out.data_ptr(), float(alpha), float(beta),
                               ws.data_ptr(), ws.numel(), current_stream(Y.device)), (Y, out, ws))


def act_grad(d, h, *, M, H, ldd, ldh, act, slope):
    _launch("usf_act_grad_f32", (d.data_ptr(), ldd, h.data_ptr(), ldh, M, H, act, float(slope),
                                 current_stream(d.device)), (d, h))


def base_logprob_grad(z, ldz, g_lp, M, D, base, loc, scale, g, ldg):
    _launch("usf_base_logprob_grad_f32", (z.data_ptr(), ldz, g_lp.data_ptr(), M, D, base, loc.data_ptr(),
                                          scale.data_ptr(), g.data_ptr(), ldg, current_stream(z.device)),
            (z, g_lp, loc, scale, g))


def base_param_grad(z, ldz, g_lp, M, D, base, loc, scale, out):
    """usf_base_param_grad_f32: out [2, D] = (sum_m g_lp[m] d/dloc, sum_m g_lp[m] d/dscale) of a Laplace / Normal base at z"""
    ws = _workspace(z.device, (M // 256 + M // 65536 + 6) * 2 * D)
    _launch("usf_base_param_grad_f32", (z.data_ptr(), ldz, g_lp.data_ptr(), M, D, base, loc.data_ptr(), scale.data_ptr(), out.data_ptr(),
                                        ws.data_ptr(), ws.numel(), current_stream(z.device)), (z, g_lp, loc, scale, out, ws))


def sophiag_step(chunks_dev: torch.Tensor, n_chunks: int, *, decay, beta1, rho_bs, lr, maximize=False) -> None:
    """usf_sophiag_step_f32 on a device table of usf_mt_chunk (see usflows_amd/sophia.py)"""
    _launch("usf_sophiag_step_f32", (chunks_dev.data_ptr(), n_chunks, decay, beta1, 1.0 - beta1, rho_bs, -lr,
                                     1 if maximize else 0, current_stream(chunks_dev.device)), keep=chunks_dev)


def sophiag_hessian(chunks_dev: torch.Tensor, n_chunks: int, *, beta2) -> None:
    _launch("usf_sophiag_hessian_f32", (chunks_dev.data_ptr(), n_chunks, beta2, 1.0 - beta2,
                                        current_stream(chunks_dev.device)), keep=chunks_dev)
