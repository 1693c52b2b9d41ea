"""ctypes binding of libppn.so (include/ppn.h).  Fails loudly when the library is missing:
there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPN_LIB", os.path.join(_HERE, "csrc", "libppn.so"))   # PPN_LIB: diagnostic builds

PPN_MAX_EDGES = 32
PPN_MAX_KP = 32
PPN_F32, PPN_BF16, PPN_F16, PPN_F16X3 = 0, 1, 2, 3
PPN_CONV_NO_FILTER_BANK, PPN_CONV_SHARED_GPU, PPN_CONV_OUT_BF16, PPN_CONV_X3_PLAIN_OUT = 1, 2, 4, 8   # ppn_conv_desc.flags
PPN_ACT_NONE, PPN_ACT_RELU, PPN_ACT_LRELU, PPN_ACT_SIGMOID = 0, 1, 2, 3
PPN_STEM_RAW_S2 = 1 << 16           # ppn_*stem012_dt dtype flag: out_raw holds only the even (row, column) pixels


class DecodeCfg(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("E", C.c_int32), ("sH", C.c_int32), ("sW", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("inH", C.c_int32), ("inW", C.c_int32),
        ("det_thr", C.c_float), ("nms_thr", C.c_float), ("min_kp", C.c_int32), ("max_humans", C.c_int32),
        ("edge_src", C.c_int32 * PPN_MAX_EDGES), ("edge_dst", C.c_int32 * PPN_MAX_EDGES),
        ("edge_order", C.c_int32 * PPN_MAX_EDGES),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("cin", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("dilation", C.c_int32), ("pad", C.c_int32),
        ("k_total", C.c_int32), ("cout_pad", C.c_int32), ("act1", C.c_int32), ("act2", C.c_int32),
        ("out_nchw_f32", C.c_int32),
        ("src", C.c_void_p), ("weight", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p),
        ("residual", C.c_void_p), ("out_raw", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p),
        ("out_act", C.c_void_p), ("zero_page", C.c_void_p),
        ("src2", C.c_void_p), ("in2_h", C.c_int32), ("in2_w", C.c_int32), ("cin2", C.c_int32), ("stride2", C.c_int32),
        ("unary_out", C.c_void_p), ("argmax_keys", C.c_void_p), ("unary_channels", C.c_int32),
        ("limb_window", C.c_int32), ("m_begin", C.c_int32), ("m_count", C.c_int32), ("limb_edge_pad", C.c_int32),
        ("flags", C.c_int32),
        ("prefetch", C.c_void_p), ("prefetch_bytes", C.c_int64),
        ("stats_partial", C.c_void_p), ("stats_mode", C.c_int32), ("stats_act", C.c_int32), ("stats_x", C.c_void_p),
        ("stats_gamma", C.c_void_p), ("stats_beta", C.c_void_p), ("stats_mean", C.c_void_p), ("stats_rstd", C.c_void_p),
        ("stats_tiles", C.POINTER(C.c_int32)),
    ]


class BlockDesc(C.Structure):
    """ppn_block_desc: a whole 64-channel stride-1 BasicBlock as one launch (csrc/block64.hip)."""
    _fields_ = [
        ("dtype", C.c_int32), ("batch", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("channels", C.c_int32),
        ("src", C.c_void_p), ("residual", C.c_void_p), ("weight1", C.c_void_p), ("scale_mid", C.c_void_p),
        ("shift_mid", C.c_void_p), ("act_mid", C.c_int32), ("weight2", C.c_void_p), ("scale1", C.c_void_p),
        ("shift1", C.c_void_p), ("act1", C.c_int32), ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("act2", C.c_int32),
        ("out_raw", C.c_void_p), ("out_act", C.c_void_p), ("flags", C.c_int32),
        ("stride", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("proj_src", C.c_void_p), ("proj_weight", C.c_void_p), ("proj_scale", C.c_void_p), ("proj_shift", C.c_void_p),
        ("w1_ld", C.c_int32), ("proj_ld", C.c_int32),
    ]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p)] + [
        (n, C.c_int32) for n in ("dtype", "cout", "cin", "ksize", "cout_pad", "k_total", "k_order", "k_step",
                                 "transposed", "reserved_")]


PPN_PACK_ITEM_BYTES = 64


class LossCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("K", "E", "sH", "sW", "H", "W", "inH", "inW")]


class BnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("channels", C.c_int32), ("pixels", C.c_int64), ("act", C.c_int32),
                ("eps", C.c_float), ("momentum", C.c_float)] + [
        (n, C.c_void_p) for n in ("x", "gamma", "beta", "running_mean", "running_var", "save_mean", "save_rstd",
                                  "scale", "shift", "y", "workspace")] + [("stats_blocks", C.c_int32),
                                                                          ("emit_blocks", C.POINTER(C.c_int32))]


class BnBwdDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("channels", C.c_int32), ("pixels", C.c_int64), ("act", C.c_int32)] + [
        (n, C.c_void_p) for n in ("x", "dy", "dx_add", "gamma", "beta", "save_mean", "save_rstd", "dgamma", "dbeta",
                                  "dx", "workspace")] + [("stats_blocks", C.c_int32)] + [
        (n, C.c_void_p) for n in ("next_x", "next_gamma", "next_beta", "next_mean", "next_rstd")] + [
        ("next_act", C.c_int32), ("next_blocks", C.POINTER(C.c_int32))]


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "batch", "in_h", "in_w", "cin", "out_h", "out_w", "cout", "ksize",
                                         "stride", "dilation", "pad")] + [
        ("beta", C.c_float), ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64)]


class PPNError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "ppn_last_error": (C.c_char_p, []),
    "ppn_version": (C.c_int, []),
    "ppn_decode_workspace_bytes": (C.c_size_t, [C.POINTER(DecodeCfg), C.c_int32]),
    "ppn_decode": (C.c_int, [C.POINTER(DecodeCfg), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_decode_fused": (C.c_int, [C.POINTER(DecodeCfg), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_decode_fused_workspace_bytes": (C.c_size_t, [C.POINTER(DecodeCfg), C.c_int32]),
    "ppn_decode_fused_ws": (C.c_int, [C.POINTER(DecodeCfg), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_limb_argmax": (C.c_int, [C.POINTER(DecodeCfg), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppn_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                          C.c_void_p]),
    "ppn_loss_workspace_bytes": (C.c_size_t, [C.POINTER(LossCfg), C.c_int32]),
    "ppn_loss_fwd_bwd": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_int32] + [C.c_void_p] * 10 +
                         [C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_loss_dz_workspace_bytes": (C.c_size_t, [C.POINTER(LossCfg), C.c_int32, C.c_int32]),
    "ppn_loss_fwd_bwd_dz": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_int32] + [C.c_void_p] * 10 +
                            [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "ppn_loss_fwd_bwd_dz_c": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_int32] + [C.c_void_p] * 9 +
                              [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "ppn_loss_fwd_bwd_dev": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_int32] + [C.c_void_p] * 10 +
                             [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_conv_tiling": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ppn_conv2d_fused": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "ppn_set_conv64_enabled": (C.c_int, [C.c_int32]),
    "ppn_basicblock64_fused": (C.c_int, [C.POINTER(BlockDesc), C.c_void_p]),
    "ppn_conv_split": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]),
    "ppn_stem7x7": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                    [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2 + [C.c_void_p, C.c_void_p]),
    "ppn_stem01": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 3 +
                   [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 5),
    "ppn_plan_add_stem01": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                            [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 4),
    "ppn_stem012": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 3 +
                    [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 11),
    "ppn_plan_add_stem012": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                             [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 10),
    "ppn_stem012_dt": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 3 +
                       [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 11),
    "ppn_plan_add_stem012_dt": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                                [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2 + [C.c_void_p] * 10),
    "ppn_plan_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ppn_plan_add_conv": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc)]),
    "ppn_plan_add_block": (C.c_int, [C.c_void_p, C.POINTER(BlockDesc)]),
    "ppn_plan_add_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ppn_plan_add_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "ppn_pack_weight_x3": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "ppn_split_f16x3": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppn_plan_add_stem": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                          [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2 + [C.c_void_p]),
    "ppn_plan_set_input": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppn_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppn_plan_run_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_int32]),
    "ppn_plan_size": (C.c_int, [C.c_void_p]),
    "ppn_plan_graph_captures": (C.c_int, [C.c_void_p]),
    "ppn_plan_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "ppn_plan_destroy": (C.c_int, [C.c_void_p]),
    "ppn_pack_weight": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppn_pack_weight_dgrad": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppn_pack_table_build": (C.c_int, [C.POINTER(PackItem), C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]),
    "ppn_pack_table_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
}

_SIGNATURES.update({
    "ppn_bn_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "ppn_bn_train_fwd": (C.c_int, [C.POINTER(BnDesc), C.c_void_p]),
    "ppn_bn_train_bwd": (C.c_int, [C.POINTER(BnBwdDesc), C.c_void_p]),
    "ppn_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "ppn_sumsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_gradnorm_probe_stats": (C.c_int, [C.c_void_p] * 5 + [C.POINTER(C.c_float), C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "ppn_gradnorm_weight_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                                           C.c_void_p, C.c_void_p]),
    "ppn_gradnorm_renorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ppn_set_conv_tile_policy": (C.c_int, [C.c_int32]),
    "ppn_set_conv_tile_override": (C.c_int, [C.c_int32, C.c_int32]),
    "ppn_last_conv_kernel": (C.c_char_p, []),
    "ppn_add_relu": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ppn_relu_mask": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ppn_upsample_zero": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p, C.c_void_p]),
    "ppn_interleave_parity": (C.c_int, [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    "ppn_interleave_parity_stacked": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    "ppn_image_to_nhwc": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppn_colsum": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_head_grad": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_loss_unary_bwd": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_int32] + [C.c_void_p] * 8 +
                           [C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_encode_targets": (C.c_int, [C.POINTER(LossCfg), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int32] + [C.c_void_p] * 11),
    "ppn_encode_targets_c": (C.c_int, [C.POINTER(LossCfg), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int32] + [C.c_void_p] * 12),
    "ppn_nchw_to_nhwc": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    "ppn_bn_act_mask": (C.c_int, [C.POINTER(BnBwdDesc), C.c_void_p]),
    "ppn_bn_dual_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "ppn_bn_dual_bwd": (C.c_int, [C.POINTER(BnBwdDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_bn_train_bwd_streams": (C.c_int, [C.POINTER(BnBwdDesc), C.c_int32, C.c_void_p]),
    "ppn_bn_act_mask_streams": (C.c_int, [C.POINTER(BnBwdDesc), C.c_int32, C.c_void_p]),
    "ppn_bn_dual_bwd_streams": (C.c_int, [C.POINTER(BnBwdDesc), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ppn_bn_dual_bwd_streams_sum": (C.c_int, [C.POINTER(BnBwdDesc), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ppn_loss_limb_dual_nhwc": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "ppn_loss_limb_dual_nhwc_c": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                            C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "ppn_loss_dual": (C.c_int, [C.POINTER(LossCfg), C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 10 +
                      [C.POINTER(C.c_float), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppn_conv_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "ppn_conv_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "ppn_ingest_frames": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_void_p]),
})

EXPORTS = tuple(_SIGNATURES)


def load():
    """Load libppn.so (built by pytorch_pose_proposal_network_amd.build); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PPNError(
            f"{LIB_PATH} is missing: build it with `python -m pytorch_pose_proposal_network_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  -- load PyTorch-ROCm's HIP runtime first: libppn.so binds to the same one
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in _SIGNATURES.items():
        if not hasattr(lib, name):
            missing.append(name)
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if missing and not os.environ.get("PPN_ALLOW_PARTIAL"):
        raise PPNError(f"{LIB_PATH} is stale, missing symbols {missing}: rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ppn_last_error()
        raise PPNError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def conv_tiling(dtype: int, cin: int, cout: int, ksize: int):
    """(k_step, cout_tile, k_order, k_total, cout_pad) for a conv of this shape."""
    ks, ct, ko = C.c_int32(), C.c_int32(), C.c_int32()
    check(load().ppn_conv_tiling(dtype, cin, cout, ksize, C.byref(ks), C.byref(ct), C.byref(ko)), "ppn_conv_tiling")
    kreal = ksize * ksize * cin
    ktot = (kreal + ks.value - 1) // ks.value * ks.value
    cpad = (cout + ct.value - 1) // ct.value * ct.value
    return ks.value, ct.value, ko.value, ktot, cpad


def current_stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
