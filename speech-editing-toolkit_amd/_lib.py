"""ctypes binding of libset_amd.so (include/set_amd.h) + the in-tree hipcc build.

The library is built in-tree (speech-editing-toolkit_amd/libset_amd.so) with
`hipcc --offload-arch=gfx950`; hipcc cross-compiles without a GPU.  The product
path fails loudly (RuntimeError) when the library cannot be loaded -- there is
no fallback implementation.
"""
import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_PATH = os.path.join(_PKG, "libset_amd.so")
# measurement only (tools/build_exp.sh): load an experimental build of the library instead; never built or rebuilt from here
_LIB_OVERRIDE = os.environ.get("SET_AMD_LIB")
SOURCES = ["conv1d.hip", "conv_x2.hip", "resblock_x2.hip", "glue.hip", "diffnet.hip", "diffnet_x3.hip", "train.hip", "attention.hip", "attention_fused.hip", "bf16.hip", "diffnet_bf16.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]

# constants mirrored from set_amd.h
OK, E_INVALID, E_UNSUPPORTED, E_LAUNCH = 0, -1, -2, -3
ACT = dict(none=0, relu=1, gelu=2, tanh=3, softplus=4, mish=5, lrelu=6)
PRO = dict(none=0, lrelu=1, div=2)
IMPL_NAIVE, IMPL_MFMA, IMPL_MFMA2, IMPL_BF16, IMPL_F16X2, IMPL_FEWOUT = 1, 2, 3, 4, 5, 6
DTYPE_F32, DTYPE_BF16, DTYPE_BF16_G16, DTYPE_BF16_G16_X16 = 0, 1, 2, 3

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)


class SetConv1dArgs(C.Structure):
    _fields_ = [
        ("inp", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("mask", C.c_void_p), ("in_chan_add", C.c_void_p), ("out", C.c_void_p),
        ("in_bs", C.c_int64), ("in_cs", C.c_int64), ("out_bs", C.c_int64), ("out_cs", C.c_int64),
        ("res_bs", C.c_int64), ("res_cs", C.c_int64),
        ("w_base", C.c_int64), ("w_sco", C.c_int64), ("w_sci", C.c_int64), ("w_stap", C.c_int64),
        ("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("K", C.c_int32), ("dil", C.c_int32),
        ("pad", C.c_int32),
        ("T_in", C.c_int32), ("T_iter", C.c_int32), ("T_out", C.c_int32), ("out_stride", C.c_int32),
        ("out_off", C.c_int32),
        ("pro", C.c_int32), ("act", C.c_int32), ("accumulate", C.c_int32), ("impl", C.c_int32),
        ("pro_param", C.c_float), ("act_param", C.c_float), ("alpha", C.c_float), ("out_div", C.c_float),
    ]


class SetDiffnetLayerArgs(C.Structure):
    _fields_ = [
        ("x_in", C.c_void_p), ("condproj", C.c_void_p), ("dstep", C.c_void_p), ("w1p", C.c_void_p),
        ("b_dil", C.c_void_p), ("w2p", C.c_void_p), ("b_out", C.c_void_p), ("x_out", C.c_void_p),
        ("skip", C.c_void_p),
        ("cp_bs", C.c_int64), ("d_bs", C.c_int64), ("d_cs", C.c_int64),
        ("B", C.c_int32), ("T", C.c_int32), ("dil", C.c_int32), ("first", C.c_int32),
        ("dbg_clock", C.c_void_p),
    ]


class SetDiffLoopArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("M", C.c_int32), ("L", C.c_int32), ("steps", C.c_int32),
        ("dilation_cycle_length", C.c_int32),
        ("x", C.c_void_p), ("noise", C.c_void_p), ("seed", C.c_uint64),
        ("condproj", C.c_void_p), ("dstep", C.c_void_p), ("coef4", C.c_void_p),
        ("w_in_p", C.c_void_p), ("b_in", C.c_void_p),
        ("w1p_all", C.c_void_p), ("w2p_all", C.c_void_p), ("b_dil_all", C.c_void_p), ("b_out_all", C.c_void_p),
        ("w1w_all", C.c_void_p), ("w2w_all", C.c_void_p),
        ("w1s_all", C.c_void_p), ("w2s_all", C.c_void_p), ("z_ws", C.c_void_p), ("wx3_all", C.c_void_p),
        ("x3_mode", C.c_int32),
        ("w_skip_x2", C.c_void_p), ("w_outp_x2", C.c_void_p), ("w_in_x2", C.c_void_p),
        ("w_skip_p", C.c_void_p), ("b_skip", C.c_void_p), ("w_outp_p", C.c_void_p), ("b_outp", C.c_void_p),
        ("ws_x0", C.c_void_p), ("ws_x1", C.c_void_p), ("ws_skip", C.c_void_p), ("ws_h", C.c_void_p),
        ("ws_x0pred", C.c_void_p),
        ("layer_span_ms", C.POINTER(C.c_float)),
        ("loop_ms", C.POINTER(C.c_float)),
        ("n_groups", C.c_int32),
        ("persistent", C.c_int32),
        ("sync_ws", C.c_void_p),
        ("err_flag", C.c_void_p),
        ("cond", C.c_void_p), ("img16_all", C.c_void_p), ("b_cond_all", C.c_void_p),
        ("bf16_ws", C.c_void_p), ("bf16_ws_floats", C.c_int64),
    ]


class SetDiffnetStackArgs(C.Structure):
    _fields_ = [
        ("xa", C.c_void_p), ("xb", C.c_void_p), ("skip", C.c_void_p), ("condproj", C.c_void_p), ("dstep", C.c_void_p),
        ("w1p_all", C.c_void_p), ("w2p_all", C.c_void_p), ("b_dil_all", C.c_void_p), ("b_out_all", C.c_void_p),
        ("sync_ws", C.c_void_p),
        ("cp_bs", C.c_int64), ("cp_ls", C.c_int64), ("d_bs", C.c_int64), ("d_cs", C.c_int64), ("d_ls", C.c_int64),
        ("B", C.c_int32), ("T", C.c_int32), ("L", C.c_int32), ("dilation_cycle_length", C.c_int32),
        ("w1w_all", C.c_void_p), ("w2w_all", C.c_void_p),
        ("x_all", C.c_void_p), ("save_y", C.c_void_p), ("save_z", C.c_void_p),
        ("err_flag", C.c_void_p),
        ("w1s_all", C.c_void_p), ("w2s_all", C.c_void_p), ("z_ws", C.c_void_p), ("wx3_all", C.c_void_p),
        ("x3_mode", C.c_int32),
    ]


class SetDiffnetLayerBf16Args(C.Structure):
    _fields_ = [
        ("x_in", C.c_void_p), ("x_out", C.c_void_p), ("skip", C.c_void_p), ("cond", C.c_void_p), ("dstep", C.c_void_p),
        ("img", C.c_void_p), ("b_dil", C.c_void_p), ("b_cond", C.c_void_p), ("b_out", C.c_void_p),
        ("y16", C.c_void_p), ("z16", C.c_void_p),
        ("d_bs", C.c_int64), ("d_cs", C.c_int64),
        ("B", C.c_int32), ("T", C.c_int32), ("dil", C.c_int32), ("first", C.c_int32),
    ]


class SetDiffnetLayersBf16Args(C.Structure):
    _fields_ = [
        ("x_in", C.c_void_p), ("x_out", C.c_void_p), ("skip", C.c_void_p), ("cond", C.c_void_p), ("dstep", C.c_void_p),
        ("img", C.c_void_p), ("b_dil", C.c_void_p), ("b_cond", C.c_void_p), ("b_out", C.c_void_p),
        ("scratch", C.c_void_p), ("scratch_floats", C.c_int64),
        ("d_bs", C.c_int64), ("d_cs", C.c_int64), ("d_ls", C.c_int64),
        ("B", C.c_int32), ("T", C.c_int32), ("l0", C.c_int32), ("nl", C.c_int32), ("dilation_cycle_length", C.c_int32),
        ("first", C.c_int32),
    ]


class SetDiffnetLayerBf16BwdArgs(C.Structure):
    _fields_ = [
        ("dx_out", C.c_void_p), ("dskip", C.c_void_p), ("y16", C.c_void_p), ("img", C.c_void_p),
        ("dx", C.c_void_p), ("dy16", C.c_void_p), ("do16", C.c_void_p), ("dcond", C.c_void_p),
        ("part_dbo", C.c_void_p), ("part_dby", C.c_void_p), ("part_dd", C.c_void_p),
        ("B", C.c_int32), ("T", C.c_int32), ("dil", C.c_int32), ("dcond_first", C.c_int32),
    ]


class SetResblockPairArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("out", C.c_void_p),
        ("x_bs", C.c_int64), ("x_cs", C.c_int64), ("out_bs", C.c_int64), ("out_cs", C.c_int64),
        ("B", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("dil", C.c_int32), ("T", C.c_int32),
        ("accumulate", C.c_int32),
        ("slope", C.c_float), ("out_div", C.c_float),
    ]


class SetPackBf16Desc(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("wp", C.c_void_p),
        ("w_base", C.c_int64), ("w_sco", C.c_int64), ("w_sci", C.c_int64), ("w_stap", C.c_int64), ("start", C.c_int64),
        ("Cout", C.c_int32), ("Cin", C.c_int32), ("K", C.c_int32), ("CoutP", C.c_int32), ("CinP", C.c_int32), ("pad_", C.c_int32),
    ]


class SetPackF32Desc(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("wp", C.c_void_p),
        ("w_base", C.c_int64), ("w_sco", C.c_int64), ("w_sci", C.c_int64), ("w_stap", C.c_int64), ("start", C.c_int64),
        ("Cout", C.c_int32), ("Cin", C.c_int32), ("K", C.c_int32), ("CinP", C.c_int32), ("kind", C.c_int32), ("RB", C.c_int32),
        ("ch_max", C.c_int32), ("pad_", C.c_int32),
    ]


class SetBmmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("a_bo", C.c_int64), ("a_bi", C.c_int64), ("a_ms", C.c_int64), ("a_ks", C.c_int64),
        ("b_bo", C.c_int64), ("b_bi", C.c_int64), ("b_ks", C.c_int64), ("b_ns", C.c_int64),
        ("c_bo", C.c_int64), ("c_bi", C.c_int64), ("c_ms", C.c_int64), ("c_ns", C.c_int64),
        ("n_outer", C.c_int32), ("n_inner", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("alpha", C.c_float), ("accumulate", C.c_int32),
    ]


class SetAttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p), ("p", C.c_void_p),
        ("kpm", C.c_void_p),
        ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64), ("o_bs", C.c_int64),
        ("q_cs", C.c_int32), ("k_cs", C.c_int32), ("v_cs", C.c_int32), ("o_cs", C.c_int32),
        ("B", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
        ("scale", C.c_float), ("fill", C.c_float),
        ("bf16", C.c_int32),
    ]


class SetAttnBwdArgs(C.Structure):
    _fields_ = [
        ("fwd", SetAttnArgs),
        ("d_o", C.c_void_p), ("delta", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("dq_bs", C.c_int64), ("dk_bs", C.c_int64), ("dv_bs", C.c_int64),
        ("dq_cs", C.c_int32), ("dk_cs", C.c_int32), ("dv_cs", C.c_int32),
    ]


_V, _I32, _I64, _U64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float

# name -> (restype, argtypes); every symbol include/set_amd.h declares
SIGNATURES = {
    "set_abi_version": (C.c_int, []),
    "set_last_error": (C.c_char_p, []),
    "set_conv1d": (C.c_int, [C.POINTER(SetConv1dArgs), _V]),
    "set_packed_conv_weight_size": (_I64, [_I32, _I32, _I32]),
    "set_pack_conv_weight": (C.c_int, [_V, _V, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _V]),
    "set_packed_conv_weight_v2_size": (_I64, [_I32, _I32, _I32]),
    "set_pack_conv_weight_v2": (C.c_int, [_V, _V, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _V]),
    "set_weight_norm_fold": (C.c_int, [_V, _V, _V, _I32, _I64, _V]),
    "set_layernorm_ch": (C.c_int, [_V, _V, _V, _V, _V, _I32, _I32, _I32, _F, _V]),
    "set_embedding_bct": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _F, _I32, _V]),
    "set_embedding_bct_dev_scale": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _V, _I32, _V]),
    "set_abs_sum_mask": (C.c_int, [_V, _V, _I32, _I32, _I32, _V]),
    "set_index_mask": (C.c_int, [_V, _V, _I64, _V]),
    "set_expand_states": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _V]),
    "set_add_chan_mask": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_masked_dur": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_pitch_coarse": (C.c_int, [_V, _V, _V, _V, _I32, _V, _V, _I64, _V]),
    "set_transpose_btc_to_bct": (C.c_int, [_V, _V, _I32, _I32, _I32, _V]),
    "set_transpose_bct_to_btc": (C.c_int, [_V, _V, _I32, _I32, _I32, _V]),
    "set_sum_scale": (C.c_int, [_V, _V, _V, _V, _F, _I64, _V]),
    "set_blend_mask": (C.c_int, [_V, _V, _V, _V, _I64, _I64, _V]),
    "set_mul_one_minus_mask": (C.c_int, [_V, _V, _V, _I64, _I64, _V]),
    "set_dur_total": (C.c_int, [_V, _V, _V, _I32, _I32, _V]),
    "set_length_regulate": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _V]),
    "set_sinusoid_embed": (C.c_int, [_V, _V, _I32, _I32, _V]),
    "set_gate": (C.c_int, [_V, _V, _I32, _I32, _I32, _V]),
    "set_res_skip": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _I32, _V]),
    "set_diffnet_layer": (C.c_int, [C.POINTER(SetDiffnetLayerArgs), _V]),
    "set_diffnet_w1p_size": (_I64, []),
    "set_diffnet_w2p_size": (_I64, []),
    "set_pack_diffnet_layer": (C.c_int, [_V, _V, _V, _V, _V]),
    "set_pack_diffnet_layers": (C.c_int, [_V, _V, _I64, _I64, _V, _V, _V, _V, _I32, _V]),
    "set_diffnet_stack_variant": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "set_diffnet_stack_x3_winograd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "set_packed_conv_weight_x2_size": (C.c_int64, [_I32, _I32, _I32]),
    "set_pack_conv_weight_x2": (C.c_int, [_V, _V, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I32, _V]),
    "set_conv_x2_range_flag": (C.c_int, [C.POINTER(C.c_int32), _I32]),
    "set_sizeof_resblock_pair_args": (_I64, []),
    "set_resblock_pair_x2_supported": (C.c_int, [_I32, _I32, _I32, _I32]),
    "set_resblock_pair_x2": (C.c_int, [C.POINTER(SetResblockPairArgs), _V]),
    "set_packed_conv_transpose_x2_size": (C.c_int64, [_I32, _I32, _I32, _I32]),
    "set_pack_conv_transpose_x2": (C.c_int, [_V, _V, _I32, _I32, _I32, _I32, _I32, _V]),
    "set_conv_transpose1d_x2": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _V]),
    "set_diffnet_layer_x3_image_size": (C.c_int64, [_I32]),
    "set_pack_diffnet_layer_x3": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _V]),
    "set_diffnet_w1w_size": (_I64, []),
    "set_pack_diffnet_layer_wino": (C.c_int, [_V, _V, _V, _V, _V]),
    "set_diffnet_stack": (C.c_int, [C.POINTER(SetDiffnetStackArgs), _V]),
    "set_sizeof_diffnet_stack_args": (_I64, []),
    "set_posterior_step": (C.c_int, [_V, _V, _V, _V, _I64, _V, _I32, _I64, _U64, _U64, _V]),
    "set_q_sample": (C.c_int, [_V, _V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_randn": (C.c_int, [_V, _I64, _U64, _U64, _V]),
    "set_rng_seed_delta": (C.c_int, [_V]),
    "set_stream_order": (C.c_int, [_V, _V, _I32]),
    "set_stream_mark": (C.c_int, [_V, _I32, _I32]),
    "set_stream_mark_done": (C.c_int, [_I32]),
    "set_stream_create_low_priority": (C.c_int, [C.POINTER(C.c_void_p)]),
    "set_diffusion_loop": (C.c_int, [C.POINTER(SetDiffLoopArgs), _V]),
    "set_selftest_mfma": (C.c_int, [C.POINTER(C.c_float), _V]),
    "set_sizeof_attn_args": (_I64, []),
    "set_sizeof_attn_bwd_args": (_I64, []),
    "set_attention": (C.c_int, [C.POINTER(SetAttnArgs), _V]),
    "set_attention_bwd": (C.c_int, [C.POINTER(SetAttnBwdArgs), _V]),
    "set_sizeof_bmm_args": (_I64, []),
    "set_bmm": (C.c_int, [C.POINTER(SetBmmArgs), _V]),
    "set_softmax_rows": (C.c_int, [_V, _V, _V, _I64, _I32, _I64, _F, _V]),
    "set_softmax_rows_bwd": (C.c_int, [_V, _V, _V, _I64, _I32, _V]),
    "set_make_positions": (C.c_int, [_V, _V, _I64, _V, _I32, _I32, _V]),
    "set_head_mean": (C.c_int, [_V, _V, _I32, _I32, _I64, _V]),
    "set_mask_fill_chan": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_masked_channel_sum": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _V]),
    "set_conv1d_wgrad": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _V]),
    "set_conv1d_wgrad_scratch_floats": (_I64, [_I32, _I32, _I32, _I32, _I32, _I32]),
    "set_conv1d_wgrad_det": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _V, _I64, _V]),
    "set_conv1d_wgrad_grouped_scratch_floats": (_I64, [_I32, _I32, _I32, _I32, _I32, _I32]),
    "set_conv1d_wgrad_det_grouped": (C.c_int, [_V, _V, _V, _V, _I32, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                                               _I32, _V, _I64, _V]),
    "set_packed_conv_weight_bf16_size": (_I64, [_I32, _I32, _I32]),
    "set_pack_conv_weight_bf16": (C.c_int, [_V, _V, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _V]),
    "set_sizeof_pack_bf16_desc": (_I64, []),
    "set_pack_conv_weights_bf16_batch": (C.c_int, [_V, _I32, _I64, _V]),
    "set_sizeof_pack_f32_desc": (_I64, []),
    "set_fill_pack_f32_desc": (_I64, [C.POINTER(SetPackF32Desc), _I32, _I32]),
    "set_pack_conv_weights_f32_batch": (C.c_int, [_V, _I32, _I64, _V]),
    "set_channel_sum_det": (C.c_int, [_V, _V, _I32, _I32, _I32, _V, _V]),
    "set_weighted_sum_det": (C.c_int, [_V, _V, _V, _I64, _I64, _V, _V]),
    "set_sumsq_det": (C.c_int, [_V, _V, _I64, _V, _V]),
    "set_dur_loss_sums_det": (C.c_int, [_V, _V, _V, _V, _V, _I32, _I32, _I32, _I32, _V, _V]),
    "set_pitch_loss_sums_det": (C.c_int, [_V, _V, _V, _V, _V, _I32, _I32, _V, _V]),
    "set_scatter_rows_segments": (_I32, [_I32]),
    "set_scatter_rows_det": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _F, _I32, _I32, _V, _V]),
    "set_sizeof_diffnet_layer_bf16_args": (_I64, []),
    "set_diffnet_layer_bf16_image_size": (_I64, []),
    "set_pack_diffnet_layer_bf16": (C.c_int, [_V, _V, _V, _V, _V]),
    "set_pack_diffnet_layers_bf16": (C.c_int, [_V, _V, _V, _I64, _I64, _I64, _V, _I32, _V]),
    "set_diffnet_layer_fwd_bf16": (C.c_int, [C.POINTER(SetDiffnetLayerBf16Args), _V]),
    "set_sizeof_diffnet_layers_bf16_args": (_I64, []),
    "set_diffnet_layers_bf16_scratch_floats": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "set_diffnet_layers_fwd_bf16": (C.c_int, [C.POINTER(SetDiffnetLayersBf16Args), _V]),
    "set_diffnet_layers_bf16_plan": (_I32, [_I32, _I32, _I32, _I32]),
    "set_debug_bf16_phase_buffer": (C.c_int, [_V]),
    "set_debug_split_phase_buffer": (C.c_int, [_V]),
    "set_debug_x3_phase_buffer": (C.c_int, [_V]),
    "set_debug_resblock_phase_buffer": (C.c_int, [_V]),
    "set_sizeof_diffnet_layer_bf16_bwd_args": (_I64, []),
    "set_diffnet_layer_bwd_bf16_tiles": (_I32, [_I32, _I32]),
    "set_diffnet_layer_bwd_bf16": (C.c_int, [C.POINTER(SetDiffnetLayerBf16BwdArgs), _V]),
    "set_partial_rows_sum": (C.c_int, [_V, _V, _I32, _I32, _I32, _I32, _F, _V]),
    "set_step_proj_fwd": (C.c_int, [_V, _V, _I64, _V, _I64, _V, _I32, _I32, _I32, _V]),
    "set_step_proj_bwd_scratch_floats": (_I64, [_I32, _I32, _I32]),
    "set_step_proj_bwd": (C.c_int, [_V, _V, _V, _I64, _V, _V, _I64, _V, _I64, _V, _I32, _I32, _I32, _V]),
    "set_step_proj_bwd_dh": (C.c_int, [_V, _V, _I64, _V, _V, _I32, _I32, _I32, _V]),
    "set_step_proj_bwd_dw": (C.c_int, [_V, _V, _V, _I64, _V, _I64, _I32, _I32, _I32, _V]),
    "set_diffnet_layer_bwd_reduce": (C.c_int, [_V, _V, _V, _I32, _I32, _V, _V, _V, _V, _I64, _V]),
    "set_diffnet_layers_bwd_reduce": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _V, _I64, _V, _I64, _V, _I64, _V, _I64, _I64, _V]),
    "set_channel_sum": (C.c_int, [_V, _V, _I32, _I32, _I32, _V]),
    "set_row_sum": (C.c_int, [_V, _V, _I64, _I32, _F, _V]),
    "set_conv_epilogue_bwd": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _I32, _F, _V]),
    "set_act_fwd": (C.c_int, [_V, _V, _I64, _I32, _F, _V]),
    "set_act_bwd": (C.c_int, [_V, _V, _V, _I64, _I32, _F, _V]),
    "set_act_bwd_scaled": (C.c_int, [_V, _V, _V, _I64, _I32, _F, _F, _V]),
    "set_gate_bwd": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _V]),
    "set_res_skip_bwd": (C.c_int, [_V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_layernorm_ch_bwd": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, _I32, _I32, _I32, _F, _V]),
    "set_layernorm_ch_bwd_add": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, _V, _I32, _I32, _I32, _F, _V]),
    "set_layernorm_ch_bwd_scratch": (C.c_int64, [_I32, _I32, _I32]),
    "set_embedding_bwd": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _F, _I32, _V]),
    "set_expand_states_bwd": (C.c_int, [_V, _V, _V, _I32, _I32, _I32, _I32, _V]),
    "set_dropout": (C.c_int, [_V, _V, _I64, _F, _U64, _U64, _V]),
    "set_frame_weight": (C.c_int, [_V, _V, _I64, _I32, _V]),
    "set_weighted_sum": (C.c_int, [_V, _V, _V, _I64, _I64, _V]),
    "set_l1_elem": (C.c_int, [_V, _V, _V, _V, _I64, _V]),
    "set_scale_bcast": (C.c_int, [_V, _V, _V, _I64, _I64, _V, _F, _V]),
    "set_ssim_filter": (C.c_int, [_V, _V, _F, _V, _V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_ssim_map": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, _V, _I64, _V]),
    "set_ssim_bwd": (C.c_int, [_V, _V, _F, _V, _V, _V, _V, _I32, _I32, _I32, _V]),
    "set_dur_loss": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _I32, _I32, _I32, _I32, _F, _F, _F, _V]),
    "set_pitch_loss": (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _I32, _I32, _F, _F, _F, _V]),
    "set_sumsq": (C.c_int, [_V, _V, _I64, _V]),
    "set_adamw": (C.c_int, [_V, _V, _V, _V, _I64, _F, _F, _F, _F, _F, _I32, _V, _F, _F, _V]),
    "set_adamw_hyper": (C.c_int, [_F, _F, _I32, C.POINTER(C.c_float)]),
    "set_adamw_dev": (C.c_int, [_V, _V, _V, _V, _I64, _V, _F, _F, _F, _F, _V, _F, _F, _V]),
    "set_sizeof_conv1d_args": (_I64, []),
    "set_sizeof_diffnet_layer_args": (_I64, []),
    "set_sizeof_diff_loop_args": (_I64, []),
}

_lib = None


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] \
        + [os.path.join(INCLUDE, "set_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile libset_amd.so for gfx950 in-tree (no GPU needed)."""
    if _LIB_OVERRIDE:
        return _LIB_OVERRIDE
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libset_amd.so")
    # one builder at a time: `torchrun --nproc-per-node 8 bench.py` calls this from every rank
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():  # another process built it while we waited
                return LIB_PATH
            tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
            # one object per source under build/obj (only stale ones are recompiled, all of them in parallel), then one link
            objdir = os.path.join(_ROOT, "build", "obj")
            os.makedirs(objdir, exist_ok=True)
            hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "set_amd.h")]
            t_hdr = max(os.path.getmtime(h) for h in hdrs)
            cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
            jobs = []
            for f in SOURCES:
                src, obj = os.path.join(CSRC, f), os.path.join(objdir, f[:-4] + ".o")
                if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
                    jobs.append([hipcc] + cflags + ["-I", INCLUDE, "-c", src, "-o", obj])
            if verbose:
                for j in jobs:
                    print(" ".join(j))
            procs = [subprocess.Popen(j, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for j in jobs]
            outs = [(p.communicate()[0], p.returncode) for p in procs]
            bad = [o for o, rc in outs if rc != 0]
            if bad:
                raise RuntimeError("hipcc failed:\n" + "\n".join(bad))
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, f[:-4] + ".o") for f in SOURCES] + ["-o", tmp]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc link failed:\n" + r.stdout)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def lib():
    """The loaded library (loads on first use; raises if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _LIB_OVERRIDE or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "libset_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU/eager fallback for this path." % path)
    # Bring torch's HIP runtime up BEFORE the library is mapped: torch ships its own libamdhip64 and the library links
    # /opt/rocm's; mapped first, the library binds a second runtime copy that later sees "no ROCm-capable device"
    # (observed when build() and smoke() run in one process).  No-op on a machine without a GPU.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    assert L.set_sizeof_conv1d_args() == C.sizeof(SetConv1dArgs), "SetConv1dArgs ABI mismatch"
    assert L.set_sizeof_diffnet_layer_args() == C.sizeof(SetDiffnetLayerArgs), "SetDiffnetLayerArgs ABI mismatch"
    assert L.set_sizeof_diff_loop_args() == C.sizeof(SetDiffLoopArgs), "SetDiffLoopArgs ABI mismatch"
    assert L.set_sizeof_diffnet_stack_args() == C.sizeof(SetDiffnetStackArgs), "SetDiffnetStackArgs ABI mismatch"
    assert L.set_sizeof_bmm_args() == C.sizeof(SetBmmArgs), "SetBmmArgs ABI mismatch"
    assert L.set_sizeof_resblock_pair_args() == C.sizeof(SetResblockPairArgs), "SetResblockPairArgs ABI mismatch"
    assert L.set_sizeof_attn_args() == C.sizeof(SetAttnArgs), "SetAttnArgs ABI mismatch"
    assert L.set_sizeof_pack_bf16_desc() == C.sizeof(SetPackBf16Desc), "SetPackBf16Desc ABI mismatch"
    assert L.set_sizeof_pack_f32_desc() == C.sizeof(SetPackF32Desc), "SetPackF32Desc ABI mismatch"
    assert L.set_sizeof_attn_bwd_args() == C.sizeof(SetAttnBwdArgs), "SetAttnBwdArgs ABI mismatch"
    assert L.set_sizeof_diffnet_layer_bf16_args() == C.sizeof(SetDiffnetLayerBf16Args), "SetDiffnetLayerBf16Args ABI mismatch"
    assert L.set_sizeof_diffnet_layers_bf16_args() == C.sizeof(SetDiffnetLayersBf16Args), "SetDiffnetLayersBf16Args ABI mismatch"
    assert L.set_sizeof_diffnet_layer_bf16_bwd_args() == C.sizeof(SetDiffnetLayerBf16BwdArgs), "SetDiffnetLayerBf16BwdArgs ABI mismatch"
    _lib = L
    return L


class SetAmdError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != OK:
        msg = lib().set_last_error()
        raise SetAmdError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
