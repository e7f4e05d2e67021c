"""ctypes signatures of the layer kernels declared in include/otgan_layers.h."""
import ctypes

c_fp = ctypes.c_void_p
c_int, c_long, c_float, c_size_t = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t
c_double = ctypes.c_double


class ConvDesc(ctypes.Structure):
    """Mirror of `otgan_conv_desc` (include/otgan_layers.h)."""
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("ldx", c_int),
                ("upsample", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int),
                ("Cout", c_int), ("ldy", c_int), ("y_coff", c_int), ("preact", c_int),
                ("list_quads", c_int), ("x_amax", ctypes.c_void_p), ("dy_amax", ctypes.c_void_p),
                ("y_accumulate", c_int), ("x_operand", ctypes.c_void_p),
                ("y_amax_out", ctypes.c_void_p), ("dx_amax_out", ctypes.c_void_p), ("w_amax", ctypes.c_void_p),
                ("x_amax_count", c_int), ("list_width", c_int),
                ("glu_out", ctypes.c_void_p), ("glu_amax_out", ctypes.c_void_p), ("dy_amax_count", c_int)]


P_DESC = ctypes.POINTER(ConvDesc)


class Dense16BwdPair(ctypes.Structure):
    """Mirror of `otgan_dense16_bwd_pair`."""
    _fields_ = [("w", ctypes.c_void_p), ("fwd_filters", ctypes.c_void_p), ("filters", ctypes.c_void_p),
                ("nslices_src", c_int), ("slice_index", c_int), ("pair_index", c_int)]


class WnFwdLayer(ctypes.Structure):
    """Mirror of `otgan_wn_fwd_layer`."""
    _fields_ = [("V", c_fp), ("g", c_fp), ("w", c_fp), ("wT", c_fp), ("inv", c_fp), ("K", c_int)]


class WnPart(ctypes.Structure):
    """Mirror of `otgan_wn_part`."""
    _fields_ = [("p", c_fp), ("perm", c_fp), ("nrows", c_int), ("rstride", c_int)]


class WnBwdLayer(ctypes.Structure):
    """Mirror of `otgan_wn_bwd_layer`."""
    _fields_ = [("V", c_fp), ("g", c_fp), ("inv", c_fp), ("dV", c_fp), ("dg", c_fp), ("Ceff", c_int), ("taps", c_int),
                ("part", WnPart * 3)]


SIGNATURES = {
    "otgan_conv2d_workspace_bytes": (c_size_t, [P_DESC, c_int]),
    "otgan_conv2d_operand_bytes": (c_size_t, [P_DESC]),
    "otgan_conv2d_glu_fused": (c_int, [P_DESC]),
    "otgan_absmax_f32": (c_int, [c_fp, c_long, c_int, c_long, c_fp, c_fp]),
    "otgan_conv2d_amax_fused": (c_int, [P_DESC, c_int]),
    "otgan_conv2d_folded_weight_elems": (c_size_t, [P_DESC]),
    "otgan_conv2d_fold_weights_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp]),
    "otgan_conv2d_fwd_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_conv2d_dgrad_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp,
                                       c_size_t, c_fp]),
    "otgan_conv2d_wgrad_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_conv2d_filter_bytes": (c_size_t, [P_DESC, c_int]),
    "otgan_dense16_filter_bytes": (c_size_t, [c_int]),
    "otgan_dense16_prepare_filters_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp]),
    "otgan_dense16_h2_ok": (c_int, [P_DESC]),
    "otgan_dense16_chain_fwd_f32": (c_int, [c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp]),
    "otgan_dense16_chain_bwd_f32": (c_int, [c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp]),
    "otgan_dense16_bwd_filter_bytes": (c_size_t, [c_int]),
    "otgan_dense16_prepare_bwd_filters_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp]),
    "otgan_dense16_bwd_slice_f32": (c_int, [c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_int,
                                            c_fp, c_int, c_fp, c_fp]),
    "otgan_conv2d_prepare_filters_f32": (c_int, [P_DESC, c_int, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_conv2d_fwd_pf_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size_t, c_fp]),
    "otgan_conv2d_dgrad_pf_f32": (c_int, [P_DESC, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp,
                                          c_size_t, c_fp]),
    "otgan_weightnorm_fwd_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp]),
    "otgan_weightnorm_fwd_amax_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "otgan_weightnorm_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp]),
    "otgan_weightnorm_fwd_batched16_f32": (c_int, [ctypes.POINTER(WnFwdLayer), c_int, c_fp]),
    "otgan_weightnorm_bwd_batched16_f32": (c_int, [ctypes.POINTER(WnBwdLayer), c_int, c_fp]),
    "otgan_colsum_f32": (c_int, [c_fp, c_long, c_int, c_long, c_fp, c_fp, c_fp]),
    "otgan_glu_fwd_f32": (c_int, [c_fp, c_long, c_int, c_fp, c_fp]),
    "otgan_glu_bwd_f32": (c_int, [c_fp, c_fp, c_long, c_int, c_fp, c_fp]),
    "otgan_glu_fwd_amax_f32": (c_int, [c_fp, c_long, c_int, c_fp, c_fp, c_fp]),
    "otgan_glu_bwd_amax_f32": (c_int, [c_fp, c_fp, c_long, c_int, c_fp, c_fp, c_fp]),
    "otgan_feature_head_bwd_amax_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "otgan_tanh_fwd_f32": (c_int, [c_fp, c_long, c_fp, c_fp]),
    "otgan_tanh_bwd_f32": (c_int, [c_fp, c_fp, c_long, c_fp, c_fp]),
    "otgan_feature_head_fwd_f32": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "otgan_feature_head_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "otgan_adam_step_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_long, c_double, c_double, c_double,
                                    c_double, c_fp]),
    "otgan_adam_step_gather_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_double, c_double, c_double, c_double,
                                           c_fp, c_double, c_fp]),
    "otgan_adam_step_gather_coef_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_double, c_double, c_double, c_double,
                                                c_fp, c_fp, c_double, c_fp]),
    "otgan_adam_coefficients": (None, [c_double, c_double, c_double, c_fp]),
    "otgan_adam_step_coef_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_long, c_double, c_double, c_double,
                                         c_double, c_fp, c_fp]),
    "otgan_glu_bwd_colsum_f32": (c_int, [c_fp, c_fp, c_long, c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "otgan_copy2d_batched_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "otgan_gather3d_batched_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_fp]),
    "otgan_adamax_step_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_long, c_double, c_double, c_double, c_fp]),
    "otgan_nesterov_step_f32": (c_int, [c_fp, c_fp, c_fp, c_long, c_double, c_double, c_fp]),
    "otgan_ema_update_f32": (c_int, [c_fp, c_fp, c_long, c_double, c_fp]),
}
