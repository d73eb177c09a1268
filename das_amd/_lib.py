"""ctypes binding of libdas_hip.so (the C ABI declared in include/das_hip.h).

The product path has no CPU fallback: if the shared library is missing or does not export
every symbol the header declares, importing the ops fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdas_hip.so')

DAS_OK, DAS_ERR_ARG, DAS_ERR_LAUNCH = 0, 1, 2
DAS_F32, DAS_BF16 = 0, 1
DAS_MAX_LEVELS = 5

vp, i32, f32, i64 = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class DasConvDesc(C.Structure):
    _fields_ = [('dtype', i32), ('out_dtype', i32),
                ('B', i32), ('H', i32), ('W', i32), ('Cin', i32), ('x_pix_stride', i32),
                ('Ho', i32), ('Wo', i32), ('Cout', i32), ('y_pix_stride', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('relu_in', i32), ('relu', i32),
                ('scale', vp), ('shift', vp), ('residual', vp), ('res_pix_stride', i32), ('stats', vp),
                ('num_levels', i32), ('lvl_H', i32 * 5), ('lvl_W', i32 * 5), ('in_up', i32), ('stats_slots', i32),
                ('bnb_raw', vp), ('bnb_y', vp), ('bnb_mean', vp), ('bnb_invstd', vp), ('bnb_gamma', vp), ('bnb_beta', vp),
                ('bnb_relu', i32), ('bnb_pix_stride', i32),
                ('out_sub', i32), ('out_ph', i32), ('out_pw', i32), ('out_H', i32), ('out_W', i32),
                ('bnb_mask_bits', vp), ('residual_mask_bits', vp)]


class DasPackEntry(C.Structure):
    _fields_ = [('off', i64), ('O', i32), ('I', i32), ('KH', i32), ('KW', i32), ('tile_start', i32), ('s2_pad', i32)]


class DasBnFinalize(C.Structure):
    _fields_ = [('stats', vp), ('stats_slots', i32), ('C', i32), ('count', i64), ('running_mean', vp), ('running_var', vp),
                ('momentum', f32), ('eps', f32), ('save_mean', vp), ('save_invstd', vp), ('num_batches_tracked', vp)]


class DasFlowJob(C.Structure):
    _fields_ = [('params', vp), ('dparams', vp), ('dst_table', vp), ('row_start', i32), ('row_end', i32)]


class DasLevels(C.Structure):
    _fields_ = [('num_levels', i32), ('B', i32), ('H', i32 * 5), ('W', i32 * 5)]


class DasHeadDesc(C.Structure):
    _fields_ = [('J', i32), ('root_idx', i32), ('raw_ps', i32), ('off_c', i32), ('depth_c', i32), ('uvd_c', i32),
                ('sigma_c', i32), ('scale', (f32 * 4) * 5), ('level_stride', f32 * 5), ('z_norm', f32),
                ('depth_factor', f32), ('scale_dev', vp)]


class DasTargetDesc(C.Structure):
    _fields_ = [('J', i32), ('background', i32), ('stride', i32 * 5), ('range_lo', f32 * 5), ('range_hi', f32 * 5),
                ('radius', f32), ('alpha', f32)]


class DasPhotometric(C.Structure):
    _fields_ = [('use_brightness', i32), ('use_contrast', i32), ('contrast_first', i32), ('use_saturation', i32),
                ('use_hue', i32), ('brightness', f32), ('contrast', f32), ('saturation', f32), ('hue', f32),
                ('perm', i32 * 3)]


class DasRleDesc(C.Structure):
    _fields_ = [('J', i32), ('sets', i32), ('npos', i32), ('pose_ps', i32), ('aux_ps', i32), ('stride2', i32),
                ('stride3', i32), ('amp', f32), ('beta', f32)]


class DasDecodeDesc(C.Structure):
    _fields_ = [('B', i32), ('J', i32), ('num_levels', i32),
                ('H', i32 * DAS_MAX_LEVELS), ('W', i32 * DAS_MAX_LEVELS), ('stride', i32 * DAS_MAX_LEVELS),
                ('cls', vp * DAS_MAX_LEVELS), ('ctr', vp * DAS_MAX_LEVELS), ('pose', vp * DAS_MAX_LEVELS),
                ('cls_ps', i32 * DAS_MAX_LEVELS), ('ctr_ps', i32 * DAS_MAX_LEVELS), ('pose_ps', i32 * DAS_MAX_LEVELS),
                ('nms_pre', i32), ('nms_post', i32), ('score_thr', f32), ('nms_thr', f32), ('scale_factor', vp), ('nms_soft', i32)]


# name -> (restype, argtypes); must list every function include/das_hip.h declares
SIGNATURES = {
    'das_abi_version': (i32, []),
    'das_target_arch': (C.c_char_p, []),
    'das_tuning_set': (i32, [C.c_char_p, i64]),
    'das_tuning_get': (i32, [C.c_char_p, C.POINTER(i64)]),
    'das_tuning_reset': (i32, []),
    'das_wgrad_pp_share': (i32, [i32]),
    'das_dev_occupy_cus': (i32, [i32, i32, i32, i32, vp]),
    'das_last_kernel': (C.c_char_p, []),
    'das_conv_last_tile_rows': (i32, []),
    'das_prof_begin': (i32, []),
    'das_prof_end': (i32, []),
    'das_prof_count': (i64, []),
    'das_prof_read': (i32, [vp, vp, i32, i64]),
    'das_img_resize_bilinear': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_img_resize_bilinear_u8': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_img_flip_horizontal': (i32, [vp, vp, i32, i32, i32, vp]),
    'das_img_photometric': (i32, [vp, i32, i32, C.POINTER(DasPhotometric), vp]),
    'das_img_warp_affine': (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(f32), vp]),
    'das_img_normalize_pad_chw': (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), i32, vp]),
    'das_conv2d_nhwc': (i32, [vp, vp, vp, C.POINTER(DasConvDesc), vp]),
    'das_conv2d_wgrad_nhwc': (i32, [vp, vp, vp, C.POINTER(DasConvDesc), i32, vp]),
    'das_conv2d_wgrad_batch': (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(DasConvDesc), i32, vp]),
    'das_wgrad_last_plan': (i32, [C.POINTER(i64), i32]),
    'das_pack_conv_weights': (i32, [vp, vp, vp, vp, i32, vp, i32, i32, vp]),
    'das_colsum': (i32, [vp, i32, i64, i32, i32, vp, vp]),
    'das_colsum_acc': (i32, [vp, i32, i64, i32, i32, vp, vp]),
    'das_bn_train_backward_bits_phase': (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i64, vp]),
    'das_bn_train_backward_bits': (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    'das_bn_train_backward': (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
    'das_bn_train_backward_phase': (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i64, vp]),
    'das_bn_backward_apply': (i32, [vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp]),
    'das_groupnorm_backward': (i32, [vp, vp, vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, vp, vp, vp, f32, i32, vp,
                                     vp, vp, vp]),
    'das_groupnorm_backward_acc': (i32, [vp, vp, vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, vp, vp, vp, f32, i32, vp,
                                         vp, vp, i32, vp]),
    'das_maxpool3x3s2_backward': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_upsample_bilinear_ac_backward': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'das_upsample_nearest_backward': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'das_pack_nchw_to_nhwc': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'das_unpack_nhwc_to_nchw': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'das_maxpool3x3s2': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_upsample_bilinear_ac_stats': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    'das_upsample_stats_lowres': (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    'das_upmerge_forward': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32] + [vp] * 8 + [vp, vp]),
    'das_upmerge_backward_reduce': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    'das_upmerge_backward_lowres': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]),
    'das_bn_dual_apply': (i32, [vp, vp, vp, i32, i64, i32, vp, i32, vp, vp]),
    'das_bn_relu_add3_forward': (i32, [vp, vp, vp, vp, i32, i64, i32, vp, vp]),
    'das_bn_relu_add3_backward': (i32, [vp, vp, vp, vp, vp, i32, i64, i32, vp, vp, i32, i64, vp, vp, vp, vp, i32, vp]),
    'das_maxpool3x3s2_argmax': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_maxpool3x3s2_backward_argmax': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'das_upsample_bilinear_ac': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'das_add_upsample_nearest': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'das_add3': (i32, [vp, vp, vp, vp, i32, i64, i32, vp]),
    'das_bn_finalize_many': (i32, [C.POINTER(DasBnFinalize), i32, vp]),
    'das_bn_train_apply': (i32, [vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, f32, f32, vp, i32, vp, vp, vp, i64, i32,
                                 vp, vp]),
    'das_groupnorm_nhwc': (i32, [vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, vp, vp, f32, i32, vp, i32, vp]),
    'das_deform_im2col3x3': (i32, [vp, vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, vp]),
    'das_dcn3x3_fused': (i32, [vp, vp, vp, vp, vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, i32, i32, vp]),
    'das_offset_sample': (i32, [vp, vp, vp, vp, C.POINTER(DasLevels), i32, i32, i32, i32, i32, i32, vp]),
    'das_sigmoid_blend': (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    'das_head_assemble': (i32, [vp, vp, vp, C.POINTER(DasLevels), C.POINTER(DasHeadDesc), vp]),
    'das_head_finalize': (i32, [vp, vp, C.POINTER(DasLevels), C.POINTER(DasHeadDesc), i32, i32, vp]),
    'das_deform_im2col3x3_backward': (i32, [vp, vp, vp, vp, vp, i32, C.POINTER(DasLevels), i32, i32, i32, i32, vp]),
    'das_offset_sample_backward': (i32, [vp, vp, vp, vp, vp, vp, vp, C.POINTER(DasLevels), i32, i32, i32, i32, i32,
                                         i32, vp]),
    'das_sigmoid_blend_backward': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    'das_head_assemble_backward': (i32, [vp, vp, vp, vp, vp, C.POINTER(DasLevels), C.POINTER(DasHeadDesc), vp]),
    'das_assign_targets': (i32, [C.POINTER(DasLevels), C.POINTER(DasTargetDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    'das_positive_rows': (i32, [vp, i32, vp, vp, C.POINTER(DasLevels), C.POINTER(DasTargetDesc), f32, f32, f32, vp, vp, vp, vp,
                                vp, vp, vp, vp]),
    'das_sigmoid_focal_loss': (i32, [vp, i32, vp, vp, i64, f32, f32, vp, vp, vp]),
    'das_smooth_l1_loss': (i32, [vp, vp, vp, i64, f32, vp, vp, vp]),
    'das_bce_logits_loss': (i32, [vp, vp, vp, i64, vp, vp, vp]),
    'das_realnvp_log_prob': (i32, [vp, i32, i32, vp, i32, C.c_uint, vp, vp, vp]),
    'das_realnvp_log_prob_multi': (i32, [vp, i32, i32, C.POINTER(DasFlowJob), i32, i32, C.c_uint, vp, vp, vp]),
    'das_realnvp_log_prob_multi_backward': (i32, [vp, vp, i32, i32, C.POINTER(DasFlowJob), i32, i32, C.c_uint, vp, vp]),
    'das_realnvp_log_prob_backward': (i32, [vp, vp, i32, i32, vp, i32, C.c_uint, vp, vp, vp, vp]),
    'das_rle_blocks': (i32, [C.POINTER(DasRleDesc)]),
    'das_rle_prepare': (i32, [vp, vp, vp, vp, vp, vp, vp, C.POINTER(DasRleDesc), vp, vp, vp, vp, vp]),
    'das_rle_loss': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(DasRleDesc), vp, vp]),
    'das_rle_backward': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(DasRleDesc), vp, vp, vp]),
    'das_grad_sumsq': (i32, [vp, i64, vp, i32, vp]),
    'das_sgd_momentum_step': (i32, [vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, i32, vp]),
    'das_decode_cap': (i32, [C.POINTER(DasDecodeDesc)]),
    'das_decode_ws_bytes': (i64, [i32, i32, i32]),
    'das_decode': (i32, [C.POINTER(DasDecodeDesc), vp, vp, vp, vp, vp, vp, vp]),
}

_lib = None


class DasHipError(RuntimeError):
    pass


def load():
    """Load libdas_hip.so and bind every declared entry point. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DasHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C das_amd/csrc`. das_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DasHipError(f'libdas_hip.so does not export {name}') from e
        fn.restype, fn.argtypes = res, args
    if lib.das_abi_version() != 4:
        raise DasHipError('libdas_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(status, what):
    if status != DAS_OK:
        raise DasHipError(f'{what} failed: ' + {DAS_ERR_ARG: 'DAS_ERR_ARG (unsupported shape/dtype/alignment)',
                                                  DAS_ERR_LAUNCH: 'DAS_ERR_LAUNCH (HIP launch error)'}.get(
                                                      status, str(status)))
