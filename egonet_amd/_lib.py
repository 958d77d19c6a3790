"""ctypes binding of ``csrc/libegonet_hip.so`` (the C ABI in include/egonet_hip.h).

The library is built in-tree by ``egonet_amd.build`` (hipcc, gfx950).  There is
no fallback: if the shared object is missing or a symbol cannot be resolved,
``lib()`` raises, and every CUDA-tensor code path of the package goes through
``lib()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EGONET_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libegonet_hip.so')   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'egonet_hip.h')

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY = 0, 1, 2, 3


class Ref(C.Structure):
    """egn_ref: (slot, byte offset); slot < 0 means NULL."""
    _fields_ = [('slot', C.c_int32), ('off', C.c_int64)]


NULL_REF = Ref(-1, 0)

_p = C.c_void_p
_i = C.c_int
_d = C.c_double

# name -> (restype, argtypes); kept in step with include/egonet_hip.h (checked
# by tests/test_abi.py, which parses the header)
SIGNATURES = {
    'egn_version': (_i, []),
    'egn_strerror': (C.c_char_p, [_i]),
    'egn_conv2d_f32': (_i, [_p] * 6 + [_i] * 14 + [_p]),
    'egn_conv_plan_query': (_i, [_i] * 13 + [C.POINTER(_i)]),
    'egn_conv_num_configs': (_i, []),
    'egn_conv_config_info': (_i, [_i, C.POINTER(_i), C.POINTER(_i)]),
    'egn_conv_config_name': (_i, [_i, C.c_char_p, _i]),
    'egn_conv_config_kind': (_i, [_i]),
    'egn_probe_build': (_i, []),
    'egn_wino_weight_floats': (C.c_long, [_i, _i, _i]),
    'egn_wino4_weight_floats': (C.c_longlong, [_i, _i]),
    'egn_wino4_pack_weight_floats': (C.c_longlong, [_i, _i, _i]),
    'egn_wino4_pack_weight_f32': (_i, [_p, _i, _i, _i, _p, _p]),
    'egn_wino_pack_weight_f32': (_i, [_p, _i, _i, _i, _p, _p]),
    'egn_fuse_sum_relu_f32': (_i, [_p, _i, _i, _i, _i, _i, _i, C.POINTER(_p), C.POINTER(_i), _i, _p]),
    'egn_nchw_to_nhwc_f32': (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'egn_nhwc_to_nchw_f32': (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'egn_pixel_shuffle_nhwc_to_nchw_f32': (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'egn_fill_coord_ramps_f32': (_i, [_p, _i, _i, _i, _i, _i, _p]),
    'egn_decode_heatmaps_f32': (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'egn_keypoints_to_screen_f64': (_i, [_p, _i, _i, _d, _d, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p]),
    'egn_unnormalize_f64': (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    'egn_pose_solve_f64': (_i, [_p, _i, _p, _d, _d, _i, _p, _p, _p]),
    'egn_keypoints_to_screen_host_f64': (_i, [_p, _i, _i, _d, _d, _p, _p, _i, _i, _p]),
    'egn_pose_solve_host_f64': (_i, [_p, _i, _p, _d, _d, _i, _p, _p]),
    'egn_kitti_eval_image': (_i, [C.c_char_p, C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                  C.POINTER(_d), C.POINTER(_d)]),
    'egn_crop_warp_normalize_u8': (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p]),
    'egn_conv2d_wgrad_ws_bytes': (C.c_long, [_i] * 11),
    'egn_conv2d_wgrad_f32': (_i, [_p, _p, _p] + [_i] * 11 + [_p, C.c_long, _p]),
    'egn_pack_matrix_f32': (_i, [_p, _i, _i, _i, _i, _p, _p]),
    'egn_transpose_f32': (_i, [_p, _i, _i, _i, _p, _i, _p]),
    'egn_colreduce_ws_bytes': (C.c_long, [_i]),
    'egn_colsum_f32': (_i, [_p, _i, _i, _i, _p, _p, _p]),
    'egn_bn_stats_f32': (_i, [_p, _i, _i, _i, C.c_float, _p, _p, _p, _p, _p, C.c_float, _p, _p]),
    'egn_conv2d_bnstats_rows': (C.c_long, [_i] * 12),
    'egn_conv2d_bnstats_f32': (_i, [_p] * 5 + [_i] * 12 + [_p, C.c_long, _p]),
    'egn_conv2d_ticket_words': (C.c_long, [_i] * 12),
    'egn_conv2d_ex_f32': (_i, [_p] * 6 + [_i] * 13 + [_p, C.c_long, _p, C.c_long, _p]),
    'egn_bn_stats_finalize_f32': (_i, [_p, C.c_long, _i, _i, C.c_float, _p, _p, _p, _p, _p, C.c_float, _p]),
    'egn_bn_act_fwd_f32': (_i, [_p, _p, _p, _p, _p, _p, C.c_float, _i, _p, _p, _i, _i, _i, _p]),
    'egn_bn_bwd_sums_f32': (_i, [_p, _p, _p, C.c_float, _p, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p]),
    'egn_bn_bwd_dz_f32': (_i, [_p, _p, _p, C.c_float, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'egn_bn_act_fwd_drop_f32': (_i, [_p, _p, _p, _p, _p, C.c_float, C.c_ulonglong, _p, _i, _i, _p, _p, _i, _i, _i, _p]),
    'egn_bn_bwd_sums_drop_f32': (_i, [_p, _p, C.c_float, C.c_ulonglong, _p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p,
                                      _p, _p]),
    'egn_bn_bwd_dz_drop_f32': (_i, [_p, _p, C.c_float, C.c_ulonglong, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i,
                                    _i, _i, _p]),
    'egn_dropout_mask_f32': (_i, [_p, C.c_long, C.c_float, C.c_ulonglong, _p, _i, _p]),
    'egn_add_f32': (_i, [_p, _p, _p, C.c_long, _p]),
    'egn_mse_f32': (_i, [_p, _p, _i, _i, _i, _i, C.c_float, _i, _p, _p, _p]),
    'egn_l1_f32': (_i, [_p, _p, C.c_long, C.c_float, _p, _p, _p]),
    'egn_elem_loss_f32': (_i, [_p, _p, _i, _i, _i, _i, _i, C.c_float, _i, _p, _p, _p]),
    'egn_cross_ratio_ws_bytes': (C.c_long, [_i, _i]),
    'egn_cross_ratio_f32': (_i, [_p, _i, _i, _p, _i, _d, C.c_float, _i, C.c_float, _p, _p, _p, _p]),
    'egn_sigmoid_bwd_f32': (_i, [_p, _p, _p, C.c_long, _p]),
    'egn_packed_weight_floats': (C.c_long, [_i] * 5),
    'egn_pack_conv_weight_f32': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'egn_pack_desc_bytes': (_i, []),
    'egn_pack_conv_weights_batch_f32': (_i, [_p, _i, C.c_long, _p]),
    'egn_zero_insert2_f32': (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'egn_fuse_bwd_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'egn_gaussian_targets_f32': (_i, [_p, _p, _i, _i, _i, _i, _d, _d, _d, _p, _p, _p]),
    'egn_ema_f32': (_i, [_p, _p, C.c_float, _i, _p]),
    'egn_adam_step_f32': (_i, [_p, _p, _p, _p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, _i, _p]),
    'egn_step_counters_i64': (_i, [_p, _i, _p, _p]),
    'egn_adam_step_dev_f32': (_i, [_p, _p, _p, _p, C.c_long, _p, C.c_float, C.c_float, C.c_float, _p, _p]),
    'egn_adam_l2_step_dev_f32': (_i, [_p, _p, _p, _p, C.c_long, _p, C.c_float, C.c_float, C.c_float, C.c_float, _p, _p]),
    'egn_sgd_step_dev_f32': (_i, [_p, _p, _p, C.c_long, _p, C.c_float, C.c_float, _p, _p]),
    'egn_program_create': (_p, [_i]),
    'egn_program_destroy': (None, [_p]),
    'egn_program_bind': (_i, [_p, _i, _p]),
    'egn_program_num_ops': (_i, [_p]),
    'egn_program_add_conv2d': (_i, [_p] + [Ref] * 6 + [_i] * 14),
    'egn_program_add_fuse': (_i, [_p, Ref, _i, _i, _i, _i, _i, _i, C.POINTER(Ref), C.POINTER(_i), _i]),
    'egn_program_add_nchw_to_nhwc': (_i, [_p, Ref, Ref, _i, _i, _i, _i, _i]),
    'egn_program_add_nhwc_to_nchw': (_i, [_p, Ref, Ref, _i, _i, _i, _i, _i]),
    'egn_program_add_pixel_shuffle': (_i, [_p, Ref, Ref, _i, _i, _i, _i, _i, _i]),
    'egn_program_add_pw_pair': (_i, [_p] + [Ref] * 8 + [_i, _i]),
    'egn_pw_pair_f32': (_i, [_p] * 8 + [_i, _i, _p]),
    'egn_program_add_ramps': (_i, [_p, Ref, _i, _i, _i, _i, _i]),
    'egn_program_add_decode': (_i, [_p, Ref, _i, _i, _i, _i, _i, Ref, Ref, Ref]),
    'egn_program_fork': (_i, [_p]),
    'egn_program_join': (_i, [_p]),
    'egn_program_set_lane': (_i, [_p, _i]),
    'egn_program_tag': (_i, [_p, C.c_char_p, _d, _d]),
    'egn_program_run': (_i, [_p, _p]),
    'egn_program_run_timed': (_i, [_p, _p, C.POINTER(C.c_float), _i]),
    'egn_program_capture': (_i, [_p, _p]),
    'egn_program_replay': (_i, [_p, _p]),
    'egn_program_ticket_ops': (_i, [_p]),
    'egn_program_poke_ticket': (_i, [_p, _i, _i, C.c_uint]),
    'egn_gemm_supported': (_i, [_i] * 7),
    'egn_gemm_ws_bytes': (C.c_long, [_i] * 4),
    'egn_gemm_f32': (_i, [_i, _p, _p, _p, _p] + [_i] * 7 + [_p, C.c_long, _p]),
    'egn_gemm_stats_rows': (C.c_long, [_i]),
    'egn_gemm_ex_f32': (_i, [_i, _p, _p, _p, _p, _p, _p, C.c_long] + [_i] * 7 + [_p, C.c_long, _p]),
    'egn_launch_count': (C.c_long, []),
    'egn_direct_conv_count': (C.c_long, []),
    'egn_program_op_info': (_i, [_p, _i, C.POINTER(_i), C.POINTER(_d), C.POINTER(_d), C.c_char_p, _i]),
}

_LIB = None


class EgonetHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if it is not built."""
    global _LIB
    if _LIB is None:
        # torch bundles its own libamdhip64; it must be the HIP runtime of the
        # process, so torch is imported BEFORE the library is dlopen'ed (loading
        # /opt/rocm's copy first gives two runtimes and "no ROCm-capable device")
        import torch  # noqa: F401
        if not os.path.isfile(LIB_PATH):
            raise EgonetHipError(
                'egonet_amd: %s is missing -- build it with `python -m egonet_amd.build` '
                '(hipcc --offload-arch=gfx950); there is no CPU/torch fallback for CUDA tensors'
                % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is absent
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(code, what=''):
    if code != 0:
        msg = lib().egn_strerror(code).decode()
        if code == -1:
            raise ValueError('%s: %s' % (what or 'egonet_hip', msg))
        raise EgonetHipError('%s: %s (code %d)' % (what or 'egonet_hip', msg, code))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
