"""ctypes binding of libstereotrack_hip.so (the C ABI declared in include/stereotrack.h).

The HIP library is the ONLY compute path of this package: if it is missing, loading raises —
there is no CPU fallback.  `import torch` happens first so that the library resolves
libamdhip64.so.7 to the HIP runtime PyTorch already loaded (one runtime, shared device pointers
and streams).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libstereotrack_hip.so')

ST_OK = 0
_ERR_NAMES = {-1: 'ST_ERR_INVALID', -2: 'ST_ERR_HIP', -3: 'ST_ERR_STATE', -4: 'ST_ERR_WORKSPACE',
              -5: 'ST_ERR_NOTFOUND'}


class StError(RuntimeError):
    """A libstereotrack_hip call returned a non-zero status."""


class StConvDesc(C.Structure):
    _fields_ = [
        ('in_dev', C.c_void_p),
        ('N', C.c_int), ('Hi', C.c_int), ('Wi', C.c_int), ('Cin', C.c_int), ('in_ld', C.c_int), ('in_off', C.c_int),
        ('wgt_dev', C.c_void_p), ('bias_dev', C.c_void_p),
        ('Cout', C.c_int), ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
        ('out1_dev', C.c_void_p), ('out1_ld', C.c_int), ('out1_off', C.c_int), ('split', C.c_int),
        ('out2_dev', C.c_void_p), ('out2_ld', C.c_int), ('out2_off', C.c_int),
        ('up_dev', C.c_void_p), ('up_ld', C.c_int), ('up_off', C.c_int),
        ('res_dev', C.c_void_p), ('res_ld', C.c_int), ('res_off', C.c_int),
        ('post_scale', C.c_float), ('act', C.c_int), ('wgt_wino_dev', C.c_void_p),
    ]


class StDetectorConfig(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int), ('widen_factor', C.c_float), ('deepen_factor', C.c_float),
        ('num_classes', C.c_int), ('batch', C.c_int), ('height', C.c_int), ('width', C.c_int),
        ('bn_eps', C.c_double), ('with_right_branch', C.c_int), ('disp_planes_identical', C.c_int),
        ('rgb_only', C.c_int),
    ]


class StTrackerConfig(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int), ('obj_score_thr', C.c_float), ('init_track_thr', C.c_float),
        ('weight_iou_with_det_scores', C.c_int), ('match_iou_thr', C.c_float), ('num_tentatives', C.c_int),
        ('vel_consist_weight', C.c_float), ('vel_delta_t', C.c_int), ('num_frames_retain', C.c_int),
    ]


class StDecodeDesc(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int), ('batch', C.c_int), ('num_levels', C.c_int),
        ('level_h', C.c_int * 4), ('level_w', C.c_int * 4), ('level_stride', C.c_int * 4),
        ('level_offset', C.c_size_t * 4),
        ('score_thr', C.c_float), ('iou_thr', C.c_float), ('max_det', C.c_int),
        ('scale_x', C.c_float), ('scale_y', C.c_float), ('pad_left', C.c_float), ('pad_top', C.c_float),
        ('ori_w', C.c_float), ('ori_h', C.c_float), ('nms_mask_rows', C.c_int), ('num_classes', C.c_int),
        ('single_label', C.c_int),
    ]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_PROTOS = {
    'st_version': (C.c_int, []),
    'st_head_row_floats': (_i, [_i]),
    'st_png_unfilter': (_i, [_vp, _i, _i, _i, _vp]),
    'st_stem_focus_conv_u8': (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    'st_detector_forward_raw': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _sz, _vp, _vp]),
    'st_detector_forward_phase0_raw': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _sz, _vp]),
    'st_pack_raw_frames': (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    'st_resize_planes': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    'st_detector_set_split': (_i, [_vp, _i]),
    'st_split_instances_available': (_i, []),
    'st_volume_agg3d': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _i, _vp]),
    'st_last_error': (C.c_char_p, []),
    'st_conv2d_nhwc': (_i, [C.POINTER(StConvDesc), _vp]),
    'st_conv2d_nhwc_variant': (_i, [C.POINTER(StConvDesc), _vp, _i]),
    'st_conv1x1_chain': (_i, [C.POINTER(StConvDesc), C.POINTER(StConvDesc), _vp]),
    'st_front_frag_floats': (_sz, [_i, _i]),
    'st_front_pack_frags': (_i, [_vp, _i, _i, _vp]),
    'st_conv3x3s2_csp_front': (_i, [C.POINTER(StConvDesc), C.POINTER(StConvDesc), C.POINTER(StConvDesc), _vp, _vp, _vp]),
    'st_csp_tail_frag_floats': (_sz, []),
    'st_csp_tail_pack_frags': (_i, [_vp, _vp]),
    'st_conv3x3_csp_tail': (_i, [C.POINTER(StConvDesc), C.POINTER(StConvDesc), _vp, _vp]),
    'st_conv_packed_floats': (_sz, [_i, _i, _i, _i]),
    'st_wino_packed_floats': (_sz, [_i, _i]),
    'st_wino_pack_weights': (_i, [_vp, _i, _i, _vp]),
    'st_conv_pack_weights': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i, _i, _i, _i, _vp, _vp]),
    'st_focus_pack': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'st_stem_packed_floats': (_sz, [_i]),
    'st_stem_pack_weights': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i, _i, _vp, _vp]),
    'st_stem_focus_conv': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    'st_pack_raw_inputs': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'st_spp_pool': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'st_detector_create': (_i, [C.POINTER(StDetectorConfig), C.POINTER(_vp)]),
    'st_detector_destroy': (_i, [_vp]),
    'st_detector_num_params': (_i, [_vp]),
    'st_detector_param_info': (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_int64), C.POINTER(_i)]),
    'st_detector_set_param': (_i, [_vp, C.c_char_p, _vp, C.c_int64]),
    'st_detector_finalize': (_i, [_vp]),
    'st_detector_workspace_bytes': (_sz, [_vp]),
    'st_detector_head_floats': (_sz, [_vp]),
    'st_detector_num_levels': (_i, [_vp]),
    'st_detector_level_info': (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz)]),
    'st_detector_forward': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'st_detector_forward_phase': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'st_detector_macs': (C.c_double, [_vp]),
    'st_detector_set_timing': (_i, [_vp, _i]),
    'st_detector_num_ops': (_i, [_vp]),
    'st_detector_op_times': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    'st_detector_autotune': (_i, [_vp, _vp, _sz, _vp, _vp, _i]),
    'st_conv_variant_name': (C.c_char_p, [_i]),
    'st_conv_variant_signature': (C.c_char_p, [_i]),
    'st_detector_get_tuning': (_i, [_vp, _vp, _i]),
    'st_detector_set_tuning': (_i, [_vp, _vp, _i]),
    'st_detector_op_desc': (_i, [_vp, _i, C.c_char_p, _i]),
    'st_detector_tap': (_i, [_vp, C.c_char_p, _vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                             C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'st_decode_nms_workspace_bytes': (_sz, [C.POINTER(StDecodeDesc)]),
    'st_lapjv_extended': (_i, [_vp, _i, _i, C.c_double, _vp, _vp]),
    'st_tracker_create': (_i, [C.POINTER(StTrackerConfig), C.POINTER(_vp)]),
    'st_tracker_destroy': (_i, [_vp]),
    'st_tracker_reset': (_i, [_vp]),
    'st_tracker_track': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, C.POINTER(_i)]),
    'st_tracker_track_records': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    'st_tracker_num_tracks': (_i, [_vp]),
    'st_tracker_next_id': (C.c_longlong, [_vp]),
    'st_tracker_get_track': (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'st_batched_tracker_create': (_i, [C.POINTER(StTrackerConfig), _i, _i, _i, C.POINTER(_vp)]),
    'st_batched_tracker_destroy': (_i, [_vp]),
    'st_batched_tracker_state_bytes': (_sz, [_vp]),
    'st_batched_tracker_scratch_bytes': (_sz, [_vp]),
    'st_batched_tracker_step': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'st_decode_nms': (_i, [C.POINTER(StDecodeDesc), _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'st_costvolume_softargmin': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    'st_softargmin': (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    'st_disp_upsample_pack': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'st_feat_upsample': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'st_costvolume_agg3d_supported': (_i, [_i, _i]),
    'st_costvolume_agg3d': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _i, _vp, _vp]),
    'st_costvolume_agg3d_softargmin': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _i, _f, _vp, _vp]),
    'st_box_depth_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'st_pack_records': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'st_box_depth': (_i, [_vp, _sz, _i, _i, _i, _vp, _vp, _i, _f, _f, _vp, _sz, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('ST_LIBRARY') or LIB_PATH   # ST_LIBRARY: tools/ load the -DST_ABLATION build
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C stereotracking_amd/csrc`).  stereotracking_amd has no CPU fallback.')
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != ST_OK:
        msg = load().st_last_error()
        raise StError(f'{what or "libstereotrack_hip"}: {_ERR_NAMES.get(rc, rc)}: '
                      f'{msg.decode() if msg else ""}')


def ptr(t):
    """Device/host pointer of a contiguous float32/int tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
