"""ctypes binding of libairvision_hip.so (the C ABI declared in include/airvision.h).

There is NO CPU fallback: if the shared library is missing this module raises, and every operator
raises when the HIP call fails.  torch is imported first so that the library binds to the HIP
runtime torch already loaded (same SONAME), i.e. one runtime / one context per process.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before libairvision_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libairvision_hip.so')

AV_MAX_LEVELS = 5
AV_PYR_BORDER = 16
AV_OK, AV_E_INVALID, AV_E_HIP, AV_E_CAPACITY, AV_E_NODEVICE, AV_E_NUMERIC = 0, -1, -2, -3, -4, -5
AV_FE_INPUTS_PERSIST = 1


DISTORTION_MODELS = {'radtan': 0, 'equidistant': 1}          # AV_DISTORTION_* (include/airvision.h)


def distortion_model_code(name):
    """'radtan' / 'equidistant' (config.py:98,117; camera_model.py:41,69: anything but 'equidistant' is radtan there) -> AV_DISTORTION_*."""
    return 1 if name == 'equidistant' else 0


class AirvisionError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, 'libairvision_hip error %d: %s' % (code, text))
        self.code = code


class PyrLayout(C.Structure):
    _fields_ = [('levels', C.c_int32),
                ('w', C.c_int32 * AV_MAX_LEVELS), ('h', C.c_int32 * AV_MAX_LEVELS),
                ('pitch', C.c_int32 * AV_MAX_LEVELS),
                ('offset', C.c_int64 * AV_MAX_LEVELS),
                ('bytes', C.c_int64)]


class FrontendConfig(C.Structure):
    _fields_ = [('width', C.c_int32), ('height', C.c_int32),
                ('grid_row', C.c_int32), ('grid_col', C.c_int32),
                ('grid_min_feature_num', C.c_int32), ('grid_max_feature_num', C.c_int32),
                ('fast_threshold', C.c_int32), ('lk_win', C.c_int32), ('lk_levels', C.c_int32),
                ('lk_max_iter', C.c_int32), ('max_corners', C.c_int32), ('flags', C.c_int32),
                ('lk_eps', C.c_double), ('lk_min_eig', C.c_double), ('stereo_threshold', C.c_double),
                ('cam0_intrinsics', C.c_double * 4), ('cam0_distortion', C.c_double * 4),
                ('cam1_intrinsics', C.c_double * 4), ('cam1_distortion', C.c_double * 4),
                ('R_cam0_imu', C.c_double * 9), ('R_cam1_imu', C.c_double * 9),
                ('R0to1', C.c_double * 9), ('E', C.c_double * 9), ('norm_unit', C.c_double),
                ('cam0_distortion_model', C.c_int32), ('cam1_distortion_model', C.c_int32)]


# name -> (restype, argtypes); the list doubles as the export check of tests/test_abi.py
_P = C.c_void_p
SIGNATURES = {
    'av_last_error': (C.c_char_p, []),
    'av_version': (C.c_char_p, []),
    'av_device_count': (C.c_int, []),
    'av_pyramid_layout': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(PyrLayout)]),
    'av_pyramid_build': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    'av_lk_track': (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int,
                              C.c_int, C.c_int, C.c_double, C.c_double, _P]),
    'av_fast_detect': (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    'av_undistort_points': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _P, _P]),
    'av_distort_points': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _P, _P]),
    'av_undistort_points_model': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _P, _P]),
    'av_distort_points_model': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _P, _P]),
    'av_frontend_create': (C.c_int, [C.POINTER(FrontendConfig), C.c_int, C.c_int, C.POINTER(_P)]),
    'av_frontend_destroy': (None, [_P]),
    'av_frontend_push_imu': (C.c_int, [_P, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    'av_frontend_push_imu_batch': (C.c_int, [_P, _P, _P, _P, C.c_int]),
    'av_frontend_step': (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double), _P]),
    'av_frontend_prestage': (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    'av_frontend_step_host': (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double), _P]),
    'av_frontend_frames_reserve': (C.c_int, [_P, C.c_int]),
    'av_frontend_frames_upload': (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int64, _P]),
    'av_frontend_step_frames': (C.c_int, [_P, _P, C.POINTER(C.c_double), _P]),
    'av_frontend_max_features': (C.c_int, [_P]),
    'av_frontend_read_features': (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    'av_frontend_read_features_begin': (C.c_int, [_P, C.c_int, _P]),
    'av_frontend_features_dev': (C.c_int, [_P, _P, _P, _P, _P]),
    'av_frontend_read_features_end': (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    'av_frontend_read_grid': (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), _P]),
    'av_frontend_read_counters': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32 * 8), _P]),
    'av_frontend_read_match_counts': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32 * 2), _P]),
    'av_msckf_create': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(_P)]),
    'av_msckf_destroy': (None, [_P]),
    'av_msckf_ld': (C.c_int, [_P]),
    'av_msckf_dim': (C.c_int, [_P]),
    'av_msckf_set_cov': (C.c_int, [_P, _P, C.c_int, _P]),
    'av_msckf_get_cov': (C.c_int, [_P, _P, C.c_int, _P]),
    'av_msckf_propagate': (C.c_int, [_P, C.c_double] + [C.POINTER(C.c_double)] * 11 + [_P]),
    'av_msckf_augment': (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    'av_msckf_remove_cam': (C.c_int, [_P, C.c_int, _P]),
    'av_msckf_triangulate': (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _P, _P, _P]),
    'av_msckf_feature_blocks': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, _P, _P, _P]),
    'av_msckf_update': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P]),
    'av_msckf_batch_create': (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 6 + [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(_P)]),
    'av_msckf_batch_destroy': (None, [_P]),
    'av_msckf_batch_push_imu': (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    'av_msckf_batch_step': (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P]),
    'av_msckf_batch_submit': (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P]),
    'av_msckf_batch_device_resident': (C.c_int, [_P]),
    'av_msckf_batch_submit_dev': (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    'av_msckf_batch_wait': (C.c_int, [_P, C.c_int]),
    'av_msckf_batch_get_cov': (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    'av_msckf_batch_sizes': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32 * 3)]),
    'av_msckf_batch_counters': (C.c_int, [_P, C.POINTER(C.c_int64 * 8)]),
    'av_png_decode_gray8': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, C.c_int, _P]),
    'av_quat_to_rotation': (C.c_int, [_P, _P]),
    'av_rotation_to_quat': (C.c_int, [_P, _P]),
    'av_quat_multiply': (C.c_int, [_P, _P, _P]),
    'av_quat_small_angle': (C.c_int, [_P, _P]),
    'av_quat_from_two_vectors': (C.c_int, [_P, _P, _P]),
    'av_msckf_batch_debug_capture': (C.c_int, [_P, C.c_int]),
    'av_msckf_batch_debug_read': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int32), _P, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'av_msckf_batch_work': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double * 8)]),
    'av_msckf_batch_work_executed': (C.c_int, [_P, C.POINTER(C.c_double * 2)]),
    'av_msckf_batch_get_state': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double * 32), _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    'av_msckf_batch_stream_status': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_char_p, C.c_int]),
    'av_frontend_enable_timing': (C.c_int, [_P, C.c_int]),
    'av_frontend_read_timing': (C.c_int, [_P, C.POINTER(C.c_double * 4), C.POINTER(C.c_int32 * 4)]),
}

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libairvision_hip.so is not built (%s); run `python -m uav_airvision_amd.build` -- '
                               'there is no CPU fallback' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise AirvisionError(rc, lib().av_last_error().decode('utf-8', 'replace'))


def current_stream():
    """hipStream_t of torch's current stream as a void pointer."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    return C.c_void_p(t.data_ptr())


def darr(a):
    return (C.c_double * len(a))(*[float(v) for v in a])
