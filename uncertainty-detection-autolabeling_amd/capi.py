"""ctypes binding of include/uda_hip.h (csrc/libuda_hip.so).

The structures below mirror the header field for field; `tests/test_capi_symbols.py`
checks that every function the header declares is exported by the library and that the
structure sizes agree with the C side's.  The product path has no CPU fallback: when the
library (or a GPU) is missing, loading / creating fails loudly.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("UDA_LIB") or os.path.join(CSRC, "libuda_hip.so")   # UDA_LIB: an alternative build for A/B runs
HEADER = os.path.join(os.path.dirname(HERE), "include", "uda_hip.h")

UDA_ABI_VERSION = 4
MAX_LEVELS = 8
MAX_FUSE = 3

OP_STEM, OP_PW, OP_DW, OP_SE, OP_FUSE, OP_POOL, OP_MBX, OP_SEP = 1, 2, 3, 4, 5, 6, 7, 8
ACT_NONE, ACT_SWISH, ACT_RELU, ACT_RELU6, ACT_HSWISH, ACT_MISH = 0, 1, 2, 3, 4, 5
RS_NONE, RS_NEAREST_UP, RS_MAXPOOL = 0, 1, 2
DECODE_PLAIN, DECODE_LNORM, DECODE_FALSEDEC, DECODE_SAMPLE = 0, 1, 2, 3
POST_GLOBAL, POST_PER_CLASS = 0, 1
CALIB_TS_ALL, CALIB_TS_PERCOO, CALIB_ISO_ALL, CALIB_ISO_PERCOO, CALIB_ISO_PERCLSCOO = 0, 1, 2, 3, 4
CLS_TS, CLS_ISO_ALL, CLS_ISO_PERCLS = 0, 1, 2
PROF_AGGREGATE, PROF_NMS, PROF_PREPROCESS = 16, 17, 18


class BufDesc(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("per_sample", C.c_int32),
                ("offset", C.c_int64), ("kind", C.c_int32), ("level", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_", C.c_int32 * MAX_FUSE), ("out", C.c_int32),
                ("se_scale", C.c_int32), ("se_partial", C.c_int32), ("residual", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("w_off", C.c_int64), ("bias_off", C.c_int64), ("bn_scale_off", C.c_int64),
                ("bn_shift_off", C.c_int64), ("se_w1_off", C.c_int64), ("se_b1_off", C.c_int64),
                ("se_w2_off", C.c_int64), ("se_b2_off", C.c_int64), ("se_mid", C.c_int32),
                ("drop_site", C.c_int32), ("resample", C.c_int32 * MAX_FUSE),
                ("fuse_w", C.c_float * MAX_FUSE), ("n_in", C.c_int32), ("drop_site2", C.c_int32),
                ("w2_off", C.c_int64), ("bn2_scale_off", C.c_int64), ("bn2_shift_off", C.c_int64),
                ("launch_group", C.c_int32), ("fuse_in", C.c_int32), ("fuse_act", C.c_int32)]


class DropSite(C.Structure):
    _fields_ = [("channels", C.c_int32), ("rate", C.c_float)]


class Model(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("image_h", C.c_int32), ("image_w", C.c_int32),
                ("mean_rgb", C.c_float * 3), ("stddev_rgb", C.c_float * 3),
                ("num_levels", C.c_int32), ("level_h", C.c_int32 * MAX_LEVELS),
                ("level_w", C.c_int32 * MAX_LEVELS), ("anchors_per_loc", C.c_int32),
                ("num_classes", C.c_int32), ("loss_attenuation", C.c_int32),
                ("mc_samples", C.c_int32), ("cls_stacked", C.c_int32), ("box_stacked", C.c_int32),
                ("has_uncert", C.c_int32), ("decode_method", C.c_int32),
                ("enable_softmax", C.c_int32), ("nms_soft_sigma", C.c_float),
                ("nms_iou_thresh", C.c_float), ("nms_score_thresh", C.c_float),
                ("max_output_size", C.c_int32), ("max_nms_inputs", C.c_int32),
                ("post_mode", C.c_int32), ("chunk_images", C.c_int32), ("max_images", C.c_int32),
                ("arena_floats", C.c_int64), ("n_drop_sites", C.c_int32), ("decode_nsamples", C.c_int32)]


_P = C.c_void_p
_SIGNATURES = {
    "uda_create": (C.c_int, [C.POINTER(Model), C.POINTER(BufDesc), C.c_int32, C.POINTER(Op), C.c_int32,
                             C.POINTER(DropSite), _P, C.c_int64, _P, C.c_int32, C.POINTER(_P)]),
    "uda_destroy": (None, [_P]),
    "uda_last_error": (C.c_char_p, [_P]),
    "uda_set_images_u8": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32]),
    "uda_set_images_u8_device": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32]),
    "uda_set_images_u8_ragged": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "uda_input_u8_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uda_prefetch_images_u8": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32]),
    "uda_prefetch_images_u8_ragged": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "uda_swap_prefetched": (C.c_int, [_P]),
    "uda_set_images_f32": (C.c_int, [_P, _P, C.c_int32, _P]),
    "uda_set_dropout_seed": (C.c_int, [_P, C.c_uint64]),
    "uda_set_dropout_image_offset": (C.c_int, [_P, C.c_int64]),
    "uda_set_dropout_sample_shard": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "uda_set_dropout_masks": (C.c_int, [_P, _P, C.c_int64]),
    "uda_get_dropout_masks": (C.c_int, [_P, _P, C.c_int64]),
    "uda_run": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "uda_synchronize": (C.c_int, [_P]),
    "uda_run_async": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "uda_collect": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "uda_collect_device": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "uda_drain": (C.c_int, [_P]),
    "uda_range_demotions": (C.c_int64, [_P]),
    "uda_nms_prefix_fallbacks": (C.c_int64, [_P]),
    "uda_nms_coop_fallbacks": (C.c_int64, [_P]),
    "uda_nms_coop_not_launched": (C.c_int64, [_P]),
    "uda_get_detections": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "uda_detection_cols": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "uda_detections_device": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "uda_get_class_probs": (C.c_int, [_P, _P, _P]),
    "uda_calibrate_box": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "uda_calibrate_class": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_uint64, _P, _P, _P]),
    "uda_serve": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "uda_get_head_outputs": (C.c_int, [_P, C.c_int32, _P, _P]),
    "uda_set_head_outputs": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_int64]),
    "uda_head_outputs_device": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "uda_set_num_images": (C.c_int, [_P, C.c_int32]),
    "uda_postprocess_heads": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "uda_copy_heads": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "uda_predict": (C.c_int, [_P, _P, C.c_int32]),
    "uda_get_candidates": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "uda_num_candidates": (C.c_int32, [_P]),
    "uda_read_buffer": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "uda_get_preprocessed": (C.c_int, [_P, _P, _P]),
    "uda_nms": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                          C.c_int32, _P, _P, _P]),
    "uda_debug_pw": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_float)]),
    "uda_nms_np": (C.c_int, [C.c_int32, _P, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, _P, C.POINTER(C.c_int32)]),
    "uda_per_class_nms_np": (C.c_int, [C.c_int32, _P, _P, _P, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.c_float, _P]),
    "uda_crc32c": (C.c_uint32, [_P, C.c_uint64, C.c_uint32]),
    "uda_profile_enable": (C.c_int, [_P, C.c_uint32]),
    "uda_profile_read": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def build(force=False):
    """Compile csrc/*.hip for gfx950 into csrc/libuda_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_PATH


def load():
    """dlopen the library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP library %s is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C %s`). There is no CPU fallback." % (LIB_PATH, CSRC))
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class UdaError(RuntimeError):
    pass


def check(lib, ctx, rc, what):
    if rc != 0:
        msg = lib.uda_last_error(ctx)
        raise UdaError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
