"""ctypes binding of include/piccolo_hip.h (the product's only compute back end).

There is deliberately NO fallback: if libpiccolo_hip.so is missing or has the wrong ABI, importing the
product's ops raises.  Build it with `python -m piccolo_amd.build` (or __graft_entry__.build()).
"""
import ctypes
import os

from . import build as _build

_c = ctypes
_vp, _i64, _int, _sz, _dbl = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t, _c.c_double

ABI_VERSION = 9
RESULT_STRIDE = 8
GD_RESULT_STRIDE = 16
GD_SEQUENTIAL, GD_BATCH = 0, 1
PANO_F32, PANO_U8, PANO_F16, PANO_U8P, PANO_U8V = 0, 1, 2, 3, 4


class GdHyper(_c.Structure):
    _fields_ = [("lr", _dbl), ("factor", _dbl), ("patience", _c.c_int32), ("mode", _c.c_int32),
                ("depth_mask", _c.c_int32), ("depth_tau", _c.c_float), ("depth_h", _c.c_int32), ("depth_w", _c.c_int32),
                ("depth_stride", _c.c_int32), ("fuse", _c.c_int32), ("images", _c.c_int32)]


# name -> (restype, argtypes); every symbol include/piccolo_hip.h declares
SIGNATURES = {
    "pcl_abi_version": (_int, []),
    "pcl_error_string": (_c.c_char_p, [_int]),
    "pcl_source_hash": (_c.c_char_p, []),
    "pcl_library_hash": (_c.c_char_p, []),
    "pcl_cloud_stride": (_i64, [_i64]),
    "pcl_cloud_bytes": (_sz, [_i64]),
    "pcl_cloud_pack": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "pcl_morton_keys": (_int, [_vp, _i64, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _vp, _vp]),
    "pcl_pano_bytes": (_sz, [_int, _int, _int]),
    "pcl_pano_pack": (_int, [_vp, _int, _int, _vp, _vp]),
    "pcl_cloud_order_workspace_bytes": (_sz, [_i64]),
    "pcl_cloud_order": (_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "pcl_pano_pack_u8": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "pcl_pano_pack_u8p": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "pcl_pano_pack_u8v": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "pcl_pano_pack_f16": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "pcl_loss_workspace_bytes": (_sz, [_i64, _int]),
    "pcl_sampling_loss": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _vp, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "pcl_loss_depth_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int, _int, _int]),
    "pcl_sampling_loss_depth": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _vp, _int, _int, _int, _int, _c.c_float, _int, _vp, _vp, _sz, _vp]),
    "pcl_depth_default": (_int, [_i64, _int, _int, _int, _c.POINTER(_int), _c.POINTER(_int), _c.POINTER(_c.c_float), _c.POINTER(_int)]),
    "pcl_gd_state_bytes": (_sz, [_int]),
    "pcl_gd_workspace_bytes": (_sz, [_i64, _int, _int, _int, _c.POINTER(GdHyper)]),
    "pcl_hist_trim_workspace_bytes": (_sz, [_int, _int, _int, _int, _int]),
    "pcl_hist_trim_workspace_bytes_n": (_sz, [_i64, _int, _int, _int, _int, _int]),
    "pcl_hist_trim_scores": (_int, [_vp, _i64, _vp, _int, _int, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcl_color_template_bytes": (_sz, [_i64]),
    "pcl_color_template_workspace_bytes": (_sz, [_i64]),
    "pcl_color_template_build": (_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "pcl_color_workspace_bytes": (_sz, []),
    "pcl_color_match": (_int, [_vp, _int, _int, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pcl_color_mod": (_int, [_vp, _int, _int, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "pcl_histogram_workspace_bytes": (_sz, [_int, _int, _int]),
    "pcl_histogram": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _c.c_float, _vp, _vp, _sz, _vp]),
    "pcl_histogram_intersection": (_int, [_vp, _vp, _int, _int, _vp, _vp]),
    "pcl_cloud_txt_rows": (_i64, [_c.c_char_p]),
    "pcl_cloud_txt_read": (_i64, [_c.c_char_p, _i64, _int, _vp, _int]),
    "pcl_hist_trim_reduce": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "pcl_hist_trim_images_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int, _int, _int]),
    "pcl_hist_trim_scores_images": (_int, [_vp, _i64, _c.POINTER(_vp), _int, _int, _int, _int, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcl_hist_trim_reduce_images": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp]),
    "pcl_trim_groups_bytes": (_sz, [_int]),
    "pcl_trim_groups": (_int, [_vp, _int, _vp, _vp]),
    "pcl_trim_loss_workspace_bytes": (_sz, [_i64, _int, _int]),
    "pcl_trim_loss": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcl_trim_order_bytes": (_sz, [_i64, _int, _int]),
    "pcl_trim_order_workspace_bytes": (_sz, [_i64, _int, _int]),
    "pcl_trim_order": (_int, [_vp, _i64, _int, _int, _int, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _sz, _vp]),
    "pcl_trim_loss_images_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "pcl_trim_loss_images": (_int, [_vp, _i64, _c.POINTER(_vp), _int, _int, _int, _int, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcl_depth_workspace_bytes": (_sz, [_int, _int, _int]),
    "pcl_depth_mask": (_int, [_vp, _i64, _vp, _vp, _int, _int, _int, _c.c_float, _int, _vp, _vp, _sz, _vp]),
    "pcl_gd_init": (_int, [_vp, _vp, _vp, _int, _c.POINTER(GdHyper), _vp]),
    "pcl_gd_run": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _int, _vp, _c.POINTER(GdHyper), _int, _vp, _vp, _sz, _vp, _vp]),
    "pcl_timer_create": (_vp, [_int]),
    "pcl_timer_destroy": (None, [_vp]),
    "pcl_timer_reset": (None, [_vp]),
    "pcl_timer_set_stride": (None, [_vp, _int]),
    "pcl_timer_read": (_int, [_vp, _c.POINTER(_dbl), _c.POINTER(_int)]),
    "pcl_timer_calibrate": (_int, [_vp, _int, _c.POINTER(_dbl), _vp]),
    "pcl_gd_result": (_int, [_vp, _int, _vp, _vp]),
    "pcl_gd_step_from_grads": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcl_gd_plan": (_int, [_i64, _int, _c.POINTER(_int), _c.POINTER(_int), _c.POINTER(_int)]),
    "pcl_gd_plan_hyper": (_int, [_i64, _int, _c.POINTER(GdHyper), _c.POINTER(_int), _c.POINTER(_int), _c.POINTER(_int)]),
    "pcl_gd_set_panos": (_int, [_vp, _vp, _int, _vp]),
    "pcl_gd_set_pano_groups": (_int, [_vp, _c.POINTER(_c.c_uint64), _int, _int, _vp]),
    "pcl_gd_winner": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pcl_select_poses": (_int, [_vp, _int, _int, _int, _int, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp]),
    "pcl_cloud2idx": (_int, [_vp, _i64, _vp, _vp]),
    "pcl_sample_from_img": (_int, [_vp, _int, _int, _int, _vp, _i64, _vp, _vp]),
    "pcl_cloud2idx_backward": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "pcl_sample_from_img_backward": (_int, [_vp, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _vp]),
    "pcl_rot_from_ypr": (_int, [_vp, _int, _vp, _vp]),
    "pcl_quantile_workspace_bytes": (_sz, []),
    "pcl_quantile_box": (_int, [_vp, _i64, _dbl, _vp, _vp, _vp]),
    "pcl_scatter_min_depth": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "pcl_scatter_min_unpack": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp]),
    "pcl_make_pano": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "pcl_transform_cloud": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
}

_lib = None


class PiccoloHipError(RuntimeError):
    pass


def so_path():
    return os.environ.get("PCL_SO", _build.SO)       # PCL_SO: A/B a differently built library (experiments)


def load():
    """Load libpiccolo_hip.so (never builds implicitly on the hot path; raises if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; import it first so this process has ONE runtime (ours resolves to the
    # already-loaded libamdhip64 by soname).  Loading ours first leaves two runtimes, one of which sees no device.
    import torch  # noqa: F401
    path = so_path()
    if not os.path.exists(path):
        raise PiccoloHipError(
            "piccolo_amd: %s not found. The MI355X HIP library is the only back end (no CPU fallback); "
            "build it with `python -m piccolo_amd.build`." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.pcl_abi_version() != ABI_VERSION:
        raise PiccoloHipError("piccolo_amd: ABI mismatch: library %d, binding %d" % (lib.pcl_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise PiccoloHipError("%s failed: %s (code %d)" % (what, load().pcl_error_string(rc).decode(), rc))
