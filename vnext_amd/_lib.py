"""ctypes binding of libvnext_hip.so (include/vnext_hip.h).

The library is the product: there is no CPU or PyTorch fallback behind these
calls.  If the .so is missing `lib()` raises; run `python -m vnext_amd.build`
(or `__graft_entry__.build()`) first.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvnext_hip.so")
# the development build (-DVNX_DEV_VARIANTS + tools/experiments/msda_tile): forced kernel variants for tests and A/B timing
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libvnext_hip_dev.so")

VNX_F32, VNX_F64, VNX_BF16, VNX_F16 = 0, 1, 2, 3
VNX_OK = 0
ABI_VERSION = 15
MSDA_LEVELS_PACKED = 1
MSDA_REF_F32 = 0x100       # or-ed into ref_dim of vnx_msda_fused_*: fp32 reference points beside 16-bit offsets / logits
MSDA_FORK = 2              # vnx_msda_backward: grad_value kernel on the library's side stream (include/vnext_hip.h)

_vp, _i, _sz, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/vnext_hip.h declaration by declaration
SIGNATURES = {
    "vnx_abi_version": (_i, []),
    "vnx_status_string": (ctypes.c_char_p, [_i]),
    "vnx_last_error": (ctypes.c_char_p, []),
    "vnx_msda_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 7 + [_vp]),
    "vnx_msda_backward_workspace_bytes": (_sz, [_i] * 10),
    "vnx_msda_backward": (_i, [_i, _i] + [_vp] * 9 + [_i] * 8 + [_vp, _sz, _vp]),
    "vnx_msda_fused_forward": (_i, [_i, _i] + [_vp] * 7 + [_i] * 9 + [_vp]),
    "vnx_msda_fused_backward_workspace_bytes": (_sz, [_i] * 7),
    "vnx_msda_fused_backward": (_i, [_i, _i] + [_vp] * 11 + [_i] * 9 + [_vp, _sz, _vp]),
    "vnx_dynamic_mask_head_forward": (_i, [_i] + [_vp] * 5 + [_i] * 7 + [_vp]),
    "vnx_dynamic_mask_head_backward": (_i, [_i] + [_vp] * 8 + [_i] * 7 + [_vp]),
    "vnx_dynamic_mask_head_forward_train": (_i, [_i] + [_vp] * 8 + [_i] * 7 + [_vp]),
    "vnx_dynamic_mask_head_backward_zeroed": (_i, [_i] + [_vp] * 8 + [_i] * 7 + [_vp]),
    "vnx_reid_similarity": (_i, [_i] + [_vp] * 3 + [_i] * 7 + [_vp]),
    "vnx_reid_bisoftmax": (_i, [_i] + [_vp] * 2 + [_i] * 4 + [_vp]),
    "vnx_mask_intersections_workspace_bytes": (_sz, [_i, _i]),
    "vnx_mask_intersections": (_i, [_vp, _i, _i, _vp, _vp, _sz, _vp]),
    "vnx_tracker_state_bytes": (_sz, [_vp]),
    "vnx_tracker_reset": (_i, [_vp, _vp, _vp]),
    "vnx_tracker_frame_workspace_bytes": (_sz, [_vp, _i, _i]),
    "vnx_tracker_frame": (_i, [_vp] * 6 + [_i] * 3 + [_vp, _vp, _sz, _vp]),
    "vnx_add_dropout_layernorm_partial_bytes": (_sz, []),
    "vnx_add_dropout_layernorm_forward": (_i, [_i] + [_vp] * 8 + [_ll, _i, ctypes.c_float, ctypes.c_float, ctypes.c_ulonglong, _vp, _vp]),
    "vnx_add_dropout_layernorm_backward": (_i, [_i] + [_vp] * 10 + [_ll, _i, ctypes.c_float, ctypes.c_ulonglong, _vp, _vp]),
    "vnx_bias_relu_dropout_partial_bytes": (_sz, [_i]),
    "vnx_refine_boxes_forward": (_i, [_i, _vp, _vp, _vp, _ll, _i, ctypes.c_float, _vp]),
    "vnx_refine_boxes_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _ll, _i, ctypes.c_float, _vp]),
    "vnx_time_weighted_sum_forward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vnx_time_weighted_sum_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vnx_query_self_attention_forward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_float, ctypes.c_ulonglong, _vp, _vp]),
    "vnx_query_self_attention_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, ctypes.c_float,
                                               ctypes.c_ulonglong, _vp, _vp]),
    "vnx_bias_relu_dropout_forward": (_i, [_i, _vp, _vp, _vp, _ll, _i, _i, ctypes.c_float, ctypes.c_ulonglong, _vp, _vp]),
    "vnx_bias_relu_dropout_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, ctypes.c_float, _vp]),
}
# measurement aids of include/vnext_hip_debug.h (bench.py, tools/): not part of the drop-in boundary
DEBUG_SIGNATURES = {
    "vnx_debug_wall_clock_khz": (_i, []),
    "vnx_debug_row_gather_probe": (_i, [_vp, _sz, _vp, _sz, _i, _vp, _vp]),
    "vnx_debug_gvtiles_units": (_i, [_vp, _i, _i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i),
                                    ctypes.POINTER(_ll), ctypes.POINTER(_ll)]),
    "vnx_debug_gvdirect_units": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), _vp, _vp, _vp]),
}
# include/vnext_hip_dev.h: exported by the development library only
DEV_SIGNATURES = {
    "vnx_set_kernel_variant": (None, [_i]),
    "vnx_get_kernel_variant": (_i, []),
    "vnx_debug_arm_stamps": (None, [_vp, _ll]),
    "vnx_debug_stamp_regions": (_i, [ctypes.POINTER(_i), ctypes.POINTER(_ll), ctypes.POINTER(_ll), _i]),
    "vnx_debug_read_rec_stamps": (_i, [ctypes.POINTER(ctypes.c_ulonglong), _i]),
    "vnx_debug_read_tile_stamps": (_i, [ctypes.POINTER(ctypes.c_ulonglong), _i]),
    "vnx_debug_read_gvd_stamps": (_i, [ctypes.POINTER(ctypes.c_ulonglong), _i]),
}



class TrackerConfig(ctypes.Structure):
    """`vnx_tracker_config` of include/vnext_hip.h, field by field."""
    _fields_ = [(k, _i) for k in ("capacity", "channels", "memory_len", "memo_tracklet_frames", "match_metric",
                                  "long_match", "frame_weight", "temporal_weight")] + \
               [(k, ctypes.c_float) for k in ("nms_thr_pre", "nms_thr_post", "init_score_thr", "addnew_score_thr",
                                              "match_score_thr", "memo_momentum")]


_lib = None          # the product library
_dev = None          # the development library, loaded on the first request for a kernel variant
_active_dev = False  # True while a non-zero kernel variant is set: lib() then hands out the development library


class VnextHipError(RuntimeError):
    pass


def _load(path: str, tables) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise VnextHipError(
            f"{path} is missing: the HIP library is the only implementation of this "
            "path (no CPU fallback). Build it with `python -m vnext_amd.build`.")
    # torch ships its own libamdhip64.so.7; import it first so this library binds to
    # the HIP runtime torch's allocator and streams live in.
    import torch  # noqa: F401
    cdll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    for table in tables:
        for name, (res, args) in table.items():
            fn = getattr(cdll, name)
            fn.restype = res
            fn.argtypes = args
    if cdll.vnx_abi_version() != ABI_VERSION:
        raise VnextHipError(f"ABI version {cdll.vnx_abi_version()} != {ABI_VERSION}; rebuild")
    return cdll


def product_lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("VNX_HIP_LIB") or LIB_PATH    # override: experiment builds (tools/wpe_sweep.py)
        _lib = _load(path, (SIGNATURES, DEBUG_SIGNATURES))
    return _lib


def dev_lib() -> ctypes.CDLL:
    """libvnext_hip_dev.so (include/vnext_hip_dev.h): tests and tools only."""
    global _dev
    if _dev is None:
        _dev = _load(os.environ.get("VNX_HIP_DEV_LIB") or DEV_LIB_PATH, (SIGNATURES, DEBUG_SIGNATURES, DEV_SIGNATURES))
    return _dev


def lib() -> ctypes.CDLL:
    """The library every op of this package calls: the product -- except while a test or a tool holds a non-zero
    kernel variant (set_kernel_variant), when it is the development build that has variants at all."""
    return dev_lib() if _active_dev else product_lib()


def check(status: int) -> None:
    if status != VNX_OK:
        l = lib()
        raise VnextHipError(
            f"{l.vnx_status_string(status).decode()}: {l.vnx_last_error().decode()}")


def current_stream(tensor) -> int:
    """hipStream_t of torch's current stream on the tensor's device."""
    import torch
    return torch.cuda.current_stream(tensor.device).cuda_stream


def set_kernel_variant(v: int) -> None:
    """Tests / tools: route this package's ops through the development library with kernel variant `v` (forced
    configurations, archived kernels, timing ablations: include/vnext_hip_dev.h); 0 returns to the product library,
    which has no variants."""
    global _active_dev
    v = int(v)
    if v != 0:
        dev_lib().vnx_set_kernel_variant(v)
        _active_dev = True
    else:
        if _dev is not None:
            _dev.vnx_set_kernel_variant(0)
        _active_dev = False
