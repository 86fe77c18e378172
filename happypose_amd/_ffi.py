"""ctypes binding of ``lib/libhappypose_amd.so`` (the C ABI of ``include/happypose_amd.h``).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
product entry point raises.  (The CPU restatement lives in ``oracle/`` and is
test infrastructure only.)
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import os as _os

_LIB_PATH = Path(_os.environ.get("HAPPYPOSE_AMD_LIB", Path(__file__).resolve().parent / "lib" / "libhappypose_amd.so"))
_lib = None

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_u8p = C.c_void_p


class Strides(C.Structure):
    """``hp_strides``: element strides (item, view, channel, row, col)."""

    _fields_ = [("s_item", C.c_int64), ("s_view", C.c_int64), ("s_chan", C.c_int64),
                ("s_row", C.c_int64), ("s_col", C.c_int64)]


class InputLayout(C.Structure):
    """``hp_input_layout``: where view ``v`` of an item writes its render / crop channels inside a pixel record."""

    _fields_ = [("view_c0", C.c_int * 8), ("crop_c0", C.c_int * 8), ("crop_src0", C.c_int * 8), ("crop_n", C.c_int * 8)]


class HipLibraryError(RuntimeError):
    pass


_PROTOS = {
    "hp_version": (C.c_int, []),
    "hp_last_error": (C.c_char_p, []),
    "hp_device_count": (C.c_int, []),
    "hp_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "hp_mesh_store_create": (C.c_void_p, [c_f32p, c_f32p, c_f32p, c_u8p, C.c_int64, c_i32p, C.c_int64,
                                           c_u8p, C.c_int64, C.c_void_p, C.c_int, c_f32p, C.c_int]),
    "hp_mesh_store_destroy": (None, [C.c_void_p]),
    "hp_mesh_store_points": (C.c_void_p, [C.c_void_p]),
    "hp_mesh_store_reserve_raster": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hp_mesh_store_scratch_generation": (C.c_int64, [C.c_void_p]),
    "hp_rasterize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int,
                               c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p,
                               C.POINTER(Strides), c_f32p, C.POINTER(Strides), c_u8p, c_f32p, C.c_int,
                               C.c_void_p]),
    "hp_render_inputs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int,
                                   C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, c_f32p, C.c_int,
                                   C.c_void_p, C.c_int, C.POINTER(InputLayout), C.c_void_p]),
    "hp_pose_prep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, c_i32p,
                               c_i32p, c_i32p, C.c_int, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                               C.c_void_p]),
    "hp_pose_prep_views": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, c_i32p,
                                     c_i32p, c_i32p, C.c_int, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                     C.c_void_p]),
    "hp_crop_roi_align": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, c_f32p, C.POINTER(Strides), c_f32p,
                                    C.c_int, C.c_void_p]),
    "hp_crop_roi_align_f16": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Strides), c_f32p,
                                        C.c_int, C.c_void_p]),
    "hp_pose_update": (C.c_int, [C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "hp_tco_init_autodepth": (C.c_int, [C.c_void_p, C.c_int, c_f32p, C.c_int, c_i32p, c_f32p, C.c_int, c_i32p, c_i32p,
                                        c_f32p, C.c_int, c_i32p, c_i32p, C.c_int, c_f32p, C.c_void_p]),
    "hp_net_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "hp_net_destroy": (None, [C.c_void_p]),
    "hp_net_add_conv": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p] + [C.c_int] * 11),
    "hp_net_add_output": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hp_net_input_channels_padded": (C.c_int, [C.c_void_p]),
    "hp_net_set_param": (C.c_int, [C.c_void_p, C.c_char_p, c_f32p, C.c_int64]),
    "hp_net_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_net_precision": (C.c_int, [C.c_void_p]),
    "hp_net_output_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hp_net_input_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hp_net_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_net_forward": (C.c_int, [C.c_void_p, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "hp_net_forward_f16in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "hp_net_input_channels_f16": (C.c_int, [C.c_void_p]),
    "hp_net_flops_per_sample": (C.c_double, [C.c_void_p]),
    "hp_net_n_feature_maps": (C.c_int, [C.c_void_p]),
    "hp_net_feature_map": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int)]),
    "hp_net_copy_feature_map": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "hp_detector_preprocess": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), c_f32p,
                                         C.c_void_p]),
    "hp_detector_preprocess_resize": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(C.c_float), C.POINTER(C.c_float), c_f32p, C.c_void_p]),
    "hp_net_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_conv_occupancy": (C.c_int, [C.c_int]),
    "hp_scratch_launches": (C.c_longlong, []),
    "hp_probe_mfma_rate": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "hp_net_set_tail_split": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_net_set_act_scale": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_net_set_conv_algo": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_net_status": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "hp_net_force_exact": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_mesh_store_set_raster_conventions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_mesh_store_get_raster_conventions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hp_mesh_store_set_backface_culling": (C.c_int, [C.c_void_p, C.c_int]),
    "hp_mesh_store_get_backface_culling": (C.c_int, [C.c_void_p]),
    "hp_profile_mark_reference": (C.c_int, [C.c_void_p]),
    "hp_net_profile_intervals": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "hp_conv_select_algo": (C.c_int, [C.c_int]),
    "hp_net_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hp_rpn_decode": (C.c_int, [c_f32p, c_i32p, C.c_int, c_f32p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float,
                                C.c_float, C.c_float, c_f32p, c_f32p, c_u8p, C.c_void_p]),
    "hp_nms": (C.c_int, [c_f32p, c_i32p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "hp_roi_align_levels": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                      C.c_int, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p, C.c_void_p]),
    "hp_box_postprocess": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p,
                                     c_f32p, C.c_void_p]),
    "hp_paste_masks": (C.c_int, [c_f32p, C.c_int, c_i32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "hp_icp_refine": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p,
                                c_f32p, c_f32p, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p, C.c_void_p, c_f32p,
                                C.c_void_p]),
    "hp_conv2d_nhwc_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "hp_conv2d_nhwc": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                 c_f32p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib_path() -> Path:
    return _LIB_PATH


def lib():
    """Load (once) and return the HIP library; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HipLibraryError(
                f"{_LIB_PATH} not found: build it with `python -m happypose_amd.build` "
                "(__graft_entry__.build()).  happypose_amd has no CPU fallback."
            )
        # torch owns the device memory and streams we are handed, so the library must bind to
        # the SAME HIP runtime instance: import torch first (its bundled libamdhip64 is then
        # the one already loaded when ours resolves its dependency).
        import torch  # noqa: F401

        try:
            handle = C.CDLL(str(_LIB_PATH))
        except OSError as e:  # e.g. libamdhip64 missing
            raise HipLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().hp_last_error().decode(errors="replace")
        if rc == -1:  # HP_ERR_ARG: the reference raises AssertionError on shape errors
            raise AssertionError(f"{what}: {msg}")
        raise HipLibraryError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Current torch HIP stream as ``void*`` (torch is plumbing: memory + streams)."""
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
