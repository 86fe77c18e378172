"""ctypes front-end of the plain-C oracle (``oracle/csrc/oracle.c``).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "_build" / "liboracle.so"
_lib = None

R_NORMALS, R_DEPTH, R_MASK, R_QUANT8, R_MSAA4, R_TEX_ANISO = 1, 2, 4, 8, 32, 64


def build(force: bool = False) -> Path:
    src = _DIR / "csrc" / "oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_DIR), "-s", "-B"])
    return _LIB_PATH


class _Meshes(C.Structure):
    _fields_ = [
        ("verts", C.c_void_p),
        ("normals", C.c_void_p),
        ("uvs", C.c_void_p),
        ("colors", C.c_void_p),
        ("faces", C.c_void_p),
        ("tex", C.c_void_p),
        ("obj", C.c_void_p),
        ("n_obj", C.c_int),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.hp_oracle_roi_align.restype = None
        _lib.hp_oracle_rasterize.restype = None
    return _lib


class RasterConventions(C.Structure):
    """``hp_oracle_raster_conventions`` (oracle.c) -- field for field ``hp_raster_conventions`` of the product header."""
    _fields_ = [("msaa_x", C.c_float * 4), ("msaa_y", C.c_float * 4), ("aniso_max", C.c_int), ("aniso_round", C.c_int),
                ("lod_from", C.c_int), ("lod_bias", C.c_float), ("aniso_ratio_bias", C.c_float), ("normal_axis", C.c_int * 3), ("normal_sign", C.c_float * 3)]


def set_raster_conventions(conv=None) -> None:
    """``conv``: ``None`` (defaults) or a dict with any of msaa_x, msaa_y, aniso_max, aniso_round, lod_from, lod_bias,
    normal_axis, normal_sign (missing keys = defaults) -- the same dict ``happypose_amd.ops.set_raster_conventions`` takes."""
    if conv is None:
        lib().hp_oracle_set_raster_conventions(None)
        return
    d = dict(msaa_x=(0.375, 0.875, 0.125, 0.625), msaa_y=(0.125, 0.375, 0.625, 0.875), aniso_max=16, aniso_round=0, lod_from=0,
             lod_bias=0.0, aniso_ratio_bias=0.0, normal_axis=(0, 1, 2), normal_sign=(1.0, -1.0, -1.0))
    d.update(conv)
    c = RasterConventions((C.c_float * 4)(*d["msaa_x"]), (C.c_float * 4)(*d["msaa_y"]), int(d["aniso_max"]), int(d["aniso_round"]),
                          int(d["lod_from"]), float(d["lod_bias"]), float(d["aniso_ratio_bias"]), (C.c_int * 3)(*d["normal_axis"]), (C.c_float * 3)(*d["normal_sign"]))
    lib().hp_oracle_set_raster_conventions(C.byref(c))


def set_threads(n: int) -> None:
    """Thread count of the C oracle's OpenMP loops."""
    lib().hp_oracle_set_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def roi_align(images, boxes, im_ids, output_size=(240, 320), sampling_ratio=4):
    """images [Bi,C,H,W] f32, boxes [n,4] xyxy, im_ids [n] -> [n,C,oh,ow] f32."""
    images = np.ascontiguousarray(images, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    im_ids = np.ascontiguousarray(im_ids, dtype=np.int32)
    Bi, Cc, H, W = images.shape
    n = boxes.shape[0]
    oh, ow = output_size
    out = np.empty((n, Cc, oh, ow), dtype=np.float32)
    lib().hp_oracle_roi_align(
        _p(images), Bi, Cc, H, W, _p(boxes), _p(im_ids), n, oh, ow, int(sampling_ratio), _p(out)
    )
    return out


def crop_images(images, boxes, im_ids, output_size=(240, 320)):
    """TB/lib3d/cropping.py:155-197 incl. the RGB-D rule: depth (channel 3) is zeroed
    wherever the roi-aligned validity mask ``depth > 0`` is ``< 0.99``."""
    crops = roi_align(images, boxes, im_ids, output_size, 4)
    if images.shape[1] == 4:
        valid = (np.asarray(images)[:, 3:4] > 0).astype(np.float32)
        vcrop = roi_align(valid, boxes, im_ids, output_size, 4)
        crops[:, 3:4] *= (vcrop >= 0.99).astype(np.float32)
    return crops


def rasterize(meshes, obj_ids, TCO, K, resolution, render_normals=False, render_depth=False,
              render_binary_mask=False, ambient=None, light_pos=None, light_col=None,
              quant8=True, msaa=False, aniso=False):
    """``msaa``: 4x multisampled colour / normal buffers (``HP_R_MSAA4`` in oracle.c); ``aniso``: mip-mapped trilinear +
    anisotropic-16 texture filtering (``HP_R_TEX_ANISO``).
    Returns dict(rgbs [n,3,h,w], normals, depths [n,1,h,w], binary_masks bool) -- the
    shapes/dtypes of ``BatchRenderOutput`` (TB/renderer/types.py:45-56).

    ``meshes`` is any object with numpy attributes ``verts normals uvs colors faces tex
    obj`` (the packed layout documented in ``oracle/csrc/oracle.c``)."""
    h, w = resolution
    obj_ids = np.ascontiguousarray(obj_ids, dtype=np.int32)
    n = len(obj_ids)
    TCO = np.ascontiguousarray(TCO, dtype=np.float32).reshape(n, 16)
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(n, 9)
    if ambient is None:
        ambient = np.ones((n, 3), np.float32)
    ambient = np.ascontiguousarray(ambient, dtype=np.float32).reshape(n, 3)
    n_lights = 0
    if light_pos is not None:
        light_pos = np.ascontiguousarray(light_pos, dtype=np.float32)
        light_col = np.ascontiguousarray(light_col, dtype=np.float32)
        n_lights = light_pos.shape[1]
    arrs = dict(
        verts=np.ascontiguousarray(meshes.verts, np.float32),
        normals=np.ascontiguousarray(meshes.normals, np.float32),
        uvs=np.ascontiguousarray(meshes.uvs, np.float32),
        colors=np.ascontiguousarray(meshes.colors, np.uint8),
        faces=np.ascontiguousarray(meshes.faces, np.int32),
        tex=np.ascontiguousarray(meshes.tex, np.uint8),
        obj=np.ascontiguousarray(meshes.obj, np.int64),
    )
    M = _Meshes(*[_p(arrs[k]) for k in ("verts", "normals", "uvs", "colors", "faces", "tex", "obj")],
                len(arrs["obj"]))
    flags = (R_NORMALS if render_normals else 0) | (R_DEPTH if render_depth else 0) | \
        (R_MASK if render_binary_mask else 0) | (R_QUANT8 if quant8 else 0) | (R_MSAA4 if msaa else 0) | \
        (R_TEX_ANISO if aniso else 0)
    if render_binary_mask:
        assert render_depth, "Binary mask can only be rendered if depth is rendered"
    rgb = np.empty((n, 3, h, w), np.float32)
    nrm = np.empty((n, 3, h, w), np.float32) if render_normals else None
    dep = np.empty((n, 1, h, w), np.float32) if render_depth else None
    msk = np.empty((n, 1, h, w), np.uint8) if render_binary_mask else None
    i64 = C.c_int64
    lib().hp_oracle_rasterize(
        C.byref(M), n, _p(obj_ids), _p(TCO), _p(K), _p(ambient), n_lights, _p(light_pos),
        _p(light_col), h, w, flags, _p(rgb), _p(nrm), _p(dep), _p(msk),
        i64(3 * h * w), i64(h * w), i64(w), i64(1), i64(h * w), i64(w), i64(1),
    )
    return dict(rgbs=rgb, normals=nrm, depths=dep,
                binary_masks=None if msk is None else msk.astype(bool))
