"""The hot-path operators registered with the PyTorch dispatcher (``torch.ops.happypose_amd.*``).

``north_star`` names the product as PyTorch-ROCm custom ops; the drop-in boundary is the C-ABI
(``include/happypose_amd.h``) and ``happypose_amd.ops`` binds it with ctypes.  This module puts the same
calls behind dispatcher schemas so that they can be called as ``torch.ops.happypose_amd.<name>``,
appear in profiler traces under their own names and carry shape functions (``Meta`` kernels) for
fake-tensor tracing.  Only the ``CUDA`` dispatch key (HIP on ROCm) has kernels: a CPU tensor raises
``NotImplementedError`` from the dispatcher -- there is no CPU fallback behind these names.

Objects that live behind C handles (the mesh store, a network plan) are passed as integer tickets
issued by :func:`ticket`; the registry holds weak references, so a ticket dies with its object.

Reference counterparts: ``crop_images`` (``TB/lib3d/cropping.py:155-197``), ``Panda3dBatchRenderer.render``
(``TB/renderer/panda3d_batch_renderer.py:271-349``), ``PosePredictor.update_pose``
(``MP/models/pose_rigid.py:456-481``), ``PosePredictor.crop_inputs`` / ``compute_crops_multiview`` /
``make_TCO_multiview`` (``:235-335``, ``TB/lib3d/multiview.py:166-251``), ``PosePredictor.net_forward``
(``:352-374``).
"""

from __future__ import annotations

import itertools
import weakref
from typing import List, Optional

import torch

from . import ops

NAMESPACE = "happypose_amd"
_lib = torch.library.Library(NAMESPACE, "DEF")
_tickets: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_counter = itertools.count(1)


def ticket(obj) -> int:
    """Integer stand-in for a ``MeshStore`` / ``Net`` in an operator call (valid while ``obj`` lives)."""
    t = getattr(obj, "_torch_ops_ticket", None)
    if t is None:
        t = next(_counter)
        obj._torch_ops_ticket = t
    _tickets[t] = obj
    return t


def _resolve(t: int, kind):
    obj = _tickets.get(int(t))
    if not isinstance(obj, kind):
        raise ValueError(f"happypose_amd op: ticket {t} does not name a live {kind.__name__}")
    return obj


def _define(schema: str, cuda_impl, meta_impl) -> None:
    name = schema.split("(", 1)[0]
    _lib.define(schema)
    _lib.impl(name, cuda_impl, "CUDA")
    _lib.impl(name, meta_impl, "Meta")


# ---- crop -------------------------------------------------------------------------------------
def _crop(images, boxes, im_ids, out_h: int, out_w: int, sampling_ratio: int = 4):
    return ops.crop_roi_align(images, boxes, im_ids, (out_h, out_w), sampling_ratio)


def _crop_meta(images, boxes, im_ids, out_h: int, out_w: int, sampling_ratio: int = 4):
    return images.new_empty((boxes.shape[0], images.shape[1], out_h, out_w))


_define("crop_roi_align(Tensor images, Tensor boxes, Tensor im_ids, int out_h, int out_w, int sampling_ratio=4) -> Tensor",
        _crop, _crop_meta)


# ---- pose update ------------------------------------------------------------------------------
def _pose_update(TCO, K_crop, pose9, tCR: Optional[torch.Tensor] = None):
    return ops.pose_update(TCO, K_crop, pose9, tCR)


def _pose_update_meta(TCO, K_crop, pose9, tCR: Optional[torch.Tensor] = None):
    return torch.empty_like(TCO)


_define("pose_update(Tensor TCO, Tensor K_crop, Tensor pose9, Tensor? tCR=None) -> Tensor", _pose_update, _pose_update_meta)


# ---- rasteriser -------------------------------------------------------------------------------
def _rasterize(store: int, obj_ids, TCO, K, height: int, width: int, normals: bool = False, depth: bool = False) -> List[torch.Tensor]:
    rgb, nrm, dep, _ = ops.rasterize(_resolve(store, ops.MeshStore), obj_ids, TCO, K, (height, width),
                                     render_normals=normals, render_depth=depth)
    return [t for t in (rgb, nrm, dep) if t is not None]


def _rasterize_meta(store: int, obj_ids, TCO, K, height: int, width: int, normals: bool = False, depth: bool = False) -> List[torch.Tensor]:
    n = TCO.shape[0]
    out = [TCO.new_empty((n, 3, height, width))]
    if normals:
        out.append(TCO.new_empty((n, 3, height, width)))
    if depth:
        out.append(TCO.new_empty((n, 1, height, width)))
    return out


_define("rasterize(int store, Tensor obj_ids, Tensor TCO, Tensor K, int height, int width, bool normals=False, "
        "bool depth=False) -> Tensor[]", _rasterize, _rasterize_meta)


# ---- per-iteration geometry ---------------------------------------------------------------------
_PREP_ORDER = ("TCO", "tCR", "TCV_O", "boxes_rend", "boxes_crop", "K_crop")


def _pose_prep(store: int, TCO, K, im_ids, obj_ids, im_h: int, im_w: int, crop_h: int, crop_w: int,
               multiview_type: str = "TCO", normalize: bool = False, lamb: float = 1.4) -> List[torch.Tensor]:
    out = ops.pose_prep(_resolve(store, ops.MeshStore), TCO, K, im_ids, obj_ids, (im_h, im_w), (crop_h, crop_w),
                        multiview_type, normalize, lamb=lamb)
    return [out[k] for k in _PREP_ORDER]


def _pose_prep_meta(store: int, TCO, K, im_ids, obj_ids, im_h: int, im_w: int, crop_h: int, crop_w: int,
                    multiview_type: str = "TCO", normalize: bool = False, lamb: float = 1.4) -> List[torch.Tensor]:
    b, V = TCO.shape[0], ops.MULTIVIEW[multiview_type][1]
    e = TCO.new_empty
    return [e((b, 4, 4)), e((b, 3)), e((b, V, 4, 4)), e((b, 4)), e((b, 4)), e((b, V, 3, 3))]


_define("pose_prep(int store, Tensor TCO, Tensor K, Tensor im_ids, Tensor obj_ids, int im_h, int im_w, int crop_h, "
        "int crop_w, str multiview_type='TCO', bool normalize=False, float lamb=1.4) -> Tensor[]",
        _pose_prep, _pose_prep_meta)


# ---- network --------------------------------------------------------------------------------
def _net_forward(net: int, x) -> List[torch.Tensor]:
    n = _resolve(net, ops.Net)
    pose, logits, _ = n.forward(x, want_pose=n.pose_dim > 0, want_logits=n.n_logits > 0)
    return [t for t in (pose, logits) if t is not None]


def _net_forward_meta(net: int, x) -> List[torch.Tensor]:
    n = _resolve(net, ops.Net)
    out = []
    if n.pose_dim > 0:
        out.append(x.new_empty((x.shape[0], n.pose_dim), dtype=torch.float32))
    if n.n_logits > 0:
        out.append(x.new_empty((x.shape[0], n.n_logits), dtype=torch.float32))
    return out


_define("net_forward(int net, Tensor x) -> Tensor[]", _net_forward, _net_forward_meta)


# ---- single conv layer (parity tests / layer-level users) ---------------------------------------
def _conv(x, w, stride: int, pad: int, bias=None, residual=None, pre_scale=None, pre_shift=None, act: int = 0):
    fn = ops.conv2d_nhwc_f16 if x.dtype == torch.float16 else ops.conv2d_nhwc
    return fn(x, w, stride, pad, bias, residual, pre_scale, pre_shift, act)


def _conv_meta(x, w, stride: int, pad: int, bias=None, residual=None, pre_scale=None, pre_shift=None, act: int = 0):
    n, h, wd, _ = x.shape
    cout, kh, kw, _ = w.shape
    return x.new_empty((n, (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1, cout))


_define("conv2d_nhwc(Tensor x, Tensor w, int stride, int pad, Tensor? bias=None, Tensor? residual=None, "
        "Tensor? pre_scale=None, Tensor? pre_shift=None, int act=0) -> Tensor", _conv, _conv_meta)

OPS = ("crop_roi_align", "pose_update", "rasterize", "pose_prep", "net_forward", "conv2d_nhwc")
