"""The hot-path operators of the PyTorch dispatcher: ``torch.ops.happypose_amd.*``, a COMPILED operator library.

``north_star`` names the product as PyTorch-ROCm custom ops; the drop-in boundary is the C-ABI
(``include/happypose_amd.h``).  ``csrc/torch_library.cpp`` (``TORCH_LIBRARY(happypose_amd)``; built by
``happypose_amd/build.py`` into ``lib/libhappypose_amd_torch.so``) registers the schemas with kernels for the
``CUDA`` dispatch key (HIP on ROCm) and shape functions for ``Meta``: each kernel checks its arguments, allocates the
outputs and makes ONE ``hp_*`` call on torch's current stream.  This module only loads that library (and fails
loudly when it has not been built) and issues the integer handles.  A CPU tensor raises ``NotImplementedError``
from the dispatcher -- there is no CPU fallback behind these names.

Objects that live behind C handles (the mesh store, a network plan) are passed as integers issued by
:func:`ticket` (the handle's address, announced to the library and withdrawn when the owning object dies: a
stale integer raises ``ValueError``).  ``pose_prep`` takes the sub-sampled point ids of the store as tensors
(``MeshStore.point_ids``); :func:`pose_prep` fills them in.

Reference counterparts: ``crop_images`` (``TB/lib3d/cropping.py:155-197``), ``Panda3dBatchRenderer.render``
(``TB/renderer/panda3d_batch_renderer.py:271-349``), ``PosePredictor.update_pose``
(``MP/models/pose_rigid.py:456-481``), ``PosePredictor.crop_inputs`` / ``compute_crops_multiview`` /
``make_TCO_multiview`` (``:235-335``, ``TB/lib3d/multiview.py:166-251``), ``PosePredictor.net_forward``
(``:352-374``).
"""

from __future__ import annotations

import weakref
from pathlib import Path
from typing import List

import torch

from . import ops

NAMESPACE = "happypose_amd"
LIBRARY = Path(__file__).resolve().parent / "lib" / "libhappypose_amd_torch.so"
OPS = ("crop_roi_align", "pose_update", "rasterize", "pose_prep", "net_forward", "conv2d_nhwc")
_KIND = {ops.MeshStore: 1, ops.Net: 2}


def _load() -> None:
    if not LIBRARY.exists():
        raise ImportError(f"{LIBRARY} is missing: build it with `python -m happypose_amd.build` (there is no Python-side "
                          "fallback for torch.ops.happypose_amd.*)")
    torch.ops.load_library(str(LIBRARY))
    missing = [n for n in OPS if not hasattr(torch.ops.happypose_amd, n)]
    if missing:
        raise ImportError(f"{LIBRARY} does not register {missing}")


_load()


def ticket(obj) -> int:
    """The integer that names a ``MeshStore`` / ``Net`` in an operator call (valid while ``obj`` lives)."""
    t = getattr(obj, "_torch_ops_ticket", None)
    if t is None:
        kind = next((k for c, k in _KIND.items() if isinstance(obj, c)), None)
        if kind is None:
            raise TypeError(f"happypose_amd op: no operator takes a {type(obj).__name__}")
        t = int(obj.handle.value)
        torch.ops.happypose_amd.register_handle(t, kind)
        weakref.finalize(obj, torch.ops.happypose_amd.release_handle, t)
        obj._torch_ops_ticket = t
    return t


def pose_prep(store: ops.MeshStore, TCO, K, im_ids, obj_ids, im_size, crop_size=(240, 320), multiview_type: str = "TCO",
              normalize: bool = False, n_points: int = 2000, n_points_extra: int = 200, lamb: float = 1.4) -> List[torch.Tensor]:
    """``torch.ops.happypose_amd.pose_prep`` with the store's deterministic point sub-samples filled in
    (``[TCO, tCR, TCV_O, boxes_rend, boxes_crop, K_crop]``)."""
    multi = ops.MULTIVIEW[multiview_type][1] > 1
    return torch.ops.happypose_amd.pose_prep(ticket(store), TCO, K, im_ids, obj_ids, store.point_ids(n_points),
                                            store.point_ids(n_points_extra) if multi else None, im_size[0], im_size[1],
                                            crop_size[0], crop_size[1], multiview_type, normalize, lamb)
