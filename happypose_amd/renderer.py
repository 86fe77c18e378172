"""Drop-in for the reference's batch renderer.

``BatchRenderer.render`` has the signature and return type of
``Panda3dBatchRenderer.render`` (``TB/renderer/panda3d_batch_renderer.py:194-286``),
``BatchRenderOutput`` / ``Panda3dLightData`` mirror ``TB/renderer/types.py:45-56,
140-151`` and ``make_scene_lights`` mirrors
``TB/renderer/panda3d_scene_renderer.py:105-141``.  There are no worker
processes: one HIP launch renders every view of the call on the compute device.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .mesh_store import RigidObjectDataset

RgbaColor = Tuple[float, float, float, float]
Resolution = Tuple[int, int]  # (h, w) as the reference's callers pass it


@dataclass
class BatchRenderOutput:
    """rgbs (n,3,h,w) f32 in [0,1]; normals (n,3,h,w) f32 in [0,1] | None;
    depths (n,1,h,w) f32 metres | None; binary_masks (n,1,h,w) bool | None."""

    rgbs: torch.Tensor
    normals: Optional[torch.Tensor]
    depths: Optional[torch.Tensor]
    binary_masks: Optional[torch.Tensor]


@dataclass
class Panda3dLightData:
    """``light_type``: "ambient" | "point".  The reference positions point lights with a
    ``positioning_function(root_node, light_node)`` that needs a Panda3D scene graph;
    here a point light carries ``direction`` (unit vector in the object frame) and is
    placed at ``direction * bounds_radius * radius_factor`` exactly like
    ``make_scene_lights`` does."""

    light_type: str
    color: RgbaColor = (1.0, 1.0, 1.0, 1.0)
    positioning_function: Optional[Callable] = None
    direction: Optional[Tuple[float, float, float]] = None
    radius_factor: float = 10.0


def make_scene_lights(ambient_light_color: RgbaColor = (0.1, 0.1, 0.1, 1.0),
                      point_lights_color: RgbaColor = (0.4, 0.4, 0.4, 1.0)) -> List[Panda3dLightData]:
    """1 ambient + 6 point lights on the +/- axes at 10 bounding radii."""
    lights = [Panda3dLightData("ambient", ambient_light_color)]
    for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        lights.append(Panda3dLightData("point", point_lights_color, direction=d))
    return lights


class BatchRenderer:
    """``Panda3dBatchRenderer`` replacement.  ``n_workers`` / ``preload_cache`` /
    ``split_objects`` are accepted for signature compatibility and ignored."""

    def __init__(self, asset_dataset: RigidObjectDataset, n_workers: int = 8, preload_cache: bool = True,
                 split_objects: bool = False, device="cuda", store: Optional[ops.MeshStore] = None, msaa: bool = False,
                 aniso: bool = False):
        """``msaa``: render colour / normals with 4x multisampling, the framebuffer state of the reference's Panda3D
        renderer (``TB/renderer/panda3d_scene_renderer.py:70-71``; semantics in ``oracle/csrc/oracle.c`` ``HP_R_MSAA4``).
        Off by default: the sample pattern is implementation-defined in OpenGL and cannot be pinned here, and the
        single-sample renders are what every parity number of this repository was measured on.  ``aniso``: the texture
        state of the same file (``texture-minfilter mipmap``, ``texture-anisotropic-degree 16``): trilinear over the mip
        chain + up to 16 probes along the footprint's major axis (``HP_R_TEX_ANISO``); off by default for the same reason."""
        assert n_workers >= 1
        self.msaa = bool(msaa)
        self.aniso = bool(aniso)
        self._object_dataset = asset_dataset
        self.store = store if store is not None else ops.MeshStore(asset_dataset, device)
        self.device = self.store.device
        self._is_closed = False

    def _lights(self, labels, light_datas, n):
        if light_datas is None:
            return None, None, None
        assert len(light_datas) == n, "Need one light list per rendered view"
        amb = np.zeros((n, 3), np.float32)
        n_pts = max(sum(1 for l in ls if l.light_type == "point") for ls in light_datas) if n else 0
        pos = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
        col = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
        radius = self.store.packed.radius
        for i, ls in enumerate(light_datas):
            k = 0
            for l in ls:
                if l.light_type == "ambient":
                    amb[i] += np.asarray(l.color[:3], np.float32)
                elif l.light_type == "point":
                    if l.direction is None:
                        raise NotImplementedError("point lights need `direction` (no Panda3D scene graph here)")
                    r = radius[self.store.label_to_id[labels[i]]]
                    pos[i, k] = np.asarray(l.direction, np.float32) * r * l.radius_factor
                    col[i, k] = np.asarray(l.color[:3], np.float32)
                    k += 1
                else:
                    raise NotImplementedError(l.light_type)
        t = lambda a: None if a is None else torch.as_tensor(a, device=self.device)  # noqa: E731
        return t(amb), t(pos), t(col)

    def render(self, labels: Sequence[str], TCO: torch.Tensor, K: torch.Tensor,
               light_datas: Optional[List[List[Panda3dLightData]]] = None,
               resolution: Resolution = (240, 320), render_normals: bool = False,
               render_depth: bool = False, render_binary_mask: bool = False) -> BatchRenderOutput:
        bsz = TCO.shape[0]
        assert TCO.shape == (bsz, 4, 4)
        assert K.shape == (bsz, 3, 3)
        assert bsz == len(labels), "Need same number of labels as TCO/K batch size"
        amb, pos, col = self._lights(labels, light_datas, bsz)
        rgb, nrm, dep, msk = ops.rasterize(
            self.store, self.store.ids_of(labels), TCO.detach(), K, tuple(resolution),
            render_normals=render_normals, render_depth=render_depth,
            render_binary_mask=render_binary_mask, ambient=amb, light_pos=pos, light_col=col, msaa=self.msaa, aniso=self.aniso)
        return BatchRenderOutput(rgbs=rgb, normals=nrm, depths=dep, binary_masks=msk)

    def stop(self) -> None:
        self._is_closed = True


# the reference's class name, so `isinstance(renderer, Panda3dBatchRenderer)` style call
# sites (MP/models/pose_rigid.py:424, CP/models/pose.py:133) keep working after a rename
Panda3dBatchRenderer = BatchRenderer
