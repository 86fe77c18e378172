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
    """``TB/renderer/types.py:139-150``.  ``light_type``: "ambient" | "point".  A point light is placed by its
    ``positioning_function(root_node, light_node)`` exactly as the reference's ``setup_lights`` calls it
    (``TB/renderer/panda3d_scene_renderer.py:294-318``); there is no Panda3D scene graph here, so the two arguments are
    duck-typed stand-ins (:class:`SceneRootProxy`: ``getBounds().radius`` = bounding-sphere radius of the object;
    :class:`LightNodeProxy`: records ``setPos``).  ``direction`` / ``radius_factor`` are this repository's shorthand for
    the same placement (``direction * bounds_radius * radius_factor``) and are used when no function is given."""

    light_type: str
    color: RgbaColor = (1.0, 1.0, 1.0, 1.0)
    positioning_function: Optional[Callable] = None
    direction: Optional[Tuple[float, float, float]] = None
    radius_factor: float = 10.0


class _Bounds:
    """What positioning functions read from ``root_node.getBounds()``: a bounding sphere."""

    def __init__(self, center, radius: float):
        self.center, self.radius = tuple(float(c) for c in center), float(radius)

    def getRadius(self) -> float:
        return self.radius

    get_radius = getRadius

    def getCenter(self):
        return self.center

    get_center = getCenter


class SceneRootProxy:
    """Stand-in for the Panda3D root ``NodePath`` of a one-object scene (``render_scene``: the object sits at the world
    origin, ``TB/renderer/panda3d_batch_renderer.py:172``)."""

    def __init__(self, center, radius: float):
        self._bounds = _Bounds(center, radius)

    def getBounds(self) -> _Bounds:
        return self._bounds

    get_bounds = getBounds

    def getTightBounds(self):
        c, r = np.asarray(self._bounds.center), self._bounds.radius
        return tuple(c - r), tuple(c + r)

    get_tight_bounds = getTightBounds


class LightNodeProxy:
    """Stand-in for the light's ``NodePath``: records the position a positioning function sets (world = object frame)."""

    def __init__(self):
        self.pos = (0.0, 0.0, 0.0)

    def setPos(self, *args) -> None:
        if len(args) == 1:  # a tuple / Vec3 / LPoint3-like
            v = args[0]
            args = tuple(v) if not hasattr(v, "x") else (v.x, v.y, v.z)
        if len(args) == 4:  # setPos(other_node, x, y, z): relative to the root, which sits at the origin
            args = args[1:]
        assert len(args) == 3, "setPos(x, y, z) or setPos((x, y, z))"
        self.pos = tuple(float(a) for a in args)

    set_pos = setPos

    def getPos(self):
        return self.pos

    get_pos = getPos

    def __getattr__(self, name):  # anything else a positioning function may try (lookAt, setHpr ...) needs a scene graph
        raise NotImplementedError(f"light positioning functions may call setPos / getBounds only (got NodePath.{name})")


def _axis_light_position(root_node, light_node, pos: np.ndarray) -> None:
    """``pos_fn`` of the reference's ``make_scene_lights`` (``TB/renderer/panda3d_scene_renderer.py:121-129``)."""
    radius = root_node.getBounds().radius
    light_node.setPos(tuple((np.asarray(pos) * radius * 10).tolist()))


def make_scene_lights(ambient_light_color: RgbaColor = (0.1, 0.1, 0.1, 1.0),
                      point_lights_color: RgbaColor = (0.4, 0.4, 0.4, 1.0)) -> List[Panda3dLightData]:
    """1 ambient + 6 point lights on the +/- axes at 10 bounding radii, each carrying a ``positioning_function`` like
    the reference's (``TB/renderer/panda3d_scene_renderer.py:105-141``)."""
    from functools import partial

    lights = [Panda3dLightData("ambient", ambient_light_color)]
    for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        lights.append(Panda3dLightData("point", point_lights_color, positioning_function=partial(_axis_light_position, pos=np.array(d))))
    return lights


class BatchRenderer:
    """``Panda3dBatchRenderer`` replacement.  ``n_workers`` / ``preload_cache`` /
    ``split_objects`` are accepted for signature compatibility and ignored."""

    def __init__(self, asset_dataset: RigidObjectDataset, n_workers: int = 8, preload_cache: bool = True,
                 split_objects: bool = False, device="cuda", store: Optional[ops.MeshStore] = None, msaa: bool = True,
                 aniso: bool = True, backface_culling: Optional[bool] = None):
        """The render state.  Default = the reference renderer's (``TB/renderer/panda3d_scene_renderer.py:68-71``):
        ``msaa``: colour / normals with 4x multisampling (``framebuffer-multisample 1``, ``multisamples 4``; semantics in
        ``oracle/csrc/oracle.c`` ``HP_R_MSAA4``); ``aniso``: ``texture-minfilter mipmap`` + ``texture-anisotropic-degree 16``
        -- trilinear over the mip chain + up to 16 probes along the footprint's major axis (``HP_R_TEX_ANISO``).  The released
        checkpoints were trained on such renders, and the measured pose difference between this state and the cheaper
        single-sample / bilinear one (``msaa=False, aniso=False``) is 10-60x the stated pose tolerance per iteration
        (``tools/parity_sharpness.py``), so the cheap state is an explicit opt-out.  Neither half can be
        pinned against Panda3D here: sample positions and the anisotropic footprint are implementation-defined in OpenGL."""
        assert n_workers >= 1
        self.msaa = bool(msaa)
        self.aniso = bool(aniso)
        self._object_dataset = asset_dataset
        # backface_culling=False: two-sided renders everywhere (object sets with self-intersecting closed meshes, see ops.MeshStore)
        self.store = store if store is not None else ops.MeshStore(asset_dataset, device, backface_culling=backface_culling)
        self.device = self.store.device
        self._is_closed = False

    def scene_light_tables(self):
        """``make_scene_lights()`` evaluated once per object of the store: ``(ambient [3], pos [n_obj, 6, 3], col [6, 3])``
        on the device.  The predictors index ``pos`` with the object ids of a batch (MegaPose without the normals
        channel lights every view with these, ``MP/models/pose_rigid.py:422``)."""
        if getattr(self, "_scene_lights", None) is None:
            n_obj = len(self.store.labels)
            amb, pos, col = self._lights(list(self.store.labels), [make_scene_lights() for _ in range(n_obj)], n_obj)
            self._scene_lights = (amb[0].contiguous(), pos.contiguous(), col[0].contiguous())
        return self._scene_lights

    def clone_for_lane(self) -> "BatchRenderer":
        """A renderer with the SAME render state on a mesh store of its own (the second lane of a
        ``TwoLanePredictor`` writes its own rasteriser scratch).  The clone's store follows this one's conventions record and
        culling switch, now and after later ``store.set_raster_conventions`` / ``set_backface_culling`` calls
        (``ops.MeshStore.clone_for_lane``)."""
        return BatchRenderer(self._object_dataset, device=self.device, store=self.store.clone_for_lane(), msaa=self.msaa, aniso=self.aniso)

    def _lights(self, labels, light_datas, n):
        if light_datas is None:
            return None, None, None
        assert len(light_datas) == n, "Need one light list per rendered view"
        amb = np.zeros((n, 3), np.float32)
        n_pts = max(sum(1 for l in ls if l.light_type == "point") for ls in light_datas) if n else 0
        pos = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
        col = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
        packed = self.store.packed
        for i, ls in enumerate(light_datas):
            k = 0
            oid = self.store.label_to_id[labels[i]]
            for l in ls:
                if l.light_type == "ambient":
                    amb[i] += np.asarray(l.color[:3], np.float32)
                elif l.light_type == "point":
                    if l.positioning_function is not None:  # the reference's way (setup_lights): call it on the scene
                        node = LightNodeProxy()
                        l.positioning_function(SceneRootProxy(packed.bounds_center[oid], packed.bounds_radius[oid]), node)
                        pos[i, k] = np.asarray(node.pos, np.float32)
                    elif l.direction is not None:
                        pos[i, k] = np.asarray(l.direction, np.float32) * packed.bounds_radius[oid] * l.radius_factor
                    else:
                        raise AssertionError("a point light needs a positioning_function")  # setup_lights asserts it (:303)
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
