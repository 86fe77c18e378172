"""Object/mesh containers of the path.

Mirrors (API names, argument meaning) the reference's
``RigidObject`` / ``RigidObjectDataset`` (``TB/datasets/object_dataset.py:32-174``),
``MeshDataBase`` / ``BatchedMeshes`` / ``Meshes``
(``TB/lib3d/rigid_mesh_database.py:52-200``) and adds ``PackedMeshes``: the flat,
device-friendly layout the HIP rasteriser and the pose-prep kernel read
(all objects concatenated; one ``int64[8]`` descriptor row per object).
"""

from __future__ import annotations

from copy import deepcopy
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from .mesh_io import MeshData, load_mesh


class RigidObject:
    """``TB/datasets/object_dataset.py:32-143``.  ``mesh_path`` may also be an
    in-memory :class:`MeshData` (synthetic scenes)."""

    def __init__(
        self,
        label: str,
        mesh_path,
        category: Optional[str] = None,
        mesh_diameter: Optional[float] = None,
        mesh_units: str = "m",
        symmetries_discrete: Sequence = (),
        symmetries_continuous: Sequence = (),
        ypr_offset_deg: Tuple[float, float, float] = (0.0, 0.0, 0.0),
        scaling_factor: float = 1.0,
        scaling_factor_mesh_units_to_meters: Optional[float] = None,
    ):
        self.label = label
        self.category = category
        self.mesh_path = mesh_path
        self.mesh_units = mesh_units
        if scaling_factor_mesh_units_to_meters is not None:
            self.scaling_factor_mesh_units_to_meters = scaling_factor_mesh_units_to_meters
        else:
            self.scaling_factor_mesh_units_to_meters = {"m": 1.0, "mm": 0.001}[mesh_units]
        self.scaling_factor = scaling_factor
        self.mesh_diameter = mesh_diameter
        self.diameter_meters = None
        self.symmetries_discrete = list(symmetries_discrete)
        self.symmetries_continuous = list(symmetries_continuous)
        self.ypr_offset_deg = ypr_offset_deg
        self._mesh: Optional[MeshData] = mesh_path if isinstance(mesh_path, MeshData) else None

    @property
    def is_symmetric(self) -> bool:
        return len(self.symmetries_discrete) > 0 or len(self.symmetries_continuous) > 0

    @property
    def scale(self) -> float:
        return self.scaling_factor_mesh_units_to_meters * self.scaling_factor

    def load(self) -> MeshData:
        if self._mesh is None:
            self._mesh = load_mesh(self.mesh_path)
        return self._mesh


class RigidObjectDataset:
    """``TB/datasets/object_dataset.py:146-174``."""

    def __init__(self, objects: List[RigidObject]):
        self.list_objects = objects
        self.label_to_objects = {obj.label: obj for obj in objects}
        if len(self.list_objects) != len(self.label_to_objects):
            raise RuntimeError("There are objects with duplicate labels")

    def __getitem__(self, idx: int) -> RigidObject:
        return self.list_objects[idx]

    def get_object_by_label(self, label: str) -> RigidObject:
        return self.label_to_objects[label]

    def __len__(self) -> int:
        return len(self.list_objects)

    @property
    def objects(self) -> List[RigidObject]:
        return self.list_objects

    def filter_objects(self, keep_labels: Set[str]) -> "RigidObjectDataset":
        return RigidObjectDataset([o for o in self.list_objects if o.label in keep_labels])


def pad_stack_points(points_list: List[np.ndarray]) -> np.ndarray:
    """``pad_stack_tensors(fill="select_random", deterministic=True)``
    (``TB/lib3d/rigid_mesh_database.py:172-200``): shorter point sets are padded by
    re-sampling their own vertices from ONE ``RandomState(0)`` stream shared by all
    objects in order."""
    n_max = max(len(p) for p in points_list)
    rs = np.random.RandomState(0)
    out = []
    for p in points_list:
        n_pad = n_max - len(p)
        if n_pad > 0:
            ids_pad = rs.choice(np.arange(len(p)), size=n_pad)
            p = np.concatenate((p, p[ids_pad]), axis=0)
        out.append(p)
    return np.stack(out)


def sample_point_ids(n_pad: int, n_points: int) -> np.ndarray:
    """Deterministic ids of ``sample_points(..., deterministic=True)``
    (``TB/lib3d/mesh_ops.py:74-84``)."""
    assert n_points <= n_pad
    return np.random.RandomState(0).choice(n_pad, size=n_points, replace=False)


class Meshes:
    """``TB/lib3d/rigid_mesh_database.py:151-169`` (tensor-collection of selected
    objects).  ``points`` is ``[b, N_pad, 3]``."""

    def __init__(self, infos, labels, points):
        self.infos = infos
        self.labels = np.asarray(labels)
        self.points = points

    def sample_points(self, n_points: int, deterministic: bool = False):
        n_pad = self.points.shape[1]
        assert n_points <= n_pad
        rs = np.random.RandomState(0) if deterministic else np.random
        ids = rs.choice(n_pad, size=n_points, replace=False)
        return self.points[:, ids]


class BatchedMeshes:
    """``TB/lib3d/rigid_mesh_database.py:133-149``: ``points [n_obj, N_pad, 3]``
    float32, metres.  ``points`` is a numpy array on the host and, after
    :meth:`to`, a torch tensor on the compute device."""

    def __init__(self, infos, labels, points):
        self.infos = infos
        self.label_to_id: Dict[str, int] = {label: n for n, label in enumerate(labels)}
        self.labels = np.asarray(labels)
        self.points = points

    def select(self, labels) -> Meshes:
        ids = [self.label_to_id[label] for label in labels]
        return Meshes([self.infos[label] for label in labels], self.labels[ids], self.points[ids])

    def ids_of(self, labels) -> np.ndarray:
        return np.asarray([self.label_to_id[label] for label in labels], dtype=np.int32)

    def to(self, device):
        import torch

        self.points = torch.as_tensor(self.points).to(device)
        return self

    def float(self):
        return self

    def cuda(self):
        return self.to("cuda")


class MeshDataBase:
    """``TB/lib3d/rigid_mesh_database.py:52-130``."""

    def __init__(self, obj_list: List[RigidObject]):
        self.obj_dict = {obj.label: obj for obj in obj_list}
        self.obj_list = obj_list
        self.infos = {obj.label: {} for obj in obj_list}
        self.meshes = {label: obj.load() for label, obj in self.obj_dict.items()}
        for label, obj in self.obj_dict.items():
            if obj.diameter_meters is None:
                points = np.asarray(self.meshes[label].vertices) * obj.scale
                extent = points.max(0) - points.min(0)
                obj.diameter_meters = float(np.linalg.norm(extent))

    @staticmethod
    def from_object_ds(object_ds: RigidObjectDataset) -> "MeshDataBase":
        return MeshDataBase([object_ds[n] for n in range(len(object_ds))])

    def batched(self) -> BatchedMeshes:
        labels, points = [], []
        new_infos = deepcopy(self.infos)
        for label, mesh in self.meshes.items():
            pts = np.asarray(mesh.vertices, dtype=np.float64) * self.obj_dict[label].scale
            new_infos[label]["n_points"] = pts.shape[0]
            points.append(pts)
            labels.append(label)
        pts = pad_stack_points(points).astype(np.float32)
        return BatchedMeshes(new_infos, np.array(labels), pts)


def mip_chain(tex: np.ndarray):
    """Levels 1.. of an RGBA8 texture ``[h,w,4]`` as ``glGenerateMipmap`` builds them: each level halves the previous one
    (``max(1, floor(n / 2))`` per side) with a 2 x 2 box filter, rounded to nearest in 8 bit; a side that is already 1
    averages 2 x 1.  Returned without level 0."""
    levels, cur = [], np.ascontiguousarray(tex, np.uint8)
    while cur.shape[0] > 1 or cur.shape[1] > 1:
        h, w = cur.shape[0], cur.shape[1]
        nh, nw = max(1, h // 2), max(1, w // 2)
        c = cur.astype(np.uint16)
        r0 = c[0:2 * nh:2] if h > 1 else c
        r1 = c[1:2 * nh:2] if h > 1 else c
        a = r0[:, 0:2 * nw:2] if w > 1 else r0
        b = r0[:, 1:2 * nw:2] if w > 1 else r0
        cc = r1[:, 0:2 * nw:2] if w > 1 else r1
        d = r1[:, 1:2 * nw:2] if w > 1 else r1
        cur = ((a + b + cc + d + 2) >> 2).astype(np.uint8)
        levels.append(cur)
    return levels


def hpr_to_matrix(ypr_deg) -> np.ndarray:
    """Rotation of Panda3D's ``NodePath.setHpr(h, p, r)`` (degrees; Z-up, Y-forward frame): heading about +Z, pitch
    about +X, roll about +Y, applied roll first, then pitch, then heading -- ``R = Rz(h) @ Rx(p) @ Ry(r)`` on column
    vectors.  **[unverified]**: Panda3D's documented convention; Panda3D is absent here."""
    h, p, r = (np.deg2rad(float(a)) for a in ypr_deg)
    ch, sh, cp, sp, cr, sr = np.cos(h), np.sin(h), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    Rz = np.array([[ch, -sh, 0.0], [sh, ch, 0.0], [0.0, 0.0, 1.0]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]])
    Ry = np.array([[cr, 0.0, sr], [0.0, 1.0, 0.0], [-sr, 0.0, cr]])
    return Rz @ Rx @ Ry


class PackedMeshes:
    """Flat layout consumed by the rasteriser (HIP and oracle).

    ``verts [Vtot,3] f32`` (metres), ``normals [Vtot,3] f32``, ``uvs [Vtot,2] f32``,
    ``colors [Vtot,4] u8``, ``faces [Ftot,3] i32`` (ids local to the object),
    ``tex`` RGBA8 pool, ``obj [n_obj,8] i64`` = (vert_off, n_verts, face_off, n_faces,
    tex_off | -1, tex_w, tex_h, n_mip_levels), ``radius [n_obj] f32`` = largest vertex norm (m), ``bounds_center`` / ``bounds_radius`` = the bounding sphere light-positioning functions see.  A texture is stored as its
    level 0 followed by its mip chain (:func:`mip_chain`; level ``k`` is ``max(1, w >> k) x max(1, h >> k)``): the
    bilinear level-0 fetch ignores the chain, ``HP_RASTER_TEX_ANISO`` (the reference's ``texture-minfilter mipmap`` +
    ``texture-anisotropic-degree 16``) walks it."""

    def __init__(self, object_ds: RigidObjectDataset):
        self.labels = [o.label for o in object_ds.list_objects]
        self.label_to_id = {l: i for i, l in enumerate(self.labels)}
        verts, normals, uvs, colors, faces, tex, rows, radius = [], [], [], [], [], [], [], []
        centers, bradius = [], []
        voff = foff = toff = 0
        for obj in object_ds.list_objects:
            m = obj.load()
            v64 = np.asarray(m.vertices, dtype=np.float64) * obj.scale
            n64 = np.asarray(m.normals, np.float64)
            if tuple(float(a) for a in obj.ypr_offset_deg) != (0.0, 0.0, 0.0):
                # the RENDER mesh only is rotated (get_object_node: setScale, setHpr --
                # TB/renderer/panda3d_scene_renderer.py:206-219); the point table of MeshDataBase is not
                # (TB/lib3d/rigid_mesh_database.py:109-111)
                R = hpr_to_matrix(obj.ypr_offset_deg)
                v64, n64 = v64 @ R.T, n64 @ R.T
            v = v64.astype(np.float32)
            nv, nf = len(v), len(m.faces)
            verts.append(v)
            normals.append(n64.astype(np.float32))
            uvs.append(np.zeros((nv, 2), np.float32) if m.uvs is None else np.asarray(m.uvs, np.float32))
            if m.colors is not None:
                colors.append(np.asarray(m.colors, np.uint8).reshape(nv, 4))
            else:
                colors.append(np.full((nv, 4), 255, np.uint8))
            faces.append(np.asarray(m.faces, np.int32))
            if m.texture is not None and m.uvs is not None:
                t = np.ascontiguousarray(m.texture, np.uint8)
                chain = mip_chain(t)
                rows.append((voff, nv, foff, nf, toff, t.shape[1], t.shape[0], 1 + len(chain)))
                for lvl in [t] + chain:
                    tex.append(lvl.reshape(-1))
                    toff += lvl.size
            else:
                rows.append((voff, nv, foff, nf, -1, 0, 0, 0))
            radius.append(float(np.linalg.norm(v, axis=1).max()) if nv else 0.0)
            # bounding sphere as Panda3D builds it for a Geom (centre of the vertices' bounding box, radius = the farthest
            # vertex from it): what ``root_node.getBounds()`` answers to a light-positioning function  [unverified]
            c = (v.min(0).astype(np.float64) + v.max(0).astype(np.float64)) / 2 if nv else np.zeros(3)
            centers.append(c)
            bradius.append(float(np.linalg.norm(v.astype(np.float64) - c, axis=1).max()) if nv else 0.0)
            voff += nv
            foff += nf
        self.verts = np.concatenate(verts)
        self.normals = np.concatenate(normals)
        self.uvs = np.concatenate(uvs)
        self.colors = np.concatenate(colors)
        self.faces = np.concatenate(faces)
        self.tex = np.concatenate(tex) if tex else np.zeros(4, np.uint8)
        self.obj = np.asarray(rows, dtype=np.int64)
        self.radius = np.asarray(radius, dtype=np.float32)
        self.bounds_center = np.asarray(centers, dtype=np.float64).reshape(-1, 3)
        self.bounds_radius = np.asarray(bradius, dtype=np.float32)

    def ids_of(self, labels) -> np.ndarray:
        return np.asarray([self.label_to_id[l] for l in labels], dtype=np.int32)
