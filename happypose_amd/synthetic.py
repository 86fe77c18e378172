"""Seeded synthetic inputs of SURVEY.md section 8(d): procedural textured meshes at
YCB-V-like complexity, a 640x480 frame, detections, noisy pose hypotheses and
name-keyed random network weights.  There are no datasets or checkpoints in
the build/bench environment, so every test and ``bench.py`` draws from here.
"""

from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

from .mesh_io import MeshData
from .mesh_store import RigidObject, RigidObjectDataset


def make_mesh(seed: int, n_lat: int = 72, n_lon: int = 112, diameter: float = 0.15,
              tex_size: int = 1024) -> MeshData:
    """Closed, bumpy superellipsoid with UVs, vertex normals and an RGBA8 texture.
    Default grid: 73 x 113 = 8249 vertices, 2*72*112 = 16128 faces (the reference's
    test asset ``tests/data/obj_000001.ply`` has 9951 / 15728)."""
    rs = np.random.RandomState(seed)
    th = np.linspace(0.0, np.pi, n_lat + 1)
    ph = np.linspace(0.0, 2 * np.pi, n_lon + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    ex = rs.uniform(0.6, 1.4, 2)
    axes = rs.uniform(0.55, 1.0, 3)

    def spow(x, e):
        return np.sign(x) * np.abs(x) ** e

    r = 1.0
    for k in range(3):  # low-frequency bumps, periodic in phi
        a, m, q = rs.uniform(0.03, 0.09), rs.randint(1, 5), rs.uniform(0, 2 * np.pi)
        r = r + a * np.sin(m * PH + q) * np.sin((k + 1) * TH) ** 2
    x = axes[0] * r * spow(np.sin(TH), ex[0]) * spow(np.cos(PH), ex[1])
    y = axes[1] * r * spow(np.sin(TH), ex[0]) * spow(np.sin(PH), ex[1])
    z = axes[2] * r * spow(np.cos(TH), ex[0])
    v = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    v -= 0.5 * (v.max(0) + v.min(0))
    v *= diameter / np.linalg.norm(v.max(0) - v.min(0))
    uv = np.stack([PH / (2 * np.pi), 1.0 - TH / np.pi], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = np.arange((n_lat + 1) * (n_lon + 1)).reshape(n_lat + 1, n_lon + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    faces = np.concatenate(
        [np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)]
    ).astype(np.int32)
    from .mesh_io import compute_vertex_normals

    normals = compute_vertex_normals(v, faces)
    # smooth-ish random texture: coarse colour blocks + fine noise (keeps bilinear
    # filtering meaningful and the render far from constant)
    blocks = rs.randint(0, 256, size=(16, 16, 3)).astype(np.float32)
    tex = np.kron(blocks, np.ones((tex_size // 16, tex_size // 16, 1), np.float32))
    tex = 0.75 * tex + 0.25 * rs.randint(0, 256, size=(tex_size, tex_size, 3))
    tex = np.concatenate([tex, np.full((tex_size, tex_size, 1), 255.0)], -1)
    return MeshData(vertices=v.astype(np.float64), faces=faces, normals=normals, uvs=uv,
                    texture=np.clip(tex, 0, 255).astype(np.uint8))


def make_object_dataset(n_objects: int = 8, seed: int = 1, tex_size: int = 1024,
                        n_lat: int = 72, n_lon: int = 112) -> RigidObjectDataset:
    rs = np.random.RandomState(seed)
    objs = []
    for i in range(n_objects):
        mesh = make_mesh(seed * 1000 + i, n_lat=n_lat, n_lon=n_lon,
                         diameter=float(rs.uniform(0.10, 0.25)), tex_size=tex_size)
        objs.append(RigidObject(label=f"obj_{i + 1:06d}", mesh_path=mesh, mesh_units="m"))
    return RigidObjectDataset(objs)


def euler_to_R(e: np.ndarray) -> np.ndarray:
    """Static-frame xyz Euler angles -> rotation matrices (Rz @ Ry @ Rx), matching
    ``transforms3d.euler.euler2mat(ai, aj, ak)`` default axes 'sxyz' used by the
    reference's pose noise (``TB/lib3d/transform_ops.py:70-104``)."""
    e = np.asarray(e, dtype=np.float64).reshape(-1, 3)
    cx, sx = np.cos(e[:, 0]), np.sin(e[:, 0])
    cy, sy = np.cos(e[:, 1]), np.sin(e[:, 1])
    cz, sz = np.cos(e[:, 2]), np.sin(e[:, 2])
    R = np.empty((len(e), 3, 3))
    R[:, 0, 0] = cy * cz; R[:, 0, 1] = sx * sy * cz - cx * sz; R[:, 0, 2] = cx * sy * cz + sx * sz
    R[:, 1, 0] = cy * sz; R[:, 1, 1] = sx * sy * sz + cx * cz; R[:, 1, 2] = cx * sy * sz - sx * cz
    R[:, 2, 0] = -sy; R[:, 2, 1] = sx * cy; R[:, 2, 2] = cx * cy
    return R


def random_rotations(rs: np.random.RandomState, n: int) -> np.ndarray:
    q = rs.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    return np.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)


def make_scene(n_detections: int = 8, n_hypotheses: int = 16, n_objects: int = 8, seed: int = 2,
               with_depth: bool = False, H: int = 480, W: int = 640,
               f: float = 600.0) -> Dict[str, np.ndarray]:
    """One frame: image ``[1,3|4,H,W]`` f32 in [0,1] (+ depth m), ``K [1,3,3]``, per
    detection a ground pose in view (z in [0.4,0.9] m), and ``n_hypotheses`` noisy
    copies of it (euler sigma 15 deg, t sigma (1,1,5) cm: the reference's training
    noise, ``TB/lib3d/transform_ops.py:70-104``)."""
    rs = np.random.RandomState(seed)
    img = np.random.RandomState(0).rand(1, 3, H, W).astype(np.float32)
    if with_depth:
        d = (0.3 + 0.4 * np.random.RandomState(3).rand(1, 1, H, W)).astype(np.float32)
        img = np.concatenate([img, d], 1)
    K = np.array([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]], dtype=np.float32)
    R = random_rotations(rs, n_detections)
    z = rs.uniform(0.4, 0.9, n_detections)
    u = rs.uniform(0.25 * W, 0.75 * W, n_detections)
    v = rs.uniform(0.25 * H, 0.75 * H, n_detections)
    t = np.stack([(u - W / 2) * z / f, (v - H / 2) * z / f, z], -1)
    T_det = np.tile(np.eye(4), (n_detections, 1, 1))
    T_det[:, :3, :3] = R
    T_det[:, :3, 3] = t
    obj_ids = (np.arange(n_detections) % n_objects).astype(np.int32)
    B = n_detections * n_hypotheses
    eul = rs.normal(0, 15.0, size=(B, 3)) * np.pi / 180
    tn = rs.normal(0, 1.0, size=(B, 3)) * np.array([0.01, 0.01, 0.05])
    T_hyp = np.repeat(T_det, n_hypotheses, axis=0)
    T_hyp[:, :3, :3] = T_hyp[:, :3, :3] @ euler_to_R(eul)
    T_hyp[:, :3, 3] += tn
    return dict(
        images=img, K=K, TCO_det=T_det.astype(np.float32), TCO_hyp=T_hyp.astype(np.float32),
        det_obj_ids=obj_ids, hyp_obj_ids=np.repeat(obj_ids, n_hypotheses).astype(np.int32),
        hyp_det_ids=np.repeat(np.arange(n_detections), n_hypotheses).astype(np.int32),
    )


def named_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0,
                  head_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Name-keyed random parameters: every tensor is drawn from its own
    ``RandomState(crc32(name) ^ seed)`` so the ~86 MB of weights can be REGENERATED
    anywhere instead of being committed (golden G6).  Conv/linear weights are
    He-scaled, BN gamma in [0.5,1.5], beta ~ 0.1 N, running_mean ~ 0.1 N,
    running_var in [0.5,1.5] -- activations stay O(1) through 34 layers."""
    out = {}
    for name, shape in shapes.items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            out[name] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * rs.normal(size=shape)).astype(np.float32)
        elif len(shape) == 1 and name.endswith("weight"):  # BN gamma
            out[name] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1:  # biases / BN beta
            out[name] = (0.1 * rs.normal(size=shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            w = rs.normal(size=shape) * np.sqrt(2.0 / fan_in)
            if len(shape) == 2 and ("pose_fc" in name or "views_logits_head" in name):
                w = w * head_scale
            if name.endswith("conv2.weight"):
                w = w * 0.25  # keeps the residual stream O(1) through 16 blocks
            if name.endswith("_project_conv.weight"):
                # EfficientNet-b3: blocks with an identity skip keep the stream O(1) with a small branch (0.3); the first
                # block of a stage has no skip and must carry the signal itself (1.25 balances the 0.5 SE gate and the
                # swish gain: with 0.3 everywhere the output was input-independent to 6 digits and a golden on it blind
                # to everything but the last blocks' biases)
                stem = name[: -len("_project_conv.weight")]
                c_in = shapes[stem + "_expand_conv.weight"][1] if stem + "_expand_conv.weight" in shapes else shape[1]
                w = w * (0.3 if c_in == shape[0] else 1.25)
            out[name] = w.astype(np.float32)
    return out


def predictor_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0,
                      update_scale: float = 0.002) -> Dict[str, np.ndarray]:
    """Weights for a whole ``PosePredictor`` (keys ``backbone.*``, ``pose_fc.*``,
    ``views_logits_head.*``).  The pose head is biased to the identity update
    ``(1,0,0, 0,1,0, 0,0,1)`` with small random weights so that refined poses stay in
    view over 5 iterations ("pose9 ~ identity + 0.01 noise")."""
    w = named_weights(shapes, seed=seed, head_scale=update_scale)
    if "pose_fc.bias" in w:
        ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
        w["pose_fc.bias"] = ident[: len(w["pose_fc.bias"])].copy()
    return w
