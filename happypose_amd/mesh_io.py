"""Mesh file ingestion (PLY ascii/binary, Wavefront OBJ) -> plain numpy arrays.

Replaces what the reference gets from trimesh (mesh-point database,
``TB/lib3d/rigid_mesh_database.py:52-82`` with ``process=False, maintain_order=True``)
and from Panda3D's assimp loader (renderer geometry,
``TB/renderer/panda3d_scene_renderer.py:206-219``).  Vertex order is preserved
(``maintain_order``) because the deterministic point sub-sampling indexes
vertices by position (``TB/lib3d/mesh_ops.py:74-84``).
"""

from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1",
    "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


@dataclass
class MeshData:
    """One object's geometry in mesh units (no scaling applied)."""

    vertices: np.ndarray  # [V,3] float64 (as trimesh holds them)
    faces: np.ndarray  # [F,3] int32
    normals: Optional[np.ndarray] = None  # [V,3] float32
    uvs: Optional[np.ndarray] = None  # [V,2] float32, v up (OpenGL convention)
    colors: Optional[np.ndarray] = None  # [V,4] uint8
    texture: Optional[np.ndarray] = None  # [Ht,Wt,4] uint8, row 0 = top
    texture_path: Optional[Path] = None


def compute_vertex_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Area-weighted vertex normals (used when the file carries none)."""
    v = np.asarray(vertices, dtype=np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-30), np.array([0.0, 0.0, 1.0]))
    return n.astype(np.float32)


def _load_texture(path: Path) -> Optional[np.ndarray]:
    if path is None or not Path(path).is_file():
        return None
    from PIL import Image  # optional dependency, only needed for textured assets

    im = Image.open(path).convert("RGBA")
    return np.asarray(im, dtype=np.uint8).copy()


def _triangulate(polys) -> np.ndarray:
    tris = []
    for p in polys:
        for k in range(1, len(p) - 1):
            tris.append((p[0], p[k], p[k + 1]))
    return np.asarray(tris, dtype=np.int32).reshape(-1, 3)


def load_ply(path) -> MeshData:
    path = Path(path)
    with open(path, "rb") as fh:
        raw = fh.read()
    end = raw.find(b"end_header")
    if not raw.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    header_end = raw.find(b"\n", end) + 1
    header = raw[:header_end].decode("ascii", errors="replace").splitlines()
    fmt = None
    elements = []  # (name, count, [(kind, name, types...)])
    texture_file = None
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "comment" and len(tok) >= 3 and tok[1].lower() == "texturefile":
            texture_file = tok[2]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1][2].append(("list", tok[4], tok[2], tok[3]))
            else:
                elements[-1][2].append(("scalar", tok[2], tok[1]))
    body = raw[header_end:]
    vert = {}
    polys = None
    if fmt == "ascii":
        tokens = body.split()
        pos = 0
        for name, count, props in elements:
            if name == "vertex":
                ncol = len(props)
                arr = np.array(tokens[pos:pos + count * ncol], dtype=np.float64).reshape(count, ncol)
                pos += count * ncol
                for k, p in enumerate(props):
                    vert[p[1]] = arr[:, k]
            elif name == "face":
                polys = []
                for _ in range(count):
                    row = None
                    for p in props:
                        if p[0] == "list":
                            n = int(tokens[pos]); pos += 1
                            vals = [int(float(t)) for t in tokens[pos:pos + n]] if p[1] in (
                                "vertex_indices", "vertex_index") else None
                            pos += n
                            if vals is not None:
                                row = vals
                        else:
                            pos += 1
                    polys.append(row)
            else:  # skip unknown element
                for _ in range(count):
                    for p in props:
                        if p[0] == "list":
                            n = int(tokens[pos]); pos += 1 + n
                        else:
                            pos += 1
    elif fmt in ("binary_little_endian", "binary_big_endian"):
        bo = "<" if fmt == "binary_little_endian" else ">"
        off = 0
        for name, count, props in elements:
            if all(p[0] == "scalar" for p in props):
                dt = np.dtype([(p[1], bo + _PLY_TYPES[p[2]]) for p in props])
                arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if name == "vertex":
                    for p in props:
                        vert[p[1]] = arr[p[1]].astype(np.float64)
            else:
                rows = []
                for _ in range(count):
                    row = None
                    for p in props:
                        if p[0] == "list":
                            cdt = np.dtype(bo + _PLY_TYPES[p[2]])
                            n = int(np.frombuffer(body, cdt, 1, off)[0]); off += cdt.itemsize
                            idt = np.dtype(bo + _PLY_TYPES[p[3]])
                            vals = np.frombuffer(body, idt, n, off); off += idt.itemsize * n
                            if p[1] in ("vertex_indices", "vertex_index"):
                                row = vals.astype(np.int64).tolist()
                        else:
                            off += np.dtype(bo + _PLY_TYPES[p[2]]).itemsize
                    rows.append(row)
                if name == "face":
                    polys = rows
    else:
        raise ValueError(f"{path}: unsupported PLY format {fmt}")

    vertices = np.stack([vert["x"], vert["y"], vert["z"]], axis=1)
    faces = _triangulate(polys) if polys else np.zeros((0, 3), np.int32)
    normals = None
    if "nx" in vert:
        normals = np.stack([vert["nx"], vert["ny"], vert["nz"]], axis=1).astype(np.float32)
    uvs = None
    for un, vn in (("texture_u", "texture_v"), ("s", "t"), ("u", "v")):
        if un in vert:
            uvs = np.stack([vert[un], vert[vn]], axis=1).astype(np.float32)
            break
    colors = None
    if "red" in vert:
        a = vert.get("alpha", np.full(len(vertices), 255.0))
        colors = np.stack([vert["red"], vert["green"], vert["blue"], a], axis=1).astype(np.uint8)
    tex_path = path.parent / texture_file if texture_file else None
    return MeshData(vertices=vertices, faces=faces, normals=normals, uvs=uvs, colors=colors,
                    texture=_load_texture(tex_path), texture_path=tex_path)


def load_obj(path) -> MeshData:
    """Wavefront OBJ with v / vt / vn / f; vertices are split per unique
    (v, vt, vn) triple, first-appearance order (triangulated fans)."""
    path = Path(path)
    vs, vts, vns, corners, polys = [], [], [], {}, []
    tex_path = None
    for line in open(path, "r", errors="replace"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            vs.append([float(t) for t in tok[1:4]])
        elif tok[0] == "vt":
            vts.append([float(t) for t in tok[1:3]])
        elif tok[0] == "vn":
            vns.append([float(t) for t in tok[1:4]])
        elif tok[0] == "f":
            poly = []
            for c in tok[1:]:
                parts = (c.split("/") + ["", ""])[:3]
                key = tuple(int(p) if p else 0 for p in parts)
                key = tuple(k + (len(a) + 1 if k < 0 else 0) for k, a in zip(key, (vs, vts, vns)))
                if key not in corners:
                    corners[key] = len(corners)
                poly.append(corners[key])
            polys.append(poly)
        elif tok[0] == "mtllib":
            mtl = path.parent / tok[1]
            if mtl.is_file():
                for ml in open(mtl, "r", errors="replace"):
                    mt = ml.split()
                    if mt and mt[0] == "map_Kd":
                        tex_path = path.parent / mt[-1]
    keys = sorted(corners, key=corners.get)
    vs_a = np.asarray(vs, dtype=np.float64).reshape(-1, 3)
    vertices = np.stack([vs_a[k[0] - 1] for k in keys]) if keys else np.zeros((0, 3))
    uvs = None
    if vts and all(k[1] > 0 for k in keys):
        vt_a = np.asarray(vts, dtype=np.float32)
        uvs = np.stack([vt_a[k[1] - 1] for k in keys])
    normals = None
    if vns and all(k[2] > 0 for k in keys):
        vn_a = np.asarray(vns, dtype=np.float32)
        normals = np.stack([vn_a[k[2] - 1] for k in keys])
    return MeshData(vertices=vertices, faces=_triangulate(polys), normals=normals, uvs=uvs,
                    texture=_load_texture(tex_path), texture_path=tex_path)


def load_mesh(path) -> MeshData:
    path = Path(path)
    ext = path.suffix.lower()
    if ext == ".ply":
        m = load_ply(path)
    elif ext == ".obj":
        m = load_obj(path)
    elif ext == ".npz":  # compact fixture format (tests/golden)
        z = np.load(path)
        m = MeshData(vertices=z["vertices"].astype(np.float64), faces=z["faces"].astype(np.int32),
                     normals=z["normals"] if "normals" in z else None,
                     uvs=z["uvs"] if "uvs" in z else None,
                     colors=z["colors"] if "colors" in z else None,
                     texture=z["texture"] if "texture" in z else None)
    else:
        raise ValueError(f"unsupported mesh format: {path}")
    if m.normals is None:
        m.normals = compute_vertex_normals(m.vertices, m.faces)
    return m
