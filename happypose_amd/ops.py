"""Thin torch-tensor front-ends of the C ABI (one function per entry point).

torch is used for device memory and streams only; every computation below
happens in the HIP library.  Each wrapper validates what the reference asserts
(shapes, dtypes) and enqueues on the current torch stream.
"""

from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from ._ffi import Strides, check, lib, ptr, stream_ptr
from .mesh_store import MeshDataBase, PackedMeshes, RigidObjectDataset, sample_point_ids

DEPTH_NORM = {None: 0, "none": 0, "tCR_scale": 1, "tCR_scale_clamp_center": 2, "tCR_center_clamp": 3}
MULTIVIEW = {"TCO": (0, 1), "1view_TCO": (0, 1), "TCO+front_1view": (1, 2),
             "TCO+front_3views": (3, 4), "TCO+front_5views": (5, 6)}
ARCH = {"vanilla_resnet34": 0, "resnet34": 1, "resnet18": 2, "efficientnet-b3": 3, "resnet50-fpn": 4}
N_FEATURES = {"vanilla_resnet34": 512, "resnet34": 512, "resnet18": 512, "efficientnet-b3": 1536, "resnet50-fpn": 256}


_GRAPH_EPOCH = 0


def graph_epoch() -> int:
    """Bumped whenever something changes WHICH kernels a forward launches (conv algorithm, profiling events, the
    non-finite guard switching a network to its exact kernels): captured hipGraphs of an older epoch are stale
    (``happypose_amd.graphs``)."""
    return _GRAPH_EPOCH


def scratch_launches() -> int:
    """``hp_scratch_launches``: launches so far of kernels that use scratch (spilled tile variants).  A captured hipGraph with
    such a launch replays wrongly on this runtime: a predictor whose eager call moves this count stays on eager launches."""
    return int(lib().hp_scratch_launches())


def bump_graph_epoch() -> None:
    global _GRAPH_EPOCH
    _GRAPH_EPOCH += 1


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def _i32(t, device) -> torch.Tensor:
    return torch.as_tensor(t).to(device=device, dtype=torch.int32).contiguous()


def _check_ids(ids, n: int, what: str) -> None:
    """Index tensors against the table they index.  Ids that still live on the host (numpy arrays, CPU tensors:
    what the estimators build from ``infos``) are checked here for free and raise like the reference's indexing;
    ids already on the device cannot be read without a synchronisation -- the kernels receive the table sizes and
    answer an out-of-range id with NaN poses / zero crops instead of touching memory outside the table."""
    if ids is None:
        return
    t = torch.as_tensor(ids)
    if t.device.type == "cpu" and t.numel():
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= n:
            raise IndexError(f"{what}: ids in [{lo}, {hi}] index a table of {n} rows")


class MeshStore:
    """Device-resident object set: geometry/textures for the rasteriser and the padded
    mesh-point table for the projection kernels (``hp_mesh_store``)."""

    def __init__(self, object_ds: RigidObjectDataset, device="cuda", backface_culling: Optional[bool] = None):
        """``backface_culling``: ``False`` renders this object set two-sided everywhere, as the reference does -- the opt-out for
        sets that may hold self-intersecting closed meshes (the per-component culling decision assumes they do not:
        ``hp_mesh_store_set_backface_culling`` in the header); ``None`` = the library default (on)."""
        self.device = torch.device(device)
        self.object_ds = object_ds
        self.packed = PackedMeshes(object_ds)
        self.mesh_db = MeshDataBase.from_object_ds(object_ds).batched()
        self.labels = list(self.packed.labels)
        self.label_to_id = dict(self.packed.label_to_id)
        pts = np.ascontiguousarray(self.mesh_db.points, dtype=np.float32)
        self.n_pad = pts.shape[1]
        p = self.packed
        with torch.cuda.device(self.device):
            self._h = lib().hp_mesh_store_create(
                _np_ptr(p.verts), _np_ptr(p.normals), _np_ptr(p.uvs), _np_ptr(p.colors), len(p.verts),
                _np_ptr(p.faces), len(p.faces), _np_ptr(p.tex), p.tex.size, _np_ptr(p.obj), len(p.obj),
                _np_ptr(pts), self.n_pad)
        if not self._h:
            raise _ffi.HipLibraryError("hp_mesh_store_create: " + lib().hp_last_error().decode())
        self._point_ids: Dict[int, torch.Tensor] = {}
        self._ids_cache: Dict[tuple, torch.Tensor] = {}
        self.radius = torch.as_tensor(p.radius, device=self.device)
        self._followers: List["weakref.ReferenceType[MeshStore]"] = []  # the lane stores cloned from this one (clone_for_lane)
        if backface_culling is not None:
            self.set_backface_culling(bool(backface_culling))

    def clone_for_lane(self) -> "MeshStore":
        """A second store on the same object set (a lane's own rasteriser scratch) that FOLLOWS this one's render state: the
        conventions record and the culling switch are copied now, and every later ``set_raster_conventions`` /
        ``set_backface_culling`` on this store reaches the clone too -- all lanes of a predictor render identically whichever
        lane runs a chunk."""
        other = MeshStore(self.object_ds, self.device)
        other.set_raster_conventions(self.get_raster_conventions())
        other.set_backface_culling(self.get_backface_culling())
        self._followers.append(weakref.ref(other))
        return other

    def _live_followers(self) -> List["MeshStore"]:
        live = [(r, r()) for r in self._followers]
        self._followers = [r for r, o in live if o is not None]
        return [o for _, o in live if o is not None]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().hp_mesh_store_destroy(C.c_void_p(h))
            except Exception:
                pass

    @property
    def handle(self):
        return C.c_void_p(self._h)

    def ids_of(self, labels: Sequence[str]) -> torch.Tensor:
        """Object ids of ``labels`` on the device.  The last few label lists are remembered: a refiner is called frame after frame
        with the same table, and the host-to-device copy of the ids sits in front of a call's first launch.  The tensor is SHARED
        between the calls that ask for the same labels: read-only for the callers (they slice it and hand it to kernels)."""
        key = tuple(labels)
        hit = self._ids_cache.get(key)
        if hit is None:
            if len(self._ids_cache) >= 16:
                self._ids_cache.pop(next(iter(self._ids_cache)))
            hit = self._ids_cache[key] = torch.as_tensor([self.label_to_id[l] for l in labels], dtype=torch.int32, device=self.device)
        return hit

    def reserve_raster(self, n_views: int, resolution: Tuple[int, int] = (240, 320), msaa: bool = False) -> None:
        """Size the rasteriser scratch for ``n_views`` views per call once (``hp_mesh_store_reserve_raster``): later
        calls of up to that size never reallocate it, so captured hipGraphs keep valid pointers."""
        with torch.cuda.device(self.device):
            check(lib().hp_mesh_store_reserve_raster(self.handle, int(n_views), resolution[0], resolution[1], 32 if msaa else 0),
                  "hp_mesh_store_reserve_raster")

    def scratch_generation(self) -> int:
        """Number of times the rasteriser scratch was reallocated (``hp_mesh_store_scratch_generation``): graphs
        captured under an older generation hold freed pointers."""
        return int(lib().hp_mesh_store_scratch_generation(self.handle))

    def set_raster_conventions(self, conv: Optional[Dict] = None) -> None:
        """``hp_mesh_store_set_raster_conventions``: the three renderer conventions nobody can pin without Panda3D --
        multisample positions, the anisotropic filter's probe-count / level-of-detail rule, the axis / sign map of the eye-normal
        code (``TB/renderer/panda3d_scene_renderer.py:68-71,221-230``, ``TB/renderer/utils.py:63-79``) -- of THIS store's renders
        (two stores in one process may differ).  ``conv``: ``None`` = defaults, or a dict overriding any of
        :data:`RASTER_CONVENTION_DEFAULTS`.  Read at launch time; captured graphs are dropped (``bump_graph_epoch``).
        ``tools/calibrate_renderer.py`` fits the record to Panda3D renders."""
        if conv is None:
            check(lib().hp_mesh_store_set_raster_conventions(self.handle, None), "hp_mesh_store_set_raster_conventions")
        else:
            d = dict(RASTER_CONVENTION_DEFAULTS)
            unknown = set(conv) - set(d)
            if unknown:
                raise KeyError(f"unknown raster convention(s): {sorted(unknown)}")
            d.update(conv)
            c = RasterConventions((C.c_float * 4)(*d["msaa_x"]), (C.c_float * 4)(*d["msaa_y"]), int(d["aniso_max"]), int(d["aniso_round"]),
                                  int(d["lod_from"]), float(d["lod_bias"]), float(d["aniso_ratio_bias"]), (C.c_int * 3)(*d["normal_axis"]),
                                  (C.c_float * 3)(*d["normal_sign"]))
            check(lib().hp_mesh_store_set_raster_conventions(self.handle, C.byref(c)), "hp_mesh_store_set_raster_conventions")
        for f in self._live_followers():
            f.set_raster_conventions(conv)
        bump_graph_epoch()

    def get_raster_conventions(self) -> Dict:
        c = RasterConventions()
        check(lib().hp_mesh_store_get_raster_conventions(self.handle, C.byref(c)), "hp_mesh_store_get_raster_conventions")
        return dict(msaa_x=tuple(c.msaa_x), msaa_y=tuple(c.msaa_y), aniso_max=c.aniso_max, aniso_round=c.aniso_round, lod_from=c.lod_from,
                    lod_bias=c.lod_bias, aniso_ratio_bias=c.aniso_ratio_bias, normal_axis=tuple(c.normal_axis), normal_sign=tuple(c.normal_sign))

    def set_backface_culling(self, on: bool = True) -> bool:
        """``hp_mesh_store_set_backface_culling``: drop triangles whose inward side is turned to the camera when they belong to a
        closed, consistently oriented connected component of the mesh seen from outside (they can never be seen; the renders stay
        two-sided like the reference's for everything else).  Per store; returns the previous setting.  Default on;
        ``HP_RASTER_NO_CULL=1`` creates stores with it off."""
        prev = bool(lib().hp_mesh_store_set_backface_culling(self.handle, 1 if on else 0))
        for f in self._live_followers():
            f.set_backface_culling(on)
        bump_graph_epoch()
        return prev

    def get_backface_culling(self) -> bool:
        """``hp_mesh_store_get_backface_culling``: the current setting."""
        return bool(lib().hp_mesh_store_get_backface_culling(self.handle))

    def point_ids(self, n_points: int) -> torch.Tensor:
        """ids of ``sample_points(n, deterministic=True)`` (TB/lib3d/mesh_ops.py:74-84)."""
        if n_points not in self._point_ids:
            ids = sample_point_ids(self.n_pad, n_points).astype(np.int32)
            self._point_ids[n_points] = torch.as_tensor(ids, device=self.device)
        return self._point_ids[n_points]


def nchw_strides(c: int, h: int, w: int) -> Strides:
    return Strides(c * h * w, 0, h * w, w, 1)


class RasterConventions(C.Structure):
    """``hp_raster_conventions`` (include/happypose_amd.h)."""
    _fields_ = [("msaa_x", C.c_float * 4), ("msaa_y", C.c_float * 4), ("aniso_max", C.c_int), ("aniso_round", C.c_int),
                ("lod_from", C.c_int), ("lod_bias", C.c_float), ("aniso_ratio_bias", C.c_float), ("normal_axis", C.c_int * 3), ("normal_sign", C.c_float * 3)]


RASTER_CONVENTION_DEFAULTS = dict(msaa_x=(0.375, 0.875, 0.125, 0.625), msaa_y=(0.125, 0.375, 0.625, 0.875), aniso_max=16,
                                  aniso_round=0, lod_from=0, lod_bias=0.0, aniso_ratio_bias=0.0, normal_axis=(0, 1, 2), normal_sign=(1.0, -1.0, -1.0))


def rasterize(store: MeshStore, obj_ids: torch.Tensor, TCO: torch.Tensor, K: torch.Tensor,
              resolution: Tuple[int, int], render_normals=False, render_depth=False,
              render_binary_mask=False, ambient: Optional[torch.Tensor] = None,
              light_pos: Optional[torch.Tensor] = None, light_col: Optional[torch.Tensor] = None,
              quant8: bool = True, msaa: bool = False, aniso: bool = False, render_rgb: bool = True):
    """NCHW outputs shaped like ``BatchRenderOutput`` (TB/renderer/types.py:45-56; ``render_rgb=False``: no colour buffer,
    ``rgbs`` is None -- the depth-only launches of a C caller).  ``msaa``: 4x multisampled colour /
    normal buffers (``HP_RASTER_MSAA4``), ``aniso``: mip-mapped trilinear + anisotropic-16 texture filtering
    (``HP_RASTER_TEX_ANISO``) -- the reference renderer's framebuffer / texture state."""
    dev = store.device
    n = TCO.shape[0]
    assert TCO.shape == (n, 4, 4) and K.shape == (n, 3, 3)
    assert obj_ids.shape == (n,)
    if render_binary_mask:
        assert render_depth, "Binary mask can only be rendered if depth is rendered"
    h, w = resolution
    TCO = _f32(TCO, dev)
    K = _f32(K, dev)
    obj_ids = _i32(obj_ids, dev)
    rgb = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev) if render_rgb else None
    nrm = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev) if render_normals else None
    dep = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev) if render_depth else None
    msk = torch.empty((n, 1, h, w), dtype=torch.uint8, device=dev) if render_binary_mask else None
    n_lights = 0
    if light_pos is not None:
        light_pos, light_col = _f32(light_pos, dev), _f32(light_col, dev)
        n_lights = light_pos.shape[1]
    if ambient is not None:
        ambient = _f32(ambient, dev)
    cs, ds = nchw_strides(3, h, w), nchw_strides(1, h, w)
    with torch.cuda.device(dev):
        check(lib().hp_rasterize(store.handle, n, 1, ptr(obj_ids), ptr(TCO), ptr(K), ptr(ambient), n_lights,
                                 ptr(light_pos), ptr(light_col), h, w, (8 if quant8 else 0) | (32 if msaa else 0) | (64 if aniso else 0), ptr(rgb), ptr(nrm),
                                 C.byref(cs), ptr(dep), C.byref(ds), ptr(msk), None, 0, stream_ptr(dev)),
              "hp_rasterize")
    return rgb, nrm, dep, (msk.bool() if msk is not None else None)


def render_inputs(store: MeshStore, x: torch.Tensor, obj_ids: torch.Tensor, TCV_O: torch.Tensor, KV: torch.Tensor,
                  render_normals: bool, render_depth: bool, *, images: Optional[torch.Tensor] = None,
                  boxes: Optional[torch.Tensor] = None, im_ids: Optional[torch.Tensor] = None, n_img_channels: int = 0,
                  depth_norm_z: Optional[torch.Tensor] = None, depth_norm_mode: int = 0, chan0: Optional[int] = None,
                  layout=None, ambient: Optional[torch.Tensor] = None, light_pos: Optional[torch.Tensor] = None,
                  light_col: Optional[torch.Tensor] = None, msaa: bool = False, aniso: bool = False,
                  sampling_ratio: int = 4) -> None:
    """``hp_render_inputs``: the network input ``x [b,h,w,c_rec]`` (fp32 or fp16) of one iteration in ONE pass -- ``V``
    rendered views per hypothesis (channels rgb, normals, depth in the reference's order,
    ``MP/models/pose_rigid.py:437-453``) and, when ``images`` is given, the observed crop (``crop_images`` =
    torchvision ``roi_align`` of frame ``im_ids[i]`` over ``boxes[i]``, ``TB/lib3d/cropping.py:155-197``, with the RGB-D
    validity rule and the depth normalisation of ``normalize_images``).  Default layout = the reference's
    ``cat((images_crop, renders))``: crop channels ``[0, n_img_channels)`` (written by view 0's workgroups), view ``v`` at
    ``chan0 + v * C_r`` (``chan0`` defaults to ``n_img_channels``).  ``layout`` = ``(view_c0, crop_c0, crop_src0,
    crop_n)`` lists per view overrides it (permuted inputs).  Every pixel record is written once."""
    dev = store.device
    b, h, w, cp = x.shape
    V = TCV_O.shape[1]
    assert TCV_O.shape == (b, V, 4, 4) and KV.shape == (b, V, 3, 3) and 1 <= V <= 8
    c_r = 3 + (3 if render_normals else 0) + (1 if render_depth else 0)
    assert x.dtype in (torch.float32, torch.float16) and x.is_contiguous()
    TCV_O, KV, obj_ids = _f32(TCV_O, dev), _f32(KV, dev), _i32(obj_ids, dev)
    crop = images is not None
    lay = _ffi.InputLayout()
    if layout is None:
        c0 = (n_img_channels if crop else 0) if chan0 is None else chan0
        assert c0 + V * c_r <= cp
        for v in range(V):
            lay.view_c0[v] = c0 + v * c_r
        if crop:
            lay.crop_n[0], lay.crop_c0[0], lay.crop_src0[0] = n_img_channels, 0, 0
    else:
        for name, vals in zip(("view_c0", "crop_c0", "crop_src0", "crop_n"), layout):
            assert len(vals) == V
            for v, val in enumerate(vals):
                getattr(lay, name)[v] = int(val)
    Bi = Ct = H = W = 0
    if crop:
        assert images.dtype == torch.float32 and images.is_contiguous() and images.dim() == 4
        Bi, Ct, H, W = images.shape
        _check_ids(im_ids, Bi, "render_inputs: im_ids -> images")
        boxes, im_ids = _f32(boxes, dev), _i32(im_ids, dev)
        assert boxes.shape == (b, 4) and im_ids.shape == (b,)
    n_lights = 0
    if light_pos is not None:  # [b*V, L, 3] object-frame positions + colours of point lights, ambient [b*V, 3]
        light_pos, light_col = _f32(light_pos, dev), _f32(light_col, dev)
        n_lights = light_pos.shape[1]
        assert light_pos.shape == (b * V, n_lights, 3) and light_col.shape == (b * V, n_lights, 3)
    if ambient is not None:
        ambient = _f32(ambient, dev)
        assert ambient.shape == (b * V, 3)
    flags = 8 | (16 if x.dtype == torch.float16 else 0) | (32 if msaa else 0) | (64 if aniso else 0) | \
        (0x1000 if render_normals else 0) | (0x2000 if render_depth else 0)
    mode = depth_norm_mode if (render_depth or (crop and n_img_channels == 4)) else 0
    with torch.cuda.device(dev):
        check(lib().hp_render_inputs(store.handle, b, V, ptr(obj_ids), ptr(TCV_O), ptr(KV), ptr(ambient), n_lights, ptr(light_pos),
                                     ptr(light_col), h, w, flags, ptr(images) if crop else None, Bi, Ct, H, W,
                                     ptr(boxes) if crop else None, ptr(im_ids) if crop else None, sampling_ratio,
                                     ptr(depth_norm_z) if mode else None, mode, C.c_void_p(x.data_ptr()), cp, C.byref(lay),
                                     stream_ptr(dev)), "hp_render_inputs")


def rasterize_into(store: MeshStore, x: torch.Tensor, chan0: int, obj_ids: torch.Tensor,
                   TCV_O: torch.Tensor, KV: torch.Tensor, render_normals: bool, render_depth: bool,
                   depth_norm_z: Optional[torch.Tensor] = None, depth_norm_mode: int = 0,
                   ambient: Optional[torch.Tensor] = None, msaa: bool = False, aniso: bool = False,
                   light_pos: Optional[torch.Tensor] = None, light_col: Optional[torch.Tensor] = None) -> None:
    """Render ``V`` views per hypothesis straight into channel slices of the NHWC network
    input ``x [b,h,w,c_pad]``: view ``v`` occupies channels ``chan0 + v*C_r ...`` in the
    reference's order rgb, normals, depth (MP/models/pose_rigid.py:437-453).  (:func:`render_inputs` without the crop.)"""
    render_inputs(store, x, obj_ids, TCV_O, KV, render_normals, render_depth, depth_norm_z=depth_norm_z,
                  depth_norm_mode=depth_norm_mode, chan0=chan0, ambient=ambient, light_pos=light_pos, light_col=light_col,
                  msaa=msaa, aniso=aniso)


def pose_prep(store: MeshStore, TCO: torch.Tensor, K: torch.Tensor, im_ids: torch.Tensor,
              obj_ids: torch.Tensor, im_size: Tuple[int, int], crop_size: Tuple[int, int] = (240, 320),
              multiview_type: str = "TCO", normalize: bool = False, n_points: int = 2000,
              n_points_extra: int = 200, lamb: float = 1.4, remove_TCO_rendering: bool = False):
    """Returns ``dict(TCO, tCR, TCV_O [b,V,4,4], boxes_rend, boxes_crop, K_crop [b,V,3,3], K_crop_main [b,3,3])``.
    ``K_crop_main`` is the K of the observed crop (``crop_inputs``): view 0 of ``K_crop`` unless
    ``remove_TCO_rendering`` (the TCO view is then not among the ``V`` rendered views; every view carries the K of its
    own 200-point crop, ``MP/models/pose_rigid.py:598-611``)."""
    dev = store.device
    b = TCO.shape[0]
    assert TCO.shape == (b, 4, 4) and K.dim() == 3 and K.shape[1:] == (3, 3)
    mv, V = MULTIVIEW[multiview_type]
    if remove_TCO_rendering:
        assert V >= 3, "remove_TCO_rendering needs a multi-view type with at least two look-at views"
        V -= 1
    TCO, K = _f32(TCO, dev), _f32(K, dev)
    _check_ids(im_ids, K.shape[0], "pose_prep: im_ids -> K")
    _check_ids(obj_ids, len(store.labels), "pose_prep: obj_ids -> objects")
    im_ids, obj_ids = _i32(im_ids, dev), _i32(obj_ids, dev)
    assert im_ids.shape == (b,) and obj_ids.shape == (b,)
    f = dict(dtype=torch.float32, device=dev)
    out = dict(TCO=torch.empty((b, 4, 4), **f), tCR=torch.empty((b, 3), **f),
               TCV_O=torch.empty((b, V, 4, 4), **f), boxes_rend=torch.empty((b, 4), **f),
               boxes_crop=torch.empty((b, 4), **f), K_crop=torch.empty((b, V, 3, 3), **f))
    ids_main = store.point_ids(n_points)
    multi = V > 1 or remove_TCO_rendering
    ids_extra = store.point_ids(n_points_extra) if multi else None
    with torch.cuda.device(dev):
        if remove_TCO_rendering:
            out["K_crop_main"] = torch.empty((b, 3, 3), **f)
            check(lib().hp_pose_prep_views(store.handle, b, V, mv, 1, int(normalize), ptr(TCO), ptr(K), K.shape[0], ptr(im_ids),
                                           ptr(obj_ids), ptr(ids_main), n_points, ptr(ids_extra), n_points_extra, im_size[0],
                                           im_size[1], crop_size[0], crop_size[1], C.c_float(lamb), ptr(out["TCO"]),
                                           ptr(out["tCR"]), ptr(out["TCV_O"]), ptr(out["boxes_rend"]), ptr(out["boxes_crop"]),
                                           ptr(out["K_crop"]), ptr(out["K_crop_main"]), stream_ptr(dev)), "hp_pose_prep_views")
        else:
            check(lib().hp_pose_prep(store.handle, b, V, mv, int(normalize), ptr(TCO), ptr(K), K.shape[0], ptr(im_ids),
                                     ptr(obj_ids), ptr(ids_main), n_points, ptr(ids_extra),
                                     n_points_extra if multi else 0, im_size[0], im_size[1], crop_size[0],
                                     crop_size[1], C.c_float(lamb), ptr(out["TCO"]), ptr(out["tCR"]),
                                     ptr(out["TCV_O"]), ptr(out["boxes_rend"]), ptr(out["boxes_crop"]),
                                     ptr(out["K_crop"]), stream_ptr(dev)), "hp_pose_prep")
            out["K_crop_main"] = out["K_crop"][:, 0]
    return out


def crop_roi_align(images: torch.Tensor, boxes: torch.Tensor, im_ids: torch.Tensor,
                   output_size=(240, 320), sampling_ratio: int = 4, out: Optional[torch.Tensor] = None,
                   depth_norm_z: Optional[torch.Tensor] = None, depth_norm_mode: int = 0,
                   n_channels: Optional[int] = None, owns_record: bool = False) -> torch.Tensor:
    """``crop_images`` (TB/lib3d/cropping.py:155-197).  ``out=None`` -> NCHW ``[n,C,oh,ow]``;
    otherwise ``out`` is the NHWC network input ``[n,oh,ow,c_pad]`` and channels 0..C-1 are
    written."""
    dev = images.device
    Bi, Ct, H, W = images.shape
    Cc = Ct if n_channels is None else n_channels
    n = boxes.shape[0]
    oh, ow = output_size
    assert images.dtype == torch.float32 and images.is_contiguous()
    _check_ids(im_ids, Bi, "crop_roi_align: im_ids -> images")
    boxes, im_ids = _f32(boxes, dev), _i32(im_ids, dev)
    assert boxes.shape == (n, 4) and im_ids.shape == (n,)
    if out is None:
        res = torch.empty((n, Cc, oh, ow), dtype=torch.float32, device=dev)
        st = Strides(Cc * oh * ow, 0, oh * ow, ow, 1)
    else:
        res = out
        assert out.shape[:3] == (n, oh, ow) and out.shape[3] >= Cc and out.is_contiguous()
        assert out.dtype in (torch.float32, torch.float16)
        cp = out.shape[3]
        st = Strides(oh * ow * cp, 0, 1, ow * cp, cp)
    fn = lib().hp_crop_roi_align_f16 if res.dtype == torch.float16 else lib().hp_crop_roi_align
    mode = depth_norm_mode if Cc == 4 else 0
    # ``owns_record``: the caller promises that the rest of every pixel record may be zeroed (the rasteriser writes it
    # afterwards): the first 8 floats of every record are then stored as a whole 32-B sector (HP_CROP_FULL_RECORD8)
    if owns_record and out is not None and res.shape[3] % (8 if res.dtype == torch.float32 else 16) == 0 and res.data_ptr() % 32 == 0:
        mode |= 0x100
    with torch.cuda.device(dev):
        check(fn(ptr(images), Bi, Ct, Cc, H, W, ptr(boxes), ptr(im_ids), n, oh, ow,
                 sampling_ratio, ptr(res), C.byref(st),
                 ptr(depth_norm_z), mode, stream_ptr(dev)),
              "hp_crop_roi_align")
    return res


def pose_update(TCO: torch.Tensor, K_crop: torch.Tensor, pose9: torch.Tensor,
                tCR: Optional[torch.Tensor] = None) -> torch.Tensor:
    dev = TCO.device
    b = TCO.shape[0]
    assert TCO.shape == (b, 4, 4) and pose9.shape == (b, 9)
    assert K_crop.shape[0] == b and K_crop.shape[-2:] == (3, 3)
    k_stride = K_crop[0].numel()
    TCO, K_crop, pose9 = _f32(TCO, dev), _f32(K_crop, dev), _f32(pose9, dev)
    if tCR is not None:
        assert tCR.shape == (b, 3)
        tCR = _f32(tCR, dev)
    out = torch.empty_like(TCO)
    with torch.cuda.device(dev):
        check(lib().hp_pose_update(b, ptr(TCO), ptr(K_crop), k_stride, ptr(pose9), ptr(tCR), ptr(out),
                                   stream_ptr(dev)), "hp_pose_update")
    return out


def tco_init_autodepth(store: MeshStore, boxes: torch.Tensor, K: torch.Tensor, im_ids, obj_ids,
                       R: Optional[torch.Tensor] = None, box_ids=None, rot_ids=None,
                       n_points: Optional[int] = None) -> torch.Tensor:
    dev = store.device
    n = len(obj_ids)
    boxes, K = _f32(boxes, dev), _f32(K, dev)
    if R is not None:
        R = _f32(R, dev)
    assert boxes.dim() == 2 and boxes.shape[1] == 4 and K.dim() == 3 and K.shape[1:] == (3, 3)
    assert R is None or (R.dim() == 3 and R.shape[1:] == (3, 3))
    assert box_ids is not None or boxes.shape[0] == n, "one box per hypothesis unless box_ids is given"
    assert R is None or rot_ids is not None or R.shape[0] == n, "one rotation per hypothesis unless rot_ids is given"
    _check_ids(im_ids, K.shape[0], "tco_init_autodepth: im_ids -> K")
    _check_ids(obj_ids, len(store.labels), "tco_init_autodepth: obj_ids -> objects")
    _check_ids(box_ids, boxes.shape[0], "tco_init_autodepth: box_ids -> boxes")
    if R is not None:
        _check_ids(rot_ids, R.shape[0], "tco_init_autodepth: rot_ids -> R")
    im_ids, obj_ids = _i32(im_ids, dev), _i32(obj_ids, dev)
    box_ids = None if box_ids is None else _i32(box_ids, dev)
    rot_ids = None if rot_ids is None else _i32(rot_ids, dev)
    out = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
    pids = None if n_points is None else store.point_ids(n_points)
    with torch.cuda.device(dev):
        check(lib().hp_tco_init_autodepth(store.handle, n, ptr(boxes), boxes.shape[0], ptr(box_ids), ptr(K), K.shape[0],
                                          ptr(im_ids), ptr(obj_ids), ptr(R), 0 if R is None else R.shape[0],
                                          ptr(rot_ids), ptr(pids), n_points or 0,
                                          ptr(out), stream_ptr(dev)),
              "hp_tco_init_autodepth")
    return out


class Net:
    """``hp_net``: backbone + heads with BN folded, on one device."""

    profiling = False  # set_profiling(True): conv stretches are timed with HIP events (no graph capture then)
    tail_split = True  # hp_net_set_tail_split state: changes the launch plan, so it is part of a graph signature
    _exact_only = False  # last seen HP_STATUS_EXACT_ONLY (the guard's switch to the exact-fp32 kernels)

    def __init__(self, arch: str, n_inputs: int, state_dict: Dict[str, "np.ndarray | torch.Tensor"],
                 max_batch: int = 128, device="cuda", h: int = 240, w: int = 320, precision: str = "f32"):
        """``precision``: ``"f32"`` (the reference's arithmetic) or ``"f16"`` (fp16 weights and
        activations, fp32 accumulation: configuration C5; inputs / outputs stay fp32)."""
        self.device = torch.device(device)
        self.arch, self.n_inputs, self.h, self.w = arch, n_inputs, h, w
        self.precision = precision
        self.max_batch = max_batch
        self.n_features = N_FEATURES[arch]
        with torch.cuda.device(self.device):
            self._h = lib().hp_net_create(ARCH[arch], n_inputs, h, w)
            if not self._h:
                raise _ffi.HipLibraryError("hp_net_create: " + lib().hp_last_error().decode())
            for name, value in state_dict.items():
                if name.endswith("num_batches_tracked"):
                    continue
                arr = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                check(lib().hp_net_set_param(self.handle, name.encode(), _np_ptr(arr), arr.size),
                      f"hp_net_set_param({name})")
            check(lib().hp_net_set_precision(self.handle, {"f32": 0, "f16": 1}[precision]), "hp_net_set_precision")
            check(lib().hp_net_finalize(self.handle, max_batch), "hp_net_finalize")
        self.c_pad = lib().hp_net_input_channels_padded(self.handle)
        self.pose_dim = state_dict["pose_fc.weight"].shape[0] if "pose_fc.weight" in state_dict else 0
        self.n_logits = (state_dict["views_logits_head.weight"].shape[0]
                         if "views_logits_head.weight" in state_dict else 0)
        self.flops_per_sample = lib().hp_net_flops_per_sample(self.handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().hp_net_destroy(C.c_void_p(h))
            except Exception:
                pass

    @property
    def handle(self):
        return C.c_void_p(self._h)

    def new_input(self, batch: int) -> torch.Tensor:
        """Zeroed NHWC input buffer (pad channels must stay 0): fp32 ``[batch,h,w,c_pad]``, or for an
        fp16 plan fp16 ``[batch,h,w,c16]`` -- crop and rasteriser write it directly and ``forward``
        skips the conversion pass (``hp_net_forward_f16in``)."""
        if self.precision == "f16":
            c16 = lib().hp_net_input_channels_f16(self.handle)
            assert c16 > 0, lib().hp_last_error().decode()
            return torch.zeros((batch, self.h, self.w, c16), dtype=torch.float16, device=self.device)
        return torch.zeros((batch, self.h, self.w, self.c_pad), dtype=torch.float32, device=self.device)

    def forward(self, x: torch.Tensor, want_pose=True, want_logits=False, want_features=False):
        b = x.shape[0]
        f = dict(dtype=torch.float32, device=self.device)
        if x.dtype == torch.float16:
            assert self.precision == "f16" and x.is_contiguous() and x.shape[:3] == (b, self.h, self.w)
            assert x.shape[3] == lib().hp_net_input_channels_f16(self.handle)
            pose = torch.empty((b, self.pose_dim), **f) if (want_pose and self.pose_dim) else None
            logits = torch.empty((b, self.n_logits), **f) if (want_logits and self.n_logits) else None
            feats = torch.empty((b, self.n_features), **f) if want_features else None
            with torch.cuda.device(self.device):
                check(lib().hp_net_forward_f16in(self.handle, C.c_void_p(x.data_ptr()), b, ptr(pose), ptr(logits), ptr(feats),
                                                 stream_ptr(self.device)), "hp_net_forward_f16in")
            return pose, logits, feats
        assert x.shape == (b, self.h, self.w, self.c_pad) and x.is_contiguous() and x.dtype == torch.float32
        pose = torch.empty((b, self.pose_dim), **f) if (want_pose and self.pose_dim) else None
        logits = torch.empty((b, self.n_logits), **f) if (want_logits and self.n_logits) else None
        feats = torch.empty((b, self.n_features), **f) if want_features else None
        with torch.cuda.device(self.device):
            check(lib().hp_net_forward(self.handle, ptr(x), b, ptr(pose), ptr(logits), ptr(feats),
                                       stream_ptr(self.device)), "hp_net_forward")
        return pose, logits, feats

    def feature_maps(self, batch: int):
        """Feature-pyramid networks (``resnet50-fpn``): the output maps of the last ``forward`` as NHWC tensors
        ``[batch, h, w, c]`` (``hp_net_copy_feature_map``)."""
        outs = []
        for i in range(lib().hp_net_n_feature_maps(self.handle)):
            h, w, c = C.c_int(0), C.c_int(0), C.c_int(0)
            check(lib().hp_net_feature_map(self.handle, i, None, C.byref(h), C.byref(w), C.byref(c)), "hp_net_feature_map")
            t = torch.empty((batch, h.value, w.value, c.value), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                check(lib().hp_net_copy_feature_map(self.handle, i, batch, ptr(t), stream_ptr(self.device)), "hp_net_copy_feature_map")
            outs.append(t)
        return outs

    def set_profiling(self, on: bool):
        check(lib().hp_net_set_profiling(self.handle, int(on)), "hp_net_set_profiling")
        self.profiling = bool(on)
        bump_graph_epoch()  # event records change the launch sequence a captured graph holds

    def set_conv_algo(self, name: Optional[str] = None):
        """Kernel families THIS network may use (``hp_net_set_conv_algo``; names of :data:`CONV_ALGOS`);
        ``None`` returns it to ``auto``."""
        check(lib().hp_net_set_conv_algo(self.handle, -1 if name is None else CONV_ALGOS[name]), "hp_net_set_conv_algo")
        bump_graph_epoch()

    def set_tail_split(self, on: bool):
        """K-slicing of the tail tiles of this network's conv launches (``hp_net_set_tail_split``): off while a
        second lane shares the GPU."""
        check(lib().hp_net_set_tail_split(self.handle, int(on)), "hp_net_set_tail_split")
        self.tail_split = bool(on)

    def set_act_scale(self, on: bool):
        """Dynamic power-of-two activation scale of the split-fp16 kernels (``hp_net_set_act_scale``; default on)."""
        check(lib().hp_net_set_act_scale(self.handle, int(on)), "hp_net_set_act_scale")
        bump_graph_epoch()  # launch arguments change

    def status(self, stream=None) -> int:
        """``hp_net_status``: waits for ``stream`` (default: the current one) and returns the guard flags --
        bit 0 (:data:`STATUS_NONFINITE`): a forward since the last call produced inf / NaN in a split-fp16
        layer (an activation beyond the fp16 range), its outputs are invalid; bit 1 (:data:`STATUS_EXACT_ONLY`):
        the network now runs the exact-fp32 kernels only, so re-running the same inputs is valid."""
        flags = C.c_int(0)
        sp = stream_ptr(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        with torch.cuda.device(self.device):
            check(lib().hp_net_status(self.handle, sp, C.byref(flags)), "hp_net_status")
        exact = bool(flags.value & STATUS_EXACT_ONLY)
        if (flags.value & STATUS_NONFINITE) or exact != self._exact_only:
            # the network switched kernels (or just poisoned a forward): captured graphs still hold the old launches.
            # Only the TRANSITION bumps: the sticky EXACT_ONLY bit alone would otherwise drop every graph cache of the
            # process after every stage of the estimators
            bump_graph_epoch()
        self._exact_only = exact
        return flags.value

    def force_exact(self, on: bool = True) -> None:
        """``hp_net_force_exact``: put the network on (or take it off) the exact-fp32 kernels the guard switches to."""
        check(lib().hp_net_force_exact(self.handle, int(bool(on))), "hp_net_force_exact")
        if bool(on) != self._exact_only:
            bump_graph_epoch()
        self._exact_only = bool(on)

    def profile_collect(self):
        """``(conv_ms, n_launches, conv_flops, mfma_flops)`` of the conv launches recorded since
        the last call (HIP events on the launch stream; waits for them): algorithmic FLOPs of the
        direct convolutions and FLOPs the matrix cores executed (Winograd layers execute 2.25x
        fewer, padded tiles more)."""
        ms, n, fl, mfl = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        check(lib().hp_net_profile_collect(self.handle, C.byref(ms), C.byref(n), C.byref(fl), C.byref(mfl)),
              "hp_net_profile_collect")
        return ms.value, n.value, fl.value, mfl.value

    def profile_intervals(self):
        """``[(t0_ms, t1_ms), ...]`` of the timed conv stretches pending for this network, relative to
        :func:`profile_mark_reference` (call before :meth:`profile_collect`)."""
        n = lib().hp_net_profile_intervals(self.handle, None, None, 0)
        if n < 0:
            check(n, "hp_net_profile_intervals")
        a, b = (C.c_double * n)(), (C.c_double * n)()
        got = lib().hp_net_profile_intervals(self.handle, a, b, n)
        if got < 0:
            check(got, "hp_net_profile_intervals")
        return list(zip(list(a), list(b)))


class GraphNet(Net):
    """``HP_ARCH_CUSTOM``: a feed-forward graph of convolutions described layer by layer (the detector's RoI heads).
    ``layers``: dicts ``weight, bias, cin, cout, k, stride, pad, relu, H, W, src, dst, res`` in execution order
    (``src = -1`` = the network input, arena slots 0..31); ``outputs``: ``(slot, H, W, C)`` read back by
    :meth:`Net.feature_maps` (``C`` rounded up to 4)."""

    def __init__(self, c_in: int, h: int, w: int, layers, outputs, state_dict, max_batch: int, device="cuda"):
        self.device = torch.device(device)
        self.arch, self.n_inputs, self.h, self.w, self.precision = "custom", c_in, h, w, "f32"
        self.n_features = self.pose_dim = self.n_logits = 0
        with torch.cuda.device(self.device):
            self._h = lib().hp_net_create(5, c_in, h, w)
            if not self._h:
                raise _ffi.HipLibraryError("hp_net_create: " + lib().hp_last_error().decode())
            for L in layers:
                check(lib().hp_net_add_conv(self.handle, L["weight"].encode(), (L.get("bias") or "").encode(), L["cin"], L["cout"], L["k"],
                                            L.get("stride", 1), L.get("pad", 0), int(L.get("relu", False)), L["H"], L["W"], L["src"],
                                            L["dst"], L.get("res", -1)), f"hp_net_add_conv({L['weight']})")
            for slot, oh, ow, oc in outputs:
                check(lib().hp_net_add_output(self.handle, slot, oh, ow, (oc + 3) // 4 * 4), "hp_net_add_output")
            for name, value in state_dict.items():
                arr = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                check(lib().hp_net_set_param(self.handle, name.encode(), _np_ptr(arr), arr.size), f"hp_net_set_param({name})")
            check(lib().hp_net_finalize(self.handle, max_batch), "hp_net_finalize")
        self.c_pad = lib().hp_net_input_channels_padded(self.handle)
        self.max_batch = max_batch
        self.flops_per_sample = lib().hp_net_flops_per_sample(self.handle)

    def run(self, x: torch.Tensor):
        """``x [n,h,w,c_in]`` NHWC fp32 -> list of output maps ``[n,oh,ow,oc4]`` (chunks of ``max_batch``)."""
        n = x.shape[0]
        outs = None
        for s in range(0, n, self.max_batch):
            xb = x[s:s + self.max_batch].contiguous()
            self.forward(xb, want_pose=False)
            maps = self.feature_maps(xb.shape[0])
            outs = [[m] for m in maps] if outs is None else [o + [m] for o, m in zip(outs, maps)]
        return [torch.cat(o) for o in outs] if outs is not None else []


STATUS_NONFINITE, STATUS_EXACT_ONLY = 1, 2


def profile_mark_reference(device) -> None:
    """Record the process-wide reference event of :meth:`Net.profile_intervals` on the current stream."""
    with torch.cuda.device(device):
        check(lib().hp_profile_mark_reference(stream_ptr(device)), "hp_profile_mark_reference")


def probe_mfma_rate(device, random_data: bool = True):
    """``hp_probe_mfma_rate``: ``(TFLOP/s, shader MHz)`` the fp16 matrix pipe sustains on zero / random operands."""
    tf, mhz = C.c_double(0), C.c_double(0)
    with torch.cuda.device(device):
        check(lib().hp_probe_mfma_rate(int(random_data), C.byref(tf), C.byref(mhz), stream_ptr(device)), "hp_probe_mfma_rate")
    return tf.value, mhz.value


CONV_ALGOS = {"auto": 0, "direct": 1, "igemm": 2, "winograd-1wave": 3, "winograd": 4, "split": 5}


def select_conv_algo(name: str = "auto") -> None:
    """The kernel family of the SINGLE-LAYER entry point ``conv2d_nhwc`` (``hp_conv_select_algo``; parity tests and
    ``tools/conv_fuzz.py`` walk the families with it).  Networks never read it -- a network's choice is
    :meth:`Net.set_conv_algo`, default ``auto``.  ``auto`` = the split-fp16 kernels where they apply, else Winograd
    F(2x2,3x3), else the patch-staged direct kernel, else the generic implicit GEMM; ``winograd`` = exact-fp32 arithmetic
    only (``winograd-1wave``: the one-wave-per-SIMD schedule of that kernel); ``direct`` = no Winograd; ``igemm`` = the
    generic implicit-GEMM kernel only; ``split`` = the split-fp16 kernels."""
    check(lib().hp_conv_select_algo(CONV_ALGOS[name]), "hp_conv_select_algo")
    bump_graph_epoch()


def conv2d_nhwc(x, w_packed, stride, pad, bias=None, residual=None, pre_scale=None, pre_shift=None, relu=False):
    """Single conv layer (parity tests).  ``x [n,h,w,cin]``, ``w_packed [cout,kh,kw,cin]``.
    ``relu``: False/0 none, True/1 ReLU, 2 swish.  ``pre_scale [cin]`` + ``pre_shift [cin]`` = the
    BN+ReLU prologue; ``pre_scale [n,cin]`` alone = a squeeze-excitation gate on the input."""
    dev = x.device
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_packed.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().hp_conv2d_nhwc(ptr(x), n, h, w, cin, ptr(w_packed), cout, kh, kw, stride, pad, ptr(bias),
                                   ptr(residual), ptr(pre_scale), ptr(pre_shift), int(relu), ptr(y),
                                   stream_ptr(dev)), "hp_conv2d_nhwc")
    return y


def conv2d_nhwc_f16(x, w_packed, stride, pad, bias=None, residual=None, pre_scale=None, pre_shift=None, relu=False):
    """Single conv layer of the fp16 kernel (parity tests): ``x [n,h,w,cin]`` and ``w_packed
    [cout,kh,kw,cin]`` (and residual / pre_scale / pre_shift) fp16, bias fp32; returns fp16."""
    dev = x.device
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_packed.shape
    assert cin == cin2 and x.dtype == torch.float16 and w_packed.dtype == torch.float16
    for t in (residual, pre_scale, pre_shift):
        assert t is None or t.dtype == torch.float16
    assert bias is None or bias.dtype == torch.float32
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        check(lib().hp_conv2d_nhwc_f16(ptr(x), n, h, w, cin, ptr(w_packed), cout, kh, kw, stride, pad, ptr(bias),
                                       ptr(residual), ptr(pre_scale), ptr(pre_shift), int(relu), ptr(y),
                                       stream_ptr(dev)), "hp_conv2d_nhwc_f16")
    return y
