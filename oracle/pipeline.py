"""CPU restatement of the refinement loop itself (``PosePredictor.forward`` /
``forward_coarse``), assembled from the oracle's pieces exactly in the reference's order.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Also the timed ``cpu_baseline`` of
``bench.py`` (kind "port").  Follows ``MP/models/pose_rigid.py:546-674,708-788`` and
``CP/models/pose.py:116-199``; the reference's batching (``bsz_objects=8``, images gathered
per hypothesis, unfused conv -> BN -> ReLU in NCHW, torch-CPU/oneDNN) is kept.
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import backbones as OB
from . import geometry as G
from . import native


def _depth_norm(d, z, mode):
    """``normalize_depth`` (MP/models/pose_rigid.py:512-544)."""
    z = z.reshape((-1,) + (1,) * (d.ndim - 1))
    if mode in (None, "none"):
        return d
    if mode == "tCR_scale":
        return d / z
    if mode == "tCR_scale_clamp_center":
        return np.clip(d / z, 0, 2) - 1
    if mode == "tCR_center_clamp":
        return np.clip(d - z, -2, 2)
    raise ValueError(mode)


class OraclePredictor:
    """``cfg`` keys: arch, n_views, multiview_type, render_normals, render_depth, input_depth,
    depth_normalization_type, cosypose (bool), heads."""

    def __init__(self, state_dict: Dict[str, np.ndarray], packed_meshes, points_table: np.ndarray, **cfg):
        self.sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
        self.meshes = packed_meshes
        self.points = np.asarray(points_table, np.float32)  # [n_obj, n_pad, 3]
        self.cfg = dict(arch="resnet34", n_views=1, multiview_type="TCO", render_normals=False,
                        render_depth=False, input_depth=False, depth_normalization_type=None,
                        cosypose=False, remove_TCO_rendering=False, msaa=True, aniso=True)  # the reference's render state (panda3d_scene_renderer.py:68-71)
        self.cfg.update(cfg)
        self.render_size = (240, 320)

    # crop_inputs (MP/models/pose_rigid.py:199-277, CP/models/pose.py:58-93)
    def _crop_inputs(self, images, K, TCO, tCR, obj_ids, im_ids, n_pts=2000):
        ids = G.sample_point_ids(self.points.shape[1], n_pts)
        pts = self.points[obj_ids][:, ids]
        boxes_rend, boxes_crop = G.crop_boxes_from_pose(pts, K, TCO, tCR, images.shape[-2:])
        K_crop = G.get_K_crop_resize(K, boxes_crop, images.shape[-2:], self.render_size)
        return boxes_rend, boxes_crop, K_crop

    def _iteration(self, images, Kb, im_ids, obj_ids, TCO_in, heads):
        c = self.cfg
        b = len(TCO_in)
        n_img = 4 if c["input_depth"] else 3
        TCO = TCO_in if c["cosypose"] else G.normalize_T(TCO_in)
        tCR = TCO[:, :3, 3].copy()
        V = c["n_views"]
        skip = bool(c["remove_TCO_rendering"]) and V > 1  # forward_refiner: the look-at views only (pose_rigid.py:578-611)
        TCV_O = G.make_TCO_multiview(TCO, tCR, c["multiview_type"], V, remove_TCO_rendering=skip) if V > 1 else TCO[:, None].copy()
        boxes_rend, boxes_crop, K_crop = self._crop_inputs(images, Kb, TCO, tCR, obj_ids, im_ids)
        images_crop = native.crop_images(np.ascontiguousarray(images[:, :n_img]), boxes_crop, im_ids, self.render_size)
        KV = np.zeros((b, V, 3, 3), np.float32)
        if not skip:
            KV[:, 0] = K_crop
        for v in range(1 if not skip else 0, V):  # compute_crops_multiview: 200 points, boxes only
            _, _, KV[:, v] = self._crop_inputs(images, Kb, TCV_O[:, v], TCV_O[:, v, :3, 3], obj_ids, im_ids, 200)
        lights = {}
        if not c["cosypose"] and not c["render_normals"]:
            # MP/models/pose_rigid.py:415-422: without the normals channel the scene is lit by make_scene_lights()
            # (TB/renderer/panda3d_scene_renderer.py:105-141): ambient 0.1 + six point lights of colour 0.4 on the +/- axes at
            # 10 radii of root_node.getBounds() (the object's bounding sphere); CosyPose and render_normals use ambient 1
            ov = np.repeat(obj_ids, V)
            dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
            rad = np.asarray(self.meshes.bounds_radius, np.float64)[ov]
            lights = dict(ambient=np.full((len(ov), 3), 0.1, np.float32),
                          light_pos=(dirs[None] * rad[:, None, None] * 10).astype(np.float32),
                          light_col=np.full((len(ov), 6, 3), 0.4, np.float32))
        r = native.rasterize(self.meshes, np.repeat(obj_ids, V), TCV_O.reshape(-1, 4, 4), KV.reshape(-1, 3, 3),
                             self.render_size, c["render_normals"], c["render_depth"], msaa=c["msaa"], aniso=c["aniso"], **lights)
        parts = [r["rgbs"]]
        if c["render_normals"]:
            parts.append(r["normals"])
        if c["render_depth"]:
            parts.append(r["depths"])
        renders = np.concatenate(parts, axis=1)
        C_r = renders.shape[1]
        renders = renders.reshape(b, V * C_r, *self.render_size)
        mode = c["depth_normalization_type"]
        if c["input_depth"]:
            images_crop[:, 3:4] = _depth_norm(images_crop[:, 3:4], tCR[:, 2], mode)
        if c["render_depth"]:
            dd = [C_r - 1 + C_r * v for v in range(V)]
            renders[:, dd] = _depth_norm(renders[:, dd], tCR[:, 2], mode)
        x = torch.as_tensor(np.concatenate([images_crop, renders], axis=1))
        with torch.no_grad():
            out = OB.net_forward(x, self.sd, c["arch"], heads=heads)
        return dict(TCO=TCO, tCR=tCR, TCV_O=TCV_O, K_crop=K_crop, KV_crop=KV, boxes_rend=boxes_rend,
                    boxes_crop=boxes_crop, x=x.numpy(), out={k: v.numpy() for k, v in out.items()})

    def forward(self, images, K, im_ids, obj_ids, TCO, n_iterations=1, bsz_objects: Optional[int] = None):
        """Returns the per-iteration dicts (TCO_input, TCO_output, K_crop, boxes_*).  Hypotheses
        are processed in chunks of ``bsz_objects`` like ``forward_refiner`` does
        (MP/inference/pose_estimator.py:142-151)."""
        images = np.asarray(images, np.float32)
        K = np.asarray(K, np.float32)
        im_ids = np.asarray(im_ids, np.int32)
        obj_ids = np.asarray(obj_ids, np.int32)
        B = len(TCO)
        bsz = B if bsz_objects is None else bsz_objects
        outs = [dict(TCO_input=[], TCO_output=[], K_crop=[], boxes_rend=[], boxes_crop=[], pose=[]) for _ in range(n_iterations)]
        for s in range(0, B, bsz):
            sl = slice(s, min(B, s + bsz))
            T_in = np.asarray(TCO[sl], np.float32)
            for n in range(n_iterations):
                it = self._iteration(images, K[im_ids[sl]], im_ids[sl], obj_ids[sl], T_in, heads=("pose",))
                pose9 = it["out"]["pose"]
                if self.cfg["cosypose"]:
                    dR = G.compute_rotation_matrix_from_ortho6d(pose9[:, :6])
                    T_out = G.apply_imagespace_predictions(it["TCO"], it["K_crop"], pose9[:, 6:9], dR)
                else:
                    T_out = G.update_pose(it["TCO"], it["K_crop"], pose9, it["tCR"])
                o = outs[n]
                o["TCO_input"].append(it["TCO"]); o["TCO_output"].append(T_out); o["K_crop"].append(it["K_crop"])
                o["boxes_rend"].append(it["boxes_rend"]); o["boxes_crop"].append(it["boxes_crop"]); o["pose"].append(pose9)
                T_in = T_out
        return [{k: np.concatenate(v) for k, v in o.items()} for o in outs]

    def forward_coarse(self, images, K, im_ids, obj_ids, TCO):
        im_ids = np.asarray(im_ids, np.int32)
        it = self._iteration(np.asarray(images, np.float32), np.asarray(K, np.float32)[im_ids], im_ids,
                             np.asarray(obj_ids, np.int32), np.asarray(TCO, np.float32), heads=("renderings_logits",))
        logits = it["out"]["renderings_logits"]
        return dict(logits=logits, scores=1.0 / (1.0 + np.exp(-logits)), x=it["x"])
