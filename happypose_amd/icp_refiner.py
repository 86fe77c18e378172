"""Depth refinement after the refiner loop (``run_depth_refiner=True``).

Mirrors ``DepthRefiner`` (``MP/inference/depth_refiner.py:27-50``) and ``ICPRefiner``
(``MP/inference/icp_refiner.py:221-303``): render the depth of every prediction at full image
resolution, mask it against the measured depth, register the two point sets, replace the pose when
the registration is accepted.  The per-prediction Python / OpenCV loop of the reference is one
batched call of ``hp_icp_refine`` here (see ``csrc/icp.hip`` for what is and is not pinned).
"""

from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ._ffi import check, lib, ptr, stream_ptr
from .renderer import BatchRenderer, Panda3dLightData


class DepthRefiner(ABC):
    @abstractmethod
    def refine_poses(self, predictions, masks: Optional[torch.Tensor] = None, depth: Optional[torch.Tensor] = None,
                     K: Optional[torch.Tensor] = None) -> Tuple[object, Dict]:
        """``predictions``: N pose estimates indexing ``depth [B,H,W]``, ``masks [B,H,W]``, ``K [B,3,3]``
        through ``infos.batch_im_id``; returns ``(refined_predictions, extra_data)``."""


class ICPRefiner(DepthRefiner):
    """``ICPRefiner(mesh_db, renderer)`` as in the reference; ``n_iterations`` / ``tolerance`` /
    ``n_min_points`` default to its constants (100 ICP iterations there are split over 4 pyramid
    levels of OpenCV's implementation; the projective ICP converges in far fewer)."""

    def __init__(self, mesh_db, renderer: BatchRenderer, n_iterations: int = 30, tolerance: float = 0.05,
                 n_min_points: int = 1000, depth_delta_thresh: float = 0.1) -> None:
        self.mesh_db = mesh_db
        self.renderer = renderer
        self.light_datas = [Panda3dLightData("ambient")]
        self.n_iterations, self.tolerance = n_iterations, tolerance
        self.n_min_points, self.depth_delta_thresh = n_min_points, depth_delta_thresh

    def refine_poses(self, predictions, masks: Optional[torch.Tensor] = None, depth: Optional[torch.Tensor] = None,
                     K: Optional[torch.Tensor] = None):
        assert depth is not None
        assert K is not None
        dev = self.renderer.device
        refined = predictions.clone()
        N = len(predictions)
        if N == 0:
            return refined, {}
        depth = depth.to(dev, torch.float32)
        if depth.dim() == 4:
            depth = depth[:, 0]
        depth = depth.contiguous()
        B, H, W = depth.shape
        df = predictions.infos
        labels = df.label.tolist()
        im_ids_h = np.ascontiguousarray(df.batch_im_id.to_numpy(), dtype=np.int32)
        im_ids = torch.as_tensor(im_ids_h, device=dev)
        TCO = predictions.poses.to(dev, torch.float32).contiguous()
        K_ = K.to(dev, torch.float32)[im_ids.long()].contiguous()
        render = self.renderer.render(labels, TCO=TCO, K=K_, light_datas=[self.light_datas] * N, resolution=(H, W),
                                      render_depth=True)
        depth_rendered = render.depths.reshape(N, H, W).contiguous()
        m = None
        if masks is not None:
            m = masks.to(dev)
            if m.dim() == 4:
                m = m[:, 0]
            m = (m != 0).to(torch.uint8).contiguous()
            assert m.shape == (B, H, W)
        out = torch.empty_like(TCO)
        retval = torch.empty(N, dtype=torch.int32, device=dev)
        residual = torch.empty(N, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib().hp_icp_refine(N, B, H, W, ptr(depth_rendered), ptr(depth), ptr(m), ptr(im_ids),
                                      im_ids_h.ctypes.data_as(C.c_void_p), ptr(K_), ptr(TCO), self.n_iterations,
                                      self.n_min_points, self.tolerance, self.depth_delta_thresh, ptr(out), ptr(retval),
                                      ptr(residual), stream_ptr(dev)), "hp_icp_refine")
        # MP/inference/icp_refiner.py:297-300: poses_input = the poses before refinement
        refined.register_tensor("poses_input", predictions.poses.clone())
        refined.register_tensor("poses", out.to(predictions.poses.device))
        return refined, {"retval": retval, "residual": residual, "depth_rendered": depth_rendered}
