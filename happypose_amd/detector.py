"""Detector: the stage in front of the render-and-compare path when ``run_detector=True``.

Reference: ``Detector`` (``MP/inference/detector.py:34-156``) over ``DetectorMaskRCNN``
(``MP/models/mask_rcnn.py:22-42`` = torchvision ``MaskRCNN`` on ``resnet_fpn_backbone("resnet50")``).

What runs on the GPU today: the image normalisation of ``GeneralizedRCNNTransform``, the ResNet-50 body, the feature
pyramid and the RPN head -- 53 + 8 + 15 convolutions on the library's MFMA conv kernels (``HP_ARCH_RESNET50_FPN``,
``csrc/net.cpp::build_graph_r50fpn``), i.e. everything of the detector that is dense arithmetic up to and including
the per-anchor objectness / box-delta maps.  What is NOT built yet: proposal decoding + NMS, the RoI heads (box and
mask branches) and mask pasting; :meth:`Detector.get_detections` therefore raises ``NotImplementedError`` after the
dense stage (it does not return made-up detections), and ``PoseEstimator.run_inference_pipeline(run_detector=True)``
keeps needing ``detections`` from the caller.  DESIGN.md section 8 tracks this.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from ._ffi import check, lib, ptr, stream_ptr

IMAGE_MEAN = (0.485, 0.456, 0.406)  # torchvision MaskRCNN defaults (models/detection/mask_rcnn.py)
IMAGE_STD = (0.229, 0.224, 0.225)
LEVELS = ("0", "1", "2", "3", "pool")


class DetectorBackbone:
    """ResNet-50 + FPN + RPN head on the device.  ``state_dict``: the reference checkpoint's keys
    (``backbone.body.*``, ``backbone.fpn.*``, ``rpn.head.*``); input size is fixed at construction (the reference
    resizes every image to ``input_resize``, ``MP/models/mask_rcnn.py:25,39-40``)."""

    def __init__(self, state_dict: Dict, input_size=(480, 640), max_batch: int = 1, device="cuda"):
        self.device = torch.device(device)
        self.h, self.w = input_size
        assert self.h % 32 == 0 and self.w % 32 == 0, "input size must be a multiple of 32 (GeneralizedRCNNTransform pads to that)"
        sd = {k: v for k, v in state_dict.items() if k.startswith(("backbone.", "rpn.head."))}
        self.net = ops.Net("resnet50-fpn", 3, sd, max_batch=max_batch, device=self.device, h=self.h, w=self.w)
        self.max_batch = max_batch

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> Dict[str, object]:
        """``images [b,3,h,w]`` fp32 in [0,1] -> ``dict(features={level: [b,256,h_l,w_l]}, objectness=[5 x [b,3,h_l,w_l]],
        deltas=[5 x [b,12,h_l,w_l]])`` (NCHW views of the NHWC maps, as torchvision's modules return them)."""
        b = images.shape[0]
        assert images.shape[1:] == (3, self.h, self.w) and images.dtype == torch.float32 and b <= self.max_batch
        images = images.to(self.device).contiguous()
        x = torch.empty((b, self.h, self.w, 4), dtype=torch.float32, device=self.device)
        mean = (C.c_float * 3)(*IMAGE_MEAN)
        std = (C.c_float * 3)(*IMAGE_STD)
        with torch.cuda.device(self.device):
            check(lib().hp_detector_preprocess(ptr(images), b, self.h, self.w, mean, std, ptr(x), stream_ptr(self.device)),
                  "hp_detector_preprocess")
        self.net.forward(x, want_pose=False)
        maps = self.net.feature_maps(b)
        nchw = [m.permute(0, 3, 1, 2) for m in maps]
        return dict(features={k: nchw[i] for i, k in enumerate(LEVELS)},
                    objectness=[m[:, :3] for m in nchw[5:10]], deltas=nchw[10:15])

    __call__ = forward


class Detector:
    """``MP/inference/detector.py:34-156``.  See the module docstring for what is implemented."""

    def __init__(self, backbone: DetectorBackbone, label_to_category_id: Optional[Dict[str, int]] = None):
        self.model = backbone
        self.category_id_to_label = {v: k for k, v in (label_to_category_id or {}).items()}

    @torch.no_grad()
    def get_detections(self, observation, detection_th: Optional[float] = None, output_masks: bool = False,
                       mask_th: float = 0.8, one_instance_per_class: bool = False):
        dense = self.model(observation.images[:, :3])
        assert all(torch.isfinite(o).all() for o in dense["objectness"])
        raise NotImplementedError(
            "Detector: backbone + FPN + RPN head run on the GPU; proposal decoding / NMS / RoI heads are not built yet -- "
            "pass `detections` to run_inference_pipeline")

    __call__ = get_detections
