"""Detector: the stage in front of the render-and-compare path when ``run_detector=True``.

Mirrors ``Detector`` (``MP/inference/detector.py:34-156``) over ``DetectorMaskRCNN`` (``MP/models/mask_rcnn.py:22-42`` =
torchvision 0.14.1 ``MaskRCNN`` on ``resnet_fpn_backbone("resnet50")``, anchors ``((32,), (64,), (128,), (256,),
(512,))`` x ratios ``(0.5, 1, 2)``), inference only.  Every arithmetic stage runs in the HIP library:

  hp_detector_preprocess   GeneralizedRCNNTransform.normalize
  hp_net (RESNET50_FPN)    ResNet-50 body + FPN + RPN head (76 convolutions)
  hp_rpn_decode / hp_nms   anchors, box decoding, clipping, per-level NMS        (rpn.py: filter_proposals)
  hp_roi_align_levels      MultiScaleRoIAlign 7x7 / 14x14                        (ops/poolers.py)
  hp_net (CUSTOM)          box head (fc6 as a 7x7 conv, fc7, predictors), mask head (4 x 3x3, deconv as 4 x 1x1, logits)
  hp_box_postprocess       softmax + per-class decoding + clipping               (roi_heads.py: postprocess_detections)
  hp_paste_masks           maskrcnn_inference + paste_masks_in_image

torch is used for device memory and for INDEX bookkeeping between the stages (top-k selection, boolean filtering,
ordering by score) -- the counterpart of the pandas bookkeeping of the pose estimators.  Images must already have the
network's input size (``input_resize``; the reference's transform resizes to it, here that resize is the caller's).
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
import torch

from . import ops
from ._ffi import check, lib, ptr, stream_ptr
from .tensor_collection import PandasTensorCollection

IMAGE_MEAN = (0.485, 0.456, 0.406)  # torchvision MaskRCNN defaults (models/detection/mask_rcnn.py)
IMAGE_STD = (0.229, 0.224, 0.225)
LEVELS = ("0", "1", "2", "3", "pool")
ANCHOR_SIZES = (32, 64, 128, 256, 512)  # MP/models/mask_rcnn.py:27
ASPECT_RATIOS = (0.5, 1.0, 2.0)

_OLD_KEYS = (  # torchvision < 0.13 checkpoints (plain Conv2d instead of Conv2dNormActivation wrappers)
    ("backbone.fpn.inner_blocks.{i}.weight", "backbone.fpn.inner_blocks.{i}.0.weight"),
    ("backbone.fpn.inner_blocks.{i}.bias", "backbone.fpn.inner_blocks.{i}.0.bias"),
    ("backbone.fpn.layer_blocks.{i}.weight", "backbone.fpn.layer_blocks.{i}.0.weight"),
    ("backbone.fpn.layer_blocks.{i}.bias", "backbone.fpn.layer_blocks.{i}.0.bias"),
    ("roi_heads.mask_head.mask_fcn{j}.weight", "roi_heads.mask_head.{i}.0.weight"),
    ("roi_heads.mask_head.mask_fcn{j}.bias", "roi_heads.mask_head.{i}.0.bias"),
)


def modernise_keys(state_dict: Dict) -> Dict:
    """Older torchvision releases saved the FPN / RPN / mask-head convolutions without the ``Conv2dNormActivation``
    wrapper level (torchvision handles that in ``_load_from_state_dict``): map those names to the 0.14.1 ones."""
    sd = dict(state_dict)
    for old, new in _OLD_KEYS:
        for i in range(4):
            o, n = old.format(i=i, j=i + 1), new.format(i=i, j=i + 1)
            if o in sd and n not in sd:
                sd[n] = sd.pop(o)
    for t in ("weight", "bias"):
        if f"rpn.head.conv.{t}" in sd and f"rpn.head.conv.0.0.{t}" not in sd:
            sd[f"rpn.head.conv.0.0.{t}"] = sd.pop(f"rpn.head.conv.{t}")
    return sd


def base_anchors(size: float) -> np.ndarray:
    """``AnchorGenerator.generate_anchors`` (models/detection/anchor_utils.py) for one scale: ``[3,4]``, rounded half to even."""
    r = np.asarray(ASPECT_RATIOS, np.float32)
    hr = np.sqrt(r)
    wr = (np.float32(1) / hr).astype(np.float32)
    ws, hs = (wr * np.float32(size)).astype(np.float32), (hr * np.float32(size)).astype(np.float32)
    return np.round(np.stack([-ws, -hs, ws, hs], 1) / np.float32(2)).astype(np.float32)


def transform_sizes(image_size, min_size: Optional[int] = None, max_size: Optional[int] = None):
    """``GeneralizedRCNNTransform.resize`` + ``batch_images`` (torchvision models/detection/transform.py) for one image
    size: ``(resized (h, w), padded (h, w))``.  The scale is formed in float32 exactly as torchvision's tensor expression
    forms it (``min(min_size / min(h, w), max_size / max(h, w))``), the resized size is ``floor(size * scale)``
    (``F.interpolate(..., recompute_scale_factor=True)``), the canvas is that rounded up to a multiple of 32."""
    h, w = int(image_size[0]), int(image_size[1])
    min_size = min(h, w) if min_size is None else min_size
    max_size = max(h, w) if max_size is None else max_size
    # torchvision divides a Python float by a 0-dim float32 tensor: Tensor.__rtruediv__ = reciprocal(tensor) * scalar, two
    # float32 roundings (320 / 500 -> 0.64000005, not the correctly rounded 0.63999999: 300 x 500 becomes 192 x 320, not 191 x 319)
    rdiv = lambda num, den: np.float32(np.float32(1.0) / np.float32(den)) * np.float32(num)  # noqa: E731
    scale = float(min(rdiv(min_size, min(h, w)), rdiv(max_size, max(h, w))))
    hr, wr = int(np.floor(float(h) * scale)), int(np.floor(float(w) * scale))
    return (hr, wr), ((hr + 31) // 32 * 32, (wr + 31) // 32 * 32)


class DetectorBackbone:
    """ResNet-50 + FPN + RPN head on the device.  ``state_dict``: the reference checkpoint's keys
    (``backbone.body.*``, ``backbone.fpn.*``, ``rpn.head.*``).  The image size is fixed at construction (the plan is built for
    it); ``min_size`` / ``max_size`` are the reference's ``input_resize`` (``MP/models/mask_rcnn.py:25,39-40``: images are
    resized so that the short side is ``min(input_resize)``, capped by ``max(input_resize)`` on the long side)."""

    def __init__(self, state_dict: Dict, input_size=(480, 640), max_batch: int = 1, device="cuda", min_size: Optional[int] = None,
                 max_size: Optional[int] = None):
        self.device = torch.device(device)
        self.in_h, self.in_w = int(input_size[0]), int(input_size[1])              # images as handed over
        (self.rh, self.rw), (self.h, self.w) = transform_sizes(input_size, min_size, max_size)  # resized; padded canvas = the plan
        sd = {k: v for k, v in modernise_keys(state_dict).items() if k.startswith(("backbone.", "rpn.head."))}
        self.net = ops.Net("resnet50-fpn", 3, sd, max_batch=max_batch, device=self.device, h=self.h, w=self.w)
        self.max_batch = max_batch

    @torch.no_grad()
    def forward_nhwc(self, images: torch.Tensor) -> List[torch.Tensor]:
        """``images [b,3,H,W]`` fp32 in [0,1] -> the 15 NHWC maps (5 pyramid levels, 5 objectness ``[..,4]``, 5 deltas ``[..,12]``)
        of the resized, normalised, zero-padded images."""
        b = images.shape[0]
        assert images.shape[1:] == (3, self.in_h, self.in_w) and images.dtype == torch.float32 and b <= self.max_batch
        images = images.to(self.device).contiguous()
        x = torch.empty((b, self.h, self.w, 4), dtype=torch.float32, device=self.device)
        mean = (C.c_float * 3)(*IMAGE_MEAN)
        std = (C.c_float * 3)(*IMAGE_STD)
        with torch.cuda.device(self.device):
            if (self.in_h, self.in_w) == (self.h, self.w):
                check(lib().hp_detector_preprocess(ptr(images), b, self.h, self.w, mean, std, ptr(x), stream_ptr(self.device)),
                      "hp_detector_preprocess")
            else:
                check(lib().hp_detector_preprocess_resize(ptr(images), b, self.in_h, self.in_w, self.rh, self.rw, self.h, self.w,
                                                          mean, std, ptr(x), stream_ptr(self.device)),
                      "hp_detector_preprocess_resize")
        self.net.forward(x, want_pose=False)
        return self.net.feature_maps(b)

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> Dict[str, object]:
        """-> ``dict(features={level: [b,256,h_l,w_l]}, objectness=[5 x [b,3,h_l,w_l]], deltas=[5 x [b,12,h_l,w_l]])``
        (NCHW views of the NHWC maps, as torchvision's modules return them)."""
        nchw = [m.permute(0, 3, 1, 2) for m in self.forward_nhwc(images)]
        return dict(features={k: nchw[i] for i, k in enumerate(LEVELS)},
                    objectness=[m[:, :3] for m in nchw[5:10]], deltas=nchw[10:15])

    __call__ = forward


class MaskRCNN:
    """``DetectorMaskRCNN`` in eval mode on the device: ``forward(images) -> [dict(boxes, labels, scores, masks)]`` like
    torchvision's ``MaskRCNN.forward`` (masks ``[n,1,H,W]`` probabilities)."""

    def __init__(self, state_dict: Dict, num_classes: int, input_size=(480, 640), max_batch: int = 1, device="cuda",
                 min_size: Optional[int] = None, max_size: Optional[int] = None, rpn_pre_nms_top_n: int = 1000, rpn_post_nms_top_n: int = 1000, rpn_nms_thresh: float = 0.7,
                 box_score_thresh: float = 0.05, box_nms_thresh: float = 0.5, box_detections_per_img: int = 100):
        sd = modernise_keys(state_dict)
        self.device = torch.device(device)
        self.num_classes = num_classes
        self.orig_size = (int(input_size[0]), int(input_size[1]))      # images as handed over; boxes / masks are returned in it
        self.backbone = DetectorBackbone(sd, input_size, max_batch, device, min_size, max_size)
        self.size = (self.backbone.rh, self.backbone.rw)               # the resized image: what boxes are clipped to
        self.padded = (self.backbone.h, self.backbone.w)               # the canvas the network runs on
        self.cfg = dict(pre=rpn_pre_nms_top_n, post=rpn_post_nms_top_n, rpn_nms=rpn_nms_thresh, score=box_score_thresh,
                        nms=box_nms_thresh, dets=box_detections_per_img)
        c4 = (num_classes + 3) // 4 * 4
        self._c4 = c4
        box_layers = [
            dict(weight="fc6.w", bias="roi_heads.box_head.fc6.bias", cin=256, cout=1024, k=7, H=7, W=7, src=-1, dst=0, relu=True),
            dict(weight="roi_heads.box_head.fc7.weight", bias="roi_heads.box_head.fc7.bias", cin=1024, cout=1024, k=1, H=1, W=1, src=0, dst=1, relu=True),
            dict(weight="roi_heads.box_predictor.cls_score.weight", bias="roi_heads.box_predictor.cls_score.bias", cin=1024,
                 cout=num_classes, k=1, H=1, W=1, src=1, dst=2),
            dict(weight="roi_heads.box_predictor.bbox_pred.weight", bias="roi_heads.box_predictor.bbox_pred.bias", cin=1024,
                 cout=4 * num_classes, k=1, H=1, W=1, src=1, dst=3),
        ]
        bsd = {k: np.asarray(v) for k, v in sd.items() if k.startswith(("roi_heads.box_head.", "roi_heads.box_predictor."))}
        # fc6 acts on x.flatten(1) of [n,256,7,7]: its weight [1024, 256*7*7] IS a 7x7 "valid" convolution [1024,256,7,7]
        bsd["fc6.w"] = bsd.pop("roi_heads.box_head.fc6.weight").reshape(1024, 256, 7, 7)
        self.box_net = ops.GraphNet(256, 7, 7, box_layers, [(2, 1, 1, num_classes), (3, 1, 1, 4 * num_classes)], bsd, max_batch=256,
                                    device=device)
        mask_layers = [dict(weight=f"roi_heads.mask_head.{i}.0.weight", bias=f"roi_heads.mask_head.{i}.0.bias", cin=256, cout=256, k=3,
                            pad=1, H=14, W=14, src=(-1 if i == 0 else (i - 1) % 2), dst=i % 2, relu=True) for i in range(4)]
        # ConvTranspose2d(256, 256, 2, 2): out[2i+a][2j+b][co] = sum_ci in[i][j][ci] W[ci][co][a][b] + bias[co] = four 1x1
        # convolutions whose outputs interleave; kept as ONE 1x1 convolution to 4 x 256 channels ordered (a, b, co): the
        # tensor [n,14,14,(a,b,co)] read as [n,14,14*4,256] is what the pointwise logits convolution consumes
        wt = np.asarray(sd["roi_heads.mask_predictor.conv5_mask.weight"])  # [ci][co][a][b]
        msd = {k: np.asarray(v) for k, v in sd.items() if k.startswith("roi_heads.mask_head.")}
        msd["deconv.w"] = np.ascontiguousarray(wt.transpose(2, 3, 1, 0).reshape(1024, 256, 1, 1))
        msd["deconv.b"] = np.tile(np.asarray(sd["roi_heads.mask_predictor.conv5_mask.bias"]), 4)
        msd["logits.w"] = np.asarray(sd["roi_heads.mask_predictor.mask_fcn_logits.weight"])
        msd["logits.b"] = np.asarray(sd["roi_heads.mask_predictor.mask_fcn_logits.bias"])
        mask_layers += [dict(weight="deconv.w", bias="deconv.b", cin=256, cout=1024, k=1, H=14, W=14, src=1, dst=2, relu=True),
                        dict(weight="logits.w", bias="logits.b", cin=256, cout=num_classes, k=1, H=14, W=56, src=2, dst=3)]
        self.mask_net = ops.GraphNet(256, 14, 14, mask_layers, [(3, 14, 56, num_classes)], msd, max_batch=128, device=device)
        self._base = [base_anchors(s) for s in ANCHOR_SIZES]

    # ---- helpers over the C ABI -------------------------------------------------------------------------------
    def _nms(self, boxes: torch.Tensor, group: torch.Tensor, thr: float) -> torch.Tensor:
        """``boxes`` sorted by decreasing score -> bool keep mask (``hp_nms``; synchronises)."""
        n = boxes.shape[0]
        keep = np.zeros(n, np.uint8)
        if n:
            boxes, group = boxes.contiguous(), group.to(torch.int32).contiguous()
            with torch.cuda.device(self.device):
                check(lib().hp_nms(ptr(boxes), ptr(group), n, C.c_float(thr), keep.ctypes.data_as(C.c_void_p), stream_ptr(self.device)), "hp_nms")
        return torch.as_tensor(keep.astype(bool), device=self.device)

    def _roi_align(self, maps: List[torch.Tensor], rois: torch.Tensor, out_size: int) -> torch.Tensor:
        K = rois.shape[0]
        out = torch.empty((K, out_size, out_size, 256), dtype=torch.float32, device=self.device)
        if K == 0:
            return out
        L = 4
        ptrs = (C.c_void_p * L)(*[m.data_ptr() for m in maps[:L]])
        hs = (C.c_int * L)(*[m.shape[1] for m in maps[:L]])
        ws = (C.c_int * L)(*[m.shape[2] for m in maps[:L]])
        scales = (C.c_float * L)(*[2.0 ** round(float(np.log2(m.shape[1] / self.size[0]))) for m in maps[:L]])
        rois = rois.contiguous()
        with torch.cuda.device(self.device):
            check(lib().hp_roi_align_levels(ptrs, hs, ws, scales, L, 2, 256, ptr(rois), K, out_size, 2, ptr(out), None,
                                            stream_ptr(self.device)), "hp_roi_align_levels")
        return out

    # ---- RegionProposalNetwork.forward (eval) -----------------------------------------------------------------
    def _proposals(self, maps: List[torch.Tensor], b: int):
        H, W = self.size
        dev = self.device
        boxes_l, scores_l, valid_l, lvl_l = [], [], [], []
        for l in range(5):
            obj_map, delta_map = maps[5 + l][b], maps[10 + l][b]            # [h,w,4], [h,w,12]
            gh, gw = obj_map.shape[0], obj_map.shape[1]
            ob = obj_map[..., :3].reshape(-1)                               # (y, x, a): torchvision's flattening order
            k = min(self.cfg["pre"], ob.numel())
            top, idx = ob.topk(k)
            idx32 = idx.to(torch.int32).contiguous()
            top = top.contiguous()
            boxes = torch.empty((k, 4), dtype=torch.float32, device=dev)
            scores = torch.empty(k, dtype=torch.float32, device=dev)
            valid = torch.empty(k, dtype=torch.uint8, device=dev)
            base = (C.c_float * 12)(*self._base[l].reshape(-1).tolist())
            delta_map = delta_map.contiguous()
            with torch.cuda.device(dev):
                check(lib().hp_rpn_decode(ptr(top), ptr(idx32), k, ptr(delta_map), gw, 3, base, self.padded[0] // gh, self.padded[1] // gw,
                                          C.c_float(H), C.c_float(W), C.c_float(1e-3), ptr(boxes), ptr(scores), ptr(valid),
                                          stream_ptr(dev)), "hp_rpn_decode")
            boxes_l.append(boxes); scores_l.append(scores); valid_l.append(valid.bool()); lvl_l.append(torch.full((k,), l, device=dev, dtype=torch.int32))
        boxes, scores, valid, lvl = torch.cat(boxes_l), torch.cat(scores_l), torch.cat(valid_l), torch.cat(lvl_l)
        boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]         # remove_small_boxes; score_thresh 0.0 keeps all
        order = scores.argsort(descending=True, stable=True)
        boxes, scores, lvl = boxes[order], scores[order], lvl[order]
        keep = self._nms(boxes, lvl, self.cfg["rpn_nms"])
        return boxes[keep][: self.cfg["post"]], scores[keep][: self.cfg["post"]]

    # ---- MaskRCNN.forward (eval) ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images: torch.Tensor, return_intermediates: bool = False):
        H, W = self.size
        dev = self.device
        maps = self.backbone.forward_nhwc(images)
        results, inter = [], []
        for b in range(images.shape[0]):
            props, pscores = self._proposals(maps, b)
            n = props.shape[0]
            rois = torch.cat([torch.full((n, 1), float(b), device=dev), props], 1)
            pooled = self._roi_align(maps, rois, 7)                                      # [n,7,7,256]
            cls, reg = self.box_net.run(pooled) if n else (torch.zeros((0, 1, 1, self._c4), device=dev), torch.zeros((0, 1, 1, 4 * self.num_classes), device=dev))
            cls, reg = cls.reshape(n, -1), reg.reshape(n, -1)
            scores = torch.empty((n, self.num_classes), dtype=torch.float32, device=dev)
            boxes = torch.empty((n, self.num_classes, 4), dtype=torch.float32, device=dev)
            if n:
                cls, reg, props = cls.contiguous(), reg.contiguous(), props.contiguous()
                with torch.cuda.device(dev):
                    check(lib().hp_box_postprocess(ptr(cls), cls.shape[1], ptr(reg), reg.shape[1], ptr(props),
                                                   n, self.num_classes, C.c_float(H), C.c_float(W), ptr(scores), ptr(boxes),
                                                   stream_ptr(dev)), "hp_box_postprocess")
            labels = torch.arange(self.num_classes, device=dev).view(1, -1).expand(n, -1)
            bx, sc, lb = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
            sel = sc > self.cfg["score"]
            bx, sc, lb = bx[sel], sc[sel], lb[sel]
            sel = (bx[:, 2] - bx[:, 0] >= 1e-2) & (bx[:, 3] - bx[:, 1] >= 1e-2)
            bx, sc, lb = bx[sel], sc[sel], lb[sel]
            order = sc.argsort(descending=True, stable=True)
            bx, sc, lb = bx[order], sc[order], lb[order]
            keep = self._nms(bx, lb, self.cfg["nms"])
            bx, sc, lb = bx[keep][: self.cfg["dets"]], sc[keep][: self.cfg["dets"]], lb[keep][: self.cfg["dets"]]
            nd = bx.shape[0]
            OH, OW = self.orig_size
            masks = torch.zeros((nd, 1, OH, OW), dtype=torch.float32, device=dev)
            ml = None
            if nd:
                mrois = torch.cat([torch.full((nd, 1), float(b), device=dev), bx], 1)
                mp = self._roi_align(maps, mrois, 14)                                    # [nd,14,14,256]
                ml = self.mask_net.run(mp)[0]                                            # [nd,14,56,c4] = (i, (j, a, b), class)
            if (OH, OW) != (H, W):  # GeneralizedRCNNTransform.postprocess: resize_boxes back to the image as handed over
                rh = torch.tensor(float(OH), dtype=torch.float32) / torch.tensor(float(H), dtype=torch.float32)
                rw = torch.tensor(float(OW), dtype=torch.float32) / torch.tensor(float(W), dtype=torch.float32)
                bx = bx * torch.stack([rw, rh, rw, rh]).to(dev)
            if nd:
                lab32, ml, bx = lb.to(torch.int32).contiguous(), ml.contiguous(), bx.contiguous()
                with torch.cuda.device(dev):
                    check(lib().hp_paste_masks(ptr(ml), ml.shape[-1], ptr(lab32), ptr(bx), nd, OH, OW, ptr(masks),
                                               stream_ptr(dev)), "hp_paste_masks")
            results.append(dict(boxes=bx, labels=lb, scores=sc, masks=masks))
            inter.append(dict(proposals=props, proposal_scores=pscores, pooled=pooled, class_logits=cls[:, : self.num_classes],
                              box_regression=reg[:, : 4 * self.num_classes]))
        return (results, inter) if return_intermediates else results

    __call__ = forward


class Detector:
    """``MP/inference/detector.py:34-156``: ``get_detections(observation, detection_th, output_masks, mask_th,
    one_instance_per_class) -> PandasTensorCollection(infos[batch_im_id, label, score, instance_id], bboxes [, masks])``."""

    def __init__(self, model: MaskRCNN, label_to_category_id: Dict[str, int]):
        self.model = model
        self.category_id_to_label = {v: k for k, v in label_to_category_id.items()}

    @torch.no_grad()
    def get_detections(self, observation, detection_th: Optional[float] = None, output_masks: bool = False,
                       mask_th: float = 0.8, one_instance_per_class: bool = False):
        from .pose_estimator import add_instance_id, filter_detections

        images = observation.images[:, :3]
        outputs = self.model(images)
        dev = self.model.device
        infos, bboxes, masks = [], [], []
        for n, out in enumerate(outputs):
            labels = [self.category_id_to_label[int(c)] for c in out["labels"].tolist()]
            scores = out["scores"].tolist()
            for i in range(len(labels)):
                infos.append(dict(batch_im_id=n, label=labels[i], score=scores[i]))
            bboxes.append(out["boxes"])
            masks.append(out["masks"][:, 0] > mask_th)
        if len(infos) > 0:
            bboxes_t, masks_t = torch.cat(bboxes).to(dev).float(), torch.cat(masks).to(dev)
            infos_df = pd.DataFrame(infos)
        else:
            infos_df = pd.DataFrame(dict(score=[], label=[], batch_im_id=[]))
            bboxes_t = torch.empty(0, 4, device=dev).float()
            masks_t = torch.empty(0, images.shape[2], images.shape[3], dtype=torch.bool, device=dev)
        detections = PandasTensorCollection(infos=infos_df, bboxes=bboxes_t)
        if output_masks:
            detections.register_tensor("masks", masks_t)
        if detection_th is not None:
            keep = np.where(detections.infos["score"] > detection_th)[0]
            detections = detections[keep]
        if one_instance_per_class:
            detections = filter_detections(detections, one_instance_per_class=True)
        return add_instance_id(detections)

    __call__ = get_detections


def maskrcnn_param_shapes(num_classes: int, n_anchors: int = 3) -> Dict[str, tuple]:
    """state-dict key -> shape of ``DetectorMaskRCNN`` (``MP/models/mask_rcnn.py:22-42`` = torchvision 0.14.1
    ``MaskRCNN`` on a ResNet-50 FPN backbone), in registration order: body (7x7 stem, bottlenecks 3-4-6-3 with the
    downsample on the first block of a stage), FPN inner 1x1 / layer 3x3 blocks, RPN head, box head / predictor, mask head /
    predictor.  What a checkpoint of the reference holds; used to make random-weight detectors for benchmarks."""
    s: Dict[str, tuple] = {}

    def bn(prefix, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}.{leaf}"] = (c,)
        s[f"{prefix}.num_batches_tracked"] = ()

    s["backbone.body.conv1.weight"] = (64, 3, 7, 7)
    bn("backbone.body.bn1", 64)
    c_in = 64
    for stage, (width, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for blk in range(blocks):
            q = f"backbone.body.layer{stage}.{blk}"
            for j, (co, ci, k) in enumerate(((width, c_in, 1), (width, width, 3), (4 * width, width, 1)), start=1):
                s[f"{q}.conv{j}.weight"] = (co, ci, k, k)
                bn(f"{q}.bn{j}", co)
            if blk == 0:
                s[f"{q}.downsample.0.weight"] = (4 * width, c_in, 1, 1)
                bn(f"{q}.downsample.1", 4 * width)
            c_in = 4 * width
    for i, width in enumerate((64, 128, 256, 512)):
        s[f"backbone.fpn.inner_blocks.{i}.0.weight"] = (256, 4 * width, 1, 1)
        s[f"backbone.fpn.inner_blocks.{i}.0.bias"] = (256,)
    for i in range(4):
        s[f"backbone.fpn.layer_blocks.{i}.0.weight"] = (256, 256, 3, 3)
        s[f"backbone.fpn.layer_blocks.{i}.0.bias"] = (256,)
    for name, shape in (("rpn.head.conv.0.0", (256, 256, 3, 3)), ("rpn.head.cls_logits", (n_anchors, 256, 1, 1)),
                        ("rpn.head.bbox_pred", (4 * n_anchors, 256, 1, 1)),
                        ("roi_heads.box_head.fc6", (1024, 256 * 7 * 7)), ("roi_heads.box_head.fc7", (1024, 1024)),
                        ("roi_heads.box_predictor.cls_score", (num_classes, 1024)),
                        ("roi_heads.box_predictor.bbox_pred", (4 * num_classes, 1024)),
                        *[(f"roi_heads.mask_head.{i}.0", (256, 256, 3, 3)) for i in range(4)],
                        ("roi_heads.mask_predictor.conv5_mask", (256, 256, 2, 2)),
                        ("roi_heads.mask_predictor.mask_fcn_logits", (num_classes, 256, 1, 1))):
        s[name + ".weight"] = shape
        s[name + ".bias"] = (shape[0],)
    return s


def synthetic_maskrcnn(device, n_classes: int, seed: int = 3, input_size=(480, 640)) -> MaskRCNN:
    """A ``MaskRCNN`` on name-keyed random weights (there are no checkpoints offline): the backbone + FPN + RPN time does
    not depend on the weights, the head time does through the number of proposals that survive, so the box / objectness
    layers are scaled to keep a realistic few hundred proposals alive.  Benchmarks only."""
    from .synthetic import named_weights

    w = named_weights(maskrcnn_param_shapes(n_classes), seed=seed)
    for k, f in (("rpn.head.bbox_pred.weight", 0.02), ("rpn.head.bbox_pred.bias", 0.5), ("roi_heads.box_predictor.bbox_pred.weight", 0.05),
                 ("roi_heads.box_predictor.cls_score.weight", 0.3), ("rpn.head.cls_logits.weight", 2.0)):
        w[k] = (w[k] * f).astype(np.float32)
    return MaskRCNN(w, n_classes, input_size=input_size, max_batch=1, device=device)
