"""CPU restatement of the detector's backbone: torchvision ResNet-50 + FPN + RPN head.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  The reference's detector is
``DetectorMaskRCNN`` (``MP/models/mask_rcnn.py:22-42``): torchvision's ``MaskRCNN`` over
``resnet_fpn_backbone("resnet50", pretrained=False)``.  torchvision (pinned 0.14.1, ``pyproject.toml:59-62``) is
absent from this image, so this file restates its published architecture with plain ``torch.nn.functional`` calls:

* ``models/resnet.py``: ``ResNet._forward_impl`` (conv1 7x7/2 - bn1 - relu - maxpool 3x3/2 - layer1..4) and
  ``Bottleneck.forward`` (1x1 - bn - relu - 3x3 with the stride - bn - relu - 1x1 - bn, + identity / (1x1 stride
  conv - bn) downsample, relu); BatchNorm2d in eval mode, eps 1e-5;
* ``ops/feature_pyramid_network.py``: ``FeaturePyramidNetwork.forward`` (lateral 1x1 convs, top-down
  ``F.interpolate(size=..., mode="nearest")`` + add, 3x3 output convs, ``LastLevelMaxPool`` =
  ``max_pool2d(x, 1, 2, 0)``), outputs named '0', '1', '2', '3', 'pool';
* ``models/detection/rpn.py``: ``RPNHead.forward`` (3x3 conv + ReLU, 1x1 ``cls_logits``, 1x1 ``bbox_pred``);
* ``models/detection/transform.py``: ``GeneralizedRCNNTransform.normalize``.

**Parity unpinned**: nothing in the reference's tests pins the detector and torchvision cannot run here.
"""

from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)


def param_shapes(n_anchors: int = 3) -> Dict[str, tuple]:
    """state-dict key -> shape of ``DetectorMaskRCNN``'s backbone + RPN head, in registration order."""
    s: Dict[str, tuple] = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)
        s[f"{p}.num_batches_tracked"] = ()

    b = "backbone.body."
    s[b + "conv1.weight"] = (64, 3, 7, 7)
    bn(b + "bn1", 64)
    inpl = 64
    for li, (pl, nb) in enumerate(zip(PLANES, LAYERS), start=1):
        for k in range(nb):
            p = f"{b}layer{li}.{k}"
            s[p + ".conv1.weight"] = (pl, inpl, 1, 1); bn(p + ".bn1", pl)
            s[p + ".conv2.weight"] = (pl, pl, 3, 3); bn(p + ".bn2", pl)
            s[p + ".conv3.weight"] = (4 * pl, pl, 1, 1); bn(p + ".bn3", 4 * pl)
            if k == 0:
                s[p + ".downsample.0.weight"] = (4 * pl, inpl, 1, 1); bn(p + ".downsample.1", 4 * pl)
            inpl = 4 * pl
    f = "backbone.fpn."
    for i, pl in enumerate(PLANES):
        s[f"{f}inner_blocks.{i}.0.weight"] = (256, 4 * pl, 1, 1)
        s[f"{f}inner_blocks.{i}.0.bias"] = (256,)
    for i in range(4):
        s[f"{f}layer_blocks.{i}.0.weight"] = (256, 256, 3, 3)
        s[f"{f}layer_blocks.{i}.0.bias"] = (256,)
    s["rpn.head.conv.0.0.weight"] = (256, 256, 3, 3)
    s["rpn.head.conv.0.0.bias"] = (256,)
    s["rpn.head.cls_logits.weight"] = (n_anchors, 256, 1, 1)
    s["rpn.head.cls_logits.bias"] = (n_anchors,)
    s["rpn.head.bbox_pred.weight"] = (4 * n_anchors, 256, 1, 1)
    s["rpn.head.bbox_pred.bias"] = (4 * n_anchors,)
    return s


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def normalize(images: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor(IMAGE_MEAN)[None, :, None, None]
    std = torch.tensor(IMAGE_STD)[None, :, None, None]
    return (images - mean) / std


def backbone_fpn_rpn(images: torch.Tensor, state_dict, normalized: bool = False) -> Dict[str, List[torch.Tensor]]:
    """images ``[b,3,h,w]`` in [0,1] (or already normalised / resized / padded by :func:`transform`) ->
    ``dict(features=[5 x [b,256,h_l,w_l]], objectness=[5 x [b,3,h_l,w_l]], deltas=[5 x [b,12,h_l,w_l]])``."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    b = "backbone.body."
    x = images if normalized else normalize(images)
    x = F.relu(_bn(F.conv2d(x, sd[b + "conv1.weight"], stride=2, padding=3), sd, b + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nb in enumerate(LAYERS, start=1):
        for k in range(nb):
            p = f"{b}layer{li}.{k}"
            stride = 2 if (k == 0 and li > 1) else 1
            idt = x
            o = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            o = F.relu(_bn(F.conv2d(o, sd[p + ".conv2.weight"], stride=stride, padding=1), sd, p + ".bn2"))
            o = _bn(F.conv2d(o, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            if k == 0:
                idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
            x = F.relu(o + idt)
        feats.append(x)
    f = "backbone.fpn."
    last = F.conv2d(feats[3], sd[f + "inner_blocks.3.0.weight"], sd[f + "inner_blocks.3.0.bias"])
    outs = [F.conv2d(last, sd[f + "layer_blocks.3.0.weight"], sd[f + "layer_blocks.3.0.bias"], padding=1)]
    for i in (2, 1, 0):
        lat = F.conv2d(feats[i], sd[f"{f}inner_blocks.{i}.0.weight"], sd[f"{f}inner_blocks.{i}.0.bias"])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        outs.insert(0, F.conv2d(last, sd[f"{f}layer_blocks.{i}.0.weight"], sd[f"{f}layer_blocks.{i}.0.bias"], padding=1))
    outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
    obj, dl = [], []
    for p in outs:
        t = F.relu(F.conv2d(p, sd["rpn.head.conv.0.0.weight"], sd["rpn.head.conv.0.0.bias"], padding=1))
        obj.append(F.conv2d(t, sd["rpn.head.cls_logits.weight"], sd["rpn.head.cls_logits.bias"]))
        dl.append(F.conv2d(t, sd["rpn.head.bbox_pred.weight"], sd["rpn.head.bbox_pred.bias"]))
    return dict(features=outs, objectness=obj, deltas=dl)


# ---------------------------------------------------------------------------------------------------------------
# The rest of MaskRCNN.forward in eval mode (torchvision 0.14.1): RPN proposal filtering, RoI heads, mask pasting.
# References: models/detection/anchor_utils.py (AnchorGenerator), _utils.py (BoxCoder), rpn.py
# (RegionProposalNetwork.forward / filter_proposals), ops/boxes.py (clip_boxes_to_image, remove_small_boxes, nms,
# batched_nms), ops/poolers.py (MultiScaleRoIAlign, LevelMapper), roi_heads.py (RoIHeads.forward,
# postprocess_detections, maskrcnn_inference, paste_masks_in_image), faster_rcnn.py (TwoMLPHead, FastRCNNPredictor),
# mask_rcnn.py (MaskRCNNHeads, MaskRCNNPredictor), transform.py (postprocess).  Defaults of MaskRCNN / FasterRCNN:
# rpn pre / post NMS top-n 1000 / 1000 (test), rpn_nms_thresh 0.7, rpn_score_thresh 0.0, box_score_thresh 0.05,
# box_nms_thresh 0.5, box_detections_per_img 100.  Parity unpinned (see the module docstring).
# ---------------------------------------------------------------------------------------------------------------
ANCHOR_SIZES = ((32,), (64,), (128,), (256,), (512,))  # MP/models/mask_rcnn.py:27
ASPECT_RATIOS = (0.5, 1.0, 2.0)
BBOX_XFORM_CLIP = float(np.log(1000.0 / 16))


def head_param_shapes(num_classes: int) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    s["roi_heads.box_head.fc6.weight"] = (1024, 256 * 7 * 7); s["roi_heads.box_head.fc6.bias"] = (1024,)
    s["roi_heads.box_head.fc7.weight"] = (1024, 1024); s["roi_heads.box_head.fc7.bias"] = (1024,)
    s["roi_heads.box_predictor.cls_score.weight"] = (num_classes, 1024); s["roi_heads.box_predictor.cls_score.bias"] = (num_classes,)
    s["roi_heads.box_predictor.bbox_pred.weight"] = (4 * num_classes, 1024); s["roi_heads.box_predictor.bbox_pred.bias"] = (4 * num_classes,)
    for i in range(4):
        s[f"roi_heads.mask_head.{i}.0.weight"] = (256, 256, 3, 3); s[f"roi_heads.mask_head.{i}.0.bias"] = (256,)
    s["roi_heads.mask_predictor.conv5_mask.weight"] = (256, 256, 2, 2); s["roi_heads.mask_predictor.conv5_mask.bias"] = (256,)
    s["roi_heads.mask_predictor.mask_fcn_logits.weight"] = (num_classes, 256, 1, 1)
    s["roi_heads.mask_predictor.mask_fcn_logits.bias"] = (num_classes,)
    return s


def base_anchors(size: float) -> np.ndarray:
    """``AnchorGenerator.generate_anchors`` for one scale: ``[3,4]`` (torch.round = half to even)."""
    r = torch.as_tensor(ASPECT_RATIOS, dtype=torch.float32)
    hr = torch.sqrt(r)
    wr = 1 / hr
    ws = (wr[:, None] * torch.as_tensor([size], dtype=torch.float32)[None, :]).view(-1)
    hs = (hr[:, None] * torch.as_tensor([size], dtype=torch.float32)[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round().numpy()


def level_anchors(image_size, grid_size, size) -> torch.Tensor:
    gh, gw = grid_size
    sh, sw = image_size[0] // gh, image_size[1] // gw
    sx = torch.arange(0, gw, dtype=torch.int32) * sw
    sy = torch.arange(0, gh, dtype=torch.int32) * sh
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1).float()
    return (shifts.view(-1, 1, 4) + torch.as_tensor(base_anchors(size)).view(1, -1, 4)).reshape(-1, 4)


def decode(rel: torch.Tensor, boxes: torch.Tensor, weights) -> torch.Tensor:
    """``BoxCoder.decode_single``: rel ``[n, 4k]``, boxes ``[n,4]`` -> ``[n, 4k]``."""
    wx, wy, ww, wh = weights
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * widths, boxes[:, 1] + 0.5 * heights
    dx, dy = rel[:, 0::4] / wx, rel[:, 1::4] / wy
    dw = torch.clamp(rel[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(rel[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx, pcy = dx * widths[:, None] + cx[:, None], dy * heights[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    hw_, hh_ = torch.tensor(0.5) * pw, torch.tensor(0.5) * ph
    return torch.stack((pcx - hw_, pcy - hh_, pcx + hw_, pcy + hh_), dim=2).flatten(1)


def clip_boxes(boxes: torch.Tensor, size) -> torch.Tensor:
    h, w = size
    b = boxes.clone()
    b[..., 0::2] = b[..., 0::2].clamp(min=0, max=w)
    b[..., 1::2] = b[..., 1::2].clamp(min=0, max=h)
    return b


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """``torchvision.ops.nms``: indices kept, by decreasing score."""
    order = np.argsort(-scores, kind="stable")
    b = boxes[order].astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    removed = np.zeros(len(b), bool)
    keep = []
    for i in range(len(b)):
        if removed[i]:
            continue
        keep.append(order[i])
        xx1, yy1 = np.maximum(b[i, 0], b[i + 1:, 0]), np.maximum(b[i, 1], b[i + 1:, 1])
        xx2, yy2 = np.minimum(b[i, 2], b[i + 1:, 2]), np.minimum(b[i, 3], b[i + 1:, 3])
        inter = np.maximum(xx2 - xx1, 0).astype(np.float32) * np.maximum(yy2 - yy1, 0).astype(np.float32)
        iou = inter / (area[i] + area[i + 1:] - inter)
        removed[i + 1:] |= iou > thr
    return np.asarray(keep, np.int64)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, thr: float) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    keep = []
    for c in torch.unique(idxs):
        sel = torch.where(idxs == c)[0]
        k = nms(boxes[sel].numpy(), scores[sel].numpy(), thr)
        keep.append(sel[torch.as_tensor(k)])
    keep = torch.cat(keep)
    return keep[scores[keep].sort(descending=True, stable=True)[1]]


def rpn_proposals(objectness: List[torch.Tensor], deltas: List[torch.Tensor], image_size, pre_nms_top_n=1000, post_nms_top_n=1000,
                  nms_thresh=0.7, min_size=1e-3, score_thresh=0.0, padded_size=None):
    """``RegionProposalNetwork.forward`` (eval): maps ``[b,3,h,w]`` / ``[b,12,h,w]``.  ``image_size`` = the (resized) image the
    boxes are clipped to; ``padded_size`` = the batched canvas the anchor strides come from (AnchorGenerator.forward uses
    ``image_list.tensors.shape[-2:]``); default: no padding."""
    padded_size = image_size if padded_size is None else padded_size
    out = []
    for b in range(objectness[0].shape[0]):
        boxes_l, scores_l, lvl_l = [], [], []
        for l, (o, d) in enumerate(zip(objectness, deltas)):
            A, gh, gw = o.shape[1], o.shape[2], o.shape[3]
            ob = o[b].permute(1, 2, 0).reshape(-1)                      # (h, w, a)
            dl = d[b].view(A, 4, gh, gw).permute(2, 3, 0, 1).reshape(-1, 4)
            k = min(pre_nms_top_n, ob.numel())
            top, idx = ob.topk(k)
            anchors = level_anchors(padded_size, (gh, gw), ANCHOR_SIZES[l][0])
            prop = decode(dl[idx], anchors[idx], (1.0, 1.0, 1.0, 1.0)).view(-1, 4)
            boxes_l.append(prop); scores_l.append(torch.sigmoid(top)); lvl_l.append(torch.full((k,), l, dtype=torch.int64))
        boxes, scores, lvl = torch.cat(boxes_l), torch.cat(scores_l), torch.cat(lvl_l)
        boxes = clip_boxes(boxes, image_size)
        keep = torch.where((boxes[:, 2] - boxes[:, 0] >= min_size) & (boxes[:, 3] - boxes[:, 1] >= min_size))[0]
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = torch.where(scores >= score_thresh)[0]
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = batched_nms(boxes, scores, lvl, nms_thresh)[:post_nms_top_n]
        out.append((boxes[keep], scores[keep]))
    return out


def map_levels(boxes: torch.Tensor, k_min=2, k_max=5) -> torch.Tensor:
    s = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(4 + torch.log2(s / 224) + torch.tensor(1e-6, dtype=s.dtype))
    return (torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - k_min)


def multiscale_roi_align(features: List[torch.Tensor], boxes_per_image: List[torch.Tensor], image_size, out_size: int, sampling_ratio=2):
    """``MultiScaleRoIAlign.forward`` over the first four pyramid levels; features NCHW."""
    from . import native

    feats = features[:4]
    scales = [2.0 ** float(torch.tensor(float(f.shape[-2]) / float(image_size[0])).log2().round()) for f in feats]
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes_per_image)])
    levels = map_levels(rois[:, 1:])
    out = torch.zeros((len(rois), feats[0].shape[1], out_size, out_size))
    for l, (f, sc) in enumerate(zip(feats, scales)):
        idx = torch.where(levels == l)[0]
        if len(idx) == 0:
            continue
        r = rois[idx]
        o = native.roi_align(f.numpy(), (r[:, 1:] * sc).numpy(), r[:, 0].numpy().astype(np.int32), (out_size, out_size), sampling_ratio)
        out[idx] = torch.as_tensor(o)
    return out, levels


def box_head(x: torch.Tensor, sd) -> tuple:
    x = x.flatten(1)
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc6.weight"], sd["roi_heads.box_head.fc6.bias"]))
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc7.weight"], sd["roi_heads.box_head.fc7.bias"]))
    return (F.linear(x, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"]),
            F.linear(x, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"]))


def mask_head(x: torch.Tensor, sd) -> torch.Tensor:
    for i in range(4):
        x = F.relu(F.conv2d(x, sd[f"roi_heads.mask_head.{i}.0.weight"], sd[f"roi_heads.mask_head.{i}.0.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, sd["roi_heads.mask_predictor.conv5_mask.weight"], sd["roi_heads.mask_predictor.conv5_mask.bias"], stride=2))
    return F.conv2d(x, sd["roi_heads.mask_predictor.mask_fcn_logits.weight"], sd["roi_heads.mask_predictor.mask_fcn_logits.bias"])


def postprocess_detections(class_logits, box_regression, proposals, image_size, score_thresh=0.05, nms_thresh=0.5, detections_per_img=100):
    """One image.  -> boxes ``[n,4]``, scores, labels."""
    num_classes = class_logits.shape[-1]
    boxes = decode(box_regression, proposals, (10.0, 10.0, 5.0, 5.0)).reshape(len(proposals), -1, 4)
    scores = F.softmax(class_logits, -1)
    boxes = clip_boxes(boxes, image_size)
    labels = torch.arange(num_classes).view(1, -1).expand_as(scores)
    boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
    inds = torch.where(scores > score_thresh)[0]
    boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
    keep = torch.where((boxes[:, 2] - boxes[:, 0] >= 1e-2) & (boxes[:, 3] - boxes[:, 1] >= 1e-2))[0]
    boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
    keep = batched_nms(boxes, scores, labels, nms_thresh)[:detections_per_img]
    return boxes[keep], scores[keep], labels[keep]


def paste_masks(mask_prob: torch.Tensor, boxes: torch.Tensor, image_size, padding: int = 1) -> torch.Tensor:
    """``paste_masks_in_image``: mask_prob ``[n,1,28,28]`` -> ``[n,1,H,W]``."""
    M = mask_prob.shape[-1]
    scale = float(M + 2 * padding) / M
    padded = F.pad(mask_prob, (padding,) * 4)
    w_half, h_half = (boxes[:, 2] - boxes[:, 0]) * 0.5 * scale, (boxes[:, 3] - boxes[:, 1]) * 0.5 * scale
    xc, yc = (boxes[:, 2] + boxes[:, 0]) * 0.5, (boxes[:, 3] + boxes[:, 1]) * 0.5
    be = torch.stack((xc - w_half, yc - h_half, xc + w_half, yc + h_half), 1).to(torch.int64)
    im_h, im_w = image_size
    res = []
    for m, box in zip(padded, be):
        w, h = max(int(box[2] - box[0] + 1), 1), max(int(box[3] - box[1] + 1), 1)
        mm = F.interpolate(m[None], size=(h, w), mode="bilinear", align_corners=False)[0][0]
        im = torch.zeros((im_h, im_w))
        x0, x1, y0, y1 = max(int(box[0]), 0), min(int(box[2]) + 1, im_w), max(int(box[1]), 0), min(int(box[3]) + 1, im_h)
        if y1 > y0 and x1 > x0:
            im[y0:y1, x0:x1] = mm[(y0 - int(box[1])):(y1 - int(box[1])), (x0 - int(box[0])):(x1 - int(box[0]))]
        res.append(im)
    return torch.stack(res)[:, None] if res else torch.zeros((0, 1, im_h, im_w))


def transform(images: torch.Tensor, min_size: int, max_size: int):
    """``GeneralizedRCNNTransform.forward`` for a batch of equally sized images: normalize, ``_resize_image_and_masks``
    (``F.interpolate(bilinear, align_corners=False, recompute_scale_factor=True)`` with the float32 scale torchvision
    forms), ``batch_images`` (zero padding to a multiple of 32) -> ``(canvas [b,3,hp,wp], resized (h, w))``."""
    h, w = images.shape[-2:]
    x = normalize(images)
    im_shape = torch.tensor([h, w])
    scale = torch.min(float(min_size) / torch.min(im_shape).to(torch.float32), float(max_size) / torch.max(im_shape).to(torch.float32))
    x = F.interpolate(x, scale_factor=scale.item(), mode="bilinear", recompute_scale_factor=True, align_corners=False)
    hr, wr = x.shape[-2:]
    hp, wp = (hr + 31) // 32 * 32, (wr + 31) // 32 * 32
    canvas = x.new_zeros((x.shape[0], 3, hp, wp))
    canvas[:, :, :hr, :wr] = x
    return canvas, (hr, wr)


def maskrcnn_forward(images: torch.Tensor, state_dict, min_size=None, max_size=None):
    """``DetectorMaskRCNN.forward`` in eval mode: list of ``dict(boxes, labels, scores, masks [n,1,H,W])`` per image (in the
    coordinates of the images as handed over) + the intermediates.  ``min_size`` / ``max_size``: the transform's resize
    (default: the image's own sides -> no resize, no padding: images that already have the network's input size)."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    orig_size = tuple(images.shape[-2:])
    if min_size is None and max_size is None:
        image_size = padded = orig_size
        dense = backbone_fpn_rpn(images, state_dict)
    else:
        canvas, image_size = transform(images, min(orig_size) if min_size is None else min_size,
                                       max(orig_size) if max_size is None else max_size)
        padded = tuple(canvas.shape[-2:])
        dense = backbone_fpn_rpn(canvas, state_dict, normalized=True)
    props = rpn_proposals(dense["objectness"], dense["deltas"], image_size, padded_size=padded)
    boxes_per_image = [p[0] for p in props]
    pooled, _ = multiscale_roi_align(dense["features"], boxes_per_image, image_size, 7)
    cls, reg = box_head(pooled, sd)
    results, start = [], 0
    for b, pb in enumerate(boxes_per_image):
        n = len(pb)
        bx, sc, lb = postprocess_detections(cls[start:start + n], reg[start:start + n], pb, image_size)
        start += n
        mp, _ = multiscale_roi_align(dense["features"], [torch.zeros((0, 4))] * b + [bx], image_size, 14) if len(bx) else (torch.zeros((0, 256, 14, 14)), None)
        if image_size != orig_size:  # GeneralizedRCNNTransform.postprocess -> resize_boxes
            rh = torch.tensor(orig_size[0], dtype=torch.float32) / torch.tensor(image_size[0], dtype=torch.float32)
            rw = torch.tensor(orig_size[1], dtype=torch.float32) / torch.tensor(image_size[1], dtype=torch.float32)
            bx = bx * torch.stack([rw, rh, rw, rh])
        if len(bx):
            ml = mask_head(mp, sd)
            prob = ml.sigmoid()[torch.arange(len(bx)), lb][:, None]
            masks = paste_masks(prob, bx, orig_size)
        else:
            masks = torch.zeros((0, 1, *orig_size))
        results.append(dict(boxes=bx, labels=lb, scores=sc, masks=masks))
    return results, dict(dense=dense, proposals=props, pooled=pooled, class_logits=cls, box_regression=reg)
