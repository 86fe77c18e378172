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


def backbone_fpn_rpn(images: torch.Tensor, state_dict) -> Dict[str, List[torch.Tensor]]:
    """images ``[b,3,h,w]`` in [0,1] -> ``dict(features=[5 x [b,256,h_l,w_l]], objectness=[5 x [b,3,h_l,w_l]],
    deltas=[5 x [b,12,h_l,w_l]])``."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    b = "backbone.body."
    x = normalize(images)
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
