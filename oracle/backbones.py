"""torch-CPU fp32 restatement of the reference backbones and heads, executed as the
reference does: the UNFUSED layer sequence conv -> BN(eval) -> ReLU ... in NCHW.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Pinned by golden G6
(``tests/golden/g6_backbones.npz``: outputs of the reference's own modules
``MP/models/torchvision_resnet.py::resnet34`` and ``MP|CP/models/wide_resnet.py::
WideResNet34/18`` on name-keyed random weights).
"""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}
_PLANES = [64, 128, 256, 512]


def _bn_shapes(prefix: str, c: int) -> Dict[str, Tuple[int, ...]]:
    return {
        f"{prefix}.weight": (c,), f"{prefix}.bias": (c,), f"{prefix}.running_mean": (c,),
        f"{prefix}.running_var": (c,), f"{prefix}.num_batches_tracked": (),
    }


def param_shapes(arch: str, n_inputs: int) -> Dict[str, Tuple[int, ...]]:
    """state-dict key -> shape, in the reference's registration order (SURVEY.md
    Appendix C).  ``arch``: ``vanilla_resnet34`` | ``resnet34`` | ``resnet18``
    (the ``backbone_str`` values of MP/training/pose_models_cfg.py:106-122)."""
    s: Dict[str, Tuple[int, ...]] = {}
    if arch == "vanilla_resnet34":
        s["conv1.weight"] = (64, n_inputs, 7, 7)
        s.update(_bn_shapes("bn1", 64))
        inpl = 64
        for li, (planes, nb) in enumerate(zip(_PLANES, _LAYERS[34]), start=1):
            for b in range(nb):
                stride = 2 if (b == 0 and li > 1) else 1
                p = f"layer{li}.{b}"
                s[f"{p}.conv1.weight"] = (planes, inpl, 3, 3)
                s.update(_bn_shapes(f"{p}.bn1", planes))
                s[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
                s.update(_bn_shapes(f"{p}.bn2", planes))
                if stride != 1 or inpl != planes:
                    s[f"{p}.downsample.0.weight"] = (planes, inpl, 1, 1)
                    s.update(_bn_shapes(f"{p}.downsample.1", planes))
                inpl = planes
        s["fc.weight"] = (512, 512)
        s["fc.bias"] = (512,)
    elif arch in ("resnet34", "resnet18"):
        depth = 34 if arch == "resnet34" else 18
        s["conv1.weight"] = (64, n_inputs, 5, 5)
        s.update(_bn_shapes("bn1", 64))
        inpl = 64
        for li, (planes, nb) in enumerate(zip(_PLANES, _LAYERS[depth]), start=1):
            for b in range(nb):
                stride = 2 if (b == 0 and li > 1) else 1
                p = f"layer{li}.{b}"
                s.update(_bn_shapes(f"{p}.bn1", inpl))
                s[f"{p}.conv1.weight"] = (planes, inpl, 3, 3)
                s.update(_bn_shapes(f"{p}.bn2", planes))
                s[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
                if stride != 1 or inpl != planes:
                    s[f"{p}.downsample.weight"] = (planes, inpl, 1, 1)
                inpl = planes
    else:
        raise ValueError(arch)
    return s


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))


def _bn(x, sd, p):
    return F.batch_norm(x, _t(sd, f"{p}.running_mean"), _t(sd, f"{p}.running_var"),
                        _t(sd, f"{p}.weight"), _t(sd, f"{p}.bias"), training=False, eps=1e-5)


def resnet34_forward(x: torch.Tensor, sd, prefix: str = "") -> torch.Tensor:
    """MP/models/torchvision_resnet.py:325-341 (``_forward_impl``) with BasicBlock
    :110-126; returns ``[B,512]`` (after avgpool + fc)."""
    g = lambda k: _t(sd, prefix + k)  # noqa: E731
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    x = F.conv2d(x, g("conv1.weight"), stride=2, padding=3)
    x = F.relu(_bn(x, sdp, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    inpl = 64
    for li, (planes, nb) in enumerate(zip(_PLANES, _LAYERS[34]), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            identity = x
            out = F.conv2d(x, _t(sdp, f"{p}.conv1.weight"), stride=stride, padding=1)
            out = F.relu(_bn(out, sdp, f"{p}.bn1"))
            out = F.conv2d(out, _t(sdp, f"{p}.conv2.weight"), stride=1, padding=1)
            out = _bn(out, sdp, f"{p}.bn2")
            if stride != 1 or inpl != planes:
                identity = _bn(F.conv2d(x, _t(sdp, f"{p}.downsample.0.weight"), stride=stride),
                               sdp, f"{p}.downsample.1")
            x = F.relu(out + identity)
            inpl = planes
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, _t(sdp, "fc.weight"), _t(sdp, "fc.bias"))


def wide_resnet_forward(x: torch.Tensor, sd, depth: int = 34, prefix: str = "") -> torch.Tensor:
    """MP/models/wide_resnet.py:120-129 with ``BasicBlockV2`` :59-65 (pre-activation;
    bare 1x1 downsample on the activated input); returns ``[B,512,h/32,w/32]``."""
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    x = F.conv2d(x, _t(sdp, "conv1.weight"), stride=2, padding=2)
    x = F.relu(_bn(x, sdp, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    inpl = 64
    for li, (planes, nb) in enumerate(zip(_PLANES, _LAYERS[depth]), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            out = F.relu(_bn(x, sdp, f"{p}.bn1"))
            if stride != 1 or inpl != planes:
                residual = F.conv2d(out, _t(sdp, f"{p}.downsample.weight"), stride=stride)
            else:
                residual = x
            out = F.conv2d(out, _t(sdp, f"{p}.conv1.weight"), stride=stride, padding=1)
            out = F.relu(_bn(out, sdp, f"{p}.bn2"))
            out = F.conv2d(out, _t(sdp, f"{p}.conv2.weight"), stride=1, padding=1)
            x = out + residual
            inpl = planes
    return x


# --------------------------------------------------------------------------- EfficientNet-b3
# CP/models/efficientnet.py:20-331 + efficientnet_utils.py (the backbone of the released
# CosyPose checkpoints, selected at CP/training/pose_models_cfg.py:33-35).  Pinned by golden G9
# (tests/golden/g9_efficientnet.npz: output of the reference module on name-keyed weights).
_EFF_BASE = [  # repeats, kernel, stride, expand, in, out  (efficientnet_utils.py:345-353; se_ratio 0.25)
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
_EFF_B3 = dict(width=1.2, depth=1.4, image_size=300, bn_eps=1e-3, divisor=8)  # :248, :357-367


def _eff_round_filters(f, width=1.2, divisor=8):
    f = f * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def efficientnet_b3_blocks():
    """[(kernel, stride, expand, in, out, se_channels)] of the 26 MBConv blocks."""
    import math

    blocks = []
    for r, k, s, e, i, o in _EFF_BASE:
        i, o = _eff_round_filters(i), _eff_round_filters(o)
        for j in range(int(math.ceil(_EFF_B3["depth"] * r))):
            cin, stride = (i, s) if j == 0 else (o, 1)
            blocks.append((k, stride, e, cin, o, max(1, int(cin * 0.25))))
    return blocks


def efficientnet_b3_param_shapes(n_inputs: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    stem = _eff_round_filters(32)
    s["_conv_stem.weight"] = (stem, n_inputs, 3, 3)
    s.update(_bn_shapes("_bn0", stem))
    for bi, (k, _st, e, cin, cout, cse) in enumerate(efficientnet_b3_blocks()):
        p, mid = f"_blocks.{bi}", cin * e
        if e != 1:
            s[f"{p}._expand_conv.weight"] = (mid, cin, 1, 1)
            s.update(_bn_shapes(f"{p}._bn0", mid))
        s[f"{p}._depthwise_conv.weight"] = (mid, 1, k, k)
        s.update(_bn_shapes(f"{p}._bn1", mid))
        s[f"{p}._se_reduce.weight"] = (cse, mid, 1, 1)
        s[f"{p}._se_reduce.bias"] = (cse,)
        s[f"{p}._se_expand.weight"] = (mid, cse, 1, 1)
        s[f"{p}._se_expand.bias"] = (mid,)
        s[f"{p}._project_conv.weight"] = (cout, mid, 1, 1)
        s.update(_bn_shapes(f"{p}._bn2", cout))
    head = _eff_round_filters(1280)
    s["_conv_head.weight"] = (head, efficientnet_b3_blocks()[-1][4], 1, 1)
    s.update(_bn_shapes("_bn1", head))
    return s


def _same_pad_static(k: int, stride: int, image_size: int = 300):
    """Conv2dStaticSamePadding (efficientnet_utils.py:183-212): the padding is computed ONCE for
    ``image_size`` (300 for b3), not for the actual feature map: (left/top, right/bottom)."""
    import math

    o = math.ceil(image_size / stride)
    pad = max((o - 1) * stride + (k - 1) + 1 - image_size, 0)
    return pad // 2, pad - pad // 2


def efficientnet_b3_forward(x: torch.Tensor, sd, prefix: str = "") -> torch.Tensor:
    """``EfficientNet.extract_features`` (CP/models/efficientnet.py:265-277) in eval mode (no drop
    connect); returns ``[B,1536,h',w']`` ([.,.,7,10] for a 240x320 input)."""
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    eps = _EFF_B3["bn_eps"]

    def conv(x, name, k, stride, groups=1, bias=None):
        lo, hi = _same_pad_static(k, stride)
        if lo or hi:
            x = F.pad(x, (lo, hi, lo, hi))
        return F.conv2d(x, _t(sdp, name), None if bias is None else _t(sdp, bias), stride=stride, groups=groups)

    def bn(x, p):
        return F.batch_norm(x, _t(sdp, f"{p}.running_mean"), _t(sdp, f"{p}.running_var"), _t(sdp, f"{p}.weight"),
                            _t(sdp, f"{p}.bias"), False, 0.0, eps)

    def swish(x):
        return x * torch.sigmoid(x)

    x = swish(bn(conv(x, "_conv_stem.weight", 3, 2), "_bn0"))
    for bi, (k, stride, e, cin, cout, _cse) in enumerate(efficientnet_b3_blocks()):
        p, inp = f"_blocks.{bi}", x
        if e != 1:
            x = swish(bn(conv(x, f"{p}._expand_conv.weight", 1, 1), f"{p}._bn0"))
        x = swish(bn(conv(x, f"{p}._depthwise_conv.weight", k, stride, groups=x.shape[1]), f"{p}._bn1"))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = conv(swish(conv(sq, f"{p}._se_reduce.weight", 1, 1, bias=f"{p}._se_reduce.bias")),
                  f"{p}._se_expand.weight", 1, 1, bias=f"{p}._se_expand.bias")
        x = torch.sigmoid(sq) * x
        x = bn(conv(x, f"{p}._project_conv.weight", 1, 1), f"{p}._bn2")
        if stride == 1 and cin == cout:
            x = x + inp
    return swish(bn(conv(x, "_conv_head.weight", 1, 1), "_bn1"))


def net_forward(x: torch.Tensor, sd, arch: str, heads=("pose",)) -> Dict[str, torch.Tensor]:
    """``PosePredictor.net_forward`` (MP/models/pose_rigid.py:352-374,
    CP/models/pose.py:108-114): backbone -> (spatial mean if 4-D) -> linear heads.
    ``sd`` uses the PosePredictor key layout: ``backbone.*``, ``pose_fc.*``,
    ``views_logits_head.*``."""
    if arch == "vanilla_resnet34":
        f = resnet34_forward(x, sd, "backbone.")
    elif arch == "efficientnet-b3":
        f = efficientnet_b3_forward(x, sd, "backbone.").flatten(2).mean(dim=-1)
    else:
        f = wide_resnet_forward(x, sd, 34 if arch == "resnet34" else 18, "backbone.")
        f = f.flatten(2).mean(dim=-1)
    out = {}
    if "features" in heads:  # the pooled backbone features the heads read (feature-level parity tests)
        out["features"] = f
    if "pose" in heads:
        out["pose"] = F.linear(f, _t(sd, "pose_fc.weight"), _t(sd, "pose_fc.bias"))
    if "renderings_logits" in heads:
        out["renderings_logits"] = F.linear(f, _t(sd, "views_logits_head.weight"),
                                            _t(sd, "views_logits_head.bias"))
    return out


def predictor_param_shapes(arch: str, n_inputs: int, pose_dim: int = 9, n_views_logits: int = 0):
    eff = arch == "efficientnet-b3"
    nf = 1536 if eff else 512  # backbone.n_features (CP/training/pose_models_cfg.py:35,42)
    s = {f"backbone.{k}": v for k, v in (efficientnet_b3_param_shapes(n_inputs) if eff else param_shapes(arch, n_inputs)).items()}
    if pose_dim:
        s["pose_fc.weight"] = (pose_dim, nf)
        s["pose_fc.bias"] = (pose_dim,)
    if n_views_logits:
        s["views_logits_head.weight"] = (n_views_logits, nf)
        s["views_logits_head.bias"] = (n_views_logits,)
    return s
