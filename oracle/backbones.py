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


def net_forward(x: torch.Tensor, sd, arch: str, heads=("pose",)) -> Dict[str, torch.Tensor]:
    """``PosePredictor.net_forward`` (MP/models/pose_rigid.py:352-374,
    CP/models/pose.py:108-114): backbone -> (spatial mean if 4-D) -> linear heads.
    ``sd`` uses the PosePredictor key layout: ``backbone.*``, ``pose_fc.*``,
    ``views_logits_head.*``."""
    if arch == "vanilla_resnet34":
        f = resnet34_forward(x, sd, "backbone.")
    else:
        f = wide_resnet_forward(x, sd, 34 if arch == "resnet34" else 18, "backbone.")
        f = f.flatten(2).mean(dim=-1)
    out = {}
    if "pose" in heads:
        out["pose"] = F.linear(f, _t(sd, "pose_fc.weight"), _t(sd, "pose_fc.bias"))
    if "renderings_logits" in heads:
        out["renderings_logits"] = F.linear(f, _t(sd, "views_logits_head.weight"),
                                            _t(sd, "views_logits_head.bias"))
    return out


def predictor_param_shapes(arch: str, n_inputs: int, pose_dim: int = 9, n_views_logits: int = 0):
    s = {f"backbone.{k}": v for k, v in param_shapes(arch, n_inputs).items()}
    if pose_dim:
        s["pose_fc.weight"] = (pose_dim, 512)
        s["pose_fc.bias"] = (pose_dim,)
    if n_views_logits:
        s["views_logits_head.weight"] = (n_views_logits, 512)
        s["views_logits_head.bias"] = (n_views_logits,)
    return s
