"""Model factories: configuration + ``state_dict`` -> predictors on the device.

Mirrors ``create_model_pose`` / ``check_update_config``
(``MP/training/pose_models_cfg.py:36-142``), ``create_pose_model_cosypose``
(``CP/training/pose_models_cfg.py:30-74``), the legacy key renames of
``TB/utils/models_compat.py:17-27`` and the checkpoint layout read by
``load_pose_models`` (``TB/inference/utils.py:84-161``: ``<run_dir>/config.yaml`` +
``<run_dir>/checkpoint.pth.tar`` holding ``{"state_dict": ...}``).
"""

from __future__ import annotations

from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .pose_predictor import CosyPosePosePredictor, PosePredictor
from .renderer import BatchRenderer

_ARCH_OF = {"vanilla_resnet34": "vanilla_resnet34", "resnet34": "resnet34", "resnet18": "resnet18"}


def change_keys_of_older_models(state_dict: Dict) -> Dict:
    """``TB/utils/models_compat.py:17-27``."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("backbone.backbone"):
            k = "backbone." + k[len("backbone.backbone."):]
        elif k.startswith("backbone.head.0."):
            k = "views_logits_head." + k[len("backbone.head.0."):]
        out[k] = v
    return out


def check_update_config(cfg) -> SimpleNamespace:
    """Defaults for configurations written by older training code
    (``MP/training/pose_models_cfg.py:36-86``).  Accepts a dict or any attribute bag."""
    d = dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))
    d["is_coarse_compat"] = False
    if d.get("input_strategy") == "input=obs+one_render":
        d.update(is_coarse_compat=True, n_rendered_views=1, multiview_type="1view_TCO",
                 predict_rendered_views_logits=True, remove_TCO_rendering=True, predict_pose_update=False)
    mv = d.get("multiview_type")
    if mv in ("front_3views", "front_5views", "front_1view"):
        d["multiview_type"] = "TCO+" + mv
    d.setdefault("predict_pose_update", True)
    d.setdefault("remove_TCO_rendering", False)
    d.setdefault("predict_rendered_views_logits", False)
    if "n_rendered_views" not in d:
        d["n_rendered_views"] = d.pop("n_views", 1)
    d.setdefault("render_normals", False)
    d.setdefault("render_depth", False)
    d.setdefault("input_depth", False)
    if "multiview_type" not in d:
        d["multiview_type"] = "TCO"
        assert not d["remove_TCO_rendering"]
    d.setdefault("views_inplane_rotations", False)
    if "depth_augmentation" not in d:  # configurations older than the depth-augmentation option
        d["depth_normalization_type"] = "tCR_scale"
    d.setdefault("depth_normalization_type", "tCR_scale_clamp_center")  # training_config.py:102
    d.setdefault("renderer", "panda3d")
    d.setdefault("backbone_str", "vanilla_resnet34")
    return SimpleNamespace(**d)


def n_input_channels(cfg) -> int:
    """``MP/training/pose_models_cfg.py:94-103``."""
    n_normals = 3 if cfg.render_normals else 0
    n_rdepth = 1 if cfg.render_depth else 0
    n_depth = 1 if cfg.input_depth else 0
    return (3 + n_depth) + (3 + n_normals + n_rdepth) * cfg.n_rendered_views


def _arch(backbone_str: str) -> str:
    if backbone_str == "vanilla_resnet34":
        return "vanilla_resnet34"
    if backbone_str == "resnet34" or "resnet34" in backbone_str and "width" not in backbone_str:
        return "resnet34"
    if "resnet18" in backbone_str:
        return "resnet18"
    if backbone_str == "efficientnet-b3":  # CP/training/pose_models_cfg.py:33-35 (the released CosyPose checkpoints)
        return "efficientnet-b3"
    raise ValueError("Unknown backbone", backbone_str)


def pose_model_param_shapes(backbone_str: str, n_inputs: int, pose_dim: int = 9,
                            n_views_logits: int = 0) -> Dict[str, tuple]:
    """state-dict key -> shape of a pose model, in the reference's registration order:
    ``vanilla_resnet34`` = ``MP/models/torchvision_resnet.py:191-344`` (7x7 stem, BasicBlock,
    BN after the 1x1 downsample, fc 512->512); ``resnet34`` / ``resnet18`` = the WideResNet of
    ``MP/models/wide_resnet.py:68-154`` (5x5 stem, pre-activation BasicBlockV2, conv-only
    downsample); heads ``MP/models/pose_rigid.py:135-149``."""
    s: Dict[str, tuple] = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)
        s[f"{p}.num_batches_tracked"] = ()

    if backbone_str == "efficientnet-b3":
        return _efficientnet_b3_param_shapes(n_inputs, pose_dim, n_views_logits)
    planes = [64, 128, 256, 512]
    vanilla = backbone_str == "vanilla_resnet34"
    layers = [2, 2, 2, 2] if backbone_str == "resnet18" else [3, 4, 6, 3]
    assert backbone_str in ("vanilla_resnet34", "resnet34", "resnet18"), backbone_str
    k1 = 7 if vanilla else 5
    s["backbone.conv1.weight"] = (64, n_inputs, k1, k1)
    bn("backbone.bn1", 64)
    inpl = 64
    for li, (pl, nb) in enumerate(zip(planes, layers), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"backbone.layer{li}.{b}"
            if vanilla:
                s[f"{p}.conv1.weight"] = (pl, inpl, 3, 3)
                bn(f"{p}.bn1", pl)
                s[f"{p}.conv2.weight"] = (pl, pl, 3, 3)
                bn(f"{p}.bn2", pl)
                if stride != 1 or inpl != pl:
                    s[f"{p}.downsample.0.weight"] = (pl, inpl, 1, 1)
                    bn(f"{p}.downsample.1", pl)
            else:
                bn(f"{p}.bn1", inpl)
                s[f"{p}.conv1.weight"] = (pl, inpl, 3, 3)
                bn(f"{p}.bn2", pl)
                s[f"{p}.conv2.weight"] = (pl, pl, 3, 3)
                if stride != 1 or inpl != pl:
                    s[f"{p}.downsample.weight"] = (pl, inpl, 1, 1)
            inpl = pl
    if vanilla:
        s["backbone.fc.weight"] = (512, 512)
        s["backbone.fc.bias"] = (512,)
    if pose_dim:
        s["pose_fc.weight"] = (pose_dim, 512)
        s["pose_fc.bias"] = (pose_dim,)
    if n_views_logits:
        s["views_logits_head.weight"] = (n_views_logits, 512)
        s["views_logits_head.bias"] = (n_views_logits,)
    return s


def efficientnet_b3_blocks():
    """``[(kernel, stride, expand, in, out, se_channels)]`` of the 26 MBConv blocks of EfficientNet-b3
    (``CP/models/efficientnet_utils.py:241-256,339-367``: width 1.2, depth 1.4, filters rounded to 8)."""
    import math

    def rf(f):
        f = f * 1.2
        new = max(8, int(f + 4) // 8 * 8)
        return int(new + 8 if new < 0.9 * f else new)

    base = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
            (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
    blocks = []
    for r, k, s, e, i, o in base:
        i, o = rf(i), rf(o)
        for j in range(int(math.ceil(1.4 * r))):
            cin, stride = (i, s) if j == 0 else (o, 1)
            blocks.append((k, stride, e, cin, o, max(1, int(cin * 0.25))))
    return blocks


def _efficientnet_b3_param_shapes(n_inputs: int, pose_dim: int, n_views_logits: int) -> Dict[str, tuple]:
    """Registration order of ``CP/models/efficientnet.py:165-233`` under ``backbone.``; the heads see
    1536 features (``CP/training/pose_models_cfg.py:33-35``)."""
    s: Dict[str, tuple] = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)
        s[f"{p}.num_batches_tracked"] = ()

    blocks = efficientnet_b3_blocks()
    s["backbone._conv_stem.weight"] = (40, n_inputs, 3, 3)
    bn("backbone._bn0", 40)
    for bi, (k, _stride, e, cin, cout, cse) in enumerate(blocks):
        p, mid = f"backbone._blocks.{bi}", cin * e
        if e != 1:
            s[f"{p}._expand_conv.weight"] = (mid, cin, 1, 1)
            bn(f"{p}._bn0", mid)
        s[f"{p}._depthwise_conv.weight"] = (mid, 1, k, k)
        bn(f"{p}._bn1", mid)
        s[f"{p}._se_reduce.weight"] = (cse, mid, 1, 1)
        s[f"{p}._se_reduce.bias"] = (cse,)
        s[f"{p}._se_expand.weight"] = (mid, cse, 1, 1)
        s[f"{p}._se_expand.bias"] = (mid,)
        s[f"{p}._project_conv.weight"] = (cout, mid, 1, 1)
        bn(f"{p}._bn2", cout)
    s["backbone._conv_head.weight"] = (1536, blocks[-1][4], 1, 1)
    bn("backbone._bn1", 1536)
    if pose_dim:
        s["pose_fc.weight"] = (pose_dim, 1536)
        s["pose_fc.bias"] = (pose_dim,)
    if n_views_logits:
        s["views_logits_head.weight"] = (n_views_logits, 1536)
        s["views_logits_head.bias"] = (n_views_logits,)
    return s


def _two_lanes(make, renderer: BatchRenderer, max_batch: int, n_lanes: int = 2):
    """``n_lanes`` predictors for :class:`TwoLanePredictor`: every lane after the first gets its own mesh store
    (rasteriser scratch) through ``renderer.clone_for_lane()`` -- same render state (msaa / aniso / ...)."""
    from .pose_predictor import TwoLanePredictor

    share = (max_batch + n_lanes - 1) // n_lanes
    return TwoLanePredictor([make(renderer if i == 0 else renderer.clone_for_lane(), share) for i in range(n_lanes)])


def create_model_pose(cfg, renderer: BatchRenderer, mesh_db=None, state_dict: Optional[Dict] = None,
                      max_batch: int = 128, precision: str = "f32", n_lanes: int = 1, graphs: bool = False) -> PosePredictor:
    """MegaPose predictor (``MP/training/pose_models_cfg.py:89-142``).  ``state_dict`` holds the
    reference's keys (``backbone.*``, ``pose_fc.*``, ``views_logits_head.*``).  ``n_lanes=2``: the refiner's
    ``forward`` runs two half-batch chains on two streams (:class:`TwoLanePredictor`).  ``graphs=True``: ``forward`` is
    captured once per call signature and replayed as a hipGraph (``happypose_amd.graphs``; pays off when the launches
    are shorter than the host's launch rate: refiner batches of <= 64)."""
    if n_lanes >= 2:
        model = _two_lanes(lambda r, mb: create_model_pose(cfg, r, mesh_db, state_dict, mb, precision), renderer, max_batch, n_lanes)
        model.use_graphs = graphs
        return model
    assert n_lanes == 1
    cfg = check_update_config(cfg)
    assert state_dict is not None, "weights are required (no training path here)"
    sd = change_keys_of_older_models(state_dict)
    net = ops.Net(_arch(cfg.backbone_str), n_input_channels(cfg), sd, max_batch=max_batch, device=renderer.device,
                  precision=precision)
    # the coarse compat model renders exactly the TCO view
    mv = cfg.multiview_type if cfg.n_rendered_views > 1 else "TCO"
    model = PosePredictor(
        backbone=net, renderer=renderer, mesh_db=mesh_db, render_size=(240, 320),
        n_rendered_views=cfg.n_rendered_views, views_inplane_rotations=cfg.views_inplane_rotations,
        multiview_type=mv, render_normals=cfg.render_normals, render_depth=cfg.render_depth,
        input_depth=cfg.input_depth, predict_rendered_views_logits=cfg.predict_rendered_views_logits,
        remove_TCO_rendering=cfg.remove_TCO_rendering, predict_pose_update=cfg.predict_pose_update,
        depth_normalization_type=cfg.depth_normalization_type)
    model.cfg = model.config = cfg
    model.use_graphs = graphs
    return model


def create_pose_model_cosypose(cfg, renderer: BatchRenderer, mesh_db=None, state_dict: Optional[Dict] = None,
                               max_batch: int = 128, precision: str = "f32", n_lanes: int = 1,
                               graphs: bool = False) -> CosyPosePosePredictor:
    """``CP/training/pose_models_cfg.py:30-53`` (6 input channels; ``n_pose_dims`` = 9).  ``n_lanes`` as in
    :func:`create_model_pose`."""
    if n_lanes >= 2:
        model = _two_lanes(lambda r, mb: create_pose_model_cosypose(cfg, r, mesh_db, state_dict, mb, precision), renderer,
                           max_batch, n_lanes)
        model.use_graphs = graphs
        return model
    assert n_lanes == 1
    d = dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))
    d.setdefault("init_method", "v0")  # check_update_config, :24-27
    d.setdefault("n_pose_dims", 9)
    cfg = SimpleNamespace(**d)
    assert state_dict is not None
    net = ops.Net(_arch(cfg.backbone_str), 6, state_dict, max_batch=max_batch, device=renderer.device, precision=precision)
    model = CosyPosePosePredictor(backbone=net, renderer=renderer, mesh_db=mesh_db, render_size=(240, 320),
                                  pose_dim=cfg.n_pose_dims)
    model.cfg = model.config = cfg
    model.use_graphs = graphs
    return model


def load_checkpoint(run_dir) -> Dict[str, torch.Tensor]:
    """``<run_dir>/checkpoint.pth.tar`` -> state dict (``TB/inference/utils.py:146-152``): tensors only
    (``weights_only=True``: a downloaded checkpoint is never unpickled as code) with the legacy key renames
    applied -- the same path ``load_model.load_pose_models`` takes."""
    from .load_model import load_state_dict

    return load_state_dict(run_dir)
