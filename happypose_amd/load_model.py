"""On-disk model ingestion: ``<models_root>/<run_id>/{config.yaml, checkpoint.pth.tar}`` ->
predictors -> :class:`PoseEstimator`.

Mirrors ``load_cfg`` / ``load_pose_models`` (``TB/inference/utils.py:77-161``) and
``NAMED_MODELS`` / ``load_named_model`` (``TB/utils/load_model.py:8-88``) of the reference
(SURVEY.md 8f-2).  Differences that follow from the design: there are no renderer worker
processes (``renderer_kwargs`` is accepted and ignored apart from validation), the weights
are folded / packed for the HIP kernels at load time instead of ``load_state_dict``, and the
training configuration is read with a RESTRICTED YAML loader: the reference uses
``yaml.UnsafeLoader`` because its ``config.yaml`` files are dumps of ``argparse.Namespace``
/ dataclass objects; only those tags (and tuples) are mapped here, nothing is executed.
"""

from __future__ import annotations

import os
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import yaml

from .mesh_store import MeshDataBase, RigidObjectDataset
from .models import change_keys_of_older_models, check_update_config, create_model_pose
from .pose_estimator import PoseEstimator
from .renderer import BatchRenderer

# name -> (coarse run, refiner run, needs depth, depth refiner, inference parameters):
# the table of TB/utils/load_model.py:8-49
_NAMED = (
    ("megapose-1.0-RGB", "coarse-rgb-906902141", "refiner-rgb-653307694", False, None, (5, 1, False)),
    ("megapose-1.0-RGBD", "coarse-rgb-906902141", "refiner-rgbd-288182519", True, None, (5, 1, False)),
    ("megapose-1.0-RGB-multi-hypothesis", "coarse-rgb-906902141", "refiner-rgb-653307694", False, None, (5, 5, False)),
    ("megapose-1.0-RGB-multi-hypothesis-icp", "coarse-rgb-906902141", "refiner-rgb-653307694", True, "ICP", (5, 5, True)),
)
NAMED_MODELS: Dict[str, Dict] = {}
for _name, _coarse, _refiner, _depth, _drefiner, (_iters, _hyp, _run_dr) in _NAMED:
    _entry = {"coarse_run_id": _coarse, "refiner_run_id": _refiner, "requires_depth": _depth,
              "inference_parameters": {"n_refiner_iterations": _iters, "n_pose_hypotheses": _hyp}}
    if _drefiner:
        _entry["depth_refiner"] = _drefiner
        _entry["inference_parameters"]["run_depth_refiner"] = _run_dr
    NAMED_MODELS[_name] = _entry


class _ConfigLoader(yaml.SafeLoader):
    """SafeLoader + the object tags the reference's training configs contain."""


def _construct_namespace(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        fields = loader.construct_mapping(node, deep=True)
        fields = fields.get("dictitems", fields) if set(fields) <= {"dictitems", "state"} else fields
        return SimpleNamespace(**{str(k): v for k, v in fields.items()})
    return SimpleNamespace()


def _construct_tuple(loader, node):
    return tuple(loader.construct_sequence(node, deep=True))


_ConfigLoader.add_multi_constructor("tag:yaml.org,2002:python/object:", _construct_namespace)
_ConfigLoader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", _construct_namespace)
_ConfigLoader.add_constructor("tag:yaml.org,2002:python/tuple", _construct_tuple)


def load_cfg(path) -> SimpleNamespace:
    """``TB/inference/utils.py:77-81``: a training ``config.yaml`` as an attribute bag."""
    cfg = yaml.load(Path(path).read_text(), Loader=_ConfigLoader)
    if isinstance(cfg, dict):
        cfg = SimpleNamespace(**cfg)
    assert isinstance(cfg, SimpleNamespace), f"{path}: unsupported configuration document"
    return cfg


def default_models_root() -> Path:
    """``LOCAL_DATA_DIR / "megapose-models"`` (``MP/config.py``; ``HAPPYPOSE_DATA_DIR``)."""
    root = os.environ.get("HAPPYPOSE_DATA_DIR") or os.environ.get("MEGAPOSE_DATA_DIR")
    assert root, "set HAPPYPOSE_DATA_DIR (the reference's data root) or pass models_root"
    return Path(root) / "megapose-models"


def load_state_dict(run_dir) -> Dict[str, torch.Tensor]:
    """``checkpoint.pth.tar`` -> state dict with the legacy key renames applied
    (``TB/inference/utils.py:146-152``).  Tensors only (``weights_only``)."""
    ckpt = torch.load(Path(run_dir) / "checkpoint.pth.tar", map_location="cpu", weights_only=True)
    return change_keys_of_older_models(ckpt["state_dict"])


def load_pose_models(coarse_run_id: Optional[str], refiner_run_id: Optional[str], object_dataset: RigidObjectDataset,
                     force_panda3d_renderer: bool = True, renderer_kwargs: Optional[Dict] = None,
                     models_root: Optional[Path] = None, device="cuda", max_batch: int = 128,
                     coarse_precision: str = "f32") -> Tuple[object, object, MeshDataBase]:
    """``(coarse_model, refiner_model, mesh_db)`` as ``TB/inference/utils.py:84-161``; a run id of
    ``None`` gives ``None`` for that model.  Both models share one device-resident mesh store."""
    assert force_panda3d_renderer, "only the Panda3D camera / light model is implemented (as in the reference)"
    for k in (renderer_kwargs or {}):
        assert k in ("split_objects", "preload_cache", "n_workers"), f"unknown renderer option {k!r}"
    models_root = Path(models_root) if models_root is not None else default_models_root()
    mesh_db = MeshDataBase.from_object_ds(object_dataset)
    renderer = BatchRenderer(object_dataset, device=device)

    def load_model(run_id, precision):
        if run_id is None:
            return None
        run_dir = models_root / run_id
        cfg = check_update_config(load_cfg(run_dir / "config.yaml"))
        model = create_model_pose(cfg, renderer, mesh_db=mesh_db, state_dict=load_state_dict(run_dir),
                                  max_batch=max_batch, precision=precision)
        model.cfg = model.config = cfg
        return model

    return load_model(coarse_run_id, coarse_precision), load_model(refiner_run_id, "f32"), mesh_db


def default_coarse_precision(model_name: str) -> str:
    """``"f16"`` for the multi-hypothesis configurations (BASELINE.json config 5 names fp16 for exactly this stage: 576 grid views
    per object are SCORED and the five best kept -- the fp16 plan's logits stay within 0.05 of the oracle's spread over an
    object's grid poses and pick the same five, ``tests/test_gpu_pipeline.py::test_c5_coarse_scoring_vs_oracle``), ``"f32"`` for
    the single-hypothesis ones, whose coarse winner alone seeds the refiner."""
    return "f16" if NAMED_MODELS[model_name]["inference_parameters"]["n_pose_hypotheses"] > 1 else "f32"


def load_named_model(model_name: str, object_dataset: RigidObjectDataset, n_workers: int = 4, bsz_images: int = 128,
                     models_root: Optional[Path] = None, device="cuda", coarse_precision: Optional[str] = None) -> PoseEstimator:
    """``TB/utils/load_model.py:52-88``.  ``coarse_precision`` (beyond the reference's signature): ``"f16"`` plans the coarse /
    scoring network in fp16, ``"f32"`` in the reference's arithmetic; ``None`` = :func:`default_coarse_precision` (fp16 for the
    multi-hypothesis configurations).  The refiner always runs in fp32.  Measured on the end-to-end frame: ``bench.py`` keys
    ``e2e`` / ``e2e_f16_coarse``, each with its parity against the oracle estimator."""
    model = NAMED_MODELS[model_name]
    if coarse_precision is None:
        coarse_precision = default_coarse_precision(model_name)
    coarse_model, refiner_model, mesh_db = load_pose_models(
        coarse_run_id=model["coarse_run_id"], refiner_run_id=model["refiner_run_id"], object_dataset=object_dataset,
        force_panda3d_renderer=True, renderer_kwargs={"preload_cache": False, "split_objects": False, "n_workers": n_workers},
        models_root=models_root, device=device, max_batch=bsz_images, coarse_precision=coarse_precision)
    depth_refiner = None
    if model.get("depth_refiner") == "ICP":
        from .icp_refiner import ICPRefiner

        depth_refiner = ICPRefiner(mesh_db, refiner_model.renderer)
    return PoseEstimator(refiner_model=refiner_model, coarse_model=coarse_model, detector_model=None,
                         depth_refiner=depth_refiner, bsz_objects=8, bsz_images=bsz_images)
