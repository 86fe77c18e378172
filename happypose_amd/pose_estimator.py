"""Inference orchestrators: the drop-in entry point ``run_inference_pipeline``.

Mirrors (signatures, defaults, returned ``PandasTensorCollection`` columns/tensors and
``extra_data`` keys) the reference's

* MegaPose  ``PoseEstimator``  ``MP/inference/pose_estimator.py:55-668``
* CosyPose  ``PoseEstimator``  ``CP/integrated/pose_estimator.py:31-470``  (here
  :class:`CosyPoseEstimator`)

Differences that are deliberate:
  * hypotheses are never gathered into per-hypothesis image copies
    (``observation.images[batch_im_ids]`` in the reference): kernels index the frame by
    ``batch_im_id``;
  * the detections x SO(3)-grid expansion is built with vectorised numpy instead of a
    python loop of ``pd.DataFrame([row] * M)`` (seconds at M = 576 in the reference);
  * ``bsz_objects`` / ``bsz_images`` only bound the working-set size (chunks are large by
    default because 288 GB of HBM are available), results are independent of them;
  * with ``torch.distributed`` initialised (more than one rank) the estimators SHARD their hypothesis rows: every
    rank runs ``forward_refiner`` / ``forward_coarse_model`` / ``forward_scoring_model`` on its contiguous
    ``distributed.shard_range`` of the rows and ONE all-gather per stage (``distributed.gather_rows``) hands every rank
    the whole result -- identical on every rank and to a single-process run.  This replaces the reference's rank files +
    barriers (``TB/utils/tensor_collection.py:166-187``, ``MP/evaluation/prediction_runner.py:65-76``).  It is a
    collective: all ranks must call with the same inputs; ``estimator.shard_hypotheses = False`` switches it off for
    scene-parallel use (each rank on its own frames).
"""

from __future__ import annotations

import time
from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from . import distributed as D
from . import ops
from .tensor_collection import PandasTensorCollection, concatenate, filter_top_pose_estimates

PoseEstimatesType = PandasTensorCollection
DetectionsType = PandasTensorCollection


# ------------------------------------------------------------------------------ datatypes
@dataclass
class InferenceConfig:
    """``TB/inference/types.py:87-98``."""

    detection_type: str = "detector"
    coarse_estimation_type: str = "SO3_grid"
    SO3_grid_size: int = 576
    n_refiner_iterations: int = 5
    n_pose_hypotheses: int = 5
    run_depth_refiner: bool = False
    depth_refiner: Optional[str] = None
    bsz_objects: int = 16
    bsz_images: int = 576


@dataclass
class ObservationTensor:
    """``TB/inference/types.py:101-235``: images ``[B,C,H,W]`` f32, C=3 (rgb in [0,1]) or 4
    (rgb + depth in metres); ``K [B,3,3]``."""

    images: torch.Tensor
    K: Optional[torch.Tensor] = None

    def cuda(self) -> "ObservationTensor":
        return self.to("cuda")

    def cpu(self) -> "ObservationTensor":
        return self.to("cpu")

    def to(self, device) -> "ObservationTensor":
        self.images = self.images.to(device)
        if self.K is not None:
            self.K = self.K.to(device)
        return self

    @property
    def batch_size(self) -> int:
        return self.images.shape[0]

    @property
    def channel_dim(self) -> int:
        return self.images.shape[1]

    @property
    def depth(self) -> torch.Tensor:
        assert self.channel_dim == 4
        return self.images[:, 3]

    def is_valid(self) -> bool:
        if self.images.ndim != 4 or self.channel_dim not in (3, 4):
            return False
        if self.K is not None and self.K.shape != torch.Size([self.batch_size, 3, 3]):
            return False
        if self.images.dtype != torch.float:
            return False
        return not bool(torch.max(self.images[:, :3]) > 1)

    @staticmethod
    def from_numpy(rgb: np.ndarray, depth: Optional[np.ndarray] = None,
                   K: Optional[np.ndarray] = None) -> "ObservationTensor":
        assert rgb.dtype == np.uint8
        t = torch.as_tensor(rgb).float() / 255
        if t.shape[-1] == 3:
            t = t.permute(2, 0, 1)
        if depth is not None:
            t = torch.cat((t, torch.as_tensor(depth).float().unsqueeze(0)), dim=0)
        return ObservationTensor(t.unsqueeze(0).contiguous(), torch.as_tensor(K).float().unsqueeze(0))

    @staticmethod
    def from_torch_batched(rgb: torch.Tensor, depth: Optional[torch.Tensor], K: torch.Tensor) -> "ObservationTensor":
        assert rgb.dtype == torch.uint8
        img = rgb.float() / 255
        if depth is not None:
            if depth.ndim == 3:
                depth = depth.unsqueeze(1)
            img = torch.cat((img, depth.float()), dim=1)
        return ObservationTensor(img.contiguous(), torch.as_tensor(K).float())


def assert_detections_valid(detections: DetectionsType) -> None:
    """``TB/inference/types.py:73-84``."""
    for f in ("batch_im_id", "label", "instance_id"):
        assert f in detections.infos, f"detections.infos missing column {f}"
    assert "bboxes" in detections.tensors, "detections missing tensor bboxes."


def add_instance_id(inputs):
    """``TB/inference/utils.py:163-183``: number repeated (image, label) pairs."""
    if "instance_id" in inputs.infos:
        return inputs
    df = inputs.infos
    df["instance_id"] = df.groupby(["batch_im_id", "label"]).cumcount()
    inputs.infos = df
    return inputs


def filter_detections(detections, labels: Optional[List[str]] = None, one_instance_per_class: bool = False):
    """``TB/inference/utils.py:186-207``."""
    if labels is not None:
        df = detections.infos
        df = df[df.label.isin(labels)]
        detections = detections[df.index.tolist()]
    if one_instance_per_class:
        df = detections.infos
        df = df.sort_values("score", ascending=False).groupby(["batch_im_id", "label"]).head(1)
        detections = detections[df.index.tolist()]
    return detections


def make_detections_from_object_data(labels: List[str], bboxes: np.ndarray) -> DetectionsType:
    """``TB/inference/utils.py:229-241`` (takes labels + ``bbox_modal`` rows)."""
    infos = pd.DataFrame({"label": list(labels), "batch_im_id": 0, "instance_id": np.arange(len(labels))})
    return PandasTensorCollection(infos=infos, bboxes=torch.as_tensor(np.stack(bboxes)).float())


def load_SO3_grid(resolution: int) -> torch.Tensor:
    """``TB/utils/transform_utils.py:24-48``: ``data_{n}.qua`` rows ``x y z w`` ->
    ``[N,3,3]`` rotation matrices (roma's unit-quaternion formula)."""
    from pathlib import Path

    fname = Path(__file__).resolve().parent / "data" / f"data_{resolution}.qua"
    assert fname.is_file(), f"File {fname} not found"
    q = torch.tensor(np.loadtxt(fname, dtype=np.float64).reshape(-1, 4).tolist())
    x, y, z, w = q.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    R = torch.stack((1 - (ty * y + tz * z), tx * y - tz * w, tx * z + ty * w,
                     tx * y + tz * w, 1 - (tx * x + tz * z), ty * z - tx * w,
                     tx * z - ty * w, ty * z + tx * w, 1 - (tx * x + ty * y)), dim=-1)
    return R.reshape(-1, 3, 3)


class _EstimatorBase:
    """``PoseEstimationModule`` (``TB/inference/pose_estimator.py:12-29``) + what both
    estimators share."""

    refiner_model = None
    coarse_model = None

    def _init_common(self, refiner_model, coarse_model, detector_model, bsz_objects, bsz_images):
        self.coarse_model = coarse_model
        self.refiner_model = refiner_model
        self.detector_model = detector_model
        self.bsz_objects = bsz_objects
        self.bsz_images = bsz_images
        model = refiner_model if refiner_model is not None else coarse_model
        if model is None:
            raise ValueError("At least one of refiner_model or  coarse_model must be specified.")
        self.cfg = getattr(model, "cfg", None)
        self.mesh_db = model.mesh_db
        self.device = model.device
        self.keep_all_outputs = False
        self.keep_all_coarse_outputs = False
        self.refiner_outputs = None
        self.coarse_outputs = None
        self.debug_dict: dict = {}

    def eval(self):
        return self

    # None: shard the hypothesis rows over the ranks whenever torch.distributed runs with more than one
    # (happypose_amd.distributed.sharding_active); False: never (ranks work on different frames)
    shard_hypotheses: Optional[bool] = None

    def _shard(self, n_rows: int) -> Tuple[int, int, bool]:
        """``(start, end, sharded)``: the rows of an ``n_rows`` table this rank computes."""
        if D.sharding_active(self.shard_hypotheses):
            s, e = D.shard_range(n_rows)
            return s, e, True
        return 0, n_rows, False

    def _guarded(self, model, stage):
        """Run ``stage()`` and ask the model's backbone for its numerical guard at the stage's end (one stream
        synchronisation, where the pipeline synchronises anyway to read scores).  If an activation left the fp16 range
        of the default split-fp16 conv kernels (``HP_STATUS_NONFINITE``) the results are invalid; the backbone has
        switched to its exact-fp32 kernels, so the stage is simply run again -- the reference's fp32 arithmetic.
        A sharded stage ends in a collective, so the ranks decide TOGETHER (MAX over ranks) whether to repeat it."""
        out = stage()
        status = getattr(model, "numerics_status", None)
        flag = int(status() & ops.STATUS_NONFINITE) if status is not None else 0
        if D.sharding_active(self.shard_hypotheses):
            flag = D.all_ranks_max(flag, self.device)
        if flag:
            # every network of every rank (and of every lane) goes exact BEFORE the repeat: only the one whose guard
            # fired has switched by itself, and a stage merged from two arithmetics -- or one rank left alone on the
            # slower kernels, stalling every later all-gather -- is not "identical to a single-process run"
            backbone = getattr(model, "backbone", None)
            if backbone is not None and hasattr(backbone, "force_exact"):
                backbone.force_exact(True)
            out = stage()
        return out

    def _run_model_chunks(self, model, observation, data_TCO_input, n_iterations, bsz, tag,
                          keep_all_outputs=False, **kw) -> Tuple[dict, dict]:
        t_start = time.time()
        preds, extra = self._guarded(model, lambda: self._run_model_chunks_once(
            model, observation, data_TCO_input, n_iterations, bsz, tag, keep_all_outputs, **kw))
        # the stage's wall time INCLUDING the guard's stream synchronisation: the bookkeeping inside runs under the kernels, so
        # without it "time" would be the host's enqueue time and the GPU's share of the stage would be booked on whatever
        # synchronises next (the reference's per-stage `time` keys, MP/inference/pose_estimator.py:196-220, are wall times)
        extra["time"] = time.time() - t_start
        return preds, extra

    _ITER_COLS = (("poses", 16, (4, 4)), ("poses_input", 16, (4, 4)), ("K_crop", 9, (3, 3)), ("boxes_rend", 4, (4,)),
                  ("boxes_crop", 4, (4,)))

    def _run_model_chunks_once(self, model, observation, data_TCO_input, n_iterations, bsz, tag,
                               keep_all_outputs=False, **kw) -> Tuple[dict, dict]:
        """Shared body of forward_refiner / CosyPose forward_coarse_model.  Rows ``[s, e)`` of the hypothesis table
        are this rank's (all of them without sharding); they run in chunks of ``bsz``; with sharding the per-iteration
        outputs of all ranks meet in one all-gather of ``[n_local, 49 * n_iterations]`` rows."""
        t_start = time.time()
        B = len(data_TCO_input)
        s, e, sharded = self._shard(B)
        cols: Dict[str, List[List[torch.Tensor]]] = {f"iteration={n}": [[] for _ in self._ITER_COLS]
                                                     for n in range(1, n_iterations + 1)}
        all_outputs = []
        model_time = 0.0
        infos = data_TCO_input.infos
        labels_all = infos["label"].tolist()
        im_all_host = np.ascontiguousarray(infos["batch_im_id"].values)
        poses_all = data_TCO_input.poses
        # slices of the table's columns: the launches of a chunk are on their way before any pandas work happens
        # (the frames below are built while the GPU runs; `data_TCO_input[ids]` cost a gather + a frame per chunk)
        bounds = [(a, min(e, a + bsz)) for a in range(s, e, bsz)]
        # ONE host -> device copy of the frame ids, BEFORE the stage's launches: a pageable H2D copy waits for everything queued
        # ahead of it on its stream, so issued after the launches (as it was) it blocked the host for the whole stage and every
        # line of pandas bookkeeping below ran with the GPU idle (round 6: 0.7 of 22.8 ms per C2 step,
        # tools/probes/estimator_overhead.py).  Now the bookkeeping runs under the stage's kernels; the only wait is the guard's.
        im_all = torch.as_tensor(im_all_host, device=self.device)
        K_all = observation.K.to(self.device)[im_all.long()]  # PosePredictorOutput.K: the intrinsics of each hypothesis' frame
        chunks = [(labels_all[a:b], poses_all[a:b], im_all[a:b]) for a, b in bounds]
        t0 = time.time()
        if len(chunks) > 1 and bsz < getattr(model, "MIN_BATCH", 0) and hasattr(model, "forward_chunks"):
            # chunks too small to be split over the model's lanes run side by side, one whole chunk per lane
            chunk_outputs = model.forward_chunks(observation.images, observation.K, chunks, n_iterations=n_iterations, **kw)
        else:
            chunk_outputs = [model(images=observation.images, K=observation.K, TCO=T, n_iterations=n_iterations, labels=lab,
                                   im_ids=ids, **kw) for lab, T, ids in chunks]
        model_time += time.time() - t0
        for outputs_ in chunk_outputs:
            if keep_all_outputs:
                all_outputs.append(outputs_)
            for n in range(1, n_iterations + 1):
                o = outputs_[f"iteration={n}"]
                for dst, t in zip(cols[f"iteration={n}"], (o.TCO_output, o.TCO_input, o.K_crop, o.boxes_rend, o.boxes_crop)):
                    dst.append(t)
        # bookkeeping columns of the reference (``MP/inference/pose_estimator.py:196-197``), by GLOBAL row so that they
        # do not depend on the number of ranks
        df_all = infos.copy()
        df_all[f"{tag}_batch_idx"] = np.arange(B) // bsz
        df_all[f"{tag}_instance_idx"] = np.arange(B) % bsz
        n_local = e - s
        f = dict(dtype=torch.float32, device=self.device)
        table = {k: [(parts[0] if len(parts) == 1 else torch.cat(parts)) if parts else torch.zeros((0,) + shape, **f)
                     for parts, (_, _, shape) in zip(v, self._ITER_COLS)]
                 for k, v in cols.items()}
        if sharded:
            # explicit widths: an EMPTY shard (more ranks than rows) must still pack to [0, 49 * n_iterations]
            packed = torch.cat([t.reshape(n_local, width).float() for k in table
                                for t, (_, width, _) in zip(table[k], self._ITER_COLS)], dim=1) if n_iterations else \
                torch.zeros((n_local, 0), **f)
            full = D.gather_rows(packed.contiguous(), s, B)
            c0 = 0
            for k in table:
                for i, (_, width, shape) in enumerate(self._ITER_COLS):
                    table[k][i] = full[:, c0:c0 + width].reshape((B,) + shape)
                    c0 += width
        preds = {}
        for k, ts in table.items():
            named = {name: t for (name, _, _), t in zip(self._ITER_COLS, ts)}
            preds[k] = PandasTensorCollection(df_all, poses=named["poses"], poses_input=named["poses_input"],
                                              K_crop=named["K_crop"], K=K_all, boxes_rend=named["boxes_rend"],
                                              boxes_crop=named["boxes_crop"])
        extra_data = {"n_iterations": n_iterations, "outputs": all_outputs, "model_time": model_time,
                      "time": time.time() - t_start, "shard": (s, e)}
        return preds, extra_data


# -------------------------------------------------------------------------------- MegaPose
class PoseEstimator(_EstimatorBase):
    """MegaPose: detections -> coarse scoring over an SO(3) grid -> top-K -> refiner ->
    re-scoring -> top-1 (``MP/inference/pose_estimator.py:55-668``)."""

    def __init__(self, refiner_model=None, coarse_model=None, detector_model=None, depth_refiner=None,
                 bsz_objects: int = 8, bsz_images: int = 256, SO3_grid_size: int = 576) -> None:
        self._init_common(refiner_model, coarse_model, detector_model, bsz_objects, bsz_images)
        self.depth_refiner = depth_refiner
        if SO3_grid_size is not None:
            self.load_SO3_grid(SO3_grid_size)

    def load_SO3_grid(self, grid_size: int) -> None:
        self._SO3_grid = load_SO3_grid(grid_size).to(self.device)

    @torch.no_grad()
    def forward_refiner(self, observation: ObservationTensor, data_TCO_input: PoseEstimatesType,
                        n_iterations: int = 5, keep_all_outputs: bool = False, cuda_timer: bool = False,
                        **refiner_kwargs) -> Tuple[dict, dict]:
        assert self.refiner_model is not None
        return self._run_model_chunks(self.refiner_model, observation, data_TCO_input, n_iterations,
                                      self.bsz_objects, "refiner", keep_all_outputs, **refiner_kwargs)

    @torch.no_grad()
    def forward_scoring_model(self, observation: ObservationTensor, data_TCO: PoseEstimatesType,
                              cuda_timer: bool = False, return_debug_data: bool = False
                              ) -> Tuple[PoseEstimatesType, dict]:
        """Adds ``pose_logit`` / ``pose_score`` (coarse net on the refined poses), in place
        (``:222-325``)."""
        return self._guarded(self.coarse_model, lambda: self._forward_scoring_model_once(
            observation, data_TCO, cuda_timer, return_debug_data))

    def _forward_scoring_model_once(self, observation, data_TCO, cuda_timer, return_debug_data):
        t_start = time.time()
        assert self.coarse_model is not None
        df = data_TCO.infos
        N = len(df)
        # debug pixels stay on the rank that made them: a call that wants them is not sharded
        s, e, sharded = (0, N, False) if return_debug_data else self._shard(N)
        logits_l, scores_l, crops_l, renders_l = [], [], [], []
        render_time = model_time = 0.0
        n_batches = 0
        for a in range(s, e, self.bsz_images):
            ids = np.arange(a, min(e, a + self.bsz_images))
            chunk = data_TCO[ids]
            im_ids = torch.as_tensor(chunk.infos["batch_im_id"].values, device=self.device)
            out_ = self.coarse_model.forward_coarse(images=observation.images, K=observation.K,
                                                    labels=chunk.infos["label"].tolist(), TCO_input=chunk.poses,
                                                    cuda_timer=cuda_timer, return_debug_data=return_debug_data,
                                                    im_ids=im_ids)
            render_time += out_["render_time"]
            model_time += out_["model_time"]
            logits_l.append(out_["logits"])
            scores_l.append(out_["scores"])
            if return_debug_data:
                crops_l.append(out_["images_crop"])
                renders_l.append(out_["renders"])
            n_batches += 1
        f = dict(dtype=torch.float32, device=self.device)
        logits = torch.cat(logits_l) if logits_l else torch.zeros((0, 1), **f)
        scores = torch.cat(scores_l) if scores_l else torch.zeros((0, 1), **f)
        if sharded:  # one all-gather of [n_local, 2] rows; the coarse model has ONE logit per row (views_logits_head of a
            # one-view model, MP/models/pose_rigid.py:144-148), so the width does not depend on what a rank happens to hold
            full = D.gather_rows(torch.cat([logits.reshape(e - s, 1), scores.reshape(e - s, 1)], dim=1), s, N)
            logits, scores = full[:, 0:1], full[:, 1:2]  # [N, 1] like the unsharded concatenation
        debug_data = {"images_crop": torch.cat(crops_l), "renders": torch.cat(renders_l)} if return_debug_data else {}
        df["pose_logit"] = logits.cpu().numpy()
        df["pose_score"] = scores.cpu().numpy()
        elapsed = time.time() - t_start
        extra_data = {"render_time": render_time, "model_time": model_time, "time": elapsed, "logits": logits,
                      "scores": scores, "debug": debug_data, "n_batches": n_batches,
                      "timing_str": f"time: {elapsed:.2f}, model_time: {model_time:.2f}, render_time: {render_time:.2f}"}
        data_TCO.infos = df
        return data_TCO, extra_data

    @torch.no_grad()
    def forward_coarse_model(self, observation: ObservationTensor, detections: DetectionsType,
                             cuda_timer: bool = False, return_debug_data: bool = False
                             ) -> Tuple[PoseEstimatesType, dict]:
        """Every detection x every grid rotation -> ``TCO_init_from_boxes_autodepth_with_R``
        -> coarse logits (``:327-485``)."""
        return self._guarded(self.coarse_model, lambda: self._forward_coarse_model_once(
            observation, detections, cuda_timer, return_debug_data))

    def _forward_coarse_model_once(self, observation, detections, cuda_timer, return_debug_data):
        t_start = time.time()
        assert_detections_valid(detections)
        coarse_model = self.coarse_model
        B, M = len(detections), self._SO3_grid.shape[0]
        df = detections.infos
        rep = np.repeat(np.arange(B), M)
        df_hyp = df.iloc[rep].copy()
        df_hyp["hypothesis_id"] = np.tile(np.arange(M), B)
        df_hyp["bbox_id"] = df.index.values[rep]
        store = coarse_model.store
        bboxes_all = detections.bboxes.to(self.device, torch.float32)
        obj_all = store.ids_of(df_hyp["label"].tolist())
        im_all = torch.as_tensor(df_hyp["batch_im_id"].values, device=self.device, dtype=torch.int32)
        box_all = torch.as_tensor(rep, device=self.device, dtype=torch.int32)
        rot_all = torch.as_tensor(df_hyp["hypothesis_id"].values, device=self.device, dtype=torch.int32)
        logits_l, scores_l, TCO_l, crops_l, renders_l = [], [], [], [], []
        render_time = model_time = 0.0
        n_batches = 0
        labels_all = df_hyp["label"].tolist()
        # the (detection x grid rotation) rows are sharded over the ranks; one all-gather of [n_local, 18] rows
        # (logit, score, the 16 floats of TCO_init) merges them.  Debug pixels are not gathered: such a call runs whole.
        s0, e0, sharded = (0, B * M, False) if return_debug_data else self._shard(B * M)
        for s in range(s0, e0, self.bsz_images):
            e = min(e0, s + self.bsz_images)
            TCO_init_ = ops.tco_init_autodepth(store, bboxes_all, observation.K, im_all[s:e], obj_all[s:e],
                                               R=self._SO3_grid, box_ids=box_all[s:e], rot_ids=rot_all[s:e])
            out_ = coarse_model.forward_coarse(images=observation.images, K=observation.K, labels=labels_all[s:e],
                                               TCO_input=TCO_init_, cuda_timer=cuda_timer,
                                               return_debug_data=return_debug_data, im_ids=im_all[s:e])
            render_time += out_["render_time"]
            model_time += out_["model_time"]
            logits_l.append(out_["logits"])
            scores_l.append(out_["scores"])
            TCO_l.append(TCO_init_)
            if return_debug_data:
                crops_l.append(out_["images_crop"])
                renders_l.append(out_["renders"])
            n_batches += 1
        f = dict(dtype=torch.float32, device=self.device)
        n_loc = e0 - s0
        logits = torch.cat(logits_l).reshape(n_loc) if logits_l else torch.zeros(0, **f)
        scores = torch.cat(scores_l).reshape(n_loc) if scores_l else torch.zeros(0, **f)
        TCO = torch.cat(TCO_l) if TCO_l else torch.zeros((0, 4, 4), **f)
        if sharded:
            full = D.gather_rows(torch.cat([logits[:, None], scores[:, None], TCO.reshape(n_loc, 16)], dim=1), s0, B * M)
            logits, scores, TCO = full[:, 0].contiguous(), full[:, 1].contiguous(), full[:, 2:].reshape(B * M, 4, 4)
        logits, scores = logits.reshape(B, M), scores.reshape(B, M)
        debug_data = {}
        if return_debug_data:
            ic, rd = torch.cat(crops_l), torch.cat(renders_l)
            debug_data = {"images_crop": ic.reshape(B, M, -1, *ic.shape[-2:]),
                          "renders": rd.reshape(B, M, -1, *rd.shape[-2:])}
        df_hyp["coarse_logit"] = logits.flatten().cpu().numpy()
        df_hyp["coarse_score"] = scores.flatten().cpu().numpy()
        elapsed = time.time() - t_start
        extra_data = {"render_time": render_time, "model_time": model_time, "time": elapsed, "logits": logits,
                      "scores": scores, "TCO": TCO.reshape(B, M, 4, 4), "debug": debug_data, "n_batches": n_batches,
                      "timing_str": f"time: {elapsed:.2f}, model_time: {model_time:.2f}, render_time: {render_time:.2f}"}
        data_TCO = PandasTensorCollection(df_hyp, poses=TCO, bboxes=bboxes_all[box_all.long()])
        return data_TCO, extra_data

    @torch.no_grad()
    def forward_detection_model(self, observation: ObservationTensor, *args: Any, **kwargs: Any) -> DetectionsType:
        if self.detector_model is None:
            raise ValueError("no detector_model: pass `detections` (the detector is outside this path)")
        return self.detector_model.get_detections(observation, *args, **kwargs)

    def run_depth_refiner(self, observation, predictions):
        assert self.depth_refiner is not None, "You must specify a depth refiner"
        return self.depth_refiner.refine_poses(predictions, depth=observation.depth, K=observation.K)

    @torch.no_grad()
    def run_inference_pipeline(self, observation: ObservationTensor, detections: Optional[DetectionsType] = None,
                               run_detector: Optional[bool] = None, n_refiner_iterations: int = 5,
                               n_pose_hypotheses: int = 1, keep_all_refiner_outputs: bool = False,
                               run_depth_refiner: bool = False, bsz_images: Optional[int] = None,
                               bsz_objects: Optional[int] = None, cuda_timer: Optional[bool] = False,
                               coarse_estimates: Optional[PoseEstimatesType] = None,
                               labels_to_keep: Optional[List[str]] = None) -> Tuple[PoseEstimatesType, dict]:
        """1 detections -> 2 coarse -> 3 top-K -> 4 refine -> 5 score -> 6 top-1 [-> depth refiner]
        (``:515-668``)."""
        timing_str = ""
        t_start = time.time()
        if bsz_images is not None:
            self.bsz_images = bsz_images
        if bsz_objects is not None:
            self.bsz_objects = bsz_objects
        if coarse_estimates is None:
            assert detections is not None or run_detector, "You must either pass in `detections` or set run_detector=True"
            if detections is None and run_detector:
                t0 = time.time()
                detections = self.forward_detection_model(observation).to(self.device)
                timing_str += f"detection={time.time() - t0:.2f}, "
            if labels_to_keep is not None:
                detections = filter_detections(detections, labels_to_keep)
            assert len(detections) > 0, "TOFIX: currently, dealing with absence of detections is not supported"
            detections = add_instance_id(detections)
            data_TCO_coarse, coarse_extra_data = self.forward_coarse_model(observation, detections, cuda_timer=cuda_timer)
            timing_str += f"coarse={coarse_extra_data['time']:.2f}, "
            data_TCO_filtered = filter_top_pose_estimates(
                data_TCO_coarse, top_K=n_pose_hypotheses, group_cols=["batch_im_id", "label", "instance_id"],
                filter_field="coarse_logit")
        else:
            data_TCO_coarse, coarse_extra_data, data_TCO_filtered = coarse_estimates, None, coarse_estimates

        preds, refiner_extra_data = self.forward_refiner(observation, data_TCO_filtered, n_iterations=n_refiner_iterations,
                                                         keep_all_outputs=keep_all_refiner_outputs, cuda_timer=cuda_timer)
        data_TCO_refined = preds[f"iteration={n_refiner_iterations}"]
        timing_str += f"refiner={refiner_extra_data['time']:.2f}, "
        data_TCO_scored, scoring_extra_data = self.forward_scoring_model(observation, data_TCO_refined, cuda_timer=cuda_timer)
        timing_str += f"scoring={scoring_extra_data['time']:.2f}, "
        data_TCO_final_scored = filter_top_pose_estimates(
            data_TCO_scored, top_K=1, group_cols=["batch_im_id", "label", "instance_id"], filter_field="pose_logit")
        if run_depth_refiner:
            t0 = time.time()
            data_TCO_depth_refiner, _ = self.run_depth_refiner(observation, data_TCO_final_scored)
            data_TCO_final = data_TCO_depth_refiner
            timing_str += f"depth refiner={time.time() - t0:.2f}"
        else:
            data_TCO_depth_refiner, data_TCO_final = None, data_TCO_final_scored
        elapsed = time.time() - t_start
        extra_data: dict = {
            "coarse": {"preds": data_TCO_coarse, "data": coarse_extra_data},
            "coarse_filter": {"preds": data_TCO_filtered},
            "refiner_all_hypotheses": {"preds": preds, "data": refiner_extra_data},
            "scoring": {"preds": data_TCO_scored, "data": scoring_extra_data},
            "refiner": {"preds": data_TCO_final_scored, "data": refiner_extra_data},
            "timing_str": f"total={elapsed:.2f}, {timing_str}", "time": elapsed,
        }
        if run_depth_refiner:
            extra_data["depth_refiner"] = {"preds": data_TCO_depth_refiner}
        return data_TCO_final, extra_data


# -------------------------------------------------------------------------------- CosyPose
class CosyPoseEstimator(_EstimatorBase):
    """CosyPose: detections -> canonical init -> coarse iterations -> refiner iterations
    (``CP/integrated/pose_estimator.py:31-470``)."""

    def __init__(self, refiner_model=None, coarse_model=None, detector_model=None,
                 bsz_objects: int = 8, bsz_images: int = 256) -> None:
        self._init_common(refiner_model, coarse_model, detector_model, bsz_objects, bsz_images)

    def make_TCO_init(self, detections: DetectionsType, K: torch.Tensor) -> PoseEstimatesType:
        """``:125-134``: ``init_method == "z-up+auto-depth"`` -> canonical z-up orientation with
        depth from the box size over 2000 mesh points; else identity at z = 1."""
        model = self.coarse_model
        init_method = getattr(getattr(model, "cfg", None), "init_method", "v0")
        boxes = detections.bboxes.to(self.device, torch.float32)
        im_ids = torch.as_tensor(detections.infos["batch_im_id"].values, device=self.device, dtype=torch.int32)
        if init_method == "z-up+auto-depth":
            store = model.store
            TCO_init = ops.tco_init_autodepth(store, boxes, K, im_ids, store.ids_of(detections.infos["label"].tolist()),
                                              n_points=2000)
        else:
            Kd = K.to(self.device, torch.float32)[im_ids.long()]
            uv = (boxes[:, [0, 1]] + boxes[:, [2, 3]]) / 2
            z = torch.ones((len(boxes), 1), device=self.device)
            xy = ((uv - Kd[:, [0, 1], [2, 2]]) * z) / Kd[:, [0, 1], [0, 1]]
            TCO_init = torch.eye(4, device=self.device).repeat(len(boxes), 1, 1)
            TCO_init[:, :2, 3] = xy
            TCO_init[:, 2, 3] = z.flatten()
        return PandasTensorCollection(infos=detections.infos, poses=TCO_init)

    @torch.no_grad()
    def forward_coarse_model(self, observation, data_TCO_input, n_iterations: int = 5,
                             keep_all_outputs: bool = False, cuda_timer: bool = False) -> Tuple[dict, dict]:
        return self._run_model_chunks(self.coarse_model, observation, data_TCO_input, n_iterations,
                                      self.bsz_objects, "coarse", keep_all_outputs)

    @torch.no_grad()
    def forward_refiner(self, observation, data_TCO_input, n_iterations: int = 5,
                        keep_all_outputs: bool = False, cuda_timer: bool = False) -> Tuple[dict, dict]:
        return self._run_model_chunks(self.refiner_model, observation, data_TCO_input, n_iterations,
                                      self.bsz_objects, "refiner", keep_all_outputs)

    def forward_detection_model(self, observation, detection_th: float = 0.7, mask_th: float = 0.8, *a, **k):
        if self.detector_model is None:
            raise ValueError("no detector_model: pass `detections` (the detector is outside this path)")
        return self.detector_model.get_detections(observation=observation, one_instance_per_class=False,
                                                  detection_th=detection_th, output_masks=False, mask_th=mask_th)

    @torch.no_grad()
    def run_inference_pipeline(self, observation: ObservationTensor, detections: Optional[DetectionsType] = None,
                               data_TCO_init: Optional[PoseEstimatesType] = None, run_detector: Optional[bool] = None,
                               n_refiner_iterations: int = 1, n_coarse_iterations: int = 1,
                               bsz_images: Optional[int] = None, bsz_objects: Optional[int] = None,
                               coarse_estimates: Optional[PoseEstimatesType] = None, detection_th: float = 0.7,
                               mask_th: float = 0.8, labels_to_keep: Optional[List[str]] = None
                               ) -> Tuple[PoseEstimatesType, dict]:
        """``:136-229``.  (The reference raises on ``data_TCO_init`` because
        ``coarse_extra_data`` is unbound there; here that path works and
        reports ``coarse.data = None``.)"""
        timing_str = ""
        t_start = time.time()
        if bsz_images is not None:
            self.bsz_images = bsz_images
        if bsz_objects is not None:
            self.bsz_objects = bsz_objects
        if coarse_estimates is None and data_TCO_init is None:
            assert detections is not None or run_detector, "You must either pass in `detections` or set run_detector=True"
            if detections is None and run_detector:
                t0 = time.time()
                detections = self.forward_detection_model(observation, detection_th, mask_th)
                timing_str += f"detection={time.time() - t0:.2f}, "
        preds = {}
        coarse_extra_data = None
        if data_TCO_init is None:
            assert detections is not None
            assert self.coarse_model is not None
            assert n_coarse_iterations > 0
            if labels_to_keep is not None:
                detections = filter_detections(detections, labels_to_keep)
            data_TCO_init = self.make_TCO_init(detections, observation.K)
            coarse_preds, coarse_extra_data = self.forward_coarse_model(observation, data_TCO_init,
                                                                        n_iterations=n_coarse_iterations)
            for n in range(1, n_coarse_iterations + 1):
                preds[f"coarse/iteration={n}"] = coarse_preds[f"iteration={n}"]
            data_TCO_coarse = coarse_preds[f"iteration={n_coarse_iterations}"]
        else:
            assert n_coarse_iterations == 0
            preds["external_coarse"] = data_TCO_init
            data_TCO_coarse = data_TCO_init
        data_TCO, refiner_extra_data = data_TCO_coarse, None
        if n_refiner_iterations >= 1:
            assert self.refiner_model is not None
            refiner_preds, refiner_extra_data = self.forward_refiner(observation, data_TCO_coarse,
                                                                     n_iterations=n_refiner_iterations)
            for n in range(1, n_refiner_iterations + 1):
                preds[f"refiner/iteration={n}"] = refiner_preds[f"iteration={n}"]
            data_TCO = refiner_preds[f"iteration={n_refiner_iterations}"]
        elapsed = time.time() - t_start
        extra_data: dict = {
            "coarse": {"preds": data_TCO_coarse, "data": coarse_extra_data},
            "refiner_all_hypotheses": {"preds": preds, "data": refiner_extra_data},
            "refiner": {"preds": data_TCO, "data": refiner_extra_data},
            "timing_str": f"total={elapsed:.2f}, {timing_str}", "time": elapsed,
        }
        return data_TCO, extra_data
