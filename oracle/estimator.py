"""CPU restatement of the orchestrators: ``PoseEstimator.run_inference_pipeline``.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  An independent CPU run of the whole
pipeline (coarse SO(3)-grid scoring -> top-K -> refiner -> re-scoring -> top-1), assembled
from the oracle's own pieces (``OraclePredictor``, ``oracle.geometry``) in the order of

* MegaPose ``MP/inference/pose_estimator.py``: ``forward_coarse_model`` :327-485,
  ``forward_refiner`` :104-220, ``forward_scoring_model`` :222-325,
  ``run_inference_pipeline`` :515-668;
* CosyPose ``CP/integrated/pose_estimator.py``: ``run_inference_pipeline`` :136-229,
  initialisation :125-134;
* top-K filtering ``TB/utils/tensor_collection.py:201-230`` (pandas ``sort_values(desc)
  .groupby(group_cols).head(K)`` -- pandas is the reference's own dependency for this step, so
  tie behaviour and output order are pandas', exactly as there);
* instance ids ``TB/inference/utils.py:163-183``.

The tests compare the product's final poses, ``hypothesis_id``s and row order with this.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import pandas as pd

from . import geometry as G
from .pipeline import OraclePredictor


def add_instance_id(df: pd.DataFrame) -> pd.DataFrame:
    """``TB/inference/utils.py:163-183``."""
    if "instance_id" not in df:
        df = df.copy()
        df["instance_id"] = df.groupby(["batch_im_id", "label"]).cumcount()
    return df


def filter_top(df: pd.DataFrame, top_K: int, field: str,
               group_cols: Sequence[str] = ("batch_im_id", "label", "instance_id")) -> np.ndarray:
    """Row numbers kept by ``filter_top_pose_estimates`` (``TB/utils/tensor_collection.py:201-230``),
    in the reference's output order (= the sorted order)."""
    d = df.reset_index(drop=True)
    d = d.sort_values(field, ascending=False).groupby(list(group_cols)).head(top_K)
    return d.index.values


class OracleEstimator:
    """MegaPose ``PoseEstimator`` on the CPU.  ``coarse`` / ``refiner`` are :class:`OraclePredictor` s over the same
    packed meshes; ``labels`` maps object ids to label strings (the mesh store's order)."""

    def __init__(self, refiner: OraclePredictor, coarse: OraclePredictor, labels: List[str], SO3_grid_size: int = 576,
                 bsz_objects: int = 8, bsz_images: int = 256):
        self.refiner, self.coarse = refiner, coarse
        self.labels = list(labels)
        self.label_to_id = {l: i for i, l in enumerate(self.labels)}
        self.grid = G.load_SO3_grid(SO3_grid_size)
        self.bsz_objects, self.bsz_images = bsz_objects, bsz_images

    # MP/inference/pose_estimator.py:327-485
    def forward_coarse_model(self, images, K, det: pd.DataFrame, bboxes: np.ndarray):
        B, M = len(det), len(self.grid)
        rep = np.repeat(np.arange(B), M)
        hyp = det.iloc[rep].copy()
        hyp["hypothesis_id"] = np.tile(np.arange(M), B)
        hyp["bbox_id"] = det.index.values[rep]
        hyp = hyp.reset_index(drop=True)
        obj = np.array([self.label_to_id[l] for l in hyp["label"]], np.int32)
        im = hyp["batch_im_id"].values.astype(np.int32)
        TCO = np.empty((B * M, 4, 4), np.float32)
        logits = np.empty(B * M, np.float32)
        pts_table = self.coarse.points
        for s in range(0, B * M, self.bsz_images):
            e = min(B * M, s + self.bsz_images)
            TCO[s:e] = G.TCO_init_from_boxes_autodepth_with_R(bboxes[rep[s:e]], pts_table[obj[s:e]], K[im[s:e]],
                                                              self.grid[hyp["hypothesis_id"].values[s:e]])
            logits[s:e] = self.coarse.forward_coarse(images, K, im[s:e], obj[s:e], TCO[s:e])["logits"].reshape(-1)
        hyp["coarse_logit"] = logits
        hyp["coarse_score"] = 1.0 / (1.0 + np.exp(-logits))
        return hyp, TCO

    # MP/inference/pose_estimator.py:222-325
    def forward_scoring_model(self, images, K, df: pd.DataFrame, poses: np.ndarray):
        obj = np.array([self.label_to_id[l] for l in df["label"]], np.int32)
        im = df["batch_im_id"].values.astype(np.int32)
        logits = np.empty(len(df), np.float32)
        for s in range(0, len(df), self.bsz_images):
            e = min(len(df), s + self.bsz_images)
            logits[s:e] = self.coarse.forward_coarse(images, K, im[s:e], obj[s:e], poses[s:e])["logits"].reshape(-1)
        df = df.copy()
        df["pose_logit"] = logits
        df["pose_score"] = 1.0 / (1.0 + np.exp(-logits))
        return df

    # MP/inference/pose_estimator.py:515-668
    def run_inference_pipeline(self, images, K, labels: List[str], bboxes, batch_im_id=None, n_refiner_iterations: int = 5,
                               n_pose_hypotheses: int = 1, instance_id=None) -> Dict[str, object]:
        images, K = np.asarray(images, np.float32), np.asarray(K, np.float32)
        bboxes = np.asarray(bboxes, np.float32)
        det = pd.DataFrame({"label": list(labels),
                            "batch_im_id": np.zeros(len(labels), int) if batch_im_id is None else np.asarray(batch_im_id)})
        if instance_id is not None:  # detections that already carry ids keep them (TB/inference/utils.py:176-177)
            det["instance_id"] = np.asarray(instance_id)
        det = add_instance_id(det)
        coarse_df, coarse_TCO = self.forward_coarse_model(images, K, det, bboxes)
        keep = filter_top(coarse_df, n_pose_hypotheses, "coarse_logit")
        filt_df, filt_TCO = coarse_df.iloc[keep].reset_index(drop=True), coarse_TCO[keep]
        obj = np.array([self.label_to_id[l] for l in filt_df["label"]], np.int32)
        im = filt_df["batch_im_id"].values.astype(np.int32)
        iters = self.refiner.forward(images, K, im, obj, filt_TCO, n_refiner_iterations, bsz_objects=self.bsz_objects)
        refined = iters[-1]["TCO_output"]
        scored_df = self.forward_scoring_model(images, K, filt_df, refined)
        best = filter_top(scored_df, 1, "pose_logit")
        return dict(coarse_df=coarse_df, coarse_TCO=coarse_TCO, filtered_df=filt_df, filtered_TCO=filt_TCO,
                    refiner_iterations=iters, scored_df=scored_df, refined_TCO=refined,
                    final_df=scored_df.iloc[best].reset_index(drop=True), final_TCO=refined[best])


class OracleCosyPoseEstimator:
    """CosyPose ``PoseEstimator`` on the CPU (``CP/integrated/pose_estimator.py:31-229``): canonical z-up
    initialisation with auto-depth over 2000 sub-sampled mesh points (:125-134), ``n_coarse_iterations`` of the coarse
    model, ``n_refiner_iterations`` of the refiner; no scoring stage."""

    def __init__(self, refiner: Optional[OraclePredictor], coarse: Optional[OraclePredictor], labels: List[str],
                 bsz_objects: int = 8):
        self.refiner, self.coarse = refiner, coarse
        self.labels = list(labels)
        self.label_to_id = {l: i for i, l in enumerate(self.labels)}
        self.bsz_objects = bsz_objects

    def make_TCO_init(self, K, labels, bboxes, batch_im_id):
        model = self.coarse if self.coarse is not None else self.refiner
        obj = np.array([self.label_to_id[l] for l in labels], np.int32)
        ids = G.sample_point_ids(model.points.shape[1], 2000)
        return G.TCO_init_from_boxes_zup_autodepth(np.asarray(bboxes, np.float32), model.points[obj][:, ids],
                                                   np.asarray(K, np.float32)[np.asarray(batch_im_id)])

    def run_inference_pipeline(self, images, K, labels, bboxes=None, batch_im_id=None, TCO_init=None,
                               n_coarse_iterations: int = 1, n_refiner_iterations: int = 1):
        images, K = np.asarray(images, np.float32), np.asarray(K, np.float32)
        im = np.zeros(len(labels), np.int32) if batch_im_id is None else np.asarray(batch_im_id, np.int32)
        obj = np.array([self.label_to_id[l] for l in labels], np.int32)
        out = {}
        if TCO_init is None:
            assert n_coarse_iterations > 0
            TCO = self.make_TCO_init(K, labels, bboxes, im)
            out["init"] = TCO
            its = self.coarse.forward(images, K, im, obj, TCO, n_coarse_iterations, bsz_objects=self.bsz_objects)
            out["coarse_iterations"] = its
            TCO = its[-1]["TCO_output"]
        else:
            assert n_coarse_iterations == 0
            TCO = np.asarray(TCO_init, np.float32)
        if n_refiner_iterations >= 1:
            its = self.refiner.forward(images, K, im, obj, TCO, n_refiner_iterations, bsz_objects=self.bsz_objects)
            out["refiner_iterations"] = its
            TCO = its[-1]["TCO_output"]
        out["final_TCO"] = TCO
        return out
