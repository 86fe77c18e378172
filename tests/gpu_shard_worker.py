"""One rank of tests/test_gpu_pipeline.py::test_estimators_shard_with_real_predictors_two_ranks: the estimators with the
REAL predictors (HIP kernels) in two processes that share cuda:0, torch.distributed over gloo (RCCL refuses two ranks on
one device): the sharded ``run_inference_pipeline`` of every rank against the same rank's unsharded run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from happypose_amd import distributed as D  # noqa: E402
from happypose_amd.models import create_model_pose, create_pose_model_cosypose, pose_model_param_shapes  # noqa: E402
from happypose_amd.pose_estimator import (CosyPoseEstimator, ObservationTensor, PoseEstimator,  # noqa: E402
                                          make_detections_from_object_data)
from happypose_amd.renderer import BatchRenderer  # noqa: E402
from happypose_amd.synthetic import make_object_dataset, make_scene, predictor_weights  # noqa: E402
from happypose_amd.tensor_collection import PandasTensorCollection  # noqa: E402

T_TOL, R_TOL, LOGIT_TOL = 2e-5, 1e-4, 5e-3


def pose_err(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    dt = float(np.linalg.norm(a[:, :3, 3] - b[:, :3, 3], axis=1).max())
    # geodesic angle through the chord (arccos of the trace has a floor of ~5e-4 rad on fp32 matrices)
    chord = np.linalg.norm(a[:, :3, :3] - b[:, :3, :3], axis=(1, 2))
    return dt, float((2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))).max())


def same_everywhere(t):
    """the tensor is bit-identical on every rank"""
    t = t.detach().float().cpu().contiguous()
    out = [torch.empty_like(t) for _ in range(D.get_world_size())]
    torch.distributed.all_gather(out, t)
    return all(torch.equal(o, out[0]) for o in out)


def main():
    rank, _, world = D.init_distributed("gloo")
    assert world == 2
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ds = make_object_dataset(3, seed=1, tex_size=256)
    renderer = BatchRenderer(ds, device=dev)
    store = renderer.store
    sc = make_scene(n_detections=3, n_hypotheses=5, n_objects=3, seed=2, with_depth=True)
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    obs = ObservationTensor(images, K)

    # ---- MegaPose: coarse grid (3 detections x 72 rotations = 216 rows), top-2, refiner (6 rows x 4 views), scoring
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True, depth_augmentation=False)
    wc = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 9, pose_dim=0, n_views_logits=1), seed=9, update_scale=1.0)
    wr = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 27), seed=2, update_scale=0.002)
    coarse = create_model_pose(ccfg, renderer, state_dict=wc, max_batch=72)
    refiner = create_model_pose(rcfg, renderer, state_dict=wr, max_batch=8)
    pts = store.mesh_db.points[sc["det_obj_ids"]].astype(np.float64)
    T = sc["TCO_det"].astype(np.float64)
    pc = np.einsum("nij,npj->npi", T[:, :3, :3], pts) + T[:, None, :3, 3]
    uv = np.einsum("ij,npj->npi", sc["K"][0].astype(np.float64), pc)
    uv = uv[..., :2] / uv[..., 2:]
    boxes = np.concatenate([uv.min(1), uv.max(1)], -1).astype(np.float32)
    det = make_detections_from_object_data([store.labels[i] for i in sc["det_obj_ids"]], boxes).to(dev)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, bsz_objects=8, bsz_images=72, SO3_grid_size=72)
    assert D.sharding_active(est.shard_hypotheses)
    f_sh, e_sh = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=2, n_pose_hypotheses=2)
    s0, s1 = e_sh["refiner_all_hypotheses"]["data"]["shard"]
    assert s1 - s0 == 3, (s0, s1)  # this rank refined its half of the 6 filtered hypotheses
    est.shard_hypotheses = False
    f_un, e_un = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=2, n_pose_hypotheses=2)
    cl_s, cl_u = e_sh["coarse"]["preds"].infos.coarse_logit.values, e_un["coarse"]["preds"].infos.coarse_logit.values
    np.testing.assert_allclose(cl_s, cl_u, rtol=0, atol=LOGIT_TOL)
    assert e_sh["coarse_filter"]["preds"].infos.hypothesis_id.tolist() == e_un["coarse_filter"]["preds"].infos.hypothesis_id.tolist()
    for n in (1, 2):
        dt, dr = pose_err(e_sh["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses, e_un["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses)
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
    assert list(f_sh.infos.columns) == list(f_un.infos.columns)
    assert f_sh.infos.label.tolist() == f_un.infos.label.tolist() and f_sh.infos.hypothesis_id.tolist() == f_un.infos.hypothesis_id.tolist()
    dt, dr = pose_err(f_sh.poses, f_un.poses)
    assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    # what the sharded run returns is the same on both ranks, bit for bit
    assert same_everywhere(f_sh.poses) and same_everywhere(torch.as_tensor(cl_s))
    assert same_everywhere(e_sh["refiner_all_hypotheses"]["preds"]["iteration=2"].poses)

    # ---- CosyPose: 15 externally generated hypotheses -> refiner on two lanes (the C2 / C4 shape, ragged shards 8 + 7)
    wz = predictor_weights(pose_model_param_shapes("resnet34", 6), seed=0, update_scale=0.002)
    cosy = create_pose_model_cosypose(dict(backbone_str="resnet34"), renderer, state_dict=wz, max_batch=16, n_lanes=2)
    N = len(sc["TCO_hyp"])
    import pandas as pd

    hyp = PandasTensorCollection(pd.DataFrame({"label": [store.labels[i] for i in sc["hyp_obj_ids"]], "batch_im_id": np.zeros(N, dtype=int),
                                               "instance_id": np.arange(N) // 5}),
                                 poses=torch.as_tensor(sc["TCO_hyp"], device=dev))
    cest = CosyPoseEstimator(refiner_model=cosy, coarse_model=None, bsz_objects=16)
    g_sh, x_sh = cest.run_inference_pipeline(obs, data_TCO_init=hyp, n_coarse_iterations=0, n_refiner_iterations=3)
    cest.shard_hypotheses = False
    g_un, x_un = cest.run_inference_pipeline(obs, data_TCO_init=hyp, n_coarse_iterations=0, n_refiner_iterations=3)
    assert g_sh.infos.label.tolist() == g_un.infos.label.tolist()
    dt, dr = pose_err(g_sh.poses, g_un.poses)
    assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    upd = pose_err(g_sh.poses, hyp.poses)
    assert upd[1] > 10 * R_TOL, upd  # the refiner moved the poses
    assert same_everywhere(g_sh.poses)
    torch.distributed.barrier()
    print(f"rank {rank} ok", flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
