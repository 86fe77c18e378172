"""End-to-end parity of the refinement path (predictors + run_inference_pipeline) against
the CPU oracle loop on the same seeded inputs.  Needs a real MI355X: ``pytest -m gpu``.

Stated tolerance on poses after n iterations: fp32 path,
translation error <= 1e-4 m and rotation geodesic <= 1e-3 rad.  The residual comes from
fp32 summation order in the conv stack and from the <= 0.05 % silhouette pixels whose
coverage may differ between the HIP rasteriser and the C oracle.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T_TOL, R_TOL = 1e-4, 1e-3


def _pose_err(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    R = A[:, :3, :3] @ np.swapaxes(B[:, :3, :3], 1, 2)
    ang = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
    return dt.max(), ang.max()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def world(dev):
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_object_dataset, make_scene

    ds = make_object_dataset(3, seed=1, tex_size=256)
    renderer = BatchRenderer(ds, device=dev)
    scene = make_scene(n_detections=3, n_hypotheses=4, n_objects=3, seed=2, with_depth=True)
    return dict(ds=ds, renderer=renderer, store=renderer.store, scene=scene)


def _weights(arch, n_in, pose=True, logits=0, seed=0, scale=0.002):
    from happypose_amd.synthetic import predictor_weights
    from oracle import backbones as ob

    shapes = ob.predictor_param_shapes(arch, n_in, pose_dim=9 if pose else 0, n_views_logits=logits)
    return predictor_weights(shapes, seed=seed, update_scale=scale)


def _labels(world, obj_ids):
    return [world["store"].labels[i] for i in obj_ids]


def test_cosypose_refiner_vs_oracle(dev, world):
    from happypose_amd.models import create_pose_model_cosypose
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    w = _weights("resnet18", 6, seed=1)
    model = create_pose_model_cosypose(dict(backbone_str="resnet18"), world["renderer"], state_dict=w, max_batch=16)
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    B = len(sc["TCO_hyp"])
    im_ids = torch.zeros(B, dtype=torch.int32)
    out = model.forward(images, K, _labels(world, sc["hyp_obj_ids"]), torch.as_tensor(sc["TCO_hyp"]), n_iterations=3,
                        im_ids=im_ids)
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="resnet18", cosypose=True)
    ref = ora.forward(sc["images"][:, :3], sc["K"], np.zeros(B, np.int32), sc["hyp_obj_ids"], sc["TCO_hyp"], 3)
    for n in range(3):
        o = out[f"iteration={n + 1}"]
        dt, dr = _pose_err(o.TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        np.testing.assert_allclose(o.boxes_crop.cpu().numpy(), ref[n]["boxes_crop"], rtol=1e-4, atol=5e-2)
        np.testing.assert_allclose(o.K_crop.cpu().numpy(), ref[n]["K_crop"], rtol=1e-4, atol=5e-2)
        assert o.TCO_input.shape == (B, 4, 4) and o.K.shape == (B, 3, 3)
    # the update is not a no-op
    assert _pose_err(out["iteration=3"].TCO_output.cpu().numpy(), sc["TCO_hyp"])[1] > 1e-3
    # reference calling convention: images/K already gathered per hypothesis
    out2 = model.forward(images.expand(B, -1, -1, -1).contiguous(), K.expand(B, -1, -1).contiguous(),
                         _labels(world, sc["hyp_obj_ids"]), torch.as_tensor(sc["TCO_hyp"]), n_iterations=1)
    assert torch.equal(out2["iteration=1"].TCO_output, out["iteration=1"].TCO_output)


def test_cosypose_efficientnet_refiner_vs_oracle(dev, world):
    """CosyPose with the EfficientNet-b3 backbone of its released checkpoints (SURVEY.md 8f-1,
    CP/training/pose_models_cfg.py:33-35): two refiner iterations against the CPU restatement."""
    from happypose_amd.models import create_pose_model_cosypose
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    w = _weights("efficientnet-b3", 6, seed=1, scale=0.01)
    model = create_pose_model_cosypose(dict(backbone_str="efficientnet-b3"), world["renderer"], state_dict=w, max_batch=8)
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    sel = np.arange(0, 12, 2)
    im_ids = torch.zeros(len(sel), dtype=torch.int32)
    out = model.forward(images, K, _labels(world, sc["hyp_obj_ids"][sel]), torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2,
                        im_ids=im_ids)
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="efficientnet-b3", cosypose=True)
    ref = ora.forward(sc["images"][:, :3], sc["K"], np.zeros(len(sel), np.int32), sc["hyp_obj_ids"][sel], sc["TCO_hyp"][sel], 2)
    for n in range(2):
        dt, dr = _pose_err(out[f"iteration={n + 1}"].TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
    assert _pose_err(out["iteration=2"].TCO_output.cpu().numpy(), sc["TCO_hyp"][sel])[1] > 1e-4


@pytest.mark.parametrize("rgbd", [False, True])
def test_megapose_refiner_vs_oracle(dev, world, rgbd):
    from happypose_amd.models import create_model_pose
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=rgbd, input_depth=rgbd, predict_pose_update=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    n_in = 32 if rgbd else 27
    w = _weights("vanilla_resnet34", n_in, seed=2)
    model = create_model_pose(cfg, world["renderer"], state_dict=w, max_batch=8)
    sel = np.arange(0, 12, 2)  # 6 hypotheses
    images = torch.as_tensor(sc["images"], device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    labels = _labels(world, sc["hyp_obj_ids"][sel])
    out = model.forward(images, K, labels, torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2,
                        im_ids=torch.zeros(len(sel), dtype=torch.int32))
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True, render_depth=rgbd,
                          input_depth=rgbd, depth_normalization_type="tCR_scale_clamp_center")
    ref = ora.forward(sc["images"], sc["K"], np.zeros(len(sel), np.int32), sc["hyp_obj_ids"][sel], sc["TCO_hyp"][sel], 2)
    for n in range(2):
        o = out[f"iteration={n + 1}"]
        dt, dr = _pose_err(o.TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        np.testing.assert_allclose(o.TCO_input.cpu().numpy(), ref[n]["TCO_input"], atol=2e-5)
        assert o.KV_crop.shape == (len(sel), 4, 3, 3) and o.TCV_O_input.shape == (len(sel), 4, 4, 4)


def test_megapose_network_input_vs_oracle(dev, world):
    """The assembled network input (crop + 4 views x rgb/normals/depth, depth normalised) is the
    tensor the reference builds with normalize_images + cat (MP/models/pose_rigid.py:624-629)."""
    from happypose_amd.models import create_model_pose
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=True, input_depth=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    w = _weights("vanilla_resnet34", 32, seed=2)
    model = create_model_pose(cfg, world["renderer"], state_dict=w, max_batch=4)
    model.keep_pixels = True
    sel = np.array([0, 5, 10])
    out = model.forward(torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev),
                        _labels(world, sc["hyp_obj_ids"][sel]), torch.as_tensor(sc["TCO_hyp"][sel]), 1,
                        im_ids=torch.zeros(3, dtype=torch.int32))["iteration=1"]
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True, render_depth=True,
                          input_depth=True, depth_normalization_type="tCR_scale_clamp_center")
    it = ora._iteration(sc["images"], np.repeat(sc["K"], 3, 0), np.zeros(3, np.int32), sc["hyp_obj_ids"][sel],
                        sc["TCO_hyp"][sel], heads=("pose",))
    x = torch.cat([out.images_crop, out.renders], 1).cpu().numpy()
    assert x.shape == it["x"].shape == (3, 32, 240, 320)
    # crop (+ depth norm): the crop boxes agree to ~1e-3 px (fp32 association order) and the frame is
    # white noise (|d image / d px| ~ 1), hence ~1e-4 on the interpolated values
    np.testing.assert_allclose(x[:, :3], it["x"][:, :3], rtol=0, atol=1e-3)
    dd = np.abs(x[:, 3] - it["x"][:, 3])
    assert (dd > 1e-3).mean() < 1e-3  # depth validity rule may flip on a few hole-border pixels
    diff = np.abs(x[:, 4:] - it["x"][:, 4:])
    assert (diff > 1.5 / 255).mean() < 2e-3  # renders: silhouette pixels only
    assert np.median(diff) == 0


def test_megapose_coarse_and_pipeline(dev, world):
    """forward_coarse logits vs oracle, then the whole run_inference_pipeline: structure of the
    reference's outputs + consistency with running the stages by hand."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import (ObservationTensor, PoseEstimator, make_detections_from_object_data)
    from oracle import geometry as G
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    wc = _weights("vanilla_resnet34", 9, pose=False, logits=1, seed=3, scale=0.05)
    wr = _weights("vanilla_resnet34", 27, seed=2)
    coarse = create_model_pose(ccfg, world["renderer"], state_dict=wc, max_batch=32)
    refiner = create_model_pose(rcfg, world["renderer"], state_dict=wr, max_batch=8)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    assert obs.is_valid()

    # coarse scoring vs oracle on a handful of poses
    sel = np.array([0, 4, 8, 9])
    oc = coarse.forward_coarse(obs.images, obs.K, _labels(world, sc["hyp_obj_ids"][sel]),
                               torch.as_tensor(sc["TCO_hyp"][sel]), im_ids=torch.zeros(4, dtype=torch.int32))
    ora = OraclePredictor(wc, store.packed, store.mesh_db.points, arch="vanilla_resnet34", render_normals=True)
    rc = ora.forward_coarse(sc["images"][:, :3], sc["K"], np.zeros(4, np.int32), sc["hyp_obj_ids"][sel], sc["TCO_hyp"][sel])
    np.testing.assert_allclose(oc["logits"].cpu().numpy(), rc["logits"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(oc["scores"].cpu().numpy(), rc["scores"], rtol=1e-3, atol=1e-3)

    # detections = boxes of the projected ground-truth objects
    pts = store.mesh_db.points[sc["det_obj_ids"]]
    boxes = G.boxes_from_uv(G.project_points(pts, np.repeat(sc["K"], 3, 0), sc["TCO_det"]))
    det = make_detections_from_object_data(_labels(world, sc["det_obj_ids"]), boxes)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, bsz_objects=8, bsz_images=64, SO3_grid_size=72)
    final, extra = est.run_inference_pipeline(obs, detections=det.to(dev), n_refiner_iterations=2, n_pose_hypotheses=2)
    assert len(final) == 3 and final.poses.shape == (3, 4, 4)
    for col in ("label", "batch_im_id", "instance_id", "hypothesis_id", "coarse_logit", "coarse_score",
                "refiner_batch_idx", "refiner_instance_idx", "pose_logit", "pose_score"):
        assert col in final.infos.columns, col
    for k in ("coarse", "coarse_filter", "refiner_all_hypotheses", "scoring", "refiner", "timing_str", "time"):
        assert k in extra
    assert len(extra["coarse"]["preds"]) == 3 * 72 and len(extra["coarse_filter"]["preds"]) == 6
    assert set(extra["refiner_all_hypotheses"]["preds"].keys()) == {"iteration=1", "iteration=2"}
    for t in ("poses", "poses_input", "K_crop", "K", "boxes_rend", "boxes_crop"):
        assert t in final.tensors
    # coarse init == oracle restatement of TCO_init_from_boxes_autodepth_with_R
    grid = G.load_SO3_grid(72)
    init = G.TCO_init_from_boxes_autodepth_with_R(np.repeat(boxes, 72, 0), np.repeat(pts, 72, 0),
                                                  np.repeat(sc["K"], 216, 0), np.tile(grid, (3, 1, 1)))
    np.testing.assert_allclose(extra["coarse"]["preds"].poses.cpu().numpy(), init, rtol=1e-5, atol=1e-6)
    # top-1 per instance by pose_logit
    scored = extra["scoring"]["preds"].infos
    best = scored.sort_values("pose_logit", ascending=False).groupby(["batch_im_id", "label", "instance_id"]).head(1)
    assert sorted(best.hypothesis_id.tolist()) == sorted(final.infos.hypothesis_id.tolist())
    # results do not depend on the chunk sizes
    final2, _ = est.run_inference_pipeline(obs, detections=det.to(dev), n_refiner_iterations=2, n_pose_hypotheses=2,
                                           bsz_images=17, bsz_objects=4)
    np.testing.assert_allclose(final2.poses.cpu().numpy(), final.poses.cpu().numpy(), atol=1e-6)
    with pytest.raises(AssertionError):  # zero detections unsupported, like the reference
        est.run_inference_pipeline(obs, detections=det[[]].to(dev))


def test_cosypose_pipeline(dev, world):
    from types import SimpleNamespace

    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor, make_detections_from_object_data
    from oracle import geometry as G

    sc, store = world["scene"], world["store"]
    w = _weights("resnet18", 6, seed=1)
    cfg = dict(backbone_str="resnet18", init_method="z-up+auto-depth")
    coarse = create_pose_model_cosypose(cfg, world["renderer"], state_dict=w, max_batch=16)
    refiner = create_pose_model_cosypose(cfg, world["renderer"], state_dict=_weights("resnet18", 6, seed=5), max_batch=16)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    pts = store.mesh_db.points[sc["det_obj_ids"]]
    boxes = G.boxes_from_uv(G.project_points(pts, np.repeat(sc["K"], 3, 0), sc["TCO_det"]))
    det = make_detections_from_object_data(_labels(world, sc["det_obj_ids"]), boxes).to(dev)
    est = CosyPoseEstimator(refiner_model=refiner, coarse_model=coarse)
    init = est.make_TCO_init(det, obs.K)
    ids = G.sample_point_ids(store.n_pad, 2000)
    ref_init = G.TCO_init_from_boxes_zup_autodepth(boxes, pts[:, ids], np.repeat(sc["K"], 3, 0))
    np.testing.assert_allclose(init.poses.cpu().numpy(), ref_init, rtol=1e-5, atol=1e-6)
    final, extra = est.run_inference_pipeline(obs, detections=det, n_coarse_iterations=1, n_refiner_iterations=2)
    assert len(final) == 3 and set(extra) >= {"coarse", "refiner_all_hypotheses", "refiner", "timing_str", "time"}
    assert "refiner/iteration=2" in extra["refiner_all_hypotheses"]["preds"]
    assert "coarse_batch_idx" in extra["coarse"]["preds"].infos.columns
    # externally generated hypotheses (16 per detection in the benchmark) go through data_TCO_init
    from happypose_amd.tensor_collection import PandasTensorCollection
    import pandas as pd

    hyp = PandasTensorCollection(pd.DataFrame({"label": _labels(world, sc["hyp_obj_ids"]), "batch_im_id": 0,
                                               "instance_id": sc["hyp_det_ids"]}),
                                 poses=torch.as_tensor(sc["TCO_hyp"], device=dev))
    f2, e2 = est.run_inference_pipeline(obs, data_TCO_init=hyp, n_coarse_iterations=0, n_refiner_iterations=2)
    assert len(f2) == 12 and e2["coarse"]["data"] is None
    direct = refiner.forward(obs.images, obs.K, _labels(world, sc["hyp_obj_ids"]), torch.as_tensor(sc["TCO_hyp"]),
                             n_iterations=2, im_ids=torch.zeros(12, dtype=torch.int32))
    np.testing.assert_allclose(f2.poses.cpu().numpy(), direct["iteration=2"].TCO_output.cpu().numpy(), atol=1e-6)


def test_load_pose_models_from_run_dirs(dev, world, tmp_path):
    """<models_root>/<run_id>/{config.yaml, checkpoint.pth.tar} -> predictors.  The
    checkpoints are written in the reference's layout (incl. the legacy key names) and the loaded
    models must behave exactly like ones built from the same weights directly."""
    from happypose_amd.load_model import load_pose_models
    from happypose_amd.models import create_model_pose

    sc = world["scene"]
    wc = _weights("vanilla_resnet34", 9, pose=False, logits=1, seed=3, scale=0.05)
    wr = _weights("vanilla_resnet34", 27, seed=2)
    legacy = {("backbone.backbone." + k[len("backbone."):] if k.startswith("backbone.") else
               k.replace("views_logits_head.", "backbone.head.0.")): torch.as_tensor(v) for k, v in wc.items()}
    for run, sd, cfg in (
        ("coarse-x", legacy, "!!python/object:argparse.Namespace\nbackbone_str: vanilla_resnet34\n"
                             "input_strategy: input=obs+one_render\nrender_normals: true\nrenderer: panda3d\n"),
        ("refiner-x", {k: torch.as_tensor(v) for k, v in wr.items()},
         "!!python/object:argparse.Namespace\nbackbone_str: vanilla_resnet34\nn_views: 4\n"
         "multiview_type: front_3views\nrender_normals: true\nrenderer: panda3d\n"),
    ):
        (tmp_path / run).mkdir()
        (tmp_path / run / "config.yaml").write_text(cfg)
        torch.save({"state_dict": sd, "epoch": 1}, tmp_path / run / "checkpoint.pth.tar")
    coarse, refiner, mesh_db = load_pose_models("coarse-x", "refiner-x", world["ds"], models_root=tmp_path, device=dev,
                                                max_batch=8)
    assert coarse.predict_rendered_views_logits and not coarse.predict_pose_update and refiner.n_rendered_views == 4
    assert len(mesh_db.batched().labels) == 3
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    sel = np.array([0, 5, 9])
    labels, T, ids = _labels(world, sc["hyp_obj_ids"][sel]), torch.as_tensor(sc["TCO_hyp"][sel]), torch.zeros(3, dtype=torch.int32)
    ref_c = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                                   predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False),
                              world["renderer"], state_dict=wc, max_batch=8)
    assert torch.equal(coarse.forward_coarse(images, K, labels, T, im_ids=ids)["logits"],
                       ref_c.forward_coarse(images, K, labels, T, im_ids=ids)["logits"])
    ref_r = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views",
                                   render_normals=True), world["renderer"], state_dict=wr, max_batch=8)
    a = refiner.forward(images, K, labels, T, n_iterations=1, im_ids=ids)["iteration=1"].TCO_output
    b = ref_r.forward(images, K, labels, T, n_iterations=1, im_ids=ids)["iteration=1"].TCO_output
    assert torch.equal(a, b)
    c2, r2, _ = load_pose_models(None, "refiner-x", world["ds"], models_root=tmp_path, device=dev, max_batch=8)
    assert c2 is None and r2 is not None


def test_pipeline_with_depth_refiner(dev, world):
    """run_depth_refiner=True (MP/inference/pose_estimator.py:404-410): the ICP refiner runs on the
    top-1 estimates and its output becomes the final prediction."""
    from happypose_amd.icp_refiner import ICPRefiner
    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator, make_detections_from_object_data
    from oracle import geometry as G

    sc, store = world["scene"], world["store"]
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    coarse = create_model_pose(ccfg, world["renderer"], state_dict=_weights("vanilla_resnet34", 9, pose=False, logits=1, seed=3, scale=0.05), max_batch=32)
    refiner = create_model_pose(rcfg, world["renderer"], state_dict=_weights("vanilla_resnet34", 27, seed=2), max_batch=8)
    obs = ObservationTensor(torch.as_tensor(sc["images"].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    assert obs.depth is not None
    pts = store.mesh_db.points[sc["det_obj_ids"]]
    boxes = G.boxes_from_uv(G.project_points(pts, np.repeat(sc["K"], 3, 0), sc["TCO_det"]))
    det = make_detections_from_object_data(_labels(world, sc["det_obj_ids"]), boxes)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, depth_refiner=ICPRefiner(store.mesh_db, world["renderer"]),
                        bsz_objects=8, bsz_images=64, SO3_grid_size=72)
    final, extra = est.run_inference_pipeline(obs, detections=det.to(dev), n_refiner_iterations=1, n_pose_hypotheses=1,
                                              run_depth_refiner=True)
    assert "depth_refiner" in extra and len(extra["depth_refiner"]["preds"]) == 3
    assert final.poses.shape == (3, 4, 4) and torch.isfinite(final.poses).all()
    # the synthetic depth is noise in [0.3, 0.7] m, unrelated to the objects: registrations are rejected
    # (or accepted with a finite pose); poses_input always holds the refiner's output
    assert torch.equal(final.poses_input.cpu(), extra["refiner"]["preds"].poses.cpu())
    with pytest.raises(AssertionError):
        PoseEstimator(refiner_model=refiner, coarse_model=coarse).run_inference_pipeline(
            obs, detections=det.to(dev), n_refiner_iterations=1, run_depth_refiner=True)


def test_two_lane_refiner_matches_single_lane(dev, world):
    """TwoLanePredictor: the halves of the hypothesis batch as two chains on two streams give the single-lane poses
    (same kernels on the same data; only the summation order inside K-sliced tiles can differ)."""
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.synthetic import make_scene

    renderer = world["renderer"]
    sc = make_scene(n_detections=6, n_hypotheses=8, n_objects=len(renderer.store.labels), seed=9)
    w = _weights("resnet18", 6, seed=3)
    labels = [renderer.store.labels[i] for i in sc["hyp_obj_ids"]]
    args = (torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev), labels, torch.as_tensor(sc["TCO_hyp"], device=dev))
    im_ids = torch.zeros(len(labels), dtype=torch.int32)
    outs = []
    for lanes in (1, 2):
        m = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=48, n_lanes=lanes)
        o = m.forward(*args, n_iterations=3, im_ids=im_ids)
        outs.append(o)
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert a.labels == b.labels
        for f in ("TCO_output", "TCO_input", "K_crop", "boxes_crop", "boxes_rend", "tCR"):
            assert getattr(a, f).shape == getattr(b, f).shape
        dT = (a.TCO_output - b.TCO_output).abs().max().item()
        assert dT < 1e-4, (k, dT)
        assert torch.allclose(a.network_outputs["pose"], b.network_outputs["pose"], atol=1e-3)
