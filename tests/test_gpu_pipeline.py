"""End-to-end parity of the refinement path (predictors + run_inference_pipeline) against
the CPU oracle loop on the same seeded inputs.  Needs a real MI355X: ``pytest -m gpu``.

Stated tolerance on poses after n iterations (SURVEY.md section 8d proposed 1e-4 m / 1e-3 rad; tightened in round 3 to
what the path measures, x5): fp32 path, translation error <= 2e-5 m and rotation geodesic <= 1e-4 rad at every
iteration, with the synthetic pose head at ``update_scale = 0.002`` (a relative feature error eps reaches the pose as
~eps x 3e-3 rad).  The residual comes from fp32 summation order in the conv stack and from the <= 0.05 % silhouette
pixels whose coverage may differ between the HIP rasteriser and the C oracle.  The HIGH-GAIN cases (``update_scale =
0.05``: every iteration moves a pose by ~0.1 rad, 1000x the tolerance of the low-gain cases) carry ``T_TOL_HI`` /
``R_TOL_HI``: there the legitimate residual is amplified 25x per iteration and fed back through the renders.  A 1 %
error injected into one conv layer's weights fails every one of these checks (``tools/parity_sharpness.py`` measures
both sides; numbers in DESIGN.md section 2).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T_TOL, R_TOL = 2e-5, 1e-4
T_TOL_HI, R_TOL_HI = 5e-4, 2.5e-3  # update_scale = 0.05 cases
T_MED, R_MED = 2e-6, 6e-6  # MEDIAN over >= 64 hypotheses, low-gain head (a 1 % weight error in one layer: >= 1.2e-5 rad)
T_MED_HI, R_MED_HI = 4e-5, 6e-5  # per iteration n: n x these, high-gain head (injected 1 %: >= 3.4e-4 rad at n = 1)
FEAT_TOL = 2e-4  # backbone features vs the CPU restatement, relative to max|ref| of the sample


def _pose_med(A, B):
    """Median over the hypotheses of (|dt|, geodesic angle): legitimate residuals are sparse outliers (a silhouette pixel
    flipped under a hypothesis), an arithmetic error in the network shifts EVERY hypothesis -- the median separates the two
    by 20-3000x where the maximum separates them by 3x (tools/parity_sharpness.py)."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    chord = np.linalg.norm(A[:, :3, :3] - B[:, :3, :3], axis=(1, 2))
    ang = 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
    return float(np.median(dt)), float(np.median(ang))


def _pose_err(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    # geodesic angle through the chord: ||R_A - R_B||_F = 2 sqrt(2) sin(theta / 2).  (arccos of the trace has a floor of
    # ~sqrt(2 * 1e-7) = 5e-4 rad on fp32 matrices -- it reported 8e-4 rad for poses that agree to 1e-6)
    chord = np.linalg.norm(A[:, :3, :3] - B[:, :3, :3], axis=(1, 2))
    ang = 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
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
        np.testing.assert_allclose(o.boxes_crop.cpu().numpy(), ref[n]["boxes_crop"], rtol=1e-5, atol=5e-3)
        np.testing.assert_allclose(o.K_crop.cpu().numpy(), ref[n]["K_crop"], rtol=1e-5, atol=5e-3)
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


# ---------------------------------------------------------------------------------------------------------------
# Parity at the BENCHMARKED sizes (BASELINE.json configs C2 / C3 / C5): the worlds are the ones bench.py times.
# ---------------------------------------------------------------------------------------------------------------
def _bench():
    import importlib
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parent.parent)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module("bench")


def test_c2_full_size_two_lanes_vs_oracle(dev):
    """C2 exactly as benchmarked: WideResNet-34, 8 detections x 16 hypotheses = 128, 5 iterations, two half-batch lanes
    (64-sample launches: the planner's tail K-slicing / half-CU slicing paths) against the CPU oracle run in the
    reference's chunks of bsz_objects = 8; T_TOL / R_TOL must hold at EVERY iteration."""
    from oracle.pipeline import OraclePredictor

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    assert B == 128
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    out = model.forward(images, K, labels, torch.as_tensor(scene["TCO_hyp"], device=dev), n_iterations=5, im_ids=im_ids)
    assert model.numerics_status() == 0
    torch.set_num_threads(bench.effective_cpu_count())
    ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="resnet34", cosypose=True)
    ref = ora.forward(scene["images"][:, :3], scene["K"], np.zeros(B, np.int32), scene["hyp_obj_ids"], scene["TCO_hyp"], 5,
                      bsz_objects=8)
    errs = []
    for n in range(5):
        got = out[f"iteration={n + 1}"].TCO_output.cpu().numpy()
        dt, dr = _pose_err(got, ref[n]["TCO_output"])
        errs.append((dt, dr) + _pose_med(got, ref[n]["TCO_output"]))
        assert dt <= T_TOL and dr <= R_TOL, (n, errs)
        assert errs[-1][2] <= T_MED and errs[-1][3] <= R_MED, (n, errs)
    # the refinement moved the poses by far more than the tolerance
    assert _pose_err(out["iteration=5"].TCO_output.cpu().numpy(), scene["TCO_hyp"])[1] > 10 * R_TOL


def test_c2_full_size_efficientnet_vs_oracle(dev):
    """The benchmark job with the backbone of the released CosyPose checkpoints (``bench.py --arch efficientnet-b3``; the
    ``c2_efficientnet_b3`` block of the default line): 128 hypotheses, two lanes of 64 -- the depthwise strip kernels, the
    fused MBConv fronts, the squeeze-excitation launches and the gated 1x1 projections at the sizes that are timed -- three
    iterations against the CPU oracle, T_TOL / R_TOL at every iteration."""
    from oracle.pipeline import OraclePredictor

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "efficientnet-b3", seed=0, workload="C2", n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    assert B == 128
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    out = model.forward(images, K, labels, torch.as_tensor(scene["TCO_hyp"], device=dev), n_iterations=3, im_ids=im_ids)
    assert model.numerics_status() == 0
    from happypose_amd import ops as _ops

    assert _ops.scratch_launches() == 0
    torch.set_num_threads(bench.effective_cpu_count())
    ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="efficientnet-b3", cosypose=True)
    ref = ora.forward(scene["images"][:, :3], scene["K"], np.zeros(B, np.int32), scene["hyp_obj_ids"], scene["TCO_hyp"], 3,
                      bsz_objects=8)
    for n in range(3):
        got = out[f"iteration={n + 1}"].TCO_output.cpu().numpy()
        dt, dr = _pose_err(got, ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        assert _pose_med(got, ref[n]["TCO_output"])[0] <= T_MED
    assert _pose_err(out["iteration=3"].TCO_output.cpu().numpy(), scene["TCO_hyp"])[1] > 10 * R_TOL


BATCH_DT, BATCH_DR = 5e-6, 2e-5  # the same hypothesis in different batches (stated in INTEGRATION.md, "Batch dependence")


@pytest.mark.parametrize("gain", [0.002, 0.05])
def test_same_hypothesis_in_different_batches(dev, gain):
    """A hypothesis' refined pose depends -- at round-off level -- on its batch-mates: the conv planner picks tiles and
    K-slices by batch size (a different fp32 summation order) and the split-fp16 kernels scale a layer's activations by ONE
    power of two per launch, taken from max|y| over the whole batch (DESIGN.md 4.1 "Activation scale").  Both are legitimate
    (the reference's cuDNN picks algorithms by batch size too) but must be BOUNDED: hypothesis 37 of C2 refined alone
    (batch 1, one lane), with 7 batch-mates and in the full 128-hypothesis two-lane batch, 5 iterations, low and high head
    gain -- final poses within BATCH_DT / BATCH_DR of one another (measured: a few 1e-7 m / 1e-6 rad), a quarter of T_TOL /
    R_TOL, the bound on HIP vs oracle.  Bitwise equality holds between two runs of the SAME batch (test_graph_replay_*,
    test_two_lane_megapose_is_reproducible)."""
    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=2, update_scale=gain)
    store = renderer.store
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    TCO = torch.as_tensor(scene["TCO_hyp"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    h = 37

    def refine(rows):
        rows = list(rows)
        out = model.forward(images, K, [labels[r] for r in rows], TCO[rows], n_iterations=5,
                            im_ids=torch.zeros(len(rows), dtype=torch.int32, device=dev))
        assert model.numerics_status() == 0
        return [out[f"iteration={n}"].TCO_output[rows.index(h)].cpu().numpy()[None] for n in range(1, 6)]

    alone = refine([h])
    eight = refine(range(32, 40))
    full = refine(range(128))
    again = refine(range(128))
    worst = [0.0, 0.0]
    for n in range(5):
        assert np.array_equal(full[n], again[n])                      # same batch: bitwise
        for a, b in ((alone, eight), (alone, full), (eight, full)):
            dt, dr = _pose_err(a[n], b[n])
            worst = [max(worst[0], dt), max(worst[1], dr)]
            assert dt <= BATCH_DT * (25 if gain > 0.01 else 1) and dr <= BATCH_DR * (25 if gain > 0.01 else 1), (gain, n, dt, dr)
    print(f"batch dependence at gain {gain}: worst dt {worst[0]:.2e} m, dR {worst[1]:.2e} rad")
    # the refinement itself moved the pose by orders of magnitude more
    assert _pose_err(full[4], scene["TCO_hyp"][h][None])[1] > 50 * BATCH_DR


def test_c3_full_size_vs_oracle(dev):
    """C3 as benchmarked: MegaPose RGB-D refiner, 64 hypotheses x 4 views x (rgb + normals + depth), ResNet-34 on 32
    channels, two lanes; all 5 iterations against the oracle (the 7x7 / 32-channel stem, 256 renders per iteration)."""
    from oracle.pipeline import OraclePredictor

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C3", n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    assert B == 64
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    out = model.forward(images, K, labels, torch.as_tensor(scene["TCO_hyp"], device=dev), n_iterations=5,
                        im_ids=torch.zeros(B, dtype=torch.int32, device=dev))
    assert model.numerics_status() == 0
    torch.set_num_threads(bench.effective_cpu_count())
    ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True, render_depth=True, input_depth=True,
                          depth_normalization_type="tCR_scale_clamp_center")
    ref = ora.forward(scene["images"], scene["K"], np.zeros(B, np.int32), scene["hyp_obj_ids"], scene["TCO_hyp"], 5, bsz_objects=8)
    errs = []
    for n in range(5):
        got = out[f"iteration={n + 1}"].TCO_output.cpu().numpy()
        errs.append(_pose_err(got, ref[n]["TCO_output"]) + _pose_med(got, ref[n]["TCO_output"]))
        assert errs[-1][0] <= T_TOL and errs[-1][1] <= R_TOL, (n, errs)
        assert errs[-1][2] <= T_MED and errs[-1][3] <= R_MED, (n, errs)


# C5 / coarse scoring: the logits of an object's 576 grid poses against the fp32 CPU oracle, in units of the ORACLE'S OWN SPREAD
# over those poses (their standard deviation; the bench world's head has update_scale 1.0 -- logits 5.5 .. 5.8, std 0.042 --,
# an absolute tolerance says nothing about a quantity whose whole range is a few tenths).  Bounds = the measured error of the
# healthy path with head-room (tools/probes/c5_parity_probe.py -> profiles/r06_c5_parity_probe.json: logits fp16 0.047 / fp32
# 0.0098 of the spread -- the fp32 figure is a handful of views with a flipped silhouette pixel, its rms is 0.0011 --, features of
# the fp16 plan 2.9e-4 of max|ref|); a network with ONE conv layer's weights x 1.01 must FAIL them (second half of the tests;
# measured 0.096 / 0.061 of the spread, features 1.7e-3).
C5_LOGIT_REL = {"f16": 0.07, "f32": 0.025}  # max |got - ref| / std(ref) over an object's 576 poses
C5_FEAT_TOL_F16 = 6e-4  # fp16 plan: max |feature - ref| / max|ref| per sample at batch 576


def _c5_reference(bench, scene, weights, store, n=576):
    from oracle.pipeline import OraclePredictor

    torch.set_num_threads(bench.effective_cpu_count())
    ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="vanilla_resnet34", render_normals=True)
    return np.concatenate([ora.forward_coarse(scene["images"][:, :3], scene["K"], np.zeros(64, np.int32), scene["hyp_obj_ids"][s:s + 64],
                                              scene["TCO_hyp"][s:s + 64])["logits"].reshape(-1) for s in range(0, n, 64)])


def _mutated(weights, factor=1.01):
    w_bad = dict(weights)
    w_bad["backbone.layer2.1.conv1.weight"] = (np.asarray(weights["backbone.layer2.1.conv1.weight"]) * factor).astype(np.float32)
    return w_bad


@pytest.mark.parametrize("precision", ["f16", "f32"])
def test_c5_coarse_scoring_vs_oracle(dev, precision):
    """C5 as benchmarked: coarse scoring of one object x the 576 SO(3)-grid poses (one chunk of the bench) -- crop and rasteriser
    writing the network input directly (fp16 records for the fp16 plan), the conv stack at its batch-576 tiles -- against the fp32
    CPU oracle: every logit within ``C5_LOGIT_REL`` of the oracle's spread, the same top-5 hypotheses; and the SAME check FAILS
    for a network with one layer's weights off by 1 % (a coarse net that returned its bias would be off by several spreads)."""
    from happypose_amd.models import create_model_pose

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C5", precision=precision, n_lanes=1)
    store = renderer.store
    sl = slice(0, 576)
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"][sl]]
    T = torch.as_tensor(scene["TCO_hyp"][sl], device=dev)
    im0 = torch.zeros(576, dtype=torch.int32, device=dev)
    got = model.forward_coarse(images, K, labels, T, im_ids=im0)["logits"].cpu().numpy().reshape(-1)
    ref = _c5_reference(bench, scene, weights, store)
    spread = float(ref.std())
    assert spread > 0.02 and np.ptp(ref) > 5 * spread, (spread, np.ptp(ref))  # the head discriminates between the grid poses
    err = float(np.abs(got - ref).max())
    assert err <= C5_LOGIT_REL[precision] * spread, (err, spread, err / spread)
    # ranking: identical top-5 set wherever the oracle separates rank 5 from rank 6 by more than twice the measured error
    order = np.argsort(-ref)
    margin = ref[order[4]] - ref[order[5]]
    top_got, top_ref = set(np.argsort(-got)[:5].tolist()), set(order[:5].tolist())
    if margin > 2 * err:
        assert top_got == top_ref, (sorted(top_got), sorted(top_ref), margin, err)
    else:  # a tie at the cut within the error: every selected hypothesis must be within it of the oracle's cut
        assert all(ref[i] >= ref[order[4]] - 2 * err for i in top_got), (sorted(top_got), sorted(top_ref), margin, err)
    assert len(top_got & top_ref) >= 4
    # the bound is sharp: one conv layer's weights x 1.01, in the HIP model only
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
               predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    bad = create_model_pose(cfg, renderer, state_dict=_mutated(weights), max_batch=576, precision=precision, n_lanes=1)
    err_bad = float(np.abs(bad.forward_coarse(images, K, labels, T, im_ids=im0)["logits"].cpu().numpy().reshape(-1) - ref).max())
    assert err_bad > C5_LOGIT_REL[precision] * spread, (err_bad, spread, err_bad / spread)


def test_backbone_features_at_benchmark_batch_f16_plan(dev):
    """The fp16 plan at ITS benchmarked batch (C5: 576 views x 9 channels, vanilla ResNet-34: ``conv3x3_pp<MODE_F16, ...>``,
    ``conv_stem7x7s2_pool_f16_pp``, ``conv_igemm_f16`` at their batch-576 tiles) against ``oracle/backbones.py`` (fp32) on the very
    same network input -- the records the product's crop + rasteriser wrote in fp16: the 512 pooled features of every 4th sample
    within ``C5_FEAT_TOL_F16`` x max|ref| per sample, and a network with one layer's weights off by 1 % outside it."""
    from happypose_amd import ops
    from oracle import backbones as ob

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C5", precision="f16", n_lanes=1)
    store = renderer.store
    sl = slice(0, 576)
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"][sl]]
    T = torch.as_tensor(scene["TCO_hyp"][sl], device=dev)
    lane = model.lanes[0] if hasattr(model, "lanes") else model
    im_ids, obj_ids = lane._ids(images, K, labels, torch.zeros(576, dtype=torch.int32, device=dev))
    _, x, _, _, _ = lane._one_pass(images, K, im_ids, obj_ids, T, n_img_channels=lane._n_img, multiview_type="TCO", normalize=True,
                                   render_normals=lane.render_normals, render_depth=lane.render_depth, depth_mode=lane._depth_mode,
                                   want_pose=False, want_logits=True)
    x = x.clone()
    assert x.dtype == torch.float16 and x.shape[0] == 576
    n_in = lane.backbone.n_inputs
    feats = lane.backbone.forward(x, want_pose=False, want_logits=False, want_features=True)[2].float().cpu().numpy()
    sub = np.arange(0, 576, 4)
    x_nchw = x[..., :n_in].float().permute(0, 3, 1, 2)[torch.as_tensor(sub, device=dev)].contiguous().cpu()
    torch.set_num_threads(bench.effective_cpu_count())
    wt = {k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()}
    with torch.no_grad():
        ref = torch.cat([ob.net_forward(x_nchw[i:i + 16], wt, "vanilla_resnet34", heads=("features",))["features"]
                         for i in range(0, len(sub), 16)]).numpy()
    scale = np.abs(ref).max(axis=1, keepdims=True)
    err = np.abs(feats[sub] - ref) / scale
    assert err.max() <= C5_FEAT_TOL_F16, (err.max(), int(sub[err.max(axis=1).argmax()]))
    bad = ops.Net("vanilla_resnet34", n_in, _mutated(weights), max_batch=576, device=dev, precision="f16")
    f_bad = bad.forward(x, want_pose=False, want_logits=False, want_features=True)[2].float().cpu().numpy()
    err_bad = (np.abs(f_bad[sub] - ref) / scale).max()
    assert err_bad > C5_FEAT_TOL_F16, (err_bad, err.max())


def test_run_inference_pipeline_vs_oracle_estimator(dev, world):
    """a-1: the whole MegaPose pipeline (coarse grid -> top-K -> refine -> score -> top-1) against an independent CPU
    run of the same pipeline (oracle/estimator.py, itself pinned by golden G10 to the reference's own orchestrator):
    coarse logits, the top-K ids and ORDER, refined poses, final ids / labels / row order / poses."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator, make_detections_from_object_data
    from oracle import geometry as G
    from oracle.estimator import OracleEstimator
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    # head weights at He scale spread the logits; seed 9 separates ranks 1 / 2 / 3 of every detection by > 2 LOGIT_TOL
    # (in the reference render state; tools: the oracle's coarse logits over seeds 5..16)
    wc = _weights("vanilla_resnet34", 9, pose=False, logits=1, seed=9, scale=1.0)
    wr = _weights("vanilla_resnet34", 27, seed=2)
    coarse = create_model_pose(ccfg, world["renderer"], state_dict=wc, max_batch=72)
    refiner = create_model_pose(rcfg, world["renderer"], state_dict=wr, max_batch=8)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    pts = store.mesh_db.points[sc["det_obj_ids"]]
    boxes = G.boxes_from_uv(G.project_points(pts, np.repeat(sc["K"], 3, 0), sc["TCO_det"]))
    labels = _labels(world, sc["det_obj_ids"])
    det = make_detections_from_object_data(labels, boxes)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, bsz_objects=8, bsz_images=72, SO3_grid_size=72)
    final, extra = est.run_inference_pipeline(obs, detections=det.to(dev), n_refiner_iterations=2, n_pose_hypotheses=2)

    oc = OraclePredictor(wc, store.packed, store.mesh_db.points, arch="vanilla_resnet34", render_normals=True)
    orf = OraclePredictor(wr, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True)
    ref = OracleEstimator(orf, oc, store.labels, SO3_grid_size=72, bsz_objects=8, bsz_images=72).run_inference_pipeline(
        sc["images"][:, :3], sc["K"], labels, boxes, n_refiner_iterations=2, n_pose_hypotheses=2, instance_id=np.arange(3))
    LOGIT_TOL = 5e-3
    cl = extra["coarse"]["preds"].infos.coarse_logit.values
    np.testing.assert_allclose(cl, ref["coarse_df"]["coarse_logit"].values, rtol=0, atol=LOGIT_TOL)
    np.testing.assert_allclose(extra["coarse"]["preds"].poses.cpu().numpy(), ref["coarse_TCO"], rtol=1e-5, atol=1e-6)
    # top-K: same ids in the same order unless the oracle's own margin at a rank boundary is inside the logit tolerance
    rl = ref["coarse_df"]["coarse_logit"].values.reshape(3, 72)
    srt = -np.sort(-rl, axis=1)
    decisive = bool((srt[:, 0] - srt[:, 1] > 2 * LOGIT_TOL).all() and (srt[:, 1] - srt[:, 2] > 2 * LOGIT_TOL).all())
    f = extra["coarse_filter"]["preds"]
    if decisive:
        assert f.infos.hypothesis_id.tolist() == ref["filtered_df"]["hypothesis_id"].tolist()
        assert f.infos.label.tolist() == ref["filtered_df"]["label"].tolist()
        for n in (1, 2):
            dt, dr = _pose_err(extra["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses.cpu().numpy(),
                               ref["refiner_iterations"][n - 1]["TCO_output"])
            assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        pl = extra["scoring"]["preds"].infos.pose_logit.values
        np.testing.assert_allclose(pl, ref["scored_df"]["pose_logit"].values, rtol=0, atol=LOGIT_TOL)
        sd = ref["scored_df"]
        gaps = sd.groupby(["batch_im_id", "label", "instance_id"])["pose_logit"].apply(lambda v: np.abs(np.diff(np.sort(v.values))).min())
        if (gaps > 2 * LOGIT_TOL).all():
            assert final.infos.hypothesis_id.tolist() == ref["final_df"]["hypothesis_id"].tolist()
            assert final.infos.label.tolist() == ref["final_df"]["label"].tolist()
            assert final.infos.instance_id.tolist() == ref["final_df"]["instance_id"].tolist()
            dt, dr = _pose_err(final.poses.cpu().numpy(), ref["final_TCO"])
            assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    assert decisive, "choose weights whose coarse logits separate the top ranks (test world)"


def test_run_inference_pipeline_f16_coarse_vs_oracle_estimator(dev, world):
    """The default of the multi-hypothesis models since round 6 (``load_model.default_coarse_precision``): coarse / scoring network
    on the fp16 plan, refiner in fp32, through the whole ``run_inference_pipeline`` -- against the fp32 CPU oracle estimator: coarse
    and scoring logits within ``C5_LOGIT_REL["f16"]`` of the oracle's spread per detection (measured 0.014 - 0.030 on this world's 72-pose
    grids), and wherever the fp16 scores pick the oracle's hypotheses (they must, where the oracle's margins
    exceed twice the measured error) the refined and final poses within the fp32 pose tolerances."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator, make_detections_from_object_data
    from oracle import geometry as G
    from oracle.estimator import OracleEstimator
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    wc = _weights("vanilla_resnet34", 9, pose=False, logits=1, seed=9, scale=1.0)
    wr = _weights("vanilla_resnet34", 27, seed=2)
    coarse = create_model_pose(ccfg, world["renderer"], state_dict=wc, max_batch=72, precision="f16")
    refiner = create_model_pose(rcfg, world["renderer"], state_dict=wr, max_batch=8)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    pts = store.mesh_db.points[sc["det_obj_ids"]]
    boxes = G.boxes_from_uv(G.project_points(pts, np.repeat(sc["K"], 3, 0), sc["TCO_det"]))
    labels = _labels(world, sc["det_obj_ids"])
    det = make_detections_from_object_data(labels, boxes)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, bsz_objects=8, bsz_images=72, SO3_grid_size=72)
    final, extra = est.run_inference_pipeline(obs, detections=det.to(dev), n_refiner_iterations=2, n_pose_hypotheses=2)

    oc = OraclePredictor(wc, store.packed, store.mesh_db.points, arch="vanilla_resnet34", render_normals=True)
    orf = OraclePredictor(wr, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True)
    ref = OracleEstimator(orf, oc, store.labels, SO3_grid_size=72, bsz_objects=8, bsz_images=72).run_inference_pipeline(
        sc["images"][:, :3], sc["K"], labels, boxes, n_refiner_iterations=2, n_pose_hypotheses=2, instance_id=np.arange(3))
    rl = ref["coarse_df"]["coarse_logit"].values.reshape(3, 72)
    cl = extra["coarse"]["preds"].infos.coarse_logit.values.reshape(3, 72)
    spread = rl.std(axis=1)
    err = np.abs(cl - rl).max(axis=1)
    assert (err <= C5_LOGIT_REL["f16"] * spread).all(), (err, spread, err / spread)
    srt = -np.sort(-rl, axis=1)
    decisive = bool((srt[:, 0] - srt[:, 1] > 2 * err).all() and (srt[:, 1] - srt[:, 2] > 2 * err).all())
    f = extra["coarse_filter"]["preds"]
    print("f16 coarse pipeline: logit error / spread per detection", (err / spread).round(4).tolist(), "spread", spread.round(4).tolist(), "decisive", decisive,
          "same top-2 as the oracle", f.infos.hypothesis_id.tolist() == ref["filtered_df"]["hypothesis_id"].tolist(),
          "same final", final.infos.hypothesis_id.tolist() == ref["final_df"]["hypothesis_id"].tolist())
    if decisive:  # the oracle's top two of every detection stand clear of the fp16 error: same hypotheses, same order
        assert f.infos.hypothesis_id.tolist() == ref["filtered_df"]["hypothesis_id"].tolist()
    if f.infos.hypothesis_id.tolist() == ref["filtered_df"]["hypothesis_id"].tolist():
        for n in (1, 2):  # the refiner is fp32: its poses meet the fp32 tolerances
            dt, dr = _pose_err(extra["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses.cpu().numpy(),
                               ref["refiner_iterations"][n - 1]["TCO_output"])
            assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        pl, rp = extra["scoring"]["preds"].infos.pose_logit.values, ref["scored_df"]["pose_logit"].values
        assert np.abs(pl - rp).max() <= C5_LOGIT_REL["f16"] * spread.max(), (np.abs(pl - rp).max(), spread)
        if final.infos.hypothesis_id.tolist() == ref["final_df"]["hypothesis_id"].tolist():
            dt, dr = _pose_err(final.poses.cpu().numpy(), ref["final_TCO"])
            assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    assert final.infos.label.tolist() == ref["final_df"]["label"].tolist()


def test_run_inference_pipeline_vs_reference_golden_g10(dev, golden_dir):
    """The product's run_inference_pipeline against outputs of the REFERENCE's own PoseEstimator.run_inference_pipeline
    (golden G10, tools/gen_golden_loop.py: the reference's orchestrator, loop and backbones executing, with roi_align /
    render / lookAt supplied by the oracle): coarse logits, top-K ids and order, refined poses, final ids and poses."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import gen_golden_loop as ggl

    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.tensor_collection import PandasTensorCollection
    import pandas as pd

    g = np.load(golden_dir / "g10_loop.npz")
    ds, packed, mesh_db, sc = ggl.world()
    renderer = BatchRenderer(ds, device=dev)
    labels_all = list(renderer.store.labels)
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                depth_augmentation=False)
    coarse = create_model_pose(ccfg, renderer, state_dict=ggl.case_weights("coarse"), max_batch=64)
    refiner = create_model_pose(rcfg, renderer, state_dict=ggl.case_weights("mp_rgb4"), max_batch=8)
    det_ids = g["e2e/det_ids"]
    infos = pd.DataFrame({"label": [labels_all[i] for i in sc["det_obj_ids"][det_ids]], "batch_im_id": 0, "instance_id": np.arange(2)})
    det = PandasTensorCollection(infos=infos, bboxes=torch.as_tensor(g["e2e/boxes"]).float()).to(dev)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, bsz_objects=8, bsz_images=64, SO3_grid_size=72)
    final, extra = est.run_inference_pipeline(obs, detections=det, n_refiner_iterations=2, n_pose_hypotheses=2)
    assert sorted(extra.keys()) == [str(k) for k in g["e2e/extra_keys"]]
    assert sorted(final.tensors.keys()) == [str(k) for k in g["e2e/final_tensors"]]
    assert set(str(c) for c in g["e2e/final_columns"]) <= set(final.infos.columns)
    np.testing.assert_allclose(extra["coarse"]["preds"].poses.cpu().numpy(), g["e2e/coarse_TCO"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(extra["coarse"]["data"]["logits"].cpu().numpy(), g["e2e/coarse_logits"], rtol=0, atol=5e-3)
    f = extra["coarse_filter"]["preds"]
    assert f.infos.hypothesis_id.tolist() == g["e2e/filtered_hyp"].tolist()
    assert f.infos.label.tolist() == [str(l) for l in g["e2e/filtered_label"]]
    for n in (1, 2):
        dt, dr = _pose_err(extra["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses.cpu().numpy(), g[f"e2e/refined_it{n}"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
    np.testing.assert_allclose(extra["scoring"]["preds"].infos.pose_logit.values, g["e2e/pose_logit"], rtol=0, atol=5e-3)
    assert final.infos.hypothesis_id.tolist() == g["e2e/final_hyp"].tolist()
    assert final.infos.label.tolist() == [str(l) for l in g["e2e/final_label"]]
    assert final.infos.instance_id.tolist() == g["e2e/final_instance"].tolist()
    dt, dr = _pose_err(final.poses.cpu().numpy(), g["e2e/final_TCO"])
    assert dt <= T_TOL and dr <= R_TOL, (dt, dr)


def test_refiners_vs_reference_golden_g10(dev, golden_dir):
    """The predictors against outputs of the REFERENCE's own PosePredictor.forward (MegaPose 1 view, MegaPose RGB-D
    4 views, CosyPose; golden G10) at every recorded iteration."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import gen_golden_loop as ggl

    from happypose_amd.models import create_model_pose, create_pose_model_cosypose
    from happypose_amd.renderer import BatchRenderer

    g = np.load(golden_dir / "g10_loop.npz")
    ds, packed, mesh_db, sc = ggl.world()
    renderer = BatchRenderer(ds, device=dev)
    labels_all = list(renderer.store.labels)
    images, K = torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev)

    def check(tag, out, n_it):
        for n in range(1, n_it + 1):
            o = out[f"iteration={n}"]
            dt, dr = _pose_err(o.TCO_output.cpu().numpy(), g[f"{tag}/it{n}/TCO_output"])
            assert dt <= T_TOL and dr <= R_TOL, (tag, n, dt, dr)
            np.testing.assert_allclose(o.boxes_crop.cpu().numpy(), g[f"{tag}/it{n}/boxes_crop"], rtol=1e-5, atol=5e-3)
            np.testing.assert_allclose(o.K_crop.cpu().numpy(), g[f"{tag}/it{n}/K_crop"], rtol=1e-5, atol=5e-3)
            np.testing.assert_allclose(o.boxes_rend.cpu().numpy(), g[f"{tag}/it{n}/boxes_rend"], rtol=1e-5, atol=5e-3)

    sel = g["cosy/sel"]
    lab = [labels_all[i] for i in sc["hyp_obj_ids"][sel]]
    ids = torch.zeros(len(sel), dtype=torch.int32)
    cosy = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=ggl.case_weights("cosy"), max_batch=8)
    check("cosy", cosy.forward(images[:, :3].contiguous(), K, lab, torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2, im_ids=ids), 2)
    m1 = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                                depth_augmentation=False), renderer, state_dict=ggl.case_weights("mp_rgb1"), max_batch=8)
    check("mp_rgb1", m1.forward(images, K, lab, torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2, im_ids=ids), 2)
    sel4 = g["mp_rgbd4/sel"]
    lab4 = [labels_all[i] for i in sc["hyp_obj_ids"][sel4]]
    m4 = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                                render_depth=True, input_depth=True, depth_augmentation=False,
                                depth_normalization_type="tCR_scale_clamp_center"), renderer, state_dict=ggl.case_weights("mp_rgbd4"),
                           max_batch=4)
    out4 = m4.forward(images, K, lab4, torch.as_tensor(sc["TCO_hyp"][sel4]), n_iterations=2, im_ids=torch.zeros(3, dtype=torch.int32))
    check("mp_rgbd4", out4, 2)
    np.testing.assert_allclose(out4["iteration=1"].KV_crop.cpu().numpy(), g["mp_rgbd4/it1/KV_crop"], rtol=1e-5, atol=5e-3)
    np.testing.assert_allclose(out4["iteration=1"].TCV_O_input.cpu().numpy(), g["mp_rgbd4/it1/TCV_O_input"], rtol=0, atol=2e-5)


def test_split_fp16_overflow_guard(dev):
    """An activation beyond the fp16 range (7e4) turns into inf inside the default split-fp16 conv kernels.  The guard
    must (1) flag it (HP_STATUS_NONFINITE), (2) switch the network to the exact-fp32 kernels, so that (3) the SAME
    call repeated gives the finite fp32 result of the reference's arithmetic."""
    from happypose_amd import ops
    from oracle import backbones as ob

    w = _weights("resnet18", 6, seed=3, scale=0.05)
    net = ops.Net("resnet18", 6, w, max_batch=2, device=dev)
    x = np.random.RandomState(1).uniform(0, 1, size=(2, 6, 240, 320)).astype(np.float32)
    xin = net.new_input(2)
    xin[..., :6] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    pose0, _, _ = net.forward(xin)
    assert net.status() == 0 and torch.isfinite(pose0).all()
    x[1, 2, 100:140, 100:160] = 7.0e4  # raw depth in a wrong unit, a saturated sensor ...
    xin[..., :6] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    net.forward(xin)
    flags = net.status()
    assert flags & ops.STATUS_NONFINITE and flags & ops.STATUS_EXACT_ONLY, flags
    pose, _, _ = net.forward(xin)  # repeated: exact-fp32 kernels now
    assert net.status() == ops.STATUS_EXACT_ONLY  # nothing flagged any more, the switch is sticky
    with torch.no_grad():
        ref = ob.net_forward(torch.as_tensor(x), w, "resnet18", heads=("pose",))["pose"].numpy()
    assert np.isfinite(ref).all()
    np.testing.assert_allclose(pose.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    # the un-poisoned sample is the same as before the switch, to the kernels' tolerance
    np.testing.assert_allclose(pose[0].cpu().numpy(), pose0[0].cpu().numpy(), rtol=1e-3, atol=1e-3)
    # a per-network choice of kernels does not leak into other networks
    net2 = ops.Net("resnet18", 6, w, max_batch=2, device=dev)
    net2.set_conv_algo("winograd")
    net3 = ops.Net("resnet18", 6, w, max_batch=2, device=dev)
    xin[..., :6] = torch.as_tensor(np.random.RandomState(1).uniform(0, 1, size=(2, 6, 240, 320)).astype(np.float32), device=dev).permute(0, 2, 3, 1)
    p2, p3 = net2.forward(xin)[0], net3.forward(xin)[0]
    assert net2.status() == 0 and net3.status() == 0
    np.testing.assert_allclose(p2.cpu().numpy(), p3.cpu().numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("lanes", [2, 3])
def test_small_estimator_chunks_run_side_by_side(dev, world, lanes):
    """``bsz_objects`` chunks below the lanes' split threshold (the reference's default of 8) run as whole chains on different
    lanes at the same time (``forward_chunks``) instead of one after the other on lane 0: same launches per chunk, so the
    stage's table is bit-identical to the single-lane estimator's, whichever lane ran which chunk; a ragged last chunk and
    more chunks than lanes included."""
    import pandas as pd

    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator
    from happypose_amd.synthetic import make_scene
    from happypose_amd.tensor_collection import PandasTensorCollection

    renderer = world["renderer"]
    sc = make_scene(n_detections=7, n_hypotheses=5, n_objects=len(renderer.store.labels), seed=31, with_depth=True)
    w = _weights("vanilla_resnet34", 32, seed=4)
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
    B = len(labels)  # 35 hypotheses: chunks of 8, 8, 8, 8, 3
    infos = pd.DataFrame({"label": labels, "batch_im_id": np.zeros(B, dtype=np.int64), "instance_id": np.arange(B) // 5,
                          "hypothesis_id": np.arange(B) % 5})
    obs = ObservationTensor(torch.as_tensor(sc["images"].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    T0 = torch.as_tensor(sc["TCO_hyp"], device=dev)

    def run(model):
        est = PoseEstimator(refiner_model=model, coarse_model=model, bsz_objects=8)
        preds, extra = est.forward_refiner(obs, PandasTensorCollection(infos=infos.copy(), poses=T0.clone()), n_iterations=3)
        return preds

    single = create_model_pose(cfg, renderer, state_dict=w, max_batch=32)
    multi = create_model_pose(cfg, renderer, state_dict=w, max_batch=32 * lanes, n_lanes=lanes)
    calls = []
    for lane in multi.lanes:
        lane.forward = (lambda f, lane=lane: lambda *a, **k: (calls.append(multi.lanes.index(lane)), f(*a, **k))[1])(lane.forward)
    ref, got = run(single), run(multi)
    assert calls == [c % lanes for c in range(5)]
    for k in ref:
        for name in ("poses", "poses_input", "K_crop", "boxes_rend", "boxes_crop"):
            assert torch.equal(getattr(ref[k], name), getattr(got[k], name)), (k, name)
        assert ref[k].infos.equals(got[k].infos)
    assert torch.isfinite(got["iteration=3"].poses).all() and not torch.equal(got["iteration=3"].poses, T0)
    # ... and with hipGraph replay per lane: the first stage runs eagerly, the second captures (two signatures per lane: the
    # full chunk and the ragged one), the third replays -- each lane's capture happens on ITS stream while the others hold work
    graphed = create_model_pose(cfg, renderer, state_dict=w, max_batch=32 * lanes, n_lanes=lanes, graphs=True)
    for _ in range(3):
        rep = run(graphed)
        for k in ref:
            assert torch.equal(ref[k].poses, rep[k].poses), k
    assert any(l._graphs is not None and l._graphs.replays > 0 for l in graphed.lanes)


def test_guard_fires_inside_a_chunked_graph_run(dev, world):
    """The numerical guard in the first chunk of a multi-chunk ``forward_refiner`` whose chunks replay / capture hipGraphs
    (advisor r4): the chunks are enqueued back to back, so the guard word of chunk 1 turns non-zero while a later chunk of the
    same signature may be CAPTURING -- the exact-fp32 weight sets are built lazily with hipMalloc, which a capture
    forbids, so the asynchronous flip must not be adopted there (net.cpp: hipStreamIsCapturing).  The stage must end without
    an exception, the estimator must see the flag and repeat it, and the repeated stage -- exact-fp32 kernels -- must give
    the finite poses an all-exact model computes."""
    import pandas as pd

    from happypose_amd import ops
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor
    from happypose_amd.synthetic import make_scene
    from happypose_amd.tensor_collection import PandasTensorCollection

    renderer = world["renderer"]
    sc = make_scene(n_detections=6, n_hypotheses=4, n_objects=len(renderer.store.labels), seed=21, with_depth=True)
    w = _weights("resnet18", 6, seed=3, scale=0.05)
    labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
    B = len(labels)
    K = torch.as_tensor(sc["K"], device=dev)
    T0 = torch.as_tensor(sc["TCO_hyp"], device=dev)
    infos = pd.DataFrame({"label": labels, "batch_im_id": np.zeros(B, dtype=np.int64), "instance_id": np.arange(B) // 4,
                          "hypothesis_id": np.arange(B) % 4})
    clean = sc["images"][:, :3].copy()
    hot = clean.copy()
    hot[:, :, 150:330, 200:440] = 7.0e4  # a saturated patch under every crop: fp16 overflow inside the split kernels

    def run(model, images, bsz=8):
        est = CosyPoseEstimator(refiner_model=model, coarse_model=model, bsz_objects=bsz)
        obs = ObservationTensor(torch.as_tensor(images, device=dev), K)
        preds, extra = est.forward_refiner(obs, PandasTensorCollection(infos=infos.copy(), poses=T0.clone()), n_iterations=2)
        return preds["iteration=2"].poses

    graphed = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=8, graphs=True)
    for _ in range(2):  # three chunks of 8 per stage: call 1 of the signature is eager, later ones capture / replay
        p_clean = run(graphed, clean)
    assert graphed.numerics_status() == 0 and torch.isfinite(p_clean).all()
    p_hot = run(graphed, hot)  # the guard fires in chunk 1 while chunks 2, 3 replay; the estimator repeats the stage
    assert torch.isfinite(p_hot).all()
    assert graphed.numerics_status() == ops.STATUS_EXACT_ONLY  # switched, nothing pending
    exact = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=8)
    exact.backbone.force_exact(True)
    p_ref = run(exact, hot)
    assert torch.isfinite(p_ref).all()
    np.testing.assert_allclose(p_hot.cpu().numpy(), p_ref.cpu().numpy(), rtol=0, atol=1e-6)
    # and the model keeps working (exact kernels, graphs re-captured under the new epoch)
    p_again = run(graphed, clean)
    np.testing.assert_allclose(p_again.cpu().numpy(), p_clean.cpu().numpy(), rtol=0, atol=2e-4)


def test_index_guards(dev, world):
    """Ids that index frames / intrinsics / objects: host-resident ids raise like the reference's indexing, device-resident
    ids are guarded by the kernels (NaN poses, zero crops) -- never an out-of-bounds read."""
    from happypose_amd import ops

    sc, store = world["scene"], world["store"]
    K = torch.as_tensor(sc["K"], device=dev)
    T = torch.as_tensor(sc["TCO_hyp"][:4], device=dev)
    obj = torch.as_tensor(sc["hyp_obj_ids"][:4])
    with pytest.raises(IndexError):
        ops.pose_prep(store, T, K, torch.tensor([0, 0, 1, 0], dtype=torch.int32), obj, (480, 640))
    with pytest.raises(IndexError):
        ops.pose_prep(store, T, K, torch.zeros(4, dtype=torch.int32), torch.tensor([0, 1, 99, 0]), (480, 640))
    out = ops.pose_prep(store, T, K, torch.tensor([0, 0, 5, 0], dtype=torch.int32, device=dev), obj.to(dev), (480, 640))
    bc = out["boxes_crop"].cpu().numpy()
    assert np.isfinite(bc[[0, 1, 3]]).all() and np.isnan(bc[2]).all() and np.isnan(out["TCO"][2].cpu().numpy()).all()
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    boxes = torch.tensor([[100.0, 100, 300, 250]] * 2, device=dev)
    with pytest.raises(IndexError):
        ops.crop_roi_align(images, boxes, torch.tensor([0, 3], dtype=torch.int32))
    crops = ops.crop_roi_align(images, boxes, torch.tensor([0, 3], dtype=torch.int32, device=dev)).cpu().numpy()
    assert np.abs(crops[0]).sum() > 0 and np.abs(crops[1]).sum() == 0
    bx = torch.as_tensor(np.array([[100.0, 100, 300, 250]], np.float32), device=dev)
    init = ops.tco_init_autodepth(store, bx, K, torch.zeros(2, dtype=torch.int32, device=dev), obj[:2].to(dev),
                                  box_ids=torch.tensor([0, 4], dtype=torch.int32, device=dev)).cpu().numpy()
    assert np.isfinite(init[0]).all() and np.isnan(init[1]).all()


def test_two_lane_steps_are_bit_reproducible(dev, world):
    """The co-scheduling non-determinism of rounds 2-4, root-caused in round 5: packed-fp32 instructions the SLP
    vectoriser formed (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel swizzles -- in the rasteriser's plane set-up, in
    round 4 its facing test, in round 2 pose_prep) intermittently returned wrong values while the SIMD co-executed the OTHER
    lane's MFMA stream: 100 - 160 of 1200 two-lane MegaPose steps differed from the first (texture coordinates of whole views:
    thousands of colour values), 0 of 2400 when the library is built with -fno-slp-vectorize (happypose_amd/build.py).  The
    stress: 150 eager two-lane steps of 3 iterations (4 views, normals + depth; > 4000 rasteriser-stage launches beside the
    other lane's conv launches), EVERY field of every iteration -- the rendered pixels included -- equal to the first step's,
    bit for bit.  tools/probes/two_lane_repro.py is the stand-alone reproducer (per-process statistics, graph replays)."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.synthetic import make_scene

    renderer = world["renderer"]
    w = _weights("vanilla_resnet34", 32, seed=4)
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    model = create_model_pose(cfg, renderer, state_dict=w, max_batch=48, n_lanes=2)
    for lane in model.lanes:
        lane.keep_pixels = True
    sc = make_scene(n_detections=6, n_hypotheses=8, n_objects=len(renderer.store.labels), seed=9, with_depth=True)
    labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
    args = (torch.as_tensor(sc["images"][:, :4].copy(), device=dev), torch.as_tensor(sc["K"], device=dev), labels,
            torch.as_tensor(sc["TCO_hyp"], device=dev))
    im_ids = torch.zeros(len(labels), dtype=torch.int32)
    fields = ("TCO_input", "TCV_O_input", "boxes_crop", "K_crop", "KV_crop", "images_crop", "renders", "TCO_output")

    def step():
        out = model.forward(*args, n_iterations=3, im_ids=im_ids)
        snap = {}
        for n in (1, 2, 3):
            o = out[f"iteration={n}"]
            for f in fields:
                t = getattr(o, f, None)
                if t is not None:
                    snap[(n, f)] = t.clone()
            snap[(n, "pose")] = o.network_outputs["pose"].clone()
        return snap

    ref = step()
    assert ("renders" in {k[1] for k in ref}) and ref[(1, "renders")].abs().sum() > 0
    for r in range(150):
        cur = step()
        for key in ref:
            assert torch.equal(ref[key], cur[key]), (r, key, int((ref[key] != cur[key]).sum()))


# ---------------------------------------------------------------------------------------------------------------
# hipGraph replay of forward() (happypose_amd.graphs): the captured step must reproduce the eager one bit for bit
# (same kernels, same order, same buffers), on new inputs, with results that survive the next replay.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flavour,lanes", [("cosypose", 1), ("cosypose", 2), ("megapose", 2)])
def test_graph_replay_matches_eager(dev, world, flavour, lanes):
    from happypose_amd.models import create_model_pose, create_pose_model_cosypose
    from happypose_amd.synthetic import make_scene

    renderer = world["renderer"]
    scenes = [make_scene(n_detections=6, n_hypotheses=8, n_objects=len(renderer.store.labels), seed=s, with_depth=True)
              for s in (9, 10, 11)]
    if flavour == "cosypose":
        w = _weights("resnet18", 6, seed=3)
        make = lambda g: create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=48,  # noqa: E731
                                                    n_lanes=lanes, graphs=g)
        chans = 3
    else:
        w = _weights("vanilla_resnet34", 32, seed=4)
        cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                   render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
                   depth_normalization_type="tCR_scale_clamp_center")
        make = lambda g: create_model_pose(cfg, renderer, state_dict=w, max_batch=48, n_lanes=lanes, graphs=g)  # noqa: E731
        chans = 4
    eager, graphed = make(False), make(True)
    kept = []
    for i, sc in enumerate(scenes):  # call 1 eager, call 2 captures + replays, call 3 replays
        labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
        args = (torch.as_tensor(sc["images"][:, :chans].copy(), device=dev), torch.as_tensor(sc["K"], device=dev), labels,
                torch.as_tensor(sc["TCO_hyp"], device=dev))
        im_ids = torch.zeros(len(labels), dtype=torch.int32)
        a = eager.forward(*args, n_iterations=3, im_ids=im_ids)
        b = graphed.forward(*args, n_iterations=3, im_ids=im_ids)
        for k in a:
            for f in ("TCO_output", "TCO_input", "K_crop", "KV_crop", "boxes_crop", "boxes_rend", "tCR", "TCV_O_input"):
                assert torch.equal(getattr(a[k], f), getattr(b[k], f)), (i, k, f)
            assert torch.equal(a[k].network_outputs["pose"], b[k].network_outputs["pose"])
            assert a[k].labels == b[k].labels
        kept.append((a["iteration=3"].TCO_output.clone(), b["iteration=3"].TCO_output))
    caches = [graphed._graphs] if lanes == 1 else [l._graphs for l in graphed.lanes]  # two lanes: a graph per lane
    assert all(c is not None and c.replays == 2 for c in caches)
    for ref, got in kept:  # outputs of earlier replays were not overwritten by later ones
        assert torch.equal(ref, got)
    # per-hypothesis calling convention (the reference's) is another signature: captured separately
    sc = scenes[0]
    labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
    n = len(labels)
    imgs = torch.as_tensor(sc["images"][:1, :chans].copy(), device=dev).expand(n, -1, -1, -1).contiguous()
    Ks = torch.as_tensor(sc["K"][:1], device=dev).expand(n, -1, -1).contiguous()
    T0 = torch.as_tensor(sc["TCO_hyp"], device=dev)
    ref = eager.forward(imgs, Ks, labels, T0, n_iterations=2)["iteration=2"].TCO_output
    for _ in range(3):
        got = graphed.forward(imgs, Ks, labels, T0, n_iterations=2)["iteration=2"].TCO_output
        assert torch.equal(ref, got)
    # anything that changes the launches drops the captured graphs
    graphed.backbone.set_profiling(True)
    graphed.forward(imgs, Ks, labels, T0, n_iterations=2)
    graphed.backbone.set_profiling(False)
    got = graphed.forward(imgs, Ks, labels, T0, n_iterations=2)["iteration=2"].TCO_output
    assert torch.equal(ref, got)


def test_two_lane_megapose_is_reproducible(dev, world):
    """Regression: with the two lanes' launches sharing the CUs, the extra views' K_crop of hp_pose_prep sporadically
    changed by up to 100 px (every lane evaluated the fp64 look-at for itself and single lanes disagreed with lane 0;
    now lane 0 evaluates it once and the block shares the projection matrix).  Same inputs -> same bits, every time."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.synthetic import make_scene

    renderer = world["renderer"]
    w = _weights("vanilla_resnet34", 32, seed=4)
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    m = create_model_pose(cfg, renderer, state_dict=w, max_batch=48, n_lanes=2)
    sc = make_scene(n_detections=6, n_hypotheses=8, n_objects=len(renderer.store.labels), seed=9, with_depth=True)
    labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
    args = (torch.as_tensor(sc["images"][:, :4].copy(), device=dev), torch.as_tensor(sc["K"], device=dev), labels,
            torch.as_tensor(sc["TCO_hyp"], device=dev))
    im_ids = torch.zeros(len(labels), dtype=torch.int32)
    fields = ("TCO_output", "KV_crop", "TCV_O_input", "boxes_crop")
    ref = m.forward(*args, n_iterations=3, im_ids=im_ids)
    ref = {k: {f: getattr(v, f).clone() for f in fields} for k, v in ref.items()}
    for run in range(40):
        out = m.forward(*args, n_iterations=3, im_ids=im_ids)
        for k in out:
            for f in fields:
                assert torch.equal(ref[k][f], getattr(out[k], f)), (run, k, f)


@pytest.mark.parametrize("H,W,flavour", [(360, 500, "cosypose"), (720, 1280, "cosypose"), (350, 470, "megapose")])
def test_refiners_on_other_frame_sizes_vs_oracle(dev, world, H, W, flavour):
    """Frames that are not 640 x 480 (odd sizes, a portrait-ish aspect, HD): the crop box aspect rule, the roi_align
    bounds and the K_crop chain depend on (H, W); poses against the oracle at T_TOL / R_TOL after 3 iterations."""
    from happypose_amd.models import create_model_pose, create_pose_model_cosypose
    from happypose_amd.synthetic import make_scene
    from oracle.pipeline import OraclePredictor

    store = world["store"]
    sc = make_scene(n_detections=3, n_hypotheses=3, n_objects=3, seed=21, with_depth=True, H=H, W=W, f=0.9 * W)
    B = len(sc["TCO_hyp"])
    labels = _labels(world, sc["hyp_obj_ids"])
    K = torch.as_tensor(sc["K"], device=dev)
    im_ids = torch.zeros(B, dtype=torch.int32)
    if flavour == "cosypose":
        w = _weights("resnet18", 6, seed=1)
        model = create_pose_model_cosypose(dict(backbone_str="resnet18"), world["renderer"], state_dict=w, max_batch=16)
        images = sc["images"][:, :3].copy()
        ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="resnet18", cosypose=True)
    else:
        w = _weights("vanilla_resnet34", 32, seed=4)
        cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                   render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
                   depth_normalization_type="tCR_scale_clamp_center")
        model = create_model_pose(cfg, world["renderer"], state_dict=w, max_batch=16)
        images = sc["images"][:, :4].copy()
        ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4, multiview_type="TCO+front_3views",
                              render_normals=True, render_depth=True, input_depth=True,
                              depth_normalization_type="tCR_scale_clamp_center")
    out = model.forward(torch.as_tensor(images, device=dev), K, labels, torch.as_tensor(sc["TCO_hyp"]), n_iterations=3, im_ids=im_ids)
    ref = ora.forward(images, sc["K"], np.zeros(B, np.int32), sc["hyp_obj_ids"], sc["TCO_hyp"], 3)
    for n in range(3):
        o = out[f"iteration={n + 1}"]
        dt, dr = _pose_err(o.TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        np.testing.assert_allclose(o.boxes_crop.cpu().numpy(), ref[n]["boxes_crop"], rtol=1e-5, atol=5e-3)


def test_megapose_refiner_remove_tco_rendering_vs_oracle(dev, world):
    """``remove_TCO_rendering`` (MP/models/pose_rigid.py:578-611): the three look-at views of "TCO+front_3views" are rendered,
    the TCO view is not; every rendered view carries the K of its own 200-point crop and the pose update uses the K of
    the observed crop."""
    from happypose_amd.models import create_model_pose
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=3, multiview_type="TCO+front_3views", render_normals=True,
               render_depth=False, input_depth=False, predict_pose_update=True, depth_augmentation=False,
               remove_TCO_rendering=True)
    w = _weights("vanilla_resnet34", 3 + 6 * 3, seed=2)
    model = create_model_pose(cfg, world["renderer"], state_dict=w, max_batch=8)
    sel = np.arange(0, 12, 2)
    images = torch.as_tensor(sc["images"][:, :3].copy(), device=dev)
    K = torch.as_tensor(sc["K"], device=dev)
    out = model.forward(images, K, _labels(world, sc["hyp_obj_ids"][sel]), torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2,
                        im_ids=torch.zeros(len(sel), dtype=torch.int32))
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=3,
                          multiview_type="TCO+front_3views", render_normals=True, remove_TCO_rendering=True)
    ref = ora.forward(sc["images"][:, :3], sc["K"], np.zeros(len(sel), np.int32), sc["hyp_obj_ids"][sel], sc["TCO_hyp"][sel], 2)
    for n in range(2):
        o = out[f"iteration={n + 1}"]
        dt, dr = _pose_err(o.TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
        assert o.KV_crop.shape == (len(sel), 3, 3, 3) and o.TCV_O_input.shape == (len(sel), 3, 4, 4)
        np.testing.assert_allclose(o.K_crop.cpu().numpy(), ref[n]["K_crop"], rtol=1e-5, atol=5e-3)
    # the first rendered view is the re-aimed camera, not the TCO view
    assert not torch.allclose(out["iteration=1"].TCV_O_input[:, 0], out["iteration=1"].TCO_input, atol=1e-4)


def test_refiner_with_reference_render_state_vs_oracle(dev, world):
    """``BatchRenderer(msaa=True, aniso=True)`` (the reference renderer's framebuffer / texture state) through the whole
    MegaPose refiner: multi-view NHWC slices, normals and depth channels, against the oracle with the same switches."""
    from happypose_amd.models import create_model_pose
    from happypose_amd.renderer import BatchRenderer
    from oracle.pipeline import OraclePredictor

    sc, store = world["scene"], world["store"]
    renderer = BatchRenderer(world["ds"], device=dev, store=store, msaa=True, aniso=True)  # (the default, spelled out)
    cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
               render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
               depth_normalization_type="tCR_scale_clamp_center")
    w = _weights("vanilla_resnet34", 32, seed=2)
    model = create_model_pose(cfg, renderer, state_dict=w, max_batch=8)
    model.keep_pixels = True
    sel = np.arange(0, 12, 2)
    images = torch.as_tensor(sc["images"], device=dev)
    out = model.forward(images, torch.as_tensor(sc["K"], device=dev), _labels(world, sc["hyp_obj_ids"][sel]),
                        torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2, im_ids=torch.zeros(len(sel), dtype=torch.int32))
    ora = OraclePredictor(w, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                          multiview_type="TCO+front_3views", render_normals=True, render_depth=True, input_depth=True,
                          depth_normalization_type="tCR_scale_clamp_center", msaa=True, aniso=True)
    ref = ora.forward(sc["images"], sc["K"], np.zeros(len(sel), np.int32), sc["hyp_obj_ids"][sel], sc["TCO_hyp"][sel], 2)
    for n in range(2):
        o = out[f"iteration={n + 1}"]
        dt, dr = _pose_err(o.TCO_output.cpu().numpy(), ref[n]["TCO_output"])
        assert dt <= T_TOL and dr <= R_TOL, (n, dt, dr)
    # the renders of the first iteration differ from the default state's (the switches reach the predictor's raster call)
    plain = create_model_pose(cfg, BatchRenderer(world["ds"], device=dev, store=store, msaa=False, aniso=False), state_dict=w, max_batch=8)
    plain.keep_pixels = True
    o2 = plain.forward(images, torch.as_tensor(sc["K"], device=dev), _labels(world, sc["hyp_obj_ids"][sel]),
                       torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=1, im_ids=torch.zeros(len(sel), dtype=torch.int32))
    r_on, r_off = out["iteration=1"].renders, o2["iteration=1"].renders
    assert r_on.shape == r_off.shape == (len(sel), 28, 240, 320)
    rgb_idx = [v * 7 + c for v in range(4) for c in range(3)]
    dep_idx = [v * 7 + 6 for v in range(4)]
    assert (r_on[:, rgb_idx] != r_off[:, rgb_idx]).float().mean() > 0.01
    assert torch.equal(r_on[:, dep_idx], r_off[:, dep_idx])  # depth stays centre-sampled


# ---------------------------------------------------------------------------------------------------------------
# Sharp parity: high-gain heads at the benchmarked sizes, features at batch 128 on two lanes,
# and the proof that the checks would see a 1 % error in one layer.
# ---------------------------------------------------------------------------------------------------------------
def _oracle_for(workload, weights, store):
    from oracle.pipeline import OraclePredictor

    if workload == "C2":
        return OraclePredictor(weights, store.packed, store.mesh_db.points, arch="resnet34", cosypose=True), 3
    return OraclePredictor(weights, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                           multiview_type="TCO+front_3views", render_normals=True, render_depth=True, input_depth=True,
                           depth_normalization_type="tCR_scale_clamp_center"), 4


@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_full_size_high_gain_head_vs_oracle(dev, workload):
    """The benchmarked worlds with the pose head at ``update_scale = 0.05``: every iteration rotates a hypothesis by
    ~0.1 rad and moves it by centimetres -- >= 100x T_TOL_HI / R_TOL_HI -- so a relative error of 1 % anywhere in the
    features shows up in the pose.  5 iterations, two lanes, against the CPU oracle in the reference's chunks of 8."""
    bench = _bench()
    gain = 0.05 if workload == "C2" else 0.02  # MegaPose's update multiplies the translation: 0.05 throws the objects out of view
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload=workload, n_lanes=2, update_scale=gain)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    out = model.forward(images, K, labels, torch.as_tensor(scene["TCO_hyp"], device=dev), n_iterations=5,
                        im_ids=torch.zeros(B, dtype=torch.int32, device=dev))
    assert model.numerics_status() == 0
    torch.set_num_threads(bench.effective_cpu_count())
    ora, n_ch = _oracle_for(workload, weights, store)
    ref = ora.forward(scene["images"][:, :n_ch], scene["K"], np.zeros(B, np.int32), scene["hyp_obj_ids"], scene["TCO_hyp"], 5, bsz_objects=8)
    errs, prev = [], scene["TCO_hyp"]
    for n in range(5):
        got = out[f"iteration={n + 1}"].TCO_output.cpu().numpy()
        errs.append(_pose_err(got, ref[n]["TCO_output"]) + _pose_med(got, ref[n]["TCO_output"]))
        assert errs[-1][0] <= T_TOL_HI and errs[-1][1] <= R_TOL_HI, (n, errs)
        assert errs[-1][2] <= (n + 1) * T_MED_HI and errs[-1][3] <= (n + 1) * R_MED_HI, (n, errs)
        # the update of THIS iteration is >= 20x the tolerance for the typical hypothesis (median over the batch)
        A, Bm = np.asarray(got, np.float64), np.asarray(prev, np.float64)
        chord = np.linalg.norm(A[:, :3, :3] - Bm[:, :3, :3], axis=(1, 2))
        assert np.median(2 * np.arcsin(np.clip(chord / (2 * np.sqrt(2)), 0, 1))) > (20 if workload == "C2" else 8) * R_TOL_HI, n
        prev = got


def _lane_features(model, x):
    """Backbone features of a batch through BOTH lanes of a TwoLanePredictor exactly as ``forward`` runs them: the two
    halves on the two streams, concurrently, tail K-slicing off (the planner paths of half-CU slicing)."""
    h = x.shape[0] // 2
    cur = torch.cuda.current_stream(x.device)
    feats = []
    model.backbone.set_tail_split(False)
    try:
        for lane, stream, sl in zip(model.lanes, model.streams, (slice(0, h), slice(h, x.shape[0]))):
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                feats.append(lane.backbone.forward(x[sl].contiguous(), want_pose=False, want_features=True)[2])
    finally:
        model.backbone.set_tail_split(True)
    for stream in model.streams:
        cur.wait_stream(stream)
    return torch.cat(feats)


@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_backbone_features_at_benchmark_batch_two_lanes(dev, workload):
    """Feature-level parity at the BENCHMARKED batch (C2: 128 x 6 ch WideResNet-34; C3: 64 x 32 ch ResNet-34), two lanes
    running concurrently: the 512 pooled features of every sample against ``oracle/backbones.py`` on the very same
    network input, FEAT_TOL x max|ref| PER SAMPLE.  No renders, no pose head in between: this is the check a wrong tile,
    a dropped K slice or a 1 % weight error cannot pass (the second half of the test injects exactly that)."""
    from oracle import backbones as ob

    bench = _bench()
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload=workload, n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    # the network input of iteration 1, assembled by the product's crop + rasteriser (one lane, eager)
    lane = model.lanes[0]
    im_ids, obj_ids = lane._ids(images, K, labels, torch.zeros(B, dtype=torch.int32, device=dev))
    xs = []
    for s0 in range(0, B, B // 2):  # a lane's network holds half the batch
        sl = slice(s0, s0 + B // 2)
        kw = dict(n_img_channels=3, multiview_type="TCO", normalize=False, render_normals=False, render_depth=False, depth_mode=0,
                  want_pose=True, want_logits=False) if workload == "C2" else \
            dict(n_img_channels=lane._n_img, multiview_type=lane.multiview_type, normalize=True, render_normals=lane.render_normals,
                 render_depth=lane.render_depth, depth_mode=lane._depth_mode, want_pose=True, want_logits=False)
        _, x, _, _, _ = lane._one_pass(images, K, im_ids[sl], obj_ids[sl], torch.as_tensor(scene["TCO_hyp"][sl], device=dev), **kw)
        xs.append(x.clone())
    x = torch.cat(xs)
    n_in = lane.backbone.n_inputs
    feats = _lane_features(model, x).cpu().numpy()
    assert model.numerics_status() == 0
    torch.set_num_threads(bench.effective_cpu_count())
    arch = "resnet34" if workload == "C2" else "vanilla_resnet34"
    x_nchw = x[..., :n_in].permute(0, 3, 1, 2).contiguous().cpu()
    with torch.no_grad():
        ref = torch.cat([ob.net_forward(x_nchw[i:i + 16], {k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()}, arch,
                                        heads=("features",))["features"] for i in range(0, B, 16)]).numpy()
    assert feats.shape == ref.shape == (B, 512)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    err = np.abs(feats - ref) / scale
    assert err.max() <= FEAT_TOL, (err.max(), int(err.max(axis=1).argmax()))
    # an injected 1 % error in ONE layer's weights is far outside that bound
    from happypose_amd import ops

    w_bad = dict(weights)
    w_bad["backbone.layer2.1.conv1.weight"] = (np.asarray(weights["backbone.layer2.1.conv1.weight"]) * 1.01).astype(np.float32)
    bad = ops.Net(arch, n_in, w_bad, max_batch=16, device=dev)
    f_bad = bad.forward(x[:16].contiguous(), want_pose=False, want_features=True)[2].cpu().numpy()
    err_bad = (np.abs(f_bad - ref[:16]) / scale[:16]).max()
    assert err_bad > 5 * FEAT_TOL, err_bad


def test_c4_refine_sharded_1024_hypotheses(dev):
    """C4 on the HIP path: ``distributed.refine_sharded`` with the real two-lane CosyPose predictor at world = 1 on a
    1024-hypothesis batch (64 detections x 16 hypotheses on one frame: 8 chunks of 128, the per-GPU share of the
    8-GPU configuration).  Poses must equal eight direct ``forward`` calls bit for bit and agree with the CPU oracle
    on a 128-hypothesis subset (16 rows of every chunk)."""
    from happypose_amd import distributed as D
    from happypose_amd.synthetic import make_scene
    from oracle.pipeline import OraclePredictor

    bench = _bench()
    ds, renderer, _, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=2)
    store = renderer.store
    scene = make_scene(n_detections=64, n_hypotheses=16, n_objects=8, seed=11)
    B = len(scene["TCO_hyp"])
    assert B == 1024 and model.max_batch == 128
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    TCO = torch.as_tensor(scene["TCO_hyp"], device=dev)
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    poses, scores = D.refine_sharded(model, images, K, labels, TCO, 5, im_ids=im_ids)
    assert poses.shape == (B, 4, 4) and scores.shape == (B,) and torch.isfinite(poses).all()
    assert model.numerics_status() == 0
    direct = torch.cat([model.forward(images, K, labels[s:s + 128], TCO[s:s + 128], n_iterations=5, im_ids=im_ids[s:s + 128])[
        "iteration=5"].TCO_output for s in range(0, B, 128)])
    assert torch.equal(poses, direct)
    sub = np.concatenate([np.arange(s, s + 16) for s in range(0, B, 128)])
    torch.set_num_threads(bench.effective_cpu_count())
    ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="resnet34", cosypose=True)
    ref = ora.forward(scene["images"][:, :3], scene["K"], np.zeros(len(sub), np.int32), scene["hyp_obj_ids"][sub], scene["TCO_hyp"][sub], 5,
                      bsz_objects=8)[-1]["TCO_output"]
    dt, dr = _pose_err(poses[torch.as_tensor(sub, device=dev)].cpu().numpy(), ref)
    assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    # the estimator entry point on the same batch (world = 1: no sharding, chunks of bsz_objects = 128)
    import pandas as pd

    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor
    from happypose_amd.tensor_collection import PandasTensorCollection

    hyp = PandasTensorCollection(pd.DataFrame({"label": labels, "batch_im_id": 0, "instance_id": scene["hyp_det_ids"]}), poses=TCO)
    est = CosyPoseEstimator(refiner_model=model, coarse_model=model, bsz_objects=128)
    final, extra = est.run_inference_pipeline(ObservationTensor(images, K), data_TCO_init=hyp, n_coarse_iterations=0, n_refiner_iterations=5)
    assert torch.equal(final.poses, direct) and extra["refiner"]["data"]["shard"] == (0, B)
    assert final.infos.refiner_batch_idx.tolist() == (np.arange(B) // 128).tolist()


# ---------------------------------------------------------------------------------------------------------------
# hipGraph replay under the conditions ADVICE r2 named
# ---------------------------------------------------------------------------------------------------------------
def test_graph_replay_survives_scratch_growth(dev, world):
    """A captured graph holds the store's rasteriser-scratch pointers.  A later, larger user of the SAME store (a
    predictor with a bigger batch) reallocates that scratch: the older graphs must not be replayed through the freed
    memory (hp_mesh_store_scratch_generation drops them); results stay equal to the eager path."""
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_scene

    renderer = BatchRenderer(world["ds"], device=dev)  # a store of its own: its scratch starts empty
    store = renderer.store
    w = _weights("resnet18", 6, seed=3)
    sc = make_scene(n_detections=8, n_hypotheses=12, n_objects=len(store.labels), seed=5)
    labels = [store.labels[j] for j in sc["hyp_obj_ids"]]
    images, K = torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev)
    T = torch.as_tensor(sc["TCO_hyp"], device=dev)
    ids = torch.zeros(len(labels), dtype=torch.int32)
    small = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=8, graphs=True)
    eager = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=8)
    g0 = store.scratch_generation()
    run = lambda m, n: m.forward(images, K, labels[:n], T[:n], n_iterations=2, im_ids=ids[:n])["iteration=2"].TCO_output  # noqa: E731
    ref8 = run(eager, 8)
    for _ in range(3):  # eager, capture, replay
        assert torch.equal(run(small, 8), ref8)
    assert small._graphs.replays >= 2 and store.scratch_generation() == g0
    big = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=96)  # reserves 96 views
    assert store.scratch_generation() > g0
    ref96 = run(big, 96)
    # poison what the allocator may hand out next, then replay the old signature
    junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(8)]
    got = run(small, 8)
    del junk
    assert torch.equal(got, ref8)
    assert torch.equal(run(big, 96), ref96)
    for _ in range(3):
        assert torch.equal(run(small, 8), ref8)


def test_graph_signature_includes_tail_split(dev, world):
    """TwoLanePredictor with graphs: a 40-hypothesis frame runs lane 0 on 20 rows with tail K-slicing OFF, a
    20-hypothesis frame runs the same lane on 20 rows with tail K-slicing ON -- same shapes, different launch plans:
    two graph signatures, each equal to its eager twin, in any interleaving."""
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.synthetic import make_scene

    renderer = world["renderer"]
    store = renderer.store
    w = _weights("resnet18", 6, seed=3)
    sc = make_scene(n_detections=5, n_hypotheses=8, n_objects=len(store.labels), seed=6)
    labels = [store.labels[j] for j in sc["hyp_obj_ids"]]
    images, K = torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev)
    T = torch.as_tensor(sc["TCO_hyp"], device=dev)
    ids = torch.zeros(len(labels), dtype=torch.int32)
    make = lambda g: create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=48, n_lanes=2, graphs=g)  # noqa: E731
    eager, graphed = make(False), make(True)
    run = lambda m, n: m.forward(images, K, labels[:n], T[:n], n_iterations=2, im_ids=ids[:n])["iteration=2"].TCO_output.clone()  # noqa: E731
    ref40, ref20 = run(eager, 40), run(eager, 20)
    for n in (40, 20, 40, 20, 20, 40, 40, 20):
        assert torch.equal(run(graphed, n), ref40 if n == 40 else ref20), n
    keys = [k for k in graphed.lanes[0]._graphs.entries]
    assert len(keys) == 2 and {dict(k[0][1:])["tail_split"] for k in keys} == {True, False}


def test_two_lanes_keep_the_render_state(dev, world):
    """``n_lanes=2`` must render BOTH halves of the batch with the renderer's state (msaa / aniso): lane 1's renderer is
    a clone of lane 0's.  One lane vs two lanes on a multisampled, mip-filtered renderer: same poses."""
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_scene

    renderer = BatchRenderer(world["ds"], device=dev, msaa=True, aniso=True)
    store = renderer.store
    w = _weights("resnet18", 6, seed=3, scale=0.05)
    sc = make_scene(n_detections=6, n_hypotheses=8, n_objects=len(store.labels), seed=9)
    labels = [store.labels[j] for j in sc["hyp_obj_ids"]]
    args = (torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev), labels, torch.as_tensor(sc["TCO_hyp"], device=dev))
    ids = torch.zeros(len(labels), dtype=torch.int32)
    outs = []
    for lanes in (1, 2):
        m = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=48, n_lanes=lanes)
        if lanes == 2:
            assert m.lanes[1].renderer.msaa and m.lanes[1].renderer.aniso
        outs.append(m.forward(*args, n_iterations=2, im_ids=ids)["iteration=2"].TCO_output)
    dt, dr = _pose_err(outs[0].cpu().numpy(), outs[1].cpu().numpy())
    assert dt <= T_TOL and dr <= R_TOL, (dt, dr)
    plain = create_pose_model_cosypose(dict(backbone_str="resnet18"), BatchRenderer(world["ds"], device=dev, msaa=False, aniso=False),
                                       state_dict=w, max_batch=48, n_lanes=2)
    off = plain.forward(*args, n_iterations=2, im_ids=ids)["iteration=2"].TCO_output
    assert _pose_err(off.cpu().numpy(), outs[1].cpu().numpy())[1] > R_TOL  # the state does reach both lanes' renders


def test_lane_stores_follow_the_parent_store_state(dev, world):
    """The conventions record and the culling switch live on the mesh store; the stores of lanes >= 1 are clones that FOLLOW
    lane 0's (``ops.MeshStore.clone_for_lane``): a non-default record / culling set BEFORE the predictor is built, and one set
    AFTER, reach every lane -- every lane's renderer gives the same pixels bit for bit."""
    from happypose_amd import ops
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_scene

    renderer = BatchRenderer(world["ds"], device=dev, msaa=True, aniso=True)
    store = renderer.store
    before = dict(normal_sign=(1.0, -1.0, 1.0), lod_bias=0.5)
    store.set_raster_conventions(before)
    store.set_backface_culling(False)
    w = _weights("resnet18", 6, seed=3, scale=0.05)
    sc = make_scene(n_detections=3, n_hypotheses=4, n_objects=len(store.labels), seed=9)
    labels = [store.labels[j] for j in sc["hyp_obj_ids"]]
    TCO = torch.as_tensor(sc["TCO_hyp"], device=dev)
    K = torch.as_tensor(sc["K"], device=dev)[:1].expand(len(labels), 3, 3).contiguous()
    K = K * torch.tensor([0.5, 0.5, 1.0], device=dev)[:, None]  # the 640 x 480 camera at the 320 x 240 render size
    m = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=w, max_batch=48, n_lanes=3)
    stores = [l.renderer.store for l in m.lanes]
    assert stores[0] is store and stores[1] is not store and stores[2] is not stores[1]

    def pixels():
        outs = [l.renderer.render(labels, TCO, K, resolution=(240, 320), render_normals=True, render_depth=True) for l in m.lanes]
        for o in outs[1:]:
            assert torch.equal(o.rgbs, outs[0].rgbs) and torch.equal(o.normals, outs[0].normals) and torch.equal(o.depths, outs[0].depths)
        return outs[0].rgbs.clone(), outs[0].normals.clone()

    for st in stores[1:]:
        assert st.get_raster_conventions() == dict(ops.RASTER_CONVENTION_DEFAULTS, **before) and st.get_backface_culling() is False
    rgb_a, nrm_a = pixels()
    after = dict(lod_bias=-0.75, aniso_max=4)  # set AFTER construction, on lane 0's store only
    store.set_raster_conventions(after)
    assert store.set_backface_culling(True) is False
    for st in stores[1:]:
        assert st.get_raster_conventions() == dict(ops.RASTER_CONVENTION_DEFAULTS, **after) and st.get_backface_culling() is True
    rgb_b, nrm_b = pixels()
    # and the record matters: the two states give different pixels (normal sign; level-of-detail bias)
    assert not torch.equal(nrm_a, nrm_b) and not torch.equal(rgb_a, rgb_b)


# ---------------------------------------------------------------------------------------------------------------
# Lights (VERDICT r2 missing #4): positioning functions through the renderer plug, and the lit MegaPose loop
# ---------------------------------------------------------------------------------------------------------------
def test_renderer_honours_positioning_functions(dev, world):
    """Light lists as REFERENCE code builds them -- a re-statement of ``make_scene_lights``'s signature and body
    (TB/renderer/panda3d_scene_renderer.py:105-141: ``partial(pos_fn, pos=...)`` reading ``root_node.getBounds().radius``
    and calling ``light_node.setPos(tuple)``) -- pass through ``BatchRenderer.render`` and light the scene like the oracle
    with the positions written out by hand."""
    from functools import partial

    from happypose_amd.renderer import Panda3dLightData, make_scene_lights
    from oracle import native

    def reference_style_lights(ambient_light_color=(0.1, 0.1, 0.1, 1.0), point_lights_color=(0.4, 0.4, 0.4, 1.0)):
        pos = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])

        def pos_fn(root_node, light_node, pos):
            radius = root_node.getBounds().radius
            xyz_ = pos * radius * 10
            light_node.setPos(tuple(xyz_.tolist()))

        out = [Panda3dLightData(light_type="ambient", color=ambient_light_color)]
        for pos_n in pos:
            out.append(Panda3dLightData(light_type="point", color=point_lights_color, positioning_function=partial(pos_fn, pos=pos_n)))
        return out

    sc, store, renderer = world["scene"], world["store"], world["renderer"]
    sel = np.array([0, 5, 10, 3])
    obj = sc["hyp_obj_ids"][sel]
    T = sc["TCO_hyp"][sel]
    K = np.tile(np.array([[700.0, 0, 160], [0, 700.0, 120], [0, 0, 1]], np.float32), (len(sel), 1, 1))
    labels = _labels(world, obj)
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    lp = dirs[None] * (10 * store.packed.bounds_radius[obj])[:, None, None]
    ref = native.rasterize(store.packed, obj, T, K, (240, 320), ambient=np.full((len(sel), 3), 0.1, np.float32), light_pos=lp.astype(np.float32),
                           light_col=np.full((len(sel), 6, 3), 0.4, np.float32), msaa=renderer.msaa, aniso=renderer.aniso)
    for lights in ([reference_style_lights() for _ in sel], [make_scene_lights() for _ in sel]):
        out = renderer.render(labels, torch.as_tensor(T, device=dev), torch.as_tensor(K, device=dev), light_datas=lights, resolution=(240, 320))
        got = out.rgbs.cpu().numpy()
        cov = ref["rgbs"].sum(1) > 0
        assert ((got.sum(1) > 0) != cov).mean() < 2e-3
        d = np.abs(got - ref["rgbs"])
        assert (d > 1.01 / 255).mean() < 3e-3 and np.median(d) == 0  # silhouette / filter-rounding pixels only
    with pytest.raises(AssertionError):  # setup_lights asserts a point light has its function (:303)
        renderer.render(labels, torch.as_tensor(T, device=dev), torch.as_tensor(K, device=dev),
                        light_datas=[[Panda3dLightData("point")] for _ in sel], resolution=(240, 320))


def test_lit_megapose_refiner_vs_reference_golden_g10(dev, golden_dir):
    """``render_normals=False``: the predictor lights every view with ``make_scene_lights()`` like
    MP/models/pose_rigid.py:422.  Against the reference's own ``PosePredictor.forward`` run with ITS lights (golden G10,
    ``mp_lit``; high-gain head) and against the high-gain CosyPose / RGB-D cases of the same golden."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import gen_golden_loop as ggl

    from happypose_amd.models import create_model_pose, create_pose_model_cosypose
    from happypose_amd.renderer import BatchRenderer

    g = np.load(golden_dir / "g10_loop.npz")
    ds, packed, mesh_db, sc = ggl.world()
    renderer = BatchRenderer(ds, device=dev)
    labels_all = list(renderer.store.labels)
    images, K = torch.as_tensor(sc["images"], device=dev), torch.as_tensor(sc["K"], device=dev)
    sel, sel4 = g["cosy/sel"], g["mp_rgbd4/sel"]
    lab = [labels_all[i] for i in sc["hyp_obj_ids"][sel]]
    ids = torch.zeros(len(sel), dtype=torch.int32)

    def check(tag, out, n_it, t_tol, r_tol):
        for n in range(1, n_it + 1):
            dt, dr = _pose_err(out[f"iteration={n}"].TCO_output.cpu().numpy(), g[f"{tag}/it{n}/TCO_output"])
            assert dt <= t_tol and dr <= r_tol, (tag, n, dt, dr)

    lit = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=False,
                                 depth_augmentation=False), renderer, state_dict=ggl.case_weights("mp_lit"), max_batch=8)
    lit.keep_pixels = True
    out = lit.forward(images[:, :3].contiguous(), K, lab, torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=2, im_ids=ids)
    check("mp_lit", out, 2, T_TOL_HI, R_TOL_HI)
    rend = out["iteration=1"].renders.cpu().numpy()
    np.testing.assert_allclose(rend.astype(np.float64).mean(axis=(0, 2, 3)), g["mp_lit/it1/renders_mean"], atol=1e-4)
    dx = np.abs(rend[:, :, ::7, ::11] - g["mp_lit/it1/renders_sample"])
    assert (dx > 1.01 / 255).mean() < 2e-3
    cosy = create_pose_model_cosypose(dict(backbone_str="resnet18"), renderer, state_dict=ggl.case_weights("cosy_hi"), max_batch=8)
    check("cosy_hi", cosy.forward(images[:, :3].contiguous(), K, lab, torch.as_tensor(sc["TCO_hyp"][sel]), n_iterations=3, im_ids=ids), 3,
          T_TOL_HI, R_TOL_HI)
    lab4 = [labels_all[i] for i in sc["hyp_obj_ids"][sel4]]
    m4 = create_model_pose(dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
                                render_depth=True, input_depth=True, depth_augmentation=False,
                                depth_normalization_type="tCR_scale_clamp_center"), renderer, state_dict=ggl.case_weights("mp_rgbd4_hi"),
                           max_batch=4)
    check("mp_rgbd4_hi", m4.forward(images, K, lab4, torch.as_tensor(sc["TCO_hyp"][sel4]), n_iterations=3,
                                    im_ids=torch.zeros(3, dtype=torch.int32)), 3, T_TOL_HI, R_TOL_HI)


def test_bench_two_ranks_on_one_device(dev):
    """``bench.py --gpus 2`` as the driver launches it (two processes, RANK / WORLD_SIZE wiring, the 2 x 128-hypothesis batch
    of C4 cut by ``distributed.refine_sharded`` with the REAL two-lane predictor in both processes, the all-gather of the
    refined poses, max-over-ranks timing, one JSON line from rank 0) -- on the one GPU of this box: both ranks on cuda:0,
    gloo instead of RCCL (which refuses two ranks on one device).  Everything but the transport of the collective."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HP_BENCH_DIST_BACKEND="gloo", HP_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-exact-fp32", "--no-extra-workloads"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["hypotheses_per_gpu"] == 128 and "all_gather_us" in d
    assert d["value"] > 0 and abs(d["value"] - 2 * 128 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]  # whole-job rate


def test_estimators_shard_with_real_predictors_two_ranks(dev):
    """8(e) behind the entry point with the HIP predictors: two processes (both on cuda:0, gloo) run MegaPose's and
    CosyPose's ``run_inference_pipeline`` sharded and unsharded (tests/gpu_shard_worker.py): same ids / labels / order,
    poses within the parity tolerance of the unsharded run, and bit-identical results on both ranks."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_shard_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in out, out[-3000:]


def test_lane_streams_are_shared_and_concurrent(dev, world):
    """Every TwoLanePredictor of a process runs its lanes on the same, probed pair of streams: fresh streams can share a
    hardware queue (HIP maps them in creation order), which serialised the lanes of every other predictor (+37 % step time)."""
    from happypose_amd.models import create_pose_model_cosypose
    from happypose_amd.pose_predictor import TwoLanePredictor

    junk = [torch.cuda.Stream(dev) for _ in range(3)]  # shift the creation order
    w = _weights("resnet34", 6)
    m1 = create_pose_model_cosypose(dict(backbone_str="resnet34"), world["renderer"], state_dict=w, max_batch=64, n_lanes=2)
    m2 = create_pose_model_cosypose(dict(backbone_str="resnet34"), world["renderer"], state_dict=w, max_batch=64, n_lanes=2)
    assert isinstance(m1, TwoLanePredictor) and len(m1.streams) == 2
    assert all(a is b for a, b in zip(m1.streams, m2.streams))
    assert m1.streams[0].cuda_stream != m1.streams[1].cuda_stream
    assert TwoLanePredictor._concurrent(dev, *m1.streams)
    del junk


def test_scratch_kernels_keep_graphs_off(dev, world):
    """A tile variant that spills to scratch (EfficientNet's narrow 1x1 layers run one) must not end up in a captured
    hipGraph: a predictor whose eager call moves ``hp_scratch_launches`` stays on eager launches -- and gives the eager result."""
    from happypose_amd import ops
    from happypose_amd.models import create_pose_model_cosypose

    sc = world["scene"]
    w = _weights("efficientnet-b3", 6)
    images, K = torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev)
    T0 = torch.as_tensor(sc["TCO_hyp"], device=dev)
    labels = _labels(world, sc["hyp_obj_ids"])
    im_ids = torch.zeros(len(labels), dtype=torch.int32, device=dev)
    eager = create_pose_model_cosypose(dict(backbone_str="efficientnet-b3"), world["renderer"], state_dict=w, max_batch=16)
    n0 = ops.scratch_launches()
    ref = eager.forward(images, K, labels, T0, n_iterations=2, im_ids=im_ids)["iteration=2"].TCO_output
    if ops.scratch_launches() == n0:
        # round 4: no tile variant EfficientNet-b3 launches uses scratch any more (the narrow 1x1 layers got their registers),
        # so hipGraph replay is ALLOWED for it -- and must reproduce the eager result
        g = create_pose_model_cosypose(dict(backbone_str="efficientnet-b3"), world["renderer"], state_dict=w, max_batch=16, graphs=True)
        for _ in range(4):
            out = g.forward(images, K, labels, T0, n_iterations=2, im_ids=im_ids)["iteration=2"].TCO_output
            assert torch.equal(out, ref)
        assert g._graphs is not None and not getattr(g, "_no_graphs", False)  # the step was captured and replayed
        return
    g = create_pose_model_cosypose(dict(backbone_str="efficientnet-b3"), world["renderer"], state_dict=w, max_batch=16, graphs=True)
    for _ in range(3):
        out = g.forward(images, K, labels, T0, n_iterations=2, im_ids=im_ids)["iteration=2"].TCO_output
        assert torch.equal(out, ref)
    assert g._graphs is None and g._no_graphs  # nothing was captured
