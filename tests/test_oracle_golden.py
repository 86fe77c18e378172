"""The CPU oracle against the golden vectors produced by the reference itself
(tools/gen_golden.py).  Tolerances: the oracle is NumPy float32 and the reference
torch float32 -- identical formulas, possibly different summation order inside
matmul/einsum, hence 1-2 ulp-level slack rather than bit equality."""

import numpy as np
import pytest

from oracle import geometry as G

RTOL, ATOL = 2e-6, 2e-6


def load(golden_dir, name):
    return np.load(golden_dir / name)


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_g1_transforms(golden_dir):
    g = load(golden_dir, "g1_transforms.npz")
    close(G.compute_rotation_matrix_from_ortho6d(g["p6"]), g["R6"])
    close(G.normalize_T(g["T_noisy"]), g["T_norm"])
    close(G.invert_transform_matrices(g["T"]), g["T_inv"])
    close(G.transform_pts(g["T"], g["pts"]), g["pts_T"])


def test_g2_projection(golden_dir):
    g = load(golden_dir, "g2_projection.npz")
    uv = G.project_points_robust(g["pts"], g["K"], g["T"])
    # pixel coordinates of O(1e2..1e4): relative tolerance
    close(uv, g["uv_robust"], rtol=1e-5, atol=1e-3)
    close(G.project_points(g["pts"][4:], g["K"][4:], g["T"][4:]), g["uv_plain"], rtol=1e-5, atol=1e-3)
    close(G.boxes_from_uv(g["uv_robust"]), g["boxes"], rtol=0, atol=0)
    close(G.get_K_crop_resize(g["K"], g["boxes_k"], (480, 640), (240, 320)), g["K_crop"],
          rtol=1e-6, atol=1e-4)


def test_g3_deepim_boxes(golden_dir):
    g = load(golden_dir, "g3_deepim_boxes.npz")
    close(G.deepim_boxes(g["center"], g["obs"], g["rend"], 1.4, (480, 640)), g["boxes"],
          rtol=1e-6, atol=1e-4)
    ex = G.deepim_boxes(np.array([[[300.0, 200.0]]]), np.array([[250.0, 150, 380, 260]]),
                        np.array([[250.0, 150, 380, 260]]), 1.4, (480, 640))
    close(ex, g["example"], rtol=0, atol=1e-4)
    close(ex, np.array([[188.0, 116.0, 412.0, 284.0]]), rtol=0, atol=1e-4)


def test_g4_pose_update_and_init(golden_dir):
    g = load(golden_dir, "g4_pose_update.npz")
    dR = G.compute_rotation_matrix_from_ortho6d(g["pose9"][:, :6])
    close(G.pose_update_with_reference_point(g["T"], g["K_crop"], g["pose9"][:, 6:], dR, g["tCR"]),
          g["upd_ref"], rtol=1e-5, atol=1e-6)
    close(G.update_pose(g["T"], g["K_crop"], g["pose9"], g["T"][:, :3, 3]), g["upd_origin"],
          rtol=1e-5, atol=1e-6)
    close(G.apply_imagespace_predictions(g["T"], g["K_crop"], g["pose9"][:, 6:], dR), g["upd_cosy"],
          rtol=1e-5, atol=1e-6)
    # MegaPose (tCR = tCO) == CosyPose update
    close(g["upd_origin"], g["upd_cosy"], rtol=1e-5, atol=1e-6)
    close(G.TCO_init_from_boxes((1.0, 1.0), g["det_boxes"], g["K"]), g["init_v0"], rtol=1e-6, atol=1e-6)
    close(G.TCO_init_from_boxes_autodepth_with_R(g["det_boxes"], g["mpts"], g["K"], g["Rg"]),
          g["init_R"], rtol=1e-5, atol=1e-6)
    close(G.TCO_init_from_boxes_zup_autodepth(g["det_boxes"], g["mpts"], g["K"]), g["init_zup"],
          rtol=1e-5, atol=1e-6)
    close(g["init_zup"], g["init_zup_cosy"], rtol=0, atol=0)


def test_g5_sampling(golden_dir):
    g = load(golden_dir, "g5_sampling.npz")
    for key in g.files:
        if key.startswith("ids_"):
            _, n_pad, n_pts = key.split("_")
            np.testing.assert_array_equal(G.sample_point_ids(int(n_pad), int(n_pts)), g[key])
    lens = g["pad_lens"]
    lst = [np.arange(n, dtype=np.float32)[:, None].repeat(3, 1) + 100 * i for i, n in enumerate(lens)]
    np.testing.assert_array_equal(G.pad_stack_points(lst), g["pad_stack"])


def test_g7_iteration_chain(golden_dir):
    g = load(golden_dir, "g7_iteration.npz")
    Tn = G.normalize_T(g["T"])
    close(Tn, g["T_norm"])
    tCR = Tn[:, :3, 3]
    br, bc = G.crop_boxes_from_pose(g["pts"], g["K"], Tn, tCR, (480, 640))
    close(br, g["boxes_rend"], rtol=1e-5, atol=1e-3)
    close(bc, g["boxes_crop"], rtol=1e-5, atol=2e-3)
    Kc = G.get_K_crop_resize(g["K"], bc, (480, 640), (240, 320))
    close(Kc, g["K_crop"], rtol=2e-5, atol=2e-3)
    close(G.update_pose(Tn, Kc, g["pose9"], tCR), g["T_out"], rtol=2e-5, atol=2e-6)


def test_g6_backbones(golden_dir):
    import torch

    from happypose_amd.synthetic import named_weights
    from oracle import backbones as ob

    g = load(golden_dir, "g6_backbones.npz")
    torch.set_num_threads(8)
    for arch, cin, tag in [("vanilla_resnet34", 27, "vanilla_resnet34_27"),
                           ("resnet34", 6, "resnet34_6"), ("resnet18", 6, "resnet18_6")]:
        shapes = ob.param_shapes(arch, cin)
        assert list(shapes.keys()) == list(g[tag + "/keys"])
        assert [str(s) for s in shapes.values()] == list(g[tag + "/shapes"])
        w = named_weights(shapes, seed=0)
        x = np.random.RandomState(100 + cin).uniform(-1, 1, size=(2, cin, 240, 320)).astype(np.float32)
        with torch.no_grad():
            if arch == "vanilla_resnet34":
                y = ob.resnet34_forward(torch.as_tensor(x), w)
            else:
                y = ob.wide_resnet_forward(torch.as_tensor(x), w, 34 if arch == "resnet34" else 18)
        np.testing.assert_allclose(y.numpy(), g[tag + "/out"], rtol=1e-4, atol=1e-4)


def test_g8_topk(golden_dir):
    pd = pytest.importorskip("pandas")
    import torch

    from happypose_amd.tensor_collection import PandasTensorCollection, filter_top_pose_estimates

    g = load(golden_dir, "g8_topk.npz")
    n = len(g["label"])
    df = pd.DataFrame({"label": g["label"], "batch_im_id": g["batch_im_id"],
                       "instance_id": g["instance_id"], "hypothesis_id": np.arange(n),
                       "coarse_logit": g["coarse_logit"]})
    coll = PandasTensorCollection(df, poses=torch.as_tensor(g["poses"]))
    for k in (1, 3, 5):
        f = filter_top_pose_estimates(coll, top_K=k, group_cols=["batch_im_id", "label", "instance_id"],
                                      filter_field="coarse_logit")
        np.testing.assert_array_equal(f.infos.hypothesis_id.values, g[f"top{k}_hyp"])
        np.testing.assert_array_equal(f.poses.numpy(), g[f"top{k}_poses"])


def test_so3_grid():
    for n in (72, 576):
        R = G.load_SO3_grid(n)
        assert R.shape == (n, 3, 3) and R.dtype == np.float32
        np.testing.assert_allclose(R @ np.swapaxes(R, 1, 2), np.tile(np.eye(3), (n, 1, 1)), atol=1e-5)
        np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)
    # w=1 quaternion -> identity
    np.testing.assert_allclose(G.unitquat_to_rotmat(np.array([0, 0, 0, 1.0])), np.eye(3))
    # 90 deg about z (xyzw)
    s = np.sqrt(0.5)
    np.testing.assert_allclose(G.unitquat_to_rotmat(np.array([0, 0, s, s])),
                               np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]), atol=1e-7)


def test_g9_efficientnet_b3_restatement(golden_dir):
    """oracle.backbones.efficientnet_b3_forward vs the reference module's own output (G9):
    key list, shapes and features must match the module exactly."""
    import torch

    from happypose_amd.synthetic import named_weights
    from oracle import backbones as ob

    g = np.load(golden_dir / "g9_efficientnet.npz")
    shapes = ob.efficientnet_b3_param_shapes(6)
    assert list(shapes.keys()) == list(g["keys"]) and [str(v) for v in shapes.values()] == list(g["shapes"])
    assert len(ob.efficientnet_b3_blocks()) == 26
    w = named_weights(shapes, seed=0)
    x = np.random.RandomState(106).uniform(-1, 1, size=(2, 6, 240, 320)).astype(np.float32)
    torch.set_num_threads(8)
    with torch.no_grad():
        y = ob.efficientnet_b3_forward(torch.as_tensor(x), w)
    assert tuple(y.shape) == tuple(g["out_shape"]) == (2, 1536, 7, 10)
    np.testing.assert_allclose(y.mean(dim=(2, 3)).numpy(), g["out_mean"], atol=1e-6)
    np.testing.assert_allclose(y.flatten()[::211].numpy(), g["out_sample"], atol=1e-6)
