#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` by running THE REFERENCE ITSELF.

Runs only in the build container (``/root/reference`` is read-only and does not travel to
the GPU box).  The reference's importable subset is imported from
where it lies through a ``sys.modules["happypose"]`` namespace shim; third-party modules
that are not installed (``transforms3d``, ``torchvision``, ``trimesh``, ``panda3d`` ...)
are replaced by EMPTY stub modules -- no arithmetic lives in a stub, and every function
recorded below is executed from the reference's own source.

Outputs are data only (inputs + the reference's outputs) as small ``.npz`` files:

  g1_transforms.npz   ortho6d / normalize_T / invert_transform_matrices / transform_pts
  g2_projection.npz   project_points(_robust) / boxes_from_uv / get_K_crop_resize
  g3_deepim_boxes.npz deepim_boxes (CosyPose copy == toolbox copy)
  g4_pose_update.npz  pose_update_with_reference_point / apply_imagespace_predictions /
                      TCO_init_from_boxes / _autodepth_with_R / _zup_autodepth
  g5_sampling.npz     sample_points(deterministic=True) ids; pad_stack_tensors
  g6_backbones.npz    resnet34(C_in in 9/27/32), WideResNet34/18(6) outputs on name-keyed
                      random weights (weights are REGENERATED from names, not stored)
  g7_iteration.npz    one refiner iteration without pixels (boxes -> K_crop -> update)
  g8_topk.npz         filter_top_pose_estimates / add_instance_id (pandas logic)
  obj_000001.npz      the reference's own test asset tests/data/obj_000001.ply re-encoded
                      (float32 arrays, ~40% of the ASCII size)

Usage:  python tools/gen_golden.py
"""

from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))


def _shim():
    os.environ.setdefault("HAPPYPOSE_DATA_DIR", tempfile.mkdtemp(prefix="hp_data_"))
    m = types.ModuleType("happypose")
    m.__path__ = [str(REF / "happypose")]
    m.__file__ = str(REF / "happypose" / "__init__.py")  # MP/config.py:28 derives PROJECT_ROOT from it
    sys.modules["happypose"] = m

    class _AnyMeta(type):
        """Attributes of a stub CLASS are stub classes again (``p3d.core.NodePath``, ``pin.SE3`` ...): usable in
        ``isinstance`` (never true for real data), annotations and as constructors of inert objects."""

        def __getattr__(cls, k):
            if k.startswith("__") and k.endswith("__"):
                raise AttributeError(k)
            return _AnyMeta(k, (_Any,), {})

        def __getitem__(cls, k):
            return cls

        def __iter__(cls):
            return iter(())

        def __or__(cls, other):
            return cls

        def __ror__(cls, other):
            return cls

    class _Any(metaclass=_AnyMeta):
        """An inert object (callable, subscriptable, usable as a decorator).  No arithmetic lives here: everything
        recorded in the golden files is computed by the reference's own source."""

        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            if k.startswith("__") and k.endswith("__"):
                raise AttributeError(k)
            return _Any()

        def __call__(self, *a, **k):
            if len(a) == 1 and callable(a[0]) and not k:  # used as a decorator
                return a[0]
            return _Any()

        def __getitem__(self, k):
            return _Any()

        def __iter__(self):
            return iter(())

    # third-party packages of the reference that are not installed here: every (sub)module of them resolves to an
    # EMPTY stub module through a meta-path finder
    STUB_TOPS = {"transforms3d", "torchvision", "trimesh", "pinocchio", "panda3d", "roma", "pybullet", "direct",
                 "simplejson", "omegaconf", "bokeh", "cv2", "open3d", "joblib_missing", "seaborn", "imageio", "pypng", "png",
                 "webdataset", "bop_toolkit_lib", "colorama", "ipdb", "meshcat", "plyfile", "pyarrow_missing", "xarray"}

    import importlib.abc
    import importlib.machinery

    class _StubLoader(importlib.abc.Loader):
        def create_module(self, spec):
            mod = types.ModuleType(spec.name)
            mod.__path__ = []  # a package: submodules resolve through the finder as well
            def _attr(k, _n=spec.name):
                if k.startswith("__") and k.endswith("__"):
                    raise AttributeError(k)
                return _AnyMeta(k, (_Any,), {})

            mod.__getattr__ = _attr
            return mod

        def exec_module(self, module):
            pass

    class _StubFinder(importlib.abc.MetaPathFinder):
        def find_spec(self, fullname, path, target=None):
            if fullname.split(".")[0] in STUB_TOPS:
                return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
            return None

    sys.meta_path.insert(0, _StubFinder())
    if not hasattr(np, "float_"):  # alias removed in NumPy 2 (reference pins numpy 1.x)
        np.float_ = np.float64


def imp(name):
    return importlib.import_module(name)


def main():
    import torch

    from happypose_amd.synthetic import named_weights, random_rotations

    _shim()
    torch.manual_seed(0)
    OUT.mkdir(parents=True, exist_ok=True)
    T = torch.as_tensor

    cg = imp("happypose.toolbox.lib3d.camera_geometry")
    rot = imp("happypose.toolbox.lib3d.rotations")
    tops = imp("happypose.toolbox.lib3d.transform_ops")
    cops = imp("happypose.toolbox.lib3d.cosypose_ops")
    mops = imp("happypose.toolbox.lib3d.mesh_ops")
    cp_crop = imp("happypose.pose_estimators.cosypose.cosypose.lib3d.cropping")
    cp_ops = imp("happypose.pose_estimators.cosypose.cosypose.lib3d.cosypose_ops")

    rs = np.random.RandomState(42)
    b = 16

    def rand_T(n):
        TT = np.tile(np.eye(4), (n, 1, 1))
        TT[:, :3, :3] = random_rotations(rs, n)
        TT[:, :3, 3] = np.stack([rs.uniform(-0.2, 0.2, n), rs.uniform(-0.15, 0.15, n),
                                 rs.uniform(0.3, 1.2, n)], -1)
        return TT.astype(np.float32)

    K = np.tile(np.array([[600.0, 0, 320], [0, 590.0, 240], [0, 0, 1]], np.float32), (b, 1, 1))
    K[:, 0, 2] += rs.uniform(-10, 10, b).astype(np.float32)

    # ---------------- G1
    p6 = rs.normal(size=(b, 6)).astype(np.float32)
    TT = rand_T(b)
    TT_noisy = TT.copy()
    TT_noisy[:, :3, :3] += 0.05 * rs.normal(size=(b, 3, 3)).astype(np.float32)
    pts = (rs.normal(size=(b, 50, 3)) * 0.05).astype(np.float32)
    np.savez_compressed(
        OUT / "g1_transforms.npz", p6=p6,
        R6=rot.compute_rotation_matrix_from_ortho6d(T(p6)).numpy(),
        T_noisy=TT_noisy, T_norm=tops.normalize_T(T(TT_noisy)).numpy(),
        T=TT, T_inv=tops.invert_transform_matrices(T(TT)).numpy(),
        pts=pts, pts_T=tops.transform_pts(T(TT), T(pts)).numpy(),
    )

    # ---------------- G2
    TT2 = rand_T(b)
    TT2[:4, 2, 3] = np.array([0.05, 0.0, -0.3, 0.12], np.float32)  # z < 0.1 clamp cases
    pts2 = (rs.normal(size=(b, 200, 3)) * 0.06).astype(np.float32)
    uv_r = cg.project_points_robust(T(pts2), T(K), T(TT2)).numpy()
    uv_p = cg.project_points(T(pts2[4:]), T(K[4:]), T(TT2[4:])).numpy()
    boxes = cg.boxes_from_uv(T(uv_r)).numpy()
    boxes_k = np.stack([rs.uniform(0, 300, b), rs.uniform(0, 200, b),
                        rs.uniform(320, 640, b), rs.uniform(220, 480, b)], -1).astype(np.float32)
    Kc = cg.get_K_crop_resize(T(K), T(boxes_k), orig_size=(480, 640), crop_resize=(240, 320)).numpy()
    np.savez_compressed(OUT / "g2_projection.npz", K=K, T=TT2, pts=pts2, uv_robust=uv_r,
                        uv_plain=uv_p, boxes=boxes, boxes_k=boxes_k, K_crop=Kc)

    # ---------------- G3
    center = np.stack([rs.uniform(100, 500, b), rs.uniform(80, 400, b)], -1).astype(np.float32)[:, None]
    obs = np.stack([center[:, 0, 0] - rs.uniform(10, 120, b), center[:, 0, 1] - rs.uniform(10, 90, b),
                    center[:, 0, 0] + rs.uniform(10, 120, b), center[:, 0, 1] + rs.uniform(10, 90, b)], -1).astype(np.float32)
    rend = obs + rs.normal(0, 8, size=(b, 4)).astype(np.float32)
    db = cp_crop.deepim_boxes(T(center), T(obs), T(rend), lamb=1.4, im_size=(480, 640)).numpy()
    # the documented worked example
    ex = cp_crop.deepim_boxes(T(np.array([[[300.0, 200.0]]], np.float32)),
                              T(np.array([[250.0, 150, 380, 260]], np.float32)),
                              T(np.array([[250.0, 150, 380, 260]], np.float32)),
                              lamb=1.4, im_size=(480, 640)).numpy()
    np.savez_compressed(OUT / "g3_deepim_boxes.npz", center=center, obs=obs, rend=rend, boxes=db,
                        example=ex)

    # ---------------- G4
    TT4 = rand_T(b)
    Kc4 = Kc.copy()
    pose9 = np.concatenate([np.eye(3)[:2].reshape(1, 6).repeat(b, 0) + 0.1 * rs.normal(size=(b, 6)),
                            0.05 * rs.normal(size=(b, 2)), 1 + 0.05 * rs.normal(size=(b, 1))], -1).astype(np.float32)
    dR = rot.compute_rotation_matrix_from_ortho6d(T(pose9[:, :6]))
    tCR = (TT4[:, :3, 3] + 0.02 * rs.normal(size=(b, 3))).astype(np.float32)
    upd_ref = cops.pose_update_with_reference_point(T(TT4), T(Kc4), T(pose9[:, 6:]), dR, T(tCR)).numpy()
    upd_origin = cops.pose_update_with_reference_point(T(TT4), T(Kc4), T(pose9[:, 6:]), dR,
                                                       T(TT4[:, :3, 3].copy())).numpy()
    upd_cosy = cp_ops.apply_imagespace_predictions(T(TT4), T(Kc4), T(pose9[:, 6:]), dR).numpy()
    det_boxes = np.stack([rs.uniform(50, 300, b), rs.uniform(50, 200, b),
                          rs.uniform(320, 600, b), rs.uniform(220, 440, b)], -1).astype(np.float32)
    mpts = (rs.normal(size=(b, 300, 3)) * 0.05).astype(np.float32)
    Rg = random_rotations(rs, b).astype(np.float32)
    np.savez_compressed(
        OUT / "g4_pose_update.npz", T=TT4, K_crop=Kc4, pose9=pose9, tCR=tCR, K=K,
        upd_ref=upd_ref, upd_origin=upd_origin, upd_cosy=upd_cosy, det_boxes=det_boxes,
        mpts=mpts, Rg=Rg,
        init_v0=cops.TCO_init_from_boxes(z_range=(1.0, 1.0), boxes=T(det_boxes), K=T(K)).numpy(),
        init_R=cops.TCO_init_from_boxes_autodepth_with_R(T(det_boxes), T(mpts), T(K), T(Rg)).numpy(),
        init_zup=cops.TCO_init_from_boxes_zup_autodepth(T(det_boxes), T(mpts), T(K)).numpy(),
        init_zup_cosy=cp_ops.TCO_init_from_boxes_zup_autodepth(T(det_boxes), T(mpts), T(K)).numpy(),
    )

    # ---------------- G5
    g5 = {}
    for n_pad, n_pts in [(2000, 2000), (2500, 2000), (9951, 2000), (8249, 2000), (8249, 200), (9951, 200)]:
        ar = torch.arange(n_pad, dtype=torch.float32).view(1, n_pad, 1).repeat(1, 1, 3)
        g5[f"ids_{n_pad}_{n_pts}"] = mops.sample_points(ar, n_pts, deterministic=True)[0, :, 0].numpy().astype(np.int64)
    rmd = imp("happypose.toolbox.lib3d.rigid_mesh_database")
    lens = [7, 12, 9, 12, 5]
    lst = [torch.arange(n, dtype=torch.float32).view(n, 1).repeat(1, 3) + 100 * i for i, n in enumerate(lens)]
    g5["pad_lens"] = np.asarray(lens)
    g5["pad_stack"] = rmd.pad_stack_tensors(lst, fill="select_random", deterministic=True).numpy()
    np.savez_compressed(OUT / "g5_sampling.npz", **g5)

    # ---------------- G6
    tvr = imp("happypose.pose_estimators.megapose.models.torchvision_resnet")
    wrn = imp("happypose.pose_estimators.megapose.models.wide_resnet")
    wrn_cp = imp("happypose.pose_estimators.cosypose.cosypose.models.wide_resnet")
    g6 = {}
    torch.set_num_threads(8)
    cases = [("vanilla_resnet34", 9), ("vanilla_resnet34", 27), ("vanilla_resnet34", 32),
             ("resnet34", 6), ("resnet18", 6), ("resnet34cp", 6)]
    for arch, cin in cases:
        if arch == "vanilla_resnet34":
            net = tvr.resnet34(num_classes=512, n_input_channels=cin)
        elif arch == "resnet34":
            net = wrn.WideResNet34(n_inputs=cin)
        elif arch == "resnet34cp":
            net = wrn_cp.WideResNet34(n_inputs=cin)
        else:
            net = wrn.WideResNet18(n_inputs=cin)
        sd = net.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items()}
        w = named_weights(shapes, seed=0)
        net.load_state_dict({k: T(v) for k, v in w.items()})
        net.eval()
        xin = np.random.RandomState(100 + cin).uniform(-1, 1, size=(2, cin, 240, 320)).astype(np.float32)
        acts = {}

        def hook(name):
            def f(_m, _i, o):
                acts[name] = o.detach()
            return f

        for nm in ["maxpool", "layer1", "layer2", "layer3", "layer4"]:
            getattr(net, nm).register_forward_hook(hook(nm))
        with torch.no_grad():
            y = net(T(xin))
        tag = f"{arch}_{cin}"
        g6[f"{tag}/keys"] = np.array(list(shapes.keys()))
        g6[f"{tag}/shapes"] = np.array([str(s) for s in shapes.values()])
        g6[f"{tag}/out"] = y.numpy()
        for nm, a in acts.items():
            g6[f"{tag}/{nm}_mean"] = np.array([a.double().mean().item(), a.double().abs().mean().item()])
            g6[f"{tag}/{nm}_sample"] = a[:, ::37].flatten()[::101].numpy()
        print(tag, "out", y.shape, float(y.abs().mean()))
    np.savez_compressed(OUT / "g6_backbones.npz", **g6)

    # ---------------- G7 : one refiner iteration without pixels
    TT7 = rand_T(b)
    pts7 = (rs.normal(size=(b, 2000, 3)) * 0.05).astype(np.float32)
    TTn = tops.normalize_T(T(TT7))
    tCR7 = TTn[:, :3, 3]
    uv7 = cg.project_points_robust(T(pts7), T(K), TTn)
    br7 = cg.boxes_from_uv(uv7)
    TCR = TTn.clone()
    TCR[:, :3, -1] = tCR7
    ctr7 = cg.project_points_robust(torch.zeros(b, 1, 3), T(K), TCR)
    bc7 = cp_crop.deepim_boxes(ctr7, br7, br7, lamb=1.4, im_size=(480, 640))
    Kc7 = cg.get_K_crop_resize(T(K).clone(), bc7, orig_size=(480, 640), crop_resize=(240, 320))
    p97 = pose9.copy()
    dR7 = rot.compute_rotation_matrix_from_ortho6d(T(p97[:, :6]))
    out7 = cops.pose_update_with_reference_point(TTn, Kc7, T(p97[:, 6:]), dR7, tCR7)
    np.savez_compressed(OUT / "g7_iteration.npz", T=TT7, K=K, pts=pts7, pose9=p97,
                        T_norm=TTn.numpy(), boxes_rend=br7.numpy(), boxes_crop=bc7.numpy(),
                        K_crop=Kc7.numpy(), T_out=out7.numpy())

    # ---------------- G8 : pandas top-K / instance ids
    import pandas as pd

    tcm = imp("happypose.toolbox.utils.tensor_collection")
    n = 40
    df = pd.DataFrame({
        "label": rs.choice(["a", "b", "c"], n), "batch_im_id": rs.randint(0, 2, n),
        "instance_id": rs.randint(0, 2, n), "hypothesis_id": np.arange(n),
        "coarse_logit": np.round(rs.normal(size=n), 3),
    })
    df.loc[5, "coarse_logit"] = df.loc[6, "coarse_logit"]  # a tie
    poses = rs.normal(size=(n, 4, 4)).astype(np.float32)
    coll = tcm.PandasTensorCollection(df.copy(), poses=T(poses))
    g8 = {"label": df.label.values.astype(str), "batch_im_id": df.batch_im_id.values,
          "instance_id": df.instance_id.values, "coarse_logit": df.coarse_logit.values, "poses": poses}
    for k in (1, 3, 5):
        f = tcm.filter_top_pose_estimates(coll, top_K=k, group_cols=["batch_im_id", "label", "instance_id"],
                                          filter_field="coarse_logit")
        g8[f"top{k}_hyp"] = f.infos.hypothesis_id.values
        g8[f"top{k}_poses"] = f.poses.numpy()
    np.savez_compressed(OUT / "g8_topk.npz", **g8)

    # ---------------- the reference's own test asset, re-encoded
    from happypose_amd.mesh_io import load_ply

    m = load_ply(REF / "tests" / "data" / "obj_000001.ply")
    np.savez_compressed(OUT / "obj_000001.npz", vertices=m.vertices.astype(np.float32),
                        faces=m.faces, normals=m.normals, uvs=m.uvs)
    print("golden vectors written to", OUT)
    for f in sorted(OUT.glob("*.npz")):
        print(f"  {f.name:24s} {f.stat().st_size / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
