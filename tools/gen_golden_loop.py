#!/usr/bin/env python3
"""Golden vectors of the LOOP and the ORCHESTRATOR, produced by running the reference's own code.

``tests/golden/g10_loop.npz`` holds outputs of the reference's

* ``PosePredictor.forward`` / ``forward_coarse`` (``MP/models/pose_rigid.py:546-788``),
* CosyPose ``PosePredictor.forward`` (``CP/models/pose.py:116-199``),
* ``PoseEstimator.run_inference_pipeline`` (``MP/inference/pose_estimator.py:515-668``) incl. its
  ``forward_coarse_model`` / ``forward_refiner`` / ``forward_scoring_model`` /
  ``filter_top_pose_estimates`` and the pandas bookkeeping,
* ``normalize_depth`` (``MP/models/pose_rigid.py:512-544``), ``change_keys_of_older_models``
  (``TB/utils/models_compat.py:17-27``), ``check_update_config`` (``MP/training/pose_models_cfg.py:36-86``)

imported from ``/root/reference`` through the namespace shim of ``tools/gen_golden.py`` (third-party packages
that are not installed resolve to EMPTY stub modules).  Three third-party operations the reference delegates
to packages that cannot run in this container are supplied by the CPU oracle's restatement, exactly and only
these:

  1. ``torchvision.ops.roi_align``            -> ``oracle.native.roi_align``   (plain C)
  2. ``Panda3dBatchRenderer.render``          -> ``oracle.native.rasterize``   (plain C)
  3. Panda3D ``NodePath.lookAt`` inside ``get_3_views_TCO_pos_front`` -> ``oracle.geometry.views_TC0_CV``
     (used by the 4-view cases only) and roma's ``unitquat_to_rotmat`` for the SO(3) grid
     -> ``oracle.geometry.load_SO3_grid``.

Everything else -- the iteration loop, crop-box chain, depth normalisation, channel bookkeeping and
concatenation, backbone + heads (the reference's modules), pose update, chunking, top-K filtering, ids and row
order -- is the reference's source executing.  The golden therefore pins ``oracle/pipeline.py`` and
``oracle/estimator.py`` (tests/test_oracle_golden.py) up to those three operations, which stay "parity
unpinned" (DESIGN.md section 2).

Inputs are regenerated from seeds by the tests (synthetic scene, name-keyed weights); only outputs are stored.
Runs in the build container only.  Usage:  python tools/gen_golden_loop.py
"""

from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))

import gen_golden as gg  # noqa: E402

OUT = REPO / "tests" / "golden"

# ---- the seeded world shared with tests/test_oracle_golden.py -------------------------------------------------
N_OBJ, TEX = 3, 128
CASES = {
    # name: (arch, n_in, cfg, weight seed, update scale)
    "cosy": dict(arch="resnet18", n_in=6, seed=1, scale=0.002),
    "mp_rgb1": dict(arch="vanilla_resnet34", n_in=9, seed=4, scale=0.002),
    "mp_rgbd4": dict(arch="vanilla_resnet34", n_in=32, seed=2, scale=0.002),
    "mp_rgb4": dict(arch="vanilla_resnet34", n_in=27, seed=5, scale=0.002),
    "coarse": dict(arch="vanilla_resnet34", n_in=9, seed=3, scale=1.0),
    # round 3: HIGH-GAIN heads (every iteration moves a pose by ~0.1 rad: a 1 % feature error becomes visible in the pose)
    "cosy_hi": dict(arch="resnet18", n_in=6, seed=1, scale=0.05),
    "mp_rgbd4_hi": dict(arch="vanilla_resnet34", n_in=32, seed=2, scale=0.05),
    # render_normals=False: the reference lights the scene with ITS make_scene_lights() (ambient 0.1 + six point lights
    # placed by positioning functions, MP/models/pose_rigid.py:422)
    "mp_lit": dict(arch="vanilla_resnet34", n_in=6, seed=6, scale=0.05),
}


def world():
    from happypose_amd.mesh_store import MeshDataBase, PackedMeshes
    from happypose_amd.synthetic import make_object_dataset, make_scene

    ds = make_object_dataset(N_OBJ, seed=1, tex_size=TEX)
    packed = PackedMeshes(ds)
    mesh_db = MeshDataBase.from_object_ds(ds).batched()
    scene = make_scene(n_detections=3, n_hypotheses=4, n_objects=N_OBJ, seed=2, with_depth=True)
    return ds, packed, mesh_db, scene


def case_weights(name):
    from happypose_amd.synthetic import predictor_weights
    from oracle import backbones as ob

    c = CASES[name]
    coarse = name == "coarse"
    shapes = ob.predictor_param_shapes(c["arch"], c["n_in"], pose_dim=0 if coarse else 9, n_views_logits=1 if coarse else 0)
    return predictor_weights(shapes, seed=c["seed"], update_scale=c["scale"])


def main():
    import torch

    gg._shim()
    from oracle import geometry as G
    from oracle import native

    torch.set_num_threads(8)
    T = torch.as_tensor

    # (1) torchvision.ops.roi_align  -> the oracle's C restatement
    def roi_align(inp, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
        assert spatial_scale == 1.0 and not aligned and sampling_ratio == 4
        b = boxes.detach().cpu().numpy().astype(np.float32)
        out = native.roi_align(inp.detach().cpu().numpy(), b[:, 1:], b[:, 0].astype(np.int32), tuple(output_size), 4)
        return T(out)

    import importlib

    importlib.import_module("torchvision").ops = types.SimpleNamespace(roi_align=roi_align)  # the empty stub module

    pr = gg.imp("happypose.pose_estimators.megapose.models.pose_rigid")
    cp_pose = gg.imp("happypose.pose_estimators.cosypose.cosypose.models.pose")
    tvr = gg.imp("happypose.pose_estimators.megapose.models.torchvision_resnet")
    wrn = gg.imp("happypose.pose_estimators.cosypose.cosypose.models.wide_resnet")
    rmd = gg.imp("happypose.toolbox.lib3d.rigid_mesh_database")
    rtypes = gg.imp("happypose.toolbox.renderer.types")
    p3dbr = gg.imp("happypose.toolbox.renderer.panda3d_batch_renderer")
    mv = gg.imp("happypose.toolbox.lib3d.multiview")
    mpe = gg.imp("happypose.pose_estimators.megapose.inference.pose_estimator")
    itypes = gg.imp("happypose.toolbox.inference.types")
    iutils = gg.imp("happypose.toolbox.inference.utils")
    cfgm = gg.imp("happypose.pose_estimators.megapose.training.pose_models_cfg")
    compat = gg.imp("happypose.toolbox.utils.models_compat")

    ds, packed, mesh_db_np, scene = world()
    labels_all = list(packed.labels)

    # (2) Panda3dBatchRenderer.render -> the oracle's C rasteriser.  A subclass, because the reference asserts
    # isinstance(renderer, Panda3dBatchRenderer) (MP/models/pose_rigid.py:424, CP/models/pose.py:133)
    class OracleRenderer(p3dbr.Panda3dBatchRenderer):
        def __init__(self):  # no worker processes
            pass

        def render(self, labels, TCO, K, light_datas, resolution, render_depth=False, render_binary_mask=False,
                   render_normals=False):
            obj = np.array([packed.label_to_id[l] for l in labels], np.int32)
            # lights as setup_lights applies them (TB/renderer/panda3d_scene_renderer.py:294-318): ambient colours add up,
            # a point light is placed by calling ITS positioning_function(root_node, light_node) -- the reference's own
            # function objects (make_scene_lights' pos_fn) run here on duck-typed stand-ins for the two NodePaths
            from happypose_amd.renderer import LightNodeProxy, SceneRootProxy

            n = len(labels)
            amb = np.zeros((n, 3), np.float32)
            n_pts = max(sum(1 for l in ls if l.light_type == "point") for ls in light_datas)
            lp = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
            lc = np.zeros((n, n_pts, 3), np.float32) if n_pts else None
            for i, ls in enumerate(light_datas):
                k = 0
                for l in ls:
                    if l.light_type == "ambient":
                        amb[i] += np.asarray(l.color[:3], np.float32)
                    else:
                        assert l.light_type == "point" and l.positioning_function is not None
                        node = LightNodeProxy()
                        l.positioning_function(SceneRootProxy(packed.bounds_center[obj[i]], packed.bounds_radius[obj[i]]), node)
                        lp[i, k], lc[i, k] = node.pos, l.color[:3]
                        k += 1
            r = native.rasterize(packed, obj, TCO.detach().cpu().numpy(), K.detach().cpu().numpy(), tuple(resolution),
                                 render_normals, render_depth, render_binary_mask, ambient=amb, light_pos=lp, light_col=lc,
                                 msaa=True, aniso=True)  # the reference's render state (TB/renderer/panda3d_scene_renderer.py:68-71)
            return rtypes.BatchRenderOutput(
                rgbs=T(r["rgbs"]), normals=None if r["normals"] is None else T(r["normals"]),
                depths=None if r["depths"] is None else T(r["depths"]),
                binary_masks=None if r["binary_masks"] is None else T(r["binary_masks"]))

        def stop(self):
            pass

    # (3) Panda3D lookAt of the extra views -> the oracle's closed form
    def get_3_views(TCO, tCR):
        return list(G.views_TC0_CV(tCR, "TCO+front_3views")[1:])

    mv.get_3_views_TCO_pos_front = get_3_views

    # the reference's own BatchedMeshes over the padded point table (the table itself is pinned by G5)
    pts = T(np.asarray(mesh_db_np.points, np.float32))
    sym = torch.eye(4).repeat(len(labels_all), 1, 1, 1)
    ref_mesh_db = rmd.BatchedMeshes({l: {} for l in labels_all}, labels_all, pts, sym).float()
    renderer = OracleRenderer()

    def load(model, w):
        sd = {k: T(np.asarray(v)) for k, v in w.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
        return model.eval()

    def mp_model(name, **kw):
        c = CASES[name]
        bb = tvr.resnet34(num_classes=512, n_input_channels=c["n_in"])
        bb.n_features = 512
        return load(pr.PosePredictor(backbone=bb, renderer=renderer, mesh_db=ref_mesh_db, **kw), case_weights(name))

    images = T(scene["images"])          # [1,4,480,640] rgb + depth
    K = T(scene["K"])
    g = {}

    def rec_iters(tag, outs, n_it, fields=("TCO_output", "TCO_input", "K_crop", "boxes_rend", "boxes_crop")):
        for n in range(1, n_it + 1):
            o = outs[f"iteration={n}"]
            for f in fields:
                g[f"{tag}/it{n}/{f}"] = getattr(o, f).detach().numpy()

    with torch.no_grad():
        # ---- CosyPose refiner: WideResNet-18, 6 channels, 2 iterations (CP/models/pose.py:116-199)
        sel = np.arange(0, 12, 2)
        nb = len(sel)
        lab = [labels_all[i] for i in scene["hyp_obj_ids"][sel]]
        bbm = wrn.WideResNet18(n_inputs=6)
        cosy = load(cp_pose.PosePredictor(backbone=bbm, renderer=renderer, mesh_db=ref_mesh_db), case_weights("cosy"))
        outs = cosy(images[:, :3].expand(nb, -1, -1, -1), K.expand(nb, -1, -1), lab, T(scene["TCO_hyp"][sel]), n_iterations=2)
        rec_iters("cosy", outs, 2)
        g["cosy/it2/pose"] = outs["iteration=2"].model_outputs["pose"].numpy()
        g["cosy/sel"] = sel

        # ---- MegaPose, one view, RGB + normals (9 channels), 2 iterations
        m1 = mp_model("mp_rgb1", multiview_type="TCO", n_rendered_views=1, render_normals=True)
        outs = m1(images.expand(nb, -1, -1, -1), K.expand(nb, -1, -1), lab, T(scene["TCO_hyp"][sel]), n_iterations=2)
        rec_iters("mp_rgb1", outs, 2)

        # ---- MegaPose RGB-D, 4 views x (rgb + normals + depth) = 32 channels, depth normalisation, 2 iterations
        sel4 = np.array([0, 5, 10])
        lab4 = [labels_all[i] for i in scene["hyp_obj_ids"][sel4]]
        m4 = mp_model("mp_rgbd4", multiview_type="TCO+front_3views", n_rendered_views=4, render_normals=True, render_depth=True,
                      input_depth=True, depth_normalization_type="tCR_scale_clamp_center")
        outs = m4(images.expand(3, -1, -1, -1), K.expand(3, -1, -1), lab4, T(scene["TCO_hyp"][sel4]), n_iterations=2)
        rec_iters("mp_rgbd4", outs, 2, ("TCO_output", "TCO_input", "K_crop", "boxes_rend", "boxes_crop", "KV_crop", "TCV_O_input"))
        x = torch.cat((outs["iteration=1"].images_crop, outs["iteration=1"].renders), 1)  # the network input of iteration 1
        g["mp_rgbd4/it1/x_shape"] = np.array(x.shape)
        g["mp_rgbd4/it1/x_chan_mean"] = x.double().mean(dim=(0, 2, 3)).numpy()
        g["mp_rgbd4/it1/x_chan_absmean"] = x.double().abs().mean(dim=(0, 2, 3)).numpy()
        g["mp_rgbd4/it1/x_sample"] = x[:, :, ::7, ::11].numpy()
        g["mp_rgbd4/sel"] = sel4

        # ---- round 3: the same two refiners with HIGH-GAIN heads, 3 iterations (a pose moves ~0.1 rad per iteration)
        cosy_hi = load(cp_pose.PosePredictor(backbone=wrn.WideResNet18(n_inputs=6), renderer=renderer, mesh_db=ref_mesh_db),
                       case_weights("cosy_hi"))
        outs = cosy_hi(images[:, :3].expand(nb, -1, -1, -1), K.expand(nb, -1, -1), lab, T(scene["TCO_hyp"][sel]), n_iterations=3)
        rec_iters("cosy_hi", outs, 3)
        m4h = mp_model("mp_rgbd4_hi", multiview_type="TCO+front_3views", n_rendered_views=4, render_normals=True, render_depth=True,
                       input_depth=True, depth_normalization_type="tCR_scale_clamp_center")
        outs = m4h(images.expand(3, -1, -1, -1), K.expand(3, -1, -1), lab4, T(scene["TCO_hyp"][sel4]), n_iterations=3)
        rec_iters("mp_rgbd4_hi", outs, 3)

        # ---- round 3: render_normals=False -> the reference's own make_scene_lights() and its positioning functions
        ml = mp_model("mp_lit", multiview_type="TCO", n_rendered_views=1, render_normals=False)
        outs = ml(images[:, :3].expand(nb, -1, -1, -1), K.expand(nb, -1, -1), lab, T(scene["TCO_hyp"][sel]), n_iterations=2)
        rec_iters("mp_lit", outs, 2)
        g["mp_lit/it1/renders_sample"] = outs["iteration=1"].renders[:, :, ::7, ::11].numpy()
        g["mp_lit/it1/renders_mean"] = outs["iteration=1"].renders.double().mean(dim=(0, 2, 3)).numpy()
        psr = gg.imp("happypose.toolbox.renderer.panda3d_scene_renderer")
        from happypose_amd.renderer import LightNodeProxy, SceneRootProxy
        lights = psr.make_scene_lights()
        g["lights/types"] = np.array([l.light_type for l in lights])
        g["lights/colors"] = np.array([l.color for l in lights], np.float32)
        pos = []
        for l in lights[1:]:
            node = LightNodeProxy()
            l.positioning_function(SceneRootProxy((0.0, 0.0, 0.0), 0.25), node)
            pos.append(node.pos)
        g["lights/pos_r025"] = np.array(pos, np.float32)

        # ---- coarse model: forward_coarse logits (MP/models/pose_rigid.py:708-788)
        selc = np.array([0, 4, 8, 9, 2, 7])
        labc = [labels_all[i] for i in scene["hyp_obj_ids"][selc]]
        mc = mp_model("coarse", multiview_type="TCO", n_rendered_views=1, render_normals=True, predict_pose_update=False,
                      predict_rendered_views_logits=True)
        oc = mc.forward_coarse(images[:, :3].expand(6, -1, -1, -1), K.expand(6, -1, -1), labc, T(scene["TCO_hyp"][selc]))
        g["coarse/logits"] = oc["logits"].numpy()
        g["coarse/scores"] = oc["scores"].numpy()
        g["coarse/sel"] = selc

        # ---- the whole MegaPose pipeline (MP/inference/pose_estimator.py:515-668): SO(3) grid of 72, 2 detections,
        # top-2 hypotheses, 2 refiner iterations with the 4-view RGB refiner, re-scoring, top-1
        mr = mp_model("mp_rgb4", multiview_type="TCO+front_3views", n_rendered_views=4, render_normals=True)
        mc.cfg = mr.cfg = types.SimpleNamespace()
        est = mpe.PoseEstimator(refiner_model=mr, coarse_model=mc, bsz_objects=8, bsz_images=64, SO3_grid_size=None)
        est._SO3_grid = T(G.load_SO3_grid(72))
        det_ids = np.array([0, 2])
        ptsd = np.asarray(mesh_db_np.points, np.float32)[scene["det_obj_ids"][det_ids]]
        boxes = G.boxes_from_uv(G.project_points(ptsd, np.repeat(scene["K"], 2, 0), scene["TCO_det"][det_ids]))
        det = iutils.make_detections_from_object_data  # needs ObjectData: build the collection directly instead
        tcm = gg.imp("happypose.toolbox.utils.tensor_collection")
        import pandas as pd

        infos = pd.DataFrame({"label": [labels_all[i] for i in scene["det_obj_ids"][det_ids]], "batch_im_id": 0,
                              "instance_id": np.arange(2)})
        detections = tcm.PandasTensorCollection(infos=infos, bboxes=T(boxes).float())
        obs = itypes.ObservationTensor(images[:, :3].contiguous(), K)
        final, extra = est.run_inference_pipeline(obs, detections=detections, n_refiner_iterations=2, n_pose_hypotheses=2)
        g["e2e/det_ids"] = det_ids
        g["e2e/boxes"] = boxes
        g["e2e/coarse_logits"] = extra["coarse"]["data"]["logits"].numpy()
        g["e2e/coarse_TCO"] = extra["coarse"]["preds"].poses.numpy()
        cf = extra["coarse_filter"]["preds"]
        g["e2e/filtered_hyp"] = cf.infos.hypothesis_id.values
        g["e2e/filtered_label"] = cf.infos.label.values.astype(str)
        g["e2e/filtered_TCO"] = cf.poses.numpy()
        for n in (1, 2):
            g[f"e2e/refined_it{n}"] = extra["refiner_all_hypotheses"]["preds"][f"iteration={n}"].poses.numpy()
        sc = extra["scoring"]["preds"]
        g["e2e/pose_logit"] = sc.infos.pose_logit.values
        g["e2e/final_hyp"] = final.infos.hypothesis_id.values
        g["e2e/final_label"] = final.infos.label.values.astype(str)
        g["e2e/final_instance"] = final.infos.instance_id.values
        g["e2e/final_TCO"] = final.poses.numpy()
        g["e2e/final_columns"] = np.array(list(final.infos.columns))
        g["e2e/extra_keys"] = np.array(sorted(extra.keys()))
        g["e2e/final_tensors"] = np.array(sorted(final.tensors.keys()))

        # ---- normalize_depth, all modes (MP/models/pose_rigid.py:512-544): the method run unbound on a bag of attributes
        rs = np.random.RandomState(7)
        d = rs.uniform(0.0, 2.5, size=(4, 2, 6, 5)).astype(np.float32)
        d[0, 0, :2] = 0.0
        z = rs.uniform(0.4, 1.3, size=(4, 3)).astype(np.float32)
        g["depthnorm/d"], g["depthnorm/tCR"] = d, z
        for mode in ("tCR_scale", "tCR_scale_clamp_center", "tCR_center_clamp", "none"):
            self_ = types.SimpleNamespace(depth_normalization_type=mode)
            g[f"depthnorm/{mode}"] = pr.PosePredictor.normalize_depth(self_, T(d.copy()), T(z)).numpy()

    # ---- legacy checkpoint keys and configuration defaults
    keys = ["backbone.backbone.conv1.weight", "backbone.backbone.layer1.0.bn1.running_var", "backbone.head.0.weight",
            "backbone.head.0.bias", "backbone.conv1.weight", "pose_fc.weight", "views_logits_head.bias",
            "backbone.backbone.fc.bias", "backbone.headx.weight"]
    new = compat.change_keys_of_older_models({k: i for i, k in enumerate(keys)})
    g["compat/keys_in"] = np.array(keys)
    g["compat/keys_out"] = np.array(list(new.keys()))
    g["compat/vals_out"] = np.array(list(new.values()))
    import argparse
    import json

    cfgs = [
        dict(input_strategy="input=obs+one_render", backbone_str="vanilla_resnet34", render_normals=True),
        dict(n_views=4, multiview_type="front_3views", render_normals=True, backbone_str="vanilla_resnet34"),
        dict(n_rendered_views=4, multiview_type="TCO+front_3views", render_normals=True, render_depth=True, input_depth=True,
             depth_augmentation=True, backbone_str="vanilla_resnet34"),
        dict(backbone_str="resnet34", multiview_type="TCO", views_inplane_rotation=False),
        dict(n_views=2, multiview_type="front_1view", depth_augmentation=False, depth_normalization_type="tCR_scale"),
    ]
    outs_cfg = []
    for c in cfgs:
        ns = cfgm.check_update_config(argparse.Namespace(**c))  # what the training code dumps to config.yaml
        outs_cfg.append({k: v for k, v in sorted(vars(ns).items())})
    g["cfg/in"] = np.array(json.dumps(cfgs))
    g["cfg/out"] = np.array(json.dumps(outs_cfg))

    np.savez_compressed(OUT / "g10_loop.npz", **g)
    print("written", OUT / "g10_loop.npz", f"{(OUT / 'g10_loop.npz').stat().st_size / 1024:.1f} KiB")
    print("e2e final hyp", g["e2e/final_hyp"], "labels", g["e2e/final_label"], "pose_logit", g["e2e/pose_logit"])


if __name__ == "__main__":
    main()
