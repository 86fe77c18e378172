"""GPU probe behind the C5 / fp16 tolerances of tests/test_gpu_pipeline.py (round 6): what the fp16 coarse plan's logits and
features look like against the fp32 CPU oracle at the benchmarked batch (576 grid poses of one object), healthy and with one
conv layer's weights x 1.01, in units of the oracle's own spread.  Prints one JSON line; `gpurun -- python tools/probes/c5_parity_probe.py`."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from happypose_amd import ops
    from happypose_amd.models import create_model_pose
    from oracle import backbones as ob
    from oracle.pipeline import OraclePredictor

    dev = torch.device("cuda:0")
    out = {}
    for prec in ("f16", "f32"):
        ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C5", precision=prec, n_lanes=1)
        store = renderer.store
        sl = slice(0, 576)
        images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
        labels = [store.labels[i] for i in scene["hyp_obj_ids"][sl]]
        T = torch.as_tensor(scene["TCO_hyp"][sl], device=dev)
        im0 = torch.zeros(576, dtype=torch.int32, device=dev)
        got = model.forward_coarse(images, K, labels, T, im_ids=im0)["logits"].cpu().numpy().reshape(-1)
        if prec == "f16":
            torch.set_num_threads(bench.effective_cpu_count())
            ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="vanilla_resnet34", render_normals=True)
            ref = np.concatenate([ora.forward_coarse(scene["images"][:, :3], scene["K"], np.zeros(64, np.int32), scene["hyp_obj_ids"][s:s + 64],
                                                     scene["TCO_hyp"][s:s + 64])["logits"].reshape(-1) for s in range(0, 576, 64)])
            out["ref"] = dict(min=float(ref.min()), max=float(ref.max()), std=float(ref.std()), top_gaps=np.diff(np.sort(ref)[::-1][:8]).tolist())
        err = np.abs(got - ref)
        cen = lambda g: np.abs((g - g.mean()) - (ref - ref.mean()))
        out[prec] = dict(max_over_std=float(err.max() / ref.std()), rms_over_std=float(np.sqrt((err ** 2).mean()) / ref.std()),
                         centered_max_over_std=float(cen(got).max() / ref.std()),
                         top5_equal=bool(set(np.argsort(-got)[:5].tolist()) == set(np.argsort(-ref)[:5].tolist())))
        # one conv layer x 1.01 in the HIP model only
        w_bad = dict(weights)
        w_bad["backbone.layer2.1.conv1.weight"] = (np.asarray(weights["backbone.layer2.1.conv1.weight"]) * 1.01).astype(np.float32)
        cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                   predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
        bad = create_model_pose(cfg, renderer, state_dict=w_bad, max_batch=576, precision=prec, n_lanes=1)
        gb = bad.forward_coarse(images, K, labels, T, im_ids=im0)["logits"].cpu().numpy().reshape(-1)
        eb = np.abs(gb - ref)
        out[prec + "_mutated"] = dict(max_over_std=float(eb.max() / ref.std()), rms_over_std=float(np.sqrt((eb ** 2).mean()) / ref.std()),
                                      centered_max_over_std=float(cen(gb).max() / ref.std()))
        # features at batch 576 on the plan's own tiles vs oracle/backbones.py on the same network input
        lane = model.lanes[0] if hasattr(model, "lanes") else model
        im_ids, obj_ids = lane._ids(images, K, labels, im0)
        kw = dict(n_img_channels=lane._n_img, multiview_type="TCO", normalize=True, render_normals=lane.render_normals,
                  render_depth=lane.render_depth, depth_mode=lane._depth_mode, want_pose=False, want_logits=True)
        _, x, _, _, _ = lane._one_pass(images, K, im_ids, obj_ids, T, **kw)
        x = x.clone()
        n_in = lane.backbone.n_inputs
        feats = lane.backbone.forward(x, want_pose=False, want_logits=False, want_features=True)[2].float().cpu().numpy()
        sub = np.arange(0, 576, 4)
        x_nchw = x[..., :n_in].float().permute(0, 3, 1, 2)[torch.as_tensor(sub, device=dev)].contiguous().cpu()
        wt = {k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()}
        with torch.no_grad():
            fref = torch.cat([ob.net_forward(x_nchw[i:i + 16], wt, "vanilla_resnet34", heads=("features",))["features"]
                              for i in range(0, len(sub), 16)]).numpy()
        scale = np.abs(fref).max(axis=1, keepdims=True)
        fe = np.abs(feats[sub] - fref) / scale
        fb = bad.lanes[0] if hasattr(bad, "lanes") else bad
        fbad = fb.backbone.forward(x, want_pose=False, want_logits=False, want_features=True)[2].float().cpu().numpy()
        fbe = np.abs(fbad[sub] - fref) / scale
        rel2 = lambda a: float(np.sqrt(((a - fref) ** 2).sum(1) / (fref ** 2).sum(1)).max())
        out[prec + "_features"] = dict(max=float(fe.max()), mean=float(fe.mean()), l2=rel2(feats[sub]), mutated_max=float(fbe.max()), mutated_mean=float(fbe.mean()),
                                       mutated_l2=rel2(fbad[sub]), feat_std_over_max=float((fref.std(axis=0) / np.abs(fref).max()).mean()))
        del model, bad
    print(json.dumps(out))


if __name__ == "__main__":
    main()
