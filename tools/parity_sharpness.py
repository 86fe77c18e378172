#!/usr/bin/env python3
"""How sharp are the pose-level parity checks?  (VERDICT r2, weak #1-#3.)  Run on the GPU box:

  python tools/parity_sharpness.py [--workload C2|C3] [--scale 0.05] > gpurun_out/sharpness.json

For the benchmarked world with the pose head scaled by ``--scale`` it prints, per iteration,
  * the error of the HIP path against the CPU oracle (max |dt| m, max geodesic rad) and the size of the update itself,
  * the same with ONE conv layer's weights of the HIP model multiplied by 1.01 (an injected 1 % error),
  * the same for the reference render state (msaa + aniso on) against the default state (HIP vs HIP),
so that the tolerances written in tests/test_gpu_pipeline.py sit between the legitimate error and the injected one.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def pose_err(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    dt = np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1)
    chord = np.linalg.norm(A[:, :3, :3] - B[:, :3, :3], axis=(1, 2))
    ang = 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
    return float(dt.max()), float(ang.max()), float(np.median(dt)), float(np.median(ang))


def run(model, scene, store, dev, n_it):
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    out = model.forward(images, K, labels, torch.as_tensor(scene["TCO_hyp"], device=dev), n_iterations=n_it,
                        im_ids=torch.zeros(B, dtype=torch.int32, device=dev))
    assert model.numerics_status() == 0
    return [out[f"iteration={n + 1}"].TCO_output.cpu().numpy() for n in range(n_it)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layer", default="backbone.layer2.1.conv1.weight")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--no-render-state", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from happypose_amd.models import create_model_pose, create_pose_model_cosypose
    from oracle.pipeline import OraclePredictor

    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload=args.workload, n_lanes=2,
                                                            update_scale=args.scale)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    res = {"workload": args.workload, "update_scale": args.scale, "hypotheses": B}
    got = run(model, scene, store, dev, args.iters)
    T0 = scene["TCO_hyp"]
    res["update_vs_input"] = [pose_err(g, T0) for g in got]
    res["update_per_iteration"] = [pose_err(g, p) for g, p in zip(got, [T0] + got[:-1])]
    if not args.no_oracle:
        torch.set_num_threads(bench.effective_cpu_count())
        if args.workload == "C2":
            ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="resnet34", cosypose=True)
            imgs = scene["images"][:, :3]
        else:
            ora = OraclePredictor(weights, store.packed, store.mesh_db.points, arch="vanilla_resnet34", n_views=4,
                                  multiview_type="TCO+front_3views", render_normals=True, render_depth=True, input_depth=True,
                                  depth_normalization_type="tCR_scale_clamp_center")
            imgs = scene["images"]
        ref = ora.forward(imgs, scene["K"], np.zeros(B, np.int32), scene["hyp_obj_ids"], scene["TCO_hyp"], args.iters, bsz_objects=8)
        res["hip_vs_oracle"] = [pose_err(g, r["TCO_output"]) for g, r in zip(got, ref)]
    # injected 1 % error in one conv layer (HIP model only)
    w_bad = dict(weights)
    w_bad[args.layer] = (np.asarray(weights[args.layer]) * 1.01).astype(np.float32)
    if args.workload == "C2":
        bad = create_pose_model_cosypose(dict(backbone_str="resnet34"), renderer, state_dict=w_bad, max_batch=B, n_lanes=2)
    else:
        bad = create_model_pose(model.cfg, renderer, state_dict=w_bad, max_batch=B, n_lanes=2)
    got_bad = run(bad, scene, store, dev, args.iters)
    res["injected_1pct_vs_clean_hip"] = [pose_err(a, b) for a, b in zip(got_bad, got)]
    if not args.no_oracle:
        res["injected_1pct_vs_oracle"] = [pose_err(g, r["TCO_output"]) for g, r in zip(got_bad, ref)]
    del bad
    if not args.no_render_state:
        for name, kw in (("msaa", dict(msaa=True)), ("aniso", dict(aniso=True)), ("msaa+aniso", dict(msaa=True, aniso=True))):
            _, r2, _, _, m2 = bench.build_world(dev, "resnet34", seed=0, workload=args.workload, n_lanes=2, update_scale=args.scale,
                                                renderer_kw=kw)
            got2 = run(m2, scene, r2.store, dev, args.iters)
            res[f"render_state_{name}_vs_default"] = [pose_err(a, b) for a, b in zip(got2, got)]
            del m2, r2
    res["columns"] = ["max_dt_m", "max_dR_rad", "median_dt_m", "median_dR_rad"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
