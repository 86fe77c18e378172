"""Phases of one CosyPoseEstimator.forward_refiner call on the C2 job (round 6): host time before the predictor is entered, inside it
(all launches enqueued), table assembly, waiting in the guard's stream synchronisation, and after it -- medians over 30 calls.
`gpurun -- python tools/probes/estimator_phases.py`"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=2)
    store = renderer.store
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    TCO0 = torch.as_tensor(scene["TCO_hyp"], device=dev)
    step = bench.estimator_entry(model, images, K, labels, TCO0, dev)
    marks = {}
    fwd, status = model.forward, model.numerics_status
    from happypose_amd import ops
    prep = ops.pose_prep

    def fwd_w(*a, **k):
        marks["fwd_in"] = time.perf_counter()
        r = fwd(*a, **k)
        marks["fwd_out"] = time.perf_counter()
        return r

    def status_w():
        marks["status_in"] = time.perf_counter()
        r = status()
        marks["status_out"] = time.perf_counter()
        return r

    def prep_w(*a, **k):
        marks.setdefault("first_launch", time.perf_counter())
        return prep(*a, **k)

    model.forward = fwd_w
    model.__class__.__call__ = lambda self, *a, **k: self.forward(*a, **k)
    model.numerics_status = status_w
    ops.pose_prep = prep_w
    rows = []
    for i in range(40):
        marks.clear()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        if i >= 10 and "fwd_in" in marks:
            rows.append([1e3 * (marks["fwd_in"] - t0), 1e3 * (marks.get("first_launch", marks["fwd_in"]) - marks["fwd_in"]), 1e3 * (marks["fwd_out"] - marks["fwd_in"]),
                         1e3 * (marks["status_in"] - marks["fwd_out"]), 1e3 * (marks["status_out"] - marks["status_in"]), 1e3 * (t1 - marks["status_out"]), 1e3 * (t1 - t0)])
    r = np.median(np.array(rows), axis=0)
    names = ["before_predictor", "predictor_entry_to_first_launch", "inside_predictor_enqueue", "table_assembly", "guard_wait", "after_guard", "call_total"]
    print(json.dumps({n: round(float(v), 3) for n, v in zip(names, r)}))


if __name__ == "__main__":
    main()
