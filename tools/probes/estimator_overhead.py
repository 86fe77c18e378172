"""Where the entry point's overhead goes (round 6): CosyPoseEstimator.forward_refiner on the C2 job against the bare model.forward,
host-side: (1) wall per step of both, (2) the host time between the end of a step (its stream synchronisation) and the first
launch of the next one, (3) cProfile of 30 estimator steps.  `gpurun -- python tools/probes/estimator_overhead.py`."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import pandas as pd

    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor
    from happypose_amd.tensor_collection import PandasTensorCollection

    dev = torch.device("cuda:0")
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    TCO0 = torch.as_tensor(scene["TCO_hyp"], device=dev)
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    est = CosyPoseEstimator(refiner_model=model, coarse_model=model, bsz_objects=B)
    obs = ObservationTensor(images, K)
    infos = pd.DataFrame({"label": list(labels), "batch_im_id": np.zeros(B, dtype=np.int64), "instance_id": np.arange(B) // 16, "hypothesis_id": np.arange(B) % 16})
    data = PandasTensorCollection(infos=infos, poses=TCO0)

    def direct():
        return model.forward(images, K, labels, TCO0, n_iterations=5, im_ids=im_ids)

    def entry():
        return est.forward_refiner(obs, data, n_iterations=5)

    def wall(fn, n, sync_each):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            if sync_each:
                torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        return 1e3 * (time.perf_counter() - t0) / n

    out = {}
    for rep in range(2):
        out[f"direct_ms_{rep}"] = wall(direct, 20, False)
        out[f"direct_sync_each_ms_{rep}"] = wall(direct, 20, True)
        out[f"entry_ms_{rep}"] = wall(entry, 20, False)
    # host time of one call while the GPU is idle = what a per-step synchronisation exposes
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); direct(); out["direct_host_enqueue_ms"] = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize(dev)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        entry()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(json.dumps(out))
    print(s.getvalue())


if __name__ == "__main__":
    main()
