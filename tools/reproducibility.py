"""Bit-reproducibility of the two-lane refiners at the benchmarked sizes (C2 WideResNet-34 / EfficientNet-b3, C3): 25 runs
of the same step must give the same bits -- the check that caught nothing after the hp_pose_prep fix;
run on the GPU box: python tools/reproducibility.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
for arch, wl in (("resnet34", "C2"), ("efficientnet-b3", "C2"), ("resnet34", "C3")):
    ds, renderer, scene, weights, model = bench.build_world(dev, arch, workload=wl, n_lanes=2)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images = torch.as_tensor(scene["images"], device=dev); K = torch.as_tensor(scene["K"], device=dev)
    T = torch.as_tensor(scene["TCO_hyp"], device=dev); labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
    im = torch.zeros(B, dtype=torch.int32, device=dev)
    ref = None; bad = 0
    for run in range(25):
        out = model.forward(images, K, labels, T, n_iterations=5, im_ids=im)
        cur = torch.stack([out[f"iteration={n}"].TCO_output for n in range(1, 6)])
        pose9 = torch.stack([out[f"iteration={n}"].network_outputs["pose"] for n in range(1, 6)])
        if ref is None: ref = (cur.clone(), pose9.clone())
        else:
            d = float((cur - ref[0]).abs().max()); dp = float((pose9 - ref[1]).abs().max())
            if d > 0 or dp > 0: bad += 1; print(arch, wl, run, "max |dT|", d, "max |dpose9|", dp, flush=True)
    print(arch, wl, "runs that differ from the first:", bad, "of 24", flush=True)
    del model, renderer
