"""Rasteriser + crop on the C2 and C3 inputs in the product layout, a few repetitions each: the command the rocprofv3
PMC passes of profiles/r03_raster_* run (tools/profile_r03.sh).  Prints the event-timed durations and the ALGORITHMIC
bytes per call as JSON so that counter bytes / algorithmic bytes can be stated."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from happypose_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(os.environ.get("HP_STAGE_REPS", "5"))
flags = dict(msaa=bool(int(os.environ.get("HP_STAGE_MSAA", "0"))), aniso=bool(int(os.environ.get("HP_STAGE_ANISO", "0"))))
only = os.environ.get("HP_STAGE_ONLY", "")  # "raster": time / launch the rasteriser stage alone (per-workload PMC passes)
out = {}
for wl in os.environ.get("HP_STAGE_WORKLOADS", "C2,C3").split(","):
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload=wl, n_lanes=1)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
    T = torch.as_tensor(scene["TCO_hyp"], device=dev)
    obj = torch.as_tensor(scene["hyp_obj_ids"], device=dev)
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    c2 = wl == "C2"
    prep = ops.pose_prep(store, T, K, im_ids, obj, (480, 640), multiview_type="TCO" if c2 else "TCO+front_3views", normalize=not c2)
    V = prep["TCV_O"].shape[1]
    n_img, c_r = (3, 3) if c2 else (4, 7)
    x = model.backbone.new_input(B)
    z = None if c2 else prep["tCR"][:, 2].contiguous()
    mode = 0 if c2 else 2

    def crop():
        ops.crop_roi_align(images, prep["boxes_crop"], im_ids, (240, 320), out=x, depth_norm_z=z, depth_norm_mode=mode if n_img == 4 else 0,
                           n_channels=n_img, owns_record=True)

    def raster():
        ops.rasterize_into(store, x, n_img, obj, prep["TCV_O"], prep["K_crop"], not c2, not c2, z, mode, **flags)

    def fused():  # what the predictors launch: crop + views, every pixel record written once
        ops.render_inputs(store, x, obj, prep["TCV_O"], prep["K_crop"], not c2, not c2, images=images, boxes=prep["boxes_crop"], im_ids=im_ids,
                          n_img_channels=n_img, depth_norm_z=z, depth_norm_mode=mode, **flags)

    depth = ops.rasterize(store, obj.repeat_interleave(V), prep["TCV_O"].reshape(-1, 4, 4), prep["K_crop"].reshape(-1, 3, 3), (240, 320),
                          render_depth=True)[2]
    covered = float((depth > 0).sum().item())
    rows = store.packed.obj[np.repeat(np.asarray(scene["hyp_obj_ids"]), V)]
    raster_bytes = float((rows[:, 1] * 32 + rows[:, 3] * 12).sum()) + B * V * 76800 * c_r * 4 + covered * 4
    crop_bytes = float(B * 76800 * n_img * 4)
    res = {}
    for name, fn, nbytes in (("crop", crop, crop_bytes), ("raster", raster, raster_bytes), ("render_inputs", fused, raster_bytes + crop_bytes)):
        if only and name != only:
            continue
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res[name] = {"us": us, "algorithmic_MB": nbytes / 1e6, "GB/s": nbytes / us / 1e3, "frac_hbm_peak": nbytes / (us * 1e-6) / 8e12}
    res["views"], res["coverage"], res["calls_each"] = B * V, covered / (B * V * 76800), reps + 2
    out[wl] = res
    del model
print(json.dumps(out))
