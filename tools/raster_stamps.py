"""Phase clocks of the band kernel (a -DHP_RASTER_STAMPS build, tools/ab_local.sh raster.hip "RSTAMP:-DHP_RASTER_STAMPS"):
shader cycles per workgroup between the phases, C2 / C3 inputs, reference state (HP_STAGE_MSAA / HP_STAGE_ANISO as in
tools/stage_workload.py).  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/RSTAMP.so python3 tools/raster_stamps.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from happypose_amd import _ffi, ops  # noqa: E402

dev = torch.device("cuda:0")
flags = dict(msaa=bool(int(os.environ.get("HP_STAGE_MSAA", "1"))), aniso=bool(int(os.environ.get("HP_STAGE_ANISO", "1"))))
lib = C.CDLL(os.environ["HAPPYPOSE_AMD_LIB"])
names = ["head", "walk", "big", "compact", "shade", "resolve", "output"]
out = {}
for wl in os.environ.get("HP_STAGE_WORKLOADS", "C2,C3").split(","):
    ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload=wl, n_lanes=1)
    store = renderer.store
    B = len(scene["TCO_hyp"])
    K = torch.as_tensor(scene["K"], device=dev)
    T = torch.as_tensor(scene["TCO_hyp"], device=dev)
    obj = torch.as_tensor(scene["hyp_obj_ids"], device=dev)
    im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    c2 = wl == "C2"
    prep = ops.pose_prep(store, T, K, im_ids, obj, (480, 640), multiview_type="TCO" if c2 else "TCO+front_3views", normalize=not c2)
    x = model.backbone.new_input(B)
    z = None if c2 else prep["tCR"][:, 2].contiguous()

    def raster():
        ops.rasterize_into(store, x, 3 if c2 else 4, obj, prep["TCV_O"], prep["K_crop"], not c2, not c2, z, 0 if c2 else 2, **flags)

    raster()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.hp_debug_raster_stamps(buf, 1)
    reps = 4
    for _ in range(reps):
        raster()
    torch.cuda.synchronize()
    lib.hp_debug_raster_stamps(buf, 1)
    n_wg = buf[15] / reps
    res = {n: round(buf[k] / max(buf[15], 1)) for k, n in enumerate(names)}
    res["workgroups_per_launch"] = n_wg
    res["cycles_per_workgroup"] = sum(buf[k] for k in range(7)) / max(buf[15], 1)
    out[wl] = res
    del model
print(json.dumps(out))
