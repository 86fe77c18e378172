"""Rasteriser timing on the C2 inputs with and without HP_RASTER_MSAA4 / HP_RASTER_TEX_ANISO (product layout: NHWC slice of the network input)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from happypose_amd import ops
dev = torch.device("cuda:0")
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet18", workload="C2")
store = renderer.store
K = torch.as_tensor(scene["K"], device=dev); T = torch.as_tensor(scene["TCO_hyp"], device=dev)
B = len(T); im = torch.zeros(B, dtype=torch.int32, device=dev); obj = torch.as_tensor(scene["hyp_obj_ids"], device=dev)
prep = ops.pose_prep(store, T, K, im, obj, (480, 640))
x = torch.zeros((B, 240, 320, 8), device=dev)
for msaa, aniso in ((False, False), (True, False), (False, True), (True, True)):
    for _ in range(3): ops.rasterize_into(store, x, 3, obj, prep["TCV_O"], prep["K_crop"], False, False, msaa=msaa, aniso=aniso)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.rasterize_into(store, x, 3, obj, prep["TCV_O"], prep["K_crop"], False, False, msaa=msaa, aniso=aniso)
    e1.record(); torch.cuda.synchronize()
    print(("msaa4" if msaa else "single-sample") + (" + mipmap/aniso16" if aniso else " + bilinear"), e0.elapsed_time(e1) / 20 * 1e3, "us for", B, "views")
