"""A/B: crop into the 8-float NHWC record as a 12-B partial store vs a whole 32-B sector (HP_CROP_FULL_RECORD8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from happypose_amd import ops
dev = torch.device("cuda:0")
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet18", workload="C2")
store = renderer.store
images = torch.as_tensor(scene["images"], device=dev); K = torch.as_tensor(scene["K"], device=dev); T = torch.as_tensor(scene["TCO_hyp"], device=dev)
B = len(T); im = torch.zeros(B, dtype=torch.int32, device=dev); obj = torch.as_tensor(scene["hyp_obj_ids"], device=dev)
prep = ops.pose_prep(store, T, K, im, obj, (480, 640))
x = torch.zeros((B, 240, 320, 8), device=dev)
def both(owns):
    ops.crop_roi_align(images, prep["boxes_crop"], im, (240, 320), out=x, n_channels=3, owns_record=owns)
    ops.rasterize_into(store, x, 3, obj, prep["TCV_O"], prep["K_crop"], False, False)
for owns in (False, True, False, True):
    for what, fn in (("crop", lambda: ops.crop_roi_align(images, prep["boxes_crop"], im, (240, 320), out=x, n_channels=3, owns_record=owns)),
                     ("crop + raster", lambda: both(owns))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"owns_record={owns} {what}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
