"""Micro-benchmark of the rasteriser and the crop kernel: NCHW tensors vs NHWC slices."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops
from happypose_amd.synthetic import make_object_dataset, make_scene

dev = torch.device("cuda:0")
ds = make_object_dataset(8, seed=1, tex_size=1024)
store = ops.MeshStore(ds, dev)
sc = make_scene()
B = 128
T = torch.as_tensor(sc["TCO_hyp"], device=dev)
K = torch.as_tensor(sc["K"], device=dev)
obj = torch.as_tensor(sc["hyp_obj_ids"], device=dev)
im_ids = torch.zeros(B, dtype=torch.int32, device=dev)
prep = ops.pose_prep(store, T, K, im_ids, obj, (480, 640))
Kc = prep["K_crop"][:, 0].contiguous()
images = torch.as_tensor(sc["images"], device=dev)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

x = torch.zeros((B, 240, 320, 8), device=dev)
cov = (ops.rasterize(store, obj, T, Kc, (240, 320), render_depth=True)[2] > 0).float().mean().item()
print(f"coverage {cov:.3f}")
for name, fn, mb in [
    ("raster NCHW rgb", lambda: ops.rasterize(store, obj, T, Kc, (240, 320)), B * 3 * 76800 * 4 / 1e6),
    ("raster NCHW rgb+n+d", lambda: ops.rasterize(store, obj, T, Kc, (240, 320), True, True), B * 7 * 76800 * 4 / 1e6),
    ("raster NHWC rgb", lambda: ops.rasterize_into(store, x, 3, obj, T[:, None], Kc[:, None], False, False), B * 3 * 76800 * 4 / 1e6),
    ("crop NCHW", lambda: ops.crop_roi_align(images, prep["boxes_crop"], im_ids), B * 3 * 76800 * 4 / 1e6),
    ("crop NHWC", lambda: ops.crop_roi_align(images, prep["boxes_crop"], im_ids, out=x), B * 3 * 76800 * 4 / 1e6),
]:
    us = timeit(fn)
    print(f"{name:22s} {us:8.1f} us   {mb / us * 1e3:7.1f} GB/s of output")
