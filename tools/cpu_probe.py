"""Where does the CPU baseline spend its time, and at which thread count is it fastest?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import backbones as ob, native
from happypose_amd.synthetic import named_weights, make_object_dataset, make_scene
from happypose_amd.mesh_store import PackedMeshes

print("cores", os.cpu_count(), "torch threads default", torch.get_num_threads())
shapes = ob.param_shapes("resnet34", 6)
w = {k: torch.as_tensor(v) for k, v in named_weights(shapes).items()}
x = torch.rand(8, 6, 240, 320)
for nt in (256, 128, 64, 32, 16):
    torch.set_num_threads(nt)
    with torch.no_grad():
        ob.wide_resnet_forward(x, w, 34)
        t = time.time(); ob.wide_resnet_forward(x, w, 34); dt = time.time() - t
    print(f"backbone b=8 threads={nt}: {dt:.3f}s  {8*11.35/dt:.0f} GFLOP/s")
ds = make_object_dataset(8, tex_size=256); pm = PackedMeshes(ds); sc = make_scene()
K = np.repeat(np.array([[[900., 0, 160], [0, 900., 120], [0, 0, 1]]], np.float32), 8, 0)
for nt in (256, 64, 16):
    os.environ["OMP_NUM_THREADS"] = str(nt)
    t = time.time(); native.rasterize(pm, sc["hyp_obj_ids"][:8], sc["TCO_hyp"][:8], K, (240, 320)); dt = time.time() - t
    print(f"raster 8 views (omp default): {dt:.3f}s")
    t = time.time(); native.crop_images(sc["images"], np.tile(np.array([[100, 80, 420, 320]], np.float32), (8, 1)), np.zeros(8, np.int32)); dt = time.time() - t
    print(f"crop 8: {dt:.3f}s")
