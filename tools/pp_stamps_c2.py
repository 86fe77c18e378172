"""In-kernel stamps of conv3x3_pp inside the REAL two-lane C2 step (library built with -DHP_PP_STAMPS:
HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/STAMPS.so python3 tools/pp_stamps_c2.py [lanes])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from happypose_amd import _ffi

dev = torch.device("cuda:0")
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=lanes)
store = renderer.store
images, K = torch.as_tensor(scene["images"], device=dev), torch.as_tensor(scene["K"], device=dev)
T = torch.as_tensor(scene["TCO_hyp"], device=dev)
labels = [store.labels[i] for i in scene["hyp_obj_ids"]]
im = torch.zeros(len(labels), dtype=torch.int32, device=dev)
for _ in range(3):
    model.forward(images, K, labels, T, n_iterations=5, im_ids=im)
torch.cuda.synchronize()
buf = (ctypes.c_double * 8)()
_ffi.lib().hp_debug_pp_stamps(buf)  # reset
for _ in range(5):
    model.forward(images, K, labels, T, n_iterations=5, im_ids=im)
torch.cuda.synchronize()
assert _ffi.lib().hp_debug_pp_stamps(buf) == 0 and buf[3] > 0
print(f"lanes {lanes}: items {buf[3]:.0f}, {buf[0] / buf[1] * 100:.0f} MHz in the K loop, {buf[0] / buf[2]:.0f} cycles/tap, {buf[2] / buf[3]:.1f} taps/item; "
      f"per item: prologue {buf[4] / buf[3]:.0f}, K loop {buf[0] / buf[3]:.0f}, epilogue {buf[5] / buf[3]:.0f} cycles; set-up per workgroup {buf[6]:.0f} total")
