"""Reproducer for the co-scheduling non-determinism: the rasteriser stage alone on one stream, rendering
the SAME views again and again, beside a co-runner on a second stream.  Every render must give the bits of the first.

  python tools/probes/raster_two_streams.py [corunner] [reps]
      corunner: none | conv (a conv-stack forward of the C2 backbone, the production neighbour) | raster (another store's
                renders) | matmul (torch.matmul: a foreign kernel family)
Env: HP_RASTER_NO_CULL=1 (no back-face culling), HP_PROBE_MSAA / HP_PROBE_ANISO (default 1 / 1), HP_PROBE_VIEWS (default 64).
Prints one JSON line: runs that differ, and for the first differing run how many pixels / which channels."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from happypose_amd import ops  # noqa: E402

corunner = sys.argv[1] if len(sys.argv) > 1 else "conv"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
msaa, aniso = bool(int(os.environ.get("HP_PROBE_MSAA", "1"))), bool(int(os.environ.get("HP_PROBE_ANISO", "1")))
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet34", seed=0, workload="C2", n_lanes=1)
store = renderer.store
nv = int(os.environ.get("HP_PROBE_VIEWS", "64"))
K = torch.as_tensor(scene["K"], device=dev)
T = torch.as_tensor(scene["TCO_hyp"], device=dev)[:nv]
obj = torch.as_tensor(scene["hyp_obj_ids"], device=dev)[:nv]
im_ids = torch.zeros(nv, dtype=torch.int32, device=dev)
prep = ops.pose_prep(store, T, K, im_ids, obj, (480, 640), multiview_type="TCO", normalize=False)
x = torch.zeros((nv, 240, 320, 8), device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

other = None
if corunner == "raster":
    from happypose_amd.synthetic import make_object_dataset
    other = ops.MeshStore(make_object_dataset(3, seed=1, tex_size=256), dev)
    xo = torch.zeros((nv, 240, 320, 8), device=dev)
    oobj = (torch.arange(nv, device=dev) % 3).to(torch.int32)
elif corunner == "conv":
    xin = model.backbone.new_input(64)
    xin.normal_()
elif corunner == "matmul":
    ma, mb = torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)


def co_step():
    if corunner == "raster":
        ops.rasterize_into(other, xo, 3, oobj, prep["TCV_O"], prep["K_crop"], False, False, None, 0, msaa=msaa, aniso=aniso)
    elif corunner == "conv":
        model.backbone.forward(xin)
    elif corunner == "matmul":
        torch.matmul(ma, mb)


full = bool(int(os.environ.get("HP_PROBE_FULL", "0")))  # rgb + normals + depth (the MegaPose render), else rgb into the 8-float record
if full:
    x = torch.zeros((nv, 240, 320, 8), device=dev)
    zn = prep["tCR"][:, 2].contiguous()


def render():
    if full:
        ops.rasterize_into(store, x, 0, obj, prep["TCV_O"], prep["K_crop"], True, True, zn, 2, msaa=msaa, aniso=aniso)
    else:
        ops.rasterize_into(store, x, 3, obj, prep["TCV_O"], prep["K_crop"], False, False, None, 0, msaa=msaa, aniso=aniso)


torch.cuda.synchronize()
with torch.cuda.stream(sa):
    render()
torch.cuda.synchronize()
ref = x.clone()
bad, first = 0, None
for r in range(reps):
    with torch.cuda.stream(sb):
        for _ in range(4):
            co_step()
    with torch.cuda.stream(sa):
        x.zero_()
        render()
        same = torch.equal(x, ref)
    if not same:
        bad += 1
        if first is None:
            d = (x != ref)
            first = dict(run=r, pixels=int(d.any(-1).sum()), views=[int(v) for v in d.any(-1).any(-1).any(-1).nonzero().flatten()[:8]],
                         channels=[int(c) for c in d.any(0).any(0).any(0).nonzero().flatten()])
torch.cuda.synchronize()
print(json.dumps(dict(corunner=corunner, reps=reps, msaa=msaa, aniso=aniso, cull=not os.environ.get("HP_RASTER_NO_CULL"), runs_that_differ=bad, first=first)))
