"""Occupancy of the multisampled coverage walk (ablation build -DHP_RABL_COUNT, HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/RCOUNT.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HP_STAGE_REPS", "1")
from happypose_amd import _ffi  # noqa: E402

lib = _ffi.lib()
buf = (C.c_ulonglong * 8)()
import bench  # noqa: E402
from happypose_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for wl in ("C2", "C3"):
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
    torch.cuda.synchronize()
    lib.hp_debug_raster_counters(buf, 1)
    ops.rasterize_into(store, x, 3 if c2 else 4, obj, prep["TCV_O"], prep["K_crop"], not c2, not c2, None if c2 else prep["tCR"][:, 2].contiguous(),
                       0 if c2 else 2, msaa=True, aniso=True)
    torch.cuda.synchronize()
    lib.hp_debug_raster_counters(buf, 1)
    it, area, tri, surv, waves, inv, probes, two = [int(buf[i]) for i in range(8)]
    pair_it, it = it >> 32, it & 0xFFFFFFFF
    pairs_hit, two = two >> 32, two & 0xFFFFFFFF
    samples_hit, probes = probes >> 32, probes & 0xFFFFFFFF
    print(wl, dict(walk_iterations=it, sum_area=area, triangles=tri, survivors=surv, waves=waves, mean_area=area / max(tri, 1),
                   iterations_per_wave=it / max(waves, 1), lane_utilisation=area / max(64 * it, 1), survivor_fraction=surv / max(area, 1), shading_invocations=inv, probes_per_invocation=probes / max(inv, 1),
                   two_level_fraction=two / max(inv, 1), probe_pair_iterations=pair_it,
                   probe_lane_utilisation=probes / max(128 * pair_it, 1),
                   tested_pairs_with_a_covered_sample=pairs_hit, covered_samples=samples_hit))
