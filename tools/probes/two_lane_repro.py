"""Where does a two-lane MegaPose step stop being bit-reproducible?  The configuration of
tests/test_gpu_pipeline.py::test_graph_replay_matches_eager[megapose-2] (4 views, normals + depth, 2 lanes, 48 hypotheses), the
same forward() over and over; every field of every iteration is compared with the first run, and the EARLIEST differing
(iteration, field) of each differing run is tallied.  keep_pixels=True also compares the network input (crop + renders).

  python tools/probes/two_lane_repro.py [runs] [lanes] [graphs]
Env: HP_RASTER_NO_CULL=1, HP_PROBE_ITERS (default 3), HP_PROBE_PIXELS=0 (the product path: renders not materialised, graphs allowed)."""
import collections
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from happypose_amd.models import create_model_pose  # noqa: E402
from happypose_amd.renderer import BatchRenderer  # noqa: E402
from happypose_amd.synthetic import make_object_dataset, make_scene  # noqa: E402
from happypose_amd.synthetic import predictor_weights  # noqa: E402
from oracle import backbones as ob  # noqa: E402  (shapes of the reference modules' parameters: test infrastructure, like the tests)

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
graphs = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
iters = int(os.environ.get("HP_PROBE_ITERS", "3"))
dev = torch.device("cuda:0")
ds = make_object_dataset(3, seed=1, tex_size=256)
renderer = BatchRenderer(ds, device=dev)
w = predictor_weights(ob.predictor_param_shapes("vanilla_resnet34", 32, pose_dim=9, n_views_logits=0), seed=4, update_scale=0.002)
cfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True,
           render_depth=True, input_depth=True, predict_pose_update=True, depth_augmentation=False,
           depth_normalization_type="tCR_scale_clamp_center")
model = create_model_pose(cfg, renderer, state_dict=w, max_batch=48, n_lanes=lanes, graphs=graphs)
pixels = bool(int(os.environ.get("HP_PROBE_PIXELS", "1")))  # 0: the product path (no materialised renders; graphs allowed)
for l in (model.lanes if lanes > 1 else [model]):
    l.keep_pixels = pixels
sc = make_scene(n_detections=6, n_hypotheses=8, n_objects=len(renderer.store.labels), seed=9, with_depth=True)
labels = [renderer.store.labels[j] for j in sc["hyp_obj_ids"]]
args = (torch.as_tensor(sc["images"][:, :4].copy(), device=dev), torch.as_tensor(sc["K"], device=dev), labels, torch.as_tensor(sc["TCO_hyp"], device=dev))
im_ids = torch.zeros(len(labels), dtype=torch.int32)
FIELDS = ("TCO_input", "tCR", "TCV_O_input", "boxes_rend", "boxes_crop", "K_crop", "KV_crop", "images_crop", "renders", "pose", "TCO_output")


def snapshot():
    out = model.forward(*args, n_iterations=iters, im_ids=im_ids)
    snap = {}
    for n in range(1, iters + 1):
        o = out[f"iteration={n}"]
        for f in FIELDS:
            t = o.network_outputs["pose"] if f == "pose" else getattr(o, f, None)
            if t is not None:
                snap[(n, f)] = t.clone()
    torch.cuda.synchronize()
    # the network input of the LAST iteration (crop + renders, NHWC records) as every lane's buffer holds it
    for li, l in enumerate(model.lanes if lanes > 1 else [model]):
        if l._x is not None:
            snap[(iters, f"x_lane{li}")] = l._x.clone()
    return snap


ref = snapshot()
tally = collections.Counter()
detail = None
for r in range(runs):
    cur = snapshot()
    order = list(FIELDS) + ["x_lane0", "x_lane1", "x_lane2"]
    for key in sorted(ref, key=lambda k: (k[0], order.index(k[1]) if not k[1].startswith("x_lane") else 8.5)):
        if not torch.equal(ref[key], cur[key]):
            tally[f"iteration={key[0]}:{key[1]}"] += 1
            if detail is None:
                d = (ref[key] != cur[key])
                detail = dict(run=r, key=list(key), n_diff=int(d.sum()), rows=[int(v) for v in d.reshape(d.shape[0], -1).any(1).nonzero().flatten()[:10]],
                              max_abs=float((ref[key].float() - cur[key].float()).abs().max()),
                              nan_in_cur=int(torch.isnan(cur[key].float()).sum()), nan_in_ref=int(torch.isnan(ref[key].float()).sum()))
                if key[1] in ("renders", "images_crop"):
                    detail["channels"] = [int(c) for c in d.any(0).reshape(d.shape[1], -1).any(1).nonzero().flatten()]
                    where = d.nonzero()[:2000]
                    detail["samples"] = [dict(at=[int(v) for v in w], ref=round(float(ref[key][tuple(w)]) * 255, 2), cur=round(float(cur[key][tuple(w)]) * 255, 2))
                                         for w in where[:: max(1, len(where) // 24)][:24]]
                    per_row = d.reshape(d.shape[0], -1).sum(1)
                    detail["per_row"] = {int(r): int(per_row[r]) for r in per_row.nonzero().flatten()}
                if key[1].startswith("x_lane"):  # [rows, h, w, record]
                    detail["channels"] = [int(c) for c in d.reshape(-1, d.shape[-1]).any(0).nonzero().flatten()]
                    detail["pixels"] = int(d.any(-1).sum())
                    pix = d.any(-1).nonzero()[:12].tolist()
                    detail["where"] = pix
                    detail["ref_vals"] = [[float(v) for v in ref[key][a, b, c]] for a, b, c in pix[:4]]
                    detail["cur_vals"] = [[float(v) for v in cur[key][a, b, c]] for a, b, c in pix[:4]]
            break
nan_anywhere = int(sum(int(torch.isnan(v.float()).sum()) for v in ref.values()))
print(json.dumps(dict(runs=runs, lanes=lanes, graphs=graphs, nan_in_reference_snapshot=nan_anywhere, canary=bool(os.environ.get("HP_RASTER_CANARY")), cull=not os.environ.get("HP_RASTER_NO_CULL"), fields_compared=sorted({k[1] for k in ref}),
                      first_difference_tally=dict(tally), first=detail)))
