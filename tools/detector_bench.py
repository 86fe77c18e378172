"""Latency of the Mask R-CNN detector (8f-4) on one 480x640 frame with random weights (the number of proposals /
detections -- and with it the head time -- depends on the weights; the backbone + FPN + RPN part does not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from happypose_amd.detector import MaskRCNN
from happypose_amd.synthetic import named_weights
from oracle import detector as od  # parameter shapes only (test infrastructure; this tool is not product code)

dev = torch.device("cuda:0")
NC = 22
shapes = dict(od.param_shapes()); shapes.update(od.head_param_shapes(NC))
w = named_weights(shapes, seed=3)
for k, f in (("rpn.head.bbox_pred.weight", 0.02), ("rpn.head.bbox_pred.bias", 0.5), ("roi_heads.box_predictor.bbox_pred.weight", 0.05),
             ("roi_heads.box_predictor.cls_score.weight", 0.3), ("rpn.head.cls_logits.weight", 2.0)):
    w[k] = (w[k] * f).astype(np.float32)
model = MaskRCNN(w, NC, input_size=(480, 640), max_batch=1, device=dev)
img = torch.as_tensor(np.random.RandomState(5).uniform(0, 1, size=(1, 3, 480, 640)).astype(np.float32), device=dev)
for _ in range(3):
    out = model(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = model(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
t1 = time.perf_counter()
for _ in range(n):
    maps = model.backbone.forward_nhwc(img)
torch.cuda.synchronize()
db = (time.perf_counter() - t1) / n
print(f"detector: {dt * 1e3:.1f} ms / frame ({len(out[0]['boxes'])} detections), of which backbone + FPN + RPN head {db * 1e3:.1f} ms")
