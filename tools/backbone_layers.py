"""Per-layer table (HP_PROFILE_LAYERS=1: stderr) of a pose backbone at a workload's chunk size.
python3 tools/backbone_layers.py vanilla_resnet34 9 f16 576    (C5)    |   ... vanilla_resnet34 32 f32 32  (a C3 lane)   |   resnet34 6 f32 64 (a C2 lane)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HP_PROFILE_LAYERS", "1")
import numpy as np, torch
from happypose_amd import ops
from happypose_amd.models import pose_model_param_shapes
from happypose_amd.synthetic import predictor_weights

arch, cin, prec, b = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
dev = torch.device("cuda:0")
w = predictor_weights(pose_model_param_shapes(arch, cin, pose_dim=9, n_views_logits=1), seed=4)
net = ops.Net(arch, cin, w, max_batch=b, device=dev, precision=prec)
x = net.new_input(b)
x[..., :cin] = torch.rand((b, 240, 320, cin), device=dev).to(x.dtype)
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
net.set_profiling(True)
for _ in range(5):
    net.forward(x)
torch.cuda.synchronize()
ms, n, fl, mfl = net.profile_collect()
print(f"{arch} cin {cin} {prec} batch {b}: conv launches {ms / 5:.3f} ms per forward ({n // 5} launches), {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s algorithmic")
