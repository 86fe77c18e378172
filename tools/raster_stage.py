"""Rasteriser / crop stage timing on the C2 inputs (bench.stage_rates) -- used with HAPPYPOSE_AMD_LIB pointing at ablation builds."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
ds, renderer, scene, weights, model = bench.build_world(dev, "resnet18", workload="C2")
images = torch.as_tensor(scene["images"], device=dev); K = torch.as_tensor(scene["K"], device=dev)
T = torch.as_tensor(scene["TCO_hyp"], device=dev); im = torch.zeros(len(T), dtype=torch.int32, device=dev)
print(json.dumps(bench.stage_rates(renderer.store, scene, images, K, T, im, dev)))
