#!/bin/bash
# squeeze-excitation squeeze + gate in one launch (se_reduce_expand_kernel) against the two launches (HP_SE_THREE_LAUNCHES=1)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06j; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -x -k "efficientnet or mbconv or squeeze" > $O/tests.log 2>&1; tail -3 $O/tests.log
python - <<'P'
import os, subprocess, sys, json
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from happypose_amd import ops
from happypose_amd.models import pose_model_param_shapes
from happypose_amd.synthetic import predictor_weights
dev = torch.device("cuda:0")
w = predictor_weights(pose_model_param_shapes("efficientnet-b3", 6), seed=3)
net = ops.Net("efficientnet-b3", 6, w, max_batch=8, device=dev)
x = net.new_input(8); x[..., :6] = torch.as_tensor(np.random.RandomState(1).rand(8, 240, 320, 6).astype(np.float32), device=dev)
outs = [net.forward(x, want_pose=True, want_features=True) for _ in range(3)]
f = outs[-1][2].cpu().numpy(); p = outs[-1][0].cpu().numpy()
assert all(torch.equal(outs[0][2], o[2]) for o in outs)
np.save(sys.argv[1], np.concatenate([f.reshape(-1), p.reshape(-1)]))
'''
import numpy as np
for name, env in (("one", {}), ("three", {"HP_SE_THREE_LAUNCHES": "1"})):
    subprocess.run([sys.executable, "-c", code, f"gpurun_out/r06j/feat_{name}.npy"], check=True, env={**os.environ, **env})
a, b = np.load("gpurun_out/r06j/feat_one.npy"), np.load("gpurun_out/r06j/feat_three.npy")
print("one launch vs two launches: bit-equal", bool(np.array_equal(a, b)), "max abs diff", float(np.abs(a - b).max()))
P
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export HP_SE_THREE_LAUNCHES=1; else unset HP_SE_THREE_LAUNCHES; fi
    timeout 400 python bench.py --arch efficientnet-b3 --steps 8 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --entry predictor > $O/bench_se_${v}_$rep.json 2> $O/bench_se_${v}_$rep.err
    python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_se_${v}_$rep.json").read().strip().splitlines()[-1])
    print("EfficientNet-b3 C2, SE as", "two launches" if $v else "one launch", "rep $rep:", round(d["value"],1), "poses/s", "launches", d["roofline"]["launches"])
except Exception as e: print("bench failed", e); print(open("$O/bench_se_${v}_$rep.err").read()[-800:])
P
  done
done
