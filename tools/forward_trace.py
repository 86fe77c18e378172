"""Kernel sequence of ONE backbone forward (eager, one stream) with durations, from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace -d <dir> -o p --output-format csv -- python3 tools/forward_trace.py run <arch> <cin> <prec> <batch>
    python3 tools/forward_trace.py show <dir>/.../p_kernel_trace.csv <launches_per_forward or 0 = autodetect>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    from happypose_amd import ops
    from happypose_amd.models import pose_model_param_shapes
    from happypose_amd.synthetic import predictor_weights

    arch, cin, prec, b = sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    dev = torch.device("cuda:0")
    w = predictor_weights(pose_model_param_shapes(arch, cin, pose_dim=9, n_views_logits=1), seed=4)
    net = ops.Net(arch, cin, w, max_batch=b, device=dev, precision=prec)
    x = net.new_input(b)
    x[..., :cin] = torch.rand((b, 240, 320, cin), device=dev).to(x.dtype)
    for _ in range(4):
        net.forward(x)
        torch.cuda.synchronize()
else:
    import csv
    import re

    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the last forward = the launches after the last but one head_kernel
    heads = [i for i, n in enumerate(names) if "head_kernel" in n]
    lo, hi = heads[-2] + 1, heads[-1] + 1
    tot = 0.0
    fam = {}
    for r in rows[lo:hi]:
        n = re.sub(r"void |hp::|\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        k = n.split("<")[0]
        fam[k] = fam.get(k, 0.0) + us
        print(f"{n:48s} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} x {r['Workgroup_Size_X']:>4s}  {us:8.1f} us")
    span = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
    print(f"launches {hi - lo}, kernel time {tot:.0f} us, span {span:.0f} us")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"  {k:32s} {v:8.1f} us  {v / tot:.3f}")
