"""Pointwise (1x1) layers of EfficientNet-b3 at the benchmark batch, one at a time through hp_conv2d_nhwc, for kernel A/B runs:
    rocprofv3 --kernel-trace -d <dir> -o p --output-format csv -- python3 tools/pw_bench.py run
    python3 tools/pw_bench.py show <dir>/.../p_kernel_trace.csv
Shapes: (h, w, cin, cout, gated); gated = a squeeze-excitation gate on the input and no activation (projection), else swish."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(120, 160, 40, 24, 1), (60, 80, 192, 32, 1), (30, 40, 48, 288, 0), (30, 40, 288, 48, 1), (15, 20, 96, 576, 0), (15, 20, 576, 96, 1),
          (15, 20, 136, 816, 0), (15, 20, 816, 136, 1), (7, 10, 232, 1392, 0), (7, 10, 1392, 232, 1), (7, 10, 384, 2304, 0), (7, 10, 2304, 384, 1)]
REPS = 4

if sys.argv[1] == "run":
    import torch
    from happypose_amd import ops

    dev = torch.device("cuda:0")
    n = 64
    for h, w, cin, cout, gated in SHAPES:
        x = torch.randn((n, h, w, cin), device=dev)
        wt = torch.randn((cout, 1, 1, cin), device=dev) * (1.0 / cin ** 0.5)
        b = torch.randn((cout,), device=dev) * 0.1
        g = torch.rand((n, cin), device=dev) if gated else None
        for _ in range(REPS):
            ops.conv2d_nhwc(x, wt, 1, 0, b, None, g, None, 0 if gated else 2)
        torch.cuda.synchronize()
else:
    import csv

    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "conv_igemm" in r["Kernel_Name"] or "pwconv" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == REPS * len(SHAPES), len(rows)
    tot = 0.0
    for i, (h, w, cin, cout, gated) in enumerate(SHAPES):
        us = min((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * REPS + 1:(i + 1) * REPS])
        mb = 64 * h * w * (cin + cout) * 4 / 1e6
        tot += us
        print(f"{h:3d}x{w:3d} {cin:4d}->{cout:4d} {'gate' if gated else 'swish'}  {us:7.1f} us  {mb:6.1f} MB  {mb / us:5.2f} TB/s  {rows[i * REPS]['Kernel_Name'][30:90]}")
    print(f"sum {tot:.1f} us")
