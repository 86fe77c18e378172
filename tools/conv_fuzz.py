"""Random-shape comparison of the Winograd kernels against the generic implicit GEMM (GPU box):
python tools/conv_fuzz.py [n_cases] [seed]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    h, w = int(rs.randint(2, 70)), int(rs.randint(2, 100))
    cin = 32 * int(rs.randint(1, 9))
    cout = 32 * int(rs.randint(1, 9))
    n = int(rs.choice([1, 2, 3, 5, 17, 40, 129]))
    if n * h * w * max(cin, cout) > 3e8:
        n = max(1, int(3e8 // (h * w * max(cin, cout))))
    pre, res, bias, relu = (bool(rs.randint(2)) for _ in range(4))
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, 3, 3, cin, device=dev) / np.sqrt(9 * cin)
    ps = torch.rand(cin, device=dev) + 0.5 if pre else None
    pb = torch.randn(cin, device=dev) * 0.3 if pre else None
    b = torch.randn(cout, device=dev) if bias else None
    r = torch.randn(n, h, w, cout, device=dev) if res else None
    out = {}
    for algo in ("igemm", "auto", "winograd-1wave"):
        ops.select_conv_algo(algo)
        out[algo] = ops.conv2d_nhwc(x, wt, 1, 1, b, r, ps, pb, relu)
    ops.select_conv_algo("auto")
    scale = max(1.0, float(out["igemm"].abs().max()))
    e8 = float((out["auto"] - out["igemm"]).abs().max()) / scale
    e1 = float((out["winograd-1wave"] - out["igemm"]).abs().max()) / scale
    worst = max(worst, e8, e1)
    flag = "" if max(e8, e1) < 6e-5 else "   <-- MISMATCH"
    print(f"{case:3d} n={n:3d} {h:2d}x{w:3d} {cin:3d}->{cout:3d} pre={int(pre)} res={int(res)} bias={int(bias)} relu={int(relu)}  "
          f"2-wave {e8:.1e}  1-wave {e1:.1e}{flag}")
print("worst relative difference:", worst)
assert worst < 6e-5
