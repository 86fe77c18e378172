"""Random-shape comparison of the conv kernel families against the exact-fp32 generic implicit GEMM (GPU box):
python tools/conv_fuzz.py [n_cases] [seed].  3x3 stride-1 cases exercise the split-fp16 patch kernel ("split"), both
Winograd schedules and the patch kernel; 3x3 stride-2 cases the space-to-depth split kernel; other filter sizes the
generic split-fp16 implicit GEMM."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    kind = ["s1", "s1", "s2", "other"][case % 4]
    h, w = int(rs.randint(2, 70)), int(rs.randint(2, 100))
    cin = 32 * int(rs.randint(1, 9))
    cout = 32 * int(rs.randint(1, 9))
    k, s, p = 3, 1, 1
    if kind == "s2":
        cin, cout, s = 64 * int(rs.randint(1, 5)), 128 * int(rs.randint(1, 4)), 2
    elif kind == "other":
        k = int(rs.choice([1, 2, 5, 7]))
        s, p = int(rs.randint(1, 3)), k // 2
        cin = 4 * int(rs.randint(1, 17))
        h, w = max(h, k), max(w, k)
    n = int(rs.choice([1, 2, 3, 5, 17, 40, 129]))
    if n * h * w * max(cin, cout) > 3e8:
        n = max(1, int(3e8 // (h * w * max(cin, cout))))
    pre, res, bias, relu = (bool(rs.randint(2)) for _ in range(4))
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, k, k, cin, device=dev) / np.sqrt(k * k * cin)
    ps = torch.rand(cin, device=dev) + 0.5 if pre else None
    pb = torch.randn(cin, device=dev) * 0.3 if pre else None
    b = torch.randn(cout, device=dev) if bias else None
    r = torch.randn(n, ho, wo, cout, device=dev) if res else None
    algos = ("igemm", "split", "auto", "winograd", "winograd-1wave", "direct") if kind == "s1" else ("igemm", "split", "auto")
    out = {}
    for algo in algos:
        ops.select_conv_algo(algo)
        out[algo] = ops.conv2d_nhwc(x, wt, s, p, b, r, ps, pb, relu)
    ops.select_conv_algo("auto")
    scale = max(1.0, float(out["igemm"].abs().max()))
    errs = {a: float((out[a] - out["igemm"]).abs().max()) / scale for a in algos[1:]}
    worst = max(worst, *errs.values())
    flag = "" if max(errs.values()) < 6e-5 else "   <-- MISMATCH"
    print(f"{case:3d} {kind:5s} n={n:3d} {h:2d}x{w:3d} {cin:3d}->{cout:3d} k{k} s{s} pre={int(pre)} res={int(res)} bias={int(bias)} "
          f"relu={int(relu)}  " + "  ".join(f"{a} {e:.1e}" for a, e in errs.items()) + flag)
print("worst relative difference:", worst)
assert worst < 6e-5

# fp16 plan: the persistent ping-pong kernel (Cin % 64 == 0, Cout % 128 == 0) and the generic fp16 kernels against torch's
# fp32 convolution of the same fp16-rounded operands; HP_PP_GRID=8 makes every workgroup walk many items
worst16 = 0.0
for case in range(max(8, n_cases // 4)):
    h, w = int(rs.randint(2, 50)), int(rs.randint(2, 90))
    cin, cout = 64 * int(rs.randint(1, 5)), 128 * int(rs.randint(1, 4))
    n = int(rs.choice([1, 3, 17, 40, 129]))
    if n * h * w * max(cin, cout) > 2e8:
        n = max(1, int(2e8 // (h * w * max(cin, cout))))
    pre, res, relu = (bool(rs.randint(2)) for _ in range(3))
    x = torch.randn(n, h, w, cin, device=dev).half()
    wt = (torch.randn(cout, 3, 3, cin, device=dev) / np.sqrt(9 * cin)).half()
    b = torch.randn(cout, device=dev)
    ps = (torch.rand(cin, device=dev) + 0.5).half() if pre else None
    pb = (torch.randn(cin, device=dev) * 0.3).half() if pre else None
    r = torch.randn(n, h, w, cout, device=dev).half() if res else None
    y = ops.conv2d_nhwc_f16(x, wt, 1, 1, b, r, ps, pb, relu).float()
    xa = x.float()
    if pre:
        xa = torch.relu((x * ps + pb).float())  # the kernel forms the prologue in fp16
    ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.float()
    if relu:
        ref = torch.relu(ref)
    e = float((y - ref).abs().max() / max(1.0, float(ref.abs().max())))
    worst16 = max(worst16, e)
    print(f"f16 {case:3d} n={n:3d} {h:2d}x{w:3d} {cin:3d}->{cout:3d} pre={int(pre)} res={int(res)} relu={int(relu)}  {e:.1e}" + ("" if e < 2e-3 else "   <-- MISMATCH"))
print("worst fp16 relative difference:", worst16)
assert worst16 < 2e-3
