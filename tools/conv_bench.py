"""Micro-benchmark of the conv kernel on the WideResNet-34 layer shapes (batch 128)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops, _ffi

dev = torch.device("cuda:0")
print("occupancy (blocks/CU): 128x128 ->", _ffi.lib().hp_conv_occupancy(0), " 128x64 ->", _ffi.lib().hp_conv_occupancy(1))
B = int(os.environ.get("B", 128))
shapes = [  # (h, w, cin, cout, k, stride, pad, pre)
    (60, 80, 64, 64, 3, 1, 1, False), (60, 80, 64, 64, 3, 1, 1, True),
    (60, 80, 64, 128, 3, 2, 1, True), (30, 40, 128, 128, 3, 1, 1, False), (30, 40, 128, 128, 3, 1, 1, True),
    (15, 20, 256, 256, 3, 1, 1, False), (15, 20, 256, 256, 3, 1, 1, True),
    (8, 10, 512, 512, 3, 1, 1, False), (8, 10, 512, 512, 3, 1, 1, True), (60, 80, 64, 128, 1, 2, 0, True),
]
for (h, w, cin, cout, k, s, p, pre) in shapes:
    x = torch.randn(B, h, w, cin, device=dev)
    wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
    ps = torch.rand(cin, device=dev) + 0.5 if pre else None
    pb = torch.randn(cin, device=dev) if pre else None
    for _ in range(3):
        y = ops.conv2d_nhwc(x, wt, s, p, pre_scale=ps, pre_shift=pb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        y = ops.conv2d_nhwc(x, wt, s, p, pre_scale=ps, pre_shift=pb)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * k * k * cin
    extra = ""
    try:  # diagnostics build (-DHP_PP_STAMPS): shader clock and cycles per tap inside the K loop of conv3x3_pp
        import ctypes
        fn = _ffi.lib().hp_debug_pp_stamps
        buf = (ctypes.c_double * 8)()
        if fn(buf) == 0 and buf[3] > 0:
            extra = f"  clock {buf[0] / buf[1] * 100:6.0f} MHz  {buf[0] / buf[2]:7.0f} cycles/tap  ({buf[2] / buf[3]:.0f} taps/WG)  prologue {buf[4] / buf[3]:6.0f}  epilogue {buf[5] / buf[3]:6.0f} cycles/WG"
    except AttributeError:
        pass
    print(f"{h:3d}x{w:3d} {cin:4d}->{cout:4d} k{k} s{s} pre={int(pre)}  {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s{extra}")
