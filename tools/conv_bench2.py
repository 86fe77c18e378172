"""A/B micro-benchmark of the 3x3 stride-1 conv kernels on the ResNet layer shapes, fp32 (split-fp16) and fp16, with and
without residual, plus a correctness check against torch's convolution.  B=<batch> F16B=<batch of the fp16 cases>."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); FB = int(os.environ.get("F16B", 576))
shapes = [(60, 80, 64, 64), (30, 40, 128, 128), (15, 20, 256, 256), (8, 10, 512, 512)]


def timeit(fn, n=10):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


for half in (False, True):
    b = FB if half else B
    dt = torch.float16 if half else torch.float32
    fn_conv = ops.conv2d_nhwc_f16 if half else ops.conv2d_nhwc
    for (h, w, cin, cout) in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(b, h, w, cin, device=dev, generator=g).to(dt)
        wt = (torch.randn(cout, 3, 3, cin, device=dev, generator=g) * 0.05).to(dt)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn(b, h, w, cout, device=dev, generator=g).to(dt)
        ps = (torch.rand(cin, device=dev, generator=g) + 0.5).to(dt); pb = torch.randn(cin, device=dev, generator=g).to(dt)
        for name, kw in (("plain", dict(bias=bias, relu=True)), ("res", dict(bias=bias, residual=res)), ("pre", dict(pre_scale=ps, pre_shift=pb, bias=bias, relu=True))):
            ms, y = timeit(lambda: fn_conv(x, wt, 1, 1, **kw))
            # reference on a slice of the batch (fp32 torch conv on the same operands)
            nb = min(b, 4)
            xa = x[:nb].float()
            if "pre_scale" in kw:
                xa = torch.relu(xa * ps.float() + pb.float())
            ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
            if "residual" in kw:
                ref = ref + res[:nb].float()
            if kw.get("relu"):
                ref = torch.relu(ref)
            err = float((y[:nb].float() - ref).abs().max() / ref.abs().max())
            fl = 2.0 * b * h * w * cout * 9 * cin
            extra = ""
            try:  # -DHP_PP_STAMPS build: cycles inside conv3x3_pp (all launches since the last read)
                import ctypes
                from happypose_amd import _ffi
                buf = (ctypes.c_double * 8)()
                if _ffi.lib().hp_debug_pp_stamps(buf) == 0 and buf[3] > 0:
                    extra = (f"  | {buf[0] / buf[1] * 100:5.0f} MHz {buf[0] / buf[2]:6.0f} cyc/tap {buf[2] / buf[3]:4.0f} taps/item; per item: prologue "
                             f"{buf[4] / buf[3]:6.0f} K loop {buf[0] / buf[3]:7.0f} epilogue {buf[5] / buf[3]:6.0f} cycles")
            except AttributeError:
                pass
            print(f"{'f16' if half else 'f32'} B={b:4d} {h:3d}x{w:3d} {cin:4d}->{cout:4d} {name:5s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s  rel err {err:.1e}{extra}")
