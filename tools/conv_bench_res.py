import os, sys, torch
sys.path.insert(0, os.getcwd())
from happypose_amd import ops
dev = torch.device("cuda:0")
B = 128
for (h, w, cin, cout, res) in [(60, 80, 64, 64, False), (60, 80, 64, 64, True), (30, 40, 128, 128, True)]:
    x = torch.randn(B, h, w, cin, device=dev); wt = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    r = torch.randn(B, h, w, cout, device=dev) if res else None
    for _ in range(3): y = ops.conv2d_nhwc(x, wt, 1, 1, residual=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.conv2d_nhwc(x, wt, 1, 1, residual=r)
    e1.record(); torch.cuda.synchronize()
    print(f"{h}x{w} {cin}->{cout} res={int(res)} {e0.elapsed_time(e1) * 100:.1f} us")
