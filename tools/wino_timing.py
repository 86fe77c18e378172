"""Per-block time stamps of the Winograd conv kernel (diagnostic build with -DHP_WABL_TIMING, see
tools/wino_timing.sh): launch ramp, prologue, and per item the set-up / K-loop / epilogue durations."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from happypose_amd import ops, _ffi

dev = torch.device("cuda:0")
lib = _ffi.lib()
lib.hp_debug_wino_stamps.restype = C.c_int
lib.hp_debug_wino_stamps.argtypes = [C.c_void_p, C.c_int]
SLOTS = 64
B = int(os.environ.get("B", 128))
for (h, w, cin, cout) in [(60, 80, 64, 64), (30, 40, 128, 128), (15, 20, 256, 256), (8, 10, 512, 512)]:
    x = torch.randn(B, h, w, cin, device=dev)
    wt = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    for _ in range(3):
        ops.conv2d_nhwc(x, wt, 1, 1)
    torch.cuda.synchronize()
    st = np.zeros(512 * SLOTS, np.int64)
    assert lib.hp_debug_wino_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(512, SLOTS)[:256].astype(np.float64) / 100.0  # us (100 MHz wall clock)
    t0 = st[:, 0].min()
    print(f"== {h}x{w} {cin}->{cout}")
    print(f"   block start - first start: mean {np.mean(st[:, 0] - t0):.2f} max {np.max(st[:, 0] - t0):.2f}")
    print(f"   prologue (start -> first K loop): mean {np.mean(st[:, 1] - st[:, 0]):.2f} max {np.max(st[:, 1] - st[:, 0]):.2f}")
    raw = np.zeros(512 * SLOTS, np.int64)
    lib.hp_debug_wino_stamps(raw.ctypes.data, raw.size)
    raw = raw.reshape(512, SLOTS)[:256]
    mhz = (raw[:, 63] - raw[:, 62]) / ((raw[:, 4] - raw[:, 0]) / 100.0)
    print(f"   shader clock over item 0: mean {mhz.mean():.0f} MHz (min {mhz.min():.0f} max {mhz.max():.0f})")
    nch = cin // 16
    cs = st[:, 32:32 + nch]
    k_end = st[:, 6]  # K loop of item 1 done
    dur = np.diff(np.concatenate([cs, k_end[:, None]], 1), axis=1).mean(0)
    print("   chunks of item 1 (us):", " ".join(f"{v:.2f}" for v in dur))
    for i in range(3):  # the first items only (later slots may be stale: the buffer is never cleared)
        s0, s1, s2 = st[:, 2 + 3 * i], st[:, 3 + 3 * i], st[:, 4 + 3 * i]
        nxt = st[:, 5 + 3 * i]
        print(f"   item {i}: K loop {np.mean(s1 - s0):.2f} (min {np.min(s1 - s0):.2f} max {np.max(s1 - s0):.2f})  "
              f"epilogue {np.mean(s2 - s1):.2f}  set-up of the next {np.mean(nxt - s2):.2f}")
