"""A/B of a kernel path against the one it replaces (an environment switch), through the whole backbone, at the product size
and at sizes whose pooled map is not a multiple of the 3 x 16 tile.  The switches are read once per process, so each side runs
in its own interpreter.
  python tools/stem_ab.py stem7f16   conv_stem7x7s2_pool_f16_pp (vanilla_resnet34, 9 channels, fp16 plan) vs HP_STEM7_F16_OLD=1
  python tools/stem_ab.py stem5      conv_stem5x5s2_pool_split_pp (CosyPose resnet34, 6 channels, fp32)    vs HP_STEM5_OLD=1
  python tools/stem_ab.py shortcut | shortcut_mp   shortcuts inside the stride-2 launch (resnet34 / vanilla_resnet34) vs HP_NET_NO_SHORTCUT_FUSION=1
prints the largest feature / pose difference per size, last line "worst <value>"; exit code 0 when below the case's bound
(fp16: 2e-3 of the feature scale -- different summation order; stem5: 0 -- same MFMA order, monotone epilogue)."""
import os, subprocess, sys, tempfile
import numpy as np

SIZES = [(240, 320, 5), (104, 136, 3), (100, 132, 2), (64, 48, 2), (240, 320, 64)]  # the last one in ONE chunk of 64 (a lane of C2:
# K-sliced tail items and more work items than workgroups); the others in chunks of <= 4
CASES = {"stem7f16": ("vanilla_resnet34", 9, "f16", "HP_STEM7_F16_OLD", 2e-3),
         "stem5": ("resnet34", 6, "f32", "HP_STEM5_OLD", 0.0),
         # the blocks' 1x1 / stride-2 shortcuts as work items of the 3x3 / stride-2 launch vs launches of their own (another kernel:
         # another summation order, fp32 round-off)
         "shortcut": ("resnet34", 6, "f32", "HP_NET_NO_SHORTCUT_FUSION", 2e-5),
         "shortcut_mp": ("vanilla_resnet34", 27, "f32", "HP_NET_NO_SHORTCUT_FUSION", 2e-5)}


def dump(case, path):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from happypose_amd import ops
    from happypose_amd.models import pose_model_param_shapes
    from happypose_amd.synthetic import predictor_weights
    arch, cin, prec, _, _ = CASES[case]
    dev = torch.device("cuda:0")
    w = predictor_weights(pose_model_param_shapes(arch, cin, pose_dim=9, n_views_logits=1), seed=4)
    out = {}
    for (h, wd, n) in SIZES:
        x = np.random.RandomState(h).uniform(-1, 1, size=(n, h, wd, cin)).astype(np.float32)
        net = ops.Net(arch, cin, w, max_batch=64 if n > 8 else 4, device=dev, h=h, w=wd, precision=prec)
        xin = net.new_input(n)
        xin[..., :cin] = torch.as_tensor(x, device=dev).to(xin.dtype)
        if prec == "f16":
            xin[..., cin:] = 7.0  # the pad channels of the 16-channel record must not matter to either kernel (zero weights)
        pose, logits, feats = net.forward(xin, want_pose=True, want_logits=True, want_features=True)
        out[f"f_{h}x{wd}_{n}"] = feats.cpu().numpy()
        out[f"p_{h}x{wd}_{n}"] = pose.cpu().numpy()
    np.savez(path, **out)


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "stem7f16"
    if len(sys.argv) > 3 and sys.argv[2] == "dump":
        return dump(case, sys.argv[3])
    env_old, bound = CASES[case][3], CASES[case][4]
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for tag, env in (("new", {}), ("old", {env_old: "1"})):
            subprocess.run([sys.executable, os.path.abspath(__file__), case, "dump", os.path.join(d, tag + ".npz")], check=True,
                           env={**os.environ, **env})
            res[tag] = np.load(os.path.join(d, tag + ".npz"))
        worst = 0.0
        for k in res["new"].files:
            a, b = res["new"][k], res["old"][k]
            rel = float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))
            print(f"{k:14s} max|new - old| / max|old| = {rel:.2e}   (max|old| {np.abs(b).max():.3f})")
            worst = max(worst, rel)
        print("worst", worst)
        return 0 if worst <= bound else 1


if __name__ == "__main__":
    sys.exit(main())
