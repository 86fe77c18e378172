"""A/B of the two fp16 MegaPose-stem kernels (conv_stem7.hip): the persistent two-group kernel (default for <= 10 real input
channels) against the tile kernel (HP_STEM7_F16_OLD=1), on the whole vanilla_resnet34 coarse backbone (9 channels), at the
product size and at sizes whose pooled map is not a multiple of the 3 x 16 tile.  The switch is read once per process, so each
side runs in its own interpreter:  python tools/stem7_ab.py            -> prints the largest feature difference per size
                                   python tools/stem7_ab.py dump <out.npz>  (one side)"""
import os, subprocess, sys, tempfile
import numpy as np

SIZES = [(240, 320, 5), (104, 136, 3), (100, 132, 2), (64, 48, 2)]


def dump(path):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from happypose_amd import ops
    from happypose_amd.models import pose_model_param_shapes
    from happypose_amd.synthetic import predictor_weights
    dev = torch.device("cuda:0")
    w = predictor_weights(pose_model_param_shapes("vanilla_resnet34", 9, pose_dim=9, n_views_logits=1), seed=4)
    out = {}
    for (h, wd, n) in SIZES:
        x = np.random.RandomState(h).uniform(-1, 1, size=(n, h, wd, 9)).astype(np.float32)
        net = ops.Net("vanilla_resnet34", 9, w, max_batch=4, device=dev, h=h, w=wd, precision="f16")
        xin = net.new_input(n)
        xin[..., :9] = torch.as_tensor(x, device=dev)
        xin[..., 9:] = 7.0  # the pad channels of the 16-channel record must not matter to either kernel ... (zero weights)
        pose, logits, feats = net.forward(xin, want_pose=True, want_logits=True, want_features=True)
        out[f"f_{h}x{wd}"] = feats.cpu().numpy()
        out[f"p_{h}x{wd}"] = pose.cpu().numpy()
    np.savez(path, **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "dump":
        return dump(sys.argv[2])
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for tag, env in (("new", {}), ("old", {"HP_STEM7_F16_OLD": "1"})):
            subprocess.run([sys.executable, os.path.abspath(__file__), "dump", os.path.join(d, tag + ".npz")], check=True, env={**os.environ, **env})
            res[tag] = np.load(os.path.join(d, tag + ".npz"))
        worst = 0.0
        for k in res["new"].files:
            a, b = res["new"][k], res["old"][k]
            rel = float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))
            print(f"{k:14s} max|new - old| / max|old| = {rel:.2e}   (max|old| {np.abs(b).max():.3f})")
            worst = max(worst, rel)
        print("worst", worst)
        return 0 if worst < 2e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
