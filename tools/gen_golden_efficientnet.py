"""Golden vector G9: the reference's EfficientNet-b3 backbone (CosyPose, 6 input channels) on
name-keyed random weights -- run in the build container only (imports /root/reference through the
namespace shim of tools/gen_golden.py); writes tests/golden/g9_efficientnet.npz.

    python tools/gen_golden_efficientnet.py
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import gen_golden as GG  # noqa: E402


def main():
    import torch

    from happypose_amd.synthetic import named_weights

    GG._shim()
    eff = GG.imp("happypose.pose_estimators.cosypose.cosypose.models.efficientnet")
    net = eff.EfficientNet.from_name("efficientnet-b3", in_channels=6)
    sd = net.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    w = named_weights(shapes, seed=0)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in w.items()})
    net.eval()
    x = np.random.RandomState(106).uniform(-1, 1, size=(2, 6, 240, 320)).astype(np.float32)
    acts = {}

    def hook(name):
        def f(_m, _i, o):
            acts[name] = o.detach()
        return f

    for i in (0, 1, 4, 7, 12, 17, 23, 25):
        net._blocks[i].register_forward_hook(hook(f"block{i}"))
    torch.set_num_threads(8)
    with torch.no_grad():
        y = net(torch.as_tensor(x))
    g = {"keys": np.array(list(shapes.keys())), "shapes": np.array([str(s) for s in shapes.values()]),
         "out_shape": np.array(y.shape), "out_mean": y.mean(dim=(2, 3)).numpy(), "out_sample": y.flatten()[::211].numpy()}
    for nm, a in acts.items():
        g[f"{nm}_shape"] = np.array(a.shape)
        g[f"{nm}_mean"] = np.array([a.double().mean().item(), a.double().abs().mean().item()])
        g[f"{nm}_sample"] = a.flatten()[::997].numpy()
        print(nm, tuple(a.shape), float(a.abs().mean()))
    print("out", tuple(y.shape), float(y.abs().mean()), float(y.abs().max()))
    np.savez_compressed(GG.OUT / "g9_efficientnet.npz", **g)


if __name__ == "__main__":
    main()
