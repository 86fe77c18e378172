"""Detector backbone: ResNet-50 + FPN + RPN head on the HIP conv kernels against the torch-CPU
restatement of torchvision's modules (oracle/detector.py; parity unpinned: torchvision is absent)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FEAT_TOL = 2e-4  # of max|ref| per map, the bound of the other backbones (tests/test_gpu_kernels.py::test_backbone_golden_g6)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("size,batch", [((128, 160), 2), ((480, 640), 1)])
def test_detector_backbone_fpn_rpn_vs_oracle(dev, size, batch):
    from happypose_amd.detector import LEVELS, DetectorBackbone
    from happypose_amd.synthetic import named_weights
    from oracle import detector as od

    w = named_weights(od.param_shapes(), seed=3)
    net = DetectorBackbone(w, input_size=size, max_batch=batch, device=dev)
    images = np.random.RandomState(5).uniform(0, 1, size=(batch, 3, *size)).astype(np.float32)
    got = net(torch.as_tensor(images, device=dev))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = od.backbone_fpn_rpn(torch.as_tensor(images), w)
    assert net.net.status() == 0
    h, w_ = size
    for i, k in enumerate(LEVELS):
        r = ref["features"][i].numpy()
        g = got["features"][k].cpu().numpy()
        exp = (batch, 256, (h // 4) >> i, (w_ // 4) >> i) if i < 4 else (batch, 256, ((h // 32) - 1) // 2 + 1, ((w_ // 32) - 1) // 2 + 1)
        assert g.shape == r.shape == exp, (k, g.shape, r.shape, exp)
        err = np.abs(g - r).max()
        assert err <= FEAT_TOL * np.abs(r).max(), (k, err, np.abs(r).max())
    for name in ("objectness", "deltas"):
        for i in range(5):
            r, g = ref[name][i].numpy(), got[name][i].cpu().numpy()
            assert g.shape == r.shape
            err = np.abs(g - r).max()
            assert err <= FEAT_TOL * max(np.abs(r).max(), np.abs(ref["features"][i].numpy()).max()), (name, i, err)


def test_detector_raises_after_dense_stage(dev):
    """The dense stage runs; the heads are not built: get_detections must say so instead of inventing detections."""
    from types import SimpleNamespace

    from happypose_amd.detector import Detector, DetectorBackbone
    from happypose_amd.synthetic import named_weights
    from oracle import detector as od

    net = DetectorBackbone(named_weights(od.param_shapes(), seed=3), input_size=(128, 160), max_batch=1, device=dev)
    det = Detector(net, {"obj_000001": 1})
    obs = SimpleNamespace(images=torch.rand(1, 3, 128, 160, device=dev))
    with pytest.raises(NotImplementedError):
        det.get_detections(obs)
