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


def _weights(num_classes=4, seed=3):
    """Name-keyed random weights of the whole DetectorMaskRCNN; box deltas damped and class logits spread so that the
    random network yields ~1000 proposals and a dozen detections of several classes."""
    from happypose_amd.synthetic import named_weights
    from oracle import detector as od

    shapes = dict(od.param_shapes())
    shapes.update(od.head_param_shapes(num_classes))
    w = named_weights(shapes, seed=seed)
    for k, f in (("rpn.head.bbox_pred.weight", 0.02), ("rpn.head.bbox_pred.bias", 0.5), ("roi_heads.box_predictor.bbox_pred.weight", 0.05),
                 ("roi_heads.box_predictor.cls_score.weight", 0.3), ("rpn.head.cls_logits.weight", 2.0)):
        w[k] = (w[k] * f).astype(np.float32)
    return w


def _iou(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa, ab = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def test_detector_stages_vs_oracle(dev):
    """Every non-convolution stage on the ORACLE's inputs (so that a discrete decision flipping upstream cannot mask a
    stage): anchor decoding, NMS, multi-level RoIAlign, the RoI heads, class / box post-processing, mask pasting."""
    import ctypes as C

    from happypose_amd._ffi import check, lib, ptr, stream_ptr
    from happypose_amd.detector import MaskRCNN, base_anchors
    from oracle import detector as od

    NC, size = 4, (256, 320)
    w = _weights(NC)
    model = MaskRCNN(w, NC, input_size=size, max_batch=1, device=dev)
    images = torch.as_tensor(np.random.RandomState(5).uniform(0, 1, size=(1, 3, *size)).astype(np.float32))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref, inter = od.maskrcnn_forward(images, w)
    dense = inter["dense"]
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in w.items()}
    for l in range(5):  # base anchors, then decode + clip + sigmoid + small-box flag of a level's top-k
        np.testing.assert_array_equal(base_anchors((32, 64, 128, 256, 512)[l]), od.base_anchors((32, 64, 128, 256, 512)[l]))
        o, d = dense["objectness"][l][0], dense["deltas"][l][0]
        gh, gw = o.shape[1:]
        ob = o.permute(1, 2, 0).reshape(-1)
        k = min(1000, ob.numel())
        top, idx = ob.topk(k)
        anchors = od.level_anchors(size, (gh, gw), (32, 64, 128, 256, 512)[l])
        want = od.clip_boxes(od.decode(d.view(3, 4, gh, gw).permute(2, 3, 0, 1).reshape(-1, 4)[idx], anchors[idx], (1.0, 1.0, 1.0, 1.0)).view(-1, 4), size)
        dmap = d.view(3, 4, gh, gw).permute(2, 3, 0, 1).reshape(gh, gw, 12).contiguous().to(dev)
        boxes = torch.empty((k, 4), device=dev); scores = torch.empty(k, device=dev); valid = torch.empty(k, dtype=torch.uint8, device=dev)
        base = (C.c_float * 12)(*base_anchors((32, 64, 128, 256, 512)[l]).reshape(-1).tolist())
        top_d, idx_d = top.to(dev).contiguous(), idx.to(dev, torch.int32).contiguous()  # keep alive: ptr() does not hold a reference
        check(lib().hp_rpn_decode(ptr(top_d), ptr(idx_d), k, ptr(dmap), gw, 3, base, size[0] // gh,
                                  size[1] // gw, C.c_float(size[0]), C.c_float(size[1]), C.c_float(1e-3), ptr(boxes), ptr(scores), ptr(valid),
                                  stream_ptr(dev)), "hp_rpn_decode")
        np.testing.assert_allclose(boxes.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(scores.cpu().numpy(), torch.sigmoid(top).numpy(), rtol=1e-5, atol=1e-6)
        wv = ((want[:, 2] - want[:, 0] >= 1e-3) & (want[:, 3] - want[:, 1] >= 1e-3)).numpy()
        assert (valid.cpu().numpy().astype(bool) == wv).mean() > 0.999
    # NMS on the oracle's candidate set (sorted by score), per level
    rs = np.random.RandomState(1)
    n = 1500
    ctr = rs.uniform(20, 300, size=(n, 2)); wh = rs.uniform(5, 60, size=(n, 2))
    bx = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    sc = np.sort(rs.uniform(size=n).astype(np.float32))[::-1].copy()
    grp = rs.randint(0, 3, n)
    keep = model._nms(torch.as_tensor(bx, device=dev), torch.as_tensor(grp, device=dev), 0.5).cpu().numpy()
    want_keep = od.batched_nms(torch.as_tensor(bx), torch.as_tensor(sc), torch.as_tensor(grp), 0.5).numpy()
    assert sorted(np.where(keep)[0].tolist()) == sorted(want_keep.tolist())
    # MultiScaleRoIAlign 7x7 on the oracle's features and proposals
    props = inter["proposals"][0][0]
    maps = [f.permute(0, 2, 3, 1).contiguous().to(dev) for f in dense["features"]]
    rois = torch.cat([torch.zeros((len(props), 1)), props], 1).to(dev)
    got = model._roi_align(maps, rois, 7).permute(0, 3, 1, 2).cpu().numpy()
    want, lv = od.multiscale_roi_align(dense["features"], [props], size, 7)
    # a box whose level index sits on the floor() boundary may round either way: compare the boxes that agree robustly
    s_ = torch.sqrt((props[:, 2] - props[:, 0]) * (props[:, 3] - props[:, 1]))
    frac = (4 + torch.log2(s_ / 224)).numpy() % 1.0
    safe = (frac > 1e-3) & (frac < 1 - 1e-3)
    assert safe.mean() > 0.99
    np.testing.assert_allclose(got[safe], want.numpy()[safe], rtol=1e-4, atol=1e-4 * np.abs(want.numpy()).max())
    # RoI heads on the oracle's pooled features
    pooled = inter["pooled"]
    cls, reg = model.box_net.run(pooled.permute(0, 2, 3, 1).contiguous().to(dev))
    cls, reg = cls.reshape(len(pooled), -1)[:, :NC].cpu().numpy(), reg.reshape(len(pooled), -1)[:, :4 * NC].cpu().numpy()
    for g, r in ((cls, inter["class_logits"].numpy()), (reg, inter["box_regression"].numpy())):
        assert np.abs(g - r).max() <= FEAT_TOL * max(np.abs(r).max(), 1.0), np.abs(g - r).max()
    # class / box post-processing
    n = len(props)
    sc_g = torch.empty((n, NC), device=dev); bx_g = torch.empty((n, NC, 4), device=dev)
    cl_d, rg_d, pr_d = inter["class_logits"].to(dev).contiguous(), inter["box_regression"].to(dev).contiguous(), props.to(dev).contiguous()
    check(lib().hp_box_postprocess(ptr(cl_d), NC, ptr(rg_d), 4 * NC,
                                   ptr(pr_d), n, NC, C.c_float(size[0]), C.c_float(size[1]), ptr(sc_g), ptr(bx_g),
                                   stream_ptr(dev)), "hp_box_postprocess")
    want_b = od.clip_boxes(od.decode(inter["box_regression"], props, (10.0, 10.0, 5.0, 5.0)).reshape(n, -1, 4), size)
    np.testing.assert_allclose(sc_g.cpu().numpy(), torch.softmax(inter["class_logits"], -1).numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bx_g.cpu().numpy(), want_b.numpy(), rtol=1e-5, atol=2e-3)
    # mask head + pasting on the oracle's detections
    det = ref[0]
    nd = len(det["boxes"])
    assert nd >= 5
    mp, _ = od.multiscale_roi_align(dense["features"], [det["boxes"]], size, 14)
    with torch.no_grad():
        ml_ref = od.mask_head(mp, sd)
    ml = model.mask_net.run(mp.permute(0, 2, 3, 1).contiguous().to(dev))[0]           # [nd,14,56,c4]
    ml_nchw = ml.reshape(nd, 14, 14, 2, 2, -1).permute(0, 5, 1, 3, 2, 4).reshape(nd, -1, 28, 28)[:, :NC].cpu().numpy()
    assert np.abs(ml_nchw - ml_ref.numpy()).max() <= FEAT_TOL * np.abs(ml_ref.numpy()).max()
    masks = torch.zeros((nd, 1, *size), device=dev)
    ml_d, lb_d, bx_d = ml.contiguous(), det["labels"].to(dev, torch.int32).contiguous(), det["boxes"].to(dev).contiguous()
    check(lib().hp_paste_masks(ptr(ml_d), ml.shape[-1], ptr(lb_d), ptr(bx_d),
                               nd, size[0], size[1], ptr(masks), stream_ptr(dev)), "hp_paste_masks")
    dm = np.abs(masks.cpu().numpy() - det["masks"].numpy())
    assert dm.max() < 2e-3, dm.max()


def test_maskrcnn_end_to_end_vs_oracle(dev):
    """DetectorMaskRCNN.forward end to end (and Detector.get_detections on top of it) against the oracle's: the same number
    of detections up to threshold / NMS ties, every oracle detection matched by one of the same label with IoU > 0.98 and the
    score within 2e-3, masks equal on > 99.9 % of the pixels at the reference's mask threshold."""
    from types import SimpleNamespace

    from happypose_amd.detector import Detector, MaskRCNN
    from oracle import detector as od

    NC, size = 4, (256, 320)
    w = _weights(NC)
    model = MaskRCNN(w, NC, input_size=size, max_batch=1, device=dev)
    images = torch.as_tensor(np.random.RandomState(5).uniform(0, 1, size=(1, 3, *size)).astype(np.float32))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref, inter = od.maskrcnn_forward(images, w)
    out, mine = model.forward(images.to(dev), return_intermediates=True)
    assert abs(len(mine[0]["proposals"]) - len(inter["proposals"][0][0])) <= 5
    r, g = ref[0], out[0]
    assert abs(len(g["boxes"]) - len(r["boxes"])) <= 2 and len(r["boxes"]) >= 5
    iou = _iou(r["boxes"].numpy(), g["boxes"].cpu().numpy())
    matched = 0
    for i in range(len(r["boxes"])):
        j = int(iou[i].argmax())
        if iou[i, j] > 0.98 and int(g["labels"][j]) == int(r["labels"][i]) and abs(float(g["scores"][j]) - float(r["scores"][i])) < 2e-3:
            matched += 1
            a, b_ = g["masks"][j, 0].cpu().numpy() > 0.8, r["masks"][i, 0].numpy() > 0.8
            assert (a != b_).mean() < 1e-3
    assert matched >= len(r["boxes"]) - 1, (matched, len(r["boxes"]))
    det = Detector(model, {f"obj_{c:06d}": c for c in range(1, NC)})
    obs = SimpleNamespace(images=images.to(dev))
    d = det.get_detections(obs, output_masks=True)
    assert len(d) == len(g["boxes"]) and set(d.infos.columns) >= {"batch_im_id", "label", "score", "instance_id"}
    assert d.bboxes.shape == (len(d), 4) and d.masks.shape == (len(d), *size) and d.masks.dtype == torch.bool
    d1 = det.get_detections(obs, detection_th=0.5, one_instance_per_class=True)
    assert len(d1) <= NC - 1 and (d1.infos.score > 0.5).all()


@pytest.mark.parametrize("size,min_size,max_size", [((300, 500), 240, 320), ((360, 400), 200, 400)])
def test_maskrcnn_transform_resize_vs_oracle(dev, size, min_size, max_size):
    """GeneralizedRCNNTransform.resize + batch_images in front and resize_boxes / paste at the original size behind:
    images that do not arrive at ``input_resize`` (300 x 500 -> 192 x 320, no padding; 360 x 400 -> 200 x 222 on a
    224 x 224 canvas).  The resized, normalised canvas is checked first (hp_detector_preprocess_resize vs F.interpolate),
    then the detections in the coordinates of the image as handed over."""
    from happypose_amd.detector import MaskRCNN, transform_sizes
    from oracle import detector as od

    NC = 4
    w = _weights(NC)
    model = MaskRCNN(w, NC, input_size=size, max_batch=1, device=dev, min_size=min_size, max_size=max_size)
    images = torch.as_tensor(np.random.RandomState(7).uniform(0, 1, size=(1, 3, *size)).astype(np.float32))
    torch.set_num_threads(8)
    with torch.no_grad():
        canvas, resized = od.transform(images, min_size, max_size)
        ref, inter = od.maskrcnn_forward(images, w, min_size=min_size, max_size=max_size)
    assert (tuple(resized), tuple(canvas.shape[-2:])) == transform_sizes(size, min_size, max_size) == (model.size, model.padded)
    # the pre-processing kernel alone: run the backbone's first step by hand
    import ctypes as C

    from happypose_amd._ffi import check, lib, ptr, stream_ptr
    from happypose_amd.detector import IMAGE_MEAN, IMAGE_STD

    x = torch.empty((1, *model.padded, 4), dtype=torch.float32, device=dev)
    img_d = images.to(dev).contiguous()
    check(lib().hp_detector_preprocess_resize(ptr(img_d), 1, size[0], size[1], resized[0], resized[1], model.padded[0], model.padded[1],
                                              (C.c_float * 3)(*IMAGE_MEAN), (C.c_float * 3)(*IMAGE_STD), ptr(x), stream_ptr(dev)),
          "hp_detector_preprocess_resize")
    np.testing.assert_allclose(x[..., :3].permute(0, 3, 1, 2).cpu().numpy(), canvas.numpy(), rtol=0, atol=2e-5)
    assert float(x[..., 3].abs().max()) == 0.0

    out, mine = model.forward(images.to(dev), return_intermediates=True)
    assert abs(len(mine[0]["proposals"]) - len(inter["proposals"][0][0])) <= 5
    r, g = ref[0], out[0]
    assert abs(len(g["boxes"]) - len(r["boxes"])) <= 2 and len(r["boxes"]) >= 3
    assert g["masks"].shape[-2:] == tuple(size) == tuple(r["masks"].shape[-2:])
    iou = _iou(r["boxes"].numpy(), g["boxes"].cpu().numpy())
    matched = 0
    for i in range(len(r["boxes"])):
        j = int(iou[i].argmax())
        if iou[i, j] > 0.98 and int(g["labels"][j]) == int(r["labels"][i]) and abs(float(g["scores"][j]) - float(r["scores"][i])) < 2e-3:
            matched += 1
            a, b_ = g["masks"][j, 0].cpu().numpy() > 0.8, r["masks"][i, 0].numpy() > 0.8
            assert (a != b_).mean() < 2e-3
    assert matched >= len(r["boxes"]) - 1, (matched, len(r["boxes"]))
    assert float(g["boxes"][:, 2].max()) <= size[1] + 1e-3 and float(g["boxes"][:, 3].max()) <= size[0] + 1e-3


def test_run_inference_pipeline_with_detector(dev):
    """run_detector=True (MP/inference/pose_estimator.py:559-566): detections come from the Mask-RCNN detector and feed the
    coarse / refiner stages; labels are mapped through label_to_category_id like the reference's Detector."""
    from happypose_amd.detector import Detector, MaskRCNN
    from happypose_amd.models import create_model_pose
    from happypose_amd.pose_estimator import ObservationTensor, PoseEstimator
    from happypose_amd.renderer import BatchRenderer
    from happypose_amd.synthetic import make_object_dataset, make_scene, predictor_weights
    from oracle import backbones as ob

    ds = make_object_dataset(3, seed=1, tex_size=128)
    renderer = BatchRenderer(ds, device=dev)
    labels = renderer.store.labels
    sc = make_scene(n_detections=3, n_hypotheses=1, n_objects=3, seed=2)
    det = Detector(MaskRCNN(_weights(4), 4, input_size=(480, 640), max_batch=1, device=dev, box_score_thresh=0.0), {l: i + 1 for i, l in enumerate(labels)})
    ccfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=1, multiview_type="TCO", render_normals=True,
                predict_rendered_views_logits=True, predict_pose_update=False, depth_augmentation=False)
    rcfg = dict(backbone_str="vanilla_resnet34", n_rendered_views=4, multiview_type="front_3views", render_normals=True, depth_augmentation=False)
    wc = predictor_weights(ob.predictor_param_shapes("vanilla_resnet34", 9, pose_dim=0, n_views_logits=1), seed=3, update_scale=0.05)
    wr = predictor_weights(ob.predictor_param_shapes("vanilla_resnet34", 27), seed=2)
    coarse = create_model_pose(ccfg, renderer, state_dict=wc, max_batch=72)
    refiner = create_model_pose(rcfg, renderer, state_dict=wr, max_batch=8)
    est = PoseEstimator(refiner_model=refiner, coarse_model=coarse, detector_model=det, bsz_objects=8, bsz_images=72, SO3_grid_size=72)
    obs = ObservationTensor(torch.as_tensor(sc["images"][:, :3].copy(), device=dev), torch.as_tensor(sc["K"], device=dev))
    dets = est.forward_detection_model(obs, one_instance_per_class=True)
    assert 1 <= len(dets) <= 3 and set(dets.infos.label) <= set(labels) and "instance_id" in dets.infos
    est.detector_model = type("D", (), {"get_detections": staticmethod(lambda o, *a, **k: dets)})()
    final, extra = est.run_inference_pipeline(obs, run_detector=True, n_refiner_iterations=1, n_pose_hypotheses=1)
    assert len(final) == len(dets) and torch.isfinite(final.poses).all() and "detection=" in extra["timing_str"]
