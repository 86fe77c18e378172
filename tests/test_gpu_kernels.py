"""Parity of every HIP kernel (through the C ABI) against the CPU oracle and the golden
vectors of the reference.  Needs a real MI355X: ``pytest -m gpu``.

Stated tolerances
  * geometry (pose prep / update / init): fp32 formulas in a different association order
    -> rtol 2e-5, atol 2e-3 px on pixel-valued quantities, 2e-6 on poses;
  * rasteriser: the kernel follows the oracle's operation order, so coverage agrees pixel
    for pixel except where a pixel centre lies within fp32 round-off of an edge; we allow
    <= 0.05 % of pixels to differ in coverage and require colour (8-bit quantised) within
    1/255 and depth within 1e-6 m elsewhere;
  * roi_align: rtol 1e-5 / atol 1e-6 (same formula, same order);
  * convolution / network: exact-fp32 MFMA with a different K order than oneDNN
    -> |err| <= 2e-4 * max|ref| per tensor.
"""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene_store(dev):
    from happypose_amd.ops import MeshStore
    from happypose_amd.synthetic import make_object_dataset

    ds = make_object_dataset(3, seed=1, tex_size=256)
    return MeshStore(ds, dev)


def _poses(n, seed, zlo=0.35, zhi=0.8):
    from happypose_amd.synthetic import random_rotations

    rs = np.random.RandomState(seed)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    T[:, :3, :3] = random_rotations(rs, n)
    T[:, :3, 3] = np.stack([rs.uniform(-0.03, 0.03, n), rs.uniform(-0.03, 0.03, n), rs.uniform(zlo, zhi, n)], -1)
    return T


# ----------------------------------------------------------------------------- rasteriser
def _compare_renders(gpu, ref, max_cov_mismatch=5e-4):
    rgb, nrm, dep, msk = [None if t is None else t.cpu().numpy() for t in gpu]
    cov_g = rgb.sum(1) > 0 if dep is None else dep[:, 0] > 0
    cov_r = ref["rgbs"].sum(1) > 0 if ref["depths"] is None else ref["depths"][:, 0] > 0
    mism = cov_g != cov_r
    assert mism.mean() <= max_cov_mismatch, f"coverage mismatch {mism.mean():.2e}"
    both = (~mism)[:, None]
    np.testing.assert_allclose(np.where(both, rgb, 0), np.where(both, ref["rgbs"], 0), atol=1.01 / 255)
    # colours are exactly equal on the overwhelming majority of pixels
    assert (np.abs(rgb - ref["rgbs"]).max(1)[~mism] > 1e-6).mean() < 2e-3
    if nrm is not None:
        np.testing.assert_allclose(np.where(both, nrm, 0), np.where(both, ref["normals"], 0), atol=1.01 / 255)
    if dep is not None:
        d_ok = np.abs(dep - ref["depths"])[:, 0][~mism]
        assert (d_ok > 1e-6).mean() < 1e-3 and d_ok.max() < 5e-3
    if msk is not None:
        assert (msk.cpu().numpy() if hasattr(msk, "cpu") else msk)[:, 0][~mism].tolist() == ref["binary_masks"][:, 0][~mism].tolist()


def test_rasterizer_vs_oracle(dev, scene_store):
    from happypose_amd import ops
    from oracle import native

    n = 12
    T = _poses(n, 5)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    K[:, 0, 2] += np.linspace(-20, 20, n)
    obj = (np.arange(n) % 3).astype(np.int32)
    gpu = ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320),
                        render_normals=True, render_depth=True, render_binary_mask=True)
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, True)
    assert gpu[0].shape == (n, 3, 240, 320) and gpu[0].dtype == torch.float32
    assert gpu[3].dtype == torch.bool and gpu[2].shape == (n, 1, 240, 320)
    assert (ref["depths"] > 0).mean() > 0.1  # the scene is not trivially empty
    _compare_renders(gpu, ref)


def test_rasterizer_msaa4_vs_oracle(dev, scene_store):
    """HP_RASTER_MSAA4 (the reference's framebuffer state): per-sample coverage / depth, one shading per pixel and
    triangle at the pixel centre, 8-bit resolve.  Against the oracle's definition, and against the single-sample render:
    depth and mask are unchanged (centre-sampled), colours only change where samples of a pixel see different triangles."""
    from happypose_amd import ops
    from oracle import native

    n = 12
    T = _poses(n, 5)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    K[:, 0, 2] += np.linspace(-20, 20, n)
    obj = (np.arange(n) % 3).astype(np.int32)
    args = (scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320))
    gpu = ops.rasterize(*args, render_normals=True, render_depth=True, render_binary_mask=True, msaa=True)
    one = ops.rasterize(*args, render_normals=True, render_depth=True, render_binary_mask=True)
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, True, msaa=True)
    assert torch.equal(gpu[2], one[2]) and torch.equal(gpu[3], one[3])
    rgb, nrm = gpu[0].cpu().numpy(), gpu[1].cpu().numpy()
    # a sample whose edge function is within fp32 round-off of zero may fall on the other side: the pixel then differs by
    # one sample's share (<= 1/4 of the colour); everything else matches to the 8-bit step
    for got, want in ((rgb, ref["rgbs"]), (nrm, ref["normals"])):
        d = np.abs(got - want).max(1)
        assert (d > 1.5 / 255).mean() < 5e-4, (d > 1.5 / 255).mean()
        assert d.max() <= 0.5 + 1e-6
    changed = np.abs(rgb - one[0].cpu().numpy()).max(1) > 0
    cov = ref["depths"][:, 0] > 0
    assert 0.02 < changed[cov].mean() < 0.6                       # edges and multi-triangle pixels only
    edge = (~cov) & changed                                        # uncovered centres that still collect samples
    assert edge.sum() > 100 and np.allclose(rgb * 255, np.round(rgb * 255), atol=1e-3)
    # NHWC slices of a network input
    x = torch.zeros((n, 240, 320, 8), device=dev)
    Tv, Kv = torch.as_tensor(T, device=dev)[:, None].contiguous(), torch.as_tensor(K, device=dev)[:, None].contiguous()
    ops.rasterize_into(scene_store, x, 3, torch.as_tensor(obj), Tv, Kv, False, False, msaa=True)
    assert torch.equal(x[..., 3:6].permute(0, 3, 1, 2), gpu[0])
    # depth / mask only (no colour buffer: nothing is shaded) and normals without colours
    dm = ops.rasterize(*args, render_depth=True, render_binary_mask=True, msaa=True, render_rgb=False)
    assert dm[0] is None and torch.equal(dm[2], gpu[2]) and torch.equal(dm[3], gpu[3])
    nm = ops.rasterize(*args, render_normals=True, msaa=True, render_rgb=False)
    assert torch.equal(nm[1], gpu[1])


@pytest.mark.parametrize("msaa,tex_size", [(False, 256), (True, 256), (True, 208)])
def test_rasterizer_texture_filter_vs_oracle(dev, scene_store, msaa, tex_size):
    """HP_RASTER_TEX_ANISO (the reference's texture state: mip-mapped trilinear + anisotropic 16) against the oracle's
    definition, alone and together with multisampling; geometry outputs do not depend on it.  Power-of-two textures take
    the mask / shift / probe-table path of the kernel, a 208 x 208 texture the generic one (integer modulo wraps)."""
    from happypose_amd import ops
    from oracle import native

    if tex_size != 256:
        from happypose_amd.synthetic import make_object_dataset

        scene_store = ops.MeshStore(make_object_dataset(3, seed=1, tex_size=tex_size), dev)
    n = 9
    T = _poses(n, 7, zlo=0.3, zhi=1.6)  # near and far: magnified and strongly minified textures
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = (np.arange(n) % 3).astype(np.int32)
    args = (scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320))
    gpu = ops.rasterize(*args, render_normals=True, render_depth=True, msaa=msaa, aniso=True)
    plain = ops.rasterize(*args, render_normals=True, render_depth=True, msaa=msaa)
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, False, msaa=msaa, aniso=True)
    assert torch.equal(gpu[1], plain[1]) and torch.equal(gpu[2], plain[2])            # normals, depth: untouched
    rgb = gpu[0].cpu().numpy()
    d = np.abs(rgb - ref["rgbs"]).max(1)
    # log2 / sqrt of the footprint differ in the last bit between libm and the device: a probe count or a level weight
    # may flip on a handful of pixels; everything else agrees to the 8-bit step
    assert (d > 1.5 / 255).mean() < 2e-3, (d > 1.5 / 255).mean()
    cov = ref["depths"][:, 0] > 0
    changed = np.abs(rgb - plain[0].cpu().numpy()).max(1) > 0
    assert changed[cov].mean() > 0.3                                                   # the filter does something
    gx = lambda im: np.abs(np.diff(im, axis=-1)).mean()                                # noqa: E731
    assert gx(rgb) < gx(plain[0].cpu().numpy())                                        # ... namely smooth the minified texture


def test_rasterizer_msaa_wider_than_640(dev, scene_store):
    """Multisampled renders wider than 640 px (720 x 540 T-LESS frames, 1280 x 720 visualisation renders) take the 512-thread /
    6400-key instantiation of the band kernel (the 256-thread / 3200-key one holds 640 five-key pixels per row): same
    definition, against the oracle; a 1281-wide multisampled render is refused with a message."""
    from happypose_amd import ops
    from oracle import native

    n = 2
    T = _poses(n, 13, zlo=0.4, zhi=0.6)
    K = np.tile(np.array([[1100.0, 0, 360], [0, 1100.0, 270], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = np.array([0, 2], np.int32)
    gpu = ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (540, 720), render_normals=True,
                        render_depth=True, msaa=True, aniso=True)
    ref = native.rasterize(scene_store.packed, obj, T, K, (540, 720), True, True, False, msaa=True, aniso=True)
    assert (ref["depths"] > 0).mean() > 0.05
    assert np.array_equal(gpu[2].cpu().numpy(), ref["depths"])                   # coverage and depth: exact integer / same-order fp32 arithmetic
    for got, want in ((gpu[0].cpu().numpy(), ref["rgbs"]), (gpu[1].cpu().numpy(), ref["normals"])):
        d = np.abs(got - want).max(1)
        assert (d > 1.5 / 255).mean() < 2e-3 and d.max() <= 0.5 + 1e-6
    with pytest.raises(Exception, match="too wide"):
        ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (64, 1281), msaa=True)


def test_rasterizer_is_watertight_at_frame_size(dev):
    """Size-independent property of the fixed-point coverage rules at the full 480 x 640 frame (the oracle's twin:
    tests/test_oracle_pixels.py::test_rasteriser_coverage_rules_are_watertight_and_order_independent): a finely tessellated
    white sheet (8192 jittered triangles, random diagonals) under a tilted camera leaves no sample uncovered -- every interior
    pixel is exactly 1.0 after the 8-bit multisample resolve -- and flipping windings / rotating index triples changes no
    pixel of mask, depth or colour; culling off (an open sheet) and on (it cannot engage: the sheet is not closed)."""
    from happypose_amd import ops
    from happypose_amd.mesh_io import MeshData
    from happypose_amd.mesh_store import RigidObject, RigidObjectDataset

    def sheet(rs, n, flip=None, rot=None):
        g = np.linspace(-0.5, 0.5, n)
        xx, yy = np.meshgrid(g, g)
        v = np.stack([xx, yy, np.zeros_like(xx)], -1).reshape(-1, 3)
        inner = (np.abs(v[:, 0]) < 0.499) & (np.abs(v[:, 1]) < 0.499)
        v[inner, :2] += rs.uniform(-0.4, 0.4, (int(inner.sum()), 2)) / (n - 1)
        faces = []
        for i in range(n - 1):
            for j in range(n - 1):
                a, b, c, d = i * n + j, i * n + j + 1, (i + 1) * n + j + 1, (i + 1) * n + j
                faces += [[a, b, c], [a, c, d]] if rs.rand() < 0.5 else [[a, b, d], [b, c, d]]
        f = np.array(faces, np.int32)
        if flip is not None:
            f[flip] = f[flip][:, ::-1]
        if rot is not None:
            f = np.stack([np.roll(t, r) for t, r in zip(f, rot)])
        nrm = np.tile([0, 0, -1.0], (len(v), 1)).astype(np.float32)
        return MeshData(v, f, nrm, None, np.full((len(v), 4), 255, np.uint8))

    n = 65
    nf = 2 * (n - 1) ** 2
    rs = np.random.RandomState(4)
    ds = RigidObjectDataset([RigidObject("sheet", sheet(np.random.RandomState(3), n)),
                             RigidObject("sheet_shuffled", sheet(np.random.RandomState(3), n, flip=rs.rand(nf) < 0.5, rot=rs.randint(0, 3, nf)))])
    store = ops.MeshStore(ds, device=dev)
    T = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    ax, ay = np.deg2rad(38.0), np.deg2rad(-21.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    T[:, :3, :3] = (Ry @ Rx).astype(np.float32)
    T[:, :3, 3] = (0.013, -0.021, 1.05)
    K = np.tile(np.array([[560.0, 0, 320.3], [0, 555.0, 239.6], [0, 0, 1]], np.float32), (2, 1, 1))
    obj = torch.as_tensor(np.array([0, 1], np.int32))
    for msaa in (False, True):
        for cull in (True, False):
            store.set_backface_culling(cull)
            rgb, _, dep, msk = ops.rasterize(store, obj, torch.as_tensor(T), torch.as_tensor(K), (480, 640), render_depth=True,
                                             render_binary_mask=True, msaa=msaa, aniso=msaa)
            rgb, dep, m = rgb.cpu().numpy(), dep.cpu().numpy(), msk.cpu().numpy()[:, 0]
            assert np.array_equal(m[0], m[1]) and 0.25 < m[0].mean() < 0.8
            assert np.array_equal(rgb[0], rgb[1])
            np.testing.assert_allclose(dep[0], dep[1], rtol=0, atol=2e-6)  # the 1/z plane is set up from the first listed corner
            inner = m[0].copy()
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    inner &= np.roll(np.roll(m[0], dy, 0), dx, 1)
            assert inner.sum() > 0.95 * m[0].sum()
            assert (rgb[0][:, inner] == 1.0).all()  # one uncovered sample would resolve to 191 / 255
    store.set_backface_culling(True)


CONVENTION_FLIPS = [
    dict(msaa_x=(0.25, 0.75, 0.25, 0.75), msaa_y=(0.25, 0.25, 0.75, 0.75)),            # ordered-grid samples
    dict(msaa_x=(0.625, 0.125, 0.875, 0.375), msaa_y=(0.125, 0.375, 0.625, 0.875)),    # the mirrored rotated grid
    dict(aniso_max=4),
    dict(aniso_round=1), dict(aniso_round=2),
    dict(lod_from=1), dict(lod_from=2), dict(lod_bias=-0.5),
    dict(normal_axis=(0, 2, 1), normal_sign=(1.0, 1.0, -1.0)),
    dict(normal_sign=(1.0, 1.0, 1.0)),                                                   # the camera frame itself
    dict(normal_axis=(2, 0, 1), normal_sign=(-1.0, -1.0, 1.0)),
]


def _render_both(scene_store, n=9, seed=7):
    from happypose_amd import ops
    from oracle import native

    T = _poses(n, seed, zlo=0.3, zhi=1.6)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = (np.arange(n) % 3).astype(np.int32)
    gpu = ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320), render_normals=True,
                        render_depth=True, msaa=True, aniso=True)
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, False, msaa=True, aniso=True)
    return gpu, ref


def test_rasterizer_conventions_record(dev, scene_store):
    """The calibration path of a-6: multisample positions, the anisotropic footprint rule and the eye-normal axis map are a
    RECORD (``hp_mesh_store_set_raster_conventions``, mirrored by ``hp_oracle_set_raster_conventions``), not compile-time constants.
    Every flipped convention keeps HIP == oracle to the tolerances of the fixed-convention tests AND changes the render; back
    on the defaults the render is bit-identical to the one made before any flip (so golden G10, generated with the
    constants, still holds: tests/test_gpu_pipeline.py::test_*_vs_reference_golden_g10 run on the defaults)."""
    from happypose_amd import ops
    from oracle import native

    assert scene_store.get_raster_conventions() == ops.RASTER_CONVENTION_DEFAULTS
    base_gpu, base_ref = _render_both(scene_store)
    try:
        for flip in CONVENTION_FLIPS:
            scene_store.set_raster_conventions(flip)
            native.set_raster_conventions(flip)
            assert scene_store.get_raster_conventions() == dict(ops.RASTER_CONVENTION_DEFAULTS, **flip)
            gpu, ref = _render_both(scene_store)
            assert torch.equal(gpu[2], base_gpu[2]), flip                       # depth stays centre-sampled, whatever the record
            for k, (got, want, base) in enumerate(((gpu[0], ref["rgbs"], base_gpu[0]), (gpu[1], ref["normals"], base_gpu[1]))):
                d = np.abs(got.cpu().numpy() - want).max(1)
                assert (d > 1.5 / 255).mean() < 2e-3 and d.max() <= 0.5 + 1e-6, (flip, k, (d > 1.5 / 255).mean(), d.max())
            touches_rgb = any(key.startswith(("msaa", "aniso", "lod")) for key in flip)
            touches_nrm = any(key.startswith(("msaa", "normal")) for key in flip)
            assert (not torch.equal(gpu[0], base_gpu[0])) == touches_rgb, flip
            assert (not torch.equal(gpu[1], base_gpu[1])) == touches_nrm, flip
    finally:
        scene_store.set_raster_conventions(None)
        native.set_raster_conventions(None)
    again_gpu, again_ref = _render_both(scene_store)
    for a, b in zip(again_gpu[:3], base_gpu[:3]):
        assert torch.equal(a, b)
    assert np.array_equal(again_ref["rgbs"], base_ref["rgbs"]) and np.array_equal(again_ref["normals"], base_ref["normals"])
    with pytest.raises(AssertionError):
        scene_store.set_raster_conventions(dict(msaa_x=(0.0, 0.5, 0.5, 0.5)))         # a sample on the pixel border
    with pytest.raises(KeyError):
        scene_store.set_raster_conventions(dict(samples=4))


def test_rasterizer_outliers_are_boundary_pixels(dev, scene_store):
    """The multisample / texture-filter tests above allow a small fraction of pixels beyond the 8-bit step.  This pins WHAT
    those pixels are: exactly the ones whose oracle value itself moves when the conventions are perturbed at round-off
    level -- a sample within 3e-5 px of a triangle edge (it may fall on either side in fp32), a footprint ratio within 1e-4 of
    an integer (the probe count flips) or a level of detail within 2e-5 of a mip boundary / zero.  Every HIP-vs-oracle
    outlier must be such a pixel, or match one of the perturbed oracle renders."""
    from happypose_amd import ops
    from oracle import native

    prev = scene_store.set_backface_culling(False)  # this test is about the conventions: the two-sided render (culling: the next test)
    try:
        gpu, ref = _render_both(scene_store, n=12, seed=5)
    finally:
        scene_store.set_backface_culling(prev)
    rgb, nrm = gpu[0].cpu().numpy(), gpu[1].cpu().numpy()
    e = 3e-5
    d0 = ops.RASTER_CONVENTION_DEFAULTS
    perturbed = [dict(msaa_x=tuple(x + sx * e for x in d0["msaa_x"]), msaa_y=tuple(y + sy * e for y in d0["msaa_y"]))
                 for sx in (-1, 0, 1) for sy in (-1, 0, 1) if (sx, sy) != (0, 0)]
    perturbed += [dict(aniso_ratio_bias=b) for b in (-1e-4, 1e-4)] + [dict(lod_bias=b) for b in (-2e-5, 2e-5)]
    sens_rgb = np.zeros(rgb.shape[:1] + rgb.shape[2:], bool)
    sens_nrm = np.zeros_like(sens_rgb)
    match_rgb = np.zeros_like(sens_rgb)
    match_nrm = np.zeros_like(sens_rgb)
    try:
        for pc in perturbed:
            native.set_raster_conventions(pc)
            pr = _oracle_only(scene_store, 12, 5)
            sens_rgb |= np.abs(pr["rgbs"] - ref["rgbs"]).max(1) > 0
            sens_nrm |= np.abs(pr["normals"] - ref["normals"]).max(1) > 0
            match_rgb |= np.abs(pr["rgbs"] - rgb).max(1) <= 1.5 / 255
            match_nrm |= np.abs(pr["normals"] - nrm).max(1) <= 1.5 / 255
    finally:
        native.set_raster_conventions(None)
    out_rgb = np.abs(rgb - ref["rgbs"]).max(1) > 1.5 / 255
    out_nrm = np.abs(nrm - ref["normals"]).max(1) > 1.5 / 255
    assert out_rgb.mean() < 2e-3 and out_nrm.mean() < 5e-4
    # explained = the oracle itself is sensitive there, or the HIP value IS one of the perturbed oracle values
    unexplained_rgb = out_rgb & ~(sens_rgb | match_rgb)
    unexplained_nrm = out_nrm & ~(sens_nrm | match_nrm)
    assert unexplained_rgb.sum() <= max(2, 0.02 * out_rgb.sum()), (int(unexplained_rgb.sum()), int(out_rgb.sum()))
    assert unexplained_nrm.sum() <= max(2, 0.02 * out_nrm.sum()), (int(unexplained_nrm.sum()), int(out_nrm.sum()))
    assert sens_rgb.mean() < 0.05  # the perturbations are round-off sized: they must not touch ordinary pixels


def _render_cull_pair(store, obj, T, K, res=(240, 320)):
    from happypose_amd import ops

    outs = []
    for cull in (True, False):
        prev = store.set_backface_culling(cull)
        try:
            out = ops.rasterize(store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), res, render_normals=True,
                                render_depth=True, msaa=True, aniso=True)
        finally:
            store.set_backface_culling(prev)
        outs.append([o.cpu().numpy() for o in out[:3]])
    a, b = outs
    diff = np.zeros(a[0].shape[:1] + a[0].shape[2:], bool)
    for x, y in zip(a, b):
        diff |= (x != y).reshape(x.shape[0], -1, *x.shape[-2:]).any(1)
    return a, b, diff


def test_rasterizer_backface_culling_keeps_the_image(dev, scene_store):
    """``hp_mesh_store_set_backface_culling`` (default on): the set-up pass drops triangles whose inward side is turned to the
    camera when they belong to a closed, consistently oriented CONNECTED COMPONENT seen from outside -- the reference renders
    two-sided (TB/renderer/panda3d_scene_renderer.py:102), and of a closed surface only outward faces are ever visible.  The
    facing test is the exact sign of the snapped area, vertices at the same position snap to the same point (texture seams are
    watertight), so culled and two-sided renders may differ only where a sample lies exactly on a silhouette edge: < 1e-5 of
    the pixels, depth / mask included.  An OPEN surface (the same mesh without a strip of faces) is never culled."""
    T = _poses(12, 5, zlo=0.3, zhi=1.6)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (12, 1, 1))
    obj = (np.arange(12) % 3).astype(np.int32)
    a, b, diff = _render_cull_pair(scene_store, obj, T, K)
    assert diff.mean() < 1e-5, diff.mean()
    assert (a[2] > 0).mean() > 0.02  # the objects are in view
    # an open surface: remove a strip of faces from every object and rebuild the store
    open_store = _open_copy(dev)
    c, d, diff_open = _render_cull_pair(open_store, obj, T, K)
    assert not diff_open.any()


def _shell(radius, centre=(0.0, 0.0, 0.0), flip=False, n_lat=24, n_lon=36, seed=0):
    """A closed UV-sphere-like shell as MeshData (outward winding; ``flip``: every face reversed)."""
    import dataclasses

    from happypose_amd.synthetic import make_mesh

    m = make_mesh(seed, n_lat=n_lat, n_lon=n_lon, diameter=2 * radius, tex_size=64)
    verts = m.vertices + np.asarray(centre, np.float32)
    faces = m.faces[:, ::-1].copy() if flip else m.faces
    normals = -m.normals if flip else m.normals
    return dataclasses.replace(m, vertices=verts.astype(np.float32), faces=np.ascontiguousarray(faces), normals=normals.astype(np.float32))


def _merge(meshes):
    import dataclasses

    off, vs, ns, uvs, cols, fs = 0, [], [], [], [], []
    for m in meshes:
        vs.append(m.vertices); ns.append(m.normals); uvs.append(m.uvs); fs.append(m.faces + off)
        if m.colors is not None:
            cols.append(m.colors)
        off += len(m.vertices)
    return dataclasses.replace(meshes[0], vertices=np.concatenate(vs), normals=np.concatenate(ns), uvs=np.concatenate(uvs),
                               faces=np.concatenate(fs).astype(np.int32), colors=np.concatenate(cols) if cols else None)


def test_backface_culling_is_decided_per_connected_component(dev, golden_dir):
    """Orientation per connected component of the welded mesh (``api.cpp: mesh_cull_flags``): a part with flipped winding or a
    nested, inverted shell must not vanish (a single sign for the whole object -- round 4 -- culled the VISIBLE faces of the
    minority part).  (i) two nested shells, the inner one inverted (invisible either way: the outer shell hides it, and its own
    flag is -1); (ii) two disjoint shells with opposite winding, both fully visible; (iii) a closed shell next to an open sheet;
    (iv) the reference's own test asset ``obj_000001``.  Culled == two-sided up to samples exactly on silhouette edges."""
    from happypose_amd.mesh_store import RigidObject, RigidObjectDataset
    from happypose_amd.ops import MeshStore

    sheet = _shell(0.04, centre=(0.0, 0.09, 0.0))
    import dataclasses
    sheet = dataclasses.replace(sheet, faces=np.ascontiguousarray(sheet.faces[: len(sheet.faces) // 2]))  # half a shell: open
    objs = [
        RigidObject("nested", _merge([_shell(0.06), _shell(0.03, flip=True, seed=1)]), mesh_units="m"),
        RigidObject("disjoint", _merge([_shell(0.04, centre=(-0.05, 0, 0)), _shell(0.04, centre=(0.05, 0, 0), flip=True, seed=2)]), mesh_units="m"),
        RigidObject("mixed", _merge([_shell(0.04, centre=(0.0, -0.03, 0.0)), sheet]), mesh_units="m"),
        RigidObject("asset", golden_dir / "obj_000001.npz", mesh_units="mm"),
    ]
    store = MeshStore(RigidObjectDataset(objs), dev)
    n = 16
    T = _poses(n, 21, zlo=0.35, zhi=0.7)
    K = np.tile(np.array([[700.0, 0, 160], [0, 700.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = (np.arange(n) % 4).astype(np.int32)
    a, b, diff = _render_cull_pair(store, obj, T, K)
    assert diff.mean() < 1e-5, [diff[obj == k].mean() for k in range(4)]
    for k in range(4):
        assert (b[2][obj == k] > 0).mean() > 0.01, k  # every object is in view
    # the flipped shell of "disjoint" is really there in the culled render: both halves of the image hold object pixels
    for v in np.nonzero(obj == 1)[0]:
        cov_c, cov_t = a[2][v, 0] > 0, b[2][v, 0] > 0
        assert abs(int(cov_c.sum()) - int(cov_t.sum())) <= 2 and cov_t.sum() > 500


def test_two_stores_hold_their_own_renderer_state(dev, scene_store):
    """No process-wide renderer state: conventions and the culling switch live on the mesh store.  Two stores of the
    same objects in one process, one with a flipped normal map and ordered-grid samples, render differently; the first
    store's output does not move."""
    from happypose_amd import ops
    from happypose_amd.synthetic import make_object_dataset

    other = ops.MeshStore(make_object_dataset(3, seed=1, tex_size=256), dev)
    base, _ = _render_both(scene_store)
    other.set_raster_conventions(dict(normal_sign=(1.0, 1.0, 1.0), msaa_x=(0.25, 0.75, 0.25, 0.75), msaa_y=(0.25, 0.25, 0.75, 0.75)))
    other.set_backface_culling(False)
    assert scene_store.get_raster_conventions() == ops.RASTER_CONVENTION_DEFAULTS
    T = _poses(9, 7, zlo=0.3, zhi=1.6)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (9, 1, 1))
    obj = (np.arange(9) % 3).astype(np.int32)
    o = ops.rasterize(other, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320), render_normals=True,
                      render_depth=True, msaa=True, aniso=True)
    again, _ = _render_both(scene_store)
    assert not torch.equal(o[1], base[1]) and not torch.equal(o[0], base[0])
    for x, y in zip(again[:3], base[:3]):
        assert torch.equal(x, y)


def test_rasterizer_chunked_launch_is_bit_identical(dev, scene_store, tmp_path):
    """``hp_rasterize`` renders in chunks of views when the per-(view, band) triangle lists of a call would exceed the list
    budget (8 GB: never at the benchmark sizes any more).  The chunked path -- scratch re-used from chunk to chunk, ``view0``
    offsets -- must give the bits of the one-chunk launch: ``HP_RASTER_CHUNK_VIEWS=5`` (read once per process: a second
    interpreter) against this process, 12 views in the reference render state."""
    import subprocess
    import sys

    gpu, _ = _render_both(scene_store, n=12, seed=5)
    here = [g.cpu().numpy() for g in gpu[:3]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_gpu_kernels as t\n"
        "from happypose_amd.ops import MeshStore\n"
        "from happypose_amd.synthetic import make_object_dataset\n"
        "store = MeshStore(make_object_dataset(3, seed=1, tex_size=256), torch.device('cuda:0'))\n"
        "gpu, _ = t._render_both(store, n=12, seed=5)\n"
        "np.savez(%r, *[g.cpu().numpy() for g in gpu[:3]])\n" % (root, os.path.join(root, "tests"), str(tmp_path / "chunked.npz")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env={**os.environ, "HP_RASTER_CHUNK_VIEWS": "5"})
    assert out.returncode == 0, out.stdout + out.stderr
    there = np.load(tmp_path / "chunked.npz")
    for k, x in enumerate(here):
        assert np.array_equal(x, there[f"arr_{k}"]), k


def _open_copy(dev):
    """The scene_store objects with a strip of faces removed: open surfaces (a boundary edge = never culled)."""
    import dataclasses

    from happypose_amd.mesh_store import RigidObject, RigidObjectDataset
    from happypose_amd.ops import MeshStore
    from happypose_amd.synthetic import make_mesh

    objs = []
    for i in range(3):
        mesh = make_mesh(1000 + i, n_lat=72, n_lon=112, diameter=float(np.random.RandomState(1).uniform(0.10, 0.25)), tex_size=256)
        mesh = dataclasses.replace(mesh, faces=np.ascontiguousarray(mesh.faces[400:]))
        objs.append(RigidObject(label=f"obj_{i + 1:06d}", mesh_path=mesh, mesh_units="m"))
    return MeshStore(RigidObjectDataset(objs), dev)


def _oracle_only(scene_store, n, seed):
    from oracle import native

    T = _poses(n, seed, zlo=0.3, zhi=1.6)
    K = np.tile(np.array([[900.0, 0, 160], [0, 900.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = (np.arange(n) % 3).astype(np.int32)
    return native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, False, msaa=True, aniso=True)


def test_rasterizer_reference_test_scene(dev, golden_dir):
    """Scene and structural asserts of the reference's renderer test
    (tests/test_batch_renderer_panda3d.py:43-69,105-122,166-179) on its own asset."""
    from happypose_amd import ops
    from happypose_amd.mesh_store import RigidObject, RigidObjectDataset
    from oracle import geometry as G
    from oracle import native

    ds = RigidObjectDataset([RigidObject("my_favorite_object_label", golden_dir / "obj_000001.npz", mesh_units="mm"),
                             RigidObject("NOT_USED", golden_dir / "obj_000001.npz", mesh_units="mm")])
    store = ops.MeshStore(ds, dev)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = G.unitquat_to_rotmat(np.array([0.5, 0.5, -0.5, 0.5]))
    T[:3, 3] = (0, 0, 0.3)
    K = np.array([[300, 0, 320], [0, 300, 240], [0, 0, 1]], np.float32)
    Nc = 4
    TT, KK = np.tile(T, (Nc, 1, 1)), np.tile(K, (Nc, 1, 1))
    rgb, nrm, dep, msk = ops.rasterize(store, torch.zeros(Nc, dtype=torch.int32), torch.as_tensor(TT),
                                       torch.as_tensor(KK), (480, 640), True, True, True)
    assert rgb.shape == (Nc, 3, 480, 640) and dep.shape == (Nc, 1, 480, 640) and msk.dtype == torch.bool
    assert torch.equal(rgb[0], rgb[1]) and torch.equal(nrm[0], nrm[1]) and torch.equal(dep[0], dep[1])
    assert rgb[0, :, 0, 0].tolist() == [0, 0, 0] and dep[0, 0, 0, 0] == 0 and not msk[0, 0, 0, 0]
    assert nrm[0, :, 0, 0].tolist() == [0, 0, 0]
    assert (rgb[0, :, 240, 320] > 0).all() and (nrm[0, :, 240, 320] > 0).all()
    assert 0 < dep[0, 0, 240, 320] < 0.3 and msk[0, 0, 240, 320]
    ref = native.rasterize(store.packed, np.zeros(Nc, np.int32), TT, KK, (480, 640), True, True, True)
    _compare_renders((rgb, nrm, dep, msk), ref)
    with pytest.raises(AssertionError):  # mask without depth (test_scene_renderer_panda3d.py:206-214)
        ops.rasterize(store, torch.zeros(1, dtype=torch.int32), torch.as_tensor(TT[:1]), torch.as_tensor(KK[:1]),
                      (480, 640), render_binary_mask=True)


def test_rasterizer_edge_cases(dev, scene_store):
    from happypose_amd import ops
    from oracle import native

    T = _poses(4, 9)
    T[0, 0, 0] = np.nan           # non-finite pose -> zero image (panda3d_batch_renderer.py:81-111)
    T[1, :3, 3] = (0, 0, 0.05)    # camera inside the object: near-plane clipping, huge triangles
    T[2, :3, 3] = (0, 0, -0.5)    # behind the camera
    T[3, :3, 3] = (0.5, 0.4, 0.6)  # mostly off-screen
    K = np.tile(np.array([[600.0, 0, 160], [0, 600.0, 120], [0, 0, 1]], np.float32), (4, 1, 1))
    obj = np.array([0, 1, 2, 0], np.int32)
    gpu = ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320),
                        True, True, False)
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), True, True, False)
    assert float(gpu[0][0].abs().max()) == 0 and float(gpu[0][2].abs().max()) == 0
    _compare_renders(gpu, ref, max_cov_mismatch=2e-3)
    # empty batch
    e = ops.rasterize(scene_store, torch.zeros(0, dtype=torch.int32), torch.zeros(0, 4, 4), torch.zeros(0, 3, 3), (240, 320))
    assert e[0].shape == (0, 3, 240, 320)


def test_rasterizer_lights_and_nhwc(dev, scene_store):
    from happypose_amd import ops
    from oracle import native

    n = 4
    T = _poses(n, 11)
    K = np.tile(np.array([[800.0, 0, 160], [0, 800.0, 120], [0, 0, 1]], np.float32), (n, 1, 1))
    obj = np.array([0, 1, 2, 1], np.int32)
    amb = np.full((n, 3), 0.1, np.float32)
    r = 10 * scene_store.packed.radius[obj]
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    lp = (dirs[None] * r[:, None, None]).astype(np.float32)
    lc = np.full((n, 6, 3), 0.4, np.float32)
    gpu = ops.rasterize(scene_store, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), (240, 320),
                        ambient=torch.as_tensor(amb), light_pos=torch.as_tensor(lp), light_col=torch.as_tensor(lc))
    ref = native.rasterize(scene_store.packed, obj, T, K, (240, 320), ambient=amb, light_pos=lp, light_col=lc)
    _compare_renders(gpu, ref)
    # NHWC slices, 2 views per item, fused depth normalisation
    x = torch.zeros((2, 240, 320, 20), device=dev)
    TV = torch.as_tensor(T).view(2, 2, 4, 4)
    KV = torch.as_tensor(K).view(2, 2, 3, 3)
    z = torch.tensor([0.5, 0.7], device=dev)
    ops.rasterize_into(scene_store, x, 4, torch.as_tensor(obj[[0, 2]]), TV, KV, True, True, z, 2)
    ref2 = native.rasterize(scene_store.packed, obj[[0, 0, 2, 2]], T, K, (240, 320), True, True)
    xs = x.cpu().numpy()
    assert np.all(xs[..., :4] == 0) and np.all(xs[..., 18:] == 0)
    for it in range(2):
        for v in range(2):
            sl = xs[it, :, :, 4 + 7 * v: 4 + 7 * (v + 1)]
            k = 2 * it + v
            cov = ref2["depths"][k, 0] > 0
            got_cov = sl[..., :3].sum(-1) > 0
            assert (cov != got_cov).mean() < 5e-4
            ok = cov == got_cov
            np.testing.assert_allclose(sl[..., :3][ok], ref2["rgbs"][k].transpose(1, 2, 0)[ok], atol=1.01 / 255)
            np.testing.assert_allclose(sl[..., 3:6][ok], ref2["normals"][k].transpose(1, 2, 0)[ok], atol=1.01 / 255)
            dn = np.clip(ref2["depths"][k, 0] / float(z[it]), 0, 2) - 1
            assert np.abs(sl[..., 6][ok] - dn[ok]).max() < 1e-2
            assert (np.abs(sl[..., 6][ok] - dn[ok]) > 1e-5).mean() < 1e-3


def test_crop_and_raster_into_f16_input(dev, scene_store):
    """fp16 destinations (the input tensor of an fp16 network plan): the same values as the fp32
    destinations, rounded to fp16 once -- bit for bit."""
    from happypose_amd import ops

    rs = np.random.RandomState(3)
    img = torch.as_tensor(rs.rand(2, 4, 480, 640).astype(np.float32), device=dev)
    boxes = torch.as_tensor(np.array([[100.3, 80.2, 420.7, 320.1], [-50, -40, 300, 222.5], [0, 0, 640, 480]], np.float32))
    ids = torch.as_tensor(np.array([0, 1, 1], np.int32))
    z = torch.tensor([0.5, 0.7, 0.6], device=dev)
    T = torch.as_tensor(_poses(6, 12)).view(3, 2, 4, 4)
    K = torch.as_tensor(np.tile(np.array([[800.0, 0, 160], [0, 800.0, 120], [0, 0, 1]], np.float32), (3, 2, 1, 1)))
    obj = torch.as_tensor(np.array([0, 2, 1], np.int32))
    xs = {}
    for dt in (torch.float32, torch.float16):
        x = torch.zeros((3, 240, 320, 24), device=dev, dtype=dt)
        ops.crop_roi_align(img, boxes, ids, out=x, depth_norm_z=z, depth_norm_mode=2, n_channels=4)
        ops.rasterize_into(scene_store, x, 4, obj, T, K, True, True, z, 2)
        xs[dt] = x
    assert xs[torch.float32][..., 4:18].abs().sum() > 0
    assert torch.equal(xs[torch.float32].half(), xs[torch.float16])


# ------------------------------------------------------------------------------------ crop
def test_crop_vs_oracle(dev):
    from happypose_amd import ops
    from oracle import native

    rs = np.random.RandomState(0)
    img = rs.rand(2, 4, 480, 640).astype(np.float32)
    img[:, 3] = np.where(rs.rand(2, 480, 640) < 0.2, 0.0, 0.3 + img[:, 3])  # depth with holes
    boxes = np.array([[100.3, 80.2, 420.7, 320.1], [-50, -40, 300, 222.5], [500, 400, 700, 550],
                      [10, 10, 10.5, 10.2], [0, 0, 640, 480], [300, 200, 340, 230]], np.float32)
    ids = np.array([0, 1, 0, 1, 1, 0], np.int32)
    for C in (3, 4):
        im = np.ascontiguousarray(img[:, :C])
        ref = native.crop_images(im, boxes, ids)
        got = ops.crop_roi_align(torch.as_tensor(im, device=dev), torch.as_tensor(boxes), torch.as_tensor(ids))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    # NHWC destination + fused depth normalisation
    net_in = torch.zeros((6, 240, 320, 8), device=dev)
    z = torch.linspace(0.4, 0.9, 6, device=dev)
    ops.crop_roi_align(torch.as_tensor(img, device=dev), torch.as_tensor(boxes), torch.as_tensor(ids), out=net_in,
                       depth_norm_z=z, depth_norm_mode=2)
    ref = native.crop_images(img, boxes, ids)
    o = net_in.cpu().numpy()
    np.testing.assert_allclose(o[..., :3], ref[:, :3].transpose(0, 2, 3, 1), rtol=1e-5, atol=1e-6)
    dn = np.clip(ref[:, 3] / z.cpu().numpy()[:, None, None], 0, 2) - 1
    np.testing.assert_allclose(o[..., 3], dn, rtol=1e-5, atol=1e-5)
    assert np.all(o[..., 4:] == 0)
    e = ops.crop_roi_align(torch.as_tensor(img, device=dev), torch.zeros(0, 4), torch.zeros(0, dtype=torch.int32))
    assert e.shape == (0, 4, 240, 320)
    # the crop as owner of an 8-float pixel record (HP_CROP_FULL_RECORD8): same three channels, the rest of the record zeroed
    net_in = torch.full((6, 240, 320, 8), 7.0, device=dev)
    ops.crop_roi_align(torch.as_tensor(img, device=dev), torch.as_tensor(boxes), torch.as_tensor(ids), out=net_in, n_channels=3,
                       owns_record=True)
    o = net_in.cpu().numpy()
    np.testing.assert_allclose(o[..., :3], ref[:, :3].transpose(0, 2, 3, 1), rtol=1e-5, atol=1e-6)
    assert np.all(o[..., 3:] == 0)
    plain = torch.full((6, 240, 320, 8), 7.0, device=dev)
    ops.crop_roi_align(torch.as_tensor(img, device=dev), torch.as_tensor(boxes), torch.as_tensor(ids), out=plain, n_channels=3)
    assert torch.equal(plain[..., :3], net_in[..., :3]) and bool((plain[..., 3:] == 7.0).all())


# -------------------------------------------------------------------------------- geometry
def _store_from_points(pts_list, dev):
    """objects whose vertices are the given point sets (faces are dummies)."""
    from happypose_amd.mesh_io import MeshData
    from happypose_amd.mesh_store import RigidObject, RigidObjectDataset
    from happypose_amd.ops import MeshStore

    objs = []
    for i, p in enumerate(pts_list):
        f = np.array([[0, 1, 2]], np.int32)
        objs.append(RigidObject(f"o{i}", MeshData(vertices=p.astype(np.float64), faces=f,
                                                  normals=np.tile([0, 0, 1.0], (len(p), 1)).astype(np.float32))))
    return MeshStore(RigidObjectDataset(objs), dev)


def test_pose_prep_golden_g7(dev, golden_dir):
    """hp_pose_prep + hp_pose_update against the reference's own outputs (golden G7)."""
    from happypose_amd import ops

    g = np.load(golden_dir / "g7_iteration.npz")
    b = len(g["T"])
    store = _store_from_points(list(g["pts"]), dev)
    assert store.n_pad == 2000
    # the kernel sub-samples with the deterministic ids; G7 projected ALL 2000 points, which is
    # the same set (a permutation) when n_points == n_pad
    out = ops.pose_prep(store, torch.as_tensor(g["T"]), torch.as_tensor(g["K"]), torch.arange(b),
                        torch.arange(b), (480, 640), normalize=True)
    c = lambda t: t.cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(c(out["TCO"]), g["T_norm"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(c(out["boxes_rend"]), g["boxes_rend"], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(c(out["boxes_crop"]), g["boxes_crop"], rtol=1e-5, atol=3e-3)
    np.testing.assert_allclose(c(out["K_crop"])[:, 0], g["K_crop"], rtol=2e-5, atol=3e-3)
    np.testing.assert_allclose(c(out["tCR"]), g["T_norm"][:, :3, 3], rtol=2e-6, atol=2e-6)
    upd = ops.pose_update(out["TCO"], out["K_crop"], torch.as_tensor(g["pose9"]), out["tCR"])
    np.testing.assert_allclose(c(upd), g["T_out"], rtol=2e-5, atol=2e-6)


def test_pose_update_and_init_golden_g4(dev, golden_dir):
    from happypose_amd import ops

    g = np.load(golden_dir / "g4_pose_update.npz")
    T, Kc, p9 = (torch.as_tensor(g[k], device=dev) for k in ("T", "K_crop", "pose9"))
    c = lambda t: t.cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(c(ops.pose_update(T, Kc, p9, torch.as_tensor(g["tCR"], device=dev))), g["upd_ref"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c(ops.pose_update(T, Kc, p9, None)), g["upd_cosy"], rtol=1e-5, atol=1e-6)
    b = len(g["T"])
    store = _store_from_points(list(g["mpts"]), dev)
    ar = torch.arange(b)
    got = ops.tco_init_autodepth(store, torch.as_tensor(g["det_boxes"]), torch.as_tensor(g["K"]), ar, ar,
                                 R=torch.as_tensor(g["Rg"]))
    np.testing.assert_allclose(c(got), g["init_R"], rtol=1e-5, atol=1e-6)
    got = ops.tco_init_autodepth(store, torch.as_tensor(g["det_boxes"]), torch.as_tensor(g["K"]), ar, ar)
    np.testing.assert_allclose(c(got), g["init_zup"], rtol=1e-5, atol=1e-6)


def test_pose_prep_multiview_vs_oracle(dev, scene_store):
    from happypose_amd import ops
    from oracle import geometry as G

    b = 9
    T = _poses(b, 21)
    T[:, :3, :3] += 0.02 * np.random.RandomState(1).normal(size=(b, 3, 3)).astype(np.float32)
    K = np.array([[[600.0, 0, 320], [0, 600.0, 240], [0, 0, 1]]], np.float32)
    obj = (np.arange(b) % 3).astype(np.int32)
    out = ops.pose_prep(scene_store, torch.as_tensor(T), torch.as_tensor(K), torch.zeros(b, dtype=torch.int32),
                        torch.as_tensor(obj), (480, 640), multiview_type="TCO+front_3views", normalize=True)
    Tn = G.normalize_T(T)
    tCR = Tn[:, :3, 3]
    TCV = G.make_TCO_multiview(Tn, tCR, "TCO+front_3views", 4)
    np.testing.assert_allclose(out["TCV_O"].cpu().numpy(), TCV, rtol=1e-5, atol=2e-6)
    pts = scene_store.mesh_db.points[obj]
    Kb = np.repeat(K, b, 0)
    for v in range(4):
        ids = G.sample_point_ids(scene_store.n_pad, 2000 if v == 0 else 200)
        br, bc = G.crop_boxes_from_pose(pts[:, ids], Kb, TCV[:, v], TCV[:, v, :3, 3], (480, 640))
        Kc = G.get_K_crop_resize(Kb, bc, (480, 640), (240, 320))
        np.testing.assert_allclose(out["K_crop"][:, v].cpu().numpy(), Kc, rtol=5e-5, atol=5e-3)
        if v == 0:
            np.testing.assert_allclose(out["boxes_rend"].cpu().numpy(), br, rtol=1e-5, atol=2e-3)
            np.testing.assert_allclose(out["boxes_crop"].cpu().numpy(), bc, rtol=1e-5, atol=3e-3)
    # extra views look at the reference point: it projects to the crop centre
    for v in range(1, 4):
        Tv = out["TCV_O"][:, v].cpu().numpy()
        assert np.allclose(Tv[:, :2, 3], 0, atol=1e-5)
        # view 1 sits at camera 0's centre, views 2/3 one radius to its right/left
        dist = np.linalg.norm(tCR, axis=1) * (1.0 if v == 1 else np.sqrt(2.0))
        assert np.allclose(np.linalg.norm(Tv[:, :3, 3], axis=1), dist, rtol=1e-5)


# ----------------------------------------------------------------------------- convolution
def _conv_ref(x_nhwc, w_oihw, stride, pad, bias, residual, pre, relu):
    import torch.nn.functional as F

    x = torch.as_tensor(x_nhwc).permute(0, 3, 1, 2).double()
    if pre is not None:
        x = F.relu(x * torch.as_tensor(pre[0]).double().view(1, -1, 1, 1) + torch.as_tensor(pre[1]).double().view(1, -1, 1, 1))
    y = F.conv2d(x, torch.as_tensor(w_oihw).double(), stride=stride, padding=pad)
    if bias is not None:
        y = y + torch.as_tensor(bias).double().view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + torch.as_tensor(residual).double()
    if relu:
        y = F.relu(y)
    return y.numpy()


@pytest.mark.parametrize("case", [
    dict(n=2, h=60, w=80, cin=64, cout=64, k=3, s=1, p=1, bias=True, res=True, pre=False, relu=True),
    dict(n=3, h=60, w=80, cin=64, cout=128, k=3, s=2, p=1, bias=True, res=False, pre=True, relu=True),    # 3x3 / stride 2, one 64-channel chunk
    dict(n=2, h=30, w=40, cin=128, cout=256, k=3, s=2, p=1, bias=True, res=False, pre=False, relu=True),  # two 64-channel chunks
    dict(n=3, h=31, w=45, cin=128, cout=128, k=3, s=2, p=1, bias=False, res=True, pre=False, relu=False), # odd size, residual
    dict(n=1, h=15, w=20, cin=256, cout=512, k=3, s=2, p=1, bias=True, res=False, pre=True, relu=True),   # few tiles: K slices
    dict(n=2, h=30, w=40, cin=128, cout=256, k=1, s=2, p=0, bias=False, res=False, pre=True, relu=False),
    dict(n=1, h=15, w=20, cin=256, cout=512, k=3, s=2, p=1, bias=False, res=True, pre=False, relu=False),
    dict(n=5, h=9, w=7, cin=32, cout=64, k=3, s=1, p=1, bias=True, res=False, pre=False, relu=False),  # ragged M
    dict(n=2, h=31, w=23, cin=8, cout=64, k=2, s=1, p=0, bias=False, res=False, pre=False, relu=False),
])
def test_conv2d_vs_fp64(dev, case):
    _check_conv(dev, case)


# the kernel families behind 3x3 / stride-1 layers must agree with fp64: Winograd F(2x2,3x3)
# ("winograd" and its one-wave-per-SIMD schedule), the split-fp16 kernel ("split": three fp16 MFMAs per fp32
# product), the patch-staged direct kernel ("direct") and the generic implicit GEMM ("igemm"); "auto" is the
# per-layer choice of the planner
@pytest.mark.parametrize("algo", ["auto", "winograd", "winograd-1wave", "split", "direct", "igemm"])
@pytest.mark.parametrize("case", [
    dict(n=3, h=30, w=40, cin=128, cout=128, k=3, s=1, p=1, bias=True, res=True, pre=True, relu=True),
    dict(n=7, h=15, w=20, cin=256, cout=256, k=3, s=1, p=1, bias=False, res=True, pre=True, relu=False),  # odd H
    dict(n=9, h=8, w=10, cin=512, cout=512, k=3, s=1, p=1, bias=True, res=False, pre=False, relu=True),    # tiles span images
    dict(n=5, h=9, w=7, cin=32, cout=64, k=3, s=1, p=1, bias=True, res=True, pre=True, relu=False),        # odd H and W, ragged
    dict(n=1, h=2, w=2, cin=32, cout=64, k=3, s=1, p=1, bias=False, res=False, pre=False, relu=False),     # a single tile
    dict(n=130, h=8, w=10, cin=64, cout=64, k=3, s=1, p=1, bias=True, res=True, pre=False, relu=True),     # > 1 item per block
    dict(n=2, h=60, w=80, cin=64, cout=64, k=3, s=1, p=1, bias=True, res=True, pre=True, relu=True),       # 640-pixel staged ranges
    dict(n=3, h=20, w=96, cin=64, cout=32, k=3, s=1, p=1, bias=False, res=False, pre=True, relu=False),    # 768: single register set
    dict(n=128, h=8, w=10, cin=64, cout=256, k=3, s=1, p=1, bias=True, res=True, pre=True, relu=True),     # 1.25 rounds: half-item tail launch
    dict(n=40, h=15, w=20, cin=256, cout=256, k=3, s=1, p=1, bias=True, res=True, pre=True, relu=False),   # split kernel: K-sliced tail tiles
    dict(n=3, h=60, w=80, cin=64, cout=128, k=3, s=2, p=1, bias=True, res=False, pre=True, relu=True),     # stride 2 (space-to-depth walk)
    dict(n=5, h=15, w=20, cin=128, cout=256, k=3, s=2, p=1, bias=False, res=True, pre=True, relu=False),   # stride 2, odd H
    dict(n=7, h=9, w=7, cin=64, cout=128, k=3, s=2, p=1, bias=True, res=False, pre=False, relu=True),      # stride 2, odd H and W, ragged
])
def test_conv3x3_kernel_families(dev, case, algo):
    from happypose_amd import ops

    ops.select_conv_algo(algo)
    try:
        # Winograd's transforms cost about one extra bit of round-off
        _check_conv(dev, case, tol=4e-5 if algo in ("auto", "winograd", "winograd-1wave") else 2e-5)
    finally:
        ops.select_conv_algo("auto")


# what the MBConv blocks of EfficientNet-b3 need from the generic kernel: channel counts that are not
# multiples of 64 (weights padded to whole tiles, stores not), K = Cin not a multiple of 32, swish,
# the squeeze-excitation gate on the input of the projection
@pytest.mark.parametrize("case", [
    dict(n=2, h=30, w=40, cin=8, cout=40, k=3, s=2, p=1, bias=True, res=False, act=2, se=False),     # stem-like
    dict(n=3, h=30, w=40, cin=24, cout=144, k=1, s=1, p=0, bias=True, res=False, act=2, se=False),   # expand
    dict(n=3, h=15, w=20, cin=144, cout=24, k=1, s=1, p=0, bias=True, res=True, act=0, se=True),     # project + skip
    dict(n=2, h=7, w=10, cin=232, cout=136, k=1, s=1, p=0, bias=False, res=False, act=0, se=True),
    dict(n=2, h=7, w=10, cin=384, cout=1536, k=1, s=1, p=0, bias=True, res=False, act=2, se=False),  # head
])
def test_conv2d_mbconv_features(dev, case):
    from happypose_amd import ops

    rs = np.random.RandomState(11)
    c = case
    x = rs.normal(size=(c["n"], c["h"], c["w"], c["cin"])).astype(np.float32)
    w = (rs.normal(size=(c["cout"], c["cin"], c["k"], c["k"])) / np.sqrt(c["cin"] * c["k"] ** 2)).astype(np.float32)
    ho, wo = (c["h"] + 2 * c["p"] - c["k"]) // c["s"] + 1, (c["w"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    bias = rs.normal(size=c["cout"]).astype(np.float32) if c["bias"] else None
    res = rs.normal(size=(c["n"], ho, wo, c["cout"])).astype(np.float32) if c["res"] else None
    gate = rs.uniform(0.1, 1.0, size=(c["n"], c["cin"])).astype(np.float32) if c["se"] else None
    xin = x.astype(np.float64) * (gate[:, None, None, :] if gate is not None else 1.0)
    ref = _conv_ref(xin, w, c["s"], c["p"], bias, res, None, False)
    if c["act"] == 2:
        ref = ref / (1.0 + np.exp(-ref))
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    y = ops.conv2d_nhwc(t(x), t(np.ascontiguousarray(w.transpose(0, 2, 3, 1))), c["s"], c["p"], t(bias), t(res), t(gate), None, c["act"])
    assert y.shape == (c["n"], ho, wo, c["cout"])
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err


def _check_conv(dev, case, tol=2e-5):
    from happypose_amd import ops

    rs = np.random.RandomState(3)
    c = case
    x = rs.normal(size=(c["n"], c["h"], c["w"], c["cin"])).astype(np.float32)
    w = (rs.normal(size=(c["cout"], c["cin"], c["k"], c["k"])) / np.sqrt(c["cin"] * c["k"] ** 2)).astype(np.float32)
    ho, wo = (c["h"] + 2 * c["p"] - c["k"]) // c["s"] + 1, (c["w"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    bias = rs.normal(size=c["cout"]).astype(np.float32) if c["bias"] else None
    res = rs.normal(size=(c["n"], ho, wo, c["cout"])).astype(np.float32) if c["res"] else None
    pre = (rs.uniform(0.5, 1.5, c["cin"]).astype(np.float32), rs.normal(size=c["cin"]).astype(np.float32)) if c["pre"] else None
    ref = _conv_ref(x, w, c["s"], c["p"], bias, res, pre, c["relu"])
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    wp = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    y = ops.conv2d_nhwc(t(x), t(wp), c["s"], c["p"], t(bias), t(res), t(pre[0]) if pre else None,
                        t(pre[1]) if pre else None, c["relu"])
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err <= tol * max(1.0, np.abs(ref).max()), err


# The split-fp16 kernels keep their bound over the dynamic range a network sees: activations spread over five
# decades (low halves down in the fp16 subnormals -- the f16 MFMA must not flush them) and folded-BN weights whose
# magnitude differs by 1e6 between output channels (per-channel power-of-two scaling).
@pytest.mark.parametrize("case", [
    dict(n=4, h=30, w=40, cin=128, cout=128, k=3, s=1),
    dict(n=4, h=30, w=40, cin=64, cout=128, k=3, s=2),
    dict(n=2, h=64, w=64, cin=8, cout=64, k=5, s=2),   # generic split kernel (stem-like)
])
def test_split_conv_dynamic_range(dev, case):
    from happypose_amd import ops

    rs = np.random.RandomState(5)
    c = case
    x = (rs.normal(size=(c["n"], c["h"], c["w"], c["cin"])) * 10.0 ** rs.uniform(-5, 0, size=(c["n"], c["h"], c["w"], 1))).astype(np.float32)
    w = (rs.normal(size=(c["cout"], c["cin"], c["k"], c["k"])) * 10.0 ** rs.uniform(-6, 0, size=(c["cout"], 1, 1, 1))).astype(np.float32)
    ref = _conv_ref(x, w, c["s"], c["k"] // 2, None, None, None, False)
    ops.select_conv_algo("split")
    try:
        y = ops.conv2d_nhwc(torch.as_tensor(x, device=dev), torch.as_tensor(np.ascontiguousarray(w.transpose(0, 2, 3, 1)), device=dev),
                            c["s"], c["k"] // 2).cpu().numpy()
    finally:
        ops.select_conv_algo("auto")
    # per output channel: 2e-5 of that channel's largest output (a global bound would hide the small-weight channels)
    err = np.abs(y - ref).reshape(-1, c["cout"]).max(0)
    scale = np.abs(ref).reshape(-1, c["cout"]).max(0)
    assert (err <= 2e-5 * scale).all(), (err / scale).max()


# fp16 kernel (configuration C5): operands rounded to fp16, fp32 accumulation, one rounding of the
# result -> compare with fp64 on the SAME fp16-rounded operands; the only error left is the
# accumulation order and the final rounding to fp16 (2^-11 relative)
@pytest.mark.parametrize("case", [
    dict(n=3, h=60, w=80, cin=64, cout=64, k=3, s=1, p=1, bias=True, res=True, pre=False, relu=True),
    dict(n=3, h=60, w=80, cin=64, cout=128, k=3, s=2, p=1, bias=True, res=False, pre=True, relu=True),    # 3x3 / stride 2, one 64-channel chunk
    dict(n=2, h=30, w=40, cin=128, cout=256, k=3, s=2, p=1, bias=True, res=False, pre=False, relu=True),  # two 64-channel chunks
    dict(n=3, h=31, w=45, cin=128, cout=128, k=3, s=2, p=1, bias=False, res=True, pre=False, relu=False), # odd size, residual
    dict(n=1, h=15, w=20, cin=256, cout=512, k=3, s=2, p=1, bias=True, res=False, pre=True, relu=True),   # few tiles: K slices
    dict(n=2, h=30, w=40, cin=128, cout=256, k=1, s=2, p=0, bias=False, res=False, pre=True, relu=False),
    dict(n=2, h=15, w=20, cin=256, cout=512, k=3, s=1, p=1, bias=False, res=True, pre=False, relu=False),
    dict(n=5, h=9, w=7, cin=64, cout=64, k=3, s=1, p=1, bias=True, res=False, pre=False, relu=False),   # ragged M
    dict(n=2, h=31, w=23, cin=16, cout=64, k=2, s=1, p=0, bias=False, res=False, pre=False, relu=False),  # one K-tile
    dict(n=1, h=24, w=32, cin=8, cout=64, k=4, s=2, p=1, bias=True, res=False, pre=False, relu=True),    # 2 K-tiles, odd taps
])
def test_conv2d_f16_vs_fp64(dev, case):
    from happypose_amd import ops

    rs = np.random.RandomState(5)
    c = case
    h16 = lambda a: a.astype(np.float16)  # noqa: E731
    x = h16(rs.normal(size=(c["n"], c["h"], c["w"], c["cin"])))
    w = h16(rs.normal(size=(c["cout"], c["cin"], c["k"], c["k"])) / np.sqrt(c["cin"] * c["k"] ** 2))
    ho, wo = (c["h"] + 2 * c["p"] - c["k"]) // c["s"] + 1, (c["w"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    bias = rs.normal(size=c["cout"]).astype(np.float32) if c["bias"] else None
    res = h16(rs.normal(size=(c["n"], ho, wo, c["cout"]))) if c["res"] else None
    pre = (h16(rs.uniform(0.5, 1.5, c["cin"])), h16(rs.normal(size=c["cin"]))) if c["pre"] else None
    xin = x.astype(np.float64)
    if pre is not None:  # the prologue itself is evaluated in fp16 (fma, relu): restate that rounding
        xin = np.maximum((x.astype(np.float32) * pre[0].astype(np.float32) + pre[1].astype(np.float32)).astype(np.float16), 0).astype(np.float64)
    ref = _conv_ref(xin, w.astype(np.float64), c["s"], c["p"], bias, None if res is None else res.astype(np.float64), None, c["relu"])
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    wp = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    y = ops.conv2d_nhwc_f16(t(x), t(wp), c["s"], c["p"], t(bias), t(res), t(pre[0]) if pre else None,
                            t(pre[1]) if pre else None, c["relu"])
    err = np.abs(y.float().cpu().numpy() - ref).max()
    assert err <= 1.5e-3 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("arch,cin", [("vanilla_resnet34", 9), ("resnet34", 6)])
def test_backbone_f16_vs_f32(dev, arch, cin):
    """The fp16 plan of a whole backbone against the fp32 plan on the same weights and input:
    features agree to the fp16 tolerance stated in DESIGN.md (2e-2 of the feature scale)."""
    from happypose_amd import ops
    from happypose_amd.models import pose_model_param_shapes
    from happypose_amd.synthetic import predictor_weights

    w = predictor_weights(pose_model_param_shapes(arch, cin, pose_dim=9, n_views_logits=1), seed=4)
    x = np.random.RandomState(7).uniform(-1, 1, size=(3, 240, 320, cin)).astype(np.float32)
    outs = {}
    for prec in ("f32", "f16"):
        net = ops.Net(arch, cin, w, max_batch=2, device=dev, precision=prec)  # 3 samples -> chunks 2 + 1
        xin = net.new_input(3)
        xin[..., :cin] = torch.as_tensor(x, device=dev)
        pose, logits, feats = net.forward(xin, want_pose=True, want_logits=True, want_features=True)
        outs[prec] = [t.cpu().numpy() for t in (pose, logits, feats)]
        if prec == "f16":
            # the fp16 plan fed with fp32 input (conversion pass inside hp_net_forward) gives the same
            # bits as the fp16 input written directly (hp_net_forward_f16in)
            assert xin.dtype == torch.float16
            x32 = torch.zeros((3, 240, 320, net.c_pad), device=dev)
            x32[..., :cin] = torch.as_tensor(x, device=dev)
            for a, b_ in zip(net.forward(x32, want_pose=True, want_logits=True, want_features=True), (pose, logits, feats)):
                assert torch.equal(a, b_)
    scale = np.abs(outs["f32"][2]).max()
    assert np.abs(outs["f16"][2] - outs["f32"][2]).max() <= 2e-2 * scale
    np.testing.assert_allclose(outs["f16"][0], outs["f32"][0], atol=2e-2 * max(1.0, np.abs(outs["f32"][0]).max()))
    np.testing.assert_allclose(outs["f16"][1], outs["f32"][1], atol=2e-2 * max(1.0, np.abs(outs["f32"][1]).max()))


@pytest.mark.parametrize("case,bound", [("stem7f16", 2e-3), ("stem5", 0.0), ("shortcut", 2e-5), ("shortcut_mp", 2e-5)])
def test_persistent_stem_kernels_vs_tile_kernels(dev, case, bound):
    """The persistent two-group stems (conv_stem7x7s2_pool_f16_pp: fp16 plan, <= 9 real channels, the MegaPose coarse model;
    conv_stem5x5s2_pool_split_pp: the CosyPose fp32 stem) against the tile kernels they replace (HP_STEM7_F16_OLD=1 /
    HP_STEM5_OLD=1), through the whole backbone: 240 x 320 and three sizes whose pooled maps are not multiples of the 3 x 16 tile;
    fp16: the pad channels of the 16-channel record filled with garbage.  The fp16 kernels sum in different orders: fp16
    rounding noise, 2e-3 of the feature scale (measured 6e-4); the fp32 kernel issues the same MFMAs in the same order and
    its epilogue is monotone: identical bits.  shortcut / shortcut_mp: the down-sampling blocks' 1 x 1 / stride-2 shortcuts as
    extra work items of the block's 3 x 3 / stride-2 launch (conv3x3s2_pp, ConvArgs::sc_w) against launches of their own
    (HP_NET_NO_SHORTCUT_FUSION=1): another kernel, another summation order, fp32 round-off (2e-5 of the feature scale).  The
    switches are read once per process: each side runs in its own interpreter (tools/stem_ab.py)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stem_ab.py")
    out = subprocess.run([sys.executable, script, case], capture_output=True, text=True, timeout=600)
    worst = float(out.stdout.strip().splitlines()[-1].split()[-1]) if out.stdout.strip() else float("nan")
    assert out.returncode == 0 and worst <= bound, out.stdout + out.stderr


@pytest.mark.parametrize("arch,cin,tag", [("vanilla_resnet34", 27, "vanilla_resnet34_27"),
                                          ("vanilla_resnet34", 9, "vanilla_resnet34_9"),
                                          ("vanilla_resnet34", 32, "vanilla_resnet34_32"),  # the C3 (RGB-D) stem
                                          ("resnet34", 6, "resnet34_6"), ("resnet18", 6, "resnet18_6"),
                                          ("resnet34", 6, "resnet34cp_6")])  # CosyPose's copy of the module
def test_backbone_golden_g6(dev, golden_dir, arch, cin, tag):
    """Whole backbone (BN folded, NHWC, MFMA) against the reference modules' outputs."""
    from happypose_amd import ops
    from happypose_amd.synthetic import named_weights
    from oracle import backbones as ob

    g = np.load(golden_dir / "g6_backbones.npz")
    shapes = ob.param_shapes(arch, cin)
    w = {f"backbone.{k}": v for k, v in named_weights(shapes, seed=0).items()}
    net = ops.Net(arch, cin, w, max_batch=2, device=dev)
    x = np.random.RandomState(100 + cin).uniform(-1, 1, size=(2, cin, 240, 320)).astype(np.float32)
    xin = net.new_input(2)
    xin[..., :cin] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    _, _, feats = net.forward(xin, want_pose=False, want_features=True)
    ref = g[tag + "/out"]
    if ref.ndim == 4:
        ref = ref.reshape(2, 512, -1).mean(-1)
    err = np.abs(feats.cpu().numpy() - ref).max()
    assert err <= 2e-4 * np.abs(ref).max(), (err, np.abs(ref).max())
    assert abs(net.flops_per_sample / 1e9 - {("vanilla_resnet34", 27): 14.236, ("vanilla_resnet34", 9): 12.068,
                                              ("vanilla_resnet34", 32): 14.838,
                                              ("resnet34", 6): 11.352, ("resnet18", 6): 5.64}[(arch, cin)]) < 0.02


def test_net_heads_and_batch_chunking(dev):
    from happypose_amd import ops
    from happypose_amd.synthetic import predictor_weights
    from oracle import backbones as ob

    shapes = ob.predictor_param_shapes("resnet18", 6, pose_dim=9, n_views_logits=1)
    w = predictor_weights(shapes, seed=3, update_scale=0.05)
    net = ops.Net("resnet18", 6, w, max_batch=2, device=dev)  # batch 5 -> chunks 2+2+1
    x = np.random.RandomState(1).uniform(0, 1, size=(5, 6, 240, 320)).astype(np.float32)
    xin = net.new_input(5)
    xin[..., :6] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    pose, logits, _ = net.forward(xin, want_pose=True, want_logits=True)
    with torch.no_grad():
        ref = ob.net_forward(torch.as_tensor(x), w, "resnet18", heads=("pose", "renderings_logits"))
    np.testing.assert_allclose(pose.cpu().numpy(), ref["pose"].numpy(), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), ref["renderings_logits"].numpy(), rtol=1e-3, atol=2e-4)


def test_efficientnet_b3_backbone_golden_g9(dev, golden_dir):
    """EfficientNet-b3 (the backbone of the released CosyPose checkpoints) against the
    reference module's own output (G9), and the pose head against the CPU restatement."""
    from happypose_amd import ops
    from happypose_amd.models import pose_model_param_shapes
    from happypose_amd.synthetic import named_weights
    from oracle import backbones as ob

    g = np.load(golden_dir / "g9_efficientnet.npz")
    # the golden weights are keyed by the backbone's own parameter names (no "backbone." prefix)
    w = {f"backbone.{k}": v for k, v in named_weights(ob.efficientnet_b3_param_shapes(6), seed=0).items()}
    w.update(named_weights({"pose_fc.weight": (9, 1536), "pose_fc.bias": (9,)}, seed=0))
    assert list(w) == list(pose_model_param_shapes("efficientnet-b3", 6, pose_dim=9))
    net = ops.Net("efficientnet-b3", 6, w, max_batch=2, device=dev)
    assert net.n_features == 1536 and abs(net.flops_per_sample / 1e9 - 2.91) < 0.1
    x = np.random.RandomState(106).uniform(-1, 1, size=(2, 6, 240, 320)).astype(np.float32)
    xin = net.new_input(2)
    xin[..., :6] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    pose, _, feats = net.forward(xin, want_pose=True, want_features=True)
    ref = g["out_mean"]  # spatial mean of the [2,1536,7,10] feature map
    err = np.abs(feats.cpu().numpy() - ref).max()
    assert err <= 2e-4 * np.abs(ref).max(), (err, np.abs(ref).max())
    with torch.no_grad():
        rp = ob.net_forward(torch.as_tensor(x), w, "efficientnet-b3", heads=("pose",))["pose"].numpy()
    np.testing.assert_allclose(pose.cpu().numpy(), rp, rtol=1e-3, atol=2e-4)
    # batch chunking (3 samples through max_batch 2) gives the same rows
    x3 = np.concatenate([x, x[:1]], 0)
    xin3 = net.new_input(3)
    xin3[..., :6] = torch.as_tensor(x3, device=dev).permute(0, 2, 3, 1)
    _, _, f3 = net.forward(xin3, want_pose=False, want_features=True)
    assert torch.equal(f3[:2], feats) and torch.equal(f3[2], feats[0])


# ------------------------------------------------------------------- dispatcher registration
def test_torch_ops_match_ctypes_path(dev, scene_store):
    """``torch.ops.happypose_amd.*`` (the compiled TORCH_LIBRARY, ``csrc/torch_library.cpp``) are the same C-ABI calls behind
    dispatcher schemas: bit-identical results."""
    from happypose_amd import ops, torch_ops
    from happypose_amd.synthetic import predictor_weights
    from oracle import backbones as ob

    o = torch.ops.happypose_amd
    rs = np.random.RandomState(0)
    img = torch.as_tensor(rs.rand(2, 3, 120, 160).astype(np.float32), device=dev)
    boxes = torch.as_tensor(np.array([[10.3, 8.2, 120.7, 100.1], [-5, -4, 90, 70.5], [0, 0, 160, 120]], np.float32), device=dev)
    ids = torch.as_tensor(np.array([0, 1, 1], np.int32), device=dev)
    assert torch.equal(o.crop_roi_align(img, boxes, ids, 60, 80), ops.crop_roi_align(img, boxes, ids, (60, 80)))

    T = torch.as_tensor(_poses(3, seed=2), device=dev)
    K = torch.as_tensor(np.tile(np.array([[150.0, 0, 80], [0, 150.0, 60], [0, 0, 1]], np.float32), (3, 1, 1)), device=dev)
    obj = torch.as_tensor(np.array([0, 1, 2], np.int32), device=dev)
    st = torch_ops.ticket(scene_store)
    got = o.rasterize(st, obj, T, K, 120, 160, True, True)
    ref = ops.rasterize(scene_store, obj, T, K, (120, 160), render_normals=True, render_depth=True)
    assert len(got) == 3 and all(torch.equal(a, b) for a, b in zip(got, ref[:3]))

    prep = torch_ops.pose_prep(scene_store, T, K, torch.arange(3, dtype=torch.int32, device=dev), obj, (120, 160), (60, 80), "TCO+front_3views", True)
    direct = o.pose_prep(st, T, K, torch.arange(3, dtype=torch.int32, device=dev), obj, scene_store.point_ids(2000), scene_store.point_ids(200),
                         120, 160, 60, 80, "TCO+front_3views", True)
    assert all(torch.equal(a, b) for a, b in zip(prep, direct))
    ref = ops.pose_prep(scene_store, T, K, torch.arange(3, dtype=torch.int32), obj, (120, 160), (60, 80), "TCO+front_3views", True)
    for a, k in zip(prep, ("TCO", "tCR", "TCV_O", "boxes_rend", "boxes_crop", "K_crop")):
        assert torch.equal(a, ref[k]), k
    pose9 = torch.as_tensor(rs.randn(3, 9).astype(np.float32) * 0.1, device=dev)
    assert torch.equal(o.pose_update(prep[0], prep[5][:, 0].contiguous(), pose9, prep[1]),
                       ops.pose_update(prep[0], prep[5][:, 0].contiguous(), pose9, prep[1]))

    w = predictor_weights(ob.predictor_param_shapes("resnet18", 6, pose_dim=9), seed=3, update_scale=0.05)
    net = ops.Net("resnet18", 6, w, max_batch=2, device=dev)
    xin = net.new_input(2)
    xin[..., :6] = torch.as_tensor(rs.rand(2, 240, 320, 6).astype(np.float32), device=dev)
    (pose,) = o.net_forward(torch_ops.ticket(net), xin)
    assert torch.equal(pose, net.forward(xin)[0])
    # the op checks H and W against the planned map (hp_net_input_dims): a smaller tensor raises, on the device and on Meta
    for bad in (xin[:, :200].contiguous(), xin[:, :, :300].contiguous(), torch.empty((2, 240, 320, 4), device=dev)):
        with pytest.raises(RuntimeError, match="net_forward"):
            o.net_forward(torch_ops.ticket(net), bad)
        with pytest.raises(RuntimeError, match="net_forward"):
            o.net_forward(torch_ops.ticket(net), torch.empty(bad.shape, device="meta"))
    assert o.net_forward(torch_ops.ticket(net), torch.empty(xin.shape, device="meta"))[0].shape == pose.shape

    x = torch.as_tensor(rs.randn(2, 12, 16, 32).astype(np.float32), device=dev)
    wt = torch.as_tensor(rs.randn(64, 3, 3, 32).astype(np.float32) * 0.05, device=dev)
    assert torch.equal(o.conv2d_nhwc(x, wt, 1, 1, act=1), ops.conv2d_nhwc(x, wt, 1, 1, relu=True))
    got = o.rasterize(st, obj, T, K, 120, 160, False, True, True, True)  # the reference renderer's state
    ref = ops.rasterize(scene_store, obj, T, K, (120, 160), render_depth=True, msaa=True, aniso=True)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[2])
    # the ops run on torch's CURRENT stream
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        y_side = o.conv2d_nhwc(x, wt, 1, 1, act=1)
    side.synchronize()
    assert torch.equal(y_side, ops.conv2d_nhwc(x, wt, 1, 1, relu=True))
    del net
    import gc
    gc.collect()
    with pytest.raises(ValueError):
        o.net_forward(10 ** 9, xin)  # not a live ticket
    with pytest.raises(NotImplementedError):
        o.pose_update(prep[0].cpu(), prep[5][:, 0].cpu().contiguous(), pose9.cpu(), prep[1].cpu())  # no CPU kernels


@pytest.mark.parametrize("case", ["cosypose6", "megapose32", "coarse_f16", "permuted32", "msaa_aniso6"])
def test_render_inputs_vs_oracle(dev, scene_store, case):
    """``hp_render_inputs``: crop + V rendered views into the pixel records of the network input in one launch, against the
    oracle's roi_align (C) and rasteriser (C) run separately: the reference's channel order for CosyPose (dense 6-float
    records) and MegaPose RGB-D (32 floats, 4 views x 7 channels, depth normalised), the 16-half records of an fp16
    plan, a permuted layout (view v = one 32-B sector: its 7 channels + crop channel v), and the reference's render state."""
    from happypose_amd import ops
    from oracle import native

    rs = np.random.RandomState(3)
    b = 5
    H, W = 480, 640
    rgbd = case in ("megapose32", "permuted32")
    Ct = 4 if rgbd else 3
    frames = rs.uniform(0, 1, size=(2, Ct, H, W)).astype(np.float32)
    if rgbd:
        frames[:, 3] = rs.uniform(0.3, 0.9, size=(2, H, W)).astype(np.float32)
        frames[:, 3, 100:140, 200:260] = 0.0  # holes: the validity rule
    im_ids = np.array([0, 1, 1, 0, 1], np.int32)
    boxes = np.array([[100, 80, 400, 305], [50.5, 20.25, 600.0, 432.5], [300, 200, 340, 230], [-40, -30, 280, 210], [0, 0, 640, 480]], np.float32)
    V = 4 if rgbd else 1
    T = _poses(b * V, 17).reshape(b, V, 4, 4)
    K = np.tile(np.array([[700.0, 0, 160], [0, 700.0, 120], [0, 0, 1]], np.float32), (b, V, 1, 1))
    obj = np.array([0, 1, 2, 1, 0], np.int32)
    z = rs.uniform(0.4, 0.8, size=b).astype(np.float32)
    nrm, dep = case in ("megapose32", "permuted32", "coarse_f16"), rgbd
    c_r = 3 + 3 * nrm + dep
    msaa = aniso = case == "msaa_aniso6"
    mode = 2 if rgbd else 0
    kw = dict(images=torch.as_tensor(frames, device=dev), boxes=torch.as_tensor(boxes, device=dev), im_ids=torch.as_tensor(im_ids),
              n_img_channels=Ct, depth_norm_z=torch.as_tensor(z, device=dev) if mode else None, depth_norm_mode=mode, msaa=msaa, aniso=aniso)
    if case == "coarse_f16":
        x = torch.zeros((b, 240, 320, 16), dtype=torch.float16, device=dev)
    elif case == "permuted32":
        x = torch.full((b, 240, 320, 32), 0.0, device=dev)
        kw["layout"] = ([8 * v for v in range(4)], [8 * v + 7 for v in range(4)], [0, 1, 2, 3], [1, 1, 1, 1])
    else:
        x = torch.zeros((b, 240, 320, 6 if not rgbd else 32), device=dev)
    ops.render_inputs(scene_store, x, torch.as_tensor(obj), torch.as_tensor(T), torch.as_tensor(K), nrm, dep, **kw)
    got = x.float().cpu().numpy()
    crop = native.crop_images(frames, boxes, im_ids, (240, 320))
    if rgbd:
        crop[:, 3] = np.clip(crop[:, 3] / z[:, None, None], 0, 2) - 1
    r = native.rasterize(scene_store.packed, np.repeat(obj, V), T.reshape(-1, 4, 4), K.reshape(-1, 3, 3), (240, 320), nrm, dep, msaa=msaa, aniso=aniso)
    parts = [r["rgbs"]] + ([r["normals"]] if nrm else []) + ([np.clip(r["depths"] / np.repeat(z, V)[:, None, None, None], 0, 2) - 1] if dep else [])
    rend = np.concatenate(parts, 1).reshape(b, V, c_r, 240, 320)
    if case == "permuted32":
        want = np.zeros((b, 32, 240, 320), np.float32)
        for v in range(4):
            want[:, 8 * v:8 * v + 7] = rend[:, v]
            want[:, 8 * v + 7] = crop[:, v]
    else:
        want = np.concatenate([crop, rend.reshape(b, V * c_r, 240, 320)], 1)
    n_ch = want.shape[1]
    got = got.transpose(0, 3, 1, 2)
    assert np.all(got[:, n_ch:] == 0)  # pads untouched
    tol = 2e-3 if case == "coarse_f16" else 0.0
    crop_ch = [8 * v + 7 for v in range(4)] if case == "permuted32" else list(range(Ct))
    colour_ch = crop_ch[:3]
    np.testing.assert_allclose(got[:, colour_ch], want[:, colour_ch], rtol=1e-5 + tol, atol=2e-6 + tol)
    if rgbd:  # the depth channel: the validity rule (mask >= 0.99) may flip on a few hole-border pixels
        dd = np.abs(got[:, crop_ch[3]] - want[:, crop_ch[3]])
        assert (dd > 1e-5).mean() < 1e-3 and np.median(dd) < 1e-6
    rend_ch = [c for c in range(n_ch) if c not in crop_ch]
    d = np.abs(got[:, rend_ch] - want[:, rend_ch])
    assert (d > 1.01 / 255 + tol).mean() < (3e-3 if msaa else 1e-3), (d > 1.01 / 255 + tol).mean()  # silhouette pixels only
    assert np.median(d) <= tol


def test_split_fp16_small_activations_keep_their_bits(dev):
    """The low side of the split-fp16 scheme.  x = x_hi + x_lo in fp16 has an absolute floor of 2^-25, so
    a layer whose activations sit at ~1e-4 loses relative bits unless they are scaled into fp16's normal range first.
    A WideResNet-18 with "trained-network" statistics -- the BN in front of six 3x3 convs shrinks their input to
    ~1e-4 .. 1e-3 (gamma, beta x 1e-4), the conv weights undo it (x 1e4) -- against the same network in float64: with the
    dynamic activation scale (default) every feature agrees to 2e-5 x max|ref| like any other network here; with
    ``hp_net_set_act_scale(net, 0)`` the very same kernels are ~100x worse."""
    from happypose_amd import ops
    from happypose_amd.synthetic import predictor_weights
    from oracle import backbones as ob

    w = predictor_weights(ob.predictor_param_shapes("resnet18", 6), seed=4)
    for blk in ("layer1.1", "layer2.0", "layer2.1", "layer3.1", "layer4.0", "layer4.1"):
        w[f"backbone.{blk}.bn2.weight"] = (w[f"backbone.{blk}.bn2.weight"] * 1e-4).astype(np.float32)
        w[f"backbone.{blk}.bn2.bias"] = (w[f"backbone.{blk}.bn2.bias"] * 1e-4).astype(np.float32)
        w[f"backbone.{blk}.conv2.weight"] = (w[f"backbone.{blk}.conv2.weight"] * 1e4).astype(np.float32)
    x = np.random.RandomState(2).uniform(0, 1, size=(4, 6, 240, 320)).astype(np.float32)
    with torch.no_grad():
        ref = ob.net_forward(torch.as_tensor(x).double(), {k: torch.as_tensor(np.asarray(v)).double() if np.asarray(v).dtype.kind == "f" else
                                                           torch.as_tensor(np.asarray(v)) for k, v in w.items()}, "resnet18", heads=("features",))["features"].numpy()
    net = ops.Net("resnet18", 6, w, max_batch=4, device=dev)
    xin = net.new_input(4)
    xin[..., :6] = torch.as_tensor(x, device=dev).permute(0, 2, 3, 1)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    errs = {}
    for on in (True, False):
        net.set_act_scale(on)
        f = net.forward(xin, want_pose=False, want_features=True)[2].cpu().numpy()
        assert net.status() == 0
        errs[on] = float((np.abs(f - ref) / scale).max())
    assert errs[True] <= 2e-5, errs
    assert errs[False] > 10 * errs[True], errs  # the floor the scale removes is real
    # and on an ordinary network the scale changes nothing beyond round-off
    w0 = predictor_weights(ob.predictor_param_shapes("resnet18", 6), seed=4)
    n0 = ops.Net("resnet18", 6, w0, max_batch=4, device=dev)
    fa = n0.forward(xin, want_pose=False, want_features=True)[2]
    n0.set_act_scale(False)
    fb = n0.forward(xin, want_pose=False, want_features=True)[2]
    assert float((fa - fb).abs().max() / fa.abs().max()) < 5e-6
