"""Renders the calibration views with the REFERENCE's renderer.  Runs where happypose + Panda3D are installed (NOT in this
repository's containers, which have neither): it is the data-collection half of tools/calibrate_renderer.py.

    python panda3d_render_calibration_views.py --mesh tests/data/obj_000001.ply [--mesh-units mm] --out panda3d_views.npz

The scene is the reference's own renderer test (tests/test_batch_renderer_panda3d.py:43-69: TWO = quat_xyzw(0.5, 0.5, -0.5,
0.5), t = (0, 0, 0.3), K = [[300, 0, 320], [0, 300, 240], [0, 0, 1]], 480 x 640, ambient light (1, 1, 1)) plus views that
exercise what the test does not: oblique and distant poses (minified, anisotropic texture footprints) and a 240 x 320
crop-like camera.  A textured asset is needed for the texture-filter rule -- put the model's texture (BOP:
``obj_000001.png``) beside the .ply; the file the reference's test ships has no texture and only calibrates the sample
pattern and the normal map.

Output (NPZ): rgb / normals uint8 [N,H,W,3] exactly as ``Panda3dBatchRenderer.render`` returns them (x 255, rounded),
depth float32 [N,H,W], TCO [N,4,4], K [N,3,3], resolution, mesh file name, mesh_units.
"""
import argparse
from pathlib import Path

import numpy as np


def calibration_views(n_extra: int = 11, seed: int = 0):
    """(TCO [N,4,4], K [N,3,3], resolutions) -- deterministic; tools/calibrate_renderer.py regenerates the same list."""
    def quat_xyzw(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    T0 = np.eye(4)
    T0[:3, :3] = quat_xyzw((0.5, 0.5, -0.5, 0.5))
    T0[:3, 3] = (0, 0, 0.3)
    K0 = np.array([[300.0, 0, 320], [0, 300.0, 240], [0, 0, 1]])
    Ts, Ks = [T0], [K0]
    rs = np.random.RandomState(seed)
    for i in range(n_extra):
        a = rs.normal(size=3)
        a /= np.linalg.norm(a)
        ang = rs.uniform(0.3, 3.0)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        T = np.eye(4)
        T[:3, :3] = R @ T0[:3, :3]
        T[:3, 3] = (rs.uniform(-0.05, 0.05), rs.uniform(-0.04, 0.04), (0.25, 0.4, 0.7, 1.2)[i % 4])
        Ts.append(T)
        Ks.append(K0 * np.array([[1 + 0.5 * (i % 3), 1, 1], [1, 1 + 0.5 * (i % 3), 1], [1, 1, 1]]))
    return np.stack(Ts).astype(np.float32), np.stack(Ks).astype(np.float32), (480, 640)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", required=True)
    ap.add_argument("--mesh-units", default="mm")
    ap.add_argument("--out", default="panda3d_views.npz")
    args = ap.parse_args()

    import torch
    from happypose.toolbox.datasets.object_dataset import RigidObject, RigidObjectDataset
    from happypose.toolbox.renderer.panda3d_batch_renderer import Panda3dBatchRenderer
    from happypose.toolbox.renderer.types import Panda3dLightData

    label = "calibration_object"
    ds = RigidObjectDataset([RigidObject(label=label, mesh_path=Path(args.mesh), mesh_units=args.mesh_units)])
    renderer = Panda3dBatchRenderer(asset_dataset=ds, n_workers=1, preload_cache=True, split_objects=False)
    TCO, K, res = calibration_views()
    n = len(TCO)
    lights = [Panda3dLightData(light_type="ambient", color=(1.0, 1.0, 1.0, 1.0))]
    out = renderer.render(labels=n * [label], TCO=torch.from_numpy(TCO), K=torch.from_numpy(K), light_datas=n * [lights],
                          resolution=res, render_normals=True, render_depth=True, render_binary_mask=True)
    to_u8 = lambda t: np.round(t.movedim(1, -1).cpu().numpy() * 255.0).astype(np.uint8)  # noqa: E731
    np.savez_compressed(args.out, rgb=to_u8(out.rgbs), normals=to_u8(out.normals), depth=out.depths[:, 0].cpu().numpy().astype(np.float32),
                        TCO=TCO, K=K, resolution=np.asarray(res), mesh=Path(args.mesh).name, mesh_units=args.mesh_units)
    print(f"wrote {args.out}: {n} views of {args.mesh}")


if __name__ == "__main__":
    main()
