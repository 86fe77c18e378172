"""Fits the rasteriser's unpinned conventions (``hp_raster_conventions``) to renders of the reference's Panda3D renderer.

Nothing in this repository's containers can run Panda3D, so three conventions of the reference-state rasteriser (a-6) rest
on a reading of OpenGL: the four multisample positions, the anisotropic filter's probe-count / level-of-detail rule and the
axis / sign map of the eye-normal code (``TB/renderer/panda3d_scene_renderer.py:68-71,221-230``,
``TB/renderer/utils.py:63-79``).  They are a run-time record on both sides (``hp_mesh_store_set_raster_conventions`` for the HIP
rasteriser, ``hp_oracle_set_raster_conventions`` for the CPU oracle), and this tool is how an owner of a Panda3D
installation pins them:

 1. where happypose + Panda3D run:   python tools/panda3d_render_calibration_views.py --mesh <obj_000001.ply> --out views.npz
    (the reference's own renderer-test scene, tests/test_batch_renderer_panda3d.py:43-69, plus oblique / distant views);
 2. anywhere (CPU is enough):        python tools/calibrate_renderer.py views.npz --mesh <obj_000001.ply> [--hip]

It renders the same views with the oracle (``--hip``: with the HIP rasteriser) under every candidate record, scores each
against the Panda3D pixels in 8-bit units on the pixels the convention can move (normal map: covered interior; sample
pattern: the silhouette band; texture rule: covered, textured interior), prints the ranking of every group and writes the
best record as JSON -- ``store.set_raster_conventions(json.load(open(...)))`` (``happypose_amd.ops.MeshStore``) applies it; to make it the
default change ``kDefaultConventions`` (csrc/raster.hip) and ``HP_ORACLE_CONV_DEFAULT`` (oracle.c) and regenerate G10.

``--self-test`` needs no Panda3D: it makes the "Panda3D" views with the oracle under a hidden non-default record and checks
that the fit recovers it.
"""
import argparse
import itertools
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))

DEFAULTS = dict(msaa_x=(0.375, 0.875, 0.125, 0.625), msaa_y=(0.125, 0.375, 0.625, 0.875), aniso_max=16, aniso_round=0, lod_from=0,
                lod_bias=0.0, aniso_ratio_bias=0.0, normal_axis=(0, 1, 2), normal_sign=(1.0, -1.0, -1.0))


def msaa_candidates():
    std = np.array([(0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875)])
    cands = {"d3d rotated grid (default)": std, "mirrored in x": std * (-1, 1) + (1, 0), "mirrored in y": std * (1, -1) + (0, 1),
             "rotated 180": 1 - std, "transposed": std[:, ::-1], "ordered grid 0.25 / 0.75": np.array([(0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75)]),
             "all at the centre (no multisampling)": np.full((4, 2), 0.5) + np.array([(-1, -1), (1, -1), (-1, 1), (1, 1)]) * 1e-3}
    out, seen = {}, set()
    for k, v in cands.items():  # the ORDER of the four samples is immaterial (the resolve is a mean): one candidate per SET
        key = tuple(sorted((round(float(x), 6), round(float(y), 6)) for x, y in v))
        if key not in seen:
            seen.add(key)
            out[k] = dict(msaa_x=tuple(float(x) for x in v[:, 0]), msaa_y=tuple(float(y) for y in v[:, 1]))
    return out


def aniso_candidates():
    out = {}
    for amax, rnd, lod, bias in itertools.product((1, 2, 4, 8, 16), (0, 1, 2), (0, 1, 2), (-0.5, 0.0, 0.5)):
        if amax == 1 and (rnd, lod) != (0, 0):
            continue  # one probe: the rounding rule is moot, lod_from 0 == 2
        out[f"max {amax:2d}, round {('ceil', 'nearest', 'floor')[rnd]}, lod log2({('Pmax/N', 'Pmin', 'Pmax')[lod]}) {bias:+.1f}"] = dict(
            aniso_max=amax, aniso_round=rnd, lod_from=lod, lod_bias=bias)
    return out


def normal_candidates():
    out = {}
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1.0, -1.0), repeat=3):
            out["RGB = (" + ", ".join(("+" if s > 0 else "-") + "xyz"[a] for a, s in zip(perm, signs)) + ") of the OpenCV camera frame"] = dict(
                normal_axis=tuple(perm), normal_sign=tuple(signs))
    return out


class Renderer:
    """Oracle (CPU) or HIP renders of the calibration views under a conventions record, as uint8 [N,H,W,3]."""

    def __init__(self, mesh, mesh_units, TCO, K, res, hip):
        from happypose_amd.mesh_store import PackedMeshes, RigidObject, RigidObjectDataset

        self.ds = RigidObjectDataset([RigidObject("calibration_object", Path(mesh), mesh_units=mesh_units)])
        self.TCO, self.K, self.res, self.hip = TCO, K, tuple(int(r) for r in res), hip
        self.obj = np.zeros(len(TCO), np.int32)
        if hip:
            import torch
            from happypose_amd import ops

            self.ops, self.torch = ops, torch
            self.store = ops.MeshStore(self.ds, torch.device("cuda:0"))
            self.packed = self.store.packed
        else:
            self.packed = PackedMeshes(self.ds)
        self.textured = bool((self.packed.obj[:, 4] >= 0).any())

    def __call__(self, conv):
        rec = dict(DEFAULTS, **conv)
        if self.hip:
            t = self.torch
            self.store.set_raster_conventions(rec)
            rgb, nrm, dep, _ = self.ops.rasterize(self.store, t.as_tensor(self.obj), t.as_tensor(self.TCO), t.as_tensor(self.K), self.res,
                                                  render_normals=True, render_depth=True, msaa=True, aniso=True)
            self.store.set_raster_conventions(None)
            r = dict(rgbs=rgb.cpu().numpy(), normals=nrm.cpu().numpy(), depths=dep.cpu().numpy())
        else:
            from oracle import native

            native.set_raster_conventions(rec)
            r = native.rasterize(self.packed, self.obj, self.TCO, self.K, self.res, True, True, False, msaa=True, aniso=True)
            native.set_raster_conventions(None)
        u8 = lambda a: np.round(np.moveaxis(a, 1, -1) * 255.0).astype(np.int16)  # noqa: E731
        return u8(r["rgbs"]), u8(r["normals"]), r["depths"][:, 0]


def erode(mask, k):
    m = mask.copy()
    for _ in range(k):
        m[:, 1:-1, 1:-1] = m[:, 1:-1, 1:-1] & m[:, :-2, 1:-1] & m[:, 2:, 1:-1] & m[:, 1:-1, :-2] & m[:, 1:-1, 2:]
        m[:, 0] = m[:, -1] = False
        m[:, :, 0] = m[:, :, -1] = False
    return m


def fit(target_rgb, target_nrm, target_depth, render, log=print):
    """Coordinate descent over the three groups (they move disjoint pixel sets / channels).  Returns (record, report)."""
    covered = target_depth > 0
    interior = erode(covered, 2)
    grown = ~erode(~covered, 2)
    band = grown & ~interior                        # silhouette band: the pixels multisampling moves
    t_rgb, t_nrm = target_rgb.astype(np.int16), target_nrm.astype(np.int16)
    best = dict(DEFAULTS)
    report = {}

    def score(cands, channel, mask, name):
        rows = []
        for label, conv in cands.items():
            rgb, nrm, _ = render(dict(best, **conv))
            img, tgt = (rgb, t_rgb) if channel == "rgb" else (nrm, t_nrm)
            rows.append((float(np.abs(img - tgt)[mask].mean()) if mask.any() else float("nan"), label, conv))
        rows.sort(key=lambda r: r[0])
        log(f"\n== {name}: mean |render - Panda3D| in 8-bit units on {int(mask.sum())} pixels (best first)")
        for sc, label, _ in rows[:8]:
            log(f"   {sc:8.4f}  {label}")
        margin = rows[1][0] - rows[0][0] if len(rows) > 1 else float("nan")
        log(f"   margin of the best over the runner-up: {margin:.4f}" + ("  (AMBIGUOUS: below 0.05 -- more / other views needed)" if margin < 0.05 else ""))
        report[name] = {"best": rows[0][1], "score": rows[0][0], "runner_up": rows[1][1] if len(rows) > 1 else None, "margin": margin,
                        "default_score": next((sc for sc, label, conv in rows if all(DEFAULTS[k] == v for k, v in conv.items())), None)}
        best.update(rows[0][2])

    score(normal_candidates(), "nrm", interior, "eye-normal axis map")
    score(msaa_candidates(), "nrm", band, "multisample positions")   # the normal image is untextured: the cleanest silhouette signal
    if render.textured:
        score(aniso_candidates(), "rgb", interior, "anisotropic filter rule")
    else:
        log("\n== anisotropic filter rule: SKIPPED -- the mesh has no texture (the reference's test asset ships without obj_000001.png)")
    return best, report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("views", nargs="?", help="NPZ written by tools/panda3d_render_calibration_views.py")
    ap.add_argument("--mesh", help="the mesh file the views were rendered from (its texture beside it)")
    ap.add_argument("--hip", action="store_true", help="render the candidates with the HIP rasteriser (needs an MI355X) instead of the CPU oracle")
    ap.add_argument("--out", default="raster_conventions.json")
    ap.add_argument("--self-test", action="store_true")
    args = ap.parse_args()
    if args.self_test:
        return self_test(args.hip)
    assert args.views and args.mesh, "give the Panda3D views (NPZ) and --mesh"
    d = np.load(args.views, allow_pickle=False)
    render = Renderer(args.mesh, str(d["mesh_units"]), d["TCO"], d["K"], d["resolution"], args.hip)
    best, report = fit(d["rgb"], d["normals"], d["depth"], render)
    json.dump(best, open(args.out, "w"), indent=1)
    print(f"\nbest record -> {args.out}\n{json.dumps(best)}\n{json.dumps(report, indent=1)}")


def self_test(hip=False, log=print):
    """No Panda3D: the 'reference' views come from the oracle under a hidden record; the fit must find it."""
    import tempfile

    from happypose_amd.synthetic import make_object_dataset
    from panda3d_render_calibration_views import calibration_views

    hidden = dict(msaa_x=(0.625, 0.125, 0.875, 0.375), msaa_y=(0.125, 0.375, 0.625, 0.875), aniso_max=8, aniso_round=1, lod_from=0,
                  lod_bias=0.5, normal_axis=(0, 2, 1), normal_sign=(1.0, 1.0, -1.0))
    TCO, K, res = calibration_views(n_extra=5)
    K = K.copy(); K[:, :2] *= 0.5; res = (240, 320)      # quarter-size views keep the CPU self-test short
    ds = make_object_dataset(1, seed=1, tex_size=256)

    class R(Renderer):
        def __init__(self):  # synthetic textured object instead of a mesh file
            from happypose_amd.mesh_store import PackedMeshes

            self.ds, self.TCO, self.K, self.res, self.hip = ds, TCO * np.array([[[1, 1, 1, 1.0]] * 3 + [[1, 1, 1, 1]]], np.float32), K, res, False
            self.obj = np.zeros(len(TCO), np.int32)
            self.packed = PackedMeshes(ds)
            self.textured = True

    render = R()
    rgb, nrm, dep = render(hidden)
    best, report = fit(rgb.astype(np.uint8), nrm.astype(np.uint8), dep, render, log=log)
    for k, v in hidden.items():
        if k in ("msaa_x", "msaa_y"):
            continue
        assert best[k] == v or (isinstance(v, tuple) and np.allclose(best[k], v)), (k, best[k], v)
    same_set = lambda r: sorted(zip(np.round(r["msaa_x"], 6), np.round(r["msaa_y"], 6)))  # noqa: E731
    assert same_set(best) == same_set(hidden), (best["msaa_x"], best["msaa_y"])
    log("self-test ok: the hidden record was recovered")
    return best, report


if __name__ == "__main__":
    main()
