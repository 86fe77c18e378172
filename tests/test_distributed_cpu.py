"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo process groups exercise the shard
partition and the single all-gather that merges refined poses (happypose_amd.distributed)."""

import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, pandas as pd
    sys.path.insert(0, os.environ["HP_ROOT"])
    from happypose_amd import distributed as D
    from happypose_amd.tensor_collection import PandasTensorCollection

    rank, local_rank, world = D.init_distributed("gloo")
    assert world == int(os.environ["WORLD_SIZE"]) and D.get_rank() == rank and D.get_world_size() == world
    n_total = int(os.environ["HP_N"])
    rs = np.random.RandomState(0)                      # same "global" result table on every rank
    poses_all = torch.as_tensor(rs.normal(size=(n_total, 4, 4)).astype(np.float32))
    scores_all = torch.as_tensor(rs.normal(size=n_total).astype(np.float32))
    s, e = D.shard_range(n_total)
    # shards are contiguous, disjoint, cover everything, sizes differ by at most one
    bounds = [D.shard_range(n_total, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_total
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in bounds]
    assert max(sizes) - min(sizes) <= 1
    poses, scores = D.gather_poses(poses_all[s:e].clone(), scores_all[s:e].clone(), s, n_total)
    assert torch.equal(poses, poses_all) and torch.equal(scores, scores_all)
    p2, s2 = D.gather_poses(poses_all[s:e].clone(), None, s, n_total)
    assert torch.equal(p2, poses_all) and float(s2.abs().sum()) == 0.0
    # collection gather (replaces the reference's rank files, TB/utils/tensor_collection.py:166-187)
    df = pd.DataFrame({"label": [f"obj{i % 3}" for i in range(s, e)], "hypothesis_id": np.arange(s, e)})
    coll = PandasTensorCollection(df, poses=poses_all[s:e].clone())
    full = coll.gather_distributed()
    assert len(full) == n_total and full.infos.hypothesis_id.tolist() == list(range(n_total))
    assert torch.equal(full.poses, poses_all)
    # sharded refinement (SURVEY.md 8e) with a stand-in predictor: the result must not depend on
    # the number of ranks
    from types import SimpleNamespace

    class FakeModel:
        def forward(self, images, K, labels, TCO, n_iterations=1, im_ids=None):
            out = TCO.clone()
            for _ in range(n_iterations):
                fr = images if im_ids is None else images[im_ids.long()]
                out = out * 0.5 + fr.mean(dim=(1, 2, 3))[:, None, None] + torch.as_tensor(
                    [float(l[3:]) for l in labels])[:, None, None]
            return {f"iteration={n_iterations}": SimpleNamespace(TCO_output=out)}

    images = torch.as_tensor(rs.normal(size=(2, 3, 4, 5)).astype(np.float32))
    Kc = torch.eye(3)[None].repeat(2, 1, 1)
    labels = [f"obj{i % 5}" for i in range(n_total)]
    im_ids = torch.as_tensor(np.arange(n_total) % 2, dtype=torch.int32)
    ref = FakeModel().forward(images, Kc, labels, poses_all, n_iterations=3, im_ids=im_ids)["iteration=3"].TCO_output
    got, sc = D.refine_sharded(FakeModel(), images, Kc, labels, poses_all, 3, im_ids=im_ids,
                               scores_fn=lambda o: o.TCO_output[:, 0, 0])
    assert torch.equal(got, ref) and torch.equal(sc, ref[:, 0, 0])
    # the reference's calling convention (im_ids=None: images / K gathered per hypothesis) shards the frames too
    got2, _ = D.refine_sharded(FakeModel(), images[im_ids.long()], Kc[im_ids.long()], labels, poses_all, 3)
    assert torch.equal(got2, ref)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(f"rank {rank} ok")
""")


ESTIMATOR_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, pandas as pd
    sys.path.insert(0, os.environ["HP_ROOT"])
    from types import SimpleNamespace
    from happypose_amd import distributed as D, ops
    from happypose_amd.pose_estimator import CosyPoseEstimator, ObservationTensor, PoseEstimator
    from happypose_amd.tensor_collection import PandasTensorCollection

    rank, local_rank, world = D.init_distributed("gloo")
    n_det, n_hyp = int(os.environ["HP_N"]), int(os.environ.get("HP_NHYP", "3"))
    n_pose_hyp = int(os.environ.get("HP_NPOSE", "2"))
    rs = np.random.RandomState(0)
    LABELS = [f"obj{i}" for i in range(4)]

    class FakeStore:
        labels = LABELS
        def ids_of(self, labels):
            return torch.as_tensor([LABELS.index(l) for l in labels], dtype=torch.int32)

    class FakeModel:
        # a deterministic stand-in predictor: every output row depends only on its own input row
        device = torch.device("cpu")
        mesh_db = None
        cfg = SimpleNamespace(init_method="v0")
        store = FakeStore()
        calls = 0
        def _obj(self, labels):
            return torch.as_tensor([float(LABELS.index(l)) for l in labels])
        def __call__(self, images, K, TCO, n_iterations, labels, im_ids):
            FakeModel.calls += 1
            out, T = {}, TCO.clone().float()
            fr = images[im_ids.long()].mean(dim=(1, 2, 3))
            for n in range(1, n_iterations + 1):
                Tn = T * 0.5 + fr[:, None, None] + self._obj(labels)[:, None, None] * 0.1
                out[f"iteration={n}"] = SimpleNamespace(TCO_output=Tn, TCO_input=T, K_crop=K[im_ids.long()] * (1 + n),
                                                        boxes_rend=Tn[:, 0, :4] * 2, boxes_crop=Tn[:, 1, :4] * 3)
                T = Tn
            return out
        def forward_coarse(self, images, K, labels, TCO_input, cuda_timer=False, return_debug_data=False, im_ids=None):
            logit = (TCO_input[:, :3, :3].reshape(len(labels), -1) * torch.arange(9.0)).sum(1, keepdim=True) + self._obj(labels)[:, None]
            return {"logits": logit, "scores": torch.sigmoid(logit), "render_time": 0.0, "model_time": 0.0, "time": 0.0}
        # the guard: `fire` lists the ranks whose NEXT status query reports a non-finite forward (once)
        fire = ()
        exact = False
        def numerics_status(self):
            if rank in FakeModel.fire:
                FakeModel.fire = ()
                return 1
            return 0
        @property
        def backbone(self):
            return self
        def force_exact(self, on=True):
            FakeModel.exact = bool(on)

    def fake_init(store, boxes, K, im_ids, obj_ids, R=None, box_ids=None, rot_ids=None, n_points=None):
        n = len(obj_ids)
        T = torch.eye(4).repeat(n, 1, 1)
        T[:, :3, :3] = R[rot_ids.long()]
        T[:, :2, 3] = boxes[box_ids.long()][:, :2] * 1e-3
        T[:, 2, 3] = 0.5 + 0.01 * obj_ids.float()
        return T
    ops.tco_init_autodepth = fake_init

    images = torch.as_tensor(rs.uniform(size=(2, 3, 6, 8)).astype(np.float32))
    Kc = torch.as_tensor(rs.uniform(1, 2, size=(2, 3, 3)).astype(np.float32))
    obs = ObservationTensor(images, Kc)

    def same(a, b):
        assert list(a.infos.columns) == list(b.infos.columns), (list(a.infos.columns), list(b.infos.columns))
        pd.testing.assert_frame_equal(a.infos, b.infos)
        assert sorted(a.tensors) == sorted(b.tensors)
        for k in a.tensors:
            assert torch.equal(a.tensors[k], b.tensors[k]), k

    # CosyPose: externally generated hypotheses -> refiner (the C2 / C4 shape)
    N = n_det * n_hyp
    hyp = lambda: PandasTensorCollection(pd.DataFrame({"label": [LABELS[i % 4] for i in range(N)], "batch_im_id": np.arange(N) % 2,
                                                       "instance_id": np.arange(N) // n_hyp}),
                                         poses=torch.as_tensor(rs.normal(size=(N, 4, 4)).astype(np.float32)))
    rs = np.random.RandomState(1); h1 = hyp(); rs = np.random.RandomState(1); h2 = hyp()
    est = CosyPoseEstimator(refiner_model=FakeModel(), coarse_model=FakeModel(), bsz_objects=4)
    assert est.shard_hypotheses is None and D.sharding_active(None)
    FakeModel.calls = 0
    f_sh, e_sh = est.run_inference_pipeline(obs, data_TCO_init=h1, n_coarse_iterations=0, n_refiner_iterations=3)
    calls_sharded = FakeModel.calls
    est.shard_hypotheses = False
    FakeModel.calls = 0
    f_1, e_1 = est.run_inference_pipeline(obs, data_TCO_init=h2, n_coarse_iterations=0, n_refiner_iterations=3)
    assert calls_sharded < FakeModel.calls or N <= 4, (calls_sharded, FakeModel.calls)   # each rank ran only its shard
    same(f_sh, f_1)
    for k in e_1["refiner_all_hypotheses"]["preds"]:
        same(e_sh["refiner_all_hypotheses"]["preds"][k], e_1["refiner_all_hypotheses"]["preds"][k])
    s, e = D.shard_range(N)
    assert e_sh["refiner"]["data"]["shard"] == (s, e) and e_1["refiner"]["data"]["shard"] == (0, N)

    # MegaPose: detections -> coarse grid (sharded over detection x rotation rows) -> top-K -> refiner -> scoring -> top-1
    det = lambda: PandasTensorCollection(pd.DataFrame({"label": [LABELS[i % 4] for i in range(n_det)], "batch_im_id": np.arange(n_det) % 2,
                                                       "instance_id": np.arange(n_det)}),
                                         bboxes=torch.as_tensor(rs.uniform(10, 200, size=(n_det, 4)).astype(np.float32)))
    rs = np.random.RandomState(2); d1 = det(); rs = np.random.RandomState(2); d2 = det()
    mp = PoseEstimator(refiner_model=FakeModel(), coarse_model=FakeModel(), bsz_objects=4, bsz_images=16, SO3_grid_size=72)
    g_sh, x_sh = mp.run_inference_pipeline(obs, detections=d1, n_refiner_iterations=2, n_pose_hypotheses=n_pose_hyp)
    mp.shard_hypotheses = False
    g_1, x_1 = mp.run_inference_pipeline(obs, detections=d2, n_refiner_iterations=2, n_pose_hypotheses=n_pose_hyp)
    same(g_sh, g_1)
    same(x_sh["coarse"]["preds"], x_1["coarse"]["preds"])
    same(x_sh["coarse_filter"]["preds"], x_1["coarse_filter"]["preds"])
    same(x_sh["scoring"]["preds"], x_1["scoring"]["preds"])
    assert torch.equal(x_sh["coarse"]["data"]["logits"], x_1["coarse"]["data"]["logits"])
    assert torch.equal(x_sh["coarse"]["data"]["TCO"], x_1["coarse"]["data"]["TCO"])
    # the guard fires on ONE rank only: every rank repeats the stage AND every rank's networks go exact first
    est.shard_hypotheses = None
    FakeModel.fire, FakeModel.exact, FakeModel.calls = (world - 1,), False, 0
    rs = np.random.RandomState(1); h3 = hyp()
    f_g, _ = est.run_inference_pipeline(obs, data_TCO_init=h3, n_coarse_iterations=0, n_refiner_iterations=3)
    assert FakeModel.exact, "a rank whose own guard stayed silent must be forced exact as well"
    same(f_g, f_1)
    # every rank holds the same final table: compare a checksum across ranks
    chk = torch.tensor([float(g_sh.poses.double().sum()), float(f_sh.poses.double().sum())], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(f"rank {rank} ok")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_total", [(2, 256), (3, 37), (2, 1), (2, 2)])
def test_shard_and_all_gather_gloo(tmp_path, world, n_total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HP_ROOT=str(ROOT), HP_N=str(n_total), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in out, out[-2000:]


@pytest.mark.parametrize("world,n_det,n_hyp,n_pose_hyp", [(2, 5, 3, 2), (3, 4, 3, 2), (2, 1, 3, 2), (3, 1, 1, 1), (2, 1, 1, 1)])
def test_estimators_shard_behind_the_entry_point_gloo(tmp_path, world, n_det, n_hyp, n_pose_hyp):
    """``run_inference_pipeline`` with torch.distributed initialised: every rank computes its shard of the hypothesis
    rows (refiner, coarse grid, scoring) and the merged result equals the unsharded run bit for bit, infos included --
    the estimator's bookkeeping driven by stand-in predictors (the real ones need the GPU: tests/test_gpu_pipeline.py).
    The (3, 1, 1, 1) / (2, 1, 1, 1) cases have MORE RANKS THAN ROWS: the refiner table has one row (the MegaPose example
    setup: one detection, ``n_pose_hypotheses=1``), so ranks hold empty shards."""
    script = tmp_path / "worker_est.py"
    script.write_text(ESTIMATOR_WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HP_ROOT=str(ROOT), HP_N=str(n_det), HP_NHYP=str(n_hyp), HP_NPOSE=str(n_pose_hyp),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in out, out[-3000:]


def test_single_process_is_identity():
    import torch

    from happypose_amd import distributed as D

    assert D.get_world_size() == 1 and D.get_rank() == 0
    assert D.shard_range(10) == (0, 10)
    p = torch.eye(4).repeat(5, 1, 1)
    out, sc = D.gather_poses(p, None, 0, 5)
    assert out is p and sc.shape == (5,)
