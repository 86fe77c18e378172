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


def test_single_process_is_identity():
    import torch

    from happypose_amd import distributed as D

    assert D.get_world_size() == 1 and D.get_rank() == 0
    assert D.shard_range(10) == (0, 10)
    p = torch.eye(4).repeat(5, 1, 1)
    out, sc = D.gather_poses(p, None, 0, 5)
    assert out is p and sc.shape == (5,)
