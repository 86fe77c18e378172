"""Multi-GPU: one process per GPU, hypotheses sharded by rank, ONE all-gather at the end.

The refinement loop has no data dependence between hypotheses, so the
path shards with no collective inside the iteration loop.  The reference merges
multi-rank inference results through rank files on a shared filesystem + barriers
(``TB/utils/tensor_collection.py:166-187``, helpers ``TB/utils/distributed.py:36-152``);
here each rank contributes ``[N_local, 18]`` fp32 rows (16 pose + score + global id) to a
single ``all_gather_into_tensor`` -- RCCL over xGMI on MI355X (backend "nccl"), gloo in the
CPU tests.  72 B per hypothesis: latency-bound, so one fused message instead of three.
"""

from __future__ import annotations

import os
from datetime import timedelta
from typing import Optional, Tuple

import torch
import torch.distributed as dist

ROW = 18  # 16 pose floats + score + global hypothesis id


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment
    (``torch.distributed.run``).  Returns ``(rank, local_rank, world_size)``."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timedelta(minutes=30))
    return rank, local_rank, world


def shard_range(n: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous ``[start, end)`` of ``n`` hypotheses owned by ``rank`` (sizes differ by at
    most one; earlier ranks take the remainder) -- ``np.array_split`` semantics like the
    reference's ``DistributedSceneSampler`` (``TB/datasets/samplers.py:40-46``)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_rows(rows: torch.Tensor, start: int, n_total: int) -> torch.Tensor:
    """The ONE collective of the path.  ``rows [N_local, C]`` fp32 = this rank's contiguous shard of a global row table
    beginning at global row ``start``; every rank returns the whole table ``[n_total, C]`` in global order.  Ragged
    shards are padded to the largest one; a trailing id column (-1 = padding) travels with the rows, so the result does
    not depend on how the ranks cut the table.  ``all_gather_into_tensor``: RCCL over xGMI with backend "nccl", gloo in
    the CPU tests."""
    world = get_world_size()
    n_local, c = rows.shape
    if world == 1:
        assert n_local == n_total
        return rows
    dev = rows.device
    cap = (n_total + world - 1) // world
    assert n_local <= cap, "a shard is larger than ceil(n_total / world): not a shard_range() partition"
    buf = torch.full((cap, c + 1), -1.0, dtype=torch.float32, device=dev)
    buf[:n_local, :c] = rows.float()
    buf[:n_local, c] = torch.arange(start, start + n_local, device=dev, dtype=torch.float32)
    out = torch.empty((world * cap, c + 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    ids = out[:, c].long()
    keep = ids >= 0
    out, ids = out[keep], ids[keep]
    assert out.shape[0] == n_total, "shards do not cover the row table"
    res = torch.empty((n_total, c), dtype=torch.float32, device=dev)
    res[ids] = out[:, :c]
    return res


def gather_poses(poses: torch.Tensor, scores: Optional[torch.Tensor], start: int, n_total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks get all results.  ``poses [N_local,4,4]`` (this rank's contiguous shard
    beginning at global index ``start``), ``scores [N_local]`` or None.  Returns
    ``(poses [n_total,4,4], scores [n_total])`` in global order: one ``[N_local, 18]`` row per hypothesis
    (16 pose floats + score + global id) through :func:`gather_rows`."""
    n_local = poses.shape[0]
    if scores is None:
        scores = torch.zeros(n_local, device=poses.device)
    if get_world_size() == 1:
        assert n_local == n_total
        return poses, scores
    rows = torch.cat([poses.reshape(n_local, 16).float(), scores.reshape(n_local, 1).float()], dim=1)
    res = gather_rows(rows, start, n_total)
    return res[:, :16].reshape(n_total, 4, 4), res[:, 16]


def sharding_active(flag: Optional[bool] = None) -> bool:
    """Whether an entry point shards its hypothesis rows over the ranks: ``flag`` if given, else whenever
    ``torch.distributed`` is initialised with more than one rank.  Sharding is a COLLECTIVE: every rank must make the
    same call with the same inputs (the frame, detections, meshes and weights are replicated).  Pass
    ``False`` where ranks work on different scenes (the reference's evaluation sampler)."""
    return get_world_size() > 1 if flag is None else bool(flag and get_world_size() > 1)


def all_ranks_max(value: int, device) -> int:
    """MAX of a small integer over the ranks (the estimators agree on a guard re-run with it)."""
    if get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def refine_sharded(model, images: torch.Tensor, K: torch.Tensor, labels, TCO: torch.Tensor, n_iterations: int,
                   im_ids: Optional[torch.Tensor] = None, scores_fn=None, bsz: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """every rank holds the frame(s), meshes and weights; the hypotheses -- sorted by
    (detection, hypothesis id) by the caller -- are split into contiguous shards, each rank runs the
    refiner loop on its shard (no collective inside the loop; chunks of ``bsz``, default the predictor's
    ``max_batch``) and ONE ``all_gather_into_tensor`` hands every rank all refined poses (``[N,4,4]`` in the caller's
    order) and scores.  Replaces the reference's rank files + barriers
    (``TB/utils/tensor_collection.py:166-187``, ``MP/evaluation/prediction_runner.py:65-76``).

    ``model`` is a predictor (``PosePredictor`` / ``CosyPosePosePredictor`` / ``TwoLanePredictor``: anything whose
    ``forward(images, K, labels, TCO, n_iterations=, im_ids=)`` returns the per-iteration outputs);
    ``scores_fn(last_output) -> [n_chunk]`` optionally attaches a score to every hypothesis."""
    n = len(labels)
    assert TCO.shape == (n, 4, 4)
    s, e = shard_range(n)
    labels = list(labels)
    if im_ids is None:  # reference convention: images / K already gathered per hypothesis -> shard them too
        assert K.shape[0] == n and images.shape[0] == n, "without im_ids, images and K must hold one row per hypothesis"
    # with im_ids the frames stay whole on every rank: im_ids[a:b] keeps indexing the global frame list
    if bsz is None:
        bsz = int(getattr(model, "max_batch", 0) or 0) or max(e - s, 1)
    poses_l, scores_l = [], []
    for a in range(s, e, bsz):
        b = min(e, a + bsz)
        if im_ids is None:
            out = model.forward(images[a:b], K[a:b], labels[a:b], TCO[a:b], n_iterations=n_iterations, im_ids=None)
        else:
            out = model.forward(images, K, labels[a:b], TCO[a:b], n_iterations=n_iterations, im_ids=im_ids[a:b])
        last = out[f"iteration={n_iterations}"]
        poses_l.append(last.TCO_output)
        if scores_fn is not None:
            scores_l.append(scores_fn(last))
    if poses_l:
        poses = torch.cat(poses_l) if len(poses_l) > 1 else poses_l[0]
        scores = (torch.cat(scores_l) if len(scores_l) > 1 else scores_l[0]) if scores_l else None
    else:  # more ranks than hypotheses
        poses, scores = torch.zeros((0, 4, 4), dtype=torch.float32, device=getattr(model, "device", TCO.device)), None
    return gather_poses(poses, scores, s, n)


def gather_collection(coll):
    """``PandasTensorCollection.gather_distributed`` without the filesystem: tensors go
    through ``all_gather`` (ragged lengths allowed), ``infos`` through ``all_gather_object``."""
    world = get_world_size()
    if world == 1:
        return coll
    from .tensor_collection import PandasTensorCollection, concatenate

    infos = [None] * world
    dist.all_gather_object(infos, coll.infos)
    lens = [len(i) for i in infos]
    cap = max(lens)
    parts = [dict() for _ in range(world)]
    for k, t in coll.tensors.items():
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad)
        for r in range(world):
            parts[r][k] = out[r][: lens[r]]
    return concatenate([PandasTensorCollection(infos[r], **parts[r]) for r in range(world)])
