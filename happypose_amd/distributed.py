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


def gather_poses(poses: torch.Tensor, scores: Optional[torch.Tensor], start: int, n_total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks get all results.  ``poses [N_local,4,4]`` (this rank's contiguous shard
    beginning at global index ``start``), ``scores [N_local]`` or None.  Returns
    ``(poses [n_total,4,4], scores [n_total])`` in global order."""
    world = get_world_size()
    n_local = poses.shape[0]
    dev = poses.device
    if scores is None:
        scores = torch.zeros(n_local, device=dev)
    if world == 1:
        assert n_local == n_total
        return poses, scores
    cap = (n_total + world - 1) // world  # ragged shards are padded to the largest one
    rows = torch.full((cap, ROW), -1.0, dtype=torch.float32, device=dev)
    rows[:n_local, :16] = poses.reshape(n_local, 16).float()
    rows[:n_local, 16] = scores.float()
    rows[:n_local, 17] = torch.arange(start, start + n_local, device=dev, dtype=torch.float32)
    out = torch.empty((world * cap, ROW), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, rows)
    ids = out[:, 17].long()
    keep = ids >= 0
    out, ids = out[keep], ids[keep]
    assert out.shape[0] == n_total, "shards do not cover the hypothesis set"
    res = torch.empty((n_total, ROW), dtype=torch.float32, device=dev)
    res[ids] = out
    return res[:, :16].reshape(n_total, 4, 4), res[:, 16]


def refine_sharded(model, images: torch.Tensor, K: torch.Tensor, labels, TCO: torch.Tensor, n_iterations: int,
                   im_ids: Optional[torch.Tensor] = None, scores_fn=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """every rank holds the frame(s), meshes and weights; the hypotheses -- sorted by
    (detection, hypothesis id) by the caller -- are split into contiguous shards, each rank runs the
    refiner loop on its shard (no collective inside the loop) and ONE ``all_gather_into_tensor``
    hands every rank all refined poses (``[N,4,4]`` in the caller's order) and scores.

    ``model`` is a predictor (``PosePredictor`` / ``CosyPosePosePredictor``: anything whose
    ``forward(images, K, labels, TCO, n_iterations=, im_ids=)`` returns the per-iteration outputs);
    ``scores_fn(last_output) -> [n_local]`` optionally attaches a score to every hypothesis."""
    n = len(labels)
    assert TCO.shape == (n, 4, 4)
    s, e = shard_range(n)
    labels = list(labels)
    ids = None if im_ids is None else im_ids[s:e]
    if im_ids is None:  # reference convention: images / K already gathered per hypothesis -> shard them too
        assert K.shape[0] == n and images.shape[0] == n, "without im_ids, images and K must hold one row per hypothesis"
        images, K = images[s:e], K[s:e]
    # with im_ids the frames stay whole on every rank: im_ids[s:e] keeps indexing the global frame list
    if e > s:
        out = model.forward(images, K, labels[s:e], TCO[s:e], n_iterations=n_iterations, im_ids=ids)
        last = out[f"iteration={n_iterations}"]
        poses = last.TCO_output
        scores = None if scores_fn is None else scores_fn(last)
    else:  # more ranks than hypotheses
        poses = TCO[:0].to(torch.float32)
        scores = None
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
