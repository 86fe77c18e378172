"""hipGraph replay of the refinement loop.

One refiner step is ``n_iterations x (pose_prep, crop, rasterise, ~40 conv launches, head, pose_update)`` --
about 250 launches per lane whose shapes and buffers do not change between calls of the same signature.
Capturing the step once and replaying it takes the host out of the loop (the reference has no counterpart: its
loop is Python + Panda3D worker processes per iteration).  Measured on MI355X (``profiles/r02j_bench_line*_graphs.json``):
+1.5 % on C2 and C3 -- with two lanes the kernels of a step add up to 1.5x its wall time, the GPU is the bound, not
the host's launch rate -- so replay is opt-in (``create_model_pose(..., graphs=True)``); it pays where the host is
slower or the batches smaller than in the benchmark configurations.

:class:`GraphCache` keeps one captured graph per call signature (constants + input shapes/dtypes):

* call 1 of a signature runs eagerly on the entry's own stream (this also sizes every per-stream workspace of the
  C library, which must not allocate during capture);
* call 2 captures (``torch.cuda.graph`` on that same stream; tensors created inside live in the graph's pool) and
  replays;
* later calls copy the inputs into the captured input buffers, replay, and return CLONES of the captured
  outputs (one fused multi-tensor copy), so results stay valid across replays like the eager path's.

Anything that changes what the launches would be -- conv algorithm, profiling events, the non-finite guard
switching a network to its exact kernels -- must :meth:`GraphCache.clear` the cache; the predictors do.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, List, Sequence, Tuple

import torch


def flatten(obj: Any, out: List[torch.Tensor]):
    """Nested lists / tuples / dicts of tensors (or None / scalars) -> spec, tensors appended to ``out``."""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return ("t", len(out) - 1)
    if isinstance(obj, dict):
        return ("d", [(k, flatten(v, out)) for k, v in obj.items()])
    if isinstance(obj, (list, tuple)):
        return ("l", [flatten(v, out) for v in obj])
    return ("c", obj)


def unflatten(spec, tensors: Sequence[torch.Tensor]):
    kind, val = spec
    if kind == "t":
        return tensors[val]
    if kind == "d":
        return {k: unflatten(v, tensors) for k, v in val}
    if kind == "l":
        return [unflatten(v, tensors) for v in val]
    return val


class _Entry:
    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.static_in: List[torch.Tensor] = []
        self.graph = None
        self.static_out: List[torch.Tensor] = []
        self.spec = None
        self.keepalive: List[Any] = []
        self.calls = 0


class GraphCache:
    MAX_ENTRIES = 8  # signatures kept (least recently used dropped): an entry owns clones of its inputs and a graph pool

    def __init__(self, device):
        self.device = torch.device(device)
        self.entries: Dict[Tuple, _Entry] = {}
        self.replays = 0

    def clear(self) -> None:
        self.entries.clear()

    @staticmethod
    def key_of(consts: Tuple, inputs: Sequence[torch.Tensor]) -> Tuple:
        return (tuple(consts), tuple((tuple(t.shape), t.dtype) for t in inputs))

    def run(self, consts: Tuple, inputs: Sequence[torch.Tensor], fn: Callable[..., Any], keepalive: Callable[[], List[Any]] = None):
        """``fn(*inputs)`` -> nested structure of tensors; ``consts`` = everything else the launches depend on.
        ``keepalive()`` returns objects whose device memory the launches reference besides inputs and outputs
        (reused scratch buffers): the entry holds them so that a later reallocation cannot free them under the graph."""
        key = self.key_of(consts, inputs)
        e = self.entries.pop(key, None)
        cur = torch.cuda.current_stream(self.device)
        if e is None:
            while len(self.entries) >= self.MAX_ENTRIES:
                self.entries.pop(next(iter(self.entries)))  # dicts keep insertion order: the first key is the oldest use
            e = _Entry(self.device)
            e.static_in = [t.clone() for t in inputs]
        self.entries[key] = e  # (re-)inserted last = most recently used
        e.calls += 1
        for s, t in zip(e.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        if e.calls == 1:  # eager, on the stream the capture will use: sizes the per-stream workspaces
            e.stream.wait_stream(cur)
            with torch.cuda.stream(e.stream):
                out = fn(*e.static_in)
            cur.wait_stream(e.stream)
            return out
        if e.graph is None:
            e.stream.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=e.stream):
                out = fn(*e.static_in)
            flat: List[torch.Tensor] = []
            e.spec = flatten(out, flat)
            e.static_out, e.graph = flat, g
            if keepalive is not None:
                e.keepalive = keepalive()
        e.graph.replay()
        self.replays += 1
        outs = [torch.empty_like(t) for t in e.static_out]
        if outs:
            torch._foreach_copy_(outs, e.static_out)
        return unflatten(e.spec, outs)
