"""Result containers of the path: a table of rows whose columns are batched tensors, optionally
with a pandas ``infos`` frame holding the non-tensor columns.

Drop-in for the names the reference's callers use (``TensorCollection``,
``PandasTensorCollection``, ``concatenate``, ``filter_top_pose_estimates``;
``TB/utils/tensor_collection.py:28-230``): tensors are reachable as attributes, ``coll[ids]``
selects the same rows of every tensor and of ``infos`` and renumbers the frame,
``len(coll) == len(coll.infos)``.  The implementation is organised differently: both classes
are one row table (``_RowTable``) whose operations are expressed as *map every column* /
*rebuild with new columns*, the conversions (``cuda``, ``half`` ...) are generated from a table, and
the multi-rank gather goes through one all-gather (``happypose_amd.distributed``) instead of the
reference's rank files (``:166-187``).
"""

from __future__ import annotations

from typing import Callable, Dict, List, Optional

import pandas as pd
import torch

_CONVERSIONS = {"cuda": "cuda", "cpu": "cpu", "float": torch.float, "double": torch.double, "half": torch.half}


class _RowTable:
    """Columns (name -> tensor with the rows on dim 0) plus optional per-row ``infos``."""

    _has_infos = False

    def _init_columns(self, tensors: Dict[str, torch.Tensor]) -> None:
        # the column dict lives in __dict__ directly so that __setattr__ can tell columns from
        # ordinary attributes
        self.__dict__["_tensors"] = dict(tensors)

    # -- columns ------------------------------------------------------------------
    @property
    def tensors(self) -> Dict[str, torch.Tensor]:
        return self.__dict__["_tensors"]

    def register_tensor(self, name: str, tensor) -> None:
        self.tensors[name] = tensor

    def delete_tensor(self, name: str) -> None:
        self.tensors.pop(name)

    @property
    def device(self):
        for t in self.tensors.values():
            return t.device
        raise ValueError("empty collection has no device")

    def __getattr__(self, name):
        # only called when normal lookup fails: columns read as attributes
        cols = self.__dict__.get("_tensors")
        if cols is None or name not in cols:
            raise AttributeError(name)
        return cols[name]

    def __setattr__(self, name, value):
        cols = self.__dict__.get("_tensors")
        if cols is None:
            raise ValueError("Please call __init__")
        if name in cols:
            cols[name] = value
        else:
            self.__dict__[name] = value

    # -- the two primitives everything else is written with ------------------------------
    def _rebuild(self, tensors: Dict[str, torch.Tensor], infos: Optional[pd.DataFrame] = None):
        if self._has_infos:
            return type(self)(self.infos if infos is None else infos, **tensors)
        return type(self)(**tensors)

    def _mapped(self, fn: Callable[[torch.Tensor], torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {name: fn(t) for name, t in self.tensors.items()}

    # -- conversions (in place, like the reference; they return self for chaining) ---------
    def to(self, torch_attr):
        self.tensors.update(self._mapped(lambda t: t.to(torch_attr)))
        return self

    def clone(self):
        return self._rebuild(self._mapped(torch.clone), self.infos.copy() if self._has_infos else None)

    def _describe_columns(self) -> str:
        return "".join(f"    {n}: {tuple(t.shape)} {t.dtype} {t.device},\n" for n, t in self.tensors.items())


for _name, _target in _CONVERSIONS.items():
    setattr(_RowTable, _name, (lambda target: lambda self: self.to(target))(_target))


class TensorCollection(_RowTable):
    def __init__(self, **tensors):
        self._init_columns(tensors)

    def __getitem__(self, ids):
        return self._rebuild(self._mapped(lambda t: t[ids]))

    def __repr__(self):
        return f"{type(self).__name__}(\n{self._describe_columns()})"

    def __getstate__(self):
        return {"tensors": self.tensors}

    def __setstate__(self, state):
        self._init_columns(state["tensors"])


class PandasTensorCollection(_RowTable):
    _has_infos = True

    def __init__(self, infos: pd.DataFrame, **tensors):
        self._init_columns(tensors)
        self.infos = infos.reset_index(drop=True)
        self.meta = {}

    def __len__(self):
        return len(self.infos)

    def __getitem__(self, ids):
        rows = ids.cpu().numpy() if isinstance(ids, torch.Tensor) else ids
        return self._rebuild(self._mapped(lambda t: t[ids]), self.infos.iloc[rows])

    def merge_df(self, df, *args, **kwargs):
        merged = self.infos.merge(df, how="left", *args, **kwargs)
        if len(merged) != len(self.infos):
            raise AssertionError("merge_df must keep one row per entry")
        return self._rebuild(self.tensors, merged)

    def __repr__(self):
        return f"{type(self).__name__}(\n{self._describe_columns()}{'-' * 40}\n    infos:\n{self.infos!r}\n)"

    def __getstate__(self):
        return {"tensors": self.tensors, "infos": self.infos, "meta": self.meta}

    def __setstate__(self, state):
        self._init_columns(state["tensors"])
        self.infos, self.meta = state["infos"], state["meta"]

    def gather_distributed(self, tmp_dir=None):
        """Reference signature (``TB/utils/tensor_collection.py:166-187``); ``tmp_dir`` is unused --
        the rows travel through one all-gather instead of per-rank files."""
        from .distributed import gather_collection

        return gather_collection(self)


def concatenate(datas: List[PandasTensorCollection]) -> PandasTensorCollection:
    """Stack collections row-wise, ignoring empty ones (``TB/utils/tensor_collection.py:28-42``)."""
    parts = [d for d in datas if len(d)]
    if not parts:
        return PandasTensorCollection(infos=pd.DataFrame())
    head = parts[0]
    if any(type(d) is not type(head) for d in parts):
        raise AssertionError("concatenate needs collections of one type")
    frame = pd.concat([d.infos for d in parts], axis=0, sort=False)
    columns = {name: torch.cat([d.tensors[name] for d in parts], dim=0) for name in head.tensors}
    return PandasTensorCollection(infos=frame, **columns)


def filter_top_pose_estimates(data_TCO: PandasTensorCollection, top_K: int, group_cols: List[str],
                              filter_field: str, ascending: bool = False) -> PandasTensorCollection:
    """Keep the ``top_K`` rows of each group ranked by ``filter_field``.  The result is in ranked
    (not input) order, as the reference's ``sort_values().groupby().head()`` leaves it
    (``TB/utils/tensor_collection.py:201-230``; golden G8 pins this including the tie case)."""
    ranked = data_TCO.infos.sort_values(filter_field, ascending=ascending)
    keep = ranked.groupby(group_cols).head(top_K).index
    return data_TCO[list(keep)]
