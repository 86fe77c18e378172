"""Result containers of the path: named tensors + a pandas ``infos`` frame.

API-compatible with the reference's ``TensorCollection`` /
``PandasTensorCollection`` / ``concatenate`` / ``filter_top_pose_estimates``
(``TB/utils/tensor_collection.py:28-230``): tensors are reachable as attributes,
``coll[ids]`` indexes every tensor and ``infos.iloc`` together and re-numbers the
index, ``len(coll) == len(coll.infos)``.  The file-based multi-rank gather of the
reference (``:166-187``) is replaced by ``happypose_amd.distributed.gather_poses``
(one RCCL all-gather); ``gather_distributed`` is kept as a thin wrapper.
"""

from __future__ import annotations

from typing import Dict, List

import pandas as pd
import torch


class TensorCollection:
    def __init__(self, **tensors):
        object.__setattr__(self, "_tensors", {})
        for name, value in tensors.items():
            self.register_tensor(name, value)

    # -- registry -----------------------------------------------------------------
    def register_tensor(self, name: str, tensor) -> None:
        self._tensors[name] = tensor

    def delete_tensor(self, name: str) -> None:
        del self._tensors[name]

    @property
    def tensors(self) -> Dict[str, torch.Tensor]:
        return self._tensors

    @property
    def device(self):
        return next(iter(self._tensors.values())).device

    # -- attribute access ---------------------------------------------------------
    def __getattr__(self, name):
        tensors = self.__dict__.get("_tensors")
        if tensors is not None and name in tensors:
            return tensors[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if "_tensors" not in self.__dict__:
            raise ValueError("Please call __init__")
        if name in self._tensors:
            self._tensors[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getitem__(self, ids):
        return TensorCollection(**{k: v[ids] for k, v in self._tensors.items()})

    def __repr__(self):
        rows = "".join(f"    {k}: {tuple(t.shape)} {t.dtype} {t.device},\n" for k, t in self._tensors.items())
        return f"{type(self).__name__}(\n{rows})"

    # -- pickling -----------------------------------------------------------------
    def __getstate__(self):
        return {"tensors": self._tensors}

    def __setstate__(self, state):
        self.__init__(**state["tensors"])

    # -- device / dtype -----------------------------------------------------------
    def to(self, torch_attr):
        for k, v in self._tensors.items():
            self._tensors[k] = v.to(torch_attr)
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def float(self):
        return self.to(torch.float)

    def double(self):
        return self.to(torch.double)

    def half(self):
        return self.to(torch.half)

    def clone(self):
        return TensorCollection(**{k: v.clone() for k, v in self._tensors.items()})


class PandasTensorCollection(TensorCollection):
    def __init__(self, infos: pd.DataFrame, **tensors):
        super().__init__(**tensors)
        self.infos = infos.reset_index(drop=True)
        self.meta = {}

    def __len__(self):
        return len(self.infos)

    def __getitem__(self, ids):
        if isinstance(ids, torch.Tensor):
            pos = ids.cpu().numpy()
        else:
            pos = ids
        infos = self.infos.iloc[pos].reset_index(drop=True)
        return PandasTensorCollection(infos, **{k: v[ids] for k, v in self._tensors.items()})

    def merge_df(self, df, *args, **kwargs):
        infos = self.infos.merge(df, how="left", *args, **kwargs)
        assert len(infos) == len(self.infos)
        return PandasTensorCollection(infos=infos, **self._tensors)

    def clone(self):
        return PandasTensorCollection(self.infos.copy(), **{k: v.clone() for k, v in self._tensors.items()})

    def __repr__(self):
        rows = "".join(f"    {k}: {tuple(t.shape)} {t.dtype} {t.device},\n" for k, t in self._tensors.items())
        return f"{type(self).__name__}(\n{rows}{'-' * 40}\n    infos:\n{self.infos!r}\n)"

    def __getstate__(self):
        state = super().__getstate__()
        state["infos"] = self.infos
        state["meta"] = self.meta
        return state

    def __setstate__(self, state):
        self.__init__(state["infos"], **state["tensors"])
        self.meta = state["meta"]

    def gather_distributed(self, tmp_dir=None):
        """Reference signature (``TB/utils/tensor_collection.py:166-187``); the rank
        files are gone -- tensors travel through one all-gather."""
        from .distributed import gather_collection

        return gather_collection(self)


def concatenate(datas: List[PandasTensorCollection]) -> PandasTensorCollection:
    """``TB/utils/tensor_collection.py:28-42``: drop empty parts, stack the rest."""
    datas = [d for d in datas if len(d) > 0]
    if not datas:
        return PandasTensorCollection(infos=pd.DataFrame())
    assert all(type(d) is type(datas[0]) for d in datas)
    infos = pd.concat([d.infos for d in datas], axis=0, sort=False).reset_index(drop=True)
    tensors = {k: torch.cat([getattr(d, k) for d in datas], dim=0) for k in datas[0].tensors}
    return PandasTensorCollection(infos=infos, **tensors)


def filter_top_pose_estimates(data_TCO: PandasTensorCollection, top_K: int, group_cols: List[str],
                              filter_field: str, ascending: bool = False) -> PandasTensorCollection:
    """Keep the ``top_K`` rows per group ranked by ``filter_field``; the output is in
    sorted (not input) order, exactly like the reference's
    ``sort_values().groupby().head()`` (``TB/utils/tensor_collection.py:201-230``,
    pinned by golden G8 including the tie case)."""
    df = data_TCO.infos
    kept = df.sort_values(filter_field, ascending=ascending).groupby(group_cols).head(top_K)
    return data_TCO[kept.index.tolist()]
