"""The plugin contract ``CellContainer`` programs against (annlite/core/index/base.py): construction from
``dim / dtype / metric / initial_size / expand_step_size / expand_mode``, ``capacity`` and ``size``, the three
mutators every index implements, ``reset``.  Growth policy: capacity starts at ``initial_size`` (or one expansion
step) and the concrete index enlarges it in multiples of ``expand_step_size``."""
import abc
from typing import List, Optional, Union

import numpy as np

from ...enums import ExpandMode, Metric

# dtype names accepted for ``dtype=`` (what annlite/helper.py understands, minus the aliases numpy 2 dropped)
_DTYPES = {name: getattr(np, name) for name in ('float64', 'float32', 'float16', 'int64', 'int32', 'int16', 'int8', 'uint8')}
_DTYPES.update(double=np.float64, float=np.float32, half=np.float16, long=np.int64, int=np.int32, bool=np.bool_)


def str2dtype(dtype_str: str):
    try:
        return _DTYPES[dtype_str]
    except KeyError:
        raise TypeError(f'Unrecognized dtype string: {dtype_str}') from None


class BaseIndex(abc.ABC):
    def __init__(self, dim: int, dtype: Union[np.dtype, str] = np.float32, metric: Metric = Metric.COSINE,
                 initial_size: Optional[int] = None, expand_step_size: int = 10240,
                 expand_mode: ExpandMode = ExpandMode.STEP, *args, **kwargs):
        if not expand_step_size > 0:
            raise AssertionError('expand_step_size must be positive')
        self.dim, self.metric = dim, metric
        self.dtype = str2dtype(dtype) if isinstance(dtype, str) else dtype
        self.expand_step_size, self.expand_mode = expand_step_size, expand_mode
        self.initial_size = initial_size if initial_size else expand_step_size
        self._capacity, self._size = self.initial_size, 0

    capacity = property(lambda self: self._capacity, doc='rows the index can hold before it has to grow')

    @property
    def size(self):
        return self._size

    def reset(self, capacity: Optional[int] = None):
        self._capacity, self._size = (capacity if capacity else self.initial_size), 0

    @abc.abstractmethod
    def add_with_ids(self, x: np.ndarray, ids: List[int], **kwargs):
        """store ``x[i]`` under offset ``ids[i]``"""

    @abc.abstractmethod
    def update_with_ids(self, x: np.ndarray, ids: List[int], **kwargs):
        """overwrite the rows of ``ids``"""

    @abc.abstractmethod
    def delete(self, ids: List[int]):
        """rows of ``ids`` are never returned again"""
