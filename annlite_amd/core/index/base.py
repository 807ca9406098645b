"""Index plugin base class -- mirrors annlite/core/index/base.py:10-57 (constructor arguments,
``capacity`` / ``size`` properties, abstract add/delete/update, ``reset``)."""
import abc
from typing import List, Optional, Union

import numpy as np

from ...enums import ExpandMode, Metric


def str2dtype(dtype_str: str):
    """annlite/helper.py:24-47 (subset that exists in numpy 2)."""
    table = {
        'double': np.float64, 'float64': np.float64, 'half': np.float16, 'float16': np.float16,
        'float': np.float32, 'float32': np.float32, 'long': np.int64, 'int64': np.int64,
        'int': np.int32, 'int32': np.int32, 'int16': np.int16, 'int8': np.int8, 'uint8': np.uint8, 'bool': np.bool_,
    }
    if dtype_str not in table:
        raise TypeError(f'Unrecognized dtype string: {dtype_str}')
    return table[dtype_str]


class BaseIndex(abc.ABC):
    def __init__(
        self,
        dim: int,
        dtype: Union[np.dtype, str] = np.float32,
        metric: Metric = Metric.COSINE,
        initial_size: Optional[int] = None,
        expand_step_size: int = 10240,
        expand_mode: ExpandMode = ExpandMode.STEP,
        *args,
        **kwargs,
    ):
        assert expand_step_size > 0
        self.initial_size = initial_size or expand_step_size
        self.expand_step_size = expand_step_size
        self.expand_mode = expand_mode
        self.dim = dim
        self.dtype = str2dtype(dtype) if isinstance(dtype, str) else dtype
        self.metric = metric
        self._size = 0
        self._capacity = self.initial_size

    @property
    def capacity(self) -> int:
        return self._capacity

    @property
    def size(self):
        return self._size

    @abc.abstractmethod
    def add_with_ids(self, x: np.ndarray, ids: List[int], **kwargs):
        ...

    @abc.abstractmethod
    def delete(self, ids: List[int]):
        ...

    @abc.abstractmethod
    def update_with_ids(self, x: np.ndarray, ids: List[int], **kwargs):
        ...

    def reset(self, capacity: Optional[int] = None):
        self._size = 0
        self._capacity = capacity or self.initial_size
