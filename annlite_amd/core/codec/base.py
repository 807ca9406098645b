"""Codec base class -- mirrors annlite/core/codec/base.py:9-38 (is_trained, pickle dump/load)."""
import pickle
from abc import ABC, abstractmethod
from pathlib import Path


class BaseCodec(ABC):
    def __init__(self, require_train: bool = True):
        self.require_train = require_train
        self._is_trained = False if require_train else True

    @abstractmethod
    def fit(self, *args, **kwargs):
        pass

    @abstractmethod
    def encode(self):
        pass

    @abstractmethod
    def decode(self):
        pass

    def dump(self, target_path: 'Path'):
        """pickle protocol 4, like the reference (codec/base.py:26-27)."""
        with Path(target_path).open('wb') as f:
            pickle.dump(self, f, protocol=4)

    @staticmethod
    def load(from_path: 'Path'):
        with Path(from_path).open('rb') as f:
            return pickle.load(f)

    @property
    def is_trained(self):
        return self._is_trained

    def _check_trained(self):
        assert self.is_trained is True, f'{self.__class__.__name__} requires training'
