"""What every codec of the path shares (the reference's contract, annlite/core/codec/base.py): a trained flag
that training flips, ``_check_trained`` in front of every use, and persistence as ONE pickle (protocol 4) that
``AnnLite`` keeps under ``data_path/parameters-<md5>/``."""
import abc
import pickle
from pathlib import Path

_PICKLE_PROTOCOL = 4


class BaseCodec(abc.ABC):
    def __init__(self, require_train: bool = True):
        self.require_train = bool(require_train)
        self._is_trained = not self.require_train  # a codec that needs no training is born trained

    # ---- state -------------------------------------------------------------------------------------
    is_trained = property(lambda self: self._is_trained, doc='True once fit / build_codebook has produced codebooks')

    def _check_trained(self):
        if self._is_trained is not True:
            raise AssertionError(f'{type(self).__name__} requires training')

    # ---- persistence -------------------------------------------------------------------------------
    def dump(self, target_path):
        Path(target_path).write_bytes(pickle.dumps(self, protocol=_PICKLE_PROTOCOL))

    @staticmethod
    def load(from_path):
        return pickle.loads(Path(from_path).read_bytes())

    # ---- what a codec has to provide -----------------------------------------------------------------
    @abc.abstractmethod
    def fit(self, *args, **kwargs):
        """learn the codebooks from training vectors"""

    @abc.abstractmethod
    def encode(self, *args, **kwargs):
        """vectors -> codes"""

    @abc.abstractmethod
    def decode(self, *args, **kwargs):
        """codes -> (approximate) vectors"""
