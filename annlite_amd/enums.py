"""Metric / ExpandMode enums -- same names and integer values as the reference
(annlite/enums.py:4-34); the integers double as the C ABI metric ids (include/annlite_hip.h)."""
from enum import IntEnum


class BetterEnum(IntEnum):
    def __str__(self):
        return self.name

    @classmethod
    def from_string(cls, s: str):
        try:
            return cls[s.upper()]
        except KeyError:
            raise ValueError(f'{s.upper()} is not a valid enum for {cls!r}, must be one of {list(cls)}')


class Metric(BetterEnum):
    EUCLIDEAN = 1
    INNER_PRODUCT = 2
    COSINE = 3


class ExpandMode(BetterEnum):
    STEP = 1
    DOUBLE = 2
    ADAPTIVE = 3
