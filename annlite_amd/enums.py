"""``Metric`` and ``ExpandMode``: the names and integer values the reference uses (annlite/enums.py) -- the
integers are also the metric ids of the C ABI (``ANNLITE_METRIC_*`` in include/annlite_hip.h), so a ``Metric``
can be handed to the library as is."""
import enum


class _ByName(enum.IntEnum):
    """Integer enum that prints as its bare member name and can be looked up by a case-insensitive name."""

    def __str__(self) -> str:
        return self.name

    @classmethod
    def from_string(cls, s: str):
        key = str(s).strip().upper()
        member = cls.__members__.get(key)
        if member is None:
            raise ValueError(f'{key} is not a valid enum for {cls!r}, must be one of {list(cls)}')
        return member


# (functional API; `module` keeps the members picklable -- codecs are pickled with their metric)
Metric = _ByName('Metric', (('EUCLIDEAN', 1), ('INNER_PRODUCT', 2), ('COSINE', 3)), module=__name__)
ExpandMode = _ByName('ExpandMode', (('STEP', 1), ('DOUBLE', 2), ('ADAPTIVE', 3)), module=__name__)
BetterEnum = _ByName  # the reference's name of the base
