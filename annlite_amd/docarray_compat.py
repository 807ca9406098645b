"""Minimal ``Document`` / ``DocumentArray`` stand-ins used ONLY when the real ``docarray`` package
is not importable (it is absent from this image and cannot be installed offline, SURVEY.md
section 7 "docarray is not installable here").  They expose exactly what the hot path touches:
``.id .embedding .tags .scores[name].value .matches`` and ``DocumentArray.embeddings`` /
``docs[:, 'id']`` (annlite/container.py:226-233, annlite/index.py:352-359, 540).
With docarray installed the real classes are used and this module is never imported.
"""
import uuid
from collections import defaultdict
from typing import Iterable, List, Optional

import numpy as np


class NamedScore:
    __slots__ = ('value', 'op_name', 'description', 'ref_id')

    def __init__(self, value=None):
        self.value = value
        self.op_name = ''
        self.description = ''
        self.ref_id = ''

    def __repr__(self):
        return f'NamedScore(value={self.value!r})'


class Document:
    def __init__(self, id: Optional[str] = None, embedding=None, tags: Optional[dict] = None, text: str = '', **kwargs):
        self.id = id if id is not None else uuid.uuid4().hex
        self.embedding = embedding
        self.tags = dict(tags) if tags else {}
        self.text = text
        self.scores = defaultdict(NamedScore)
        self.matches = DocumentArray()
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __repr__(self):
        return f'<Document id={self.id!r}>'


class DocumentArray(list):
    def __init__(self, docs: Optional[Iterable[Document]] = None):
        super().__init__(docs if docs is not None else [])

    @property
    def embeddings(self):
        if len(self) == 0:
            return None
        return np.stack([np.asarray(d.embedding) for d in self])

    @embeddings.setter
    def embeddings(self, value):
        for d, e in zip(self, value):
            d.embedding = e

    def __getitem__(self, item):
        if isinstance(item, tuple) and len(item) == 2 and isinstance(item[1], str):
            sel = self[item[0]]
            sel = sel if isinstance(sel, list) else [sel]
            return [getattr(d, item[1]) for d in sel]
        if isinstance(item, slice):
            return DocumentArray(list.__getitem__(self, item))
        if isinstance(item, (list, np.ndarray)):
            return DocumentArray([list.__getitem__(self, int(i)) for i in item])
        if isinstance(item, str):
            for d in self:
                if d.id == item:
                    return d
            raise KeyError(item)
        return list.__getitem__(self, item)

    def __contains__(self, item):
        if isinstance(item, str):
            return any(d.id == item for d in self)
        return list.__contains__(self, item)


def to_numpy_array(value) -> np.ndarray:
    """docarray.math.ndarray.to_numpy_array for the array kinds we meet."""
    if value is None:
        raise ValueError('documents have no embeddings')
    if hasattr(value, 'detach'):
        value = value.detach().cpu().numpy()
    return np.asarray(value)
