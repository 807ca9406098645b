"""Minimal ``Document`` / ``DocumentArray`` stand-ins used ONLY when the real ``docarray`` package
is not importable (it is absent from this image and cannot be installed offline, SURVEY.md
section 7 "docarray is not installable here").  They expose exactly what the hot path touches:
``.id .embedding .tags .scores[name].value .matches`` and ``DocumentArray.embeddings`` /
``docs[:, 'id']`` (annlite/container.py:226-233, annlite/index.py:352-359, 540).
With docarray installed the real classes are used and this module is never imported.
"""
import threading
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
    """id / embedding / tags / text / scores / matches.  ``scores`` (a ``defaultdict(NamedScore)``) and ``matches`` (a
    ``DocumentArray``) are created when first touched: a search result is ten thousand ``Document(id)`` objects per 1024-query
    batch (annlite/container.py:226-233), most of which are only ever asked for their id and one score."""
    __slots__ = ('id', 'embedding', 'tags', 'text', '_scores', '_matches', '_score1', '__dict__')

    def __init__(self, id: Optional[str] = None, embedding=None, tags: Optional[dict] = None, text: str = '', **kwargs):
        self.id = id if id is not None else uuid.uuid4().hex
        self.embedding = embedding
        self.tags = dict(tags) if tags else {}
        self.text = text
        self._scores = None
        self._matches = None
        self._score1 = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def match(cls, id, score_name: str, value, embedding=None, tags=None) -> 'Document':
        """A search match: ``Document(id)`` with ``scores[score_name].value = value`` (the score object itself is made when
        ``scores`` is first read)."""
        d = cls.__new__(cls)
        d.id, d.embedding, d.tags, d.text = id, embedding, (dict(tags) if tags else {}), ''
        d._scores, d._matches, d._score1 = None, None, (score_name, value)
        return d

    @property
    def scores(self):
        sc = self._scores
        if sc is None:
            sc = self._scores = defaultdict(NamedScore)
            if self._score1 is not None:
                sc[self._score1[0]].value = self._score1[1]
                self._score1 = None
        return sc

    @scores.setter
    def scores(self, value):
        self._scores, self._score1 = value, None

    @property
    def matches(self):
        if self._matches is None:
            self._matches = DocumentArray()
        return self._matches

    @matches.setter
    def matches(self, value):
        self._matches = value

    def __getstate__(self):
        st = dict(self.__dict__)
        st.update(id=self.id, embedding=self.embedding, tags=self.tags, text=self.text, scores=dict(self.scores),
                  matches=list(self.matches) if self._matches is not None else [])
        return st

    def __setstate__(self, st):
        sc = st.pop('scores', {})
        ms = st.pop('matches', [])
        self.id, self.embedding, self.tags, self.text = st.pop('id'), st.pop('embedding', None), st.pop('tags', {}), st.pop('text', '')
        self._scores, self._matches, self._score1 = None, None, None
        for k, v in st.items():
            setattr(self, k, v)
        for k, v in sc.items():
            self.scores[k] = v
        if ms:
            self._matches = DocumentArray(ms)

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


# (one lock for all lists: materialisation is microseconds of python per list, and a per-instance lock would cost 1024 lock objects
# per batch on the path this class exists to keep cheap)
_MATERIALISE_LOCK = threading.RLock()


class LazyMatches(DocumentArray):
    """``doc.matches`` of a search result, materialised on first use.

    The reference builds ``Document(id=doc_id)`` + ``doc.scores[metric].value = dist`` for every match of every query in a
    python loop (annlite/container.py:226-233): 10 240 objects per 1024-query batch, tens of milliseconds next to a 1.4 ms
    scan.  Here a query's matches hold the row of (offset, distance) arrays the GPU returned and a resolver; the
    ``Document`` objects -- same ``.id``, ``.scores[metric].value``, metadata -- are built when the list is first read
    (indexing, iteration, ``in``, comparison, mutation); ``len()`` answers without building anything.

    Known limit of a ``list`` subclass with deferred storage: CPython routines that read a list's item array directly
    (``[] + m``, ``lst[a:b] = m``, ``np.array(m)`` -- the ``PySequence_Fast`` users) bypass the overrides and see an empty list
    until the first python-level access; call ``list(m)`` (or touch ``m[0]``) first.  ``AnnLite.search`` with the real docarray
    never builds these (eager matches, as the reference).
    """

    def __init__(self, offsets, dists, resolve):
        list.__init__(self)
        self._pending = (offsets, dists, resolve)

    def _ensure(self):
        if self.__dict__.get('_pending') is not None:
            # ONE thread materialises, under the lock: check -> extend -> clear is not atomic otherwise (two readers could both pass
            # the check and both extend: every match twice).  A reader that loses the race waits here and finds `_pending` cleared.
            with _MATERIALISE_LOCK:
                p = self.__dict__.get('_pending')
                if p is not None:
                    offsets, dists, resolve = p
                    list.extend(self, resolve(offsets, dists))  # storage first, the flag afterwards
                    self._pending = None
        return self

    @property
    def materialised(self) -> bool:
        return self.__dict__.get('_pending') is None

    def __len__(self):
        p = self.__dict__.get('_pending')
        return len(p[0]) if p is not None else list.__len__(self)

    def __bool__(self):
        return len(self) > 0

    def __iter__(self):
        return list.__iter__(self._ensure())

    def __reversed__(self):
        return list.__reversed__(self._ensure())

    def __getitem__(self, item):
        self._ensure()
        return DocumentArray.__getitem__(self, item)

    def __contains__(self, item):
        self._ensure()
        return DocumentArray.__contains__(self, item)

    def __eq__(self, other):
        if isinstance(other, LazyMatches):
            other._ensure()
        return list.__eq__(self._ensure(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return list.__repr__(self._ensure())

    def __reduce_ex__(self, protocol):  # pickles / copies as a plain DocumentArray of the materialised documents
        return (DocumentArray, (list(self._ensure()),))


def _lazy_mutator(name):
    fn = getattr(list, name)

    def method(self, *a, **kw):
        return fn(self._ensure(), *a, **kw)

    method.__name__ = name
    return method


for _n in ('append', 'extend', 'insert', 'pop', 'remove', 'sort', 'reverse', 'index', 'count', 'copy', 'clear', '__setitem__',
           '__delitem__', '__add__', '__iadd__', '__mul__', '__imul__', '__lt__', '__le__', '__gt__', '__ge__'):
    setattr(LazyMatches, _n, _lazy_mutator(_n))
del _n


def to_numpy_array(value) -> np.ndarray:
    """docarray.math.ndarray.to_numpy_array for the array kinds we meet."""
    if value is None:
        raise ValueError('documents have no embeddings')
    if hasattr(value, 'detach'):
        value = value.detach().cpu().numpy()
    return np.asarray(value)
