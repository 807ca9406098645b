"""Mongo-style filter evaluation over document tags -> row offsets.

The reference translates the same filter dicts to SQL over a SQLite cell table
(annlite/filter.py:1-100, annlite/storage/table.py) and hands the resulting offsets to the index
as ``indices`` (annlite/container.py:107-120).  Storage is out of scope here (SURVEY.md section 2
rows 19-20); this is the small in-memory equivalent that feeds the GPU validity bitmap
(section 8f-3).  Supported: $eq $ne $gt $gte $lt $lte $in $nin and the logical $and / $or / $not.
"""
from typing import Any, Dict, List

_CMP = {
    '$eq': lambda a, b: a == b,
    '$ne': lambda a, b: a != b,
    '$gt': lambda a, b: a is not None and a > b,
    '$gte': lambda a, b: a is not None and a >= b,
    '$lt': lambda a, b: a is not None and a < b,
    '$lte': lambda a, b: a is not None and a <= b,
    '$in': lambda a, b: a in b,
    '$nin': lambda a, b: a not in b,
}
LOGICAL = ('$and', '$or', '$not')


def match(tags: Dict[str, Any], flt: Dict) -> bool:
    for key, cond in flt.items():
        if key == '$and':
            if not all(match(tags, c) for c in cond):
                return False
        elif key == '$or':
            if not any(match(tags, c) for c in cond):
                return False
        elif key == '$not':
            if match(tags, cond):
                return False
        elif key.startswith('$'):
            raise ValueError(f'The operator {key} is not supported')
        else:
            val = tags.get(key)
            if isinstance(cond, dict):
                for op, ref in cond.items():
                    if op not in _CMP:
                        raise ValueError(f'The operator {op} is not supported')
                    if not _CMP[op](val, ref):
                        return False
            elif val != cond:
                return False
    return True


def select(all_tags: List[Dict[str, Any]], flt: Dict) -> List[int]:
    return [i for i, t in enumerate(all_tags) if t is not None and match(t, flt)]
