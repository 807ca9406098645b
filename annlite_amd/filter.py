"""Filter dicts over document tags -> row offsets for the GPU validity bitmap.

The reference compiles the same dicts to an SQL WHERE clause over a SQLite cell table (annlite/filter.py,
annlite/storage/table.py) and hands the selected offsets to the index as ``indices`` (container.py:107-120).  Storage
is out of scope here; this module evaluates the reference's filter GRAMMAR directly on
the in-memory tags, with the semantics the generated SQL has:

  * a dict is a sequence of conditions joined by the enclosing logic (AND at the top); ``{'$and': X}`` / ``{'$or': X}``
    evaluate ``X`` with that logic and are joined to what precedes them by that same operator;
  * ``field: {op: value, ...}`` -- several operators on one field are AND-ed; ``$lt $gt $lte $gte $eq $neq``,
    membership ``$in $nin``, or a logical operator whose value is evaluated with that logic;
  * a list is a parenthesised group of its elements joined by the enclosing logic;
  * conditions are spliced into ONE flat clause, so SQL precedence applies: AND binds tighter than OR;
  * a document without the tag (SQL NULL) satisfies no comparison, ``$nin`` and ``$neq`` included;
  * unknown ``$`` operators raise ``ValueError`` like the reference's parser.
Extensions kept from this build's earlier filter: ``$ne`` (alias of ``$neq``), ``$not`` and a bare value as equality.
"""
from typing import Any, Dict, List

_LOGIC = {'$and': 'AND', '$or': 'OR'}
_COMPARE = {
    '$lt': lambda a, b: a < b,
    '$gt': lambda a, b: a > b,
    '$lte': lambda a, b: a <= b,
    '$gte': lambda a, b: a >= b,
    '$eq': lambda a, b: a == b,
    '$neq': lambda a, b: a != b,
    '$ne': lambda a, b: a != b,
}
_MEMBER = {'$in': lambda a, b: a in b, '$nin': lambda a, b: a not in b}
LOGICAL = ('$and', '$or', '$not')


def _unsupported(op):
    return ValueError(f'The operator {op} is not supported yet, please double check the given filters!')


def _reduce(tokens: List) -> bool:
    """a flat clause ``t0 AND t1 OR t2 ...`` under SQL precedence; the empty clause selects everything"""
    if not tokens:
        return True
    result, group = False, True  # OR over groups, AND inside a group
    for tok in tokens:
        if tok == 'OR':
            result, group = result or group, True
        elif tok != 'AND':
            group = group and bool(tok)
    return result or group


def _compare(val, op, ref) -> bool:
    if val is None:  # NULL never compares true
        return False
    try:
        return bool((_COMPARE.get(op) or _MEMBER[op])(val, ref))
    except TypeError:  # SQLite orders mixed types instead of failing; an in-memory store just says no
        return False


def _tokens(tags: Dict[str, Any], data, logic: str = 'AND') -> List:
    out: List = []
    if isinstance(data, dict):
        for i, (key, value) in enumerate(data.items()):
            if key in _LOGIC:
                if i > 0:
                    out.append(_LOGIC[key])
                out.extend(_tokens(tags, value, _LOGIC[key]))
            elif key == '$not':
                if i > 0:
                    out.append(logic)
                out.append(not _reduce(_tokens(tags, value)))
            elif key.startswith('$'):
                raise _unsupported(key)
            else:
                if i > 0:
                    out.append(logic)
                if not isinstance(value, dict):
                    out.append(_compare(tags.get(key), '$eq', value))
                    continue
                if not value:
                    raise ValueError(f'The query express is illegal: {data}')
                parts: List = []
                for op, ref in value.items():
                    if parts:
                        parts.append('AND')
                    if op in _LOGIC:
                        parts.extend(_tokens(tags, ref, _LOGIC[op]))
                    elif op in _COMPARE or op in _MEMBER:
                        parts.append(_compare(tags.get(key), op, ref))
                    else:
                        raise _unsupported(op)
                out.extend(parts)
    elif isinstance(data, (list, tuple)):
        inner: List = []
        for d in data:
            if inner:
                inner.append(logic)
            inner.extend(_tokens(tags, d))
        out.append(_reduce(inner))  # (the parenthesised group is ONE operand)
    else:
        raise ValueError(f'The query express is illegal: {data}')
    return out


def match(tags: Dict[str, Any], flt: Dict) -> bool:
    return _reduce(_tokens(tags, flt or {}))


def select(all_tags: List[Dict[str, Any]], flt: Dict) -> List[int]:
    flt = flt or {}
    _tokens({}, flt)  # malformed filters raise even when the table is empty (the reference parses before it queries)
    return [i for i, t in enumerate(all_tags) if t is not None and match(t, flt)]
