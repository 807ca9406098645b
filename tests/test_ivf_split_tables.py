"""Host logic of IvfPQGpuIndex.rerank_split (no GPU): the cell table that lists every cell ALSO as S contiguous parts -- what
annlite_ivf_search_candidates is handed when the nearest cells are probed in parts.  The parts of a cell must tile
its row range exactly, begin at multiples of 64 rows like the cells themselves (the scan's work items start on 64-row blocks), and the
order array must list every entry, longest first."""
import types

import numpy as np
import pytest
import torch

from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex


def _fake(counts):
    counts = torch.tensor(counts, dtype=torch.int64)
    padded = (counts + 63) // 64 * 64
    begin = torch.cumsum(padded, 0) - padded
    o = types.SimpleNamespace(_split_cache=None, _cell_rows=torch.stack([begin, begin + counts], dim=1).contiguous())
    return o, counts, begin


@pytest.mark.parametrize('S', [2, 3, 4, 8])
def test_parts_tile_their_cell(S):
    rs = np.random.RandomState(S)
    counts = [0, 1, 63, 64, 65, 127, 128, 129, 200, 257, 1000, 39063] + rs.randint(0, 5000, 40).tolist()
    o, counts, begin = _fake(counts)
    rows, order = IvfPQGpuIndex._split_tables(o, S)
    C = counts.numel()
    assert rows.shape == (C * (1 + S), 2) and torch.equal(rows[:C], o._cell_rows)
    assert sorted(order.tolist()) == list(range(C * (1 + S)))
    ln = (rows[:, 1] - rows[:, 0])[order.long()]
    assert (ln[1:] <= ln[:-1]).all()
    for c in range(C):
        p = rows[C + c * S:C + (c + 1) * S]
        n = p[:, 1] - p[:, 0]
        assert (n >= 0).all() and int(n.sum()) == int(counts[c])
        assert (p[:, 0] % 64 == 0).all()
        ne = p[n > 0]
        if ne.shape[0]:
            assert int(ne[0, 0]) == int(begin[c]) and int(ne[-1, 1]) == int(begin[c] + counts[c])
            assert torch.equal(ne[1:, 0], ne[:-1, 1])  # contiguous, in order
            assert (n[:ne.shape[0]] > 0).all()        # the empty parts are the LAST ones
        else:
            assert int(counts[c]) == 0
        # balanced up to the 64-row granularity
        if int(counts[c]) >= 64 * S:
            assert int(n.max()) - int(n[n > 0].min()) <= 64 * S


def test_split_tables_are_cached_per_S_and_dropped_by_a_new_seal():
    o, _, _ = _fake([100, 200, 300])
    a = IvfPQGpuIndex._split_tables(o, 4)
    assert IvfPQGpuIndex._split_tables(o, 4)[0] is a[0]
    b = IvfPQGpuIndex._split_tables(o, 2)
    assert b[0].shape[0] == 3 * 3 and o._split_cache[0] == 2
    o._split_cache = None  # what _seal does
    assert IvfPQGpuIndex._split_tables(o, 4)[0] is not a[0]
