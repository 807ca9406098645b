"""The pruned search over cells on the BYTE-TABLE kernel (annlite_ivf_search_topk, round 6): cell tiles of 32 (query, cell)
pairs, exact sums inside the tile, bounds shared by query, one merge launch.  Pinned against the oracle's restatement of
`CellContainer.ivf_search` (annlite/container.py:88-144) over the probed cells -- bit-exact ids and distances -- and against
the u16 tile scan + re-score it replaces (`IvfPQGpuIndex.byte_tiles = False`)."""
import numpy as np
import pytest

from conftest import has_gpu
from test_ivf import _build, _check_against_oracle, _data

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]


def _both(idx, q, k, P, indices=None):
    idx.byte_tiles = True
    d1, i1 = idx.search_batch(q, limit=k, n_probe=P, indices=indices)
    idx.byte_tiles = False
    d0, i0 = idx.search_batch(q, limit=k, n_probe=P, indices=indices)
    idx.byte_tiles = True
    assert np.array_equal(i1, i0)
    assert np.array_equal(d1, d0)
    return d1, i1


@pytest.mark.parametrize('D,C,P,B,k,N', [
    (64, 32, 4, 100, 10, 20000),
    (128, 64, 16, 1024, 10, 60000),   # the bench's probe count; a full 1024-query batch: every tile shape
    (128, 16, 15, 257, 16, 30000),    # all but one cell; the largest k; a ragged batch
    (64, 8, 1, 3, 1, 5000),           # a handful of queries, one probe, k = 1
    (32, 40, 7, 33, 5, 9000),         # 2-float sub-vectors do not take this path (dsub % 4): the fallback answers
    (256, 32, 8, 129, 10, 12000),     # the widest vectors the fused build takes
    (64, 32, 20, 1000, 10, 40000),    # more than 16384 (query, cell) pairs: the plan's loops through memory
], ids=lambda v: str(v))
def test_byte_tiles_equal_oracle_and_u16_path(oracle, D, C, P, B, k, N):
    from annlite_amd import Metric

    idx, codec, vq, x = _build(N, D, 16, C, Metric.EUCLIDEAN, seed=3)
    _, q = _data(np.random.RandomState(4), 1, D, B)
    d, i = _both(idx, q, k, P)
    _check_against_oracle(oracle, idx, codec, q, k, P, d, i)
    assert (np.diff(d, axis=1) >= 0).all()


def test_byte_tiles_is_the_path_taken(monkeypatch):
    """the index really calls annlite_ivf_search_topk for M = 16 / L2 / k <= 16 (and not for the shapes it does not serve)"""
    from annlite_amd import Metric, ops

    calls = []
    real = ops.ivf_search_topk
    monkeypatch.setattr(ops, 'ivf_search_topk', lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    idx, codec, vq, x = _build(6000, 64, 16, 16, Metric.EUCLIDEAN, seed=5)
    _, q = _data(np.random.RandomState(6), 1, 64, 20)
    idx.search_batch(q, limit=10, n_probe=4)
    assert len(calls) == 1
    idx.search_batch(q, limit=20, n_probe=4)  # k > 16: the u16 tile scan
    assert len(calls) == 1
    idx2, *_ = _build(6000, 64, 16, 16, Metric.COSINE, seed=5)
    idx2.search_batch(q, limit=10, n_probe=4)  # cosine: inner-product tables, built in the same launch
    assert len(calls) == 2
    idx3, *_ = _build(6000, 64, 8, 16, Metric.EUCLIDEAN, seed=5)
    idx3.search_batch(q, limit=10, n_probe=4)  # M = 8: the u16 tile scan
    assert len(calls) == 2


@pytest.mark.parametrize('metric_name,D,C,P,B,k', [
    ('COSINE', 64, 32, 4, 200, 10),
    ('INNER_PRODUCT', 128, 16, 5, 77, 16),
    ('COSINE', 128, 64, 16, 1024, 10),
], ids=lambda v: str(v))
def test_byte_tiles_inner_product_tables(oracle, metric_name, D, C, P, B, k):
    """COSINE (the reference's default metric) and INNER_PRODUCT: float32(1 / Ks) - <q_sub, codeword> tables (pq.py:316-322) built inside
    the preparation launch -- negative entries, the byte tables' minima / ranges as for L2"""
    from annlite_amd import Metric

    idx, codec, vq, x = _build(30000, D, 16, C, Metric[metric_name], seed=9)
    _, q = _data(np.random.RandomState(10), 1, D, B)
    d, i = _both(idx, q, k, P)
    _check_against_oracle(oracle, idx, codec, q, k, P, d, i)


def test_byte_tiles_deletes_filter_and_ties(oracle):
    from annlite_amd import Metric

    rng = np.random.RandomState(21)
    base, q = _data(rng, 400, 64, 40)
    x = np.repeat(base, 25, axis=0)  # every vector 25 times: exact distance ties inside and ACROSS cells' lists
    idx, codec, vq, _ = _build(x.shape[0], 64, 16, 12, Metric.EUCLIDEAN, seed=21, x=x)
    for P in (1, 3, 11):
        d, i = _both(idx, q, 10, P)
        _check_against_oracle(oracle, idx, codec, q, 10, P, d, i)
    dead = rng.choice(x.shape[0], 3000, replace=False)
    idx.delete(dead)
    d, i = _both(idx, q, 10, 3)
    _check_against_oracle(oracle, idx, codec, q, 10, 3, d, i)
    assert not np.isin(i, dead).any()
    keep = rng.choice(x.shape[0], 2500, replace=False)  # an `indices` filter: most cells keep only a few rows
    d, i = _both(idx, q, 10, 3, indices=keep)
    _check_against_oracle(oracle, idx, codec, q, 10, 3, d, i, indices=keep)
    assert np.isin(i[i >= 0], keep).all()


def test_byte_tiles_ties_across_cells_break_by_id(oracle):
    """the SAME vectors stored under ids that interleave two cells: a distance tie between rows of different cells must be
    broken by the external id, not by the position in the cell-sorted table"""
    import torch

    from annlite_amd import Metric

    rng = np.random.RandomState(31)
    base, q = _data(rng, 600, 64, 32)
    x = np.concatenate([base, base], axis=0)
    idx, codec, vq, _ = _build(x.shape[0], 64, 16, 8, Metric.EUCLIDEAN, seed=31, x=x)
    # move every second copy to another cell by hand: equal codes, equal distances, different cells
    cells = idx._cell_of[: x.shape[0]].clone()
    second = torch.arange(600, 1200, device=cells.device)
    cells[second[::2]] = (cells[second[::2]] + 1) % 8
    idx._cell_of[: x.shape[0]] = cells
    idx._sealed = False
    for P in (2, 7):
        d, i = _both(idx, q, 10, P)
        _check_against_oracle(oracle, idx, codec, q, 10, P, d, i)


def test_byte_tiles_small_and_empty_cells(oracle):
    from annlite_amd import Metric

    idx, codec, vq, x = _build(900, 64, 16, 64, Metric.EUCLIDEAN, seed=13)
    _, q = _data(np.random.RandomState(14), 1, 64, 19)
    for P in (1, 3, 63):
        d, i = _both(idx, q, 10, P)
        _check_against_oracle(oracle, idx, codec, q, 10, P, d, i)


def test_c_abi_argument_checks():
    import torch

    from annlite_amd import ops

    dev = torch.device('cuda', 0)
    q = torch.zeros((4, 64), device=dev)
    cb = torch.zeros((8, 256, 8), device=dev)
    codes = torch.zeros((64, 8), dtype=torch.uint8, device=dev)
    cells = torch.zeros((4, 1), dtype=torch.int32, device=dev)
    rows = torch.tensor([[0, 64]], dtype=torch.int64, device=dev)
    order = torch.zeros((1,), dtype=torch.int32, device=dev)
    with pytest.raises(Exception, match='M = 16'):
        ops.ivf_search_topk(1, q, cb, codes, cells, 1, rows, order, 10, 8, 256)
