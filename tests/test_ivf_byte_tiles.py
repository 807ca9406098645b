"""The pruned search over cells on the BYTE-TABLE kernel (annlite_ivf_search_topk, round 6): cell tiles of 32 (query, cell)
pairs, exact sums inside the tile, bounds shared by query, one merge launch.  Pinned against the oracle's restatement of
`CellContainer.ivf_search` (annlite/container.py:88-144) over the probed cells -- bit-exact ids and distances -- and against
the u16 tile scan + re-score it replaces (`IvfPQGpuIndex.byte_tiles = False`)."""
import numpy as np
import pytest
import torch

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


def test_candidate_lists_are_private_prefixes_and_hold_the_adc_topk(oracle):
    """annlite_ivf_search_candidates: every (query, probed cell) list is a PREFIX of that cell's exact ADC ranking (the cell's best rows at
    or below the query's first bound, at most k), whatever the other tiles do; the union of a query's lists holds its exact ADC top-k of
    the probed cells"""
    from annlite_amd import Metric, ops
    from annlite_amd._capi import CODES_SKEWED, LUT_L2

    N, D, C, P, B, k = 30000, 64, 24, 5, 24, 16
    idx, codec, vq, x = _build(N, D, 16, C, Metric.EUCLIDEAN, seed=41)
    _, q = _data(np.random.RandomState(42), 1, D, B)
    idx._seal()
    qd = idx._pre(q)
    cells = idx.probe_cells(qd, P)
    args = (LUT_L2, qd, codec.codebooks_dev, idx._table, cells, C, idx._cell_rows, idx._cell_order, k, 16, 256)
    kw = dict(row_ids=idx._row_ids, n_rows=idx._n_table, codes_layout=CODES_SKEWED)
    ids = ops.ivf_search_candidates(*args, bound_rank=1, **kw).cpu().numpy().reshape(B, P, k)
    # (a second call may cut the far cells' lists at another length: the first bound is the k-th smallest of per-(wave, lane) minima over
    # seed blocks the waves DRAW -- any such bound is valid; the prefix property below holds for both calls)
    ids2 = ops.ivf_search_candidates(*args, bound_rank=4, **kw).cpu().numpy().reshape(B, P, k)
    assert np.array_equal(ids[:, 0], ids2[:, 0])  # the nearest cell's list: complete either way
    assert (ids2 >= 0).sum() > (ids >= 0).sum()  # the looser bound (4k-th seed sum against the k-th) lengthens the far cells' lists
    codes = ops.codes_to_numpy(idx._plain_codes(N))
    cell_of = idx._cell_of[:N].cpu().numpy()
    probe = cells.cpu().numpy()
    _, top = oracle.ivf_search(q, codec.codebooks, codes, cell_of, probe, oracle.EUCLIDEAN, 10)
    n_short = 0
    for b in range(B):
        for p in range(P):
            lst = ids[b, p]
            n = int((lst >= 0).sum())
            assert (lst[:n] >= 0).all() and (lst[n:] == -1).all()
            _, own = oracle.ivf_search(q[b:b + 1], codec.codebooks, codes, cell_of, probe[b:b + 1, p:p + 1], oracle.EUCLIDEAN, k)
            assert np.array_equal(lst[:n], own[0][:n]), (b, p)
            n_short += n < min(k, int((cell_of == probe[b, p]).sum()))
        assert set(top[b].tolist()) <= set(ids[b].reshape(-1).tolist())
        assert n >= 0
    assert n_short > 0  # (lists of far cells ARE cut by the first bound: that is the point)
    # the nearest cell's list is never cut below k rows by the bound (the bound comes from ITS rows)
    for b in range(B):
        assert (ids[b, 0] >= 0).sum() == min(k, int((cell_of == probe[b, 0]).sum()))


def test_candidate_lists_with_the_nearest_cells_in_parts(oracle):
    """IvfPQGpuIndex.rerank_split: the query's nearest cells probed as S contiguous row ranges each (entries C .. of the split cell table),
    every range with a private list of its own: every list is the prefix of ITS range's exact ADC ranking, the ranges of a query are
    disjoint (no row twice), the first range's list is complete, the union still holds the oracle's ADC top-k of the probed cells --
    and it is a larger pool than whole cells give"""
    from annlite_amd import Metric, ops
    from annlite_amd._capi import CODES_SKEWED, LUT_L2

    N, D, C, P, B, k, n_split, S = 30000, 64, 24, 5, 24, 16, 2, 4
    idx, codec, vq, x = _build(N, D, 16, C, Metric.EUCLIDEAN, seed=43)
    _, q = _data(np.random.RandomState(44), 1, D, B)
    idx._seal()
    qd = idx._pre(q)
    cells = idx.probe_cells(qd, P)
    rows_t, order_t = idx._split_tables(S)
    assert rows_t.shape[0] == C * (1 + S) and torch.equal(rows_t[:C], idx._cell_rows)
    parts = C + cells[:, :n_split, None].to(torch.int64) * S + torch.arange(S, device=cells.device)
    cells_x = torch.cat([parts.reshape(B, -1).to(torch.int32), cells[:, n_split:]], dim=1).contiguous()
    Px = cells_x.shape[1]
    kw = dict(row_ids=idx._row_ids, n_rows=idx._n_table, codes_layout=CODES_SKEWED, bound_rank=2)
    ids = ops.ivf_search_candidates(LUT_L2, qd, codec.codebooks_dev, idx._table, cells_x, C * (1 + S), rows_t, order_t, k, 16, 256,
                                    **kw).cpu().numpy().reshape(B, Px, k)
    whole = ops.ivf_search_candidates(LUT_L2, qd, codec.codebooks_dev, idx._table, cells, C, idx._cell_rows, idx._cell_order, k, 16, 256,
                                      **kw).cpu().numpy()
    assert (ids >= 0).sum() > (whole >= 0).sum()
    # the first bound from the WHOLE nearest cell (seed_cells) instead of its first part: a tighter bound, the same promises but the last
    ids_ws = ops.ivf_search_candidates(LUT_L2, qd, codec.codebooks_dev, idx._table, cells_x, C * (1 + S), rows_t, order_t, k, 16, 256,
                                       seed_cells=cells[:, 0].contiguous(), **kw).cpu().numpy().reshape(B, Px, k)
    assert (whole >= 0).sum() < (ids_ws >= 0).sum() <= (ids >= 0).sum()
    codes = ops.codes_to_numpy(idx._plain_codes(N))
    cell_of = idx._cell_of[:N].cpu().numpy()
    probe = cells.cpu().numpy()
    row_ids = idx._row_ids.cpu().numpy()
    rows_np, cx = rows_t.cpu().numpy(), cells_x.cpu().numpy()
    lut = oracle.get_dist_mat_c(q, codec.codebooks, oracle.EUCLIDEAN)
    _, top = oracle.ivf_search(q, codec.codebooks, codes, cell_of, probe, oracle.EUCLIDEAN, k)
    for b in range(B):
        dist = oracle.dist_pqcodes_to_codebooks_c(lut[b], codes)
        for which, first_complete in ((ids, True), (ids_ws, False)):
            got = which[b].reshape(-1)
            got = got[got >= 0]
            assert len(set(got.tolist())) == got.size and np.isin(cell_of[got], probe[b]).all()
            assert set(top[b][top[b] >= 0].tolist()) <= set(got.tolist())
            for p in range(Px):
                lst = which[b, p]
                n = int((lst >= 0).sum())
                assert (lst[n:] == -1).all()
                own = row_ids[rows_np[cx[b, p], 0]:rows_np[cx[b, p], 1]]
                own = own[own >= 0]
                ranked = own[np.lexsort((own, dist[own]))]
                assert np.array_equal(lst[:n], ranked[:n]), (b, p)
                if p == 0 and first_complete:
                    assert n == min(k, own.size)


def test_float_rerank_on_the_cell_tiles(oracle):
    """IvfPQGpuIndex(rerank=True) with limit <= 16: candidates from annlite_ivf_search_candidates, exact distances + top-k fused; the
    distances are the true ones, the recall is at least the ADC search's, and the u16 pipeline's re-rank is in the same range"""
    from annlite_amd import Metric

    N, D, C, P, B, k = 40000, 64, 32, 8, 200, 10
    idx, codec, vq, x = _build(N, D, 16, C, Metric.EUCLIDEAN, seed=51, rerank=True)
    _, q = _data(np.random.RandomState(52), 1, D, B)
    d, i = idx.search_batch(q, limit=k, n_probe=P)
    assert idx.last_pruned_path.startswith('annlite_ivf_search_candidates')
    assert (i >= 0).all() and (np.diff(d, axis=1) >= -1e-6).all()
    for b in range(0, B, 17):
        np.testing.assert_allclose(d[b], np.sqrt(((x[i[b]] - q[b]) ** 2).sum(1)), rtol=1e-4, atol=1e-5)
        assert len(set(i[b].tolist())) == k
    truth = np.argsort(((q[:, None, :] - x[None, :, :]) ** 2).sum(2), axis=1)[:, :k] if N * B <= 8_000_000 else None
    if truth is None:
        dd = (q ** 2).sum(1)[:, None] + (x ** 2).sum(1)[None, :] - 2.0 * q @ x.T
        truth = np.argsort(dd, axis=1)[:, :k]
    rec = lambda ids: float(np.mean([len(set(ids[b]) & set(truth[b])) / k for b in range(B)]))
    r_new = rec(i)
    assert idx.rerank_split == (2, 4) and 'nearest 2 cells in 4 parts' in idx.last_pruned_path
    idx.rerank_split = (0, 1)  # whole cells only: 16 rows per cell at most -- the smaller pool
    r_whole = rec(idx.search_batch(q, limit=k, n_probe=P)[1])
    assert 'parts' not in idx.last_pruned_path and r_new >= r_whole - 0.01, (r_new, r_whole)
    idx.rerank_split = (2, 4)
    by_rank = {}
    assert idx.rerank_bound_rank == 1
    for rank in (2, 4):  # (the default is 1) a looser first bound = longer lists from the far cells: a superset pool, up to the seed draw
        idx.rerank_bound_rank = rank
        by_rank[rank] = rec(idx.search_batch(q, limit=k, n_probe=P)[1])
    by_rank[1] = r_new
    idx.rerank_bound_rank = 0  # the pool = exactly the ADC top-16 (annlite_ivf_search_topk with k = 16): a subset of every rank's pool
    d0, i0 = idx.search_batch(q, limit=k, n_probe=P)
    assert idx.last_pruned_path.startswith('annlite_ivf_search_topk') and idx.last_pruned_path.endswith('annlite_rerank_topk')
    by_rank[0] = rec(i0)
    idx.rerank, idx.rerank_bound_rank = False, 1
    _, i16 = idx.search_batch(q, limit=16, n_probe=P)
    idx.rerank = True
    for b in range(0, B, 13):  # the exact top-10 of those 16 rows, nothing else
        e = np.sqrt(((x[i16[b]] - q[b]) ** 2).sum(1))
        assert set(i0[b].tolist()) <= set(i16[b].tolist())
        np.testing.assert_allclose(d0[b], np.sort(e)[:k], rtol=1e-4, atol=1e-5)
    assert by_rank[0] <= by_rank[1] + 0.02 and by_rank[1] <= by_rank[2] + 0.02 and by_rank[2] <= by_rank[4] + 0.02, by_rank
    idx.rerank = False
    _, i_adc = idx.search_batch(q, limit=k, n_probe=P)
    idx.rerank = True
    idx.byte_tiles = False
    _, i_u16 = idx.search_batch(q, limit=k, n_probe=P)
    assert idx.last_pruned_path.startswith('annlite_pq_search_tiles')
    idx.byte_tiles = True
    # (the u16 pipeline re-ranks every tile's whole candidate list, a larger pool than P lists of 16: rank 4 is within 0.05 of it)
    assert by_rank[1] >= rec(i_adc) and by_rank[4] >= rec(i_u16) - 0.05, (by_rank, r_new, rec(i_adc), rec(i_u16))


def test_facade_pruned_search_with_the_float_rerank(tmp_path):
    """AnnLite(n_cells, n_probe, ivf_prune=True, rerank=True[, rerank_split, rerank_bound_rank]): the keyword channel of the reference's
    container (container.py:56) reaches IvfPQGpuIndex; a search with limit <= 16 takes the cell tiles' candidate lists + the exact
    re-rank, and the scores are the TRUE distances of the returned documents"""
    from annlite_amd import AnnLite
    from annlite_amd.docarray_compat import Document, DocumentArray

    rs = np.random.RandomState(15)
    N, D = 8000, 64
    x, q = _data(rs, N, D, 24)
    for kw, tag in ((dict(), 'nearest 2 cells in 4 parts'), (dict(rerank_split=(1, 2), rerank_bound_rank=2), 'nearest 1 cells in 2 parts'),
                    (dict(rerank_split=(0, 1)), None)):
        ann = AnnLite(D, metric='euclidean', n_subvectors=16, n_cells=8, n_probe=3, ivf_prune=True, rerank=True,
                      data_path=str(tmp_path / ('a%d' % len(kw))), **kw)
        ann.train(x[:4096])
        ann.index(DocumentArray([Document(id=str(i), embedding=x[i]) for i in range(N)]))
        docs = DocumentArray([Document(id='q%d' % i, embedding=q[i]) for i in range(len(q))])
        ann.search(docs, limit=10)
        idx = ann._vec_indexes[0]
        assert idx.last_pruned_path.startswith('annlite_ivf_search_candidates'), idx.last_pruned_path
        assert (tag in idx.last_pruned_path) if tag else ('parts' not in idx.last_pruned_path)
        for b, d in enumerate(docs):
            ids = [int(m.id) for m in d.matches]
            assert len(ids) == 10 and len(set(ids)) == 10
            got = np.array([m.scores['euclidean'].value for m in d.matches], dtype=np.float64)
            np.testing.assert_allclose(got, np.sqrt(((x[ids] - q[b]) ** 2).sum(1)), rtol=1e-4, atol=1e-5)
            assert (np.diff(got) >= -1e-6).all()
