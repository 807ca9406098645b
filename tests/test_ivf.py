"""Pruned (IVF) PQ search: ``IvfPQGpuIndex`` / ``VQCodec`` / the ``annlite_ivf_*`` entry points against the CPU
oracle's restatement (``pq_oracle.select_cells / assign_cells / ivf_search``).

Reference structure: AnnLite(n_cells > 1) -- VQCodec (annlite/core/codec/vq.py), _cell_selection
(annlite/index.py:458-466), CellContainer.ivf_search (annlite/container.py:88-144).  The reference probes every
cell (index.py:94); with every cell probed the GPU result must equal the flat index's, with fewer cells it must
equal the exact top-k of the probed rows.  bar: bit-exact ids and distances (EUCLIDEAN; cosine / inner product
within the north star's 1e-4 because l2_normalize sums in another order than numpy).
"""
import numpy as np
import pytest

from conftest import has_gpu



# ------------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_select_cells_matches_float64_ranking(oracle):
    rng = np.random.RandomState(3)
    q = rng.randn(9, 24).astype(np.float32)
    c = rng.randn(13, 24).astype(np.float32)
    d2 = ((q[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(oracle.select_cells(q, c, 0, 5), np.argsort(d2, axis=1, kind='stable')[:, :5])
    ip = -(q.astype(np.float64) @ c.astype(np.float64).T)
    assert np.array_equal(oracle.select_cells(q, c, 1, 4), np.argsort(ip, axis=1, kind='stable')[:, :4])
    assert np.array_equal(oracle.assign_cells(q, c), np.argmin(d2, axis=1))


def test_oracle_ivf_search_with_every_cell_is_the_flat_search(oracle):
    rng = np.random.RandomState(5)
    N, D, M, Ks, C = 3000, 32, 8, 256, 7
    cb = rng.randn(M, Ks, D // M).astype(np.float32)
    x = rng.randn(N, D).astype(np.float32)
    q = rng.randn(6, D).astype(np.float32)
    codes = oracle.encode_c(x, cb)
    cells = rng.randint(0, C, size=N)
    every = np.tile(np.arange(C), (6, 1))
    for metric in (oracle.EUCLIDEAN, oracle.COSINE, oracle.INNER_PRODUCT):
        d0, i0 = oracle.index_search(q, cb, codes, metric, 10)
        d1, i1 = oracle.ivf_search(q, cb, codes, cells, every, metric, 10)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    # a pruned search returns rows of the probed cells only, in (distance, id) order
    probe = np.array([[1, 4]] * 6)
    d2, i2 = oracle.ivf_search(q, cb, codes, cells, probe, oracle.EUCLIDEAN, 10)
    assert np.isin(cells[i2], [1, 4]).all()
    assert (np.diff(d2, axis=1) >= 0).all()


def test_vq_codec_host_logic():
    import pickle

    from annlite_amd import Metric
    from annlite_amd.core.codec.vq import VQCodec

    vq = VQCodec(8, metric=Metric.EUCLIDEAN)
    assert not vq.is_trained
    assert hash(vq) == hash(VQCodec(8, metric=Metric.EUCLIDEAN)) != hash(VQCodec(9, metric=Metric.EUCLIDEAN))
    with pytest.raises(AssertionError):
        vq.encode(np.zeros((2, 4), np.float32))  # untrained
    vq._codebook = np.arange(32, dtype=np.float32).reshape(8, 4)
    vq._is_trained = True
    vq2 = pickle.loads(pickle.dumps(vq))
    assert vq2.is_trained and np.array_equal(vq2.codebook, vq.codebook)
    assert vq.decode(None) is None


# ------------------------------------------------------------------------------------------- GPU
def _data(rng, N, D, B, r=8):
    A = rng.randn(r, D).astype(np.float32)
    x = (rng.randn(N, r).astype(np.float32) @ A + 0.1 * rng.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rng.randn(B, r).astype(np.float32) @ A + 0.1 * rng.randn(B, D).astype(np.float32)).astype(np.float32)
    return x, q


def _build(N, D, M, C, metric, seed, x=None, **kw):
    from annlite_amd import PQCodec
    from annlite_amd.core.codec.vq import VQCodec
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    rng = np.random.RandomState(seed)
    if x is None:
        x, _ = _data(rng, N, D, 1)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=metric, n_init=1)
    codec.seed = 1
    codec.fit(x[:4096], iter=4)
    vq = VQCodec(C, metric=metric, iter=8, n_init=1)
    vq.seed = 2
    vq.fit(x[:4096])
    idx = IvfPQGpuIndex(dim=D, metric=metric, pq_codec=codec, vq_codec=vq, initial_size=N, **kw)
    idx.add_with_ids(x, np.arange(N))
    return idx, codec, vq, x


def _oracle_metric(oracle, metric):
    from annlite_amd import Metric

    return {Metric.EUCLIDEAN: oracle.EUCLIDEAN, Metric.COSINE: oracle.COSINE, Metric.INNER_PRODUCT: oracle.INNER_PRODUCT}[metric]


def _check_against_oracle(oracle, idx, codec, q, k, P, d, i, indices=None):
    from annlite_amd import Metric, ops

    N = idx._n_rows
    codes = ops.codes_to_numpy(idx._plain_codes(N))
    cells_of = idx._cell_of[:N].cpu().numpy()
    probe = idx.probe_cells(idx._pre(q), P).cpu().numpy()
    valid = idx._valid_bool[:N].cpu().numpy()
    if indices is not None:
        sel = np.zeros(N, bool)
        sel[np.asarray(indices)] = True
        valid = valid & sel
    od, oi = oracle.ivf_search(q, codec.codebooks, codes, cells_of, probe, _oracle_metric(oracle, idx.metric), k, valid=valid)
    assert np.array_equal(oi, i)
    if idx.metric == Metric.EUCLIDEAN:
        assert np.array_equal(od, d)
    else:
        assert np.allclose(od, d, rtol=1e-4, atol=1e-6)  # north-star tolerance (l2_normalize summation order)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
@pytest.mark.parametrize('M,D,metric_name,C,P,B,k', [
    (16, 64, 'EUCLIDEAN', 32, 4, 100, 10),
    (16, 64, 'COSINE', 32, 4, 37, 10),
    (8, 64, 'INNER_PRODUCT', 16, 3, 64, 5),
    (32, 128, 'EUCLIDEAN', 32, 4, 50, 10),
    (64, 768, 'COSINE', 16, 4, 20, 10),
    (16, 64, 'EUCLIDEAN', 32, 1, 1, 1),      # one query, one probe, k = 1
    (16, 64, 'EUCLIDEAN', 8, 7, 33, 20),     # k above the number of waves: no seed bound
    (64, 256, 'EUCLIDEAN', 8, 2, 9, 16),     # M = 64: k above its 12 waves
    (16, 64, 'EUCLIDEAN', 32, 4, 300, 64),   # the largest k
], ids=lambda v: str(v))
def test_pruned_search_equals_oracle(oracle, M, D, metric_name, C, P, B, k):
    from annlite_amd import Metric

    metric = Metric[metric_name]
    N = 20000 if D < 512 else 12000
    idx, codec, vq, x = _build(N, D, M, C, metric, seed=0)
    _, q = _data(np.random.RandomState(1), 1, D, B)
    # cells: nearest centroid of the vectors as given (vq.py:81-90); probes: cdist ranking (index.py:462-465)
    assert np.array_equal(oracle.assign_cells(x, vq.codebook), idx._cell_of[:N].cpu().numpy())
    d, i = idx.search_batch(q, limit=k, n_probe=P)
    _check_against_oracle(oracle, idx, codec, q, k, P, d, i)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_probe_selection_equals_oracle(oracle):
    from annlite_amd import Metric, ops

    for metric, kind in ((Metric.EUCLIDEAN, 0), (Metric.INNER_PRODUCT, 1), (Metric.COSINE, 1)):
        idx, codec, vq, x = _build(6000, 64, 16, 48, metric, seed=3)
        _, q = _data(np.random.RandomState(4), 1, 64, 77)
        qq = oracle.l2_normalize(q) if metric == Metric.COSINE else q
        cent = vq.codebook
        if metric == Metric.COSINE:
            cent = ops.l2_normalize(ops.to_dev(cent)).cpu().numpy()
        for P in (1, 5, 48):
            got = idx.probe_cells(idx._pre(q), P).cpu().numpy()
            if metric == Metric.COSINE:  # GPU-normalised queries differ from numpy's in the last bit: compare as sets
                want = oracle.select_cells(idx._pre(q).cpu().numpy(), cent, kind, P)
            else:
                want = oracle.select_cells(qq, cent, kind, P)
            assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_every_cell_probed_equals_flat_index(oracle):
    """the reference's own behaviour (n_probe = max(n_probe, n_cells), index.py:94)"""
    from annlite_amd import Metric
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex

    for metric in (Metric.EUCLIDEAN, Metric.COSINE):
        idx, codec, vq, x = _build(15000, 64, 16, 12, metric, seed=5)
        _, q = _data(np.random.RandomState(6), 1, 64, 40)
        flat = PQFlatGpuIndex(dim=64, metric=metric, pq_codec=codec, initial_size=15000)
        flat.add_with_ids(x, np.arange(15000))
        d0, i0 = flat.search_batch(q, limit=10)
        d1, i1 = idx.search_batch(q, limit=10)                 # n_probe None -> every cell
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
        d2, i2 = idx.search_batch(q, limit=10, n_probe=12)      # explicit n_probe >= n_cells
        assert np.array_equal(i0, i2) and np.array_equal(d0, d2)
        # the pruned machinery over ALL cells but one short: still a subset search that the oracle reproduces
        d3, i3 = idx.search_batch(q, limit=10, n_probe=11)
        _check_against_oracle(oracle, idx, codec, q, 10, 11, d3, i3)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_delete_filter_and_reseal(oracle):
    from annlite_amd import Metric

    idx, codec, vq, x = _build(12000, 64, 16, 16, Metric.EUCLIDEAN, seed=7)
    _, q = _data(np.random.RandomState(8), 1, 64, 25)
    d, i = idx.search_batch(q, limit=10, n_probe=4)
    # delete the current best hits: they must disappear, the rest must still match the oracle
    gone = np.unique(i[:, :3].ravel())
    gone = gone[gone >= 0]
    idx.delete(gone.tolist())
    d2, i2 = idx.search_batch(q, limit=10, n_probe=4)
    assert not np.isin(i2, gone).any()
    _check_against_oracle(oracle, idx, codec, q, 10, 4, d2, i2)
    # `indices` subset (pq_index.py:42-44)
    subset = np.arange(0, 12000, 3)
    d3, i3 = idx.search_batch(q, limit=10, n_probe=4, indices=subset)
    assert np.isin(i3[i3 >= 0], subset).all()
    _check_against_oracle(oracle, idx, codec, q, 10, 4, d3, i3, indices=subset)
    # add more rows (new offsets): the sealed view is rebuilt
    x2, _ = _data(np.random.RandomState(9), 3000, 64, 1)
    idx.add_with_ids(x2, np.arange(12000, 15000))
    d4, i4 = idx.search_batch(q, limit=10, n_probe=4)
    _check_against_oracle(oracle, idx, codec, q, 10, 4, d4, i4)
    assert idx.size == 15000 - gone.size


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_candidate_list_overflow_falls_back_to_the_whole_cell(oracle):
    """Heavy ties (every vector stored 40 times) make far more rows pass the integer filter than a 64-entry list
    holds: the overflow flag sends those slots through the whole-cell re-score; results stay exact, ties by id."""
    from annlite_amd import Metric

    rng = np.random.RandomState(11)
    base, q = _data(rng, 300, 64, 30)
    x = np.repeat(base, 40, axis=0)
    idx, codec, vq, _ = _build(x.shape[0], 64, 16, 6, Metric.EUCLIDEAN, seed=11, x=x)
    idx.cand_cap = 64
    d, i = idx.search_batch(q, limit=10, n_probe=2)
    _check_against_oracle(oracle, idx, codec, q, 10, 2, d, i)
    assert (np.diff(d, axis=1) >= 0).all()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_candidate_list_overflow_with_float_rerank_keeps_the_probed_cells(oracle):
    """Same heavy ties, `rerank=True`: an overflowed list used to contribute NOTHING to the float re-rank (the query
    lost its nearest cell).  The candidates now come from the exact path (ADC top-k of the probed cells), so the true
    nearest rows of the probed cells are returned."""
    from annlite_amd import Metric

    rng = np.random.RandomState(11)
    base, q = _data(rng, 300, 64, 30)
    x = np.repeat(base, 40, axis=0)
    idx, codec, vq, _ = _build(x.shape[0], 64, 16, 6, Metric.EUCLIDEAN, seed=11, x=x, rerank=True)
    idx.cand_cap = 64
    from annlite_amd import ops

    P, k, rk = 2, 10, 16
    d, i = idx.search_batch(q, limit=k, n_probe=P, rerank_k=rk)
    assert (i >= 0).all() and (np.diff(d, axis=1) >= -1e-6).all()
    # the candidate pool of an overflowed batch holds the ADC top-`rerank_k` rows of every query's probed cells (the
    # oracle's pruned search); the result = the pool ranked by exact distance
    N = idx._n_rows
    codes = ops.codes_to_numpy(idx._plain_codes(N))
    cells_of = idx._cell_of[:N].cpu().numpy()
    probe = idx.probe_cells(idx._pre(q), P).cpu().numpy()
    _, pool = oracle.ivf_search(q, codec.codebooks, codes, cells_of, probe, oracle.EUCLIDEAN, rk)
    for b in range(q.shape[0]):
        np.testing.assert_allclose(d[b], np.sqrt(((x[i[b]] - q[b]) ** 2).sum(1)), rtol=1e-4, atol=1e-5)
        # at least as good as the exact ranking of that pool, row for row (a list that did not overflow contributes
        # every row it emitted, a superset; among the 40 identical copies of a vector any id may be returned)
        ex = np.sort(np.sqrt(((x[pool[b]] - q[b]) ** 2).sum(1)))[:k]
        assert (d[b] <= ex * (1 + 1e-4) + 1e-5).all(), (b, d[b], ex)
        assert len(set(i[b].tolist())) == k


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_small_and_empty_cells(oracle):
    """more cells than the data has clusters: some cells hold a handful of rows, queries still get exact answers"""
    from annlite_amd import Metric

    idx, codec, vq, x = _build(900, 64, 16, 64, Metric.EUCLIDEAN, seed=13)
    _, q = _data(np.random.RandomState(14), 1, 64, 19)
    for P in (1, 3, 63):
        d, i = idx.search_batch(q, limit=10, n_probe=P)
        _check_against_oracle(oracle, idx, codec, q, 10, P, d, i)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_plan_groups_every_pair_into_one_slot():
    import torch

    from annlite_amd import ops

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    B, P, C, qt = 123, 5, 37, 16
    cells = torch.stack([torch.randperm(C, generator=g, device=dev)[:P] for _ in range(B)]).to(torch.int32).contiguous()
    sizes = torch.randint(0, 500, (C,), generator=g, device=dev)
    begin = torch.cumsum((sizes + 63) // 64 * 64, 0) - (sizes + 63) // 64 * 64
    cell_rows = torch.stack([begin, begin + sizes], 1).contiguous()
    order = torch.sort(sizes, descending=True, stable=True).indices.to(torch.int32).contiguous()
    vmap, slot_of, tile_rows, used = ops.ivf_plan(cells, C, qt, cell_rows, order)
    vmap, slot_of, tile_rows = vmap.cpu().numpy(), slot_of.cpu().numpy(), tile_rows.cpu().numpy()
    cells_h, rows_h = cells.cpu().numpy(), cell_rows.cpu().numpy()
    assert len(np.unique(slot_of)) == B * P                       # every pair has its own slot
    assert np.array_equal(vmap[slot_of], np.repeat(np.arange(B), P).reshape(B, P))
    assert (vmap >= 0).sum() == B * P
    assert np.array_equal(tile_rows[slot_of // qt], rows_h[cells_h])   # the slot's tile scans the pair's cell
    n_used = int(used.item())
    assert n_used == sum(-(-np.bincount(cells_h.ravel(), minlength=C) // qt))
    assert (tile_rows[n_used:, 0] == -1).all()
    lens = tile_rows[:n_used, 1] - tile_rows[:n_used, 0]
    assert (np.diff(lens) <= 0).all()                             # longest tiles first


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_rerank_and_persistence(oracle, tmp_path):
    from annlite_amd import Metric
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    idx, codec, vq, x = _build(20000, 64, 16, 16, Metric.EUCLIDEAN, seed=17, rerank=True)
    _, q = _data(np.random.RandomState(18), 1, 64, 64)
    d, i = idx.search_batch(q, limit=10, n_probe=8, rerank_k=64)
    # exact float distances of the returned ids, ascending; recall against brute force over the probed cells
    ex = np.sqrt(((x[i] - q[:, None, :]) ** 2).sum(-1))
    assert np.allclose(ex, d, rtol=1e-4, atol=1e-5) and (np.diff(d, axis=1) >= -1e-6).all()
    probe = idx.probe_cells(idx._pre(q), 8).cpu().numpy()
    cells_of = idx._cell_of[:20000].cpu().numpy()
    rec = []
    for b in range(q.shape[0]):
        rows = np.nonzero(np.isin(cells_of, probe[b]))[0]
        truth = rows[np.argsort(((x[rows] - q[b]) ** 2).sum(-1), kind='stable')[:10]]
        rec.append(len(set(truth) & set(i[b])) / 10)
    assert np.mean(rec) >= 0.8
    # dump / load keeps cells and results
    f = tmp_path / 'ivf.idx'
    idx.dump(f)
    idx2 = IvfPQGpuIndex(dim=64, metric=Metric.EUCLIDEAN, pq_codec=codec, vq_codec=vq, initial_size=20000, rerank=True)
    idx2.load(f)
    d2, i2 = idx2.search_batch(q, limit=10, n_probe=8, rerank_k=64)
    assert np.array_equal(i, i2) and np.array_equal(d, d2)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_vq_codec_training_quality():
    """statistical parity with sklearn KMeans (vq.py:41-50): inertia within 10 % on clustered data"""
    from sklearn.cluster import KMeans

    from annlite_amd import Metric
    from annlite_amd.core.codec.vq import VQCodec

    rng = np.random.RandomState(21)
    centres = rng.randn(16, 32).astype(np.float32) * 4
    x = (centres[rng.randint(0, 16, 6000)] + rng.randn(6000, 32).astype(np.float32)).astype(np.float32)
    vq = VQCodec(16, metric=Metric.EUCLIDEAN, iter=50, n_init=3)
    vq.seed = 5
    vq.fit(x)
    codes = vq.encode(x)
    mine = float(((x - vq.codebook[codes]) ** 2).sum())
    ref = KMeans(16, n_init=3, max_iter=50, random_state=0).fit(x).inertia_
    assert mine <= 1.1 * ref
    # streaming variant (vq.py:52-76)
    vq2 = VQCodec(16, metric=Metric.EUCLIDEAN)
    vq2.seed = 6
    for s in range(0, 6000, 1000):
        vq2.partial_fit(x[s:s + 1000])
    vq2.build_codebook()
    assert vq2.is_trained and vq2.codebook.shape == (16, 32)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_annlite_facade_with_cells(tmp_path):
    """AnnLite(n_cells > 1): like the reference every cell is visited, so the matches equal the single-cell
    facade's; ivf_prune=True honours n_probe (matches come from the probed cells, most of them the same)."""
    from annlite_amd import AnnLite
    from annlite_amd.docarray_compat import Document, DocumentArray

    rs = np.random.RandomState(5)
    N, D = 6000, 64
    x, q = _data(rs, N, D, 20)
    res = {}
    for name, kw in (('flat', {}), ('cells', dict(n_cells=8, n_probe=2)), ('pruned', dict(n_cells=8, n_probe=3, ivf_prune=True))):
        ann = AnnLite(D, metric='euclidean', n_subvectors=8, data_path=str(tmp_path / name), **kw)
        ann._pq_codec.seed = 1  # same PQ codebooks for the three facades
        if ann._vq_codec is not None:
            ann._vq_codec.seed = 2
        ann.train(x[:4096])
        assert ann.is_trained
        ann.index(DocumentArray([Document(id=str(i), embedding=x[i]) for i in range(N)]))
        docs = DocumentArray([Document(id='q%d' % i, embedding=q[i]) for i in range(len(q))])
        ann.search(docs, limit=10)
        res[name] = [[(m.id, m.scores['euclidean'].value) for m in d.matches] for d in docs]
        if name == 'cells':  # a second facade over the same data_path picks both trained codecs up (index.py:125-140)
            again = AnnLite(D, metric='euclidean', n_subvectors=8, data_path=str(tmp_path / name), **kw)
            assert again.is_trained and np.array_equal(again._vq_codec.codebook, ann._vq_codec.codebook)
    # (every facade trains its own PQ codebooks: same seed, but the k-means sums are fp32 atomics, so the codewords
    # agree only to the last bits -- ids must agree, distances to 1e-5)
    same = np.mean([[a[0] == b[0] for a, b in zip(ra, rb)] for ra, rb in zip(res['cells'], res['flat'])])
    assert same >= 0.99, same
    assert np.allclose([[m[1] for m in r] for r in res['cells']], [[m[1] for m in r] for r in res['flat']], rtol=1e-5)
    agree = np.mean([len({m for m, _ in a} & {m for m, _ in b}) / 10 for a, b in zip(res['pruned'], res['flat'])])
    assert 0.5 <= agree <= 1.0


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_pruned_index_behind_the_row_sharded_wrapper(oracle):
    """sharded.py: a rank's shard may be an IvfPQGpuIndex -- with pruning active the general (non-packed) path runs
    and ids come back as global rows (row_base + local offset)."""
    import torch

    from annlite_amd import Metric, ops
    from annlite_amd.sharded import ShardedPQIndex

    idx, codec, vq, x = _build(9000, 64, 16, 16, Metric.EUCLIDEAN, seed=23, n_probe=4)
    _, q = _data(np.random.RandomState(24), 1, 64, 31)
    qd = ops.to_dev(q, torch.float32)
    assert idx.search_batch_packed(qd, 10, row_base=0) is None
    d0, i0 = idx.search_batch(qd, limit=10)
    d1, i1 = ShardedPQIndex(idx, row_base=5000).search_batch(qd, limit=10)
    assert torch.equal(d0, d1) and torch.equal(torch.where(i0 >= 0, i0 + 5000, i0), i1)
    _check_against_oracle(oracle, idx, codec, q, 10, 4, d0.cpu().numpy(), i0.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
@pytest.mark.parametrize('M', [16, 64])
def test_pruned_search_with_skewed_value_ranges(oracle, M):
    """One sub-space dominates the table range (coarse integer steps for all others), another is constant (zero
    range): the integer bound must stay valid -- ids and distances still equal the oracle's."""
    from annlite_amd import Metric

    rng = np.random.RandomState(31)
    D = 4 * M
    x, q = _data(rng, 16000, D, 45)
    x[:, :4] *= 300.0
    q[:, :4] *= 300.0
    x[:, 4:8] = 0.25
    q[:, 4:8] = 0.25
    idx, codec, vq, _ = _build(x.shape[0], D, M, 24, Metric.EUCLIDEAN, seed=31, x=x)
    for P, k in ((3, 10), (23, 5)):
        d, i = idx.search_batch(q, limit=k, n_probe=P)
        _check_against_oracle(oracle, idx, codec, q, k, P, d, i)


# ------------------------------------------------------------------------- golden fixture from the REAL reference
def _cells_fixture():
    import os

    from conftest import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, 'cells', 'cells_m16_d64.npz'))
    g = {k: z[k] for k in z.files}
    g['M'], g['dsub'], g['Ks'], g['N'], g['B'], g['C'], g['P'], g['K'], _ = (int(v) for v in g['meta'])
    return g


def _same_outside_ties(ids, want_ids, dists):
    neq = ids != want_ids
    return all((dists == dists[j]).sum() > 1 for j in np.where(neq)[0])


def test_oracle_cells_against_reference_fixture(oracle):
    """tests/golden/cells/*.npz was produced by the reference's VQCodec.encode / math.cdist + top_k / per-cell
    PQIndex.search merged like CellContainer.ivf_search (tests/golden/make_golden.py:make_cells_case)."""
    g = _cells_fixture()
    got = oracle.assign_cells(g['x'], g['centroids'])
    d2 = oracle.cell_distances(g['x'], g['centroids'], 0)
    for n in np.nonzero(got != g['cells_of'])[0]:  # scipy's GEMM expansion vs the fp32 chain: near ties only
        assert abs(d2[n, got[n]] - d2[n, g['cells_of'][n]]) <= 1e-5 * d2[n, got[n]]
    for name, kind, qq, cc in (('euclidean', 0, g['queries'], g['centroids']),
                               ('cosine', 1, oracle.l2_normalize(g['queries']), oracle.l2_normalize(g['centroids']))):
        mine = oracle.select_cells(qq, cc, kind, g['P'])
        for b in range(g['B']):
            if set(mine[b]) != set(g['probe_' + name][b]):
                s = np.sort(g['cdist_' + name][b])
                assert abs(s[g['P']] - s[g['P'] - 1]) <= 1e-4 * max(abs(s[g['P'] - 1]), 1e-6)
    assert np.array_equal(oracle.encode_c(g['x'], g['codebooks']), g['codes'])
    od, oi = oracle.ivf_search(g['queries'], g['codebooks'], g['codes'], g['cells_of'], g['probe_euclidean'],
                               oracle.EUCLIDEAN, g['K'], sqrt_euclidean=False)
    assert np.array_equal(od, g['merged_d'])
    assert all(_same_outside_ties(oi[b], g['merged_i'][b], od[b]) for b in range(g['B']))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
def test_gpu_cells_against_reference_fixture():
    """the GPU index, fed the fixture's codebooks / centroids, reproduces the reference's merged per-cell results"""
    import torch

    from annlite_amd import Metric, PQCodec
    from annlite_amd.core.codec.vq import VQCodec
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    g = _cells_fixture()
    D = g['M'] * g['dsub']
    codec = PQCodec(dim=D, n_subvectors=g['M'], n_clusters=g['Ks'], metric=Metric.EUCLIDEAN)
    codec.set_codebooks(torch.from_numpy(g['codebooks']))
    vq = VQCodec(g['C'], metric=Metric.EUCLIDEAN)
    vq._codebook = g['centroids']
    vq._is_trained = True
    idx = IvfPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, vq_codec=vq, initial_size=g['N'])
    idx.add_with_ids(g['x'], np.arange(g['N']))
    cells = idx._cell_of[: g['N']].cpu().numpy()
    assert (cells != g['cells_of']).mean() < 1e-3  # (near-tie assignments of scipy's expansion aside)
    probe = idx.probe_cells(idx._pre(g['queries']), g['P']).cpu().numpy()
    same_probe = [set(probe[b]) == set(g['probe_euclidean'][b]) for b in range(g['B'])]
    same_cells = np.array_equal(cells, g['cells_of'])
    d, i = idx.search_batch(g['queries'], limit=g['K'], n_probe=g['P'])
    for b in range(g['B']):
        if not (same_probe[b] and same_cells):
            continue  # another cell set is another (equally valid) search; the oracle tests cover it
        assert np.array_equal(np.sqrt(g['merged_d'][b]), d[b])  # hnsw/index.py:164-165: sqrt for EUCLIDEAN
        assert _same_outside_ties(i[b], g['merged_i'][b], d[b])
    assert sum(same_probe) >= g['B'] - 1


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')
@pytest.mark.parametrize('kw', [{}, dict(n_cells=6, n_probe=2, ivf_prune=True), dict(graph=True)], ids=['flat', 'cells', 'graph'])
def test_annlite_dump_and_reopen(tmp_path, kw):
    """index.py:689-714 / 769-777: ``dump()`` writes the codecs and a snapshot; a new AnnLite over the same
    data_path comes back trained, with its documents, and answers the same."""
    from annlite_amd import AnnLite
    from annlite_amd.docarray_compat import Document, DocumentArray

    rs = np.random.RandomState(41)
    N, D = 3000, 64
    x, q = _data(rs, N, D, 12)
    ann = AnnLite(D, metric='euclidean', n_subvectors=8, data_path=str(tmp_path / 'idx'), **kw)
    ann.train(x[:2048])
    ann.index(DocumentArray([Document(id=str(i), embedding=x[i], tags={'parity': i % 2}) for i in range(N)]))
    ann.delete(['5', '6'])
    d0, i0 = ann.search_numpy(q, limit=10)
    f0 = ann.search_numpy(q, filter={'parity': {'$eq': 1}}, limit=5)
    assert ann.snapshot_path is None
    ann.dump()
    again = AnnLite(D, metric='euclidean', n_subvectors=8, data_path=str(tmp_path / 'idx'), **kw)
    assert again.is_trained and again.total_docs == N - 2 and again.index_size == ann.index_size
    d1, i1 = again.search_numpy(q, limit=10)
    assert all(np.array_equal(a, b) for a, b in zip(i0, i1)) and all(np.array_equal(a, b) for a, b in zip(d0, d1))
    f1 = again.search_numpy(q, filter={'parity': {'$eq': 1}}, limit=5)
    assert all(np.array_equal(a, b) for a, b in zip(f0[1], f1[1]))
    assert again.get_doc_by_id('7').tags['parity'] == 1 and again.get_doc_by_id('5') is None
