"""Live pinning of the oracle against the REAL reference, run only where /root/reference and
the build of oracle/build_ref.sh exist (the build container).  Skipped on the GPU box -- parity there rests on the
golden fixtures.  Fresh seeds and larger sizes than the fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference tree / its build (oracle/build_ref.sh) not present')


@pytest.fixture(scope='module')
def ref():
    return ref_import.load()


@pytest.mark.parametrize('M,dsub,Ks,N,B', [(16, 8, 256, 20000, 16), (8, 16, 256, 5000, 8), (64, 12, 256, 2000, 4),
                                           (8, 8, 768, 3000, 4), (3, 5, 17, 500, 3)])
def test_live_luts_scan_search(oracle, ref, M, dsub, Ks, N, B):
    rs = np.random.RandomState(M * 1000 + dsub)
    D = M * dsub
    cb = rs.rand(M, Ks, dsub).astype(np.float32)
    q = rs.rand(B, D).astype(np.float32)
    x = rs.rand(N, D).astype(np.float32)
    pb = ref.pq_bind
    assert np.array_equal(oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb), np.asarray(pb.batch_precompute_adc_table(q, dsub, Ks, cb)))
    assert np.array_equal(oracle.batch_precompute_adc_table_ip_c(q, dsub, Ks, cb), np.asarray(pb.batch_precompute_adc_table_ip(q, dsub, Ks, cb)))
    for metric in (ref.Metric.EUCLIDEAN, ref.Metric.INNER_PRODUCT, ref.Metric.COSINE):
        c = ref.PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=metric)
        c._codebooks = cb
        c._is_trained = True
        assert np.array_equal(oracle.get_dist_mat_c(q, cb, int(metric)), c.get_dist_mat(q))
    codes = c.encode(x)
    got = oracle.encode_c(x, cb)
    bad = np.argwhere(got != codes)
    if len(bad):
        best, second = oracle.encode_gap(x, cb)
        for n, m in bad:
            assert (second[n, m] - best[n, m]) / second[n, m] < 1e-5
    lut = np.asarray(pb.batch_precompute_adc_table(q, dsub, Ks, cb))
    for b in range(min(B, 3)):
        want = np.asarray(pb.dist_pqcodes_to_codebooks(lut[b], codes), dtype=np.float32)
        assert np.array_equal(oracle.dist_pqcodes_to_codebooks_c(lut[b], codes), want)
    if codes.dtype == np.uint8:
        ce = ref.PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=ref.Metric.EUCLIDEAN)
        ce._codebooks = cb
        ce._is_trained = True
        idx = ref.PQIndex(D, ce, initial_size=N)
        idx.add_with_ids(x, np.arange(N))
        d, i = oracle.adc_search_c(lut, codes, 10)
        for b in range(min(B, 3)):
            rd, ri = idx.search(q[b], limit=10)
            assert np.array_equal(np.asarray(rd), d[b].astype(np.float64))
            neq = np.asarray(ri) != i[b]
            assert all((np.asarray(rd) == rd[j]).sum() > 1 for j in np.where(neq)[0])


def test_live_cells_vq_selection_and_merge(oracle, ref):
    """The oracle's restatement of the n_cells > 1 structure against the reference's own pieces: VQCodec.encode
    (vq.py:78-90), cdist + top_k of _cell_selection (index.py:462-465), per-cell PQIndex.search + the
    concatenate/argsort merge of CellContainer.ivf_search (container.py:101-138, python loop restated here)."""
    rs = np.random.RandomState(77)
    N, D, M, Ks, C, B, k = 6000, 32, 8, 256, 12, 9, 10
    cent = rs.randn(C, D).astype(np.float32)
    x = (cent[rs.randint(0, C, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
    q = (cent[rs.randint(0, C, B)] + 0.7 * rs.randn(B, D)).astype(np.float32)

    # cell assignment: scipy vq (GEMM expansion) vs the fp32 chain -- any mismatch must be a near tie
    vq = ref.VQCodec(C, metric=ref.Metric.EUCLIDEAN)
    vq._codebook = cent
    vq._is_trained = True
    want = np.asarray(vq.encode(x))
    got = oracle.assign_cells(x, cent)
    d2 = oracle.cell_distances(x, cent, 0)
    for n in np.nonzero(want != got)[0]:
        assert abs(d2[n, want[n]] - d2[n, got[n]]) <= 1e-5 * d2[n, got[n]]

    # probe selection: the reference ranks by cdist(..., metric) then top_k; same SETS unless the boundary is a near tie
    for name, kind, qq, cc in (('euclidean', 0, q, cent),
                               ('cosine', 1, oracle.l2_normalize(q), oracle.l2_normalize(cent))):
        dists = ref.math.cdist(q, cent, metric=name)
        _, rcells = ref.math.top_k(dists, k=4)
        mine = oracle.select_cells(qq, cc, kind, 4)
        for b in range(B):
            if set(rcells[b]) != set(mine[b]):
                s = np.sort(dists[b])
                assert abs(s[4] - s[3]) <= 1e-4 * max(abs(s[3]), 1e-6)

    # search: one reference PQIndex per cell, merged the way ivf_search does it
    cb = rs.rand(M, Ks, D // M).astype(np.float32)
    codec = ref.PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=ref.Metric.EUCLIDEAN)
    codec._codebooks = cb
    codec._is_trained = True
    codes = oracle.encode_c(x, cb)
    cells_of = got
    per_cell, offsets = {}, {}
    for c in range(C):
        rows = np.nonzero(cells_of == c)[0]
        offsets[c] = rows
        idx = ref.PQIndex(D, codec, initial_size=max(len(rows), 1))
        if len(rows):
            idx.add_with_ids(x[rows], np.arange(len(rows)))
        per_cell[c] = idx
    probe = oracle.select_cells(q, cent, 0, 4)
    od, oi = oracle.ivf_search(q, cb, codes, cells_of, probe, oracle.EUCLIDEAN, k, sqrt_euclidean=False)
    for b in range(B):
        ds, ids = [], []
        for c in probe[b]:
            n_c = len(offsets[c])
            if n_c == 0:
                continue
            dd, ii = per_cell[c].search(q[b], limit=min(k, n_c))
            ds.append(np.asarray(dd, dtype=np.float32))
            ids.append(offsets[c][np.asarray(ii)])
        ds, ids = np.concatenate(ds), np.concatenate(ids)
        order = np.lexsort((ids, ds))[:k]  # ivf_search: argsort of the concatenation (ties: the build's id order)
        assert np.array_equal(ds[order], od[b])
        neq = ids[order] != oi[b]
        assert all((ds == ds[order][j]).sum() > 1 for j in np.where(neq)[0])


def test_live_filter_semantics_match_reference_sql(ref):
    """annlite_amd.filter evaluates the filter grammar in memory; the reference compiles it to SQL
    (annlite/filter.py) and lets SQLite select the offsets (storage/table.py).  Same offsets for every filter,
    documents with missing tags (NULL) included."""
    import importlib
    import sqlite3

    from annlite_amd.filter import select

    Filter = importlib.import_module('annlite.filter').Filter
    rs = np.random.RandomState(5)
    brands = ['Nike', 'Gucci', 'Puma', 'Asics']
    tags = []
    for i in range(400):
        t = {'price': int(rs.randint(0, 100)), 'rating': float(np.round(rs.rand() * 5, 2)), 'year': int(rs.randint(2000, 2015)),
             'brand': brands[rs.randint(0, 4)]}
        for key in list(t):
            if rs.rand() < 0.1:
                del t[key]  # NULL in the table
        tags.append(t)
    con = sqlite3.connect(':memory:')
    con.execute('CREATE TABLE t (_id INTEGER PRIMARY KEY, price INTEGER, rating FLOAT, year INTEGER, brand TEXT)')
    con.executemany('INSERT INTO t VALUES (?, ?, ?, ?, ?)',
                    [(i, t.get('price'), t.get('rating'), t.get('year'), t.get('brand')) for i, t in enumerate(tags)])
    filters = [
        {},
        {'brand': {'$eq': 'Nike'}},
        {'price': {'$lt': 30}},
        {'price': {'$neq': 30}},
        {'brand': {'$lt': 'O'}, 'price': {'$gte': 50}},
        {'$and': {'brand': {'$eq': 'Puma'}, 'price': {'$gte': 50}}},
        {'$or': {'brand': {'$eq': 'Puma'}, 'price': {'$gte': 90}}},
        {'$and': {'brand': {'$in': ['Nike', 'Gucci']}, 'price': {'$gte': 50}}},
        {'$or': {'brand': {'$nin': ['Nike', 'Gucci']}, 'price': {'$gte': 50}}},
        {'$and': {'price': {'$gte': 0, '$lte': 54}, 'rating': {'$gte': 1}, 'year': {'$gte': 2007, '$lte': 2010}}},
        {'$and': {'price': {'$or': [{'price': {'$gte': 80}}, {'price': {'$lte': 20}}]}, 'rating': {'$gte': 1},
                  'year': {'$gte': 2007, '$lte': 2010}}},
        {'$and': {'$or': [{'price': {'$gte': 80}}, {'price': {'$lte': 20}}], 'rating': {'$gte': 1}}},
        {'rating': {'$gt': 2.5}, '$or': {'brand': {'$eq': 'Asics'}, 'year': {'$lt': 2003}}},   # AND binds tighter than OR
        {'$or': {'brand': {'$eq': 'Asics'}, 'year': {'$lt': 2003}}, 'rating': {'$gt': 2.5}},
        {'$or': [{'price': {'$lt': 10}}, {'brand': {'$eq': 'Nike'}, 'rating': {'$gte': 4}}]},
    ]
    for flt in filters:
        where, params = Filter(flt).parse_where_clause()
        sql = 'SELECT _id FROM t' + (f' WHERE {where}' if where else '') + ' ORDER BY _id'
        want = [r[0] for r in con.execute(sql, params)]
        assert select(tags, flt) == want, flt
    with pytest.raises(ValueError):
        select(tags, {'$may': {'brand': {'$lt': 1}}})
    with pytest.raises(ValueError):
        Filter({'$may': {'brand': {'$lt': 1}}}).parse_where_clause()


def test_live_filter_fuzz_against_reference_sql(ref):
    """random filters from the grammar (nested dict / list logic, several operators per field, field-level logic):
    same offsets as the reference's SQL on SQLite"""
    import importlib
    import sqlite3

    from annlite_amd.filter import select

    Filter = importlib.import_module('annlite.filter').Filter
    rs = np.random.RandomState(11)
    tags = []
    for i in range(300):
        t = {'a': int(rs.randint(0, 10)), 'b': int(rs.randint(0, 10)), 'c': ['x', 'y', 'z'][rs.randint(0, 3)]}
        for key in list(t):
            if rs.rand() < 0.15:
                del t[key]
        tags.append(t)
    con = sqlite3.connect(':memory:')
    con.execute('CREATE TABLE t (_id INTEGER PRIMARY KEY, a INTEGER, b INTEGER, c TEXT)')
    con.executemany('INSERT INTO t VALUES (?, ?, ?, ?)', [(i, t.get('a'), t.get('b'), t.get('c')) for i, t in enumerate(tags)])

    def leaf():
        f = ['a', 'b', 'c'][rs.randint(0, 3)]
        if f == 'c':
            op = ['$eq', '$neq', '$in', '$nin'][rs.randint(0, 4)]
            val = ['x', 'y'][: rs.randint(1, 3)] if op in ('$in', '$nin') else ['x', 'y', 'z'][rs.randint(0, 3)]
            return {f: {op: val}}
        ops = {}
        for _ in range(rs.randint(1, 3)):
            op = ['$lt', '$gt', '$lte', '$gte', '$eq', '$neq', '$in', '$nin'][rs.randint(0, 8)]
            ops[op] = [int(v) for v in rs.randint(0, 10, size=rs.randint(1, 4))] if op in ('$in', '$nin') else int(rs.randint(0, 10))
        return {f: ops}

    def node(depth):
        r = rs.rand()
        if depth == 0 or r < 0.3:
            return leaf()
        logic = ['$and', '$or'][rs.randint(0, 2)]
        if r < 0.55:   # logic over a dict of conditions (distinct fields)
            d = {}
            for _ in range(rs.randint(1, 4)):
                d.update(leaf())
            return {logic: d}
        if r < 0.8:    # logic over a list of sub-filters
            return {logic: [node(depth - 1) for _ in range(rs.randint(1, 4))]}
        d = dict(leaf())  # conditions + a trailing logic group in one dict (flat clause: SQL precedence)
        d[logic] = [node(depth - 1) for _ in range(rs.randint(1, 3))]
        return d

    n_checked = 0
    for _ in range(300):
        flt = node(3)
        where, params = Filter(flt).parse_where_clause()
        want = [r[0] for r in con.execute(f'SELECT _id FROM t WHERE {where} ORDER BY _id', params)]
        assert select(tags, flt) == want, flt
        n_checked += 1
    assert n_checked == 300
