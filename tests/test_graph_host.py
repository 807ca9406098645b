"""libannlite_graph.so (HNSW over PQ codes, host code; BASELINE config 5) -- no GPU needed.

Parity with the reference here is statistical (graph ids are build-order dependent):
edge distances must be the PQLookup sums bit-for-bit, recall of the candidate lists against the exhaustive
ADC ranking must be high, and -- when the reference's own hnsw_bind was built by oracle/build_ref.sh -- not worse
than the reference's on the same data and parameters."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def _data(n=20000, nq=200, D=64, M=8, Ks=256, seed=0, unit=False):
    rs = np.random.RandomState(seed)
    A = rs.randn(12, D).astype(np.float32)
    x = (rs.randn(n, 12).astype(np.float32) @ A + 0.05 * rs.randn(n, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(nq, 12).astype(np.float32) @ A + 0.05 * rs.randn(nq, D).astype(np.float32)).astype(np.float32)
    if unit:  # cosine indexes see unit rows (hnsw/index.py:28-29)
        x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
        q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    # codebooks: per sub-space, Ks training points (a crude but valid codebook)
    cb = np.stack([x[rs.choice(n, Ks, replace=False), m * (D // M):(m + 1) * (D // M)] for m in range(M)]).astype(np.float32)
    return np.ascontiguousarray(x), np.ascontiguousarray(q), np.ascontiguousarray(cb)


def _graph(gc, cb, cap, mc=16, efc=200):
    M, Ks, dsub = cb.shape
    h = gc.lib().annlite_hnsw_create(cb.ctypes.data, M, Ks, dsub, cap, mc, efc, 100)
    assert h, gc.lib().annlite_hnsw_last_error()
    return ctypes.c_void_p(h)


def _search(gc, g, q, ef, threads=0):
    B = q.shape[0]
    ids = np.empty((B, ef), np.int64)
    d = np.empty((B, ef), np.float32)
    gc.check(gc.lib().annlite_hnsw_search(g, q.ctypes.data, B, ef, ids.ctypes.data, d.ctypes.data, threads), 'search')
    return ids, d


def test_graph_library_symbols():
    from annlite_amd import _graph_capi as gc

    L = gc.lib()
    hdr = open(os.path.join(ROOT, 'include', 'annlite_graph.h')).read()
    for s in gc.SYMBOLS:
        assert hasattr(L, s) and s in hdr


@pytest.mark.parametrize('kind', [1, 3])
def test_candidates_recall_and_pqlookup_bits(kind):
    """kind 1: EUCLIDEAN index; kind 3: COSINE index (unit rows, tables 1/Ks - dot).  The graph always walks
    with L2 tables; what must hold is that the exhaustive ADC top-10 UNDER THE INDEX METRIC is inside the
    ef_search = 128 candidates (the GPU then ranks the candidates with the metric's own tables)."""
    import pq_oracle
    from annlite_amd import _graph_capi as gc

    x, q, cb = _data(unit=(kind == 3))
    codes = pq_oracle.encode_c(x, cb)
    g = _graph(gc, cb, len(x))
    labels = np.arange(len(x), dtype=np.int64)
    gc.check(gc.lib().annlite_hnsw_add(g, x.ctypes.data, codes.ctypes.data, labels.ctypes.data, len(x), 0), 'add')
    assert gc.lib().annlite_hnsw_size(g) == len(x)
    ids, d = _search(gc, g, q, 128)
    # exhaustive ADC ranking under the index metric (oracle = the reference's arithmetic)
    M, Ks, dsub = cb.shape
    lut_l2 = pq_oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    lut = lut_l2 if kind == 1 else pq_oracle.get_dist_mat_c(q, cb, 3)
    td, ti = pq_oracle.adc_search_c(lut, codes, 10)
    rec = np.mean([len(set(ids[b]) & set(ti[b])) / 10 for b in range(len(q))])
    assert rec >= 0.95, rec
    if kind == 1:
        rec10 = np.mean([len(set(ids[b, :10]) & set(ti[b])) / 10 for b in range(len(q))])
        assert rec10 >= 0.9, rec10
    # the distance the graph reports for a node IS the PQLookup sum over its L2 table (fp32, sub-space order)
    for b in range(0, len(q), 17):
        ok = ids[b] >= 0
        want = pq_oracle.dist_pqcodes_to_codebooks_c(lut_l2[b], codes[ids[b][ok]])
        assert np.array_equal(d[b][ok], want)
        assert np.all(np.diff(d[b][ok]) >= 0)
    # deletions are never returned; save / load round trip gives the same candidates
    dead = ids[0, :5].copy()
    for i in dead:
        gc.check(gc.lib().annlite_hnsw_mark_deleted(g, int(i)), 'delete')
    ids2, _ = _search(gc, g, q[:1], 128)
    assert not (set(dead.tolist()) & set(ids2[0].tolist()))
    path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'annlite_graph_test_%d.bin' % os.getpid())
    gc.check(gc.lib().annlite_hnsw_save(g, path.encode()), 'save')
    g2 = ctypes.c_void_p(gc.lib().annlite_hnsw_load(path.encode()))
    os.unlink(path)
    ids3, d3 = _search(gc, g2, q, 64, threads=1)
    ids4, d4 = _search(gc, g, q, 64, threads=1)
    assert np.array_equal(ids3, ids4) and np.array_equal(d3, d4)
    gc.lib().annlite_hnsw_free(g)
    gc.lib().annlite_hnsw_free(g2)


def test_recall_not_worse_than_reference_hnsw_pq():
    """Same data, codebooks, max_connection / ef_construction / ef_search through the reference's own
    HnswIndex(pq_codec=...) (oracle/build_ref.sh build of hnsw_bind; skipped where /root/reference is absent)."""
    ref = pytest.importorskip('ref_import', reason='reference not available')
    try:
        mods = ref.load()
    except Exception as e:  # pragma: no cover
        pytest.skip('reference not importable: %r' % (e,))
    import pq_oracle
    from annlite_amd import _graph_capi as gc

    x, q, cb = _data(n=8000, nq=100)
    M, Ks, dsub = cb.shape
    codec = mods.PQCodec(dim=x.shape[1], n_subvectors=M, n_clusters=Ks, metric=mods.Metric.EUCLIDEAN)
    codec._codebooks = cb.copy()
    codec._is_trained = True
    idx = mods.HnswIndex(dim=x.shape[1], metric=mods.Metric.EUCLIDEAN, pq_codec=codec, ef_construction=200, ef_search=64,
                         max_connection=16, initial_size=len(x))
    idx.add_with_ids(x, list(range(len(x))))
    codes = pq_oracle.encode_c(x, cb)
    lut = pq_oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    _, ti = pq_oracle.adc_search_c(lut, codes, 10)
    ref_rec = np.mean([len(set(idx.search(q[b], limit=10)[1].tolist()) & set(ti[b])) / 10 for b in range(len(q))])
    g = _graph(gc, cb, len(x))
    labels = np.arange(len(x), dtype=np.int64)
    gc.check(gc.lib().annlite_hnsw_add(g, x.ctypes.data, codes.ctypes.data, labels.ctypes.data, len(x), 0), 'add')
    ids, _ = _search(gc, g, q, 64)
    our_rec = np.mean([len(set(ids[b, :10]) & set(ti[b])) / 10 for b in range(len(q))])
    gc.lib().annlite_hnsw_free(g)
    assert our_rec >= ref_rec - 0.03, (our_rec, ref_rec)


def test_save_load_delete_and_export_roundtrip(tmp_path):
    """hnsw/index.py:116-122 analogue (dump / load of the graph), mark_deleted, and the link export the GPU walk
    consumes: a reloaded graph answers exactly like the one that was saved."""
    import pq_oracle
    from annlite_amd import _graph_capi as gc

    x, q, cb = _data(n=6000, nq=40)
    codes = pq_oracle.encode_c(x, cb)
    L = gc.lib()
    g = _graph(gc, cb, 6000)
    labels = np.arange(6000, dtype=np.int64)
    gc.check(L.annlite_hnsw_add(g, x.ctypes.data, codes.ctypes.data, labels.ctypes.data, 6000, 1), 'add')  # 1 thread: reproducible
    assert L.annlite_hnsw_size(g) == 6000
    ids0, d0 = _search(gc, g, q, 64, threads=2)
    path = str(tmp_path / 'g.bin').encode()
    gc.check(L.annlite_hnsw_save(g, path), 'save')
    h = L.annlite_hnsw_load(path)
    assert h, L.annlite_hnsw_last_error()
    g2 = ctypes.c_void_p(h)
    ids1, d1 = _search(gc, g2, q, 64, threads=2)
    assert np.array_equal(ids0, ids1) and np.array_equal(d0, d1)
    # deleted labels never come back, in either copy
    gone = np.unique(ids0[:, 0])
    for lab in gone:
        gc.check(L.annlite_hnsw_mark_deleted(g2, int(lab)), 'mark_deleted')
    ids2, _ = _search(gc, g2, q, 64)
    assert not np.isin(ids2, gone).any()
    # export: (count, ids) per row, counts within the level-0 degree, ids in range, seeds distinct
    lpn = L.annlite_hnsw_links_per_node(g)
    links = np.zeros((6000, lpn + 1), np.uint32)
    seeds = np.full(1024, -1, np.int64)
    n_seeds = ctypes.c_int64(0)
    gc.check(L.annlite_hnsw_export(g, 6000, links.ctypes.data, seeds.ctypes.data, 1024, ctypes.byref(n_seeds)), 'export')
    assert links[:, 0].max() <= lpn and links[:, 0].min() >= 1
    for r in (0, 17, 5999):
        row = links[r, 1:1 + links[r, 0]]
        assert (row < 6000).all() and r not in row and len(set(row.tolist())) == len(row)
    ns = n_seeds.value
    assert 1 <= ns <= 1024 and len(set(seeds[:ns].tolist())) == ns and (seeds[:ns] < 6000).all()
    L.annlite_hnsw_free(g)
    L.annlite_hnsw_free(g2)
