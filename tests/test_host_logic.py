"""CPU: host-side logic that needs no GPU (enums, filters, docarray stand-in, shard ranges,
constructor/assertion behaviour mirrored from the reference)."""
import numpy as np
import pytest


def test_enums_match_reference_values():
    from annlite_amd.enums import ExpandMode, Metric

    assert (int(Metric.EUCLIDEAN), int(Metric.INNER_PRODUCT), int(Metric.COSINE)) == (1, 2, 3)  # enums.py:25-28
    assert Metric.from_string('cosine') is Metric.COSINE and str(Metric.EUCLIDEAN) == 'EUCLIDEAN'
    assert int(ExpandMode.STEP) == 1
    with pytest.raises(ValueError):
        Metric.from_string('manhattan')


def test_codec_constructor_contract():
    from annlite_amd import Metric, PQCodec

    with pytest.raises(AssertionError):
        PQCodec(dim=130, n_subvectors=8)  # pq.py:51-53
    c = PQCodec(dim=128, n_subvectors=16, n_clusters=256, metric=Metric.COSINE)
    assert c.d_subvector == 8 and c.code_dtype == np.uint8 and c.normalize_input and not c.is_trained
    assert PQCodec(dim=64, n_subvectors=8, n_clusters=512).code_dtype == np.uint16  # pq.py:56-60
    assert PQCodec(dim=64, n_subvectors=8, n_clusters=70000).code_dtype == np.uint32
    assert c.get_subspace_splitting() == (16, 256, 8)
    assert c.get_codebook().shape == (16, 256, 8) and c.get_codebook().dtype == np.float32
    assert hash(c) == hash(PQCodec(dim=128, n_subvectors=16, n_clusters=256, metric=Metric.COSINE))
    c.set_codebooks(np.ones((16, 256, 8), np.float32))
    assert c.is_trained


def test_codec_pickle_roundtrip(tmp_path):
    from annlite_amd import Metric, PQCodec
    from annlite_amd.core.codec.base import BaseCodec

    c = PQCodec(dim=32, n_subvectors=4, n_clusters=16, metric=Metric.INNER_PRODUCT)
    c.set_codebooks(np.random.RandomState(0).rand(4, 16, 8).astype(np.float32))
    p = tmp_path / 'pq_codec.params'
    c.dump(p)
    c2 = BaseCodec.load(p)
    assert c2.is_trained and c2.metric == Metric.INNER_PRODUCT and np.array_equal(c2.codebooks, c.codebooks)


def test_annlite_constructor_and_untrained_errors(tmp_path):
    from annlite_amd import AnnLite
    from annlite_amd.index import Document, DocumentArray

    with pytest.raises(AssertionError):
        AnnLite(100, n_subvectors=8, data_path=tmp_path / 'a')  # index.py:86-89
    ivf = AnnLite(128, n_subvectors=8, n_cells=4, n_probe=2, data_path=tmp_path / 'b')  # index.py:125-133
    assert ivf.n_probe == 4 and not ivf.is_trained and ivf.stat['n_cells'] == 4  # index.py:94: max(n_probe, n_cells)
    with pytest.raises(NotImplementedError):
        AnnLite(128, n_subvectors=8, n_components=16, data_path=tmp_path / 'b2')
    ann = AnnLite(64, n_subvectors=8, data_path=tmp_path / 'c', dim=64)
    assert not ann.is_trained and ann.stat['total_docs'] == 0 and ann.stat['metric'] == 'COSINE'
    docs = DocumentArray([Document(id=str(i), embedding=np.zeros(64, np.float32)) for i in range(3)])
    with pytest.raises(RuntimeError):
        ann.index(docs)  # tests/test_pq_index.py:52-65
    with pytest.raises(RuntimeError):
        ann.search(docs)
    with pytest.raises(RuntimeError):
        ann.search_numpy(np.zeros((1, 64), np.float32))
    ro = AnnLite(64, n_subvectors=8, data_path=tmp_path / 'd', read_only=True)
    assert ro.index(docs) is None  # logs and returns, index.py:280-282


def test_filter_and_docarray_standin():
    from annlite_amd.docarray_compat import Document, DocumentArray
    from annlite_amd.filter import match, select

    tags = [{'price': 10, 'cat': 'a'}, {'price': 50, 'cat': 'b'}, None, {'price': 70, 'cat': 'a'}]
    assert select(tags, {'price': {'$lt': 60}}) == [0, 1]
    assert select(tags, {'$and': [{'cat': {'$eq': 'a'}}, {'price': {'$gte': 20}}]}) == [3]
    assert select(tags, {'$or': [{'price': {'$gt': 60}}, {'cat': {'$in': ['b']}}]}) == [1, 3]
    assert match({'x': 1}, {'x': 1}) and not match({'x': 1}, {'x': {'$ne': 1}})
    with pytest.raises(ValueError):
        select(tags, {'price': {'$regex': 'x'}})
    da = DocumentArray([Document(id=str(i), embedding=np.full(4, i, np.float32)) for i in range(5)])
    assert da.embeddings.shape == (5, 4) and da[:, 'id'] == ['0', '1', '2', '3', '4']
    d = da[1]
    d.scores['cosine'].value = 0.5
    assert d.scores['cosine'].value == 0.5 and len(d.matches) == 0 and '3' in da


def test_shard_ranges_cover_exactly():
    from annlite_amd.sharded import shard_range

    for n, g in ((10_000_000, 8), (1000, 3), (7, 8), (0, 2)):
        rs = [shard_range(n, g, r) for r in range(g)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:])) and all(lo <= hi for lo, hi in rs)


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from annlite_amd import ops

    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.device()
    from annlite_amd import Metric, PQCodec

    c = PQCodec(dim=8, n_subvectors=2, n_clusters=4).set_codebooks(np.zeros((2, 4, 4), np.float32))
    with pytest.raises(RuntimeError):
        c.encode(np.zeros((3, 8), np.float32))


def test_annlite_filter_and_get_docs_host_logic(tmp_path):
    """index.py:389-456: document filtering without a vector search (host-side table only, no GPU)."""
    from annlite_amd import AnnLite
    from annlite_amd.index import Document

    ann = AnnLite(64, n_subvectors=8, data_path=tmp_path / 'f')
    for i in range(10):  # fill the in-memory table the way index() does
        d = Document(id=str(i), embedding=np.zeros(64, np.float32), tags={'price': 10 - i, 'cat': i % 3})
        ann._offset2id.append(d.id)
        ann._id2offset[d.id] = i
        ann._tags.append(dict(d.tags))
        ann._docs[d.id] = d
    got = ann.filter({'cat': {'$eq': 1}}, limit=10)
    assert [d.id for d in got] == ['1', '4', '7']
    got = ann.filter({'price': {'$gte': 5}}, limit=2, offset=1, order_by='price', ascending=True)
    assert [d.id for d in got] == ['4', '3']  # prices 5,6,7,.. -> ids 5,4,3: skip one, take two
    assert [d.id for d in ann.get_docs(limit=3)] == ['0', '1', '2']
    assert len(ann.get_docs(limit=-1)) == 10
    assert [d.id for d in ann.filter({}, limit=2, order_by='price', ascending=False, include_metadata=False)] == ['0', '1']


def test_filter_grammar_of_the_reference():
    """tests/test_filter.py:12-110 restated on documents instead of SQL strings: logic operators with dict and
    list operands, several operators on one field, membership, field-level $or, unsupported operators."""
    from annlite_amd.filter import match, select

    docs = [{'brand': 0, 'price': 60}, {'brand': 2, 'price': 60}, {'brand': 0, 'price': 10}, {'brand': 5, 'price': 5}, {'price': 70}]
    assert select(docs, {}) == [0, 1, 2, 3, 4]                                            # test_empty_filter
    assert select(docs, {'brand': {'$lt': 1}}) == [0, 2]                                  # test_simple_filter (NULL: no)
    both = [0]
    assert select(docs, {'$and': {'brand': {'$lt': 1}, 'price': {'$gte': 50}}}) == both   # test_logic_operator
    assert select(docs, {'brand': {'$lt': 1}, 'price': {'$gte': 50}}) == both
    assert select(docs, {'$or': {'brand': {'$lt': 1}, 'price': {'$gte': 50}}}) == [0, 1, 2, 4]
    shoes = [{'brand': 'Nike', 'price': 40}, {'brand': 'Gucci', 'price': 90}, {'brand': 'Puma', 'price': 20}, {'brand': 'Puma', 'price': 55}]
    assert select(shoes, {'$and': {'brand': {'$in': ['Nike', 'Gucci']}, 'price': {'$gte': 50}}}) == [1]   # test_membership_operator
    assert select(shoes, {'$or': {'brand': {'$nin': ['Nike', 'Gucci']}, 'price': {'$gte': 50}}}) == [1, 2, 3]
    cars = [{'price': 30, 'rating': 2, 'year': 2008}, {'price': 60, 'rating': 2, 'year': 2008}, {'price': 30, 'rating': 0, 'year': 2008},
            {'price': 30, 'rating': 2, 'year': 2012}]
    rng = {'$and': {'price': {'$gte': 0, '$lte': 54}, 'rating': {'$gte': 1}, 'year': {'$gte': 2007, '$lte': 2010}}}
    assert select(cars, rng) == [0]                                                       # test_cases
    either = {'$and': {'price': {'$or': [{'price': {'$gte': 55}}, {'price': {'$lte': 20}}]}, 'rating': {'$gte': 1},
                       'year': {'$gte': 2007, '$lte': 2010}}}
    assert select(cars, either) == [1]
    either2 = {'$and': {'$or': [{'price': {'$gte': 55}}, {'price': {'$lte': 20}}], 'rating': {'$gte': 1}, 'year': {'$gte': 2007, '$lte': 2010}}}
    assert select(cars, either2) == [1]
    with pytest.raises(ValueError):                                                       # test_error_filter
        select(cars, {'$may': {'brand': {'$lt': 1}, 'price': {'$gte': 50}}})
    with pytest.raises(ValueError):
        match(cars[0], {'price': {'$between': [1, 2]}})
    assert not match({}, {'price': {'$neq': 3}}) and not match({}, {'price': {'$nin': [3]}})  # NULL satisfies nothing
    # SQL precedence of the flat clause: a AND b OR c
    assert match({'a': 0, 'b': 0, 'c': 1}, {'a': {'$eq': 1}, 'b': {'$eq': 1}, '$or': {'c': {'$eq': 1}}})


def test_update_and_delete_of_unknown_documents(tmp_path):
    """tests/test_crud.py:66-105: what update / delete do with ids the index does not hold (container.py:349-365,
    389-405) -- decided on the host before anything reaches the GPU."""
    from annlite_amd import AnnLite
    from annlite_amd.index import Document, DocumentArray

    ann = AnnLite(64, n_subvectors=8, data_path=tmp_path / 'u')
    ann._pq_codec._is_trained = True  # (host logic only: nothing below touches the codebooks)
    ghosts = DocumentArray([Document(id=f'{i}_wrong', embedding=np.zeros(64, np.float32)) for i in range(3)])
    with pytest.raises(Exception):
        ann.update(ghosts, raise_errors_on_not_found=True, insert_if_not_found=False)
    with pytest.warns(RuntimeWarning):
        ann.update(ghosts, raise_errors_on_not_found=False, insert_if_not_found=False)
    assert ann.total_docs == 0
    with pytest.raises(Exception):
        ann.delete(ghosts, raise_errors_on_not_found=True)
    ann.delete(ghosts)  # silently ignored
    ann.delete(['nope'])


def test_seed_rows_are_spread_over_the_table():
    """scan_prep.hip: seed_bound_kernel draws ceil(S / 64) blocks of 64 rows in runs of 2^c blocks, the runs spread evenly over the
    extent (block_row).  The mapping restated: every block starts inside the extent on a multiple of 64 (the lanes keep their skew
    residues), no two blocks overlap, the last run starts in the last 1/n_runs of the table (its tail is sampled), and S >= extent
    degenerates to the contiguous scan."""
    def block_rows(S, ext, c):
        S = min(S, ext)
        n_blocks = (S + 63) >> 6
        mask = (1 << c) - 1
        run_step = ((ext >> 6) // ((n_blocks + mask) >> c)) << 6
        run_step = max(run_step, 64 << c)
        return [(b >> c) * run_step + ((b & mask) << 6) for b in range(n_blocks)], run_step

    for ext in (4_100, 65_600, 1_250_000, 10_000_000, 9_999_937):
        for S in (100, 8192, 32768, 131072, ext, ext + 5):
            for c in (0, 3, 6):
                rows, step = block_rows(S, ext, c)
                assert all(r % 64 == 0 for r in rows)
                assert len(set(rows)) == len(rows)
                srt = sorted(rows)
                assert all(b - a >= 64 for a, b in zip(srt, srt[1:]))
                inside = [r for r in rows if r < ext]
                assert len(inside) >= len(rows) - (1 << c), (ext, S, c)  # (only a last partial run may stick out: its lanes are masked)
                if min(S, ext) * 2 <= ext and len(rows) >= (2 << c):
                    assert max(inside) >= ext - 2 * step - 64 * len(rows), (ext, S, c)  # (the run step is rounded down to blocks)
                if S >= ext:
                    assert rows == list(range(0, ((ext + 63) >> 6) << 6, 64))


def test_graph_index_build_side_is_resolved_without_a_gpu():
    """HnswPQGpuIndex(build=None): the GPU builds the level-0 graph where the GPU walk applies (max_connection <= 16, M in {8, 16, 32},
    uint8 codes, walk='gpu' -- M in {8, 16, 32, 64}); everything else keeps the host library -- decided at construction, no device touched."""
    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    def mk(m=8, ks=256, dim=64, **kw):
        return HnswPQGpuIndex(dim=dim, metric=Metric.EUCLIDEAN, pq_codec=PQCodec(dim=dim, n_subvectors=m, n_clusters=ks), **kw)

    assert mk().build == 'gpu' and mk().expand_width == 2
    assert mk(walk='host').build == 'host'
    assert mk(max_connection=32).build == 'host'
    assert mk(m=64, dim=128).build == 'gpu'   # (round 6: 64 sub-spaces walk and build on the GPU too)
    assert mk(m=4, dim=64).build == 'host'    # (no walk kernel for this width)
    assert mk(ks=512).build == 'host'
    assert mk(build='host').build == 'host' and mk(m=16, build='gpu').build == 'gpu'
    with pytest.raises(AssertionError):
        mk(build='cpu')
    with pytest.raises(AssertionError):
        mk(expand_width=3)
