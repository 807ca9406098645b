"""CPU: the single-process multi-GPU index (annlite_amd/core/index/multi_gpu.py) behind the ``AnnLite`` facade, with two FAKE
devices -- shards that keep their rows in numpy and scan them with the oracle -- and the numpy restatement of the packed
merge kernel: the block-cyclic dealing of rows, the local <-> global id mapping, ties across shard boundaries, a shard
holding fewer than k rows, deletes on either shard, filters -- everything but the kernels (those: tests/test_gpu_parity.py
``test_multi_gpu_index_on_one_device``)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))


class FakeShard:
    """What MultiGpuPQIndex asks of a shard (PQFlatGpuIndex's methods), rows in numpy, scans by the oracle."""

    def __init__(self, codec, metric):
        import pq_oracle

        self.o = pq_oracle
        self._codec, self.metric = codec, metric
        self.codes = np.zeros((0, 0), np.uint8)
        self.valid = np.zeros((0,), bool)

    @property
    def codec(self):
        return self._codec[0] if isinstance(self._codec, list) else self._codec

    @property
    def size(self):
        return int(self.valid.sum())

    def reset(self):
        self.codes, self.valid = self.codes[:0], self.valid[:0]

    def add_with_ids(self, x, ids, **kw):
        ids = np.asarray(ids, np.int64)
        n = int(ids.max()) + 1
        if self.codes.shape[1] == 0:
            self.codes = np.zeros((0, self.codec.n_subvectors), np.uint8)
        if n > len(self.valid):
            self.codes = np.concatenate([self.codes, np.zeros((n - len(self.codes), self.codes.shape[1]), np.uint8)])
            self.valid = np.concatenate([self.valid, np.zeros(n - len(self.valid), bool)])
        self.codes[ids] = self.o.encode_c(np.asarray(x, np.float32), self.codec.codebooks)
        self.valid[ids] = True

    def delete(self, ids):
        self.valid[np.asarray(ids, np.int64)] = False

    def _scan(self, x, k, keep):
        lut = self.o.get_dist_mat_c(np.asarray(x, np.float32), self.codec.codebooks, self.o.EUCLIDEAN)
        rows = np.nonzero(keep)[0]
        if len(rows) == 0:
            return np.full((lut.shape[0], k), np.inf, np.float32), np.full((lut.shape[0], k), -1, np.int64)
        d, i = self.o.adc_search_c(lut, self.codes[rows], k)
        return d, np.where(i >= 0, rows[np.clip(i, 0, max(len(rows) - 1, 0))] if len(rows) else -1, -1)

    def search_batch_packed(self, x, k, row_base=0):
        d, i = self._scan(x, k, self.valid)
        out = np.empty(d.shape + (2,), np.int64)
        out[..., 0] = i
        out[..., 1] = d.astype(np.float32).view(np.uint32).astype(np.int64)
        return torch.from_numpy(out)

    def search_batch(self, x, limit=10, indices=None, **kw):
        keep = self.valid.copy()
        if indices is not None:
            sel = np.zeros_like(keep)
            sel[np.asarray(indices, np.int64)] = True
            keep &= sel
        d, i = self._scan(x, limit, keep)
        return np.sqrt(d), i  # EUCLIDEAN epilogue (hnsw/index.py:164-165)


@pytest.fixture()
def world():
    import pq_oracle
    from annlite_amd import Metric, PQCodec
    from annlite_amd.core.index.multi_gpu import MultiGpuPQIndex
    from annlite_amd.sharded import numpy_merge_packed

    rs = np.random.RandomState(5)
    D, M = 32, 8
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN)
    codec.set_codebooks_host(rs.randn(M, 256, D // M).astype(np.float32)) if hasattr(codec, 'set_codebooks_host') else None
    return pq_oracle, codec, MultiGpuPQIndex, numpy_merge_packed, rs


def _make(world, block=64, G=2):
    o, codec, MultiGpuPQIndex, merge, rs = world
    from annlite_amd import Metric

    return MultiGpuPQIndex(dim=32, pq_codec=codec, metric=Metric.EUCLIDEAN, devices=list(range(G)), block=block,
                           shard_factory=lambda g, dev: FakeShard(codec, Metric.EUCLIDEAN), merge_packed=merge)


def _codec_with_books(world):
    o, codec, *_ = world
    rs = np.random.RandomState(9)
    codec._codebooks = rs.randn(8, 256, 4).astype(np.float32)
    codec._is_trained = True
    return codec


def test_dealing_is_a_bijection(world):
    idx = _make(world, block=64, G=3)
    o = np.arange(0, 64 * 3 * 5 + 17)
    sh, loc = idx.shard_of(o)
    back = np.empty_like(o)
    for g in range(3):
        back[sh == g] = idx._global_ids(torch.from_numpy(loc[sh == g]), g).numpy()
    assert np.array_equal(back, o)
    for g in range(3):  # every shard's local rows are dense from 0
        assert np.array_equal(np.sort(loc[sh == g])[:64 * 5], np.arange(64 * 5))


def test_two_fake_devices_equal_one_flat_scan(world):
    o, codec, *_ = world
    codec = _codec_with_books(world)
    rs = np.random.RandomState(1)
    idx = _make(world, block=64, G=2)
    N, B, k = 64 * 7 + 5, 9, 10
    x = rs.randn(N, 32).astype(np.float32)
    x[70:90] = x[3]      # ties across the shard boundary (rows 0-63 -> shard 0, 64-127 -> shard 1)
    x[130:135] = x[3]
    idx.add_with_ids(x, np.arange(N))
    assert idx.size == N
    q = np.concatenate([x[3:4], rs.randn(B - 1, 32).astype(np.float32)])
    codes = o.encode_c(x, codec.codebooks)
    lut = o.get_dist_mat_c(q, codec.codebooks, o.EUCLIDEAN)
    rd, ri = o.adc_search_c(lut, codes, k)
    d, i = idx.search_batch(q, limit=k)
    assert np.array_equal(i, ri) and np.array_equal(d, np.sqrt(rd))
    # deletes on either shard
    gone = [3, 70, 71, 200, 64 * 6 + 1]
    idx.delete(gone)
    keep = np.ones(N, bool)
    keep[gone] = False
    rows = np.nonzero(keep)[0]
    rd, ri = o.adc_search_c(lut, codes[rows], k)
    d, i = idx.search_batch(q, limit=k)
    assert np.array_equal(i, rows[ri]) and np.array_equal(d, np.sqrt(rd))
    assert idx.size == N - len(gone)
    # a filter whose rows all live on ONE shard, and one spread over both (general path: merged on the final distances)
    for sel in (np.arange(64, 100), np.array([1, 2, 65, 66, 130, 300, 301])):
        sel = sel[keep[sel]]
        rd, ri = o.adc_search_c(lut, codes[sel], min(k, len(sel)))
        d, i = idx.search_batch(q, limit=min(k, len(sel)), indices=sel)
        assert np.array_equal(i, sel[ri]) and np.allclose(d, np.sqrt(rd), rtol=0, atol=0)
    # one query, reference signature
    d1, i1 = idx.search(q[0], limit=5)
    assert len(i1) == 5 and np.array_equal(i1, idx.search_batch(q[:1], limit=5)[1][0])


def test_a_shard_with_fewer_rows_than_k(world):
    o, codec, *_ = world
    codec = _codec_with_books(world)
    rs = np.random.RandomState(2)
    idx = _make(world, block=64, G=2)
    N, k = 64 + 3, 10  # shard 1 holds 3 rows
    x = rs.randn(N, 32).astype(np.float32)
    idx.add_with_ids(x, np.arange(N))
    q = rs.randn(4, 32).astype(np.float32)
    lut = o.get_dist_mat_c(q, codec.codebooks, o.EUCLIDEAN)
    rd, ri = o.adc_search_c(lut, o.encode_c(x, codec.codebooks), k)
    d, i = idx.search_batch(q, limit=k)
    assert np.array_equal(i, ri) and np.array_equal(d, np.sqrt(rd))
    # fewer rows than k in the WHOLE table: (+inf, -1) padding survives the merge
    idx2 = _make(world, block=64, G=2)
    idx2.add_with_ids(x[:4], np.arange(4))
    d, i = idx2.search_batch(q, limit=k)
    assert (i[:, 4:] == -1).all() and np.isinf(d[:, 4:]).all() and (i[:, :4] >= 0).all()


@pytest.mark.parametrize('G', [4, 8])
def test_four_and_eight_fake_devices_uneven_empty_and_short_shards(world, G):
    """Round 5: the single-process index over G = 4 / 8 devices (what ``AnnLite(devices=[0..7])`` builds on the 8-GPU node) --
    a table that does not divide into the block-cyclic deal, shards EMPTY (fewer blocks than devices), every shard shorter than
    k, k beyond the merge kernel's 64 (merge_lists_sorted); deletes that empty a whole shard."""
    o, codec, *_ = world
    codec = _codec_with_books(world)
    rs = np.random.RandomState(10 + G)
    B = 6
    q = rs.randn(B, 32).astype(np.float32)
    lut = o.get_dist_mat_c(q, codec.codebooks, o.EUCLIDEAN)
    for N in (64 * (2 * G + 1) + 7, 64 * 2 + 3, G + 1, 5):  # uneven / empty shards beyond the third / a few rows in shard 0 only
        x = rs.randn(N, 32).astype(np.float32)
        if N > 70:
            x[60:70] = x[1]  # ties across the first block boundary
        idx = _make(world, block=64, G=G)
        idx.add_with_ids(x, np.arange(N))
        assert idx.size == N
        codes = o.encode_c(x, codec.codebooks)
        for k in (10, 1, 70):
            rd, ri = o.adc_search_c(lut, codes, min(k, 64)) if k <= 64 else (None, None)
            if k > 64:  # beyond the oracle's lists: full sort by (distance, id)
                dd = np.stack([o.dist_pqcodes_to_codebooks_c(lut[b], codes) for b in range(B)])
                order = np.stack([np.lexsort((np.arange(N), dd[b]))[:k] for b in range(B)])
                ri = np.full((B, k), -1, np.int64)
                rd = np.full((B, k), np.inf, np.float32)
                ri[:, :order.shape[1]] = order
                rd[:, :order.shape[1]] = np.take_along_axis(dd, order, axis=1)
            d, i = idx.search_batch(q, limit=k)
            assert np.array_equal(i, ri) and np.array_equal(d, np.sqrt(rd)), (G, N, k)
        if N > 64 * 2:  # empty the whole second block (= all of shard 1's rows when the table has at most G blocks)
            gone = np.arange(64, 128)
            idx.delete(gone)
            keep = np.ones(N, bool)
            keep[gone] = False
            rows = np.nonzero(keep)[0]
            rd, ri = o.adc_search_c(lut, codes[rows], 10)
            d, i = idx.search_batch(q, limit=10)
            assert np.array_equal(i, np.where(ri >= 0, rows[np.clip(ri, 0, len(rows) - 1)], -1)) and np.array_equal(d, np.sqrt(rd)), (G, N)


def test_annlite_facade_over_two_fake_devices(world, tmp_path):
    """``AnnLite(..., devices=[0, 1])``: index() / search() / delete() / a filter through the reference's one-object API
    (annlite/index.py:274-359), the code table dealt over two (fake) devices -- results equal one flat oracle scan."""
    o, codec, MultiGpuPQIndex, merge, _ = world
    from annlite_amd import AnnLite, Metric
    from annlite_amd.index import Document, DocumentArray

    rs = np.random.RandomState(11)
    D, M, N = 32, 8, 300
    holder = []  # (the facade creates the codec the shards need)
    ann = AnnLite(D, metric='euclidean', n_subvectors=M, data_path=tmp_path / 'mg', devices=[0, 1], shard_block=64,
                  shard_factory=lambda g, dev: FakeShard(holder, Metric.EUCLIDEAN), merge_packed=merge)
    holder.append(ann._pq_codec)
    ann._pq_codec._codebooks = rs.randn(M, 256, D // M).astype(np.float32)
    ann._pq_codec._is_trained = True
    assert isinstance(ann.vec_index(0), MultiGpuPQIndex) and ann.vec_index(0).n_shards == 2
    x = rs.randn(N, D).astype(np.float32)
    docs = DocumentArray([Document(id=str(i), embedding=x[i], tags={'price': i % 7}) for i in range(N)])
    ann.index(docs)
    assert ann.index_size == N
    q = rs.randn(6, D).astype(np.float32)
    codes = o.encode_c(x, ann._pq_codec.codebooks)
    lut = o.get_dist_mat_c(q, ann._pq_codec.codebooks, o.EUCLIDEAN)
    rd, ri = o.adc_search_c(lut, codes, 10)
    qd = DocumentArray([Document(id=f'q{i}', embedding=q[i]) for i in range(len(q))])
    ann.search(qd, limit=10)
    for b, doc in enumerate(qd):
        assert [m.id for m in doc.matches] == [str(i) for i in ri[b]]
        assert np.allclose([m.scores['euclidean'].value for m in doc.matches], np.sqrt(rd[b]), rtol=0, atol=0)
    # lazy match lists resolve what they named WHEN THE SEARCH RAN, also after a delete (container.py:226-233 builds them eagerly)
    qd2 = DocumentArray([Document(id=f'p{i}', embedding=q[i]) for i in range(len(q))])
    ann.search(qd2, limit=10)
    assert not getattr(qd2[0].matches, 'materialised', False) and len(qd2[0].matches) == 10
    ann.delete([str(i) for i in (int(ri[0][0]), int(ri[1][0]))])
    assert [m.id for m in qd2[0].matches] == [str(i) for i in ri[0]] and qd2[0].matches[0].tags == {'price': int(ri[0][0]) % 7}
    assert [m.id for m in qd2[1].matches] == [str(i) for i in ri[1]]
    # ... and the tombstones live only as long as a handed-out list has not been read: the other queries' lists are still pending
    assert set(ann._tomb) == {int(ri[0][0]), int(ri[1][0])} and len(ann._live_resolvers) == 1
    for doc in qd2:
        doc.matches[0]
    del doc
    assert len(ann._live_resolvers) == 0
    ann.delete([])
    assert ann._tomb == {}
    keep = np.ones(N, bool)
    keep[[int(ri[0][0]), int(ri[1][0])]] = False
    rows = np.nonzero(keep)[0]
    rd2, ri2 = o.adc_search_c(lut, codes[rows], 10)
    dists, ids = ann.search_numpy(q, limit=10)
    for b in range(len(q)):
        assert np.array_equal(np.asarray(ids[b]), rows[ri2[b]]) and np.array_equal(dists[b], np.sqrt(rd2[b]))
    sel = np.array([i for i in rows if i % 7 < 2])
    rd3, ri3 = o.adc_search_c(lut, codes[sel], 5)
    qd = DocumentArray([Document(id=f'q{i}', embedding=q[i]) for i in range(len(q))])
    ann.search(qd, filter={'price': {'$lt': 2}}, limit=5)
    for b, doc in enumerate(qd):
        assert [m.id for m in doc.matches] == [str(i) for i in sel[ri3[b]]]
