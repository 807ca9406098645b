"""CPU tests of round 4's host-side pieces: bench.py's self-launch command line, the lazy match lists of
``AnnLite.search``, the any-k merge of the single-process multi-GPU index, the oracle's NaN order."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


# ------------------------------------------------------------------ bench.py --gpus N without a launcher
def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_reexec_command_line():
    """`python bench.py --gpus 8 --steps K --warmup W` (the form the driver uses for N = 1) must become the N-rank job the
    contract describes: torch.distributed.run, one node, N processes, rendezvous on 127.0.0.1, the arguments unchanged."""
    bench = _bench_module()
    argv = ['--gpus', '8', '--steps', '50', '--warmup', '5']
    cmd = bench.torchrun_argv(8, argv, 29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == argv
    port = bench.free_port()
    assert 1024 < port < 65536


def test_bench_main_reexecs_itself_when_no_launcher(monkeypatch):
    bench = _bench_module()
    seen = {}

    def fake_execv(exe, cmd):
        seen['exe'], seen['cmd'] = exe, cmd
        raise SystemExit(0)

    for var in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(bench.os, 'execv', fake_execv)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen['exe'] == sys.executable and '--nproc-per-node=4' in seen['cmd']
    assert seen['cmd'][-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']


def test_bench_does_not_reexec_under_a_launcher(monkeypatch):
    """Under torchrun (RANK / WORLD_SIZE set) bench.py must NOT re-exec: it IS a rank."""
    bench = _bench_module()
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setattr(bench.os, 'execv', lambda *a: (_ for _ in ()).throw(AssertionError('re-exec under a launcher')))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4'])

    class Stop(Exception):
        pass

    def stop(*a, **k):
        raise Stop()

    monkeypatch.setattr(bench.torch.cuda, 'set_device', stop)  # (first thing main() does as a rank)
    with pytest.raises(Stop):
        bench.main()


# ------------------------------------------------------------------ lazy match lists
def _resolver(calls):
    from annlite_amd.docarray_compat import Document

    def resolve(offs, dists):
        calls.append(len(offs))
        out = []
        for o, d in zip(offs, dists):
            doc = Document(id=str(int(o)))
            doc.scores['euclidean'].value = d
            out.append(doc)
        return out

    return resolve


def test_lazy_matches_build_documents_on_first_use_only():
    from annlite_amd.docarray_compat import Document, DocumentArray, LazyMatches

    calls = []
    m = LazyMatches(np.array([3, 5, 7]), np.array([.1, .2, .3], np.float32), _resolver(calls))
    assert len(m) == 3 and bool(m) and not calls and not m.materialised  # len() builds nothing
    assert m[0].id == '3' and m[0].scores['euclidean'].value == np.float32(.1) and calls == [3] and m.materialised
    assert [d.id for d in m] == ['3', '5', '7'] and m[:, 'id'] == ['3', '5', '7'] and '5' in m and calls == [3]
    assert isinstance(m, DocumentArray) and isinstance(m[1:], DocumentArray)
    m2 = LazyMatches(np.array([1]), np.array([.5], np.float32), _resolver(calls))
    m2.append(Document(id='z'))
    assert len(m2) == 2 and m2[1].id == 'z'
    m3 = LazyMatches(np.array([1, 2]), np.array([.5, .6], np.float32), _resolver(calls))
    p = pickle.loads(pickle.dumps(m3))
    assert type(p) is DocumentArray and [d.id for d in p] == ['1', '2']
    empty = LazyMatches(np.empty((0,), np.int64), np.empty((0,), np.float32), _resolver(calls))
    assert not empty and len(empty) == 0 and list(empty) == []
    a = LazyMatches(np.array([1, 2]), np.array([.5, .6], np.float32), _resolver(calls))
    assert [d.id for d in reversed(a)] == ['2', '1'] and a.index(a[1]) == 1


def test_facade_result_rows_and_vectorised_ids():
    """``_valid_rows`` cuts every query's row at its first missing entry; ``search_numpy`` maps offsets to int(doc id) with ONE
    gather (container.py:260) and falls back to the per-id conversion (and its ValueError) for ids that are not integers."""
    from annlite_amd.index import AnnLite

    d = np.array([[.1, .2, np.inf], [.3, np.inf, np.inf]], np.float32)
    i = np.array([[4, 1, -1], [1, -1, -1]], np.int64)
    dd, ii = AnnLite._valid_rows(d, i)
    assert [x.tolist() for x in ii] == [[4, 1], [1]] and [len(x) for x in dd] == [2, 1]
    full_d, full_i = AnnLite._valid_rows(d[:, :1], i[:, :1])
    assert [x.tolist() for x in full_i] == [[4], [1]]

    class Stub(AnnLite):
        def __init__(self, ids, d, i):
            self._offset2id, self._d, self._i = ids, d, i

        is_trained = True

        def _search_arrays(self, q, f, k):
            return self._d, self._i

    s = Stub(['10', '11', None, '13', '14'], d, i)
    dists, ids = s.search_numpy(np.zeros((2, 4), np.float32), limit=3)
    assert [x.tolist() for x in ids] == [[14, 11], [11]] and ids[0].dtype == np.dtype(int)
    assert np.array_equal(dists[0], d[0, :2])
    s2 = Stub(['a', 'b', 'c', 'd', 'e'], d, i)
    with pytest.raises(ValueError):
        s2.search_numpy(np.zeros((2, 4), np.float32), limit=3)


# ------------------------------------------------------------------ merge of G lists for any k (multi-GPU index, limit > 64)
@pytest.mark.parametrize('k', [5, 100])
def test_merge_lists_sorted_equals_the_lexsort_merge(k):
    from annlite_amd.core.index.multi_gpu import merge_lists_sorted
    from annlite_amd.sharded import numpy_merge

    rs = np.random.RandomState(k)
    G, B = 3, 7
    d = np.sort(rs.randint(0, 12, size=(G, B, k)).astype(np.float32), axis=2)  # heavy ties, within and across lists
    ids = np.stack([np.stack([np.sort(rs.choice(10_000, size=k, replace=False)) * G + g for _ in range(B)]) for g in range(G)]).astype(np.int64)
    short = rs.randint(0, k, size=(G, B))  # every list ends in padding of its own length
    for g in range(G):
        for b in range(B):
            d[g, b, k - short[g, b]:] = np.inf
            ids[g, b, k - short[g, b]:] = -1
    d[0, 0, :] = np.inf
    ids[0, 0, :] = -1
    td, ti = torch.from_numpy(d), torch.from_numpy(ids)
    od, oi = merge_lists_sorted(td, ti, k)
    rd, ri = numpy_merge(td, ti)
    rd, ri = rd.numpy(), ri.numpy()
    ri = np.where(np.isinf(rd), -1, ri)  # (numpy_merge keeps a padding entry's id slot as it sorted it)
    assert np.array_equal(od.numpy(), rd) and np.array_equal(oi.numpy(), ri)
    od2, oi2 = merge_lists_sorted(td[:, :, :3], ti[:, :, :3], 20)  # fewer entries than k: padded
    assert od2.shape == (B, 20) and bool((oi2[:, 9:] == -1).all()) and bool(torch.isinf(od2[:, 9:]).all())


# ------------------------------------------------------------------ the oracle's order with non-finite distances
def test_oracle_topk_sorts_nan_last_like_numpy(oracle):
    """math.py:94-120 selects with argpartition / argsort: numpy puts NaN behind every number (+inf included), whatever its
    sign bit.  The oracle -- and through it the kernels -- use that order, ties (all NaNs tie) by row id."""
    neg_nan = np.array([0xFFC00000], dtype=np.uint32).view(np.float32)[0]
    v = np.array([3.0, np.nan, -np.inf, np.inf, neg_nan, 1.0, np.inf, -2.5, np.nan, 0.0], dtype=np.float32)
    for k in (1, 4, 7, 10, 12):
        d, i = oracle.top_k_c(v, k)
        order = np.argsort(v, kind='stable')[:k]
        assert np.array_equal(i[:len(order)], order)
        assert np.array_equal(d[:len(order)], v[order], equal_nan=True)
        dn, in_ = oracle.top_k_numpy(v, k)
        assert np.array_equal(in_, i) and np.array_equal(dn, d, equal_nan=True)
    d, i = oracle.top_k_c(np.full((50,), np.nan, np.float32), 5)
    assert i.tolist() == [0, 1, 2, 3, 4] and np.isnan(d).all()


# ------------------------------------------------------------------ seed exchange: what every rank must agree on without talking
def test_split_search_conditions_are_static_configuration():
    """``split_supported`` decides whether the ranks of a row-sharded search run the seed collective at all: it may depend on the
    configuration, the batch's shape and the input kind only -- never on how many rows THIS rank holds (a short last shard
    answers ``keys is None`` inside the protocol instead, sharded.py ``_split_search``)."""
    import torch

    from annlite_amd import Metric, PQCodec
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex

    def index(dim=128, m=16, metric=Metric.EUCLIDEAN, rerank=False, ks=256):
        return PQFlatGpuIndex(dim=dim, pq_codec=PQCodec(dim=dim, n_subvectors=m, n_clusters=ks, metric=metric), metric=metric,
                              rerank=rerank)

    x = torch.zeros((4, 128))
    a, b = index(), index()
    b._n_rows = 10  # (no storage touched: the answer must not look at it)
    assert a.split_supported(x, 10) and b.split_supported(x, 10) and a.split_supported(x, 16) and a.split_supported(x, 1)
    assert not a.split_supported(x, 17) and not a.split_supported(x, 0)
    assert not a.split_supported(x.numpy(), 10) and not a.split_supported(x[:0], 10) and not a.split_supported(x[0], 10)
    assert not index(m=32).split_supported(x, 10) and not index(m=8).split_supported(x, 10)
    assert not index(metric=Metric.COSINE).split_supported(x, 10) and not index(metric=Metric.INNER_PRODUCT).split_supported(x, 10)
    assert not index(dim=512, m=16).split_supported(torch.zeros((4, 512)), 10)  # (dim > 256: the preparation launch's LDS)
    assert not index(dim=96, m=16).split_supported(torch.zeros((4, 96)), 10)  # (6-dim sub-vectors: not a multiple of 4)
    assert index(dim=64, m=16).split_supported(torch.zeros((4, 64)), 10)


def test_seed_rows_deal_the_single_gpu_seed_over_the_ranks():
    """``ShardedPQIndex.seed_rows``: the single-GPU rule (N / 32 clamped to [8192, 32768], scan.hip plan) on the WHOLE table, divided
    by the ranks (emulated peers count), in multiples of 1024 and not below 4096."""
    import torch
    import torch.distributed as dist

    from annlite_amd.sharded import SEED_KEYS, ShardedPQIndex

    class Shard:
        _n_rows = 1_250_000
        sqrt_epilogue = True

    port = _bench_module().free_port()
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        ident = lambda *a, **k: None  # noqa: E731  (merges are not called)
        s = ShardedPQIndex(Shard(), row_base=0, merge=ident, merge_packed=ident, seed_exchange=True)
        assert s.seed_rows() == 32768  # one rank, 1.25M rows: the single-GPU seed
        s._peer_keys = torch.zeros((7, 4, SEED_KEYS), dtype=torch.int64)
        assert s.seed_rows() == 4096  # 8 ranks x 1.25M = 10M rows: 32768 / 8
        s._peer_keys = torch.zeros((1, 4, SEED_KEYS), dtype=torch.int64)
        assert s.seed_rows() == 16384
        s.n_total = 200_000  # a small table: 8192 rows in all, never below 4096 per rank
        assert s.seed_rows() == 4096
        s._peer_keys, s.n_total = None, 400_000
        assert s.seed_rows() == 13312  # ceil(12500 / 1024) * 1024
    finally:
        dist.destroy_process_group()
