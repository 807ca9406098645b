"""GPU tests added in round 4 (run with ``-m gpu``): non-finite inputs, the ``math.top_k`` wrapper's edge cases, limit > 64
through the single-process multi-GPU index, the default candidate pool of a re-rank index with few row slices, lazy matches."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


# ------------------------------------------------------------------------------------ non-finite inputs
# The reference has no guard: pq_bindings.pyx:30-47 sums whatever the tables hold, math.py:94-120 selects with numpy's order --
# NaN behind every number, +inf included.  The kernels must return the same rows (fixed tie-break: distance, then row id) and
# the same distances (NaN for NaN) for: a query with an inf coordinate (its tables are all inf in one sub-space: every row's
# sum is inf), an all-inf query, a NaN query, inner-product tables with +-inf / NaN mixed, code books with a few / with only
# huge code words (table entries overflow to inf).  Every scan kernel family: byte tables (M = 16 / 8 / 64, k <= 16), u16
# tables (M = 32; M = 16 with k = 32), the generic kernel (M = 12), both entry points (tables prebuilt / built by the call).
CASES = ['inf_coordinate', 'inf_query', 'nan_query', 'ip_inf_query', 'huge_codewords_some', 'huge_codewords_all']


def _nonfinite_inputs(case, M, dsub, N, B, Ks, seed):
    rs = np.random.RandomState(seed)
    D = M * dsub
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    lat = rs.randn(N, 6).astype(np.float32) @ rs.randn(6, D).astype(np.float32)
    x = (lat + 0.1 * rs.randn(N, D)).astype(np.float32)
    q = (rs.randn(B, 6).astype(np.float32) @ rs.randn(6, D).astype(np.float32)).astype(np.float32)
    kind = 1
    if case == 'inf_coordinate':
        q[3, 5] = np.inf
        q[9, D - 1] = -np.inf
    elif case == 'inf_query':
        q[3, :] = np.inf
    elif case == 'nan_query':
        q[4, 0] = np.nan
        q[11, :] = np.nan
    elif case == 'ip_inf_query':
        kind = 3  # float32(1/Ks) - <q, c>  (pq.py:316-322): +-inf and (0 * inf, inf - inf) NaN entries mixed
        q[3, 5] = np.inf
        q[9, 2] = -np.inf
        cb[0, 7, :] = 0.0
    elif case == 'huge_codewords_some':
        cb[2, 17, :] = 1e30   # its L2 entries are (1e30)^2 = inf for every query
        cb[5, 200, 1] = -3e25
    elif case == 'huge_codewords_all':
        cb *= 1e25
    return cb, x, q, kind


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('M,dsub,k', [(16, 8, 10), (16, 8, 32), (32, 4, 10), (8, 8, 10), (64, 4, 10), (12, 4, 10)])
def test_non_finite_inputs_equal_the_oracle(ops, oracle, case, M, dsub, k):
    import torch
    from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, scan_plan

    N, B, Ks = 30_000, 21, 256
    cb, x, q, kind = _nonfinite_inputs(case, M, dsub, N, B, Ks, seed=M * 100 + k)
    codes = oracle.encode_c(x, np.where(np.isfinite(cb), cb, 0).astype(np.float32) if case.startswith('huge') else cb)
    if case.startswith('huge'):
        codes[::7, 2] = 17      # rows that use the overflowing code words ...
        codes[::11, 5] = 200    # ... (their distances are +inf for every query)
    omet = {1: oracle.EUCLIDEAN, 3: oracle.INNER_PRODUCT}[kind]
    with np.errstate(all='ignore'):
        lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb) if kind == 1 else oracle.get_dist_mat_c(q, cb, omet)
        rd, ri = oracle.adc_search_c(lut, codes, k)
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    lut_d = ops.lut_build(q_d, cb_d, kind, LAYOUT_BMK).cpu().numpy()
    assert np.array_equal(lut_d, lut, equal_nan=True), 'tables differ'
    for layout in ((0, 1) if M in (8, 16, 32, 64) else (0,)):
        cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
        d, i = ops.pq_search_topk(kind, q_d, cb_d, cd, k, M, Ks, codes_layout=layout)   # tables built by the call
        torch.cuda.synchronize()
        assert np.array_equal(i.cpu().numpy(), ri), (case, layout, 'ids')
        assert np.array_equal(d.cpu().numpy(), rd, equal_nan=True), (case, layout, 'distances')
        plan = scan_plan(N, M, Ks, 1, B, k)
        lt = ops.lut_build(q_d, cb_d, kind, LAYOUT_TILED if plan.fast else LAYOUT_BMK, plan.qi)
        d, i = ops.adc_scan_topk(cd, lt, B, k, M, Ks, codes_layout=layout)               # tables handed over
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd, equal_nan=True), (case, layout, 'prebuilt')


@pytest.mark.parametrize('case', ['inf_query', 'nan_query', 'huge_codewords_some'])
def test_non_finite_inputs_through_the_index_plugin(ops, oracle, case):
    """The same through ``PQFlatGpuIndex.search_batch`` (numpy in, sqrt epilogue of EUCLIDEAN, hnsw/index.py:164-165) incl. limit > 64
    (the all-distances path) and the single-query ``search``: every valid row is a legitimate neighbour, whatever its distance."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    M, dsub, Ks, N, B = 16, 8, 256, 5_000, 13
    cb, x, q, _ = _nonfinite_inputs(case, M, dsub, N, B, Ks, seed=77)
    codec = PQCodec(dim=M * dsub, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN).set_codebooks(cb)
    idx = PQFlatGpuIndex(dim=M * dsub, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    codes = ops.codes_to_numpy(ops.pq_encode(ops.to_dev(x), codec.codebooks_dev))
    for k in (10, 100):
        with np.errstate(all='ignore'):
            rd, ri = oracle.index_search(q, cb, codes, oracle.EUCLIDEAN, k)
        d, i = idx.search_batch(q, limit=k)
        assert np.array_equal(i, ri), (case, k)
        assert np.array_equal(d, rd, equal_nan=True), (case, k)
    d1, i1 = idx.search(q[3], limit=10)
    assert np.array_equal(i1, ri[3][:10]) and len(d1) == 10


# ------------------------------------------------------------------------------------ math.top_k wrapper
def test_math_top_k_wrapper_edge_cases(ops, oracle):
    """annlite/math.py:94-120: ``descending``, ``k >= n`` (full argsort branch), k beyond the wave kernel's 64 (device sort) --
    values as the reference returns them, indices under the fixed tie-break (value, then index)."""
    import torch
    from annlite_amd import math as amath

    rs = np.random.RandomState(3)
    v = rs.randint(0, 40, size=(5, 300)).astype(np.float32)  # heavy ties
    v[2, 17] = -np.inf
    v[3, 5] = np.inf

    def want(vals, k, descending):
        w = -vals if descending else vals
        idx = np.stack([np.lexsort((np.arange(w.shape[1]), w[b]))[:k] for b in range(w.shape[0])])
        return np.take_along_axis(vals, idx, axis=1), idx

    for k in (1, 10, 64, 65, 100, 300, 450):
        for desc in (False, True):
            d, i = amath.top_k(v, k, descending=desc)
            wd, wi = want(v, min(k, v.shape[1]), desc)
            assert d.shape == wd.shape and np.array_equal(i, wi), (k, desc)
            assert np.array_equal(d, wd), (k, desc)
            dt, it = amath.top_k(torch.from_numpy(v).cuda(), k, descending=desc)  # torch in -> torch out
            assert isinstance(dt, torch.Tensor) and np.array_equal(it.cpu().numpy(), wi) and np.array_equal(dt.cpu().numpy(), wd)
    # the oracle's own top-k (pinned to the reference's values by the golden fixture) agrees on a row
    od, oi = oracle.top_k_c(v[1], 100)
    d, i = amath.top_k(v[1:2], 100)
    assert np.array_equal(d[0], od) and np.array_equal(i[0], oi)


# ------------------------------------------------------------------------------------ limit > 64 across shards, re-rank pool
def test_multi_gpu_index_limit_above_64(ops, oracle, tmp_path):
    """``AnnLite(devices=[0, 0]).search(limit=100)``: every shard answers through its batched large-k path and the lists are merged by
    (distance, id) on the device -- equal to the flat index (the merge kernel's wave lists end at k = 64)."""
    from annlite_amd import AnnLite

    rs = np.random.RandomState(21)
    D, M, N, B = 64, 16, 6_000, 9
    A = rs.randn(8, D).astype(np.float32)
    x = (rs.randn(N, 8).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    x[3000:3030] = x[5]  # ties across shard blocks
    q = np.concatenate([x[5:6], (rs.randn(B - 1, 8).astype(np.float32) @ A).astype(np.float32)])
    from annlite_amd.index import Document, DocumentArray

    flat = AnnLite(D, metric='euclidean', n_subvectors=M, data_path=tmp_path / 'flat')
    flat.train(x[:4096])
    multi = AnnLite(D, metric='euclidean', n_subvectors=M, data_path=tmp_path / 'multi', devices=[0, 0], shard_block=512)
    multi._pq_codec.set_codebooks(flat._pq_codec.codebooks)
    docs = lambda: DocumentArray([Document(id=str(i), embedding=x[i]) for i in range(N)])
    flat.index(docs())
    multi.index(docs())
    for k in (65, 100, 700):
        fd, fi = flat.search_numpy(q, limit=k)
        md, mi = multi.search_numpy(q, limit=k)
        for b in range(B):
            assert np.array_equal(fd[b], md[b]), (k, b)
            # equal distances may tie differently only where the sqrt merged two raw sums: ids agree as sets per distance value
            if not np.array_equal(fi[b], mi[b]):
                for val in np.unique(fd[b]):
                    sel = fd[b] == val
                    if val == fd[b][-1]:
                        continue  # (a tie cut by k: either member may stay)
                    assert set(fi[b][sel]) == set(mi[b][sel]), (k, b, val)


def test_rerank_default_pool_with_few_slices(ops):
    """A re-rank index over a SMALL table (the plan has one or two row slices) asked for k = 32: the default candidate pool must
    exceed k (16 keys x 1 slice would hand the exact stage fewer rows than it has to return) -- k real hits, recall >= 0.9."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(8)
    D, M, N, B, k = 64, 16, 9_000, 64, 32
    A = rs.randn(8, D).astype(np.float32)
    x = (rs.randn(N, 8).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 8).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 2
    codec.fit(x[:4096], iter=10)
    idx = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, rerank=True, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    d, i = idx.search_batch(q, limit=k)
    assert (i >= 0).all() and np.isfinite(d).all(), 'padding in a re-rank result although the table has the rows'
    exact = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    truth = np.argsort(exact, axis=1)[:, :k]
    recall = np.mean([len(set(i[b]) & set(truth[b])) / k for b in range(B)])
    assert recall >= 0.9, recall
    assert all(len(set(i[b])) == k for b in range(B))
    d10, i10 = idx.search_batch(q, limit=10)  # k <= 16 on a table with few slices: the pool is still wider than k
    recall10 = np.mean([len(set(i10[b]) & set(truth[b][:10])) / 10 for b in range(B)])
    assert recall10 >= 0.9, recall10


# ------------------------------------------------------------------------------------ lazy matches through the facade
def test_facade_lazy_matches_equal_the_eager_documents(ops, tmp_path):
    """``AnnLite.search`` attaches LAZY match lists: same ids / scores / metadata as documents built on the spot
    (container.py:226-233), built only when read; ``search_numpy`` returns the same ids as ints."""
    from annlite_amd import AnnLite
    from annlite_amd.docarray_compat import LazyMatches
    from annlite_amd.index import Document, DocumentArray

    rs = np.random.RandomState(9)
    N, D, B = 3000, 64, 50
    X = rs.rand(N, D).astype(np.float32)
    ann = AnnLite(D, data_path=tmp_path / 'idx', n_subvectors=8, metric='euclidean')
    ann._pq_codec.seed = 1
    ann.train(X)
    ann.index(DocumentArray([Document(id=str(1000 + i), embedding=X[i], tags={'n': i}) for i in range(N)]))
    ann.delete([str(1000 + 7)])
    query = DocumentArray([Document(embedding=X[i]) for i in range(B)])
    ann.search(query, limit=6)
    dists, ids = ann.search_numpy(X[:B], limit=6)
    d_raw, i_raw = ann._search_arrays(X[:B], None, 6)
    for b, qd in enumerate(query):
        assert isinstance(qd.matches, LazyMatches) and not qd.matches.materialised and len(qd.matches) == 6
        got = [(m.id, m.scores['euclidean'].value, m.tags.get('n')) for m in qd.matches]
        assert qd.matches.materialised
        assert [g[0] for g in got] == [str(1000 + int(o)) for o in i_raw[b]]
        assert [g[1] for g in got] == list(d_raw[b])
        assert [g[2] for g in got] == [int(o) for o in i_raw[b]]
        assert ids[b].tolist() == [1000 + int(o) for o in i_raw[b]] and np.array_equal(dists[b], d_raw[b])
    assert '1007' not in [m.id for qd in query for m in qd.matches]
    ann.search(query, limit=6, include_metadata=False)
    assert query[0].matches[0].tags == {}


# ------------------------------------------------------------------------------------ early merger of a tile's row slices
@pytest.mark.parametrize('patience', [None, '0', '3', '300', '2000'])
@pytest.mark.parametrize('M,dsub,k', [(16, 8, 10), (16, 8, 16), (64, 4, 10), (8, 8, 3)])
def test_early_merger_and_its_give_up_path(ops, oracle, monkeypatch, patience, M, dsub, k):
    """The first workgroup of a query tile to finish folds the other slices' lists as they arrive (scan_q8.hip: q8_early_merge);
    with no patience (ANNLITE_EARLY_MERGE_PATIENCE=0 / 3 ticks) it leaves at once and the last slice to arrive merges the tile
    from global memory -- both roads, and the switch that turns the early merger off, return the oracle's bits; heavy ties across
    slices, deleted rows, a ragged batch.  Patience of 3 / 20 us (300 / 2000 ticks of the 100 MHz clock) runs out WHILE slices keep
    arriving: the leave decision is taken by one thread for the whole workgroup (round 5; per-wave clock readings could
    split the workgroup at the threshold)."""
    import torch
    from annlite_amd._capi import LUT_L2

    rs = np.random.RandomState(M + k)
    N, B, Ks, D = 400_000, 70, 256, M * dsub
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    base = rs.randint(0, Ks, size=(500, M)).astype(np.uint8)
    codes = base[rs.randint(0, 500, N)]  # 500 distinct code rows: every distance ties ~800 times, across all slices
    valid = np.ones(((N + 31) // 32 + 2) * 32, dtype=bool)
    valid[N:] = False
    valid[rs.choice(N, 50_000, replace=False)] = False
    q = rs.randn(B, D).astype(np.float32)
    lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    live = np.nonzero(valid[:N])[0]
    rd, ri = oracle.adc_search_c(lut, codes[live], k, threads=oracle.max_threads())
    ri = live[ri]
    bits = ops.to_dev(np.packbits(valid.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))  # bit n of word n / 32
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    envs = [{}] if patience is None else [{'ANNLITE_EARLY_MERGE_PATIENCE': patience}]
    envs.append({'ANNLITE_NO_EARLY_MERGE': '1'})
    for env in envs:
        for key in ('ANNLITE_EARLY_MERGE_PATIENCE', 'ANNLITE_NO_EARLY_MERGE'):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        for layout in (0, 1):
            cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
            for rep in range(3):
                d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, valid_bits=bits)
                assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd), (env, layout, rep)


# ------------------------------------------------------------------------------------ the search in two halves with a seed exchange
def test_split_search_with_seed_union_equals_the_plain_search(ops, oracle):
    """annlite_pq_search_split (PREPARE -> annlite_pq_search_seed_union -> SCAN): a rank that seeds from few rows and tightens
    its bound with the union of "peer" key sets (here: the seeds of other row ranges of the same table -- valid bounds for it)
    returns the plain search's bits; NOT_APPLICABLE before the table's state has settled and for shapes without the fused
    preparation launch (nothing launched: the plain call follows)."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2, PHASE_PREPARE, PHASE_SCAN, SEED_KEYS

    rs = np.random.RandomState(5)
    N, M, dsub, Ks, B, k = 600_000, 16, 8, 256, 70, 10
    D = M * dsub
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    A = rs.randn(8, D).astype(np.float32)
    cb_d = ops.to_dev(cb)
    codes_d = torch.empty((N, M), dtype=torch.uint8, device='cuda')
    for c0 in range(0, N, 100_000):
        x = (rs.randn(100_000, 8).astype(np.float32) @ A + 0.05 * rs.randn(100_000, D).astype(np.float32)).astype(np.float32)
        codes_d[c0:c0 + 100_000] = ops.pq_encode(ops.to_dev(x), cb_d)
    q = (rs.randn(B, 8).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    q_d = ops.to_dev(q)
    codes = codes_d.cpu().numpy()
    lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    rd, ri = oracle.adc_search_c(lut, codes, k, threads=oracle.max_threads())
    for layout in (1, 0):
        cd = ops.codes_skew(codes_d) if layout else codes_d
        ws, state = ops.ScanWorkspace(), _capi.ScanState()
        call = lambda phase, **kw: ops.pq_search_split(phase, LUT_L2, q_d, cb_d, cd, k, M, Ks, state, ws, codes_layout=layout, **kw)
        assert call(PHASE_PREPARE, seed_rows=4096) is None  # the state has not seen a launch yet: not applicable
        keys = None
        for _ in range(12):  # a few plain calls settle the kernel choice (read without synchronising: give it launches)
            d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, workspace=ws, state=state)
            torch.cuda.synchronize()
            assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
            keys = call(PHASE_PREPARE, seed_rows=4096)
            if keys is not None:
                break
        assert keys is not None and keys.shape == (B, SEED_KEYS), 'the state never settled on the byte-table kernel'
        torch.cuda.synchronize()
        kh = keys.cpu().numpy().view(np.uint64)
        assert (np.diff(kh[:, :k].astype(np.float64), axis=1) >= 0).all() and (kh[:, k:] == np.uint64(2 ** 64 - 1)).all()
        # "peers": the seeds of three other row ranges (each a PREPARE on a view of the table), then this rank's own
        peers = []
        for off in (64_000, 128_000, 256_000):  # (views of more than half the table: the state's size watch stays quiet)
            sub = cd[off:]
            pk = ops.pq_search_split(PHASE_PREPARE, LUT_L2, q_d, cb_d, sub, k, M, Ks, state, ws, codes_layout=layout, seed_rows=4096)
            assert pk is not None
            peers.append(pk.clone())
        own = call(PHASE_PREPARE, seed_rows=4096)
        allk = torch.stack([own] + peers).contiguous()
        ops.pq_search_seed_union(allk, cd, B, k, M, Ks, ws)
        packed = call(PHASE_SCAN)
        torch.cuda.synchronize()
        p = packed.cpu().numpy()
        assert np.array_equal(p[..., 0], ri), layout
        assert np.array_equal((p[..., 1] & 0xFFFFFFFF).astype(np.uint32).view(np.float32), rd), layout
        # the union bound is at least as tight as the own one: k-th smallest of 4 k keys <= own k-th
        uk = np.sort(allk.cpu().numpy().view(np.uint64).transpose(1, 0, 2).reshape(B, -1), axis=1)[:, k - 1]
        assert (uk <= kh[:, k - 1]).all()
    # a shape without the fused preparation launch: not applicable, whatever the state
    cb32 = ops.to_dev(rs.randn(32, Ks, 4).astype(np.float32))
    c32 = torch.randint(0, Ks, (50_000, 32), dtype=torch.uint8, device='cuda')
    q32 = ops.to_dev(rs.randn(5, 128).astype(np.float32))
    assert ops.pq_search_split(PHASE_PREPARE, LUT_L2, q32, cb32, c32, k, 32, Ks, _capi.ScanState(), ops.ScanWorkspace()) is None


def test_bench_seed_exchange_under_torchrun_one_rank(tmp_path):
    """bench.py as one rank under torchrun with the exchange forced and 8 emulated seed peers: the seed collective + union + scan
    pipeline on two streams, results = the CPU oracle for every query of the batch."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, ANNLITE_FORCE_GATHER='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29537', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--rows', '400000', '--steps', '12',
           '--warmup', '2', '--recall-queries', '0', '--cpu-queries', '2', '--no-rerank', '--legs', 'none', '--seed-exchange',
           '--emulate-seed-peers', '8']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert rec['config']['seed_exchange'] is True and rec['config']['seed_peers_emulated'] == 8 and rec['config']['seed_rows'] == 4096
    assert rec['per_rank'][0]['exchange_ms'] is not None and rec['exchange_ms'] > 0


# ------------------------------------------------------------------------------------ round 5: reproducible training
def test_deterministic_fit_repeats_bit_for_bit(ops, oracle):
    """``PQCodec.deterministic = True`` (+ a seed): the Lloyd steps accumulate in a fixed order -- two fits give the SAME codebooks,
    bit for bit (with the float atomics of the default path they agree to the last few bits only), at the default path's quality.
    bench.py trains this way so that its ``result_sha256`` can be compared across runs (an N = 8 line against the N = 1 line)."""
    from annlite_amd import Metric, PQCodec

    rs = np.random.RandomState(0)
    N, D, M = 20480, 128, 16
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    books, mse = [], []
    for det in (True, True, False):
        c = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
        c.seed = 7
        c.deterministic = det
        c.fit(x, iter=12)
        books.append(c.codebooks.copy())
        codes = oracle.encode_c(x, c.codebooks)
        rec = np.concatenate([c.codebooks[m][codes[:, m]] for m in range(M)], axis=1)
        mse.append(float(((x - rec) ** 2).mean()))
    assert np.array_equal(books[0].view(np.uint32), books[1].view(np.uint32))
    assert abs(mse[0] - mse[2]) <= 0.02 * mse[2], mse


def test_measured_shader_clock_and_kernel_revisions(ops, oracle):
    """bench.py's evidence hooks: the byte-table kernel leaves its own cycle / wall-clock stamps when profiling is on -- a plausible
    MI355X shader clock --, another kernel's launch says "none"; ``annlite_kernel_rev`` knows the kernels profiles/traffic.json
    records and every recorded entry names a revision."""
    import json
    import os

    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2
    from conftest import ROOT

    rs = np.random.RandomState(3)
    N, M, dsub, Ks, B, k = 300_000, 16, 8, 256, 64, 10
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    base = rs.randint(0, Ks, size=(2000, M)).astype(np.uint8)
    codes = base[rs.randint(0, 2000, N)]
    q = rs.randn(B, M * dsub).astype(np.float32)
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    _capi.profile_enable(True)
    try:
        os.environ['ANNLITE_SCAN_VARIANT'] = '50'
        _capi.knobs_reload()  # (the library parses its switches at load: tell it)
        ops.pq_search_topk(LUT_L2, q_d, cb_d, codes_d, k, M, Ks)
        torch.cuda.synchronize()
        ms, mhz = _capi.profile_last_scan_ms(), _capi.profile_last_scan_clock_mhz()
        assert ms > 0 and mhz is not None and 800.0 < mhz < 2600.0, (ms, mhz)
        os.environ['ANNLITE_SCAN_VARIANT'] = '31'  # the u16-table kernel: no stamps of its own
        _capi.knobs_reload()
        ops.pq_search_topk(LUT_L2, q_d, cb_d, codes_d, k, M, Ks)
        torch.cuda.synchronize()
        assert _capi.profile_last_scan_ms() > 0 and _capi.profile_last_scan_clock_mhz() is None
    finally:
        os.environ.pop('ANNLITE_SCAN_VARIANT', None)
        _capi.knobs_reload()
        _capi.profile_enable(False)
    for name in ('adc_scan_q8_kernel', 'adc_scan_qfilter_kernel', 'adc_scan_qfilter64_kernel', 'graph_beam_search_kernel'):
        assert _capi.kernel_rev(name) >= 1
    assert _capi.kernel_rev('no_such_kernel') == 0
    table = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    for key, ent in table.items():
        if isinstance(ent, dict):
            assert isinstance(ent.get('kernel_rev'), int) and ent['hbm_bytes_per_launch'] > 0, key
