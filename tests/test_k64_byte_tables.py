"""16 < k <= 64 on the byte-table kernel (round 5): ``adc_scan_q8_kernel<16, 16, SKEWED, 2, 1, true, 64>`` (launch id 1664,
scan_q8.hip) -- the M = 16 kernel with 64-key lists (one insertion per wave operation), the keys at four list positions
published to the sibling slices (``q8_weighted_bound``), the slices merged by ``merge_partial_kernel``.  The reference's own PQ
test asks for ``topk = 50`` (tests/test_pq_index.py:83-135).  The PUBLIC plan of these k stays the u16 plan; the library's search
picks the byte tables itself (guarded, like k <= 16); ``ANNLITE_SCAN_VARIANT=50`` pins the kernel under test.
Bit-exact against the oracle and against the u16-table kernel: random / tied / deleted / short tables, both layouts, structured
data at 2M rows, non-finite tables, forced epochs, the guarded first launch on a table the byte filter leaks on."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]

M = 16


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


@pytest.fixture(autouse=True)
def byte_tables(monkeypatch):
    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '50')


def _bits(valid):
    bits = np.zeros(((len(valid) + 31) // 32 + 2) * 32, dtype=bool)
    bits[:len(valid)] = valid
    return np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1)


def _scan(ops, codes, lut, k, layout, valid=None, row_base=0):
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import scan_plan

    B, Ks = lut.shape[0], lut.shape[2]
    plan = scan_plan(codes.shape[0], M, Ks, 1, B, k)
    assert plan.fast and plan.qt == 32, (plan.fast, plan.qt)  # (variant 50: the byte-table plan, 32 queries per tile)
    lut_d = ops.lut_retile(ops.to_dev(lut), plan.qi)
    codes_d = ops.to_dev(codes)
    if layout == 1:
        codes_d = ops.codes_skew(codes_d)
    vb = ops.to_dev(_bits(valid)) if valid is not None else None
    os.environ['ANNLITE_DEBUG_COUNTERS'] = '2'
    _capi.knobs_reload()  # (the library parses its switches at load: tell it)
    try:
        d, i = ops.adc_scan_topk(codes_d, lut_d, B, k, M, Ks, valid_bits=vb, row_base=row_base, codes_layout=layout)
        torch.cuda.synchronize()
        items = _capi.debug_timeline()['items'] if codes.shape[0] > 0 else 1
    finally:
        del os.environ['ANNLITE_DEBUG_COUNTERS']
        _capi.knobs_reload()
    assert items > 0, 'the byte-table kernel did not run'
    return d.cpu().numpy(), i.cpu().numpy()


SHAPES = [  # Ks, N, B, k
    (256, 70_000, 20, 50), (256, 130_000, 33, 64), (256, 66_000, 5, 17), (100, 80_000, 48, 33), (256, 300_000, 9, 50),
    (256, 63, 9, 64), (256, 1, 3, 20), (256, 5000, 37, 50), (17, 9000, 64, 40), (256, 4097, 16, 63),
]


@pytest.mark.parametrize('Ks,N,B,k', SHAPES)
@pytest.mark.parametrize('layout', [0, 1])
def test_random_shapes_equal_the_oracle(ops, oracle, Ks, N, B, k, layout):
    rs = np.random.RandomState(Ks * 31 + N + k)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut[B // 2] -= 0.5  # negative entries (inner-product style tables)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    d, i = _scan(ops, codes, lut, k, layout, row_base=1000)
    rd, ri = oracle.adc_search_c(lut, codes, k, id_base=1000)
    assert np.array_equal(d, rd)
    assert np.array_equal(i, ri)


def test_ties_delete_marks_and_short_tables(ops, oracle):
    rs = np.random.RandomState(12)
    Ks, N, B, k = 256, 90_000, 21, 50
    base = rs.randint(0, Ks, size=(64, M)).astype(np.uint8)
    codes = base[rs.randint(0, 64, size=N)]  # every row has ~1400 exact duplicates: the 50 best all tie
    lut = rs.rand(B, M, Ks).astype(np.float32)
    for layout in (0, 1):
        d, i = _scan(ops, codes, lut, k, layout)
        rd, ri = oracle.adc_search_c(lut, codes, k)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
    valid = rs.rand(N) < 0.3
    d, i = _scan(ops, codes, lut, k, 1, valid=valid)
    idx = np.where(valid)[0]
    rd, ri = oracle.adc_search_c(lut, codes[idx], k)
    assert np.array_equal(d, rd) and np.array_equal(i, idx[ri])
    valid2 = np.zeros(N, bool)
    valid2[[5, 77, 80_000]] = True
    d, i = _scan(ops, codes, lut, k, 0, valid=valid2)
    assert (i[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all()
    assert (np.sort(i[:, :3], axis=1) == np.array([5, 77, 80_000])).all()


def _structured(ops, N, B, dsub, seed):
    import torch
    from annlite_amd import Metric, PQCodec

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    D = M * dsub
    A = torch.randn((16, D), generator=g, device=dev)

    def gen(n):
        return (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()

    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.fit(gen(20480), iter=10)
    cb = codec.codebooks_dev
    codes = torch.empty((N, M), dtype=torch.uint8, device=dev)
    for c0 in range(0, N, 500_000):
        n = min(500_000, N - c0)
        codes[c0:c0 + n] = ops.pq_encode(gen(n), cb)
    return cb, codes, gen(B)


@pytest.mark.parametrize('N,B,k', [(300_000, 100, 50), (2_000_000, 256, 50), (1_000_000, 1000, 64), (2_500_000, 70, 17)])
def test_structured_data_equals_the_u16_table_kernel_and_the_oracle(ops, oracle, monkeypatch, N, B, k):
    """``annlite_pq_search_topk`` (tables built by the call, seed bound for THIS k, shared bounds incl. the weighted bound of the
    sibling slices, merge of the row slices): the byte-table plan returns the bits of the u16-table plan on the same inputs --
    all queries --, and of the oracle for a sample; deleted rows; several calls on one workspace."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    cb, codes, q = _structured(ops, N, B, 8, seed=N % 1000 + k)
    rs = np.random.RandomState(k)
    valid = np.ones(N, bool)
    valid[rs.choice(N, N // 20, replace=False)] = False
    vb = ops.to_dev(_bits(valid))
    out = {}
    for layout in (0, 1):
        cd = ops.codes_skew(codes) if layout == 1 else codes
        for variant in ('50', '31'):
            monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
            for rep in range(2):
                monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '2')
                d, i = ops.pq_search_topk(LUT_L2, q, cb, cd, k, M, 256, codes_layout=layout, valid_bits=vb)
                torch.cuda.synchronize()
                items = _capi.debug_timeline()['items']
                monkeypatch.delenv('ANNLITE_DEBUG_COUNTERS')
                assert (items > 0) == (variant == '50'), (variant, items)
                got = (d.cpu().numpy(), i.cpu().numpy())
                if rep:
                    assert np.array_equal(got[0], out[(layout, variant)][0]) and np.array_equal(got[1], out[(layout, variant)][1])
                out[(layout, variant)] = got
        assert np.array_equal(out[(layout, '50')][0], out[(layout, '31')][0]), layout
        assert np.array_equal(out[(layout, '50')][1], out[(layout, '31')][1]), layout
    assert np.array_equal(out[(0, '50')][1], out[(1, '50')][1])
    nq = min(B, 8)
    cb_h, codes_h, q_h = cb.cpu().numpy(), ops.codes_to_numpy(codes), q[:nq].cpu().numpy()
    lut = oracle.batch_precompute_adc_table_c(q_h, 8, 256, cb_h)
    live = np.nonzero(valid)[0]
    rd, ri = oracle.adc_search_c(lut, codes_h[live], k, threads=oracle.max_threads())
    assert np.array_equal(out[(1, '50')][0][:nq], rd) and np.array_equal(out[(1, '50')][1][:nq], live[ri])


@pytest.mark.parametrize('case', ['inf_coordinate', 'inf_query', 'nan_query', 'ip_inf_query', 'huge_codewords_some', 'huge_codewords_all'])
def test_non_finite_tables(ops, oracle, case):
    """the reference has no guard (pq_bindings.pyx:30-47, math.py:94-120: NaN sorts last): same rows, same distances at k = 50"""
    import torch
    from test_round4_gpu import _nonfinite_inputs

    N, B, Ks, dsub, k = 70_000, 21, 256, 8, 50
    cb, x, q, kind = _nonfinite_inputs(case, M, dsub, N, B, Ks, seed=M * 100 + k)
    codes = oracle.encode_c(x, np.where(np.isfinite(cb), cb, 0).astype(np.float32) if case.startswith('huge') else cb)
    if case.startswith('huge'):
        codes[::7, 2] = 17
        codes[::11, 5] = 200
    omet = {1: oracle.EUCLIDEAN, 3: oracle.INNER_PRODUCT}[kind]
    with np.errstate(all='ignore'):
        lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb) if kind == 1 else oracle.get_dist_mat_c(q, cb, omet)
        rd, ri = oracle.adc_search_c(lut, codes, k)
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    for layout in (0, 1):
        cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
        d, i = ops.pq_search_topk(kind, q_d, cb_d, cd, k, M, Ks, codes_layout=layout)
        torch.cuda.synchronize()
        assert np.array_equal(i.cpu().numpy(), ri), (case, layout, 'ids')
        assert np.array_equal(d.cpu().numpy(), rd, equal_nan=True), (case, layout, 'distances')


@pytest.mark.parametrize('tune', [('1,2,192,0', '64', '7'), ('100000,2,384,3', '96', '4')])
def test_epochs_and_rebuilds(ops, oracle, tune, monkeypatch):
    """an epoch end every other step with the tables rebuilt as soon as a bound moves, and a schedule without any epoch"""
    import torch
    from annlite_amd._capi import LUT_L2

    monkeypatch.setenv('ANNLITE_Q8_TUNE', tune[0])
    monkeypatch.setenv('ANNLITE_Q8_TARGET', tune[1])
    monkeypatch.setenv('ANNLITE_Q8_REBUILD', tune[2])
    N, B, k = 400_000, 48, 50
    cb, codes, q = _structured(ops, N, B, 8, seed=21)
    d, i = ops.pq_search_topk(LUT_L2, q, cb, ops.codes_skew(codes), k, M, 256, codes_layout=1)
    torch.cuda.synchronize()
    lut = oracle.batch_precompute_adc_table_c(q.cpu().numpy(), 8, 256, cb.cpu().numpy())
    rd, ri = oracle.adc_search_c(lut, ops.codes_to_numpy(codes), k, threads=oracle.max_threads())
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)


def test_the_library_picks_it_guards_it_and_falls_back(ops, oracle, monkeypatch):
    """WITHOUT the variant switch, through the index plug-in: the library runs the byte tables for k = 50 guarded (a u16 pass gated
    behind it), settles on them for a table with structure -- and on a table the byte filter leaks on (independent uniform
    codes) the guarded launch gives up and the gated pass answers.  The oracle's bits either way, every batch."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    rs = np.random.RandomState(6)
    N, D, B, k = 1_200_000, 128, 100, 50
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    idx = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    codes = ops.codes_to_numpy(ops.pq_encode(ops.to_dev(x), codec.codebooks_dev))
    rd, ri = oracle.index_search(q, codec.codebooks, codes, oracle.EUCLIDEAN, k, threads=oracle.max_threads())
    for _ in range(4):
        d, i = idx.search_batch(q, limit=k)
        assert np.array_equal(i, ri) and np.array_equal(d, rd)
    assert idx.scan_kernel in ('byte tables', 'u16 tables')  # (settled; which one is the library's measurement on this box)
    # a table without structure: every code drawn independently
    N2 = 300_000
    idx2 = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N2)
    idx2.add_with_ids(x[:N2], np.arange(N2))
    rnd = rs.randint(0, 256, size=(N2, M)).astype(np.uint8)
    idx2._codes[:N2] = ops.codes_skew(ops.to_dev(rnd)) if idx2._layout() == 1 else ops.to_dev(rnd)
    rd2, ri2 = oracle.index_search(q, codec.codebooks, rnd, oracle.EUCLIDEAN, k, threads=oracle.max_threads())
    for _ in range(3):
        d, i = idx2.search_batch(q, limit=k)
        assert np.array_equal(i, ri2) and np.array_equal(d, rd2)


def test_rerank_pool_from_the_global_top_50(ops, oracle, monkeypatch):
    """``PQFlatGpuIndex(rerank=True, rerank_pool='global')``: the exact re-rank takes the GLOBAL ADC top-50 (one shared-bound search
    on the 64-key lists) instead of the slices' own top-16 lists: every returned id is one of the oracle's ADC top-50, the order
    is the exact distances', and recall against brute force is that of a 50-row pool."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    rs = np.random.RandomState(8)
    N, D, B, k = 600_000, 128, 64, 10
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    idx = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True, rerank_pool='global')
    idx.add_with_ids(x, np.arange(N))
    codes = ops.codes_to_numpy(ops.pq_encode(ops.to_dev(x), codec.codebooks_dev))
    _, pool = oracle.index_search(q, codec.codebooks, codes, oracle.EUCLIDEAN, 50, threads=oracle.max_threads())
    d, i = idx.search_batch(q, limit=k)
    exact = np.sqrt(((x[pool].astype(np.float64) - q[:, None, :].astype(np.float64)) ** 2).sum(-1))  # [B, 50]
    best = np.full((B, k), np.inf)
    best_i = np.full((B, k), -1, np.int64)
    for c0 in range(0, N, 100_000):
        dd = ((x[c0:c0 + 100_000, None, :] - q[None, :, :]) ** 2).sum(-1).T  # [B, chunk]
        md = np.concatenate([best, dd], 1)
        mi = np.concatenate([best_i, np.arange(c0, c0 + dd.shape[1])[None, :].repeat(B, 0)], 1)
        o = np.argsort(md, axis=1)[:, :k]
        best, best_i = np.take_along_axis(md, o, 1), np.take_along_axis(mi, o, 1)
    for b in range(B):
        assert set(i[b]) <= set(pool[b]), b
        order = np.argsort(exact[b], kind='stable')[:k]
        # the k best of the pool by exact distance (float64 here, float32 on the GPU: compare as sets unless a near-tie at the cut)
        want = set(pool[b][order])
        if set(i[b]) != want:
            gap = np.sort(exact[b])[k] - np.sort(exact[b])[k - 1]
            assert gap < 1e-4 * np.sort(exact[b])[k - 1], (b, gap)
        assert (np.diff(d[b]) >= 0).all()
    rec_global = np.mean([len(set(i[b]) & set(best_i[b])) / k for b in range(B)])
    idx.rerank_pool = 'slices'
    _, i2 = idx.search_batch(q, limit=k)
    rec_slices = np.mean([len(set(i2[b]) & set(best_i[b])) / k for b in range(B)])
    # (this small table plans 128 row slices for its two query tiles: the slice pool is 2048 rows and finds everything -- at the
    # bench's 10M x 1024 it is 8 x 16 = 128 rows, see the `rerank.global_pool` leg)
    assert rec_global >= 0.85 and rec_slices >= rec_global - 0.02, (rec_global, rec_slices)
