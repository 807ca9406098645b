"""M = 32 on the byte-table kernel: ``adc_scan_q8_kernel<32, 16, SKEWED, 1, 1>`` (launch id 3250, scan_q8.hip) -- 16 queries per
workgroup, entries clipped at 7, two half tables of 16 sub-spaces, one ``v_perm_b32`` per look-up address.  The reference's own
look-up-table test runs this sub-space count (tests/test_pq_bind.py:36-59).  The library's default for M = 32, k <= 16 (``annlite_scan_plan_query`` says 16 queries per tile); the tests here pin it with
``ANNLITE_SCAN_VARIANT=50`` (no guarded first launch: the kernel that runs is the one under test) except the index plug-in test.
Bit-exact against the oracle, against the u16-table kernel on the same inputs, for both code layouts, ragged batches, Ks < 256,
delete marks, exact ties, non-finite tables; and the debug counters prove which kernel ran."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]

M = 32


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


def _scan(ops, codes, lut, k, layout, valid=None, row_base=0, want_kernel=True):
    """``annlite_adc_scan_topk`` on tables handed over; asserts that the byte-table kernel did the work (its work-item counter)."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import scan_plan

    B, Ks = lut.shape[0], lut.shape[2]
    plan = scan_plan(codes.shape[0], M, Ks, 1, B, k)
    assert plan.fast and plan.qt == 16, (plan.fast, plan.qt)  # (u16 tables: 8 queries per tile)
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
    if want_kernel:
        assert items > 0, 'the byte-table kernel did not run'
    return d.cpu().numpy(), i.cpu().numpy()


SHAPES = [  # Ks, N, B, k
    (256, 2500, 6, 10), (256, 30000, 17, 16), (200, 999, 5, 3), (100, 70000, 33, 1), (256, 64, 1, 1), (256, 63, 9, 16),
    (256, 1, 3, 5), (256, 130000, 20, 10), (256, 4097, 16, 7), (17, 5000, 48, 10),
]


@pytest.mark.parametrize('Ks,N,B,k', SHAPES)
@pytest.mark.parametrize('layout', [0, 1])
def test_random_shapes_equal_the_oracle(ops, oracle, Ks, N, B, k, layout):
    rs = np.random.RandomState(Ks * 31 + N)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut[B // 2] -= 0.5  # negative entries (inner-product style tables)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    d, i = _scan(ops, codes, lut, k, layout, row_base=1000)
    rd, ri = oracle.adc_search_c(lut, codes, k, id_base=1000)
    assert np.array_equal(d, rd)
    assert np.array_equal(i, ri)


def test_ties_delete_marks_and_short_tables(ops, oracle):
    """duplicate rows => exact distance ties (ids ascending), rows masked by the validity bitmap never returned, fewer valid rows
    than k padded with (+inf, -1)"""
    rs = np.random.RandomState(11)
    Ks, N, B, k = 256, 40_000, 21, 16
    base = rs.randint(0, Ks, size=(64, M)).astype(np.uint8)
    codes = base[rs.randint(0, 64, size=N)]  # every row has ~600 exact duplicates
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
    valid2[[5, 77, 39_000]] = True
    d, i = _scan(ops, codes, lut, k, 0, valid=valid2)
    assert (i[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all()
    assert (np.sort(i[:, :3], axis=1) == np.array([5, 77, 39_000])).all()


def _structured(ops, N, B, dsub, seed):
    """the bench's kind of data: rank-16 latent vectors + noise through a trained codec (what the byte filter is made for)"""
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


@pytest.mark.parametrize('N,B,k', [(300_000, 100, 10), (2_000_000, 256, 16), (1_000_000, 1000, 1)])
def test_structured_data_equals_the_u16_table_kernel_and_the_oracle(ops, oracle, monkeypatch, N, B, k):
    """``annlite_pq_search_topk`` (tables built by the call, seed bound, shared bounds, in-kernel merge of the row slices): the
    byte-table plan returns the bits of the u16-table plan on the same inputs -- all queries --, and of the oracle for a sample."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    cb, codes, q = _structured(ops, N, B, 4, seed=N % 1000 + k)
    out = {}
    for layout in (0, 1):
        cd = ops.codes_skew(codes) if layout == 1 else codes
        for variant in ('50', '31'):
            monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
            monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '2')
            d, i = ops.pq_search_topk(LUT_L2, q, cb, cd, k, M, 256, codes_layout=layout)
            torch.cuda.synchronize()
            items = _capi.debug_timeline()['items']
            monkeypatch.delenv('ANNLITE_DEBUG_COUNTERS')
            assert (items > 0) == (variant == '50'), (variant, items)
            out[(layout, variant)] = (d.cpu().numpy(), i.cpu().numpy())
        assert np.array_equal(out[(layout, '50')][0], out[(layout, '31')][0]), layout
        assert np.array_equal(out[(layout, '50')][1], out[(layout, '31')][1]), layout
    assert np.array_equal(out[(0, '50')][1], out[(1, '50')][1])
    nq = min(B, 8)
    cb_h, codes_h, q_h = cb.cpu().numpy(), ops.codes_to_numpy(codes), q[:nq].cpu().numpy()
    lut = oracle.batch_precompute_adc_table_c(q_h, 4, 256, cb_h)
    rd, ri = oracle.adc_search_c(lut, codes_h, k)
    assert np.array_equal(out[(1, '50')][0][:nq], rd) and np.array_equal(out[(1, '50')][1][:nq], ri)


@pytest.mark.parametrize('case', ['inf_coordinate', 'inf_query', 'nan_query', 'ip_inf_query', 'huge_codewords_some', 'huge_codewords_all'])
def test_non_finite_tables(ops, oracle, case):
    """the reference has no guard (pq_bindings.pyx:30-47, math.py:94-120: NaN sorts last): same rows, same distances"""
    import torch
    from test_round4_gpu import _nonfinite_inputs

    N, B, Ks, dsub, k = 30_000, 21, 256, 4, 10
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


def test_index_plugin_settles_on_a_kernel_for_m32(ops, oracle, monkeypatch):
    """Through ``PQFlatGpuIndex`` WITHOUT the variant switch: whatever the library's default for M = 32 is, the results are the
    oracle's -- several batches, so that a guarded first launch and the settled choice are both exercised."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    rs = np.random.RandomState(5)
    N, D, B, k = 60_000, 128, 40, 10
    x = (rs.randn(N, 8) @ rs.randn(8, D) + 0.1 * rs.randn(N, D)).astype(np.float32)
    q = (rs.randn(B, 8) @ rs.randn(8, D)).astype(np.float32)
    cb = np.stack([x[rs.choice(N, 256, replace=False), m * (D // M):(m + 1) * (D // M)] for m in range(M)]).astype(np.float32)  # rows as code words
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN).set_codebooks(cb)
    idx = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    codes = ops.codes_to_numpy(ops.pq_encode(ops.to_dev(x), codec.codebooks_dev))
    rd, ri = oracle.index_search(q, cb, codes, oracle.EUCLIDEAN, k)
    for _ in range(4):
        d, i = idx.search_batch(q, limit=k)
        assert np.array_equal(i, ri) and np.array_equal(d, rd)


@pytest.mark.parametrize('layout', [0, 1])
def test_candidate_lists_with_slice_bounds(ops, oracle, layout):
    """the kernel as the candidate generator of the re-rank stage (k <= 16 per row slice, nothing shared between the slices,
    every (query, slice) seeded from the slice's own first rows): each list is the exact top-k of its slice, bit for bit"""
    import torch
    from annlite_amd import Metric, PQCodec
    from annlite_amd._capi import scan_plan

    rs = np.random.RandomState(34)
    N, D, B, k = 300_000, 128, 70, 16
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    codes = oracle.encode_c(x, codec.codebooks)
    lut = oracle.get_dist_mat_c(q, codec.codebooks, oracle.EUCLIDEAN)
    plan = scan_plan(N, M, 256, 1, B, k)
    assert plan.qt == 16 and plan.n_slices >= 4
    codes_d = ops.to_dev(codes)
    if layout == 1:
        codes_d = ops.codes_skew(codes_d)
    lut_t = ops.lut_retile(ops.to_dev(lut), plan.qi)
    valid = np.ones(N, dtype=bool)
    valid[rs.randint(0, N, size=N // 50)] = False
    cd, ci = ops.adc_scan_candidates(codes_d, lut_t, B, k, M, 256, valid_bits=ops.to_dev(_bits(valid)), codes_layout=layout)
    torch.cuda.synchronize()
    cd, ci = cd.cpu().numpy(), ci.cpu().numpy()
    ns = plan.n_slices
    assert ci.shape == (B, ns * k)
    rows = ((N + ns - 1) // ns + 63) // 64 * 64
    for sl in range(ns):
        a, b = sl * rows, min(N, (sl + 1) * rows)
        keep = np.nonzero(valid[a:b])[0]
        rd, ri = oracle.adc_search_c(lut, codes[a:b][keep], k)
        assert np.array_equal(cd[:, sl * k:(sl + 1) * k], rd), sl
        assert np.array_equal(ci[:, sl * k:(sl + 1) * k], keep[ri] + a), sl


@pytest.mark.parametrize('tune', [('1,2,192,0', '64', '7'), ('100000,2,384,3', '96', '4')])
def test_epochs_and_rebuilds(ops, oracle, tune, monkeypatch):
    """an epoch end every other step with the tables rebuilt as soon as a bound moves (the default schedule is sparse: small
    tables never reach its first epoch end), and a schedule without any epoch: the oracle's bits either way"""
    from annlite_amd import Metric, PQCodec, _capi

    rs = np.random.RandomState(21)
    N, D, B, k = 200_000, 128, 48, 10
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    codes = oracle.encode_c(x, codec.codebooks)
    lut = oracle.get_dist_mat_c(q, codec.codebooks, oracle.EUCLIDEAN)
    rd, ri = oracle.adc_search_c(lut, codes, k)
    monkeypatch.setenv('ANNLITE_Q8_TUNE', tune[0])
    monkeypatch.setenv('ANNLITE_Q8_TARGET', tune[1])
    monkeypatch.setenv('ANNLITE_Q8_REBUILD', tune[2])
    monkeypatch.setenv('ANNLITE_SEED_ROWS', '4096')  # (a loose first bound: the tables have something to follow)
    for layout in (0, 1):
        d, i = _scan(ops, codes, lut, k, layout)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
        if tune[0].startswith('1,2'):
            assert _capi.debug_counters()[5] > 0  # tables were rebuilt


def test_give_up_path_and_kernel_choice_on_uniform_codes(ops, oracle, monkeypatch):
    """Independent uniform codes (the byte filter leaks on them).  Without the variant switch the library's first launch is the
    GUARDED byte-table kernel with the gated u16-table pass queued behind it; forced to give up at once (a budget of 8 candidates per
    workgroup) the gated pass must deliver the same bits; a per-table state then settles on the u16 kernel.  Every path returns what
    the u16 kernel forced through the environment returns, and the oracle's bits on a sample."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    torch.manual_seed(6)
    N, Ks, B, k, D = 1_000_000, 256, 200, 10, 128
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device='cuda')
    cb = torch.randn((M, Ks, D // M), device='cuda')
    q = torch.randn((B, D), device='cuda')
    sk = ops.codes_skew(codes)
    ws = ops.ScanWorkspace()

    def run(state=None):
        return ops.pq_search_topk(LUT_L2, q, cb, sk, k, M, Ks, codes_layout=1, workspace=ws, state=state)

    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '31')
    d31, i31 = run()
    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '50')
    d50, i50 = run()
    assert torch.equal(d50, d31) and torch.equal(i50, i31)
    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    d0, i0 = run()  # stateless: guarded
    assert torch.equal(d0, d31) and torch.equal(i0, i31)
    monkeypatch.setenv('ANNLITE_GUARD_BASE', '8')
    dg, ig = run()
    assert torch.equal(dg, d31) and torch.equal(ig, i31)
    st = _capi.ScanState()
    for _ in range(3):
        dg, ig = run(st)
        torch.cuda.synchronize()
        assert torch.equal(dg, d31) and torch.equal(ig, i31)
    assert st.info()[0] == 2, st.info()
    monkeypatch.delenv('ANNLITE_GUARD_BASE')
    st2 = _capi.ScanState()
    for _ in range(6):
        ds, is_ = run(st2)
        torch.cuda.synchronize()
        assert torch.equal(ds, d31) and torch.equal(is_, i31)
    assert st2.info()[0] in (1, 2), st2.info()
    lut = ops.lut_build(q[:4], cb, LUT_L2).cpu().numpy()
    rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy(), k)
    assert np.array_equal(d0[:4].cpu().numpy(), rd) and np.array_equal(i0[:4].cpu().numpy(), ri)
