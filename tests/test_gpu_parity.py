"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI
(annlite_amd.ops -> libannlite_hip.so), against the CPU oracle and the golden fixtures produced by
the compiled reference.

bar: bit-exact (np.array_equal) for look-up tables, ADC distances, codes (except near-tie encode
mismatches), and neighbour ids at the fixed tie-break (distance asc, row id asc).  Floating-point
tolerance appears only where stated (l2_normalize: 1e-6 relative).
"""
import os

import numpy as np
import pytest

from conftest import golden_names, has_gpu, load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


def untile(lut_tiled: np.ndarray, B, M, Ks, qi):
    """[ceil16(B)/qi][Ks][M][qi] -> [B][M][Ks]"""
    bp = ((B + 15) // 16) * 16
    t = lut_tiled.reshape(bp // qi, Ks, M, qi)
    return np.ascontiguousarray(t.transpose(0, 3, 2, 1).reshape(bp, M, Ks)[:B])


def test_native_library_is_loaded(ops):
    from annlite_amd import _capi

    assert _capi.lib().annlite_hip_abi_version() == 2
    assert _capi.device_count() >= 1
    assert 'gfx950' in _capi.device_arch(0)
    with open('/proc/self/maps') as f:
        assert 'libannlite_hip.so' in f.read()


# ------------------------------------------------------------------------------------------- LUT
@pytest.mark.parametrize('name', golden_names())
def test_lut_bit_exact_vs_reference_fixture(ops, name):
    from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, LUT_IP, LUT_IPDIST, LUT_L2

    g = load_golden(name)
    q, cb = ops.to_dev(g['queries']), ops.to_dev(g['codebooks'])
    assert np.array_equal(ops.lut_build(q, cb, LUT_L2, LAYOUT_BMK).cpu().numpy(), g['lut_l2_batch'])
    assert np.array_equal(ops.lut_build(q, cb, LUT_IP, LAYOUT_BMK).cpu().numpy(), g['lut_ip_batch'])  # MFMA
    assert np.array_equal(ops.lut_build(q, cb, LUT_IPDIST, LAYOUT_BMK).cpu().numpy(), g['dist_mat_inner_product'])
    for qi in (4, 2):
        for kind, want in ((LUT_L2, g['lut_l2_batch']), (LUT_IP, g['lut_ip_batch']), (LUT_IPDIST, g['dist_mat_inner_product'])):
            t = ops.lut_build(q, cb, kind, LAYOUT_TILED, qi).cpu().numpy()
            assert np.array_equal(untile(t, g['B'], g['M'], g['Ks'], qi), want)
        t = ops.lut_retile(ops.to_dev(g['lut_l2_batch']), qi).cpu().numpy()
        assert np.array_equal(untile(t, g['B'], g['M'], g['Ks'], qi), g['lut_l2_batch'])


@pytest.mark.parametrize('name', golden_names())
def test_pq_bind_module_mirror(ops, name):
    """numpy in / numpy out like annlite.pq_bind; the reference's own checks
    (tests/test_pq_bind.py:36-75, tests/test_pq_index.py:30-49) restated."""
    from annlite_amd import pq_bind

    g = load_golden(name)
    t1 = pq_bind.precompute_adc_table(g['queries'][0], g['dsub'], g['Ks'], g['codebooks'])
    assert isinstance(t1, np.ndarray) and np.array_equal(t1, g['lut_l2_single'])
    tb = pq_bind.batch_precompute_adc_table(g['queries'], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(tb, g['lut_l2_batch'])
    assert np.array_equal(pq_bind.batch_precompute_adc_table_ip(g['queries'], g['dsub'], g['Ks'], g['codebooks']), g['lut_ip_batch'])
    ref = np.empty((g['M'], g['Ks']), np.float32)
    for m in range(g['M']):
        ref[m] = np.linalg.norm(g['codebooks'][m] - g['queries'][0][m * g['dsub']:(m + 1) * g['dsub']], axis=1) ** 2
    np.testing.assert_array_almost_equal(ref, t1, decimal=4)
    d = pq_bind.dist_pqcodes_to_codebooks(g['lut_l2_batch'][1], g['codes'])
    assert np.array_equal(np.asarray(d, dtype=np.float32), g['adist'][1])


# ------------------------------------------------------------------------------------ codec
@pytest.mark.parametrize('name', golden_names())
def test_codec_against_fixture(ops, oracle, name):
    from annlite_amd import Metric, PQCodec

    g = load_golden(name)
    for mname, metric in (('euclidean', Metric.EUCLIDEAN), ('inner_product', Metric.INNER_PRODUCT), ('cosine', Metric.COSINE)):
        c = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=metric)
        c.set_codebooks(g['codebooks_cos'] if mname == 'cosine' else g['codebooks'])
        dm = c.get_dist_mat(g['queries'])
        assert dm.dtype == np.float32 and dm.flags['C_CONTIGUOUS'] and dm.shape == (g['B'], g['M'], g['Ks'])
        if mname == 'cosine':
            # host buffers are normalised on the host in the reference's numpy arithmetic: tables bit-equal to the oracle's
            # (= the reference's, tests/test_oracle_vs_reference.py) computed with THIS host's numpy; the committed fixture
            # was produced by another host's numpy build (einsum's summation order is the build's): north-star tolerance
            assert np.array_equal(dm, oracle.get_dist_mat_c(g['queries'], g['codebooks_cos'], oracle.COSINE))
            np.testing.assert_allclose(dm, g['dist_mat_cosine'], rtol=2e-5, atol=2e-6)
        else:
            assert np.array_equal(dm, g['dist_mat_' + mname])
    c = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=Metric.EUCLIDEAN).set_codebooks(g['codebooks'])
    assert c.get_subspace_splitting() == (g['M'], g['Ks'], g['dsub'])
    assert c.get_codebook().shape == (g['M'], g['Ks'], g['dsub'])
    assert np.array_equal(c.precompute_adc(g['queries'][0]).dtable, g['lut_l2_single'])
    codes = c.encode(g['x'])
    assert codes.dtype == g['codes'].dtype and codes.shape == g['codes'].shape
    bad = np.argwhere(codes != g['codes'])
    if len(bad):
        best, second = oracle.encode_gap(g['x'], g['codebooks'])
        for n, m in bad:
            assert (second[n, m] - best[n, m]) / max(second[n, m], 1e-30) < 1e-5
    assert np.array_equal(codes, oracle.encode_c(g['x'], g['codebooks']))  # same fmaf chain => identical
    assert np.array_equal(c.decode(g['codes'][:64]), g['decoded'])
    for b in range(g['B']):
        adist = c.precompute_adc(g['queries'][b]).adist(g['codes'])
        assert np.array_equal(np.asarray(adist, np.float32), g['adist'][b])


def test_l2_normalize(ops, oracle):
    from annlite_amd import math as amath

    rs = np.random.RandomState(3)
    x = rs.randn(257, 96).astype(np.float32)
    x[5] = 0.0  # norm < 10*eps row stays unscaled (math.py:14-16)
    x[6] = 1e-9
    got = amath.l2_normalize(x)  # host buffer: the reference's arithmetic, bit-equal
    assert np.array_equal(got, oracle.l2_normalize(x))
    import torch

    got_dev = amath.l2_normalize(torch.from_numpy(x).cuda()).cpu().numpy()  # device tensor: the kernel (~1 ulp)
    np.testing.assert_allclose(got_dev, oracle.l2_normalize(x), rtol=1e-6, atol=1e-12)
    assert np.array_equal(got[5], x[5]) and np.array_equal(got_dev[5], x[5])


# ------------------------------------------------------------------------------------ scan + top-k
def _scan(ops, codes, lut_bmk, k, layout=0, valid=None, row_base=0):
    import torch
    from annlite_amd._capi import scan_plan

    B, M, Ks = lut_bmk.shape
    codes_d = ops.to_dev(codes)
    plan = scan_plan(codes.shape[0], M, Ks, codes.dtype.itemsize, B, k)
    lut_d = ops.to_dev(lut_bmk)
    if plan.fast:
        lut_d = ops.lut_retile(lut_d, plan.qi)
    if layout == 1:
        codes_d = ops.codes_skew(codes_d)
    vb = None
    if valid is not None:
        bits = np.zeros(((len(valid) + 31) // 32 + 2) * 32, dtype=bool)
        bits[:len(valid)] = valid
        vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    d, i = ops.adc_scan_topk(codes_d, lut_d, B, k, M, Ks, valid_bits=vb, row_base=row_base, codes_layout=layout)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy(), plan


@pytest.mark.parametrize('name', [n for n in golden_names()])
@pytest.mark.parametrize('layout', [0, 1])
def test_scan_topk_vs_fixture(ops, oracle, name, layout):
    from annlite_amd._capi import scan_plan

    g = load_golden(name)
    if g['codes'].dtype != np.uint8 and layout == 1:
        pytest.skip('SKEWED layout is uint8-only')
    k = g['K']
    if layout == 1 and not scan_plan(g['N'], g['M'], g['Ks'], 1, g['B'], k).fast:
        pytest.skip('SKEWED needs the fast plan')  # (M = 128: the generic kernel)
    d, i, plan = _scan(ops, g['codes'], g['lut_l2_batch'], k, layout)
    if g['codes'].dtype == np.uint8:
        rd, ri = oracle.adc_search_c(g['lut_l2_batch'], g['codes'], k)
    else:
        rd, ri = oracle.adc_search_numpy(g['lut_l2_batch'], g['codes'], k)
    assert np.array_equal(d, rd)
    assert np.array_equal(i, ri)
    # and against the reference's own PQIndex.search output (zero rows up to capacity included)
    if g['codes'].dtype == np.uint8:
        cap = int(g['pqindex_capacity'][0])
        table = np.zeros((cap, g['M']), np.uint8)
        table[:g['N']] = g['codes']
        d2, i2, _ = _scan(ops, table, g['lut_l2_batch'], k, layout)
        assert np.array_equal(d2.astype(np.float64), g['pqindex_d'])


SHAPES = [
    # M, Ks, N, B, k
    (16, 256, 5000, 37, 10), (16, 256, 64, 1, 1), (16, 256, 63, 9, 64), (16, 256, 1, 3, 5), (16, 256, 130, 8, 10),
    (8, 256, 3000, 17, 10), (8, 200, 999, 5, 3), (32, 256, 2500, 6, 10), (64, 256, 1500, 5, 10), (64, 100, 700, 2, 7),
    (16, 256, 40000, 24, 50), (4, 256, 1000, 5, 10), (3, 17, 500, 4, 10), (12, 256, 800, 3, 10),
    (64, 256, 70000, 9, 10), (64, 256, 8200, 4, 64), (8, 256, 50000, 33, 10), (32, 256, 30000, 17, 16),
    # round 5: the reference example's own shapes -- examples/pq_benchmark.py:44 loops n_subvectors in [64, 128] at D = 128
    # (dsub 2 and 1): M = 128 has no fast kernel (the generic one), M = 64 at k = 50 the u16 tables
    (128, 256, 3000, 9, 10), (128, 256, 40000, 5, 50), (128, 256, 129, 3, 1), (64, 256, 20000, 6, 50),
]


@pytest.mark.parametrize('M,Ks,N,B,k', SHAPES)
@pytest.mark.parametrize('layout', [0, 1])
def test_scan_topk_random_shapes(ops, oracle, M, Ks, N, B, k, layout):
    from annlite_amd._capi import scan_plan

    if layout == 1 and not scan_plan(N, M, Ks, 1, B, k).fast:
        pytest.skip('SKEWED needs the fast plan')
    rs = np.random.RandomState(M * 7919 + N)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    if M % 2 == 0:
        lut[B // 2] -= 0.5  # negative entries (inner-product style tables)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    d, i, _ = _scan(ops, codes, lut, k, layout, row_base=1000)
    rd, ri = oracle.adc_search_c(lut, codes, k, id_base=1000)
    ri = np.where(ri == -1, -1, ri)
    assert np.array_equal(d, rd)
    assert np.array_equal(i, ri)


@pytest.mark.parametrize('M,dsub', [(64, 2), (128, 1)])
@pytest.mark.parametrize('metric', ['euclidean', 'cosine', 'inner_product'])
def test_example_shapes_queries_in_neighbours_out(ops, oracle, M, dsub, metric):
    """examples/pq_benchmark.py:44 (`for n_subvectors in [64, 128]` at D = 128): sub-vectors of 2 floats and of ONE float,
    queries in -> neighbours out through annlite_pq_search_topk (no fused preparation launch for them: dsub % 4 != 0) and through
    the index plug-in, all three metrics, against the oracle's restatement of the reference pipeline."""
    from annlite_amd import Metric, PQCodec
    from annlite_amd._capi import LUT_IPDIST, LUT_L2
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex

    rs = np.random.RandomState(M + len(metric))
    N, B, k, Ks = 30_000, 19, 10, 256
    D = M * dsub
    A = rs.randn(12, D).astype(np.float32)
    x = (rs.randn(N, 12).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 12).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    xn = oracle.l2_normalize(x) if metric == 'cosine' else x  # (what the index stores codes of: hnsw/index.py:28-29)
    cb = np.stack([xn[rs.choice(N, Ks, replace=False), m * dsub:(m + 1) * dsub] for m in range(M)]).astype(np.float32)
    omet = {'euclidean': oracle.EUCLIDEAN, 'cosine': oracle.COSINE, 'inner_product': oracle.INNER_PRODUCT}[metric]
    met = {'euclidean': Metric.EUCLIDEAN, 'cosine': Metric.COSINE, 'inner_product': Metric.INNER_PRODUCT}[metric]
    codes = oracle.encode_c(xn, cb)
    rd, ri = oracle.index_search(q, cb, codes, omet, k)
    # the C entry point (tables from the queries inside the call)
    qn = oracle.l2_normalize(oracle.l2_normalize(q)) if metric == 'cosine' else q  # (index.py:28-29 + pq.py:309-310: normalised twice)
    kind = LUT_L2 if metric == 'euclidean' else LUT_IPDIST
    d, i = ops.pq_search_topk(kind, ops.to_dev(np.ascontiguousarray(qn, np.float32)), ops.to_dev(cb), ops.to_dev(codes), k, M, Ks,
                              sqrt=(metric == 'euclidean'))
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    # the index plug-in (encode on the GPU, numpy in -> numpy out)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=met).set_codebooks(cb)
    idx = PQFlatGpuIndex(dim=D, metric=met, pq_codec=codec, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    d2, i2 = idx.search_batch(q, limit=k)
    assert np.array_equal(np.asarray(i2), ri) and np.array_equal(np.asarray(d2), rd)


@pytest.mark.parametrize('M', [16, 8, 32, 64])
def test_scan_ties_and_valid_bits(ops, oracle, M):
    """duplicate rows => exact distance ties: ids must come out ascending (the fixed tie-break);
    rows masked out by the validity bitmap (delete marks / `indices` filter) are never returned."""
    rs = np.random.RandomState(11)
    Ks, N, B, k = 256, 4096, 12, 20
    base = rs.randint(0, Ks, size=(64, M)).astype(np.uint8)
    codes = base[rs.randint(0, 64, size=N)]  # every row has ~64 exact duplicates
    lut = rs.rand(B, M, Ks).astype(np.float32)
    for layout in (0, 1):
        d, i, _ = _scan(ops, codes, lut, k, layout)
        rd, ri = oracle.adc_search_c(lut, codes, k)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
    valid = rs.rand(N) < 0.3
    d, i, _ = _scan(ops, codes, lut, k, 1, valid=valid)
    idx = np.where(valid)[0]
    rd, ri = oracle.adc_search_c(lut, codes[idx], k)
    assert np.array_equal(d, rd) and np.array_equal(i, idx[ri])
    # fewer valid rows than k: padded with (+inf, -1)
    valid2 = np.zeros(N, bool)
    valid2[[5, 77, 4000]] = True
    d, i, _ = _scan(ops, codes, lut, k, 0, valid=valid2)
    assert (i[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all()
    assert (np.sort(i[:, :3], axis=1) == np.array([5, 77, 4000])).all()


@pytest.mark.parametrize('shape', [(8, 768, 3000, 7, 10, True), (8, 512, 120_000, 37, 50, True), (16, 512, 90_000, 20, 10, True),
                                   (8, 1024, 50_000, 9, 10, True), (16, 768, 4000, 5, 10, False), (4, 512, 4000, 5, 10, False),
                                   # M = 8, Ks <= 512, k <= 16: the byte-table kernel's uint16-code shape (scan_q8.hip, C16)
                                   (8, 512, 300_000, 70, 10, True), (8, 300, 100_000, 33, 16, True), (8, 512, 5000, 3, 1, True),
                                   # ... and 512 < Ks <= 1024: one entry group, 16 queries per workgroup
                                   (8, 768, 300_000, 70, 10, True), (8, 1024, 100_000, 21, 16, True), (8, 600, 5000, 3, 1, True)])
def test_scan_uint16_codes(ops, oracle, shape):
    """uint16 codes (n_clusters > 256; the reference's own PQ tests run 512 and 768 at n_subvectors = 8,
    tests/test_pq_index.py:80-163): M = 8 up to Ks = 512 with k <= 16 runs the byte-table kernel (adc_scan_q8_kernel<8,..>), M = 8
    up to Ks = 1024 and M = 16 up to Ks = 512 the u16-table kernel with the tables in LDS (adc_scan_qfilter_kernel<.., CODE16>),
    what does not fit runs the generic kernel; all bit-exact."""
    M, Ks, N, B, k, fast = shape
    rs = np.random.RandomState(5)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint16)
    codes[100:164] = codes[100]  # a wave-step of exact ties
    valid = np.ones(N, dtype=bool)
    valid[[3, 100, 101, N - 1]] = False
    d, i, plan = _scan(ops, codes, lut, k, valid=valid)
    assert bool(plan.fast) == fast
    dist = np.stack([oracle.dist_pqcodes_to_codebooks_c(lut[b], codes) for b in range(B)])
    dist[:, ~valid] = np.inf
    for b in range(B):
        rd, ri = oracle.top_k_c(dist[b], k)
        assert np.array_equal(d[b], rd) and np.array_equal(i[b], ri)


@pytest.mark.parametrize('seed', range(8))
def test_scan_uint16_codes_byte_tables_random_shapes(ops, oracle, seed):
    """The byte-table kernel's two uint16-code shapes (M = 8; Ks <= 512: 32 queries per workgroup, the two entry groups read in
    lane-dependent order; Ks <= 1024: 16 queries) on random sizes: ragged N (not a multiple of 64), B (not a multiple of a tile),
    k = 1 .. 16, heavy exact ties, negative table entries, a validity bitmap, a row base -- ids and distances = the oracle's."""
    from annlite_amd._capi import scan_plan

    rs = np.random.RandomState(1000 + seed)
    M = 8
    Ks = int(rs.choice([257, 300, 512, 513, 700, 1024]))
    N = int(rs.choice([1, 63, 64, 65, 1000, 4097, 30_000, 70_001, 200_000]))
    B = int(rs.choice([1, 15, 16, 17, 31, 33, 70]))
    k = int(rs.randint(1, 17))
    plan = scan_plan(N, M, Ks, 2, B, k)
    assert plan.fast == 1 and plan.qt == (32 if Ks <= 512 else 16)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut[B // 2] -= 0.5
    if seed % 2:  # few distinct rows: exact ties everywhere
        base = rs.randint(0, Ks, size=(17, M)).astype(np.uint16)
        codes = base[rs.randint(0, 17, size=N)]
    else:
        codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint16)
    valid = rs.rand(N) < 0.8 if seed % 3 == 0 else None
    d, i, _ = _scan(ops, codes, lut, k, 0, valid=valid, row_base=77)
    if valid is None:
        rd, ri = oracle.adc_search_c(lut, codes, k, id_base=77)
    else:
        idx = np.where(valid)[0]
        if len(idx):
            rd, ri = oracle.adc_search_c(lut, codes[idx], k)
            ri = np.where(ri >= 0, idx[np.clip(ri, 0, len(idx) - 1)] + 77, -1)
        else:
            rd, ri = np.full((B, k), np.inf, np.float32), np.full((B, k), -1, np.int64)
    assert np.array_equal(d, rd), (Ks, N, B, k)
    assert np.array_equal(i, ri), (Ks, N, B, k)


def test_candidates_superset_and_gather(ops, oracle):
    import torch
    from annlite_amd._capi import scan_plan

    rs = np.random.RandomState(21)
    M, Ks, N, B, k = 16, 256, 30000, 9, 32
    lut = rs.rand(B, M, Ks).astype(np.float32)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    plan = scan_plan(N, M, Ks, 1, B, k)
    codes_d = ops.to_dev(codes)
    lut_t = ops.lut_retile(ops.to_dev(lut), plan.qi)
    cd, ci = ops.adc_scan_candidates(codes_d, lut_t, B, k, M, Ks)
    torch.cuda.synchronize()
    cd, ci = cd.cpu().numpy(), ci.cpu().numpy()
    assert ci.shape == (B, plan.n_slices * k)
    rd, ri = oracle.adc_search_c(lut, codes, k)
    for b in range(B):
        assert set(ri[b]).issubset(set(ci[b]))
    # gathered ADC over the candidates reproduces their distances bit-for-bit (space_pq.h PQLookup)
    gd = ops.adc_gather(ops.to_dev(lut), codes_d, ops.to_dev(ci)).cpu().numpy()
    assert np.array_equal(gd, cd)
    for b in range(2):
        want = oracle.adc_gather_c(lut[b], codes, ci[b])
        assert np.array_equal(gd[b], want)


@pytest.mark.parametrize('layout', [0, 1])
def test_byte_table_candidate_lists_with_slice_bounds(ops, oracle, layout):
    """The byte-table kernel as the candidate generator of the re-rank stage (k <= 16 per row slice, nothing shared between
    the slices): every (query, slice) starts from its OWN first bound -- the k-th of the slice's first rows, seed_bound_kernel
    with blockIdx.y = slice -- and its list is the exact top-k of that slice's rows, ascending, bit for bit the oracle's."""
    import torch
    from annlite_amd import Metric, PQCodec
    from annlite_amd._capi import scan_plan

    rs = np.random.RandomState(33)
    N, D, M, B, k = 300_000, 128, 16, 70, 16
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 16).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    codes = oracle.encode_c(x, codec.codebooks)
    lut = oracle.get_dist_mat_c(q, codec.codebooks, oracle.EUCLIDEAN)
    plan = scan_plan(N, M, 256, 1, B, k)
    assert plan.qt == 32 and plan.n_slices >= 8
    codes_d = ops.to_dev(codes)
    if layout == 1:
        codes_d = ops.codes_skew(codes_d)
    lut_t = ops.lut_retile(ops.to_dev(lut), plan.qi)
    valid = np.ones(N, dtype=bool)
    valid[rs.randint(0, N, size=N // 50)] = False  # deleted rows: inside the seed rows of the slices too
    bits = np.packbits(np.concatenate([valid, np.zeros((-N) % 32 + 64, dtype=bool)]), bitorder='little').view(np.int32)
    cd, ci = ops.adc_scan_candidates(codes_d, lut_t, B, k, M, 256, valid_bits=ops.to_dev(bits), codes_layout=layout)
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


def test_topk_merge_and_rows(ops, oracle):
    rs = np.random.RandomState(8)
    G, B, k = 8, 33, 10
    vals = rs.rand(B, G * 500).astype(np.float32)
    vals[:, ::7] = vals[:, 1::7]  # ties
    d, i = ops.topk_rows(ops.to_dev(vals), k)
    for b in range(B):
        rd, ri = oracle.top_k_c(vals[b], k)
        assert np.array_equal(d[b].cpu().numpy(), rd) and np.array_equal(i[b].cpu().numpy(), ri)
    # shard-wise top-k then merge == global top-k
    sd, si = [], []
    for g in range(G):
        dd, ii = ops.topk_rows(ops.to_dev(vals[:, g * 500:(g + 1) * 500]), k, id_base=g * 500 + (1 << 33))
        sd.append(dd)
        si.append(ii)
    import torch

    md, mi = ops.topk_merge(torch.stack(sd), torch.stack(si))
    assert np.array_equal(md.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(mi.cpu().numpy() - (1 << 33), i.cpu().numpy())


def test_packed_scan_and_packed_merge_equal_the_two_tensor_path(ops):
    """The one-buffer form used by the row-sharded search: same bits as (dist, id), and shard-wise packed
    scans merged == one scan over the whole table (ties across shards resolved by global id)."""
    import torch

    from annlite_amd._capi import LAYOUT_TILED, LUT_L2, scan_plan

    torch.manual_seed(5)
    N, M, Ks, B, k, G = 6000, 16, 256, 37, 10, 3
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device='cuda')
    codes[N // 2:N // 2 + 40] = codes[:40]  # duplicates in different shards -> distance ties across shards
    cb = torch.randn((M, Ks, 8), device='cuda')
    q = torch.randn((B, M * 8), device='cuda')

    def lut_for(n):
        plan = scan_plan(n, M, Ks, 1, B, k)
        return ops.lut_build(q, cb, LUT_L2, LAYOUT_TILED, plan.qi) if plan.fast else ops.lut_build(q, cb, LUT_L2)

    d, i = ops.adc_scan_topk(codes, lut_for(N), B, k, M, Ks)
    p = ops.adc_scan_topk_packed(codes, lut_for(N), B, k, M, Ks, row_base=7)
    assert np.array_equal(p[..., 0].cpu().numpy(), i.cpu().numpy() + 7)
    assert np.array_equal(p[..., 1].cpu().numpy().astype(np.uint32).view(np.float32), d.cpu().numpy())
    per = N // G
    parts = [ops.adc_scan_topk_packed(codes[g * per:(g + 1) * per].contiguous(), lut_for(per), B, k, M, Ks,
                                      row_base=g * per) for g in range(G)]
    md, mi = ops.topk_merge_packed(torch.stack(parts))
    assert np.array_equal(md.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(mi.cpu().numpy(), i.cpu().numpy())


@pytest.mark.parametrize('M,dsub,kind', [(16, 8, 1), (8, 16, 1), (32, 4, 1), (16, 6, 1), (16, 8, 3), (64, 12, 1), (64, 12, 3)])
def test_fused_search_entry_equals_lut_build_plus_scan(ops, M, dsub, kind):
    """annlite_pq_search_topk (tables built + quantised inside one launch for L2 on the quantised-filter plan)
    returns the same bits as annlite_lut_build + annlite_adc_scan_topk, and leaves the same fp32 tables."""
    import torch

    from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, scan_plan

    torch.manual_seed(M * 100 + dsub + kind)
    N, Ks, B, k = 30000, 256, 45, 10
    D = M * dsub
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device='cuda')
    cb = torch.randn((M, Ks, dsub), device='cuda')
    q = torch.randn((B, D), device='cuda')
    plan = scan_plan(N, M, Ks, 1, B, k)
    lut = ops.lut_build(q, cb, kind, LAYOUT_TILED if plan.fast else LAYOUT_BMK, plan.qi)
    d0, i0 = ops.adc_scan_topk(codes, lut, B, k, M, Ks)
    d1, i1 = ops.pq_search_topk(kind, q, cb, codes, k, M, Ks)
    assert np.array_equal(d0.cpu().numpy(), d1.cpu().numpy()) and np.array_equal(i0.cpu().numpy(), i1.cpu().numpy())
    p = ops.pq_search_topk(kind, q, cb, codes, k, M, Ks, row_base=3, packed=True)
    assert np.array_equal(p[..., 0].cpu().numpy(), i0.cpu().numpy() + 3)
    assert np.array_equal(p[..., 1].cpu().numpy().astype(np.uint32).view(np.float32), d0.cpu().numpy())


# ------------------------------------------------------------------------------------ index plugin
@pytest.mark.parametrize('name', ['c1_m8_d128', 'c2_m16_d128', 'c4_m64_d768'])
@pytest.mark.parametrize('mname,metric', [('euclidean', 1), ('inner_product', 2), ('cosine', 3)])
def test_index_plugin_vs_oracle_and_hnsw_fixture(ops, oracle, name, mname, metric):
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    g = load_golden(name)
    cb = g['codebooks_cos'] if metric == 3 else g['codebooks']
    codec = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=Metric(metric)).set_codebooks(cb)
    idx = PQFlatGpuIndex(dim=g['D'], metric=Metric(metric), pq_codec=codec, initial_size=512, expand_step_size=512)
    idx.add_with_ids(g['x'], np.arange(g['N']))
    assert idx.size == g['N'] and idx.capacity >= g['N']
    ref_codes = g['codes_cos'] if metric == 3 else g['codes']
    d, i = idx.search_batch(g['queries'], limit=g['K'])
    # oracle on the reference's codes: exact parity for every metric -- ids AND distances (encode agrees on these fixtures;
    # cosine queries are normalised on the host in the reference's numpy arithmetic, twice, as hnsw/index.py:28-29 +
    # pq.py:309-310 do)
    rd, ri = oracle.index_search(g['queries'], cb, ref_codes, metric, g['K'])
    if metric == 3:
        # the stored rows were normalised by this host's numpy too; their codes can differ from the fixture's only at
        # encode near-ties (the fixture's vectors were normalised by another numpy build)
        mine = oracle.encode_c(oracle.l2_normalize(g['x']), cb)
        rd, ri = oracle.index_search(g['queries'], cb, mine, metric, g['K'])
    assert np.array_equal(d, rd) and np.array_equal(i, ri)
    # HnswIndex(PQ) fixture: the exhaustive scan is never worse than the graph walk, and for EUCLIDEAN /
    # IP its distance for every id the reference returned is bit-identical
    key = 'hnsw_%s_d' % mname
    if key in g:
        assert (d <= g[key] + (1e-5 if metric == 3 else 0)).all()
    # single-query reference signature
    d1, i1 = idx.search(g['queries'][0], limit=g['K'])
    assert np.array_equal(d1, d[0]) and np.array_equal(i1, i[0])
    # delete marks
    idx.delete([int(i[0][0])])
    d2, i2 = idx.search(g['queries'][0], limit=g['K'])
    assert int(i[0][0]) not in i2 and idx.size == g['N'] - 1
    # subset search (`indices`), pq_index.py:42-44
    sub = np.arange(0, g['N'], 3)
    d3, i3 = idx.search(g['queries'][1], limit=5, indices=sub)
    assert set(i3).issubset(set(sub))
    # ... against the oracle ON THE SUBSET (minus the row deleted above): ids and distances, not just membership
    live = sub[sub != int(i[0][0])]
    my_codes = mine if metric == 3 else ref_codes
    rd3, ri3 = oracle.index_search(g['queries'][1:2], cb, my_codes[live], metric, 5)
    assert np.array_equal(i3, live[ri3[0]]) and np.array_equal(d3, rd3[0])


def test_index_untrained_raises(ops):
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    codec = PQCodec(dim=64, n_subvectors=8)
    idx = PQFlatGpuIndex(dim=64, metric=Metric.EUCLIDEAN, pq_codec=codec)
    with pytest.raises(RuntimeError):
        idx.add_with_ids(np.zeros((3, 64), np.float32), [0, 1, 2])
    with pytest.raises(RuntimeError):
        idx.search(np.zeros(64, np.float32))


def test_index_dump_load_large_k_and_rerank(ops, oracle, tmp_path):
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    g = load_golden('c2_m16_d128')
    codec = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=Metric.EUCLIDEAN).set_codebooks(g['codebooks'])
    idx = PQFlatGpuIndex(dim=g['D'], metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=256, expand_step_size=300)
    idx.add_with_ids(g['x'][:500], np.arange(500))
    idx.add_with_ids(g['x'][500:], np.arange(500, g['N']))  # forces capacity expansion
    d, i = idx.search_batch(g['queries'], limit=10)
    p = tmp_path / 'cell_0.pqgpu'
    idx.dump(p)
    idx2 = PQFlatGpuIndex(dim=g['D'], metric=Metric.EUCLIDEAN, pq_codec=codec, index_file=p)
    d2, i2 = idx2.search_batch(g['queries'], limit=10)
    assert np.array_equal(d, d2) and np.array_equal(i, i2) and idx2.size == idx.size
    # k > 64 path
    dk, ik = idx.search_batch(g['queries'], limit=100)
    rd, ri = oracle.index_search(g['queries'], g['codebooks'], g['codes'], 1, 100)
    assert np.array_equal(dk, rd) and np.array_equal(ik, ri)
    # exact re-rank: result = exact distances of the best candidates
    idr = PQFlatGpuIndex(dim=g['D'], metric=Metric.EUCLIDEAN, pq_codec=codec, rerank=True, initial_size=2048)
    idr.add_with_ids(g['x'], np.arange(g['N']))
    dr, ir = idr.search_batch(g['queries'], limit=10, rerank_k=64)
    exact = np.sqrt(((g['queries'][:, None, :] - g['x'][None, :, :]) ** 2).sum(-1))
    truth = np.argsort(exact, axis=1)[:, :10]
    recall = np.mean([len(set(ir[b]) & set(truth[b])) / 10 for b in range(g['B'])])
    assert recall >= 0.9, recall
    np.testing.assert_allclose(dr, np.take_along_axis(exact, ir, axis=1), rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('metric_name', ['EUCLIDEAN', 'INNER_PRODUCT'])
def test_large_k_batched_path_ties_and_deletes(ops, oracle, metric_name):
    """limit > 64 (pq_flat_gpu._search_large_k: chunked distance matrix + one batched top-k over (sum, row) keys): heavy ties
    (50 distinct code rows among 3000), deleted rows, negative sums (inner product) -- ids and distances = the oracle's."""
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    g = load_golden('c2_m16_d128')
    metric = getattr(Metric, metric_name)
    rs = np.random.RandomState(11)
    N = 3000
    base = g['x'][:50]
    x = base[rs.randint(0, 50, N)]
    codec = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=metric).set_codebooks(g['codebooks'])
    idx = PQFlatGpuIndex(dim=g['D'], metric=metric, pq_codec=codec, initial_size=4096)
    idx.add_with_ids(x, np.arange(N))
    gone = rs.choice(N, 300, replace=False)
    idx.delete(gone.tolist())
    keep = np.setdiff1d(np.arange(N), gone)
    codes = ops.codes_to_numpy(ops.pq_encode(ops.to_dev(x), codec.codebooks_dev))
    q = g['queries'][:7]
    for k in (65, 200, 2900):
        d, i = idx.search_batch(q, limit=k)
        rd, ri = oracle.index_search(q, g['codebooks'], codes[keep], int(metric), k)
        ri = np.where(ri >= 0, keep[np.clip(ri, 0, len(keep) - 1)], -1)
        assert np.array_equal(i, ri), (metric_name, k)
        fin = np.isfinite(rd)
        assert np.array_equal(np.isfinite(d), fin)
        np.testing.assert_array_equal(d[fin], rd[fin])


# ------------------------------------------------------------------------------------ AnnLite facade
def test_annlite_facade_end_to_end(ops, tmp_path):
    """tests/test_pq_index.py:52-77 restated + result shape of AnnLite.search (container.py:226-233)."""
    from annlite_amd import AnnLite
    from annlite_amd.index import Document, DocumentArray

    rs = np.random.RandomState(4)
    N, D = 1000, 64
    X = rs.rand(N, D).astype(np.float32)
    docs = DocumentArray([Document(id=f'{i}', embedding=X[i], tags={'x': str(i), 'price': float(i)}) for i in range(N)])
    ann = AnnLite(D, data_path=tmp_path / 'idx', n_subvectors=8, metric='euclidean')
    with pytest.raises(RuntimeError):
        ann.index(docs)
    with pytest.raises(RuntimeError):
        ann.search(docs)
    ann._pq_codec.seed = 1
    ann.train(X)
    assert ann.is_trained and ann._pq_codec_path.exists()
    ann.index(docs)
    assert ann.stat['total_docs'] == N and ann.stat['index_size'] == N and ann.stat['metric'] == 'EUCLIDEAN'
    query = DocumentArray([Document(embedding=X[i]) for i in range(10)])
    ann.search(query, limit=7)
    for qi, qd in enumerate(query):
        assert len(qd.matches) == 7
        vals = [m.scores['euclidean'].value for m in qd.matches]
        assert vals == sorted(vals)
        assert qd.matches[0].id == str(qi)  # a vector's own code is its nearest ADC neighbour here
    dists, ids = ann.search_numpy(X[:4], limit=5)
    assert len(dists) == 4 and ids[0].dtype.kind == 'i' and ids[0][0] == 0
    # filter -> GPU bitmap
    ann.search(query, filter={'price': {'$lt': 50.0}}, limit=5)
    assert all(int(m.id) < 50 for qd in query for m in qd.matches)
    # delete / update
    ann.delete(['0'])
    ann.search(query, limit=3)
    assert query[0].matches[0].id != '0'
    # delete through a DocumentArray of INDEXED documents (index.py:389-414 takes ids or documents): they are gone
    # from the table and from the results
    before = ann.stat['total_docs']
    ann.delete(DocumentArray([Document(id='1'), Document(id='2')]))
    assert ann.stat['total_docs'] == before - 2 and '1' not in ann._id2offset
    ann.search(query, limit=3)
    assert query[1].matches[0].id != '1' and query[2].matches[0].id != '2'
    # an id that is already indexed (or twice in one batch) violates the reference table's UNIQUE(_doc_id)
    # (storage/table.py:203): nothing is inserted
    import sqlite3

    n_docs, n_rows = ann.stat['total_docs'], len(ann._offset2id)
    with pytest.raises(sqlite3.IntegrityError):
        ann.index(DocumentArray([Document(id='5', embedding=X[5])]))
    with pytest.raises(sqlite3.IntegrityError):
        ann.index(DocumentArray([Document(id='n1', embedding=X[5]), Document(id='n1', embedding=X[6])]))
    assert ann.stat['total_docs'] == n_docs and len(ann._offset2id) == n_rows and 'n1' not in ann._id2offset
    # a second AnnLite over the same data_path picks the trained codec up (index.py:136-140)
    ann2 = AnnLite(D, data_path=tmp_path / 'idx', n_subvectors=8, metric='euclidean')
    assert ann2.is_trained
    assert np.array_equal(ann2._pq_codec.codebooks, ann._pq_codec.codebooks)


def test_kmeans_fit_quality_vs_sklearn(ops):
    """Training parity is statistical (pq.py:89-115 is unseeded sklearn): reconstruction MSE of the
    GPU k-means must be within 5 % of sklearn KMeans on the same data."""
    from sklearn.cluster import KMeans
    from annlite_amd import Metric, PQCodec

    rs = np.random.RandomState(0)
    z = rs.randn(4000, 8).astype(np.float32)
    x = (z @ rs.randn(8, 32).astype(np.float32) + 0.05 * rs.randn(4000, 32)).astype(np.float32)
    c = PQCodec(dim=32, n_subvectors=4, n_clusters=64, metric=Metric.EUCLIDEAN, n_init=2)
    c.seed = 7
    c.fit(x, iter=50)
    assert c.is_trained and c.codebooks.shape == (4, 64, 8)
    rec = c.decode(c.encode(x))
    mse = float(((rec - x) ** 2).mean())
    ref_mse = 0.0
    for m in range(4):
        km = KMeans(n_clusters=64, n_init=2, max_iter=50, random_state=0).fit(x[:, m * 8:(m + 1) * 8])
        ref_mse += km.inertia_ / x.shape[0] / 32
    assert mse <= 1.05 * ref_mse, (mse, ref_mse)
    # partial_fit + build_codebook give the same codebook shape (tests/test_codec.py:65-70)
    c2 = PQCodec(dim=32, n_subvectors=4, n_clusters=64)
    for s in range(0, 4000, 500):
        c2.partial_fit(x[s:s + 500])
    c2.build_codebook()
    assert c2.codebooks.shape == c.codebooks.shape and c2.is_trained


def test_partial_fit_quality_and_search_vs_minibatch_kmeans(ops, oracle):
    """The streaming path (pq.py:117-156: MiniBatchKMeans.partial_fit per sub-space, then build_codebook): parity is
    statistical like `fit`'s -- reconstruction MSE within 10 % of sklearn's MiniBatchKMeans fed the same stream -- and
    a search over codes of the streamed codec equals the oracle on those codebooks bit for bit."""
    from sklearn.cluster import MiniBatchKMeans
    from annlite_amd import Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(1)
    z = rs.randn(6000, 8).astype(np.float32)
    x = (z @ rs.randn(8, 32).astype(np.float32) + 0.05 * rs.randn(6000, 32)).astype(np.float32)
    stream = np.array_split(x[:5000], 10)
    c = PQCodec(dim=32, n_subvectors=4, n_clusters=64, metric=Metric.EUCLIDEAN)
    c.seed = 3
    for b in stream:
        c.partial_fit(b)
    assert not c.is_trained
    c.build_codebook()
    assert c.is_trained and c.codebooks.shape == (4, 64, 8)
    mse = float(((c.decode(c.encode(x)) - x) ** 2).mean())
    ref_mse = 0.0
    for m in range(4):
        km = MiniBatchKMeans(n_clusters=64, random_state=0, n_init=1)
        for b in stream:
            km.partial_fit(b[:, m * 8:(m + 1) * 8])
        sub = x[:, m * 8:(m + 1) * 8]
        d2 = ((sub[:, None, :] - km.cluster_centers_[None, :, :].astype(np.float32)) ** 2).sum(2)
        ref_mse += float(d2.min(1).mean()) / 32  # squared error per row, averaged over the 32 coordinates
    assert mse <= 1.10 * ref_mse, (mse, ref_mse)
    idx = PQFlatGpuIndex(dim=32, metric=Metric.EUCLIDEAN, pq_codec=c, initial_size=6000)
    idx.add_with_ids(x, np.arange(6000))
    q = x[5000:5040] + 0.01
    d, i = idx.search_batch(q.astype(np.float32), limit=10)
    rd, ri = oracle.index_search(q.astype(np.float32), c.codebooks, oracle.encode_c(x, c.codebooks), oracle.EUCLIDEAN, 10)
    assert np.array_equal(d, rd) and np.array_equal(i, ri)


def test_kmeanspp_seeding_beats_random_rows(ops):
    """k-means++ (sklearn's default init behind pq.py:106-110) vs random rows after the SAME few Lloyd iterations:
    the seeded run must not be worse (it is the default; `init='random'` keeps the old behaviour)."""
    from annlite_amd import Metric, PQCodec

    rs = np.random.RandomState(5)
    cent = rs.randn(40, 32).astype(np.float32) * 4
    x = (cent[rs.randint(0, 40, 8000)] + 0.3 * rs.randn(8000, 32)).astype(np.float32)
    res = {}
    for init in ('k-means++', 'random'):
        c = PQCodec(dim=32, n_subvectors=4, n_clusters=64, metric=Metric.EUCLIDEAN, n_init=1)
        c.seed, c.init = 11, init
        c.fit(x, iter=3)
        res[init] = float(c.inertia_.sum())
    assert res['k-means++'] <= 1.02 * res['random'], res


# ------------------------------------------------------------------------------------ filter-kernel specific
def test_filter_slack_with_badly_conditioned_tables(ops, oracle):
    """The default scan kernel filters with a sum taken in a per-lane ROTATED order and recomputes the
    exact ascending-m sum only for rows inside `thr + slack`.  Tables whose entries span 12 orders of
    magnitude make the two summation orders differ in the leading bits; ids and distances must still be
    the oracle's, bit for bit, for every lane skew (rows 0..63 of a wave-step have different orders)."""
    rs = np.random.RandomState(77)
    M, Ks, N, B, k = 16, 256, 20000, 16, 25
    mag = 10.0 ** rs.uniform(-6, 6, size=(B, M, 1))
    lut = (rs.rand(B, M, Ks) * mag).astype(np.float32)
    lut[1] = -lut[1]                      # all-negative table
    lut[2, ::2] = -lut[2, ::2]            # mixed signs: catastrophic cancellation between sub-spaces
    lut[3] = np.float32(1e-30) * rs.rand(M, Ks).astype(np.float32)  # denormal-range sums
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[5000:5064] = codes[5000]        # a full wave-step of duplicates: 64 exact ties, 64 different skews
    for layout in (0, 1):
        d, i, plan = _scan(ops, codes, lut, k, layout)
        assert plan.fast
        rd, ri = oracle.adc_search_c(lut, codes, k)
        assert np.array_equal(d, rd)
        assert np.array_equal(i, ri)


def test_pad_queries_of_a_ragged_batch_generate_no_candidates(ops, oracle, monkeypatch):
    """B not a multiple of the 16-query tile: the pad queries (all-zero tables) must not pass the integer filter.
    They once made a 1-query search 15x slower than a 16-query one; counted through the debug counters."""
    import torch

    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '1')
    torch.manual_seed(2)
    N, M, Ks, k = 200_000, 16, 256, 10
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device='cuda')
    cb = torch.randn((M, Ks, 8), device='cuda')
    for B in (1, 3):
        q = torch.randn((B, M * 8), device='cuda')
        d, i = ops.pq_search_topk(LUT_L2, q, cb, ops.codes_skew(codes), k, M, Ks, codes_layout=1)
        c = _capi.debug_counters()
        # candidate rows of the whole launch (the pads alone used to contribute 15 * N): the u16 kernels count them in
        # [4], the byte-table kernel (32-query tiles: 31 pads here) counts the pushed candidates in [1]
        n_cand = c[1] if _capi.scan_plan(N, M, Ks, 1, B, k).qt == 32 else c[4]
        assert n_cand < N // 4, c
        lut = ops.lut_build(q, cb, LUT_L2).cpu().numpy()
        rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy(), k)
        assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)


@pytest.mark.parametrize('variant', ['0', '50', '31', '30', '32'])
def test_scan_kernel_variants_agree(ops, oracle, variant, monkeypatch):
    """every selectable M=16 scan kernel (byte filter tables = the default / 50; u16 filter tables with 16 / 12 / 8 waves)
    is bit-exact; selected through the environment (ANNLITE_SCAN_VARIANT: A/B measurements).  Without the variable the
    library chooses itself (test_library_picks_the_scan_kernel)."""
    from annlite_amd import _capi

    rs = np.random.RandomState(3)
    M, Ks, N, B, k = 16, 256, 70000, 40, 10
    lut = rs.rand(B, M, Ks).astype(np.float32)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    rd, ri = oracle.adc_search_c(lut, codes, k)
    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
    assert _capi.scan_plan(N, M, Ks, 1, B, k).qt == (32 if variant in ('0', '50') else 16)
    for layout in (1, 0):
        d, i, _ = _scan(ops, codes, lut, k, layout)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)


@pytest.mark.parametrize('variant', ['0', '31'])
def test_m64_scan_kernels_agree(ops, oracle, variant, monkeypatch):
    """M = 64 (BASELINE config 4): the byte-table kernel (the default for k <= 16: 8 queries per 8-byte entry, byte sums widened
    into u16 sums) and the u16-table kernel, on random tables / codes and on structured data, both layouts, ragged batch --
    bit for bit the oracle's."""
    from annlite_amd import Metric, PQCodec, _capi

    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
    rs = np.random.RandomState(13)
    M, Ks, N, B, k = 64, 256, 60000, 21, 10
    assert _capi.scan_plan(N, M, Ks, 1, B, k).qt == (8 if variant == '0' else 4)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[1000:1040] = codes[17]  # ties
    rd, ri = oracle.adc_search_c(lut, codes, k)
    for layout in (1, 0):
        d, i, _ = _scan(ops, codes, lut, k, layout)
        assert np.array_equal(d, rd) and np.array_equal(i, ri), layout
    D = 768
    A = rs.randn(64, D).astype(np.float32)
    x = (rs.randn(120_000, 64).astype(np.float32) @ A + 0.05 * rs.randn(120_000, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(40, 64).astype(np.float32) @ A + 0.05 * rs.randn(40, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=4)
    codes = oracle.encode_c(x, codec.codebooks)
    lut = oracle.get_dist_mat_c(q, codec.codebooks, oracle.EUCLIDEAN)
    rd, ri = oracle.adc_search_c(lut, codes, k)
    for layout in (1, 0):
        d, i, _ = _scan(ops, codes, lut, k, layout)
        assert np.array_equal(d, rd) and np.array_equal(i, ri), layout


def _event_ms(fn, reps=5):
    import torch

    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def test_library_picks_the_scan_kernel(ops, oracle, monkeypatch):
    """Kernel choice inside the library (annlite_hip.h: annlite_scan_state).  2M rows of independent
    uniform codes -- the byte filter leaks on them (~38 ms per 1024-query batch at 10M rows against 3.4 for the u16 kernel):
      * a plain C-ABI call (no variant, no state) must stay within 1.3x of the u16 kernel forced through the environment:
        its guarded byte-table launch gives up within microseconds and the gated u16 pass behind it redoes the scan;
      * with a per-table state the library settles on the u16 kernel after the first launch has completed;
      * data with structure settles on byte tables;
      * all of it bit-identical to the CPU oracle on a sample of the queries."""
    import torch
    from annlite_amd import Metric, PQCodec, _capi
    from annlite_amd._capi import LUT_L2

    monkeypatch.delenv('ANNLITE_SCAN_VARIANT', raising=False)
    torch.manual_seed(5)
    N, M, Ks, B, k, D = 2_000_000, 16, 256, 1024, 10, 128
    codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device='cuda')
    cb = torch.randn((M, Ks, D // M), device='cuda')
    q = torch.randn((B, D), device='cuda')
    sk = ops.codes_skew(codes)
    ws = ops.ScanWorkspace()

    def run(state=None):
        return ops.pq_search_topk(LUT_L2, q, cb, sk, k, M, Ks, codes_layout=1, workspace=ws, state=state)

    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '31')
    ms_u16 = _event_ms(run)
    d31, i31 = run()
    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    ms_plain = _event_ms(run)
    d0, i0 = run()
    assert torch.equal(d0, d31) and torch.equal(i0, i31)
    assert ms_plain <= 1.3 * ms_u16, (ms_plain, ms_u16)
    st = _capi.ScanState()
    assert st.info()[0] == 0
    for _ in range(6):  # (a launch that gave up settles it at the next call; the grey zone takes two timed calls more)
        ds, is_ = run(st)
        torch.cuda.synchronize()
    assert st.info()[0] in (1, 2), st.info()  # settled (which kernel depends on how badly THIS table leaks)
    ms_state = _event_ms(lambda: run(st))
    assert ms_state <= 1.3 * ms_u16, (ms_state, ms_u16, st.info())
    assert torch.equal(ds, d31) and torch.equal(is_, i31)
    # the give-up path itself, forced: a budget of 8 candidates per workgroup trips at once -- the gated u16 pass must deliver
    # the same bits, and a state must settle on the u16 kernel.  (With the early epoch end: the scanning waves stop at step 15
    # while the consumer counts the transient's candidates -- without it the rows drawn raise the budget as fast as this table leaks.)
    monkeypatch.setenv('ANNLITE_GUARD_BASE', '8')
    monkeypatch.setenv('ANNLITE_Q8_TUNE', '15,16,384,3')
    dg, ig = run()
    assert torch.equal(dg, d31) and torch.equal(ig, i31)
    st3 = _capi.ScanState()
    for _ in range(3):
        dg, ig = run(st3)
        torch.cuda.synchronize()
        assert torch.equal(dg, d31) and torch.equal(ig, i31)
    assert st3.info()[0] == 2, st3.info()
    monkeypatch.delenv('ANNLITE_GUARD_BASE')
    monkeypatch.delenv('ANNLITE_Q8_TUNE')
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/kernel_choice_2m_uniform_codes.txt', 'w') as f:
        f.write('2M x 16 uniform codes, 1024 queries, ms per batch: u16 forced %.3f, stateless (guarded) %.3f, with state %.3f (%s)\n'
                % (ms_u16, ms_plain, ms_state, st.info()))
    lut = ops.lut_build(q[:4], cb, LUT_L2).cpu().numpy()
    rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy(), k)
    assert np.array_equal(d0[:4].cpu().numpy(), rd) and np.array_equal(i0[:4].cpu().numpy(), ri)
    # data with structure: byte tables stay
    rs = np.random.RandomState(21)
    A = rs.randn(16, D).astype(np.float32)
    x = (rs.randn(400_000, 16).astype(np.float32) @ A + 0.05 * rs.randn(400_000, D).astype(np.float32)).astype(np.float32)
    qs = (rs.randn(64, 16).astype(np.float32) @ A + 0.05 * rs.randn(64, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 4
    codec.fit(x[:8192], iter=5)
    codes2 = oracle.encode_c(x, codec.codebooks)
    st2 = _capi.ScanState()
    sk2 = ops.codes_skew(ops.to_dev(codes2))
    for _ in range(3):
        d2, i2 = ops.pq_search_topk(LUT_L2, ops.to_dev(qs), codec.codebooks_dev, sk2, k, M, Ks, codes_layout=1, workspace=ws, state=st2)
        torch.cuda.synchronize()
    assert st2.info()[0] == 1, st2.info()  # byte tables
    lut2 = oracle.get_dist_mat_c(qs, codec.codebooks, oracle.EUCLIDEAN)
    rd2, ri2 = oracle.adc_search_c(lut2, codes2, k)
    assert np.array_equal(d2.cpu().numpy(), rd2) and np.array_equal(i2.cpu().numpy(), ri2)


@pytest.mark.parametrize('tune', [('1,2,192,0', '64', '7'), ('3,4,448,3', '127', '8'), ('100000,2,384,3', '96', '4')])
def test_byte_table_kernel_epochs_and_rebuilds(ops, oracle, tune, monkeypatch):
    """The byte-table kernel's epoch schedule, ring limit, table resolution and rebuild rule are tunable through the
    environment (ANNLITE_Q8_TUNE / _TARGET / _REBUILD: DESIGN 3.1).  Whatever they are the result is the oracle's, bit for
    bit -- including a schedule that ends an epoch every other step and rebuilds the tables as soon as a bound moves
    (the default schedule is sparse: small tables never reach its first epoch end), and one without any epoch."""
    from annlite_amd import Metric, PQCodec, _capi

    rs = np.random.RandomState(21)
    N, D, M, B, k = 200_000, 128, 16, 96, 10
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
    monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '1')
    monkeypatch.setenv('ANNLITE_SEED_ROWS', '4096')  # (a loose first bound: the tables have something to follow)
    assert _capi.scan_plan(N, M, 256, 1, B, k).qt == 32
    for layout in (0, 1):
        d, i, _ = _scan(ops, codes, lut, k, layout)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
        if tune[0].startswith('1,2'):
            assert _capi.debug_counters()[5] > 0  # tables were rebuilt


def test_state_times_both_kernels_in_the_grey_zone(ops, oracle, monkeypatch):
    """Uniform VECTORS (the reference's own test distribution, tests/test_pq_bind.py:19) through a trained codec: the byte-table
    launch completes within its give-up budget but with ~10^4 candidates per query -- the grey zone, where counting candidates
    cannot say which kernel is faster.  A per-table state then TIMES one call of each kernel (events in the caller's stream,
    read without waiting) and keeps the faster one: after six calls it has settled, the results are the oracle's bit for
    bit in every phase, and the settled path is within 15 % of the better of the two kernels forced through the environment."""
    import torch
    from annlite_amd import Metric, PQCodec, _capi
    from annlite_amd._capi import LUT_L2

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    N, D, M, Ks, B, k = 2_000_000, 128, 16, 256, 512, 10
    gen = lambda n: torch.rand((n, D), generator=g, device=dev)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 3
    codec.fit(gen(20480), iter=10)
    cb = codec.codebooks_dev
    codes = ops.codes_skew(torch.cat([ops.pq_encode(gen(500_000), cb) for _ in range(4)]))
    q = gen(B)

    def run(state=None):
        return ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=1, state=state)

    for name in ('ANNLITE_SCAN_VARIANT', 'ANNLITE_Q8_TUNE', 'ANNLITE_Q8_TARGET', 'ANNLITE_SEED_ROWS', 'ANNLITE_GUARD_BASE'):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '31')
    ms_u16 = _event_ms(run)
    d31, i31 = run()
    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '50')
    ms_byte = _event_ms(run)
    monkeypatch.delenv('ANNLITE_SCAN_VARIANT')
    st = _capi.ScanState()
    for _ in range(6):
        ds, is_ = run(st)
        assert torch.equal(ds, d31) and torch.equal(is_, i31)
        torch.cuda.synchronize()
    kernel = st.info()[0]
    assert kernel in (1, 2), st.info()
    ms_state = _event_ms(lambda: run(st))
    assert ms_state <= 1.15 * min(ms_byte, ms_u16), (ms_state, ms_byte, ms_u16, st.info())
    lut = ops.lut_build(q[:8], cb, LUT_L2).cpu().numpy()
    rd, ri = oracle.adc_search_c(lut, ops.codes_skew(codes, inverse=True).cpu().numpy(), k, threads=oracle.max_threads())
    assert np.array_equal(d31[:8].cpu().numpy(), rd) and np.array_equal(i31[:8].cpu().numpy(), ri)
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/kernel_choice_2m_uniform_vectors.txt', 'w') as f:
        f.write('2M x 16 codes of uniform vectors, %d queries: u16 tables %.3f ms, byte tables %.3f ms, with state %.3f ms, state %s\n'
                % (B, ms_u16, ms_byte, ms_state, st.info()))


def test_byte_table_kernel_on_the_bench_distribution_2m_rows(ops, oracle, monkeypatch):
    """The headline kernel on the bench's own data (rank-16 latent Gaussian + noise, trained codec) at 2M rows, DEFAULT epoch
    schedule / table resolution / rebuild rule, 256 queries, bit for bit against the CPU oracle -- and the debug counters
    must show that the machinery the small tests never reach has run: an epoch end (all waves met at the barrier) and a
    table rebuild.  The two knobs changed to make a rebuild certain at this size: the first bound is made loose (512 seed rows)
    and a slot asks for a new table when its T has fallen below 6/8 (default 4/8) of what the table was built for -- 8 tiles x
    32 slices of 62 500 rows: at the epoch end after step 15 the slices have seen 491k rows, 960x the seed's.  (With the
    defaults a rebuild is rare: 10-20 of a 10M-row launch's 256 workgroups.)"""
    import torch
    from annlite_amd import Metric, PQCodec, _capi
    from annlite_amd._capi import LUT_L2

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    N, D, M, Ks, B, k = 2_000_000, 128, 16, 256, 256, 10
    A = torch.randn((16, D), generator=g, device=dev)
    gen = lambda n: (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.fit(gen(20480), iter=20)
    cb = codec.codebooks_dev
    codes = torch.cat([ops.pq_encode(gen(500_000), cb) for _ in range(4)])
    q = gen(B)
    monkeypatch.delenv('ANNLITE_SCAN_VARIANT', raising=False)
    for name in ('ANNLITE_Q8_TUNE', 'ANNLITE_Q8_TARGET'):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv('ANNLITE_SEED_ROWS', '512')
    monkeypatch.setenv('ANNLITE_Q8_REBUILD', '6')
    monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '1')
    assert _capi.scan_plan(N, M, Ks, 1, B, k).qt == 32
    st = _capi.ScanState()
    d, i = ops.pq_search_topk(LUT_L2, q, cb, ops.codes_skew(codes), k, M, Ks, codes_layout=1, state=st)
    torch.cuda.synchronize()
    c = _capi.debug_counters()
    assert c[7] > 0, c   # wave 0 spent cycles at epoch ends: the barriers were reached
    assert c[5] > 0, c   # tables were rebuilt
    monkeypatch.delenv('ANNLITE_DEBUG_COUNTERS')
    lut = ops.lut_build(q, cb, LUT_L2).cpu().numpy()
    rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy(), k, threads=oracle.max_threads())
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)
    # the defaults too, PLAIN layout
    monkeypatch.delenv('ANNLITE_SEED_ROWS')
    monkeypatch.delenv('ANNLITE_Q8_REBUILD')
    d, i = ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=0, state=st)
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)


def test_full_size_properties_config2(ops, oracle):
    """BASELINE config 2 at full size (1M x 128-d, PQ m=16, batch 1024, k=10) through size-independent
    properties: ascending order, every returned distance equals the gathered ADC distance of that row
    (adc_gather kernel = space_pq.h PQLookup), a second run is identical (idempotence), PLAIN and SKEWED
    tables agree, and a sample of queries equals the CPU oracle exactly."""
    import torch
    from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, LUT_L2, scan_plan

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    N, D, M, Ks, B, k = 1_000_000, 128, 16, 256, 1024, 10
    cb = torch.randn((M, Ks, D // M), generator=g, device=dev)
    x = torch.randn((N, D), generator=g, device=dev)
    q = torch.randn((B, D), generator=g, device=dev)
    codes = ops.pq_encode(x, cb)
    plan = scan_plan(N, M, Ks, 1, B, k)
    lut_t = ops.lut_build(q, cb, LUT_L2, LAYOUT_TILED, plan.qi)
    lut_b = ops.lut_build(q, cb, LUT_L2, LAYOUT_BMK)
    d, i = ops.adc_scan_topk(codes, lut_t, B, k, M, Ks)
    d2, i2 = ops.adc_scan_topk(ops.codes_skew(codes), lut_t, B, k, M, Ks, codes_layout=1)
    d3, i3 = ops.adc_scan_topk(codes, lut_t, B, k, M, Ks)
    torch.cuda.synchronize()
    assert torch.equal(d, d2) and torch.equal(i, i2) and torch.equal(d, d3) and torch.equal(i, i3)
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    ties = d[:, 1:] == d[:, :-1]
    assert bool((i[:, 1:][ties] > i[:, :-1][ties]).all())
    assert bool(((i >= 0) & (i < N)).all())
    assert torch.equal(ops.adc_gather(lut_b, codes, i), d)
    # threshold property on a few queries: no row outside the result beats the k-th distance
    for b in (0, 511, 1023):
        full = ops.adc_dist(lut_b[b], codes)
        assert int((full < d[b, -1]).sum().item()) <= k - 1
    codes_np = codes.cpu().numpy()
    sel = [0, 1, 500, 1023]
    rd, ri = oracle.adc_search_c(lut_b[sel].cpu().numpy(), codes_np, k)
    assert np.array_equal(d[sel].cpu().numpy(), rd) and np.array_equal(i[sel].cpu().numpy(), ri)


@pytest.mark.parametrize('name,N,M,dsub,B,kind', [
    ('config3', 10_000_000, 16, 8, 1024, 1),    # 10M x 128-d, PQ m=16, L2, batch 1024
    ('config4', 10_000_000, 64, 12, 256, 3),    # 10M x 768-d, PQ m=64, cosine tables, batch 256
])
def test_full_size_properties_config3_config4(ops, oracle, name, N, M, dsub, B, kind):
    """BASELINE configs 3 (one GPU's view: the whole 10M-row table; its 8-way sharding is the merge property
    below) and 4 at FULL size through size-independent properties: ascending (distance, id) order, every
    returned distance equals the gathered ADC distance of that row (adc_gather = space_pq.h PQLookup), the
    one-call entry equals table build + scan, two row shards scanned separately and merged equal the full
    scan (sharding linearity), no unreturned row beats the k-th distance, a few queries equal the CPU oracle."""
    import torch
    from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, scan_plan

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    Ks, k, D = 256, 10, M * dsub
    cb = torch.randn((M, Ks, dsub), generator=g, device=dev)
    q = torch.randn((B, D), generator=g, device=dev)
    if kind == 3:
        q = q / q.norm(dim=1, keepdim=True)
    # codes drawn directly (an encode of 10M x 768 floats would need 30 GB of inputs); every code equally likely
    codes = torch.randint(0, Ks, (N, M), generator=g, device=dev, dtype=torch.uint8)
    skewed = ops.codes_skew(codes)
    assert torch.equal(ops.codes_skew(skewed, inverse=True), codes)  # layout round trip (wrap-coded for M = 64)
    plan = scan_plan(N, M, Ks, 1, B, k)
    lut_t = ops.lut_build(q, cb, kind, LAYOUT_TILED, plan.qi)
    lut_b = ops.lut_build(q, cb, kind, LAYOUT_BMK)
    d, i = ops.adc_scan_topk(skewed, lut_t, B, k, M, Ks, codes_layout=1)
    d1, i1 = ops.pq_search_topk(kind, q, cb, skewed, k, M, Ks, codes_layout=1)
    assert torch.equal(d, d1) and torch.equal(i, i1)
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    ties = d[:, 1:] == d[:, :-1]
    assert bool((i[:, 1:][ties] > i[:, :-1][ties]).all())
    assert bool(((i >= 0) & (i < N)).all())
    assert torch.equal(ops.adc_gather(lut_b, codes, i), d)
    half = N // 2
    parts = [ops.pq_search_topk(kind, q, cb, ops.codes_skew(codes[a:b].contiguous()), k, M, Ks, codes_layout=1,
                                row_base=a, packed=True) for a, b in ((0, half), (half, N))]
    md, mi = ops.topk_merge_packed(torch.stack(parts))
    assert torch.equal(md, d) and torch.equal(mi, i)
    for b in (0, B - 1):
        full = ops.adc_dist(lut_b[b], codes)
        assert int((full < d[b, -1]).sum().item()) <= k - 1
    # EVERY query of the batch against an independent device path: all N distances of the query by the operator-seam kernel
    # (annlite_adc_dist = pq_bindings.pyx:52-80, one thread per row, ascending-m sum from the reference-layout table), then the k
    # smallest (distance, row) keys by torch.topk -- no scan kernel, no filter tables, no lists (math.py:94-120 with the fixed
    # tie-break)
    rows = torch.arange(N, device=dev, dtype=torch.int64)
    chunk = max(1, (1 << 26) // N)  # <= 64M keys (512 MB) per chunk
    n_diff = 0
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        dist = torch.empty((nb, N), dtype=torch.float32, device=dev)
        for j in range(nb):
            ops.adc_dist(lut_b[b0 + j], codes, out=dist[j])
        bits = (dist + 0.0).view(torch.int32)
        bits = bits ^ ((bits >> 31) & 0x7FFFFFFF)  # signed-comparable image of the float order
        top = torch.topk((bits.to(torch.int64) << 32) | rows[None, :], k, dim=1, largest=False, sorted=True).values
        ti = top & 0xFFFFFFFF
        td = torch.gather(dist, 1, ti)
        n_diff += int((~((ti == i[b0:b0 + nb]).all(dim=1) & (td == d[b0:b0 + nb]).all(dim=1))).sum().item())
        del dist, bits, top
    assert n_diff == 0, f'{n_diff} of {B} queries differ from the all-distances + top-k path'
    # ... and against the CPU oracle: every query as well (all host cores; ~3 s at 16 threads)
    rd, ri = oracle.adc_search_c(lut_b.cpu().numpy(), codes.cpu().numpy(), k, threads=oracle.max_threads())
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)


def test_bench_under_torchrun_rccl_gather_path(tmp_path):
    """The driver launches bench.py with torch.distributed.run; exercise that launch mode with one rank and
    the all-gather + merge path forced on (RCCL all_gather_into_tensor + merge_lists_kernel), small shape.
    The JSON line must carry the contract fields and the GPU result must equal the CPU oracle."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, ANNLITE_FORCE_GATHER='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--rows', '300000', '--steps', '3',
           '--warmup', '1', '--recall-queries', '32', '--cpu-queries', '2']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in rec
    assert rec['n_gpus'] == 1 and rec['value'] > 0 and rec['roofline']['frac'] > 0
    assert rec['cpu_baseline']['gpu_matches_cpu_bit_exact'] is True
    assert rec['rerank']['recall_at_10'] >= 0.9


# ------------------------------------------------------------------------------------ config 5: HNSW-over-PQ
@pytest.mark.parametrize('walk', ['gpu', 'host'])
@pytest.mark.parametrize('mname,metric', [('euclidean', 1), ('cosine', 3)])
def test_hnsw_pq_candidates_with_gpu_rerank(ops, mname, metric, walk):
    """HnswPQGpuIndex: graph candidates (host) + distances / top-k on the GPU.  Against the exhaustive scan with
    the same codec: the ids agree for >= 90 % of the top-10, and where an id is returned its distance is the
    exhaustive scan's distance bit for bit (same a2 / PQLookup sum); with rerank, recall vs exact search."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(3)
    N, D, M, B, k = 20000, 64, 8, 64, 10
    A = rs.randn(12, D).astype(np.float32)
    x = (rs.randn(N, 12).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 12).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric(metric), n_init=1)
    codec.seed = 1
    codec.fit(x[:8192], iter=10)
    flat = PQFlatGpuIndex(dim=D, metric=Metric(metric), pq_codec=codec, initial_size=N)
    hn = HnswPQGpuIndex(dim=D, metric=Metric(metric), pq_codec=codec, initial_size=N, ef_search=128, rerank=True, walk=walk)
    ids = list(range(N))
    flat.add_with_ids(x, ids)
    hn.add_with_ids(x[:N // 2], ids[:N // 2])  # two batches: the second inserts into a populated graph
    hn.add_with_ids(x[N // 2:], ids[N // 2:])
    assert hn.size == N
    fd, fi = flat.search_batch(q, limit=k)
    hn.rerank = False
    hd, hi = hn.search_batch(q, limit=k)
    rec = np.mean([len(set(hi[b]) & set(fi[b])) / k for b in range(B)])
    assert rec >= 0.9, rec
    for b in range(B):  # same distance bits for the ids both return
        pos = {int(i): j for j, i in enumerate(fi[b])}
        for j, i in enumerate(hi[b]):
            if int(i) in pos:
                assert hd[b][j] == fd[b][pos[int(i)]]
    # the walk's own output: ascending, and every distance is the PQLookup sum of that row under the L2 tables
    # (bit-equal to the gather kernel), whichever side walked the graph
    from annlite_amd._capi import LAYOUT_BMK, LUT_L2
    qd = hn._pre(torch.from_numpy(q).cuda())
    cid, cd = hn.candidates(qd, 128)
    _, xg = codec.scan_inputs(qd)
    lut_l2 = ops.lut_build(xg, codec.codebooks_dev, LUT_L2, LAYOUT_BMK)
    want = ops.adc_gather(lut_l2, hn._plain_table(N), cid)
    ok = cid >= 0
    assert torch.equal(cd[ok], want[ok]) and bool((cd[:, 1:] >= cd[:, :-1]).all())
    assert bool(((cid[:, 1:] != cid[:, :-1]) | ~ok[:, 1:]).all())  # no node twice
    # reference single-query signature + deletions
    d1, i1 = hn.search(q[0], limit=k)
    assert np.array_equal(i1, hi[0][hi[0] >= 0])
    hn.delete([int(i1[0])])
    d2, i2 = hn.search(q[0], limit=k)
    assert int(i1[0]) not in set(i2.tolist())
    # exact re-rank of the candidates: recall against brute force on the float vectors
    hn.rerank = True
    rd, ri = hn.search_batch(q, limit=k)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    if metric == 3:
        xt, qt = xt / xt.norm(dim=1, keepdim=True), qt / qt.norm(dim=1, keepdim=True)
    truth = torch.cdist(qt, xt).topk(k + 1, largest=False).indices.cpu().numpy()
    rr = np.mean([len(set(ri[b]) & (set(truth[b]) - {int(i1[0])} if b == 0 else set(truth[b][:k]))) / k for b in range(B)])
    assert rr >= 0.9, rr


@pytest.mark.gpu
@pytest.mark.parametrize('mname,metric', [('inner_product', 2), ('cosine', 3)])
def test_hnsw_pq_ip_cosine_pinned_to_the_reference_fixture(ops, oracle, mname, metric):
    """PQ_Space semantics for IP / cosine graphs (include/hnswlib/space_pq.h:15-37, hnsw/index.py:20-48): this build walks a
    graph built with L2 tables for every metric -- a departure.  What is pinned against the REFERENCE's own
    HnswIndex(PQ) run (tests/golden: hnsw_<metric>_i / _d): every distance reported is the metric's own PQLookup sum of that
    row (= the reference's number for every id both return: bit for bit for IP, 1e-5 for cosine -- another numpy build
    normalised the fixture's rows), the result is ranked by it, and against the exhaustive ADC top-k under the metric's tables
    it is at least as complete as the reference's graph was."""
    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    g = load_golden('c2_m16_d128')
    cb = g['codebooks_cos'] if metric == 3 else g['codebooks']
    K, N, B = int(g['K']), int(g['N']), int(g['B'])
    codec = PQCodec(dim=g['D'], n_subvectors=g['M'], n_clusters=g['Ks'], metric=Metric(metric)).set_codebooks(cb)
    hn = HnswPQGpuIndex(dim=g['D'], metric=Metric(metric), pq_codec=codec, initial_size=N, ef_search=128, rerank=False)
    hn.add_with_ids(g['x'], np.arange(N))
    d, i = hn.search_batch(g['queries'], limit=K)
    ref_i, ref_d = g['hnsw_%s_i' % mname], g['hnsw_%s_d' % mname]
    codes = oracle.encode_c(oracle.l2_normalize(g['x']), cb) if metric == 3 else g['codes']
    full_d, full_i = oracle.index_search(g['queries'], cb, codes, metric, N)  # every row's distance, (distance, id) order
    tol = 1e-5 if metric == 3 else 0.0
    n_common = 0
    for b in range(B):
        mine = {int(r): float(x) for r, x in zip(i[b], d[b]) if r >= 0}
        assert len(mine) == K
        row_d = dict(zip(full_i[b].tolist(), full_d[b].tolist()))
        for r, x in mine.items():  # the metric's own PQLookup sum of that row
            assert x == row_d[r], (b, r, x, row_d[r])
        assert list(d[b]) == sorted(d[b])
        for r, x in zip(ref_i[b], ref_d[b]):  # the reference's number for the ids both return
            if int(r) in mine:
                n_common += 1
                assert abs(mine[int(r)] - float(x)) <= tol, (b, int(r), mine[int(r)], float(x))
    assert n_common >= 0.8 * B * K, n_common
    top = full_i[:, :K]
    mine_rec = np.mean([len(set(i[b]) & set(top[b])) / K for b in range(B)])
    ref_rec = np.mean([len(set(ref_i[b]) & set(top[b])) / K for b in range(B)])
    assert mine_rec >= ref_rec - 1e-9 and mine_rec >= 0.9, (mine_rec, ref_rec)


@pytest.mark.parametrize('N,min_overlap,min_recall', [(1_000_000, 0.9, 0.9), (5_000_000, 0.85, 0.88)])
def test_config5_hnsw_pq_oracle_side(ops, oracle, N, min_overlap, min_recall):
    """BASELINE config 5 (HNSW-over-PQ, 128-d, PQ m=16, ef_search=128) at 1M rows and at its STATED size, 5M rows (graph
    construction ~60 s on the box's host cores; measured there: candidate overlap 0.95+, re-rank recall 0.92 -- the
    thresholds leave room for the sample of 256 queries): graph walk on the GPU, checked from the ORACLE side --
      * every distance the walk reports is hnswlib::PQLookup of that row (space_pq.h:15-37): the CPU oracle's
        adc_gather_c on the walk's candidate ids, bit for bit;
      * the top-10 of the graph search (ADC ranking) overlaps the oracle's EXHAUSTIVE ADC top-10 >= 0.9;
      * with the exact re-rank of the 128 candidates recall@10 vs brute force >= 0.9;
      * result conventions of HnswIndex.search: sqrt for EUCLIDEAN (hnsw/index.py:164-165), closest first
        (hnswalg.h:1286-1294 pops the heap into ascending order)."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec
    from annlite_amd._capi import LAYOUT_BMK, LUT_L2

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    D, M, B, k, ef = 128, 16, 256, 10, 128
    A = torch.randn((16, D), generator=g, device=dev)

    def gen(n):
        return (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()

    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    x = gen(N)
    codec.fit(x[:20480], iter=15)
    hn = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=ef, rerank=True)
    for c0 in range(0, N, 250_000):
        hn.add_with_ids(x[c0:c0 + 250_000], torch.arange(c0, c0 + 250_000, device=dev, dtype=torch.int64))
    assert hn.size == N
    q = gen(B)
    # -- the walk's candidates against the oracle's PQLookup
    qd = hn._pre(q)
    cid, cd = hn.candidates(qd, ef)
    codes_np = ops.codes_to_numpy(hn._plain_codes(N))
    lut_np = ops.lut_build(codec.scan_inputs(qd)[1], codec.codebooks_dev, LUT_L2, LAYOUT_BMK).cpu().numpy()
    cid_np, cd_np = cid.cpu().numpy(), cd.cpu().numpy()
    assert (cid_np >= 0).all() and (np.diff(cd_np, axis=1) >= 0).all()
    for b in range(0, B, 4):  # 64 queries x 128 candidates
        assert np.array_equal(cd_np[b], oracle.adc_gather_c(lut_np[b], codes_np, cid_np[b])), b
    # -- ADC ranking of the graph search vs the oracle's exhaustive ADC top-10
    hn.rerank = False
    hd, hi = hn.search_batch(q, limit=k)
    hd, hi = hd.cpu().numpy(), hi.cpu().numpy()
    nq = 64
    od, oi = oracle.adc_search_c(lut_np[:nq], codes_np, k, threads=oracle.max_threads())
    overlap = np.mean([len(set(hi[b]) & set(oi[b])) / k for b in range(nq)])
    assert overlap >= min_overlap, overlap
    assert (np.diff(hd, axis=1) >= 0).all()
    for b in range(nq):  # same ids => same (sqrt of the) oracle distance
        pos = {int(i): j for j, i in enumerate(oi[b])}
        for j, i in enumerate(hi[b]):
            if int(i) in pos:
                assert hd[b][j] == np.sqrt(od[b][pos[int(i)]])
    # -- exact re-rank of the candidates: recall@10 vs brute force
    hn.rerank = True
    rd, ri = hn.search_batch(q, limit=k)
    best = torch.cat([torch.cdist(q, x[c0:c0 + 250_000]) for c0 in range(0, N, 250_000)], dim=1)
    truth = best.topk(k, largest=False).indices.cpu().numpy()
    ri = ri.cpu().numpy()
    rec = np.mean([len(set(ri[b]) & set(truth[b])) / k for b in range(B)])
    assert rec >= min_recall, rec
    assert bool((rd[:, 1:] >= rd[:, :-1]).all())


def test_annlite_facade_with_graph_index(ops, tmp_path):
    """AnnLite(..., graph=True): same API, HnswPQGpuIndex underneath; its matches agree with the exhaustive facade."""
    from annlite_amd import AnnLite
    from annlite_amd.docarray_compat import Document, DocumentArray

    rs = np.random.RandomState(5)
    N, D = 6000, 64
    A = rs.randn(10, D).astype(np.float32)
    x = (rs.randn(N, 10).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(20, 10).astype(np.float32) @ A + 0.05 * rs.randn(20, D).astype(np.float32)).astype(np.float32)
    res = []
    for graph in (False, True):
        ann = AnnLite(D, metric='euclidean', n_subvectors=8, data_path=str(tmp_path / ('g%d' % graph)), graph=graph,
                      ef_search=128)
        ann._pq_codec.seed = 1  # same codebooks for both facades
        ann.train(x[:4096])
        ann.index(DocumentArray([Document(id=str(i), embedding=x[i]) for i in range(N)]))
        docs = DocumentArray([Document(id='q%d' % i, embedding=q[i]) for i in range(len(q))])
        ann.search(docs, limit=10)
        res.append([[m.id for m in d.matches] for d in docs])
    agree = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(*res)])
    assert agree >= 0.9, agree


@pytest.mark.parametrize('mname,metric', [('euclidean', 'EUCLIDEAN'), ('cosine', 'COSINE')])
def test_multi_gpu_index_on_one_device(ops, oracle, mname, metric, tmp_path):
    """``AnnLite(..., devices=[0, 0])`` -- the single-process multi-GPU index with both shards on the one device of the box --
    against the flat index: identical ids AND distances (the packed per-shard results are merged on the raw sums by the
    kernel the multi-process path uses, metric epilogue last), through index / search / delete / filter / dump + reopen."""
    import torch
    from annlite_amd import AnnLite, Metric
    from annlite_amd.core.index.multi_gpu import MultiGpuPQIndex
    from annlite_amd.index import Document, DocumentArray

    rs = np.random.RandomState(12)
    D, M, N, B, k = 64, 16, 20_000, 33, 10
    A = rs.randn(8, D).astype(np.float32)
    x = (rs.randn(N, 8).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    x[5000:5040] = x[17]  # ties across shard blocks (block = 1024 rows)
    q = np.concatenate([x[17:18], (rs.randn(B - 1, 8).astype(np.float32) @ A).astype(np.float32)])
    flat = AnnLite(D, metric=mname, n_subvectors=M, data_path=tmp_path / 'flat')
    flat.train(x[:8192])
    multi = AnnLite(D, metric=mname, n_subvectors=M, data_path=tmp_path / 'multi', devices=[0, 0], shard_block=1024)
    multi._pq_codec.set_codebooks(flat._pq_codec.codebooks)
    assert isinstance(multi.vec_index(0), MultiGpuPQIndex)
    docs = lambda: DocumentArray([Document(id=str(i), embedding=x[i], tags={'price': i % 11}) for i in range(N)])
    flat.index(docs())
    multi.index(docs())
    assert multi.index_size == flat.index_size == N
    for filt in (None, {'price': {'$lt': 3}}):
        fd, fi = flat.search_numpy(q, filter=filt or {}, limit=k)
        md, mi = multi.search_numpy(q, filter=filt or {}, limit=k)
        for b in range(B):
            assert np.array_equal(fi[b], mi[b]) and np.array_equal(fd[b], md[b]), (filt, b)
    gone = [str(int(i)) for i in fi[0][:3]] + ['17', '5001']
    flat.delete(gone)
    multi.delete(gone)
    fd, fi = flat.search_numpy(q, limit=k)
    md, mi = multi.search_numpy(q, limit=k)
    for b in range(B):
        assert np.array_equal(fi[b], mi[b]) and np.array_equal(fd[b], md[b])
    # device tensors in -> device tensors out
    td, ti = multi.vec_index(0).search_batch(ops.to_dev(q), limit=k)
    assert td.is_cuda and ti.shape == (B, k)
    # snapshot + reopen onto the same number of shards
    multi.dump()
    again = AnnLite(D, metric=mname, n_subvectors=M, data_path=tmp_path / 'multi', devices=[0, 0], shard_block=1024)
    ad, ai = again.search_numpy(q, limit=k)
    for b in range(B):
        assert np.array_equal(ai[b], mi[b]) and np.array_equal(ad[b], md[b])
