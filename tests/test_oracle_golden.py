"""Pin the CPU oracle (oracle/pq_oracle.c + .py) against the golden fixtures that
tests/golden/make_golden.py produced by running the compiled reference.  CPU only.

bar: bit-exact for every fp32 table / distance / code / id (np.array_equal), except encode, where
scipy's BLAS expansion is "parity unpinned" upstream and a mismatch is excused only on a near-tie.
"""
import numpy as np
import pytest


def test_lut_l2_batch_bit_exact(oracle, golden):
    g = golden
    got = oracle.batch_precompute_adc_table_c(g['queries'], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(got, g['lut_l2_batch'])
    got_np = oracle.batch_precompute_adc_table_numpy(g['queries'][:2], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(got_np, g['lut_l2_batch'][:2])


def test_lut_l2_single_bit_exact(oracle, golden):
    g = golden
    got = oracle.precompute_adc_table_c(g['queries'][0], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(got, g['lut_l2_single'])
    # reference's own check (tests/test_pq_index.py:30-49): batch row == single-query table
    assert np.array_equal(g['lut_l2_batch'][0], g['lut_l2_single'])


def test_lut_ip_batch_bit_exact(oracle, golden):
    g = golden
    got = oracle.batch_precompute_adc_table_ip_c(g['queries'], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(got, g['lut_ip_batch'])
    got_np = oracle.batch_precompute_adc_table_ip_numpy(g['queries'][:2], g['dsub'], g['Ks'], g['codebooks'])
    assert np.array_equal(got_np, g['lut_ip_batch'][:2])


def test_lut_matches_reference_numpy_check(oracle, golden):
    """The reference's own known-answer test, tests/test_pq_bind.py:36-59 (decimal=5)."""
    g = golden
    q = g['queries'][0]
    ref = np.empty((g['M'], g['Ks']), dtype=np.float32)
    for m in range(g['M']):
        ref[m] = np.linalg.norm(g['codebooks'][m] - q[m * g['dsub']:(m + 1) * g['dsub']], axis=1) ** 2
    got = oracle.precompute_adc_table_c(q, g['dsub'], g['Ks'], g['codebooks'])
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('mname,metric', [('euclidean', 1), ('inner_product', 2), ('cosine', 3)])
def test_get_dist_mat_bit_exact(oracle, golden, mname, metric):
    g = golden
    cb = g['codebooks_cos'] if metric == 3 else g['codebooks']
    got = oracle.get_dist_mat_c(g['queries'], cb, metric)
    assert got.dtype == np.float32 and got.flags['C_CONTIGUOUS']
    assert np.array_equal(got, g['dist_mat_' + mname])


def test_l2_normalize(oracle, golden):
    assert np.array_equal(oracle.l2_normalize(golden['queries']), golden['l2norm_q'])


def test_adc_scan_bit_exact(oracle, golden):
    g = golden
    for b in range(g['B']):
        got = oracle.dist_pqcodes_to_codebooks_c(g['lut_l2_batch'][b], g['codes'])
        assert np.array_equal(got, g['adist'][b])
    got_np = oracle.dist_pqcodes_to_codebooks_numpy(g['lut_l2_batch'][0], g['codes'])
    assert np.array_equal(got_np, g['adist'][0])


def test_decode_bit_exact(oracle, golden):
    g = golden
    assert np.array_equal(oracle.decode_numpy(g['codes'][:64], g['codebooks']), g['decoded'])
    if g['codes'].dtype == np.uint8:
        assert np.array_equal(oracle.decode_c(g['codes'][:64], g['codebooks']), g['decoded'])


def _check_encode(oracle, x, codebooks, want):
    got = oracle.encode_c(x, codebooks)
    assert got.dtype == want.dtype
    bad = np.argwhere(got != want)
    if len(bad):
        best, second = oracle.encode_gap(x, codebooks)
        for n, m in bad:
            gap = (second[n, m] - best[n, m]) / max(second[n, m], 1e-30)
            assert gap < 1e-5, 'encode mismatch at (%d,%d) with top-2 gap %.3g' % (n, m, gap)
    return len(bad)


def test_encode_matches_scipy_fixture(oracle, golden):
    g = golden
    _check_encode(oracle, g['x'], g['codebooks'], g['codes'])
    # and the literal scipy call reproduces the fixture exactly on this image (scipy 1.15.3)
    assert np.array_equal(oracle.encode_scipy(g['x'], g['codebooks']), g['codes'])


def test_encode_cosine_fixture(oracle, golden):
    g = golden
    xn = oracle.l2_normalize(g['x']).astype(np.float32)
    _check_encode(oracle, xn, g['codebooks_cos'], g['codes_cos'])


def test_topk_values_match_reference_topk(oracle, golden):
    g = golden
    d, i = oracle.top_k_c(g['adist'][0], g['K'])
    assert np.array_equal(d.astype(np.float64), g['topk_d'])
    d2, i2 = oracle.top_k_numpy(g['adist'][0], g['K'])
    assert np.array_equal(d, d2) and np.array_equal(i, i2)


def test_pqindex_search_semantics(oracle, golden):
    """PQIndex.search (pq_index.py:29-56): scans ALL capacity rows (zero rows included), no sqrt,
    distances come back float64-typed.  ids may differ from the fixed tie-break only on exact ties."""
    g = golden
    if g['codes'].dtype != np.uint8:
        pytest.skip('C search path is uint8-only; u16 covered by test_adc_scan_bit_exact')
    cap = int(g['pqindex_capacity'][0])
    table = np.zeros((cap, g['M']), dtype=np.uint8)
    table[:g['N']] = g['codes']
    d, i = oracle.adc_search_c(g['lut_l2_batch'], table, g['K'])
    assert np.array_equal(d.astype(np.float64), g['pqindex_d'])
    for b in range(g['B']):
        same = i[b] == g['pqindex_i'][b]
        if not same.all():  # only legal where the distance is tied
            dd = g['pqindex_d'][b]
            for j in np.where(~same)[0]:
                assert (dd == dd[j]).sum() > 1
    dn, in_ = oracle.adc_search_numpy(g['lut_l2_batch'][:2], table, g['K'])
    assert np.array_equal(dn, d[:2]) and np.array_equal(in_, i[:2])


@pytest.mark.parametrize('mname,metric', [('euclidean', 1), ('inner_product', 2), ('cosine', 3)])
def test_hnsw_pq_distances_equal_flat_adc(oracle, golden, mname, metric):
    """HnswIndex(pq_codec).search (hnsw/index.py:139-167) returns, for each id it found, exactly
    the flat-ADC distance of that row (space_pq.h:15-37 == pyx:30-47), sqrt'ed for EUCLIDEAN.
    The exhaustive scan must therefore (a) reproduce those distances bit-for-bit on those ids and
    (b) never be worse than the graph walk."""
    g = golden
    key = 'hnsw_%s_d' % mname
    if key not in g:
        pytest.skip('fixture built without hnsw_bind')
    codes = g['codes_cos'] if metric == 3 else g['codes']
    cb = g['codebooks_cos'] if metric == 3 else g['codebooks']
    x = g['queries']
    if metric == 3:
        x = oracle.l2_normalize(x).astype(np.float32)  # hnsw/index.py:28-29
    lut = oracle.get_dist_mat_c(x, cb, metric)         # normalises again for cosine (pq.py:309-310)
    for b in range(g['B']):
        ids = g['hnsw_%s_i' % mname][b]
        flat = oracle.dist_pqcodes_to_codebooks_c(lut[b], codes)
        want = flat[ids]
        if metric == 1:
            want = np.sqrt(want)
        assert np.array_equal(want.astype(np.float32), g[key][b])
        d, i = oracle.top_k_c(flat, g['K'])
        if metric == 1:
            d = np.sqrt(d)
        assert (d <= g[key][b]).all()


def test_c_and_numpy_agree_random(oracle):
    rs = np.random.RandomState(7)
    M, dsub, Ks = 4, 5, 19  # odd sizes
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    q = rs.randn(3, M * dsub).astype(np.float32)
    assert np.array_equal(oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb),
                          oracle.batch_precompute_adc_table_numpy(q, dsub, Ks, cb))
    assert np.array_equal(oracle.batch_precompute_adc_table_ip_c(q, dsub, Ks, cb),
                          oracle.batch_precompute_adc_table_ip_numpy(q, dsub, Ks, cb))
    codes = rs.randint(0, Ks, size=(57, M)).astype(np.uint8)
    lut = oracle.get_dist_mat_c(q, cb, 2)
    d, i = oracle.adc_search_c(lut, codes, 60)  # k > N pads with (+inf, -1)
    dn, in_ = oracle.adc_search_numpy(lut, codes, 60)
    assert np.array_equal(d, dn) and np.array_equal(i, in_)
    assert np.isinf(d[:, 57:]).all() and (i[:, 57:] == -1).all()


def test_adc_search_wide_codes(oracle):
    """n_clusters > 256 means uint16 codes (pq.py:56-60): the C search helper must not narrow them (it once cast every code
    table to uint8 -- a checker that was wrong exactly where the reference's own PQ tests run, Ks = 512 / 768)."""
    import numpy as np

    rs = np.random.RandomState(0)
    lut = rs.rand(3, 8, 768).astype(np.float32)
    codes = rs.randint(0, 768, size=(5000, 8)).astype(np.uint16)
    codes[7] = codes[9]  # a tie
    d, i = oracle.adc_search_c(lut, codes, 10)
    d2, i2 = oracle.adc_search_numpy(lut, codes, 10)
    assert np.array_equal(d, d2) and np.array_equal(i, i2)
    d3, i3 = oracle.adc_search_c(lut, codes.view(np.int16), 10)  # (torch hands uint16 tables over as int16)
    assert np.array_equal(d, d3) and np.array_equal(i, i3)
