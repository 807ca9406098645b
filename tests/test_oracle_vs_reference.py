"""Live pinning of the oracle against the REAL reference, run only where /root/reference and
oracle/_ref exist (the build container).  Skipped on the GPU box -- parity there rests on the
golden fixtures.  Fresh seeds and larger sizes than the fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference tree / oracle/_ref not present')


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
