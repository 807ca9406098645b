"""Round 6: annlite_pq_search_candidates -- the re-rank stage's candidate generator with the tables built inside the call and, for
M = 16 / L2, ONE table-wide first bound from the plain search's preparation launch instead of per-slice seeds.

Reference semantics of the stage it feeds: FlatIndex.search (annlite/core/index/flat_index.py:15-39) re-scores candidates exactly;
the candidates are PQ / ADC rankings (pq_bindings.pyx:52-80 sums).  What is pinned: every distance is the oracle's ascending-m sum
of that row; every slice's list is a PREFIX of that slice's own complete top-k (the rows at or below the table-wide bound); the
union holds the table's exact top-k under the fixed tie-break; shapes without the fused path return the old call's lists."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


def _setup(ops, rs, N, D, M, B, structured=True):
    from annlite_amd._capi import CODES_SKEWED

    cb = rs.randn(M, 256, D // M).astype(np.float32)
    if structured:
        A = rs.randn(12, D).astype(np.float32)
        x = (rs.randn(N, 12).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
        q = (rs.randn(B, 12).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
        cb = x[rs.choice(N, M * 256, replace=True)].reshape(M, 256, D)[:, :, :D // M].copy()
        for m in range(M):
            cb[m] = x[rs.choice(N, 256, replace=False), m * (D // M):(m + 1) * (D // M)]
    else:
        x = rs.rand(N, D).astype(np.float32)
        q = rs.rand(B, D).astype(np.float32)
    cb_d = ops.to_dev(cb)
    codes = ops.pq_encode(ops.to_dev(x), cb_d)
    sk = ops.codes_skew(codes)
    return cb, cb_d, q, codes, sk, CODES_SKEWED


@pytest.mark.parametrize('k', [16, 10, 1])
@pytest.mark.parametrize('layout', ['skewed', 'plain'])
def test_fused_candidates_are_prefixes_of_the_slices_own_lists(ops, oracle, monkeypatch, k, layout):
    import torch
    from annlite_amd._capi import CODES_PLAIN, LAYOUT_BMK, LAYOUT_TILED, LUT_L2, scan_plan

    rs = np.random.RandomState(100 + k)
    N, D, M, B = 300_000, 128, 16, 70
    cb, cb_d, q, codes, sk, SK = _setup(ops, rs, N, D, M, B)
    table, lay = (sk, SK) if layout == 'skewed' else (codes, CODES_PLAIN)
    valid = rs.rand(N) < 0.95
    bits = np.zeros(((N + 31) // 32 + 2) * 32, bool)
    bits[:N] = valid
    vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    qd = ops.to_dev(q)
    plan = scan_plan(N, M, 256, 1, B, k)
    nd, ni = ops.pq_search_candidates(LUT_L2, qd, cb_d, table, k, M, 256, valid_bits=vb, codes_layout=lay)
    # the slices' complete lists: the same call with the table-wide seed switched off, and the round-5 entry point
    monkeypatch.setenv('ANNLITE_NO_CAND_SEED', '1')
    od, oi = ops.pq_search_candidates(LUT_L2, qd, cb_d, table, k, M, 256, valid_bits=vb, codes_layout=lay)
    monkeypatch.delenv('ANNLITE_NO_CAND_SEED')
    lut = ops.lut_build(qd, cb_d, LUT_L2, LAYOUT_TILED if plan.fast else LAYOUT_BMK, plan.qi)
    rd, ri = ops.adc_scan_candidates(table, lut, B, k, M, 256, valid_bits=vb, codes_layout=lay)
    torch.cuda.synchronize()
    nd, ni, od, oi, rd, ri = (t.cpu().numpy() for t in (nd, ni, od, oi, rd, ri))
    assert np.array_equal(oi, ri) and np.array_equal(od.view(np.uint32), rd.view(np.uint32))
    S = plan.n_slices
    assert ni.shape == (B, S * k)
    luts = np.asarray(oracle.batch_precompute_adc_table_c(q, D // M, 256, cb))
    codes_np = codes.cpu().numpy()
    shorter = 0
    for b in range(B):
        for s in range(S):
            new, old = ni[b, s * k:(s + 1) * k], oi[b, s * k:(s + 1) * k]
            n = int((new >= 0).sum())
            assert (new[n:] == -1).all() and np.array_equal(new[:n], old[:n]), (b, s)
            assert np.array_equal(nd[b, s * k:s * k + n].view(np.uint32), od[b, s * k:s * k + n].view(np.uint32))
            shorter += n < int((old >= 0).sum())
        got = ni[b][ni[b] >= 0]
        assert valid[got].all() and len(np.unique(got)) == len(got)
        assert np.array_equal(nd[b][ni[b] >= 0], oracle.adc_gather_c(luts[b], codes_np, got))
        # the table's exact top-k (deleted rows out) is in the union
        full = oracle.dist_pqcodes_to_codebooks_c(luts[b], codes_np)
        full[~valid] = np.inf
        top = np.lexsort((np.arange(N), full))[:k]
        assert np.isin(top, got).all(), b
    assert shorter > 0  # (the bound does cut: slices away from a query return fewer rows)


def test_shapes_without_the_fused_path_return_the_old_lists(ops):
    import torch
    from annlite_amd._capi import CODES_PLAIN, LAYOUT_BMK, LAYOUT_TILED, LUT_IPDIST, LUT_L2, scan_plan

    rs = np.random.RandomState(5)
    for M, D, kind, k in ((8, 64, LUT_L2, 10), (16, 128, LUT_IPDIST, 16), (32, 128, LUT_L2, 8), (16, 128, LUT_L2, 40)):
        N, B = 100_000, 33
        cb, cb_d, q, codes, sk, SK = _setup(ops, rs, N, D, M, B, structured=False)
        qd = ops.to_dev(q)
        plan = scan_plan(N, M, 256, 1, B, k)
        nd, ni = ops.pq_search_candidates(kind, qd, cb_d, sk, k, M, 256, codes_layout=SK)
        lut = ops.lut_build(qd, cb_d, kind, LAYOUT_TILED if plan.fast else LAYOUT_BMK, plan.qi)
        rd, ri = ops.adc_scan_candidates(sk, lut, B, k, M, 256, codes_layout=SK)
        torch.cuda.synchronize()
        assert np.array_equal(ni.cpu().numpy(), ri.cpu().numpy()), (M, kind, k)
        assert np.array_equal(nd.cpu().numpy().view(np.uint32), rd.cpu().numpy().view(np.uint32))
