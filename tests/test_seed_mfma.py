"""Round 6: the seed bound's rows are NOMINATED by one bf16 MFMA contraction (seed_mfma.hip: per query the best row, by approximate
ADC distance, of each of 512 disjoint groups of seed rows) and the preparation launch takes the k-th smallest EXACT sum of the
nominees -- any k distinct valid rows give a valid bound, so the results must not move.

What is checked here: (1) the nomination launch itself -- every nominee is a valid row of ITS group and (nearly) the group's best
by the exact ascending-m fp32 sum (a wrong operand layout nominates random rows); (2) the search with the nominated seed equals the
search with the exact seed scan (the default; ANNLITE_MFMA_SEED=1 opts in) and the CPU oracle, bit for bit, on both code layouts, with deleted rows,
ragged batches, k on the 16- and the 64-key lists, and a table whose seed rows must be rounded up to the group grid
(reference: annlite/core/codec/pq.py:316-322 tables, pq_bindings.pyx:30-47 sums, math.py:94-120 selection)."""
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


def _table(N, seed, Ks=256):
    """Low-rank 128-d vectors through code books drawn from the same latent model (M = 16, 8-float sub-vectors)."""
    rs = np.random.RandomState(seed)
    M, dsub, D = 16, 8, 128
    A = rs.randn(12, D).astype(np.float32)
    x = (rs.randn(N, 12).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    cb = (rs.randn(Ks, 12).astype(np.float32) @ A + 0.02 * rs.randn(Ks, D).astype(np.float32)).reshape(Ks, M, dsub).transpose(1, 0, 2).copy()
    return cb, x, A, rs


def _bits(valid_bool):
    n = len(valid_bool)
    assert n % 32 == 0
    w = valid_bool.reshape(-1, 32).astype(np.uint64)
    return (w << np.arange(32, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint32).view(np.int32)


def _seed_rows(S, N, clog=3):
    """Table row of seed index s (scan_prep.hip / seed_mfma.hip: 64-row blocks in runs of 2^clog blocks spread evenly over the table)."""
    n_blocks = S >> 6
    cmask = (1 << clog) - 1
    run_step = ((N >> 6) // ((n_blocks + cmask) >> clog)) << 6
    run_step = max(run_step, 64 << clog)
    s = np.arange(S, dtype=np.int64)
    b = s >> 6
    return (b >> clog) * run_step + ((b & cmask) << 6) + (s & 63)


def _group_of(S):
    """Group index (column of the nominee array) of seed index s: slice, row quarter, lane half of the MFMA tile, half of the quarter."""
    s = np.arange(S, dtype=np.int64)
    RW = S // 32
    nt = RW // 128  # 32-row tiles per wave
    sl, w = s // RW, s % RW
    h4, r = w // (RW // 4), w % (RW // 4)
    t, i = r // 32, r % 32
    kh = (i >> 2) & 1  # rows {0-3, 8-11, ...} of a tile sit in lanes 0..31, rows {4-7, 12-15, ...} in lanes 32..63
    g = (t >= nt // 2).astype(np.int64)
    return sl * 16 + h4 * 4 + kh * 2 + g


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('N,S,B', [(200_000, 32768, 200), (70_000, 8192, 64), (300_000, 65536, 129)])
def test_nominees_are_their_groups_best_rows(ops, N, S, B, layout):
    import torch
    from annlite_amd._capi import LAYOUT_BMK, LUT_L2

    cb, x, A, rs = _table(N, 5 + B)
    cb_d = ops.to_dev(cb)
    codes = ops.pq_encode(ops.to_dev(x), cb_d)
    q = (rs.randn(B, 12).astype(np.float32) @ A + 0.05 * rs.randn(B, 128).astype(np.float32)).astype(np.float32)
    q_d = ops.to_dev(q)
    valid = np.ones(((N + 31) // 32 + 2) * 32, dtype=bool)
    valid[N:] = False
    valid[rs.choice(N, N // 10, replace=False)] = False
    rows = _seed_rows(S, N)
    assert rows.max() < N and len(np.unique(rows)) == S
    grp = _group_of(S)
    assert grp.max() == 511 and np.all(np.bincount(grp) == S // 512)
    valid[rows[grp == 7]] = False  # one group without a valid row: its nominee is "none"
    stored = ops.codes_skew(codes) if layout == 1 else codes
    cand = ops.debug_seed_candidates(q_d, cb_d, stored, S, valid_bits=ops.to_dev(_bits(valid)), n_rows=N, codes_layout=layout)
    assert cand is not None and cand.shape == (B, 512)
    cand = cand.cpu().numpy()
    # exact ADC sums of all seed rows (reference arithmetic through the operator-seam kernels)
    lut = ops.lut_build(q_d, cb_d, LUT_L2, LAYOUT_BMK)  # [B, M, Ks]
    rows_d = torch.from_numpy(rows).cuda()
    dist = torch.stack([ops.adc_dist(lut[b], codes)[rows_d] for b in range(B)]).cpu().numpy()  # [B, S]
    dist = np.where(valid[rows][None, :], dist, np.inf)
    order = np.argsort(grp, kind='stable')
    per = S // 512
    dg = dist[:, order].reshape(B, 512, per)
    rg = rows[order].reshape(512, per)
    best = dg.min(axis=2)  # [B, 512]
    assert np.all(cand[:, 7] == -1) and np.all(np.isinf(best[:, 7]))
    live = np.ones(512, dtype=bool)
    live[7] = False
    c = cand[:, live]
    assert np.all(c >= 0)
    # every nominee is a VALID row of ITS group
    in_group = (rg[None, live, :] == c[:, :, None])
    assert in_group.any(axis=2).all()
    assert valid[c].all()
    # ... and (nearly) the group's best: the exact sum of the nominee against the group's exact minimum
    dn = np.take_along_axis(dg[:, live, :], in_group.argmax(axis=2)[:, :, None], axis=2)[:, :, 0]
    excess = dn - best[:, live]
    scale = float(np.median(dist[np.isfinite(dist)]))
    hit = float((excess == 0).mean())
    assert hit >= 0.6, hit  # (bf16 operands: the exact argmin in most groups ...)
    assert float((excess <= 0.02 * scale).mean()) >= 0.995, (float(np.quantile(excess, 0.999)), scale)  # ... a near-tie in the rest
    assert float(excess.max()) <= 0.25 * scale, (float(excess.max()), scale)
    # distinct nominees per query (the groups are disjoint)
    for b in range(0, B, 17):
        assert len(np.unique(c[b])) == c.shape[1]


def test_not_applicable_shapes(ops):
    import torch

    rs = np.random.RandomState(0)
    q = ops.to_dev(rs.randn(70, 128).astype(np.float32))
    cb = ops.to_dev(rs.randn(16, 256, 8).astype(np.float32))
    codes = torch.randint(0, 256, (20_000, 16), dtype=torch.uint8, device='cuda')
    assert ops.debug_seed_candidates(q, cb, codes, 8192) is not None
    assert ops.debug_seed_candidates(q, cb, codes, 8192 + 1024) is None      # not on the group grid
    assert ops.debug_seed_candidates(q, cb, codes, 32768) is None            # more seed rows than the table has
    cb8 = ops.to_dev(rs.randn(8, 256, 16).astype(np.float32))
    codes8 = torch.randint(0, 256, (20_000, 8), dtype=torch.uint8, device='cuda')
    assert ops.debug_seed_candidates(q, cb8, codes8, 8192) is None           # M = 8


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('N,B,k', [(300_001, 128, 10), (1_000_000, 257, 10), (70_000, 64, 16), (600_000, 100, 50), (9_000, 90, 10)])
def test_search_with_nominated_seed_equals_exact_seed_and_oracle(ops, oracle, monkeypatch, N, B, k, layout):
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    cb, x, A, rs = _table(N, 40 + k)
    cb_d = ops.to_dev(cb)
    codes = ops.pq_encode(ops.to_dev(x), cb_d)
    q = (rs.randn(B, 12).astype(np.float32) @ A + 0.05 * rs.randn(B, 128).astype(np.float32)).astype(np.float32)
    q_d = ops.to_dev(q)
    valid = np.ones(((N + 31) // 32 + 2) * 32, dtype=bool)
    valid[N:] = False
    valid[rs.choice(N, N // 7, replace=False)] = False
    vb = ops.to_dev(_bits(valid))
    stored = ops.codes_skew(codes) if layout == 1 else codes
    kw = dict(valid_bits=vb, n_rows=N, codes_layout=layout)
    out = {}
    for name, env in (('mfma', '1'), ('exact', None)):
        if env:
            monkeypatch.setenv('ANNLITE_MFMA_SEED', env)
        else:
            monkeypatch.delenv('ANNLITE_MFMA_SEED', raising=False)
        st = _capi.ScanState()
        for _ in range(3):  # (guarded first call, then the settled byte-table kernel)
            d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, stored, k, 16, 256, state=st, **kw)
        torch.cuda.synchronize()
        out[name] = (d.cpu().numpy(), i.cpu().numpy())
    assert np.array_equal(out['mfma'][1], out['exact'][1])
    assert np.array_equal(out['mfma'][0].view(np.uint32), out['exact'][0].view(np.uint32))
    lut = oracle.batch_precompute_adc_table_c(q, 8, 256, cb)
    live = np.nonzero(valid[:N])[0]
    rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy()[live], k, threads=oracle.max_threads())
    assert np.array_equal(out['mfma'][1], live[ri]) and np.array_equal(out['mfma'][0], rd)


def test_preparation_launch_is_shorter_with_nominees(ops, monkeypatch):
    """The point of the exercise: the seed phase of the preparation launch (stamps of its first workgroup) shrinks from a scan of
    32768 rows x 4 queries to 2048 nominee rows -- at least 3x, with a wide margin for a busy box."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    N, B, k = 1_100_000, 1024, 10
    cb, x, A, rs = _table(N, 77)
    cb_d = ops.to_dev(cb)
    codes = ops.codes_skew(ops.pq_encode(ops.to_dev(x), cb_d))
    q_d = ops.to_dev((rs.randn(B, 12).astype(np.float32) @ A).astype(np.float32))
    seed_us = {}
    monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '2')
    for name, env in (('mfma', '1'), ('exact', None)):
        if env:
            monkeypatch.setenv('ANNLITE_MFMA_SEED', env)
        else:
            monkeypatch.delenv('ANNLITE_MFMA_SEED', raising=False)
        st = _capi.ScanState()
        for _ in range(4):
            ops.pq_search_topk(LUT_L2, q_d, cb_d, codes, k, 16, 256, codes_layout=1, state=st)
        torch.cuda.synchronize()
        seed_us[name] = _capi.debug_prep_timeline()['first']['seed_us']
    assert seed_us['mfma'] * 3.0 <= seed_us['exact'], seed_us
