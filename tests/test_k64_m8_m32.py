"""16 < k <= 64 on the byte-table kernels of M = 8 (uint8 and uint16 codes) and M = 32 (round 6): the shapes 852 / 850 / 851 / 3250
with 64-key lists (launch ids 864 / 8650 / 8651 / 3264, scan_q8.hip) -- what round 5 built for M = 16.  The reference's own PQ test
searches ``topk = 50`` at ``n_subvectors = 8`` with 256, 512 and 768 clusters (tests/test_pq_index.py:78-135); until this round
those searches ran the u16 tables.  ``ANNLITE_SCAN_VARIANT=50`` pins the kernel under test (the library's own choice needs a table
of >= 65536 rows and is guarded like k <= 16).  Bit-exact against the CPU oracle (pq_bindings.pyx:30-47 sums, math.py:94-120
selection) and against the u16-table kernels: random / tied / deleted / short tables, both layouts, structured data through
``annlite_pq_search_topk``, forced epochs and rebuilds, the guarded give-up."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs an AMD GPU')]

# (M, bytes per code, Ks, queries per tile of the byte-table plan)
KERNELS = [(8, 1, 256, 32), (8, 1, 100, 32), (32, 1, 256, 16), (8, 2, 512, 32), (8, 2, 300, 32), (8, 2, 768, 16), (8, 2, 1024, 16)]


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


def _bits(valid):
    bits = np.zeros(((len(valid) + 31) // 32 + 2) * 32, dtype=bool)
    bits[:len(valid)] = valid
    return np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1)


def _scan(ops, monkeypatch, codes, lut, k, layout, qt, valid=None, row_base=0, variant='50'):
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import scan_plan

    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
    B, M, Ks = lut.shape
    cb = codes.dtype.itemsize
    plan = scan_plan(codes.shape[0], M, Ks, cb, B, k)
    if variant == '50':
        assert plan.fast and plan.qt == qt, (plan.fast, plan.qt)
    lut_d = ops.lut_retile(ops.to_dev(lut), plan.qi)
    codes_d = ops.to_dev(codes)
    if layout == 1:
        codes_d = ops.codes_skew(codes_d)
    vb = ops.to_dev(_bits(valid)) if valid is not None else None
    monkeypatch.setenv('ANNLITE_DEBUG_COUNTERS', '2')
    d, i = ops.adc_scan_topk(codes_d, lut_d, B, k, M, Ks, valid_bits=vb, row_base=row_base, codes_layout=layout)
    torch.cuda.synchronize()
    items = _capi.debug_timeline()['items'] if codes.shape[0] > 0 else 1
    monkeypatch.delenv('ANNLITE_DEBUG_COUNTERS')
    if variant == '50':
        assert items > 0, 'the byte-table kernel did not run'
    return d.cpu().numpy(), i.cpu().numpy()


def _oracle(oracle, lut, codes, k, valid=None, row_base=0):
    if valid is None:
        return oracle.adc_search_c(lut, codes, k, id_base=row_base)
    idx = np.where(valid)[0]
    B = lut.shape[0]
    if not len(idx):
        return np.full((B, k), np.inf, np.float32), np.full((B, k), -1, np.int64)
    rd, ri = oracle.adc_search_c(lut, codes[idx], k)
    return rd, np.where(ri >= 0, idx[np.clip(ri, 0, len(idx) - 1)] + row_base, -1)


SHAPES = [  # N, B, k
    (70_000, 20, 50), (130_000, 33, 64), (66_000, 5, 17), (300_000, 9, 50), (63, 9, 64), (1, 3, 20), (5000, 37, 50), (4097, 16, 63),
]


@pytest.mark.parametrize('N,B,k', SHAPES)
@pytest.mark.parametrize('M,cb,Ks,qt', KERNELS)
def test_random_shapes_equal_the_oracle(ops, oracle, monkeypatch, M, cb, Ks, qt, N, B, k):
    rs = np.random.RandomState(Ks * 31 + N + k + M)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut[B // 2] -= 0.5  # negative entries (inner-product style tables)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8 if cb == 1 else np.uint16)
    for layout in ((0, 1) if cb == 1 else (0,)):  # (uint16 codes: PLAIN rows only)
        d, i = _scan(ops, monkeypatch, codes, lut, k, layout, qt, row_base=1000)
        rd, ri = _oracle(oracle, lut, codes, k, row_base=1000)
        assert np.array_equal(d, rd), (M, cb, Ks, N, B, k, layout)
        assert np.array_equal(i, ri), (M, cb, Ks, N, B, k, layout)


@pytest.mark.parametrize('M,cb,Ks,qt', [(8, 1, 256, 32), (32, 1, 256, 16), (8, 2, 512, 32), (8, 2, 768, 16)])
def test_ties_delete_marks_and_short_tables(ops, oracle, monkeypatch, M, cb, Ks, qt):
    rs = np.random.RandomState(12 + M + Ks)
    N, B, k = 90_000, 21, 50
    dt = np.uint8 if cb == 1 else np.uint16
    base = rs.randint(0, Ks, size=(64, M)).astype(dt)
    codes = base[rs.randint(0, 64, size=N)]  # every row has ~1400 exact duplicates: the 50 best all tie
    lut = rs.rand(B, M, Ks).astype(np.float32)
    layouts = (0, 1) if cb == 1 else (0,)
    for layout in layouts:
        d, i = _scan(ops, monkeypatch, codes, lut, k, layout, qt)
        rd, ri = _oracle(oracle, lut, codes, k)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
    valid = rs.rand(N) < 0.3
    d, i = _scan(ops, monkeypatch, codes, lut, k, layouts[-1], qt, valid=valid)
    rd, ri = _oracle(oracle, lut, codes, k, valid=valid)
    assert np.array_equal(d, rd) and np.array_equal(i, ri)
    valid2 = np.zeros(N, bool)
    valid2[[5, 77, 80_000]] = True
    d, i = _scan(ops, monkeypatch, codes, lut, k, 0, qt, valid=valid2)
    assert (i[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all()
    assert (np.sort(i[:, :3], axis=1) == np.array([5, 77, 80_000])).all()


def _structured(ops, N, B, M, dsub, Ks, seed):
    import torch
    from annlite_amd import Metric, PQCodec

    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    D = M * dsub
    A = torch.randn((16, D), generator=g, device=dev)

    def gen(n):
        return (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()

    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.fit(gen(max(20480, 40 * Ks)), iter=8)
    cb = codec.codebooks_dev
    parts = []
    for c0 in range(0, N, 500_000):
        parts.append(ops.pq_encode(gen(min(500_000, N - c0)), cb))
    return cb, torch.cat(parts), gen(B)


@pytest.mark.parametrize('M,dsub,Ks,N,B,k', [(8, 8, 256, 300_000, 100, 50), (8, 16, 256, 1_000_000, 300, 64), (32, 4, 256, 300_000, 70, 50),
                                             (32, 4, 256, 1_000_000, 130, 20), (8, 8, 512, 300_000, 100, 50), (8, 8, 768, 400_000, 40, 33)])
def test_structured_data_equals_the_u16_table_kernel_and_the_oracle(ops, oracle, monkeypatch, M, dsub, Ks, N, B, k):
    """``annlite_pq_search_topk`` (tables built by the call, seed bound for THIS k, shared bounds incl. the weighted bound of the
    sibling slices, merge of the row slices): the byte-table plan returns the bits of the u16-table plan on the same inputs -- all
    queries --, and of the oracle; deleted rows; two calls on one workspace; and the library's OWN choice (no variant pinned: guarded
    first call, then the settled kernel) says byte tables for this data."""
    import torch
    from annlite_amd import _capi
    from annlite_amd._capi import LUT_L2

    cb, codes, q = _structured(ops, N, B, M, dsub, Ks, seed=N % 1000 + k + M)
    wide = codes.dtype != torch.uint8
    rs = np.random.RandomState(k)
    valid = np.ones(N, bool)
    valid[rs.choice(N, N // 20, replace=False)] = False
    vb = ops.to_dev(_bits(valid))
    out = {}
    for layout in ((0,) if wide else (0, 1)):
        cd = ops.codes_skew(codes) if layout == 1 else codes
        for variant in ('50', '31', None):
            if variant is None:
                monkeypatch.delenv('ANNLITE_SCAN_VARIANT', raising=False)
            else:
                monkeypatch.setenv('ANNLITE_SCAN_VARIANT', variant)
            st = _capi.ScanState() if variant is None else None
            ws = ops.ScanWorkspace()
            for rep in range(4 if variant is None else 2):
                d, i = ops.pq_search_topk(LUT_L2, q, cb, cd, k, M, Ks, valid_bits=vb, codes_layout=layout, workspace=ws, state=st)
                if variant is None:
                    torch.cuda.synchronize()  # (the next call reads what this launch left in the state's host-mapped block)
            torch.cuda.synchronize()
            out[(layout, variant)] = (d.cpu().numpy(), i.cpu().numpy())
            if variant is None:
                assert st.info()[0] == 1, st.info()  # (settled on the byte tables)
    ref = out[(0, '31')]
    for key, (d, i) in out.items():
        assert np.array_equal(i, ref[1]), key
        assert np.array_equal(d.view(np.uint32), ref[0].view(np.uint32)), key
    lut = oracle.batch_precompute_adc_table_c(q.cpu().numpy(), dsub, Ks, cb.cpu().numpy())
    rd, ri = _oracle(oracle, lut, ops.codes_to_numpy(codes), k, valid=valid)
    assert np.array_equal(ref[1], ri) and np.array_equal(ref[0], rd)


@pytest.mark.parametrize('M,cb,Ks,qt', [(8, 1, 256, 32), (32, 1, 256, 16), (8, 2, 512, 32)])
def test_forced_epochs_rebuilds_and_the_give_up_path(ops, oracle, monkeypatch, M, cb, Ks, qt):
    """Early epoch ends with a rebuild at every one of them, many slices, a tiny seed -- the 64-key lists must come out the same;
    and a launch whose guard gives up at once is redone by the gated u16 pass behind it."""
    import torch
    from annlite_amd._capi import LUT_L2

    rs = np.random.RandomState(3 + M)
    N, B, k = 200_000, 40, 50
    dt = np.uint8 if cb == 1 else np.uint16
    base = rs.randint(0, Ks, size=(3000, M)).astype(dt)
    codes = base[rs.randint(0, 3000, size=N)]
    lut = rs.rand(B, M, Ks).astype(np.float32)
    rd, ri = _oracle(oracle, lut, codes, k)
    for env in ({'ANNLITE_Q8_TUNE': '3,2,192,0', 'ANNLITE_Q8_REBUILD': '8'}, {'ANNLITE_SCAN_SLICES': '24', 'ANNLITE_SEED_ROWS': '100'},
                {'ANNLITE_Q8_TARGET': '40'}):
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        d, i = _scan(ops, monkeypatch, codes, lut, k, 0, qt)
        for key in env:
            monkeypatch.delenv(key)
        assert np.array_equal(d, rd) and np.array_equal(i, ri), env
    # the library's own search, guard forced to give up at the first candidate: the gated u16 pass answers
    monkeypatch.delenv('ANNLITE_SCAN_VARIANT', raising=False)
    monkeypatch.setenv('ANNLITE_GUARD_BASE', '0')
    dsub = 4
    cbk = rs.randn(M, Ks, dsub).astype(np.float32)
    q = rs.randn(B, M * dsub).astype(np.float32)
    d, i = ops.pq_search_topk(LUT_L2, ops.to_dev(q), ops.to_dev(cbk), ops.to_dev(codes), k, M, Ks)
    torch.cuda.synchronize()
    lut2 = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cbk)
    rd2, ri2 = _oracle(oracle, lut2, codes, k)
    assert np.array_equal(d.cpu().numpy(), rd2) and np.array_equal(i.cpu().numpy(), ri2)
