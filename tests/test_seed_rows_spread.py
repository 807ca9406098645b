"""The first bound of a scan comes from S seed rows spread over the whole table in runs of 64-row blocks (scan_prep.hip:
seed_bound_kernel; round 5) instead of from its first S rows.  Any subset of the valid rows gives a correct bound, so the results
must not move: a table filled in cluster order (the case the spread is for), a deleted head, ragged sizes, both code layouts,
k on the 16-key and the 64-key lists, every run length, more seed rows than the table has, and the opt-in interleaved row slices
of the scan itself -- bit-exact against the CPU oracle
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


def _sorted_table(N, M, dsub, Ks, seed):
    """Low-rank vectors through random code books, rows ordered along the first latent direction."""
    rs = np.random.RandomState(seed)
    D = M * dsub
    A = rs.randn(8, D).astype(np.float32)
    z = rs.randn(N, 8).astype(np.float32)
    x = z @ A + 0.05 * rs.randn(N, D).astype(np.float32)
    cb = (rs.randn(Ks, 8).astype(np.float32) @ A).reshape(Ks, M, dsub).transpose(1, 0, 2).copy()
    order = np.argsort(z[:, 0], kind='stable')
    return cb, x[order], A, rs


ENVS = [{}, {'ANNLITE_SEED_CONTIGUOUS': '1'}, {'ANNLITE_SEED_CHUNK_LOG': '0'}, {'ANNLITE_SEED_CHUNK_LOG': '6'},
        {'ANNLITE_SEED_ROWS': '100'}, {'ANNLITE_SEED_ROWS': '100000000'}, {'ANNLITE_SEED_ROWS': '0'},
        # interleaved row slices (scan.hip at ANNLITE_Q8_ILV; opt-in): runs of 2 / 16 / 256 blocks, 8 and 24 slices
        {'ANNLITE_Q8_ILV': '1'}, {'ANNLITE_Q8_ILV': '4'}, {'ANNLITE_Q8_ILV': '8', 'ANNLITE_SCAN_SLICES': '8'},
        {'ANNLITE_Q8_ILV': '2', 'ANNLITE_SCAN_SLICES': '24'}]


@pytest.mark.parametrize('N,k', [(300_001, 10), (300_001, 50), (65_600, 16), (4_100, 10)])
def test_spread_seed_rows_keep_the_results(ops, oracle, monkeypatch, N, k):
    import torch
    from annlite_amd._capi import LUT_L2

    M, dsub, Ks, B = 16, 8, 256, 100
    cb, x, A, rs = _sorted_table(N, M, dsub, Ks, 11 + k)
    codes = ops.pq_encode(ops.to_dev(x), ops.to_dev(cb)).cpu().numpy()
    q = (rs.randn(B, 8).astype(np.float32) @ A).astype(np.float32)
    q[:, :] += 0.5 * A[0]  # (queries off to one side of the order: their neighbours sit far from the head of the table)
    valid = np.ones(((N + 31) // 32 + 2) * 32, dtype=bool)
    valid[N:] = False
    valid[:N // 5] = False  # the head of the table is deleted: its first seed rows are all invalid
    valid[rs.choice(N, N // 20, replace=False)] = False
    lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    live = np.nonzero(valid[:N])[0]
    rd, ri = oracle.adc_search_c(lut, codes[live], k, threads=oracle.max_threads())
    ri = live[ri]
    rd0, ri0 = oracle.adc_search_c(lut, codes, k, threads=oracle.max_threads())
    bits = ops.to_dev(np.packbits(valid.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    for env in ENVS:
        for key in ('ANNLITE_SEED_CONTIGUOUS', 'ANNLITE_SEED_CHUNK_LOG', 'ANNLITE_SEED_ROWS', 'ANNLITE_Q8_ILV', 'ANNLITE_SCAN_SLICES'):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        for layout in (0, 1):
            cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
            for vb in (bits, None):
                d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, valid_bits=vb)
                if vb is None:
                    assert np.array_equal(i.cpu().numpy(), ri0) and np.array_equal(d.cpu().numpy(), rd0), (env, layout)
                else:
                    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd), (env, layout)
    torch.cuda.synchronize()


def test_spread_seed_rows_on_a_table_in_cluster_order_is_faster(ops, oracle, monkeypatch):
    """1M rows ordered along one latent direction, 1024 queries: the bound from the head of the table is far off for most queries
    (their tables stay coarse), the bound from spread rows is not -- measured 0.645 against 0.330 ms per batch at 1.25M rows
    (profiles/r05/seed_rows_spread.txt).  Same bits either way."""
    import torch
    from annlite_amd._capi import LUT_L2

    M, dsub, Ks, B, k, N = 16, 8, 256, 1024, 10, 1_000_000
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    A = torch.randn((16, M * dsub), generator=g, device='cuda')
    z = torch.randn((N, 16), generator=g, device='cuda')
    z = z[torch.argsort(z[:, 0])]
    cb = (torch.randn((Ks, 16), generator=g, device='cuda') @ A).reshape(Ks, M, dsub).permute(1, 0, 2).contiguous()
    codes = torch.empty((N, M), dtype=torch.uint8, device='cuda')
    for c0 in range(0, N, 250_000):
        zz = z[c0:c0 + 250_000]
        codes[c0:c0 + 250_000] = ops.pq_encode((zz @ A + 0.05 * torch.randn((zz.shape[0], M * dsub), generator=g, device='cuda')).contiguous(), cb)
    q = (torch.randn((B, 16), generator=g, device='cuda') @ A).contiguous()
    sk = ops.codes_skew(codes)
    ws = ops.ScanWorkspace()

    def run():
        return ops.pq_search_topk(LUT_L2, q, cb, sk, k, M, Ks, codes_layout=1, workspace=ws)

    def ms():
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 20

    monkeypatch.delenv('ANNLITE_SEED_CONTIGUOUS', raising=False)
    t_spread = ms()
    d0, i0 = run()
    monkeypatch.setenv('ANNLITE_SEED_CONTIGUOUS', '1')
    t_head = ms()
    d1, i1 = run()
    monkeypatch.delenv('ANNLITE_SEED_CONTIGUOUS')
    assert torch.equal(d0, d1) and torch.equal(i0, i1)
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/seed_rows_spread_test.txt', 'w') as f:
        f.write('1M rows in cluster order, 1024 queries, ms per batch: seed rows spread %.4f, first rows %.4f\n' % (t_spread, t_head))
    assert t_spread < 0.9 * t_head, (t_spread, t_head)
    lut = oracle.batch_precompute_adc_table_c(q[:8].cpu().numpy(), dsub, Ks, cb.cpu().numpy())
    rd, ri = oracle.adc_search_c(lut, codes.cpu().numpy(), k, threads=oracle.max_threads())
    assert np.array_equal(i0[:8].cpu().numpy(), ri) and np.array_equal(d0[:8].cpu().numpy(), rd)
