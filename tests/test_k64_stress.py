"""Round 5: randomised shapes for the byte-table kernel's 64-key lists (16 < k <= 64) -- slice counts 1 ... 24 (forced through
ANNLITE_SCAN_SLICES as well as planned), ragged last tiles, structured and tied data, delete marks, epoch schedules that rebuild
the tables -- every query against the oracle; and for the packed graph walk against the plain walk on random graphs of every
width.  Kept short enough for the suite; `scripts/stress_fixture.py` is the long-running sibling for the k <= 16 paths."""
import os

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


def _bits(valid):
    bits = np.zeros(((len(valid) + 31) // 32 + 2) * 32, dtype=bool)
    bits[:len(valid)] = valid
    return np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1)


CASES = [  # seed, N, B, k, forced slices (0 = the plan's), epoch tune or None
    (1, 35_000, 7, 50, 0, None), (2, 35_000, 40, 17, 2, None), (3, 120_000, 33, 64, 4, None), (4, 500_000, 70, 33, 8, None),
    (5, 500_000, 5, 50, 16, None), (6, 900_000, 129, 50, 0, None), (7, 1_500_000, 64, 41, 24, None), (8, 300_000, 200, 50, 1, None),
    (9, 700_000, 96, 50, 8, '1,2,192,0'), (10, 700_000, 31, 64, 0, '3,4,256,1'), (11, 250_000, 17, 20, 0, '100000,2,384,7'),
    (12, 2_000_000, 300, 50, 0, None),
]


@pytest.mark.parametrize('seed,N,B,k,slices,tune', CASES)
def test_k64_random_configurations_equal_the_oracle(ops, oracle, monkeypatch, seed, N, B, k, slices, tune):
    import torch
    from annlite_amd._capi import LUT_L2

    monkeypatch.setenv('ANNLITE_SCAN_VARIANT', '50')
    if slices:
        monkeypatch.setenv('ANNLITE_SCAN_SLICES', str(slices))
    if tune:
        monkeypatch.setenv('ANNLITE_Q8_TUNE', tune)
        monkeypatch.setenv('ANNLITE_Q8_REBUILD', '7')
    rs = np.random.RandomState(seed)
    M, dsub, Ks = 16, 8, 256
    D = M * dsub
    A = rs.randn(12, D).astype(np.float32)
    cb = np.stack([(rs.randn(Ks, 12).astype(np.float32) @ A)[:, m * dsub:(m + 1) * dsub] for m in range(M)]).astype(np.float32)
    # codes with structure (a few thousand distinct rows + noise in some sub-spaces): ties and near-ties at every rank
    base = rs.randint(0, Ks, size=(3000, M)).astype(np.uint8)
    codes = base[rs.randint(0, 3000, N)]
    noise = rs.rand(N, M) < 0.15
    codes = np.where(noise, rs.randint(0, Ks, size=(N, M)), codes).astype(np.uint8)
    q = (rs.randn(B, 12).astype(np.float32) @ A + 0.1 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    valid = rs.rand(N) < 0.93
    valid[:64] = rs.rand(64) < 0.5  # holes in the seed rows too
    lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    live = np.nonzero(valid)[0]
    rd, ri = oracle.adc_search_c(lut, codes[live], k, threads=oracle.max_threads())
    ri = np.where(ri >= 0, live[np.clip(ri, 0, len(live) - 1)], -1)
    cb_d, q_d, codes_d, vb = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes), ops.to_dev(_bits(valid))
    for layout in (1, 0):
        cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
        for rep in range(2):
            d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, valid_bits=vb)
            torch.cuda.synchronize()
            assert np.array_equal(i.cpu().numpy(), ri), (layout, rep, 'ids')
            assert np.array_equal(d.cpu().numpy(), rd), (layout, rep, 'distances')


def test_k64_guarded_launch_gives_up_and_the_gated_pass_answers(ops, oracle, monkeypatch):
    """ANNLITE_GUARD_BASE=0 (no variant switch): the guarded byte-table launch declares itself lost at once -- its partial lists are
    garbage, merge_partial_kernel merges garbage -- and the gated u16 pass behind it overwrites the outputs: the oracle's bits."""
    import torch
    from annlite_amd._capi import LUT_L2

    monkeypatch.delenv('ANNLITE_SCAN_VARIANT', raising=False)
    monkeypatch.setenv('ANNLITE_GUARD_BASE', '0')
    rs = np.random.RandomState(77)
    N, M, dsub, Ks, B, k = 400_000, 16, 8, 256, 50, 50
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)  # no structure at all
    q = rs.randn(B, M * dsub).astype(np.float32)
    lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb)
    rd, ri = oracle.adc_search_c(lut, codes, k, threads=oracle.max_threads())
    cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
    for rep in range(2):
        d, i = ops.pq_search_topk(LUT_L2, q_d, cb_d, codes_d, k, M, Ks)
        torch.cuda.synchronize()
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd), rep


@pytest.mark.parametrize('seed', range(10))
def test_packed_walk_random_graphs(ops, oracle, monkeypatch, seed):
    """random width / ef / sub-space count / seed count / visited-table size: packed == plain == one-at-a-time insertion"""
    import torch

    rs = np.random.RandomState(1000 + seed)
    M = int(rs.choice([8, 16, 32]))
    L = int(rs.choice([1, 2, 7, 16, 31, 32, 33, 48, 64]))
    ef = int(rs.choice([1, 5, 32, 63, 64, 65, 127, 128, 129, 200, 256]))
    N = int(rs.choice([500, 5000, 40_000]))
    B = int(rs.choice([1, 3, 64, 90]))
    n_seeds = int(rs.choice([1, 10, 64, 65, 500]))
    hb = int(rs.choice([5, 8, 12, 13]))
    Ks = 256
    links = np.zeros((N, L + 1), np.uint32)
    cnt = rs.randint(0, L + 1, N).astype(np.uint32)
    cnt[rs.rand(N) < 0.5] = L
    links[:, 0] = cnt
    near = (np.arange(N)[:, None] + rs.randint(1, 30, size=(N, L))) % N
    far = rs.randint(0, N, size=(N, L))
    links[:, 1:] = np.where(rs.rand(N, L) < 0.8, near, far)
    if L > 1:
        links[:, 2] = np.where(rs.rand(N) < 0.1, links[:, 1], links[:, 2])  # duplicate neighbours
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[: N // 10] = codes[0]  # a block of exact ties
    lut = rs.rand(B, M, Ks).astype(np.float32)
    seeds = rs.choice(N, min(n_seeds, N), replace=False).astype(np.int32)
    valid = rs.rand(N) < 0.8
    vb = ops.to_dev(_bits(valid))
    monkeypatch.setenv('ANNLITE_GRAPH_HASH_BITS', str(hb))
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links.view(np.int32)), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    pi, pd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef, valid_bits=vb)
    qi, qd = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, valid_bits=vb)
    monkeypatch.setenv('ANNLITE_GRAPH_SEQ_INSERT', '1')
    si, sd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef, valid_bits=vb)
    torch.cuda.synchronize()
    cfg = (M, L, ef, N, B, n_seeds, hb)
    assert np.array_equal(pi.cpu().numpy(), qi.cpu().numpy()) and np.array_equal(pd.cpu().numpy().view(np.uint32), qd.cpu().numpy().view(np.uint32)), cfg
    assert np.array_equal(si.cpu().numpy(), qi.cpu().numpy()) and np.array_equal(sd.cpu().numpy().view(np.uint32), qd.cpu().numpy().view(np.uint32)), cfg
    ids, dd = qi.cpu().numpy(), qd.cpu().numpy()
    for b in range(B):
        ok = ids[b] >= 0
        assert valid[ids[b][ok]].all() and len(np.unique(ids[b][ok])) == ok.sum(), cfg
        assert np.array_equal(dd[b][ok], oracle.adc_gather_c(lut[b], codes, ids[b][ok])), cfg
