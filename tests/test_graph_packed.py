"""Round 5: the GPU graph walk over PACKED node records (annlite_graph_pack / annlite_graph_search_packed) -- every node's
neighbours' code rows inline behind its link list, the next record prefetched -- against the plain walk
(annlite_graph_search: link lists + the code table): same walk order, same hnswlib::PQLookup arithmetic
(space_pq.h:15-37, hnswalg.h:243-329), so the candidate lists must be BIT-EQUAL, ids and distances."""
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


def _random_graph(rs, N, L, full_frac=0.6):
    """links i32 [N, L+1]: a random regular-ish graph (what the walk needs is a graph, not a good one) with ragged counts,
    a few empty lists, ids beyond the table (never followed) and duplicate neighbours"""
    links = np.zeros((N, L + 1), np.uint32)
    cnt = np.where(rs.rand(N) < full_frac, L, rs.randint(0, L + 1, N)).astype(np.uint32)
    links[:, 0] = cnt
    nb = rs.randint(0, N, size=(N, L)).astype(np.uint32)
    near = (np.arange(N)[:, None] + rs.randint(1, 50, size=(N, L))) % N  # locality: walks make progress
    nb = np.where(rs.rand(N, L) < 0.7, near, nb).astype(np.uint32)
    nb[rs.rand(N, L) < 0.002] = N + 5  # beyond the table
    nb[:, 1] = np.where(rs.rand(N) < 0.05, nb[:, 0], nb[:, 1])  # duplicates
    links[:, 1:] = nb
    return links.view(np.int32)


@pytest.mark.parametrize('M,L,ef', [(16, 32, 128), (16, 32, 64), (16, 32, 200), (8, 32, 128), (32, 24, 100), (16, 64, 128), (16, 5, 10),
                                    (16, 32, 33)])
def test_packed_walk_bit_equal_to_plain_walk(ops, oracle, monkeypatch, M, L, ef):
    import torch

    rs = np.random.RandomState(M * 100 + L + ef)
    N, B, Ks = 60_000, 70, 256
    links = _random_graph(rs, N, L)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[1000:1100] = codes[1000]  # exact distance ties inside the walk
    lut = rs.rand(B, M, Ks).astype(np.float32)
    seeds = rs.choice(N, 700, replace=False).astype(np.int32)
    valid = rs.rand(N) < 0.9  # deleted rows route the walk and are dropped from the result
    bits = np.zeros(((N + 31) // 32 + 2) * 32, bool)
    bits[:N] = valid
    vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    # the records are what the header says: [L x M code bytes][L ids][count]
    rec = packed.cpu().numpy()
    lk = links.view(np.uint32)
    for n in (0, 17, N - 1):
        cnt = int(lk[n, 0])
        ids = rec[n, L * M:L * M + 4 * L].view(np.uint32)
        assert int(rec[n, L * M + 4 * L:L * M + 4 * L + 4].view(np.uint32)[0]) == cnt
        assert np.array_equal(ids[:cnt], lk[n, 1:1 + cnt]) and (ids[cnt:] == 0xFFFFFFFF).all()
        for j in range(cnt):
            if lk[n, 1 + j] < N:
                assert np.array_equal(rec[n, j * M:(j + 1) * M], codes[lk[n, 1 + j]])
    for vbits in (None, vb):
        pi, pd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef, valid_bits=vbits)
        qi, qd = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, valid_bits=vbits)
        # ... and the plain walk with the one-at-a-time list insertion of rounds 2-4 (the merge insertion builds the same list)
        monkeypatch.setenv('ANNLITE_GRAPH_SEQ_INSERT', '1')
        si, sd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef, valid_bits=vbits)
        monkeypatch.delenv('ANNLITE_GRAPH_SEQ_INSERT')
        torch.cuda.synchronize()
        assert np.array_equal(pi.cpu().numpy(), qi.cpu().numpy())
        assert np.array_equal(pd.cpu().numpy().view(np.uint32), qd.cpu().numpy().view(np.uint32))
        assert np.array_equal(si.cpu().numpy(), qi.cpu().numpy())
        assert np.array_equal(sd.cpu().numpy().view(np.uint32), qd.cpu().numpy().view(np.uint32))
        ids = qi.cpu().numpy()
        # and the distances are the oracle's PQLookup of those rows
        dd = qd.cpu().numpy()
        for b in range(0, B, 9):
            ok = ids[b] >= 0
            assert np.array_equal(dd[b][ok], oracle.adc_gather_c(lut[b], codes, ids[b][ok]))
            if vbits is not None:
                assert valid[ids[b][ok]].all()


def test_a_full_visited_table_changes_nothing_but_the_speed(ops, oracle, monkeypatch):
    """ANNLITE_GRAPH_HASH_BITS=6: 64 visited slots -- the table is full after the first expansions, nodes are re-evaluated and the
    list's duplicate check keeps them out (one-at-a-time insertion: per candidate; merge insertion: against the list and against
    the earlier candidates of the same expansion).  All three walks still agree, and equal the walk with a large table."""
    import torch

    rs = np.random.RandomState(77)
    N, B, Ks, M, L, ef = 20_000, 33, 256, 16, 32, 96
    links = _random_graph(rs, N, L)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    seeds = rs.choice(N, 300, replace=False).astype(np.int32)
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    ref_i, ref_d = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef)
    torch.cuda.synchronize()
    monkeypatch.setenv('ANNLITE_GRAPH_HASH_BITS', '6')
    pi, pd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef)
    qi, qd = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef)
    monkeypatch.setenv('ANNLITE_GRAPH_SEQ_INSERT', '1')
    si, sd = ops.graph_search(links_d, seeds_d, codes_d, lut_d, ef)
    torch.cuda.synchronize()
    for ai, ad in ((pi, pd), (qi, qd)):
        assert np.array_equal(ai.cpu().numpy(), si.cpu().numpy())
        assert np.array_equal(ad.cpu().numpy().view(np.uint32), sd.cpu().numpy().view(np.uint32))
    ids = si.cpu().numpy()
    for b in range(B):  # no node twice in a list
        real = ids[b][ids[b] >= 0]
        assert len(np.unique(real)) == len(real)
    # a table that never fills records every node once: the same walk unless the full table's re-evaluations changed the order
    # of ties -- they cannot: a re-evaluated node has the same key
    assert np.array_equal(ref_i.cpu().numpy(), ids) and np.array_equal(ref_d.cpu().numpy().view(np.uint32), sd.cpu().numpy().view(np.uint32))


def test_index_uses_packed_records_and_rebuilds_them_after_inserts(ops, oracle):
    """HnswPQGpuIndex walks packed records by default; ``packed_graph=False`` is the plain walk -- same candidates -- and the
    records follow the graph through inserts and deletes."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    rs = np.random.RandomState(4)
    N, D, M, B = 30_000, 64, 16, 40
    A = rs.randn(8, D).astype(np.float32)
    x = (rs.randn(N, 8).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 8).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 3
    codec.fit(x[:8192], iter=8)
    hn = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=100, rerank=False,
                        expand_width=1, build='host')  # (one node per step: the walk that equals the plain walk; the pair walk:
    # test_graph_pair.py.  build='host': this test is about the records packed from the host library's exported lists)
    hn.add_with_ids(x[:20_000], np.arange(20_000))
    qd = hn._pre(q)

    def both():
        hn.packed_graph = True
        a = hn.candidates(qd, 100)
        hn.packed_graph = False
        b = hn.candidates(qd, 100)
        hn.packed_graph = True
        torch.cuda.synchronize()
        assert np.array_equal(a[0].cpu().numpy(), b[0].cpu().numpy())
        assert np.array_equal(a[1].cpu().numpy().view(np.uint32), b[1].cpu().numpy().view(np.uint32))
        return a[0].cpu().numpy()

    first = both()
    assert hn._packed is not None and hn._packed.shape[0] == 20_000
    hn.add_with_ids(x[20_000:], np.arange(20_000, N))  # the graph grows: new export, new records
    second = both()
    assert hn._packed.shape[0] == N and (second >= 20_000).any()
    hn.delete(np.unique(second[:, 0]))  # the best candidate of every query: gone from the lists, still routing
    third = both()
    assert not np.isin(third[third >= 0], np.unique(second[:, 0])).any()
    assert first.shape == second.shape == third.shape
