"""Round 6: the PAIR walk (annlite_graph_search_packed_ex, expand_width = 2) -- the two best unexpanded entries of the list are
expanded together, one half wave per packed record -- pinned BIT FOR BIT against a plain restatement of its order.

The restatement below is hnswlib's searchBaseLayerST (include/hnswlib/hnswalg.h:243-329: a list of the ef best nodes seen, the
best unexpanded one expanded until none is left) with the two deviations this build's GPU walk has had since round 2 -- the seeds
(the top of the hierarchy) are scanned flat in rounds of 64, and a step's neighbours are tested against the list's worst entry
BEFORE any of them is inserted -- and a width: W nodes per step.  W = 1 must reproduce the one-at-a-time kernels (which checks the
restatement against three walks that were pinned in round 5), W = 2 the pair walk.  Edge distances are hnswlib::PQLookup
(space_pq.h:15-37) through the oracle's ``adc_gather_c``: the keys compared here are the exact bits the kernels compare."""
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
    links = np.zeros((N, L + 1), np.uint32)
    cnt = np.where(rs.rand(N) < full_frac, L, rs.randint(0, L + 1, N)).astype(np.uint32)
    links[:, 0] = cnt
    nb = rs.randint(0, N, size=(N, L)).astype(np.uint32)
    near = (np.arange(N)[:, None] + rs.randint(1, 50, size=(N, L))) % N
    nb = np.where(rs.rand(N, L) < 0.7, near, nb).astype(np.uint32)
    nb[rs.rand(N, L) < 0.002] = N + 5  # beyond the table: never followed
    nb[:, 1] = np.where(rs.rand(N) < 0.05, nb[:, 0], nb[:, 1])  # a neighbour listed twice
    links[:, 1:] = nb
    return links.view(np.int32)


def walk_restated(oracle, links, seeds, codes, lut, ef, width, valid=None):
    """One query.  ``links`` u32 [N, L+1] (count, ids), ``lut`` f32 [M, Ks].  Returns (ids i64 [ef], dist f32 [ef])."""
    N = codes.shape[0]
    E = 1 if ef <= 64 else 2 if ef <= 128 else 3 if (ef <= 192 and width == 2) else 4  # registers per lane of the kernel's list
    slots, cap = 64 * E, min(ef, 64 * E)
    lst = []  # sorted [(distance bits as ordered key, node, expanded)]: the kernels' list of 64 E entries, `cap` of them live
    seen = set()

    def key(d):  # f32_to_ordered for the non-negative finite sums of these tests: the float's own bits ascend with it
        return int(np.float32(d).view(np.uint32))

    def offer(nodes):
        if not len(nodes):
            return
        nodes = np.asarray(nodes, dtype=np.int64)
        d = oracle.adc_gather_c(lut, codes, nodes)
        worst = (lst[cap - 1][0], lst[cap - 1][1]) if len(lst) >= cap else (1 << 40, 1 << 40)
        new = [(key(x), int(n), False) for x, n in zip(d, nodes) if (key(x), int(n)) < worst]
        lst.extend(new)
        lst.sort(key=lambda t: (t[0], t[1]))
        del lst[slots:]

    seeds = [int(s) for s in seeds]
    for s0 in range(0, len(seeds), 64):
        offer([s for s in seeds[s0:s0 + 64] if s < N])
    for t in lst:
        seen.add(t[1])
    while True:
        pick = [i for i, t in enumerate(lst[:cap]) if not t[2]][:width]
        if not pick:
            break
        for i in pick:
            lst[i] = (lst[i][0], lst[i][1], True)
        fresh = []
        for i in pick:
            row = links[lst[i][1]]
            for nb in row[1:1 + int(row[0])]:
                nb = int(nb)
                if nb < N and nb not in seen:
                    seen.add(nb)
                    fresh.append(nb)
        offer(fresh)
    ids = np.full(ef, -1, np.int64)
    dist = np.full(ef, np.inf, np.float32)
    for i, t in enumerate(lst[:cap]):
        if valid is None or valid[t[1]]:
            ids[i] = t[1]
            dist[i] = np.uint32(t[0]).view(np.float32)
    return ids, dist


@pytest.mark.parametrize('M,L,ef', [(16, 32, 128), (16, 32, 64), (16, 32, 200), (8, 32, 100), (32, 24, 128), (16, 5, 10), (16, 32, 33),
                                    (16, 17, 70), (16, 32, 160), (8, 32, 192), (16, 32, 129), (64, 32, 128), (64, 20, 50)])
def test_pair_walk_equals_its_restatement(ops, oracle, M, L, ef):
    import torch

    rs = np.random.RandomState(M * 1000 + L * 7 + ef)
    N, B, Ks = 6000, 10, 256
    links = _random_graph(rs, N, L)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[1000:1040] = codes[1000]  # exact ties inside the walk: the key's low word (the node id) decides
    lut = rs.rand(B, M, Ks).astype(np.float32)
    seeds = rs.choice(N, 150, replace=False).astype(np.int32)
    valid = rs.rand(N) < 0.9
    bits = np.zeros(((N + 31) // 32 + 2) * 32, bool)
    bits[:N] = valid
    vb = ops.to_dev(np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    lk = links.view(np.uint32)
    for width in (1, 2):
        for vbits, vmask in ((None, None), (vb, valid)):
            gi, gd = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, valid_bits=vbits, expand_width=width)
            torch.cuda.synchronize()
            gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
            for b in range(B):
                ri, rd = walk_restated(oracle, lk, seeds, codes, lut[b], ef, width, vmask)
                assert np.array_equal(gi[b], ri), (width, b)
                assert np.array_equal(gd[b].view(np.uint32), rd.view(np.uint32)), (width, b)
    # width 1 through the _ex entry is the round-5 entry point's walk
    a = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef)
    torch.cuda.synchronize()
    assert np.array_equal(a[0].cpu().numpy(), ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, expand_width=1)[0].cpu().numpy())


def test_pair_walk_needs_half_wave_records(ops):
    rs = np.random.RandomState(3)
    N, M, L, Ks = 500, 16, 40, 256
    links = _random_graph(rs, N, L)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    lut = rs.rand(2, M, Ks).astype(np.float32)
    seeds = np.arange(20, dtype=np.int32)
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    with pytest.raises(Exception, match='width'):
        ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, 32, expand_width=2)
    with pytest.raises(Exception, match='width'):
        ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, 32, expand_width=3)


def test_pair_walk_with_a_full_visited_table(ops, oracle, monkeypatch):
    """64 visited slots: nodes are evaluated again and again, the merge's duplicate tests keep every node once; the list is a sorted
    list of distinct nodes with exact sums (the ORDER of a walk that re-evaluates is the table's business, not pinned)."""
    import torch

    rs = np.random.RandomState(78)
    N, B, Ks, M, L, ef = 20_000, 33, 256, 16, 32, 96
    links = _random_graph(rs, N, L)
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    lut = rs.rand(B, M, Ks).astype(np.float32)
    seeds = rs.choice(N, 300, replace=False).astype(np.int32)
    links_d, codes_d, lut_d, seeds_d = ops.to_dev(links), ops.to_dev(codes), ops.to_dev(lut), ops.to_dev(seeds)
    packed = ops.graph_pack(links_d, codes_d)
    ref_i, _ = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, expand_width=2)
    torch.cuda.synchronize()
    monkeypatch.setenv('ANNLITE_GRAPH_HASH_BITS', '6')
    qi, qd = ops.graph_search_packed(packed, L, seeds_d, codes_d, lut_d, ef, expand_width=2)
    torch.cuda.synchronize()
    ids, dd = qi.cpu().numpy(), qd.cpu().numpy()
    same = 0
    for b in range(B):
        real = ids[b][ids[b] >= 0]
        assert len(np.unique(real)) == len(real)
        assert np.array_equal(dd[b][ids[b] >= 0], oracle.adc_gather_c(lut[b], codes, real))
        keys = list(zip(dd[b][ids[b] >= 0].view(np.uint32).tolist(), real.tolist()))
        assert keys == sorted(keys)
        same += len(np.intersect1d(real, ref_i[b].cpu().numpy()))
    assert same >= 0.98 * B * ef  # a re-evaluated node has the same key: the walk is the same walk up to the order of re-visits


def test_pair_walk_on_a_real_graph_finds_what_the_one_at_a_time_walk_finds(ops, oracle):
    """HnswPQGpuIndex: expand_width 2 (default) against 1 -- candidate overlap and the recall of the ADC top-10 they lead to."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    rs = np.random.RandomState(5)
    N, D, M, B = 40_000, 64, 16, 64
    A = rs.randn(8, D).astype(np.float32)
    x = (rs.randn(N, 8).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rs.randn(B, 8).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 3
    codec.fit(x[:8192], iter=8)
    hn = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=128, rerank=False, build='host')
    assert hn.expand_width == 2
    hn.add_with_ids(x, np.arange(N))
    qd = hn._pre(q)
    two = hn.candidates(qd, 128)
    hn.expand_width = 1
    one = hn.candidates(qd, 128)
    hn.packed_graph = False
    plain = hn.candidates(qd, 128)
    torch.cuda.synchronize()
    assert np.array_equal(one[0].cpu().numpy(), plain[0].cpu().numpy())  # width 1 IS the plain walk
    i1, i2 = one[0].cpu().numpy(), two[0].cpu().numpy()
    overlap = np.mean([len(np.intersect1d(i1[b][i1[b] >= 0], i2[b][i2[b] >= 0])) / max(1, (i1[b] >= 0).sum()) for b in range(B)])
    assert overlap >= 0.97, overlap
    # the ten best of the two lists (ascending ADC distance): the pair walk's are at least as good on average
    d1, d2 = one[1].cpu().numpy()[:, :10], two[1].cpu().numpy()[:, :10]
    assert d2.mean() <= d1.mean() * 1.001
    # and exact sums: the walk's L2 tables (pyx:149-210) and PQLookup over the stored code rows
    codes = hn._plain_table(N).cpu().numpy()
    luts = np.asarray(oracle.batch_precompute_adc_table_c(q, D // M, 256, np.ascontiguousarray(codec.codebooks, dtype=np.float32)))
    dd = two[1].cpu().numpy()
    for b in range(0, B, 7):
        ok = i2[b] >= 0
        assert np.array_equal(dd[b][ok], oracle.adc_gather_c(luts[b], codes, i2[b][ok]))
