"""Round 6: the level-0 graph BUILT ON THE GPU in batches (annlite_amd/csrc/graph_build.hip, core/index/graph_gpu_build.py).

The rules are hnswlib's addPoint (include/hnswlib/hnswalg.h:1108-1235: search with ef_construction, getNeighborsByHeuristic2
378-429, mutuallyConnectNewElement 431-553) in the form libannlite_graph.so restates them (hnsw_host.cpp select_neighbors /
connect: triangle tests on the symmetric code-to-code L2 table).  The kernels are checked against plain restatements of those
rules here (same fp32 sums, same order => same decisions), the whole build against the host-built graph by what a graph is for:
the candidates a walk finds."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]
SENT = np.iinfo(np.int64).max


@pytest.fixture(scope='module')
def ops():
    import torch
    from annlite_amd import ops as _ops

    torch.cuda.set_device(0)
    return _ops


def _sdc_np(cb):
    M, Ks, dsub = cb.shape
    out = np.zeros((M, Ks, Ks), np.float32)
    for j in range(dsub):  # the kernel's chain: acc += (a_j - b_j) * (a_j - b_j), fp32, j ascending
        d = (cb[:, :, None, j] - cb[:, None, :, j]).astype(np.float32)
        out = (out + (d * d).astype(np.float32)).astype(np.float32)
    return out


def _sym(sdc, ca, cb_):
    r = np.float32(0)
    for m in range(len(ca)):
        r = np.float32(r + sdc[m, ca[m], cb_[m]])
    return r


def _select(sdc, codes, base, pool, m_max):
    """hnsw_host.cpp select_neighbors over a pool that is already in visiting order."""
    pool = [int(p) for p in pool if p >= 0 and p != base]
    if len(pool) <= m_max:
        return pool
    keep = []
    for c in pool:
        if len(keep) >= m_max:
            break
        tb = _sym(sdc, codes[base], codes[c])
        if all(not (_sym(sdc, codes[k], codes[c]) < tb) for k in keep):
            keep.append(c)
    return keep


@pytest.mark.parametrize('M,ef,keep', [(16, 200, 16), (8, 64, 16), (32, 256, 12), (16, 10, 16), (16, 40, 4), (64, 200, 16)])
def test_select_kernel_is_algorithm_4(ops, M, ef, keep):
    import torch

    rs = np.random.RandomState(M + ef)
    N, Ks, dsub, b = 3000, 256, 4, 37
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    # clustered rows: many candidates are closer to a kept neighbour than to the base point
    centers = rs.randint(0, Ks, size=(12, M))
    codes = centers[rs.randint(0, 12, N)].astype(np.uint8)
    flip = rs.rand(N, M) < 0.25
    codes[flip] = rs.randint(0, Ks, int(flip.sum())).astype(np.uint8)
    cb_d, codes_d = ops.to_dev(cb), ops.to_dev(codes)
    sdc_d = ops.graph_build_sdc(cb_d)
    sdc = _sdc_np(cb)
    torch.cuda.synchronize()
    assert np.array_equal(sdc_d.cpu().numpy(), sdc)
    base0 = N - b
    cand = np.full((b, ef), -1, np.int64)
    for i in range(b):
        n_c = ef if i % 5 else rs.randint(0, ef + 1)  # ragged lists, some shorter than `keep`
        ids = rs.choice(base0, n_c, replace=False)
        # visiting order = the base point's symmetric distance (stand-in for the search distance), with a few holes
        ids = ids[np.argsort([_sym(sdc, codes[base0 + i], codes[c]) for c in ids], kind='stable')]
        cand[i, :n_c] = ids
        if n_c > 8:
            cand[i, rs.randint(0, n_c, 2)] = -1
    lpn = 32
    links = torch.zeros((N, lpn + 1), dtype=torch.int32, device='cuda')
    pairs = ops.graph_build_select(ops.to_dev(cand), base0, codes_d, sdc_d, keep, links)
    torch.cuda.synchronize()
    lk, pr = links.cpu().numpy().view(np.uint32), pairs.cpu().numpy()
    for i in range(b):
        want = _select(sdc, codes, base0 + i, cand[i], keep)
        row = lk[base0 + i]
        assert int(row[0]) == len(want), (i, row[:8], want)
        assert row[1:1 + len(want)].tolist() == want, i
        got_pairs = [int(p) for p in pr[i] if p != SENT]
        assert got_pairs == [(t << 32) | (base0 + i) for t in want]
    assert not lk[:base0].any()  # nobody else's row is touched


def test_reverse_kernel_appends_dedupes_and_shrinks(ops):
    import torch

    rs = np.random.RandomState(9)
    N, M, Ks, dsub, lpn = 2000, 16, 256, 4, 32
    cb = rs.randn(M, Ks, dsub).astype(np.float32)
    centers = rs.randint(0, Ks, size=(8, M))
    codes = centers[rs.randint(0, 8, N)].astype(np.uint8)
    flip = rs.rand(N, M) < 0.3
    codes[flip] = rs.randint(0, Ks, int(flip.sum())).astype(np.uint8)
    sdc = _sdc_np(cb)
    links = np.zeros((N, lpn + 1), np.uint32)
    targets = rs.choice(1000, 120, replace=False)
    pairs = []
    want = {}
    for t_i, t in enumerate(targets):
        c = [0, 5, 20, 31, 32, 32, 32][t_i % 7]
        cur = [int(v) for v in rs.choice(np.setdiff1d(np.arange(1000), [t]), c, replace=False)]
        links[t, 0] = c
        links[t, 1:1 + c] = cur
        k = [1, 3, 12, 1, 1, 7, 40][(t_i // 7) % 7]
        src = [int(v) for v in rs.choice(np.arange(1000, N), k, replace=False)]
        if cur and t_i % 3 == 0:
            src.append(cur[0])  # a source that is in the list already: dropped
        if t_i % 11 == 0:
            src.append(int(t))  # the target itself: dropped
        src = sorted(set(src))
        pairs += [(int(t) << 32) | s for s in src]
        fresh = [s for s in src if s not in cur and s != t][: 64 - c]
        pool = cur + [s for s in sorted(src)[: 64 - c] if s not in cur and s != t]
        if len(pool) == len(cur):
            want[int(t)] = cur
        elif len(pool) <= lpn:
            want[int(t)] = pool
        else:
            order = sorted(pool, key=lambda x: (_sym(sdc, codes[t], codes[x]).view(np.uint32), x))
            want[int(t)] = _select(sdc, codes, int(t), order, lpn)
        del fresh
    keys = np.array(sorted(pairs), dtype=np.int64)
    tg = keys >> 32
    bounds = np.concatenate([[0], np.nonzero(np.diff(tg))[0] + 1, [len(keys)]]).astype(np.int64)
    links_d = ops.to_dev(links.view(np.int32))
    ops.graph_build_reverse(ops.to_dev(keys), ops.to_dev(bounds), ops.to_dev(codes), ops.graph_build_sdc(ops.to_dev(cb)), links_d)
    torch.cuda.synchronize()
    got = links_d.cpu().numpy().view(np.uint32)
    for t, w in want.items():
        assert int(got[t, 0]) == len(w), (t, got[t, :6], w[:6])
        assert got[t, 1:1 + len(w)].tolist() == w, t
    untouched = np.setdiff1d(np.arange(N), targets)
    assert np.array_equal(got[untouched], links[untouched])


def test_pack_nodes_equals_the_full_pack(ops):
    import torch

    rs = np.random.RandomState(2)
    N, M, L = 5000, 16, 32
    links = np.zeros((N, L + 1), np.uint32)
    links[:, 0] = rs.randint(0, L + 1, N)
    links[:, 1:] = rs.randint(0, N, size=(N, L))
    codes = rs.randint(0, 256, size=(N, M)).astype(np.uint8)
    links_d, codes_d = ops.to_dev(links.view(np.int32)), ops.to_dev(codes)
    full = ops.graph_pack(links_d, codes_d)
    part = torch.zeros_like(full)
    nodes = torch.from_numpy(rs.choice(N, 700, replace=False).astype(np.int64)).cuda()
    ops.graph_pack_nodes(links_d, codes_d, nodes, part, N)
    torch.cuda.synchronize()
    sel = nodes.cpu().numpy()
    assert np.array_equal(part.cpu().numpy()[sel], full.cpu().numpy()[sel])
    rest = np.setdiff1d(np.arange(N), sel)
    assert not part.cpu().numpy()[rest].any()


def _data(rs, N, D, r=8):
    A = rs.randn(r, D).astype(np.float32)
    return lambda n: (rs.randn(n, r).astype(np.float32) @ A + 0.05 * rs.randn(n, D).astype(np.float32)).astype(np.float32)


def test_gpu_built_graph_against_the_host_built_graph(ops, oracle, tmp_path):
    """60k clustered points, inserted in three calls (exact lists below 8192 nodes, walk batches after): the GPU-built level 0 finds
    what the host-built hierarchy finds -- candidate overlap with the exhaustive ADC top-10 -- lists are well formed, deletes /
    dump / load / later inserts work, and what the GPU build does not have (host walks, sparse ids) is refused."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(11)
    N, D, M, B, k = 60_000, 64, 16, 128, 10
    gen = _data(rs, N, D)
    x, q = gen(N), gen(B)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 3
    codec.fit(x[:8192], iter=8)
    flat = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
    flat.add_with_ids(x, np.arange(N))
    _, fi = flat.search_batch(q, limit=k)

    def overlap(hn):
        hn.rerank = False
        _, hi = hn.search_batch(q, limit=k)
        return np.mean([len(set(hi[b]) & set(fi[b])) / k for b in range(B)])

    host = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=128, rerank=False, build='host')
    host.add_with_ids(x, np.arange(N))
    gpu = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=128, rerank=False)
    assert gpu.build == 'gpu' and host._gg is None  # (the default where the GPU walk applies)
    gpu.add_with_ids(x[:3000], np.arange(3000))          # exact lists
    gpu.add_with_ids(x[3000:20_000], np.arange(3000, 20_000))  # crosses 8192 inside one call
    gpu.add_with_ids(x[20_000:], np.arange(20_000, N))
    assert gpu.size == N and gpu._gg.n == N
    o_host, o_gpu = overlap(host), overlap(gpu)
    assert o_gpu >= 0.9 and o_gpu >= o_host - 0.02, (o_gpu, o_host)
    # well-formed lists: counts in range, no self links, no duplicates, ids inside the table
    lk = gpu._gg.links[:N].cpu().numpy().view(np.uint32)
    cnt = lk[:, 0]
    assert cnt.max() <= 32 and cnt.min() >= 1
    for n in rs.randint(0, N, 300):
        row = lk[n, 1:1 + cnt[n]]
        assert (row < N).all() and n not in row and len(np.unique(row)) == len(row)
    assert cnt.mean() > 12  # (reverse links fill the lists)
    # the records the walk reads are the records of these lists
    full = ops.graph_pack(gpu._gg.links[:N], gpu._gg.codes[:N])
    torch.cuda.synchronize()
    assert torch.equal(full, gpu._gg.packed[:N])
    # candidates carry exact PQLookup sums (space_pq.h:15-37)
    qd = gpu._pre(torch.from_numpy(q).cuda())
    cid, cd = gpu.candidates(qd, 128)
    luts = np.asarray(oracle.batch_precompute_adc_table_c(q, D // M, 256, np.ascontiguousarray(codec.codebooks, dtype=np.float32)))
    codes_np = gpu._gg.codes[:N].cpu().numpy()
    ci, cdn = cid.cpu().numpy(), cd.cpu().numpy()
    for b in range(0, B, 16):
        ok = ci[b] >= 0
        assert np.array_equal(cdn[b][ok], oracle.adc_gather_c(luts[b], codes_np, ci[b][ok]))
    # deletes route but are not returned
    _, hi = gpu.search_batch(q, limit=k)
    gone = np.unique(hi[:, 0])
    gpu.delete(gone.tolist())
    _, hi2 = gpu.search_batch(q, limit=k)
    assert not np.isin(hi2, gone).any()
    # persistence
    gpu.dump(tmp_path / 'g.idx')
    again = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=128, rerank=False, build='gpu')
    again.load(tmp_path / 'g.idx')
    _, hi3 = again.search_batch(q, limit=k)
    assert np.array_equal(hi2, hi3)
    # later inserts into the loaded graph
    more = gen(500)
    again.add_with_ids(more, np.arange(N, N + 500))
    d4, hi4 = again.search_batch(more[:32], limit=1)
    assert (hi4[:, 0] >= 0).all() and np.mean(hi4[:, 0] == np.arange(N, N + 32)) > 0.8  # a point finds itself
    # refusals
    with pytest.raises(RuntimeError, match='insertion order'):
        again.add_with_ids(more[:2], [N + 900, N + 901])
    again.walk = 'host'
    with pytest.raises(RuntimeError, match='level 0 only'):
        again.search_batch(q[:2], limit=k)


def test_gpu_build_tiny_and_one_at_a_time(ops):
    """1, 2, 17 points; then points one by one: every list stays well formed and every point finds itself."""
    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    rs = np.random.RandomState(5)
    D, M = 32, 8
    gen = _data(rs, 0, D, r=4)
    x = gen(4000)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 1
    codec.fit(x, iter=5)
    hn = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=256, ef_search=32, rerank=False, build='gpu',
                        max_connection=8, ef_construction=40)
    n = 0
    for step in (1, 1, 2, 17, 1, 1, 1, 300, 1):
        hn.add_with_ids(x[n:n + step], np.arange(n, n + step))
        n += step
        d, i = hn.search_batch(x[:n][-min(n, 20):], limit=1)
        assert (i[:, 0] >= 0).all()
    lk = hn._gg.links[:n].cpu().numpy().view(np.uint32)
    assert lk[:, 0].max() <= 16 and (lk[1:, 0] >= 1).all()
    d, i = hn.search_batch(x[:n], limit=1)
    codes = hn._gg.codes[:n].cpu().numpy()
    # a point's nearest stored row is itself or a row with the same code (distance ties go to the smaller id)
    same = [np.array_equal(codes[i[b, 0]], codes[b]) for b in range(n)]
    assert np.mean(same) > 0.97


@pytest.mark.parametrize('case', ['cluster_order', 'duplicates'])
def test_gpu_build_on_awkward_insertion_orders(ops, case):
    """Rows inserted in CLUSTER order (sorted along one latent direction: every batch lands next to the previous one, far from most
    of the graph -- and from most seeds) and a table where half of the rows are exact duplicates (symmetric distance 0 between
    candidates): the GPU-built graph answers like the host-built one."""
    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(21)
    N, D, M, B, k = 120_000, 64, 16, 128, 10
    A = rs.randn(8, D).astype(np.float32)
    z = rs.randn(N, 8).astype(np.float32)
    if case == 'cluster_order':
        z = z[np.argsort(z[:, 0])]
    x = (z @ A + 0.05 * rs.randn(N, D).astype(np.float32)).astype(np.float32)
    if case == 'duplicates':
        x[N // 2:] = x[rs.randint(0, N // 2, N - N // 2)]
    q = (rs.randn(B, 8).astype(np.float32) @ A + 0.05 * rs.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 3
    codec.fit(x[rs.choice(N, 8192, replace=False)], iter=8)
    flat = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
    flat.add_with_ids(x, np.arange(N))
    fd, fi = flat.search_batch(q, limit=k)
    got = {}
    for where in ('gpu', 'host'):
        hn = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=128, rerank=False, build=where)
        for c0 in range(0, N, 30_000):
            hn.add_with_ids(x[c0:c0 + 30_000], np.arange(c0, min(N, c0 + 30_000)))
        hd, hi = hn.search_batch(q, limit=k)
        # ids may differ where distances tie (duplicates): compare the DISTANCES found with the exhaustive scan's
        got[where] = float(np.mean(np.isclose(hd, fd, rtol=0, atol=0) | (hd <= fd)))
        if where == 'gpu':
            lk = hn._gg.links[:N].cpu().numpy().view(np.uint32)
            assert lk[:, 0].max() <= 32 and lk[:, 0].min() >= 1
    assert got['gpu'] >= 0.9 and got['gpu'] >= got['host'] - 0.03, got


def test_snapshots_across_the_two_builds(ops, tmp_path):
    """A snapshot of a host-built graph opened by an index whose default is the GPU build keeps being served (build switches to
    'host'); a dump without any graph file is rebuilt from the stored rows; a GPU-built snapshot opened with build='host' is
    refused with a message -- never an index that silently answers nothing."""
    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec

    rs = np.random.RandomState(8)
    N, D, M, B, k = 20_000, 64, 16, 32, 10
    gen = _data(rs, N, D)
    x, q = gen(N), gen(B)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 2
    codec.fit(x[:8192], iter=6)

    def mk(**kw):
        return HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, ef_search=96, rerank=False, **kw)

    host = mk(build='host')
    host.add_with_ids(x, np.arange(N))
    _, want = host.search_batch(q, limit=k)
    host.dump(tmp_path / 'h.idx')
    a = mk()
    assert a.build == 'gpu'
    a.load(tmp_path / 'h.idx')
    assert a.build == 'host'
    assert np.array_equal(a.search_batch(q, limit=k)[1], want)
    # no graph file at all: rebuilt (from the decoded rows here: no float vectors kept)
    (tmp_path / 'h.idx.graph').unlink()
    b = mk()
    b.load(tmp_path / 'h.idx')
    assert b.build == 'gpu' and b._gg.n == N
    got = b.search_batch(q, limit=k)[1]
    assert np.mean([len(set(got[i]) & set(want[i])) / k for i in range(B)]) >= 0.9
    b.dump(tmp_path / 'g.idx')
    assert (tmp_path / 'g.idx.level0.npy').exists() and not (tmp_path / 'g.idx.graph').exists()
    c = mk(build='host')
    with pytest.raises(RuntimeError, match='GPU-built'):
        c.load(tmp_path / 'g.idx')
    d = mk()
    d.load(tmp_path / 'g.idx')
    assert np.array_equal(d.search_batch(q, limit=k)[1], got)


def test_gpu_graph_for_64_sub_spaces(ops, oracle):
    """M = 64 (config 4's code width: 768-d text embeddings, cosine): 64 KB of table per wave -- one wave per CU --, graph built and walked
    on the GPU like the narrower codes; against the exhaustive scan of the same index and the oracle's PQLookup."""
    import torch

    from annlite_amd import HnswPQGpuIndex, Metric, PQCodec, PQFlatGpuIndex

    rs = np.random.RandomState(64)
    N, D, M, B, k = 30_000, 256, 64, 48, 10
    gen = _data(rs, N, D, r=24)
    x, q = gen(N), gen(B)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.COSINE, n_init=1)
    codec.seed = 6
    codec.fit(x[:8192], iter=6)
    flat = PQFlatGpuIndex(dim=D, metric=Metric.COSINE, pq_codec=codec, initial_size=N)
    flat.add_with_ids(x, np.arange(N))
    fd, fi = flat.search_batch(q, limit=k)
    hn = HnswPQGpuIndex(dim=D, metric=Metric.COSINE, pq_codec=codec, initial_size=N, ef_search=128, rerank=False)
    assert hn.build == 'gpu' and hn._gpu_walk_ok()
    hn.add_with_ids(x[:10_000], np.arange(10_000))
    hn.add_with_ids(x[10_000:], np.arange(10_000, N))
    hd, hi = hn.search_batch(q, limit=k)
    assert np.mean([len(set(hi[b]) & set(fi[b])) / k for b in range(B)]) >= 0.9
    for b in range(B):  # the ids both return carry the same (cosine) distance bits
        pos = {int(i): j for j, i in enumerate(fi[b])}
        for j, i in enumerate(hi[b]):
            if int(i) in pos:
                assert hd[b][j] == fd[b][pos[int(i)]]
    # pair walk == one at a time up to the last places; exact sums under the walk's L2 tables
    qd = hn._pre(torch.from_numpy(q).cuda())
    two = hn.candidates(qd, 128)
    hn.expand_width = 1
    one = hn.candidates(qd, 128)
    i1, i2 = one[0].cpu().numpy(), two[0].cpu().numpy()
    assert np.mean([len(np.intersect1d(i1[b], i2[b])) / 128 for b in range(B)]) >= 0.97
    from annlite_amd._capi import LAYOUT_BMK, LUT_L2
    lut = ops.lut_build(codec.scan_inputs(qd)[1], codec.codebooks_dev, LUT_L2, LAYOUT_BMK).cpu().numpy()
    codes = hn._gg.codes[:N].cpu().numpy()
    dd = two[1].cpu().numpy()
    for b in range(0, B, 11):
        ok = i2[b] >= 0
        assert np.array_equal(dd[b][ok], oracle.adc_gather_c(lut[b], codes, i2[b][ok]))
