import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import pq_oracle
from annlite_amd import Metric, PQCodec, ops
from annlite_amd.core.codec.vq import VQCodec
from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex
from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex

def run(N, D, M, C, P, B, k, metric, seed=0):
    rng = np.random.RandomState(seed)
    A = rng.randn(8, D).astype(np.float32)
    x = (rng.randn(N, 8).astype(np.float32) @ A + 0.1 * rng.randn(N, D).astype(np.float32)).astype(np.float32)
    q = (rng.randn(B, 8).astype(np.float32) @ A + 0.1 * rng.randn(B, D).astype(np.float32)).astype(np.float32)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=metric, n_init=1); codec.seed = 1
    codec.fit(x[:4096], iter=5)
    vq = VQCodec(C, metric=metric, iter=10, n_init=1); vq.seed = 2
    vq.fit(x[:4096])
    idx = IvfPQGpuIndex(dim=D, metric=metric, pq_codec=codec, vq_codec=vq, n_probe=P, initial_size=N)
    idx.add_with_ids(x, np.arange(N))
    d, i = idx.search_batch(q, limit=k)
    torch.cuda.synchronize()
    # oracle
    om = {Metric.EUCLIDEAN: pq_oracle.EUCLIDEAN, Metric.COSINE: pq_oracle.COSINE, Metric.INNER_PRODUCT: pq_oracle.INNER_PRODUCT}[metric]
    cells_of = pq_oracle.assign_cells(x, vq.codebook)
    got_cells = idx._cell_of[:N].cpu().numpy()
    print('cell assignment equal:', np.array_equal(cells_of, got_cells), 'mismatch', int((cells_of != got_cells).sum()))
    xin = x / np.linalg.norm(x, axis=1, keepdims=True) if False else x
    qq = pq_oracle.l2_normalize(q) if metric == Metric.COSINE else q
    kind = 0 if metric == Metric.EUCLIDEAN else 1
    cent = vq.codebook
    if metric == Metric.COSINE:
        cent = ops.l2_normalize(ops.to_dev(cent)).cpu().numpy()
    pc = pq_oracle.select_cells(qq, cent, kind, P)
    gpc = idx.probe_cells(idx._pre(q), P).cpu().numpy()
    print('probe cells equal:', np.array_equal(pc, gpc))
    codes = ops.codes_to_numpy(idx._plain_codes(N))
    xo = pq_oracle.l2_normalize(x) if metric == Metric.COSINE else x
    od, oi = pq_oracle.ivf_search(q, codec.codebooks, codes, got_cells, gpc, om, k)
    print(metric, 'ids equal', np.array_equal(oi, i), 'dist equal', np.array_equal(od, d) if metric == Metric.EUCLIDEAN else np.allclose(od, d, rtol=1e-4, atol=1e-6))
    if not np.array_equal(oi, i):
        bad = np.nonzero((oi != i).any(1))[0]
        print('bad queries', bad[:10], 'of', B)
        b = bad[0]; print(oi[b], i[b]); print(od[b], d[b])
    # all cells == flat
    d2, i2 = idx.search_batch(q, limit=k, n_probe=C)
    flat = PQFlatGpuIndex(dim=D, metric=metric, pq_codec=codec, initial_size=N)
    flat.add_with_ids(x, np.arange(N))
    d3, i3 = flat.search_batch(q, limit=k)
    print('all-cells == flat:', np.array_equal(i2, i3), np.array_equal(d2, d3))
    # forced tiles with P = C-1?  full probe through tiles
    idx.n_probe = None
    return idx

run(20000, 64, 16, 32, 4, 100, 10, Metric.EUCLIDEAN)
run(20000, 64, 16, 32, 4, 37, 10, Metric.COSINE)
run(30000, 64, 8, 16, 3, 64, 5, Metric.INNER_PRODUCT)
run(20000, 128, 32, 32, 4, 50, 10, Metric.EUCLIDEAN)
run(20000, 768, 64, 16, 4, 20, 10, Metric.COSINE)
