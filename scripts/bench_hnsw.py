#!/usr/bin/env python3
"""BASELINE config 5: HNSW-over-PQ, ef_search=128, GPU ADC / exact re-rank of the candidate lists,
recall@10 vs exact brute force -- next to the exhaustive GPU scan over the same codes.

    python scripts/bench_hnsw.py --rows 5000000        # one JSON line on stdout
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from annlite_amd import HnswPQGpuIndex, Metric, PQCodec  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--rows', type=int, default=1_000_000)
p.add_argument('--dim', type=int, default=128)
p.add_argument('--m', type=int, default=16)
p.add_argument('--batch', type=int, default=1024)
p.add_argument('--k', type=int, default=10)
p.add_argument('--ef-search', type=int, default=128)
p.add_argument('--ef-construction', type=int, default=200)
p.add_argument('--max-connection', type=int, default=16)
p.add_argument('--steps', type=int, default=5)
a = p.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, D, M, B, k = a.rows, a.dim, a.m, a.batch, a.k
r_lat = 16 if D <= 128 else 64
g = torch.Generator(device=dev)
g.manual_seed(99)
A = torch.randn((r_lat, D), generator=g, device=dev)


def gen(chunk, rows):
    gg = torch.Generator(device=dev)
    gg.manual_seed(1234 + chunk)
    z = torch.randn((rows, r_lat), generator=gg, device=dev)
    e = torch.randn((rows, D), generator=gg, device=dev)
    return (z @ A + 0.05 * e).contiguous()


codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
codec.seed = 7
codec.fit(gen(0, 250_000)[:20480], iter=20)
index = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True,
                       ef_search=a.ef_search, ef_construction=a.ef_construction, max_connection=a.max_connection)
CH = 250_000
t0 = time.time()
for c in range((N + CH - 1) // CH):
    rows = min(CH, N - c * CH)
    index.add_with_ids(gen(c, rows), torch.arange(c * CH, c * CH + rows, device=dev, dtype=torch.int64))
torch.cuda.synchronize()
build_s = time.time() - t0

gq = torch.Generator(device=dev)
gq.manual_seed(4321)
q = (torch.randn((B, r_lat), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()

# exact truth
best_d = torch.full((B, k), float('inf'), device=dev)
best_i = torch.full((B, k), -1, dtype=torch.int64, device=dev)
qn = (q * q).sum(1)[:, None]
for c in range((N + CH - 1) // CH):
    rows = min(CH, N - c * CH)
    x = gen(c, rows)
    dd = qn + (x * x).sum(1)[None, :] - 2.0 * (q @ x.T)
    cd, ci = torch.topk(dd, k, dim=1, largest=False)
    md, mi = torch.cat([best_d, cd], 1), torch.cat([best_i, ci + c * CH], 1)
    o = torch.argsort(md, dim=1)[:, :k]
    best_d, best_i = torch.gather(md, 1, o), torch.gather(mi, 1, o)
truth = best_i.cpu().numpy()


def recall(ids):
    ids = ids.cpu().numpy()
    return float(np.mean([len(set(ids[b]) & set(truth[b])) / k for b in range(B)]))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        out = fn()
    torch.cuda.synchronize()
    return out, B * a.steps / (time.perf_counter() - t)


res = {}
for name, rerank, graph, walk in (('hnsw_gpu_walk_adc', False, True, 'gpu'), ('hnsw_gpu_walk_exact_rerank', True, True, 'gpu'),
                                  ('hnsw_host_walk_adc', False, True, 'host'), ('hnsw_host_walk_exact_rerank', True, True, 'host'),
                                  ('exhaustive_adc', False, False, None), ('exhaustive_exact_rerank', True, False, None)):
    index.rerank = rerank
    if walk:
        index.walk = walk
    fn = (lambda: index.search_batch(q, limit=k)) if graph else (lambda: index.search_exhaustive(q, limit=k))
    (d, i), qps = timed(fn)
    res[name] = {'queries_per_s': qps, 'recall_at_10': recall(i)}
# the walks alone
walks = {}
for walk in ('gpu', 'host'):
    index.walk = walk
    index.candidates(q, a.ef_search)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        index.candidates(q, a.ef_search)
    torch.cuda.synchronize()
    walks[walk] = B * a.steps / (time.perf_counter() - t)
walk_qps = walks
print(json.dumps({'config': f'HNSW-over-PQ: {N} x {D}-dim, PQ m={M} ks=256, L2, max_connection={a.max_connection}, '
                            f'ef_construction={a.ef_construction}, ef_search={a.ef_search}, batch {B}, k={k}',
                  'build_s': build_s, 'build_rows_per_s': N / build_s, 'host_cpus_reported': os.cpu_count(), 'note': 'the graph library starts min(CPUs, affinity, cgroup quota) threads',
                  'graph_walk_queries_per_s': walk_qps, **res}))
