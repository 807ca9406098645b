#!/usr/bin/env python3
"""Config 5's graph walk ALONE, for counter passes: building the graph (a minute of host work at 5M rows, far longer under a
profiler) and walking it are two invocations, so that rocprofv3 only ever sees the second one.

    python scripts/prof_graph_walk.py --build /tmp/g5m --rows 5000000          # no profiler: fit, encode, build, dump
    rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- \
        python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000       # load, walk `--iters` times

Same vectors, codec settings, graph parameters and queries as scripts/bench_hnsw.py (the `c5` leg of bench.py).
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from annlite_amd import HnswPQGpuIndex, Metric, PQCodec, _capi, ops  # noqa: E402
from annlite_amd._capi import LAYOUT_BMK, LUT_L2  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--build', default=None)
p.add_argument('--walk', default=None)
p.add_argument('--rows', type=int, default=5_000_000)
p.add_argument('--dim', type=int, default=128)
p.add_argument('--m', type=int, default=16)
p.add_argument('--batch', type=int, default=1024)
p.add_argument('--ef-search', type=int, default=128)
p.add_argument('--ef-construction', type=int, default=200)
p.add_argument('--max-connection', type=int, default=16)
p.add_argument('--iters', type=int, default=4)
p.add_argument('--layout', choices=['packed', 'plain'], default='packed', help='node records with the neighbours\' code rows inline (round 5) / link lists + code table')
a = p.parse_args()
assert (a.build is None) != (a.walk is None), 'exactly one of --build / --walk'
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, D, M, B = a.rows, a.dim, a.m, a.batch
r_lat = 16 if D <= 128 else 64
g = torch.Generator(device=dev)
g.manual_seed(99)
A = torch.randn((r_lat, D), generator=g, device=dev)
CH = 250_000


def gen(chunk, rows):
    gg = torch.Generator(device=dev)
    gg.manual_seed(1234 + chunk)
    z = torch.randn((rows, r_lat), generator=gg, device=dev)
    e = torch.randn((rows, D), generator=gg, device=dev)
    return (z @ A + 0.05 * e).contiguous()


def new_index(codec):
    return HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=False, ef_search=a.ef_search,
                          ef_construction=a.ef_construction, max_connection=a.max_connection)


if a.build:
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.deterministic = True
    codec.fit(gen(0, CH)[:20480], iter=20)
    index = new_index(codec)
    t0 = time.time()
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        index.add_with_ids(gen(c, rows), torch.arange(c * CH, c * CH + rows, device=dev, dtype=torch.int64))
    torch.cuda.synchronize()
    print(f'built {N} rows in {time.time() - t0:.1f} s', flush=True)
    with open(a.build + '.codec', 'wb') as f:
        pickle.dump(codec, f)
    index.dump(a.build)
    print('dumped', {q: os.path.getsize(q) for q in (a.build, a.build + '.graph', a.build + '.level0.npy', a.build + '.codec') if os.path.exists(q)},
          'graph built on', index.build, flush=True)
    sys.exit(0)

with open(a.walk + '.codec', 'rb') as f:
    codec = pickle.load(f)
index = new_index(codec)
index.load(a.walk)
gq = torch.Generator(device=dev)
gq.manual_seed(4321)
q = (torch.randn((B, r_lat), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()
index.walk = 'gpu'
qd = index._pre(q)
_, xg = codec.scan_inputs(qd)
links, seeds = index._export_graph()
lut = ops.lut_build(xg, codec.codebooks_dev, LUT_L2, LAYOUT_BMK)
plain = index._plain_table(index._n_rows)
lpn = links.shape[1] - 1
packed = index._packed_records(links, plain) if a.layout == 'packed' else None


def walk():
    if packed is not None:
        return ops.graph_search_packed(packed, lpn, seeds, plain, lut, a.ef_search, valid_bits=index._valid, n_rows=index._n_rows,
                                       expand_width=index.expand_width if lpn <= 32 else 1)
    return ops.graph_search(links, seeds, plain, lut, a.ef_search, valid_bits=index._valid, n_rows=index._n_rows)


os.environ['ANNLITE_DEBUG_COUNTERS'] = '1'
_capi.knobs_reload()  # (the library parses its switches at load)
walk()
n_expand, n_eval, n_hit = _capi.graph_search_stats_ex()
del os.environ['ANNLITE_DEBUG_COUNTERS']
_capi.knobs_reload()
kms = []
for _ in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    walk()
    e1.record()
    e1.synchronize()
    kms.append(e0.elapsed_time(e1))
lpn = links.shape[1] - 1
alg = n_expand * 4.0 * (lpn + 1) + n_eval * float(M) + B * seeds.numel() * float(M)
print(f'graph walk: {index._n_rows} rows, batch {B}, ef_search {a.ef_search}: kernel ms {np.round(kms, 3).tolist()}; '
      f'expansions/query {n_expand / B:.1f}, rows evaluated/query {n_eval / B:.1f}, algorithmic bytes/launch {alg:.4g}; layout {a.layout}, '
      f'graph built on {index.build}, expand_width {index.expand_width if (packed is not None and lpn <= 32) else 1}, seeds {seeds.numel()}'
      + (f', record {packed.shape[1]} B, bytes read by design {n_expand * float(packed.shape[1]) + B * seeds.numel() * float(M):.4g}, '
         f'prefetched records used {n_hit / max(n_expand, 1):.3f}' if packed is not None else ''), flush=True)
