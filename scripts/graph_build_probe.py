#!/usr/bin/env python3
"""Round 6 probe: the GPU-built level-0 graph at --rows -- build time by phase, degree statistics, and the number of seeds the walk scans
(GpuLevel0Graph.MAX_SEEDS) against walk time and recall.  One line per setting on stdout."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from annlite_amd import HnswPQGpuIndex, Metric, PQCodec  # noqa: E402
from annlite_amd.core.index import graph_gpu_build as gb  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--rows', type=int, default=5_000_000)
p.add_argument('--seeds', default='32,64,128,256,1024')
p.add_argument('--build-seeds', type=int, default=0, help='MAX_SEEDS while building (0: the class default)')
p.add_argument('--batch', type=int, default=0, help='GpuLevel0Graph.BATCH (0: the class default)')
p.add_argument('--ef', default='128', help='comma list of ef_search values (each with the default visited table and, beyond 128, with 4096 entries)')
a = p.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, D, M, B, k = a.rows, 128, 16, 1024, 10
g = torch.Generator(device=dev)
g.manual_seed(99)
A = torch.randn((16, D), generator=g, device=dev)


def gen(chunk, rows):
    gg = torch.Generator(device=dev)
    gg.manual_seed(1234 + chunk)
    return (torch.randn((rows, 16), generator=gg, device=dev) @ A + 0.05 * torch.randn((rows, D), generator=gg, device=dev)).contiguous()


codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
codec.seed = 7
codec.deterministic = True
codec.fit(gen(0, 250_000)[:20480], iter=20)
if a.build_seeds:
    gb.GpuLevel0Graph.MAX_SEEDS = a.build_seeds
if a.batch:
    gb.GpuLevel0Graph.BATCH = a.batch
# phase timing: wrap the ops the builder calls
phase = {}
from annlite_amd import ops  # noqa: E402


def wrap(name):
    fn = getattr(ops, name)

    def w(*args, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        phase[name] = phase.get(name, 0.0) + time.perf_counter() - t
        return out

    setattr(ops, name, w)


if os.environ.get('PROBE_PHASES'):
    for nm in ('lut_build', 'graph_search_packed', 'graph_build_select', 'graph_build_reverse', 'graph_pack_nodes', 'pq_encode'):
        wrap(nm)
index = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True, ef_search=128, build='gpu')
CH = 250_000
torch.cuda.synchronize()
t0 = time.time()
for c in range((N + CH - 1) // CH):
    rows = min(CH, N - c * CH)
    index.add_with_ids(gen(c, rows), torch.arange(c * CH, c * CH + rows, device=dev, dtype=torch.int64))
torch.cuda.synchronize()
print('build_s %.2f' % (time.time() - t0), {kk: round(v, 3) for kk, v in phase.items()}, flush=True)
cnt = index._gg.links[:N, 0].to(torch.int64)
print('degree: mean %.2f min %d max %d; share of full lists %.3f' % (cnt.float().mean().item(), cnt.min().item(), cnt.max().item(),
                                                                     (cnt == index._gg.lpn).float().mean().item()), flush=True)
gq = torch.Generator(device=dev)
gq.manual_seed(4321)
q = (torch.randn((B, 16), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()
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
from annlite_amd import _capi  # noqa: E402

for S in [int(v) for v in a.seeds.split(',')]:
    index._gg.MAX_SEEDS = S
    index._gg._seeds = None
    for ef in [int(v) for v in a.ef.split(',')]:
        for hb in ([None] if ef <= 128 else [None, 12]):
            if hb is None:
                os.environ.pop('ANNLITE_GRAPH_HASH_BITS', None)
            else:
                os.environ['ANNLITE_GRAPH_HASH_BITS'] = str(hb)
            _capi.knobs_reload()
            index.ef_search = ef
            qd = index._pre(q)
            for _ in range(2):
                index.candidates(qd, ef)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                index.candidates(qd, ef)
            torch.cuda.synchronize()
            walk_ms = (time.perf_counter() - t) / 10 * 1e3
            d, i = index.search_batch(q, limit=k)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                d, i = index.search_batch(q, limit=k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 10 * 1e3
            ids = i.cpu().numpy()
            rec = float(np.mean([len(set(ids[b]) & set(truth[b])) / k for b in range(B)]))
            print('seeds %4d ef %3d visited table %s: candidates() %.4f ms, search_batch %.4f ms = %.0f q/s, recall@10 %.4f' % (
                S, ef, 'default' if hb is None else '2^%d' % hb, walk_ms, ms, B / ms * 1e3, rec), flush=True)
os.environ.pop('ANNLITE_GRAPH_HASH_BITS', None)
