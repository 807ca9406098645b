#!/usr/bin/env python3
"""Host time per batch of the graph index's search (config 5 shape at --rows): the python + enqueue time of a batch against the device time,
and a cProfile of 200 batches."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from annlite_amd import HnswPQGpuIndex, Metric, PQCodec  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--rows', type=int, default=1_000_000)
a = p.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, D, M, B, k = a.rows, 128, 16, 1024, 10
g = torch.Generator(device=dev)
g.manual_seed(99)
A = torch.randn((16, D), generator=g, device=dev)


def gen(n, seed):
    gg = torch.Generator(device=dev)
    gg.manual_seed(seed)
    return (torch.randn((n, 16), generator=gg, device=dev) @ A + 0.05 * torch.randn((n, D), generator=gg, device=dev)).contiguous()


codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
codec.seed = 7
x = gen(N, 1)
codec.fit(x[:20480], iter=10)
index = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True, ef_search=128)
index.add_with_ids(x, torch.arange(N, device=dev, dtype=torch.int64))
qs = [gen(B, 100 + j) for j in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for j in range(8):
    with torch.cuda.stream(streams[j % 2]):
        index.search_batch(qs[j % 2], limit=k)
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for j in range(n):
    with torch.cuda.stream(streams[j % 2]):
        index.search_batch(qs[j % 2], limit=k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('two streams: host enqueue %.4f ms per batch, whole %.4f ms per batch (%.0f q/s)' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, B * n / (t2 - t0)))
pr = cProfile.Profile()
pr.enable()
for j in range(200):
    with torch.cuda.stream(streams[j % 2]):
        index.search_batch(qs[j % 2], limit=k)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(18)
