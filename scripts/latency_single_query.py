#!/usr/bin/env python3
"""Single-query latency of the index plugin's reference-signature search() (numpy in, numpy out) at 10M rows."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annlite_amd import Metric, PQCodec, PQFlatGpuIndex  # noqa: E402

dev = torch.device('cuda', 0)
N, D, M = int(os.environ.get('ROWS', 10_000_000)), 128, 16
g = torch.Generator(device=dev)
g.manual_seed(0)
codec = PQCodec(dim=D, n_subvectors=M, n_clusters=256, metric=Metric.EUCLIDEAN, n_init=1)
codec.seed = 1
codec.fit(torch.randn((20480, D), generator=g, device=dev), iter=5)
index = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N)
CH = 1_000_000
for c in range(N // CH):
    index.add_with_ids(torch.randn((CH, D), generator=g, device=dev), torch.arange(c * CH, (c + 1) * CH, device=dev))
q = np.random.RandomState(0).randn(100, D).astype(np.float32)
for i in range(5):
    index.search(q[i], limit=10)
t = time.perf_counter()
for i in range(100):
    d, ids = index.search(q[i], limit=10)
dt = (time.perf_counter() - t) / 100
print('search(x[D], limit=10) over %d rows: %.3f ms per call (numpy in -> numpy out), %d ids' % (N, dt * 1e3, len(ids)))
