"""Per-tile phase stamps and event counters of the cell-tile scan (annlite_ivf_search_topk under ANNLITE_DEBUG_COUNTERS=1), and a plain
loop of pruned searches for rocprofv3 --kernel-trace --stats.  `python scripts/prof_ivf_bytes.py [--rows N] [--probe P] [--loop R]`."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=10_000_000)
    p.add_argument('--cells', type=int, default=256)
    p.add_argument('--probe', type=int, default=16)
    p.add_argument('--loop', type=int, default=0)
    p.add_argument('--rerank', type=int, default=0, help='> 0: the float re-rank on the cell tiles (annlite_ivf_search_candidates + '
                   'annlite_rerank_topk, rerank_k 16) with this bound_rank instead of the plain pruned search')
    args = p.parse_args()
    from annlite_amd import Metric, PQCodec, _capi
    from annlite_amd.core.codec.vq import VQCodec
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    dev = torch.device('cuda', 0)
    N, D, M, Ks, B, k, C, P = args.rows, 128, 16, 256, 1024, 10, args.cells, args.probe
    gA = torch.Generator(device=dev)
    gA.manual_seed(99)
    A = torch.randn((16, D), generator=gA, device=dev)
    CH = 250_000
    train = bench.gen_chunk(0, CH, D, A, dev)[:100_000]
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.fit(train, iter=10)
    vq = VQCodec(C, metric=Metric.EUCLIDEAN, iter=15, n_init=1)
    vq.seed = 11
    vq.fit(train)
    idx = IvfPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, vq_codec=vq, initial_size=N, rerank=args.rerank > 0)
    idx.rerank_bound_rank = max(1, args.rerank)
    kw = dict(rerank_k=16) if args.rerank else {}
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        idx.add_with_ids(bench.gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev))
    idx._seal()
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = (torch.randn((B, 16), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()
    for _ in range(3):
        idx.search_batch(queries, limit=k, n_probe=P, **kw)
    torch.cuda.synchronize()
    if args.rerank:
        print('path:', idx.last_pruned_path, 'bound_rank', idx.rerank_bound_rank, flush=True)
    if args.loop:
        for _ in range(args.loop):
            idx.search_batch(queries, limit=k, n_probe=P, **kw)
        torch.cuda.synchronize()
        return
    idx.search_batch(queries, limit=k, n_probe=P, **kw)
    torch.cuda.synchronize()
    it = _capi.debug_items()
    cnt = _capi.debug_counters()
    t0 = it[:, 2].min()
    rows = (idx._cell_rows[:, 1] - idx._cell_rows[:, 0]).cpu().numpy()
    print(json.dumps({'items': int(it.shape[0]), 'counters[wave-steps with a candidate, pushed, exact sums, offered, consumer cycles, rebuilds, batches, wave0 wait]': cnt,
                      'cell_rows_min_mean_max': [int(rows.min()), float(rows.mean()), int(rows.max())]}))
    us = lambda v: (v - t0) / 100.0
    order = np.argsort(it[:, 2])
    print('tile  block  start_us  build_us  scan_us  wait_us  end_us(total)')
    sel = list(order[:12]) + list(order[len(order) // 2 - 4: len(order) // 2 + 4]) + list(order[-12:])
    for j in sel:
        r = it[j]
        print('%5d %5d %9.1f %8.1f %8.1f %8.1f %8.1f' % (r[0], r[7], us(r[2]), (r[3] - r[2]) / 100.0, (r[4] - r[3]) / 100.0, (r[5] - r[4]) / 100.0, (r[6] - r[2]) / 100.0))
    tot = (it[:, 6] - it[:, 2]) / 100.0
    print('per item us: mean %.1f  p50 %.1f  p90 %.1f  max %.1f; build mean %.1f scan mean %.1f wait mean %.1f; span %.1f us' % (
        tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), tot.max(), ((it[:, 3] - it[:, 2]) / 100.0).mean(),
        ((it[:, 4] - it[:, 3]) / 100.0).mean(), ((it[:, 5] - it[:, 4]) / 100.0).mean(), (it[:, 6].max() - t0) / 100.0))
    # by the block's n-th item
    for nth in range(4):
        m = np.array([j for j in range(it.shape[0]) if (it[:, 7] == it[j, 7]).sum() > nth and np.argsort(it[it[:, 7] == it[j, 7], 2]).tolist().index(
            int(np.where(np.where(it[:, 7] == it[j, 7])[0] == j)[0][0])) == nth])
        if m.size:
            print('item #%d of its block: n=%d mean total %.1f us (scan %.1f, wait %.1f)' % (nth, m.size, tot[m].mean(), ((it[m, 4] - it[m, 3]) / 100.0).mean(),
                                                                                          ((it[m, 5] - it[m, 4]) / 100.0).mean()))


if __name__ == '__main__':
    main()
