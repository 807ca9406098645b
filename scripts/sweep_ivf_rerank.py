"""The float re-rank of a pruned search on the byte-table cell tiles (annlite_ivf_search_candidates + annlite_rerank_topk): pool shape against
rate and recall@10 -- bound_rank x (nearest cells split, parts) -- at the headline's rows.  `python scripts/sweep_ivf_rerank.py [--rows N]`."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=10_000_000)
    p.add_argument('--cells', type=int, default=256)
    p.add_argument('--probe', type=int, default=16)
    p.add_argument('--reps', type=int, default=20)
    p.add_argument('--configs', default='2:0x1,2:1x4,2:2x4,2:4x4,2:2x8,1:2x4,4:2x4,2:16x2,1:4x4')
    args = p.parse_args()
    from annlite_amd import Metric, PQCodec
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
    idx = IvfPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, vq_codec=vq, initial_size=N, rerank=True)
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        idx.add_with_ids(bench.gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev))
    idx._seal()
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = (torch.randn((B, 16), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()
    # exact top-10 of the first nq queries over the stored vectors
    nq = 256
    best_d = torch.full((nq, k), float('inf'), device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    for lo in range(0, N, 1_000_000):
        xs = idx._vectors[lo:lo + 1_000_000]
        d = (xs * xs).sum(1)[None, :] - 2.0 * queries[:nq] @ xs.T
        dd, ii = torch.topk(d, k, dim=1, largest=False)
        alld, alli = torch.cat([best_d, dd], 1), torch.cat([best_i, ii + lo], 1)
        sel = torch.topk(alld, k, dim=1, largest=False).indices
        best_d, best_i = torch.gather(alld, 1, sel), torch.gather(alli, 1, sel)
    truth = best_i.cpu().numpy()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(**kw):
        for j in range(4):
            with torch.cuda.stream(streams[j % 2]):
                out = idx.search_batch(queries, limit=k, n_probe=P, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(2 * args.reps):
            with torch.cuda.stream(streams[j % 2]):
                out = idx.search_batch(queries, limit=k, n_probe=P, **kw)
        torch.cuda.synchronize()
        qps = B * 2 * args.reps / (time.perf_counter() - t0)
        got = out[1][:nq].cpu().numpy()
        rec = sum(len(set(got[b]) & set(truth[b])) for b in range(nq)) / (nq * k)
        return qps, rec

    for cfg in args.configs.split(','):
        rank, sp = cfg.split(':')
        idx.rerank_seed_whole_cell = not sp.endswith('p')  # (suffix p: the first bound from the nearest cell's first PART)
        n, S = sp.rstrip('p').split('x')
        idx.rerank_bound_rank, idx.rerank_split = int(rank), (int(n), int(S))
        qps, rec = run(rerank_k=16)
        print(json.dumps({'bound_rank': int(rank), 'split': [int(n), int(S)], 'qps_two_streams': round(qps), 'recall_at_10': round(rec, 4),
                          'seed': 'whole cell' if idx.rerank_seed_whole_cell else 'first part', 'path': idx.last_pruned_path}), flush=True)
    qps, rec = run(rerank_k=32)
    print(json.dumps({'rerank_k': 32, 'qps_two_streams': round(qps), 'recall_at_10': round(rec, 4), 'path': idx.last_pruned_path}), flush=True)


if __name__ == '__main__':
    main()
