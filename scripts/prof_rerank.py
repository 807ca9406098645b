"""Exact re-rank leg at the bench shape: time per stage and recall@k for several candidate
counts.  `python scripts/prof_rerank.py [--rows 10000000] [--rk 8,16,32,64]` on a GPU box."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (data generator of the headline bench)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=10_000_000)
    p.add_argument('--batch', type=int, default=1024)
    p.add_argument('--k', type=int, default=10)
    p.add_argument('--rk', default='8,16,24,32,48,64')
    p.add_argument('--truth-queries', type=int, default=256)
    p.add_argument('--reps', type=int, default=5)
    args = p.parse_args()
    from annlite_amd import Metric, PQCodec, ops
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex, scan_plan

    dev = torch.device('cuda', 0)
    N, D, M, Ks, B, k = args.rows, 128, 16, 256, args.batch, args.k
    gA = torch.Generator(device=dev)
    gA.manual_seed(99)
    A = torch.randn((16, D), generator=gA, device=dev)
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    CH = 250_000
    codec.fit(bench.gen_chunk(0, CH, D, A, dev)[:100_000], iter=10)
    index = PQFlatGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True, skewed=True)
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        index.add_with_ids(bench.gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev))
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    zq = torch.randn((B, 16), generator=gq, device=dev)
    eq = torch.randn((B, D), generator=gq, device=dev)
    queries = (zq @ A + 0.05 * eq).contiguous()

    nq = min(args.truth_queries, B)
    qs = queries[:nq]
    qn = (qs * qs).sum(1)[:, None]
    best_d = torch.full((nq, k), float('inf'), device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        x = index._vectors[c * CH: c * CH + rows]
        dd = qn + (x * x).sum(1)[None, :] - 2.0 * (qs @ x.T)
        cd, ci = torch.topk(dd, k, dim=1, largest=False)
        md, mi = torch.cat([best_d, cd], 1), torch.cat([best_i, ci + c * CH], 1)
        o = torch.argsort(md, dim=1)[:, :k]
        best_d, best_i = torch.gather(md, 1, o), torch.gather(mi, 1, o)
    truth = best_i.cpu().numpy()

    def recall(ids):
        got = ids[:nq].cpu().numpy()
        return float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps * 1e3, r

    index.rerank = False
    ms, r = timed(lambda: index.search_batch(queries, limit=k))
    out = {'rows': N, 'batch': B, 'k': k, 'adc_only': {'ms': ms, 'recall': recall(r[1])}, 'rerank': []}
    index.rerank = True
    N_ = index._n_rows
    for rk in [int(v) for v in args.rk.split(',')]:
        ms, r = timed(lambda: index.search_batch(queries, limit=k, rerank_k=rk))
        plan = scan_plan(N_, M, Ks, 1, B, rk)
        q = index._pre(queries)
        st = {}
        st['lut'], lut = timed(lambda: codec.get_dist_mat_tiled(q, plan.qi))
        st['scan_candidates'], (_, cand) = timed(lambda: ops.adc_scan_candidates(
            index._codes, lut, B, rk, M, Ks, valid_bits=index._valid, n_rows=N_, codes_layout=index._layout(), workspace=index._ws))
        st['exact_gather'], exact = timed(lambda: ops.exact_gather_dist(int(index.metric), q, index._vectors, cand))
        st['topk_rows'], _ = timed(lambda: ops.topk_rows(exact, k))
        out['rerank'].append({'rk': rk, 'candidates': int(cand.shape[1]), 'ms': ms, 'qps': B / ms * 1e3,
                              'recall': recall(r[1]), 'stages_ms': st})
        print(json.dumps(out['rerank'][-1]), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
