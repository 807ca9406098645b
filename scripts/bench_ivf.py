"""Pruned (IVF) PQ search at the headline bench's shape: queries/s, recall@10 and time per stage for several
(n_cells, n_probe).  `python scripts/bench_ivf.py [--rows 10000000] [--cells 256] [--probes 4,8,16,32]`."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=10_000_000)
    p.add_argument('--batch', type=int, default=1024)
    p.add_argument('--k', type=int, default=10)
    p.add_argument('--cells', type=int, default=256)
    p.add_argument('--probes', default='4,8,16,32')
    p.add_argument('--truth-queries', type=int, default=256)
    p.add_argument('--reps', type=int, default=10)
    p.add_argument('--no-rerank', action='store_true')
    p.add_argument('--dim', type=int, default=128)
    p.add_argument('--m', type=int, default=16)
    p.add_argument('--metric', choices=['euclidean', 'cosine', 'inner_product'], default='euclidean')
    args = p.parse_args()
    from annlite_amd import Metric, PQCodec, ops
    from annlite_amd import _capi
    from annlite_amd._capi import CODES_SKEWED, scan_plan, scan_plan_tiles
    from annlite_amd.core.codec.vq import VQCodec
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    dev = torch.device('cuda', 0)
    N, D, M, Ks, B, k, C = args.rows, args.dim, args.m, 256, args.batch, args.k, args.cells
    metric = {'euclidean': Metric.EUCLIDEAN, 'cosine': Metric.COSINE, 'inner_product': Metric.INNER_PRODUCT}[args.metric]
    r_lat = 16 if D <= 128 else 64
    gA = torch.Generator(device=dev)
    gA.manual_seed(99)
    A = torch.randn((r_lat, D), generator=gA, device=dev)
    CH = 250_000
    train = bench.gen_chunk(0, CH, D, A, dev)[:100_000]
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=metric, n_init=1)
    codec.seed = 7
    codec.fit(train, iter=10)
    vq = VQCodec(C, metric=metric, iter=15, n_init=1)
    vq.seed = 11
    t0 = time.time()
    vq.fit(train)
    vq_s = time.time() - t0
    idx = IvfPQGpuIndex(dim=D, metric=metric, pq_codec=codec, vq_codec=vq, initial_size=N,
                        rerank=not args.no_rerank)
    t0 = time.time()
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        idx.add_with_ids(bench.gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev))
    torch.cuda.synchronize()
    add_s = time.time() - t0
    t0 = time.time()
    idx._seal()
    torch.cuda.synchronize()
    seal_s = time.time() - t0
    counts = (idx._cell_rows[:, 1] - idx._cell_rows[:, 0]).cpu().numpy()

    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    zq = torch.randn((B, r_lat), generator=gq, device=dev)
    eq = torch.randn((B, D), generator=gq, device=dev)
    queries = (zq @ A + 0.05 * eq).contiguous()

    nq = min(args.truth_queries, B)
    qs = queries[:nq]
    qn = (qs * qs).sum(1)[:, None]
    best_d = torch.full((nq, k), float('inf'), device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    if idx._vectors is not None:
        for c in range((N + CH - 1) // CH):
            x = idx._vectors[c * CH: min(N, (c + 1) * CH)]
            dd = qn + (x * x).sum(1)[None, :] - 2.0 * (qs @ x.T)
            cd, ci = torch.topk(dd, k, dim=1, largest=False)
            md, mi = torch.cat([best_d, cd], 1), torch.cat([best_i, ci + c * CH], 1)
            o = torch.argsort(md, dim=1)[:, :k]
            best_d, best_i = torch.gather(md, 1, o), torch.gather(mi, 1, o)
    truth = best_i.cpu().numpy()

    def recall(ids, ref):
        got = ids[:nq].cpu().numpy() if isinstance(ids, torch.Tensor) else ids[:nq]
        return float(np.mean([len(set(got[b]) & set(ref[b])) / k for b in range(nq)]))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps * 1e3, r

    idx.rerank = False
    ms_flat, r_flat = timed(lambda: idx.search_batch(queries, limit=k, n_probe=C))
    adc_truth = r_flat[1].cpu().numpy()
    out = {'rows': N, 'batch': B, 'k': k, 'n_cells': C, 'vq_fit_s': vq_s, 'add_s': add_s, 'seal_s': seal_s,
           'cell_rows_min_mean_max': [int(counts.min()), float(counts.mean()), int(counts.max())],
           'exhaustive': {'ms': ms_flat, 'qps': B / ms_flat * 1e3,
                          'recall_vs_exact': recall(r_flat[1], truth) if idx._vectors is not None else None}}
    print(json.dumps(out), flush=True)
    for P in [int(v) for v in args.probes.split(',')]:
        idx.rerank = False
        ms, r = timed(lambda: idx.search_batch(queries, limit=k, n_probe=P))
        q = idx._pre(queries)
        st = {}
        st['select'], cells = timed(lambda: idx.probe_cells(q, P))
        qt = scan_plan_tiles(idx._n_table, M, Ks, 1, 16, k).qt
        st['plan'], (vmap, slot_of, tile_rows, used) = timed(lambda: ops.ivf_plan(cells, C, qt, idx._cell_rows, idx._cell_order))
        kind, xq = codec.scan_inputs(q)
        st['tables_scan'], (cand, count) = timed(lambda: ops.pq_search_tiles(
            kind, xq, codec.codebooks_dev, idx._table, k, M, Ks, tile_rows, vmap, n_rows=idx._n_table,
            codes_layout=CODES_SKEWED, workspace=idx._tws, cand_cap=idx.cand_cap))
        _capi.profile_enable(True)
        ops.pq_search_tiles(kind, xq, codec.codebooks_dev, idx._table, k, M, Ks, tile_rows, vmap, n_rows=idx._n_table,
                            codes_layout=CODES_SKEWED, workspace=idx._tws, cand_cap=idx.cand_cap)
        st['scan_kernel'] = _capi.profile_last_scan_ms()
        _capi.profile_enable(False)
        if os.environ.get('ANNLITE_DEBUG_COUNTERS'):
            st['counters'] = _capi.debug_counters()[:5]
        st['lut_real'], lut = timed(lambda: codec.get_dist_mat(q))
        st['rescore'], _ = timed(lambda: ops.ivf_rescore(lut, idx._table_plain, cand, count, slot_of, tile_rows, qt, k,
                                                         idx._row_ids, sqrt=True))
        cnt = count[vmap >= 0].to(torch.int64)
        st['cand_per_slot_mean_max_overflow'] = [float(cnt[cnt >= 0].float().mean().item()), int(cnt.max().item()),
                                                 int((cnt < 0).sum().item())]
        # throughput with consecutive batches alternating between two streams (one batch's single-workgroup plan /
        # selection / re-score launches run beside the other's tile scan)
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st_):
                idx.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(2 * args.reps):
            with torch.cuda.stream(streams[it & 1]):
                idx.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / (2 * args.reps) * 1e3
        st['two_streams_ms_per_batch'] = ms2
        rec = {'n_probe': P, 'ms': ms, 'qps': B / ms * 1e3, 'tiles_used': int(used.item()), 'slots': int(vmap.numel()),
               'recall_vs_exhaustive_adc': recall(r[1], adc_truth),
               'recall_vs_exact': recall(r[1], truth) if idx._vectors is not None else None, 'stages_ms': st}
        if idx._vectors is not None:
            idx.rerank = True
            for rk in (10, 16, 32):
                ms_r, rr = timed(lambda: idx.search_batch(queries, limit=k, n_probe=P, rerank_k=rk))
                rec[f'rerank{rk}'] = {'ms': ms_r, 'qps': B / ms_r * 1e3, 'recall_vs_exact': recall(rr[1], truth)}
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
