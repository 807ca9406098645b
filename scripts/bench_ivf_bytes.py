"""Pruned search over cells: the byte-table cell tiles (annlite_ivf_search_topk, round 6) beside the u16 tile scan + re-score they
replace, at the headline bench's shape.  `python scripts/bench_ivf_bytes.py [--rows 10000000] [--cells 256] [--probes 8,16,32]`.
Prints one JSON line per probe count: ms per 1024-query batch and queries/s of both paths (one stream / two caller streams), whether
the two paths agree bit for bit, the agreement with the exhaustive ADC top-k, and the new path's stages."""
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
    p.add_argument('--probes', default='8,16,32')
    p.add_argument('--reps', type=int, default=20)
    args = p.parse_args()
    from annlite_amd import Metric, PQCodec, ops, _capi
    from annlite_amd.core.codec.vq import VQCodec
    from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

    dev = torch.device('cuda', 0)
    N, D, M, Ks, B, k, C = args.rows, 128, 16, 256, args.batch, args.k, args.cells
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
    idx = IvfPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, vq_codec=vq, initial_size=N, rerank=False)
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        idx.add_with_ids(bench.gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev))
    idx._seal()
    torch.cuda.synchronize()
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = (torch.randn((B, 16), generator=gq, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq, device=dev)).contiguous()

    def timed(fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r

    def two_streams(fn, reps=args.reps):
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                fn()
                fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(2 * reps):
            with torch.cuda.stream(streams[it & 1]):
                fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * reps) * 1e3

    ms_flat, r_flat = timed(lambda: idx.search_batch(queries, limit=k, n_probe=C), 5)
    adc = r_flat[1].cpu().numpy()
    print(json.dumps({'rows': N, 'cells': C, 'batch': B, 'k': k, 'exhaustive_ms': ms_flat, 'exhaustive_qps': B / ms_flat * 1e3}), flush=True)
    for P in [int(v) for v in args.probes.split(',')]:
        rec = {'n_probe': P}
        res = {}
        for name, flag in (('bytes', True), ('u16', False)):
            idx.byte_tiles = flag
            ms, r = timed(lambda: idx.search_batch(queries, limit=k, n_probe=P))
            ms2 = two_streams(lambda: idx.search_batch(queries, limit=k, n_probe=P))
            res[name] = r
            rec[name] = {'ms': round(ms, 4), 'qps': round(B / ms * 1e3), 'two_streams_ms': round(ms2, 4), 'two_streams_qps': round(B / ms2 * 1e3)}
        rec['paths_bit_equal'] = bool(torch.equal(res['bytes'][0], res['u16'][0]) and torch.equal(res['bytes'][1], res['u16'][1]))
        got = res['bytes'][1].cpu().numpy()
        rec['agreement_with_exhaustive_adc_topk'] = float(np.mean([len(set(got[b]) & set(adc[b])) / k for b in range(B)]))
        # stages of the new path
        idx.byte_tiles = True
        q = idx._pre(queries)
        st = {}
        st['select_ms'], cells = timed(lambda: idx.probe_cells(q, P))
        st['search_topk_ms'], _ = timed(lambda: ops.ivf_search_topk(_capi.LUT_L2, q, codec.codebooks_dev, idx._table, cells, C, idx._cell_rows, idx._cell_order,
                                                                   k, M, Ks, row_ids=idx._row_ids, n_rows=idx._n_table,
                                                                   codes_layout=_capi.CODES_SKEWED, sqrt=True, workspace=idx._tws))
        _capi.profile_enable(True)
        ops.ivf_search_topk(_capi.LUT_L2, q, codec.codebooks_dev, idx._table, cells, C, idx._cell_rows, idx._cell_order, k, M, Ks, row_ids=idx._row_ids,
                            n_rows=idx._n_table, codes_layout=_capi.CODES_SKEWED, sqrt=True, workspace=idx._tws)
        st['scan_kernel_ms'] = _capi.profile_last_scan_ms()
        _capi.profile_enable(False)
        rec['stages'] = {a: round(b, 4) for a, b in st.items()}
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
