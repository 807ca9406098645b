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
p.add_argument('--streams', type=int, default=2,
               help='caller streams the timed GPU-walk batches alternate between (2: as the other legs of bench.py -- a walk launch lasts as '
                    'long as its slowest query, the next batch fills the SIMDs the others have left; 1: one batch at a time)')
p.add_argument('--gpu-build-batch', type=int, default=0, help='GpuLevel0Graph.BATCH (0: the class default)')
p.add_argument('--gpu-build-grow', type=int, default=0, help='GpuLevel0Graph.GROW (0: the class default)')
p.add_argument('--gpu-seeds', type=int, default=0, help='GpuLevel0Graph.MAX_SEEDS, build and search (0: the class default)')
p.add_argument('--build', choices=['both', 'gpu', 'host'], default='both',
               help="where the graph is built: 'gpu' = level 0 in batches on the GPU (round 6), 'host' = libannlite_graph.so; 'both' "
                    "(default): the GPU-built graph is what is measured, the host-built one serves the host walks / the CPU baseline "
                    "and the graph-quality comparison")
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
codec.deterministic = True
codec.fit(gen(0, 250_000)[:20480], iter=20)
CH = 250_000
if a.gpu_build_batch or a.gpu_build_grow or a.gpu_seeds:
    from annlite_amd.core.index import graph_gpu_build as _gb

    if a.gpu_build_batch:
        _gb.GpuLevel0Graph.BATCH = a.gpu_build_batch
    if a.gpu_build_grow:
        _gb.GpuLevel0Graph.GROW = a.gpu_build_grow
    if a.gpu_seeds:
        _gb.GpuLevel0Graph.MAX_SEEDS = a.gpu_seeds


def build_index(where):
    ix = HnswPQGpuIndex(dim=D, metric=Metric.EUCLIDEAN, pq_codec=codec, initial_size=N, rerank=True, ef_search=a.ef_search,
                        ef_construction=a.ef_construction, max_connection=a.max_connection, build=where)
    torch.cuda.synchronize()
    t0 = time.time()
    for c in range((N + CH - 1) // CH):
        rows = min(CH, N - c * CH)
        ix.add_with_ids(gen(c, rows), torch.arange(c * CH, c * CH + rows, device=dev, dtype=torch.int64))
    torch.cuda.synchronize()
    return ix, time.time() - t0


index_gpu, gpu_build_s = build_index('gpu') if a.build in ('both', 'gpu') else (None, None)
index_host, host_build_s = build_index('host') if a.build in ('both', 'host') else (None, None)
index = index_gpu if index_gpu is not None else index_host  # what is measured
build_s = gpu_build_s if index_gpu is not None else host_build_s

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


gq2 = torch.Generator(device=dev)
gq2.manual_seed(4321 + 7919)
q_alt = (torch.randn((B, r_lat), generator=gq2, device=dev) @ A + 0.05 * torch.randn((B, D), generator=gq2, device=dev)).contiguous()
q_rot = [q, q_alt]  # the timed steps rotate through two query batches (recall is the first batch's)
side = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]


def timed(fn, streams=1):
    """fn(batch) -> result; `streams` > 1: consecutive batches on alternating caller streams."""
    for j in range(4):  # (warm-up ON the streams that are timed: torch's allocator keeps a pool per stream, the first allocations of a
        with torch.cuda.stream(side[j % streams] if streams > 1 else torch.cuda.current_stream()):  # stream are hipMalloc calls)
            fn(q_rot[j % 2])
    torch.cuda.synchronize()
    n = max(a.steps, 2)
    t = time.perf_counter()
    if streams > 1:
        for j in range(n):
            with torch.cuda.stream(side[j % streams]):
                fn(q_rot[j % 2])
    else:
        for j in range(n):
            fn(q_rot[j % 2])
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    out = fn(q)
    torch.cuda.synchronize()
    return out, B * n / el


res = {}
for name, rerank, graph, walk in (('hnsw_gpu_walk_adc', False, True, 'gpu'), ('hnsw_gpu_walk_exact_rerank', True, True, 'gpu'),
                                  ('hnsw_host_walk_adc', False, True, 'host'), ('hnsw_host_walk_exact_rerank', True, True, 'host'),
                                  ('exhaustive_adc', False, False, None), ('exhaustive_exact_rerank', True, False, None)):
    ix = index_host if walk == 'host' else index  # (host walks need the hierarchy: the host-built graph)
    if ix is None:
        continue
    ix.rerank = rerank
    if walk:
        ix.walk = walk
    fn = (lambda qq: ix.search_batch(qq, limit=k)) if graph else (lambda qq: ix.search_exhaustive(qq, limit=k))
    ns = a.streams if (graph and walk == 'gpu') else 1
    (d, i), qps = timed(fn, ns)
    res[name] = {'queries_per_s': qps, 'recall_at_10': recall(i), 'streams': ns}
    if graph and walk == 'gpu' and ns > 1:  # ... and one batch at a time, for the record
        res[name]['one_stream_queries_per_s'] = timed(fn, 1)[1]
if index_gpu is not None and index_host is not None:  # graph quality: the same GPU walk over the host-built graph
    index_host.walk, index_host.rerank = 'gpu', True
    (d, i), qps = timed(lambda qq: index_host.search_batch(qq, limit=k), a.streams)
    res['hnsw_gpu_walk_exact_rerank_on_host_built_graph'] = {'queries_per_s': qps, 'recall_at_10': recall(i), 'streams': a.streams}
    index_host.rerank = False
    (d, i), qps = timed(lambda qq: index_host.search_batch(qq, limit=k), a.streams)
    res['hnsw_gpu_walk_adc_on_host_built_graph'] = {'queries_per_s': qps, 'recall_at_10': recall(i), 'streams': a.streams}
# the walks alone
walks = {}
for walk in ('gpu', 'host'):
    ix = index_host if walk == 'host' else index
    if ix is None:
        continue
    ix.walk = walk
    ix.candidates(q, a.ef_search)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        ix.candidates(q, a.ef_search)
    torch.cuda.synchronize()
    walks[walk] = B * a.steps / (time.perf_counter() - t)
walk_qps = walks

# ---- roofline of the dominant kernel (graph_beam_search_kernel): HIP-event time of the walk launch alone, algorithmic
# bytes from the walk's own counters (expansions x one link list + evaluated rows x M code bytes) against the HBM
# peak -- a pointer chase is latency-bound, the fraction says how far from streaming it is -----------------------------
from annlite_amd import _capi, ops  # noqa: E402
from annlite_amd._capi import LAYOUT_BMK, LUT_L2  # noqa: E402

index.walk = 'gpu'
qd = index._pre(q)
_, xg = codec.scan_inputs(qd)
links, seeds = index._export_graph()
lut = ops.lut_build(xg, codec.codebooks_dev, LUT_L2, LAYOUT_BMK)
plain = index._plain_table(index._n_rows)
lpn = links.shape[1] - 1
packed = index._packed_records(links, plain) if index.packed_graph and lpn <= 64 else None


width = index.expand_width if (packed is not None and lpn <= 32) else 1  # nodes per step of the packed walk (2: the pair walk)


def walk_once(use_packed, w=None):
    if use_packed:
        return ops.graph_search_packed(packed, lpn, seeds, plain, lut, a.ef_search, valid_bits=index._valid, n_rows=index._n_rows,
                                       expand_width=width if w is None else w)
    return ops.graph_search(links, seeds, plain, lut, a.ef_search, valid_bits=index._valid, n_rows=index._n_rows)


def walk_ms(use_packed, w=None):
    for _ in range(2):
        walk_once(use_packed, w)
    torch.cuda.synchronize()
    out = []
    for _ in range(max(3, a.steps)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        walk_once(use_packed, w)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.mean(out))


os.environ['ANNLITE_DEBUG_COUNTERS'] = '1'
_capi.knobs_reload()  # (the library parses its switches at load)
walk_once(packed is not None)
n_expand, n_eval, n_hit = _capi.graph_search_stats_ex()
phase_cycles = dict(_capi.graph_search_stats_ex.cycles)
del os.environ['ANNLITE_DEBUG_COUNTERS']
_capi.knobs_reload()
kernel_ms = walk_ms(packed is not None)
plain_ms = walk_ms(False) if packed is not None else kernel_ms
same = None
one_ms = pair_overlap = None
if packed is not None:  # the two layouts walk the same graph in the same order (one node per step): bit-equal candidate lists
    (i1, d1), (i2, d2) = walk_once(True, 1), walk_once(False)
    same = bool(torch.equal(i1, i2) and torch.equal(d1.view(torch.int32), d2.view(torch.int32)))
    if width == 2:  # the pair walk against the one-at-a-time walk: time, and the share of its candidates found by both
        one_ms = walk_ms(True, 1)
        ip, _ = walk_once(True, 2)
        both = (ip[:, :, None] == i1[:, None, :]).any(dim=2) & (ip >= 0)
        pair_overlap = float(both.sum().item() / max(1, int((i1 >= 0).sum().item())))
rec_bytes = int(packed.shape[1]) if packed is not None else None
# algorithmic bytes: what the WALK needs -- one link list per expansion + M code bytes per evaluated row (+ the seed rows);
# the packed layout reads a whole record per expansion by design (`bytes_read_by_design`)
alg_bytes = n_expand * 4.0 * (lpn + 1) + n_eval * float(M) + B * seeds.numel() * float(M)
design_bytes = (n_expand * float(rec_bytes) + B * seeds.numel() * float(M)) if packed is not None else alg_bytes
# measured HBM traffic of the walk launch: the committed rocprofv3 PMC pass (FETCH_SIZE x 2 + WRITE_SIZE, profiles/traffic.json),
# refused when it was taken on another revision of the kernel
traffic = None
traffic_note = None
try:
    _tt = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'traffic.json')))
    ent = _tt.get(f'graph_beam_search_kernel:{N}x{M}x{B}', {})
    traffic = ent.get('hbm_bytes_per_launch')
    rev = _capi.kernel_rev('graph_beam_search_kernel')
    if traffic is not None and ent.get('kernel_rev') != rev:
        traffic_note = f'STALE: measured on kernel revision {ent.get("kernel_rev")}, the library is at {rev}'
        print('bench_hnsw.py: profiles/traffic.json ' + traffic_note, file=sys.stderr)
        traffic = None
except Exception as ex:  # noqa: BLE001
    traffic_note = str(ex)
roofline = {'bound': 'hbm', 'achieved': alg_bytes / (kernel_ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
            'frac': alg_bytes / (kernel_ms * 1e-3) / 1e9 / 8000.0, 'traffic': traffic, 'traffic_note': traffic_note,
            'kernel': 'graph_beam_search_kernel', 'kernel_rev': _capi.kernel_rev('graph_beam_search_kernel'), 'kernel_ms': kernel_ms,
            'layout': 'packed node records (neighbours\' code rows inline, next record prefetched)' if packed is not None else 'plain',
            'plain_layout_kernel_ms': plain_ms, 'packed_equals_plain_bit_exact': same, 'record_bytes': rec_bytes,
            'expand_width': width, 'one_at_a_time_kernel_ms': one_ms, 'pair_candidates_shared_with_one_at_a_time': pair_overlap,
            'prefetched_records_used': (n_hit / max(n_expand, 1)),
            # shader cycles per query by phase of the walk (ANNLITE_DEBUG_COUNTERS run: the stamps themselves cost a few per cent)
            'cycles_per_query_by_phase': {kk: vv / B for kk, vv in phase_cycles.items()},
            'algorithmic_bytes_per_launch': alg_bytes, 'bytes_read_by_design': design_bytes,
            'expansions_per_query': n_expand / B, 'rows_evaluated_per_query': n_eval / B,
            'note': 'a pointer chase, latency-bound by design: one wave per query, one dependent record read per expansion'}

# ---- CPU baseline: the host-built graph walked on the host by libannlite_graph.so (C++ restatement of hnswlib's searchKnn /
# searchBaseLayerST with PQLookup distances), ONE thread = the reference's execution model (knn_query runs single-threaded
# for AnnLite's one-query calls, hnsw_bindings.cpp:332-334), bounded sample; all cores beside it ------------------------
def usable_threads() -> int:
    """The threads the graph library actually starts (hnsw_host.cpp usable_threads): min(CPUs, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, max(1, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max' and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


cpu_baseline = None
if index_host is not None:
    index_host.walk = 'host'
    nq1 = min(B, 256)
    threads_all = index_host.n_threads
    index_host.n_threads = 1
    index_host.candidates(q[:8], a.ef_search)
    t0 = time.perf_counter()
    index_host.candidates(q[:nq1], a.ef_search)
    cpu1 = nq1 / (time.perf_counter() - t0)
    index_host.n_threads = threads_all
    cpu_baseline = {'value': cpu1, 'unit': 'queries/s', 'cores': 1, 'kind': 'port',
                    'sample': f'{nq1} queries, graph walk ef_search={a.ef_search} over {N} rows on the host (candidate lists only), host-built graph',
                    'all_cores': {'value': walks['host'], 'cores': usable_threads(), 'sample': f'{B} queries x {a.steps}'}}
print(json.dumps({'config': f'HNSW-over-PQ: {N} x {D}-dim, PQ m={M} ks=256, L2, max_connection={a.max_connection}, '
                            f'ef_construction={a.ef_construction}, ef_search={a.ef_search}, batch {B}, k={k}',
                  'metric': 'queries/sec', 'value': res['hnsw_gpu_walk_exact_rerank']['queries_per_s'], 'unit': 'queries/s',
                  'streams': res['hnsw_gpu_walk_exact_rerank']['streams'], 'query_batches': 2,
                  'ms_per_step': B / res['hnsw_gpu_walk_exact_rerank']['queries_per_s'] * 1e3,
                  'recall_at_10': res['hnsw_gpu_walk_exact_rerank']['recall_at_10'],
                  'graph_built_on': 'gpu (level 0, batches: graph_build.hip)' if index_gpu is not None else 'host (libannlite_graph.so)',
                  'build_s': build_s, 'build_rows_per_s': N / build_s, 'host_build_s': host_build_s, 'gpu_build_s': gpu_build_s,
                  'host_cpus_reported': os.cpu_count(), 'note': 'the graph library starts min(CPUs, affinity, cgroup quota) threads',
                  'graph_walk_queries_per_s': walk_qps, 'roofline': roofline, 'cpu_baseline': cpu_baseline, **res}))
