#!/usr/bin/env python3
"""bench.py -- queries/sec of the PQ/ADC search hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
             --master-port P bench.py --gpus N --steps K --warmup W)

Workload (BASELINE.json `metric`: "queries/sec + recall@10, 10M x 128-dim PQ-m=16, batch-1k,
1/2/4/8 GPU"): 10M synthetic float32 vectors of dimension 128 (low-rank mixture, SURVEY.md
section 8d), PQ m=16 ks=256 trained on the first 20 480 rows, squared-L2 tables, batch of 1024
queries, k=10.  The code table is row-sharded over the N ranks (STRONG scaling: the 10M rows are
fixed), every rank scans its shard, one RCCL all-gather of [B,k] + a merge kernel gives every rank
the global top-k.

One "step" = the whole hot path for one batch: LUT build (tiled layout) -> seed bound -> ADC scan +
per-shard top-k -> (N>1: all-gather + merge).  Inputs (queries, codebooks, codes) are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0 with the driver's contract fields plus
`roofline` (dominant kernel = adc_scan_q8_kernel at the headline shape: B*N_local*M table look-ups per
launch over its HIP-event duration against the LDS look-up rate the kernel's entry width allows --
the code rows are shared by the 32 queries of a tile, so the scan runs out of LDS, not HBM; the
algorithmic code bytes per second and the measured HBM traffic are reported beside it, DESIGN.md
section 7) and `cpu_baseline` (the C oracle, single thread = the reference's execution model,
bounded sample).  Extra legs (rank 0, N=1, never `value`): `rerank` (recall target), `ivf`, and --
with the default workload only -- `c2`, `c4`, `c5` (the other BASELINE configurations, each a
sub-run with its own `roofline` and `cpu_baseline`), `m32` (the default workload at m = 32) and `uniform` (the reference's
own test distribution); `--legs` selects them.

Set-up before the W warm-up steps includes `--prewarm-steps` (64) untimed steps: a GPU that idled through index
construction needs milliseconds of work to reach its sustained clock -- at a 0.27 ms step (one rank's shard of 8) W = 2 /
K = 20 measured 0.311 ms per step without it and 0.266 with it (K = 200: 0.265 / 0.257).  The timed region is unchanged:
exactly K steps between two synchronisations (+ barriers), max over ranks.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--rows', type=int, default=10_000_000)
    p.add_argument('--dim', type=int, default=128)
    p.add_argument('--m', type=int, default=16)
    p.add_argument('--ks', type=int, default=256)
    p.add_argument('--batch', type=int, default=1024)
    p.add_argument('--k', type=int, default=10)
    p.add_argument('--query-batches', type=int, default=4,
                   help='DISTINCT query batches the warm-up and the timed steps rotate through (step i searches batch i mod this); every '
                        'one of them is checked against the CPU oracle and enters `result_sha256`')
    p.add_argument('--train-rows', type=int, default=20480)
    p.add_argument('--train-iters', type=int, default=20)
    p.add_argument('--recall-queries', type=int, default=128, help='queries used for recall@10 (0 = skip)')
    p.add_argument('--cpu-queries', type=int, default=64,
                   help='queries of the bounded single-thread CPU-baseline sample (0 = skip); 64 x 10M rows = ~2.5 s per run')
    p.add_argument('--cpu-repeats', type=int, default=5, help='timed runs of the single-thread sample (median reported; BASELINE.md section 3: >= 5)')
    p.add_argument('--data', choices=['lowrank', 'uniform'], default='lowrank',
                   help='database / query distribution (SURVEY.md section 8d): lowrank = rank-16 (rank-64 above 128-d) latent Gaussian '
                        '+ noise; uniform = U[0,1)^D, the reference\'s own test distribution (tests/test_pq_bind.py:19)')
    p.add_argument('--legs', default='auto',
                   help='comma list of extra legs (rank 0, N=1, never `value`): rerank, graph, ivf, facade, uniform, c2, c4, c5, m32, k50; "none"; "auto" = all '
                        'of them for the default workload, rerank + ivf otherwise')
    p.add_argument('--metric', choices=['euclidean', 'cosine', 'inner_product'], default='euclidean',
                   help="BASELINE config 2/3: euclidean; config 4 (10M x 768, m=64, batch 256): cosine")
    p.add_argument('--streams', type=int, choices=list(range(0, 9)), default=0,
                   help='HIP streams the timed, independent batches alternate on (0 = auto: 2 when the exchange runs, else 1).  On two '
                        'streams the next batch\'s preparation launch fills the CUs the previous scan\'s tail leaves idle (-1.8 %% per batch '
                        'at 10M rows, -7 %% at 1.25M) -- but a scan launched while the previous one still holds the CUs is TIMED from its '
                        'launch: rocprofv3 then reports 1.77 ms per scan kernel where HIP events around a lone launch say 1.34 '
                        '(profiles/r04/bench_10m_n1_two_streams_rocprof_kernel_stats.txt), so the default single-GPU line stays on one stream')
    p.add_argument('--prewarm-steps', type=int, default=64, help='untimed steps of set-up before the W warm-up steps (clock ramp); 0 = none')
    p.add_argument('--layout', choices=['skewed', 'plain'], default='skewed')
    p.add_argument('--rerank-k', type=int, default=0, help='ADC candidates per row slice of the exact re-rank leg (0 = the index default, 64)')
    p.add_argument('--no-rerank', action='store_true', help='skip the (untimed-in-value) exact re-rank leg')
    p.add_argument('--ivf-cells', type=int, default=256,
                   help='extra leg (N=1, never `value`): pruned search over this many cells; 0 = skip')
    p.add_argument('--ivf-probe', type=int, default=16)
    p.add_argument('--seed-exchange', action='store_true',
                   help='N > 1 (opt-in): every rank seeds from 1 / N of the single-GPU seed rows and the ranks all-gather their seeds\' '
                        'k smallest bounds before the scan (sharded.py: measured slower than the plain search on this runtime)')
    p.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                   help='torch.distributed backend of an N > 1 run.  nccl (= RCCL over xGMI) is the product path and what the driver '
                        'launches.  gloo exists so that the script\'s N > 1 branches (shard ranges, broadcast of the codebooks, the packed '
                        'exchange + merge kernel, max-over-ranks timing, per-rank records, the merged recall) can run on ONE GPU: '
                        'RCCL refuses two ranks on one device, gloo does not care -- the ranks share cuda:(local_rank mod device count) '
                        'and every collective goes through host memory (tests/test_bench_two_ranks.py)')
    p.add_argument('--emulate-seed-peers', type=int, default=0,
                   help='ONE rank with the exchange forced (ANNLITE_FORCE_GATHER=1 under torchrun): stand-ins for P - 1 peers in the '
                        'seed exchange -- key sets precomputed, untimed, from P - 1 other row ranges of THIS shard (valid bounds for it) '
                        '-- so that a single GPU runs a rank of a P-rank job at that job\'s cost: seed from 1 / P of the rows, the '
                        'collective, the union over P key sets, a scan that starts from a P-rank bound')
    return p.parse_args()


UNIFORM = False  # --data uniform


def gen_chunk(chunk: int, rows: int, D: int, A: torch.Tensor, dev) -> torch.Tensor:
    """x = z.A + 0.05 eps, z ~ N(0, I_r): identical for any number of ranks (seeded per chunk)."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + chunk)
    if UNIFORM:
        return torch.rand((rows, D), generator=g, device=dev)
    z = torch.randn((rows, A.shape[0]), generator=g, device=dev)
    e = torch.randn((rows, D), generator=g, device=dev)
    return (z @ A + 0.05 * e).contiguous()


def sub_run(cmd, timeout_s):
    """One of the other BASELINE configurations as a sub-run of this file / scripts/bench_hnsw.py: its JSON line, or the
    reason it is missing."""
    env = {k: v for k, v in os.environ.items()  # (a sub-run is a plain single-process run, whatever launched this one)
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                        'TORCHELASTIC_RUN_ID', 'ANNLITE_FORCE_GATHER')}
    try:
        r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            return {'error': f'rc={r.returncode}', 'stderr_tail': r.stderr[-400:]}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {'error': f'timeout after {timeout_s} s'}


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def torchrun_argv(n_gpus: int, argv, port: int):
    """The command line a plain ``python bench.py --gpus N`` (N > 1, no RANK / WORLD_SIZE in the environment) re-executes
    itself as: one process per GPU over RCCL, rendezvous on 127.0.0.1 (the container hostname may not resolve) -- the form
    the driver uses for N > 1.  ``argv`` = this script's own arguments, passed on unchanged."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + list(argv)


def main():
    global UNIFORM
    args = parse()
    UNIFORM = args.data == 'uniform'
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the N-rank job (same arguments, same JSON line from rank 0)
        cmd = torchrun_argv(args.gpus, sys.argv[1:], free_port())
        print('bench.py: --gpus %d without a launcher, re-executing as: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr)
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:  # the launcher decides how many ranks exist; the line reports what actually ran (n_gpus = world)
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)', file=sys.stderr)
    N_, D_, M_, Ks_, B_, k_ = args.rows, args.dim, args.m, args.ks, args.batch, args.k
    default_workload = (N_, D_, M_, Ks_, B_, k_, args.metric, args.data) == (10_000_000, 128, 16, 256, 1024, 10, 'euclidean', 'lowrank')
    legs = args.legs.split(',') if args.legs not in ('auto', 'none') else (
        [] if args.legs == 'none' else ['rerank', 'ivf'] + (['graph', 'facade', 'uniform', 'c2', 'c4', 'c5', 'm32', 'k50'] if default_workload else []))
    if args.no_rerank and 'rerank' in legs:
        legs.remove('rerank')
    if args.ivf_cells <= 1 and 'ivf' in legs:
        legs.remove('ivf')
    if world > 1:
        legs = []  # (the extra legs are single-GPU figures)
    # ---- the other BASELINE configurations and the reference's own test distribution, as sub-runs (rank 0, N=1), run BEFORE this
    # process creates its own GPU context (a second resident process costs the sub-run's kernels 25-40 %) ------------
    sub = {}
    if rank == 0 and world == 1:
        me = os.path.join(ROOT, 'bench.py')
        # (the legs' batches alternate between two streams, as the multi-GPU runs do; each leg's own config says so)
        common = ['--legs', 'none', '--gpus', '1', '--streams', '2', '--query-batches', '2']
        if 'c2' in legs:  # config 2: 1M x 128, m=16, L2, batch 1024
            sub['c2'] = sub_run([me, '--rows', '1000000', '--steps', '40', '--warmup', '10'] + common, 240)
        if 'c4' in legs:  # config 4: 10M x 768, m=64, cosine, batch 256
            sub['c4'] = sub_run([me, '--rows', '10000000', '--dim', '768', '--m', '64', '--batch', '256', '--metric', 'cosine',
                                 '--steps', '20', '--warmup', '5', '--cpu-queries', '16', '--cpu-repeats', '3',
                                 '--recall-queries', '32'] + common, 400)
        if 'c5' in legs:  # config 5: HNSW-over-PQ, 5M x 128, ef_search 128, GPU walk + exact re-rank
            sub['c5'] = sub_run([os.path.join(ROOT, 'scripts', 'bench_hnsw.py'), '--rows', '5000000', '--steps', '20'], 600)
        if 'm32' in legs:  # the default workload at m = 32 (the reference's own table-test shape): byte tables of one entry group
            sub['m32'] = sub_run([me, '--m', '32', '--steps', '20', '--warmup', '5', '--cpu-queries', '16', '--cpu-repeats', '3',
                                  '--recall-queries', '32'] + common, 300)
        if 'k50' in legs:  # the default workload at k = 50 (the reference's own PQ test asks for topk = 50: tests/test_pq_index.py:83-135)
            sub['k50'] = sub_run([me, '--k', '50', '--steps', '20', '--warmup', '5', '--cpu-queries', '16', '--cpu-repeats', '3',
                                  '--recall-queries', '32'] + common, 300)
        if 'uniform' in legs:  # U[0,1)^D: unstructured codes -- the kernel the library picks for them
            sub['uniform'] = sub_run([me, '--data', 'uniform', '--steps', '10', '--warmup', '3', '--cpu-queries', '16', '--cpu-repeats', '3',
                                      '--recall-queries', '32'] + common, 300)

    gloo = args.backend == 'gloo'
    dev_idx = local_rank % max(1, torch.cuda.device_count()) if gloo else local_rank  # (gloo: the ranks may share a device)
    torch.cuda.set_device(dev_idx)
    dev = torch.device('cuda', dev_idx)
    use_dist = world > 1 or 'RANK' in os.environ  # torchrun with 1 rank also initialises RCCL
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if gloo:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    # the script's own collectives (never inside the timed region's data path: that is sharded.py's packed exchange).  Under gloo
    # they go through host memory -- gloo's support for device tensors differs per collective
    def coll_bcast(t):
        if not gloo:
            dist.broadcast(t, src=0)
            return t
        h = t.cpu()
        dist.broadcast(h, src=0)
        t.copy_(h)
        return t

    def coll_max(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device='cpu' if gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def coll_gather(t):
        src = t.cpu() if gloo else t
        outl = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(outl, src)
        return [o.to(dev) for o in outl]

    from annlite_amd import Metric, PQCodec, _capi, ops
    from annlite_amd.core.index.pq_flat_gpu import PQFlatGpuIndex
    from annlite_amd.sharded import ShardedPQIndex, shard_range

    N, D, M, Ks, B, k = args.rows, args.dim, args.m, args.ks, args.batch, args.k
    r_lat = 16 if D <= 128 else 64
    gA = torch.Generator(device=dev)
    gA.manual_seed(99)
    A = torch.randn((r_lat, D), generator=gA, device=dev)

    # ---- train the codec on the first rows (rank 0), broadcast the codebooks ----------------------
    metric = {'euclidean': Metric.EUCLIDEAN, 'cosine': Metric.COSINE, 'inner_product': Metric.INNER_PRODUCT}[args.metric]
    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=metric, n_init=1)
    codec.seed = 7
    codec.deterministic = True  # (bit-identical codebooks run after run: the result digests compare across runs)
    CH = 250_000
    t0 = time.time()
    if rank == 0:
        xt = gen_chunk(0, CH, D, A, dev)[: args.train_rows]
        codec.fit(xt, iter=args.train_iters)
        cb = codec.codebooks_dev.clone()
    else:
        cb = torch.empty((M, Ks, D // M), dtype=torch.float32, device=dev)
    if world > 1:
        coll_bcast(cb)
    codec.set_codebooks(cb)
    train_s = time.time() - t0

    # ---- build this rank's shard --------------------------------------------------------------------
    lo, hi = shard_range(N, world, rank)
    n_local = hi - lo
    keep_vectors = 'rerank' in legs or world > 1 and not args.no_rerank
    index = PQFlatGpuIndex(dim=D, metric=metric, pq_codec=codec, initial_size=max(n_local, 64),
                           rerank=keep_vectors, skewed=(args.layout == 'skewed'))
    t0 = time.time()
    c0, c1 = lo // CH, (hi + CH - 1) // CH
    for c in range(c0, c1):
        rows = min(CH, N - c * CH)
        x = gen_chunk(c, rows, D, A, dev)
        a, b = max(lo, c * CH), min(hi, c * CH + rows)
        xs = x[a - c * CH: b - c * CH]
        index.add_with_ids(xs, torch.arange(a - lo, b - lo, device=dev, dtype=torch.int64))
    torch.cuda.synchronize()
    index_s = time.time() - t0
    sharded = ShardedPQIndex(index, row_base=lo, seed_exchange=args.seed_exchange, n_total=N)

    # NB distinct query batches (same on every rank): the steps rotate through them, so that nothing the library keeps per
    # table (annlite_scan_state: its kernel choice rests on statistics of the batches it has seen) or per stream is tuned to
    # ONE batch; batch 0 is round 1-4's batch (seed 4321)
    NB = max(1, args.query_batches)
    q_sets = []
    for j in range(NB):
        gq = torch.Generator(device=dev)
        gq.manual_seed(4321 + 7919 * j)
        if UNIFORM:
            q_sets.append(torch.rand((B, D), generator=gq, device=dev))
        else:
            zq = torch.randn((B, r_lat), generator=gq, device=dev)
            eq = torch.randn((B, D), generator=gq, device=dev)
            q_sets.append((zq @ A + 0.05 * eq).contiguous())
    queries = q_sets[0]

    def barrier():
        if world > 1:
            dist.barrier()

    # the product path is plain ADC (no re-rank): that is the reference's PQ search semantics
    index.rerank = False

    step_no = [0]

    def step():
        step_no[0] += 1
        return sharded.search_batch(q_sets[step_no[0] % NB], limit=k)

    n_streams = args.streams or (2 if (world > 1 or os.environ.get('ANNLITE_FORCE_GATHER')) else 1)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams >= 2 else [torch.cuda.current_stream(dev)]
    _leg_streams = []

    def leg_stream_pair():
        # ONE pair of caller streams for every in-process leg that alternates batches (graph, ivf): HIP maps a process's streams onto a few
        # hardware queues, and a fresh pair per leg can land both of its streams on one queue (the ivf leg's two-stream figure then
        # equalled its one-stream figure while scripts/bench_ivf_bytes.py, two streams in a fresh process, measured 1.24x)
        if not _leg_streams:
            _leg_streams.extend(streams if len(streams) >= 2 else [torch.cuda.Stream(device=dev) for _ in range(2)])
        return _leg_streams
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    # clocks: a GPU that has idled through index construction and host-side set-up takes milliseconds of work to reach its
    # sustained clock -- at a 0.27 ms step a short warm-up (W = 10) left the first timed steps 8 % slow.  Untimed set-up work,
    # before the W warm-up steps the contract asks for:
    # (a fixed number of steps, not a time: with the exchange every rank must run the same collectives)
    for _ in range(args.prewarm_steps):
        step()
    torch.cuda.synchronize()
    n_emulated = 0
    if args.emulate_seed_peers > 1 and world == 1 and use_dist and os.environ.get('ANNLITE_FORCE_GATHER') and args.seed_exchange:
        # (after the set-up steps: the table's kernel state has settled by now, PREPARE applies)
        from annlite_amd._capi import PHASE_PREPARE

        P = args.emulate_seed_peers
        s_rows = max(4096, -(-(min(32768, max(8192, -(-(N * P // 32) // 1024) * 1024)) // P) // 1024) * 1024)
        kind, xq = index._scan_inputs(queries, index._pre(queries))
        peers = []
        for j in range(1, P):
            # (row ranges in the first half of the shard: a view must keep more than half the table's rows, or the table's kernel
            # state takes it for another table and measures again)
            off = (j * (n_local // (2 * P))) // 64 * 64
            pk = ops.pq_search_split(PHASE_PREPARE, kind, xq, codec.codebooks_dev, index._codes[off:], k, M, Ks, index.scan_state,
                                     index._ws, valid_bits=index._valid[off // 32:], n_rows=n_local - off,
                                     codes_layout=index._layout(), seed_rows=s_rows)
            if pk is not None:
                peers.append(pk.clone())
        if len(peers) == P - 1:
            sharded._peer_keys = torch.stack(peers).contiguous()
            sharded.n_total = N * P  # (this shard stands for 1 / P of a P-times larger table)
            n_emulated = P
        torch.cuda.synchronize()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    for w_i in range(max(args.warmup, len(streams))):  # (every stream warms its own scratch buffer up)
        with torch.cuda.stream(streams[w_i % len(streams)]):
            step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    # K independent batches: the exchange of batch i (all-gather + merge, side stream) overlaps the scan of
    # batch i+1 (compute stream, never waits); every batch's merged result exists before the closing sync
    # Consecutive batches alternate between two streams: each batch's kernels are in order on their own stream,
    # and the next batch's table build / seed / first workgroups fill the CUs the previous scan's tail leaves idle
    pending = None
    outs = [None] * NB  # the LAST result of every distinct batch (read only after the closing synchronisation)
    for s_i in range(args.steps):
        with torch.cuda.stream(streams[s_i % len(streams)]):
            nxt = sharded.search_batch_async(q_sets[s_i % NB], limit=k)
        if pending is not None:
            outs[(s_i - 1) % NB] = pending.result(wait=False)
        pending = nxt
    outs[(args.steps - 1) % NB] = pending.result(wait=False)
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3  # host time per step to ENQUEUE the K steps (no device wait in it)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed  # (this rank's clock; `elapsed` becomes the max over the ranks)
    if world > 1:
        elapsed = coll_max(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    qps = B * args.steps / elapsed
    for j in range(NB):  # (fewer timed steps than batches: the rest untimed, so that every batch has a result to check)
        if outs[j] is None:
            outs[j] = sharded.search_batch(q_sets[j], limit=k)
    torch.cuda.synchronize()
    out = outs[0]

    # ---- what was computed, as checksums a reader can compare ACROSS runs: the merged result of every distinct batch (ids +
    # distance bits; identical on every rank and for every N -- a SCALE line must print the N = 1 line's digest), and an
    # ADDITIVE checksum of each rank's shard of the code table (the ranks' sums add up to the N = 1 figure mod 2^64) ----------
    import hashlib

    h_all = hashlib.sha256()
    sha_batches = []
    for j in range(NB):
        blob = outs[j][1].to(torch.int64).cpu().numpy().tobytes() + outs[j][0].to(torch.float32).cpu().numpy().view(np.uint32).tobytes()
        h_all.update(blob)
        sha_batches.append(hashlib.sha256(blob).hexdigest()[:16])
    result_sha256 = h_all.hexdigest()

    def shard_checksum():
        total = 0
        plain = index._plain_codes(n_local)
        wcol = (torch.arange(M * index.code_bytes, device=dev, dtype=torch.int64) * 2 + 1)[None, :]
        for a0 in range(0, n_local, 1 << 20):
            a1 = min(n_local, a0 + (1 << 20))
            rows8 = plain[a0:a1].contiguous().view(torch.uint8).view(a1 - a0, -1).to(torch.int64)
            gid = torch.arange(lo + a0, lo + a1, device=dev, dtype=torch.int64)
            wrow = ((gid * 2654435761 + 12345) & 0x7fffffff) | 1
            total = (total + int(((rows8 * wcol).sum(1) * wrow).sum().item())) & 0xFFFFFFFFFFFFFFFF
        return total

    my_checksum = shard_checksum()

    # ---- second figure: the same steps with the host buffers the AnnLite API hands over --
    # numpy queries in (H2D), numpy results out (D2H); never `value`
    host_qps = None
    if world == 1:
        q_host = queries.cpu().numpy()
        n_h = max(3, min(args.steps, 10))
        index.search_batch(q_host, limit=k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_h):
            _host_d, _host_i = index.search_batch(q_host, limit=k)  # numpy in -> numpy out (synchronous)
        host_qps = B * n_h / (time.perf_counter() - t0)

    if os.environ.get('ANNLITE_DEBUG_COUNTERS') and rank == 0:
        c = _capi.debug_counters()  # of the last step (debug aid; the counters slow the kernel down)
        print('counters: slow-block entries %d, flush query-groups %d, inserting %d, publications %d, candidate rows %d' %
              (c[0], c[1], c[2], c[3], c[4]), file=sys.stderr)

    # ---- roofline leg: HIP events around the dominant kernel, live, over the same steps ------------
    for _ in range(min(args.prewarm_steps, 32)):  # (the host-transfer leg above left the GPU waiting on the host: sustained clock first)
        step()
    torch.cuda.synchronize()
    _capi.profile_enable(True)
    kms, clks = [], []
    for _ in range(max(3, min(args.steps, 10))):
        step()
        kms.append(_capi.profile_last_scan_ms())
        clks.append(_capi.profile_last_scan_clock_mhz())  # (byte-table kernel: the shader clock that launch held; else None)
    _capi.profile_enable(False)
    kernel_ms = float(np.mean(kms))
    clock_mhz = float(np.mean([c for c in clks if c])) if any(clks) else None

    # ---- N > 1 (or the exchange forced on one rank): what the exchange alone costs, and every rank's own figures ----
    gathering = use_dist and (world > 1 or bool(os.environ.get('ANNLITE_FORCE_GATHER')))
    exchange_ms = None
    per_rank = None
    if gathering and not gloo:  # (gloo: the exchange bounces through the host, nothing to price)
        packed = index.search_batch_packed(queries, k, lo)
        if packed is not None:
            G_ = dist.get_world_size()
            gathered = torch.empty((G_ * B, k, 2), dtype=torch.int64, device=dev)
            for _ in range(3):
                dist.all_gather_into_tensor(gathered, packed)
                ops.topk_merge_packed(gathered.view(G_, B, k, 2), sqrt=index.sqrt_epilogue)
            torch.cuda.synchronize()
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_x = 20
            e0.record()
            for _ in range(n_x):
                dist.all_gather_into_tensor(gathered, packed)  # (in stream order: torch makes the current stream wait for RCCL's)
                ops.topk_merge_packed(gathered.view(G_, B, k, 2), sqrt=index.sqrt_epilogue)
            e1.record()
            torch.cuda.synchronize()
            exchange_ms = e0.elapsed_time(e1) / n_x  # one packed all-gather of [B, k, 2] i64 + the merge kernel, back to back
    if use_dist:
        mine = torch.tensor([own_elapsed / args.steps * 1e3, kernel_ms, exchange_ms if exchange_ms is not None else -1.0, float(n_local),
                             float(my_checksum >> 32), float(my_checksum & 0xFFFFFFFF), float(int(result_sha256[:8], 16)),
                             clock_mhz if clock_mhz else -1.0],
                            dtype=torch.float64, device=dev)
        allr = coll_gather(mine) if world > 1 else [mine]
        per_rank = [{'rank': r, 'ms_per_step': float(t[0]), 'kernel_ms': float(t[1]),
                     'exchange_ms': None if float(t[2]) < 0 else float(t[2]), 'rows': int(t[3]),
                     'shard_codes_checksum': '%016x' % ((int(t[4]) << 32) | int(t[5])),
                     # (every rank holds the merged result: the first 32 bits of ITS digest -- all equal, or a rank merged differently)
                     'result_sha256_head': '%08x' % int(t[6]),
                     'shader_clock_mhz': None if float(t[7]) < 0 else float(t[7])} for r, t in enumerate(allr)]
    scan_bytes = float(B) * n_local * M  # algorithmic code bytes consumed per launch
    achieved = scan_bytes / (kernel_ms * 1e-3) / 1e9
    lookups_per_s = float(B) * n_local * M / (kernel_ms * 1e-3)

    # measured HBM traffic of the same launch, from the committed rocprofv3 PMC pass (FETCH_SIZE x2
    # gfx950 correction + WRITE_SIZE, profiles/*/traffic.json); bench.py cannot run rocprof on itself
    traffic_table = {}
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            traffic_table = json.load(f)
    except Exception:
        traffic_table = {}

    # ---- recall@10 vs exact brute force (subset of the queries), ADC-only and with re-rank ---------
    recall_adc = recall_rr = None
    rr_qps = None
    rr_global = None
    nq = min(args.recall_queries, B)
    if nq > 0:
        qs = queries[:nq]
        best_d = torch.full((nq, k), float('inf'), device=dev)
        best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
        if args.metric == 'cosine':
            qs = qs / qs.norm(dim=1, keepdim=True)
        qn = (qs * qs).sum(1)[:, None]
        for c in range(c0, c1):
            rows = min(CH, N - c * CH)
            x = gen_chunk(c, rows, D, A, dev)
            a, b = max(lo, c * CH), min(hi, c * CH + rows)
            xs = x[a - c * CH: b - c * CH]
            if args.metric == 'cosine':
                xs = xs / xs.norm(dim=1, keepdim=True)  # cosine order == L2 order of the normalised vectors
            if args.metric == 'inner_product':
                dd = -(qs @ xs.T)
            else:
                dd = qn + (xs * xs).sum(1)[None, :] - 2.0 * (qs @ xs.T)
            cd, ci = torch.topk(dd, k, dim=1, largest=False)
            md = torch.cat([best_d, cd], 1)
            mi = torch.cat([best_i, ci + a], 1)
            o = torch.argsort(md, dim=1)[:, :k]
            best_d, best_i = torch.gather(md, 1, o), torch.gather(mi, 1, o)
        if world > 1:
            gd, gi = coll_gather(best_d), coll_gather(best_i)
            md, mi = torch.cat(gd, 1), torch.cat(gi, 1)
            o = torch.argsort(md, dim=1)[:, :k]
            best_i = torch.gather(mi, 1, o)
        truth = best_i.cpu().numpy()
        got = out[1][:nq].cpu().numpy()
        recall_adc = float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))
        if keep_vectors:
            index.rerank = True
            index.rerank_k = args.rerank_k or None
            for _ in range(2):
                rr = sharded.search_batch(queries, limit=k)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            n_rr = max(3, args.steps // 4)
            for _ in range(n_rr):
                rr = sharded.search_batch(queries, limit=k)
            torch.cuda.synchronize()
            barrier()
            rr_el = time.perf_counter() - t0
            if world > 1:
                rr_el = coll_max(rr_el)
            rr_qps = B * n_rr / rr_el
            got = rr[1][:nq].cpu().numpy()
            recall_rr = float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))
            # ... and with the pool taken from the GLOBAL ADC top-50 (round 5: the shared-bound search at k = 50 runs on the
            # byte-table kernel's 64-key lists) instead of the slices' own top-16 lists
            rr_global = None
            if world == 1 and M == 16:
                index.rerank_pool = 'global'
                index.rerank_k = 50
                for _ in range(2):
                    rr = sharded.search_batch(queries, limit=k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_rr):
                    rr = sharded.search_batch(queries, limit=k)
                torch.cuda.synchronize()
                got = rr[1][:nq].cpu().numpy()
                rr_global = {'value': B * n_rr / (time.perf_counter() - t0), 'unit': 'queries/s', 'pool': 'global ADC top-50',
                             'recall_at_10': float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))}
                index.rerank_pool = 'slices'
                index.rerank_k = args.rerank_k or None
            index.rerank = False

    # ---- extra leg, never `value`: HNSW-over-PQ over the SAME rows (config 5's index at the headline's size): the level-0 graph built
    # on the GPU in batches (round 6), walked by the pair walk (ef_search 128), candidates re-ranked exactly -- the north-star's
    # recall target (>= 0.90 recall@10) answered by the graph index instead of the exhaustive scan's candidate pool -----------------
    graph_rec = None
    if world == 1 and 'graph' in legs and nq > 0 and M in (8, 16, 32) and Ks <= 256 and args.metric != 'inner_product':
        try:
            from annlite_amd import HnswPQGpuIndex

            gidx = HnswPQGpuIndex(dim=D, metric=metric, pq_codec=codec, initial_size=max(N, 64), rerank=True, ef_search=128,
                                  ef_construction=200, max_connection=16, build='gpu')
            torch.cuda.synchronize()
            t0 = time.time()
            for c in range((N + CH - 1) // CH):
                rows = min(CH, N - c * CH)
                gidx.add_with_ids(gen_chunk(c, rows, D, A, dev), torch.arange(c * CH, c * CH + rows, device=dev, dtype=torch.int64))
            torch.cuda.synchronize()
            g_build_s = time.time() - t0  # (vector generation, encode, storage and the graph)
            g_streams = leg_stream_pair()  # (consecutive batches on two caller streams, as the other legs)
            for j in range(4):  # (warm-up on the timed streams: the allocator keeps a pool per stream)
                with torch.cuda.stream(g_streams[j % 2]):
                    gr = gidx.search_batch(q_sets[j % NB], limit=k)
            torch.cuda.synchronize()
            n_g = max(8, args.steps // 2)
            t0 = time.perf_counter()
            for j in range(n_g):
                with torch.cuda.stream(g_streams[j % 2]):
                    gr = gidx.search_batch(q_sets[j % NB], limit=k)
            torch.cuda.synchronize()
            g_el = time.perf_counter() - t0
            gr = gidx.search_batch(queries, limit=k)
            got = gr[1][:nq].cpu().numpy()
            graph_rec = {'index': 'HnswPQGpuIndex(build="gpu", ef_search=128, max_connection=16, ef_construction=200, rerank=True)',
                         'value': B * n_g / g_el, 'unit': 'queries/s', 'ms_per_step': g_el / n_g * 1e3,
                         'recall_at_10': float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)])),
                         'build_s': g_build_s, 'rows': N, 'streams': 2, 'answers': 'north_star recall target (>= 0.90 recall@10)'}
            # ... and with a longer candidate list (config 5 fixes ef_search = 128 at 5M rows; at 10M rows the 128 best by PQ distance
            # hold fewer of the true neighbours): lists beyond 128 entries take four registers per lane
            gidx.ef_search = 160
            for j in range(4):
                with torch.cuda.stream(g_streams[j % 2]):
                    gr = gidx.search_batch(q_sets[j % NB], limit=k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(n_g):
                with torch.cuda.stream(g_streams[j % 2]):
                    gr = gidx.search_batch(q_sets[j % NB], limit=k)
            torch.cuda.synchronize()
            g_el2 = time.perf_counter() - t0
            got = gidx.search_batch(queries, limit=k)[1][:nq].cpu().numpy()
            graph_rec['ef_search_160'] = {'value': B * n_g / g_el2, 'unit': 'queries/s', 'ms_per_step': g_el2 / n_g * 1e3,
                                          'recall_at_10': float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))}
            del gidx
            torch.cuda.empty_cache()
        except Exception as ex:  # noqa: BLE001  (a leg never takes the line down)
            graph_rec = {'error': repr(ex)[:300]}

    # ---- extra leg, never `value`: the pruned (IVF) search over the same rows (SURVEY.md 8f follow-on) --------
    ivf_rec = None
    if world == 1 and 'ivf' in legs and M in (8, 16, 32, 64) and nq > 0:
        from annlite_amd.core.codec.vq import VQCodec
        from annlite_amd.core.index.ivf_pq_gpu import IvfPQGpuIndex

        vq = VQCodec(args.ivf_cells, metric=metric, iter=15, n_init=1)
        vq.seed = 11
        vq.fit(gen_chunk(0, CH, D, A, dev)[:100_000])
        ivf = IvfPQGpuIndex(dim=D, metric=metric, pq_codec=codec, vq_codec=vq, initial_size=64, rerank=keep_vectors,
                            skewed=(args.layout == 'skewed'))
        # adopt the flat index's storage (same codes, validity, float vectors); add the cell of every row
        for name in ('_codes', '_valid_bool', '_valid_bits_cache', '_vectors', '_capacity', '_n_rows', '_size'):
            setattr(ivf, name, getattr(index, name))
        ivf._cell_of = torch.zeros((index._capacity,), dtype=torch.int32, device=dev)
        for c in range(c0, c1):
            rows = min(CH, N - c * CH)
            x = gen_chunk(c, rows, D, A, dev)
            a_, b_ = max(lo, c * CH), min(hi, c * CH + rows)
            ivf._cell_of[a_ - lo: b_ - lo] = vq.encode(x[a_ - c * CH: b_ - c * CH]).to(torch.int32)
        ivf.rerank = False
        P = min(args.ivf_probe, args.ivf_cells - 1)
        for _ in range(2):
            iv = ivf.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        n_iv = max(4, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(n_iv):
            iv = ivf.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        iv_qps_1 = B * n_iv / (time.perf_counter() - t0)
        # consecutive batches on two caller streams, as the other legs (`value`); the one-stream figure stays beside it
        i_streams = leg_stream_pair()
        for j in range(4):  # (warm-up on the timed streams: scratch and the allocator's pool are per stream)
            with torch.cuda.stream(i_streams[j % 2]):
                ivf.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(2 * n_iv):
            with torch.cuda.stream(i_streams[j % 2]):
                ivf.search_batch(queries, limit=k, n_probe=P)
        torch.cuda.synchronize()
        iv_qps = B * 2 * n_iv / (time.perf_counter() - t0)
        got_iv = iv[1][:nq].cpu().numpy()
        adc_ids = out[1][:nq].cpu().numpy()
        ivf_rec = {'n_cells': args.ivf_cells, 'n_probe': P, 'value': iv_qps, 'unit': 'queries/s', 'streams': 2, 'one_stream_value': iv_qps_1,
                   'path': ivf.last_pruned_path,
                   'recall_at_10': float(np.mean([len(set(got_iv[b]) & set(truth[b])) / k for b in range(nq)])),
                   'agreement_with_exhaustive_adc_top10': float(np.mean([len(set(got_iv[b]) & set(adc_ids[b])) / k for b in range(nq)]))}
        if ivf.last_pruned_path and ivf.last_pruned_path.startswith('annlite_ivf_search_topk'):
            # the u16 tile scan + re-score it replaces: same results (compared), its rate beside the new one; and the cell-tile launch's
            # duration against the look-up roof (the units it processes: B x P / C x N x M look-ups)
            try:
                from annlite_amd import _capi as _c

                _c.profile_enable(True)
                ivf.search_batch(queries, limit=k, n_probe=P)
                scan_ms = _c.profile_last_scan_ms()
                _c.profile_enable(False)
                ivf.byte_tiles = False
                for _ in range(2):
                    iv16 = ivf.search_batch(queries, limit=k, n_probe=P)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_iv):
                    iv16 = ivf.search_batch(queries, limit=k, n_probe=P)
                torch.cuda.synchronize()
                ivf_rec['u16_tiles_one_stream_value'] = B * n_iv / (time.perf_counter() - t0)
                ivf_rec['equal_to_u16_tiles'] = bool(torch.equal(iv16[0], iv[0]) and torch.equal(iv16[1], iv[1]))
                ivf.byte_tiles = True
                lookups = float(B) * P / args.ivf_cells * N * M
                ivf_rec['scan_kernel'] = {'ms': scan_ms, 'lookups': lookups, 'frac_of_lookup_roof': lookups / (scan_ms * 1e-3) / 1.573e14}
            except Exception as ex:  # noqa: BLE001
                ivf_rec['u16_tiles_error'] = repr(ex)[:200]
        if keep_vectors:
            ivf.rerank = True
            for _ in range(2):
                iv = ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_iv):
                iv = ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=32)
            torch.cuda.synchronize()
            got_iv = iv[1][:nq].cpu().numpy()
            ivf_rec['rerank'] = {'value': B * n_iv / (time.perf_counter() - t0), 'unit': 'queries/s', 'rerank_k': 32,
                                 'recall_at_10': float(np.mean([len(set(got_iv[b]) & set(truth[b])) / k for b in range(nq)]))}
            ivf_rec['rerank']['path'] = ivf.last_pruned_path
            # (round 6) the re-rank on the byte-table cell tiles: every probed cell's own ADC list of 16 (private lists) as the candidates,
            # exact distances + top-k fused; one stream, then consecutive batches on the two caller streams
            try:
                def _rr16():
                    for _ in range(2):
                        iv = ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=16)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n_iv):
                        iv = ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=16)
                    torch.cuda.synchronize()
                    q1 = B * n_iv / (time.perf_counter() - t0)
                    for j in range(4):
                        with torch.cuda.stream(i_streams[j % 2]):
                            ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=16)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for j in range(2 * n_iv):
                        with torch.cuda.stream(i_streams[j % 2]):
                            iv = ivf.search_batch(queries, limit=k, n_probe=P, rerank_k=16)
                    torch.cuda.synchronize()
                    q2 = B * 2 * n_iv / (time.perf_counter() - t0)
                    got = iv[1][:nq].cpu().numpy()
                    return {'value': q2, 'unit': 'queries/s', 'streams': 2, 'one_stream_value': q1, 'rerank_k': 16,
                            'bound_rank': ivf.rerank_bound_rank, 'path': ivf.last_pruned_path,
                            'recall_at_10': float(np.mean([len(set(got[b]) & set(truth[b])) / k for b in range(nq)]))}

                ivf_rec['rerank16'] = _rr16()  # the index's default: private lists, first bound at the 16-th seed sum, nearest 2 cells in 4 parts
                ivf_rec['rerank16']['split'] = list(ivf.rerank_split)
                split0 = ivf.rerank_split
                ivf.rerank_split = (0, 1)      # whole cells only (16 rows per cell at most)
                ivf_rec['rerank16_whole_cells'] = _rr16()
                ivf.rerank_split = split0
                ivf.rerank_bound_rank = 2      # a looser first bound (the 32nd seed sum): longer lists from the far cells
                ivf_rec['rerank16_rank2'] = _rr16()
                ivf.rerank_bound_rank = 0      # the pool = exactly the ADC top-16 (shared bounds: annlite_ivf_search_topk's ids)
                ivf_rec['rerank16_top16'] = _rr16()
                ivf.rerank_bound_rank = 1
            except Exception as ex:  # noqa: BLE001
                ivf_rec['rerank16'] = {'error': repr(ex)[:200]}
        del ivf

    # ---- the drop-in API itself (north_star: "keeping the AnnLite(...)/index()/search() Python API and DocArray result shape"):
    # AnnLite.search(docs) -- numpy embeddings in, doc.matches out -- and search_numpy over the SAME table; never `value` -------
    facade = None
    if world == 1 and 'facade' in legs:
        import shutil
        import tempfile

        from annlite_amd import AnnLite
        from annlite_amd.index import Document, DocumentArray

        tmp = tempfile.mkdtemp(prefix='annlite_bench_')
        try:
            ann = AnnLite(n_dim=D, metric=args.metric, n_subvectors=M, n_clusters=Ks, data_path=tmp)
            # adopt the bench's codec and table (indexing 10M python Documents is not what this leg measures): document id = str(row)
            ann._pq_codec = codec
            ann._vec_indexes = [index]
            ann._offset2id = list(map(str, range(n_local)))
            docs = DocumentArray([Document(id=f'q{b}', embedding=q_host[b]) for b in range(B)])
            name = metric.name.lower()
            n_f = max(3, min(args.steps, 20))

            def timed(fn):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_f):
                    fn()
                return B * n_f / (time.perf_counter() - t0)

            def read_all():
                ann.search(docs, limit=k)
                return sum(1 for d in docs for m in d.matches if m.id is not None and m.scores[name].value is not None)

            f_search = timed(lambda: ann.search(docs, limit=k))
            f_numpy = timed(lambda: ann.search_numpy(q_host, limit=k))
            f_read = timed(read_all)
            ann.search(docs, limit=k)
            ok = all([m.id for m in docs[b].matches] == [str(int(x)) for x in _host_i[b] if x >= 0] for b in range(0, B, 37))
            facade = {
                'search': {'value': f_search, 'unit': 'queries/s',
                           'note': 'AnnLite.search(docs, limit=k): 1024 Documents with numpy embeddings in, doc.matches out (lazy: '
                                   'the match Documents are built when a list is first read)'},
                'search_all_matches_read': {'value': f_read, 'unit': 'queries/s',
                                            'note': 'the same, then EVERY match\'s id and scores[metric].value read: what the '
                                                    'reference\'s eager loop always pays (container.py:226-233)'},
                'search_numpy': {'value': f_numpy, 'unit': 'queries/s', 'note': 'AnnLite.search_numpy(x, limit=k): lists of dists[k] / int ids[k]'},
                'matches_equal_index_result': bool(ok),
            }
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    # ---- CPU baseline: the oracle (C port of the reference loops) on a bounded sample, rank 0, N=1 --
    cpu = None
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import pq_oracle

        nqc = min(args.cpu_queries, B)
        codes_np = ops.codes_to_numpy(index._plain_codes(n_local))
        q_np = queries[:nqc].cpu().numpy()
        cb_np = codec.codebooks
        omet = {'euclidean': pq_oracle.EUCLIDEAN, 'cosine': pq_oracle.COSINE, 'inner_product': pq_oracle.INNER_PRODUCT}[args.metric]
        # single thread = the reference's execution model (Cython under the GIL): warm-up + median of 3
        n_rep = max(1, args.cpu_repeats)
        pq_oracle.index_search(q_np[:2], cb_np, codes_np, omet, k, threads=1)
        runs = []
        for _ in range(n_rep):
            t0 = time.perf_counter()
            cd, ci = pq_oracle.index_search(q_np, cb_np, codes_np, omet, k, threads=1)
            runs.append(time.perf_counter() - t0)
        cpu_s = float(np.median(runs))
        gd, gi = out[0][:nqc].cpu().numpy(), out[1][:nqc].cpu().numpy()
        if args.metric == 'cosine':
            # the timed batches hand over DEVICE queries (normalised by the kernel: ~1 ulp off numpy's einsum); the parity
            # sample goes through the host-buffer path -- the reference's numpy normalisation before the upload -- and
            # must then equal the CPU result exactly, ids and distances
            gd, gi = index.search_batch(q_np, limit=k)
        parity = bool(np.array_equal(cd, gd) and np.array_equal(ci, gi))
        # the same with the reference's result materialisation (Python list of N floats -> float64 array -> argpartition,
        # pq_index.py:46-49): a few queries are enough, the cost is per query
        lut_np = pq_oracle.get_dist_mat_c(pq_oracle.l2_normalize(q_np[:4]) if args.metric == 'cosine' else q_np[:4], cb_np, omet)
        pq_oracle.pqindex_search_reference_style(lut_np[0], codes_np[:1000], k)
        t0 = time.perf_counter()
        for b in range(lut_np.shape[0]):
            pq_oracle.pqindex_search_reference_style(lut_np[b], codes_np, k)
        ref_style_s = (time.perf_counter() - t0) / lut_np.shape[0]
        threads = pq_oracle.max_threads()
        # all cores: the WHOLE batch where the host has the cores for it (16 CPUs: ~2.5 s at 10M rows) -- and its result is
        # kept: every query of the timed batch is compared with the CPU oracle, not a sample
        nq_all = B if threads >= 8 else min(B, max(nqc, threads * 64))
        cpu_all_s = 0.0
        n_bad = 0
        bad_batches = []
        for j in range(NB):  # EVERY distinct batch of the timed loop, not only batch 0
            q_all = q_sets[j][:nq_all].cpu().numpy()
            t0 = time.perf_counter()
            cd_all, ci_all = pq_oracle.index_search(q_all, cb_np, codes_np, omet, k, threads=threads)
            cpu_all_s += time.perf_counter() - t0
            if args.metric == 'cosine':
                gd_all, gi_all = index.search_batch(q_all, limit=k)  # (host-buffer path, see above)
            else:
                gd_all, gi_all = outs[j][0][:nq_all].cpu().numpy(), outs[j][1][:nq_all].cpu().numpy()
            nb_j = int(np.sum(np.any(ci_all != gi_all, axis=1) | np.any(cd_all != gd_all, axis=1)))
            n_bad += nb_j
            if nb_j:
                bad_batches.append(j)
        parity_all = n_bad == 0
        nq_all_total = nq_all * NB
        cpu = {
            'value': nqc / cpu_s, 'unit': 'queries/s', 'cores': 1, 'kind': 'port',
            'sample': f'{nqc} queries x {n_local} rows (LUT + flat ADC scan + top-{k}), single thread = the reference execution '
                      f'model; kernel only, median of {n_rep} runs after a warm-up',
            'with_reference_materialisation': {
                'value': 1.0 / ref_style_s, 'unit': 'queries/s', 'cores': 1,
                'sample': f'{lut_np.shape[0]} queries: ADC kernel -> Python list of N floats -> np.expand_dims -> argpartition top-k '
                          '(pq_bindings.pyx:75-80, pq_index.py:46-49)'},
            'all_cores': {'value': nq_all_total / cpu_all_s, 'cores': threads,
                          'sample': f'{NB} batches x {nq_all} queries, OpenMP over queries, one run each'},
            'gpu_matches_cpu_bit_exact': parity,
            # ids AND distances of the timed batch's result against the CPU oracle, every one of `queries_checked` queries
            'gpu_matches_cpu_bit_exact_all': parity_all, 'queries_checked': nq_all_total, 'queries_differing': n_bad,
            'batches_checked': NB, 'batches_differing': bad_batches,
        }

    gathering_cfg = use_dist and (world > 1 or bool(os.environ.get('ANNLITE_FORCE_GATHER')))
    if rank == 0:
        # The scan does not stream its algorithmic bytes from HBM (every code row is shared by the 16 / 32 queries of a
        # tile and stays in L2): its roof is the LDS look-up rate.  One ds_read_b128 (4 LDS cycles per wave64) serves
        # 64 lanes x 16 byte entries (byte-table kernel) or x 8 u16 entries (u16 kernels); M=64: ds_read_b64, 2 cycles,
        # 64 x 4.  256 CUs at 2.4 GHz (MI355X_MICROARCH.md: 256 B / clk / CU).
        plan_k = _capi.scan_plan(n_local, M, Ks, index.code_bytes, B, k)
        # (which M = 16 kernel served the table is the library's choice, from what its launches measured: index.scan_kernel)
        # (16 < k <= 64 at M = 16: the public plan is the u16 plan, the library's search runs the byte-table kernel with 64-key lists)
        lk64 = ((M in (8, 16, 32) and index.code_bytes == 1 and Ks <= 256) or (M == 8 and index.code_bytes == 2 and Ks <= 1024)) and 16 < k <= 64 and n_local >= 65536
        byte_tables = (plan_k.qt == 32 or (M == 64 and plan_k.qt == 8) or (M == 8 and index.code_bytes == 2 and plan_k.qt == 16) or
                       (M == 32 and plan_k.qt == 16) or lk64) and \
            index.scan_kernel != 'u16 tables' and \
            os.environ.get('ANNLITE_SCAN_VARIANT', '0') in ('0', '50')
        per_clk = 256 if byte_tables else 128
        lds_peak = 256 * per_clk * 2.4e9
        kernel_name = (('adc_scan_lds_kernel' if M * Ks * 4 <= 144 * 1024 else 'adc_scan_generic_kernel') if not plan_k.fast else 'adc_scan_q8_kernel' if byte_tables else
                       'adc_scan_qfilter64_kernel' if M == 64 else 'adc_scan_qfilter_kernel')
        # measured HBM traffic: a committed PMC pass of the same kernel / shape (bench.py cannot run rocprof on itself).  An entry
        # measured on ANOTHER revision of the kernel (the library's ANNLITE_KERNEL_REV for it has moved on since) is refused, loudly
        t_key = f'{kernel_name}:{n_local}x{M}x{B}' + ('' if k <= 16 else f':k{k}') + ('' if args.data == 'lowrank' else f':{args.data}')
        t_ent = traffic_table.get(t_key, {})
        traffic = t_ent.get('hbm_bytes_per_launch')
        traffic_note = None if traffic is not None else f'no PMC pass kept for {t_key}'
        lib_rev = _capi.kernel_rev(kernel_name)
        if traffic is not None and t_ent.get('kernel_rev') != lib_rev:
            traffic_note = (f'STALE: profiles/traffic.json[{t_key}] was measured on kernel revision {t_ent.get("kernel_rev")}, '
                            f'the library is at revision {lib_rev}: re-run the PMC passes (scripts/r06_profiles.sh)')
            print('bench.py: ' + traffic_note, file=sys.stderr)
            traffic = None
        # shapes with both a byte-table and a u16-table kernel (scan.hip: search_policy) have a choice to report
        has_choice = (k <= 16 or lk64) and index.code_bytes in (1, 2) and ((M in (8, 16, 32) and index.code_bytes == 1 and Ks <= 256) or
                                                                 (M == 8 and index.code_bytes == 2 and Ks <= 1024))
        roof = {
            'bound': 'lds', 'achieved': lookups_per_s, 'peak': lds_peak, 'unit': 'look-ups/s', 'frac': lookups_per_s / lds_peak,
            'traffic': traffic, 'traffic_note': traffic_note, 'kernel_rev': lib_rev,
            'kernel': kernel_name, 'kernel_ms': kernel_ms, 'lookups_per_clk_per_cu': per_clk,
            'kernel_choice': index.scan_kernel if has_choice else 'fixed',
            'peak_note': 'design-relative: one ds_read_b128 (M=64: ds_read_b64) per wave64 per 4 (2) LDS cycles x 64 lanes x the '
                         'entries this kernel packs per read (byte tables: 16 one-byte entries per 16 B, M=64 8 per 8 B = 256 look-ups / clk / CU; '
                         'u16 tables: 8 per 16 B, M=64 4 per 8 B = 128) x 256 CUs x 2.4 GHz: the roof of THIS table format, not a chip constant',
            # the M = 64 byte-table kernel is VALU-bound and holds 2.1 GHz, not the 2.4 GHz the roof is priced at (PMC:
            # GRBM_GUI_ACTIVE / 8 XCDs / duration, profiles/r03/scan_c4_10m_m64_q8_pmc_*.csv): the fraction at THAT clock beside it
            # round 5: the clock is MEASURED in the run (workgroup 0's s_memtime over its 100 MHz wall clock, mean over the profiled
            # launches) for every byte-table shape -- a slow box shows here, not as a slow kernel
            'shader_clock_mhz': clock_mhz,
            'frac_at_measured_clock': (lookups_per_s / (256 * per_clk * clock_mhz * 1e6)) if clock_mhz else None,
            # SURVEY.md 8(d)'s per-unit figure: M code bytes per (query, row) evaluation -- what a one-query-at-a-time scan
            # (the reference) streams; reported for comparison, not a fraction of anything
            'algorithmic': {'bytes_per_launch': scan_bytes, 'GB_per_s': achieved},
            'hbm': None if traffic is None else {'bytes_per_launch': traffic, 'frac': traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        rec = {
            'metric': 'queries/sec', 'value': qps, 'unit': 'queries/s', 'n_gpus': world, 'gpus_arg': args.gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic' if args.data == 'lowrank' else 'synthetic (uniform)',
            'config': {
                'workload': f'{N} x {D}-dim float32, PQ m={M} ks={Ks}, {args.metric}, batch {B}, k={k}, exhaustive ADC scan + exact top-k',
                'rows_total': N, 'rows_per_gpu': n_local, 'batch': B, 'k': k, 'parallelism': f'row-shard x{world}',
                'codes_layout': args.layout,
                # what torch.distributed itself reports (one process per GPU over RCCL); 1 / None without a process group
                'n_ranks': dist.get_world_size() if use_dist else 1,
                'backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ' (collectives through host memory; ranks may share a device)')) if use_dist else None,
                # independent batches alternate between this many HIP streams (each batch's kernels in order on its own)
                'streams': n_streams,
                'query_batches': NB,  # distinct batches the timed steps rotate through
                'prewarm_steps': args.prewarm_steps,  # untimed set-up steps BEFORE the W warm-up steps (clock ramp)
                # N > 1: every rank seeds from 1 / N of the single-GPU seed rows and the ranks all-gather their seeds' k smallest
                # bounds before the scan (sharded.py); seed_peers_emulated: --emulate-seed-peers (one rank standing for P)
                'seed_exchange': bool(gathering_cfg and sharded.seed_exchange and index.split_supported(queries, k)),
                'seed_rows': sharded.seed_rows() if gathering_cfg else None,
                'seed_peers_emulated': n_emulated,
            },
            # N > 1: every rank's own clock over the K steps, its scan kernel (HIP events) and the exchange alone (one packed
            # all-gather + merge, 20 back to back) -- a first multi-GPU run says where its time went
            'per_rank': per_rank,
            # digest of (ids, distance bits) of the LAST result of each of the `query_batches` distinct batches, in order: the same
            # for every N (the merged result is the single-GPU result) -- compare a SCALE line with the N = 1 line
            'result_sha256': result_sha256, 'result_sha256_per_batch': sha_batches,
            # additive checksum of the rank's shard of the code table (sum over ranks mod 2^64 == the N = 1 value)
            'shard_codes_checksum_sum': '%016x' % (sum(int(r['shard_codes_checksum'], 16) for r in per_rank) & 0xFFFFFFFFFFFFFFFF)
            if per_rank else '%016x' % my_checksum,
            'exchange_ms': exchange_ms,
            'host_enqueue_ms_per_step': host_enqueue_ms,  # (close to ms_per_step: the host, not the GPU, paces the loop)
            'recall_at_10': recall_adc,
            # `value` is the reference's own search semantics (plain ADC top-k, the parity quantity); the north-star's
            # ">= 90 % recall@10" figure is the re-rank leg below
            'rerank': None if recall_rr is None else {'recall_at_10': recall_rr, 'value': rr_qps, 'unit': 'queries/s',
                                                       'candidates_per_query': 'n_slices * rerank_k per shard (rerank_k = %d: the byte-table kernel\'s 16-key lists)' % (args.rerank_k or 16),
                                                       'answers': 'north_star recall target (>= 0.90 recall@10)',
                                                       'global_pool': rr_global},
            'graph': graph_rec,
            'roofline': roof,
            'cpu_baseline': cpu,
            'ivf': ivf_rec,
            'facade': facade,
            'with_host_transfer': {'value': host_qps, 'unit': 'queries/s',
                                   'note': 'numpy queries in, numpy results out per batch (PCIe both ways, synchronous)'},
            'setup': {'train_s': train_s, 'index_s': index_s},
        }
        for name, r in sub.items():  # (never `value`: the other configurations, each with its own roofline / cpu_baseline)
            if 'error' in r:
                rec[name] = r
            elif name == 'c5':
                rec[name] = {kk: r[kk] for kk in ('config', 'value', 'unit', 'streams', 'ms_per_step', 'recall_at_10', 'graph_built_on', 'build_s', 'gpu_build_s',
                                                 'host_build_s', 'graph_walk_queries_per_s', 'roofline', 'cpu_baseline',
                                                 'hnsw_gpu_walk_adc', 'hnsw_gpu_walk_exact_rerank', 'hnsw_gpu_walk_exact_rerank_on_host_built_graph',
                                                 'exhaustive_exact_rerank') if kk in r}
            else:
                rec[name] = {'config': r['config']['workload'], 'streams': r['config'].get('streams'), 'value': r['value'], 'unit': r['unit'],
                             'result_sha256': r.get('result_sha256'),
                             'ms_per_step': r['ms_per_step'],
                             'recall_at_10': r.get('recall_at_10'), 'roofline': r['roofline'], 'cpu_baseline': r['cpu_baseline']}
        # LAST key of the line: every leg in a few numbers -- a reader who keeps only the tail of the line (the driver's record
        # keeps its last kilobytes) still sees all of them: q/s, ms per batch, kernel's fraction of its roof, queries checked against
        # the CPU oracle / differing, the first 8 hex digits of the result digest; N > 1: every rank's digest head and checksum
        def _r(x, n):
            return None if x is None else round(float(x), n)

        def leg_summary(r):
            if not isinstance(r, dict) or 'error' in r:
                return {'error': (r or {}).get('error', 'missing')} if isinstance(r, dict) else None
            rf, cb = r.get('roofline') or {}, r.get('cpu_baseline') or {}
            o = {'qps': _r(r.get('value'), 0), 'ms': _r(r.get('ms_per_step'), 4), 'frac': _r(rf.get('frac'), 3),
                 'chk': cb.get('queries_checked'), 'diff': cb.get('queries_differing'), 'cpu_qps': _r(cb.get('value'), 1),
                 'sha': (r.get('result_sha256') or '')[:8] or None}
            if r.get('recall_at_10') is not None:
                o['recall'] = _r(r['recall_at_10'], 3)
            return {kk: v for kk, v in o.items() if v is not None}

        summ = {'main': leg_summary(rec)}
        summ['main']['n'] = world
        if rec.get('rerank'):
            summ['rerank'] = {'qps': _r(rec['rerank']['value'], 0), 'recall': _r(rec['rerank']['recall_at_10'], 3)}
        if graph_rec and 'error' not in graph_rec:
            summ['graph'] = {'qps': _r(graph_rec['value'], 0), 'recall': _r(graph_rec['recall_at_10'], 3), 'build_s': _r(graph_rec['build_s'], 1)}
            if 'ef_search_160' in graph_rec:
                summ['graph']['ef160'] = [_r(graph_rec['ef_search_160']['value'], 0), _r(graph_rec['ef_search_160']['recall_at_10'], 3)]
        if ivf_rec:
            summ['ivf'] = {'qps': _r(ivf_rec['value'], 0), 'qps_1s': _r(ivf_rec.get('one_stream_value', 0), 0),
                           'agree': _r(ivf_rec['agreement_with_exhaustive_adc_top10'], 3)}
            if isinstance(ivf_rec.get('rerank16'), dict) and 'value' in ivf_rec['rerank16']:
                summ['ivf']['rr16'] = [_r(ivf_rec['rerank16']['value'], 0), _r(ivf_rec['rerank16']['recall_at_10'], 3)]
            if isinstance(ivf_rec.get('rerank16_rank2'), dict) and 'value' in ivf_rec['rerank16_rank2']:
                summ['ivf']['rr16_r2'] = [_r(ivf_rec['rerank16_rank2']['value'], 0), _r(ivf_rec['rerank16_rank2']['recall_at_10'], 3)]
        if facade:
            summ['facade'] = {'search_qps': _r(facade['search']['value'], 0), 'numpy_qps': _r(facade['search_numpy']['value'], 0)}
        for name in ('c2', 'c4', 'c5', 'm32', 'k50', 'uniform'):
            if name in rec:
                summ[name] = leg_summary(rec[name])
                if name == 'c5' and isinstance(rec[name], dict) and 'build_s' in rec[name]:
                    summ[name]['build_s'] = _r(rec[name]['build_s'], 1)
                    if rec[name].get('host_build_s') is not None:
                        summ[name]['host_build_s'] = _r(rec[name]['host_build_s'], 1)
        if per_rank and world > 1:
            summ['ranks'] = {'sha': [r['result_sha256_head'] for r in per_rank], 'rows': [r['rows'] for r in per_rank],
                             'ms': [_r(r['ms_per_step'], 4) for r in per_rank], 'checksum_sum': rec['shard_codes_checksum_sum']}
        rec['summary'] = summ
        print(json.dumps(rec))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
