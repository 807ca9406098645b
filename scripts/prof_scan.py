#!/usr/bin/env python3
"""Run only the hot path (LUT build + ADC scan + merge) a few times -- the command rocprofv3 wraps.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python scripts/prof_scan.py --rows 10000000
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU ... -d gpurun_out/pmc -- python scripts/prof_scan.py
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annlite_amd import _capi, ops  # noqa: E402
from annlite_amd._capi import LAYOUT_TILED, LUT_L2, scan_plan  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--rows', type=int, default=1_000_000)
p.add_argument('--m', type=int, default=16)
p.add_argument('--batch', type=int, default=1024)
p.add_argument('--k', type=int, default=10)
p.add_argument('--ks', type=int, default=256, help='codewords per sub-space (> 256: uint16 codes, PLAIN layout)')
p.add_argument('--dsub', type=int, default=8, help='floats per sub-vector (D = m * dsub)')
p.add_argument('--iters', type=int, default=5)
p.add_argument('--layout', type=int, default=1)
p.add_argument('--valid', action='store_true', help='pass an all-ones validity bitmap (what the index plugin does)')
p.add_argument('--fused', action='store_true', help='annlite_pq_search_topk (tables built inside)')
p.add_argument('--data', choices=['random', 'lowrank'], default='random',
               help="random: uniform codes + gaussian codebooks; lowrank: the bench's data (rank-16 latent + noise, trained codec)")
p.add_argument('--order', choices=['iid', 'sorted'], default='iid',
               help='lowrank data: sorted = the table is filled in order of the first latent coordinate (an insertion order that makes '
                    'the head of the table unrepresentative: the seed rows of the first bound must not be its first rows)')
a = p.parse_args()
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
N, M, Ks, B, k = a.rows, a.m, a.ks, a.batch, a.k
D = M * a.dsub
if Ks > 256:
    a.layout = 0
if a.data == 'random':
    codes = torch.randint(0, Ks, (N, M), generator=g, device=dev, dtype=torch.int32).to(torch.uint8 if Ks <= 256 else torch.int16)
    cb = torch.randn((M, Ks, D // M), generator=g, device=dev)
    q = torch.randn((B, D), generator=g, device=dev)
else:
    from annlite_amd import Metric, PQCodec
    A = torch.randn((16, D), generator=g, device=dev)

    def gen(n):
        return (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()

    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.deterministic = True
    codec.fit(gen(20480), iter=20)
    cb = codec.codebooks_dev
    codes = torch.empty((N, M), dtype=torch.uint8 if Ks <= 256 else torch.int16, device=dev)
    for c0 in range(0, N, 500_000):
        n = min(500_000, N - c0)
        codes[c0:c0 + n] = ops.pq_encode(gen(n), cb)
    if a.order == 'sorted':  # (rows ordered by their projection on the first latent direction)
        key = torch.empty((N,), device=dev)
        cbf = cb.reshape(M, Ks, -1)
        u = (A[0] / A[0].norm()).reshape(M, -1)
        proj = torch.einsum('mkd,md->mk', cbf, u)  # contribution of every codeword to the projection
        for c0 in range(0, N, 500_000):
            cc = codes[c0:c0 + 500_000].long()
            key[c0:c0 + cc.shape[0]] = proj.gather(1, cc.t()).sum(0)
        codes = codes[torch.argsort(key)].contiguous()
    q = gen(B)
    if a.layout == 1:
        codes = ops.codes_skew(codes)
plan = scan_plan(N, M, Ks, 1 if Ks <= 256 else 2, B, k)
ws = ops.ScanWorkspace()
state = _capi.ScanState()  # (as the index plug-in does: the library settles on a kernel after the first launches)
valid = torch.full(((N + 31) // 32 + 1,), -1, dtype=torch.int32, device=dev) if a.valid else None
_capi.profile_enable(True)
ms = []
for it in range(a.iters):
    if a.fused:
        d, i = ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid, state=state)
    else:
        lut = ops.lut_build(q, cb, LUT_L2, LAYOUT_TILED, plan.qi)
        d, i = ops.adc_scan_topk(codes, lut, B, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid)
    ms.append(_capi.profile_last_scan_ms())
torch.cuda.synchronize()
if a.fused and not os.environ.get('ANNLITE_DEBUG_COUNTERS'):  # the whole call (preparation launch, scan, merge), back to back
    import time
    _capi.profile_enable(False)
    n_w = 20
    t0 = time.perf_counter()
    for _ in range(n_w):
        ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid, state=state)
    torch.cuda.synchronize()
    print('whole call, %d back to back: %.4f ms per batch' % (n_w, (time.perf_counter() - t0) / n_w * 1e3))
    _capi.profile_enable(True)
look = B * N * M
med = float(np.median(ms[len(ms) // 2:]))  # (the first iterations include module load and the clock ramp)
print('kernel choice:', _capi.ScanState.KERNELS[state.info()[0]], state.info())
print('scan kernel ms: median of the last %d: %.4f  min %.4f  first %.3f' % (len(ms) - len(ms) // 2, med, min(ms), ms[0]),
      ' lookups/s %.3e' % (look / (med * 1e-3)), ' alg GB/s %.1f' % (look / (med * 1e-3) / 1e9))

if os.environ.get('ANNLITE_DEBUG_COUNTERS') and plan.qt == 32:
    c = _capi.debug_counters()
    nwg = 256
    print('byte-table kernel: wave-steps with candidates %d, pushed %d, exact sums %d, queued for a list %d, table rebuilds %d, consumer batches %d; '
          'per workgroup: consumer inside batches %.1f us, wave 0 at epoch ends %.1f us' % (c[0], c[1], c[2], c[3], c[5], c[6], c[4] / nwg / 2400., c[7] / nwg / 2400.))
    print('byte-table kernel timeline (us; per-work-item averages but the span):', _capi.debug_timeline())
    if a.fused:
        try:
            print('preparation launch (us; first / last workgroup):', _capi.debug_prep_timeline())
        except Exception as ex:  # (a plan without the fused preparation launch)
            print('preparation launch: no stamps (%s)' % ex)
    it = _capi.debug_items()
    if len(it):
        t0 = it[:, 2].min()
        tot = (it[:, 6] - it[:, 2]) / 100.0
        scan = (it[:, 4] - it[:, 3]) / 100.0
        print('items: total us min %.1f mean %.1f max %.1f; step loop us min %.1f mean %.1f max %.1f; last end %.1f' %
              (tot.min(), tot.mean(), tot.max(), scan.min(), scan.mean(), scan.max(), (it[:, 6].max() - t0) / 100.0))
        order = np.argsort(-it[:, 6])[:6]
        print('  the last to end (tile, slice: build / loop / wait / merge us, end at):',
              ['%d,%d: %.1f/%.1f/%.1f/%.1f @%.1f' % (it[j, 0], it[j, 1], (it[j, 3] - it[j, 2]) / 100., (it[j, 4] - it[j, 3]) / 100.,
                                                   (it[j, 5] - it[j, 4]) / 100., (it[j, 6] - it[j, 5]) / 100., (it[j, 6] - t0) / 100.) for j in order])
        mg = (it[:, 6] - it[:, 5]) / 100.0
        print('  merge phase us: median %.1f, the 32 longest mean %.1f max %.1f' % (np.median(mg), np.sort(mg)[-32:].mean(), mg.max()))
        for name, col in (('tile', 0), ('slice', 1), ('xcd', None)):
            key = it[:, 7] % 8 if col is None else it[:, col]
            means = [(int(v), float(scan[key == v].mean())) for v in np.unique(key)]
            means.sort(key=lambda x: x[1])
            print('  step loop by %s: fastest %s ... slowest %s' % (name, ['%d:%.1f' % m for m in means[:4]], ['%d:%.1f' % m for m in means[-4:]]))
elif os.environ.get('ANNLITE_DEBUG_COUNTERS'):
    c = _capi.debug_counters()
    n_wave_steps = (N // 64) * ((B + plan.qt - 1) // plan.qt)
    print('counters: slow-block entries %d, events %d, inserting events %d, publications %d, candidate rows %d ; wave-steps %d'
          ' ; table rebuilds %d, ring-full waits %d, consumer batches %d' % (c[0], c[1], c[2], c[3], c[4], n_wave_steps, c[5], c[6], c[7]))
