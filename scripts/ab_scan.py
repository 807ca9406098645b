#!/usr/bin/env python3
"""A/B of environment switches of the scan on ONE data set in ONE process (the library reads its switches per call):

    python scripts/ab_scan.py --rows 10000000 --m 64 --dsub 12 --batch 256 --data lowrank --envs "|ANNLITE_Q8_MAP=0|ANNLITE_Q8_MAP=1"

Every setting: a few untimed calls, then `--iters` calls of annlite_pq_search_topk (tables built inside); prints the median
HIP-event time of the scan kernel and of the whole call, and whether the results equal the first setting's."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annlite_amd import _capi, ops  # noqa: E402
from annlite_amd._capi import LUT_L2  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--rows', type=int, default=1_250_000)
p.add_argument('--m', type=int, default=16)
p.add_argument('--dsub', type=int, default=8)
p.add_argument('--batch', type=int, default=1024)
p.add_argument('--k', type=int, default=10)
p.add_argument('--ks', type=int, default=256)
p.add_argument('--iters', type=int, default=40)
p.add_argument('--rank', type=int, default=0, help='latent rank of the lowrank data (0: 16 up to 128-d, 64 above)')
p.add_argument('--data', choices=['random', 'lowrank'], default='lowrank')
p.add_argument('--layout', type=int, default=1)
p.add_argument('--kind', type=int, default=LUT_L2)
p.add_argument('--envs', default='', help="settings separated by '|', each 'A=1,B=2' (empty = the defaults)")
a = p.parse_args()
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
N, M, Ks, B, k = a.rows, a.m, a.ks, a.batch, a.k
D = M * a.dsub
if a.data == 'random':
    codes = torch.randint(0, Ks, (N, M), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    cb = torch.randn((M, Ks, a.dsub), generator=g, device=dev)
    q = torch.randn((B, D), generator=g, device=dev)
else:
    from annlite_amd import Metric, PQCodec

    r = a.rank or (16 if D <= 128 else 64)
    A = torch.randn((r, D), generator=g, device=dev)

    def gen(n):
        return (torch.randn((n, r), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()

    codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1)
    codec.seed = 7
    codec.fit(gen(20480), iter=15)
    cb = codec.codebooks_dev
    codes = torch.empty((N, M), dtype=torch.uint8, device=dev)
    step = max(1, 200_000_000 // D)
    for c0 in range(0, N, step):
        n = min(step, N - c0)
        codes[c0:c0 + n] = ops.pq_encode(gen(n), cb)
    q = gen(B)
if a.layout == 1:
    codes = ops.codes_skew(codes)
valid = torch.full(((N + 31) // 32 + 2,), -1, dtype=torch.int32, device=dev)
base = None
switches = set()
settings = a.envs.split('|') if a.envs else ['']
for s in settings:
    for kv in filter(None, s.split(',')):
        switches.add(kv.split('=')[0])
for s in settings:
    for name in switches:
        os.environ.pop(name, None)
    for kv in filter(None, s.split(',')):
        name, val = kv.split('=')
        os.environ[name] = val
    ws = ops.ScanWorkspace()
    state = _capi.ScanState()
    for _ in range(6):
        d, i = ops.pq_search_topk(a.kind, q, cb, codes, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid, state=state)
    torch.cuda.synchronize()
    _capi.profile_enable(True)
    kms, cms = [], []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d, i = ops.pq_search_topk(a.kind, q, cb, codes, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid, state=state)
        e1.record()
        kms.append(_capi.profile_last_scan_ms())
        torch.cuda.synchronize()
        cms.append(e0.elapsed_time(e1))
    _capi.profile_enable(False)
    # back-to-back throughput (no sync between the calls)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        d, i = ops.pq_search_topk(a.kind, q, cb, codes, k, M, Ks, codes_layout=a.layout, workspace=ws, valid_bits=valid, state=state)
    e1.record()
    torch.cuda.synchronize()
    thr = e0.elapsed_time(e1) / a.iters
    res = (d.cpu().numpy(), i.cpu().numpy())
    same = True if base is None else (np.array_equal(res[0], base[0], equal_nan=True) and np.array_equal(res[1], base[1]))
    if base is None:
        base = res
    look = float(B) * N * M
    per_clk = 256
    print('%-50s kernel ms median %.4f min %.4f | call %.4f | back-to-back %.4f ms | of the byte roof %.3f | same result %s | %s' % (
        s or '(defaults)', float(np.median(kms)), min(kms), float(np.median(cms)), thr, look / (float(np.median(kms)) * 1e-3) / (256 * per_clk * 2.4e9),
        same, _capi.ScanState.KERNELS[state.info()[0]]), flush=True)
