#!/usr/bin/env python3
"""Repeat the capacity-padded fixture scans (thousands of tied all-zero rows: the candidate path under flood) and count
the runs that differ from the oracle -- a race shows up as an occasional mismatch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pq_oracle as oracle
from annlite_amd import ops
from annlite_amd._capi import scan_plan
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name in ('c4_m64_d768', 'c2_m16_d128'):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    g = {kk: z[kk] for kk in z.files}
    M, dsub, Ks, N, B_, seed, k = (int(v) for v in g['meta'])
    cap = int(g['pqindex_capacity'][0])
    table = np.zeros((cap, M), np.uint8); table[:N] = g['codes']
    lut = g['lut_l2_batch']
    rd, ri = oracle.adc_search_c(lut, table, k)
    B = lut.shape[0]
    plan = scan_plan(cap, M, 256, 1, B, k)
    for layout in (1, 0, 1, 0):
        codes_d = ops.to_dev(table)
        if layout == 1: codes_d = ops.codes_skew(codes_d)
        lut_d = ops.lut_retile(ops.to_dev(lut), plan.qi)
        bad = 0
        which = []
        for r in range(reps):
            d, i = ops.adc_scan_topk(codes_d, lut_d, B, k, M, 256, codes_layout=layout)
            torch.cuda.synchronize()
            if not (np.array_equal(d.cpu().numpy(), rd) and np.array_equal(i.cpu().numpy(), ri)):
                bad += 1
                which.append(r)
                if bad == 1:
                    dd, ii = d.cpu().numpy(), i.cpu().numpy()
                    qb = [b for b in range(B) if not np.array_equal(ii[b], ri[b])]
                    print('  first mismatch: queries', qb[:4], 'got', ii[qb[0]], 'want', ri[qb[0]], 'N', N, 'cap', cap)
        print(name, 'layout', layout, 'B', B, 'k', k, 'qt', plan.qt, 'slices', plan.n_slices, ': %d of %d runs differ' % (bad, reps), which[:20])
