#!/usr/bin/env python3
"""Randomised parity run (GPU box): random shapes, table orders, validity patterns, k, layouts and entry points against the CPU
oracle, bit for bit, until the time budget is spent.  The oracle is the checker (oracle/pq_oracle.py restates pq_bindings.pyx:30-47,
149-274 and math.py:94-120 with the fixed tie-break distance, then row id).

Test infrastructure (it is the oracle that checks): `tests/test_fuzz_parity.py` runs a short budget inside the GPU suite;

    python tests/fuzz_parity.py --seconds 200 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import pq_oracle as oracle  # noqa: E402
from annlite_amd import _capi, ops  # noqa: E402
from annlite_amd._capi import LAYOUT_BMK, LAYOUT_TILED, scan_plan  # noqa: E402



def run(seconds, seed, verbose=True):
    """Random cases until `seconds` are spent; returns (cases, calls, mismatches, cases by M)."""
    oracle.build()
    torch.cuda.set_device(0)
    rs = np.random.RandomState(seed)
    t_end = time.time() + seconds
    n_cases = n_calls = n_bad = 0
    by_m = {}
    while time.time() < t_end:
        M = int(rs.choice([8, 16, 16, 16, 32, 64, 12, 24, 128]))
        dsub = int(rs.choice([2, 4, 8])) if M < 128 else int(rs.choice([1, 2]))
        # (round 6: fewer code words than 256; uint16 codes -- the M = 8 byte-table shapes incl. their 64-key lists, M = 16 on the u16 tables,
        # the LDS-table / generic kernels for the rest)
        Ks = int(rs.choice([256, 256, 256, 100, 512, 700])) if M in (8, 12, 16) else int(rs.choice([256, 256, 100]))
        cdt = np.uint8 if Ks <= 256 else np.uint16
        N = int(np.exp(rs.uniform(np.log(1), np.log(600_000))))
        B = int(np.exp(rs.uniform(np.log(1), np.log(300))))
        B = max(1, min(B, int(1.5e9 / (N * M))))  # (the oracle's share of the time budget)
        k = int(rs.choice([1, 3, 10, 10, 16, 17, 33, 50, 64]))
        kind = int(rs.choice([1, 1, 3]))
        order = rs.choice(['iid', 'sorted', 'few_distinct', 'uniform_codes'])
        D = M * dsub
        r = 6
        A = rs.randn(r, D).astype(np.float32)
        cb = (rs.randn(Ks, r).astype(np.float32) @ A).reshape(Ks, M, dsub).transpose(1, 0, 2).copy()
        if order == 'uniform_codes':
            codes = rs.randint(0, Ks, size=(N, M)).astype(cdt)
            cb = rs.randn(M, Ks, dsub).astype(np.float32)
        elif order == 'few_distinct':
            base = rs.randint(0, Ks, size=(max(1, min(N, 300)), M)).astype(cdt)
            codes = base[rs.randint(0, base.shape[0], N)]
        else:
            z = rs.randn(N, r).astype(np.float32)
            if order == 'sorted':
                z = z[np.argsort(z[:, 0], kind='stable')]
            x = z @ A + 0.05 * rs.randn(N, D).astype(np.float32)
            codes = oracle.encode_c(x, cb, threads=oracle.max_threads()).astype(cdt)
        q = (rs.randn(B, r).astype(np.float32) @ A).astype(np.float32)
        if rs.rand() < 0.3:
            q += 0.7 * A[0]
        vmode = rs.choice(['none', 'random', 'head', 'most'])
        valid = np.ones(((N + 31) // 32 + 2) * 32, dtype=bool)
        valid[N:] = False
        if vmode == 'random':
            valid[:N] &= rs.rand(N) > 0.2
        elif vmode == 'head':
            valid[:N // 3] = False
        elif vmode == 'most':
            valid[:N] &= rs.rand(N) > 0.95
        live = np.nonzero(valid[:N])[0]
        if live.size == 0:
            continue
        k = min(k, int(live.size))
        omet = {1: oracle.EUCLIDEAN, 3: oracle.INNER_PRODUCT}[kind]
        lut = oracle.batch_precompute_adc_table_c(q, dsub, Ks, cb) if kind == 1 else oracle.get_dist_mat_c(q, cb, omet)
        rd, ri = oracle.adc_search_c(lut, codes[live], k, threads=oracle.max_threads())
        ri = live[ri]
        bits = None if vmode == 'none' else ops.to_dev(np.packbits(valid.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
        cb_d, q_d, codes_d = ops.to_dev(cb), ops.to_dev(q), ops.to_dev(codes)
        n_cases += 1
        by_m[M] = by_m.get(M, 0) + 1
        # (round 6) a third of the cases search with the table's scan state -- guarded first call, then the settled, unguarded kernel --,
        # and the opt-in MFMA-nominated seed is switched on for half of the cases (it applies to M = 16 / 128-d / >= 64 queries)
        mfma = rs.rand() < 0.5
        if mfma:
            os.environ['ANNLITE_MFMA_SEED'] = '1'
        else:
            os.environ.pop('ANNLITE_MFMA_SEED', None)
        _capi.knobs_reload()
        for layout in ((0, 1) if (M in (8, 16, 32, 64) and Ks <= 256) else (0,)):
            cd = ops.codes_skew(codes_d) if layout == 1 else codes_d
            outs = [('fused', ops.pq_search_topk(kind, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, valid_bits=bits))]
            if rs.rand() < 0.34:
                st = _capi.ScanState()
                for rep in range(3):
                    o3 = ops.pq_search_topk(kind, q_d, cb_d, cd, k, M, Ks, codes_layout=layout, valid_bits=bits, state=st)
                    torch.cuda.synchronize()
                outs.append(('fused+state', o3))
            plan = scan_plan(N, M, Ks, codes.dtype.itemsize, B, k)
            lt = ops.lut_build(q_d, cb_d, kind, LAYOUT_TILED if plan.fast else LAYOUT_BMK, plan.qi)
            outs.append(('prebuilt', ops.adc_scan_topk(cd, lt, B, k, M, Ks, codes_layout=layout, valid_bits=bits)))
            for name, (d, i) in outs:
                n_calls += 1
                if not (np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd, equal_nan=True)):
                    n_bad += 1
                    print('MISMATCH', dict(M=M, dsub=dsub, Ks=Ks, mfma=mfma, N=N, B=B, k=k, kind=kind, order=str(order), valid=str(vmode), layout=layout, entry=name), flush=True)
    os.environ.pop('ANNLITE_MFMA_SEED', None)
    _capi.knobs_reload()
    return n_cases, n_calls, n_bad, dict(sorted(by_m.items()))


def run_cells(seconds, seed, verbose=True, candidates=True):
    """Random pruned searches over cells through the C ABI (annlite_ivf_search_topk, the byte-table cell tiles) against the oracle's
    restatement of CellContainer.ivf_search (container.py:88-144) over the probed cells, bit for bit: random cell sizes (empty and
    one-row cells included), ANY distinct probed cells in any order (the first one seeds the bound), both table layouts, both table
    kinds, validity bitmaps, Ks below 256, heavy ties.  candidates: every other call also runs annlite_ivf_search_candidates on the same
    inputs and checks its lists (check_candidate_lists).  Returns (cases, calls, mismatches)."""
    oracle.build()
    torch.cuda.set_device(0)
    rs = np.random.RandomState(seed)
    t_end = time.time() + seconds
    n_cases = n_calls = n_bad = 0
    M = 16
    while time.time() < t_end:
        dsub = int(rs.choice([4, 8, 16]))
        Ks = int(rs.choice([256, 256, 100, 37]))
        N = int(np.exp(rs.uniform(np.log(1), np.log(300_000))))
        C = int(rs.choice([1, 2, 7, 32, 100, 256]))
        P = int(rs.randint(1, C + 1)) if C <= 7 else int(rs.choice([1, 2, 5, 16, min(C, 40)]))
        B = int(np.exp(rs.uniform(np.log(1), np.log(1100))))
        B = max(1, min(B, int(2e9 / (max(N * P // C, 1) * M + 1)), 1100))
        k = int(rs.choice([1, 3, 10, 10, 16]))
        kind = int(rs.choice([1, 1, 3]))
        D = M * dsub
        r = 6
        A = rs.randn(r, D).astype(np.float32)
        cb = (rs.randn(Ks, r).astype(np.float32) @ A).reshape(Ks, M, dsub).transpose(1, 0, 2).copy()
        order = rs.choice(['iid', 'few_distinct', 'uniform_codes'])
        if order == 'uniform_codes':
            codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
        elif order == 'few_distinct':
            base = rs.randint(0, Ks, size=(max(1, min(N, 50)), M)).astype(np.uint8)
            codes = base[rs.randint(0, base.shape[0], N)]
        else:
            x = rs.randn(N, r).astype(np.float32) @ A + 0.05 * rs.randn(N, D).astype(np.float32)
            codes = oracle.encode_c(x, cb, threads=oracle.max_threads()).astype(np.uint8)
        q = (rs.randn(B, r).astype(np.float32) @ A).astype(np.float32)
        # cells of uneven size (some empty): a row's cell by a skewed draw
        w = rs.rand(C) ** 3 + 1e-3
        w[rs.rand(C) < 0.1] = 0.0
        if w.sum() == 0:
            w[0] = 1.0
        cell_of = rs.choice(C, size=N, p=w / w.sum()).astype(np.int32)
        probe = np.stack([rs.permutation(C)[:P] for _ in range(B)]).astype(np.int32)
        vmode = rs.choice(['none', 'random', 'most'])
        valid = np.ones(N, dtype=bool)
        if vmode == 'random':
            valid &= rs.rand(N) > 0.2
        elif vmode == 'most':
            valid &= rs.rand(N) > 0.95
        omet = {1: oracle.EUCLIDEAN, 3: oracle.INNER_PRODUCT}[kind]
        rd, ri = oracle.ivf_search(q, cb, codes, cell_of, probe, omet, k, valid=valid, sqrt_euclidean=False)
        # the cell-sorted table: every cell starts at a multiple of 64 rows, ascending ids inside a cell
        counts = np.bincount(cell_of, minlength=C)
        padded = (counts + 63) // 64 * 64
        begin = np.cumsum(padded) - padded
        srt = np.argsort(cell_of, kind='stable')
        rank = np.arange(N) - (np.cumsum(counts) - counts)[cell_of[srt]]
        pos = begin[cell_of[srt]] + rank
        Nt = max(64, int(padded.sum()))
        table = np.zeros((Nt, M), dtype=np.uint8)
        table[pos] = codes[srt]
        row_ids = np.full(Nt, -1, dtype=np.int64)
        row_ids[pos] = srt
        vt = np.zeros(((Nt + 31) // 32 + 2) * 32, dtype=bool)
        vt[pos] = valid[srt]
        bits = None if vmode == 'none' else ops.to_dev(np.packbits(vt.reshape(-1, 32), axis=1, bitorder='little').view(np.int32).reshape(-1))
        cell_rows = ops.to_dev(np.stack([begin, begin + counts], axis=1).astype(np.int64))
        cell_order = ops.to_dev(np.argsort(-counts, kind='stable').astype(np.int32))
        table_d = ops.to_dev(table)
        n_cases += 1
        for layout in (0, 1):
            td = ops.codes_skew(table_d) if layout == 1 else table_d
            d, i = ops.ivf_search_topk(kind, ops.to_dev(q), ops.to_dev(cb), td, ops.to_dev(probe), C, cell_rows, cell_order, k, M, Ks,
                                       row_ids=ops.to_dev(row_ids), valid_bits=bits, n_rows=Nt, codes_layout=layout)
            n_calls += 1
            if not (np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd, equal_nan=True)):
                n_bad += 1
                print('MISMATCH cells', dict(dsub=dsub, Ks=Ks, N=N, C=C, P=P, B=B, k=k, kind=kind, order=str(order), valid=str(vmode), layout=layout),
                      flush=True)
            # the same pipeline as the re-rank's candidate generator (annlite_ivf_search_candidates, private lists): every (query, cell)
            # list is a PREFIX of the oracle's ranking of that cell alone, the nearest cell's list is complete, and the union of a query's
            # lists holds the oracle's top-k of the probed cells
            if candidates and (n_cases + layout) % 2 == 0 and B * P * N <= 3e8:  # (the check is P oracle passes)
                ids = ops.ivf_search_candidates(kind, ops.to_dev(q), ops.to_dev(cb), td, ops.to_dev(probe), C, cell_rows, cell_order, k, M, Ks,
                                                row_ids=ops.to_dev(row_ids), valid_bits=bits, n_rows=Nt, codes_layout=layout,
                                                bound_rank=int(rs.choice([1, 2, 4, 64]))).cpu().numpy()
                n_calls += 1
                why = check_candidate_lists(ids.reshape(B, P, k), q, cb, codes, cell_of, probe, omet, k, valid, ri)
                if why:
                    n_bad += 1
                    print('MISMATCH candidates (%s)' % why, dict(dsub=dsub, Ks=Ks, N=N, C=C, P=P, B=B, k=k, kind=kind, order=str(order),
                                                                 valid=str(vmode), layout=layout), flush=True)
                # ... and with every probed cell in S parts (IvfPQGpuIndex.rerank_split's cell table: entries C + c S + s = contiguous row
                # ranges of cell c, first rows at multiples of 64): the same promises per PART
                S = int(rs.choice([2, 3, 4]))
                if B * P * S * N <= 3e8 and C * (1 + S) <= 16384:
                    chunk = ((counts + S - 1) // S + 63) // 64 * 64
                    rank_of_row = np.empty(N, dtype=np.int64)
                    rank_of_row[srt] = rank
                    vcell = (C + cell_of.astype(np.int64) * S + rank_of_row // np.maximum(chunk[cell_of], 1)).astype(np.int64)
                    sidx = np.arange(S)[None, :]
                    lo = np.minimum(sidx * chunk[:, None], counts[:, None])
                    hi = np.minimum((sidx + 1) * chunk[:, None], counts[:, None])
                    some = hi > lo
                    lo, hi = np.where(some, lo, 0), np.where(some, hi, 0)
                    parts = np.stack([begin[:, None] + lo, begin[:, None] + hi], axis=2).reshape(-1, 2)
                    rows_x = np.concatenate([np.stack([begin, begin + counts], axis=1), parts]).astype(np.int64)
                    order_x = np.argsort(-(rows_x[:, 1] - rows_x[:, 0]), kind='stable').astype(np.int32)
                    probe_x = (C + probe[:, :, None].astype(np.int64) * S + sidx[None]).reshape(B, -1).astype(np.int32)
                    # (seed_cells: the first bound from the WHOLE nearest cell -- entry probe[b][0] -- instead of its first part: the first
                    # list may then be cut like the others)
                    whole_seed = bool(rs.rand() < 0.5)
                    ids = ops.ivf_search_candidates(kind, ops.to_dev(q), ops.to_dev(cb), td, ops.to_dev(probe_x), C * (1 + S), ops.to_dev(rows_x),
                                                    ops.to_dev(order_x), k, M, Ks, row_ids=ops.to_dev(row_ids), valid_bits=bits, n_rows=Nt,
                                                    codes_layout=layout, bound_rank=int(rs.choice([1, 2, 4])),
                                                    seed_cells=ops.to_dev(np.ascontiguousarray(probe[:, 0])) if whole_seed else None).cpu().numpy()
                    n_calls += 1
                    why = check_candidate_lists(ids.reshape(B, P * S, k), q, cb, codes, vcell, probe_x, omet, k, valid, ri,
                                                first_complete=not whole_seed)
                    if why:
                        n_bad += 1
                        print('MISMATCH candidates in %d parts (%s)' % (S, why), dict(dsub=dsub, Ks=Ks, N=N, C=C, P=P, B=B, k=k, kind=kind,
                                                                                    order=str(order), valid=str(vmode), layout=layout), flush=True)
    return n_cases, n_calls, n_bad


def check_candidate_lists(ids, q, cb, codes, cell_of, probe, omet, k, valid, top, first_complete=True):
    """'' if ids [B][P][k] (annlite_ivf_search_candidates) are what the header promises, else the first broken promise."""
    B, P, _ = ids.shape
    n = (ids >= 0).sum(axis=2)
    if not (np.sort(ids >= 0, axis=2)[:, :, ::-1] == (ids >= 0)).all():
        return 'a gap inside a list'
    for p in range(P):
        _, own = oracle.ivf_search(q, cb, codes, cell_of, probe[:, p:p + 1], omet, k, valid=valid, sqrt_euclidean=False)
        mask = np.arange(k)[None, :] < n[:, p, None]
        if not np.array_equal(np.where(mask, ids[:, p], -1), np.where(mask, own, -1)):
            return 'list of probe %d is not a prefix of the cell\'s own ranking' % p
        if p == 0 and first_complete and not np.array_equal(n[:, 0], (own >= 0).sum(axis=1)):
            return 'the nearest cell\'s list is cut'
    flat = ids.reshape(B, -1)
    for b in range(B):
        want = top[b][top[b] >= 0]
        if not np.isin(want, flat[b]).all():
            return 'query %d: the union of the lists misses a row of the exact top-k' % b
    return ''


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--seconds', type=float, default=200)
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--cells', action='store_true', help='the pruned search over cells (annlite_ivf_search_topk) instead of the flat search')
    a = p.parse_args()
    if a.cells:
        n_cases, n_calls, n_bad = run_cells(a.seconds, a.seed)
        print('fuzz_parity (cells): seed %d, %d cases, %d calls compared with the oracle bit for bit, %d mismatches' % (a.seed, n_cases, n_calls, n_bad))
        sys.exit(1 if n_bad else 0)
    n_cases, n_calls, n_bad, by_m = run(a.seconds, a.seed)
    print('fuzz_parity: seed %d, %d cases (by M: %s), %d calls compared with the oracle bit for bit, %d mismatches' % (a.seed, n_cases, by_m, n_calls, n_bad))
    sys.exit(1 if n_bad else 0)
