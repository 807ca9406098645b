"""CPU, world_size 2, gloo: the multi-GPU search path's partition -> scan -> all-gather -> merge
logic (annlite_amd/sharded.py) with the oracle standing in for the HIP scan kernel."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import pq_oracle
    from annlite_amd.sharded import ShardedSearcher, numpy_merge, shard_range

    rs = np.random.RandomState(0)  # same data on every rank
    N, M, Ks, B, k = 5003, 16, 256, 11, 10
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[100:140] = codes[100]  # ties across the shard boundary region
    codes[2500:2510] = codes[100]
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lo, hi = shard_range(N, world, rank)

    def scan(queries_lut, kk):
        d, i = pq_oracle.adc_search_c(queries_lut, codes[lo:hi], kk, id_base=lo)
        return torch.from_numpy(d), torch.from_numpy(i)

    d, i = ShardedSearcher(scan, numpy_merge).search(lut, k)
    rd, ri = pq_oracle.adc_search_c(lut, codes, k)
    ok = bool(np.array_equal(d.numpy(), rd) and np.array_equal(i.numpy(), ri))
    # a shard smaller than k pads with (+inf, -1) and must not pollute the merge
    small = codes[:13]
    lo2, hi2 = shard_range(13, world, rank)

    def scan2(queries_lut, kk):
        d, i = pq_oracle.adc_search_c(queries_lut, small[lo2:hi2], kk, id_base=lo2)
        return torch.from_numpy(d), torch.from_numpy(i)

    d2, i2 = ShardedSearcher(scan2, numpy_merge).search(lut, k)
    rd2, ri2 = pq_oracle.adc_search_c(lut, small, k)
    ok = ok and bool(np.array_equal(d2.numpy(), rd2) and np.array_equal(i2.numpy(), ri2))
    # ---- the PRODUCT's exchange: ShardedPQIndex.search_batch_async's packed path (ONE all-gather of [B, k, 2] int64 =
    # (global id, bits of the raw ADC sum), merge on the raw sums, sqrt epilogue last) -- the local scan injected
    from annlite_amd.sharded import ShardedPQIndex, numpy_merge_packed

    class FakeShard:  # what ShardedPQIndex needs of PQFlatGpuIndex
        sqrt_epilogue = True  # EUCLIDEAN: hnsw/index.py:164-165

        def __init__(self, rows, base):
            self.rows, self.base = rows, base

        def search_batch_packed(self, queries_lut, kk, row_base):
            assert row_base == self.base
            d, i = pq_oracle.adc_search_c(queries_lut.numpy(), self.rows, kk, id_base=row_base)
            out = np.empty(d.shape + (2,), dtype=np.int64)
            out[..., 0] = i  # (a shard shorter than k pads with -1 / +inf)
            out[..., 1] = d.view(np.uint32).astype(np.int64)
            return torch.from_numpy(out)

    lut_t = torch.from_numpy(lut)
    for rows_all in (codes, codes[:13]):  # ties across the shard boundary; a shard smaller than k
        l3, h3 = shard_range(rows_all.shape[0], world, rank)
        sh = ShardedPQIndex(FakeShard(rows_all[l3:h3], l3), row_base=l3, merge=numpy_merge, merge_packed=numpy_merge_packed)
        pd, pi = sh.search_batch_async(lut_t, limit=k).result()
        rd3, ri3 = pq_oracle.adc_search_c(lut, rows_all, k)
        ok = ok and bool(np.array_equal(pd.numpy(), np.sqrt(rd3)) and np.array_equal(pi.numpy(), ri3))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_search_world2_gloo(oracle):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
