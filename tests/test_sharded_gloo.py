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


# ---------------------------------------------------------------------------------------------------------------------
# the row-sharded search WITH the seed exchange (ShardedPQIndex._split_search / _exchange_seeds): two ranks, the
# product's collectives (a second process group for the seeds, the result all-gather on the first), the kernels restated in numpy
def _split_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import pq_oracle
    from annlite_amd.sharded import SEED_KEYS, ShardedPQIndex, numpy_merge, numpy_merge_packed, shard_range

    rs = np.random.RandomState(1)
    N, M, Ks, B, k = 40_000, 16, 256, 9, 10
    codes = rs.randint(0, Ks, size=(N, M)).astype(np.uint8)
    codes[19_990:20_010] = codes[7]  # ties across the shard boundary
    lo, hi = shard_range(N, world, rank)
    NONE = np.uint64(0xFFFFFFFFFFFFFFFF)

    def key_of(d):  # monotone u64 image of a non-negative f32 distance, as a bound: (bits + 2) << 32 (the kernels' form)
        return (d.astype(np.float32).view(np.uint32).astype(np.uint64) + np.uint64(2)) << np.uint64(32)

    class FakeShard:
        sqrt_epilogue = True
        _n_rows = hi - lo

        def __init__(self):
            self.batch = 0
            self.log = []

        def _dist(self, lut):
            return np.stack([pq_oracle.dist_pqcodes_to_codebooks_c(lut[b], codes[lo:hi]) for b in range(lut.shape[0])])

        def _packed(self, lut, bound=None):
            d = self._dist(lut)
            out = np.empty((lut.shape[0], k, 2), dtype=np.int64)
            for b in range(lut.shape[0]):
                dd = d[b]
                order = np.lexsort((np.arange(dd.shape[0]), dd))
                if bound is not None:  # the scan under a bound: rows above it never reach the lists
                    order = order[key_of(dd[order]) <= bound[b]]
                order = order[:k]
                ids = np.full(k, -1, np.int64)
                ds = np.full(k, np.inf, np.float32)
                ids[:len(order)], ds[:len(order)] = order + lo, dd[order]
                out[b, :, 0], out[b, :, 1] = ids, ds.view(np.uint32).astype(np.int64)
            return torch.from_numpy(out)

        def search_batch_packed(self, lut_t, kk, row_base):
            return self._packed(lut_t.numpy())

        def split_prepare(self, lut_t, kk, row_base, seed_rows, workspace):
            lut = lut_t.numpy()
            self.batch += 1
            settled = not (rank == 1 and self.batch <= 2)  # rank 1's kernel choice settles two batches later than rank 0's
            keys = None
            if settled:
                d = self._dist(lut)[:, :seed_rows]
                kk_ = np.full((lut.shape[0], SEED_KEYS), NONE, dtype=np.uint64)
                kk_[:, :k] = key_of(np.sort(d, axis=1)[:, :k])
                keys = torch.from_numpy(kk_.view(np.int64))
            shard = self

            class Batch:
                n_queries = lut.shape[0]
                device = torch.device('cpu')

                def union(self, all_keys):
                    u = all_keys.numpy().view(np.uint64)  # [G, B, 16]
                    G = u.shape[0]
                    flat = np.sort(u.transpose(1, 0, 2).reshape(lut.shape[0], G * SEED_KEYS), axis=1)
                    self.bound = flat[:, k - 1]
                    shard.log.append((G, int((u[:, 0, 0] != NONE).sum())))

                def scan(self):
                    return shard._packed(lut, self.bound)

                def plain(self):
                    return shard._packed(lut)

            bt = Batch()
            bt.keys = keys
            return bt

    ok = True
    sh = ShardedPQIndex(FakeShard(), row_base=lo, merge=numpy_merge, merge_packed=numpy_merge_packed, n_total=N, seed_exchange=True)
    ok = ok and sh.seed_rows() == 4096  # (N / 32 clamped to 8192, over two ranks, not below 4096)
    for step in range(5):  # several batches: the seed collective and the result collective alternate, on their own groups
        lut = rs.rand(B, M, Ks).astype(np.float32)
        pd, pi = sh.search_batch_async(torch.from_numpy(lut), limit=k).result()
        rd, ri = pq_oracle.adc_search_c(lut, codes, k)
        ok = ok and bool(np.array_equal(pd.numpy(), np.sqrt(rd)) and np.array_equal(pi.numpy(), ri))
    # rank 1 sat out two exchanges (contributing "no bound"), rank 0 none; from batch 3 on both ranks' keys are in the union
    log = sh.index.log
    ok = ok and sh._seed_group is not None and len(log) == (5 if rank == 0 else 3)
    ok = ok and all(g == 2 for g, _ in log) and [n for _, n in log][-3:] == [2, 2, 2]
    if rank == 0:
        ok = ok and [n for _, n in log][:2] == [1, 1]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_search_with_seed_exchange_world2_gloo(oracle):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---------------------------------------------------------------------------------------------------------------------
# round 5: world sizes 4 and 8 (what the 8-GPU node will run), through BOTH exchanges -- ShardedSearcher's general gather and
# ShardedPQIndex's packed one -- on tables that do not divide by the rank count, leave a rank EMPTY, leave every rank shorter
# than k, and (seed exchange) leave ranks shorter than their share of the seed rows
def _wide_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import pq_oracle
    from annlite_amd.core.index.multi_gpu import merge_lists_sorted
    from annlite_amd.sharded import SEED_KEYS, ShardedPQIndex, ShardedSearcher, numpy_merge, numpy_merge_packed, shard_range

    rs = np.random.RandomState(3)  # same data on every rank
    M, Ks, B = 16, 256, 7
    full = rs.randint(0, Ks, size=(5003, M)).astype(np.uint8)
    full[600:660] = full[600]    # ties across the first shard boundaries of both world sizes (626 / 1251)
    full[1240:1260] = full[600]
    lut = rs.rand(B, M, Ks).astype(np.float32)
    lut_t = torch.from_numpy(lut)
    ok = True
    why = []

    class FakeShard:
        sqrt_epilogue = True

        def __init__(self, rows, base):
            self.rows, self.base = rows, base

        def search_batch_packed(self, queries_lut, kk, row_base):
            d, i = pq_oracle.adc_search_c(queries_lut.numpy(), self.rows, kk, id_base=row_base)
            out = np.empty(d.shape + (2,), dtype=np.int64)
            out[..., 0], out[..., 1] = i, d.view(np.uint32).astype(np.int64)
            return torch.from_numpy(out)

    # N: not divisible by G / one row per rank and EMPTY ranks (N < G at world 8) / every rank shorter than k / empty table
    for N in (5003, 5, 13, 2 * world + 1, 0):
        rows_all = full[:N]
        lo, hi = shard_range(N, world, rank)
        for k in (10, 1, 33):
            rd, ri = pq_oracle.adc_search_c(lut, rows_all, k)

            def scan(queries_lut, kk):
                d, i = pq_oracle.adc_search_c(queries_lut, rows_all[lo:hi], kk, id_base=lo)
                return torch.from_numpy(d), torch.from_numpy(i)

            d, i = ShardedSearcher(scan, numpy_merge).search(lut, k)
            good = np.array_equal(d.numpy(), rd) and np.array_equal(i.numpy(), ri)
            # ... the same gather with the product's any-k merge (MultiGpuPQIndex / limit > 64: merge_lists_sorted)
            d2, i2 = ShardedSearcher(scan, lambda ad, ai: merge_lists_sorted(ad, ai, k)).search(lut, k)
            good = good and np.array_equal(d2.numpy(), rd) and np.array_equal(i2.numpy(), ri)
            sh = ShardedPQIndex(FakeShard(rows_all[lo:hi], lo), row_base=lo, merge=numpy_merge, merge_packed=numpy_merge_packed)
            pd, pi = sh.search_batch_async(lut_t, limit=k).result()
            good = good and np.array_equal(pd.numpy(), np.sqrt(rd)) and np.array_equal(pi.numpy(), ri)
            if not good:
                ok = False
                why.append(('gather', N, k))

    # ---- the seed exchange at this world size: ranks whose shard is shorter than seed_rows() (and, at the small table, empty
    # ranks) contribute the keys they have; the k-th smallest of the union bounds every rank's scan
    NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
    k = 10

    def key_of(d):
        return (d.astype(np.float32).view(np.uint32).astype(np.uint64) + np.uint64(2)) << np.uint64(32)

    for N in (5003, world + 3):  # 5003 / G < 4096 seed rows: every rank is shorter than its seed; world + 3: ranks of 2, 1 and 0 rows
        rows_all = full[:N]
        lo, hi = shard_range(N, world, rank)

        class SplitShard(FakeShard):
            _n_rows = hi - lo

            def split_prepare(self, lut_tt, kk, row_base, seed_rows, workspace):
                lq = lut_tt.numpy()
                mine = self.rows[:seed_rows]  # (shorter than seed_rows: all the rank has)
                dd = np.stack([pq_oracle.dist_pqcodes_to_codebooks_c(lq[b], mine) for b in range(lq.shape[0])]) if len(mine) else \
                    np.zeros((lq.shape[0], 0), np.float32)
                kk_ = np.full((lq.shape[0], SEED_KEYS), NONE, dtype=np.uint64)
                n = min(k, dd.shape[1])
                if n:
                    kk_[:, :n] = key_of(np.sort(dd, axis=1)[:, :n])
                shard = self

                class Batch:
                    n_queries = lq.shape[0]
                    keys = torch.from_numpy(kk_.view(np.int64))

                    def union(self, all_keys):
                        u = all_keys.numpy().view(np.uint64)
                        flat = np.sort(u.transpose(1, 0, 2).reshape(lq.shape[0], -1), axis=1)
                        self.bound = flat[:, k - 1]  # (NONE when the whole table has fewer than k rows: no bound)

                    def scan(self):
                        d, i = pq_oracle.adc_search_c(lq, shard.rows, k, id_base=shard.base)
                        keep = key_of(d) <= self.bound[:, None]  # rows above the bound never reach the lists
                        i = np.where(keep, i, -1)
                        d = np.where(keep, d, np.inf).astype(np.float32)
                        out = np.empty(d.shape + (2,), dtype=np.int64)
                        out[..., 0], out[..., 1] = i, d.view(np.uint32).astype(np.int64)
                        return torch.from_numpy(out)

                    def plain(self):
                        return shard.search_batch_packed(lut_tt, k, shard.base)

                return Batch()

        sh = ShardedPQIndex(SplitShard(rows_all[lo:hi], lo), row_base=lo, merge=numpy_merge, merge_packed=numpy_merge_packed,
                            n_total=max(N, 1), seed_exchange=True)
        for _ in range(2):
            pd, pi = sh.search_batch_async(lut_t, limit=k).result()
            rd, ri = pq_oracle.adc_search_c(lut, rows_all, k)
            if not (np.array_equal(pd.numpy(), np.sqrt(rd)) and np.array_equal(pi.numpy(), ri)):
                ok = False
                why.append(('seed', N))
    q.put((rank, ok, why))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('world', [4, 8])
def test_sharded_search_world4_world8_gloo(oracle, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True, []) for r in range(world)], res
