"""Row-sharded PQ/ADC search across the GPUs of one node: one process per GPU
(``torch.distributed``, backend "nccl" == RCCL over xGMI), each rank scans its own contiguous slice
of the code table and keeps a local top-k; ONE all-gather of ``[B, k]`` (f32 distance, i64 global
row id) per batch, then every rank merges the ``G`` lists with the same (distance, id) order rule.

The reference has no collective of any kind; its only multi-worker mode is Jina ``shards=N`` with
``polling ALL`` and a gateway-side merge (tests/executor/test_executor.py:326-350), whose per-cell
analogue inside one process is the hstack + argsort merge of annlite/container.py:130-138.  The
exchange is 16 B * B * k per rank (160 KB at B=1024, k=10): latency-bound, so ONE
``all_gather_into_tensor`` per batch is used and nothing is bucketed.

The scan and the merge are injected callables so that the partition / gather logic can be covered
by world_size-2 ``gloo`` tests on CPU (tests/test_sharded_gloo.py) with the oracle standing in for
the kernels; the product wiring (``ShardedPQIndex``) always uses the HIP kernels.
"""
import os
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row ranges: rank g holds rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G)))
    (global id = row_base + local row; row ids are insertion offsets, storage/table.py:251-257)."""
    per = (n_total + world_size - 1) // world_size
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def gather_and_merge(local_d: torch.Tensor, local_i: torch.Tensor, merge_fn: Callable,
                     group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """all-gather the per-shard top-k lists and merge them; every rank returns the global result."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_d, local_i
    if dist.get_world_size(group) == 1 and not os.environ.get('ANNLITE_FORCE_GATHER'):
        return local_d, local_i  # (the env switch lets a 1-GPU box exercise the RCCL + merge path)
    G = dist.get_world_size(group)
    B, k = local_d.shape
    # ONE collective here too (the general path: re-rank, limit > 64): (id, bits of the f32 distance) pairs in one int64 buffer,
    # concatenation form ([G*B, k, 2]: accepted by both RCCL and gloo), unpacked into [G, B, k] for the merge
    # (the distance's own bits: f32 in the low word, f64 as the whole int64 -- a float64 caller gets float64 back)
    f64 = local_d.dtype == torch.float64
    bits = local_d.contiguous().view(torch.int64) if f64 else local_d.to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    both = torch.stack([local_i.to(torch.int64), bits], dim=2)
    gathered = _all_gather_rows(both.contiguous(), G, group)
    all_i = gathered[..., 0].contiguous().view(G, B, k)
    all_d = (gathered[..., 1].contiguous().view(torch.float64) if f64 else gathered[..., 1].to(torch.int32).view(torch.float32)).view(G, B, k)
    return merge_fn(all_d.contiguous(), all_i)


def _through_host(t: torch.Tensor, group) -> bool:
    """Device tensors under the gloo backend: the collective runs on host copies (gloo's support for device tensors differs per
    collective).  That configuration exists for tests -- two ranks sharing one GPU, which RCCL refuses -- not for production."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _all_gather_rows(t: torch.Tensor, G: int, group) -> torch.Tensor:
    """ONE ``all_gather_into_tensor`` in concatenation form: ``[n, ...] -> [G * n, ...]`` on ``t``'s device."""
    if _through_host(t, group):
        h = t.cpu()
        out = torch.empty((G * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h, group=group)
        return out.to(t.device)
    out = torch.empty((G * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


SEED_KEYS = 16  # keys per query in the ranks' seed exchange (annlite_hip.h: ANNLITE_SEED_KEYS)


class ShardedSearcher:
    """``scan_fn(queries, k) -> (dist [B,k], global ids [B,k])`` over the local shard, then the
    gather + ``merge_fn([G,B,k], [G,B,k]) -> ([B,k], [B,k])``."""

    def __init__(self, scan_fn: Callable, merge_fn: Callable, group: Optional[dist.ProcessGroup] = None):
        self.scan_fn = scan_fn
        self.merge_fn = merge_fn
        self.group = group

    def search(self, queries, k: int):
        d, i = self.scan_fn(queries, k)
        return gather_and_merge(d, i, self.merge_fn, self.group)


class ShardedPQIndex:
    """Product wiring: a ``PQFlatGpuIndex`` per rank holding rows [row_base, row_base + n_local) of
    the global table; ``search_batch`` returns global row ids on every rank."""

    def __init__(self, index, row_base: int, group: Optional[dist.ProcessGroup] = None,
                 merge: Optional[Callable] = None, merge_packed: Optional[Callable] = None, seed_exchange: bool = False,
                 n_total: Optional[int] = None, seed_group: Optional[dist.ProcessGroup] = None):
        self.index = index
        self.row_base = int(row_base)
        self.group = group
        # SEED EXCHANGE (opt-in): every rank repeats the per-batch work for all B queries, and the largest part
        # of it is the seed bound -- the exact k-th distance of the table's first ~32768 rows.  With the exchange a rank seeds from
        # 1/G of those rows, the ranks all-gather the bounds of their seeds' k smallest rows ([B, 16] keys, 128 KB at 1024
        # queries: one more small collective per batch, on its OWN process group so that it never queues behind a result
        # all-gather that waits for a scan) and every rank starts its scan with the k-th smallest of the union.  OFF by default:
        # measured on one MI355X (one rank, 7 emulated peers: bench.py --seed-exchange --emulate-seed-peers 8) the collective's
        # cross-stream hops between the preparation launch and the scan cost more than the smaller seed saves.
        self.seed_exchange = bool(seed_exchange)
        self.n_total = n_total  # rows of the whole table (default: this shard's rows x world size)
        # The seed exchange's own process group.  ``dist.new_group`` is collective over the WHOLE default group (members and
        # non-members alike), so it is made HERE -- construction with ``seed_exchange=True`` is collective over the world, as
        # torch.distributed requires -- never on the first batch's critical path.  A caller whose ``group`` is a strict
        # sub-group creates the seed group itself (every process of the world calling ``new_group``) and passes it in.
        self._seed_group = seed_group
        if self.seed_exchange and self._seed_group is None and dist.is_available() and dist.is_initialized():
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self._seed_group = dist.new_group(ranks=ranks)
        self._peer_keys = None  # (measurement aid, bench.py --emulate-seed-peers: precomputed key sets standing in for peers)
        if merge is None or merge_packed is None:  # the product: the merge kernels (tests inject numpy restatements)
            from . import ops

            merge = merge or ops.topk_merge
            merge_packed = merge_packed or ops.topk_merge_packed
        self._merge = merge
        self._merge_packed = merge_packed
        self._xstream = None  # side stream of the exchange (all-gather + merge)

    def _scan(self, queries, k):
        return self.index.search_batch(queries, limit=k, row_base=self.row_base)

    def search_batch(self, queries: torch.Tensor, limit: int = 10):
        return self.search_batch_async(queries, limit).result()

    def search_batch_async(self, queries: torch.Tensor, limit: int = 10) -> 'PendingSearch':
        """Enqueue the local scan and start the exchange; ``.result()`` merges.  The exchange (all-gather + merge)
        runs on a side stream that waits for the scan through an event; the compute stream itself never waits,
        so batch i+1's kernels follow batch i's scan back to back (a cross-stream wait on the compute stream
        measured ~15 us of idle time per batch on one MI355X, twice per batch)."""
        gather = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(self.group) > 1 or bool(os.environ.get('ANNLITE_FORCE_GATHER')))
        packed = None
        if gather and isinstance(queries, torch.Tensor):
            if self.seed_exchange and hasattr(self.index, 'split_prepare'):
                packed = self._split_search(queries, limit)
            if packed is None:
                packed = self.index.search_batch_packed(queries, limit, self.row_base)
        if packed is None:
            return PendingSearch(self, value=ShardedSearcher(self._scan, self._merge, self.group).search(queries, limit))
        # ONE collective per batch: (global id, raw ADC sum) pairs, 16 B each; merged on the raw sums (the
        # single-GPU order), the metric epilogue (sqrt for EUCLIDEAN) comes last
        G = dist.get_world_size(self.group)
        B, k, _ = packed.shape
        if not packed.is_cuda:  # host tensors (the gloo tests of this path): same exchange, no streams to overlap
            gathered = torch.empty((G * B, k, 2), dtype=torch.int64)
            dist.all_gather_into_tensor(gathered, packed.contiguous(), group=self.group)
            return PendingSearch(self, value=self._merge_packed(gathered.view(G, B, k, 2), sqrt=self.index.sqrt_epilogue))
        if _through_host(packed, self.group):  # (gloo with device tensors: synchronous, through the host -- tests only)
            gathered = _all_gather_rows(packed.contiguous(), G, self.group)
            return PendingSearch(self, value=self._merge_packed(gathered.view(G, B, k, 2), sqrt=self.index.sqrt_epilogue))
        return self._exchange_result(packed, None)

    def _exchange_result(self, packed: torch.Tensor, scanned) -> 'PendingSearch':
        G = dist.get_world_size(self.group)
        B, k, _ = packed.shape
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=packed.device)
        xs = self._xstream
        # (hand-rolled `with torch.cuda.stream(xs)`: ONE current_stream() per batch instead of the context manager's own + the
        # event's + three inside the ops -- python-level stream look-ups were a third of the 0.10 ms a batch cost the host with the
        # exchange on, profiles/r06/host_overhead.txt)
        cur = torch.cuda.current_stream(packed.device)
        if scanned is None:  # (the scan ran on the caller's stream: the side stream waits for what has been enqueued there so far)
            xs.wait_stream(cur)
        else:
            xs.wait_event(scanned)
        torch.cuda.set_stream(xs)
        try:
            packed.record_stream(xs)
            gathered = torch.empty((G * B, k, 2), dtype=torch.int64, device=packed.device)
            work = dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=True)
            work.wait()  # the SIDE stream waits for the collective
            value = self._merge_packed(gathered.view(G, B, k, 2), sqrt=self.index.sqrt_epilogue)
            done = torch.cuda.Event()
            done.record(xs)
        finally:
            torch.cuda.set_stream(cur)
        return PendingSearch(self, value=value, done=done, keep=(packed, gathered))

    # ------------------------------------------------------------------ seed exchange
    def seed_rows(self) -> int:
        """Rows of this rank's seed: the single-GPU rule (N / 32 clamped to [8192, 32768]) applied to the WHOLE table, dealt out
        over the ranks (not below 4096: the seed costs ~0.7 us per 1024 rows, the bound's rank is what the scan pays for)."""
        G = dist.get_world_size(self.group) + (0 if self._peer_keys is None else int(self._peer_keys.shape[0]))
        n_total = self.n_total or getattr(self.index, '_n_rows', 0) * G
        whole = min(32768, max(8192, -(-(n_total // 32) // 1024) * 1024))
        return max(4096, -(-(whole // G) // 1024) * 1024)

    def _split_search(self, queries: torch.Tensor, limit: int) -> Optional[torch.Tensor]:
        """This rank's packed result through the search in two halves (index.split_prepare), everything in the calling stream's
        order.  The per-batch protocol -- identical on every rank whatever its private state:
            batch = index.split_prepare(...)     None: the split is not for this configuration (every rank says so: no collective)
            all_keys = exchange(batch.keys or "no bound" keys)      ALWAYS: the ranks' collectives stay aligned although
                                                 `keys is None` (table too small, kernel choice not settled) is a per-rank fact
            batch.union(all_keys); batch.scan()  -- or batch.plain(), the ordinary search, where keys is None
        (Measured on one MI355X, DESIGN.md section 8: the collective between the two halves costs two cross-stream hops of
        15-40 us each on this runtime; a caller that alternates batches between two streams hides part of them.  A deeper
        pipeline -- preparation side, scans and result exchange on streams of their own, one call apart -- was built and measured
        SLOWER: HIP streams share a few in-order hardware queues, and a preparation launch needs every CU a scan holds.)"""
        batch = self.index.split_prepare(queries, limit, self.row_base, self.seed_rows(), None)
        if batch is None:
            return None
        keys = batch.keys if batch.keys is not None else torch.full((batch.n_queries, SEED_KEYS), -1, dtype=torch.int64,
                                                                    device=queries.device)
        all_keys = self._exchange_seeds(keys)
        if batch.keys is None:
            return batch.plain()
        batch.union(all_keys)
        return batch.scan()

    def _exchange_seeds(self, keys: torch.Tensor) -> torch.Tensor:
        """ONE all-gather of the ranks' seed keys [B, 16] -> [G, B, 16], in the calling (preparation) stream's order."""
        G = dist.get_world_size(self.group)
        if self._seed_group is None:
            raise RuntimeError('seed exchange without its process group: construct ShardedPQIndex(seed_exchange=True) after '
                               'init_process_group (collective over the world), or pass seed_group=')
        out = _all_gather_rows(keys.contiguous(), G, self._seed_group).view(G, keys.shape[0], keys.shape[1])
        if self._peer_keys is not None:
            out = torch.cat([out, self._peer_keys.to(out.device)], dim=0)
        return out


class PendingSearch:
    """Handle of one in-flight row-sharded batch (see ``ShardedPQIndex.search_batch_async``)."""

    def __init__(self, owner, value=None, done=None, keep=None):
        self._owner, self._value, self._done, self._keep = owner, value, done, keep

    def result(self, wait: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(dists [B,k], ids [B,k])``.  With ``wait`` (default) the current stream is made to wait for the
        exchange, so the tensors can be used right away; ``wait=False`` skips that (a throughput loop that only
        reads results after a device synchronisation keeps its compute stream free of waits)."""
        if self._done is not None and wait:
            torch.cuda.current_stream().wait_event(self._done)
            for t in self._value:
                t.record_stream(torch.cuda.current_stream())
        return self._value


def numpy_merge(all_d: torch.Tensor, all_i: torch.Tensor):
    """Reference merge used by the CPU (gloo) tests only: lexsort by (distance, id), -1 ids last."""
    G, B, k = all_d.shape
    d = all_d.permute(1, 0, 2).reshape(B, G * k).cpu().numpy()
    i = all_i.permute(1, 0, 2).reshape(B, G * k).cpu().numpy()
    od = np.empty((B, k), dtype=d.dtype)
    oi = np.empty((B, k), dtype=i.dtype)
    for b in range(B):
        key_i = np.where(i[b] < 0, np.iinfo(np.int64).max, i[b])
        order = np.lexsort((key_i, d[b]))[:k]
        od[b], oi[b] = d[b][order], i[b][order]
    return torch.from_numpy(od), torch.from_numpy(oi)


def numpy_merge_packed(gathered: torch.Tensor, sqrt: bool = False):
    """Restatement of ``annlite_topk_merge_packed`` (merge_lists_kernel) for the CPU (gloo) tests: ``gathered`` is
    [G, B, k, 2] int64 = (global row id or -1, bits of the raw fp32 ADC sum); merge by (sum, id) ascending, padding
    entries dropped, then the metric epilogue."""
    G, B, k, _ = gathered.shape
    g = gathered.permute(1, 0, 2, 3).reshape(B, G * k, 2).cpu().numpy()
    ids = g[:, :, 0]
    d = (g[:, :, 1] & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    od = np.full((B, k), np.inf, dtype=np.float32)
    oi = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        keep = np.nonzero(ids[b] >= 0)[0]
        order = keep[np.lexsort((ids[b][keep], d[b][keep]))][:k]
        od[b, :len(order)], oi[b, :len(order)] = d[b][order], ids[b][order]
    if sqrt:
        od = np.sqrt(od)
    return torch.from_numpy(od), torch.from_numpy(oi)
