"""Row-sharded PQ/ADC search across the GPUs of one node: one process per GPU
(``torch.distributed``, backend "nccl" == RCCL over xGMI), each rank scans its own contiguous slice
of the code table and keeps a local top-k; ONE all-gather of ``[B, k]`` (f32 distance, i64 global
row id) per batch, then every rank merges the ``G`` lists with the same (distance, id) order rule.

The reference has no collective of any kind; its only multi-worker mode is Jina ``shards=N`` with
``polling ALL`` and a gateway-side merge (tests/executor/test_executor.py:326-350), whose per-cell
analogue inside one process is the hstack + argsort merge of annlite/container.py:130-138.  The
exchange is 12 B * B * k per rank (120 KB at B=1024, k=10): latency-bound, so a single
``all_gather_into_tensor`` per tensor is used and nothing is bucketed.

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
    # concatenation form ([G*B, k]): accepted by both RCCL and gloo; viewed as [G, B, k] for the merge
    all_d = torch.empty((G * B, k), dtype=local_d.dtype, device=local_d.device)
    all_i = torch.empty((G * B, k), dtype=local_i.dtype, device=local_i.device)
    dist.all_gather_into_tensor(all_d, local_d.contiguous(), group=group)
    dist.all_gather_into_tensor(all_i, local_i.contiguous(), group=group)
    return merge_fn(all_d.view(G, B, k), all_i.view(G, B, k))


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
                 merge: Optional[Callable] = None, merge_packed: Optional[Callable] = None):
        self.index = index
        self.row_base = int(row_base)
        self.group = group
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
        packed = self.index.search_batch_packed(queries, limit, self.row_base) if gather and isinstance(
            queries, torch.Tensor) else None
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
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=packed.device)
        scanned = torch.cuda.Event()
        scanned.record(torch.cuda.current_stream(packed.device))
        with torch.cuda.stream(self._xstream):
            self._xstream.wait_event(scanned)
            packed.record_stream(self._xstream)
            gathered = torch.empty((G * B, k, 2), dtype=torch.int64, device=packed.device)
            work = dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=True)
            work.wait()  # the SIDE stream waits for the collective
            value = self._merge_packed(gathered.view(G, B, k, 2), sqrt=self.index.sqrt_epilogue)
            done = torch.cuda.Event()
            done.record(self._xstream)
        return PendingSearch(self, value=value, done=done, keep=(packed, gathered))


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
