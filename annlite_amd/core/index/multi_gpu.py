"""``MultiGpuPQIndex`` -- ONE index object, ONE process, G devices: the row-sharded PQ/ADC search behind the
reference's single-object API (``AnnLite(...).index()/search()``, annlite/index.py:334-359; SURVEY.md section 8e: "single
process driving G devices with one stream each ... matching AnnLite's single-object API").

Rows are dealt to the G shards block-cyclically (``block`` rows at a time: a table grows by appends -- row ids are
insertion offsets, storage/table.py:251-257 -- and every shard grows with it); shard g is a ``PQFlatGpuIndex`` living on
``devices[g]``.  A search hands every shard the whole query batch -- each device builds its tables and scans its rows on its
own stream, all G launched back to back from this thread, none waited for --, collects the G packed results
``[B, k, 2]`` (global row id, bits of the raw ADC sum) on the first device by peer copies and merges them with
``annlite_topk_merge_packed`` (the kernel the multi-process path runs after its RCCL all-gather, sharded.py): the same
(sum, id) order rule as one flat index, metric epilogue last -- bit-identical results.

The multi-PROCESS variant (one process per GPU, ``torch.distributed`` over RCCL) is ``annlite_amd.sharded``; this class is
its single-process sibling for callers that keep the reference's one-object API.  Shards and the merge are injectable
(``shard_factory`` / ``merge_packed``) so that the dealing, the id mapping, deletes, filters and the merge order are covered
by CPU tests with the oracle standing in for the kernels (tests/test_multi_gpu_facade.py).
"""
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ...enums import Metric
from .base import BaseIndex


class _OnDevice:
    """``with _OnDevice(dev):`` -- make ``dev`` the current HIP device (what ``ops`` allocates on); no-op for None / CPU."""

    def __init__(self, dev):
        self._ctx = torch.cuda.device(dev) if dev is not None and torch.cuda.is_available() else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def merge_lists_sorted(d: torch.Tensor, i: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """G lists ``[G, B, kk]`` of (distance, global id; -1 = padding) -> the k smallest per query under the (distance, id)
    order, padding last: a stable sort by id followed by a stable sort by distance.  Any k -- the merge of ``limit > 64``
    searches, which the wave-list kernel (``annlite_topk_merge``, k <= 64) does not take; device or host tensors."""
    G, B, kk = d.shape
    dd = d.permute(1, 0, 2).reshape(B, G * kk)
    ii = i.permute(1, 0, 2).reshape(B, G * kk)
    pad = ii < 0
    by_id = torch.argsort(torch.where(pad, torch.full_like(ii, torch.iinfo(torch.int64).max), ii), dim=1, stable=True)
    dd, ii, pad = torch.gather(dd, 1, by_id), torch.gather(ii, 1, by_id), torch.gather(pad, 1, by_id)
    dd = torch.where(pad, torch.full_like(dd, float('inf')), dd)
    # The library's order everywhere else (f32_to_key, annlite_topk_merge): numbers ascending, +inf, then a real row's NaN,
    # and only then "none".  torch sorts NaN behind +inf too -- i.e. behind the PADDING's +inf: a real row with a NaN distance
    # would be cut in favour of padding.  So: stable sort by the distance with NaN read as +inf (ties keep the id order, the
    # padding stays behind every real id), then a stable sort by class (number or +inf / NaN / padding).
    nan = torch.isnan(dd) & ~pad
    by_d = torch.argsort(torch.where(nan, torch.full_like(dd, float('inf')), dd), dim=1, stable=True)
    cls = torch.gather(nan.to(torch.int8) + 2 * pad.to(torch.int8), 1, by_d)
    by_d = torch.gather(by_d, 1, torch.argsort(cls, dim=1, stable=True))[:, :k]
    od, oi = torch.gather(dd, 1, by_d), torch.gather(ii, 1, by_d)
    if od.shape[1] < k:
        od = torch.cat([od, torch.full((B, k - od.shape[1]), float('inf'), dtype=od.dtype, device=od.device)], dim=1)
        oi = torch.cat([oi, torch.full((B, k - oi.shape[1]), -1, dtype=oi.dtype, device=oi.device)], dim=1)
    return od, oi


class MultiGpuPQIndex(BaseIndex):
    def __init__(self, dim: int, pq_codec=None, metric: Metric = Metric.COSINE, devices: Sequence[int] = (0,), block: int = 65536,
                 shard_factory: Optional[Callable] = None, merge_packed: Optional[Callable] = None, **kwargs):
        for kk in ('ef_construction', 'ef_search', 'max_connection'):
            kwargs.pop(kk, None)
        super().__init__(dim, metric=metric, **{k: v for k, v in kwargs.items() if k in ('dtype', 'initial_size', 'expand_step_size', 'expand_mode')})
        assert pq_codec is not None, 'MultiGpuPQIndex needs a PQCodec'
        assert len(devices) >= 1 and block >= 64 and block % 64 == 0
        self.pq_codec = pq_codec
        self.devices = [int(d) for d in devices]
        self.block = int(block)
        G = len(self.devices)
        per_shard = dict(kwargs)
        if per_shard.get('initial_size'):
            per_shard['initial_size'] = max(64, -(-int(per_shard['initial_size']) // G))
        if shard_factory is None:
            from .pq_flat_gpu import PQFlatGpuIndex

            def shard_factory(g, dev):
                with _OnDevice(dev):
                    return PQFlatGpuIndex(dim=dim, metric=metric, pq_codec=pq_codec, **per_shard)
        self._shards = [shard_factory(g, d) for g, d in enumerate(self.devices)]
        self._merge_packed = merge_packed
        self._fake = merge_packed is not None  # (CPU tests: injected shards + numpy merge)

    # ------------------------------------------------------------------ the dealing: global offset <-> (shard, local row)
    @property
    def n_shards(self) -> int:
        return len(self._shards)

    def shard_of(self, offsets: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """global row offsets -> (shard, local row): block-cyclic, ``block`` consecutive rows per turn"""
        o = np.asarray(offsets, dtype=np.int64)
        blk = o // self.block
        return (blk % self.n_shards).astype(np.int64), (blk // self.n_shards) * self.block + o % self.block

    def _global_ids(self, local: torch.Tensor, g: int) -> torch.Tensor:
        """local rows of shard g -> global offsets (-1 stays -1)"""
        blk = torch.div(local, self.block, rounding_mode='floor')
        glob = (blk * self.n_shards + g) * self.block + local % self.block
        return torch.where(local >= 0, glob, local)

    # ------------------------------------------------------------------ mutation
    def add_with_ids(self, x, ids, **kwargs):
        ids = np.asarray(ids.cpu() if isinstance(ids, torch.Tensor) else ids, dtype=np.int64)
        if len(ids) == 0:
            return
        if isinstance(x, torch.Tensor) and x.is_cuda:
            xs = x
        else:
            xs = np.ascontiguousarray(x.reshape(1, -1) if getattr(x, 'ndim', 2) == 1 else x, dtype=np.float32)
        sh, loc = self.shard_of(ids)
        for g, (shard, dev) in enumerate(zip(self._shards, self.devices)):
            sel = np.nonzero(sh == g)[0]
            if len(sel) == 0:
                continue
            with _OnDevice(dev):
                part = xs[torch.from_numpy(sel).to(xs.device)] if isinstance(xs, torch.Tensor) else xs[sel]
                if isinstance(part, torch.Tensor) and part.device.index != dev:
                    part = part.to(torch.device('cuda', dev))
                shard.add_with_ids(part, loc[sel])
        self._size = sum(s.size for s in self._shards)

    def update_with_ids(self, x, ids, **kwargs):
        self.add_with_ids(x, ids)

    def delete(self, ids):
        ids = np.asarray(list(ids), dtype=np.int64)
        if len(ids) == 0:
            return
        sh, loc = self.shard_of(ids)
        for g, (shard, dev) in enumerate(zip(self._shards, self.devices)):
            sel = np.nonzero(sh == g)[0]
            if len(sel):
                with _OnDevice(dev):
                    shard.delete(loc[sel].tolist())
        self._size = sum(s.size for s in self._shards)

    def reset(self, capacity: Optional[int] = None):
        super().reset(capacity=capacity)
        for shard, dev in zip(getattr(self, '_shards', []), getattr(self, 'devices', [])):
            with _OnDevice(dev):
                shard.reset()

    @property
    def size(self):
        return sum(s.size for s in self._shards)

    # ------------------------------------------------------------------ search
    def _split_filter(self, indices):
        if indices is None:
            return [None] * self.n_shards
        sh, loc = self.shard_of(np.asarray(indices, dtype=np.int64))
        return [loc[sh == g] for g in range(self.n_shards)]

    def search_batch(self, x, limit: int = 10, indices=None, **kwargs):
        """All queries, all shards: ``(dists [B, k], global ids [B, k])`` ascending by (distance, id); numpy in -> numpy out,
        device tensor in -> tensors on the first device."""
        is_np = not isinstance(x, torch.Tensor)
        k = int(limit)
        assert k >= 1
        per = self._split_filter(indices)
        packed, plain = [], []
        for g, (shard, dev) in enumerate(zip(self._shards, self.devices)):
            with _OnDevice(dev):
                xg = x
                if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != dev:
                    xg = x.to(torch.device('cuda', dev), non_blocking=True)
                if per[g] is not None and len(per[g]) == 0:  # nothing of the filter lives here
                    B = x.shape[0] if getattr(x, 'ndim', 1) == 2 else 1
                    p = torch.empty((B, k, 2), dtype=torch.int64, device=None if self._fake else torch.device('cuda', dev))
                    p[..., 0], p[..., 1] = -1, 0x7F800000
                    packed.append(p)
                    continue
                p = shard.search_batch_packed(xg, k, 0) if per[g] is None else None
                if p is not None:
                    p = p.clone() if self._fake else p
                    p[..., 0] = self._global_ids(p[..., 0], g)
                    packed.append(p)
                else:  # filtered / re-rank / k > 64: the general path of the shard, merged on the final distances
                    d, i = shard.search_batch(xg, limit=k, indices=per[g], **kwargs)
                    d, i = torch.as_tensor(d), torch.as_tensor(i)
                    plain.append((d, self._global_ids(i, g)))
        dev0 = None if self._fake else torch.device('cuda', self.devices[0])
        with _OnDevice(self.devices[0]):
            if plain or not packed:
                # (general path) every shard's (distance, id) lists -> one (distance, id)-ordered list
                lists = plain + [(self._unpack_d(p), p[..., 0]) for p in packed]
                d = torch.stack([a.to(dev0) if dev0 is not None else a for a, _ in lists])
                i = torch.stack([b.to(dev0) if dev0 is not None else b for _, b in lists])
                if self._fake:
                    from ...sharded import numpy_merge

                    od, oi = numpy_merge(d, i)
                elif k > 64:  # (beyond the merge kernel's wave lists: every shard answered through its batched large-k path)
                    od, oi = merge_lists_sorted(d, i, k)
                else:
                    from ... import ops

                    od, oi = ops.topk_merge(d.contiguous(), i.contiguous())
            else:
                gathered = torch.stack([p.to(dev0, non_blocking=True) if dev0 is not None else p for p in packed])
                sqrt = self.metric == Metric.EUCLIDEAN
                if self._merge_packed is not None:
                    od, oi = self._merge_packed(gathered, sqrt=sqrt)
                else:
                    from ... import ops

                    od, oi = ops.topk_merge_packed(gathered.contiguous(), sqrt=sqrt)
        if is_np:
            return od.cpu().numpy(), oi.cpu().numpy()
        return od, oi

    def _unpack_d(self, p: torch.Tensor) -> torch.Tensor:
        d = (p[..., 1] & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
        return torch.sqrt(d) if self.metric == Metric.EUCLIDEAN else d

    def search(self, x, limit: int = 10, indices=None):
        """ONE query, reference signature (hnsw/index.py:139-167)."""
        if indices is not None and len(indices) < limit:
            limit = len(indices)
        if limit <= 0:
            return np.empty((0,), np.float32), np.empty((0,), np.int64)
        d, i = self.search_batch(x if getattr(x, 'ndim', 1) == 2 else np.asarray(x).reshape(1, -1), limit=limit, indices=indices)
        if isinstance(d, torch.Tensor):
            d, i = d.cpu().numpy(), i.cpu().numpy()
        keep = i[0] >= 0
        return d[0][keep], i[0][keep]

    # ------------------------------------------------------------------ persistence: one file per shard
    def dump(self, index_file: Union[str, Path]):
        index_file = str(index_file)
        with open(index_file, 'wb') as f:
            np.save(f, np.array([{'format': 'annlite_amd.MultiGpuPQIndex/1', 'n_shards': self.n_shards, 'block': self.block}], dtype=object),
                    allow_pickle=True)
        for g, (shard, dev) in enumerate(zip(self._shards, self.devices)):
            with _OnDevice(dev):
                shard.dump(f'{index_file}.shard{g}')

    def load(self, index_file: Union[str, Path]):
        index_file = str(index_file)
        with open(index_file, 'rb') as f:
            state = np.load(f, allow_pickle=True)[0]
        assert state['format'] == 'annlite_amd.MultiGpuPQIndex/1'
        assert state['n_shards'] == self.n_shards and state['block'] == self.block, 'an index is reloaded onto the same number of shards'
        for g, (shard, dev) in enumerate(zip(self._shards, self.devices)):
            with _OnDevice(dev):
                shard.load(f'{index_file}.shard{g}')
        self._size = sum(s.size for s in self._shards)
