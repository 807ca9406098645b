"""``PQFlatGpuIndex`` -- exhaustive PQ/ADC index on one MI355X, plugging in where the reference
constructs ``HnswIndex(pq_codec=...)`` per cell (annlite/container.py:48-59).

Semantics = the reference's flat ADC index ``PQIndex`` (annlite/core/index/pq_index.py:11-56) made
metric-aware and batched, with ``HnswIndex``'s pre/post-processing so it is a drop-in for the seam
``CellContainer`` uses (SURVEY.md section 8b "Index plugin seam"):

  * ``search(x[D], limit, indices)`` -> ``(dists[k], ids[k])`` ascending        hnsw/index.py:139-167
      reshape to (1, D), cast f32, ``l2_normalize`` if cosine (index.py:28-29), tables through
      ``PQCodec.get_dist_mat`` (which normalises again, pq.py:309-310), ``sqrt`` of the distances
      for EUCLIDEAN (index.py:164-165); cosine / inner product distances are ``M/Ks - <q, x^>``.
  * ``add_with_ids(data[N,D], offsets)``   encode + store rows at ``offsets``      hnsw/index.py:124-137
  * ``delete(ids)``  marks rows (never returned again)                            hnsw/index.py:169-171
  * ``reset`` / ``dump`` / ``load`` / ``size`` / ``capacity``                      hnsw/index.py:116-122,185-191
  * untrained codec -> ``RuntimeError('Please train the PQ ...')``                hnsw/index.py:32-35
plus ``search_batch(X[B,D], limit)`` so that a whole ``AnnLite.search(docs)`` call is one launch
instead of the reference's per-query python loop (annlite/container.py:214).

Data layout in HBM: one ``uint8 [capacity, M]`` code table, stored pre-SKEWED (row n
rotated left by n mod M bytes) when the fast scan plan applies; a ``uint32`` validity bitmap (delete
marks / never-written rows); optionally the raw ``float32 [capacity, D]`` vectors for the exact
re-rank stage (``rerank=True``; 10M x 128 fp32 = 5.1 GB of the 288 GB).
"""
import math
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from ... import ops
from ..._capi import CODES_PLAIN, CODES_SKEWED, ScanState, scan_plan
from ...enums import ExpandMode, Metric
from ...math import l2_normalize_host
from ..codec.pq import PQCodec
from .base import BaseIndex


class SplitBatch:
    """A batch between the two halves of the split search (``PQFlatGpuIndex.split_prepare``; protocol: sharded.py)."""

    def __init__(self, keys, n_queries, device, union, scan, plain, keep=None):
        self.keys, self.n_queries, self.device = keys, n_queries, device
        self.union, self.scan, self.plain, self._keep = union, scan, plain, keep


class PQFlatGpuIndex(BaseIndex):
    def __init__(
        self,
        dim: int,
        pq_codec: Optional[PQCodec] = None,
        dtype: np.dtype = np.float32,
        metric: Metric = Metric.COSINE,
        rerank: bool = False,
        skewed: bool = True,
        index_file: Optional[Union[str, Path]] = None,
        rerank_pool: str = 'slices',
        **kwargs,
    ):
        # HNSW-only kwargs the reference forwards (ef_construction, ef_search, max_connection) are accepted and ignored
        for k in ('ef_construction', 'ef_search', 'max_connection'):
            kwargs.pop(k, None)
        super().__init__(dim, dtype=dtype, metric=metric, **kwargs)
        assert pq_codec is not None, 'PQFlatGpuIndex needs a PQCodec'
        self.pq_codec = pq_codec
        self.rerank = bool(rerank)
        # candidates of the exact re-rank stage: 'slices' = the union of the row slices' own ADC top-rerank_k lists (n_slices x
        # rerank_k rows); 'global' = the global ADC top-rerank_k (<= 64, default 50) of the shared-bound search
        assert rerank_pool in ('slices', 'global')
        self.rerank_pool = rerank_pool
        self._want_skew = bool(skewed)
        self._ws = ops.ScanWorkspace()
        # device storage is allocated on first use so that constructing an index (and the host-side
        # error paths, e.g. "not trained") does not need a GPU
        self._codes = None
        self._valid_bool = None
        self._valid_bits_cache = None
        self._vectors = None
        self._n_rows = 0
        self._scan_state = None  # the library's per-table kernel choice (created with the device storage)
        if index_file:
            self.load(index_file)

    # ------------------------------------------------------------------ storage
    @property
    def M(self) -> int:
        return self.pq_codec.n_subvectors

    @property
    def Ks(self) -> int:
        return self.pq_codec.n_clusters

    @property
    def code_bytes(self) -> int:
        return np.dtype(self.pq_codec.code_dtype).itemsize

    def _layout(self) -> int:
        fast = self.code_bytes == 1 and self.M in (8, 16, 32, 64) and self.Ks <= 256
        return CODES_SKEWED if (fast and self._want_skew) else CODES_PLAIN

    def _alloc(self, capacity: int):
        dev = ops.device()
        tdt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[self.code_bytes]
        self._codes = torch.zeros((capacity, self.M), dtype=tdt, device=dev)
        # validity: bool per row is the source of truth, the uint32 bitmap the kernels read is packed lazily
        self._valid_bool = torch.zeros((((capacity + 31) // 32 + 2) * 32,), dtype=torch.bool, device=dev)
        self._valid_bits_cache: Optional[torch.Tensor] = None
        self._vectors = torch.zeros((capacity, self.dim), dtype=torch.float32, device=dev) if self.rerank else None
        self._capacity = capacity
        self._n_rows = 0  # scan range = highest written row id + 1
        self._size = 0

    def _ensure_alloc(self):
        if self._codes is None:
            self._alloc(self._capacity)

    def _expand_capacity(self, new_capacity: int):
        self._ensure_alloc()
        old_codes, old_valid, old_vec, n_rows, size = self._codes, self._valid_bool, self._vectors, self._n_rows, self._size
        self._alloc(new_capacity)
        self._codes[: old_codes.shape[0]] = old_codes
        self._valid_bool[: old_codes.shape[0]] = old_valid[: old_codes.shape[0]]
        if old_vec is not None:
            self._vectors[: old_vec.shape[0]] = old_vec
        self._n_rows, self._size = n_rows, size

    # ------------------------------------------------------------------ pre-processing (hnsw/index.py:20-48)
    def _pre(self, x) -> torch.Tensor:
        if not self.pq_codec.is_trained:
            raise RuntimeError('Please train the PQ before using HNSW quantization backend')
        if isinstance(x, np.ndarray) and self.metric == Metric.COSINE:
            # host buffers are normalised with the reference's own numpy expression before the upload (bit-equal
            # vectors => bit-equal codes / tables / ids); device tensors by the kernel
            xh = np.ascontiguousarray(x.reshape(1, -1) if x.ndim == 1 else x, dtype=np.float32)
            assert xh.shape[-1] == self.dim, (
                f'the query embedding dimension does not match with index dimension: {xh.shape[-1]} vs {self.dim}')
            return ops.to_dev(l2_normalize_host(xh), torch.float32)
        x = ops.to_dev(x, torch.float32)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        assert x.shape[-1] == self.dim, (
            f'the query embedding dimension does not match with index dimension: {x.shape[-1]} vs {self.dim}')
        if self.metric == Metric.COSINE:
            x = ops.l2_normalize(x)
        return x

    def _scan_inputs(self, x, q: torch.Tensor):
        """(LUT kind, queries as the table build sees them): ``PQCodec.get_dist_mat`` normalises cosine queries a second
        time (pq.py:309-310); for host buffers that happens on the host too, in the reference's arithmetic."""
        # (what the table build normalises is the CODEC's business -- pq.py:67-69 normalize_input --, not the index metric's)
        if isinstance(x, np.ndarray) and self.pq_codec.normalize_input:
            kind, _ = self.pq_codec.scan_inputs(q[:0])
            return kind, ops.to_dev(l2_normalize_host(l2_normalize_host(
                np.ascontiguousarray(x.reshape(1, -1) if x.ndim == 1 else x, dtype=np.float32))), torch.float32)
        return self.pq_codec.scan_inputs(q)

    @staticmethod
    def _pack_bits(flags: torch.Tensor) -> torch.Tensor:
        """bool [32*W] -> int32 [W] bitmap words (bit i of word w = flags[32*w + i]); plumbing only."""
        shifts = torch.arange(32, device=flags.device, dtype=torch.int64)
        packed = (flags.reshape(-1, 32).to(torch.int64) << shifts[None, :]).sum(dim=1)
        return torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)

    @property
    def _valid(self) -> torch.Tensor:
        if self._valid_bits_cache is None:
            self._valid_bits_cache = self._pack_bits(self._valid_bool)
        return self._valid_bits_cache

    def _set_bits(self, ids: torch.Tensor, value: bool):
        self._valid_bool[ids] = value
        self._valid_bits_cache = None

    def _get_bits(self, ids: torch.Tensor) -> torch.Tensor:
        return self._valid_bool[ids]

    # ------------------------------------------------------------------ mutation
    def add_with_ids(self, x, ids: List[int], **kwargs):
        x = self._pre(x)
        self._ensure_alloc()
        ids_t = ops.to_dev(np.asarray(ids, dtype=np.int64) if not isinstance(ids, torch.Tensor) else ids, torch.int64)
        assert ids_t.numel() == x.shape[0]
        if ids_t.numel() == 0:
            return
        max_id = int(ids_t.max().item()) + 1
        if max_id > self.capacity:
            steps = math.ceil(max_id / self.expand_step_size)  # hnsw/index.py:132-135
            self._expand_capacity(steps * self.expand_step_size)
        codes = ops.pq_encode(x, self.pq_codec.codebooks_dev)
        if self._layout() == CODES_SKEWED:
            ops.codes_skew(codes, ids_t, out=self._codes)
        else:
            self._codes[ids_t] = codes
        if self._vectors is not None:
            self._vectors[ids_t] = x
        was_valid = self._get_bits(ids_t)
        self._set_bits(ids_t, True)
        self._size += int((~was_valid).sum().item())
        self._n_rows = max(self._n_rows, max_id)

    def update_with_ids(self, x, ids: List[int], **kwargs):
        """flat_index.py:70-71 semantics (overwrite rows); HnswIndex refuses updates, PQIndex allows."""
        self.add_with_ids(x, ids)

    def delete(self, ids: List[int]):
        if self._codes is None or len(ids) == 0:
            return
        ids_t = ops.to_dev(np.asarray(list(ids), dtype=np.int64), torch.int64)
        was_valid = self._get_bits(ids_t)
        self._set_bits(ids_t, False)
        self._size -= int(was_valid.sum().item())

    def reset(self, capacity: Optional[int] = None):
        super().reset(capacity=capacity)
        self._codes = None
        self._valid_bool = None
        self._valid_bits_cache = None
        self._vectors = None
        self._n_rows = 0
        if self._scan_state is not None:
            self._scan_state.reset()

    @property
    def size(self):
        return self._size

    # ------------------------------------------------------------------ search
    def _filter_bits(self, indices) -> torch.Tensor:
        """`indices` argument of search (pq_index.py:42-44, container.py:107-120): restrict to a subset."""
        idx = ops.to_dev(np.asarray(indices, dtype=np.int64) if not isinstance(indices, torch.Tensor) else indices, torch.int64)
        sel = torch.zeros_like(self._valid_bool)
        sel[idx] = True
        return self._pack_bits(sel & self._valid_bool)

    # ------------------------------------------------------------------ kernel choice
    # Which M = 16 scan kernel serves this table -- byte filter tables (data with structure: what PQ is for) or u16 filter
    # tables (independent uniform codes: the byte filter leaks) -- is decided INSIDE the library, from what its own launches
    # measure (annlite_hip.h: annlite_scan_state): the first call(s) run the byte-table kernel guarded, later calls read
    # the verdict from the state's host-mapped block.  No calibration runs, no extra latency on a user's query, and
    # ANNLITE_SCAN_VARIANT in the environment still overrides everything for A/B measurements.
    @property
    def scan_state(self) -> ScanState:
        if self._scan_state is None:
            self._scan_state = ScanState()
        return self._scan_state

    @property
    def scan_kernel(self) -> str:
        """'undecided' | 'byte tables' | 'u16 tables' -- what the library has settled on for this table"""
        return ScanState.KERNELS[self.scan_state.info()[0]] if self._scan_state is not None else 'undecided'

    def search_batch(self, x, limit: int = 10, indices=None, rerank_k: Optional[int] = None, row_base: int = 0
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
        """All queries of ``x`` [B, D] in one launch.  Returns device tensors
        ``(dists f32 [B, k], ids i64 [B, k])``, ascending by (distance, id); missing -> (+inf, -1).
        With ``rerank=True`` indexes, the ADC scan produces ``n_slices * rerank_k`` candidates per query
        that are re-scored exactly on the stored float vectors."""
        is_np = not isinstance(x, torch.Tensor)
        q = self._pre(x)
        B = q.shape[0]
        k = int(limit)
        assert k >= 1
        N = self._n_rows
        valid = None
        if N > 0:
            valid = self._valid if indices is None else self._filter_bits(indices)
        dev = q.device
        if N == 0 or B == 0:
            d = torch.full((B, k), float('inf'), dtype=torch.float32, device=dev)
            i = torch.full((B, k), -1, dtype=torch.int64, device=dev)
        elif self.rerank and self._vectors is not None:
            d, i = self._search_rerank(q, k, valid, N, rerank_k, self._scan_inputs(x, q), filtered=indices is not None)
        elif k <= 64:
            # table build + scan + top-k: one C call (annlite_pq_search_topk)
            kind, xq = self._scan_inputs(x, q)
            base = row_base
            d, i = ops.pq_search_topk(
                kind, xq, self.pq_codec.codebooks_dev, self._codes, k, self.M, self.Ks, valid_bits=valid, n_rows=N,
                codes_layout=self._layout(), workspace=self._ws, row_base=base,
                sqrt=self.metric == Metric.EUCLIDEAN,  # hnsw/index.py:164-165
                state=self.scan_state if indices is None else None)  # (a filtered call says nothing about the table)
            row_base = 0
        else:
            d, i = self._search_large_k(q, k, valid, N, self._scan_inputs(x, q))
        if row_base:
            i = torch.where(i >= 0, i + row_base, i)  # (paths that do not take row_base natively)
        if is_np:
            return d.cpu().numpy(), i.cpu().numpy()
        return d, i

    def search_batch_packed(self, x, limit: int, row_base: int = 0) -> Optional[torch.Tensor]:
        """Plain ADC top-k of a device batch as ONE i64 tensor [B, k, 2] = (row_base + id or -1, bits of the RAW
        ADC sum -- no sqrt): the per-rank contribution to the single all-gather of the row-sharded search.
        ``None`` when this index state needs the general path (re-rank, k > 64)."""
        k = int(limit)
        if (self.rerank and self._vectors is not None) or k > 64:
            return None
        q = self._pre(x)
        B, N = q.shape[0], self._n_rows
        if N == 0 or B == 0:
            out = torch.empty((B, k, 2), dtype=torch.int64, device=q.device)
            out[..., 0] = -1
            out[..., 1] = 0x7F800000  # +inf
            return out
        kind, xq = self._scan_inputs(x, q)
        return ops.pq_search_topk(
            kind, xq, self.pq_codec.codebooks_dev, self._codes, k, self.M, self.Ks, valid_bits=self._valid,
            row_base=row_base, n_rows=N, codes_layout=self._layout(), workspace=self._ws, packed=True, state=self.scan_state)

    def split_supported(self, x, limit: int) -> bool:
        """STATIC part of the split search's conditions -- the same on every rank of a row-sharded search (configuration, batch
        shape, input kind), so that all ranks agree on whether the seed collective takes place at all."""
        k = int(limit)
        dsub = self.dim // self.M if self.M else 0
        return (isinstance(x, torch.Tensor) and x.ndim == 2 and x.shape[0] > 0 and not (self.rerank and self._vectors is not None) and
                1 <= k <= 16 and self.M == 16 and self.code_bytes == 1 and self.Ks <= 256 and self.dim <= 256 and dsub % 4 == 0 and
                self.metric == Metric.EUCLIDEAN and not self.pq_codec.normalize_input)

    def split_prepare(self, x, limit: int, row_base: int, seed_rows: int, workspace) -> Optional['SplitBatch']:
        """First half of ``search_batch_packed`` for a rank of a row-sharded search WITH a seed exchange (sharded.py runs the
        protocol): the preparation launch, seeding from this rank's first ``seed_rows`` rows only, into ``workspace`` (None: the index's own,
        per-stream scratch).  Returns the batch's handle -- ``keys`` i64 [B, 16] (this
        rank's contribution to the seed all-gather) or ``keys is None`` when the split does not apply to THIS rank NOW (fewer
        than 4096 rows, the library has not settled on the byte-table kernel yet: nothing was launched, ``plain()`` is the
        search) -- or ``None`` when ``split_supported`` says no (the same answer on every rank: no collective at all)."""
        from ..._capi import PHASE_PREPARE, PHASE_SCAN

        if not self.split_supported(x, limit):
            return None
        k = int(limit)
        q = self._pre(x)
        B = q.shape[0]
        kind, xq = self._scan_inputs(x, q)
        args = dict(valid_bits=self._valid, row_base=row_base, n_rows=self._n_rows, codes_layout=self._layout())
        workspace = workspace if workspace is not None else self._ws
        call = lambda phase, **kw: ops.pq_search_split(phase, kind, xq, self.pq_codec.codebooks_dev, self._codes, k, self.M, self.Ks,
                                                       self.scan_state, workspace, **args, **kw)
        keys = call(PHASE_PREPARE, seed_rows=seed_rows) if self._n_rows >= 4096 else None

        def union(all_keys):
            ops.pq_search_seed_union(all_keys.contiguous(), self._codes, B, k, self.M, self.Ks, workspace, n_rows=self._n_rows)

        return SplitBatch(keys=keys, n_queries=B, device=q.device, union=union, scan=lambda: call(PHASE_SCAN),
                          plain=lambda: self.search_batch_packed(x, limit, row_base), keep=(q, xq))

    @property
    def sqrt_epilogue(self) -> bool:
        """Metric epilogue of ``search`` on raw ADC sums (hnsw/index.py:164-165)."""
        return self.metric == Metric.EUCLIDEAN

    def _plain_codes(self, N: int) -> torch.Tensor:
        if self._layout() == CODES_SKEWED:
            return ops.codes_skew(self._codes[:N], inverse=True)
        return self._codes[:N]

    def _search_large_k(self, q, k, valid, N, scan_in=None):
        """k > 64 (beyond the scan kernels' lists): the distances of a CHUNK of queries to every row (adc_dist launches into one
        [chunk, N] buffer) -> i64 keys (order-preserving bits of the f32 sum << 32 | row: unique, so the k smallest keys ARE
        the (distance, id)-ordered top-k) -> one batched ``torch.topk`` per chunk.  No per-query sort, no host round trip."""
        from ..._capi import LAYOUT_BMK, LAYOUT_TILED  # noqa: F401

        kind, xq = scan_in if scan_in is not None else self.pq_codec.scan_inputs(q)
        lut = ops.lut_build(xq, self.pq_codec.codebooks_dev, kind, LAYOUT_BMK)
        codes = self._plain_codes(N)
        dev = q.device
        shifts = torch.arange(32, device=dev, dtype=torch.int64)
        vb = (((valid.to(torch.int64) & 0xFFFFFFFF)[:, None] >> shifts[None, :]) & 1).bool().reshape(-1)[:N]  # unpack
        kk = min(k, N)
        B = q.shape[0]
        chunk = max(1, min(B, (1 << 26) // max(N, 1)))  # <= 64M keys (512 MB) + their f32 sums per chunk
        rows = torch.arange(N, device=dev, dtype=torch.int64)
        inf = torch.tensor(float('inf'), device=dev)
        nan = torch.tensor(float('nan'), device=dev)
        key_none = torch.tensor(torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
        ds, is_ = [], []
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            dist = torch.empty((nb, N), dtype=torch.float32, device=dev)
            for j in range(nb):
                ops.adc_dist(lut[b0 + j], codes, out=dist[j])
            dist = dist + 0.0  # (-0.0 -> +0.0: equal VALUES tie-break by id)
            dist = torch.where(torch.isnan(dist), nan, dist)  # one NaN, sign bit clear: it sorts behind +inf (numpy's order)
            bits = dist.view(torch.int32)
            bits = bits ^ ((bits >> 31) & 0x7FFFFFFF)  # signed-comparable image of the float order
            keys = (bits.to(torch.int64) << 32) | rows[None, :]
            keys = torch.where(vb[None, :], keys, key_none)  # deleted / never-written rows: behind every real row, NaN ones included
            top = torch.topk(keys, kk, dim=1, largest=False, sorted=True).values
            none = top == key_none
            si = torch.where(none, torch.zeros_like(top), top & 0xFFFFFFFF)
            sd = torch.where(none, inf, torch.gather(dist, 1, si))
            si = torch.where(none, torch.full_like(si, -1), si)
            ds.append(sd)
            is_.append(si)
        d, i = torch.cat(ds), torch.cat(is_)
        if kk < k:
            d = torch.cat([d, torch.full((d.shape[0], k - kk), float('inf'), device=d.device)], dim=1)
            i = torch.cat([i, torch.full((i.shape[0], k - kk), -1, dtype=torch.int64, device=i.device)], dim=1)
        if self.metric == Metric.EUCLIDEAN:
            d = torch.sqrt(d)
        return d, i

    @staticmethod
    def _topk_rows_any(values: torch.Tensor, k: int):
        """Row-wise k smallest with positions, (value, position) ascending: the wave kernel for k <= 64, a stable device
        sort beyond (a large `limit` must not be cut to 64 silently)."""
        if k <= 64:
            return ops.topk_rows(values, k)
        sd, si = torch.sort(values, dim=1, stable=True)
        sd, si = sd[:, :k], si[:, :k]
        return sd, torch.where(torch.isinf(sd), torch.full_like(si, -1), si)

    def _search_rerank(self, q, k, valid, N, rerank_k, scan_in=None, filtered=False):
        B = q.shape[0]
        # candidates per row slice: 16 where the byte-table kernel generates them (M = 16: 8 slices x 16 keys = 128 per query
        # at 1024 queries; its lists hold 16 keys), 64 otherwise (u16-table kernels)
        byte_tables = self.M == 16 and self.code_bytes == 1 and self.Ks <= 256 and self.scan_kernel != 'u16 tables'
        asked = rerank_k or getattr(self, 'rerank_k', None)
        rk = max(1, min(64, int(asked or (16 if byte_tables else 64))))
        plan = scan_plan(N, self.M, self.Ks, self.code_bytes, B, rk)
        if not asked:
            # The pool is n_slices * rk rows per query, and a small table or a huge batch plans 1-4 slices: the default must
            # still hand the re-rank MORE than k candidates (k of k re-ranks nothing, fewer than k returns padding) -- at
            # least max(64, 4 k) where the table has them.  Beyond the byte-table lists' 16 keys the u16-table generator
            # (up to 64 keys per slice) takes over: scan_plan picks it from rk.
            want = max(64, 4 * k)
            for _ in range(2):  # (the tile width, hence n_slices, can change with the generator)
                if plan.n_slices * rk >= want or rk >= 64:
                    break
                rk = min(64, -(-want // max(plan.n_slices, 1)))
                plan = scan_plan(N, self.M, self.Ks, self.code_bytes, B, rk)
        from ..._capi import LAYOUT_BMK, LAYOUT_TILED

        kind, xq = scan_in if scan_in is not None else self.pq_codec.scan_inputs(q)
        # (the global pool holds at most 64 rows: a larger `limit` takes the slice pool, n_slices * rk rows -- never 64 real rows
        # plus padding; a filtered call -- `valid` is the caller's mask, not the table's own bitmap -- keeps its candidate counts out
        # of the table's kernel-choice state, as search_batch does)
        if getattr(self, 'rerank_pool', 'slices') == 'global' and k <= 64:
            # Pool = the GLOBAL ADC top-R (R <= 64) of the shared-bound search -- since round 5 the byte-table kernel serves it
            # (64-key lists) -- instead of the union of per-slice top-16 lists: 50 rows that are the 50 best by ADC distance
            # against 128 of which only the 16 best are guaranteed.  One C call, then the exact re-score as below.
            R = max(min(k, 64), min(64, int(asked or 50)))
            _, cand = ops.pq_search_topk(kind, xq, self.pq_codec.codebooks_dev, self._codes, R, self.M, self.Ks, valid_bits=valid,
                                         n_rows=N, codes_layout=self._layout(), workspace=self._ws,
                                         state=None if filtered else self.scan_state)
        else:
            # (round 6) tables + candidate generator in one C call; M = 16 / L2: through the plain search's ONE preparation launch
            _, cand = ops.pq_search_candidates(kind, xq, self.pq_codec.codebooks_dev, self._codes, rk, self.M, self.Ks, valid_bits=valid,
                                               n_rows=N, codes_layout=self._layout(), workspace=self._ws)
        if k <= 64:  # (round 6) exact distances + top-k + ids + sqrt in one launch, a wave per query; same numbers as below
            return ops.rerank_topk(int(self.metric), q, self._vectors, cand, k, sqrt=self.metric == Metric.EUCLIDEAN)
        exact = ops.exact_gather_dist(int(self.metric), q, self._vectors, cand)  # [B, R]
        kk = min(k, cand.shape[1])
        d, pos = self._topk_rows_any(exact, kk)
        i = torch.gather(cand, 1, pos.clamp(min=0))
        i = torch.where(pos < 0, torch.full_like(i, -1), i)
        i = torch.where(torch.isinf(d), torch.full_like(i, -1), i)
        if self.metric == Metric.EUCLIDEAN:
            d = torch.sqrt(d)
        if kk < k:
            d = torch.cat([d, torch.full((B, k - kk), float('inf'), device=d.device)], dim=1)
            i = torch.cat([i, torch.full((B, k - kk), -1, dtype=torch.int64, device=i.device)], dim=1)
        return d, i

    def search(self, x, limit: int = 10, indices=None):
        """ONE query, reference signature (hnsw/index.py:139-167): ``(dists[k'], ids[k'])`` numpy,
        ``k' <= limit`` valid entries only."""
        if indices is not None and len(indices) < limit:
            limit = len(indices)  # hnsw/index.py:153-154
        if limit <= 0:
            return np.empty((0,), np.float32), np.empty((0,), np.int64)
        d, i = self.search_batch(x, limit=limit, indices=indices)
        if isinstance(d, torch.Tensor):
            d, i = d.cpu().numpy(), i.cpu().numpy()
        d, i = d[0], i[0]
        keep = i >= 0
        return d[keep], i[keep]

    # ------------------------------------------------------------------ persistence (own format)
    def dump(self, index_file: Union[str, Path]):
        """hnsw/index.py:121-122 analogue.  Codes are saved in the PLAIN (reference) layout."""
        self._ensure_alloc()
        N = self._n_rows
        state = {
            'format': 'annlite_amd.PQFlatGpuIndex/1',
            'dim': self.dim, 'M': self.M, 'Ks': self.Ks, 'metric': int(self.metric),
            'n_rows': N, 'size': self._size, 'capacity': self._capacity,
            'codes': ops.codes_to_numpy(self._plain_codes(N)) if N else np.zeros((0, self.M), self.pq_codec.code_dtype),
            'valid': self._valid_bool[:N].cpu().numpy(),
            'vectors': self._vectors[:N].cpu().numpy() if self._vectors is not None else None,
        }
        with open(str(index_file), 'wb') as f:
            np.save(f, np.array([state], dtype=object), allow_pickle=True)

    def load(self, index_file: Union[str, Path]):
        with open(str(index_file), 'rb') as f:
            state = np.load(f, allow_pickle=True)[0]
        assert state['format'] == 'annlite_amd.PQFlatGpuIndex/1'
        assert state['dim'] == self.dim and state['M'] == self.M and state['Ks'] == self.Ks
        self._alloc(max(int(state['capacity']), self._capacity))
        N = int(state['n_rows'])
        if N:
            codes = ops.to_dev(state['codes'])
            if self._layout() == CODES_SKEWED:
                ops.codes_skew(codes, ids=None, id_base=0, out=self._codes)
            else:
                self._codes[:N] = codes
            if self._vectors is not None and state['vectors'] is not None:
                self._vectors[:N] = ops.to_dev(state['vectors'])
        v = ops.to_dev(state['valid'])
        self._valid_bool[: v.numel()] = v
        self._valid_bits_cache = None
        self._n_rows, self._size = N, int(state['size'])
        if self._scan_state is not None:
            self._scan_state.reset()  # (another table: what was measured no longer applies)
