"""``IvfPQGpuIndex`` -- PQ/ADC index over coarse cells on one MI355X: the structure of the reference's
``AnnLite(n_cells > 1)`` (``VQCodec`` coarse quantiser annlite/core/codec/vq.py, ``_cell_selection``
annlite/index.py:458-466, one vector index per cell + ``CellContainer.ivf_search`` merge, container.py:88-144)
as ONE code table and one scan launch for all queries and cells.

Semantics:
  * ``n_probe >= n_cells`` (what the reference always does: ``n_probe = max(n_probe, n_cells)``, index.py:94):
    every cell is visited -- the result is the exhaustive scan's (``PQFlatGpuIndex.search_batch``), which is
    what runs.
  * ``n_probe < n_cells`` (this build's extension, opt-in): a query scans only the rows of its ``n_probe``
    nearest cells; the result is the exact top-k of those rows under the fixed order (distance asc, id asc).

Layout in HBM on top of the flat index's storage (codes by offset, validity, optional float vectors):
  * ``_cell_of``  i32 [capacity]        cell of every offset (nearest centroid in squared L2, vq.py:81-90)
  * sealed view, rebuilt lazily after a mutation (one sort + one gather over the live rows):
      ``_table``     u8 [Nt, M] SKEWED by table row: the live rows grouped by cell, every cell starts at a
                     multiple of 64 rows, ascending offset inside a cell
      ``_row_ids``   i64 [Nt] offset of every table row (-1: padding)
      ``_cell_rows`` i64 [C, 2] (begin, end) of every cell, ``_cell_order`` i32 [C] cells by descending size
      ``_table_plain`` the same rows in the PLAIN layout (read by the exact re-score)
Search, ``n_subvectors = 16`` and ``limit <= 16`` (round 6) = ``annlite_ivf_select_cells`` -> ``annlite_ivf_search_topk`` (plan of
cell tiles of 32 (query, cell) pairs; ONE preparation launch: the queries' tables, a first bound from each query's nearest cell, a
byte table per query; the byte-table kernel over the tiles with exact sums and bounds shared between a query's tiles; merge of the
per-cell lists).  Other shapes and the float re-rank = ``annlite_ivf_select_cells`` -> ``annlite_ivf_plan`` (query tiles of one cell
each) -> ``annlite_pq_search_tiles`` (quantised tables of the queries + integer scan: per-slot candidate lists) ->
``annlite_lut_build`` for the real queries -> ``annlite_ivf_rescore`` (exact sums of the candidates, top-k).  Same results.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from ... import ops
from ..._capi import CODES_SKEWED, LUT_IPDIST, LUT_L2, scan_plan, scan_plan_tiles
from ...enums import Metric
from ..codec.pq import PQCodec
from ..codec.vq import VQCodec
from .pq_flat_gpu import PQFlatGpuIndex


class IvfPQGpuIndex(PQFlatGpuIndex):
    R_CAP = 4096  # rows a query re-ranks at most (float re-rank of a pruned search); grows with k * n_probe: see _search_pruned
    rerank_truncated = 0  # queries (so far) whose candidate set exceeded the cap and lost the later cells' lists
    def __init__(self, dim: int, pq_codec: Optional[PQCodec] = None, vq_codec: Optional[VQCodec] = None,
                 n_probe: Optional[int] = None, rerank_bound_rank: int = 1, rerank_split: Tuple[int, int] = (2, 4), **kwargs):
        super().__init__(dim, pq_codec=pq_codec, **kwargs)
        assert vq_codec is not None, 'IvfPQGpuIndex needs a VQCodec'
        self.vq_codec = vq_codec
        self.n_probe = n_probe  # None: every cell (the reference's behaviour)
        self._cell_of = None
        self._sealed = False
        self._table = self._table_plain = self._row_ids = self._cell_rows = self._cell_order = self._pos_of = None
        self.cand_cap = 256  # emitted candidates per (query, cell) list; an overflowing list falls back to the whole cell
        self.byte_tiles = True  # M = 16, k <= 16: annlite_ivf_search_topk / _candidates (False: the u16 tile scan + re-score)
        # float re-rank on the cell tiles: the candidate lists' first bound = the (rank x k)-th smallest seed sum of the nearest cell (1: the
        # smallest pool that still holds the ADC top-k; larger: longer lists from the far cells, better recall; include/annlite_hip.h).
        # 0: no private lists at all -- the pool is exactly the ADC top-`rerank_k` (annlite_ivf_search_topk's ids), the fastest
        self.rerank_bound_rank = int(rerank_bound_rank)
        # ... and the nearest cells in parts: a list holds 16 keys, so a whole cell hands the re-rank 16 rows at most -- the cap on the
        # recall of that path, since the nearest cells hold most true neighbours.  (n, S): each of the query's n nearest cells is probed as
        # S contiguous row ranges ("sub-cells": entries C .. C (1 + S) of the split cell table, _split_tables), each with a list of its own
        # -- up to 16 S rows from such a cell.  (0, 1): whole cells only.
        self.rerank_split = (int(rerank_split[0]), int(rerank_split[1]))
        self.rerank_seed_whole_cell = True
        self._split_cache = None
        self._tws = ops.ScanWorkspace()
        self.last_pruned_path = None  # which kernels served the last pruned search (measurement scripts)

    @property
    def n_cells(self) -> int:
        return self.vq_codec.n_clusters

    # ------------------------------------------------------------------ storage
    def _alloc(self, capacity: int):
        old = self._cell_of
        super()._alloc(capacity)
        self._cell_of = torch.zeros((capacity,), dtype=torch.int32, device=self._codes.device)
        if old is not None:
            n = min(old.numel(), capacity)
            self._cell_of[:n] = old[:n]
        self._sealed = False

    def add_with_ids(self, x, ids, **kwargs):
        ids_t = ops.to_dev(np.asarray(ids, dtype=np.int64) if not isinstance(ids, torch.Tensor) else ids, torch.int64)
        if ids_t.numel() == 0:
            return
        super().add_with_ids(x, ids_t)  # codes / validity / float vectors exactly as the flat index stores them
        # the reference assigns cells on the vectors as given (index.py:291-292, vq.py:81-90)
        raw = ops.to_dev(x, torch.float32)
        raw = raw.reshape(1, -1) if raw.ndim == 1 else raw
        self._cell_of[ids_t] = self.vq_codec.encode(raw).to(torch.int32)
        self._sealed = False

    def delete(self, ids):
        super().delete(ids)
        self._sealed = False

    def reset(self, capacity: Optional[int] = None):
        super().reset(capacity=capacity)
        self._cell_of = None
        self._sealed = False

    # ------------------------------------------------------------------ sealed (cell-sorted) view
    def _seal(self):
        if self._sealed:
            return
        dev = self._codes.device
        N, C = self._n_rows, self.n_cells
        live = torch.nonzero(self._valid_bool[:N]).flatten()  # ascending offsets
        cell = self._cell_of[:N][live].to(torch.int64)
        order = torch.sort(cell, stable=True).indices          # by cell, offsets ascending inside a cell
        offs = live[order]
        cell_sorted = cell[order]
        counts = torch.bincount(cell_sorted, minlength=C)
        padded = (counts + 63) // 64 * 64
        begin = torch.cumsum(padded, 0) - padded
        rank = torch.arange(offs.numel(), device=dev) - (torch.cumsum(counts, 0) - counts)[cell_sorted]
        pos = begin[cell_sorted] + rank
        Nt = max(64, int(padded.sum().item()))
        self._table = torch.zeros((Nt, self.M), dtype=torch.uint8, device=dev)
        self._table_plain = torch.zeros((Nt, self.M), dtype=torch.uint8, device=dev)
        if offs.numel():
            plain = self._plain_codes(N)[offs].contiguous()
            ops.codes_skew(plain, pos.contiguous(), out=self._table)
            self._table_plain[pos] = plain
        self._row_ids = torch.full((Nt,), -1, dtype=torch.int64, device=dev)
        self._row_ids[pos] = offs
        self._pos_of = torch.full((max(N, 1),), -1, dtype=torch.int64, device=dev)
        self._pos_of[offs] = pos
        self._cell_rows = torch.stack([begin, begin + counts], dim=1).contiguous()
        self._cell_order = torch.sort(counts, descending=True, stable=True).indices.to(torch.int32).contiguous()
        self._n_table = Nt
        self._split_cache = None
        self._sealed = True

    def _split_tables(self, S: int):
        """(cell_rows [C (1 + S)][2], cell_order) with every cell ALSO listed as S contiguous parts: entry C + c S + s = part s of cell c (begins at
        multiples of 64 rows like the cells themselves; a cell of fewer than S blocks leaves empty parts).  A query probes a cell either whole
        or through its parts, never both."""
        if self._split_cache is None or self._split_cache[0] != S:
            begin, ln = self._cell_rows[:, 0], self._cell_rows[:, 1] - self._cell_rows[:, 0]
            chunk = ((ln + S - 1) // S + 63) // 64 * 64
            s = torch.arange(S, device=begin.device)[None, :]
            lo = torch.minimum(s * chunk[:, None], ln[:, None])
            hi = torch.minimum((s + 1) * chunk[:, None], ln[:, None])
            some = hi > lo  # (an empty part is [begin, begin): a multiple of 64 like every other range's first row)
            lo, hi = torch.where(some, lo, torch.zeros_like(lo)), torch.where(some, hi, torch.zeros_like(hi))
            parts = torch.stack([begin[:, None] + lo, begin[:, None] + hi], dim=2).reshape(-1, 2)
            rows = torch.cat([self._cell_rows, parts]).contiguous()
            order = torch.sort(rows[:, 1] - rows[:, 0], descending=True, stable=True).indices.to(torch.int32).contiguous()
            self._split_cache = (S, rows, order)
        return self._split_cache[1], self._split_cache[2]

    def _select_kind_and_centroids(self) -> Tuple[int, torch.Tensor]:
        """cdist(query, vq codebook, metric) of ``_cell_selection`` (index.py:462-464) as a ranking."""
        cb = self.vq_codec.codebook_dev
        if self.metric == Metric.EUCLIDEAN:
            return 0, cb
        if self.metric == Metric.COSINE:
            return 1, ops.l2_normalize(cb)  # queries are normalised by _pre: 1 - cos ranks like -<q, c/|c|>
        return 1, cb

    # ------------------------------------------------------------------ search
    def search_batch(self, x, limit: int = 10, indices=None, rerank_k: Optional[int] = None, row_base: int = 0,
                     n_probe: Optional[int] = None):
        P = self.n_probe if n_probe is None else n_probe
        C = self.n_cells
        if P is None or P >= C:
            return super().search_batch(x, limit=limit, indices=indices, rerank_k=rerank_k, row_base=row_base)
        is_np = not isinstance(x, torch.Tensor)
        q = self._pre(x)
        B, k = q.shape[0], int(limit)
        assert 1 <= k <= 64, 'pruned search supports limit <= 64'
        assert self.M in (8, 16, 32, 64) and self.Ks <= 256 and self.code_bytes == 1, \
            'pruned search needs the quantised-filter scan plan (M in {8,16,32,64}, Ks <= 256)'
        dev = q.device
        if self._n_rows == 0 or B == 0:
            d = torch.full((B, k), float('inf'), dtype=torch.float32, device=dev)
            i = torch.full((B, k), -1, dtype=torch.int64, device=dev)
        else:
            self._seal()
            d, i = self._search_pruned(q, k, max(1, int(P)), indices, rerank_k)
            if row_base:
                i = torch.where(i >= 0, i + row_base, i)
        if is_np:
            return d.cpu().numpy(), i.cpu().numpy()
        return d, i

    def search_batch_packed(self, x, limit: int, row_base: int = 0):
        """(row-sharded search, sharded.py) the packed fast path is the exhaustive scan's: with pruning active the
        caller takes the general path through ``search_batch``."""
        if self.n_probe is not None and self.n_probe < self.n_cells:
            return None
        return super().search_batch_packed(x, limit, row_base=row_base)

    def probe_cells(self, q: torch.Tensor, n_probe: int) -> torch.Tensor:
        kind, cent = self._select_kind_and_centroids()
        return ops.ivf_select_cells(kind, q, cent, n_probe)

    def _table_bits(self, indices) -> Optional[torch.Tensor]:
        """``indices`` filter (offsets) as a bitmap over TABLE rows; None = every stored row."""
        if indices is None:
            return None  # the cell ranges hold live rows only; padding lies outside every range
        idx = ops.to_dev(np.asarray(indices, dtype=np.int64) if not isinstance(indices, torch.Tensor) else indices, torch.int64)
        idx = idx[(idx >= 0) & (idx < self._pos_of.numel())]
        pos = self._pos_of[idx]
        pos = pos[pos >= 0]
        sel = torch.zeros((((self._n_table + 31) // 32 + 2) * 32,), dtype=torch.bool, device=pos.device)
        sel[pos] = True
        return self._pack_bits(sel)

    def _search_pruned(self, q, k, P, indices, rerank_k):
        rerank = self.rerank and self._vectors is not None
        k_out = k
        if rerank:  # the scan keeps the rows that can be in a cell's ADC top-`rerank_k`; all of them are re-ranked
            k = max(k, min(32, int(rerank_k or 16)))
        cells = self.probe_cells(q, P)
        kind_l, xq_l = self.pq_codec.scan_inputs(q)
        if (self.byte_tiles and rerank and kind_l in (LUT_L2, LUT_IPDIST) and self.M == 16 and k <= 16 and k_out <= 16
                and self.dim <= 256 and (self.dim // self.M) % 4 == 0 and self._n_table < 2 ** 31):
            # (round 6) float re-rank on the byte-table cell tiles: every probed cell's own list (its best <= k rows by ADC sum at or
            # below the query's first bound -- private lists: a function of the cell, whatever else runs) -> P x k candidate ids ->
            # exact distances + top-k in one launch.  The union of a query's lists holds its exact ADC top-k of the probed cells.
            if self.rerank_bound_rank == 0:
                # exactly the ADC top-`rerank_k` of the probed cells as the pool: the plain pruned search (bounds SHARED by a query's
                # tiles, the fastest scan) with k = rerank_k, its ids re-ranked
                self.last_pruned_path = 'annlite_ivf_search_topk (byte-table cell tiles) + annlite_rerank_topk'
                _, ids = ops.ivf_search_topk(kind_l, xq_l, self.pq_codec.codebooks_dev, self._table, cells, self.n_cells, self._cell_rows,
                                             self._cell_order, k, self.M, self.Ks, row_ids=self._row_ids,
                                             valid_bits=self._table_bits(indices), n_rows=self._n_table, codes_layout=CODES_SKEWED,
                                             workspace=self._tws)
                return ops.rerank_topk(int(self.metric), q, self._vectors, ids, k_out, sqrt=self.metric == Metric.EUCLIDEAN)
            self.last_pruned_path = 'annlite_ivf_search_candidates (byte-table cell tiles) + annlite_rerank_topk'
            n_split, S = self.rerank_split
            n_split = min(int(n_split), P)
            cell_rows, cell_order, n_entries = self._cell_rows, self._cell_order, self.n_cells
            seed_cells = None
            if n_split > 0 and S > 1 and self.n_cells * (1 + int(S)) <= 16384:  # (the plan's limit on cell-table entries)
                if self.rerank_seed_whole_cell:  # the first bound from the WHOLE nearest cell's rows, not from its first part's
                    seed_cells = cells[:, 0].contiguous()
                cell_rows, cell_order = self._split_tables(int(S))
                n_entries = cell_rows.shape[0]
                parts = self.n_cells + cells[:, :n_split, None].to(torch.int64) * S + torch.arange(S, device=cells.device)
                cells = torch.cat([parts.reshape(cells.shape[0], -1).to(torch.int32), cells[:, n_split:]], dim=1).contiguous()
                self.last_pruned_path += ' (nearest %d cells in %d parts)' % (n_split, S)
            ids = ops.ivf_search_candidates(kind_l, xq_l, self.pq_codec.codebooks_dev, self._table, cells, n_entries, cell_rows,
                                            cell_order, k, self.M, self.Ks, row_ids=self._row_ids, valid_bits=self._table_bits(indices),
                                            n_rows=self._n_table, codes_layout=CODES_SKEWED, workspace=self._tws,
                                            bound_rank=self.rerank_bound_rank, seed_cells=seed_cells)
            return ops.rerank_topk(int(self.metric), q, self._vectors, ids, k_out, sqrt=self.metric == Metric.EUCLIDEAN)
        if (self.byte_tiles and not rerank and kind_l in (LUT_L2, LUT_IPDIST) and self.M == 16 and k <= 16 and self.dim <= 256
                and (self.dim // self.M) % 4 == 0 and self._n_table < 2 ** 31):
            # (round 6) the byte-table kernel in cell tiles: exact sums inside the tile, bounds shared by query, one merge
            # launch -- the same results as the u16 tile scan + re-score below
            self.last_pruned_path = 'annlite_ivf_search_topk (byte-table cell tiles)'
            return ops.ivf_search_topk(kind_l, xq_l, self.pq_codec.codebooks_dev, self._table, cells, self.n_cells, self._cell_rows,
                                       self._cell_order, k, self.M, self.Ks, row_ids=self._row_ids, valid_bits=self._table_bits(indices),
                                       n_rows=self._n_table, codes_layout=CODES_SKEWED, sqrt=self.metric == Metric.EUCLIDEAN,
                                       workspace=self._tws)
        self.last_pruned_path = 'annlite_pq_search_tiles + annlite_ivf_rescore (u16 tables)'
        qt = scan_plan_tiles(self._n_table, self.M, self.Ks, 1, 16, k).qt
        vmap, slot_of, tile_rows, _ = ops.ivf_plan(cells, self.n_cells, qt, self._cell_rows, self._cell_order)
        kind, xq = self.pq_codec.scan_inputs(q)
        bits = self._table_bits(indices)
        cand, count = ops.pq_search_tiles(kind, xq, self.pq_codec.codebooks_dev, self._table, k, self.M, self.Ks,
                                          tile_rows, vmap, valid_bits=bits, n_rows=self._n_table,
                                          codes_layout=CODES_SKEWED, workspace=self._tws,
                                          cand_cap=max(self.cand_cap, 32 * k))  # ~k ln(rows / k) rows pass per list
        if not rerank:
            lut = self.pq_codec.get_dist_mat(q)  # [B, M, Ks], the reference's tables for the real queries
            return ops.ivf_rescore(lut, self._table_plain, cand, count, slot_of, tile_rows, qt, k, self._row_ids, bits,
                                   sqrt=self.metric == Metric.EUCLIDEAN)
        # float re-rank: every row the integer scan let through (a superset of each probed cell's ADC top-k, 3-4 k
        # rows per list) is scored exactly on the stored vectors
        cnt = count.to(torch.int64)[slot_of.to(torch.int64)]  # [B, P]; an overflowed list counts 0xffffffff (-1 as int32)
        over_rows = (cnt < 0).any(dim=1)                      # queries with a list that holds only part of its cell
        per_query = torch.where(cnt < 0, torch.zeros_like(cnt), cnt).sum(dim=1)
        # ONE host round trip per batch: the widest candidate row and whether any list overflowed.  R is bounded: a query
        # re-ranks at most R_CAP rows (default 4096 = 8 x the ~540 a 16-cell probe emits at k = 10; beyond it the lists of
        # the later cells are cut, never the exact ADC top-k of the affected rows below)
        # (the cap follows the work asked for -- 8 x the ~3.4 k rows a list emits, per probed cell -- so that a wide probe with a
        # large k is not cut silently; what IS cut is counted in `rerank_truncated` and logged once)
        cap = max(self.R_CAP, 32 * k * P)
        stats = torch.stack([per_query.max(), over_rows.any().to(torch.int64), (per_query > cap).sum()]).cpu()
        R = max(min(int(stats[0]), cap), k)
        if int(stats[2]):
            if not self.rerank_truncated:
                import logging

                logging.getLogger('annlite_amd').warning(
                    'pruned re-rank: %d queries emitted more than %d candidate rows (k=%d, n_probe=%d); the later cells\' lists were cut',
                    int(stats[2]), cap, k, P)
            self.rerank_truncated += int(stats[2])
        ids = ops.ivf_candidate_ids(cand, count, slot_of, R, self._row_ids)  # (an overflowed list contributes nothing)
        if bool(stats[1]):
            # a list that overflowed (many ties / a loose bound): THOSE QUERIES take their candidates from the exact path
            # as well -- ivf_rescore walks an overflowed cell completely: the ADC top-k of the probed cells -- merged into
            # their row; the other queries of the batch keep their wide candidate set (recall does not depend on who shares
            # the batch)
            lut = self.pq_codec.get_dist_mat(q)
            _, ids_x = ops.ivf_rescore(lut, self._table_plain, cand, count, slot_of, tile_rows, qt, k, self._row_ids, bits,
                                       sqrt=False)
            kx = ids_x.shape[1]
            if R < kx + 1:
                ids = torch.cat([ids, torch.full((ids.shape[0], kx + 1 - R), -1, dtype=ids.dtype, device=ids.device)], dim=1)
            # the exact ids go in front; what the lists emitted follows (duplicates are harmless for a top-k by position:
            # equal distances, the first position wins; they are dropped below)
            # (only the overflowed queries' rows are merged: a B x R x k comparison over the whole batch was 268 MB at
            # 1024 x 4096 x 64)
            sel = torch.nonzero(over_rows).flatten()
            tail = ids[sel]
            dup = (tail[:, :, None] == ids_x[sel][:, None, :]).any(dim=2)
            tail = torch.where(dup, torch.full_like(tail, -1), tail)
            ids = torch.cat([ids, torch.full_like(ids_x, -1)], dim=1)
            ids[sel] = torch.cat([ids_x[sel], tail], dim=1)
        if k_out <= 64:  # (round 6) exact distances + top-k + ids + sqrt in one launch (annlite_rerank_topk): the same numbers as below
            return ops.rerank_topk(int(self.metric), q, self._vectors, ids.contiguous(), k_out, sqrt=self.metric == Metric.EUCLIDEAN)
        exact = ops.exact_gather_dist(int(self.metric), q, self._vectors, ids)
        d, pos = self._topk_rows_any(exact, min(k_out, ids.shape[1]))  # (k_out > 64: a stable device sort, never cut silently)
        i = torch.gather(ids, 1, pos.clamp(min=0))
        i = torch.where((pos < 0) | torch.isinf(d), torch.full_like(i, -1), i)
        if self.metric == Metric.EUCLIDEAN:
            d = torch.sqrt(d)
        if d.shape[1] < k_out:
            d = torch.cat([d, torch.full((d.shape[0], k_out - d.shape[1]), float('inf'), device=d.device)], dim=1)
            i = torch.cat([i, torch.full((i.shape[0], k_out - i.shape[1]), -1, dtype=torch.int64, device=i.device)], dim=1)
        return d, i

    # ------------------------------------------------------------------ persistence
    def dump(self, index_file):
        super().dump(index_file)
        np.save(str(index_file) + '.cells.npy', self._cell_of[: self._n_rows].cpu().numpy())

    def load(self, index_file):
        super().load(index_file)
        cells = np.load(str(index_file) + '.cells.npy')
        self._cell_of[: cells.shape[0]] = ops.to_dev(cells)
        self._sealed = False
