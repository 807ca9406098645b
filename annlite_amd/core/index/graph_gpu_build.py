"""``GpuLevel0Graph`` -- the level-0 graph of the HNSW-over-PQ index BUILT ON THE GPU in batches (round 6; kernels:
``annlite_amd/csrc/graph_build.hip``; the searches of an insertion are the same packed pair walk queries use, ``graph.hip``).

What it replaces: hnswlib ``addPoint`` (include/hnswlib/hnswalg.h:1108-1235) as ``HnswIndex.add_with_ids`` drives it
(annlite/core/index/hnsw/index.py:124-137), in the form ``libannlite_graph.so`` restates it (hnsw_host.cpp ``insert`` /
``select_neighbors`` / ``connect``): search with ``ef_construction``, keep at most ``max_connection`` candidates by the
diversification heuristic (Algorithm 4) under the symmetric code-to-code L2 distance, link back with a shrink to
``2 * max_connection``.  Same knobs, same rules -- for a BATCH of points at a time:

  * a graph of fewer than ``BRUTE`` (8192) nodes grows by exact candidate lists: the ``ef_construction`` nearest stored points of
    every new point (ADC distance: the point's vector against the DECODED rows), batch mates included -- one ``cdist``;
  * beyond that a batch is ``min(16384, nodes / 4)`` points: their L2 tables (``annlite_lut_build``) are the queries of ONE packed
    pair walk over the graph as it is, ``annlite_graph_build_select`` writes their lists, the (target, source) pairs are sorted on
    the device and ``annlite_graph_build_reverse`` updates every target once, ``annlite_graph_pack_nodes`` re-packs the records
    that changed.  The points of a batch do not see each other (they meet through the neighbours both link to and through later
    batches' reverse links); a batch is at most a quarter of the graph while it is small and 16384 points later (0.3 % at 5M).

Only level 0 exists: the GPU walk scans a seed sample flat instead of descending upper layers (graph.hip), so none are built --
``seeds()`` is up to 128 nodes spread evenly over the insertion order.
"""
from typing import Optional, Tuple

import torch

from ... import ops
from ..._capi import LAYOUT_BMK, LUT_L2


class GpuLevel0Graph:
    BRUTE = 8192       # below this many nodes: exact candidate lists
    BATCH = 16384      # points per walk batch
    GROW = 4           # ... and at most 1 / GROW of the graph (while it is small)
    MAX_SEEDS = 128    # seeds the walk scans flat (5M rows: 32 ... 1024 seeds give the same recall, 1024 cost 16 rounds of 64 per query)

    def __init__(self, codebooks_dev: torch.Tensor, max_connection: int = 16, ef_construction: int = 200):
        M, Ks, dsub = codebooks_dev.shape
        assert M in (8, 16, 32, 64) and Ks <= 256, 'the GPU graph build supports M in {8, 16, 32, 64} and uint8 codes'
        assert 2 <= max_connection <= 16, 'the GPU graph build holds 2 * max_connection <= 32 links per node'
        self.cb = codebooks_dev.contiguous()
        self.M, self.Ks, self.dsub = M, Ks, dsub
        self.Mc = int(max_connection)
        self.lpn = 2 * self.Mc
        self.efc = int(min(max(ef_construction, self.Mc), 256))
        self.sdc = ops.graph_build_sdc(self.cb)
        self.rec = ops.graph_record_bytes(self.lpn, M)
        self.n = 0
        self.links = self.codes = self.packed = None
        self._seeds: Optional[Tuple[int, torch.Tensor]] = None

    # ------------------------------------------------------------------ storage
    def reserve(self, cap: int):
        cur = 0 if self.links is None else self.links.shape[0]
        if cap <= cur:
            return
        cap = max(cap, cur * 2, 1024)
        dev = self.cb.device
        links = torch.zeros((cap, self.lpn + 1), dtype=torch.int32, device=dev)
        codes = torch.zeros((cap, self.M), dtype=torch.uint8, device=dev)
        packed = torch.zeros((cap, self.rec), dtype=torch.uint8, device=dev)
        if self.n:
            links[: self.n] = self.links[: self.n]
            codes[: self.n] = self.codes[: self.n]
            packed[: self.n] = self.packed[: self.n]
        self.links, self.codes, self.packed = links, codes, packed

    def seeds(self) -> torch.Tensor:
        """Up to ``MAX_SEEDS`` distinct nodes spread evenly over the insertion order (i32)."""
        if self._seeds is None or self._seeds[0] != self.n:
            n, s = self.n, min(self.n, self.MAX_SEEDS)
            idx = (torch.arange(s, device=self.cb.device, dtype=torch.int64) * n) // max(s, 1)
            self._seeds = (n, idx.to(torch.int32).contiguous())
        return self._seeds[1]

    # ------------------------------------------------------------------ build
    def add(self, x: torch.Tensor, codes: torch.Tensor):
        """Insert ``x`` f32 [n, D] (what the walk's tables are built from: pre-processed vectors) with their PLAIN code rows u8 [n, M];
        the points become nodes ``self.n .. self.n + n``."""
        n_new = x.shape[0]
        if n_new == 0:
            return
        assert codes.shape == (n_new, self.M) and codes.dtype == torch.uint8
        self.reserve(self.n + n_new)
        self.codes[self.n: self.n + n_new] = codes
        pos = 0
        while pos < n_new:
            n = self.n
            if n < self.BRUTE:
                b = min(n_new - pos, self.BRUTE - n)
                cand = self._exact_candidates(x[pos: pos + b], n, b)
            else:
                b = min(n_new - pos, self.BATCH, max(1024, n // self.GROW))
                lut = ops.lut_build(x[pos: pos + b].contiguous(), self.cb, LUT_L2, LAYOUT_BMK)
                cand, _ = ops.graph_search_packed(self.packed, self.lpn, self.seeds(), self.codes, lut, self.efc, n_rows=n,
                                                  expand_width=2)
            self._link(cand, n, b)
            pos += b

    def _exact_candidates(self, x: torch.Tensor, n: int, b: int) -> torch.Tensor:
        """The ``ef_construction`` nearest of the n + b stored points (batch mates included, the point itself not) for each of the b
        new points, ascending: ADC distance as |x - decode(row)|^2."""
        total = n + b
        xhat = ops.pq_decode(self.codes[:total], self.cb)
        d = torch.cdist(x.to(torch.float32), xhat).square_()
        d[torch.arange(b, device=d.device), torch.arange(n, total, device=d.device)] = float('inf')
        k = min(self.efc, total - 1)
        cand = torch.full((b, self.efc), -1, dtype=torch.int64, device=d.device)
        if k > 0:
            cand[:, :k] = torch.topk(d, k, dim=1, largest=False, sorted=True).indices
        return cand

    def _link(self, cand: torch.Tensor, n: int, b: int):
        pairs = ops.graph_build_select(cand.contiguous(), n, self.codes, self.sdc, self.Mc, self.links)
        keys = torch.sort(pairs.reshape(-1)).values
        n_keys = int((keys != torch.iinfo(torch.int64).max).sum().item())
        new_nodes = torch.arange(n, n + b, device=keys.device, dtype=torch.int64)
        if n_keys:
            keys = keys[:n_keys].contiguous()
            targets, counts = torch.unique_consecutive(keys >> 32, return_counts=True)
            seg = torch.zeros((targets.numel() + 1,), dtype=torch.int64, device=keys.device)
            seg[1:] = torch.cumsum(counts, 0)
            ops.graph_build_reverse(keys, seg, self.codes, self.sdc, self.links)
            changed = torch.cat([targets, new_nodes])
        else:
            changed = new_nodes
        self.n = n + b
        ops.graph_pack_nodes(self.links, self.codes, changed.contiguous(), self.packed, self.n)

    # ------------------------------------------------------------------ persistence
    def state(self):
        return {'n': self.n, 'links': self.links[: self.n].cpu().numpy() if self.n else None}

    def load_state(self, st, codes: torch.Tensor):
        """``codes``: the PLAIN code rows of nodes 0..n (the index's own table)."""
        import numpy as np  # noqa: F401

        n = int(st['n'])
        self.n = 0
        self.links = self.codes = self.packed = None
        self._seeds = None
        if n == 0:
            return
        self.reserve(n)
        self.links[:n] = ops.to_dev(st['links'])
        self.codes[:n] = codes[:n]
        self.n = n
        ops.graph_pack_nodes(self.links, self.codes, torch.arange(n, device=self.cb.device, dtype=torch.int64), self.packed, n)
