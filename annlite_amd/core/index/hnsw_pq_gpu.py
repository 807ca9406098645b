"""``HnswPQGpuIndex`` -- BASELINE config 5: HNSW-over-PQ candidate lists + GPU ADC / exact re-rank.

Mirrors the reference's ``HnswIndex(pq_codec=...)`` (annlite/core/index/hnsw/index.py:52-191) where it
differs from the exhaustive ``PQFlatGpuIndex``: a navigable graph proposes ``ef_search`` candidate rows per
query instead of scanning every row.

  * graph: the reference's knobs -- ``max_connection`` (16), ``ef_construction`` (200), ``ef_search`` (50) (hnsw/index.py:66-69) --
    and the same form of edge distance, hnswlib::PQLookup over the stored code bytes (include/hnswlib/space_pq.h:15-37), always
    with L2 tables (inner-product tables break the graph, see hnsw_host.cpp).  BUILT on the GPU in batches (round 6,
    ``graph_gpu_build.GpuLevel0Graph``: level 0 only, the packed pair walk as the insertion search) where the GPU walk applies,
    else -- or with ``build='host'`` -- by ``libannlite_graph.so`` (``annlite_amd/csrc/hnsw_host.cpp``: host code like the
    reference's hnswlib, the full hierarchy); WALKED on the GPU (``graph.hip``: packed node records, two nodes per step) or, for a
    host-built graph, by the library (``walk='host'``);
  * the candidates' distances and the final top-k come from the GPU: ``annlite_adc_gather`` (the a2 sum of
    SURVEY.md section 8a, bit-equal to the flat scan's distances) + ``annlite_topk_rows``, or, with
    ``rerank=True``, the exact distances on the stored float vectors fused with the top-k (``annlite_rerank_topk``).

Everything else (storage, encode, validity bitmap, dump/load of the code table, metric pre/post
processing) is inherited from ``PQFlatGpuIndex``; ``search_exhaustive`` keeps the full scan available.
"""
import ctypes
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from ... import _graph_capi as gc
from ... import ops
from ...enums import Metric
from .pq_flat_gpu import PQFlatGpuIndex


class HnswPQGpuIndex(PQFlatGpuIndex):
    def __init__(self, dim: int, pq_codec=None, metric: Metric = Metric.COSINE, ef_construction: int = 200,
                 ef_search: int = 50, max_connection: int = 16, n_threads: int = 0, seed: int = 100,
                 walk: str = 'gpu', packed_graph: bool = True, expand_width: int = 2, build: Optional[str] = None, **kwargs):
        super().__init__(dim, pq_codec=pq_codec, metric=metric, **kwargs)
        self.ef_construction = int(ef_construction)  # hnsw/index.py:66-69
        self.ef_search = int(ef_search)
        self.max_connection = int(max_connection)
        self.n_threads = int(n_threads)
        self.seed = int(seed)
        assert walk in ('gpu', 'host')
        self.walk = walk  # where the graph is walked at query time: GPU kernel (default) or the host library
        # GPU walk over packed node records (neighbours' code rows inline: 656 B per node at M = 16, max_connection 16) -- the
        # default; False = the plain lists + code table (the parity tests compare the two)
        self.packed_graph = bool(packed_graph)  # (False: the plain walk; `release_packed()` gives the records' memory back)
        # nodes expanded per step of the packed GPU walk (round 6): 2 = the pair walk (the two best unexpanded entries together: half
        # as many dependent steps per query; its own expansion order, > 0.99 of the one-at-a-time walk's candidates at ef = 128);
        # 1 = one at a time, bit-equal to the plain walk.  Lists of more than 32 links per node are walked one at a time.
        assert expand_width in (1, 2)
        self.expand_width = int(expand_width)
        # where the graph is BUILT: 'gpu' (round 6) = graph_gpu_build.GpuLevel0Graph -- level 0 only, inserted in batches by the GPU
        # (the searches of an insertion are the packed pair walk itself: 5M rows in 3.5 s against 60 s on 16 host cores, same
        # recall); GPU walks only; ids must be dense in insertion order (0, 1, 2, ...: what AnnLite's offsets are,
        # storage/table.py:251-257).  'host' = libannlite_graph.so: the full hierarchy (host walks, the reference's structure).
        # None: 'gpu' where it applies -- walk='gpu', max_connection <= 16, M in {8, 16, 32, 64}, uint8 codes -- else 'host'.
        assert build in (None, 'host', 'gpu')
        if build is None:
            build = 'gpu' if (walk == 'gpu' and 2 <= self.max_connection <= 16 and self.M in (8, 16, 32, 64) and self.Ks <= 256) else 'host'
        self.build = build
        self._gg = None         # GpuLevel0Graph
        self._packed = None
        self._packed_key = None
        self._graph = None
        self._gpu_graph = None  # (key, links i32 [N, L+1], seeds i32 [S]) exported for the GPU walk
        self._mutations = 0     # bumped by every add / delete / reset / load: the key of the exported graph (a delete changes the seeds)
        self._structure = 0     # bumped by add / reset / load only: the key of the caches a delete leaves valid -- the PLAIN code
                                # table and the packed node records (N x 656 B at 32 links, M = 16: 3.3 GB at 5M rows; a delete only
                                # clears a validity bit, the records and the link lists stay as they are)

    # ------------------------------------------------------------------ graph handle
    def _ensure_graph(self):
        if self._graph is None:
            if not self.pq_codec.is_trained:
                raise RuntimeError('Please train the PQ before using HNSW quantization backend')  # hnsw/index.py:32-35
            assert self.Ks <= 256, 'the graph stores uint8 codes'
            cb = np.ascontiguousarray(self.pq_codec.codebooks, dtype=np.float32)
            h = gc.lib().annlite_hnsw_create(cb.ctypes.data, self.M, self.Ks, self.dim // self.M,
                                             max(int(self.capacity), 1), self.max_connection, self.ef_construction,
                                             self.seed)
            if not h:
                raise RuntimeError('annlite_hnsw_create: ' + gc.lib().annlite_hnsw_last_error().decode())
            self._graph = ctypes.c_void_p(h)
        return self._graph

    def _ensure_gpu_graph(self):
        if self._gg is None:
            if not self.pq_codec.is_trained:
                raise RuntimeError('Please train the PQ before using HNSW quantization backend')  # hnsw/index.py:32-35
            from .graph_gpu_build import GpuLevel0Graph

            self._gg = GpuLevel0Graph(self.pq_codec.codebooks_dev, self.max_connection, self.ef_construction)
        return self._gg

    def _rebuild_gpu_graph(self):
        N = self._n_rows
        gg = self._ensure_gpu_graph()
        codes = self._plain_codes(N).contiguous()
        gg.reserve(int(self.capacity))
        for c0 in range(0, N, 1 << 18):
            c1 = min(N, c0 + (1 << 18))
            x = self._vectors[c0:c1] if self._vectors is not None else ops.pq_decode(codes[c0:c1], self.pq_codec.codebooks_dev)
            _, xg = self.pq_codec.scan_inputs(x.contiguous())
            gg.add(xg, codes[c0:c1])

    def __del__(self):
        try:
            if getattr(self, '_graph', None) is not None:
                gc.lib().annlite_hnsw_free(self._graph)
                self._graph = None
        except Exception:
            pass

    # ------------------------------------------------------------------ build
    def add_with_ids(self, x, ids: List[int], **kwargs):
        ids_np = np.ascontiguousarray(np.asarray(ids.cpu() if isinstance(ids, torch.Tensor) else ids, dtype=np.int64))
        xq = self._pre(x)  # device, f32, normalised for cosine (hnsw/index.py:28-29)
        if xq.shape[0] == 0:
            return
        super().add_with_ids(x, ids)  # code table / validity / float vectors (the parent pre-processes itself)
        self._mutations += 1
        self._structure += 1
        # the graph sees what the reference's add_items sees: the pre-processed vectors (+ their code bytes);
        # PQCodec.get_dist_mat would normalise them once more for cosine (pq.py:309-310) -- do the same
        _, xg = self.pq_codec.scan_inputs(xq)
        codes = ops.pq_encode(xq, self.pq_codec.codebooks_dev)
        if self.build == 'gpu':
            gg = self._ensure_gpu_graph()
            if not (ids_np[0] == gg.n and (np.diff(ids_np) == 1).all()):
                raise RuntimeError(f"build='gpu' takes ids in insertion order (next: {gg.n}, got {ids_np[:3]}...)")
            gg.reserve(int(self.capacity))
            gg.add(xg, codes)
            return
        x_np = np.ascontiguousarray(xg.cpu().numpy(), dtype=np.float32)
        c_np = np.ascontiguousarray(codes.cpu().numpy(), dtype=np.uint8)
        g = self._ensure_graph()
        gc.check(gc.lib().annlite_hnsw_add(g, x_np.ctypes.data, c_np.ctypes.data, ids_np.ctypes.data, len(ids_np),
                                           self.n_threads), 'annlite_hnsw_add')

    def update_with_ids(self, x, ids: List[int], **kwargs):
        raise RuntimeError('the HNSW graph does not support in-place updates (delete + add a new offset): '
                           'hnsw/index.py:173-177')

    def delete(self, ids: List[int]):
        super().delete(ids)
        self._mutations += 1
        if self._graph is not None:
            for i in ids:
                gc.lib().annlite_hnsw_mark_deleted(self._graph, int(i))  # hnsw/index.py:169-171 mark_deleted

    def reset(self, capacity: Optional[int] = None):
        super().reset(capacity=capacity)
        self._mutations += 1
        self._structure = getattr(self, '_structure', 0) + 1
        self._gpu_graph = None
        self._plain_cache = None
        self._plain_cache_key = None
        self._packed = None
        self._packed_key = None
        self._gg = None
        if self._graph is not None:
            gc.lib().annlite_hnsw_free(self._graph)
            self._graph = None

    # ------------------------------------------------------------------ search
    def _export_graph(self):
        """Level-0 lists + seed set of the host graph on the device (re-exported after inserts / deletes)."""
        if self.build == 'gpu':
            gg = self._ensure_gpu_graph()
            return gg.links[: gg.n], gg.seeds()
        key = self._mutations  # (sizes alone collide: clear() + the same number of documents again, restore())
        if self._gpu_graph is None or self._gpu_graph[0] != key:
            n = self._n_rows
            lpn = int(gc.lib().annlite_hnsw_links_per_node(self._graph))
            links = np.empty((n, lpn + 1), dtype=np.uint32)
            seeds = np.empty((1024,), dtype=np.int64)
            ns = ctypes.c_int64(0)
            gc.check(gc.lib().annlite_hnsw_export(self._graph, n, links.ctypes.data, seeds.ctypes.data, seeds.size,
                                                  ctypes.byref(ns)), 'annlite_hnsw_export')
            dev = self._codes.device
            self._gpu_graph = (key, torch.from_numpy(links.view(np.int32)).to(dev),
                               torch.from_numpy(seeds[:ns.value].astype(np.int32)).to(dev))
        return self._gpu_graph[1], self._gpu_graph[2]

    def _packed_records(self, links: torch.Tensor, plain: torch.Tensor) -> torch.Tensor:
        """The packed node records of the current graph (cached; rebuilt after INSERTS -- deletes leave links and code rows alone)."""
        if self.build == 'gpu':
            return self._gg.packed  # (kept current by every insertion batch)
        key = (self._n_rows, self._structure)
        if getattr(self, '_packed_key', None) != key:
            self._packed = ops.graph_pack(links, plain, n_rows=self._n_rows)
            self._packed_key = key
        return self._packed

    def release_packed(self):
        """Free the packed node records (N x 656 B at 32 links, M = 16: 3.3 GB at 5M rows); the next packed walk rebuilds them."""
        self._packed, self._packed_key = None, None

    def _gpu_walk_ok(self) -> bool:
        return self.walk == 'gpu' and self.M in (8, 16, 32, 64) and self.Ks <= 256

    def candidates(self, q_dev: torch.Tensor, ef: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Graph walk: ``(ids i64 [B, ef], L2 pq distance f32 [B, ef])`` on the device, -1 / +inf padded, ascending.
        ``walk='gpu'``: ``annlite_graph_search`` (one wave per query); ``walk='host'``: libannlite_graph.so."""
        ef = int(ef or self.ef_search)
        _, xg = self.pq_codec.scan_inputs(q_dev)
        if self._gpu_walk_ok() and ef <= 256:
            from ..._capi import LAYOUT_BMK, LUT_L2

            if self.build != 'gpu':
                self._ensure_graph()
            links, seeds = self._export_graph()
            lut = ops.lut_build(xg, self.pq_codec.codebooks_dev, LUT_L2, LAYOUT_BMK)
            plain = self._plain_table(self._n_rows)
            if self.packed_graph and links.shape[1] - 1 <= 64:
                # packed node records (round 5): the neighbours' code rows sit behind every node's link list -- one contiguous
                # read per expansion, the next record prefetched; with expand_width = 1 the candidate lists are the plain walk's, bit
                # for bit
                lpn = links.shape[1] - 1
                return ops.graph_search_packed(self._packed_records(links, plain), lpn, seeds, plain, lut, ef,
                                               valid_bits=self._valid, n_rows=self._n_rows,
                                               expand_width=self.expand_width if lpn <= 32 else 1)
            return ops.graph_search(links, seeds, plain, lut, ef, valid_bits=self._valid, n_rows=self._n_rows)
        if self.build == 'gpu':
            raise RuntimeError("build='gpu' keeps level 0 only: walk='gpu' with M in {8, 16, 32, 64}, Ks <= 256, ef <= 256")
        x_np = np.ascontiguousarray(xg.cpu().numpy(), dtype=np.float32)
        B = x_np.shape[0]
        ids = np.empty((B, ef), dtype=np.int64)
        dist = np.empty((B, ef), dtype=np.float32)
        gc.check(gc.lib().annlite_hnsw_search(self._ensure_graph(), x_np.ctypes.data, B, ef, ids.ctypes.data,
                                              dist.ctypes.data, self.n_threads), 'annlite_hnsw_search')
        return ops.to_dev(ids, torch.int64), ops.to_dev(dist, torch.float32)

    def search_exhaustive(self, x, limit: int = 10, **kw):
        return super().search_batch(x, limit=limit, **kw)

    def search_batch(self, x, limit: int = 10, indices=None, ef_search: Optional[int] = None, **kwargs):
        """``(dists [B,k], ids [B,k])``; candidates from the graph (``max(ef_search, k)`` per query), distances
        and top-k from the GPU.  ``indices`` (a filter) falls back to the exhaustive masked scan, which is what
        the reference's pre-filter does to small candidate sets as well (container.py:107-120)."""
        if indices is not None:
            return super().search_batch(x, limit=limit, indices=indices)
        is_np = not isinstance(x, torch.Tensor)
        q = self._pre(x)
        B, k = q.shape[0], int(limit)
        N = self._n_rows
        if N == 0 or B == 0 or (self._graph is None and (self._gg is None or self._gg.n == 0)):
            d = torch.full((B, k), float('inf'), dtype=torch.float32, device=q.device)
            i = torch.full((B, k), -1, dtype=torch.int64, device=q.device)
        else:
            ef = max(int(ef_search or self.ef_search), k)
            cand, cand_d = self.candidates(q, ef)
            if not (self.rerank and self._vectors is not None) and self.metric == Metric.EUCLIDEAN and self._gpu_walk_ok() and ef <= 256 \
                    and k <= 64:
                # (round 6) EUCLIDEAN, ADC ranking: the walk's own distances ARE the metric's PQLookup sums (the graph is walked with
                # L2 tables: same table, same ascending-m fp32 chain as annlite_adc_gather, bit for bit) and its list is ascending
                # with the deleted rows blanked (-1, +inf) -- the result is the list's first k real entries: no table build, no
                # gather.  (cosine / inner product: the metric's tables differ from the walk's -- the gather below.)
                kk = min(k, ef)
                d, pos = ops.topk_rows(cand_d, kk)
                i = torch.gather(cand, 1, pos.clamp(min=0))
                i = torch.where((pos < 0) | torch.isinf(d), torch.full_like(i, -1), i)
                d = torch.sqrt(d)  # hnsw/index.py:164-165
                if kk < k:
                    d = torch.cat([d, torch.full((B, k - kk), float('inf'), device=d.device)], dim=1)
                    i = torch.cat([i, torch.full((B, k - kk), -1, dtype=torch.int64, device=i.device)], dim=1)
                return (d.cpu().numpy(), i.cpu().numpy()) if is_np else (d, i)
            if self.rerank and self._vectors is not None and k <= 64:
                # (round 6) exact distances of the candidates + validity screen + top-k + sqrt in ONE launch, one wave per query:
                # the same numbers as the steps below, bit for bit (annlite_rerank_topk)
                d, i = ops.rerank_topk(int(self.metric), q, self._vectors, cand, k, valid_bits=self._valid,
                                       sqrt=self.metric == Metric.EUCLIDEAN)
                return (d.cpu().numpy(), i.cpu().numpy()) if is_np else (d, i)
            # rows deleted after the walk started / never written are masked here as well
            ok = (cand >= 0) & self._valid_bool[cand.clamp(min=0)]
            cand = torch.where(ok, cand, torch.full_like(cand, -1))
            if self.rerank and self._vectors is not None:
                dist = ops.exact_gather_dist(int(self.metric), q, self._vectors, cand)
            else:
                from ..._capi import LAYOUT_BMK

                kind, xq = self._scan_inputs(x, q)  # (host buffers: normalised like the flat index's queries, bit for bit)
                lut = ops.lut_build(xq, self.pq_codec.codebooks_dev, kind, LAYOUT_BMK)  # [B, M, Ks] on the device
                dist = ops.adc_gather(lut, self._plain_table(N), cand)
            kk = min(k, ef)
            d, pos = self._topk_rows_any(dist, kk)
            i = torch.gather(cand, 1, pos.clamp(min=0))
            i = torch.where((pos < 0) | torch.isinf(d), torch.full_like(i, -1), i)
            if self.metric == Metric.EUCLIDEAN:
                d = torch.sqrt(d)  # hnsw/index.py:164-165
            if kk < k:
                d = torch.cat([d, torch.full((B, k - kk), float('inf'), device=d.device)], dim=1)
                i = torch.cat([i, torch.full((B, k - kk), -1, dtype=torch.int64, device=i.device)], dim=1)
        if is_np:
            return d.cpu().numpy(), i.cpu().numpy()
        return d, i

    def _plain_table(self, N: int) -> torch.Tensor:
        """Code rows in sub-space order for the gather kernel (cached; rebuilt after inserts)."""
        if getattr(self, 'build', 'host') == 'gpu' and self._gg is not None and self._gg.n >= N and N > 0:
            return self._gg.codes[:N]  # (the GPU-built graph keeps the PLAIN rows it links: the same bytes, no third copy)
        key = (N, self._structure)
        if getattr(self, '_plain_cache_key', None) != key:
            self._plain_cache = self._plain_codes(N).contiguous()
            self._plain_cache_key = key
        return self._plain_cache

    # ------------------------------------------------------------------ persistence
    def dump(self, index_file: Union[str, Path]):
        super().dump(index_file)
        if getattr(self, 'build', 'host') == 'gpu':
            Path(str(index_file) + '.graph').unlink(missing_ok=True)  # (one graph file per dump: the other build's would be stale)
            if self._gg is not None:
                with open(str(index_file) + '.level0.npy', 'wb') as f:
                    np.save(f, np.array([self._gg.state()], dtype=object), allow_pickle=True)
            return
        Path(str(index_file) + '.level0.npy').unlink(missing_ok=True)
        if self._graph is not None:
            gc.check(gc.lib().annlite_hnsw_save(self._graph, (str(index_file) + '.graph').encode()), 'annlite_hnsw_save')

    def load(self, index_file: Union[str, Path]):
        super().load(index_file)
        self._mutations = getattr(self, '_mutations', 0) + 1
        self._structure = getattr(self, '_structure', 0) + 1
        lpath, gpath = str(index_file) + '.level0.npy', str(index_file) + '.graph'
        if getattr(self, 'build', 'host') == 'gpu':
            self._gg = None
            if Path(lpath).exists():
                st = np.load(lpath, allow_pickle=True)[0]
                if int(st['n']) == self._n_rows:  # (a file left behind by an older dump of another table is not this table's graph)
                    self._ensure_gpu_graph().load_state(st, self._plain_codes(self._n_rows).contiguous())
                    return
            if not Path(gpath).exists():
                # a dump without a graph file (written by the flat index, or the graph file was lost): the level-0 graph is
                # rebuilt from what IS stored -- the float vectors where the index keeps them, else the decoded code rows
                # (the walk's tables are then built from reconstructions: the distances between stored rows are the same)
                if self._n_rows:
                    self._rebuild_gpu_graph()
                return
            self.build = 'host'  # the file of a host-built graph (a snapshot of an earlier build): keep serving it as it is
        if getattr(self, 'build', 'host') == 'host' and not Path(gpath).exists() and Path(lpath).exists():
            raise RuntimeError(f'{index_file}: the snapshot holds a GPU-built level-0 graph; open it with build="gpu" (or build=None)')
        if Path(gpath).exists():
            if self._graph is not None:
                gc.lib().annlite_hnsw_free(self._graph)
            h = gc.lib().annlite_hnsw_load(gpath.encode())
            if not h:
                raise RuntimeError('annlite_hnsw_load: ' + gc.lib().annlite_hnsw_last_error().decode())
            self._graph = ctypes.c_void_p(h)
