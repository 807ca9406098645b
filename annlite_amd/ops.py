"""Functional wrappers over the C ABI on torch device tensors.

Every function takes/returns ``torch`` tensors living on the current HIP device and launches the
hand-written gfx950 kernels of ``libannlite_hip.so`` on torch's current stream.  No function here
computes anything on the CPU; without a GPU they raise ``RuntimeError`` (``_capi.require_gpu``).
"""
from typing import Optional, Tuple

import numpy as np
import ctypes

import torch

from . import _capi
from ._capi import (CODES_PLAIN, CODES_SKEWED, LAYOUT_BMK, LAYOUT_TILED, LUT_IP, LUT_IPDIST, LUT_L2,
                    check, lib, scan_plan, stream_ptr)

_CODE_TORCH = {1: torch.uint8, 2: torch.int16, 4: torch.int32}  # torch has no uint16/32 arithmetic; bits are what matter


def device() -> torch.device:
    _capi.require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def to_dev(a, dtype=None) -> torch.Tensor:
    """numpy / torch (any device) -> contiguous tensor on the current HIP device."""
    dev = device()
    if isinstance(a, np.ndarray):
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        elif a.dtype == np.uint64:
            a = a.view(np.int64)
        t = torch.from_numpy(np.ascontiguousarray(a))
    else:
        t = a
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(dev, non_blocking=True).contiguous()


def code_bytes_of(t: torch.Tensor) -> int:
    return t.element_size()


def codes_to_numpy(t: torch.Tensor) -> np.ndarray:
    a = t.cpu().numpy()
    return {1: a, 2: a.view(np.uint16), 4: a.view(np.uint32)}[a.dtype.itemsize] if a.dtype.kind == 'i' else a


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------- LUT
def lut_build(queries: torch.Tensor, codebooks: torch.Tensor, kind: int, layout: int = LAYOUT_BMK,
              qi: int = 4) -> torch.Tensor:
    """f32 [B,D] x f32 [M,Ks,dsub] -> LUT.  BMK: [B,M,Ks]; TILED: flat padded buffer (see header)."""
    assert queries.dtype == torch.float32 and codebooks.dtype == torch.float32
    assert queries.ndim == 2 and codebooks.ndim == 3
    B, D = queries.shape
    M, Ks, dsub = codebooks.shape
    assert D == M * dsub, 'input dimension must be Ds * M'
    if layout == LAYOUT_BMK:
        out = torch.empty((B, M, Ks), dtype=torch.float32, device=queries.device)
    else:
        out = torch.empty((((B + 15) // 16) * 16 * M * Ks,), dtype=torch.float32, device=queries.device)
    check(lib().annlite_lut_build(kind, queries.data_ptr(), B, D, codebooks.data_ptr(), M, Ks, out.data_ptr(),
                                  layout, qi, stream_ptr()), 'lut_build')
    return out


def lut_retile(lut_bmk: torch.Tensor, qi: int) -> torch.Tensor:
    B, M, Ks = lut_bmk.shape
    out = torch.empty((((B + 15) // 16) * 16 * M * Ks,), dtype=torch.float32, device=lut_bmk.device)
    check(lib().annlite_lut_retile(lut_bmk.data_ptr(), B, M, Ks, out.data_ptr(), qi, stream_ptr()), 'lut_retile')
    return out


# ---------------------------------------------------------------------------------------------- ADC
def adc_dist(adtable: torch.Tensor, codes: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pq_bind.dist_pqcodes_to_codebooks on device: f32 [M,Ks], codes [N,M] -> f32 [N] (into ``out`` when given)."""
    assert adtable.ndim == 2 and codes.ndim == 2 and adtable.dtype == torch.float32
    M, Ks = adtable.shape
    N = codes.shape[0]
    assert codes.shape[1] == M
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=codes.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == N and out.device == codes.device
    check(lib().annlite_adc_dist(adtable.data_ptr(), M, Ks, codes.data_ptr(), code_bytes_of(codes), N,
                                 out.data_ptr(), stream_ptr()), 'adc_dist')
    return out


def adc_gather(lut_bmk: torch.Tensor, codes: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    B, M, Ks = lut_bmk.shape
    assert cand.dtype == torch.int64 and cand.shape[0] == B
    R = cand.shape[1]
    out = torch.empty((B, R), dtype=torch.float32, device=codes.device)
    check(lib().annlite_adc_gather(lut_bmk.data_ptr(), B, M, Ks, codes.data_ptr(), code_bytes_of(codes),
                                   codes.shape[0], cand.data_ptr(), R, out.data_ptr(), stream_ptr()), 'adc_gather')
    return out


def graph_search(links: torch.Tensor, seeds: torch.Tensor, codes: torch.Tensor, lut_bmk: torch.Tensor, ef: int,
                 valid_bits: Optional[torch.Tensor] = None, n_rows: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """GPU beam search over exported HNSW level-0 lists (``annlite_graph_search``): ``links`` i32 [N, L+1]
    (count, ids), ``seeds`` i32 [S], PLAIN ``codes`` u8 [N, M], L2 tables f32 [B, M, Ks].  Returns
    (ids i64 [B, ef] ascending by PQ distance, -1 padded; dist f32 [B, ef])."""
    B, M, Ks = lut_bmk.shape
    N = codes.shape[0] if n_rows is None else n_rows
    out_i = torch.empty((B, ef), dtype=torch.int64, device=codes.device)
    out_d = torch.empty((B, ef), dtype=torch.float32, device=codes.device)
    check(lib().annlite_graph_search(links.data_ptr(), links.shape[1] - 1, seeds.data_ptr(), seeds.numel(), codes.data_ptr(),
                                     N, M, Ks, _ptr(valid_bits), lut_bmk.data_ptr(), B, int(ef), out_i.data_ptr(),
                                     out_d.data_ptr(), stream_ptr()), 'graph_search')
    return out_i, out_d


def graph_pack(links: torch.Tensor, codes: torch.Tensor, n_rows: Optional[int] = None) -> torch.Tensor:
    """Packed node records (``annlite_graph_pack``): u8 [N, record_bytes] = every node's neighbours' code rows + its link
    list, from the exported lists ``links`` i32 [N, L+1] and the PLAIN ``codes`` u8 [N, M]."""
    import ctypes

    N = codes.shape[0] if n_rows is None else n_rows
    L, M = links.shape[1] - 1, codes.shape[1]
    nb = ctypes.c_int64(0)
    check(lib().annlite_graph_record_bytes(L, M, ctypes.byref(nb)), 'graph_record_bytes')
    out = torch.empty((max(N, 1), int(nb.value)), dtype=torch.uint8, device=codes.device)
    check(lib().annlite_graph_pack(links.data_ptr(), L, codes.data_ptr(), N, M, out.data_ptr(), stream_ptr()), 'graph_pack')
    return out


def graph_search_packed(packed: torch.Tensor, links_per_node: int, seeds: torch.Tensor, codes: torch.Tensor, lut_bmk: torch.Tensor,
                        ef: int, valid_bits: Optional[torch.Tensor] = None, n_rows: Optional[int] = None, expand_width: int = 1
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``graph_search`` over packed node records (``annlite_graph_search_packed``): one contiguous read per expansion, the
    next record prefetched; candidate lists bit-equal to ``graph_search``'s.  ``expand_width=2``
    (``annlite_graph_search_packed_ex``): the pair walk -- the two best unexpanded entries per step, half as many dependent steps;
    its own (pinned) expansion order."""
    B, M, Ks = lut_bmk.shape
    N = codes.shape[0] if n_rows is None else n_rows
    out_i = torch.empty((B, ef), dtype=torch.int64, device=codes.device)
    out_d = torch.empty((B, ef), dtype=torch.float32, device=codes.device)
    if int(expand_width) != 1:
        check(lib().annlite_graph_search_packed_ex(packed.data_ptr(), int(links_per_node), seeds.data_ptr(), seeds.numel(),
                                                   codes.data_ptr(), N, M, Ks, _ptr(valid_bits), lut_bmk.data_ptr(), B, int(ef),
                                                   int(expand_width), out_i.data_ptr(), out_d.data_ptr(), stream_ptr()),
              'graph_search_packed_ex')
        return out_i, out_d
    check(lib().annlite_graph_search_packed(packed.data_ptr(), int(links_per_node), seeds.data_ptr(), seeds.numel(), codes.data_ptr(),
                                            N, M, Ks, _ptr(valid_bits), lut_bmk.data_ptr(), B, int(ef), out_i.data_ptr(),
                                            out_d.data_ptr(), stream_ptr()), 'graph_search_packed')
    return out_i, out_d


def graph_record_bytes(links_per_node: int, M: int) -> int:
    import ctypes

    nb = ctypes.c_int64(0)
    check(lib().annlite_graph_record_bytes(int(links_per_node), int(M), ctypes.byref(nb)), 'graph_record_bytes')
    return int(nb.value)


def graph_build_sdc(codebooks: torch.Tensor) -> torch.Tensor:
    """Symmetric code-to-code L2 table f32 [M, Ks, Ks] of the GPU graph build (``annlite_graph_build_sdc``)."""
    M, Ks, dsub = codebooks.shape
    out = torch.empty((M, Ks, Ks), dtype=torch.float32, device=codebooks.device)
    check(lib().annlite_graph_build_sdc(codebooks.data_ptr(), M, Ks, dsub, out.data_ptr(), stream_ptr()), 'graph_build_sdc')
    return out


def graph_build_select(cand: torch.Tensor, base0: int, codes: torch.Tensor, sdc: torch.Tensor, max_keep: int,
                       links: torch.Tensor) -> torch.Tensor:
    """``annlite_graph_build_select``: the lists of the batch's points (nodes ``base0 + i``) from their candidate lists ``cand``
    i64 [b, ef]; writes ``links`` rows in place, returns the (target << 32 | source) pairs i64 [b, max_keep] (INT64_MAX = none)."""
    b, ef = cand.shape
    M, Ks = codes.shape[1], sdc.shape[1]
    pairs = torch.empty((b, int(max_keep)), dtype=torch.int64, device=codes.device)
    check(lib().annlite_graph_build_select(cand.data_ptr(), int(ef), b, int(base0), codes.data_ptr(), M, Ks, sdc.data_ptr(),
                                           int(max_keep), links.data_ptr(), links.shape[1] - 1, pairs.data_ptr(), stream_ptr()),
          'graph_build_select')
    return pairs


def graph_build_reverse(keys_sorted: torch.Tensor, seg: torch.Tensor, codes: torch.Tensor, sdc: torch.Tensor, links: torch.Tensor):
    """``annlite_graph_build_reverse``: reverse links for the sorted pairs; ``seg`` i64 [S + 1] bounds the runs of equal target."""
    M, Ks = codes.shape[1], sdc.shape[1]
    check(lib().annlite_graph_build_reverse(keys_sorted.data_ptr(), seg.data_ptr(), seg.numel() - 1, codes.data_ptr(), M, Ks,
                                            sdc.data_ptr(), links.data_ptr(), links.shape[1] - 1, stream_ptr()), 'graph_build_reverse')


def graph_pack_nodes(links: torch.Tensor, codes: torch.Tensor, nodes: torch.Tensor, packed: torch.Tensor, n_rows: int):
    """``annlite_graph_pack_nodes``: re-pack the records of ``nodes`` i64 [n] in place."""
    check(lib().annlite_graph_pack_nodes(links.data_ptr(), links.shape[1] - 1, codes.data_ptr(), int(n_rows), codes.shape[1],
                                         nodes.data_ptr(), nodes.numel(), packed.data_ptr(), stream_ptr()), 'graph_pack_nodes')


class ScanWorkspace:
    """Re-usable device scratch for the scan (avoids an allocation per search call).  One buffer PER STREAM:
    batches issued on different streams run concurrently (the next batch's kernels fill the CUs the previous
    scan's tail leaves idle) and must not share scratch."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, dev) -> torch.Tensor:
        key = stream_ptr()
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != dev:
            buf = torch.empty((max(nbytes, 8),), dtype=torch.uint8, device=dev)
            self.bufs[key] = buf
        return buf


def adc_scan_topk(codes: torch.Tensor, lut: torch.Tensor, B: int, k: int, M: int, Ks: int,
                  valid_bits: Optional[torch.Tensor] = None, row_base: int = 0, n_rows: Optional[int] = None,
                  codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None,
                  out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched flat ADC scan + exact top-k.  ``lut`` must be in the plan's layout (TILED when
    ``scan_plan(...).fast`` else BMK).  Returns (f32 [B,k], i64 [B,k]) ascending by (dist, id)."""
    N = codes.shape[0] if n_rows is None else n_rows
    cb = code_bytes_of(codes)
    plan = scan_plan(N, M, Ks, cb, B, k)
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(plan.workspace_bytes), dev)
    if out is None:
        od = torch.empty((B, k), dtype=torch.float32, device=dev)
        oi = torch.empty((B, k), dtype=torch.int64, device=dev)
    else:
        od, oi = out
    check(lib().annlite_adc_scan_topk(codes.data_ptr(), cb, codes_layout, N, M, Ks, _ptr(valid_bits), lut.data_ptr(), B, k,
                                      row_base, od.data_ptr(), oi.data_ptr(), ws.data_ptr(), ws.numel(),
                                      stream_ptr()), 'adc_scan_topk')
    return od, oi


def adc_scan_topk_packed(codes: torch.Tensor, lut: torch.Tensor, B: int, k: int, M: int, Ks: int,
                         valid_bits: Optional[torch.Tensor] = None, row_base: int = 0, n_rows: Optional[int] = None,
                         codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None) -> torch.Tensor:
    """``adc_scan_topk`` with the result in ONE i64 tensor [B, k, 2] = (global id, f32 distance bits): the
    buffer a rank contributes to the single all-gather of the row-sharded search."""
    N = codes.shape[0] if n_rows is None else n_rows
    cb = code_bytes_of(codes)
    plan = scan_plan(N, M, Ks, cb, B, k)
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(plan.workspace_bytes), dev)
    out = torch.empty((B, k, 2), dtype=torch.int64, device=dev)
    check(lib().annlite_adc_scan_topk_packed(codes.data_ptr(), cb, codes_layout, N, M, Ks, _ptr(valid_bits),
                                             lut.data_ptr(), B, k, row_base, out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), stream_ptr()), 'adc_scan_topk_packed')
    return out


def pq_search_topk(lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, k: int, M: int,
                   Ks: int, valid_bits: Optional[torch.Tensor] = None, row_base: int = 0, n_rows: Optional[int] = None,
                   codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None, packed: bool = False,
                   sqrt: bool = False, state=None):
    """LUT build + scan + top-k in one C call (``annlite_pq_search_topk_ex``).  ``queries`` f32 [B, D] already
    pre-processed (normalised for cosine).  Returns (f32 [B,k], i64 [B,k]) or, with ``packed``, i64 [B,k,2].
    ``state``: the table's ``_capi.ScanState`` (the library's kernel choice remembers what earlier launches measured)."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    cb = code_bytes_of(codes)
    need = ctypes.c_int64(0)
    check(lib().annlite_pq_search_workspace_bytes(N, M, Ks, cb, B, k, ctypes.byref(need)), 'pq_search_workspace_bytes')
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(need.value), dev)
    od = oi = op = None
    if packed:
        op = torch.empty((B, k, 2), dtype=torch.int64, device=dev)
    else:
        od = torch.empty((B, k), dtype=torch.float32, device=dev)
        oi = torch.empty((B, k), dtype=torch.int64, device=dev)
    check(lib().annlite_pq_search_topk_ex(lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), codes.data_ptr(), cb,
                                          codes_layout, N, M, Ks, _ptr(valid_bits), k, row_base, _ptr(od), _ptr(oi), _ptr(op),
                                          1 if sqrt else 0, ws.data_ptr(), ws.numel(), stream_ptr(),
                                          state.ptr if state is not None else None), 'pq_search_topk')
    return op if packed else (od, oi)


def debug_seed_candidates(queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, seed_rows: int,
                          valid_bits: Optional[torch.Tensor] = None, n_rows: Optional[int] = None,
                          codes_layout: int = CODES_PLAIN) -> Optional[torch.Tensor]:
    """Test hook (``annlite_debug_seed_candidates``): the rows the MFMA launch nominates for the seed bound, int64 [B, 512]
    (-1 = none); ``None`` where the launch does not apply (not M = 16 / 128-d, seed_rows not a multiple of 8192 ...)."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    M, Ks = codebooks.shape[0], codebooks.shape[1]
    out = torch.empty((B, 512), dtype=torch.int32, device=codes.device)
    rc = lib().annlite_debug_seed_candidates(queries.data_ptr(), B, D, codebooks.data_ptr(), codes.data_ptr(), codes_layout, N, M, Ks,
                                             _ptr(valid_bits), int(seed_rows), out.data_ptr(), stream_ptr())
    from ._capi import NOT_APPLICABLE

    if rc == NOT_APPLICABLE:
        return None
    check(rc, 'debug_seed_candidates')
    o = out.to(torch.int64)
    return torch.where(o < 0, o + (1 << 32), o).masked_fill(out == -1, -1)


def pq_search_split(phase: int, lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, k: int, M: int,
                    Ks: int, state, workspace: ScanWorkspace, valid_bits: Optional[torch.Tensor] = None, row_base: int = 0,
                    n_rows: Optional[int] = None, codes_layout: int = CODES_PLAIN, seed_rows: int = 0,
                    seed_keys: Optional[torch.Tensor] = None):
    """One half of the row-sharded search with a seed exchange (``annlite_pq_search_split``).
    ``PHASE_PREPARE`` -> the rank's seed keys i64 [B, SEED_KEYS], or None when the split does not apply to this batch (nothing
    was launched: make the plain call); ``PHASE_SCAN`` -> the packed result i64 [B, k, 2] of the prepared batch (same
    arguments, same workspace object, same stream)."""
    from ._capi import NOT_APPLICABLE, PHASE_PREPARE, SEED_KEYS

    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    cb = code_bytes_of(codes)
    need = ctypes.c_int64(0)
    check(lib().annlite_pq_search_workspace_bytes(N, M, Ks, cb, B, k, ctypes.byref(need)), 'pq_search_workspace_bytes')
    dev = codes.device
    ws = workspace.get(int(need.value), dev)
    op = None
    if phase == PHASE_PREPARE:
        seed_keys = torch.empty((B, SEED_KEYS), dtype=torch.int64, device=dev) if seed_keys is None else seed_keys
    else:
        op = torch.empty((B, k, 2), dtype=torch.int64, device=dev)
    rc = lib().annlite_pq_search_split(phase, int(seed_rows), lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), codes.data_ptr(),
                                       cb, codes_layout, N, M, Ks, _ptr(valid_bits), k, row_base, None, None, _ptr(op), 0,
                                       ws.data_ptr(), ws.numel(), stream_ptr(), state.ptr, _ptr(seed_keys))
    if rc == NOT_APPLICABLE:
        return None
    check(rc, 'pq_search_split')
    return seed_keys if phase == PHASE_PREPARE else op


def pq_search_seed_union(all_keys: torch.Tensor, codes: torch.Tensor, B: int, k: int, M: int, Ks: int, workspace: ScanWorkspace,
                         n_rows: Optional[int] = None) -> None:
    """``all_keys`` i64 [G, B, SEED_KEYS] (the all-gathered seed keys of the G ranks): tighten the prepared batch's bounds to the
    k-th smallest of every query's G * k keys (``annlite_pq_search_seed_union``; between the two halves of ``pq_search_split``)."""
    N = codes.shape[0] if n_rows is None else n_rows
    cb = code_bytes_of(codes)
    need = ctypes.c_int64(0)
    check(lib().annlite_pq_search_workspace_bytes(N, M, Ks, cb, B, k, ctypes.byref(need)), 'pq_search_workspace_bytes')
    ws = workspace.get(int(need.value), codes.device)
    assert all_keys.dtype == torch.int64 and all_keys.is_contiguous() and all_keys.shape[1] == B
    check(lib().annlite_pq_search_seed_union(all_keys.data_ptr(), all_keys.shape[0], N, M, Ks, cb, B, k, ws.data_ptr(), ws.numel(),
                                             stream_ptr()), 'pq_search_seed_union')


def ivf_select_cells(kind: int, queries: torch.Tensor, centroids: torch.Tensor, n_probe: int) -> torch.Tensor:
    """The ``n_probe`` nearest cells of every query, i32 [B, P] ascending in (distance, cell)
    (``AnnLite._cell_selection``, annlite/index.py:458-466).  kind 0: squared L2, 1: negative inner product."""
    B, D = queries.shape
    C = centroids.shape[0]
    cells = torch.empty((B, n_probe), dtype=torch.int32, device=queries.device)
    check(lib().annlite_ivf_select_cells(kind, queries.data_ptr(), B, D, centroids.data_ptr(), C, n_probe,
                                         cells.data_ptr(), stream_ptr()), 'ivf_select_cells')
    return cells


def ivf_max_tiles(B: int, P: int, C: int, qt: int) -> int:
    return int(lib().annlite_ivf_max_tiles(B, P, C, qt))


def ivf_plan(cells: torch.Tensor, n_cells: int, qt: int, cell_rows: torch.Tensor, cell_order: torch.Tensor):
    """(query, cell) pairs -> query tiles of ``qt`` slots probing one cell each.  Returns
    ``(vmap i32 [T*qt], slot_of i32 [B, P], tile_rows i64 [T, 2], n_tiles_used i32 [1])``."""
    B, P = cells.shape
    T = ivf_max_tiles(B, P, n_cells, qt)
    dev = cells.device
    vmap = torch.empty((T * qt,), dtype=torch.int32, device=dev)
    slot_of = torch.empty((B, P), dtype=torch.int32, device=dev)
    tile_rows = torch.empty((T, 2), dtype=torch.int64, device=dev)
    used = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib().annlite_ivf_plan(cells.data_ptr(), B, P, n_cells, qt, cell_rows.data_ptr(), cell_order.data_ptr(), T,
                                 vmap.data_ptr(), slot_of.data_ptr(), tile_rows.data_ptr(), used.data_ptr(),
                                 stream_ptr()), 'ivf_plan')
    return vmap, slot_of, tile_rows, used


def pq_search_tiles(lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, k: int,
                    M: int, Ks: int, tile_rows: torch.Tensor, vmap: torch.Tensor,
                    valid_bits: Optional[torch.Tensor] = None, n_rows: Optional[int] = None,
                    codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None, cand_cap: int = 256
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``annlite_pq_search_tiles``: REAL queries f32 [B, D] (their tables are built once); slot s of the V =
    ``vmap.numel()`` slots scans rows ``tile_rows[s // qt]`` with the tables of query ``vmap[s]``, integer sums only.
    Returns the per-slot candidate lists ``(cand i32 [V, cand_cap] table rows, count i32 [V])`` (count -1: overflow,
    re-score the whole cell)."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    V = vmap.numel()
    cb = code_bytes_of(codes)
    need = ctypes.c_int64(0)
    check(lib().annlite_pq_search_tiles_workspace_bytes(N, M, Ks, cb, V, k, ctypes.byref(need)),
          'pq_search_tiles_workspace_bytes')
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(need.value), dev)
    cand = torch.empty((V, cand_cap), dtype=torch.int32, device=dev)
    count = torch.empty((V,), dtype=torch.int32, device=dev)
    check(lib().annlite_pq_search_tiles(lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), codes.data_ptr(),
                                        cb, codes_layout, N, M, Ks, _ptr(valid_bits), k, V, tile_rows.data_ptr(),
                                        vmap.data_ptr(), cand.data_ptr(), cand_cap, count.data_ptr(), ws.data_ptr(),
                                        ws.numel(), stream_ptr()), 'pq_search_tiles')
    return cand, count


def ivf_rescore(lut_bmk: torch.Tensor, codes_plain: torch.Tensor, cand: torch.Tensor, count: torch.Tensor,
                slot_of: torch.Tensor, tile_rows: torch.Tensor, qt: int, k: int, row_ids: Optional[torch.Tensor] = None,
                valid_bits: Optional[torch.Tensor] = None, id_base: int = 0, sqrt: bool = False
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact distances of every query's candidate rows + top-k: ``lut_bmk`` f32 [B, M, Ks] (``get_dist_mat``),
    ``codes_plain`` u8 [N, M] (cell-sorted, PLAIN).  Returns ([B,k] f32, [B,k] i64 ids)."""
    B, M, Ks = lut_bmk.shape
    P = slot_of.shape[1]
    od = torch.empty((B, k), dtype=torch.float32, device=lut_bmk.device)
    oi = torch.empty((B, k), dtype=torch.int64, device=lut_bmk.device)
    check(lib().annlite_ivf_rescore(lut_bmk.data_ptr(), B, M, Ks, codes_plain.data_ptr(), codes_plain.shape[0],
                                    _ptr(valid_bits), cand.data_ptr(), cand.shape[1], count.data_ptr(), slot_of.data_ptr(),
                                    P, tile_rows.data_ptr(), qt, _ptr(row_ids), id_base, k, od.data_ptr(), oi.data_ptr(),
                                    1 if sqrt else 0, stream_ptr()), 'ivf_rescore')
    return od, oi


def ivf_candidate_ids(cand: torch.Tensor, count: torch.Tensor, slot_of: torch.Tensor, R: int,
                      row_ids: Optional[torch.Tensor] = None, id_base: int = 0) -> torch.Tensor:
    """Every query's candidate lists as one dense i64 row [B, R] of external ids, padded with -1."""
    B, P = slot_of.shape
    out = torch.empty((B, R), dtype=torch.int64, device=cand.device)
    check(lib().annlite_ivf_candidate_ids(cand.data_ptr(), cand.shape[1], count.data_ptr(), slot_of.data_ptr(), B, P,
                                          _ptr(row_ids), id_base, out.data_ptr(), R, stream_ptr()), 'ivf_candidate_ids')
    return out


def ivf_search_topk(lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, cells: torch.Tensor, n_cells: int,
                    cell_rows: torch.Tensor, cell_order: torch.Tensor, k: int, M: int, Ks: int,
                    row_ids: Optional[torch.Tensor] = None, valid_bits: Optional[torch.Tensor] = None,
                    n_rows: Optional[int] = None, codes_layout: int = CODES_PLAIN, id_base: int = 0, sqrt: bool = False,
                    workspace: Optional[ScanWorkspace] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``annlite_ivf_search_topk``: the pruned search over cells on the byte-table kernel in one call -- plan, preparation
    launch (tables, per-cell seed bounds, per-query byte tables), scan in cell tiles with exact sums, merge of the per-cell
    lists.  ``queries`` f32 [B, D] (the tables of ``lut_kind`` -- LUT_L2 or LUT_IPDIST -- are built from them and ``codebooks`` [M, Ks, D/M]), ``codes`` the
    CELL-SORTED table, ``cells`` i32 [B, P] nearest first.  M = 16, Ks <= 256, k <= 16.  Returns ([B,k] f32, [B,k] i64)."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    P = cells.shape[1]
    need = ctypes.c_int64(0)
    check(lib().annlite_ivf_search_topk_workspace_bytes(B, P, n_cells, M, Ks, k, ctypes.byref(need)), 'ivf_search_topk_workspace_bytes')
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(need.value), dev)
    od = torch.empty((B, k), dtype=torch.float32, device=dev)
    oi = torch.empty((B, k), dtype=torch.int64, device=dev)
    check(lib().annlite_ivf_search_topk(lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), M, Ks, codes.data_ptr(), codes_layout, N,
                                        _ptr(valid_bits), cells.data_ptr(), P, n_cells, cell_rows.data_ptr(), cell_order.data_ptr(),
                                        _ptr(row_ids), id_base, k, od.data_ptr(), oi.data_ptr(), 1 if sqrt else 0, ws.data_ptr(),
                                        ws.numel(), stream_ptr()), 'ivf_search_topk')
    return od, oi


def ivf_search_candidates(lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, cells: torch.Tensor,
                          n_cells: int, cell_rows: torch.Tensor, cell_order: torch.Tensor, k: int, M: int, Ks: int,
                          row_ids: Optional[torch.Tensor] = None, valid_bits: Optional[torch.Tensor] = None,
                          n_rows: Optional[int] = None, codes_layout: int = CODES_PLAIN, id_base: int = 0,
                          workspace: Optional[ScanWorkspace] = None, bound_rank: int = 1,
                          seed_cells: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``annlite_ivf_search_candidates``: the pruned search's pipeline as the candidate generator of an exact re-rank -- every
    (query, probed cell) list on its own (the cell's best <= k rows at or below the query's first bound, the
    ``min(bound_rank * k, 64)``-th smallest seed sum of its nearest cell -- of entry ``seed_cells[b]`` (i32 [B]) of the cell table when given).
    Returns i64 [B, P * k] external ids, -1 = none."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    P = cells.shape[1]
    need = ctypes.c_int64(0)
    check(lib().annlite_ivf_search_topk_workspace_bytes(B, P, n_cells, M, Ks, k, ctypes.byref(need)), 'ivf_search_topk_workspace_bytes')
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(need.value), dev)
    out = torch.empty((B, P * k), dtype=torch.int64, device=dev)
    check(lib().annlite_ivf_search_candidates(lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), M, Ks, codes.data_ptr(), codes_layout,
                                              N, _ptr(valid_bits), cells.data_ptr(), P, n_cells, cell_rows.data_ptr(), cell_order.data_ptr(),
                                              _ptr(row_ids), id_base, k, bound_rank, _ptr(seed_cells), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                              stream_ptr()),
          'ivf_search_candidates')
    return out


def topk_merge_packed(packed: torch.Tensor, sqrt: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """[G,B,k,2] i64 (id, distance bits) -> ([B,k] f32, [B,k] i64), same order rule as ``topk_merge``."""
    G, B, k, _ = packed.shape
    od = torch.empty((B, k), dtype=torch.float32, device=packed.device)
    oi = torch.empty((B, k), dtype=torch.int64, device=packed.device)
    check(lib().annlite_topk_merge_packed(packed.contiguous().data_ptr(), G, B, k, od.data_ptr(), oi.data_ptr(),
                                          1 if sqrt else 0, stream_ptr()), 'topk_merge_packed')
    return od, oi


def adc_scan_candidates(codes: torch.Tensor, lut: torch.Tensor, B: int, k: int, M: int, Ks: int,
                        valid_bits: Optional[torch.Tensor] = None, row_base: int = 0, n_rows: Optional[int] = None,
                        codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Unmerged per-slice top-k lists: (f32 [B, n_slices*k], i64 [B, n_slices*k]); superset of the top-k."""
    N = codes.shape[0] if n_rows is None else n_rows
    cb = code_bytes_of(codes)
    plan = scan_plan(N, M, Ks, cb, B, k)
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(plan.workspace_bytes), dev)
    R = plan.n_slices * k
    od = torch.empty((B, R), dtype=torch.float32, device=dev)
    oi = torch.empty((B, R), dtype=torch.int64, device=dev)
    check(lib().annlite_adc_scan_candidates(codes.data_ptr(), cb, codes_layout, N, M, Ks, _ptr(valid_bits),
                                            lut.data_ptr(), B, k, row_base, od.data_ptr(), oi.data_ptr(),
                                            ws.data_ptr(), ws.numel(), stream_ptr()), 'adc_scan_candidates')
    return od, oi


def pq_search_candidates(lut_kind: int, queries: torch.Tensor, codebooks: torch.Tensor, codes: torch.Tensor, k: int, M: int, Ks: int,
                         valid_bits: Optional[torch.Tensor] = None, row_base: int = 0, n_rows: Optional[int] = None,
                         codes_layout: int = CODES_PLAIN, workspace: Optional[ScanWorkspace] = None
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``annlite_pq_search_candidates``: table build + candidate generator in one C call -- (f32 [B, n_slices*k], i64 [B, n_slices*k]),
    every slice's k best rows at or below the table-wide first bound (a superset of the table's top-k; -1 / +inf padded)."""
    N = codes.shape[0] if n_rows is None else n_rows
    B, D = queries.shape
    cb = code_bytes_of(codes)
    plan = scan_plan(N, M, Ks, cb, B, k)
    need = ctypes.c_int64(0)
    check(lib().annlite_pq_search_workspace_bytes(N, M, Ks, cb, B, k, ctypes.byref(need)), 'pq_search_workspace_bytes')
    dev = codes.device
    ws = (workspace or ScanWorkspace()).get(int(need.value), dev)
    R = plan.n_slices * k
    od = torch.empty((B, R), dtype=torch.float32, device=dev)
    oi = torch.empty((B, R), dtype=torch.int64, device=dev)
    check(lib().annlite_pq_search_candidates(lut_kind, queries.data_ptr(), B, D, codebooks.data_ptr(), codes.data_ptr(), cb, codes_layout,
                                             N, M, Ks, _ptr(valid_bits), k, row_base, od.data_ptr(), oi.data_ptr(), ws.data_ptr(),
                                             ws.numel(), stream_ptr()), 'pq_search_candidates')
    return od, oi


def topk_merge(dist: torch.Tensor, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[G,B,k] x2 -> [B,k] x2 (the merge after the RCCL all-gather of per-shard top-k)."""
    G, B, k = dist.shape
    od = torch.empty((B, k), dtype=torch.float32, device=dist.device)
    oi = torch.empty((B, k), dtype=torch.int64, device=dist.device)
    check(lib().annlite_topk_merge(dist.data_ptr(), ids.data_ptr(), G, B, k, od.data_ptr(), oi.data_ptr(),
                                   stream_ptr()), 'topk_merge')
    return od, oi


def topk_rows(values: torch.Tensor, k: int, id_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    B, N = values.shape
    od = torch.empty((B, k), dtype=torch.float32, device=values.device)
    oi = torch.empty((B, k), dtype=torch.int64, device=values.device)
    check(lib().annlite_topk_rows(values.data_ptr(), B, N, k, id_base, od.data_ptr(), oi.data_ptr(), stream_ptr()),
          'topk_rows')
    return od, oi


def codes_skew(codes: torch.Tensor, ids: Optional[torch.Tensor] = None, id_base: int = 0,
               out: Optional[torch.Tensor] = None, inverse: bool = False) -> torch.Tensor:
    """PLAIN rows -> SKEWED table rows (scatter by ids) or back (inverse gather)."""
    assert codes.dtype == torch.uint8
    if inverse:
        n = ids.shape[0] if ids is not None else codes.shape[0]
        M = codes.shape[1]
        out = torch.empty((n, M), dtype=torch.uint8, device=codes.device) if out is None else out
    else:
        n, M = codes.shape
        out = torch.empty_like(codes) if out is None else out
    check(lib().annlite_codes_skew(codes.data_ptr(), n, M, _ptr(ids), id_base, out.data_ptr(), int(inverse),
                                   stream_ptr()), 'codes_skew')
    return out


# -------------------------------------------------------------------------------------------- codec
def pq_encode(x: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    N, D = x.shape
    M, Ks, dsub = codebooks.shape
    assert D == M * dsub, 'input dimension must be Ds * M'
    cb = 1 if Ks <= 256 else (2 if Ks <= 65536 else 4)
    out = torch.empty((N, M), dtype=_CODE_TORCH[cb], device=x.device)
    check(lib().annlite_pq_encode(x.data_ptr(), N, D, codebooks.data_ptr(), M, Ks, out.data_ptr(), cb, stream_ptr()),
          'pq_encode')
    return out


def pq_decode(codes: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    N, M = codes.shape
    Mc, Ks, dsub = codebooks.shape
    assert M == Mc
    out = torch.empty((N, M * dsub), dtype=torch.float32, device=codes.device)
    check(lib().annlite_pq_decode(codes.data_ptr(), code_bytes_of(codes), N, M, Ks, codebooks.data_ptr(), dsub,
                                  out.data_ptr(), stream_ptr()), 'pq_decode')
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    N, D = x.shape
    out = torch.empty_like(x)
    check(lib().annlite_l2_normalize(x.data_ptr(), N, D, out.data_ptr(), stream_ptr()), 'l2_normalize')
    return out


def kmeans_assign_accumulate(x, codebooks, sums, counts, inertia=None):
    N, D = x.shape
    M, Ks, dsub = codebooks.shape
    check(lib().annlite_kmeans_assign_accumulate(x.data_ptr(), N, D, codebooks.data_ptr(), M, Ks, sums.data_ptr(),
                                                 counts.data_ptr(), _ptr(inertia), stream_ptr()),
          'kmeans_assign_accumulate')


def kmeans_update(sums, counts, codebooks):
    M, Ks, dsub = codebooks.shape
    check(lib().annlite_kmeans_update(sums.data_ptr(), counts.data_ptr(), M, Ks, dsub, codebooks.data_ptr(),
                                      stream_ptr()), 'kmeans_update')


def exact_gather_dist(metric: int, queries: torch.Tensor, vectors: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    B, D = queries.shape
    N = vectors.shape[0]
    R = cand.shape[1]
    out = torch.empty((B, R), dtype=torch.float32, device=queries.device)
    check(lib().annlite_exact_gather_dist(int(metric), queries.data_ptr(), B, D, vectors.data_ptr(), N,
                                          cand.data_ptr(), R, out.data_ptr(), stream_ptr()), 'exact_gather_dist')
    return out


def rerank_topk(metric: int, queries: torch.Tensor, vectors: torch.Tensor, cand: torch.Tensor, k: int,
                valid_bits: Optional[torch.Tensor] = None, sqrt: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """``annlite_rerank_topk``: exact distances of the candidate lists ``cand`` i64 [B, R] fused with the top-k (k <= 64):
    ``(dist f32 [B, k], ids i64 [B, k])`` in (distance, list position) order, (+inf, -1) padded."""
    B, D = queries.shape
    R = cand.shape[1]
    out_d = torch.empty((B, k), dtype=torch.float32, device=queries.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=queries.device)
    check(lib().annlite_rerank_topk(int(metric), queries.data_ptr(), B, D, vectors.data_ptr(), vectors.shape[0], cand.data_ptr(), R,
                                    _ptr(valid_bits), int(k), 1 if sqrt else 0, out_d.data_ptr(), out_i.data_ptr(),
                                    stream_ptr()), 'rerank_topk')
    return out_d, out_i
