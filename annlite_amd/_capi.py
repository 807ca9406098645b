"""ctypes binding of ``libannlite_hip.so`` (C ABI declared in ``include/annlite_hip.h``).

The product path has NO fallback: if the shared library is missing, or a kernel launch fails
(e.g. no MI355X visible), a ``RuntimeError`` is raised -- nothing here ever routes to CPU code.

Device pointers come from ``torch.Tensor.data_ptr()``; tensors stay owned by PyTorch.  The HIP
stream handed to every call is ``torch.cuda.current_stream().cuda_stream`` so launches order with
the caller's torch work (PyTorch is plumbing here: memory, streams, ``torch.distributed``).
"""
import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libannlite_hip.so'
# ANNLITE_HIP_LIB: another build of the same library (A/B measurements of compile-time variants); default: the in-tree one
LIB_PATH = os.environ.get('ANNLITE_HIP_LIB') or os.path.join(_HERE, LIB_NAME)

ANNLITE_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = 1, 2, 3, 4
NOT_APPLICABLE = 5  # (not an error: annlite_pq_search_split has no effect for this shape / state)
PHASE_PREPARE, PHASE_SCAN, SEED_KEYS = 1, 2, 16

LUT_L2, LUT_IP, LUT_IPDIST = 1, 2, 3
LAYOUT_BMK, LAYOUT_TILED = 0, 1
CODES_PLAIN, CODES_SKEWED = 0, 1

# every symbol include/annlite_hip.h declares (tests/test_capi_symbols.py checks header == this list)
SYMBOLS = (
    'annlite_hip_abi_version',
    'annlite_hip_last_error',
    'annlite_hip_device_count',
    'annlite_hip_device_arch',
    'annlite_knobs_reload',
    'annlite_scan_plan_query',
    'annlite_scan_plan_tiles',
    'annlite_scan_state_create',
    'annlite_scan_state_destroy',
    'annlite_scan_state_reset',
    'annlite_scan_state_info',
    'annlite_lut_build',
    'annlite_lut_retile',
    'annlite_adc_dist',
    'annlite_adc_gather',
    'annlite_graph_search',
    'annlite_graph_search_stats',
    'annlite_graph_search_stats_ex',
    'annlite_graph_record_bytes',
    'annlite_graph_pack',
    'annlite_graph_search_packed',
    'annlite_graph_search_packed_ex',
    'annlite_rerank_topk',
    'annlite_graph_build_sdc',
    'annlite_graph_build_select',
    'annlite_graph_build_reverse',
    'annlite_graph_pack_nodes',
    'annlite_adc_scan_topk',
    'annlite_adc_scan_topk_packed',
    'annlite_pq_search_workspace_bytes',
    'annlite_pq_search_topk',
    'annlite_pq_search_topk_ex',
    'annlite_pq_search_split',
    'annlite_pq_search_seed_union',
    'annlite_adc_scan_candidates',
    'annlite_pq_search_candidates',
    'annlite_topk_merge',
    'annlite_topk_merge_packed',
    'annlite_topk_rows',
    'annlite_pq_encode',
    'annlite_pq_decode',
    'annlite_l2_normalize',
    'annlite_kmeans_assign_accumulate',
    'annlite_kmeans_update',
    'annlite_exact_gather_dist',
    'annlite_ivf_select_cells',
    'annlite_ivf_max_tiles',
    'annlite_ivf_plan',
    'annlite_ivf_max_tiles_first',
    'annlite_ivf_plan_first',
    'annlite_pq_search_tiles_workspace_bytes',
    'annlite_pq_search_tiles',
    'annlite_ivf_rescore',
    'annlite_ivf_candidate_ids',
    'annlite_ivf_search_topk_workspace_bytes',
    'annlite_ivf_search_topk',
    'annlite_ivf_search_candidates',
    'annlite_ivf_merge_lists',
    'annlite_codes_skew',
    'annlite_profile_enable',
    'annlite_profile_last_scan_ms',
    'annlite_profile_last_scan_clock_mhz',
    'annlite_kernel_rev',
    'annlite_debug_counters',
    'annlite_debug_timeline',
    'annlite_debug_items',
    'annlite_debug_prep_timeline',
    'annlite_debug_seed_candidates',
)


class ScanPlan(ctypes.Structure):
    _fields_ = [
        ('fast', ctypes.c_int32),
        ('qi', ctypes.c_int32),
        ('qt', ctypes.c_int32),
        ('waves', ctypes.c_int32),
        ('n_slices', ctypes.c_int32),
        ('max_k', ctypes.c_int32),
        ('lut_floats', ctypes.c_int64),
        ('workspace_bytes', ctypes.c_int64),
    ]


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load the HIP library; raise loudly if it has not been built (``python __graft_entry__.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension has not been built '
            f'(run `make -C annlite_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`). '
            f'annlite_amd has no CPU fallback.'
        )
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already mapped
    # when our library's DT_NEEDED libamdhip64 is resolved, otherwise two runtimes coexist and device
    # pointers allocated by torch are foreign to our launches ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    i64, i32, vp, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    L.annlite_hip_abi_version.restype = i32
    L.annlite_hip_last_error.restype = ctypes.c_char_p
    L.annlite_hip_device_count.argtypes = [ctypes.POINTER(i32)]
    L.annlite_hip_device_arch.argtypes = [i32, ctypes.c_char_p, sz]
    L.annlite_scan_plan_query.argtypes = [i64, i64, i64, i32, i64, i64, ctypes.POINTER(ScanPlan)]
    L.annlite_scan_plan_tiles.argtypes = [i64, i64, i64, i32, i64, i64, ctypes.POINTER(ScanPlan)]
    L.annlite_lut_build.argtypes = [i32, vp, i64, i64, vp, i64, i64, vp, i32, i32, vp]
    L.annlite_lut_retile.argtypes = [vp, i64, i64, i64, vp, i32, vp]
    L.annlite_adc_dist.argtypes = [vp, i64, i64, vp, i32, i64, vp, vp]
    L.annlite_adc_gather.argtypes = [vp, i64, i64, i64, vp, i32, i64, vp, i64, vp, vp]
    L.annlite_graph_search.argtypes = [vp, i32, vp, i64, vp, i64, i64, i64, vp, vp, i64, i32, vp, vp, vp]
    L.annlite_graph_search_packed.argtypes = [vp, i32, vp, i64, vp, i64, i64, i64, vp, vp, i64, i32, vp, vp, vp]
    L.annlite_graph_search_packed_ex.argtypes = [vp, i32, vp, i64, vp, i64, i64, i64, vp, vp, i64, i32, i32, vp, vp, vp]
    L.annlite_graph_pack.argtypes = [vp, i32, vp, i64, i64, vp, vp]
    L.annlite_rerank_topk.argtypes = [i32, vp, i64, i64, vp, i64, vp, i64, vp, i64, i32, vp, vp, vp]
    L.annlite_graph_build_sdc.argtypes = [vp, i64, i64, i64, vp, vp]
    L.annlite_graph_build_select.argtypes = [vp, i32, i64, i64, vp, i64, i64, vp, i32, vp, i32, vp, vp]
    L.annlite_graph_build_reverse.argtypes = [vp, vp, i64, vp, i64, i64, vp, vp, i32, vp]
    L.annlite_graph_pack_nodes.argtypes = [vp, i32, vp, i64, i64, vp, i64, vp, vp]
    L.annlite_graph_record_bytes.argtypes = [i32, i64, ctypes.POINTER(ctypes.c_int64)]
    L.annlite_adc_scan_topk.argtypes = [vp, i32, i32, i64, i64, i64, vp, vp, i64, i64, i64, vp, vp, vp, sz, vp]
    L.annlite_adc_scan_candidates.argtypes = L.annlite_adc_scan_topk.argtypes
    L.annlite_pq_search_candidates.argtypes = [i32, vp, i64, i64, vp, vp, i32, i32, i64, i64, i64, vp, i64, i64, vp, vp, vp, sz, vp]
    L.annlite_adc_scan_topk_packed.argtypes = [vp, i32, i32, i64, i64, i64, vp, vp, i64, i64, i64, vp, vp, sz, vp]
    L.annlite_pq_search_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i64, ctypes.POINTER(ctypes.c_int64)]
    L.annlite_pq_search_topk.argtypes = [i32, vp, i64, i64, vp, vp, i32, i32, i64, i64, i64, vp, i64, i64, vp, vp, vp, i32,
                                         vp, sz, vp]
    L.annlite_pq_search_topk_ex.argtypes = L.annlite_pq_search_topk.argtypes + [vp]
    L.annlite_pq_search_split.argtypes = [i32, i64] + L.annlite_pq_search_topk_ex.argtypes + [vp]
    L.annlite_pq_search_seed_union.argtypes = [vp, i64, i64, i64, i64, i32, i64, i64, vp, sz, vp]
    L.annlite_scan_state_create.argtypes = [ctypes.POINTER(vp)]
    L.annlite_scan_state_destroy.argtypes = [vp]
    L.annlite_scan_state_reset.argtypes = [vp]
    L.annlite_scan_state_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_uint64)]
    L.annlite_topk_merge.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
    L.annlite_topk_merge_packed.argtypes = [vp, i64, i64, i64, vp, vp, i32, vp]
    L.annlite_topk_rows.argtypes = [vp, i64, i64, i64, i64, vp, vp, vp]
    L.annlite_pq_encode.argtypes = [vp, i64, i64, vp, i64, i64, vp, i32, vp]
    L.annlite_pq_decode.argtypes = [vp, i32, i64, i64, i64, vp, i64, vp, vp]
    L.annlite_l2_normalize.argtypes = [vp, i64, i64, vp, vp]
    L.annlite_kmeans_assign_accumulate.argtypes = [vp, i64, i64, vp, i64, i64, vp, vp, vp, vp]
    L.annlite_kmeans_update.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    L.annlite_exact_gather_dist.argtypes = [i32, vp, i64, i64, vp, i64, vp, i64, vp, vp]
    L.annlite_codes_skew.argtypes = [vp, i64, i64, vp, i64, vp, i32, vp]
    L.annlite_ivf_select_cells.argtypes = [i32, vp, i64, i64, vp, i64, i64, vp, vp]
    L.annlite_ivf_max_tiles.argtypes = [i64, i64, i64, i64]
    L.annlite_ivf_plan.argtypes = [vp, i64, i64, i64, i64, vp, vp, i64, vp, vp, vp, vp, vp]
    L.annlite_ivf_max_tiles_first.argtypes = [i64, i64, i64, i64]
    L.annlite_ivf_plan_first.argtypes = [vp, i64, i64, i64, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.annlite_pq_search_tiles_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i64, ctypes.POINTER(ctypes.c_int64)]
    L.annlite_pq_search_tiles.argtypes = [i32, vp, i64, i64, vp, vp, i32, i32, i64, i64, i64, vp, i64, i64, vp, vp, vp,
                                          i64, vp, vp, sz, vp]
    L.annlite_ivf_candidate_ids.argtypes = [vp, i64, vp, vp, i64, i64, vp, i64, vp, i64, vp]
    L.annlite_ivf_rescore.argtypes = [vp, i64, i64, i64, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, vp, i64, i64, vp,
                                      vp, i32, vp]
    L.annlite_ivf_search_topk_workspace_bytes.argtypes = [i64, i64, i64, i64, i64, i64, ctypes.POINTER(ctypes.c_int64)]
    L.annlite_ivf_search_topk.argtypes = [i32, vp, i64, i64, vp, i64, i64, vp, i32, i64, vp, vp, i64, i64, vp, vp, vp, i64, i64, vp, vp,
                                          i32, vp, sz, vp]
    L.annlite_ivf_search_candidates.argtypes = [i32, vp, i64, i64, vp, i64, i64, vp, i32, i64, vp, vp, i64, i64, vp, vp, vp, i64, i64, i64,
                                                vp, vp, vp, sz, vp]
    L.annlite_ivf_merge_lists.argtypes = [vp, i64, vp, i64, i64, vp, i64, vp, vp, i32, vp]
    L.annlite_profile_enable.argtypes = [i32]
    L.annlite_profile_last_scan_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.annlite_profile_last_scan_clock_mhz.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.annlite_kernel_rev.argtypes = [ctypes.c_char_p]
    L.annlite_debug_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.annlite_debug_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.annlite_debug_items.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.annlite_debug_prep_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.annlite_debug_seed_candidates.argtypes = [vp, i64, i64, vp, vp, i32, i64, i64, i64, vp, i64, vp, vp]
    L.annlite_graph_search_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.annlite_graph_search_stats_ex.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError here == the .so does not export a declared symbol
        if name in ('annlite_ivf_max_tiles', 'annlite_ivf_max_tiles_first'):
            fn.restype = i64
        elif name not in ('annlite_hip_last_error',):
            fn.restype = i32
    _lib = L
    return L


def last_error() -> str:
    return lib().annlite_hip_last_error().decode('utf-8', 'replace')


def check(rc: int, what: str = '') -> None:
    """Map the C status codes onto the exception types the reference raises for the same
    conditions (SURVEY.md section 8b "Error conventions"): argument/shape problems are
    ``AssertionError`` in the reference (python ``assert``), everything else ``RuntimeError``."""
    if rc == ANNLITE_OK:
        return
    msg = f'{what}: {last_error()}' if what else last_error()
    if rc == ERR_INVALID:
        raise AssertionError(msg)
    raise RuntimeError(f'annlite_hip error {rc}: {msg}')


def knobs_reload() -> None:
    """The library parses its ANNLITE_* switches once, when it is loaded; a process that changes them afterwards (tests, A/B
    measurements) has them parsed again with this.  No-op before the library is loaded (it will read the environment then)."""
    if _lib is not None:
        check(_lib.annlite_knobs_reload(), 'knobs_reload')


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = lib().annlite_hip_device_count(ctypes.byref(n))
    return int(n.value) if rc == ANNLITE_OK else 0


def device_arch(dev: int = 0) -> str:
    buf = ctypes.create_string_buffer(256)
    check(lib().annlite_hip_device_arch(dev, buf, 256), 'device_arch')
    return buf.value.decode()


_gpu_seen = False


def require_gpu() -> None:
    global _gpu_seen
    if _gpu_seen:  # (torch.cuda.is_available() costs microseconds of python per call: seven calls per batch on the search path)
        return
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError('annlite_amd needs an AMD GPU (MI355X / gfx950); no HIP device is visible and there is no CPU fallback')
    _gpu_seen = True


_raw_stream = None


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device, as an integer.  torch.cuda.current_stream() builds a python Stream
    object through three layers of device-index resolution (~3 us, five times per batch with the exchange on); the raw getter is one C
    call."""
    global _raw_stream
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return int(_raw_stream(torch._C._cuda_getDevice()))
    return int(torch.cuda.current_stream().cuda_stream)


def scan_plan(N: int, M: int, Ks: int, code_bytes: int, B: int, k: int) -> ScanPlan:
    p = ScanPlan()
    check(lib().annlite_scan_plan_query(N, M, Ks, code_bytes, B, k, ctypes.byref(p)), 'scan_plan')
    return p


def scan_plan_tiles(N: int, M: int, Ks: int, code_bytes: int, V: int, k: int) -> ScanPlan:
    """The plan of ``annlite_pq_search_tiles`` (cells): its query tiles hold ``qt`` slots each."""
    p = ScanPlan()
    check(lib().annlite_scan_plan_tiles(N, M, Ks, code_bytes, V, k, ctypes.byref(p)), 'scan_plan_tiles')
    return p


class ScanState:
    """The caller-owned, per-code-table memory of the library's kernel choice (``annlite_scan_state``): which of the two
    M = 16 scan kernels suits the table is measured by the launches themselves and read back from host-mapped memory by
    the next calls.  One per index; not for two threads at once."""

    KERNELS = {0: 'undecided', 1: 'byte tables', 2: 'u16 tables'}

    def __init__(self):
        p = ctypes.c_void_p()
        check(lib().annlite_scan_state_create(ctypes.byref(p)), 'scan_state_create')
        self._p = p

    @property
    def ptr(self):
        return self._p

    def reset(self):
        check(lib().annlite_scan_state_reset(self._p), 'scan_state_reset')

    def info(self):
        """(kernel: 0 undecided / 1 byte tables / 2 u16 tables, rows it was decided at, candidates of the deciding launch)"""
        kern, rows, cand = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_uint64(0)
        check(lib().annlite_scan_state_info(self._p, ctypes.byref(kern), ctypes.byref(rows), ctypes.byref(cand)), 'scan_state_info')
        return int(kern.value), int(rows.value), int(cand.value)

    def __del__(self):
        try:
            if self._p:
                import torch

                torch.cuda.synchronize()  # (launches that were given the state may still write its block)
                lib().annlite_scan_state_destroy(self._p)
                self._p = None
        except Exception:
            pass


def profile_enable(on: bool) -> None:
    check(lib().annlite_profile_enable(int(bool(on))), 'profile_enable')


def profile_last_scan_ms() -> float:
    ms = ctypes.c_float(0.0)
    check(lib().annlite_profile_last_scan_ms(ctypes.byref(ms)), 'profile_last_scan_ms')
    return float(ms.value)


def kernel_rev(kernel: str) -> int:
    """Revision of the kernel's memory behaviour (0: unknown name) -- what profiles/traffic.json entries are tagged with."""
    return int(lib().annlite_kernel_rev(kernel.encode()))


def profile_last_scan_clock_mhz():
    """Shader clock (MHz) the last profiled byte-table scan held, or None (another kernel served the launch)."""
    mhz = ctypes.c_float(0.0)
    if lib().annlite_profile_last_scan_clock_mhz(ctypes.byref(mhz)) != 0:
        return None
    return float(mhz.value)


def graph_search_stats():
    """(expansions, rows evaluated) of the last GPU graph walk (ANNLITE_DEBUG_COUNTERS=1)."""
    out = (ctypes.c_uint64 * 2)()
    check(lib().annlite_graph_search_stats(out), 'graph_search_stats')
    return int(out[0]), int(out[1])


def graph_search_stats_ex():
    """(expansions, rows evaluated, prefetched records used) of the last GPU graph walk (ANNLITE_DEBUG_COUNTERS=1)."""
    out = (ctypes.c_uint64 * 8)()
    check(lib().annlite_graph_search_stats_ex(out), 'graph_search_stats_ex')
    graph_search_stats_ex.cycles = {'seed': int(out[3]), 'record': int(out[4]), 'visited': int(out[5]), 'sums': int(out[6]), 'merge': int(out[7])}
    return int(out[0]), int(out[1]), int(out[2])


def debug_timeline():
    """Phase stamps of the byte-table kernel's workgroups (see ``annlite_debug_timeline``), as microseconds:
    dict(span, start_spread, build, scan, wait, merge) -- per-work-item averages except the span."""
    out = (ctypes.c_uint64 * 8)()
    check(lib().annlite_debug_timeline(out), 'debug_timeline')
    v = [int(x) for x in out]
    n = max(v[7], 1)
    t0 = (1 << 62) - v[0]
    return {'items': v[7], 'span_us': (v[1] - t0) / 100.0, 'avg_start_us': (v[2] / n - t0) / 100.0,
            'build_us': v[3] / n / 100.0, 'scan_us': v[4] / n / 100.0, 'wait_us': v[5] / n / 100.0, 'merge_us': v[6] / n / 100.0}


def debug_prep_timeline():
    """Phase durations (microseconds) of the preparation launch's first and last workgroup (``annlite_debug_prep_timeline``):
    [(build, seed rows, selection [+ byte tables]), ...] and the start of the last workgroup relative to the first."""
    out = (ctypes.c_uint64 * 8)()
    check(lib().annlite_debug_prep_timeline(out), 'debug_prep_timeline')
    v = [int(x) for x in out]
    ph = lambda o: {'build_us': (v[o + 1] - v[o]) / 100.0, 'seed_us': (v[o + 2] - v[o + 1]) / 100.0, 'select_us': (v[o + 3] - v[o + 2]) / 100.0}
    return {'first': ph(0), 'last': ph(4), 'last_start_after_first_us': (v[4] - v[0]) / 100.0, 'span_us': (max(v[3], v[7]) - v[0]) / 100.0}


def debug_items():
    """Per-work-item records of the byte-table kernel (``annlite_debug_items``): int64 array [n, 8] = tile, slice, five
    100 MHz stamps (start, table built, step loop left, last barrier passed, end), block index."""
    import numpy as np

    buf = (ctypes.c_uint64 * (4096 * 8))()
    n = ctypes.c_int64(0)
    check(lib().annlite_debug_items(buf, 4096, ctypes.byref(n)), 'debug_items')
    return np.frombuffer(buf, dtype=np.uint64)[: n.value * 8].reshape(-1, 8).astype(np.int64)


def debug_counters():
    out = (ctypes.c_uint64 * 8)()
    check(lib().annlite_debug_counters(out), 'debug_counters')
    return [int(v) for v in out]
