"""ctypes binding of ``libannlite_graph.so`` (``include/annlite_graph.h``): the host-side HNSW-over-PQ
candidate generator of BASELINE config 5.  Pure CPU library (g++/OpenMP); it needs no GPU to load."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libannlite_graph.so')

SYMBOLS = (
    'annlite_hnsw_create',
    'annlite_hnsw_free',
    'annlite_hnsw_last_error',
    'annlite_hnsw_size',
    'annlite_hnsw_reserve',
    'annlite_hnsw_add',
    'annlite_hnsw_search',
    'annlite_hnsw_mark_deleted',
    'annlite_hnsw_links_per_node',
    'annlite_hnsw_export',
    'annlite_hnsw_save',
    'annlite_hnsw_load',
)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
        L.annlite_hnsw_create.restype = vp
        L.annlite_hnsw_create.argtypes = [vp, i64, i64, i64, i64, i32, i32, u64]
        L.annlite_hnsw_free.argtypes = [vp]
        L.annlite_hnsw_free.restype = None
        L.annlite_hnsw_last_error.restype = ctypes.c_char_p
        L.annlite_hnsw_size.argtypes = [vp]
        L.annlite_hnsw_size.restype = i64
        L.annlite_hnsw_reserve.argtypes = [vp, i64]
        L.annlite_hnsw_add.argtypes = [vp, vp, vp, vp, i64, i32]
        L.annlite_hnsw_search.argtypes = [vp, vp, i64, i32, vp, vp, i32]
        L.annlite_hnsw_mark_deleted.argtypes = [vp, i64]
        L.annlite_hnsw_links_per_node.argtypes = [vp]
        L.annlite_hnsw_export.argtypes = [vp, i64, vp, vp, i64, vp]
        L.annlite_hnsw_save.argtypes = [vp, ctypes.c_char_p]
        L.annlite_hnsw_load.argtypes = [ctypes.c_char_p]
        L.annlite_hnsw_load.restype = vp
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what}: {lib().annlite_hnsw_last_error().decode()}')
