"""annlite_amd -- MI355X (gfx950) implementation of jina-ai/annlite's PQ codec + ADC scan hot path.

Public surface (mirrors the reference's names, see DESIGN.md / INTEGRATION.md):

    from annlite_amd import AnnLite, PQCodec, PQFlatGpuIndex, HnswPQGpuIndex, Metric, pq_bind

The compute path is the hand-written HIP library ``annlite_amd/libannlite_hip.so`` (C ABI in
``include/annlite_hip.h``), loaded through ctypes; there is no CPU fallback.
"""
__version__ = '0.1.0'

from .enums import ExpandMode, Metric  # noqa: F401
from . import _capi  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import annlite_amd` stays cheap
    if name == 'AnnLite':
        from .index import AnnLite
        return AnnLite
    if name in ('PQCodec', 'DistanceTable'):
        from .core.codec import pq
        return getattr(pq, name)
    if name == 'PQFlatGpuIndex':
        from .core.index.pq_flat_gpu import PQFlatGpuIndex
        return PQFlatGpuIndex
    if name == 'HnswPQGpuIndex':
        from .core.index.hnsw_pq_gpu import HnswPQGpuIndex
        return HnswPQGpuIndex
    if name in ('pq_bind', 'ops', 'math', 'sharded'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
