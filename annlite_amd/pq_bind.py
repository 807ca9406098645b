"""Drop-in for the reference's native operator module ``annlite.pq_bind``
(bindings/pq_bindings.pyx) -- same four function names, argument order and meaning, computed by
the HIP kernels of ``libannlite_hip.so``.

numpy in -> numpy out, so the reference's call sites (annlite/core/codec/pq.py:220-222, 313-320,
361) and its tests (tests/test_pq_bind.py, tests/test_pq_index.py:30-49) run unchanged; torch
device tensors in -> torch device tensors out (no host round trip).

Differences kept deliberately small:
  * ``dist_pqcodes_to_codebooks`` returns a float32 array instead of a python list of floats
    (the reference's ``vector[float]`` -> list conversion, pyx:75-80, is pure overhead).
"""
import numpy as np
import torch

from . import ops
from ._capi import LAYOUT_BMK, LUT_IP, LUT_L2


def _dev(a, dtype=torch.float32):
    return ops.to_dev(a, dtype), isinstance(a, np.ndarray) or not isinstance(a, torch.Tensor)


def _codebooks(codebooks, d_subvector, n_clusters):
    cb, _ = _dev(codebooks)
    assert cb.ndim == 3 and cb.shape[1] == n_clusters and cb.shape[2] == d_subvector
    return cb


def precompute_adc_table(query, d_subvector, n_clusters, codebooks):
    """bindings/pq_bindings.pyx:85-145 -- single-query squared-L2 table [M, Ks]."""
    q, is_np = _dev(np.asarray(query) if not isinstance(query, torch.Tensor) else query)
    assert q.ndim == 1
    out = ops.lut_build(q[None, :], _codebooks(codebooks, d_subvector, n_clusters), LUT_L2, LAYOUT_BMK)[0]
    return out.cpu().numpy() if is_np else out


def batch_precompute_adc_table(queries, d_subvector, n_clusters, codebooks):
    """bindings/pq_bindings.pyx:149-210 -- batched squared-L2 tables [B, M, Ks]."""
    q, is_np = _dev(queries)
    assert q.ndim == 2
    out = ops.lut_build(q, _codebooks(codebooks, d_subvector, n_clusters), LUT_L2, LAYOUT_BMK)
    return out.cpu().numpy() if is_np else out


def batch_precompute_adc_table_ip(queries, d_subvector, n_clusters, codebooks):
    """bindings/pq_bindings.pyx:214-274 -- batched inner-product tables [B, M, Ks] (MFMA)."""
    q, is_np = _dev(queries)
    assert q.ndim == 2
    out = ops.lut_build(q, _codebooks(codebooks, d_subvector, n_clusters), LUT_IP, LAYOUT_BMK)
    return out.cpu().numpy() if is_np else out


def dist_pqcodes_to_codebooks(adtable, pq_codes):
    """bindings/pq_bindings.pyx:52-80 -- d[n] = sum_m adtable[m, pq_codes[n, m]], fp32, m ascending."""
    t, is_np = _dev(adtable)
    c = ops.to_dev(pq_codes)
    assert t.ndim == 2 and c.ndim == 2
    out = ops.adc_dist(t, c)
    return out.cpu().numpy() if is_np else out
