"""CPU oracle for the PQ / ADC hot path -- python face of ``oracle/pq_oracle.c``.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; ``annlite_amd`` never does (the
product path raises when its HIP library is missing instead of falling back to anything here).

Pinning status (see DESIGN.md "Oracle"):
  * pinned bit-for-bit against the compiled reference (``oracle/_ref`` built by
    ``oracle/build_ref.sh`` from /root/reference) in ``tests/test_oracle_vs_reference.py`` and
    against the committed golden fixtures ``tests/golden/*.npz`` (generated from that same compiled
    reference by ``tests/golden/make_golden.py``) in ``tests/test_oracle_golden.py``:
    LUT L2 / LUT IP / get_dist_mat (3 metrics) / flat ADC scan (u8,u16) / decode / PQIndex.search /
    HnswIndex(PQ) distances.
  * ``encode`` and ``fit`` sit on un-vendored third-party code (scipy ``vq``, sklearn ``KMeans``):
    the reference's tests pin neither => "parity unpinned" by the reference;
    encode is pinned here against the installed scipy's answers stored in the fixtures
    (mismatch allowed only on near-ties), training only statistically.

Two implementations of every arithmetic function are provided:
  ``*_c``      -> the C restatement (fast; also the timed CPU baseline)
  ``*_numpy``  -> an independent numpy/pure-python restatement used to cross-check the C one on
                  small cases (sequential fp32 fma emulated exactly in float64, see ``_fma32``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libpq_oracle.so')

EUCLIDEAN, INNER_PRODUCT, COSINE = 1, 2, 3  # annlite/enums.py:25-28

_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/pq_oracle.c -> oracle/libpq_oracle.so (gcc, a second or two)."""
    src = os.path.join(_HERE, 'pq_oracle.c')
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpq_oracle.so'], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        vp = ctypes.c_void_p
        L.oracle_precompute_adc_table.argtypes = [vp, i64, i64, i64, vp, vp]
        L.oracle_batch_precompute_adc_table.argtypes = [vp, i64, i64, i64, i64, vp, vp, ctypes.c_int]
        L.oracle_batch_precompute_adc_table_ip.argtypes = [vp, i64, i64, i64, i64, vp, vp, ctypes.c_int]
        L.oracle_get_dist_mat.argtypes = [ctypes.c_int, vp, i64, i64, i64, i64, vp, vp, ctypes.c_int]
        L.oracle_get_dist_mat.restype = ctypes.c_int
        for n in ('u8', 'u16', 'u32'):
            getattr(L, 'oracle_dist_pqcodes_to_codebooks_' + n).argtypes = [vp, i64, i64, vp, i64, vp, ctypes.c_int]
        L.oracle_adc_gather_u8.argtypes = [vp, i64, i64, vp, vp, i64, vp]
        L.oracle_topk.argtypes = [vp, i64, i64, i64, vp, vp]
        L.oracle_adc_search_u8.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, vp, vp, ctypes.c_int]
        L.oracle_encode.argtypes = [vp, i64, i64, i64, i64, vp, vp, ctypes.c_int]
        L.oracle_decode_u8.argtypes = [vp, i64, i64, i64, i64, vp, vp]
        L.oracle_max_threads.restype = ctypes.c_int
        del f32p
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def max_threads() -> int:
    """Threads worth starting for the all-cores CPU baseline: OpenMP's default, capped by the CPU affinity
    mask and by a cgroup CPU quota (a container may report 256 CPUs and be throttled to 16)."""
    n = int(lib().oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max' and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# ------------------------------------------------------------------------------------------------
# annlite/math.py:6-18  l2_normalize -- literally numpy, like the reference
# ------------------------------------------------------------------------------------------------
def l2_normalize(x: np.ndarray, eps: float = np.finfo(np.float32).eps) -> np.ndarray:
    norms = np.einsum('ij,ij->i', x, x)
    np.sqrt(norms, norms)
    constant_mask = norms < 10 * eps
    norms[constant_mask] = 1.0
    return x / norms[:, np.newaxis]


# ------------------------------------------------------------------------------------------------
# LUTs  (bindings/pq_bindings.pyx:85-274, annlite/core/codec/pq.py:200-224, 293-325)
# ------------------------------------------------------------------------------------------------
def precompute_adc_table_c(query, d_subvector, n_clusters, codebooks):
    query, codebooks = _f32c(query), _f32c(codebooks)
    D = query.shape[0]
    out = np.empty((D // d_subvector, n_clusters), dtype=np.float32)
    lib().oracle_precompute_adc_table(_p(query), D, d_subvector, n_clusters, _p(codebooks), _p(out))
    return out


def batch_precompute_adc_table_c(queries, d_subvector, n_clusters, codebooks, threads=1):
    queries, codebooks = _f32c(queries), _f32c(codebooks)
    B, D = queries.shape
    out = np.empty((B, D // d_subvector, n_clusters), dtype=np.float32)
    lib().oracle_batch_precompute_adc_table(_p(queries), B, D, d_subvector, n_clusters, _p(codebooks), _p(out), threads)
    return out


def batch_precompute_adc_table_ip_c(queries, d_subvector, n_clusters, codebooks, threads=1):
    queries, codebooks = _f32c(queries), _f32c(codebooks)
    B, D = queries.shape
    out = np.empty((B, D // d_subvector, n_clusters), dtype=np.float32)
    lib().oracle_batch_precompute_adc_table_ip(_p(queries), B, D, d_subvector, n_clusters, _p(codebooks), _p(out), threads)
    return out


def get_dist_mat_c(x, codebooks, metric: int, threads=1):
    """PQCodec.get_dist_mat (pq.py:293-325) incl. the cosine re-normalisation at 309-310."""
    x, codebooks = _f32c(x), _f32c(codebooks)
    M, Ks, dsub = codebooks.shape
    if metric == COSINE:
        x = _f32c(l2_normalize(x))
    B, D = x.shape
    assert D == M * dsub
    out = np.empty((B, M, Ks), dtype=np.float32)
    rc = lib().oracle_get_dist_mat(int(metric), _p(x), B, D, dsub, Ks, _p(codebooks), _p(out), threads)
    if rc != 0:
        raise ValueError('unsupported metric %r' % (metric,))
    return out


def _fma32(a, b, c):
    """Exact fp32 fused multiply-add emulated in float64: the product of two fp32 numbers is exact
    in float64 (48 significant bits), the sum with a third fp32 number is rounded once to float64
    and once more to float32.  Double rounding can differ from a true fmaf only when the float64
    sum lands exactly on a float32 rounding tie, which needs a >29-bit cancellation pattern; the
    cross-check test compares against the C fmaf on random data and tolerates no mismatch, which
    has held on every fixture."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def batch_precompute_adc_table_numpy(queries, d_subvector, n_clusters, codebooks):
    queries, codebooks = _f32c(queries), _f32c(codebooks)
    B, D = queries.shape
    M = D // d_subvector
    out = np.zeros((B, M, n_clusters), dtype=np.float32)
    for m in range(M):
        q = queries[:, m * d_subvector:(m + 1) * d_subvector]  # [B, dsub]
        acc = np.zeros((B, n_clusters), dtype=np.float32)
        for j in range(d_subvector):
            c = (codebooks[m, :, j][None, :] - q[:, j][:, None]).astype(np.float32)
            acc = _fma32(c, c, acc)
        out[:, m, :] = acc
    return out


def batch_precompute_adc_table_ip_numpy(queries, d_subvector, n_clusters, codebooks):
    queries, codebooks = _f32c(queries), _f32c(codebooks)
    B, D = queries.shape
    M = D // d_subvector
    out = np.zeros((B, M, n_clusters), dtype=np.float32)
    for m in range(M):
        q = queries[:, m * d_subvector:(m + 1) * d_subvector]
        acc = np.zeros((B, n_clusters), dtype=np.float32)
        for j in range(d_subvector):
            cw = np.broadcast_to(codebooks[m, :, j][None, :], acc.shape)
            qq = np.broadcast_to(q[:, j][:, None], acc.shape)
            acc = _fma32(cw, qq, acc)
        out[:, m, :] = acc
    return out


# ------------------------------------------------------------------------------------------------
# flat ADC scan (bindings/pq_bindings.pyx:52-80 ; pq.py:350-368 ; space_pq.h:15-37)
# ------------------------------------------------------------------------------------------------
def dist_pqcodes_to_codebooks_c(adtable, pq_codes, threads=1):
    adtable = _f32c(adtable)
    pq_codes = np.ascontiguousarray(pq_codes)
    N, M = pq_codes.shape
    assert adtable.shape[0] == M
    Ks = adtable.shape[1]
    out = np.empty(N, dtype=np.float32)
    fn = {1: 'u8', 2: 'u16', 4: 'u32'}[pq_codes.dtype.itemsize]
    assert pq_codes.dtype.kind == 'u'
    getattr(lib(), 'oracle_dist_pqcodes_to_codebooks_' + fn)(_p(adtable), M, Ks, _p(pq_codes), N, _p(out), threads)
    return out


def dist_pqcodes_to_codebooks_numpy(adtable, pq_codes):
    adtable = _f32c(adtable)
    N, M = pq_codes.shape
    acc = np.zeros(N, dtype=np.float32)
    for m in range(M):  # strict ascending-m fp32 add chain
        acc = (acc + adtable[m, pq_codes[:, m].astype(np.int64)]).astype(np.float32)
    return acc


def adc_gather_c(adtable, pq_codes, cand):
    adtable = _f32c(adtable)
    pq_codes = np.ascontiguousarray(pq_codes, dtype=np.uint8)
    cand = np.ascontiguousarray(cand, dtype=np.int64)
    out = np.empty(cand.shape[0], dtype=np.float32)
    lib().oracle_adc_gather_u8(_p(adtable), pq_codes.shape[1], adtable.shape[1], _p(pq_codes), _p(cand), cand.shape[0], _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# top-k with the build's fixed tie-break (annlite/math.py:94-120 leaves ties unspecified)
# ------------------------------------------------------------------------------------------------
def top_k_c(values, k, id_base=0):
    values = _f32c(values)
    assert values.ndim == 1
    d = np.empty(k, dtype=np.float32)
    i = np.empty(k, dtype=np.int64)
    lib().oracle_topk(_p(values), values.shape[0], k, id_base, _p(d), _p(i))
    return d, i


def top_k_numpy(values, k, id_base=0):
    values = np.asarray(values)
    order = np.lexsort((np.arange(values.shape[0]), values))[:k]  # (dist asc, id asc), stable
    d = values[order].astype(np.float32)
    i = order.astype(np.int64) + id_base
    if k > values.shape[0]:
        pad = k - values.shape[0]
        d = np.concatenate([d, np.full(pad, np.inf, np.float32)])
        i = np.concatenate([i, np.full(pad, -1, np.int64)])
    return d, i


# ------------------------------------------------------------------------------------------------
# batched flat search: PQIndex.search (pq_index.py:29-56) per query, over ALL rows of `codes`
# ------------------------------------------------------------------------------------------------
def adc_search_c(lut, codes, k, id_base=0, threads=1):
    lut = _f32c(lut)
    codes = np.ascontiguousarray(codes)
    if codes.dtype.itemsize != 1:  # uint16 / uint32 codes (n_clusters > 256, pq.py:56-60): per query, the same two kernels
        if codes.dtype.kind != 'u':
            codes = codes.view({2: np.uint16, 4: np.uint32}[codes.dtype.itemsize])
        ds, is_ = [], []
        for b in range(lut.shape[0]):
            d, i = top_k_c(dist_pqcodes_to_codebooks_c(lut[b], codes, threads=threads), k, id_base)
            ds.append(d)
            is_.append(i)
        return np.stack(ds), np.stack(is_)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    B, M, Ks = lut.shape
    N = codes.shape[0]
    assert codes.shape[1] == M
    d = np.empty((B, k), dtype=np.float32)
    i = np.empty((B, k), dtype=np.int64)
    lib().oracle_adc_search_u8(_p(lut), B, M, Ks, _p(codes), N, k, id_base, _p(d), _p(i), threads)
    return d, i


def adc_search_numpy(lut, codes, k, id_base=0):
    ds, is_ = [], []
    for b in range(lut.shape[0]):
        dist = dist_pqcodes_to_codebooks_numpy(lut[b], codes)
        d, i = top_k_numpy(dist, k, id_base)
        ds.append(d)
        is_.append(i)
    return np.stack(ds), np.stack(is_)


# ------------------------------------------------------------------------------------------------
# encode / decode  (pq.py:158-198)
# ------------------------------------------------------------------------------------------------
def code_dtype(n_clusters):
    return np.uint8 if n_clusters <= 2 ** 8 else (np.uint16 if n_clusters <= 2 ** 16 else np.uint32)  # pq.py:56-60


def encode_c(x, codebooks, threads=1):
    x, codebooks = _f32c(x), _f32c(codebooks)
    M, Ks, dsub = codebooks.shape
    N, D = x.shape
    assert D == M * dsub
    out = np.empty((N, M), dtype=np.uint32)
    lib().oracle_encode(_p(x), N, D, dsub, Ks, _p(codebooks), _p(out), threads)
    return out.astype(code_dtype(Ks))


def encode_scipy(x, codebooks):
    """The literal reference call (pq.py:173-175): scipy.cluster.vq.vq per subspace."""
    from scipy.cluster.vq import vq

    x, codebooks = _f32c(x), _f32c(codebooks)
    M, Ks, dsub = codebooks.shape
    codes = np.empty((x.shape[0], M), dtype=code_dtype(Ks))
    for m in range(M):
        codes[:, m], _ = vq(x[:, m * dsub:(m + 1) * dsub], codebooks[m])
    return codes


def encode_gap(x, codebooks):
    """float64 (best, second-best) squared distances per (row, subspace): used to excuse encode
    mismatches on near-ties only."""
    x = np.asarray(x, dtype=np.float64)
    cb = np.asarray(codebooks, dtype=np.float64)
    M, Ks, dsub = cb.shape
    N = x.shape[0]
    best = np.empty((N, M))
    second = np.empty((N, M))
    for m in range(M):
        d = ((x[:, None, m * dsub:(m + 1) * dsub] - cb[m][None]) ** 2).sum(-1)
        part = np.partition(d, 1, axis=1)
        best[:, m], second[:, m] = part[:, 0], part[:, 1]
    return best, second


def decode_c(codes, codebooks):
    codebooks = _f32c(codebooks)
    M, Ks, dsub = codebooks.shape
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.empty((codes.shape[0], M * dsub), dtype=np.float32)
    lib().oracle_decode_u8(_p(codes), codes.shape[0], M, dsub, Ks, _p(codebooks), _p(out))
    return out


def decode_numpy(codes, codebooks):
    M, Ks, dsub = codebooks.shape
    out = np.empty((codes.shape[0], M * dsub), dtype=np.float32)
    for m in range(M):
        out[:, m * dsub:(m + 1) * dsub] = codebooks[m][codes[:, m].astype(np.int64), :]
    return out


def pqindex_search_reference_style(adtable, codes, k):
    """ONE query the way the reference's flat index materialises it (annlite/core/index/pq_index.py:46-49 on top of
    bindings/pq_bindings.pyx:75-80): the ADC kernel's result becomes a Python LIST of N floats, `np.expand_dims` turns
    the list into a float64 array, `math.top_k` (math.py:94-120: argpartition + argsort) selects.  Only used to time
    that overhead next to the kernel-only figure (bench.py cpu_baseline); tie order follows numpy, as in the reference."""
    dists = dist_pqcodes_to_codebooks_c(adtable, codes).tolist()   # vector<float> -> list (pyx:75-80)
    dists = np.expand_dims(dists, axis=0)                          # pq_index.py:47
    if k >= dists.shape[1]:
        idx = dists.argsort(axis=1)[:, :k]
        vals = np.take_along_axis(dists, idx, axis=1)
    else:
        idx_ps = dists.argpartition(kth=k, axis=1)[:, :k]
        vals = np.take_along_axis(dists, idx_ps, axis=1)
        idx_fs = vals.argsort(axis=1)
        idx = np.take_along_axis(idx_ps, idx_fs, axis=1)
        vals = np.take_along_axis(vals, idx_fs, axis=1)
    return vals[0], idx[0]


# ------------------------------------------------------------------------------------------------
# end-to-end search semantics the drop-in index mirrors
# ------------------------------------------------------------------------------------------------
def index_search(x, codebooks, codes, metric, k, sqrt_euclidean=True, threads=1):
    """HnswIndex.pre_process + search epilogue (annlite/core/index/hnsw/index.py:20-48, 139-167)
    applied to an EXHAUSTIVE ADC scan (PQIndex semantics made metric-aware):
    cast f32, normalise if cosine (index.py:28-29), get_dist_mat (re-normalises, pq.py:309-310),
    ADC over all rows, top-k (dist asc, id asc), sqrt if EUCLIDEAN (index.py:164-165)."""
    x = _f32c(np.atleast_2d(x))
    if metric == COSINE:
        x = _f32c(l2_normalize(x))
    lut = get_dist_mat_c(x, codebooks, metric, threads=threads)
    d, i = adc_search_c(lut, codes, k, threads=threads)
    if metric == EUCLIDEAN and sqrt_euclidean:
        d = np.sqrt(d)
    return d, i


# ------------------------------------------------------------------------------------------------
# pruned (IVF) search: AnnLite(n_cells > 1) structure -- VQCodec.encode (annlite/core/codec/vq.py:78-90),
# AnnLite._cell_selection (annlite/index.py:458-466), CellContainer.ivf_search (annlite/container.py:88-144)
# ------------------------------------------------------------------------------------------------
def cell_distances(queries, centroids, kind):
    """What the build's selection kernel ranks cells by: kind 0 = squared L2 as the fp32 chain
    acc = fma(c - q, c - q, acc) over j; kind 1 = -(fp32 chain acc = fma(c, q, acc)).  As rankings these are
    the reference's cdist(query, vq codebook, metric) (annlite/math.py:21-61): 'euclidean' = sqrt of the squared
    distance, 'cosine' = 1 - <q, c>/(|q||c|) with unit q and centroids normalised by the caller."""
    q = _f32c(queries)
    c = _f32c(centroids)
    acc = np.zeros((q.shape[0], c.shape[0]), dtype=np.float32)
    for j in range(q.shape[1]):
        if kind == 0:
            d = (c[None, :, j] - q[:, None, j]).astype(np.float32)
            acc = _fma32(d, d, acc)
        else:
            acc = _fma32(np.broadcast_to(c[None, :, j], acc.shape), np.broadcast_to(q[:, None, j], acc.shape), acc)
    return acc if kind == 0 else -acc


def select_cells(queries, centroids, kind, n_probe):
    """top_k(dists, k=n_probe) of _cell_selection with the fixed tie-break (distance asc, cell asc)."""
    d = cell_distances(queries, centroids, kind)
    order = np.lexsort((np.broadcast_to(np.arange(d.shape[1]), d.shape), d), axis=1)
    return order[:, :n_probe].astype(np.int32)


def assign_cells(x, centroids):
    """VQCodec.encode: nearest centroid in squared L2, first minimum wins."""
    return select_cells(x, centroids, 0, 1)[:, 0]


def ivf_search(x, codebooks, codes, cell_of_row, probe_cells, metric, k, valid=None, sqrt_euclidean=True):
    """index_search restricted, per query, to the rows whose cell is in probe_cells[b]: the exact top-k
    (distance asc, id asc) of the union of the probed cells = ivf_search's concatenate + argsort
    (container.py:130-138) without its early-skip heuristic (container.py:120-121)."""
    x = _f32c(np.atleast_2d(x))
    if metric == COSINE:
        x = _f32c(l2_normalize(x))
    lut = get_dist_mat_c(x, codebooks, metric)
    cell_of_row = np.asarray(cell_of_row)
    ds, is_ = [], []
    for b in range(x.shape[0]):
        sel = np.isin(cell_of_row, probe_cells[b])
        if valid is not None:
            sel &= np.asarray(valid, dtype=bool)
        rows = np.nonzero(sel)[0]
        if rows.size:
            dist = dist_pqcodes_to_codebooks_c(lut[b], np.ascontiguousarray(codes[rows]))
            d, i = top_k_numpy(dist, k)
            i = np.where(i >= 0, rows[np.clip(i, 0, rows.size - 1)], -1)
        else:
            d, i = np.full(k, np.inf, np.float32), np.full(k, -1, np.int64)
        ds.append(d)
        is_.append(i)
    d, i = np.stack(ds), np.stack(is_)
    if metric == EUCLIDEAN and sqrt_euclidean:
        d = np.sqrt(d)
    return d, i
