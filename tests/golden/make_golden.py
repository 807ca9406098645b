#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz by RUNNING THE REAL REFERENCE.

Run in the build container only (needs /root/reference and oracle/_ref, see oracle/build_ref.sh):

    ./oracle/build_ref.sh && python tests/golden/make_golden.py

The reference's own tests hold no golden vectors for this path (all data is
unseeded ``np.random.random``), so the fixtures are produced here by importing the reference
(oracle/ref_import.py) on seeded inputs.  Fixtures are DATA only: seeded inputs + the reference's
outputs.  Codebooks are drawn directly (sub-vectors of seeded data), not fitted, so everything is
deterministic (sklearn KMeans in the reference is unseeded, pq.py:108-110).

Each case stores, for a (M, dsub, Ks) shape of SURVEY.md section 8c:
  inputs   codebooks[M,Ks,dsub]  queries[B,D]  x[N,D]
  pq_bind  lut_l2_batch  lut_ip_batch  lut_l2_single(query 0)         (pq_bindings.pyx:85-274)
  PQCodec  dist_mat_{euclidean,inner_product,cosine}  codes  codes_cos  decoded (pq.py:158-325)
  adist    DistanceTable.adist for every query over `codes`           (pq.py:350-368, pyx:52-80)
  PQIndex  pqindex_d / pqindex_i  (limit=10, scans all capacity rows) (pq_index.py:29-56)
  Hnsw     hnsw_{metric}_d / _i   HnswIndex(pq_codec).search          (hnsw/index.py:139-167)
  math     l2norm_q, topk_d (math.top_k values of adist[0])           (math.py:6-18, 94-120)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import ref_import  # noqa: E402

CASES = {
    # name: (M, dsub, Ks, N, B, seed)
    'c1_m8_d128': (8, 16, 256, 1000, 6, 101),      # BASELINE config 1 shape (1k docs, M=8)
    'c2_m16_d128': (16, 8, 256, 1024, 6, 102),     # configs 2/3/5 shape
    'c4_m64_d768': (64, 12, 256, 192, 3, 103),     # config 4 shape
    't_m32_d128': (32, 4, 256, 1024, 5, 104),      # tests/test_pq_bind.py shape
    'ks512_m8_d64': (8, 8, 512, 1000, 5, 105),     # tests/test_pq_index.py:80 (uint16 codes)
    'ks768_m8_d64': (8, 8, 768, 1000, 5, 106),     # tests/test_pq_index.py:80, the third n_clusters (round 3: 16-query byte tables)
    'ex_m64_d128': (64, 2, 256, 512, 4, 107),      # examples/pq_benchmark.py:44 `for n_subvectors in [64, 128]` at D = 128: dsub = 2 ...
    'ex_m128_d128': (128, 1, 256, 512, 4, 108),    # ... and dsub = 1 (round 5)
}
K = 10


def make_case(ref, name, M, dsub, Ks, N, B, seed):
    rs = np.random.RandomState(seed)
    D = M * dsub
    # structured-ish data: a few gaussian blobs + uniform noise, float32
    centers = rs.randn(32, D).astype(np.float32)
    x = (centers[rs.randint(0, 32, size=N)] + 0.35 * rs.randn(N, D)).astype(np.float32)
    queries = (centers[rs.randint(0, 32, size=B)] + 0.35 * rs.randn(B, D)).astype(np.float32)
    pool = (centers[rs.randint(0, 32, size=Ks * 4)] + 0.35 * rs.randn(Ks * 4, D)).astype(np.float32)
    codebooks = np.empty((M, Ks, dsub), dtype=np.float32)
    for m in range(M):
        pick = rs.choice(pool.shape[0], size=Ks, replace=False)
        codebooks[m] = pool[pick, m * dsub:(m + 1) * dsub]
    # normalised-data codebooks for the cosine index
    pool_n = ref.math.l2_normalize(pool).astype(np.float32)
    codebooks_cos = np.empty((M, Ks, dsub), dtype=np.float32)
    for m in range(M):
        pick = rs.choice(pool.shape[0], size=Ks, replace=False)
        codebooks_cos[m] = pool_n[pick, m * dsub:(m + 1) * dsub]

    out = dict(codebooks=codebooks, codebooks_cos=codebooks_cos, queries=queries, x=x,
               meta=np.array([M, dsub, Ks, N, B, seed, K], dtype=np.int64))

    pb = ref.pq_bind
    out['lut_l2_batch'] = np.asarray(pb.batch_precompute_adc_table(queries, dsub, Ks, codebooks), dtype=np.float32)
    out['lut_ip_batch'] = np.asarray(pb.batch_precompute_adc_table_ip(queries, dsub, Ks, codebooks), dtype=np.float32)
    out['lut_l2_single'] = np.asarray(pb.precompute_adc_table(queries[0], dsub, Ks, codebooks), dtype=np.float32)

    codecs = {}
    for mname, metric, cb in (('euclidean', ref.Metric.EUCLIDEAN, codebooks),
                              ('inner_product', ref.Metric.INNER_PRODUCT, codebooks),
                              ('cosine', ref.Metric.COSINE, codebooks_cos)):
        c = ref.PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=metric)
        c._codebooks = cb.copy()
        c._is_trained = True
        codecs[mname] = c
        dm = c.get_dist_mat(queries)
        assert dm.dtype == np.float32 and dm.flags['C_CONTIGUOUS']
        out['dist_mat_' + mname] = dm

    ce = codecs['euclidean']
    codes = ce.encode(x)
    out['codes'] = codes
    xn = ref.math.l2_normalize(x).astype(np.float32)
    out['codes_cos'] = codecs['cosine'].encode(xn)
    out['decoded'] = ce.decode(codes)[:64]
    out['l2norm_q'] = ref.math.l2_normalize(queries)

    adist = np.empty((B, N), dtype=np.float32)
    for b in range(B):
        dt = ref.DistanceTable(out['lut_l2_batch'][b])
        adist[b] = np.asarray(dt.adist(codes), dtype=np.float32)
    out['adist'] = adist
    out['topk_d'] = ref.math.top_k(adist[:1].astype(np.float64), K)[0][0]

    # PQIndex (deprecated linear scan) -- scans ALL capacity rows incl. never-written zero rows
    cap = N + 24
    pqi = ref.PQIndex(D, ce, initial_size=cap)
    pqi.add_with_ids(x, np.arange(N))
    pd_, pi_ = [], []
    for b in range(B):
        d, i = pqi.search(queries[b], limit=K)
        pd_.append(np.asarray(d, dtype=np.float64))
        pi_.append(np.asarray(i, dtype=np.int64))
    out['pqindex_d'] = np.stack(pd_)
    out['pqindex_i'] = np.stack(pi_)
    out['pqindex_capacity'] = np.array([cap], dtype=np.int64)

    # HnswIndex over PQ: same ADC arithmetic through space_pq.h; ids depend on the graph walk, the
    # distance of every returned id must equal the flat ADC distance of that row bit-for-bit.
    if ref.HnswIndex is not None:
        for mname, metric in (('euclidean', ref.Metric.EUCLIDEAN),
                              ('inner_product', ref.Metric.INNER_PRODUCT),
                              ('cosine', ref.Metric.COSINE)):
            h = ref.HnswIndex(dim=D, metric=metric, pq_codec=codecs[mname], initial_size=N,
                              ef_search=128, ef_construction=200)
            h.add_with_ids(x, np.arange(N))
            hd, hi = [], []
            for b in range(B):
                d, i = h.search(queries[b], limit=K)
                hd.append(np.asarray(d, dtype=np.float32))
                hi.append(np.asarray(i, dtype=np.int64))
            out['hnsw_%s_d' % mname] = np.stack(hd)
            out['hnsw_%s_i' % mname] = np.stack(hi)

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-16s -> %s  (%.1f KB)' % (name, os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


def make_cells_case(ref, name='cells_m16_d64', M=16, dsub=4, Ks=256, N=4000, B=8, C=12, P=4, seed=201):
    """n_cells > 1 structure (tests/golden/cells/*.npz): VQCodec.encode (vq.py:78-90), cdist + top_k of
    _cell_selection (index.py:462-465), one reference PQIndex per cell and the concatenate + argsort merge of
    CellContainer.ivf_search (container.py:101-138; its python loop is restated here, the early-skip of lines
    120-121 left out -- it is a shortcut, not a result)."""
    rs = np.random.RandomState(seed)
    D = M * dsub
    centroids = rs.randn(C, D).astype(np.float32)
    x = (centroids[rs.randint(0, C, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
    queries = (centroids[rs.randint(0, C, B)] + 0.7 * rs.randn(B, D)).astype(np.float32)
    codebooks = np.empty((M, Ks, dsub), dtype=np.float32)
    for m in range(M):
        codebooks[m] = x[rs.choice(N, size=Ks, replace=False), m * dsub:(m + 1) * dsub]
    vq = ref.VQCodec(C, metric=ref.Metric.EUCLIDEAN)
    vq._codebook = centroids
    vq._is_trained = True
    cells_of = np.asarray(vq.encode(x)).astype(np.int32)
    out = dict(centroids=centroids, x=x, queries=queries, codebooks=codebooks, cells_of=cells_of,
               meta=np.array([M, dsub, Ks, N, B, C, P, K, seed], dtype=np.int64))
    for metric in ('euclidean', 'cosine'):
        dists = ref.math.cdist(queries, centroids, metric=metric)
        out['cdist_' + metric] = np.asarray(dists, dtype=np.float64)
        out['probe_' + metric] = np.asarray(ref.math.top_k(dists, k=P)[1], dtype=np.int32)
    codec = ref.PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=ref.Metric.EUCLIDEAN)
    codec._codebooks = codebooks
    codec._is_trained = True
    out['codes'] = codec.encode(x)
    per_cell, rows_of = {}, {}
    for c in range(C):
        rows = np.nonzero(cells_of == c)[0]
        rows_of[c] = rows
        per_cell[c] = ref.PQIndex(D, codec, initial_size=max(len(rows), 1))
        if len(rows):
            per_cell[c].add_with_ids(x[rows], np.arange(len(rows)))
    md, mi = np.full((B, K), np.inf, np.float32), np.full((B, K), -1, np.int64)
    for b in range(B):
        ds, ids = [], []
        for c in out['probe_euclidean'][b]:
            if len(rows_of[c]) == 0:
                continue
            dd, ii = per_cell[c].search(queries[b], limit=min(K, len(rows_of[c])))
            ds.append(np.asarray(dd, dtype=np.float32))
            ids.append(rows_of[c][np.asarray(ii)])
        ds, ids = np.concatenate(ds), np.concatenate(ids)
        order = ds.argsort(axis=0)[:K]
        md[b, :len(order)], mi[b, :len(order)] = ds[order], ids[order]
    out['merged_d'], out['merged_i'] = md, mi  # raw ADC sums (PQIndex returns no sqrt)
    os.makedirs(os.path.join(HERE, 'cells'), exist_ok=True)
    path = os.path.join(HERE, 'cells', name + '.npz')
    np.savez_compressed(path, **out)
    print('%-16s -> %s  (%.1f KB)' % (name, os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


def main():
    ref = ref_import.load()
    only = set(sys.argv[1:])  # (no arguments: every fixture; names: only those -- e.g. a case added later)
    for name, spec in CASES.items():
        if not only or name in only:
            make_case(ref, name, *spec)
    if not only or 'cells' in only:
        make_cells_case(ref)


if __name__ == '__main__':
    main()
