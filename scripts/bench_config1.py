#!/usr/bin/env python3
"""BASELINE config 1: the plumbing of the reference's examples/pq_benchmark.py at its CPU-runnable size
(1k docs x 128-dim float32, PQ m=8 ks=256, cosine, top-10), GPU path and CPU path side by side.

What is reproduced (reference file:line):
  * data            np.random.seed(123); make_blobs(n_samples, n_features=128); train_test_split(test_size=20)
                    -> 1000 index vectors, 20 queries                              examples/pq_benchmark.py:26-28
  * training set    Xtr[:20480]                                                    examples/pq_benchmark.py:50
  * documents       ids '0', '1', ... (strings), embeddings = the rows             examples/pq_benchmark.py:32-37, 54
  * ground truth    cdist(Xte, Xtr, metric) + top_k                                examples/pq_benchmark.py:57-58
  * query           pq.search(docs, limit=top_k); ids from doc.matches             examples/pq_benchmark.py:60-67
  * recall/precision  examples/utils.py:40-71 (restated below: `_precision` divides by len(predicted))

Two paths on the SAME codebooks:
  gpu   annlite_amd.AnnLite(128, metric='cosine', n_subvectors=8).train / index / search  (HIP kernels through the C ABI)
  cpu   the oracle's restatement of the reference CPU path (oracle/pq_oracle.py: l2_normalize, encode, get_dist_mat,
        flat ADC scan, top-k) -- what the reference computes on this host's cores, single thread
and the two must return the same neighbour ids, hence identical recall / precision.

    python scripts/bench_config1.py              # GPU box: both paths
    python scripts/bench_config1.py --cpu-only   # no GPU: CPU path only, codebooks trained like the reference does
                                                 # (sklearn KMeans per sub-space, pq.py:89-115)
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def _precision(predicted, relevant, eval_at):  # examples/utils.py:40-49
    if eval_at == 0:
        return 0.0
    return len(set(predicted[:eval_at]).intersection(set(relevant))) / len(predicted)


def _recall(predicted, relevant, eval_at):  # examples/utils.py:52-60
    if eval_at == 0:
        return 0.0
    return len(set(predicted[:eval_at]).intersection(set(relevant))) / len(relevant)


def evaluate(predicts, relevants, top_k):  # examples/utils.py:63-71
    recall = precision = 0
    for p, r in zip(predicts, relevants):
        p = np.array([int(x) for x in p])
        recall += _recall(p, r, top_k)
        precision += _precision(p, r, top_k)
    return recall / len(predicts), precision / len(predicts)


def cosine_cdist(x, y, eps=np.finfo(np.float32).eps):  # annlite/math.py:21-38
    return 1 - np.clip((np.dot(x, y.T) + eps) / (np.outer(np.linalg.norm(x, axis=1), np.linalg.norm(y, axis=1)) + eps), -1, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--docs', type=int, default=1000)
    ap.add_argument('--cpu-only', action='store_true')
    ap.add_argument('--repeat', type=int, default=5, help='timed repetitions of the query step (median)')
    args = ap.parse_args()
    from sklearn.datasets import make_blobs
    from sklearn.model_selection import train_test_split

    import pq_oracle

    D, M, Ks, top_k, n_test = 128, 8, 256, 10, 20
    np.random.seed(123)
    Xtr, Xte = train_test_split(make_blobs(n_samples=args.docs + n_test, n_features=D)[0].astype(np.float32), test_size=n_test)
    true_ids = np.argsort(cosine_cdist(Xte, Xtr), axis=1, kind='stable')[:, :top_k]
    rec = {'config': 'BASELINE config 1: %d docs x %d-dim, PQ m=%d ks=%d, cosine, top-%d, %d queries (examples/pq_benchmark.py plumbing)'
                     % (len(Xtr), D, M, Ks, top_k, len(Xte))}

    gpu = None
    if not args.cpu_only:
        import tempfile

        from annlite_amd import AnnLite
        from annlite_amd.index import Document, DocumentArray

        def get_documents(emb):
            return DocumentArray([Document(id=f'{i}', embedding=emb[i]) for i in range(len(emb))])

        pq = AnnLite(D, metric='cosine', n_subvectors=M, data_path=tempfile.mkdtemp(prefix='annlite_c1_'))
        pq._pq_codec.seed = 123
        t0 = time.time()
        pq.train(Xtr[:20480])
        train_time = time.time() - t0
        t0 = time.time()
        pq.index(get_documents(Xtr))
        index_time = time.time() - t0
        times = []
        for _ in range(args.repeat + 1):  # (first run: module load / allocation, dropped like executor/benchmark.py:61)
            docs = get_documents(Xte)
            t0 = time.time()
            pq.search(docs, limit=top_k)
            times.append(time.time() - t0)
        query_time = float(np.median(times[1:]))
        gpu_ids = [[m.id for m in d.matches] for d in docs]
        gpu_d = np.array([[m.scores['cosine'].value for m in d.matches] for d in docs], dtype=np.float32)
        recall, precision = evaluate(gpu_ids, true_ids, top_k)
        gpu = {'recall': recall, 'precision': precision, 'train_time': train_time, 'index_time': index_time,
               'query_time': query_time, 'query_qps': len(Xte) / query_time, 'index_qps': len(Xtr) / index_time}
        codebooks = pq._pq_codec.codebooks
        rec['gpu'] = gpu
    else:
        from sklearn.cluster import KMeans

        xt = pq_oracle.l2_normalize(Xtr[:20480])  # pq.py:100-101
        codebooks = np.zeros((M, Ks, D // M), dtype=np.float32)
        t0 = time.time()
        for m in range(M):  # pq.py:103-113
            km = KMeans(n_clusters=Ks, max_iter=100, n_init=4, random_state=m).fit(xt[:, m * (D // M):(m + 1) * (D // M)])
            codebooks[m] = km.cluster_centers_
        rec['cpu_train_time_sklearn'] = time.time() - t0

    # ---- CPU path (the reference's arithmetic, oracle restatement), same codebooks --------------------------------
    t0 = time.time()
    codes = pq_oracle.encode_c(pq_oracle.l2_normalize(Xtr), codebooks)  # hnsw/index.py:28-29 + pq.py:158-177
    cpu_index_time = time.time() - t0
    times = []
    for _ in range(args.repeat + 1):
        t0 = time.time()
        cd, ci = pq_oracle.index_search(Xte, codebooks, codes, pq_oracle.COSINE, top_k, threads=1)
        times.append(time.time() - t0)
    cpu_query = float(np.median(times[1:]))
    cpu_ids = [[str(j) for j in row] for row in ci]
    recall, precision = evaluate(cpu_ids, true_ids, top_k)
    rec['cpu'] = {'recall': recall, 'precision': precision, 'index_time': cpu_index_time, 'query_time': cpu_query,
                  'query_qps': len(Xte) / cpu_query, 'cores': 1, 'kind': 'port',
                  'note': 'oracle restatement of the reference CPU path: normalise, encode, get_dist_mat, flat ADC scan, top-k'}
    if gpu is not None:
        same_ids = [list(a) == list(b) for a, b in zip(gpu_ids, cpu_ids)]
        rec['gpu_equals_cpu'] = {'ids_identical': bool(all(same_ids)), 'queries': len(same_ids),
                                 'distances_identical': bool(np.array_equal(gpu_d, cd.astype(np.float32))),
                                 'recall_identical': gpu['recall'] == recall, 'precision_identical': gpu['precision'] == precision}
    print(json.dumps(rec))
    return rec


if __name__ == '__main__':
    main()
