/* annlite_graph.h -- C ABI of libannlite_graph.so: the HNSW-over-PQ candidate generator of BASELINE config 5.
 *
 * Host (CPU, OpenMP) code: a from-scratch HNSW graph over PQ codes whose edge walks use the asymmetric PQ
 * distance of the reference's hnswlib::PQLookup (include/hnswlib/space_pq.h:15-37): the look-up table of the
 * query (or of the point being inserted) against the stored code bytes, fp32 adds in sub-space order.  It
 * replaces hnsw_bind.Index(space='pq') as used by HnswIndex (annlite/core/index/hnsw/index.py:60-167;
 * bindings/hnsw_bindings.cpp:206-375).  The graph only proposes ef_search candidates per query; their
 * distances and the final top-k come from the GPU (annlite_adc_gather / annlite_exact_gather_dist +
 * annlite_topk_rows in annlite_hip.h).  Graph ids are build-order dependent in the reference too
 * (SURVEY.md section 8c): parity is recall, not bits.
 */
#ifndef ANNLITE_GRAPH_H
#define ANNLITE_GRAPH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define ANNLITE_GRAPH_API __attribute__((visibility("default")))

typedef struct annlite_hnsw annlite_hnsw;

/* codebooks f32 [M][Ks][dsub] (copied).  max_connection / ef_construction: hnsw/index.py:66-69 defaults 16 / 200.
 * The graph is built and walked with L2 tables whatever the metric of the index (see hnsw_host.cpp: inner-
 * product tables make half of the points unreachable); the metric's own distances are computed on the GPU. */
ANNLITE_GRAPH_API annlite_hnsw *annlite_hnsw_create(const float *codebooks, int64_t M, int64_t Ks, int64_t dsub,
                                                    int64_t capacity, int max_connection, int ef_construction,
                                                    uint64_t seed);
ANNLITE_GRAPH_API void annlite_hnsw_free(annlite_hnsw *g);
ANNLITE_GRAPH_API const char *annlite_hnsw_last_error(void);
ANNLITE_GRAPH_API int64_t annlite_hnsw_size(const annlite_hnsw *g);
ANNLITE_GRAPH_API int annlite_hnsw_reserve(annlite_hnsw *g, int64_t capacity);
/* Insert n points: x f32 [n][M*dsub] (already pre-processed: normalised for cosine), their codes u8 [n][M] and
 * labels (row ids, dense from 0: label == slot).  Replaces HnswIndex.add_with_ids -> Index.add_items(x, ids,
 * dtables) (hnsw/index.py:124-137).  Thread-parallel (n_threads <= 0: all cores). */
ANNLITE_GRAPH_API int annlite_hnsw_add(annlite_hnsw *g, const float *x, const uint8_t *codes, const int64_t *labels,
                                       int64_t n, int n_threads);
/* Candidate lists: for each of B queries (f32 [B][M*dsub]) the ef best nodes found by the level-0 beam search,
 * ascending by the L2 PQ distance of the walk; out_ids i64 [B][ef] padded with -1, out_dist f32 [B][ef] (+inf).
 * Replaces the graph part of Index.knn_query(query, k, dtables) (hnsw/index.py:139-167). */
ANNLITE_GRAPH_API int annlite_hnsw_search(const annlite_hnsw *g, const float *queries, int64_t B, int ef,
                                          int64_t *out_ids, float *out_dist, int n_threads);
/* For the GPU walk (annlite_graph_search in annlite_hip.h): the level-0 link lists of rows [0, n_rows) as
 * u32 [n_rows][links_per_node + 1] (count, ids) and a seed set -- all nodes of the top levels of the hierarchy,
 * as many levels as fit max_seeds -- which the GPU scans flat instead of descending the upper layers. */
ANNLITE_GRAPH_API int annlite_hnsw_links_per_node(const annlite_hnsw *g);
ANNLITE_GRAPH_API int annlite_hnsw_export(const annlite_hnsw *g, int64_t n_rows, uint32_t *links_out, int64_t *seeds_out,
                                          int64_t max_seeds, int64_t *n_seeds_out);
ANNLITE_GRAPH_API int annlite_hnsw_mark_deleted(annlite_hnsw *g, int64_t label);
ANNLITE_GRAPH_API int annlite_hnsw_save(const annlite_hnsw *g, const char *path);
ANNLITE_GRAPH_API annlite_hnsw *annlite_hnsw_load(const char *path);
#ifdef __cplusplus
}
#endif
#endif
