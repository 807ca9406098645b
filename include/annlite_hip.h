/*
 * annlite_hip.h -- C ABI of libannlite_hip.so: the MI355X (gfx950) implementation of annlite's
 * PQ codec + asymmetric-distance (ADC) scan hot path.
 *
 * This is the drop-in boundary.  Every entry point is what a binding for
 * the reference's `annlite.pq_bind` operator seam / `PQCodec` / index plugin would call; the
 * reference interface each one replaces is cited as (file:line) relative to jina-ai/annlite
 * v0.5.11.  The reference-side ctypes stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no exceptions across the boundary: every function returns an int status
 *     (ANNLITE_OK == 0); annlite_hip_last_error() gives a thread-local message.
 *   - all `*_dev` pointers are DEVICE pointers (HBM) owned by the caller (e.g. torch tensors'
 *     data_ptr()); outputs are caller-allocated.  No torch / HIP types in the signatures:
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all launches are asynchronous on `stream`; the caller synchronises.
 *   - no shared mutable state inside the library: calls on different streams / host threads are
 *     independent (the reference's pq_bind is serial under the GIL, hnsw_bind attaches the LUT to a
 *     shared space object and is not re-entrant: include/hnswlib/space_pq.h:55-64).  What a call
 *     remembers for the next one -- which scan kernel suits a code table -- lives in an
 *     annlite_scan_state the CALLER owns (one per table), never in the process.  The thread-local
 *     items are the last-error message and the measurement hooks (annlite_profile_*).  The ANNLITE_* environment
 *     variables (A/B switches of measurements and tests, listed in DESIGN.md) are parsed ONCE when the library is
 *     loaded; nothing on the search path reads the environment (annlite_knobs_reload() below).
 *   - there is NO CPU fallback in this library: if no gfx950 device is present the launches fail
 *     with ANNLITE_ERR_HIP.
 *
 * Distances are float32, ADC sums are accumulated in float32 in ascending sub-space order, the LUT
 * entries are sequential fused multiply-add chains -- bit-identical to the reference built with its
 * own flags (setup.py:125-144) -- see DESIGN.md "Numerics".
 */
#ifndef ANNLITE_HIP_H_
#define ANNLITE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANNLITE_HIP_ABI_VERSION 2

/* exported symbols (the library is built with -fvisibility=hidden) */
#define ANNLITE_API __attribute__((visibility("default")))

/* status codes */
#define ANNLITE_OK 0
#define ANNLITE_ERR_INVALID 1     /* bad argument (shape, null pointer, k out of range ...)        */
#define ANNLITE_ERR_UNSUPPORTED 2 /* valid request this build has no kernel for                    */
#define ANNLITE_ERR_HIP 3         /* HIP runtime error (no device, launch failure ...)             */
#define ANNLITE_ERR_WORKSPACE 4   /* workspace too small: call the matching *_workspace_bytes()    */
#define ANNLITE_NOT_APPLICABLE 5  /* not an error: the request has no effect for this shape / state; nothing was launched
                                     (annlite_pq_search_split: make the plain call instead)                              */

/* metric ids == annlite/enums.py:25-28 (Metric.EUCLIDEAN / INNER_PRODUCT / COSINE) */
#define ANNLITE_METRIC_EUCLIDEAN 1
#define ANNLITE_METRIC_INNER_PRODUCT 2
#define ANNLITE_METRIC_COSINE 3

/* look-up-table kinds (what one table entry holds) */
#define ANNLITE_LUT_L2 1     /* sum_j (C[m,k,j]-q[j])^2    bindings/pq_bindings.pyx:149-210 (and 85-145) */
#define ANNLITE_LUT_IP 2     /* sum_j  C[m,k,j]*q[j]        bindings/pq_bindings.pyx:214-274              */
#define ANNLITE_LUT_IPDIST 3 /* float32(1/Ks) - LUT_IP      annlite/core/codec/pq.py:316-322              */

/* code-table layouts in HBM */
#define ANNLITE_CODES_PLAIN 0  /* [N][M] row-major, byte m of row n = code of sub-space m (the reference's
                                  PQIndex._data / HNSW level-0 record layout)                              */
#define ANNLITE_CODES_SKEWED 1 /* [N][M], byte j of row n = code of sub-space (j + n) mod M: every row is
                                  pre-rotated by its own row id so that the scan kernel's lane l (row n,
                                  n mod M == l mod M) reads sub-space (l + t) mod M at byte t with no
                                  in-register rotation (annlite_codes_skew converts; uint8 codes only)     */

/* look-up-table layouts in HBM */
#define ANNLITE_LAYOUT_BMK 0 /* [B][M][Ks]  -- the reference's layout (what get_dist_mat returns)       */
#define ANNLITE_LAYOUT_TILED 1 /* [ceil16(B)/QI][Ks][M][QI] -- scan layout (B padded to 16), QI from the plan */

ANNLITE_API int annlite_hip_abi_version(void);
ANNLITE_API const char *annlite_hip_last_error(void);
/* number of visible HIP devices and the gfx arch name of device `dev` (e.g. "gfx950:sramecc+:xnack-") */
ANNLITE_API int annlite_hip_device_count(int *count);
ANNLITE_API int annlite_hip_device_arch(int dev, char *buf, size_t buf_len);
/* The measurement / test switches (ANNLITE_SCAN_VARIANT, ANNLITE_Q8_*, ANNLITE_SEED_*, ANNLITE_NO_*, ANNLITE_GUARD_BASE, ...) are
 * read from the environment once, at load.  A process that changes them afterwards calls this to have them parsed again: the new
 * block is published atomically, calls already running keep the one they started with.  No reference counterpart (the reference has
 * no such switches); a production deployment never calls it. */
ANNLITE_API int annlite_knobs_reload(void);

/* ------------------------------------------------------------------------------------------------
 * Scan plan: how the ADC scan kernel wants its inputs for a (M, Ks, code width, k) problem.
 * ---------------------------------------------------------------------------------------------- */
typedef struct annlite_scan_plan {
    int32_t fast;          /* 1: LDS-resident skewed-gather kernel, 0: generic (LUT read through L2)  */
    int32_t qi;            /* queries interleaved per LUT entry in the tiled layout (4, 2 or 1)       */
    int32_t qt;            /* queries per workgroup (multiple of qi)                                  */
    int32_t waves;         /* waves per workgroup                                                     */
    int32_t n_slices;      /* row slices the table is cut into (>= 8: one per XCD)                    */
    int32_t max_k;         /* largest k this plan supports                                            */
    int64_t lut_floats;    /* number of floats of a TILED LUT buffer for B queries                    */
    int64_t workspace_bytes; /* scratch needed by annlite_adc_scan_topk()                             */
} annlite_scan_plan;

ANNLITE_API int annlite_scan_plan_query(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                            annlite_scan_plan *plan);
/* the plan annlite_pq_search_tiles() runs with for V scan slots (its query tiles hold plan.qt slots each) */
ANNLITE_API int annlite_scan_plan_tiles(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t V, int64_t k,
                            annlite_scan_plan *plan);

/* ------------------------------------------------------------------------------------------------
 * LUT construction.
 * replaces: pq_bind.batch_precompute_adc_table      (bindings/pq_bindings.pyx:149-210)  kind=L2
 *           pq_bind.batch_precompute_adc_table_ip   (bindings/pq_bindings.pyx:214-274)  kind=IP
 *           pq_bind.precompute_adc_table            (bindings/pq_bindings.pyx:85-145)   kind=L2, B=1
 *           PQCodec.get_dist_mat                    (annlite/core/codec/pq.py:293-325)  kind=L2|IPDIST
 * queries_dev   f32 [B][D]  (already normalised by the caller for COSINE, as pq.py:309-310 does)
 * codebooks_dev f32 [M][Ks][dsub], dsub = D / M     (PQCodec.get_codebook(), pq.py:231-237)
 * out_dev       f32, layout BMK: [B][M][Ks];  TILED: plan.lut_floats floats (pad queries zeroed)
 * `qi` is only read for the TILED layout (take it from annlite_scan_plan_query()).
 * Numerics: sequential fmaf chain over j, bit-exact vs the reference; the IP kinds run on the
 * f32 MFMA (v_mfma_f32_16x16x4_f32), which is bitwise that same chain.
 * ---------------------------------------------------------------------------------------------- */
ANNLITE_API int annlite_lut_build(int kind, const float *queries_dev, int64_t B, int64_t D,
                      const float *codebooks_dev, int64_t M, int64_t Ks, float *out_dev, int layout,
                      int qi, void *stream);

/* re-layout a [B][M][Ks] table (e.g. one the caller computed himself) into the scan layout */
ANNLITE_API int annlite_lut_retile(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks, float *out_tiled_dev,
                       int qi, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Flat ADC scan, all distances (the operator seam, one query).
 * replaces: pq_bind.dist_pqcodes_to_codebooks (bindings/pq_bindings.pyx:52-80) == DistanceTable.adist
 *           (annlite/core/codec/pq.py:350-368):  out[n] = sum_{m ascending} adtable[m][codes[n][m]]
 * adtable_dev f32 [M][Ks] ; codes_dev [N][M] of code_bytes (1|2|4) ; out_dev f32 [N]
 * ---------------------------------------------------------------------------------------------- */
ANNLITE_API int annlite_adc_dist(const float *adtable_dev, int64_t M, int64_t Ks, const void *codes_dev,
                     int code_bytes, int64_t N, float *out_dev, void *stream);

/* Gathered ADC: out[b][r] = ADC distance of row cand[b][r] under query b's table (cand < 0 -> +inf).
 * replaces: hnswlib::PQLookup (include/hnswlib/space_pq.h:15-37) evaluated over a candidate list --
 * the "GPU ADC rerank of HNSW candidate lists" of BASELINE config 5.
 * lut_bmk_dev f32 [B][M][Ks] ; cand_dev i64 [B][R] ; out_dev f32 [B][R] */
ANNLITE_API int annlite_adc_gather(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks, const void *codes_dev,
                       int code_bytes, int64_t N, const int64_t *cand_dev, int64_t R, float *out_dev,
                       void *stream);

/* Beam search over the level-0 lists of an HNSW-over-PQ graph, on the GPU (BASELINE config 5): one wave per
 * query, its L2 table and a visited set in LDS, the ef best nodes as a sorted list in registers.
 * replaces: the graph walk of hnsw_bind.Index.knn_query (bindings/hnsw_bindings.cpp:302-375 ->
 * include/hnswlib/hnswalg.h searchBaseLayerST) for a whole batch; edge distances are hnswlib::PQLookup
 * (space_pq.h:15-37) bit for bit.  The graph comes from libannlite_graph.so (annlite_hnsw_export).
 * links_dev u32 [N][links_per_node + 1] (count, ids) ; seeds_dev u32 [n_seeds] DISTINCT nodes (top of the
 * hierarchy, scanned flat -- REQUIRED distinct, as annlite_hnsw_export produces them: the seed scan inserts without a duplicate
 * test, a node listed twice would enter the beam twice and come back twice) ; codes_dev u8 [N][M] PLAIN ; lut_bmk_dev f32 [B][M][Ks] L2 tables ; ef <= 256 ; M in {8, 16, 32, 64} (64, round 6: 64 KB of table per wave, one wave per CU)
 * out_ids_dev i64 [B][ef] ascending by distance (-1 padded, deleted rows per valid_bits dropped),
 * out_dist_dev f32 [B][ef] (+inf padded). */
ANNLITE_API int annlite_graph_search(const uint32_t *links_dev, int links_per_node, const uint32_t *seeds_dev,
                         int64_t n_seeds, const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                         const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef,
                         int64_t *out_ids_dev, float *out_dist_dev, void *stream);
/* The same walk over PACKED node records (round 5): a record holds the node's neighbours' CODE ROWS behind its link list --
 * [L x M code bytes][L x u32 ids][u32 count][pad to 16 B], annlite_graph_record_bytes() per node (656 B at L = 32, M = 16) --
 * so one expansion is ONE contiguous read instead of the link list followed by L random 16-byte rows (a 128-byte line
 * each), and the record of the next node is prefetched while the current one's neighbours are evaluated.  Same walk order,
 * same hnswlib::PQLookup arithmetic (space_pq.h:15-37, hnswalg.h:243-329): candidate lists bit-equal to annlite_graph_search's.
 * annlite_graph_pack builds the records from the exported lists and the PLAIN code table (rebuild after inserts; deletes
 * need none: deleted rows keep routing the walk, as in hnswlib).  links_per_node <= 64; codes_dev is still read for the seeds. */
ANNLITE_API int annlite_graph_record_bytes(int links_per_node, int64_t M, int64_t *bytes);
ANNLITE_API int annlite_graph_pack(const uint32_t *links_dev, int links_per_node, const void *codes_dev, int64_t N, int64_t M,
                       void *packed_dev, void *stream);
ANNLITE_API int annlite_graph_search_packed(const void *packed_dev, int links_per_node, const uint32_t *seeds_dev,
                         int64_t n_seeds, const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                         const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef,
                         int64_t *out_ids_dev, float *out_dist_dev, void *stream);
/* ... with expand_width nodes expanded per step (round 6).  1 = annlite_graph_search_packed.  2 = the PAIR walk (links_per_node <= 32):
 * the two best unexpanded entries of the list are expanded TOGETHER -- one half wave per record, one pass through the visited table,
 * one merge of up to 64 neighbours -- so the chain of dependent steps of a walk (what a launch of one wave per query lasts) is half as
 * long.  The stop rule (no unexpanded entry among the ef best, hnswalg.h searchBaseLayerST) and the arithmetic are unchanged; the
 * ORDER of the expansions is not: the runner-up is expanded before the winner's neighbours are known, so the list can differ from
 * the one-at-a-time walk's in its last places (candidate overlap > 0.99 at ef = 128; tests/test_graph_pair.py pins the pair order
 * bit for bit against a restatement). */
ANNLITE_API int annlite_graph_search_packed_ex(const void *packed_dev, int links_per_node, const uint32_t *seeds_dev,
                         int64_t n_seeds, const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                         const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef, int expand_width,
                         int64_t *out_ids_dev, float *out_dist_dev, void *stream);
/* ---- the level-0 graph BUILT ON THE GPU, in batches (round 6; graph_build.hip) -------------------------------------------------
 * replaces, for a batch of points: hnswlib addPoint (include/hnswlib/hnswalg.h:1108-1235) = searchBaseLayer with ef_construction
 * (the walk above over the graph as it is, the points' own L2 tables as queries), getNeighborsByHeuristic2 (378-429) and
 * mutuallyConnectNewElement (431-553), in the form libannlite_graph.so gives them (hnsw_host.cpp select_neighbors / connect: the
 * triangle tests on the SYMMETRIC code-to-code L2 table).  Only level 0 exists -- the GPU walk scans a seed sample flat and never
 * descends upper layers.  Node id = row of the PLAIN code table.
 *   annlite_graph_build_sdc      sdc f32 [M][Ks][Ks] <- sum_j (C[m][a][j] - C[m][b][j])^2
 *   annlite_graph_build_select   point i of the batch = node base0 + i; cand_dev i64 [b][ef] = its candidates in ascending search
 *                                distance (-1 = none; ef <= 256): keeps at most max_keep of them (every one if they are that few),
 *                                writes links[base0 + i] = (count, ids) and pairs[i][0..max_keep) = (target << 32) | source
 *                                (INT64_MAX = none)
 *   annlite_graph_build_reverse  keys_dev i64 [P] = the pairs SORTED ascending; seg_dev i64 [S + 1] = the boundaries of the S runs
 *                                of equal target: every target appends its sources while its list has room, else shrinks
 *                                (list + sources) to links_per_node (<= 32) entries with the same heuristic; a target takes at most
 *                                64 - (its list's length) sources of one call (the lowest ids; the rest are dropped: a link the
 *                                heuristic would most likely have pruned -- only the all-at-once start of a graph sees such hubs)
 *   annlite_graph_pack_nodes     the packed records (annlite_graph_pack) of the nodes in nodes_dev i64 [n_nodes] only */
ANNLITE_API int annlite_graph_build_sdc(const float *codebooks_dev, int64_t M, int64_t Ks, int64_t dsub, float *sdc_dev, void *stream);
ANNLITE_API int annlite_graph_build_select(const int64_t *cand_dev, int ef, int64_t b, int64_t base0, const void *codes_dev, int64_t M,
                         int64_t Ks, const float *sdc_dev, int max_keep, uint32_t *links_dev, int links_per_node,
                         int64_t *pairs_dev, void *stream);
ANNLITE_API int annlite_graph_build_reverse(const int64_t *keys_dev, const int64_t *seg_dev, int64_t n_segments, const void *codes_dev,
                         int64_t M, int64_t Ks, const float *sdc_dev, uint32_t *links_dev, int links_per_node, void *stream);
ANNLITE_API int annlite_graph_pack_nodes(const uint32_t *links_dev, int links_per_node, const void *codes_dev, int64_t N, int64_t M,
                         const int64_t *nodes_dev, int64_t n_nodes, void *packed_dev, void *stream);
/* Debug aid: with ANNLITE_DEBUG_COUNTERS=1 the walk counts [0] expansions (link lists read) and [1] rows evaluated
 * (PQLookup sums) over the batch; this copies the two counters of the last walk to the host (the roofline of
 * scripts/bench_hnsw.py: algorithmic bytes = expansions * 4 (links_per_node + 1) + evaluations * M). */
ANNLITE_API int annlite_graph_search_stats(uint64_t *out2);
/* ... EIGHT counters: [2] the expansions whose record had been prefetched (packed walk); shader cycles summed over the queries'
 * waves: [3] seed phase, [4] pick + wait for the record, [5] visited table, [6] PQLookup sums, [7] list merge. */
ANNLITE_API int annlite_graph_search_stats_ex(uint64_t *out8);

/* ------------------------------------------------------------------------------------------------
 * Batched flat ADC scan + top-k: the hot path.
 * replaces, for B queries in ONE launch: PQIndex.search (annlite/core/index/pq_index.py:29-56:
 * LUT -> adist over the code table -> math.top_k, annlite/math.py:94-120), i.e. the per-query loop
 * of CellContainer.search_cells (annlite/container.py:214) over the index plugin's .search().
 *
 * codes_dev   [N][M] codes, row-major, code_bytes 1 (Ks<=256) or 2; codes_layout PLAIN or SKEWED
 *             (SKEWED only with the fast plan)
 * valid_bits  optional u32 bitmap, bit n set = row n may be returned (NULL = all N rows).  Carries
 *             delete() marks (annlite/core/index/hnsw/index.py:169-171) and the `indices` filter
 *             argument of search (pq_index.py:42-44).
 * lut_dev     the tables in plan.layout order: TILED for plan.fast, else BMK
 * out_dist_dev f32 [B][k], out_id_dev i64 [B][k]: ascending by (distance, row id) -- the fixed
 *             tie-break -- ids are row_base + row; missing entries (k > #valid rows) are (+inf, -1).
 * workspace   plan.workspace_bytes of device scratch.
 * ---------------------------------------------------------------------------------------------- */
ANNLITE_API int annlite_adc_scan_topk(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M,
                          int64_t Ks, const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B, int64_t k,
                          int64_t row_base, float *out_dist_dev, int64_t *out_id_dev,
                          void *workspace_dev, size_t workspace_bytes, void *stream);

/* Same scan and result, written as ONE buffer out_packed_dev i64 [B][k][2] = (row_base + row or -1, the f32
 * distance's bits zero-extended): what a rank contributes to the single all-gather of the row-sharded
 * search; annlite_topk_merge_packed consumes the gathered [G][B][k][2]. */
ANNLITE_API int annlite_adc_scan_topk_packed(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M,
                                 int64_t Ks, const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B,
                                 int64_t k, int64_t row_base, int64_t *out_packed_dev, void *workspace_dev,
                                 size_t workspace_bytes, void *stream);

/* The index plugin's search() in ONE call: query batch in, neighbours out.
 * replaces: PQIndex.search / HnswIndex.search's LUT step + scan + top_k for B queries
 * (annlite/core/index/pq_index.py:29-56, hnsw/index.py:139-167: get_dist_mat -> adist/knn -> top_k).
 * Equivalent to annlite_lut_build(lut_kind, ...) into the plan's layout followed by annlite_adc_scan_topk
 * (or _packed when out_packed_dev != NULL; then out_dist_dev/out_id_dev may be NULL) -- same bits --
 * but for L2 tables on the quantised-filter plan the tables are built, reduced and quantised by one
 * launch.  queries_dev must already carry the caller's pre-processing (l2-normalised for COSINE).
 * flags: ANNLITE_FLAG_SQRT.
 * workspace: annlite_pq_search_workspace_bytes() bytes (scan scratch + the fp32 tables). */
#define ANNLITE_FLAG_SQRT 1 /* metric epilogue of EUCLIDEAN search: out_dist = sqrt(ADC sum) (hnsw/index.py:164-165);
                               applied after all merging, ignored for packed output (raw sums travel) */
ANNLITE_API int annlite_pq_search_workspace_bytes(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t B, int64_t k,
                                      int64_t *bytes);
ANNLITE_API int annlite_pq_search_topk(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                           const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                           int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                           int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                           int flags, void *workspace_dev, size_t workspace_bytes, void *stream);
/* ... with the caller's per-table state (see annlite_scan_state below; NULL = annlite_pq_search_topk) */
struct annlite_scan_state;
ANNLITE_API int annlite_pq_search_topk_ex(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                              const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                              int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                              int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                              int flags, void *workspace_dev, size_t workspace_bytes, void *stream,
                              struct annlite_scan_state *state);

/* The row-sharded search of one process per GPU, with a SEED EXCHANGE between the ranks: the search above
 * in two halves, so that a rank can seed from 1/G of the rows a single GPU would and still start its scan with the bound of
 * all G ranks' seed rows together.  (Every rank repeats the per-batch work for all B queries; the seed bound -- the exact
 * k-th distance of S rows spread over the table -- is the largest part of it: 22 of 39 us at 32768 rows x 1024 queries.)
 *   phase ANNLITE_PHASE_PREPARE  tables, parameters, reset, the seed bound from `seed_rows` rows spread over this rank's table (<= 0: the
 *       single-GPU default) and -- the rank's contribution to one all-gather -- seed_keys_dev [B][ANNLITE_SEED_KEYS] u64: the
 *       bounds implied by the seed's k smallest rows, ascending (all-ones where it has fewer).  Outputs untouched.
 *   annlite_pq_search_seed_union  all_keys_dev [G][B][ANNLITE_SEED_KEYS] (the all-gathered keys): the k-th smallest of a
 *       query's G * k keys has k distinct rows of the GLOBAL table at or below it -- the prepared batch's bound becomes
 *       min(own, that).
 *   phase ANNLITE_PHASE_SCAN     the scan of the prepared batch (same arguments, same workspace, same stream order).
 * PREPARE returns ANNLITE_NOT_APPLICABLE -- nothing launched, make the plain call -- unless the batch runs the byte-table
 * plan with the fused preparation launch (M = 16, uint8 codes, L2 tables, k <= 16, D <= 256, N >= 4096) and `state` has
 * settled on the byte-table kernel.  Results are those of the plain call, bit for bit (a bound only prunes). */
#define ANNLITE_PHASE_PREPARE 1
#define ANNLITE_PHASE_SCAN 2
#define ANNLITE_SEED_KEYS 16
ANNLITE_API int annlite_pq_search_split(int phase, int64_t seed_rows, int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                            const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                            int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k,
                            int64_t row_base, float *out_dist_dev, int64_t *out_id_dev, int64_t *out_packed_dev,
                            int flags, void *workspace_dev, size_t workspace_bytes, void *stream,
                            struct annlite_scan_state *state, uint64_t *seed_keys_dev);
ANNLITE_API int annlite_pq_search_seed_union(const uint64_t *all_keys_dev, int64_t G, int64_t N, int64_t M, int64_t Ks,
                                 int code_bytes, int64_t B, int64_t k, void *workspace_dev, size_t workspace_bytes,
                                 void *stream);

/* Same scan, but return the UNMERGED per-slice lists: plan.n_slices * k candidates per query
 * ([B][n_slices*k], unordered across slices, (+inf,-1) where a slice had fewer rows).  The set is a
 * superset of the exact top-k; it is the candidate generator of the exact re-rank stage
 *  -- the reference has no such stage. */
ANNLITE_API int annlite_adc_scan_candidates(const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M,
                                int64_t Ks, const uint32_t *valid_bits_dev, const float *lut_dev, int64_t B,
                                int64_t k, int64_t row_base, float *out_dist_dev, int64_t *out_id_dev,
                                void *workspace_dev, size_t workspace_bytes, void *stream);
/* ... with the tables built inside the call (round 6): queries_dev f32 [B][D] + codebooks_dev instead of a prebuilt table; same
 * output, out[b][0 .. n_slices * k) with n_slices of annlite_scan_plan_query(N, M, Ks, code_bytes, B, k); workspace of
 * annlite_pq_search_workspace_bytes.  M = 16, L2 tables, k <= 16: ONE preparation launch (tables, list reset, ONE first bound from
 * rows spread over the whole table, prebuilt byte tables) instead of four; every slice then keeps the k best of ITS rows that are at
 * or below that bound -- the bound lies at or above the k-th key of the whole table, so the lists still hold the table's top-k
 * (and a row beyond it is of no use to a re-rank: a slice far from the query returns fewer than k rows, -1 padded). */
ANNLITE_API int annlite_pq_search_candidates(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                             const void *codes_dev, int code_bytes, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                             const uint32_t *valid_bits_dev, int64_t k, int64_t row_base, float *out_dist_dev,
                             int64_t *out_id_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* Measurement hooks (bench.py): when enabled on the calling thread, annlite_adc_scan_topk /
 * _candidates bracket the scan kernel launch -- the dominant kernel -- with HIP events on the
 * launch stream; annlite_profile_last_scan_ms() waits for the last one and returns its duration. */
ANNLITE_API int annlite_profile_enable(int on);
ANNLITE_API int annlite_profile_last_scan_ms(float *ms);
/* ... and, for the byte-table kernel, the shader clock that launch actually held (MHz): workgroup 0's s_memtime delta over
 * its 100 MHz wall-clock delta.  ANNLITE_ERR_INVALID when the last profiled scan was not a byte-table launch. */
ANNLITE_API int annlite_profile_last_scan_clock_mhz(float *mhz);
/* Revision (> 0) of a kernel's memory behaviour, by kernel name ("adc_scan_q8_kernel", ...); 0 = unknown name.  Recorded with every
 * kept PMC pass (profiles/traffic.json): bench.py's roofline.traffic refuses a pass taken on another revision. */
ANNLITE_API int annlite_kernel_rev(const char *kernel);
/* Kernel choice (M = 8 / 16 / 32 with uint8 codes, M = 8 with uint16 codes up to Ks = 1024; k <= 16): byte filter tables (the default)
 * or u16 filter tables -- made INSIDE the library, per call, never by a process-wide switch.  Without a state every
 * byte-table launch is guarded: it gives up when a workgroup has seen more than 1024 + (rows it has drawn) / 16 candidates
 * (a byte filter that leaks) and a gated u16-table pass queued behind it redoes the scan (~10 us per batch of gated
 * launches that return at once otherwise).  With a state -- ONE PER CODE TABLE, created once, passed to
 * annlite_pq_search_topk_ex -- every byte-table launch leaves its candidate count in the state's host-mapped block; later
 * calls read it without synchronising: a launch that gave up settles the table on the u16 kernel, one with few candidates
 * on the byte tables, and in between the state times ONE call of each kernel (events in the caller's stream, queried,
 * never waited for) and keeps the faster -- until the table has doubled or halved.  Results are identical whatever runs.  (Calls
 * that carry a state record events and read a host-mapped block: do not issue them inside a stream capture.)  A state must
 * not be used by two host threads at once (one searcher per index, as the reference has: SURVEY.md section 8b);
 * destroy it only after the launches that were given it have completed.
 * annlite_scan_state_info: kernel = 0 undecided, 1 byte tables, 2 u16 tables; rows / candidates of the deciding launch.
 * (ANNLITE_SCAN_VARIANT in the environment, read per call, still forces one instantiation for A/B measurements.) */
typedef struct annlite_scan_state annlite_scan_state;
ANNLITE_API int annlite_scan_state_create(annlite_scan_state **out);
ANNLITE_API int annlite_scan_state_destroy(annlite_scan_state *state);
ANNLITE_API int annlite_scan_state_reset(annlite_scan_state *state);
ANNLITE_API int annlite_scan_state_info(annlite_scan_state *state, int32_t *kernel, int64_t *rows, uint64_t *candidates);
/* Debug aid: with ANNLITE_DEBUG_COUNTERS=1 in the environment the scan counts events; this copies the 8 uint64
 * counters of the last scan to the host.  u16 kernels: [0] slow-block entries [1] (wave,query) candidate events
 * [2] events that inserted [3] bound publications [4] candidate rows.  Byte-table kernel: [0] wave-steps with a
 * candidate [1] candidates pushed [2] exact sums [3] candidates queued for a list [4] consumer-wave cycles inside
 * batches [5] table rebuilds [6] consumer batches [7] cycles of wave 0 at epoch ends. */
ANNLITE_API int annlite_debug_counters(uint64_t *out8);
/* Debug aid, same switch: phase stamps of the byte-table kernel's workgroups (thread 0, 100 MHz wall clock): [0] 2^62 -
 * earliest start, [1] latest end; sums over the work items of [2] start stamp, [3] initialisation + first table build,
 * [4] step loop, [5] wait at the last barrier, [6] list store + merge; [7] work items. */
ANNLITE_API int annlite_debug_timeline(uint64_t *out8);
/* ... and per work item (up to 4096 records of 8 uint64): query tile, row slice, the stamps start / table built / step loop
 * left / last barrier passed / end, block index. */
ANNLITE_API int annlite_debug_items(uint64_t *out, int64_t max_items, int64_t *n_items);
/* ... and of the byte-table plan's preparation launch (tables + parameters + reset + seed bound [+ byte tables], one launch):
 * stamps of its first workgroup [0..3] and its last one [4..7]: start, tables built, seed rows scanned, end. */
ANNLITE_API int annlite_debug_prep_timeline(uint64_t *out8);
/* Test hook of the seed bound's MFMA nomination launch (round 6, seed_mfma.hip; M = 16, 128-d, uint8 codes): `seed_rows` seed rows
 * (a multiple of 8192, <= 131072, <= N) spread over the table are cut into 512 disjoint groups; cand_dev u32 [B][512] receives, per
 * query and group, the table row with the smallest APPROXIMATE (bf16 contraction) ADC distance -- 0xffffffff where the group holds
 * no valid row.  The search itself never returns these: it takes the k-th smallest EXACT sum of a query's nominees as its first
 * bound (any k distinct valid rows give a valid one).  ANNLITE_NOT_APPLICABLE for other shapes.  No reference counterpart. */
ANNLITE_API int annlite_debug_seed_candidates(const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                                  const void *codes_dev, int codes_layout, int64_t N, int64_t M, int64_t Ks,
                                  const uint32_t *valid_bits_dev, int64_t seed_rows, uint32_t *cand_dev, void *stream);

/* Convert between the PLAIN and the SKEWED code-table layout (uint8 codes).
 * forward (inverse=0): table_out[id][j] = codes_in[i][(j + id) mod M]   -- scatter rows i -> id
 *                      (M = 64: each 32-byte half is rotated by id mod 32 on its own and "wrap-coded" -- minus 1 mod 256
 *                      where j mod 32 + id mod 32 >= 32 -- the form the M = 64 scan kernel turns into an LDS address with
 *                      one byte permute; SKEWED is an opaque, per-M storage format: always produce and undo it with this
 *                      function)
 * inverse (inverse=1): codes_out[i][j]  = table_in[id][(j - id) mod M]  -- gather rows id -> i
 * id = ids_dev[i] if ids_dev != NULL else id_base + i.  This is the storage step of the index
 * plugin's add_with_ids (annlite/core/index/pq_index.py:25-27 -> flat_index.py:41-50 `_data[ids] = x`). */
ANNLITE_API int annlite_codes_skew(const void *in_dev, int64_t N, int64_t M, const int64_t *ids_dev, int64_t id_base,
                       void *out_dev, int inverse, void *stream);

/* Merge G sorted candidate lists per query into one: in [G][B][k] -> out [B][k], same order rule.
 * This is the step after the RCCL all-gather of per-shard top-k; the
 * reference's analogue is the hstack+argsort merge of per-cell results (annlite/container.py:130-138). */
ANNLITE_API int annlite_topk_merge(const float *dist_dev, const int64_t *id_dev, int64_t G, int64_t B, int64_t k,
                       float *out_dist_dev, int64_t *out_id_dev, void *stream);

/* The same merge over the packed form of annlite_adc_scan_topk_packed: in i64 [G][B][k][2]. */
ANNLITE_API int annlite_topk_merge_packed(const int64_t *packed_dev, int64_t G, int64_t B, int64_t k,
                              float *out_dist_dev, int64_t *out_id_dev, int flags, void *stream);

/* Row-wise k smallest of a dense f32 matrix [B][N] -> ([B][k], [B][k]) with the fixed tie-break.
 * replaces: annlite.math.top_k(values, k, descending=False) (annlite/math.py:94-120). k <= 64. */
ANNLITE_API int annlite_topk_rows(const float *values_dev, int64_t B, int64_t N, int64_t k, int64_t id_base,
                      float *out_dist_dev, int64_t *out_id_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Codec.
 * ---------------------------------------------------------------------------------------------- */
/* replaces: PQCodec.encode (annlite/core/codec/pq.py:158-177): codes[n][m] = argmin_k |x_sub - C[m,k]|^2,
 * first minimum wins.  out codes [N][M] of code_bytes (1 if Ks<=256, 2 if Ks<=65536: pq.py:56-60). */
ANNLITE_API int annlite_pq_encode(const float *x_dev, int64_t N, int64_t D, const float *codebooks_dev, int64_t M,
                      int64_t Ks, void *out_codes_dev, int code_bytes, void *stream);

/* replaces: PQCodec.decode (annlite/core/codec/pq.py:179-198): gather codewords. out f32 [N][D] */
ANNLITE_API int annlite_pq_decode(const void *codes_dev, int code_bytes, int64_t N, int64_t M, int64_t Ks,
                      const float *codebooks_dev, int64_t dsub, float *out_dev, void *stream);

/* replaces: annlite.math.l2_normalize (annlite/math.py:6-18): rows with norm < 10*eps unscaled.
 * In-place allowed (out_dev == x_dev). */
ANNLITE_API int annlite_l2_normalize(const float *x_dev, int64_t N, int64_t D, float *out_dev, void *stream);

/* One Lloyd iteration building block for PQCodec.fit (annlite/core/codec/pq.py:89-115, sklearn KMeans
 * per sub-space): assign every training row to its nearest codeword in every sub-space and accumulate
 * per-codeword sums/counts.  sums_dev f32 [M][Ks][dsub], counts_dev i32 [M][Ks] (both zeroed by the
 * caller), inertia_dev f64 [M] (zeroed by caller, may be NULL). */
ANNLITE_API int annlite_kmeans_assign_accumulate(const float *x_dev, int64_t N, int64_t D, const float *codebooks_dev,
                                     int64_t M, int64_t Ks, float *sums_dev, int32_t *counts_dev,
                                     double *inertia_dev, void *stream);
/* codebooks[m][k] = sums/counts where counts > 0 (empty clusters keep their old centre). */
ANNLITE_API int annlite_kmeans_update(const float *sums_dev, const int32_t *counts_dev, int64_t M, int64_t Ks,
                          int64_t dsub, float *codebooks_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Exact re-rank (GPU analogue of FlatIndex.search,
 * annlite/core/index/flat_index.py:15-39, and hnswlib's space_l2.h/space_ip.h distance functions):
 * out[b][r] = exact distance between queries[b] and vectors[cand[b][r]] (cand < 0 -> +inf).
 * metric EUCLIDEAN -> squared L2 ; INNER_PRODUCT / COSINE -> 1 - <q, x> (inputs already normalised
 * for COSINE).  vectors_dev f32 [N][D]. */
ANNLITE_API int annlite_exact_gather_dist(int metric, const float *queries_dev, int64_t B, int64_t D,
                              const float *vectors_dev, int64_t N, const int64_t *cand_dev, int64_t R,
                              float *out_dev, void *stream);
/* The same distances FUSED with the top-k (round 6): out[b][0..k) = the k smallest exact distances among cand[b][0..R) in
 * (distance, position in the list) order -- what annlite_exact_gather_dist + annlite_topk_rows + a gather of the ids give, bit for
 * bit -- as (distance, cand id); candidates < 0, >= N or cleared in valid_bits_dev (may be NULL) are skipped, missing places hold
 * (+inf, -1).  k <= 64.  flags: ANNLITE_FLAG_SQRT (EUCLIDEAN results, hnsw/index.py:164-165).  One wave per query. */
ANNLITE_API int annlite_rerank_topk(int metric, const float *queries_dev, int64_t B, int64_t D, const float *vectors_dev, int64_t N,
                        const int64_t *cand_dev, int64_t R, const uint32_t *valid_bits_dev, int64_t k, int flags,
                        float *out_dist_dev, int64_t *out_id_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Pruned (IVF) search over cells (follow-on of rank 4).
 * Reference structure: AnnLite(n_cells > 1): VQCodec coarse quantiser (annlite/core/codec/vq.py),
 * AnnLite._cell_selection (annlite/index.py:458-466: cdist(query, vq codebook) -> top_k(n_probe)),
 * CellContainer.ivf_search (annlite/container.py:88-144: one index per cell, lists merged).  The
 * reference always probes every cell (n_probe = max(n_probe, n_cells), index.py:94); probing fewer
 * is this build's extension of the same structure.
 * Table layout: the rows of a cell are CONTIGUOUS in the code table, every cell starts at a multiple
 * of 64 rows (padding rows are invalid in the bitmap), rows of a cell ascend in external id.
 * ---------------------------------------------------------------------------------------------- */
/* cells[b][0..P) = the P nearest centroids of query b, ascending in (distance, cell).
 * kind 0: squared L2; kind 1: negative inner product (cosine: pass normalised centroids).
 * centroids_dev f32 [C][D] (VQCodec.codebook). */
ANNLITE_API int annlite_ivf_select_cells(int kind, const float *queries_dev, int64_t B, int64_t D,
                             const float *centroids_dev, int64_t C, int64_t P, int32_t *cells_dev, void *stream);

/* Upper bound of the query tiles annlite_ivf_plan can produce (qt = tile size of the scan plan). */
ANNLITE_API int64_t annlite_ivf_max_tiles(int64_t B, int64_t P, int64_t C, int64_t qt);

/* Group the B*P (query, cell) pairs into query tiles of qt slots that probe ONE cell each.
 *   cell_rows_dev  i64 [C][2]  (begin, end) rows of every cell in the code table
 *   cell_order_dev i32 [C]     cells in descending size (tiles are scanned longest first)
 *   vmap_dev       i32 [n_tiles_max*qt]  out: query of every slot, -1 = padding slot
 *   slot_of_dev    i32 [B*P]   out: slot of pair (b, p)
 *   tile_rows_dev  i64 [n_tiles_max][2]  out: row range of every tile (begin -1: unused tile)
 *   n_tiles_used_dev i32 [1]   out (may be NULL) */
ANNLITE_API int annlite_ivf_plan(const int32_t *cells_dev, int64_t B, int64_t P, int64_t C, int64_t qt,
                     const int64_t *cell_rows_dev, const int32_t *cell_order_dev, int64_t n_tiles_max,
                     int32_t *vmap_dev, int32_t *slot_of_dev, int64_t *tile_rows_dev, int32_t *n_tiles_used_dev,
                     void *stream);

/* ... with two classes of tiles: the pairs of every query's n_first nearest cells (probe ranks < n_first) in tiles of their own that
 * come FIRST, the other pairs behind them (annlite_ivf_search_topk: the scan shares a query's bound between its tiles; the k best
 * rows of the nearest cells then bound the candidates of the others from the start).  n_tiles_max >= annlite_ivf_max_tiles_first.
 * n_first = 0 or >= P: annlite_ivf_plan. */
ANNLITE_API int64_t annlite_ivf_max_tiles_first(int64_t B, int64_t P, int64_t C, int64_t qt);
ANNLITE_API int annlite_ivf_plan_first(const int32_t *cells_dev, int64_t B, int64_t P, int64_t C, int64_t qt,
                           const int64_t *cell_rows_dev, const int32_t *cell_order_dev, int64_t n_tiles_max,
                           int32_t *vmap_dev, int32_t *slot_of_dev, int64_t *tile_rows_dev, int32_t *n_tiles_used_dev,
                           int64_t n_first, void *stream);

/* The scan of annlite_pq_search_topk where query tile t = slots [t*qt, (t+1)*qt) scans ONLY rows
 * tile_rows[t] (one work item per tile, handed out dynamically) -- with INTEGER sums only: a tile is
 * too short to amortise exact fp32 recomputes.  The kernel keeps, per slot, the k smallest integer
 * sums (which bound the slot's k-th exact distance from above: any k rows with S <= Sk give
 * d_kth <= L + step*(Sk + 1.002 M) + slack) and EMITS every row that can still be in the slot's exact
 * top-k:  cand[v][0..cand_count[v])  (table rows; cand_count 0xffffffff = the list overflowed
 * cand_cap, re-score the whole cell).  annlite_ivf_rescore turns the lists into exact results.
 * queries_dev f32 [B][D]: the REAL queries -- their tables are built and quantised once; slot s of
 * the V = n_tiles * qt slots scans with the tables of query vmap[s] (-1: padding slot).
 * Quantised-filter plans only (M in {8,16,32,64}, Ks <= 256, uint8 codes). */
ANNLITE_API int annlite_pq_search_tiles_workspace_bytes(int64_t N, int64_t M, int64_t Ks, int code_bytes, int64_t V,
                                            int64_t k, int64_t *bytes);
ANNLITE_API int annlite_pq_search_tiles(int lut_kind, const float *queries_dev, int64_t B, int64_t D,
                            const float *codebooks_dev, const void *codes_dev, int code_bytes, int codes_layout,
                            int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev, int64_t k, int64_t V,
                            const int64_t *tile_rows_dev, const int32_t *vmap_dev, uint32_t *cand_dev,
                            int64_t cand_cap, uint32_t *cand_count_dev, void *workspace_dev, size_t workspace_bytes,
                            void *stream);

/* Exact re-score + merge, one workgroup per query: the query's fp32 table lut_bmk[b] ([M][Ks], what
 * get_dist_mat returns) in LDS, exact ascending-m ADC sums (space_pq.h:32-35) of the candidate rows of
 * its P probed slots (slot_of_dev i32 [B][P]), top-k under the fixed tie-break -> [B][k] (distance,
 * id_base + row_ids[row]).  codes_plain_dev: the cell-sorted table in the PLAIN layout [N][M];
 * valid_bits_dev (may be NULL) is consulted for overflowed slots, which are re-scored over
 * tile_rows[slot / qt].  flags: ANNLITE_FLAG_SQRT.
 * replaces: the per-cell search + hstack/argsort merge of CellContainer.ivf_search
 * (annlite/container.py:88-144). */
ANNLITE_API int annlite_ivf_rescore(const float *lut_bmk_dev, int64_t B, int64_t M, int64_t Ks,
                        const void *codes_plain_dev, int64_t N, const uint32_t *valid_bits_dev,
                        const uint32_t *cand_dev, int64_t cand_cap, const uint32_t *cand_count_dev,
                        const int32_t *slot_of_dev, int64_t P, const int64_t *tile_rows_dev, int64_t qt,
                        const int64_t *row_ids_dev, int64_t id_base, int64_t k, float *out_dist_dev,
                        int64_t *out_id_dev, int flags, void *stream);

/* The candidate lists of every query as ONE dense id row: out_ids[b][0..R) = its P lists back to
 * back (id_base + row_ids[row]), padded with -1 -- the input of annlite_exact_gather_dist when the
 * index keeps the float vectors (re-rank).  Overflowed lists contribute
 * nothing and entries beyond R are dropped (the re-rank candidate set is a heuristic). */
ANNLITE_API int annlite_ivf_candidate_ids(const uint32_t *cand_dev, int64_t cand_cap, const uint32_t *cand_count_dev,
                              const int32_t *slot_of_dev, int64_t B, int64_t P, const int64_t *row_ids_dev,
                              int64_t id_base, int64_t *out_ids_dev, int64_t R, void *stream);

/* (round 6) The pruned search over cells on the BYTE-TABLE kernel, one call: annlite_ivf_plan (tiles of 32 slots that probe one
 * cell each) -> ONE preparation launch (L2 tables of the B real queries, quantisation parameters, every query's first bound from
 * rows of its NEAREST cell -- a bound from rows outside the probed cells would be wrong --, its byte table) -> the scan in cell
 * tiles (exact ascending-m fp32 sums, space_pq.h:32-35 / pq_bindings.pyx:44-45, of the rows that pass the byte filter; a 16-key
 * list per slot; the bounds shared between the tiles of a QUERY's cells) -> annlite_ivf_merge_lists.  Result: for every query the
 * exact top-k, under the fixed order (distance asc, id asc), of the rows of its P probed cells -- bit-equal to
 * annlite_pq_search_tiles + annlite_ivf_rescore and to the oracle's ivf_search.
 *   cells_dev      i32 [B][P]  the probed cells, nearest first (annlite_ivf_select_cells)
 *   codes_dev      u8 [N][16]  the CELL-SORTED table (cell c = rows [cell_rows[c][0], cell_rows[c][1]), begin a multiple of 64), in
 *                  codes_layout; valid_bits_dev (may be NULL): bitmap over TABLE rows; row_ids_dev i64 [N] (may be NULL): external id
 *                  of every table row, ascending inside a cell
 *   cell_order_dev i32 [C]     cells by descending size (the plan's tile order)
 *   lut_kind: ANNLITE_LUT_L2 (pq_bindings.pyx:149-210) or ANNLITE_LUT_IPDIST (float32(1 / Ks) - <q_sub, codeword>, pq_bindings.pyx:214-274 +
 *   pq.py:316-322: INNER_PRODUCT, and COSINE on normalised queries) -- the tables are built inside the preparation launch, the reference's
 *   j-ascending fmaf chains.  flags: ANNLITE_FLAG_SQRT.  Serves M = 16, Ks <= 256, k <= 16, D <= 256, sub-vectors of a multiple of 4
 *   floats; other shapes: annlite_pq_search_tiles + annlite_ivf_rescore.
 * replaces: _cell_selection's consumers -- the per-cell search loop and the hstack / argsort merge of CellContainer.ivf_search
 * (annlite/container.py:88-144) with n_probe < n_cells. */
ANNLITE_API int annlite_ivf_search_topk_workspace_bytes(int64_t B, int64_t P, int64_t C, int64_t M, int64_t Ks, int64_t k,
                                            int64_t *bytes);
ANNLITE_API int annlite_ivf_search_topk(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev, int64_t M,
                            int64_t Ks, const void *codes_dev, int codes_layout, int64_t N, const uint32_t *valid_bits_dev,
                            const int32_t *cells_dev, int64_t P, int64_t C, const int64_t *cell_rows_dev,
                            const int32_t *cell_order_dev, const int64_t *row_ids_dev, int64_t id_base, int64_t k,
                            float *out_dist_dev, int64_t *out_id_dev, int flags, void *workspace_dev, size_t workspace_bytes,
                            void *stream);

/* The same pipeline as the candidate generator of an exact re-rank (IvfPQGpuIndex(rerank=True)): the slots keep PRIVATE lists -- nothing is
 * shared between a query's tiles --, so every (query, probed cell) list is a function of its own cell: the cell's best <= k rows, by
 * exact ADC sum (ties by id), among those at or below the query's first bound (the min(bound_rank * k, 64)-th smallest exact sum of seed
 * rows of its nearest cell: at or above the k-th key of the probed rows for every bound_rank >= 1, so the union of a query's lists holds
 * its exact ADC top-k; the nearest cell's list is never cut).  bound_rank 1 .. 64 trades pool size for time: a looser bound lengthens
 * the far cells' lists (10M rows, 16 of 256 cells, k = 16: rank 1 / 2 / 4 -> 2.44 / 1.72 / 1.21 M q/s, re-ranked recall@10 0.803 / 0.813 /
 * 0.813).  seed_cells_dev (may be NULL) i32 [B]: the entry of the cell table whose rows seed query b's bound instead of cells[b][0] --
 * for a cell table that lists a cell's rows ALSO as parts (IvfPQGpuIndex.rerank_split): the probe names the parts, the seed the whole
 * cell, whose rows must all be probed by that query; the first list may then be cut like the others.  (Entries of the cell table may
 * overlap -- a cell and its parts -- as long as the entries ONE query probes are disjoint row ranges: a row then appears in one list.)
 * out_ids_dev i64 [B][P * k]: list (b, p) at [p * k, (p + 1) * k), ascending by (sum, id), -1 where it is shorter -- the input of
 * annlite_rerank_topk.  Workspace and shapes: annlite_ivf_search_topk's.  (Not in the reference: its cells hold exact vectors or PQ
 * codes, never both.) */
ANNLITE_API int annlite_ivf_search_candidates(int lut_kind, const float *queries_dev, int64_t B, int64_t D, const float *codebooks_dev,
                                  int64_t M, int64_t Ks, const void *codes_dev, int codes_layout, int64_t N,
                                  const uint32_t *valid_bits_dev, const int32_t *cells_dev, int64_t P, int64_t C,
                                  const int64_t *cell_rows_dev, const int32_t *cell_order_dev, const int64_t *row_ids_dev,
                                  int64_t id_base, int64_t k, int64_t bound_rank, const int32_t *seed_cells_dev,
                                  int64_t *out_ids_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* Merge of per-slot lists: lists_dev u64 [V][k] (ordered distance << 32 | table row, ascending, ~0 = none: what the cell-tile scan
 * leaves per slot) -> per query the k smallest of its P slots' keys re-keyed by external id (id_base + row_ids[row]); one wave per
 * query.  replaces: the concatenate + sort at the end of CellContainer.ivf_search (annlite/container.py:131-144). */
ANNLITE_API int annlite_ivf_merge_lists(const uint64_t *lists_dev, int64_t k, const int32_t *slot_of_dev, int64_t B, int64_t P,
                            const int64_t *row_ids_dev, int64_t id_base, float *out_dist_dev, int64_t *out_id_dev, int flags,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANNLITE_HIP_H_ */
