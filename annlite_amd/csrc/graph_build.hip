// graph_build.hip -- the level-0 graph of BASELINE config 5 (HNSW over PQ codes) BUILT ON THE GPU, in batches (round 6).
//
// The reference builds its graph on the host, one point at a time (include/hnswlib/hnswalg.h:1108-1235 addPoint: greedy descent,
// searchBaseLayer with ef_construction, getNeighborsByHeuristic2 378-429, mutuallyConnectNewElement 431-553); libannlite_graph.so
// (hnsw_host.cpp) restates that with OpenMP threads -- 56-65 s for 5M rows on a 16-core host, a third of a bench run.  Here the
// insertion of a BATCH of points is three launches on top of the walk kernel the searches use anyway:
//
//   1. candidates: graph_beam_search_kernel (graph.hip, packed records, pair walk) with ef = ef_construction over the graph AS IT
//      IS -- the points of a batch do not see each other (they meet through the common neighbours both link to, and through later
//      batches' reverse links: batches are a small fraction of the graph, 16384 points or a quarter of it while it is small);
//   2. graph_select_kernel: Algorithm 4 of the HNSW paper / hnsw_host.cpp select_neighbors -- the candidates in the order of their
//      search distance, one kept only if it is closer to the new point than to every neighbour kept before it; the triangle
//      comparisons use the symmetric code-to-code L2 table `sdc` [M][Ks][Ks] (see hnsw_host.cpp: the asymmetric PQ distance is not
//      a metric) -- writes the new point's link list and emits (target, source) pairs;
//   3. graph_reverse_kernel: the pairs, SORTED BY TARGET on the device (one torch sort per batch), one wave per target: append the
//      sources while the list has room, else shrink (list + sources) with the same heuristic to links_per_node entries -- what
//      mutuallyConnectNewElement does under a per-node lock, here without locks: a target's incoming links of a batch arrive together;
//   + graph_pack_nodes_kernel: the packed records (graph.hip) of the nodes whose lists changed.
//
// Only level 0 exists: the GPU walk never descends the upper layers (it scans a seed sample flat, graph.hip), so none are built.
// One wave per work item; the pool of a work item (<= 256 candidates: ids, code rows, distances to the base point) lives in LDS,
// the kept neighbours' code rows in the lanes' registers (lane j = j-th kept neighbour), a candidate's test against all of them is
// one round of M gathers per lane from the 4 MB `sdc` table (L2 / MALL resident).
#include "scan_common.h"

namespace annlite {

constexpr int kPoolMax = 256;

template <int M>
struct CodeRow {
    uint32_t w[M / 4];
};

template <int M>
__device__ __forceinline__ CodeRow<M> load_code(const uint8_t *__restrict__ codes, uint32_t node) {
    CodeRow<M> r;
    const uint32_t *p = (const uint32_t *)(codes + (int64_t)node * M);
#pragma unroll
    for (int i = 0; i < M / 4; ++i) r.w[i] = p[i];
    return r;
}

// symmetric distance between two stored rows: sum over the sub-spaces of sdc[m][a_m][b_m]
template <int M>
__device__ __forceinline__ float sym_dist(const float *__restrict__ sdc, int Ks, const CodeRow<M> &a, const CodeRow<M> &b) {
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint32_t ca = (a.w[m / 4] >> (8 * (m % 4))) & 0xffu, cb = (b.w[m / 4] >> (8 * (m % 4))) & 0xffu;
        r += sdc[((int64_t)m * Ks + ca) * Ks + cb];
    }
    return r;
}

// Algorithm 4 over a pool that is ALREADY in the order it is to be visited (LDS: s_id / s_code / s_tb = id, code row, distance to
// the base point of entry t; id 0xffffffff = skip).  Lane j < n_kept holds the j-th kept neighbour.  Wave-uniform control flow.
template <int M>
__device__ __forceinline__ void heuristic_select(const float *__restrict__ sdc, int Ks, const uint32_t *s_id, const uint32_t *s_code,
                                                 const float *s_tb, int n_pool, int m_max, int lane, uint32_t &kept_id, int &n_kept) {
    CodeRow<M> kc;
#pragma unroll
    for (int i = 0; i < M / 4; ++i) kc.w[i] = 0u;
    kept_id = 0xffffffffu;
    n_kept = 0;
    for (int t = 0; t < n_pool && n_kept < m_max; ++t) {
        const uint32_t id = s_id[t];
        if (id == 0xffffffffu) continue;
        CodeRow<M> ct;
#pragma unroll
        for (int i = 0; i < M / 4; ++i) ct.w[i] = s_code[t * (M / 4) + i];
        const float tb = s_tb[t];
        bool closer_to_kept = false;
        if (lane < n_kept) closer_to_kept = sym_dist<M>(sdc, Ks, kc, ct) < tb;
        if (__ballot(closer_to_kept) == 0ull) {
            if (lane == n_kept) {
                kc = ct;
                kept_id = id;
            }
            ++n_kept;
        }
    }
}

// ---- 2. the new points' own lists ---------------------------------------------------------------------------------------------
// cand i64 [b][ef]: the walk's list for point i (ascending search distance, -1 = none); the point is node base0 + i.
// links u32 [.][lpn + 1]: row base0 + i <- (count, ids); pairs i64 [b][m_keep] <- (target << 32) | source, INT64_MAX = none.
template <int M, int WPB>
__global__ __launch_bounds__(WPB * 64) void graph_select_kernel(const int64_t *__restrict__ cand, int ef, int64_t b, int64_t base0,
                                                          const uint8_t *__restrict__ codes, const float *__restrict__ sdc, int Ks,
                                                          int m_keep, uint32_t *__restrict__ links, int lpn, int64_t *__restrict__ pairs) {
    constexpr int CW = M / 4;
    __shared__ uint32_t s_all[WPB][kPoolMax * (2 + CW)];  // (M = 64: 18 KB per wave -- two waves per workgroup stay under 64 KB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * WPB + wave;
    if (i >= b) return;
    uint32_t *s_id = s_all[wave], *s_code = s_id + kPoolMax;
    float *s_tb = (float *)(s_code + kPoolMax * CW);
    const uint32_t base = (uint32_t)(base0 + i);
    const CodeRow<M> bc = load_code<M>(codes, base);
    int n_real = 0;
    for (int t0 = 0; t0 < ef; t0 += 64) {
        const int t = t0 + lane;
        const int64_t id = t < ef ? cand[i * ef + t] : -1;
        const bool ok = id >= 0 && (uint32_t)id != base;
        if (t < ef) {
            s_id[t] = ok ? (uint32_t)id : 0xffffffffu;
            if (ok) {
                const CodeRow<M> c = load_code<M>(codes, (uint32_t)id);
#pragma unroll
                for (int j = 0; j < CW; ++j) s_code[t * CW + j] = c.w[j];
                s_tb[t] = sym_dist<M>(sdc, Ks, bc, c);
            }
        }
        n_real += __popcll(__ballot(ok));
    }
    asm volatile("" ::: "memory");  // (lanes read each other's LDS stores below: one wave, program order)
    uint32_t kept_id = 0xffffffffu;
    int n_kept = 0;
    if (n_real <= m_keep) {  // (hnsw_host.cpp select_neighbors: a pool that fits is kept whole)
        // compact the real entries into the lanes in list order
        int basecnt = 0;
        for (int t0 = 0; t0 < ef; t0 += 64) {
            const int t = t0 + lane;
            const uint32_t id = t < ef ? s_id[t] : 0xffffffffu;
            const unsigned long long m = __ballot(id != 0xffffffffu);
            const int mypos = basecnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (id != 0xffffffffu) s_tb[mypos] = __uint_as_float(id);  // (s_tb is free now: the ids, compacted)
            basecnt += __popcll(m);
        }
        asm volatile("" ::: "memory");
        n_kept = n_real;
        if (lane < n_kept) kept_id = __float_as_uint(s_tb[lane]);
    } else {
        heuristic_select<M>(sdc, Ks, s_id, s_code, s_tb, ef, m_keep, lane, kept_id, n_kept);
    }
    uint32_t *ll = links + (int64_t)base * (lpn + 1);
    if (lane == 0) ll[0] = (uint32_t)n_kept;
    if (lane < lpn) ll[1 + lane] = lane < n_kept ? kept_id : 0u;
    if (lane < m_keep)
        pairs[i * m_keep + lane] = lane < n_kept ? (int64_t)(((uint64_t)kept_id << 32) | (uint64_t)base) : (int64_t)0x7fffffffffffffffll;
}

// ---- 3. the reverse links -----------------------------------------------------------------------------------------------------
// keys i64 [P] ascending: (target << 32) | source; seg i64 [S + 1]: segment s = keys[seg[s] .. seg[s+1]) all share one target.
template <int M>
__global__ __launch_bounds__(256) void graph_reverse_kernel(const int64_t *__restrict__ keys, const int64_t *__restrict__ seg, int64_t S,
                                                           const uint8_t *__restrict__ codes, const float *__restrict__ sdc, int Ks,
                                                           uint32_t *__restrict__ links, int lpn) {
    constexpr int CW = M / 4;
    __shared__ uint32_t s_all[4][64 * (2 + CW)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= S) return;
    uint32_t *s_id = s_all[wave], *s_code = s_id + 64;
    float *s_tb = (float *)(s_code + 64 * CW);
    const int64_t k0 = seg[s], k1 = seg[s + 1];
    const uint32_t o = (uint32_t)((uint64_t)keys[k0] >> 32);
    uint32_t *ll = links + (int64_t)o * (lpn + 1);
    const int c = (int)ll[0];  // (<= lpn <= 32)
    // pool: lanes [0, c) the list, lanes [c, c + k) the sources (as many as the 64 lanes hold), duplicates of the list dropped
    int k = (int)((k1 - k0) < (int64_t)(64 - c) ? (k1 - k0) : (int64_t)(64 - c));
    uint32_t x = 0xffffffffu;
    if (lane < c) x = ll[1 + lane];
    else if (lane < c + k) x = (uint32_t)((uint64_t)keys[k0 + (lane - c)] & 0xffffffffull);
    bool dup = x == o;
    for (int j = 0; j < c; ++j) {
        const uint32_t e = __builtin_amdgcn_readlane(x, j);
        dup = dup || (lane >= c && lane < c + k && x == e);
    }
    if (dup && lane >= c) x = 0xffffffffu;
    // compact the surviving sources behind the list
    const unsigned long long live = __ballot(x != 0xffffffffu);
    const int n_pool = __popcll(live);
    if (n_pool == c) return;  // (nothing new)
    if (n_pool <= lpn) {
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        if (x != 0xffffffffu && lane >= c) ll[1 + pos] = x;
        if (lane == 0) ll[0] = (uint32_t)n_pool;
        return;
    }
    // shrink: the pool ordered by its symmetric distance to the target (ties: id), then the heuristic to lpn entries
    const CodeRow<M> oc = load_code<M>(codes, o);
    CodeRow<M> xc;
#pragma unroll
    for (int j = 0; j < CW; ++j) xc.w[j] = 0u;
    float d = __builtin_inff();
    if (x != 0xffffffffu) {
        xc = load_code<M>(codes, x);
        d = sym_dist<M>(sdc, Ks, oc, xc);
    }
    const uint32_t dk = x != 0xffffffffu ? f32_to_ordered(d) : 0xffffffffu;
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
        const uint32_t ek = __builtin_amdgcn_readlane(dk, j), ex = __builtin_amdgcn_readlane(x, j);
        rank += (ek < dk || (ek == dk && (ex < x || (ex == x && j < lane)))) ? 1 : 0;
    }
    s_id[rank] = x;  // (dead lanes: key and id 0xffffffff rank last, among themselves by lane)
    s_tb[rank] = d;
#pragma unroll
    for (int j = 0; j < CW; ++j) s_code[rank * CW + j] = xc.w[j];
    asm volatile("" ::: "memory");
    uint32_t kept_id = 0xffffffffu;
    int n_kept = 0;
    heuristic_select<M>(sdc, Ks, s_id, s_code, s_tb, n_pool, lpn, lane, kept_id, n_kept);
    if (lane < lpn) ll[1 + lane] = lane < n_kept ? kept_id : 0u;
    if (lane == 0) ll[0] = (uint32_t)n_kept;
}

// ---- packed records of the nodes in a list (graph.hip graph_pack_kernel, with an indirection) ---------------------------------
__global__ __launch_bounds__(256) void graph_pack_nodes_kernel(const uint32_t *__restrict__ links, int L, const uint8_t *__restrict__ codes,
                                                              int64_t N, int M, const int64_t *__restrict__ nodes, int64_t n_nodes,
                                                              uint8_t *__restrict__ out, int64_t stride) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes * L) return;
    const int64_t node = nodes[t / L];
    const int j = (int)(t % L);
    if (node < 0 || node >= N) return;
    const uint32_t *ll = links + node * (L + 1);
    const uint32_t cnt = ll[0];
    const uint32_t nb = (uint32_t)j < cnt ? ll[1 + j] : 0xffffffffu;
    const bool ok = (uint32_t)j < cnt && (int64_t)nb < N;
    uint8_t *r = out + node * stride;
    uint32_t *dst = (uint32_t *)(r + (int64_t)j * M);
    const uint32_t *src = (const uint32_t *)(codes + (int64_t)(ok ? nb : 0u) * M);
    for (int i = 0; i < M / 4; ++i) dst[i] = ok ? src[i] : 0u;
    uint32_t *hdr = (uint32_t *)(r + (int64_t)L * M);
    hdr[j] = nb;
    if (j == 0) {
        hdr[L] = cnt;
        for (int64_t bb = (int64_t)L * M + 4 * L + 4; bb + 4 <= stride; bb += 4) *(uint32_t *)(r + bb) = 0u;
    }
}

// sdc[m][a][b] = sum_j (C[m][a][j] - C[m][b][j])^2  (hnsw_host.cpp build_sdc)
__global__ __launch_bounds__(256) void graph_sdc_kernel(const float *__restrict__ cb, int M, int Ks, int dsub, float *__restrict__ sdc) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * Ks * Ks) return;
    const int b = (int)(t % Ks), a = (int)((t / Ks) % Ks), m = (int)(t / ((int64_t)Ks * Ks));
    const float *pa = cb + ((int64_t)m * Ks + a) * dsub, *pb = cb + ((int64_t)m * Ks + b) * dsub;
    float acc = 0.f;
    for (int j = 0; j < dsub; ++j) acc += (pa[j] - pb[j]) * (pa[j] - pb[j]);
    sdc[t] = acc;
}

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_graph_build_sdc(const float *codebooks_dev, int64_t M, int64_t Ks, int64_t dsub, float *sdc_dev, void *stream) {
    ANNLITE_REQUIRE(codebooks_dev && sdc_dev && M >= 1 && Ks >= 1 && Ks <= 256 && dsub >= 1, "bad arguments (Ks <= 256)");
    const int64_t total = M * Ks * Ks;
    hipLaunchKernelGGL(graph_sdc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codebooks_dev, (int)M,
                       (int)Ks, (int)dsub, sdc_dev);
    return launch_status("graph_sdc_kernel");
}

extern "C" int annlite_graph_build_select(const int64_t *cand_dev, int ef, int64_t b, int64_t base0, const void *codes_dev, int64_t M,
                                          int64_t Ks, const float *sdc_dev, int max_keep, uint32_t *links_dev, int links_per_node,
                                          int64_t *pairs_dev, void *stream) {
    ANNLITE_REQUIRE(b >= 0 && base0 >= 0 && ef >= 1 && ef <= kPoolMax, "bad b=%lld base0=%lld ef=%d (ef <= 256)", (long long)b,
                    (long long)base0, ef);
    ANNLITE_REQUIRE((M == 8 || M == 16 || M == 32 || M == 64) && Ks >= 1 && Ks <= 256, "graph build supports M in {8,16,32,64}, Ks <= 256");
    ANNLITE_REQUIRE(max_keep >= 1 && max_keep <= links_per_node && links_per_node <= 64, "bad max_keep=%d links_per_node=%d", max_keep,
                    links_per_node);
    if (b == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(cand_dev && codes_dev && sdc_dev && links_dev && pairs_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
#define ANNLITE_SEL(MM, WPB) \
    hipLaunchKernelGGL((graph_select_kernel<MM, WPB>), dim3((unsigned)((b + WPB - 1) / WPB)), dim3(WPB * 64), 0, st, cand_dev, ef, b, base0, \
                       (const uint8_t *)codes_dev, sdc_dev, (int)Ks, max_keep, links_dev, links_per_node, pairs_dev)
    if (M == 8) ANNLITE_SEL(8, 4);
    else if (M == 16) ANNLITE_SEL(16, 4);
    else if (M == 32) ANNLITE_SEL(32, 4);
    else ANNLITE_SEL(64, 2);
#undef ANNLITE_SEL
    return launch_status("graph_select_kernel");
}

extern "C" int annlite_graph_build_reverse(const int64_t *keys_dev, const int64_t *seg_dev, int64_t n_segments, const void *codes_dev,
                                           int64_t M, int64_t Ks, const float *sdc_dev, uint32_t *links_dev, int links_per_node,
                                           void *stream) {
    ANNLITE_REQUIRE(n_segments >= 0, "bad n_segments");
    ANNLITE_REQUIRE((M == 8 || M == 16 || M == 32 || M == 64) && Ks >= 1 && Ks <= 256, "graph build supports M in {8,16,32,64}, Ks <= 256");
    ANNLITE_REQUIRE(links_per_node >= 1 && links_per_node <= 32, "reverse links: links_per_node in [1, 32] (%d)", links_per_node);
    if (n_segments == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(keys_dev && seg_dev && codes_dev && sdc_dev && links_dev, "null device pointer");
    const dim3 grid((unsigned)((n_segments + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define ANNLITE_REV(MM) \
    hipLaunchKernelGGL(graph_reverse_kernel<MM>, grid, block, 0, st, keys_dev, seg_dev, n_segments, (const uint8_t *)codes_dev, sdc_dev, \
                       (int)Ks, links_dev, links_per_node)
    if (M == 8) ANNLITE_REV(8);
    else if (M == 16) ANNLITE_REV(16);
    else if (M == 32) ANNLITE_REV(32);
    else ANNLITE_REV(64);
#undef ANNLITE_REV
    return launch_status("graph_reverse_kernel");
}

extern "C" int annlite_graph_pack_nodes(const uint32_t *links_dev, int links_per_node, const void *codes_dev, int64_t N, int64_t M,
                                        const int64_t *nodes_dev, int64_t n_nodes, void *packed_dev, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && n_nodes >= 0 && links_per_node >= 1 && links_per_node <= 64 && (M == 8 || M == 16 || M == 32 || M == 64),
                    "packed records: links_per_node in [1, 64], M in {8,16,32,64}");
    if (N == 0 || n_nodes == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(links_dev && codes_dev && nodes_dev && packed_dev, "null device pointer");
    int64_t stride = 0;
    if (int rc = annlite_graph_record_bytes(links_per_node, M, &stride)) return rc;
    const int64_t total = n_nodes * links_per_node;
    hipLaunchKernelGGL(graph_pack_nodes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, links_dev,
                       links_per_node, (const uint8_t *)codes_dev, N, (int)M, nodes_dev, n_nodes, (uint8_t *)packed_dev, stride);
    return launch_status("graph_pack_nodes_kernel");
}
