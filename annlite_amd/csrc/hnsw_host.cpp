// hnsw_host.cpp -- libannlite_graph.so: HNSW over PQ codes with the asymmetric PQ distance (host code).
//
// Written from the published algorithm (Malkov & Yashunin, "Efficient and robust approximate nearest
// neighbor search using Hierarchical Navigable Small World graphs", Algorithms 1-5) with the parameter
// conventions of the reference's HnswIndex (annlite/core/index/hnsw/index.py:60-100: max_connection M,
// 2M links on level 0, ef_construction, ef_search; level = floor(-ln(U) / ln(M))).  Edge-walk distances
// have the form of the reference's hnswlib::PQLookup (include/hnswlib/space_pq.h:15-37): sum over
// sub-spaces of the look-up table entry selected by the stored code byte, fp32, sub-space order; the
// table of a point being inserted / of a query is built here with the reference's L2 fmaf chain
// (bindings/pq_bindings.pyx:149-210), so no table crosses PCIe.  Neighbour diversification (Algorithm 4)
// needs distances between STORED points; those use a symmetric code-to-code L2 table [M][Ks][Ks].
//
// The graph is built AND walked in L2 geometry for every metric of the index.  Walking with the
// reference's inner-product tables (1/Ks - dot, not a metric: the large-norm reconstructions win every
// insertion search, low-norm points end up without in-links) left half of 20k clustered unit vectors
// unreachable here (candidate recall 0.48 for any ef; the reference's own PQ space: 0.53 at 8k points,
// tests/test_graph_host.py).  For cosine indexes (unit rows) L2 and cosine neighbourhoods coincide, and
// the distances that are RETURNED never come from the graph: the GPU re-evaluates the candidates with
// the metric's own tables (annlite_adc_gather) or exactly (annlite_exact_gather_dist).
#include "../../include/annlite_graph.h"

#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <vector>

namespace {

thread_local char g_err[512];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

typedef std::pair<float, uint32_t> Cand;  // (distance, node)

// threads worth starting: the CPUs this process may run on, capped by a cgroup CPU quota (a container that
// reports 256 CPUs but is throttled to a few makes 256 spinning builders crawl)
int usable_threads() {
    int n = omp_get_max_threads();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            const long long q = atoll(quota);
            if (q > 0) n = std::min<long long>(n, std::max<long long>(1, (q + period - 1) / period));
        }
        fclose(f);
    }
    return std::max(1, n);
}

struct SpinLock {
    std::atomic_flag f = ATOMIC_FLAG_INIT;
    void lock() {
        while (f.test_and_set(std::memory_order_acquire)) {
        }
    }
    void unlock() { f.clear(std::memory_order_release); }
};

}  // namespace

struct annlite_hnsw {
    int64_t M = 0, Ks = 0, dsub = 0, D = 0;
    std::vector<float> cb;   // [M][Ks][dsub]
    std::vector<float> cbT;  // [M][dsub][Ks] (table builds vectorise over the codewords)
    std::vector<float> sdc;  // [M][Ks][Ks] code-to-code distances (diversification heuristic)
    int64_t cap = 0;
    std::atomic<int64_t> n{0};
    int Mc = 16, M0 = 32, efc = 200;
    double mult = 0;
    std::vector<uint8_t> codes;                   // [cap][M]
    std::vector<int32_t> level;                   // [cap], -1 = empty slot
    std::vector<uint32_t> link0;                  // [cap][M0 + 1]: count, ids
    std::vector<std::vector<uint32_t>> linkU;     // per node: level l >= 1 at [(l-1)*(Mc+1)]: count, ids
    std::unique_ptr<SpinLock[]> locks;
    std::mutex global;
    int64_t enter = -1;
    int maxlevel = -1;
    std::vector<uint8_t> deleted;
    std::mt19937_64 rng;

    // ---- tables ---------------------------------------------------------------------------------
    // the walking table of a point: L2 between its sub-vectors and the centroids, the reference's fmaf chain
    // (bindings/pq_bindings.pyx:149-210)
    void build_lut(const float *x, float *lut) const {  // [M][Ks]
        // per entry the chain acc = fma(c_j - q_j, c_j - q_j, acc) over j, as in the reference; the loop over the
        // Ks codewords is the vector dimension (codebooks transposed once to [M][dsub][Ks]): 8 chains per AVX2 op
        for (int64_t m = 0; m < M; ++m) {
            const float *q = x + m * dsub;
            float *acc = lut + m * Ks;
            for (int64_t k = 0; k < Ks; ++k) acc[k] = 0.f;
            for (int64_t j = 0; j < dsub; ++j) {
                const float qj = q[j];
                const float *c = cbT.data() + (m * dsub + j) * Ks;
#pragma omp simd
                for (int64_t k = 0; k < Ks; ++k) {
                    const float d = c[k] - qj;
                    acc[k] = __builtin_fmaf(d, d, acc[k]);
                }
            }
        }
    }
    void build_cbT() {
        cbT.resize(cb.size());
        for (int64_t m = 0; m < M; ++m)
            for (int64_t k = 0; k < Ks; ++k)
                for (int64_t j = 0; j < dsub; ++j) cbT[(m * dsub + j) * Ks + k] = cb[(m * Ks + k) * dsub + j];
    }
    void build_sdc() {
        sdc.resize((size_t)M * Ks * Ks);
#pragma omp parallel for schedule(static)
        for (int64_t mk = 0; mk < M * Ks; ++mk) {
            const int64_t m = mk / Ks;
            const float *a = cb.data() + mk * dsub;
            for (int64_t k2 = 0; k2 < Ks; ++k2) {
                const float *b = cb.data() + (m * Ks + k2) * dsub;
                float acc = 0.f;
                for (int64_t j = 0; j < dsub; ++j) acc += (a[j] - b[j]) * (a[j] - b[j]);
                sdc[(size_t)mk * Ks + k2] = acc;  // always L2: the heuristic needs a metric (see select_neighbors)
            }
        }
    }
    inline float adc(const float *lut, uint32_t node) const {  // PQLookup
        const uint8_t *c = codes.data() + (size_t)node * M;
        float r = 0.f;
        for (int64_t m = 0; m < M; ++m) r += lut[m * Ks + c[m]];
        return r;
    }
    // the same sums for several nodes at once: a single sum is a chain of M dependent adds (latency-bound); four
    // independent chains fill the FP pipes.  Order per node unchanged (ascending m): identical bits.
    inline void adc_many(const float *lut, const uint32_t *nodes, size_t n, float *out) const {
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
            const uint8_t *c0 = codes.data() + (size_t)nodes[i] * M, *c1 = codes.data() + (size_t)nodes[i + 1] * M;
            const uint8_t *c2 = codes.data() + (size_t)nodes[i + 2] * M, *c3 = codes.data() + (size_t)nodes[i + 3] * M;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            for (int64_t m = 0; m < M; ++m) {
                const float *row = lut + m * Ks;
                r0 += row[c0[m]];
                r1 += row[c1[m]];
                r2 += row[c2[m]];
                r3 += row[c3[m]];
            }
            out[i] = r0, out[i + 1] = r1, out[i + 2] = r2, out[i + 3] = r3;
        }
        for (; i < n; ++i) out[i] = adc(lut, nodes[i]);
    }
    // is sym(a, b) < bound ?  All terms are >= 0 and fp32 sums of non-negative terms never decrease, so the sum can be
    // abandoned as soon as a partial sum reaches the bound: same answer as the full sum, most pairs stop after a few terms
    inline bool sym_below(uint32_t a, uint32_t b, float bound) const {
        const uint8_t *ca = codes.data() + (size_t)a * M, *cbp = codes.data() + (size_t)b * M;
        float r = 0.f;
        for (int64_t m = 0; m < M; ++m) {
            r += sdc[((size_t)m * Ks + ca[m]) * Ks + cbp[m]];
            if (!(r < bound)) return false;
        }
        return true;
    }
    inline float sym(uint32_t a, uint32_t b) const {
        const uint8_t *ca = codes.data() + (size_t)a * M, *cbp = codes.data() + (size_t)b * M;
        float r = 0.f;
        for (int64_t m = 0; m < M; ++m) r += sdc[((size_t)m * Ks + ca[m]) * Ks + cbp[m]];
        return r;
    }

    // ---- link lists -------------------------------------------------------------------------------
    uint32_t *links(uint32_t node, int lv) {
        return lv == 0 ? link0.data() + (size_t)node * (M0 + 1) : linkU[node].data() + (size_t)(lv - 1) * (Mc + 1);
    }
    const uint32_t *links(uint32_t node, int lv) const {
        return lv == 0 ? link0.data() + (size_t)node * (M0 + 1) : linkU[node].data() + (size_t)(lv - 1) * (Mc + 1);
    }

    struct Visited {
        std::vector<uint32_t> mark;
        uint32_t epoch = 0;
        void begin(size_t n) {
            if (mark.size() < n) mark.assign(n, 0), epoch = 0;
            if (++epoch == 0) {
                std::fill(mark.begin(), mark.end(), 0);
                epoch = 1;
            }
        }
        bool test_set(uint32_t i) {
            if (mark[i] == epoch) return true;
            mark[i] = epoch;
            return false;
        }
    };

    // Algorithm 2: beam search on one layer; returns up to ef (distance, node), ascending.
    // The ef best nodes live in ONE array sorted by (distance, node) with an "expanded" flag per entry, and the
    // next node to expand is the first unflagged entry (a cursor that only moves back when something is inserted in
    // front of it) -- the paper's two heaps hold the same sets: an entry pushed out of the ef best would never be
    // expanded there either (it is farther than the ef-th best).  A heap push + pop per accepted neighbour was 60 %
    // of the build's cycles (3350 evaluations and 214 expansions per insertion at 1M rows); an ordered insert is one
    // binary search and one short memmove.
    void search_layer(const float *lut, uint32_t ep, float ep_d, int ef, int lv, Visited &vis, std::vector<Cand> &out,
                      bool locked) const {
        constexpr uint32_t kExpanded = 0x80000000u;  // (node ids < 2^31)
        std::vector<Cand> &best = out;               // sorted ascending; .second carries the flag
        best.clear();
        best.reserve((size_t)ef + 1);
        vis.begin((size_t)cap);
        vis.test_set(ep);
        best.emplace_back(ep_d, ep);
        size_t cur = 0;
        std::vector<uint32_t> nb;
        std::vector<float> dist;
        for (;;) {
            while (cur < best.size() && (best[cur].second & kExpanded)) ++cur;
            if (cur == best.size()) break;
            const uint32_t node = best[cur].second;
            best[cur].second |= kExpanded;
            // the link list of the likely NEXT node (first unflagged entry behind this one): a DRAM miss per expansion
            // at millions of rows (links alone are 132 MB per million), started now, it returns while this node's
            // neighbours are scored
            for (size_t nx = cur + 1; nx < best.size(); ++nx)
                if (!(best[nx].second & kExpanded)) {
                    const uint32_t *pl = links(best[nx].second, lv);
                    __builtin_prefetch(pl);
                    __builtin_prefetch(pl + 16);
                    break;
                }
            {
                const uint32_t *ll = links(node, lv);
                if (locked) {
                    SpinLock &l = const_cast<SpinLock &>(locks[node]);
                    l.lock();
                    nb.assign(ll + 1, ll + 1 + ll[0]);
                    l.unlock();
                } else {
                    nb.assign(ll + 1, ll + 1 + ll[0]);
                }
            }
            // two passes so that the random accesses overlap: visited marks first, then the code rows of the
            // unvisited neighbours (a 5M-row graph: 20 MB of marks, 80 MB of codes -- every access a cache miss)
            for (uint32_t v : nb) __builtin_prefetch(vis.mark.data() + v);
            size_t n_new = 0;
            for (uint32_t v : nb) {
                if (vis.test_set(v)) continue;
                __builtin_prefetch(codes.data() + (size_t)v * M);
                nb[n_new++] = v;
            }
            dist.resize(n_new);
            adc_many(lut, nb.data(), n_new, dist.data());
            for (size_t i = 0; i < n_new; ++i) {
                const Cand c(dist[i], nb[i]);
                if ((int)best.size() >= ef) {
                    const Cand &w = best.back();
                    if (!(c.first < w.first || (c.first == w.first && c.second < (w.second & ~kExpanded)))) continue;
                    best.pop_back();
                }
                // position by (distance, node); flags do not take part in the order
                size_t lo = 0, hi = best.size();
                while (lo < hi) {
                    const size_t mid = (lo + hi) >> 1;
                    const Cand &e = best[mid];
                    if (e.first < c.first || (e.first == c.first && (e.second & ~kExpanded) < c.second)) lo = mid + 1;
                    else hi = mid;
                }
                best.insert(best.begin() + (std::ptrdiff_t)lo, c);
                if (lo < cur) cur = lo;
            }
        }
        for (Cand &c : best) c.second &= ~kExpanded;
    }

    // Algorithm 4: keep a candidate only if it is closer to the base point than to every kept neighbour.
    // Candidates are visited in the order of their search distance (asymmetric table of the base point, any
    // kind); the triangle comparisons themselves use the symmetric L2 distance between the STORED codes of
    // base, kept and candidate -- a metric for every table kind (1/Ks - dot is none: a point is not at
    // distance 0 from itself; keeping just the closest M, which is what the reference's PQ space effectively
    // does -- hnswalg.h:465-473 evaluates PQLookup(table of the inserted point, candidate) on both sides --
    // left clustered data with recall 0.48 at ef 128 here, this rule 0.9+).
    void select_neighbors(uint32_t base, std::vector<Cand> &cands, int Mmax) const {
        if ((int)cands.size() <= Mmax) return;
        std::sort(cands.begin(), cands.end());
        std::vector<Cand> keep;
        keep.reserve(Mmax);
        for (const Cand &c : cands) {
            if ((int)keep.size() >= Mmax) break;
            const float to_base = sym(base, c.second);
            bool good = true;
            for (const Cand &k : keep)
                if (sym_below(k.second, c.second, to_base)) {
                    good = false;
                    break;
                }
            if (good) keep.push_back(c);
        }
        cands.swap(keep);
    }

    void connect(uint32_t node, std::vector<Cand> &cands, int lv) {
        const int Mmax = lv == 0 ? M0 : Mc;
        select_neighbors(node, cands, Mc);
        {
            locks[node].lock();
            uint32_t *ll = links(node, lv);
            ll[0] = (uint32_t)cands.size();
            for (size_t i = 0; i < cands.size(); ++i) ll[1 + i] = cands[i].second;
            locks[node].unlock();
        }
        for (const Cand &c : cands) {
            const uint32_t o = c.second;
            locks[o].lock();
            uint32_t *ll = links(o, lv);
            if ((int)ll[0] < Mmax) {
                ll[1 + ll[0]] = node;
                ll[0]++;
            } else {
                // shrink o's list with the heuristic over its neighbours + the new node (symmetric distances)
                std::vector<Cand> pool;
                pool.reserve(ll[0] + 1);
                pool.emplace_back(sym(o, node), node);
                for (uint32_t i = 0; i < ll[0]; ++i) pool.emplace_back(sym(o, ll[1 + i]), ll[1 + i]);
                select_neighbors(o, pool, Mmax);
                ll[0] = (uint32_t)pool.size();
                for (size_t i = 0; i < pool.size(); ++i) ll[1 + i] = pool[i].second;
            }
            locks[o].unlock();
        }
    }

    // Algorithm 1
    void insert(uint32_t node, int lv, const float *lut, Visited &vis) {
        std::unique_lock<std::mutex> glock(global);
        const int64_t ep0 = enter;
        const int ml = maxlevel;
        if (ep0 < 0) {
            enter = node;
            maxlevel = lv;
            return;
        }
        if (lv <= ml) glock.unlock();  // only an insertion that raises the top level keeps the graph exclusive
        uint32_t ep = (uint32_t)ep0;
        float ep_d = adc(lut, ep);
        std::vector<uint32_t> nb;
        std::vector<float> hop;
        for (int l = ml; l > lv; --l) {  // greedy descent
            bool changed = true;
            while (changed) {
                changed = false;
                locks[ep].lock();
                const uint32_t *ll = links(ep, l);
                nb.assign(ll + 1, ll + 1 + ll[0]);
                locks[ep].unlock();
                for (uint32_t v : nb) __builtin_prefetch(codes.data() + (size_t)v * M);
                hop.resize(nb.size());
                adc_many(lut, nb.data(), nb.size(), hop.data());
                for (size_t i = 0; i < nb.size(); ++i) {
                    if (hop[i] < ep_d) {
                        ep_d = hop[i];
                        ep = nb[i];
                        changed = true;
                    }
                }
            }
        }
        std::vector<Cand> cands;
        for (int l = std::min(lv, ml); l >= 0; --l) {
            search_layer(lut, ep, ep_d, efc, l, vis, cands, true);
            // next layer starts from the closest found
            Cand best = cands[0];
            for (const Cand &c : cands)
                if (c < best) best = c;
            ep = best.second;
            ep_d = best.first;
            // concurrent insertions may already have linked to this node on an upper layer: never link to itself
            cands.erase(std::remove_if(cands.begin(), cands.end(), [&](const Cand &c) { return c.second == node; }),
                        cands.end());
            connect(node, cands, l);
        }
        if (lv > ml) {
            enter = node;
            maxlevel = lv;
        }
    }
};

extern "C" {

const char *annlite_hnsw_last_error(void) { return g_err; }

annlite_hnsw *annlite_hnsw_create(const float *codebooks, int64_t M, int64_t Ks, int64_t dsub, int64_t capacity,
                                  int max_connection, int ef_construction, uint64_t seed) {
    if (!codebooks || M < 1 || Ks < 1 || Ks > 256 || dsub < 1 || capacity < 1 ||
        max_connection < 2 || ef_construction < 1) {
        set_error("bad arguments (Ks <= 256 => uint8 codes, capacity >= 1, max_connection >= 2)");
        return nullptr;
    }
    annlite_hnsw *g = new annlite_hnsw();
    g->M = M;
    g->Ks = Ks;
    g->dsub = dsub;
    g->D = M * dsub;
    g->cb.assign(codebooks, codebooks + M * Ks * dsub);
    g->Mc = max_connection;
    g->M0 = 2 * max_connection;
    g->efc = std::max(ef_construction, max_connection);
    g->mult = 1.0 / std::log((double)max_connection);
    g->rng.seed(seed);
    g->build_sdc();
    g->build_cbT();
    if (annlite_hnsw_reserve(g, capacity) != 0) {
        delete g;
        return nullptr;
    }
    return g;
}

void annlite_hnsw_free(annlite_hnsw *g) { delete g; }

int64_t annlite_hnsw_size(const annlite_hnsw *g) { return g ? g->n.load() : 0; }

int annlite_hnsw_reserve(annlite_hnsw *g, int64_t capacity) {
    if (!g || capacity < g->cap) return 0;
    if (capacity >= (int64_t)1 << 31) {  // (bit 31 of a node id is the beam's "expanded" flag)
        set_error("capacity must be < 2^31");
        return 1;
    }
    g->codes.resize((size_t)capacity * g->M);
    g->level.resize((size_t)capacity, -1);
    g->link0.resize((size_t)capacity * (g->M0 + 1), 0);
    g->linkU.resize((size_t)capacity);
    g->deleted.resize((size_t)capacity, 0);
    std::unique_ptr<SpinLock[]> nl(new SpinLock[capacity]);
    g->locks.swap(nl);
    g->cap = capacity;
    return 0;
}

int annlite_hnsw_add(annlite_hnsw *g, const float *x, const uint8_t *codes, const int64_t *labels, int64_t n,
                     int n_threads) {
    if (!g || (n > 0 && (!x || !codes || !labels))) {
        set_error("null argument");
        return 1;
    }
    int64_t need = g->cap;
    for (int64_t i = 0; i < n; ++i) {
        if (labels[i] < 0) {
            set_error("negative label");
            return 1;
        }
        need = std::max(need, labels[i] + 1);
    }
    if (need > g->cap && annlite_hnsw_reserve(g, std::max(need, g->cap + g->cap / 2)) != 0) return 1;
    // levels are drawn up front, in label order of the batch: the build is reproducible for a fixed thread count 1
    std::vector<int> lv((size_t)n);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) {
        if (g->level[(size_t)labels[i]] >= 0) {
            set_error("label %lld already present (update = delete + add with a new offset, like the reference's table)",
                      (long long)labels[i]);
            return 1;
        }
        double u = U(g->rng);
        if (u < 1e-300) u = 1e-300;
        lv[(size_t)i] = (int)(-std::log(u) * g->mult);
    }
    for (int64_t i = 0; i < n; ++i) {
        const size_t s = (size_t)labels[i];
        std::memcpy(g->codes.data() + s * g->M, codes + (size_t)i * g->M, (size_t)g->M);
        g->level[s] = lv[(size_t)i];
        g->linkU[s].assign((size_t)lv[(size_t)i] * (g->Mc + 1), 0);
        g->link0[s * (g->M0 + 1)] = 0;
    }
    const int nt = n_threads > 0 ? n_threads : usable_threads();
    int64_t start = 0;
    if (g->enter < 0 && n > 0) {  // the very first point, single-threaded
        std::vector<float> lut((size_t)g->M * g->Ks);
        annlite_hnsw::Visited vis;
        g->build_lut(x, lut.data());
        g->insert((uint32_t)labels[0], lv[0], lut.data(), vis);
        start = 1;
    }
#pragma omp parallel num_threads(nt)
    {
        std::vector<float> lut((size_t)g->M * g->Ks);
        static thread_local annlite_hnsw::Visited vis;
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = start; i < n; ++i) {
            g->build_lut(x + (size_t)i * g->D, lut.data());
            g->insert((uint32_t)labels[i], lv[(size_t)i], lut.data(), vis);
        }
    }
    g->n += n;
    return 0;
}

int annlite_hnsw_search(const annlite_hnsw *g, const float *queries, int64_t B, int ef, int64_t *out_ids, float *out_dist,
                        int n_threads) {
    if (!g || (B > 0 && (!queries || !out_ids || !out_dist)) || ef < 1) {
        set_error("bad arguments");
        return 1;
    }
    const int nt = n_threads > 0 ? n_threads : usable_threads();
#pragma omp parallel num_threads(nt)
    {
        std::vector<float> lut((size_t)g->M * g->Ks);
        // one visited array per THREAD for the life of the process (epoch-stamped, never cleared): allocating and
        // zeroing cap x 4 bytes per thread per call cost 87 ms per 1024-query batch at 5M rows x 256 threads
        static thread_local annlite_hnsw::Visited vis;
        std::vector<Cand> cands;
        std::vector<uint32_t> nb;
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < B; ++b) {
            int64_t *oi = out_ids + (size_t)b * ef;
            float *od = out_dist + (size_t)b * ef;
            for (int i = 0; i < ef; ++i) {
                oi[i] = -1;
                od[i] = INFINITY;
            }
            if (g->enter < 0) continue;
            g->build_lut(queries + (size_t)b * g->D, lut.data());
            uint32_t ep = (uint32_t)g->enter;
            float ep_d = g->adc(lut.data(), ep);
            for (int l = g->maxlevel; l > 0; --l) {
                bool changed = true;
                while (changed) {
                    changed = false;
                    const uint32_t *ll = g->links(ep, l);
                    for (uint32_t i = 0; i < ll[0]; ++i) {
                        const float d = g->adc(lut.data(), ll[1 + i]);
                        if (d < ep_d) {
                            ep_d = d;
                            ep = ll[1 + i];
                            changed = true;
                        }
                    }
                }
            }
            g->search_layer(lut.data(), ep, ep_d, ef, 0, vis, cands, false);
            std::sort(cands.begin(), cands.end());
            int o = 0;
            for (const Cand &c : cands) {
                if (g->deleted[c.second]) continue;
                oi[o] = (int64_t)c.second;
                od[o] = c.first;
                ++o;
            }
        }
    }
    return 0;
}

int annlite_hnsw_export(const annlite_hnsw *g, int64_t n_rows, uint32_t *links_out, int64_t *seeds_out, int64_t max_seeds,
                        int64_t *n_seeds_out) {
    if (!g || n_rows < 0 || n_rows > g->cap || !links_out || !seeds_out || !n_seeds_out || max_seeds < 1) {
        set_error("bad arguments");
        return 1;
    }
    const size_t stride = (size_t)g->M0 + 1;
    std::memcpy(links_out, g->link0.data(), (size_t)n_rows * stride * 4);
    // seed set: every node of the highest levels, as many levels as fit max_seeds (the top of the hierarchy,
    // which the GPU walk scans flat instead of descending)
    int lv = g->maxlevel;
    std::vector<int64_t> count((size_t)std::max(lv, 0) + 2, 0);
    for (int64_t i = 0; i < n_rows; ++i)
        if (g->level[(size_t)i] >= 0 && !g->deleted[(size_t)i]) count[(size_t)std::min(g->level[(size_t)i], lv)]++;
    int64_t acc = 0;
    int min_level = lv;
    for (int l = lv; l >= 0; --l) {
        if (acc + count[(size_t)l] > max_seeds && acc > 0) break;
        acc += count[(size_t)l];
        min_level = l;
        if (acc >= max_seeds) break;
    }
    int64_t ns = 0;
    for (int64_t i = 0; i < n_rows && ns < max_seeds; ++i)
        if (g->level[(size_t)i] >= min_level && !g->deleted[(size_t)i]) seeds_out[ns++] = i;
    *n_seeds_out = ns;
    return 0;
}

int annlite_hnsw_links_per_node(const annlite_hnsw *g) { return g ? g->M0 : 0; }

int annlite_hnsw_mark_deleted(annlite_hnsw *g, int64_t label) {
    if (!g || label < 0 || label >= g->cap || g->level[(size_t)label] < 0) {
        set_error("label %lld not in the graph", (long long)label);
        return 1;
    }
    g->deleted[(size_t)label] = 1;
    return 0;
}

static const uint64_t kMagic = 0x31474e4e41ull;  // "ANNG1"

int annlite_hnsw_save(const annlite_hnsw *g, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("cannot open %s", path);
        return 1;
    }
    auto w = [&](const void *p, size_t n) { return fwrite(p, 1, n, f) == n; };
    const int64_t hdr[12] = {(int64_t)kMagic, 1, g->M, g->Ks, g->dsub, g->cap, g->n.load(), g->Mc, g->efc,
                             g->enter, g->maxlevel, 0};
    bool ok = w(hdr, sizeof(hdr)) && w(g->cb.data(), g->cb.size() * 4) && w(g->codes.data(), g->codes.size()) &&
              w(g->level.data(), g->level.size() * 4) && w(g->link0.data(), g->link0.size() * 4) &&
              w(g->deleted.data(), g->deleted.size());
    for (int64_t i = 0; ok && i < g->cap; ++i) {
        const uint64_t sz = g->linkU[(size_t)i].size();
        ok = w(&sz, 8) && (sz == 0 || w(g->linkU[(size_t)i].data(), sz * 4));
    }
    fclose(f);
    if (!ok) set_error("short write to %s", path);
    return ok ? 0 : 1;
}

annlite_hnsw *annlite_hnsw_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open %s", path);
        return nullptr;
    }
    auto r = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
    int64_t hdr[12];
    if (!r(hdr, sizeof(hdr)) || (uint64_t)hdr[0] != kMagic) {
        fclose(f);
        set_error("%s is not an annlite graph file", path);
        return nullptr;
    }
    std::vector<float> cb((size_t)(hdr[2] * hdr[3] * hdr[4]));
    if (!r(cb.data(), cb.size() * 4)) {
        fclose(f);
        set_error("truncated file");
        return nullptr;
    }
    annlite_hnsw *g = annlite_hnsw_create(cb.data(), hdr[2], hdr[3], hdr[4], hdr[5], (int)hdr[7], (int)hdr[8], 0);
    if (!g) {
        fclose(f);
        return nullptr;
    }
    g->n = hdr[6];
    g->enter = hdr[9];
    g->maxlevel = (int)hdr[10];
    bool ok = r(g->codes.data(), g->codes.size()) && r(g->level.data(), g->level.size() * 4) &&
              r(g->link0.data(), g->link0.size() * 4) && r(g->deleted.data(), g->deleted.size());
    for (int64_t i = 0; ok && i < g->cap; ++i) {
        uint64_t sz = 0;
        ok = r(&sz, 8);
        if (ok && sz) {
            g->linkU[(size_t)i].resize(sz);
            ok = r(g->linkU[(size_t)i].data(), sz * 4);
        }
    }
    fclose(f);
    if (!ok) {
        set_error("truncated file");
        delete g;
        return nullptr;
    }
    return g;
}

}  // extern "C"
