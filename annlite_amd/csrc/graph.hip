// graph.hip -- beam search over the level-0 lists of the HNSW-over-PQ graph ON THE GPU (BASELINE config 5).
//
// The reference walks its graph on the host, one query per call (hnsw_bindings.cpp:302-375 -> hnswalg.h
// searchBaseLayerST); libannlite_graph.so does the same, batch-parallel over CPU threads.  This kernel
// moves the walk next to the code table: ONE WAVE PER QUERY, its L2 look-up table (M*Ks fp32 = 16 KB at
// M=16) and a 4096-entry visited hash set (16 KB) in LDS, the ef_search best nodes seen so far as a sorted list spread over
// the lanes' registers (two entries per lane for ef = 128).  The upper layers of the hierarchy are not
// descended: the nodes of its top levels (annlite_hnsw_export: up to 4096 "seeds") are scanned flat --
// a coalesced gather of a few thousand code rows -- and the walk starts from the best of them.
//
// Per expansion (best unexpanded node of the list): its link list is one coalesced 132-byte read, the
// lanes that hold unseen neighbours fetch those rows' code bytes (the only random HBM traffic: 16 x 64-byte
// sectors) and evaluate hnswlib::PQLookup from LDS -- fp32 adds in ascending sub-space order, the same
// bits as the flat scan (space_pq.h:15-37) -- and the survivors are inserted into the sorted list.
// The walk ends when the list holds no unexpanded node (hnswalg.h searchBaseLayerST's stop rule).
//
// PACKED layout (round 5, annlite_graph_pack / annlite_graph_search_packed).  With the plain layout an expansion is TWO dependent
// random round trips -- the node's link list, then the 16-byte code rows of its unseen neighbours, each of which costs a
// 128-byte line: 385 MB of HBM traffic per 5M-row launch against 62 MB algorithmic, L2 hit rate 12 % -- and one wave per SIMD
// hides none of it.  A packed node record holds the neighbours' CODE ROWS behind its link list,
//     [L x M code bytes of neighbour 0 .. L-1][L x u32 link ids][u32 count][pad to 16 B]      (656 B at L = 32, M = 16)
// so an expansion is ONE contiguous read (lane j: neighbour j's 16 code bytes + its id), and the record of the best
// still-unexpanded node is PREFETCHED into registers while the current node's neighbours are evaluated: if the next pick is
// that node -- it is, once the walk has reached its plateau -- no round trip is exposed at all.  The order of the walk and
// the arithmetic are unchanged: candidate lists are bit-equal to the plain kernel's (tests/test_graph_packed.py).
#include "scan_common.h"

namespace annlite {

constexpr uint32_t kEmpty = 0xffffffffu;

// sorted list of 64 * E entries across the lanes: entry i lives in lane i % 64, slot i / 64
template <int E>
struct BeamList {
    uint32_t hi[E], lo[E];  // ordered distance key, node id
    bool exp[E];            // expanded already
};

template <int E>
__device__ __forceinline__ void beam_insert(BeamList<E> &L, uint32_t chi, uint32_t clo, int lane) {
    // entries smaller than the candidate form a prefix of the list
    int pos = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) pos += __popcll(__ballot(key_less(L.hi[e], L.lo[e], chi, clo)));
    // shift entries [pos, end) up by one, dropping the last; entry i - 1 of slot e lane l is (e, l-1), or (e-1, 63) for l = 0
    uint32_t phi[E], plo[E];
    bool pex[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        phi[e] = __shfl_up(L.hi[e], 1);
        plo[e] = __shfl_up(L.lo[e], 1);
        pex[e] = __shfl_up((int)L.exp[e], 1) != 0;
    }
#pragma unroll
    for (int e = E - 1; e >= 1; --e) {
        const uint32_t bh = __builtin_amdgcn_readlane(L.hi[e - 1], 63), bl = __builtin_amdgcn_readlane(L.lo[e - 1], 63);
        const bool bx = __builtin_amdgcn_readlane((int)L.exp[e - 1], 63) != 0;
        if (lane == 0) {
            phi[e] = bh;
            plo[e] = bl;
            pex[e] = bx;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = e * 64 + lane;
        if (idx == pos) {
            L.hi[e] = chi;
            L.lo[e] = clo;
            L.exp[e] = false;
        } else if (idx > pos) {
            L.hi[e] = phi[e];
            L.lo[e] = plo[e];
            L.exp[e] = pex[e];
        }
    }
}

// BATCH (round 5): the neighbours of an expansion that beat the list's worst entry are MERGED into the sorted list in one pass --
// every list entry counts the candidates below it, every candidate the entries and candidates below it, all move to their
// final positions through a per-wave LDS scratch -- instead of one full-wave shift per candidate (~110 instructions each, the
// largest single item of the walk: one wave per SIMD executes them back to back).  The result is the same sorted list --
// the top of (old list + candidates) -- so the walk is unchanged; BATCH = false keeps the one-at-a-time insertion (the
// comparison path of tests/test_graph_packed.py).
// W (round 6, PACKED only): nodes expanded per step.  W = 2 = the PAIR walk: the two best unexpanded entries are expanded TOGETHER --
// lanes 0..31 take one record, lanes 32..63 the other (links_per_node <= 32), one pass through the visited table, one merge of up
// to 64 neighbours -- so a walk's chain of dependent steps is half as long (one wave per SIMD hides no latency: the launch lasts as
// long as one query's chain).  Every entry that stays inside the ef best is expanded sooner or later in either walk; the pair walk
// expands the runner-up before it knows the winner's neighbours, which changes the ORDER of expansions (and, rarely, expands a node
// the one-at-a-time walk would have seen pushed out of the list first).  The list is the same kind of object -- the ef best of the
// nodes evaluated, exact PQLookup sums -- and tests/test_graph_pair.py pins it bit for bit against a restatement of this order.
template <int M, int E, bool PACKED, bool BATCH, int W = 1>
__global__ __launch_bounds__(256) void graph_beam_search_kernel(const uint32_t *__restrict__ links, int links_per_node,
                                                               const uint8_t *__restrict__ packed, int64_t rec_stride,
                                                               const uint32_t *__restrict__ seeds, int n_seeds,
                                                               const uint8_t *__restrict__ codes, int64_t N,
                                                               const uint32_t *__restrict__ valid,
                                                               const float *__restrict__ lut_bmk, int B, int Ks, int ef,
                                                               int hash_bits, int64_t *__restrict__ out_ids,
                                                               float *__restrict__ out_dist,
                                                               unsigned long long *__restrict__ stats) {
    constexpr int CW = M / 4;
    unsigned int n_expand = 0, n_eval = 0, n_hit = 0;  // (ANNLITE_DEBUG_COUNTERS: link lists read, rows evaluated, prefetched records used)
    unsigned long long t_seed = 0, t_rec = 0, t_visit = 0, t_sum = 0, t_offer = 0;  // ... and shader cycles by phase (packed walk)
    auto now = [&]() -> unsigned long long { return stats ? __builtin_readcyclecounter() : 0ull; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;  // 1..4 waves per workgroup, as many as the LDS holds
    if (b >= B) return;  // (whole waves leave; no block-wide barrier below)
    const uint32_t hash_n = 1u << hash_bits;
    const size_t per_wave = (size_t)M * Ks * 4 + (size_t)hash_n * 4 + (BATCH ? (size_t)E * 64 * 12 + 1024 : 0);  // (same formula as launch_beam)
    float *s_lut = (float *)(smem + wave * per_wave);
    uint32_t *s_hash = (uint32_t *)(smem + wave * per_wave + (size_t)M * Ks * 4);
    uint32_t *s_mrg = s_hash + hash_n;  // BATCH: the merge's scratch, u32 [3][64 E]: keys hi, ids, expanded flags; then u32 [4][64]: the
                                        // candidates compacted (hi, id) and by rank (hi, id)
    {
        const f32x4 *src = (const f32x4 *)(lut_bmk + (int64_t)b * M * Ks);
        for (int i = lane; i < M * Ks / 4; i += 64) ((f32x4 *)s_lut)[i] = src[i];
        for (uint32_t i = lane; i < hash_n / 4; i += 64) ((u32x4 *)s_hash)[i] = (u32x4){kEmpty, kEmpty, kEmpty, kEmpty};
    }
    // (one wave: its LDS writes are visible to its own later reads in program order)

    // true if the node was not seen before (it is recorded now).  A (nearly) full table cannot record any more: the
    // node is then reported as new every time -- evaluated again, never lost -- and `offer` keeps it out of the list
    // if it is already there, so a walk that outgrows the table only gets slower, not worse.
    // BUCKETS of four entries (round 5): one 16-byte read shows a bucket, the node is looked for in it and, if there is room,
    // claimed with ONE compare-and-swap.  (One entry per probe: at the end of a 5M-row walk the 4096-entry table holds ~2600
    // nodes and the slowest of an expansion's 32 lanes needed a dozen dependent LDS round trips.)
    // (by LDS byte address in the LDS address space: through a generic volatile pointer the bucket read was a FLAT load, and a flat
    // load counts on the vector-memory counter too -- every probe then waited for the prefetched record)
    bool tbl_ovf = false;   // this lane has met a full stretch of the visited table
    bool tbl_full = false;  // ... some lane of the wave has (wave-uniform, sticky)
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    const uint32_t hash_ad = (uint32_t)(uintptr_t)(lds_bytes)smem + (uint32_t)(wave * per_wave) + (uint32_t)(M * Ks * 4);
    auto visit = [&](uint32_t node) -> bool {
        const uint32_t n_buckets = hash_n >> 2;
        uint32_t bk = ((node * 2654435761u) >> (32 - hash_bits)) >> 2;
        for (uint32_t probe = 0; probe < 16; ++probe) {
            const u32x4 v = *(volatile __attribute__((address_space(3))) u32x4 *)(uintptr_t)(hash_ad + 16u * bk);
            if (v.x == node || v.y == node || v.z == node || v.w == node) return false;
            const int slot = v.x == kEmpty ? 0 : v.y == kEmpty ? 1 : v.z == kEmpty ? 2 : v.w == kEmpty ? 3 : -1;
            if (slot >= 0) {
                uint32_t old = kEmpty;  // (expected; the call leaves what it found here)
                __hip_atomic_compare_exchange_strong((__attribute__((address_space(3))) uint32_t *)(uintptr_t)(hash_ad + 16u * bk + 4u * (uint32_t)slot),
                                                     &old, node, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == kEmpty) return true;
                if (old == node) return false;
                continue;  // (another lane of this expansion took the slot: look at the bucket again)
            }
            bk = (bk + 1) & (n_buckets - 1);
        }
        tbl_ovf = true;  // (no room within 16 buckets: from now on a node can be evaluated twice -- the merge checks for duplicates)
        return true;
    };
    auto load_row = [&](uint32_t node, uint32_t (&c)[CW]) {
        const uint32_t *p = (const uint32_t *)(codes + (int64_t)node * M);
#pragma unroll
        for (int i = 0; i < CW; ++i) c[i] = p[i];
    };
    auto pq_sum = [&](const uint32_t (&c)[CW]) -> float {  // hnswlib::PQLookup: ascending-m fp32 adds
        float d = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) d += s_lut[m * Ks + ((c[m / 4] >> (8 * (m % 4))) & 0xffu)];
        return d;
    };
    auto pq_lookup = [&](uint32_t node) -> float {
        uint32_t c[CW];
        load_row(node, c);
        return pq_sum(c);
    };
    auto is_valid = [&](uint32_t node) -> bool { return !valid || ((valid[node >> 5] >> (node & 31)) & 1u); };

    BeamList<E> L;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        L.hi[e] = kKeyInfHi;
        L.lo[e] = kIdNone;
        L.exp[e] = true;  // empty slots are never picked
    }
    const int cap = ef < 64 * E ? ef : 64 * E;  // entries beyond `cap` are ignored at the end
    auto worst = [&](uint32_t &whi, uint32_t &wlo) {
        const int i = cap - 1;
        whi = kKeyInfHi;
        wlo = kIdNone;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (i / 64 == e) {
                whi = __builtin_amdgcn_readlane(L.hi[e], i % 64);
                wlo = __builtin_amdgcn_readlane(L.lo[e], i % 64);
            }
    };
    auto offer = [&](bool mine, uint32_t node, float d) {
        uint32_t whi, wlo;
        worst(whi, wlo);
        const uint32_t khi = f32_to_ordered(d);
        unsigned long long pm = __ballot(mine && key_less(khi, node, whi, wlo));
        if constexpr (!BATCH) {
            while (pm) {
                const int src = __builtin_ctzll(pm);
                pm &= pm - 1;
                const uint32_t chi = __builtin_amdgcn_readlane(khi, src), clo = __builtin_amdgcn_readlane(node, src);
                worst(whi, wlo);
                bool dup = false;
#pragma unroll
                for (int e = 0; e < E; ++e) dup = dup || __ballot(L.lo[e] == clo && L.hi[e] == chi) != 0ull;
                if (!dup && key_less(chi, clo, whi, wlo)) beam_insert<E>(L, chi, clo, lane);
            }
        } else {
            if (!pm) return;
            // FAST PATH: ranks by counting, no wave-wide vote per candidate.  The candidates' keys go to the LDS compacted; every
            // lane reads them back one by one (same address in all lanes: a broadcast) and counts -- a list entry the candidates
            // below it (its shift), a candidate lane the candidates below its own key (its rank r).  Entries move to position +
            // shift through the scratch; the positions nobody moved to are the candidates', in rank order: the h-th hole takes
            // the candidate of rank h.  A candidate equal to a list entry or to another candidate (only when the visited table
            // is full) sends the whole expansion down the careful path below.
            {
                uint32_t *cu_hi = s_mrg + 192 * E, *cu_lo = cu_hi + 64, *cs_hi = cu_hi + 128, *cs_lo = cu_hi + 192;
                const bool in = ((pm >> lane) & 1ull) != 0;
                const int n = __popcll(pm);
                const int myidx = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                if (in) cu_hi[myidx] = khi, cu_lo[myidx] = node;
                // (LANES talk to each other through the scratch: the compiler, which sees one thread, must not forward a lane's own
                // store to its later load of the same address -- another lane may have written there in between)
                asm volatile("" ::: "memory");
                int shift[E];
#pragma unroll
                for (int e = 0; e < E; ++e) shift[e] = 0;
                int r = 0;
                bool bad = false;
                // (duplicates exist only once the visited table has overflowed: until then the equality tests are left out)
                if (!tbl_full && __ballot(tbl_ovf)) tbl_full = true;
                if (tbl_full) {
#pragma unroll 2
                    for (int j = 0; j < n; ++j) {
                        const uint32_t chi = cu_hi[j], clo = cu_lo[j];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            shift[e] += key_less(chi, clo, L.hi[e], L.lo[e]) ? 1 : 0;
                            bad = bad || (chi == L.hi[e] && clo == L.lo[e]);
                        }
                        r += (in && key_less(chi, clo, khi, node)) ? 1 : 0;
                        bad = bad || (in && j != myidx && chi == khi && clo == node);
                    }
                } else {
#pragma unroll 2
                    for (int j = 0; j < n; ++j) {
                        const uint32_t chi = cu_hi[j], clo = cu_lo[j];
#pragma unroll
                        for (int e = 0; e < E; ++e) shift[e] += key_less(chi, clo, L.hi[e], L.lo[e]) ? 1 : 0;
                        r += (in && key_less(chi, clo, khi, node)) ? 1 : 0;
                    }
                }
                if (!__ballot(bad)) {
                    if (in) cs_hi[r] = khi, cs_lo[r] = node;
#pragma unroll
                    for (int e = 0; e < E; ++e) s_mrg[128 * E + e * 64 + lane] = 2u;  // "nobody moved here"
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int np = e * 64 + lane + shift[e];
                        if (np < 64 * E) {
                            s_mrg[np] = L.hi[e];
                            s_mrg[64 * E + np] = L.lo[e];
                            s_mrg[128 * E + np] = L.exp[e] ? 1u : 0u;
                        }
                    }
                    asm volatile("" ::: "memory");
                    int hbase = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int P = e * 64 + lane;
                        const uint32_t x = s_mrg[128 * E + P];
                        uint32_t nhi = s_mrg[P], nlo = s_mrg[64 * E + P];
                        const bool hole = x == 2u;
                        const unsigned long long hm = __ballot(hole);
                        if (hole) {
                            const int h = hbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                            nhi = cs_hi[h];
                            nlo = cs_lo[h];
                        }
                        L.hi[e] = nhi;
                        L.lo[e] = nlo;
                        L.exp[e] = !hole && x != 0u;
                        hbase += __popcll(hm);
                    }
                    return;
                }
            }
            // CAREFUL PATH (a duplicate somewhere): one pass over the candidates with wave-wide votes.
            // (wave-uniform: their keys are broadcast): a list entry counts the candidates below
            // it, a candidate lane the list entries and the other candidates below its key.  A candidate that is already in the
            // list, or equals an earlier candidate (both only when the visited table is full: see visit), is dropped --
            // what the one-at-a-time insertion's duplicate check does.
            int shift[E];
#pragma unroll
            for (int e = 0; e < E; ++e) shift[e] = 0;
            int mypos = 0;
            bool in = ((pm >> lane) & 1ull) != 0;
            unsigned long long rem = pm;
            while (rem) {
                const int src = __builtin_ctzll(rem);
                rem &= rem - 1;
                const uint32_t chi = __builtin_amdgcn_readlane(khi, src), clo = __builtin_amdgcn_readlane(node, src);
                bool dup = false;
#pragma unroll
                for (int e = 0; e < E; ++e) dup = dup || __ballot(L.lo[e] == clo && L.hi[e] == chi) != 0ull;
                if (dup) {
                    if (lane == src) in = false;
                    continue;
                }
                const unsigned long long same = __ballot(in && lane > src && khi == chi && node == clo) & rem;
                if (same) {
                    if ((same >> lane) & 1ull) in = false;
                    rem &= ~same;
                }
                int below = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool lt = key_less(L.hi[e], L.lo[e], chi, clo);  // entry < candidate (never equal: see above)
                    below += __popcll(__ballot(lt));
                    shift[e] += lt ? 0 : 1;
                }
                if (lane == src) mypos += below;
                else if (in && key_less(chi, clo, khi, node)) mypos += 1;
            }
            // every element of (list + kept candidates) now knows its rank in the union: the first 64 E go back into the list
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int np = e * 64 + lane + shift[e];
                if (np < 64 * E) {
                    s_mrg[np] = L.hi[e];
                    s_mrg[64 * E + np] = L.lo[e];
                    s_mrg[128 * E + np] = L.exp[e] ? 1u : 0u;
                }
            }
            if (in && mypos < 64 * E) {
                s_mrg[mypos] = khi;
                s_mrg[64 * E + mypos] = node;
                s_mrg[128 * E + mypos] = 0u;
            }
            asm volatile("" ::: "memory");  // (see the fast path: no store-to-load forwarding across lanes)
            // (one wave: its LDS writes are visible to its own later reads in program order)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                L.hi[e] = s_mrg[e * 64 + lane];
                L.lo[e] = s_mrg[64 * E + e * 64 + lane];
                L.exp[e] = s_mrg[128 * E + e * 64 + lane] != 0u;
            }
        }
    };

    // ---- seeds: the top of the hierarchy, scanned flat -------------------------------------------------
    // (seeds are distinct: no visited check while scanning them; only the ones that made the list are recorded --
    // recording all of them would fill the hash table before the walk starts)
    // (the round's code rows are requested one round ahead, its seed ids two: the rounds' two dependent round trips overlap
    // the previous rounds' insertions; the ORDER of the offers is unchanged)
    {
        const unsigned long long t_s0 = now();
        auto seed_at = [&](int s0) -> uint32_t { return s0 + lane < n_seeds ? seeds[s0 + lane] : 0xffffffffu; };
        uint32_t node_cur = seed_at(0), node_nxt = seed_at(64);
        uint32_t c_cur[CW];
        load_row(((int64_t)node_cur < N) ? node_cur : 0u, c_cur);
        for (int s0 = 0; s0 < n_seeds; s0 += 64) {
            const uint32_t node = node_cur;
            const bool mine = s0 + lane < n_seeds && (int64_t)node < N;
            uint32_t c[CW];
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = c_cur[i];
            node_cur = node_nxt;
            if (s0 + 64 < n_seeds) load_row(((int64_t)node_cur < N) ? node_cur : 0u, c_cur);
            node_nxt = seed_at(s0 + 128);
            const float d = mine ? pq_sum(c) : 0.f;
            n_eval += (unsigned int)__popcll(__ballot(mine));
            offer(mine, node, d);
        }
        t_seed = now() - t_s0;
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (L.lo[e] != kIdNone) visit(L.lo[e]);

    // ---- walk ------------------------------------------------------------------------------------------
    // the best unexpanded entry of the list (marked expanded) and -- `second` -- the next best one, unmarked: the prefetch target
    auto pick_next = [&](uint32_t &node, uint32_t &second) -> bool {
        int p0 = -1, p1 = -1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (p1 < 0) {
                unsigned long long m = __ballot(!L.exp[e] && (e * 64 + lane) < cap);
                if (m && p0 < 0) {
                    p0 = e * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                }
                if (m && p0 >= 0) p1 = e * 64 + __builtin_ctzll(m);
            }
        }
        if (p0 < 0) return false;
        node = 0;
        second = kEmpty;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (p0 / 64 == e) {
                node = __builtin_amdgcn_readlane(L.lo[e], p0 % 64);
                if (lane == p0 % 64) L.exp[e] = true;
            }
            if (p1 >= 0 && p1 / 64 == e) second = __builtin_amdgcn_readlane(L.lo[e], p1 % 64);
        }
        return true;
    };
    if constexpr (PACKED && W == 2) {
        const int Lc = links_per_node;  // (<= 32: one HALF wave per record)
        const int half = lane >> 5, hj = lane & 31;
        // the record of `node` as seen by this lane's half: neighbour hj's code row and id (slots beyond the count hold 0xffffffff)
        auto load_rec = [&](uint32_t node, uint32_t (&c)[CW], uint32_t &nb) {
            const uint8_t *r = packed + (int64_t)node * rec_stride;
            const int j = hj < Lc ? hj : 0;
            const uint32_t *pc = (const uint32_t *)(r + (int64_t)j * M);
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = pc[i];
            nb = ((const uint32_t *)(r + (int64_t)Lc * M))[j];
        };
        // the two best unexpanded entries (marked expanded) and the two after them (the halves' prefetch targets)
        auto pick_pair = [&](uint32_t (&nd)[4]) -> bool {
            int p[4] = {-1, -1, -1, -1};
            int found = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (found < 4) {
                    unsigned long long m = __ballot(!L.exp[e] && (e * 64 + lane) < cap);
                    while (m && found < 4) {
                        p[found++] = e * 64 + __builtin_ctzll(m);
                        m &= m - 1;
                    }
                }
            }
            if (found == 0) return false;
#pragma unroll
            for (int i = 0; i < 4; ++i) nd[i] = kEmpty;
#pragma unroll
            for (int e = 0; e < E; ++e) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p[i] >= 0 && p[i] / 64 == e) nd[i] = __builtin_amdgcn_readlane(L.lo[e], p[i] % 64);
                if ((p[0] >= 0 && p[0] / 64 == e && lane == p[0] % 64) || (p[1] >= 0 && p[1] / 64 == e && lane == p[1] % 64)) L.exp[e] = true;
            }
            return true;
        };
        uint32_t pf_c[CW], pf_nb = kEmpty, pf_node = kEmpty;  // this HALF's prefetched record (pf_node == kEmpty: none)
#pragma unroll
        for (int i = 0; i < CW; ++i) pf_c[i] = 0;
        for (;;) {
            uint32_t nd[4];
            const unsigned long long t0 = now();
            if (!pick_pair(nd)) break;
            // a prefetched record stays in the half that holds it: the halves swap nodes when that keeps more of them (the
            // candidates of a step are merged as a SET -- which half evaluated a neighbour does not matter)
            const uint32_t pf0 = __builtin_amdgcn_readlane(pf_node, 0), pf1 = __builtin_amdgcn_readlane(pf_node, 32);
            const bool two = nd[1] != kEmpty;
            const int keep_straight = (pf0 == nd[0] ? 1 : 0) + (two && pf1 == nd[1] ? 1 : 0);
            const int keep_swapped = (two && pf0 == nd[1] ? 1 : 0) + (pf1 == nd[0] ? 1 : 0);
            const bool swapped = keep_swapped > keep_straight;
            const uint32_t node = ((half != 0) != swapped) ? nd[1] : nd[0];
            const bool have = node != kEmpty;
            uint32_t c[CW], nb = kEmpty;
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = 0;
            if (have && pf_node == node) {
#pragma unroll
                for (int i = 0; i < CW; ++i) c[i] = pf_c[i];
                nb = pf_nb;
            } else if (have) {
                load_rec(node, c, nb);
            }
            n_hit += (unsigned int)__popcll(__ballot(have && pf_node == node && hj == 0));
            // (the current records must have ARRIVED before the prefetches are issued: see the one-at-a-time loop below)
            asm volatile("" : "+v"(nb));
#pragma unroll
            for (int i = 0; i < CW; ++i) asm volatile("" : "+v"(c[i]));
            // half 0 requests the best node that is unexpanded NOW, half 1 the one after it: the next pair unless this step's
            // neighbours turn out better
            pf_node = half ? nd[3] : nd[2];
            if (pf_node != kEmpty) load_rec(pf_node, pf_c, pf_nb);
            n_expand += two ? 2u : 1u;
            const unsigned long long t1 = now();
            // (a neighbour of BOTH nodes is probed by two lanes at once: the table's compare-and-swap lets exactly one of them through)
            bool mine = have && hj < Lc;
            mine = mine && (int64_t)nb < N && visit(nb);
            n_eval += (unsigned int)__popcll(__ballot(mine));
            const unsigned long long t2 = now();
            float d = mine ? pq_sum(c) : 0.f;
            if (stats) asm volatile("" : "+v"(d));
            const unsigned long long t3 = now();
            offer(mine, nb, d);
            if (stats) {
                const unsigned long long t4 = now();
                t_rec += t1 - t0, t_visit += t2 - t1, t_sum += t3 - t2, t_offer += t4 - t3;
            }
        }
    } else if constexpr (PACKED) {
        // record of a node: lane j < L holds neighbour j's code row and id; the count is a wave-uniform load
        const int Lc = links_per_node;  // (<= 64: one lane per neighbour)
        // (slots beyond the node's count hold the id 0xffffffff, which fails the `nb < N` test like any id beyond the table:
        // the walk never reads the count -- a wave-uniform, i.e. SCALAR load would share its counter with the LDS traffic of
        // visit() and stall it for a full memory round trip whenever a prefetch is in flight)
        auto load_rec = [&](uint32_t node, uint32_t (&c)[CW], uint32_t &nb) {
            const uint8_t *r = packed + (int64_t)node * rec_stride;
            const int j = lane < Lc ? lane : 0;
            const uint32_t *pc = (const uint32_t *)(r + (int64_t)j * M);
#pragma unroll
            for (int i = 0; i < CW; ++i) c[i] = pc[i];
            nb = ((const uint32_t *)(r + (int64_t)Lc * M))[j];
        };
        uint32_t pf_c[CW], pf_nb = 0, pf_node = kEmpty;  // the prefetched record (pf_node == kEmpty: none)
#pragma unroll
        for (int i = 0; i < CW; ++i) pf_c[i] = 0;
        for (;;) {
            uint32_t node = 0, second = kEmpty;
            const unsigned long long t0 = now();
            if (!pick_next(node, second)) break;
            uint32_t c[CW], nb;
            if (pf_node == node) {
                ++n_hit;
#pragma unroll
                for (int i = 0; i < CW; ++i) c[i] = pf_c[i];
                nb = pf_nb;
            } else {
                load_rec(node, c, nb);
            }
            // The current record must have ARRIVED before the prefetch below is issued: the memory counter is in order and the
            // compiler, merging the paths with and without a prefetch, otherwise waits for everything outstanding -- the
            // prefetch included -- at the first use of the current record.  (An empty asm that reads the registers.)
            asm volatile("" : "+v"(nb));
#pragma unroll
            for (int i = 0; i < CW; ++i) asm volatile("" : "+v"(c[i]));
            // request the record of the best node that is unexpanded NOW: the next pick unless one of this node's neighbours
            // turns out better
            pf_node = second;
            if (second != kEmpty) load_rec(second, pf_c, pf_nb);
            ++n_expand;
            const unsigned long long t1 = now();
            bool mine = lane < Lc;
            mine = mine && (int64_t)nb < N && visit(nb);
            n_eval += (unsigned int)__popcll(__ballot(mine));
            const unsigned long long t2 = now();
            float d = mine ? pq_sum(c) : 0.f;
            if (stats) asm volatile("" : "+v"(d));
            const unsigned long long t3 = now();
            offer(mine, nb, d);
            if (stats) {
                const unsigned long long t4 = now();
                t_rec += t1 - t0, t_visit += t2 - t1, t_sum += t3 - t2, t_offer += t4 - t3;
            }
        }
    } else {
        for (;;) {
            uint32_t node = 0, second = kEmpty;
            if (!pick_next(node, second)) break;
            const uint32_t *ll = links + (int64_t)node * (links_per_node + 1);
            const uint32_t cnt = ll[0];
            ++n_expand;
            bool mine = false;
            uint32_t nb = 0;
            float d = 0.f;
            for (uint32_t base = 0; base < cnt; base += 64) {  // (links_per_node <= 64 in practice: one round)
                mine = base + lane < cnt;
                nb = mine ? ll[1 + base + lane] : 0u;
                mine = mine && (int64_t)nb < N && visit(nb);
                n_eval += (unsigned int)__popcll(__ballot(mine));
                d = mine ? pq_lookup(nb) : 0.f;
                offer(mine, nb, d);
            }
        }
    }

    // ---- result: the list, ascending; deleted rows dropped here (they still route the walk, like hnswlib) ---
    // compacting is left to the host-side top-k (ids of dropped rows become -1 / +inf)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        if (i < ef) {
            const bool none = (L.hi[e] == kKeyInfHi && L.lo[e] == kIdNone) || i >= cap || !is_valid(L.lo[e]);
            out_ids[(int64_t)b * ef + i] = none ? (int64_t)-1 : (int64_t)L.lo[e];
            out_dist[(int64_t)b * ef + i] = none ? __builtin_inff() : ordered_to_f32(L.hi[e]);
        }
    }
    if (stats && lane == 0) {
        atomicAdd(stats + 0, (unsigned long long)n_expand);
        atomicAdd(stats + 1, (unsigned long long)n_eval);
        atomicAdd(stats + 2, (unsigned long long)n_hit);
        atomicAdd(stats + 3, t_seed), atomicAdd(stats + 4, t_rec), atomicAdd(stats + 5, t_visit), atomicAdd(stats + 6, t_sum), atomicAdd(stats + 7, t_offer);
    }
}

}  // namespace annlite

using namespace annlite;

// Packed node records for graph_beam_search_kernel<.., PACKED>: [L x M code bytes][L x u32 ids][u32 count][pad to 16 B].
// One thread per (node, neighbour slot): copies the neighbour's code row (slots beyond the count / ids beyond N: zeros).
__global__ __launch_bounds__(256) void graph_pack_kernel(const uint32_t *__restrict__ links, int L, const uint8_t *__restrict__ codes,
                                                        int64_t N, int M, uint8_t *__restrict__ out, int64_t stride) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * L) return;
    const int64_t node = t / L;
    const int j = (int)(t - node * L);
    const uint32_t *ll = links + node * (L + 1);
    const uint32_t cnt = ll[0];
    const uint32_t nb = (uint32_t)j < cnt ? ll[1 + j] : kEmpty;  // (an unused slot: an id no table holds)
    const bool ok = (uint32_t)j < cnt && (int64_t)nb < N;
    uint8_t *r = out + node * stride;
    uint32_t *dst = (uint32_t *)(r + (int64_t)j * M);
    const uint32_t *src = (const uint32_t *)(codes + (int64_t)(ok ? nb : 0u) * M);
    for (int i = 0; i < M / 4; ++i) dst[i] = ok ? src[i] : 0u;
    uint32_t *hdr = (uint32_t *)(r + (int64_t)L * M);
    hdr[j] = nb;
    if (j == 0) {
        hdr[L] = cnt;
        for (int64_t b = (int64_t)L * M + 4 * L + 4; b + 4 <= stride; b += 4) *(uint32_t *)(r + b) = 0u;
    }
}

static int64_t graph_record_stride(int64_t L, int64_t M) { return ((L * M + 4 * L + 4 + 15) / 16) * 16; }

template <int M, int E, bool PACKED, bool BATCH, int W = 1>
static int launch_beam(const uint32_t *links, int lpn, const uint8_t *packed, const uint32_t *seeds, int n_seeds, const uint8_t *codes,
                       int64_t N, const uint32_t *valid, const float *lut, int64_t B, int64_t Ks, int ef, int hash_bits,
                       int64_t *out_ids, float *out_dist, unsigned long long *stats, hipStream_t st) {
    const size_t per_wave = (size_t)M * Ks * 4 + ((size_t)4 << hash_bits) + (BATCH ? (size_t)E * 64 * 12 + 1024 : 0);
    int wpb = (int)((size_t)160 * 1024 / per_wave);
    if (wpb > 4) wpb = 4;
    ANNLITE_REQUIRE(wpb >= 1, "M * Ks tables do not fit the LDS");
    const size_t lds = wpb * per_wave;
    auto fn = graph_beam_search_kernel<M, E, PACKED, BATCH, W>;
    ANNLITE_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)((B + wpb - 1) / wpb)), dim3(wpb * 64), lds, st, links, lpn, packed,
                       graph_record_stride(lpn, M), seeds, n_seeds, codes, N, valid, lut, (int)B, (int)Ks, ef, hash_bits, out_ids,
                       out_dist, stats);
    return launch_status("graph_beam_search_kernel");
}

static unsigned long long *g_graph_stats = nullptr;  // debug only (ANNLITE_DEBUG_COUNTERS): leaked 64-byte device buffer

extern "C" int annlite_graph_search_stats(uint64_t *out2) {
    ANNLITE_REQUIRE(out2 != nullptr, "out2 is NULL");
    if (!g_graph_stats) {
        set_error("no counters recorded (set ANNLITE_DEBUG_COUNTERS=1 before the walk)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    ANNLITE_HIP_TRY(hipMemcpy(out2, g_graph_stats, 16, hipMemcpyDeviceToHost));
    return ANNLITE_OK;
}

static int graph_search_impl(const uint32_t *links_dev, const uint8_t *packed_dev, int links_per_node, const uint32_t *seeds_dev,
                             int64_t n_seeds, const void *codes_dev, int64_t N, int64_t M, int64_t Ks, const uint32_t *valid_bits_dev,
                             const float *lut_bmk_dev, int64_t B, int ef, int64_t *out_ids_dev, float *out_dist_dev, void *stream,
                             int width = 1) {
    ANNLITE_REQUIRE(B >= 0 && N >= 0 && ef >= 1 && ef <= 256, "bad B=%lld N=%lld ef=%d (ef <= 256)", (long long)B,
                    (long long)N, ef);
    ANNLITE_REQUIRE(width == 1 || (width == 2 && packed_dev && links_per_node <= 32),
                    "expansion width %d: 1, or 2 over packed records of at most 32 neighbours (links_per_node = %d)", width, links_per_node);
    ANNLITE_REQUIRE(Ks >= 1 && Ks <= 256 && (M == 8 || M == 16 || M == 32 || M == 64), "graph search supports M in {8,16,32,64}, Ks <= 256");
    ANNLITE_REQUIRE(links_per_node >= 1 && n_seeds >= 1 && N < (1ll << 32) - 1, "bad graph");
    ANNLITE_REQUIRE(!packed_dev || links_per_node <= 64, "packed records hold at most 64 neighbours (links_per_node = %d)", links_per_node);
    if (B == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE((links_dev || packed_dev) && seeds_dev && codes_dev && lut_bmk_dev && out_ids_dev && out_dist_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    // 4096 entries up to ef = 128 (a walk records 2-3k nodes: 5M rows, ef 128 gave the same recall as 8192 entries at
    // 1.5x the speed -- 5 instead of 3 waves per CU), 8192 beyond; a full table only costs re-evaluations (see visit)
    // round 6: 4096 entries up to ef = 208 as well -- with 8192 entries beside a list of more than 128 entries only three waves fit a CU
    // and a 1024-query batch runs in two rounds (10M rows, ef 144 / 160 / 192: 0.594 / 0.658 / 0.755 ms per batch with 8192 entries and
    // four registers per lane, 0.387 / 0.437 / 0.544 with 4096, 0.360 / 0.405 / 0.510 with three registers per lane on top -- the table
    // fills towards the end of such a walk and the merge tests for duplicates from then on; same lists, same recall:
    // profiles/r06/graph_10m_ef_sweep.txt)
    int hash_bits = ef <= 208 ? 12 : 13;  // (ef 200, the reference's ef_construction: 0.785 -> 0.568 ms per 1024 walks at 10M rows)
    if (knobs().graph_hash_bits >= 0) hash_bits = knobs().graph_hash_bits;  // (ANNLITE_GRAPH_HASH_BITS: tests force a tiny table)
    if (hash_bits < 4) hash_bits = 4;    // (buckets of four entries, 16-byte initialisation)
    if (hash_bits > 15) hash_bits = 15;  // (128 KB: what is left of the LDS beside the smallest table)
    const uint8_t *codes = (const uint8_t *)codes_dev;
    unsigned long long *stats = nullptr;
    if (knobs().debug_counters) {
        if (!g_graph_stats) ANNLITE_HIP_TRY(hipMalloc((void **)&g_graph_stats, 64));
        ANNLITE_HIP_TRY(hipMemsetAsync(g_graph_stats, 0, 64, st));
        stats = g_graph_stats;
    }
#define ANNLITE_BEAM_ARGS links_dev, links_per_node, packed_dev, seeds_dev, (int)n_seeds, codes, N, valid_bits_dev, lut_bmk_dev, B, Ks, ef, \
                          hash_bits, out_ids_dev, out_dist_dev, stats, st
#define ANNLITE_BEAM(MM, PK, BT) \
    (ef <= 64 ? launch_beam<MM, 1, PK, BT>(ANNLITE_BEAM_ARGS) : ef <= 128 ? launch_beam<MM, 2, PK, BT>(ANNLITE_BEAM_ARGS) : launch_beam<MM, 4, PK, BT>(ANNLITE_BEAM_ARGS))
    // (pair walk: a list of three registers per lane for 128 < ef <= 192 -- every list operation is per register)
#define ANNLITE_BEAM2(MM) \
    (ef <= 64 ? launch_beam<MM, 1, true, true, 2>(ANNLITE_BEAM_ARGS) : ef <= 128 ? launch_beam<MM, 2, true, true, 2>(ANNLITE_BEAM_ARGS) : \
     ef <= 192 ? launch_beam<MM, 3, true, true, 2>(ANNLITE_BEAM_ARGS) : launch_beam<MM, 4, true, true, 2>(ANNLITE_BEAM_ARGS))
    if (packed_dev && width == 2) {
        if (M == 8) return ANNLITE_BEAM2(8);
        if (M == 16) return ANNLITE_BEAM2(16);
        if (M == 32) return ANNLITE_BEAM2(32);
        return ANNLITE_BEAM2(64);  // (round 6: 64 KB of table per wave -- one wave per CU: a 256-query batch fills the chip)
    }
#undef ANNLITE_BEAM2
    if (packed_dev) {
        if (M == 8) return ANNLITE_BEAM(8, true, true);
        if (M == 16) return ANNLITE_BEAM(16, true, true);
        if (M == 32) return ANNLITE_BEAM(32, true, true);
        return ANNLITE_BEAM(64, true, true);
    }
    // ANNLITE_GRAPH_SEQ_INSERT=1 (plain layout): the one-at-a-time list insertion of rounds 2-4 -- the parity tests walk with
    // both and compare
    if (knobs().graph_seq_insert) {
        if (M == 8) return ANNLITE_BEAM(8, false, false);
        if (M == 16) return ANNLITE_BEAM(16, false, false);
        if (M == 32) return ANNLITE_BEAM(32, false, false);
        return ANNLITE_BEAM(64, false, false);
    }
    if (M == 8) return ANNLITE_BEAM(8, false, true);
    if (M == 16) return ANNLITE_BEAM(16, false, true);
    if (M == 32) return ANNLITE_BEAM(32, false, true);
    return ANNLITE_BEAM(64, false, true);
#undef ANNLITE_BEAM
#undef ANNLITE_BEAM_ARGS
}

// [0] expansions, [1] rows evaluated, [2] prefetched records used; shader cycles summed over the queries' waves: [3] seed phase, packed
// walk: [4] pick + wait for the record, [5] visited table, [6] PQLookup sums, [7] list merge
extern "C" int annlite_graph_search_stats_ex(uint64_t *out4) {
    ANNLITE_REQUIRE(out4 != nullptr, "out8 is NULL");
    if (!g_graph_stats) {
        set_error("no counters recorded (set ANNLITE_DEBUG_COUNTERS=1 before the walk)");
        return ANNLITE_ERR_INVALID;
    }
    ANNLITE_HIP_TRY(hipDeviceSynchronize());
    ANNLITE_HIP_TRY(hipMemcpy(out4, g_graph_stats, 64, hipMemcpyDeviceToHost));
    return ANNLITE_OK;
}

extern "C" int annlite_graph_search(const uint32_t *links_dev, int links_per_node, const uint32_t *seeds_dev, int64_t n_seeds,
                                    const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                                    const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef,
                                    int64_t *out_ids_dev, float *out_dist_dev, void *stream) {
    return graph_search_impl(links_dev, nullptr, links_per_node, seeds_dev, n_seeds, codes_dev, N, M, Ks, valid_bits_dev, lut_bmk_dev,
                             B, ef, out_ids_dev, out_dist_dev, stream);
}

extern "C" int annlite_graph_record_bytes(int links_per_node, int64_t M, int64_t *bytes) {
    ANNLITE_REQUIRE(bytes != nullptr && links_per_node >= 1 && links_per_node <= 64 && (M == 8 || M == 16 || M == 32 || M == 64),
                    "packed records: links_per_node in [1, 64], M in {8,16,32,64}");
    *bytes = graph_record_stride(links_per_node, M);
    return ANNLITE_OK;
}

extern "C" int annlite_graph_pack(const uint32_t *links_dev, int links_per_node, const void *codes_dev, int64_t N, int64_t M,
                                  void *packed_dev, void *stream) {
    ANNLITE_REQUIRE(N >= 0 && links_per_node >= 1 && links_per_node <= 64 && (M == 8 || M == 16 || M == 32 || M == 64),
                    "packed records: links_per_node in [1, 64], M in {8,16,32,64}");
    if (N == 0) return ANNLITE_OK;
    ANNLITE_REQUIRE(links_dev && codes_dev && packed_dev, "null device pointer");
    const int64_t total = N * links_per_node;
    hipLaunchKernelGGL(graph_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, links_dev,
                       links_per_node, (const uint8_t *)codes_dev, N, (int)M, (uint8_t *)packed_dev, graph_record_stride(links_per_node, M));
    return launch_status("graph_pack_kernel");
}

extern "C" int annlite_graph_search_packed(const void *packed_dev, int links_per_node, const uint32_t *seeds_dev, int64_t n_seeds,
                                           const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                                           const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef,
                                           int64_t *out_ids_dev, float *out_dist_dev, void *stream) {
    ANNLITE_REQUIRE(packed_dev != nullptr || B == 0, "packed_dev is NULL");
    return graph_search_impl(nullptr, (const uint8_t *)packed_dev, links_per_node, seeds_dev, n_seeds, codes_dev, N, M, Ks,
                             valid_bits_dev, lut_bmk_dev, B, ef, out_ids_dev, out_dist_dev, stream);
}

extern "C" int annlite_graph_search_packed_ex(const void *packed_dev, int links_per_node, const uint32_t *seeds_dev, int64_t n_seeds,
                                              const void *codes_dev, int64_t N, int64_t M, int64_t Ks,
                                              const uint32_t *valid_bits_dev, const float *lut_bmk_dev, int64_t B, int ef,
                                              int expand_width, int64_t *out_ids_dev, float *out_dist_dev, void *stream) {
    ANNLITE_REQUIRE(packed_dev != nullptr || B == 0, "packed_dev is NULL");
    return graph_search_impl(nullptr, (const uint8_t *)packed_dev, links_per_node, seeds_dev, n_seeds, codes_dev, N, M, Ks,
                             valid_bits_dev, lut_bmk_dev, B, ef, out_ids_dev, out_dist_dev, stream, expand_width);
}
