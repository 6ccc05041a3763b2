// collector.cpp -- per-query top-N hit lists: what is left in the reference's HSP
// stream after the preliminary stage (rank 0 replays the gathered per-shard records
// through this in ascending OID order).
//
// Replaces BlastHSPStreamWrite -> s_BlastHSPCollectorRun -> Blast_HitListUpdate
// (CORE/blast_hspstream.c:316-365, CORE/hspfilter_collector.c:86-170,
// CORE/blast_hits.c:2924-2981) and the read-out of BlastHSPStreamClose
// (CORE/blast_hspstream.c:136-209).  The keep/evict decisions depend on arrival
// order and on a fuzzy (non-transitive) e-value comparison, so the worst-first heap
// below performs the same sift-down steps as the reference's (CORE/blast_hits.c:
// 1470-1521): std::make_heap / pop_heap would be a different, equally valid heap
// and would evict different lists when e-values tie within 1e-6.
#include "gbn_host.hpp"
#include "gbn_guard.hpp"
#include <algorithm>
#include <climits>
#include <memory>

namespace {

struct SubjectHits {                // one (query, subject) HSP list
    int32_t oid = 0, query = 0;
    double best_evalue = 0;
    std::vector<GbnHSP> hsps;       // sorted by score on arrival
};

inline int fuzzy_cmp(double a, double b) {          // CORE/blast_hits.c:1238-1253
    if (a < (1 - 1e-6) * b) return -1;
    if (a > (1 + 1e-6) * b) return 1;
    return 0;
}

inline int cmp3(int32_t a, int32_t b) { return a < b ? -1 : (a > b ? 1 : 0); }

int score_order(const GbnHSP &a, const GbnHSP &b) { // ScoreCompareHSPs, CORE/blast_hits.c:1182-1208
    if (int r = cmp3(b.score, a.score)) return r;
    if (int r = cmp3(a.s_offset, b.s_offset)) return r;
    if (int r = cmp3(b.s_end, a.s_end)) return r;
    if (int r = cmp3(a.q_offset, b.q_offset)) return r;
    return cmp3(b.q_end, a.q_end);
}

int evalue_order(const GbnHSP &a, const GbnHSP &b) {   // CORE/blast_hits.c:1263-1284
    if (int r = fuzzy_cmp(a.evalue, b.evalue)) return r;
    return score_order(a, b);
}

// > 0: a is the worse list (CORE/blast_hits.c:2757-2788)
int list_order(const SubjectHits &a, const SubjectHits &b) {
    if (a.hsps.empty() || b.hsps.empty()) return (int)a.hsps.empty() - (int)b.hsps.empty();
    if (int r = fuzzy_cmp(a.best_evalue, b.best_evalue)) return r;
    if (a.hsps[0].score != b.hsps[0].score) return a.hsps[0].score > b.hsps[0].score ? -1 : 1;
    return cmp3(b.oid, a.oid);
}

void order_by_evalue(SubjectHits &l) {                 // CORE/blast_hits.c:1286-1306
    auto &v = l.hsps;
    bool sorted = true;
    for (size_t i = 0; i + 1 < v.size(); i++) if (evalue_order(v[i], v[i + 1]) > 0) { sorted = false; break; }
    if (!sorted) std::stable_sort(v.begin(), v.end(), [](const GbnHSP &x, const GbnHSP &y) { return evalue_order(x, y) < 0; });
}

struct QueryHits {
    std::vector<std::unique_ptr<SubjectHits>> lists;
    double worst_evalue = 0; int32_t low_score = INT32_MAX; bool heap = false;

    // worst list at index 0; `node` sinks while a child is worse, the worse child
    // being the LEFT one on ties -- the reference's sift-down
    void sink(size_t node, size_t last_parent, size_t last) {
        while (node <= last_parent) {
            size_t l = 2 * node + 1, pick = l;
            if (l != last && list_order(*lists[l], *lists[l + 1]) < 0) pick = l + 1;
            if (list_order(*lists[node], *lists[pick]) >= 0) break;
            std::swap(lists[node], lists[pick]);
            node = pick;
        }
    }
    void build_heap() {
        size_t n = lists.size();
        if (n < 2) return;
        for (size_t i = n / 2; i-- > 0;) sink(i, (n - 2) / 2, n - 1);
    }
    void offer(std::unique_ptr<SubjectHits> l, size_t cap) {
        l->best_evalue = (double)INT32_MAX;
        for (auto &h : l->hsps) l->best_evalue = std::min(h.evalue, l->best_evalue);
        if (lists.size() < cap) {
            worst_evalue = std::max(l->best_evalue, worst_evalue);
            low_score = std::min(l->hsps[0].score, low_score);
            lists.push_back(std::move(l));
            return;
        }
        order_by_evalue(*l);
        int ord = fuzzy_cmp(l->best_evalue, worst_evalue);
        if (ord > 0 || (ord == 0 && l->hsps[0].score < low_score)) return;    // worse than everything kept
        if (!heap) { for (auto &x : lists) order_by_evalue(*x); build_heap(); heap = true; }
        lists[0] = std::move(l);
        size_t n = lists.size();
        if (n >= 2) sink(0, n / 2 - 1, n - 1);
        worst_evalue = lists[0]->best_evalue;
        low_score = lists[0]->hsps[0].score;
    }
};

}  // namespace

struct GbnCollector {
    int32_t nq = 0; size_t cap = 0; bool closed = false;
    std::vector<QueryHits> per_query;
    std::vector<GbnHSP> out; std::vector<int64_t> list_start; std::vector<int32_t> list_query;
};

extern "C" {

int32_t gbn_prelim_hitlist_size(int32_t hitlist_size) {
    return gbn::guard_as<int32_t>(__func__, (int32_t)-1, (int32_t)-1, [&]() -> int32_t {
    // SBlastHitsParametersNew, CORE/blast_hits.c:45-76 (gapped, no composition statistics)
    return std::max(std::min(2 * hitlist_size, hitlist_size + 50), 10);
    });
}

int gbn_collector_new(GbnCollector **out, int32_t num_queries, int32_t hitlist_size) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || num_queries <= 0 || hitlist_size <= 0) { gbn::set_error("gbn_collector_new: bad argument"); return GBN_ERR_ARG; }
    auto *c = new GbnCollector();
    c->nq = num_queries; c->cap = (size_t)gbn_prelim_hitlist_size(hitlist_size);
    c->per_query.resize((size_t)num_queries);
    *out = c;
    return GBN_OK;
    });
}

void gbn_collector_free(GbnCollector *c) { delete c; }

// records grouped by oid (as gbn_results_hsps yields them); every oid group is one
// BlastHSPStreamWrite.  Writing after close is an error (CORE/blast_hspstream.c:332-335).
int gbn_collector_write(GbnCollector *c, const GbnHSP *h, int64_t n) {
    return gbn::guard(__func__, [&]() -> int {
    gbn::CpuScope cpu(gbn::GBN_CPU_END_COLLECT);
    if (!c || (n > 0 && !h)) { gbn::set_error("gbn_collector_write: bad argument"); return GBN_ERR_ARG; }
    if (c->closed) { gbn::set_error("gbn_collector_write: collector already closed"); return GBN_ERR_ARG; }
    std::vector<std::unique_ptr<SubjectHits>> split((size_t)c->nq);
    std::vector<int32_t> touched;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        touched.clear();
        for (; j < n && h[j].oid == h[i].oid; j++) {
            int32_t q = h[j].context / 2;               // Blast_GetQueryIndexFromContext, blastn
            if (q < 0 || q >= c->nq) { gbn::set_error("gbn_collector_write: context out of range"); return GBN_ERR_ARG; }
            if (!split[q]) { split[q].reset(new SubjectHits()); split[q]->oid = h[i].oid; split[q]->query = q; touched.push_back(q); }
            split[q]->hsps.push_back(h[j]);
        }
        std::sort(touched.begin(), touched.end());      // hit lists are updated in query order
        for (int32_t q : touched) c->per_query[q].offer(std::move(split[q]), c->cap);
        i = j;
    }
    return GBN_OK;
    });
}

// surviving lists in (oid, query) ascending order -- the order BlastHSPStreamRead hands
// them to the traceback stage (ascending oid)
int gbn_collector_close(GbnCollector *c) {
    return gbn::guard(__func__, [&]() -> int {
    gbn::CpuScope cpu(gbn::GBN_CPU_END_COLLECT);
    if (!c) return GBN_ERR_ARG;
    if (c->closed) return GBN_OK;
    std::vector<const SubjectHits *> all;
    for (auto &q : c->per_query) for (auto &l : q.lists) all.push_back(l.get());
    std::sort(all.begin(), all.end(), [](const SubjectHits *a, const SubjectHits *b) {
        return a->oid != b->oid ? a->oid < b->oid : a->query < b->query; });
    for (auto *l : all) {
        c->list_start.push_back((int64_t)c->out.size());
        c->list_query.push_back(l->query);
        c->out.insert(c->out.end(), l->hsps.begin(), l->hsps.end());
    }
    c->list_start.push_back((int64_t)c->out.size());
    c->closed = true;
    return GBN_OK;
    });
}

int64_t gbn_collector_num_lists(const GbnCollector *c) { return c && c->closed ? (int64_t)c->list_query.size() : 0; }
const int64_t *gbn_collector_list_starts(const GbnCollector *c) { return c->list_start.data(); }
const int32_t *gbn_collector_list_queries(const GbnCollector *c) { return c->list_query.data(); }
int64_t gbn_collector_num_hsps(const GbnCollector *c) { return c && c->closed ? (int64_t)c->out.size() : 0; }
const GbnHSP *gbn_collector_hsps(const GbnCollector *c) { return c->out.data(); }

}  // extern "C"
