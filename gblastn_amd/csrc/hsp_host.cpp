// hsp_host.cpp -- host side of the gapped stage: BLAST_GetGappedScore's
// sequential acceptance loop replayed over gapped extensions that the GPU has
// already computed for every initial hit, then the per-subject HSP list rules.
//
// The GPU extends ALL initial hits speculatively (an extension does not depend
// on earlier ones); only the decision "was this initial hit skipped because an
// already accepted HSP envelops it" is order dependent, and that is replayed
// here in the reference's order (CORE/blast_gapalign.c:3351-3548) with an index
// that makes the same decisions as its interval tree (CORE/blast_itree.c).
#include "gbn_host.hpp"
#include "gbn_dev.h"
#include "hsp_host.hpp"
#include <algorithm>
#include <cstring>

namespace gbn {

// ---------------------------------------------------------------------------
// Envelope index: a two-level centred interval tree (query range, then subject
// range) with the reference's insertion, common-endpoint eviction and
// containment rules, including their order dependence.
// ---------------------------------------------------------------------------
class EnvelopeIndex {
    struct Node { int32_t lo, hi, left, mid, right, hsp; };   // hsp < 0: internal
    std::vector<Node> nd_;
    const std::vector<GbnHSP> *pool_ = nullptr;
    int32_t smin_ = 0, smax_ = 0;
    const GbnBatch *b_;

    int32_t raw_node() { nd_.push_back(Node{0, 0, 0, 0, 0, -1}); return (int32_t)nd_.size() - 1; }
    int32_t child(int32_t parent, bool left_half) {
        int32_t i = raw_node();
        int32_t mid = (nd_[parent].lo + nd_[parent].hi) / 2;
        if (left_half) { nd_[i].lo = nd_[parent].lo; nd_[i].hi = mid; }
        else { nd_[i].lo = mid + 1; nd_[i].hi = nd_[parent].hi; }
        return i;
    }
    int32_t root(int32_t a, int32_t b) { int32_t i = raw_node(); nd_[i].lo = a; nd_[i].hi = b; return i; }
    const GbnHSP &H(int32_t node) const { return (*pool_)[nd_[node].hsp]; }

    int32_t strand_start(int32_t context) const {
        int32_t c = context;
        while (c) {
            int32_t f = b_->ctx[c].frame, fp = b_->ctx[c - 1].frame;
            if (f == 0 || ((f > 0) != (fp > 0))) break;
            c--;
        }
        return b_->ctx[c].query_offset;
    }
    // 0 = no shared end, 1 = newcomer wins, 2 = resident wins
    static int shared_end(const GbnHSP &in, int32_t inq, const GbnHSP &tr, int32_t trq, bool left_end) {
        if (inq != trq) return 0;
        bool same = left_end ? (in.q_offset == tr.q_offset && in.s_offset == tr.s_offset)
                             : (in.q_end == tr.q_end && in.s_end == tr.s_end);
        if (!same) return 0;
        if (in.score != tr.score) return in.score > tr.score ? 1 : 2;
        int32_t a = in.q_end - in.q_offset, b = tr.q_end - tr.q_offset;
        if (a != b) return a > b ? 2 : 1;
        a = in.s_end - in.s_offset; b = tr.s_end - tr.s_offset;
        if (a != b) return a > b ? 2 : 1;
        return 2;
    }
    bool subject_level_blocks(int32_t rootn, const GbnHSP &in, int32_t inq, bool left_end) {
        const int32_t target = left_end ? in.s_offset : in.s_end;
        int32_t r = rootn;
        for (;;) {
            int32_t t = nd_[r].mid, prevn = r, cur = t;
            while (t != 0) {
                int w = shared_end(in, inq, H(cur), nd_[cur].left, left_end);
                t = nd_[cur].mid;
                if (w == 2) return true;
                if (w == 1) nd_[prevn].mid = t;
                prevn = cur; cur = t;
            }
            int32_t mid = (nd_[r].lo + nd_[r].hi) / 2, nx = 0;
            if (target < mid) nx = nd_[r].left; else if (target > mid) nx = nd_[r].right;
            if (nx == 0) return false;
            if (nd_[nx].hsp >= 0) {
                int w = shared_end(in, inq, H(nx), nd_[nx].left, left_end);
                if (w == 2) return true;
                if (w == 1) { if (target < mid) nd_[r].left = 0; else if (target > mid) nd_[r].right = 0; }
                return false;
            }
            r = nx;
        }
    }
    bool query_level_blocks(const GbnHSP &in, int32_t inq, bool left_end) {
        const int32_t target = left_end ? inq + in.q_offset : inq + in.q_end;
        int32_t r = 0;
        for (;;) {
            int32_t t = nd_[r].mid;
            if (t != 0 && subject_level_blocks(t, in, inq, left_end)) return true;
            int32_t mid = (nd_[r].lo + nd_[r].hi) / 2, nx = 0;
            if (target < mid) nx = nd_[r].left; else if (target > mid) nx = nd_[r].right;
            if (nx == 0) return false;
            if (nd_[nx].hsp >= 0) {
                int w = shared_end(in, inq, H(nx), nd_[nx].left, left_end);
                if (w == 2) return true;
                if (w == 1) { if (target < mid) nd_[r].left = 0; else if (target > mid) nd_[r].right = 0; }
                return false;
            }
            r = nx;
        }
    }
    static bool envelops(const GbnHSP &in, int32_t inq, const GbnHSP &tr, int32_t trq, int32_t mds) {
        if (inq != trq) return false;
        auto inside = [&](int32_t qp, int32_t sp) {
            return tr.q_offset <= qp && tr.q_end >= qp && tr.s_offset <= sp && tr.s_end >= sp;
        };
        if (in.score <= tr.score && inside(in.q_offset, in.s_offset) && inside(in.q_end, in.s_end)) {
            if (mds == 0) return true;
            int32_t d1 = std::abs((tr.q_offset - tr.s_offset) - (in.q_offset - in.s_offset));
            int32_t d2 = std::abs((tr.q_end - tr.s_end) - (in.q_end - in.s_end));
            return d1 < mds || d2 < mds;
        }
        return false;
    }
    bool subject_level_envelops(int32_t rootn, const GbnHSP &in, int32_t inq, int32_t mds) const {
        int32_t n = rootn;
        while (nd_[n].hsp < 0) {
            for (int32_t t = nd_[n].mid; t != 0; t = nd_[t].mid)
                if (envelops(in, inq, H(t), nd_[t].left, mds)) return true;
            int32_t mid = (nd_[n].lo + nd_[n].hi) / 2, nx = 0;
            if (in.s_end < mid) nx = nd_[n].left; else if (in.s_offset > mid) nx = nd_[n].right;
            if (nx == 0) return false;
            n = nx;
        }
        return envelops(in, inq, H(n), nd_[n].left, mds);
    }

public:
    EnvelopeIndex(const GbnBatch *b, const std::vector<GbnHSP> *pool, int32_t qmax, int32_t smax)
        : pool_(pool), smin_(0), smax_(smax), b_(b) { nd_.reserve(128); root(0, qmax); }

    bool enveloped(const GbnHSP &in, int32_t mds) const {
        const int32_t inq = strand_start(in.context);
        const int32_t rs = inq + in.q_offset, re = inq + in.q_end;
        int32_t n = 0;
        while (nd_[n].hsp < 0) {
            int32_t t = nd_[n].mid;
            if (t > 0 && subject_level_envelops(t, in, inq, mds)) return true;
            int32_t mid = (nd_[n].lo + nd_[n].hi) / 2, nx = 0;
            if (re < mid) nx = nd_[n].left; else if (rs > mid) nx = nd_[n].right;
            if (nx == 0) return false;
            n = nx;
        }
        return envelops(in, inq, H(n), nd_[n].left, mds);
    }

    void insert(int32_t hsp_idx) {
        const GbnHSP hsp = (*pool_)[hsp_idx];
        const int32_t inq = strand_start(hsp.context);
        if (query_level_blocks(hsp, inq, true)) return;
        if (query_level_blocks(hsp, inq, false)) return;
        int32_t rs = inq + hsp.q_offset, re = inq + hsp.q_end;
        int32_t leaf = raw_node();
        nd_[leaf].left = inq; nd_[leaf].hsp = hsp_idx;
        int32_t r = 0; bool on_subject = false;
        for (;;) {
            int32_t mid = (nd_[r].lo + nd_[r].hi) / 2, old; bool left_half;
            if (re < mid) {
                if (nd_[r].left == 0) { nd_[r].left = leaf; return; }
                old = nd_[r].left;
                if (nd_[old].hsp < 0) { r = old; continue; }
                left_half = true;
            } else if (rs > mid) {
                if (nd_[r].right == 0) { nd_[r].right = leaf; return; }
                old = nd_[r].right;
                if (nd_[old].hsp < 0) { r = old; continue; }
                left_half = false;
            } else {
                if (on_subject) { nd_[leaf].mid = nd_[r].mid; nd_[r].mid = leaf; return; }
                on_subject = true;
                if (nd_[r].mid == 0) { int32_t m = root(smin_, smax_); nd_[r].mid = m; }
                r = nd_[r].mid;
                rs = hsp.s_offset; re = hsp.s_end;
                continue;
            }
            int32_t m = child(r, left_half);
            if (left_half) nd_[r].left = m; else nd_[r].right = m;
            const GbnHSP oh = H(old);
            int32_t ors, ore;
            if (on_subject) { ors = oh.s_offset; ore = oh.s_end; }
            else { ors = nd_[old].left + oh.q_offset; ore = nd_[old].left + oh.q_end; }
            r = m;
            mid = (nd_[r].lo + nd_[r].hi) / 2;
            if (ore < mid) nd_[m].left = old;
            else if (ors > mid) nd_[m].right = old;
            else if (on_subject) nd_[m].mid = old;
            else {
                int32_t m2 = root(smin_, smax_);
                nd_[m].mid = m2;
                int32_t mid2 = (nd_[m2].lo + nd_[m2].hi) / 2;
                if (oh.s_end < mid2) nd_[m2].left = old;
                else if (oh.s_offset > mid2) nd_[m2].right = old;
                else nd_[m2].mid = old;
            }
        }
    }
};

// ---------------------------------------------------------------------------
// HSP list rules (CORE/blast_hits.c:2037-2302, :1182-1236, :2734-2750)
// ---------------------------------------------------------------------------
static bool by_query_start(const GbnHSP &a, const GbnHSP &b) {
    if (a.context != b.context) return a.context < b.context;
    if (a.q_offset != b.q_offset) return a.q_offset < b.q_offset;
    if (a.s_offset != b.s_offset) return a.s_offset < b.s_offset;
    if (a.score != b.score) return a.score > b.score;
    if (a.q_end != b.q_end) return a.q_end > b.q_end;
    if (a.s_end != b.s_end) return a.s_end > b.s_end;
    return false;
}
static bool by_query_end(const GbnHSP &a, const GbnHSP &b) {
    if (a.context != b.context) return a.context < b.context;
    if (a.q_end != b.q_end) return a.q_end < b.q_end;
    if (a.s_end != b.s_end) return a.s_end < b.s_end;
    if (a.score != b.score) return a.score > b.score;
    if (a.q_offset != b.q_offset) return a.q_offset > b.q_offset;
    if (a.s_offset != b.s_offset) return a.s_offset > b.s_offset;
    return false;
}
static bool by_score(const GbnHSP &a, const GbnHSP &b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.s_offset != b.s_offset) return a.s_offset < b.s_offset;
    if (a.s_end != b.s_end) return a.s_end > b.s_end;
    if (a.q_offset != b.q_offset) return a.q_offset < b.q_offset;
    return a.q_end > b.q_end;
}

void purge_common_endpoints(std::vector<GbnHSP> &v) {
    std::stable_sort(v.begin(), v.end(), by_query_start);
    auto same_start = [](const GbnHSP &a, const GbnHSP &b) {
        return a.context == b.context && a.q_offset == b.q_offset && a.s_offset == b.s_offset;
    };
    v.erase(std::unique(v.begin(), v.end(), same_start), v.end());
    std::stable_sort(v.begin(), v.end(), by_query_end);
    auto same_end = [](const GbnHSP &a, const GbnHSP &b) {
        return a.context == b.context && a.q_end == b.q_end && a.s_end == b.s_end;
    };
    v.erase(std::unique(v.begin(), v.end(), same_end), v.end());
}

void sort_by_score(std::vector<GbnHSP> &v) {
    if (!std::is_sorted(v.begin(), v.end(), by_score)) std::stable_sort(v.begin(), v.end(), by_score);
}

// initial-hit order: score desc, s_start asc, length desc, q_start asc, then
// arrival (scan) order -- CORE/blast_extend.c:259-315 under a stable qsort
static bool ihit_before(const GbnDevInitHit &a, const GbnDevInitHit &b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.s_start != b.s_start) return a.s_start < b.s_start;
    if (a.length != b.length) return a.length > b.length;
    if (a.q_start != b.q_start) return a.q_start < b.q_start;
    return a.seq < b.seq;
}

void finish_subject(const GbnBatch &b, int32_t oid, int32_t slen,
                    std::vector<std::pair<GbnDevInitHit, GbnDevGapped>> &hits,
                    std::vector<GbnHSP> &out, GbnDiagnostics *diag)
{
    std::sort(hits.begin(), hits.end(), [](const auto &x, const auto &y) { return ihit_before(x.first, y.first); });
    std::vector<GbnHSP> accepted;
    EnvelopeIndex index(&b, &accepted, b.qlen + 1, slen + 1);
    for (auto &pr : hits) {
        const GbnDevInitHit &ih = pr.first; const GbnDevGapped &g = pr.second;
        const int32_t context = g.context, qstart = b.ctx[context].query_offset;
        GbnHSP probe{};
        probe.context = context; probe.score = ih.score;
        probe.q_offset = ih.q_start - qstart; probe.q_end = probe.q_offset + ih.length;
        probe.s_offset = ih.s_start; probe.s_end = ih.s_start + ih.length;
        if (index.enveloped(probe, b.opt.min_diag_separation)) continue;
        if (diag) diag->gapped_extensions++;
        if (g.score >= b.ctx[context].gap_cutoff_score) {
            GbnHSP h{};
            h.oid = oid; h.context = context; h.score = g.score;
            h.q_offset = g.q_start; h.q_end = g.q_stop; h.s_offset = g.s_start; h.s_end = g.s_stop;
            h.q_gapped_start = g.seed_q; h.s_gapped_start = g.seed_s;
            accepted.push_back(h);
            index.insert((int32_t)accepted.size() - 1);
        }
    }
    purge_common_endpoints(accepted);
    if (b.round_down) for (auto &h : accepted) h.score &= ~1;
    sort_by_score(accepted);
    size_t kept = 0;
    for (auto &h : accepted) {
        h.evalue = evalue_for_score(h.score, b.kbp_gap, b.ctx[h.context].eff_searchsp);
        if (h.evalue > b.opt.evalue) continue;
        out.push_back(h); kept++;
    }
    if (diag && kept) { diag->seqs_passed++; diag->good_extensions += (int64_t)kept; }
}

}  // namespace gbn
