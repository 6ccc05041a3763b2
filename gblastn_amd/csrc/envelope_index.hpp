// envelope_index.hpp -- the containment index of the gapped and traceback stages (CORE/blast_itree.c restated).
#pragma once
#include "gbn_host.hpp"
#include <cstdlib>
#include <vector>

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


}  // namespace gbn
