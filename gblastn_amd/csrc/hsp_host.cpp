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
#include "envelope_index.hpp"
#include <algorithm>
#include <cstring>

namespace gbn {

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
