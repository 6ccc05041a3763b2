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

// Blast_HSPListsMerge (CORE/blast_hits.c:2545-2716) for a subject searched in chunks: `cur` = the list of the chunk
// that starts at `split` (sequence coordinates already), `comb` = what the chunks before it left.  HSPs that reach
// into the overlap strip come to the front of either list (in the reference's swap order, which fixes the order the
// pairs are tried in); a pair of one context whose end / start diagonals are closer than OVERLAP_DIAG_CLOSE and
// that touch (s_BlastMergeTwoHSPs, :1337-1378) becomes one HSP with the joint extent and the better score.
static void merge_two_lists(std::vector<GbnHSP> &comb, std::vector<GbnHSP> &cur, int32_t split, int32_t overlap)
{
    if (cur.empty()) return;
    if (comb.empty()) { comb.swap(cur); return; }
    size_t n1 = 0, n2 = 0;
    for (size_t i = 0; i < comb.size(); i++) if (comb[i].s_end > split) { std::swap(comb[n1], comb[i]); n1++; }
    for (size_t i = 0; i < cur.size(); i++) if (cur[i].s_offset < split + overlap) { std::swap(cur[n2], cur[i]); n2++; }
    std::vector<char> gone(cur.size(), 0);
    auto inside = [](int32_t a, int32_t b, int32_t c, int32_t d, int32_t e, int32_t f) { return a <= c && b >= c && d <= f && e >= f; };
    for (size_t i = 0; i < n1; i++) {
        GbnHSP &h1 = comb[i];
        for (size_t j = 0; j < n2; j++) {
            if (gone[j] || h1.context != cur[j].context) continue;
            const GbnHSP &h2 = cur[j];
            const int32_t end_diag = h1.q_end - h1.s_end, start_diag = h2.q_offset - h2.s_offset;
            if (std::abs(end_diag - start_diag) >= 10) continue;                     // OVERLAP_DIAG_CLOSE
            if (inside(h1.q_offset, h1.q_end, h2.q_offset, h1.s_offset, h1.s_end, h2.s_offset) ||
                inside(h1.q_offset, h1.q_end, h2.q_end, h1.s_offset, h1.s_end, h2.s_end)) {
                h1.q_offset = std::min(h1.q_offset, h2.q_offset); h1.s_offset = std::min(h1.s_offset, h2.s_offset);
                h1.q_end = std::max(h1.q_end, h2.q_end); h1.s_end = std::max(h1.s_end, h2.s_end);
                if (h2.score > h1.score) { h1.q_gapped_start = h2.q_gapped_start; h1.s_gapped_start = h2.s_gapped_start; h1.score = h2.score; h1.evalue = h2.evalue; }
                gone[j] = 1;
            }
        }
    }
    for (size_t j = 0; j < cur.size(); j++) if (!gone[j]) comb.push_back(cur[j]);
    std::stable_sort(comb.begin(), comb.end(), by_score);
    cur.clear();
}

// The chunk lists of a sequence -> its one list: merged in chunk order, then what the engine does with a subject's
// list after its chunk loop (GB/gpu_blastn_pre_search_engine.cpp:772-810): e-values, e-value reap, counters.
void merge_chunk_lists(std::vector<GbnHSP> &hsps, int32_t chunk_len, const GbnResults::ChunkMerge &b, GbnDiagnostics *diag)
{
    std::vector<GbnHSP> out; out.reserve(hsps.size());
    const int32_t stride = chunk_len - kDbseqChunkOverlap;
    for (size_t i = 0; i < hsps.size();) {
        if (hsps[i].pad_ == 0) { out.push_back(hsps[i++]); continue; }
        // the chunk lists of one sequence follow each other in chunk order
        const int32_t oid = hsps[i].oid;
        std::vector<GbnHSP> comb;
        while (i < hsps.size() && hsps[i].pad_ != 0 && hsps[i].oid == oid) {
            const int32_t ord = hsps[i].pad_ - 1;
            std::vector<GbnHSP> cur;
            while (i < hsps.size() && hsps[i].oid == oid && hsps[i].pad_ == ord + 1) { cur.push_back(hsps[i]); cur.back().pad_ = 0; i++; }
            // the first chunk starts where its range does: no overlap strip on its left (CORE/blast_engine.c:532-533)
            merge_two_lists(comb, cur, ord * stride, ord == 0 ? 0 : kDbseqChunkOverlap);
        }
        size_t kept = 0;
        for (GbnHSP &h : comb) {
            h.evalue = evalue_for_score(h.score, b.kbp_gap, b.eff_searchsp[(size_t)h.context]);
            if (h.evalue > b.evalue) continue;
            out.push_back(h); kept++;
        }
        if (diag && kept) { diag->seqs_passed++; diag->good_extensions += (int64_t)kept; }
    }
    hsps.swap(out);
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
                    std::vector<GbnHSP> &out, GbnDiagnostics *diag, bool chunk)
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
    if (chunk) { out.insert(out.end(), accepted.begin(), accepted.end()); return; }
    size_t kept = 0;
    for (auto &h : accepted) {
        h.evalue = evalue_for_score(h.score, b.kbp_gap, b.ctx[h.context].eff_searchsp);
        if (h.evalue > b.opt.evalue) continue;
        out.push_back(h); kept++;
    }
    if (diag && kept) { diag->seqs_passed++; diag->good_extensions += (int64_t)kept; }
}

}  // namespace gbn
