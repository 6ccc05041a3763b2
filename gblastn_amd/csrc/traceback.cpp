// traceback.cpp -- the traceback stage of a nucleotide search on the host (the reference keeps it on the
// CPU too; in pipeline mode it is the consumer that overlaps the GPU's preliminary stage).
//
// For every (query, subject) list the preliminary stage left in the collector: unpack the subject, redo each
// HSP's gapped extension with the final X-drop and an edit script, apply the list rules, compute identities,
// e-values and bit scores.  Replaces BLAST_ComputeTraceback -> Blast_TracebackFromHSPList ->
// {BLAST_GappedAlignmentWithTraceback -> ALIGN_EX | BLAST_GreedyGappedAlignment with traceback} ->
// s_HSPListPostTracebackUpdate (CORE/blast_traceback.c:1375-1639, :336-790, :278-334;
// CORE/blast_gapalign.c:350-708, :3994-4155, :2619-2751; CORE/greedy_align.c:385-753) and the final order of the
// results (CORE/blast_hits.c:2757-2788, CORE/blast_traceback.c:907-922).
//
// Greedy traceback covers both forms (gap costs 0 / 0 and BLAST_AffineGreedyAlign with explicit costs).  Subjects
// come from the 2-bit shard; the ambiguity codes of a real database travel with the shard as runs
// (gbn_db_set_ambiguities) and are put back over the stretch read from HBM before anything is aligned.
#include <hip/hip_runtime.h>
#include "gbn_host.hpp"
#include "gbn_guard.hpp"
#include "envelope_index.hpp"
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

namespace gbn {
namespace {

enum : uint8_t { kDel = 0, kSub = 3, kIns = 6 };     // EGapAlignOpType values (COREI/gapinfo.h): gap in query / pair / gap in subject
struct EditOp { uint8_t op; int32_t n; };
typedef std::vector<EditOp> Script;
inline void append_op(Script &s, uint8_t op, int32_t n) {       // GapPrelimEditBlockAdd
    if (n <= 0) return;
    if (!s.empty() && s.back().op == op) s.back().n += n; else s.push_back(EditOp{op, n});
}
// left half as produced (far end first), right half produced near end first: reversed, joined (CORE/blast_gapalign.c:2456-2516)
Script join_halves(const Script &left, const Script &right) {
    Script out(left);
    for (size_t i = right.size(); i-- > 0;) {
        if (i + 1 == right.size() && !out.empty() && out.back().op == right[i].op) out.back().n += right[i].n;
        else out.push_back(right[i]);
    }
    return out;
}

struct Extent { int32_t q_start = 0, q_stop = 0, s_start = 0, s_stop = 0, score = 0; };
const int32_t kFloor = INT32_MIN / 2;

// ---------------------------------------------------------------------------------------------------
// One quadrant of the X-drop dynamic programme with its actions kept (ALIGN_EX).  Rows = query letters, columns
// = subject letters; `mirrored`: both sequences are walked backwards from their ends.  The actions of a row
// start at its first live column; the path from the best cell back to the origin is appended to `path`.
// ---------------------------------------------------------------------------------------------------
struct Cell { int32_t best, gap; };
int32_t xdrop_quadrant(const int32_t (*mat)[16], const uint8_t *A, const uint8_t *B, int32_t M, int32_t N, bool mirrored,
                       int32_t X, int32_t gap_open, int32_t gap_extend, int32_t &a_best, int32_t &b_best, Script &path)
{
    enum : uint8_t { kOpMask = 0x07, kStayA = 0x10, kStayB = 0x40 };
    a_best = 0; b_best = 0;
    const int32_t goe = gap_open + gap_extend;
    if (X < goe) X = goe;
    if (N <= 0 || M <= 0) return 0;
    const int32_t spare = gap_extend > 0 ? X / gap_extend + 3 : N + 3;
    std::vector<Cell> cell((size_t)spare + 100);
    std::vector<uint8_t> acts;                      // all rows' actions, back to back
    struct RowRef { size_t at; int32_t first; };
    std::vector<RowRef> rows; rows.reserve(256);
    rows.push_back(RowRef{0, 0});
    acts.assign((size_t)spare + 4, 0);
    cell[0] = Cell{0, -goe};
    int32_t width = 1;
    for (int32_t s = -goe; width <= N && s >= -X; width++, s -= gap_extend) {
        cell[(size_t)width] = Cell{s, s - goe};
        acts[(size_t)width] = kDel;
    }
    int32_t first = 0, best = 0;
    for (int32_t a = 1; a <= M; a++) {
        const size_t span = (size_t)(gap_extend > 0 ? width - first + spare : N + 3 - first) + 4;
        rows.push_back(RowRef{acts.size(), first});
        acts.resize(acts.size() + span);
        uint8_t *act = acts.data() + rows.back().at - first;       // act[b]
        const int32_t *mrow = mat[mirrored ? A[M - a] : A[a]];
        int32_t score = kFloor, gap_row = kFloor, last = first;
        const int32_t row_first = first;
        int32_t b = row_first;
        for (; b < width; b++) {
            const uint8_t letter = mirrored ? B[N - 1 - b] : B[b + 1];
            int32_t gap_col = cell[(size_t)b].gap;
            const int32_t diag_next = cell[(size_t)b].best + mrow[letter];
            uint8_t what = kSub;
            if (score < gap_col) { what = kIns; score = gap_col; }
            if (score < gap_row) { what = kDel; score = gap_row; }
            if (best - score > X) {
                if (first == b) first++; else cell[(size_t)b].best = kFloor;
            } else {
                last = b;
                if (score > best) { best = score; a_best = a; b_best = b; }
                gap_row -= gap_extend; gap_col -= gap_extend;
                if (gap_col < score - goe) cell[(size_t)b].gap = score - goe;
                else { cell[(size_t)b].gap = gap_col; what += kStayB; }
                if (gap_row < score - goe) gap_row = score - goe; else what += kStayA;
                cell[(size_t)b].best = score;
            }
            score = diag_next;
            act[b] = what;
        }
        if (first == width) break;
        if ((size_t)(last + spare + 3) >= cell.size()) cell.resize(std::max<size_t>((size_t)(last + spare + 100), 2 * cell.size()));
        if (last < width - 1) width = last + 1;
        else for (; gap_row >= best - X && width <= N; width++, gap_row -= gap_extend) {
            cell[(size_t)width] = Cell{gap_row, gap_row - goe};
            act[width] = kDel;
        }
        if (width <= N) { cell[(size_t)width] = Cell{kFloor, kFloor}; width++; }
    }
    // walk back
    int32_t a = a_best, b = b_best; uint8_t state = kSub;
    while (a > 0 || b > 0) {
        const uint8_t rec = acts[rows[(size_t)a].at + (size_t)(b - rows[(size_t)a].first)];
        if (state == kDel) state = (rec & kStayA) ? kDel : (rec & kOpMask);
        else if (state == kIns) state = (rec & kStayB) ? kIns : (rec & kOpMask);
        else state = rec & kOpMask;
        if (state == kDel) b--; else if (state == kIns) a--; else { a--; b--; }
        append_op(path, state, 1);
    }
    return best;
}

// BLAST_GappedAlignmentWithTraceback: left quadrant includes the start point, right one starts after it
Extent gapped_traceback(const int32_t (*mat)[16], const uint8_t *q, const uint8_t *s, int32_t qlen, int32_t slen,
                        int32_t q0, int32_t s0, int32_t X, int32_t gap_open, int32_t gap_extend, Script &script)
{
    Extent e; Script left, right; int32_t da = 0, db = 0;
    int32_t score_left = xdrop_quadrant(mat, q, s, q0 + 1, s0 + 1, true, X, gap_open, gap_extend, da, db, left);
    e.q_start = q0 - da + 1; e.s_start = s0 - db + 1;
    int32_t score_right = 0;
    if (q0 < qlen && s0 < slen) {
        score_right = xdrop_quadrant(mat, q + q0, s + s0, qlen - q0 - 1, slen - s0 - 1, false, X, gap_open, gap_extend, da, db, right);
        e.q_stop = q0 + da + 1; e.s_stop = s0 + db + 1;
    } else { e.q_stop = q0 - 1; e.s_stop = s0 - 1; }
    script = join_halves(left, right);
    // a gap at either end is given back (CORE/blast_gapalign.c:4115-4151)
    if (!script.empty() && script.front().op != kSub) {
        score_left += gap_open + script.front().n * gap_extend;
        if (script.front().op == kDel) e.s_start += script.front().n; else e.q_start += script.front().n;
        script.erase(script.begin());
    }
    if (!script.empty() && script.back().op != kSub) {
        score_right += gap_open + script.back().n * gap_extend;
        if (script.back().op == kDel) e.s_stop -= script.back().n; else e.q_stop -= script.back().n;
        script.pop_back();
    }
    e.score = score_left + score_right;
    return e;
}

// ---------------------------------------------------------------------------------------------------
// Greedy extension (gap costs 0 / 0) with every distance row kept, then walked back (BLAST_GreedyAlign with an
// edit block; unpacked subject: ambiguous query letters never match, CORE/greedy_align.c:318-381)
// ---------------------------------------------------------------------------------------------------
inline int32_t run_of_matches(const uint8_t *a, const uint8_t *b, int32_t la, int32_t lb, int32_t i, int32_t j, bool backwards) {
    const int32_t i0 = i;
    if (backwards) while (i < la && j < lb && a[la - 1 - i] < 4 && a[la - 1 - i] == b[lb - 1 - j]) { i++; j++; }
    else while (i < la && j < lb && a[i] < 4 && a[i] == b[j]) { i++; j++; }
    return i - i0;
}
// ... with the direction known at compile time, eight bases per step where both sequences have them (forwards: the first byte that
// differs or is no base, from the low end of the words; backwards: from the high end of the words that END at the positions)
template <bool kBackwards>
inline int32_t run_of_matches_dir(const uint8_t *a, const uint8_t *b, int32_t la, int32_t lb, int32_t i, int32_t j) {
    const int32_t i0 = i;
    while (i + 8 <= la && j + 8 <= lb) {
        uint64_t x, y;
        if (kBackwards) { std::memcpy(&x, a + la - 8 - i, 8); std::memcpy(&y, b + lb - 8 - j, 8); }
        else { std::memcpy(&x, a + i, 8); std::memcpy(&y, b + j, 8); }
        const uint64_t bad = (x ^ y) | (x & 0xFCFCFCFCFCFCFCFCull);     // a byte is non-zero where the letters differ or a's is no base
        if (bad == 0) { i += 8; j += 8; continue; }
        const int32_t n = kBackwards ? (__builtin_clzll(bad) >> 3) : (__builtin_ctzll(bad) >> 3);
        return i + n - i0;
    }
    if (kBackwards) while (i < la && j < lb && a[la - 1 - i] < 4 && a[la - 1 - i] == b[lb - 1 - j]) { i++; j++; }
    else while (i < la && j < lb && a[i] < 4 && a[i] == b[j]) { i++; j++; }
    return i - i0;
}
int32_t greedy_half(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb, bool backwards, int32_t xdrop,
                    int32_t match2, int32_t mismatch2, int32_t &used_a, int32_t &used_b, Script &path)
{
    const int32_t kNone = -2;
    const int32_t dmax = std::min(10000, lb / 2 + 1), origin = dmax + 2;
    const int32_t lookback = (xdrop + match2 / 2) / (match2 + mismatch2) + 1;
    int32_t run = run_of_matches(a, b, la, lb, 0, 0, backwards);
    used_a = used_b = run;
    if (run == la || run == lb) { append_op(path, kSub, run); return 0; }
    // furthest subject offset per (distance, diagonal); a row spans the diagonals its distance can reach plus two
    // either side (the reference's allocation for the rows it adds, :677-683; its first two rows span everything)
    // (the buffers stay with the thread: `top` is 10,000 entries for a subject of 20 kb and more -- 40 KB allocated and cleared per half
    // and alignment, a quarter of a traceback worker's time at 2,469 alignments per batch -- and only its first `lookback` entries are
    // read before they are written)
    struct RowRef { size_t at; int32_t lo; };
    thread_local std::vector<int32_t> tl_store, tl_top; thread_local std::vector<RowRef> tl_rows;
    std::vector<int32_t> &store = tl_store, &top = tl_top; std::vector<RowRef> &rows = tl_rows;    // (one look-up of the thread's copies, not one per use)
    store.clear(); rows.clear();
    auto new_row = [&](int32_t lo, int32_t n) { rows.push_back(RowRef{store.size(), lo}); store.resize(store.size() + (size_t)n, 0); };
    auto at = [&](int32_t d, int32_t k) -> int32_t & { return store[rows[(size_t)d].at + (size_t)(k - rows[(size_t)d].lo)]; };
    new_row(origin - 2, 7); new_row(origin - 3, 9);
    if (top.size() < (size_t)(dmax + 2 + lookback)) top.resize((size_t)(dmax + 2 + lookback));
    std::fill(top.begin(), top.begin() + lookback + 1, 0);          // best score per distance, `lookback` zeros in front
    auto best_at = [&](int32_t d) -> int32_t & { return top[(size_t)(d + lookback)]; };
    at(0, origin) = run;
    best_at(0) = run * match2;
    int32_t lower = origin - 1, upper = origin + 1, best_d = 0, best_k = 0;
    bool hit_end_a = false, hit_end_b = false;
    for (int32_t d = 1; d <= dmax; d++) {
        const int32_t from = lower, to = upper;
        // (rows d - 1 and d through pointers of their own: both exist, and nothing is appended to `store` inside the loop over the
        // diagonals; round 6)
        int32_t *const prev = store.data() + rows[(size_t)d - 1].at - rows[(size_t)d - 1].lo;
        int32_t *const cur = store.data() + rows[(size_t)d].at - rows[(size_t)d].lo;
        prev[lower - 1] = kNone; prev[lower] = kNone; prev[upper] = kNone; prev[upper + 1] = kNone;
        int32_t floor_sum = best_at(d - lookback) + (match2 + mismatch2) * d - xdrop;
        floor_sum = (int32_t)std::ceil((double)floor_sum / (match2 / 2));
        int32_t far = 0, far_j = 0, far_k = 0;
        for (int32_t k = from; k <= to; k++) {
            int32_t j = std::max(prev[k + 1], prev[k]) + 1;
            j = std::max(j, prev[k - 1]);
            int32_t i = j + k - origin;
            if (j < 0 || i + j < floor_sum) { if (k == lower) lower++; else cur[k] = kNone; continue; }
            upper = k;
            const int32_t more = backwards ? run_of_matches_dir<true>(a, b, la, lb, i, j) : run_of_matches_dir<false>(a, b, la, lb, i, j);
            i += more; j += more;
            cur[k] = j;
            if (i + j > far) { far = i + j; far_j = j; far_k = k; }
            if (j == lb) { lower = k + 1; hit_end_b = true; }
            if (i == la) { upper = k - 1; hit_end_a = true; }
        }
        const int32_t sc = far * (match2 / 2) - d * (match2 + mismatch2);
        if (sc > best_at(d - 1)) { best_at(d) = sc; best_d = d; best_k = far_k; used_b = far_j; used_a = far_j + far_k - origin; }
        else best_at(d) = best_at(d - 1);
        if (lower > upper) break;
        if (!hit_end_b) lower--;
        if (!hit_end_a) upper++;
        new_row(lower - 2, upper - lower + 7);
    }
    // back from the best (distance, diagonal): the neighbour with the largest offset at distance - 1
    int32_t j = used_b, k = best_k;
    for (int32_t d = best_d; d > 0; d--) {
        const int32_t below = at(d - 1, k - 1), same = at(d - 1, k), above = at(d - 1, k + 1);
        if (below > std::max(same, above)) { append_op(path, kSub, j - below); append_op(path, kIns, 1); j = below; k--; }
        else if (same > above) { append_op(path, kSub, j - same); j = same; }
        else { append_op(path, kSub, j - above - 1); append_op(path, kDel, 1); j = above; k++; }
    }
    append_op(path, kSub, at(0, origin));
    return best_d;
}
// s_ReduceGaps (CORE/blast_gapalign.c:2547-2617): an insertion and a deletion around a short run are traded for mismatches
void reduce_gaps(Script &sc, const uint8_t *q, const uint8_t *s)
{
    for (size_t i = 0; i < sc.size(); i++) {
        if (sc[i].op == kSub) { q += sc[i].n; s += sc[i].n; continue; }
        if (i > 1 && sc[i].op != sc[i - 2].op && sc[i - 2].n > 0) {
            int32_t d = sc[i].n + sc[i - 1].n + sc[i - 2].n;
            if (d == 3) {
                sc[i - 2].n = 0; sc[i - 1].n = 2; sc[i].n = 0;
                if (sc[i].op == kIns) ++q; else ++s;
            } else if (d < 12) {
                int32_t same_now = 0, same_then = 0;
                d = std::min(sc[i].n, sc[i - 2].n);
                q -= sc[i - 1].n; s -= sc[i - 1].n;
                const uint8_t *q1 = q, *s1 = s;
                if (sc[i].op == kIns) s -= d; else q -= d;
                for (int32_t j = 0; j < sc[i - 1].n; ++j, ++q1, ++s1, ++q, ++s) { if (*q1 == *s1) same_now++; if (*q == *s) same_then++; }
                for (int32_t j = 0; j < d; ++j, ++q, ++s) if (*q == *s) same_then++;
                if (same_then >= same_now - d) { sc[i - 2].n -= d; sc[i - 1].n += d; sc[i].n -= d; }
                else { q = q1; s = s1; }
            }
        }
        if (sc[i].op == kIns) q += sc[i].n; else s += sc[i].n;
    }
    Script out;
    for (size_t i = 0; i < sc.size(); i++) {
        if (sc[i].n > 0) out.push_back(sc[i]);
        else if (++i < sc.size() && !out.empty()) out.back().n += sc[i].n;
    }
    sc.swap(out);
}
// The affine form (BLAST_AffineGreedyAlign with an edit block, CORE/greedy_align.c:755-1236): per (distance, diagonal)
// the furthest subject offset of a path ending in a match run, in a gap in the subject ("ins") and in a gap in the
// query ("del"); distances are in units of the costs' common factor, a step back reaches max(mismatch, open + extend)
// distances.  Every row is kept; the script is read back from the best cell by asking, state by state, which
// predecessor reaches furthest (s_GetNextAffineTbackFromMatch / FromIndel, :153-262).  Returns the half's score.
int32_t greedy_affine_half(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb, bool backwards, int32_t xdrop,
                           int32_t match2, int32_t mismatch2, int32_t open_in, int32_t extend_in,
                           int32_t &used_a, int32_t &used_b, Script &path)
{
    const int32_t kNone = -2, kNoDiag = 100000000;
    const int32_t half = match2 / 2;
    int32_t sub = match2 + mismatch2, open = open_in, ext = extend_in + half;
    auto gcd = [](int32_t x, int32_t y) { y = std::abs(y); if (y > x) std::swap(x, y); while (y) { const int32_t t = x % y; x = y; y = t; } return x; };
    const int32_t unit = open == 0 ? gcd(sub, ext) : gcd(sub, gcd(open, ext));       // BLAST_Gdb3
    if (unit > 1) { sub /= unit; open /= unit; ext /= unit; }
    const int32_t open_ext = open + ext, reach = std::max(sub, open_ext);
    const int32_t dmax = std::min(10000, lb / 2 + 1), dlast = dmax * ext, origin = dmax + 2;
    const int32_t lookback = (xdrop + half) / unit + 1;
    const int32_t run = run_of_matches(a, b, la, lb, 0, 0, backwards);
    used_a = used_b = run;
    if (run == la || run == lb) { append_op(path, kSub, run); return run * match2; }
    struct Cell { int32_t ins, match, del; };
    std::vector<Cell> store; struct RowRef { size_t at; int32_t lo; };
    std::vector<RowRef> rows;
    auto new_row = [&](int32_t lo, int32_t n) { rows.push_back(RowRef{store.size(), lo}); store.resize(store.size() + (size_t)std::max(n, 1)); };
    auto at = [&](int32_t d, int32_t k) -> Cell & { return store[rows[(size_t)d].at + (size_t)(k - rows[(size_t)d].lo)]; };
    // bounds of the diagonals alive at a distance; `reach` empty ones in front stand for negative distances
    std::vector<int32_t> lo_of((size_t)(dlast + 2 + reach), kNoDiag), hi_of((size_t)(dlast + 2 + reach), -kNoDiag);
    auto lo = [&](int32_t d) -> int32_t & { return lo_of[(size_t)(d + reach)]; };
    auto hi = [&](int32_t d) -> int32_t & { return hi_of[(size_t)(d + reach)]; };
    auto alive = [&](int32_t d, int32_t k) { return k >= lo(d) && k <= hi(d); };
    std::vector<int32_t> top((size_t)(dlast + 2 + lookback), 0);
    auto best_at = [&](int32_t d) -> int32_t & { return top[(size_t)(d + lookback)]; };
    new_row(origin, 1);
    at(0, origin) = Cell{kNone, run, kNone};
    best_at(0) = run * match2; lo(0) = hi(0) = origin;
    int32_t from = origin - 1, to = origin + 1, end_a = 0, end_b = 0, live = 1, best_d = 0, best_k = 0;
    for (int32_t d = 1; d <= dlast;) {
        const int32_t first = from, last = to;
        new_row(first, last - first + 1);
        int32_t floor_sum = best_at(d - lookback) + unit * d - xdrop;
        floor_sum = std::max(0, (int32_t)std::ceil((double)floor_sum / half));
        int32_t far = 0, far_j = 0, far_k = 0;
        for (int32_t k = first; k <= last; k++) {
            Cell &c = at(d, k);
            int32_t j = alive(d - open_ext, k + 1) ? at(d - open_ext, k + 1).match : kNone;
            if (alive(d - ext, k + 1)) j = std::max(j, at(d - ext, k + 1).del);
            c.del = j == kNone ? kNone : j + 1;
            j = alive(d - open_ext, k - 1) ? at(d - open_ext, k - 1).match : kNone;
            if (alive(d - ext, k - 1)) j = std::max(j, at(d - ext, k - 1).ins);
            c.ins = j;
            j = std::max(c.ins, c.del);
            if (alive(d - sub, k)) j = std::max(j, at(d - sub, k).match + 1);
            int32_t i = j + k - origin;
            if (j < 0 || i + j < floor_sum) { if (k == from) from++; else c.match = kNone; continue; }
            to = k;
            const int32_t more = run_of_matches(a, b, la, lb, i, j, backwards);
            i += more; j += more;
            c.match = j;
            if (i + j > far) { far = i + j; far_j = j; far_k = k; }
            if (i == la) { to = k; end_a = k - 1; }
            if (j == lb) { from = k; end_b = k + 1; }
        }
        const int32_t sc = far * half - d * unit;
        if (sc > best_at(d - 1)) { best_at(d) = sc; best_d = d; best_k = far_k; used_b = far_j; used_a = far_j + far_k - origin; }
        else best_at(d) = best_at(d - 1);
        if (from <= to) { live++; lo(d) = from; hi(d) = to; }
        if (lo(d - reach) <= hi(d - reach)) live--;
        if (live == 0) break;
        d++;
        from = std::min({lo(d - open_ext) - 1, lo(d - ext) - 1, lo(d - sub)});
        if (end_b > 0) from = std::max(from, end_b);
        to = std::max({hi(d - open_ext) + 1, hi(d - ext) + 1, hi(d - sub)});
        if (end_a > 0) to = std::min(to, end_a);
    }
    // back from the best cell
    int32_t j = used_b, k = best_k, d = best_d;
    enum { InMatch, InIns, InDel } state = InMatch;
    while (d > 0) {
        if (state == InMatch) {
            const Cell &c = at(d, k);
            int32_t prev;
            if (alive(d - sub, k) && at(d - sub, k).match >= std::max(c.ins, c.del)) { prev = at(d - sub, k).match; d -= sub; }
            else if (c.ins > c.del) { prev = c.ins; state = InIns; }
            else { prev = c.del; state = InDel; }
            append_op(path, kSub, j - prev);
            j = prev;
        } else {
            const bool ins = state == InIns;
            const int32_t nk = ins ? k - 1 : k + 1;
            append_op(path, ins ? kIns : kDel, 1);
            int32_t through_gap = kNone;
            if (alive(d - ext, nk)) through_gap = ins ? at(d - ext, nk).ins : at(d - ext, nk).del;
            if (alive(d - open_ext, nk) && through_gap < at(d - open_ext, nk).match) { d -= open_ext; state = InMatch; }
            else d -= ext;
            if (ins) k--; else { k++; j--; }
        }
    }
    append_op(path, kSub, at(0, origin).match);
    return best_at(best_d);
}

Extent greedy_traceback(const uint8_t *q, const uint8_t *s, int32_t qlen, int32_t slen, int32_t q0, int32_t s0, int32_t X,
                        int32_t reward, int32_t penalty, int32_t gap_open, int32_t gap_extend, Script &script)
{
    int32_t m2 = reward, mm2 = -penalty, x2 = X, go2 = gap_open, ge2 = gap_extend;
    if (m2 % 2 == 1) { m2 *= 2; mm2 *= 2; x2 *= 2; go2 *= 2; ge2 *= 2; }
    Script left, right; int32_t qr, sr, ql, sl;
    Extent e;
    if (go2 == 0 && ge2 == 0) {
        int32_t dist = greedy_half(q + q0, qlen - q0, s + s0, slen - s0, false, x2, m2, mm2, qr, sr, right);
        dist += greedy_half(q, q0, s, s0, true, x2, m2, mm2, ql, sl, left);
        e.score = (qr + sr + ql + sl) * reward / 2 - dist * (reward - penalty);
    } else {
        int32_t sc = greedy_affine_half(q + q0, qlen - q0, s + s0, slen - s0, false, x2, m2, mm2, go2, ge2, qr, sr, right);
        sc += greedy_affine_half(q, q0, s, s0, true, x2, m2, mm2, go2, ge2, ql, sl, left);
        e.score = (reward % 2 == 1) ? sc / 2 : sc;
    }
    e.q_start = q0 - ql; e.s_start = s0 - sl; e.q_stop = q0 + qr; e.s_stop = s0 + sr;
    script = join_halves(left, right);
    if (!script.empty()) reduce_gaps(script, q + e.q_start, s + e.s_start);
    return e;
}

// ---------------------------------------------------------------------------------------------------
// start point of an extension
// ---------------------------------------------------------------------------------------------------
const int32_t kWindow = 11;     // HSP_MAX_WINDOW
bool start_scores_positive(const int32_t (*mat)[16], const GbnHSP &h, const uint8_t *q, const uint8_t *s) {   // BLAST_CheckStartForGappedAlignment
    int32_t lo = std::max({-kWindow / 2, h.q_offset - h.q_gapped_start, h.s_offset - h.s_gapped_start});
    int32_t hi = std::min({kWindow / 2 + 1, h.q_end - h.q_gapped_start, h.s_end - h.s_gapped_start});
    int32_t sum = 0;
    for (int32_t i = lo; i < hi; i++) sum += mat[q[h.q_gapped_start + i]][s[h.s_gapped_start + i]];
    return sum > 0;
}
bool best_window_start(const int32_t (*mat)[16], const uint8_t *q, const uint8_t *s, const GbnHSP &h, int32_t &qo, int32_t &so) {   // BlastGetOffsetsForGappedAlignment
    const int32_t ql = h.q_end - h.q_offset, sl = h.s_end - h.s_offset;
    if (ql <= kWindow) { qo = h.q_offset + ql / 2; so = h.s_offset + ql / 2; return true; }
    int32_t sum = 0;
    for (int32_t i = 0; i < kWindow; i++) sum += mat[q[h.q_offset + i]][s[h.s_offset + i]];
    int32_t top = sum, where = h.q_offset + kWindow - 1;
    const int32_t n = std::min(ql, sl);
    for (int32_t i = kWindow; i < n; i++) {
        sum += mat[q[h.q_offset + i]][s[h.s_offset + i]] - mat[q[h.q_offset + i - kWindow]][s[h.s_offset + i - kWindow]];
        if (sum > top) { top = sum; where = h.q_offset + i; }
    }
    if (top > 0) { qo = where; so = where - h.q_offset + h.s_offset; return true; }
    sum = 0;
    for (int32_t i = 0; i < kWindow; i++) sum += mat[q[h.q_end - kWindow + i]][s[h.s_end - kWindow + i]];
    if (sum > 0) { qo = h.q_end - kWindow / 2; so = h.s_end - kWindow / 2; return true; }
    return false;
}
void longest_identity_run_start(const uint8_t *q, const uint8_t *s, GbnHSP &h) {     // BlastGetStartForGappedAlignmentNucl
    const int32_t kRun = 20;
    const int32_t back = std::min(h.s_gapped_start - h.s_offset, h.q_gapped_start - h.q_offset);
    const int32_t q0 = h.q_gapped_start - back, s0 = h.s_gapped_start - back;
    const int32_t n = std::min(h.s_end - s0, h.q_end - q0);
    int32_t run = 0, top = 0, where = q0; bool same = false, before = false; int32_t i = q0;
    for (; i < q0 + n; i++) {
        same = q[i] == s[i - q0 + s0];
        if (same != before) {
            before = same;
            if (same) run = 1; else if (run > top) { top = run; where = i - run / 2; }
        } else if (same) {
            if (++run > kRun) { h.q_gapped_start = i - kRun / 2; h.s_gapped_start = h.q_gapped_start + s0 - q0; return; }
        }
    }
    if (same && run > top) { top = run; where = i - run / 2; }
    if (top > 0) { h.q_gapped_start = where; h.s_gapped_start = where + s0 - q0; }
}

// ---------------------------------------------------------------------------------------------------
// one HSP with its script
// ---------------------------------------------------------------------------------------------------
struct Item { GbnHSP h; Script sc; int32_t ident = 0, alen = 0; bool live = true; };

void count_identities(const uint8_t *q, const uint8_t *s, Item &it) {
    const uint8_t *a = q + it.h.q_offset, *b = s + it.h.s_offset; int32_t same = 0, len = 0;
    for (const EditOp &o : it.sc) {
        len += o.n;
        if (o.op == kSub) { for (int32_t i = 0; i < o.n; i++) same += a[i] == b[i]; a += o.n; b += o.n; }
        else if (o.op == kDel) b += o.n; else a += o.n;
    }
    it.ident = same; it.alen = len;
}
// s_CutOffGapEditScript: drop the part before (q_cut, s_cut) or after it
void cut_script(Item &it, int32_t q_cut, int32_t s_cut, bool drop_front) {
    q_cut -= it.h.q_offset; s_cut -= it.h.s_offset;
    int32_t qn = 0, sn = 0, in_op = 0; size_t at = 0; bool found = false;
    for (; at < it.sc.size() && !found; at++) {
        const EditOp &o = it.sc[at];
        for (in_op = 0; in_op < o.n;) {
            if (o.op == kSub) { qn++; sn++; in_op++; }
            else if (o.op == kDel) { sn += o.n; in_op += o.n; }
            else { qn += o.n; in_op += o.n; }
            if (qn >= q_cut && sn >= s_cut) { found = true; break; }
        }
        if (found) break;
    }
    if (!found) return;
    if (drop_front) {
        Script rest;
        if (in_op < it.sc[at].n) rest.push_back(EditOp{it.sc[at].op, it.sc[at].n - in_op});
        rest.insert(rest.end(), it.sc.begin() + (long)at + 1, it.sc.end());
        it.sc.swap(rest);
        it.h.q_offset += qn; it.h.s_offset += sn;
    } else {
        if (in_op < it.sc[at].n) it.sc[at].n = in_op;
        it.sc.resize(at + 1);
        it.h.q_end = it.h.q_offset + qn; it.h.s_end = it.h.s_offset + sn;
    }
}
// Blast_HSPReevaluateWithAmbiguitiesGapped: best-scoring stretch of the script, grown over exact matches; true: drop
bool rescore_along_script(Item &it, const int32_t (*mat)[16], const uint8_t *q, int32_t qlen, const uint8_t *s, int32_t slen,
                          int32_t cutoff, int32_t reward, int32_t penalty, int32_t gap_open_in, int32_t gap_extend_in)
{
    if (it.sc.empty()) return true;
    int32_t factor = 1, gap_open = gap_open_in, gap_extend = gap_extend_in;
    if (gap_open_in == 0 && gap_extend_in == 0) { if (reward % 2 == 1) factor = 2; gap_open = 0; gap_extend = (reward - 2 * penalty) * factor / 2; }
    int32_t qi = it.h.q_offset, si = it.h.s_offset, sum = 0, score = 0;
    int32_t bq0 = qi, bq1 = qi, bs0 = si, bs1 = si, cq = qi, cs = si;
    int best_first = 0, best_last = 0, cur_first = 0, best_last_n = -1;
    Script &sc = it.sc;
    for (int index = 0; index < (int)sc.size(); index++) {
        for (int done = 0; done < sc[(size_t)index].n;) {
            const EditOp o = sc[(size_t)index];
            if (o.op == kSub) { sum += factor * mat[q[qi] & 0x0f][s[si]]; qi++; si++; done++; }
            else if (o.op == kDel) { sum -= gap_open + gap_extend * o.n; si += o.n; done += o.n; }
            else { sum -= gap_open + gap_extend * o.n; qi += o.n; done += o.n; }
            if (sum < 0) {
                if (done < sc[(size_t)index].n) { sc[(size_t)index].n -= done; cur_first = index; done = 0; }
                else cur_first = index + 1;
                sum = 0; cq = qi; cs = si;
                if (score < cutoff) { bq0 = qi; bs0 = si; score = 0; best_first = cur_first; best_last = cur_first; }
            } else if (sum > score) {
                score = sum; bq0 = cq; bs0 = cs; bq1 = qi; bs1 = si;
                best_first = cur_first; best_last = index; best_last_n = done;
            }
        }
    }
    score /= factor;
    if (best_first < (int)sc.size() && best_last < (int)sc.size()) {
        int32_t a = bq0, b = bs0, ext = 0;
        while (a > 0 && b > 0 && q[a - 1] == s[b - 1] && q[a - 1] < 4) { a--; b--; ext++; }
        bq0 -= ext; bs0 -= ext; sc[(size_t)best_first].n += ext;
        if (best_last == best_first) best_last_n += ext;
        score += ext * reward;
        a = bq1; b = bs1; ext = 0;
        while (a < qlen && b < slen && q[a] < 4 && q[a] == s[b]) { a++; b++; ext++; }
        bq1 += ext; bs1 += ext; sc[(size_t)best_last].n += ext; best_last_n += ext;
        score += ext * reward;
    }
    it.h.score = score;
    if (score < cutoff) return true;
    it.h.q_offset = bq0; it.h.q_end = bq1; it.h.s_offset = bs0; it.h.s_end = bs1;
    if (best_last != (int)sc.size() - 1 || best_first > 0) sc = Script(sc.begin() + best_first, sc.begin() + best_last + 1);
    sc.back().n = best_last_n;
    return false;
}

bool before_by_start(const Item &x, const Item &y) {     // s_QueryOffsetCompareHSPs
    const GbnHSP &a = x.h, &b = y.h;
    if (a.context != b.context) return a.context < b.context;
    if (a.q_offset != b.q_offset) return a.q_offset < b.q_offset;
    if (a.s_offset != b.s_offset) return a.s_offset < b.s_offset;
    if (a.score != b.score) return a.score > b.score;
    if (a.q_end != b.q_end) return a.q_end > b.q_end;
    return a.s_end > b.s_end;
}
bool before_by_end(const Item &x, const Item &y) {       // s_QueryEndCompareHSPs
    const GbnHSP &a = x.h, &b = y.h;
    if (a.context != b.context) return a.context < b.context;
    if (a.q_end != b.q_end) return a.q_end < b.q_end;
    if (a.s_end != b.s_end) return a.s_end < b.s_end;
    if (a.score != b.score) return a.score > b.score;
    if (a.q_offset != b.q_offset) return a.q_offset > b.q_offset;
    return a.s_offset > b.s_offset;
}
bool before_by_score(const Item &x, const Item &y) {     // ScoreCompareHSPs
    const GbnHSP &a = x.h, &b = y.h;
    if (a.score != b.score) return a.score > b.score;
    if (a.s_offset != b.s_offset) return a.s_offset < b.s_offset;
    if (a.s_end != b.s_end) return a.s_end > b.s_end;
    if (a.q_offset != b.q_offset) return a.q_offset < b.q_offset;
    return a.q_end > b.q_end;
}
// HSPs sharing a start or an end with a better one: the longer one keeps its other part, the rest go
// (Blast_HSPListPurgeHSPsWithCommonEndpoints with purge = FALSE).  Returns how many HSPs were left untouched
// (they come first); the trimmed ones follow in `v`, the dropped ones are gone.
size_t trim_shared_ends(std::vector<Item> &v)
{
    std::vector<Item> set_aside;
    auto pass = [&](bool by_start) {
        std::stable_sort(v.begin(), v.end(), by_start ? before_by_start : before_by_end);
        std::vector<Item> kept;
        for (size_t i = 0; i < v.size();) {
            kept.push_back(std::move(v[i]));
            const Item &lead = kept.back();
            size_t j = i + 1;
            for (; j < v.size(); j++) {
                const GbnHSP &a = lead.h, &b = v[j].h;
                const bool shared = a.context == b.context && (by_start ? (a.q_offset == b.q_offset && a.s_offset == b.s_offset)
                                                                        : (a.q_end == b.q_end && a.s_end == b.s_end));
                if (!shared) break;
                if (by_start ? b.q_end > a.q_end : b.q_offset < a.q_offset) {
                    Item t = std::move(v[j]);
                    if (by_start) cut_script(t, a.q_end, a.s_end, true); else cut_script(t, a.q_offset, a.s_offset, false);
                    set_aside.push_back(std::move(t));
                }
            }
            i = j;
        }
        v.swap(kept);
    };
    pass(true); pass(false);
    const size_t untouched = v.size();
    for (size_t i = set_aside.size(); i-- > 0;) v.push_back(std::move(set_aside[i]));     // (the reference parks them from the back of its array)
    return untouched;
}

// Blast_TracebackFromHSPList + s_HSPListPostTracebackUpdate for the HSPs of one query against one subject
int traceback_list(const GbnBatch &b, const uint8_t *subject, int32_t slen, const GbnHSP *in, size_t nin, std::vector<Item> &out)
{
    const GbnOptions &o = b.opt;
    const bool greedy = o.greedy != 0;
    const int32_t X = b.gap_x_dropoff_final;
    std::vector<Item> items(nin);
    for (size_t i = 0; i < nin; i++) items[i].h = in[i];
    std::vector<GbnHSP> accepted; accepted.reserve(nin);
    {
        EnvelopeIndex index(&b, &accepted, b.qlen + 1, slen + 1);
        for (Item &it : items) {
            GbnHSP &h = it.h;
            const GbnContext &cx = b.ctx[(size_t)h.context];
            const uint8_t *q = b.query() + cx.query_offset;
            if (index.enveloped(h, o.min_diag_separation)) { it.live = false; continue; }
            int32_t q0, s0;
            { gbn::CpuScope c1(gbn::GBN_CPU_TB_START);
            if ((h.q_gapped_start == 0 && h.s_gapped_start == 0) || !start_scores_positive(b.matrix, h, q, subject)) {
                if (!best_window_start(b.matrix, q, subject, h, q0, s0)) { it.live = false; continue; }
                h.q_gapped_start = q0; h.s_gapped_start = s0;
            } else {
                longest_identity_run_start(q, subject, h);
                q0 = h.q_gapped_start; s0 = h.s_gapped_start;
            }
            }
            // long subjects: only the stretch an extension can reach (AdjustSubjectRange)
            int32_t shift = 0, sub_len = slen;
            if (slen >= 90000) {
                const int32_t reach_left = q0 + 3000, reach_right = cx.query_length - q0 + 3000, s_at = s0;
                if (s_at > reach_left) { shift = s_at - reach_left; s0 = reach_left; }
                sub_len = std::min(slen, s_at + reach_right) - shift;
            }
            const uint8_t *sub = subject + shift;
            h.s_gapped_start = s0;
            gbn::CpuScope c2(gbn::GBN_CPU_TB_ALIGN);
            Extent e = greedy ? greedy_traceback(q, sub, cx.query_length, sub_len, q0, s0, X, o.reward, o.penalty, o.gap_open, o.gap_extend, it.sc)
                              : gapped_traceback(b.matrix, q, sub, cx.query_length, sub_len, q0, s0, X, o.gap_open, o.gap_extend, it.sc);
            h.score = e.score; h.q_offset = e.q_start; h.q_end = e.q_stop; h.s_offset = e.s_start; h.s_end = e.s_stop;
            if (!greedy) count_identities(q, sub, it);
            if (shift > 0) { h.s_offset += shift; h.s_end += shift; h.s_gapped_start += shift; }
            accepted.push_back(h);
            index.insert((int32_t)accepted.size() - 1);
        }
    }
    items.erase(std::remove_if(items.begin(), items.end(), [](const Item &t) { return !t.live; }), items.end());
    gbn::CpuScope c3(gbn::GBN_CPU_TB_RESCORE);
    size_t first_to_rescore = trim_shared_ends(items);
    if (greedy) first_to_rescore = 0;           // the greedy aligner ignored ambiguities: every HSP is re-scored
    for (size_t i = first_to_rescore; i < items.size(); i++) {
        Item &it = items[i];
        const GbnContext &cx = b.ctx[(size_t)it.h.context];
        const uint8_t *q = b.query() + cx.query_offset;
        if (rescore_along_script(it, b.matrix, q, cx.query_length, subject, slen, cx.gap_cutoff_score, o.reward, o.penalty, o.gap_open, o.gap_extend))
            it.live = false;
        else count_identities(q, subject, it);
    }
    items.erase(std::remove_if(items.begin(), items.end(), [](const Item &t) { return !t.live; }), items.end());
    if (!std::is_sorted(items.begin(), items.end(), before_by_score)) std::stable_sort(items.begin(), items.end(), before_by_score);
    {   // what a better HSP envelops goes
        accepted.clear();
        EnvelopeIndex index(&b, &accepted, b.qlen + 1, slen + 1);
        for (Item &it : items) {
            if (index.enveloped(it.h, o.min_diag_separation)) it.live = false;
            else { accepted.push_back(it.h); index.insert((int32_t)accepted.size() - 1); }
        }
        items.erase(std::remove_if(items.begin(), items.end(), [](const Item &t) { return !t.live; }), items.end());
    }
    for (Item &it : items) {
        if (b.round_down) it.h.score &= ~1;
        it.h.evalue = evalue_for_score(it.h.score, b.kbp_gap, b.ctx[(size_t)it.h.context].eff_searchsp);
        if (it.h.evalue <= o.evalue) out.push_back(std::move(it));
    }
    return GBN_OK;
}

inline int fuzzy_order(double a, double b) { return a < (1 - 1e-6) * b ? -1 : (a > (1 + 1e-6) * b ? 1 : 0); }

}  // namespace
}  // namespace gbn

using namespace gbn;

struct GbnTraceback {
    std::vector<GbnTbHSP> hsps;         // per query: subjects by best e-value, HSPs by score inside
    std::vector<uint8_t> op; std::vector<int32_t> op_len;
    std::vector<int64_t> query_start;   // [num_queries + 1] into hsps
};

extern "C" {

int gbn_traceback_new(GbnTraceback **out) { return gbn::guard(__func__, [&]() -> int { if (!out) return GBN_ERR_ARG; *out = new GbnTraceback(); return GBN_OK; }); }
void gbn_traceback_free(GbnTraceback *t) { delete t; }
int64_t gbn_traceback_num_hsps(const GbnTraceback *t) { return t ? (int64_t)t->hsps.size() : 0; }
const GbnTbHSP *gbn_traceback_hsps(const GbnTraceback *t) { return t->hsps.data(); }
const uint8_t *gbn_traceback_ops(const GbnTraceback *t) { return t->op.data(); }
const int32_t *gbn_traceback_op_lengths(const GbnTraceback *t) { return t->op_len.data(); }
const int64_t *gbn_traceback_query_starts(const GbnTraceback *t) { return t->query_start.data(); }

// lists: the collector's output -- HSPs of one (subject, query) pair contiguous, sorted by score, pairs in ascending
// (oid, query) order (list_start[nlists + 1]).  Subjects are read back from the shard in HBM, unpacked and traced
// on `threads` host threads (0: one per four CPUs this process may use -- gbn_host_cpus --, at most 16).
int gbn_traceback_run(GbnBatch *batch, GbnDb *db, const GbnHSP *hsps, const int64_t *list_start, int64_t nlists,
                      int32_t threads, GbnTraceback *out)
{
    return gbn::guard(__func__, [&]() -> int {
    if (!batch || !db || !out || nlists < 0 || (nlists > 0 && (!hsps || !list_start))) { set_error("gbn_traceback_run: bad argument"); return GBN_ERR_ARG; }
    out->hsps.clear(); out->op.clear(); out->op_len.clear(); out->query_start.assign((size_t)batch->nq + 1, 0);
    if (nlists == 0) return GBN_OK;
    gbn::CpuScope cpu(gbn::GBN_CPU_TRACEBACK);
    trace_mark("traceback: starts");
    // subjects that have lists, each fetched once -- of a long subject only the stretch the extensions can reach
    // (AdjustSubjectRange: query length + 3000 either side of an HSP) is read back and unpacked
    struct Work { int32_t local; int64_t first_list, end_list; int32_t lo; std::vector<uint8_t> bases; int32_t hi = 0; size_t packed_at = 0; };
    std::vector<Work> work;
    for (int64_t l = 0; l < nlists;) {
        const int32_t oid = hsps[list_start[l]].oid;
        int64_t e = l;
        while (e < nlists && hsps[list_start[e]].oid == oid) e++;
        const int32_t local = db->local_of(oid);           // the sequence's index in the shard (not a chunk's: db->real_*)
        if (local < 0) { set_error("gbn_traceback_run: subject id outside this shard"); return GBN_ERR_ARG; }
        work.push_back(Work{local, l, e, 0, {}});
        l = e;
    }
    std::vector<uint8_t> packed;
    {
        std::vector<int64_t> src_off; std::vector<int32_t> nbytes; std::vector<int32_t> his, wbytes;
        const bool chunked = !db->real_of.empty();
        auto seq_len = [&](int32_t local) { return chunked ? db->real_len[(size_t)local] : db->len[(size_t)local]; };
        for (Work &w : work) {
            const int32_t len = seq_len(w.local);
            int32_t lo = 0, hi = len;
            if (len >= 90000) {
                lo = len; hi = 0;
                for (int64_t i = list_start[w.first_list]; i < list_start[w.end_list]; i++) {
                    const int32_t reach = batch->ctx[(size_t)hsps[i].context].query_length + 3000 + 64;
                    lo = std::min(lo, std::max(0, std::min(hsps[i].s_offset, hsps[i].s_gapped_start) - reach));
                    hi = std::max(hi, std::min(len, std::max(hsps[i].s_end, hsps[i].s_gapped_start) + reach));
                }
                lo &= ~3;
            }
            w.lo = lo; his.push_back(hi);
            if (!chunked) {
                src_off.push_back(db->byte_off[(size_t)w.local] + (lo >> 2));
                nbytes.push_back((hi - lo + 3) / 4); wbytes.push_back(nbytes.back());
            } else {
                // the sequence exists as overlapping chunk copies (a chunk starts on a byte of its sequence): the
                // stretch is put together from the chunks it crosses
                const int64_t stride = (int64_t)db->chunk_len - kDbseqChunkOverlap;
                int32_t total = 0;
                for (int64_t pos = lo; pos < hi;) {
                    int64_t c = std::min<int64_t>(pos / stride, (int64_t)(w.local + 1 < db->real_seqs ? db->first_virt[(size_t)w.local + 1] : db->num_seqs) - db->first_virt[(size_t)w.local] - 1);
                    const int32_t v = db->first_virt[(size_t)w.local] + (int32_t)c;
                    const int64_t coff = c * stride, cend = coff + db->len[(size_t)v];
                    const int64_t upto = std::min<int64_t>(hi, cend);
                    const int32_t nb = (int32_t)((upto - pos + 3) / 4);
                    src_off.push_back(db->byte_off[(size_t)v] + ((pos - coff) >> 2)); nbytes.push_back(nb); total += nb;
                    pos = upto;                                             // (a multiple of 4 unless it is `hi`)
                }
                wbytes.push_back(total);
            }
        }
        const int rc = gather_shard_bytes(*db, src_off, nbytes, packed);
        if (rc != GBN_OK) return rc;
        trace_mark("traceback: subject stretches read back");
        size_t at = 0;
        for (size_t k = 0; k < work.size(); k++) { work[k].hi = his[k]; work[k].packed_at = at; at += (size_t)wbytes[k]; }
    }
    // a subject's stretch one base per byte, with the codes the 2-bit data cannot hold: made by the thread that aligns the subject's
    // lists (round 6; one thread used to unpack all stretches of a batch -- 20 MB for 2,500 subjects -- in front of the parallel part:
    // the serial third of a batch's traceback at sixteen threads)
    auto unpack = [&](Work &w) {
        const int32_t n = w.hi - w.lo; const uint8_t *p = packed.data() + w.packed_at;
        w.bases.resize(((size_t)n + 3) / 4 * 4 + 4);
        uint8_t *o = w.bases.data();
        // (a packed byte's four bases as one 32-bit store out of a 1 KB table: the byte-wise loop was a fifth of a worker's time)
        static const std::array<uint32_t, 256> four = [] {
            std::array<uint32_t, 256> t{};
            for (int c = 0; c < 256; c++) { const uint8_t b[4] = {(uint8_t)(c >> 6), (uint8_t)((c >> 4) & 3), (uint8_t)((c >> 2) & 3), (uint8_t)(c & 3)}; std::memcpy(&t[(size_t)c], b, 4); }
            return t; }();
        for (int32_t i = 0; i < (n + 3) / 4; i++, o += 4) std::memcpy(o, &four[p[i]], 4);
        if (!db->amb.empty())
            for (const GbnDb::AmbRun &r : db->amb[(size_t)w.local]) {
                const int32_t a = std::max(r.start, w.lo), e = std::min(r.start + r.length, w.hi);
                for (int32_t x = a; x < e; x++) w.bases[(size_t)(x - w.lo)] = r.code;
            }
    };
    struct Done { int32_t oid, query; std::vector<Item> items; };
    std::vector<std::vector<Done>> per_work(work.size());
    std::atomic<size_t> next{0}; std::atomic<int> failed{GBN_OK}; std::string err;
    std::mutex err_mu;
    auto body = [&]() {
        gbn::CpuScope cpu_w(gbn::GBN_CPU_TRACEBACK_WORKERS);
        for (size_t k; (k = next.fetch_add(1)) < work.size();) {
            Work &w = work[k];
            { gbn::CpuScope cpu_u(gbn::GBN_CPU_TB_UNPACK); unpack(w); }
            for (int64_t l = w.first_list; l < w.end_list; l++) {
                const GbnHSP *first = hsps + list_start[l]; const size_t n = (size_t)(list_start[l + 1] - list_start[l]);
                Done d; d.oid = first->oid; d.query = first->context / 2;
                // (base 0 of the subject sits at w.bases - w.lo; nothing outside the stretch read back is touched)
                const int rc = traceback_list(*batch, w.bases.data() - w.lo, db->real_of.empty() ? db->len[(size_t)w.local] : db->real_len[(size_t)w.local], first, n, d.items);
                if (rc != GBN_OK) { std::lock_guard<std::mutex> lk(err_mu); failed = rc; err = gbn_last_error(); return; }
                for (Item &it : d.items) it.h.oid = d.oid;
                if (!d.items.empty()) per_work[k].push_back(std::move(d));
            }
        }
    };
    unsigned nthreads = threads > 0 ? (unsigned)threads : std::min(16u, std::max(1u, gbn::host_cpus() / 4));
    nthreads = (unsigned)std::min<size_t>(nthreads, work.size());
    if (nthreads <= 1) body();
    else { std::vector<std::thread> pool; for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(body); for (auto &th : pool) th.join(); }
    if (failed != GBN_OK) { set_error(err); return failed; }
    trace_mark("traceback: lists aligned");
    gbn::CpuScope c4(gbn::GBN_CPU_TB_SORT_OUT);
    // per query: subjects by (best e-value, best score, oid descending), at most hitlist_size of them
    std::vector<std::vector<Done *>> by_query((size_t)batch->nq);
    for (auto &v : per_work) for (Done &d : v) by_query[(size_t)d.query].push_back(&d);
    for (int32_t qi = 0; qi < batch->nq; qi++) {
        auto &lists = by_query[(size_t)qi];
        auto best_e = [](const Done *d) { double e = d->items[0].h.evalue; for (const Item &it : d->items) e = std::min(e, it.h.evalue); return e; };
        std::stable_sort(lists.begin(), lists.end(), [&](const Done *a, const Done *c) {
            if (int r = fuzzy_order(best_e(a), best_e(c))) return r < 0;
            if (a->items[0].h.score != c->items[0].h.score) return a->items[0].h.score > c->items[0].h.score;
            return a->oid > c->oid; });
        if ((int32_t)lists.size() > batch->opt.hitlist_size) lists.resize((size_t)batch->opt.hitlist_size);
        out->query_start[(size_t)qi] = (int64_t)out->hsps.size();
        for (const Done *d : lists) for (const Item &it : d->items) {
            GbnTbHSP r; std::memset(&r, 0, sizeof(r));
            r.hsp = it.h; r.num_ident = it.ident;
            r.ops_first = (int64_t)out->op.size(); r.ops_count = (int32_t)it.sc.size();
            int32_t len = it.h.q_end - it.h.q_offset;
            for (const EditOp &o : it.sc) {
                out->op.push_back(o.op); out->op_len.push_back(o.n);
                if (o.op == kDel) { len += o.n; r.gaps += o.n; r.gap_opens++; }
                else if (o.op == kIns) { r.gaps += o.n; r.gap_opens++; }
            }
            r.align_length = len;
            r.bit_score = (batch->kbp_gap.lambda * it.h.score - batch->kbp_gap.logK) / 0.69314718055994530941723212145818;
            out->hsps.push_back(r);
        }
    }
    out->query_start[(size_t)batch->nq] = (int64_t)out->hsps.size();
    trace_mark("traceback: done");
    return GBN_OK;
    });
}

// Final results of several shards of one database (one gbn_traceback_run per shard, same query batch) -> the
// results of the whole database: per query the subjects of all parts by (best e-value, best score, oid
// descending), at most hitlist_size (CORE/blast_hits.c:2757-2788, 3093-3131: the order and the cut the
// single-shard stage applies).  Edit scripts stay with the parts (ops_first / ops_count are cleared).
// out must hold the sum of the parts' HSPs; out_query_start nq + 1 offsets.  Returns the number written.
int64_t gbn_traceback_merge(int32_t nparts, const GbnTbHSP *const *hsps, const int64_t *const *query_start, int32_t nq,
                            int32_t hitlist_size, GbnTbHSP *out, int64_t *out_query_start)
{
    return gbn::guard_as<int64_t>(__func__, (int64_t)-1, (int64_t)-1, [&]() -> int64_t {
    if (nparts < 0 || nq < 0 || !out_query_start || (nparts > 0 && (!hsps || !query_start))) { set_error("gbn_traceback_merge: bad argument"); return -1; }
    struct List { const GbnTbHSP *first; int64_t n; double best_e; };
    int64_t w = 0;
    std::vector<List> lists;
    for (int32_t qi = 0; qi < nq; qi++) {
        lists.clear();
        for (int32_t p = 0; p < nparts; p++) {
            const int64_t a = query_start[p][qi], e = query_start[p][qi + 1];
            for (int64_t i = a; i < e;) {
                int64_t j = i; double be = hsps[p][i].hsp.evalue;
                while (j < e && hsps[p][j].hsp.oid == hsps[p][i].hsp.oid) { be = std::min(be, hsps[p][j].hsp.evalue); j++; }
                lists.push_back(List{hsps[p] + i, j - i, be});
                i = j;
            }
        }
        std::sort(lists.begin(), lists.end(), [](const List &a, const List &c) { return a.first->hsp.oid < c.first->hsp.oid; });
        std::stable_sort(lists.begin(), lists.end(), [](const List &a, const List &c) {
            if (int r = fuzzy_order(a.best_e, c.best_e)) return r < 0;
            if (a.first->hsp.score != c.first->hsp.score) return a.first->hsp.score > c.first->hsp.score;
            return a.first->hsp.oid > c.first->hsp.oid; });
        if (hitlist_size > 0 && (int64_t)lists.size() > hitlist_size) lists.resize((size_t)hitlist_size);
        out_query_start[qi] = w;
        for (const List &l : lists) for (int64_t i = 0; i < l.n; i++) { out[w] = l.first[i]; out[w].ops_first = 0; out[w].ops_count = 0; w++; }
    }
    out_query_start[nq] = w;
    return w;
    });
}

}  // extern "C"
