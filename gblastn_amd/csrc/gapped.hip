// gapped.hip -- the score-only gapped extensions of the preliminary search (gfx950 / CDNA4), one initial hit each:
//   greedy_kernel        megablast: greedy gapped extension, linear and affine
//                        (CORE/greedy_align.c:385-753, CORE/blast_gapalign.c:2619-2751)
//   dynprog_lane_kernel / dynprog_wave_kernel / dynprog_kernel: blastn: X-drop dynamic programming on the packed subject
//                        (s_BlastDynProgNtGappedAlignment / s_BlastAlignPackedNucl, CORE/blast_gapalign.c:2762-3056): an
//                        extension per lane with the band in LDS, per wave with the band in registers, per thread with it in
//                        scratch memory -- each kernel takes what the one before leaves
// (a translation unit of its own since round 4; kernels.hip keeps the scans and the seed stage).  Integer work only: no MFMA.
#include "scan_dev.hpp"

// ---------------------------------------------------------------------------
// gapped extensions, one thread per initial hit (score-only)
// ---------------------------------------------------------------------------
namespace {
// query as the gapped kernels see it: 2 bits per base plus a bitmap of the codes that can
// never match a subject base (ambiguity codes, sentinels); a0 = index of "q[0]"
struct GQ { const uint8_t *q2; const uint8_t *qinv; int64_t a0; };
__device__ __forceinline__ GQ operator+(GQ q, int32_t d) { q.a0 += d; return q; }

// s_FindFirstMismatch (CORE/greedy_align.c:318-381), 32 bases per step
__device__ __forceinline__ int32_t match_run_fwd(const GQ &q, const uint8_t *subj, int32_t len1, int32_t len2,
                                                 int32_t i1, int32_t i2, int32_t s_base)
{
    const int32_t maxn = min(len1 - i1, len2 - i2);
    const int64_t qa = q.a0 + i1, sa = (int64_t)s_base + i2;
    for (int32_t n = 0; n < maxn; n += 32) {
        const uint64_t x = bases32(q.q2, qa + n) ^ bases32(subj, sa + n);
        const uint32_t inv = bits32(q.qinv, qa + n);
        const int32_t m = min(x ? (__clzll((long long)x) >> 1) : 32, inv ? __clz((int)inv) : 32);
        if (m < 32) return min(n + m, maxn);
    }
    return max(maxn, 0);
}
__device__ __forceinline__ int32_t match_run_rev(const GQ &q, const uint8_t *subj, int32_t len1, int32_t len2,
                                                 int32_t i1, int32_t i2)
{
    const int32_t maxn = min(len1 - i1, len2 - i2);
    const int64_t qa = q.a0 + len1 - 1 - i1, sa = (int64_t)len2 - 1 - i2;     // last pair compared first
    for (int32_t n = 0; n < maxn; n += 32) {
        const uint64_t x = bases32(q.q2, qa - n - 31) ^ bases32(subj, sa - n - 31);
        const uint32_t inv = bits32(q.qinv, qa - n - 31);
        const int32_t m = min(x ? ((__ffsll((long long)x) - 1) >> 1) : 32, inv ? (__ffs((int)inv) - 1) : 32);
        if (m < 32) return min(n + m, maxn);
    }
    return max(maxn, 0);
}

struct GSeed { int32_t start_q, start_s, match_length; };

// non-affine greedy (BLAST_GreedyAlign), score only
__device__ int32_t greedy_linear(const GQ &q, int32_t len1, const uint8_t *subj, int32_t s_base, int32_t len2,
                                 bool reverse, int32_t xdrop, int32_t match_cost, int32_t mismatch_cost,
                                 int32_t *l1, int32_t *l2, GSeed &seed, int32_t *row0, int32_t *row1,
                                 int32_t *max_score_base)
{
    const int32_t kInvalid = -2;
    int32_t max_dist = min(10000, len2 / 2 + 1);
    int32_t diag_origin = max_dist + 2;
    int32_t xdrop_offset = (xdrop + match_cost / 2) / (match_cost + mismatch_cost) + 1;
    int32_t index = reverse ? match_run_rev(q, subj, len1, len2, 0, 0) : match_run_fwd(q, subj, len1, len2, 0, 0, s_base);
    *l1 = index; *l2 = index;
    int32_t seq1_index = index, seq2_index, best_dist = 0, best_diag = 0;
    seed.start_q = 0; seed.start_s = 0; seed.match_length = index;
    int32_t longest = index;
    if (index == len1 || index == len2) return 0;
    int32_t *max_score = max_score_base + xdrop_offset;
    for (int32_t t = 0; t < xdrop_offset; t++) max_score_base[t] = 0;
    row0[diag_origin] = seq1_index;
    max_score[0] = seq1_index * match_cost;
    int32_t diag_lower = diag_origin - 1, diag_upper = diag_origin + 1;
    bool end1 = false, end2 = false;
    for (int32_t d = 1; d <= max_dist; d++) {
        int32_t curr_extent = 0, curr_seq2 = 0, curr_diag = 0;
        const int32_t tl = diag_lower, tu = diag_upper;
        int32_t *prev = ((d - 1) & 1) ? row1 : row0, *cur = (d & 1) ? row1 : row0;
        prev[diag_lower - 1] = kInvalid; prev[diag_lower] = kInvalid;
        prev[diag_upper] = kInvalid; prev[diag_upper + 1] = kInvalid;
        int32_t xs = max_score[d - xdrop_offset] + (match_cost + mismatch_cost) * d - xdrop;
        xs = (int32_t)ceil((double)xs / (double)(match_cost / 2));
        for (int32_t k = tl; k <= tu; k++) {
            seq2_index = max(prev[k + 1], prev[k]) + 1;
            seq2_index = max(seq2_index, prev[k - 1]);
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xs) {
                if (k == diag_lower) diag_lower++; else cur[k] = kInvalid;
                continue;
            }
            diag_upper = k;
            index = reverse ? match_run_rev(q, subj, len1, len2, seq1_index, seq2_index)
                            : match_run_fwd(q, subj, len1, len2, seq1_index, seq2_index, s_base);
            if (index > longest) { seed.start_q = seq1_index; seed.start_s = seq2_index; seed.match_length = longest = index; }
            seq1_index += index; seq2_index += index;
            cur[k] = seq2_index;
            if (seq1_index + seq2_index > curr_extent) { curr_extent = seq1_index + seq2_index; curr_seq2 = seq2_index; curr_diag = k; }
            if (seq2_index == len2) { diag_lower = k + 1; end2 = true; }
            if (seq1_index == len1) { diag_upper = k - 1; end1 = true; }
        }
        int32_t curr_score = curr_extent * (match_cost / 2) - d * (match_cost + mismatch_cost);
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score; best_dist = d; best_diag = curr_diag;
            *l2 = curr_seq2; *l1 = curr_seq2 + best_diag - diag_origin;
        } else max_score[d] = max_score[d - 1];
        if (diag_lower > diag_upper) break;
        if (!end2) diag_lower--;
        if (!end1) diag_upper++;
    }
    return best_dist;
}
}  // namespace

namespace {
struct GOff { int32_t insert_off, match_off, delete_off; };

__device__ int gcd_dev(int a, int b) { b = abs(b); if (b > a) { int c = a; a = b; b = c; } while (b) { int c = a % b; a = b; b = c; } return a; }

// affine greedy (BLAST_AffineGreedyAlign, CORE/greedy_align.c:755-1236), score only.
// scratch: rows[(max_penalty+1) * row_len] GOff, bounds[2 * (scaled_max + 1 + max_penalty)], max_score
__device__ int32_t greedy_affine(const GQ &q, int32_t len1, const uint8_t *subj, int32_t s_base, int32_t len2,
                                 bool reverse, int32_t xdrop, int32_t match_score, int32_t mismatch_score,
                                 int32_t in_gap_open, int32_t in_gap_extend, int32_t *l1, int32_t *l2, GSeed &seed,
                                 int32_t *scratch, int32_t row_len_alloc)
{
    const int32_t kInvalid = -2, kInvalidDiag = 100000000;
    const int32_t half = match_score / 2;
    int32_t op_cost = match_score + mismatch_score, gap_open = in_gap_open, gap_extend = in_gap_extend + half;
    int32_t g = (gap_open == 0) ? gcd_dev(op_cost, gap_extend) : gcd_dev(op_cost, gcd_dev(gap_open, gap_extend));
    if (g > 1) { op_cost /= g; gap_open /= g; gap_extend /= g; }
    const int32_t scf = g, goe = gap_open + gap_extend, max_penalty = max(op_cost, goe);
    const int32_t max_dist = min(10000, len2 / 2 + 1), scaled_max = max_dist * gap_extend;
    const int32_t diag_origin = max_dist + 2;
    const int32_t xoff = (xdrop + half) / scf + 1;
    int32_t index = reverse ? match_run_rev(q, subj, len1, len2, 0, 0) : match_run_fwd(q, subj, len1, len2, 0, 0, s_base);
    *l1 = index; *l2 = index;
    int32_t seq1_index = index, seq2_index, best_dist = 0, best_diag = 0, longest = index;
    seed.start_q = 0; seed.start_s = 0; seed.match_length = index;
    if (index == len1 || index == len2) return index * match_score;
    const int32_t nrows = max_penalty + 1;
    GOff *rows = reinterpret_cast<GOff *>(scratch);
    int32_t *bounds = scratch + (size_t)nrows * row_len_alloc * 3;
    int32_t *diag_lower = bounds, *diag_upper = bounds + scaled_max + 1 + max_penalty;
    int32_t *msb = bounds + 2 * (size_t)(scaled_max + 1 + max_penalty);
    int32_t *max_score = msb + xoff;
    for (int32_t t = 0; t < xoff; t++) msb[t] = 0;
    for (int32_t t = 0; t < max_penalty; t++) { diag_lower[t] = kInvalidDiag; diag_upper[t] = -kInvalidDiag; }
    diag_lower += max_penalty; diag_upper += max_penalty;
#define GROW(dd) (rows + (size_t)((dd) % nrows) * row_len_alloc)
    GROW(0)[diag_origin].match_off = seq1_index; GROW(0)[diag_origin].insert_off = kInvalid; GROW(0)[diag_origin].delete_off = kInvalid;
    max_score[0] = seq1_index * match_score;
    diag_lower[0] = diag_origin; diag_upper[0] = diag_origin;
    int32_t cdl = diag_origin - 1, cdu = diag_origin + 1, end1_diag = 0, end2_diag = 0, nonempty = 1, d = 1;
    while (d <= scaled_max) {
        int32_t curr_extent = 0, curr_seq2 = 0, curr_diag = 0;
        const int32_t tl = cdl, tu = cdu;
        GOff *cur = GROW(d);
        int32_t xs = max_score[d - xoff] + scf * d - xdrop;
        xs = (int32_t)ceil((double)xs / (double)half);
        if (xs < 0) xs = 0;
        for (int32_t k = tl; k <= tu; k++) {
            seq2_index = kInvalid;
            if (k + 1 <= diag_upper[d - goe] && k + 1 >= diag_lower[d - goe]) seq2_index = GROW(d - goe)[k + 1].match_off;
            if (k + 1 <= diag_upper[d - gap_extend] && k + 1 >= diag_lower[d - gap_extend] &&
                seq2_index < GROW(d - gap_extend)[k + 1].delete_off) seq2_index = GROW(d - gap_extend)[k + 1].delete_off;
            cur[k].delete_off = (seq2_index == kInvalid) ? kInvalid : seq2_index + 1;
            seq2_index = kInvalid;
            if (k - 1 <= diag_upper[d - goe] && k - 1 >= diag_lower[d - goe]) seq2_index = GROW(d - goe)[k - 1].match_off;
            if (k - 1 <= diag_upper[d - gap_extend] && k - 1 >= diag_lower[d - gap_extend] &&
                seq2_index < GROW(d - gap_extend)[k - 1].insert_off) seq2_index = GROW(d - gap_extend)[k - 1].insert_off;
            cur[k].insert_off = seq2_index;
            seq2_index = max(cur[k].insert_off, cur[k].delete_off);
            if (k <= diag_upper[d - op_cost] && k >= diag_lower[d - op_cost])
                seq2_index = max(seq2_index, GROW(d - op_cost)[k].match_off + 1);
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xs) {
                if (k == cdl) cdl++; else cur[k].match_off = kInvalid;
                continue;
            }
            cdu = k;
            index = reverse ? match_run_rev(q, subj, len1, len2, seq1_index, seq2_index)
                            : match_run_fwd(q, subj, len1, len2, seq1_index, seq2_index, s_base);
            if (index > longest) { seed.start_q = seq1_index; seed.start_s = seq2_index; seed.match_length = longest = index; }
            seq1_index += index; seq2_index += index;
            cur[k].match_off = seq2_index;
            if (seq1_index + seq2_index > curr_extent) { curr_extent = seq1_index + seq2_index; curr_seq2 = seq2_index; curr_diag = k; }
            if (seq1_index == len1) { cdu = k; end1_diag = k - 1; }
            if (seq2_index == len2) { cdl = k; end2_diag = k + 1; }
        }
        const int32_t curr_score = curr_extent * half - d * scf;
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score; best_dist = d; best_diag = curr_diag;
            *l2 = curr_seq2; *l1 = curr_seq2 + best_diag - diag_origin;
        } else max_score[d] = max_score[d - 1];
        if (cdl <= cdu) { nonempty++; diag_lower[d] = cdl; diag_upper[d] = cdu; }
        else { diag_lower[d] = kInvalidDiag; diag_upper[d] = -kInvalidDiag; }
        if (diag_lower[d - max_penalty] <= diag_upper[d - max_penalty]) nonempty--;
        if (nonempty == 0) break;
        d++;
        cdl = min(diag_lower[d - goe], diag_lower[d - gap_extend]) - 1;
        cdl = min(cdl, diag_lower[d - op_cost]);
        if (end2_diag > 0) cdl = max(cdl, end2_diag);
        cdu = max(diag_upper[d - goe], diag_upper[d - gap_extend]) + 1;
        cdu = max(cdu, diag_upper[d - op_cost]);
        if (end1_diag > 0) cdu = min(cdu, end1_diag);
    }
#undef GROW
    return max_score[best_dist];
}
}  // namespace

namespace {
// one initial hit; `slot` = the thread's scratch slot
__device__ void greedy_hit(const GbnGapParams &P, int64_t i, int64_t slot)
{
    const GbnDevInitHit h = P.ihits[P.first + i];
    const uint8_t *__restrict__ subj = P.db + P.byte_off[h.subj];
    const int32_t slen = P.len[h.subj];
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > h.q_off) hi = m; else lo = m; }
    const int32_t qstart = P.ctx_off[lo], qlen = P.ctx_len[lo];
    const GQ q = {P.q2, P.qinv, (int64_t)qstart};
    const int32_t q_start_u = h.q_start - qstart;
    // start in the middle of the ungapped HSP (CORE/blast_gapalign.c:3466-3471)
    const int32_t q_off = q_start_u + h.length / 2, s_off = h.s_start + h.length / 2;
    int32_t *scratch = P.scratch + (size_t)slot * P.scratch_per_thread;
    int32_t *row0 = scratch, *row1 = scratch + P.row_len, *msb = scratch + 2 * (size_t)P.row_len;
    int32_t reward = P.reward, pen = -P.penalty, X = P.xdrop;
    int32_t mc = reward, mm = pen;
    if (mc % 2 == 1) { mc *= 2; mm *= 2; X *= 2; }
    int32_t qr, sr, ql, sl; GSeed fwd, rev;
    int32_t score;
    if (P.gap_open == 0 && P.gap_extend == 0) {
        int32_t dist = greedy_linear(q + q_off, qlen - q_off, subj, s_off, slen - s_off, false, X, mc, mm, &qr, &sr, fwd, row0, row1, msb);
        dist += greedy_linear(q, q_off, subj, 0, s_off, true, X, mc, mm, &ql, &sl, rev, row0, row1, msb);
        score = (qr + sr + ql + sl) * reward / 2 - dist * (reward - P.penalty);
    } else {
        int32_t go = P.gap_open, ge = P.gap_extend;
        if (reward % 2 == 1) { go *= 2; ge *= 2; }
        score = greedy_affine(q + q_off, qlen - q_off, subj, s_off, slen - s_off, false, X, mc, mm, go, ge, &qr, &sr, fwd, scratch, P.row_len);
        score += greedy_affine(q, q_off, subj, 0, s_off, true, X, mc, mm, go, ge, &ql, &sl, rev, scratch, P.row_len);
        if (reward % 2 == 1) score /= 2;
    }
    int32_t q_box_l = q_off - ql, s_box_l = s_off - sl, q_box_r = q_off + qr, s_box_r = s_off + sr;
    int32_t qsl = q_off - rev.start_q, ssl = s_off - rev.start_s;
    int32_t qsr = q_off + fwd.start_q, ssr = s_off + fwd.start_s;
    int32_t vl = 0, vr = 0;
    if (qsr < q_box_r && ssr < s_box_r) { vr = min(min(q_box_r - qsr, s_box_r - ssr), fwd.match_length) / 2; }
    else { qsr = q_off; ssr = s_off; }
    if (qsl > q_box_l && ssl > s_box_l) { vl = min(min(qsl - q_box_l, ssl - s_box_l), rev.match_length) / 2; }
    else { qsl = q_off; ssl = s_off; }
    GbnDevGapped g;
    if (vr > vl) { g.seed_q = qsr + vr; g.seed_s = ssr + vr; } else { g.seed_q = qsl - vl; g.seed_s = ssl - vl; }
    g.q_start = q_box_l; g.s_start = s_box_l; g.q_stop = q_box_r; g.s_stop = s_box_r;
    g.score = score; g.context = lo;
    P.out[P.first + i] = g;
}
}  // namespace

// ---------------------------------------------------------------------------
// greedy_wave_kernel: BLAST_GreedyAlign with the diagonals of a distance across the lanes of a wave.
// The thread-per-hit kernel below walks ~d diagonals per distance d one after the other, each with a dependent chain of
// scattered loads (the match run): ~2,000 round trips to memory per initial hit, 2.1 ms for the 1,900 hits of a C2 pass on 30
// waves.  Here a workgroup of two waves takes one hit -- wave 0 its right half, wave 1 its left half -- and a lane takes one
// diagonal: the furthest-reaching offsets of two distances live in a window of the wave's LDS, the match runs of a distance
// are in flight together, and what the reference's loop over the diagonals decides one by one (diag_lower / diag_upper, which
// cells are marked invalid, the furthest point, the longest run) comes out of four lane masks and two wave-wide maxima in
// closed form.  Same reads and writes of the same cells as greedy_linear (the oracle's order), so the same scores and boxes;
// a half that needs more distance than the window holds (GBN_GW_DMAX) is left to greedy_kernel (GBN_GAP_REDO).
// ---------------------------------------------------------------------------
namespace {
// (6.4 KB of LDS per workgroup: the probe kernel of the next pass, next to which this one runs, leaves 8 KB of a CU's 160)
#define GBN_GW_C    128         // LDS row window: diagonals -126 .. +126 around the start diagonal
#define GBN_GW_DMAX 122
#define GBN_GW_MS   256         // max_score window

__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    #pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ int32_t greedy_linear_wave(const GQ &q, int32_t len1, const uint8_t *subj, int32_t s_base, int32_t len2,
                                      bool reverse, int32_t xdrop, int32_t match_cost, int32_t mismatch_cost,
                                      int32_t *l1, int32_t *l2, GSeed &seed, int32_t *row0, int32_t *row1,
                                      int32_t *max_score_base, bool &redo)
{
    const int32_t kInvalid = -2;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long lane_lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int32_t max_dist = min(10000, len2 / 2 + 1);
    const int32_t xdrop_offset = (xdrop + match_cost / 2) / (match_cost + mismatch_cost) + 1;
    int32_t index = reverse ? match_run_rev(q, subj, len1, len2, 0, 0) : match_run_fwd(q, subj, len1, len2, 0, 0, s_base);
    index = __builtin_amdgcn_readfirstlane(index);
    *l1 = index; *l2 = index;
    seed.start_q = 0; seed.start_s = 0; seed.match_length = index;
    int32_t longest = index, best_dist = 0;
    if (index == len1 || index == len2) return 0;
    if (xdrop_offset > GBN_GW_MS - GBN_GW_DMAX - 2) { redo = true; return 0; }
    int32_t *max_score = max_score_base + xdrop_offset;
    for (int32_t t = lane; t < xdrop_offset; t += 64) max_score_base[t] = 0;
    if (lane == 0) { row0[GBN_GW_C] = index; max_score[0] = index * match_cost; }
    // diagonals relative to the start diagonal: kk = k - diag_origin, seq1_index = seq2_index + kk
    int32_t dl = -1, du = 1;
    bool end1 = false, end2 = false;
    for (int32_t d = 1; d <= max_dist; d++) {
        if (d > GBN_GW_DMAX) { redo = true; return 0; }
        const int32_t tl = dl, tu = du;
        int32_t *prev = ((d - 1) & 1) ? row1 : row0, *cur = (d & 1) ? row1 : row0;
        if (lane == 0) {
            prev[GBN_GW_C + dl - 1] = kInvalid; prev[GBN_GW_C + dl] = kInvalid;
            prev[GBN_GW_C + du] = kInvalid; prev[GBN_GW_C + du + 1] = kInvalid;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int32_t xs = max_score[d - xdrop_offset] + (match_cost + mismatch_cost) * d - xdrop;
        xs = (int32_t)ceil((double)xs / (double)(match_cost / 2));
        int32_t curr_extent = 0, curr_seq2 = 0, curr_kk = 0;
        bool open = true, any_pass = false, last_h1 = false; int32_t last_pass_kk = 0;
        for (int32_t base = tl; base <= tu; base += 64) {
            const int32_t kk = base + lane;
            const bool valid = kk <= tu;
            int32_t s2 = kInvalid, s1 = 0, idx = 0;
            if (valid) {
                s2 = max(prev[GBN_GW_C + kk + 1], prev[GBN_GW_C + kk]) + 1;
                s2 = max(s2, prev[GBN_GW_C + kk - 1]);
                s1 = s2 + kk;
            }
            const bool pass = valid && !(s2 < 0 || s1 + s2 < xs);
            if (pass) idx = reverse ? match_run_rev(q, subj, len1, len2, s1, s2) : match_run_fwd(q, subj, len1, len2, s1, s2, s_base);
            const int32_t e1 = s1 + idx, e2 = s2 + idx;
            const unsigned long long P = __ballot(pass), F = __ballot(valid && !pass);
            const unsigned long long H2 = __ballot(pass && e2 == len2), H1 = __ballot(pass && e1 == len1);
            // cells: a passing diagonal stores its offset; a failing one is marked invalid unless it sits in the run of failures
            // that moves diag_lower up (nothing passed below it since the start of the row or since a diagonal that reached
            // the end of the subject: there the reference leaves the cell as it is)
            if (pass) cur[GBN_GW_C + kk] = e2;
            else if (valid) {
                const unsigned long long below = P & lane_lt;
                const bool moves_lower = below ? (((H2 >> (63 - __clzll((long long)below))) & 1ull) != 0) : open;
                if (!moves_lower) cur[GBN_GW_C + kk] = kInvalid;
            }
            if (H2) {
                const int hl = 63 - __clzll((long long)H2);
                const unsigned long long after = hl == 63 ? 0ull : (F >> (hl + 1));
                const int run = (~after) ? (__ffsll((long long)~after) - 1) : 64;
                dl = base + hl + 1 + run;
            } else if (open) {
                const int nvalid = min(64, tu - base + 1);
                dl += P ? (__ffsll((long long)P) - 1) : nvalid;
            }
            if (P) {
                const int hp = 63 - __clzll((long long)P);
                open = ((H2 >> hp) & 1ull) != 0; any_pass = true; last_pass_kk = base + hp; last_h1 = ((H1 >> hp) & 1ull) != 0;
            }
            end1 = end1 || H1 != 0; end2 = end2 || H2 != 0;
            // the furthest point of this distance (the first diagonal that reaches it) and the longest run so far
            const int32_t ext = pass ? e1 + e2 : -1;
            const int32_t m = wave_max_i32(ext);
            if (m > curr_extent) {
                const int who = __ffsll((long long)__ballot(pass && ext == m)) - 1;
                curr_extent = m; curr_seq2 = __shfl(e2, who, 64); curr_kk = base + who;
            }
            const int32_t mi = wave_max_i32(pass ? idx : -1);
            if (mi > longest) {
                const int who = __ffsll((long long)__ballot(pass && idx == mi)) - 1;
                longest = mi; seed.start_q = __shfl(s1, who, 64); seed.start_s = __shfl(s2, who, 64); seed.match_length = mi;
            }
        }
        if (any_pass) du = last_h1 ? last_pass_kk - 1 : last_pass_kk;
        const int32_t curr_score = curr_extent * (match_cost / 2) - d * (match_cost + mismatch_cost);
        const int32_t before = max_score[d - 1];
        int32_t now_best = before;
        if (curr_score > before) { now_best = curr_score; best_dist = d; *l2 = curr_seq2; *l1 = curr_seq2 + curr_kk; }
        if (lane == 0) max_score[d] = now_best;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (dl > du) break;
        if (!end2) dl--;
        if (!end1) du++;
    }
    return best_dist;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(128) greedy_wave_kernel(GbnGapParams P)
{
    __shared__ int32_t rows[2][2][2 * GBN_GW_C + 8];
    __shared__ int32_t msb[2][GBN_GW_MS + 8];
    __shared__ int32_t half_out[2][8];        // dist, l1, l2, seed.start_q, seed.start_s, seed.match_length, redo
    const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    for (int64_t i = blockIdx.x; i < P.n; i += gridDim.x) {
        const GbnDevInitHit h = P.ihits[P.first + i];
        const uint8_t *__restrict__ subj = P.db + P.byte_off[h.subj];
        const int32_t slen = P.len[h.subj];
        int lo = 0, hi = P.nctx;
        while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > h.q_off) hi = m; else lo = m; }
        const int32_t qstart = P.ctx_off[lo], qlen = P.ctx_len[lo];
        const GQ q = {P.q2, P.qinv, (int64_t)qstart};
        const int32_t q_off = h.q_start - qstart + h.length / 2, s_off = h.s_start + h.length / 2;
        const int32_t reward = P.reward, pen = -P.penalty;
        int32_t X = P.xdrop, mc = reward, mm = pen;
        if (mc % 2 == 1) { mc *= 2; mm *= 2; X *= 2; }
        int32_t a1 = 0, a2 = 0; GSeed sd; bool redo = false;
        int32_t dist;
        if (w == 0) dist = greedy_linear_wave(q + q_off, qlen - q_off, subj, s_off, slen - s_off, false, X, mc, mm, &a1, &a2, sd, rows[0][0], rows[0][1], msb[0], redo);
        else dist = greedy_linear_wave(q, q_off, subj, 0, s_off, true, X, mc, mm, &a1, &a2, sd, rows[1][0], rows[1][1], msb[1], redo);
        if (lane == 0) {
            int32_t *o = half_out[w];
            o[0] = dist; o[1] = a1; o[2] = a2; o[3] = sd.start_q; o[4] = sd.start_s; o[5] = sd.match_length; o[6] = redo ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int32_t *f = half_out[0], *r = half_out[1];
            GbnDevGapped g;
            if (f[6] || r[6]) { g.q_start = g.q_stop = g.s_start = g.s_stop = g.seed_q = g.seed_s = 0; g.context = lo; g.score = GBN_GAP_REDO; }
            else {
                const int32_t qr = f[1], sr = f[2], ql = r[1], sl = r[2];
                const int32_t score = (qr + sr + ql + sl) * reward / 2 - (f[0] + r[0]) * (reward - P.penalty);
                const int32_t q_box_l = q_off - ql, s_box_l = s_off - sl, q_box_r = q_off + qr, s_box_r = s_off + sr;
                int32_t qsl = q_off - r[3], ssl = s_off - r[4];
                int32_t qsr = q_off + f[3], ssr = s_off + f[4];
                int32_t vl = 0, vr = 0;
                if (qsr < q_box_r && ssr < s_box_r) { vr = min(min(q_box_r - qsr, s_box_r - ssr), f[5]) / 2; }
                else { qsr = q_off; ssr = s_off; }
                if (qsl > q_box_l && ssl > s_box_l) { vl = min(min(qsl - q_box_l, ssl - s_box_l), r[5]) / 2; }
                else { qsl = q_off; ssl = s_off; }
                if (vr > vl) { g.seed_q = qsr + vr; g.seed_s = ssr + vr; } else { g.seed_q = qsl - vl; g.seed_s = ssl - vl; }
                g.q_start = q_box_l; g.s_start = s_box_l; g.q_stop = q_box_r; g.s_stop = s_box_r;
                g.score = score; g.context = lo;
            }
            P.out[P.first + i] = g;
        }
        __syncthreads();
    }
}

// The gapped kernels run on the second stream underneath the next range's scan, whose kernels need whole
// CUs (one 1024-thread workgroup + most of the LDS each): a grid sized to the initial hits would fill
// every wave slot for as long as its longest extension lasts and keep those workgroups waiting.  So the
// grid is capped (the engine asks for 24 waves per CU) and the threads walk the hits with a grid stride; the
// scratch then depends on the grid, not on the number of hits.
extern "C" __global__ void greedy_kernel(GbnGapParams P)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = slot; i < P.n; i += total) {
        if (P.redo_only && P.out[P.first + i].score != GBN_GAP_REDO) continue;     // second launch: what greedy_wave_kernel left
        greedy_hit(P, i, slot);
    }
}

namespace {
struct GapDP { int32_t best, best_gap; };
#define GBN_MININT (INT32_MIN / 2)

// s_BlastAlignPackedNucl: forward reads query[q0+b] / subject[s0+a-1];
// reverse reads query[N-1-b] / subject[M-a].  `sa` is this lane's column of a wave-interleaved array
// (stride 64 cells, see dynprog_kernel).
__device__ int32_t align_packed(const GbnGapParams &P, const uint8_t *q, const uint8_t *subj,
                                int32_t q0, int32_t s0, int32_t N, int32_t M, int32_t *b_off, int32_t *a_off,
                                bool reverse, GapDP *sa, int32_t cap, int *overflow)
{
    const int32_t gap_open = P.gap_open, gap_extend = P.gap_extend, goe = gap_open + gap_extend;
    auto SA = [&](int32_t b) -> GapDP & { return sa[(size_t)b * 64]; };
    int32_t x_dropoff = P.xdrop;
    *a_off = 0; *b_off = 0;
    if (x_dropoff < goe) x_dropoff = goe;
    if (N <= 0 || M <= 0) return 0;
    int32_t score = -goe, i;
    SA(0).best = 0; SA(0).best_gap = -goe;
    for (i = 1; i <= N; i++) {
        if (score < -x_dropoff) break;
        if (i >= cap) { *overflow = 1; return 0; }
        SA(i).best = score; SA(i).best_gap = score - goe; score -= gap_extend;
    }
    int32_t b_size = i, best_score = 0, first_b = 0, last_b;
    for (int32_t a = 1; a <= M; a++) {
        const int ab = reverse ? base_at(subj, M - a) : base_at(subj, (int64_t)s0 + a - 1);
        const int32_t *row = P.matrix + ab * 16;
        int32_t sc = GBN_MININT, sgr = GBN_MININT;
        last_b = first_b;
        for (int32_t b = first_b; b < b_size; b++) {
            const uint8_t bl = reverse ? q[N - 1 - b] : q[q0 + b];
            int32_t sgc = SA(b).best_gap;
            int32_t next = SA(b).best + row[bl];
            if (sc < sgc) sc = sgc;
            if (sc < sgr) sc = sgr;
            if (best_score - sc > x_dropoff) {
                if (b == first_b) first_b++; else SA(b).best = GBN_MININT;
            } else {
                last_b = b;
                if (sc > best_score) { best_score = sc; *a_off = a; *b_off = b; }
                sgr -= gap_extend; sgc -= gap_extend;
                SA(b).best_gap = max(sc - goe, sgc);
                sgr = max(sc - goe, sgr);
                SA(b).best = sc;
            }
            sc = next;
        }
        if (first_b == b_size) break;
        if (last_b < b_size - 1) {
            b_size = last_b + 1;
        } else {
            while (sgr >= (best_score - x_dropoff) && b_size <= N) {
                if (b_size >= cap) { *overflow = 1; return 0; }
                SA(b_size).best = sgr; SA(b_size).best_gap = sgr - goe; sgr -= gap_extend; b_size++;
            }
        }
        if (b_size <= N) {
            if (b_size >= cap) { *overflow = 1; return 0; }
            SA(b_size).best = GBN_MININT; SA(b_size).best_gap = GBN_MININT; b_size++;
        }
    }
    return best_score;
}
}  // namespace

namespace {
__device__ void dynprog_hit(const GbnGapParams &P, int64_t i)
{
    const GbnDevInitHit h = P.ihits[P.first + i];
    const uint8_t *__restrict__ subj = P.db + P.byte_off[h.subj];
    const int32_t slen = P.len[h.subj];
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > h.q_off) hi = m; else lo = m; }
    const int32_t qstart = P.ctx_off[lo], qlen = P.ctx_len[lo];
    const uint8_t *q = P.q8 + qstart;
    int32_t q_off = h.q_off - qstart, s_off = h.s_off;
    const int32_t s_end = h.s_start + h.length;
    if (s_end >= s_off + 8) { s_off += 3; q_off += 3; }       // CORE/blast_gapalign.c:3494-3497
    // DP rows of the 64 lanes of a wave interleaved cell by cell (cell b of lane l at [b * 64 + l]): lanes
    // walk their bands roughly in step, so a wave's accesses to cell b fall into one 512-byte stretch
    GapDP *sa = reinterpret_cast<GapDP *>(P.scratch + (size_t)blockIdx.x * 64 * P.scratch_per_thread) + (threadIdx.x & 63);
    const int32_t cap = P.scratch_per_thread / 2;
    int overflow = 0;
    int32_t adj = 4 - (s_off & 3);
    int32_t q_length = q_off + adj, s_length = s_off + adj;
    if (q_length > qlen || s_length > slen) { q_length -= 4; s_length -= 4; }
    int32_t pq, ps;
    GbnDevGapped g; g.context = lo; g.seed_q = q_off; g.seed_s = s_off;
    int32_t left = align_packed(P, q, subj, 0, 0, q_length, s_length, &pq, &ps, true, sa, cap, &overflow);
    g.q_start = q_length - pq; g.s_start = s_length - ps;
    int32_t right = 0;
    if (q_length < qlen && s_length < slen) {
        right = align_packed(P, q, subj, q_length, s_length, qlen - q_length, slen - s_length, &pq, &ps, false, sa, cap, &overflow);
        g.q_stop = pq + q_length; g.s_stop = ps + s_length;
    } else { g.q_stop = q_length; g.s_stop = s_length; }
    g.score = overflow ? INT32_MIN : left + right;
    P.out[P.first + i] = g;
}
}  // namespace

extern "C" __global__ void dynprog_kernel(GbnGapParams P)
{
    const int64_t total = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += total) {
        if (P.redo_only && P.out[P.first + i].score != GBN_GAP_REDO) continue;     // second launch: what the wave kernel left
        dynprog_hit(P, i);
    }
}

// ---------------------------------------------------------------------------------------------------
// The same extension (s_BlastAlignPackedNucl, CORE/blast_gapalign.c:2842-3056), one extension per WAVE:
// lane l holds the DP cell {best, best_gap} of the query column b with b mod 64 == l inside the active
// window [first_b, b_size) -- the band lives in registers, no scratch memory -- and a subject row is
// evaluated by all lanes at once.  The reference walks a row left to right with three running values;
// they become prefix operations over the lanes, in the row's own order (the window is circular in the
// lanes), with exactly the reference's results:
//   score entering column b     = old best[b-1] + match(b-1): one lane rotation
//   score_gap_row entering b    = max over KEPT columns b' < b of S[b'] - gap_open_extend - gap_extend * (kept
//                                 columns between b' and b): the running value only decays on columns that pass the
//                                 X-drop test.  Opening a gap out of a gap never beats extending it, so S[b'] may be
//                                 replaced by H[b'] = max(diagonal, vertical gap), which does not depend on the row's
//                                 horizontal gaps: a prefix maximum of H[b'] + gap_extend * K(b'+1), K = kept columns
//                                 before a column (a population count of the kept mask)
//   best_score seen by column b = max(best so far, S of the kept columns b' < b): a prefix maximum
//   kept(b)                     = best seen - S[b] <= X
// kept depends on S and the running best, S on kept: the row is iterated from "all kept" until the kept mask
// repeats.  Column j of the window is final after j rounds (its inputs lie left of it), so the loop ends with the
// one solution the sequential walk has; two rounds are typical.  Query letters: a lane keeps the four match
// scores of its column of the current 64-column block and of the next two, reloaded a block ahead.  Subject bases:
// 256 at a time in the lanes, read with v_readlane.  A window wider than 62 columns (or gap_extend 0) is left to
// dynprog_kernel (GBN_GAP_REDO).
namespace {
constexpr int32_t kDpNeg = GBN_MININT;
// inclusive prefix maximum over the 64 lanes of a fully active wave: six v_max_i32_dpp steps (row shifts, then row
// broadcasts; a lane without a source keeps its value).  Written as assembly: the compiler turns the same steps
// into three instructions each.  (s_nop 1: a DPP source needs two wait states after the VALU write before it.)
__device__ __forceinline__ int32_t wave_scan_max_incl(int32_t v)
{
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return v;
}
// value of the lane below (lane 0: kDpNeg)
__device__ __forceinline__ int32_t lane_below(int32_t v) { return __builtin_amdgcn_update_dpp(kDpNeg, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1

__device__ int32_t align_packed_wave(const GbnGapParams &P, const uint8_t *q, const uint8_t *subj,
                                     int32_t q0, int32_t s0, int32_t N, int32_t M, int32_t *b_off, int32_t *a_off,
                                     bool reverse, int *redo, int32_t *maxw_stat = nullptr)
{
    const int lane = (int)(threadIdx.x & 63);
    const int32_t gap_extend = P.gap_extend, goe = P.gap_open + P.gap_extend;
    int32_t x = P.xdrop;
    *a_off = 0; *b_off = 0;
    if (x < goe) x = goe;
    if (N <= 0 || M <= 0) return 0;
    if (gap_extend <= 0) { *redo = 1; return 0; }
    constexpr int32_t W = 62;                                   // widest window kept here (64 lanes: the window, one sentinel, one spare)
    // match scores of a column against the four subject bases
    auto load_scores = [&](int32_t col, int32_t (&m)[4]) {
        const int32_t c = min(col, N);                          // column N is the sentinel cell: the letter past the query end
        const uint8_t letter = reverse ? q[N - 1 - c] : q[q0 + c];
        #pragma unroll
        for (int t = 0; t < 4; t++) m[t] = P.matrix[t * 16 + letter];
    };
    // scores of column cb * 64 + lane, of the block after it (the window straddles two blocks) and, loaded a block
    // ahead of its first use, of the one after that
    int32_t mA[4], mB[4], mC[4]; int32_t cb = 0;
    load_scores(lane, mA); load_scores(64 + lane, mB); load_scores(128 + lane, mC);
    // row 0
    const int32_t ninit = min(N, (x - goe) / gap_extend + 1);
    int32_t first_b = 0, b_size = ninit + 1;
    if (b_size > W) { *redo = 1; return 0; }
    int32_t best = lane == 0 ? 0 : -goe - (lane - 1) * gap_extend, bgap = best - goe;     // lane l: column b with b mod 64 == l
    int32_t best_score = 0;
    // subject bytes: 64 at a time (256 bases), byte `sb_lo + l` of the subject in lane l
    int32_t sb_lo = INT32_MIN; uint32_t sbytes = 0;
    for (int32_t a = 1; a <= M; a++) {
        const int32_t pos = reverse ? (M - a) : (s0 + a - 1), byte = pos >> 2;
        if (byte < sb_lo || byte >= sb_lo + 64) {
            sb_lo = reverse ? max(byte - 63, 0) : byte;
            sbytes = subj[(int64_t)sb_lo + lane];
        }
        const int ab = (int)((__builtin_amdgcn_readlane((int)sbytes, byte - sb_lo) >> (2 * (3 - (pos & 3)))) & 3);
        const int f = first_b & 63, width = b_size - first_b;
        const int32_t col = first_b + ((lane - f) & 63);
        const bool inwin_p = col < b_size;
        const bool blkA = (col >> 6) == cb;
        const int32_t m0 = blkA ? mA[0] : mB[0], m1 = blkA ? mA[1] : mB[1], m2 = blkA ? mA[2] : mB[2], m3 = blkA ? mA[3] : mB[3];
        const int32_t msel = (ab & 2) ? ((ab & 1) ? m3 : m2) : ((ab & 1) ? m1 : m0);
        // score entering a column = what the column before it hands on (lane rotation; the window's first column: none)
        int32_t D = __builtin_amdgcn_update_dpp(kDpNeg, inwin_p ? best + msel : kDpNeg, 0x13C, 0xf, 0xf, false);   // wave_ror:1
        if (lane == f) D = kDpNeg;
        const int32_t C = bgap;
        // from here to the state update: lane j = column first_b + j (one crossbar move in, one out)
        const int32_t H = __builtin_amdgcn_ds_bpermute(((lane + f) & 63) << 2, inwin_p ? max(max(D, C), kDpNeg) : kDpNeg);
        const bool inwin = lane < width;
        unsigned long long km = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
        int32_t S = kDpNeg, Ttot = kDpNeg, Stot = kDpNeg; bool kept = false;
        for (int round = 0; round < 66; round++) {
            kept = (km >> lane) & 1ull;
            const int32_t K = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));   // kept columns before this one
            const int32_t Ti = wave_scan_max_incl(kept ? max(H + gap_extend * (K + 1), kDpNeg) : kDpNeg);
            Ttot = __builtin_amdgcn_readlane(Ti, 63);
            const int32_t PM = lane_below(Ti);
            const int32_t G = PM <= kDpNeg ? kDpNeg : PM - goe - gap_extend * K;
            S = max(H, G);
            const int32_t Si = wave_scan_max_incl(kept ? S : kDpNeg);
            Stot = __builtin_amdgcn_readlane(Si, 63);
            const bool keep = inwin && !(max(best_score, lane_below(Si)) - S > x);
            const unsigned long long km2 = __ballot(keep);
#if GBN_DP_STATS
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 1, 1ull);
#endif
            if (km2 == km) break;
            km = km2;
        }
#if GBN_DP_STATS
        if (lane == 0) { atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch), 1ull); atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 4, (unsigned long long)(b_size - first_b));
            const int wd = b_size - first_b; atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 8 + min(wd / 8, 7), 1ull);
            if (maxw_stat && wd > *maxw_stat) *maxw_stat = wd; }
#endif
        if (Stot > best_score) {
            best_score = Stot; *a_off = a;
            *b_off = first_b + (int32_t)__builtin_ctzll(__ballot(kept && S == Stot));
        }
        {   // back to the lanes the columns live in
            const int j = (lane - f) & 63;
            const int32_t Sp = __builtin_amdgcn_ds_bpermute(j << 2, S);
            const bool kept_p = (km >> j) & 1ull;
            if (kept_p) { bgap = max(Sp - goe, C - gap_extend); best = Sp; }
            else if (inwin_p) best = kDpNeg;
        }
        if (km == 0) break;                                     // every column failed: first_b == b_size
        const int32_t new_first = first_b + (int32_t)__builtin_ctzll(km), last_b = first_b + 63 - (int32_t)__builtin_clzll(km);
        if (last_b < b_size - 1) {
            b_size = last_b + 1;
        } else {
            // the horizontal gap runs on past the window
            const int32_t Gfin = Ttot - goe - gap_extend * (int32_t)__popcll(km), thr = best_score - x;
            int32_t k = Gfin >= thr ? (Gfin - thr) / gap_extend + 1 : 0;
            k = min(k, max(N - b_size + 1, 0));
            if (b_size + k + 1 - new_first > W) { *redo = 1; return 0; }
            const int32_t cn = new_first + ((lane - new_first) & 63);
            if (cn >= b_size && cn < b_size + k) { best = Gfin - (cn - b_size) * gap_extend; bgap = best - goe; }
            b_size += k;
        }
        if (b_size <= N) {
            if (b_size + 1 - new_first > W + 1) { *redo = 1; return 0; }
            if (lane == (b_size & 63)) { best = kDpNeg; bgap = kDpNeg; }
            b_size++;
        }
        first_b = new_first;
        if ((first_b >> 6) > cb) {                              // the window has left block cb
            cb++;
            #pragma unroll
            for (int t = 0; t < 4; t++) { mA[t] = mB[t]; mB[t] = mC[t]; }
            load_scores((cb + 2) * 64 + lane, mC);
        }
    }
    return best_score;
}

// one half of an extension (side 0: left of the start point, reversed; side 1: right of it) by one wave; the two waves
// of a 128-thread workgroup take the two halves of the same initial hit -- what this kernel is left with are the
// long extensions, and a launch lasts as long as its longest wave
struct HalfOut { int32_t score, pq, ps, redo; };
__device__ HalfOut dynprog_half_wave(const GbnGapParams &P, int64_t i, int side, GbnDevGapped &g, int32_t &q_length, int32_t &s_length)
{
    const GbnDevInitHit h = P.ihits[P.first + i];
    const int32_t subj_id = __builtin_amdgcn_readfirstlane(h.subj), h_q_off = __builtin_amdgcn_readfirstlane(h.q_off),
                  h_s_off = __builtin_amdgcn_readfirstlane(h.s_off), h_s_start = __builtin_amdgcn_readfirstlane(h.s_start),
                  h_length = __builtin_amdgcn_readfirstlane(h.length);
    const uint8_t *__restrict__ subj = P.db + P.byte_off[subj_id];
    const int32_t slen = P.len[subj_id];
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > h_q_off) hi = m; else lo = m; }
    const int32_t qstart = P.ctx_off[lo], qlen = P.ctx_len[lo];
    const uint8_t *q = P.q8 + qstart;
    int32_t q_off = h_q_off - qstart, s_off = h_s_off;
    const int32_t s_end = h_s_start + h_length;
    if (s_end >= s_off + 8) { s_off += 3; q_off += 3; }       // CORE/blast_gapalign.c:3494-3497
    const int32_t adj = 4 - (s_off & 3);
    q_length = q_off + adj; s_length = s_off + adj;
    if (q_length > qlen || s_length > slen) { q_length -= 4; s_length -= 4; }
    g.context = lo; g.seed_q = q_off; g.seed_s = s_off;
    HalfOut o; o.score = 0; o.pq = 0; o.ps = 0; o.redo = 0;
    int32_t maxw = 0;
    if (side == 0) o.score = align_packed_wave(P, q, subj, 0, 0, q_length, s_length, &o.pq, &o.ps, true, &o.redo, &maxw);
    else if (q_length < qlen && s_length < slen)
        o.score = align_packed_wave(P, q, subj, q_length, s_length, qlen - q_length, slen - s_length, &o.pq, &o.ps, false, &o.redo, &maxw);
#if GBN_DP_STATS
    if ((threadIdx.x & 63) == 0 && side == 0) { atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 2, 1ull); if (o.redo) atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 3, 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(P.scratch) + 16 + min(maxw / 8, 7), 1ull); }
#endif
    return o;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(128) dynprog_wave_kernel(GbnGapParams P, const unsigned long long *redo_count, const int32_t *redo_list)
{
    __shared__ HalfOut s_half[2];
    const int side = threadIdx.x >> 6;
    // after dynprog_lane_kernel: the extensions it listed; else every one
    const int64_t todo = redo_list ? (int64_t)*redo_count : P.n;
    for (int64_t k = blockIdx.x; k < todo; k += gridDim.x) {
        const int64_t i = redo_list ? (int64_t)redo_list[k] : k;
        if (!redo_list && P.redo_only && P.out[P.first + i].score != GBN_GAP_REDO) continue;
        GbnDevGapped g; int32_t q_length, s_length;
        const HalfOut o = dynprog_half_wave(P, i, side, g, q_length, s_length);
        if ((threadIdx.x & 63) == 0) s_half[side] = o;
        __syncthreads();
        if (threadIdx.x == 0) {
            const HalfOut l = s_half[0], r = s_half[1];
            g.q_start = q_length - l.pq; g.s_start = s_length - l.ps;
            g.q_stop = q_length + r.pq; g.s_stop = s_length + r.ps;
            g.score = (l.redo || r.redo) ? GBN_GAP_REDO : l.score + r.score;
            P.out[P.first + i] = g;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The same extension once more, one extension per LANE: the reference's own sequential walk of a row
// (CORE/blast_gapalign.c:2957-3052) as a per-lane state machine, every lane of a wave on an extension of its own.
// What makes the thread-per-extension form slow in dynprog_kernel -- the band in scratch memory, rows of different
// lengths in lock step -- is avoided: the band {best, best_gap} and the column's four match scores live in LDS
// (a circular window of GBN_LANE_W columns per lane), one loop iteration is one CELL of whatever row the lane is
// in (or one column appended at the end of a row), and a lane that finishes its extension takes the next one from
// a counter -- no lane waits for its neighbours' rows.  The chance hits that make up the blastn workload (some
// 40 rows of 20 columns either side) keep all 64 lanes busy; an extension whose window outgrows the LDS slots or
// that runs longer than GBN_LANE_ROWS rows in one direction (a real homolog: better on a whole wave) is left to
// dynprog_wave_kernel (GBN_GAP_REDO).  Global loads never sit in the cell loop: query letters arrive through a
// four-deep FIFO ahead of the window's right edge, subject bases 16 at a time with the next word in flight, and
// the set-up of new work (hit, context, first letters) is done for several lanes at once.
// ---------------------------------------------------------------------------------------------------
#ifndef GBN_LANE_W
#define GBN_LANE_W 32
#endif
#ifndef GBN_LANE_ROWS
#define GBN_LANE_ROWS 96          // measured: 192 rows 2.45 + 0.72 ms (lane + wave kernel), 128: 2.18 + 0.72, 96: 2.10 + 0.71, 64: 2.02 + 1.01
#endif
#ifndef GBN_LANE_LOOP_V1
#define GBN_LANE_LOOP_V1 0       // 1: the cell loop as it was before round 3's instruction diet (A/B builds)
#endif
#ifndef GBN_LANE_DUP
#define GBN_LANE_DUP 0        // timing experiment: the lane DP's cell loop executed twice (results unchanged)
#endif
#ifndef GBN_LANE_BATCH
#define GBN_LANE_BATCH 8            // lanes waiting for set-up before the (long-latency) set-up code runs
#endif

// context of every initial hit, found once by a thread of its own (a binary search = a chain of dependent loads)
extern "C" __global__ void gap_context_kernel(GbnGapParams P, int32_t *ctx_of)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const int32_t q_off = P.ihits[P.first + i].q_off;
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > q_off) hi = m; else lo = m; }
    ctx_of[i] = lo;
}

extern "C" __global__ void __launch_bounds__(64) dynprog_lane_kernel(GbnGapParams P, unsigned long long *counter, const int32_t *ctx_of, int32_t *redo_list)
{
    constexpr int W = GBN_LANE_W;
    constexpr int32_t NEG = GBN_MININT;
    enum { RUN = 0, START = 1, DONE = 2 };
    // {best, best_gap} of column c at [c mod W][lane], 16 bits each: 10 KB of LDS per wave instead of 18, i.e. 16
    // waves per CU instead of 8.  Scores of this kernel's extensions fit (the launcher checks reward x the longest
    // context); "dead" is -32768: a dead value is never the larger operand of a maximum that is kept, and whatever is
    // added to it stays below every live score, so it acts exactly as the reference's INT4_MIN / 2 does.
    __shared__ uint32_t s_cell[W][64];
    auto pack_cell = [](int32_t best, int32_t gap) -> uint32_t {
        return ((uint32_t)max(best, -32768) & 0xffffu) | ((uint32_t)max(gap, -32768) << 16); };
    auto cell_best = [](uint32_t w) -> int32_t { return (int32_t)(int16_t)(w & 0xffffu); };
    auto cell_gap = [](uint32_t w) -> int32_t { return (int32_t)w >> 16; };
    __shared__ uint8_t s_let[W][64];            // its query letter
    __shared__ uint32_t s_pm[16];               // per query letter: the match scores against the four subject bases, a byte each (-128: the sentinel's NEG)
    const int lane = threadIdx.x;
    if (lane < 16) {
        uint32_t v = 0;
        for (int t = 0; t < 4; t++) { const int32_t m = P.matrix[t * 16 + lane]; v |= (uint32_t)((m < -127 ? -128 : m) & 0xff) << (8 * t); }
        s_pm[lane] = v;
    }
    __syncthreads();
    const int32_t ge = P.gap_extend, goe = P.gap_open + P.gap_extend;
    const int32_t x = P.xdrop < goe ? goe : P.xdrop;
    const unsigned long long lt = (1ull << lane) - 1ull;

#if GBN_LANE_DUP
    uint32_t dup_sink = 0, dup_zero = 0;
    asm volatile("" : "+v"(dup_zero));
#endif
    // ---- per-lane state
    int mode = START;
    bool need_new = true, reverse = false, row0 = false;
    int est = 2;                                                      // the row's end: 0 the gap run goes on, 1 the sentinel column is due, 2 nothing (left) to do
    int64_t i = -1;
    const uint8_t *q = P.q8; const uint32_t *sp = reinterpret_cast<const uint32_t *>(P.db);   // context's query, subject (as words); always readable
    int32_t qlen = 0, slen = 0, q_length = 0, s_length = 0, ctx = 0, seed_q = 0, seed_s = 0, left_score = 0, g_q_start = 0, g_s_start = 0;
    const uint8_t *qp = P.q8; int32_t qs = 1;                         // letter of column c = qp[qs * c]
    int32_t s0 = 0, N = 0, M = 0;
    int32_t a = 0, first_b = 0, b_size = 0, hw = 0;
    int fix = 0, six = 0;                                             // first_b, b_size modulo W
    int32_t best_score = 0, a_off = 0, b_off = 0, sgr = NEG;
    // Global loads are issued once per round of the loop below by all lanes at once and taken over a round later,
    // when they have long arrived: nothing in between waits on memory.  Letters of the columns hw .. hw + ln - 1
    // wait in `lq` (4 bits each), eight more are fetched when eight or fewer are left; the subject word after the
    // one in use is fetched as soon as that one is taken.
    unsigned long long lq = 0; int ln = 0;
    uint32_t pb[8]; bool pend_l = false;
    uint32_t sw = 0, swn = 0, pw = 0; int32_t wi = 0, wmax = 0; bool have_next = false, pend_w = false;
    auto inc = [](int v) { return (W & (W - 1)) == 0 ? ((v + 1) & (W - 1)) : (v + 1 == W ? 0 : v + 1); };
    auto letter_at = [&](int32_t c) -> uint32_t { return qp[(int64_t)qs * max(min(c, N), 0)]; };
    auto subject_word = [&](int32_t w) -> uint32_t { return sp[min(max(w, 0), wmax)]; };

    // One round = one subject row of every running lane, in three phases the lanes go through together:
    //   rows begin (or a half ends) -> the cells of the row, as many steps as the widest window of the wave has
    //   columns -> the row's end: the horizontal gap runs on, one column per step, and the sentinel column.
    for (;;) {
        const unsigned long long m_start = __ballot(mode == START), m_run = __ballot(mode == RUN);
        if (!m_start && !m_run) break;                                 // every lane is DONE

        // ---------------- set-up: a new extension (left half) or the right half of the current one
        if (m_start && (__popcll(m_start) >= GBN_LANE_BATCH || !m_run)) {
            const unsigned long long m_new = __ballot(mode == START && need_new);
            unsigned long long base = 0;
            if (m_new) {
                if (lane == (int)__builtin_ctzll(m_new)) base = atomicAdd(counter, (unsigned long long)__popcll(m_new));
                base = __shfl(base, (int)__builtin_ctzll(m_new));
            }
            if (mode == START) {
                bool go = true;
                if (need_new) {
                    i = (int64_t)(base + (unsigned long long)__popcll(m_new & lt));
                    if (i >= P.n) { mode = DONE; go = false; }
                    else {
                        const GbnDevInitHit h = P.ihits[P.first + i];
                        ctx = ctx_of[i];
                        const int32_t qstart = P.ctx_off[ctx];
                        qlen = P.ctx_len[ctx]; slen = P.len[h.subj];
                        q = P.q8 + qstart; sp = reinterpret_cast<const uint32_t *>(P.db + P.byte_off[h.subj]);
                        wmax = max(slen - 1, 0) >> 4;
                        int32_t q_off = h.q_off - qstart, s_off = h.s_off;
                        if (h.s_start + h.length >= s_off + 8) { s_off += 3; q_off += 3; }      // CORE/blast_gapalign.c:3494-3497
                        const int32_t adj = 4 - (s_off & 3);
                        q_length = q_off + adj; s_length = s_off + adj;
                        if (q_length > qlen || s_length > slen) { q_length -= 4; s_length -= 4; }
                        seed_q = q_off; seed_s = s_off;
                        reverse = true; N = q_length; M = s_length; s0 = 0; qp = q + N - 1; qs = -1;
                    }
                } else {
                    reverse = false; N = qlen - q_length; M = slen - s_length; s0 = s_length; qp = q + q_length; qs = 1;
                }
                if (go) {
                    best_score = 0; a_off = 0; b_off = 0; pend_l = false; pend_w = false;
                    mode = RUN;
                    if (N <= 0 || M <= 0) {
                        a = M; first_b = 0; b_size = 0; row0 = false; est = 2; sgr = NEG; ln = 16;      // nothing to align on this side: over at the next row begin
                    } else {
                        a = 0; first_b = 0; fix = 0; b_size = 1; six = 1; hw = 1;
                        lq = 0;
                        #pragma unroll
                        for (int k = 0; k < 16; k++) lq |= (unsigned long long)(letter_at(k) & 15u) << (4 * k);
                        s_cell[0][lane] = pack_cell(0, -goe);
                        s_let[0][lane] = (uint8_t)(lq & 15u);
                        lq >>= 4; ln = 15;
                        const int32_t pos1 = reverse ? (M - 1) : s0;      // row 1's base
                        wi = pos1 >> 4; sw = subject_word(wi); swn = subject_word(wi + (reverse ? -1 : 1)); have_next = true;
                        sgr = -goe; row0 = true; est = 0;                   // row 0 = a gap run from column 1 on, no sentinel
                    }
                }
            }
            // nothing of the set-up stays in flight: a load left pending into a register the row code reads makes
            // that code wait for whatever else is outstanding, the loads of the memory point included
            __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0)
        }

        // ---------------- the memory point: take over last round's loads, issue this round's
        {
            if (pend_l) {
                uint32_t add = 0;
                #pragma unroll
                for (int k = 0; k < 8; k++) add |= (pb[k] & 15u) << (4 * k);
                lq |= (unsigned long long)add << (4 * ln); ln += 8; pend_l = false;
            }
            if (pend_w) { swn = pw; have_next = true; pend_w = false; }
            // every lane loads (clamped, always valid addresses): a conditional load ends in a register copy behind
            // it and with that in a wait right here; the flags say whose results count
            pend_l = mode == RUN && ln <= 8; pend_w = mode == RUN && !have_next;
            const int32_t c0 = hw + ln;
            #pragma unroll
            for (int k = 0; k < 8; k++) pb[k] = letter_at(c0 + k);
            pw = subject_word(wi + (reverse ? -1 : 1));
        }

        // ---------------- rows begin
        bool over = false, redo = false;
        int32_t width = 0; int ab = 0;
        bool in_row = false;                                            // this lane walks a row in this round
        if (mode == RUN && est == 2) {
            const int32_t an = a + 1;
            if (an > M) over = true;
            else if (an > GBN_LANE_ROWS) { over = true; redo = true; }  // a long one: a whole wave does it faster
            else {
                const int32_t pos = reverse ? (M - an) : (s0 + an - 1);
                const int32_t w = pos >> 4;
                bool ready = true;
                if (w != wi) {
                    if (have_next) { sw = swn; wi = w; have_next = false; } else ready = false;      // (not yet: next round)
                }
                if (ready) {
                    a = an; in_row = true; width = b_size - first_b;
                    ab = (int)((sw >> (8 * ((pos >> 2) & 3) + 6 - 2 * (pos & 3))) & 3u);
                }
            }
        }

        // ---------------- the cells of the row (CORE/blast_gapalign.c:2957-3020): as many steps as the widest window
        // of the wave has columns
        int32_t last_b = first_b; int lix = fix;
#if GBN_LANE_LOOP_V1 || (GBN_LANE_W & (GBN_LANE_W - 1))
        {
            const int32_t reward = P.reward, penalty = P.penalty;
            int32_t sc = NEG, b = first_b; int ix = fix;
            if (in_row) sgr = NEG;
            // (straight-line selects: the divergent if / else of the reference's loop body costs three times the
            // instructions once the compiler has structurized it)
            uint32_t cw = s_cell[ix][lane]; uint32_t letter = s_let[ix][lane];
            for (int32_t t = 0; t < width; t++) {                       // (a lane leaves the loop after its last column)
                {
                    // the next column's cell is on its way while this one is worked on (its slot is not written here)
                    const int ixn = inc(ix);
                    const uint32_t cwn = s_cell[ixn][lane]; const uint32_t letter_n = s_let[ixn][lane];
                    const int32_t c_best = cell_best(cw), c_gap = cell_gap(cw);
                    int32_t msel = (int)letter == ab ? reward : penalty;
                    if (__ballot(letter >= 4u)) {                       // ambiguity codes, the sentinel: from the matrix
                        const uint32_t mm = s_pm[letter];
                        int32_t m2 = (int32_t)(int8_t)(mm >> (8 * ab));
                        m2 = m2 == -128 ? NEG : m2;
                        msel = letter >= 4u ? m2 : msel;
                    }
                    const int32_t next = c_best + msel;
                    sc = max(sc, max(c_gap, sgr));
                    const bool keep = !(best_score - sc > x);
                    const bool drop_first = !keep && b == first_b;
                    const bool better = keep && sc > best_score;
                    const int32_t open = sc - goe;
                    s_cell[ix][lane] = pack_cell(keep ? sc : NEG, keep ? max(open, c_gap - ge) : c_gap);   // (a failed first column leaves the window: what is stored there does not matter)
                    sgr = keep ? max(open, sgr - ge) : sgr;
                    last_b = keep ? b : last_b; lix = keep ? ix : lix;
                    best_score = better ? sc : best_score; a_off = better ? a : a_off; b_off = better ? b : b_off;
                    first_b += drop_first ? 1 : 0; fix = drop_first ? inc(fix) : fix;
                    sc = next; b++; ix = ixn; cw = cwn; letter = letter_n;
                }
            }
        }

#else
        {
            // The kernel is bound by VALU issue (93 % of the SIMDs' issue slots, 1.18e9 instructions per range), so the loop
            // carries only what cannot be had afterwards: the slots follow from the columns (slot = column mod W), the row of
            // the best score from whether the best score moved in this row, "the window's first column failed" is a flag
            // that stays up while the columns fail from the left.
            int32_t rew_v = P.reward, pen_v = P.penalty;
            asm volatile("" : "+v"(rew_v), "+v"(pen_v));                // (kept in VGPRs: the select below needs them there every step)
            const int32_t first_b0 = first_b, b_end = first_b + width, best_before = best_score;
            int32_t sc = NEG, b = first_b;
            bool lead = true;
#if GBN_LANE_DUP
            const int32_t dup_sgr = in_row ? NEG : sgr; const int dup_fix = fix;
#endif
            if (in_row) sgr = NEG;
            int ix = fix;
            uint32_t cw = s_cell[ix][lane]; uint32_t letter = s_let[ix][lane];
            // (written out twice per turn, or with the slot's LDS address carried instead of the slot, the compiler turns
            // the selects below into divergent branches and the loop is no faster than it was: measured)
            while (b < b_end) {                                         // (a lane leaves the loop after its last column)
                // the next column's cell is on its way while this one is worked on (its slot is not written here)
                const int ixn = inc(ix);
                const uint32_t cwn = s_cell[ixn][lane]; const uint32_t letter_n = s_let[ixn][lane];
                const int32_t c_best = cell_best(cw), c_gap = cell_gap(cw);
                int32_t msel = (int)letter == ab ? rew_v : pen_v;
                if (__ballot(letter >= 4u)) {                           // ambiguity codes, the sentinel: from the matrix
                    const uint32_t mm = s_pm[letter];
                    int32_t m2 = (int32_t)(int8_t)(mm >> (8 * ab));
                    m2 = m2 == -128 ? NEG : m2;
                    msel = letter >= 4u ? m2 : msel;
                }
                const int32_t next = c_best + msel;
                sc = max(sc, max(c_gap, sgr));
                const bool keep = !(best_score - sc > x);
                const bool better = keep & (sc > best_score);
                const int32_t open = sc - goe;
                lead = lead & !keep;
                s_cell[ix][lane] = pack_cell(keep ? sc : NEG, keep ? max(open, c_gap - ge) : c_gap);   // (a failed first column leaves the window: what is stored there does not matter)
                sgr = keep ? max(open, sgr - ge) : sgr;
                last_b = keep ? b : last_b;
                best_score = better ? sc : best_score; b_off = better ? b : b_off;
                first_b += lead ? 1 : 0;
                sc = next; b++; ix = ixn; cw = cwn; letter = letter_n;
            }
#if GBN_LANE_DUP
            // timing experiment (round 5): the cell loop ONCE MORE on shadow state, same trip counts, same LDS reads, its writes
            // into the slots it has just read (the values that are there); the kernel's results do not change.  The kernel's
            // time with it minus its time without = what the cell loop costs in the kernel as it runs.
            {
                int32_t sc2 = NEG, b2 = first_b0, sgr2 = dup_sgr, best2 = best_before, boff2 = 0, last2 = first_b0, first2 = first_b0;
                bool lead2 = true; int ix2 = dup_fix;
                uint32_t cw2 = s_cell[ix2][lane]; uint32_t letter2 = s_let[ix2][lane];
                while (b2 < b_end) {
                    const int ixn = inc(ix2);
                    const uint32_t cwn = s_cell[ixn][lane]; const uint32_t letter_n = s_let[ixn][lane];
                    const int32_t c_best = cell_best(cw2), c_gap = cell_gap(cw2);
                    int32_t msel = (int)letter2 == ab ? rew_v : pen_v;
                    if (__ballot(letter2 >= 4u)) {
                        const uint32_t mm = s_pm[letter2];
                        int32_t m2 = (int32_t)(int8_t)(mm >> (8 * ab));
                        m2 = m2 == -128 ? NEG : m2;
                        msel = letter2 >= 4u ? m2 : msel;
                    }
                    const int32_t next = c_best + msel;
                    sc2 = max(sc2, max(c_gap, sgr2));
                    const bool keep = !(best2 - sc2 > x);
                    const bool better = keep & (sc2 > best2);
                    const int32_t open = sc2 - goe;
                    lead2 = lead2 & !keep;
                    const uint32_t wv = pack_cell(keep ? sc2 : NEG, keep ? max(open, c_gap - ge) : c_gap);
                    s_cell[ix2][lane] = (wv & dup_zero) | cw2;          // (dup_zero = 0, unknown to the compiler: the slot keeps its value)
                    sgr2 = keep ? max(open, sgr2 - ge) : sgr2;
                    last2 = keep ? b2 : last2;
                    best2 = better ? sc2 : best2; boff2 = better ? b2 : boff2;
                    first2 += lead2 ? 1 : 0;
                    sc2 = next; b2++; ix2 = ixn; cw2 = cwn; letter2 = letter_n;
                }
                dup_sink += (uint32_t)(best2 + boff2 + last2 + first2 + sgr2);
            }
#endif
            a_off = best_score > best_before ? a : a_off;
            lix = (fix + (last_b - first_b0)) & (W - 1);
            fix = (fix + (first_b - first_b0)) & (W - 1);
        }
#endif

        // ---------------- the row's end (CORE/blast_gapalign.c:3022-3052): every column failed -> the half is over;
        // the window shrinks, or the horizontal gap runs on past it, one column per step; then the sentinel column
        if (in_row) {
            if (first_b == b_size) over = true;
            else {
                if (last_b < b_size - 1) { b_size = last_b + 1; six = inc(lix); sgr = NEG; }
                est = 0;
            }
        }
        for (;;) {
            const bool more = sgr >= best_score - x && b_size <= N;
            est = (est == 0 && !more) ? ((!row0 && b_size <= N) ? 1 : 2) : est;
            // column c lives in slot c mod W: [first_b, max(b_size, hw)) must stay within W columns
            const bool full = est < 2 && max(b_size + 1, hw) - first_b > W;       // no slot left: leave it to the wave kernel
            over = over || full; redo = redo || full; est = full ? 2 : est;
            const bool fresh = b_size == hw;
            const bool step = est < 2 && !(fresh && ln == 0);           // (its letter has not arrived yet: next round)
            if (!__ballot(step)) break;
            if (step) {
                s_cell[six][lane] = est == 0 ? pack_cell(sgr, sgr - goe) : pack_cell(NEG, NEG);
                if (fresh) { s_let[six][lane] = (uint8_t)(lq & 15u); lq >>= 4; ln--; hw++; }
                sgr -= ge; b_size++; six = inc(six);
                est = est == 1 ? 2 : est;
            }
        }
        if (est == 2) row0 = false;

        // ---------------- a half is over
        if (over) {
            est = 2;
            if (redo) {                                                 // given up: dynprog_wave_kernel redoes the whole extension
                P.out[P.first + i].score = GBN_GAP_REDO;
                redo_list[atomicAdd(counter + 1, 1ull)] = (int32_t)i;  // (a few per cent: listed, so that nobody has to look for them)
                need_new = true; mode = START;
            } else if (reverse) {
                left_score = best_score; g_q_start = q_length - b_off; g_s_start = s_length - a_off;
                if (q_length < qlen && s_length < slen) { need_new = false; mode = START; }
                else {
                    GbnDevGapped g; g.q_start = g_q_start; g.s_start = g_s_start; g.q_stop = q_length; g.s_stop = s_length;
                    g.score = left_score; g.seed_q = seed_q; g.seed_s = seed_s; g.context = ctx;
                    P.out[P.first + i] = g;
                    need_new = true; mode = START;
                }
            } else {
                GbnDevGapped g; g.q_start = g_q_start; g.s_start = g_s_start; g.q_stop = b_off + q_length; g.s_stop = a_off + s_length;
                g.score = left_score + best_score; g.seed_q = seed_q; g.seed_s = seed_s; g.context = ctx;
                P.out[P.first + i] = g;
                need_new = true; mode = START;
            }
        }
    }
#if GBN_LANE_DUP
    if (dup_sink == 0x9e3779b9u) redo_list[0] = (int32_t)dup_sink;     // (keeps the shadow loop's results alive)
#endif
}

namespace gbn {
hipError_t launch_gapped(const GbnGapParams &p, bool greedy, hipStream_t st, GbnKernelTimer *kt)
{
    auto mark = [&](int t) { if (kt) kt->mark(t, st); };
    if (p.n <= 0) return hipSuccess;
    const int64_t need = (p.n + 63) / 64;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, p.max_blocks > 0 ? std::min<int64_t>(need, p.max_blocks) : need);
    if (greedy) {
        // linear gap costs (megablast's default 0 / 0): a workgroup of two waves per initial hit, the diagonals of a distance across
        // the lanes (greedy_wave_kernel); what needs more distance than its LDS window holds, and affine costs: a thread per hit
        GbnGapParams r = p;
        const bool wave = p.gap_open == 0 && p.gap_extend == 0 && gbn::switch_value("GBN_GREEDY_WAVE", 1) != 0;
        if (wave) {
            const unsigned wblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(p.n, p.max_blocks > 0 ? (int64_t)p.max_blocks * 8 : p.n));
            mark(GBN_KT_WAVE_DP);
            hipLaunchKernelGGL(greedy_wave_kernel, dim3(wblocks), dim3(128), 0, st, p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            r.redo_only = 1;
        }
        mark(GBN_KT_THREAD_GAP); hipLaunchKernelGGL(greedy_kernel, dim3(blocks), dim3(64), 0, st, r); mark(-1); return hipGetLastError();
    }
    // blastn, three kernels: one extension per LANE with the band in LDS (dynprog_lane_kernel: the many short
    // extensions of chance hits); what it leaves (GBN_GAP_REDO: a window wider than its LDS slots, a long run) one
    // extension per WAVE with the band in registers; what that leaves (a band wider than a wave, gap_extend 0) the
    // thread-per-extension kernel with its band in scratch memory.  GBN_GAP_LANE=0: start with the wave kernel.
    if (!p.redo_only) {
        const bool lane_on = gbn::switch_value("GBN_GAP_LANE", 1) != 0;
        // 16-bit band cells, "dead" = -32768.  A half of a lane extension has at most GBN_LANE_ROWS rows, so no live value
        // exceeds GBN_LANE_ROWS * reward, none that is kept lies more than xdrop below the best, and what is stored beside
        // it (a gap opened or extended from a kept cell) at most gap_open + gap_extend lower still: all of it, and the
        // distance from the best score to a dead cell, must fit 16 bits -- whatever the length of the queries
        // (a bound by the longest context sent every batch with a 15 kb query through the wave kernel).
        const int64_t live_top = (int64_t)GBN_LANE_ROWS * std::max(std::abs(p.reward), std::abs(p.penalty));
        const bool lane = lane_on && p.gap_extend > 0 && std::abs(p.reward) <= 127 && std::abs(p.penalty) <= 127 &&
                          live_top + (int64_t)std::max(p.xdrop, p.gap_open + p.gap_extend) + p.gap_open + p.gap_extend < 30000 &&
                          (int64_t)p.scratch_per_thread * 64 * blocks >= 2 * p.n + 16;
        GbnGapParams w = p;
        int32_t *redo_list = nullptr;
        if (lane) {
            // scratch: [0, 1] the work counter, [2, 3] the number of extensions left to the wave kernel, [16, 16 + n) the
            // context of every hit, [16 + n, 16 + 2 n) the list of those extensions (the third kernel reuses it all later)
            hipError_t e = hipMemsetAsync(p.scratch, 0, 16 * sizeof(int32_t), st);
            if (e != hipSuccess) return e;
            int32_t *ctx_of = p.scratch + 16;
            mark(GBN_KT_LANE_DP);
            hipLaunchKernelGGL(gap_context_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p, ctx_of);
            const int64_t lblocks = std::max<int64_t>(1, std::min<int64_t>(need, p.max_blocks > 0 ? std::max(1, p.max_blocks * 2 / 3) : need));   // 16 of its workgroups fit a CU (LDS)
            redo_list = p.scratch + 16 + p.n;
            hipLaunchKernelGGL(dynprog_lane_kernel, dim3((unsigned)lblocks), dim3(64), 0, st, p, reinterpret_cast<unsigned long long *>(p.scratch), ctx_of, redo_list);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            w.redo_only = 1;
        }
        // a workgroup of two waves per extension (its two halves)
        const int64_t wblocks = std::max<int64_t>(1, std::min<int64_t>(p.n, p.max_blocks > 0 ? (int64_t)p.max_blocks * 4 : p.n));
        mark(GBN_KT_WAVE_DP);
        hipLaunchKernelGGL(dynprog_wave_kernel, dim3((unsigned)(redo_list ? std::min<int64_t>(wblocks, 16384) : wblocks)), dim3(128), 0, st, w,
                           reinterpret_cast<const unsigned long long *>(p.scratch) + 1, (const int32_t *)redo_list);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    GbnGapParams r = p; r.redo_only = 1;
    mark(GBN_KT_THREAD_GAP);
    hipLaunchKernelGGL(dynprog_kernel, dim3(blocks), dim3(64), 0, st, r);
    mark(-1);
    return hipGetLastError();
}
}  // namespace gbn
