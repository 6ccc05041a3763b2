/* orc_align.c -- ORACLE (test infrastructure): score-only gapped aligners
 * restated from CORE/greedy_align.c and CORE/blast_gapalign.c. */
#include "orc_int.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GREEDY_MAX_COST 10000           /* COREI/greedy_align.h:50 */
#define GREEDY_MAX_COST_FRACTION 2      /* COREI/greedy_align.h:47 */
#define MININT (ORC_INT4_MIN / 2)
static const int32_t kInvalidOffset = -2;   /* CORE/greedy_align.c:134 */

typedef struct { int32_t start_q, start_s, match_length; } GSeed;

/* CORE/greedy_align.c:318-381 (s_FindFirstMismatch), compressed-subject
 * branches only.  seq2 is the packed subject; `rem` shifts forward reads. */
static int32_t first_mismatch(const uint8_t *seq1, const uint8_t *seq2, int32_t len1, int32_t len2,
                              int32_t i1, int32_t i2, int reverse, int rem)
{
    int32_t tmp = i1;
    if (reverse) {
        while (i1 < len1 && i2 < len2 &&
               seq1[len1 - 1 - i1] == ORC_BASE(seq2, len2 - 1 - i2)) { ++i1; ++i2; }
    } else {
        while (i1 < len1 && i2 < len2 &&
               seq1[i1] == ORC_BASE(seq2, i2 + rem)) { ++i1; ++i2; }
    }
    return i1 - tmp;
}

/* CORE/greedy_align.c:385-753 (BLAST_GreedyAlign), score-only path
 * (edit_block == NULL): two rows of last_seq2_off are reused alternately. */
static int32_t greedy_linear(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2,
                             int reverse, int32_t xdrop_threshold, int32_t match_cost,
                             int32_t mismatch_cost, int32_t *seq1_align_len,
                             int32_t *seq2_align_len, int rem, GSeed *seed)
{
    int32_t seq1_index, seq2_index, index, d, k;
    int32_t diag_lower, diag_upper, max_dist, diag_origin, best_dist = 0, best_diag = 0;
    int32_t *row[2], *rows, *max_score_base, *max_score;
    int32_t xdrop_offset, longest_match_run;
    int end1_reached, end2_reached;

    max_dist = ORC_MIN(GREEDY_MAX_COST, len2 / GREEDY_MAX_COST_FRACTION + 1);
    diag_origin = max_dist + 2;
    xdrop_offset = (xdrop_threshold + match_cost / 2) / (match_cost + mismatch_cost) + 1;

    index = first_mismatch(seq1, seq2, len1, len2, 0, 0, reverse, rem);
    *seq1_align_len = index; *seq2_align_len = index;
    seq1_index = index;
    seed->start_q = 0; seed->start_s = 0;
    seed->match_length = longest_match_run = index;
    if (index == len1 || index == len2) return 0;

    rows = (int32_t *)malloc((size_t)(2 * max_dist + 6) * 2 * sizeof(int32_t));
    row[0] = rows; row[1] = rows + 2 * max_dist + 6;
    max_score_base = (int32_t *)malloc((size_t)(max_dist + 2 + xdrop_offset) * sizeof(int32_t));
    max_score = max_score_base + xdrop_offset;
    for (index = 0; index < xdrop_offset; index++) max_score_base[index] = 0;

    row[0][diag_origin] = seq1_index;
    max_score[0] = seq1_index * match_cost;
    diag_lower = diag_origin - 1;
    diag_upper = diag_origin + 1;
    end1_reached = end2_reached = 0;

    for (d = 1; d <= max_dist; d++) {
        int32_t xdrop_score, curr_score, curr_extent = 0, curr_seq2_index = 0, curr_diag = 0;
        int32_t tmp_diag_lower = diag_lower, tmp_diag_upper = diag_upper;
        int32_t *prev = row[(d - 1) & 1], *cur = row[d & 1];

        prev[diag_lower - 1] = kInvalidOffset;
        prev[diag_lower] = kInvalidOffset;
        prev[diag_upper] = kInvalidOffset;
        prev[diag_upper + 1] = kInvalidOffset;

        xdrop_score = max_score[d - xdrop_offset] + (match_cost + mismatch_cost) * d - xdrop_threshold;
        xdrop_score = (int32_t)ceil((double)xdrop_score / (match_cost / 2));

        for (k = tmp_diag_lower; k <= tmp_diag_upper; k++) {
            seq2_index = ORC_MAX(prev[k + 1], prev[k]) + 1;
            seq2_index = ORC_MAX(seq2_index, prev[k - 1]);
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xdrop_score) {
                if (k == diag_lower) diag_lower++;
                else cur[k] = kInvalidOffset;
                continue;
            }
            diag_upper = k;
            index = first_mismatch(seq1, seq2, len1, len2, seq1_index, seq2_index, reverse, rem);
            if (index > longest_match_run) {
                seed->start_q = seq1_index; seed->start_s = seq2_index;
                seed->match_length = longest_match_run = index;
            }
            seq1_index += index; seq2_index += index;
            cur[k] = seq2_index;
            if (seq1_index + seq2_index > curr_extent) {
                curr_extent = seq1_index + seq2_index;
                curr_seq2_index = seq2_index;
                curr_diag = k;
            }
            if (seq2_index == len2) { diag_lower = k + 1; end2_reached = 1; }
            if (seq1_index == len1) { diag_upper = k - 1; end1_reached = 1; }
        }
        curr_score = curr_extent * (match_cost / 2) - d * (match_cost + mismatch_cost);
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score;
            best_dist = d; best_diag = curr_diag;
            *seq2_align_len = curr_seq2_index;
            *seq1_align_len = curr_seq2_index + best_diag - diag_origin;
        } else {
            max_score[d] = max_score[d - 1];
        }
        if (diag_lower > diag_upper) break;
        if (!end2_reached) diag_lower--;
        if (!end1_reached) diag_upper++;
    }
    free(rows); free(max_score_base);
    return best_dist;
}

/* CORE/greedy_align.c:755-1236 (BLAST_AffineGreedyAlign), score-only path:
 * three furthest-reaching offsets (match / insert / delete) per diagonal and
 * distance, rows reused with period max_penalty + 1. */
typedef struct { int32_t insert_off, match_off, delete_off; } GOff;

static int32_t greedy_affine(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2,
                             int reverse, int32_t xdrop_threshold, int32_t match_score,
                             int32_t mismatch_score, int32_t in_gap_open, int32_t in_gap_extend,
                             int32_t *seq1_align_len, int32_t *seq2_align_len, int rem, GSeed *seed)
{
    int32_t seq1_index, seq2_index, index, d, k;
    int32_t max_dist, scaled_max_dist, diag_origin, best_dist = 0, best_diag = 0;
    int32_t longest_match_run, xdrop_offset, end1_diag, end2_diag;
    int32_t op_cost, gap_open, gap_extend, gap_open_extend, max_penalty, score_common_factor;
    int32_t match_score_half, curr_diag_lower, curr_diag_upper, num_nonempty_dist, result;
    int32_t *bounds, *diag_lower, *diag_upper, *max_score_base, *max_score;
    GOff *rows; int32_t row_len, nrows;
    const int32_t kInvalidDiag = 100000000;
#define ROW(dd) (rows + (size_t)((dd) % nrows) * row_len)

    match_score_half = match_score / 2;
    op_cost = match_score + mismatch_score;
    gap_open = in_gap_open;
    gap_extend = in_gap_extend + match_score_half;
    score_common_factor = orc_gdb3(&op_cost, &gap_open, &gap_extend);
    gap_open_extend = gap_open + gap_extend;
    max_penalty = ORC_MAX(op_cost, gap_open_extend);
    max_dist = ORC_MIN(GREEDY_MAX_COST, len2 / GREEDY_MAX_COST_FRACTION + 1);
    scaled_max_dist = max_dist * gap_extend;
    diag_origin = max_dist + 2;
    xdrop_offset = (xdrop_threshold + match_score_half) / score_common_factor + 1;

    index = first_mismatch(seq1, seq2, len1, len2, 0, 0, reverse, rem);
    *seq1_align_len = index; *seq2_align_len = index;
    seq1_index = index;
    seed->start_q = 0; seed->start_s = 0;
    seed->match_length = longest_match_run = index;
    if (index == len1 || index == len2) return index * match_score;

    nrows = max_penalty + 1; row_len = 2 * max_dist + 6;
    rows = (GOff *)malloc((size_t)nrows * row_len * sizeof(GOff));
    bounds = (int32_t *)malloc((size_t)2 * (scaled_max_dist + 1 + max_penalty) * sizeof(int32_t));
    max_score_base = (int32_t *)malloc((size_t)(scaled_max_dist + 2 + xdrop_offset) * sizeof(int32_t));
    max_score = max_score_base + xdrop_offset;
    for (index = 0; index < xdrop_offset; index++) max_score_base[index] = 0;
    diag_lower = bounds; diag_upper = bounds + scaled_max_dist + 1 + max_penalty;
    for (index = 0; index < max_penalty; index++) { diag_lower[index] = kInvalidDiag; diag_upper[index] = -kInvalidDiag; }
    diag_lower += max_penalty; diag_upper += max_penalty;

    ROW(0)[diag_origin].match_off = seq1_index;
    ROW(0)[diag_origin].insert_off = kInvalidOffset;
    ROW(0)[diag_origin].delete_off = kInvalidOffset;
    max_score[0] = seq1_index * match_score;
    diag_lower[0] = diag_origin; diag_upper[0] = diag_origin;
    curr_diag_lower = diag_origin - 1; curr_diag_upper = diag_origin + 1;
    end1_diag = 0; end2_diag = 0; num_nonempty_dist = 1; d = 1;

    while (d <= scaled_max_dist) {
        int32_t xdrop_score, curr_score, curr_extent = 0, curr_seq2_index = 0, curr_diag = 0;
        int32_t tmp_diag_lower = curr_diag_lower, tmp_diag_upper = curr_diag_upper;
        GOff *cur = ROW(d);
        xdrop_score = max_score[d - xdrop_offset] + score_common_factor * d - xdrop_threshold;
        xdrop_score = (int32_t)ceil((double)xdrop_score / match_score_half);
        if (xdrop_score < 0) xdrop_score = 0;
        for (k = tmp_diag_lower; k <= tmp_diag_upper; k++) {
            seq2_index = kInvalidOffset;
            if (k + 1 <= diag_upper[d - gap_open_extend] && k + 1 >= diag_lower[d - gap_open_extend])
                seq2_index = ROW(d - gap_open_extend)[k + 1].match_off;
            if (k + 1 <= diag_upper[d - gap_extend] && k + 1 >= diag_lower[d - gap_extend] &&
                seq2_index < ROW(d - gap_extend)[k + 1].delete_off)
                seq2_index = ROW(d - gap_extend)[k + 1].delete_off;
            if (seq2_index == kInvalidOffset) cur[k].delete_off = kInvalidOffset;
            else cur[k].delete_off = seq2_index + 1;

            seq2_index = kInvalidOffset;
            if (k - 1 <= diag_upper[d - gap_open_extend] && k - 1 >= diag_lower[d - gap_open_extend])
                seq2_index = ROW(d - gap_open_extend)[k - 1].match_off;
            if (k - 1 <= diag_upper[d - gap_extend] && k - 1 >= diag_lower[d - gap_extend] &&
                seq2_index < ROW(d - gap_extend)[k - 1].insert_off)
                seq2_index = ROW(d - gap_extend)[k - 1].insert_off;
            cur[k].insert_off = seq2_index;

            seq2_index = ORC_MAX(cur[k].insert_off, cur[k].delete_off);
            if (k <= diag_upper[d - op_cost] && k >= diag_lower[d - op_cost])
                seq2_index = ORC_MAX(seq2_index, ROW(d - op_cost)[k].match_off + 1);
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xdrop_score) {
                if (k == curr_diag_lower) curr_diag_lower++;
                else cur[k].match_off = kInvalidOffset;
                continue;
            }
            curr_diag_upper = k;
            index = first_mismatch(seq1, seq2, len1, len2, seq1_index, seq2_index, reverse, rem);
            if (index > longest_match_run) {
                seed->start_q = seq1_index; seed->start_s = seq2_index;
                seed->match_length = longest_match_run = index;
            }
            seq1_index += index; seq2_index += index;
            cur[k].match_off = seq2_index;
            if (seq1_index + seq2_index > curr_extent) {
                curr_extent = seq1_index + seq2_index; curr_seq2_index = seq2_index; curr_diag = k;
            }
            if (seq1_index == len1) { curr_diag_upper = k; end1_diag = k - 1; }
            if (seq2_index == len2) { curr_diag_lower = k; end2_diag = k + 1; }
        }
        curr_score = curr_extent * match_score_half - d * score_common_factor;
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score; best_dist = d; best_diag = curr_diag;
            *seq2_align_len = curr_seq2_index;
            *seq1_align_len = curr_seq2_index + best_diag - diag_origin;
        } else max_score[d] = max_score[d - 1];
        if (curr_diag_lower <= curr_diag_upper) {
            num_nonempty_dist++; diag_lower[d] = curr_diag_lower; diag_upper[d] = curr_diag_upper;
        } else { diag_lower[d] = kInvalidDiag; diag_upper[d] = -kInvalidDiag; }
        if (diag_lower[d - max_penalty] <= diag_upper[d - max_penalty]) num_nonempty_dist--;
        if (num_nonempty_dist == 0) break;
        d++;
        curr_diag_lower = ORC_MIN(diag_lower[d - gap_open_extend], diag_lower[d - gap_extend]) - 1;
        curr_diag_lower = ORC_MIN(curr_diag_lower, diag_lower[d - op_cost]);
        if (end2_diag > 0) curr_diag_lower = ORC_MAX(curr_diag_lower, end2_diag);
        curr_diag_upper = ORC_MAX(diag_upper[d - gap_open_extend], diag_upper[d - gap_extend]) + 1;
        curr_diag_upper = ORC_MAX(curr_diag_upper, diag_upper[d - op_cost]);
        if (end1_diag > 0) curr_diag_upper = ORC_MIN(curr_diag_upper, end1_diag);
    }
    result = max_score[best_dist];
    free(rows); free(bounds); free(max_score_base);
#undef ROW
    return result;
}

/* CORE/greedy_align.c:795-815: doubling of odd match scores and dispatch */
static int32_t greedy_dispatch(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2,
                               int reverse, int32_t xdrop, int32_t match_score, int32_t mismatch_score,
                               int32_t gap_open, int32_t gap_extend, int32_t *l1, int32_t *l2,
                               int rem, GSeed *seed, int *unsupported)
{
    (void)unsupported;
    if (match_score % 2 == 1) {
        match_score *= 2; mismatch_score *= 2; xdrop *= 2; gap_open *= 2; gap_extend *= 2;
    }
    if (gap_open == 0 && gap_extend == 0)
        return greedy_linear(seq1, len1, seq2, len2, reverse, xdrop, match_score,
                             mismatch_score, l1, l2, rem, seed);
    return greedy_affine(seq1, len1, seq2, len2, reverse, xdrop, match_score, mismatch_score,
                         gap_open, gap_extend, l1, l2, rem, seed);
}

/* CORE/blast_gapalign.c:2619-2751 (BLAST_GreedyGappedAlignment), compressed
 * subject, no traceback */
int orc_greedy_gapped(const uint8_t *query, const uint8_t *subj, int32_t qlen, int32_t slen,
                      int32_t q_off, int32_t s_off, int32_t X, int32_t reward, int32_t penalty,
                      int32_t gap_open, int32_t gap_extend, OrcGapResult *r)
{
    int32_t score, q_ext_l, q_ext_r, s_ext_l, s_ext_r;
    int32_t q_avail = qlen - q_off, s_avail = slen - s_off;
    GSeed fwd, rev; int unsupported = 0;
    int32_t q_seed_start = q_off, s_seed_start = s_off;

    /* right: seq2 = subject + s_off/4 with rem = s_off % 4 */
    score = greedy_dispatch(query + q_off, q_avail, subj + s_off / 4, s_avail, 0, X,
                            reward, -penalty, gap_open, gap_extend, &q_ext_r, &s_ext_r,
                            s_off % 4, &fwd, &unsupported);
    /* left: whole prefixes, rem = 0 */
    score += greedy_dispatch(query, q_off, subj, s_off, 1, X, reward, -penalty,
                             gap_open, gap_extend, &q_ext_l, &s_ext_l, 0, &rev, &unsupported);
    if (unsupported) return -1;
    if (gap_open == 0 && gap_extend == 0)
        score = (q_ext_r + s_ext_r + q_ext_l + s_ext_l) * reward / 2 - score * (reward - penalty);
    else if (reward % 2 == 1)
        score /= 2;
    {
        int32_t q_box_l = q_off - q_ext_l, s_box_l = s_off - s_ext_l;
        int32_t q_box_r = q_off + q_ext_r, s_box_r = s_off + s_ext_r;
        int32_t q_seed_start_l = q_off - rev.start_q, s_seed_start_l = s_off - rev.start_s;
        int32_t q_seed_start_r = q_off + fwd.start_q, s_seed_start_r = s_off + fwd.start_s;
        int32_t valid_l = 0, valid_r = 0;
        if (q_seed_start_r < q_box_r && s_seed_start_r < s_box_r) {
            valid_r = ORC_MIN(q_box_r - q_seed_start_r, s_box_r - s_seed_start_r);
            valid_r = ORC_MIN(valid_r, fwd.match_length) / 2;
        } else { q_seed_start_r = q_off; s_seed_start_r = s_off; }
        if (q_seed_start_l > q_box_l && s_seed_start_l > s_box_l) {
            valid_l = ORC_MIN(q_seed_start_l - q_box_l, s_seed_start_l - s_box_l);
            valid_l = ORC_MIN(valid_l, rev.match_length) / 2;
        } else { q_seed_start_l = q_off; s_seed_start_l = s_off; }
        if (valid_r > valid_l) {
            q_seed_start = q_seed_start_r + valid_r; s_seed_start = s_seed_start_r + valid_r;
        } else {
            q_seed_start = q_seed_start_l - valid_l; s_seed_start = s_seed_start_l - valid_l;
        }
    }
    r->q_start = q_off - q_ext_l; r->s_start = s_off - s_ext_l;
    r->q_stop = q_off + q_ext_r;  r->s_stop = s_off + s_ext_r;
    r->seed_q = q_seed_start; r->seed_s = s_seed_start;
    r->score = score;
    return 0;
}

int orc_greedy_extend(const uint8_t *query, int32_t qlen, const uint8_t *subj_packed, int32_t slen,
                      int32_t q_off, int32_t s_off, int32_t xdrop, int32_t reward, int32_t penalty,
                      int32_t gap_open, int32_t gap_extend, OrcHSP *out)
{
    OrcGapResult r;
    int rc = orc_greedy_gapped(query, subj_packed, qlen, slen, q_off, s_off, xdrop, reward,
                               penalty, gap_open, gap_extend, &r);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    out->q_offset = r.q_start; out->q_end = r.q_stop; out->s_offset = r.s_start; out->s_end = r.s_stop;
    out->q_gapped_start = r.seed_q; out->s_gapped_start = r.seed_s; out->score = r.score;
    return 0;
}

/* CORE/blast_gapalign.c:2842-3056 (s_BlastAlignPackedNucl).  Letters are
 * addressed through position formulas instead of the reference's pre-offset
 * pointers: forward pass reads query[q0 + b] / subject[s0 + a - 1], reverse
 * pass reads query[N-1-b] / subject[M-a]. */
typedef struct { int32_t best, best_gap; } GapDP;

static int32_t align_packed(const int32_t matrix[16][16], const uint8_t *query, const uint8_t *subj,
                            int32_t q0, int32_t s0, int32_t N, int32_t M,
                            int32_t *b_offset, int32_t *a_offset, int32_t x_dropoff,
                            int32_t gap_open, int32_t gap_extend, int reverse)
{
    int32_t i, a_index, b_index, b_size, first_b_index, last_b_index;
    GapDP *score_array; int32_t alloc, num_extra_cells;
    int32_t gap_open_extend = gap_open + gap_extend;
    int32_t score, score_gap_row, score_gap_col, next_score, best_score;

    *a_offset = 0; *b_offset = 0;
    if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
    if (N <= 0 || M <= 0) return 0;
    if (gap_extend > 0) num_extra_cells = x_dropoff / gap_extend + 3;
    else num_extra_cells = N + 3;
    alloc = num_extra_cells + 100;
    score_array = (GapDP *)malloc((size_t)alloc * sizeof(GapDP));
    score = -gap_open_extend;
    score_array[0].best = 0;
    score_array[0].best_gap = -gap_open_extend;
    for (i = 1; i <= N; i++) {
        if (score < -x_dropoff) break;
        score_array[i].best = score;
        score_array[i].best_gap = score - gap_open_extend;
        score -= gap_extend;
    }
    b_size = i; best_score = 0; first_b_index = 0;
    for (a_index = 1; a_index <= M; a_index++) {
        int a_base = reverse ? ORC_BASE(subj, M - a_index) : ORC_BASE(subj, s0 + a_index - 1);
        const int32_t *matrix_row = matrix[a_base];
        score = MININT; score_gap_row = MININT; last_b_index = first_b_index;
        for (b_index = first_b_index; b_index < b_size; b_index++) {
            uint8_t b_letter = reverse ? query[N - 1 - b_index] : query[q0 + b_index];
            score_gap_col = score_array[b_index].best_gap;
            next_score = score_array[b_index].best + matrix_row[b_letter];
            if (score < score_gap_col) score = score_gap_col;
            if (score < score_gap_row) score = score_gap_row;
            if (best_score - score > x_dropoff) {
                if (b_index == first_b_index) first_b_index++;
                else score_array[b_index].best = MININT;
            } else {
                last_b_index = b_index;
                if (score > best_score) {
                    best_score = score; *a_offset = a_index; *b_offset = b_index;
                }
                score_gap_row -= gap_extend;
                score_gap_col -= gap_extend;
                score_array[b_index].best_gap = ORC_MAX(score - gap_open_extend, score_gap_col);
                score_gap_row = ORC_MAX(score - gap_open_extend, score_gap_row);
                score_array[b_index].best = score;
            }
            score = next_score;
        }
        if (first_b_index == b_size) break;
        if (last_b_index + num_extra_cells + 3 >= alloc) {
            alloc = ORC_MAX(last_b_index + num_extra_cells + 100, 2 * alloc);
            score_array = (GapDP *)realloc(score_array, (size_t)alloc * sizeof(GapDP));
        }
        if (last_b_index < b_size - 1) {
            b_size = last_b_index + 1;
        } else {
            while (score_gap_row >= (best_score - x_dropoff) && b_size <= N) {
                score_array[b_size].best = score_gap_row;
                score_array[b_size].best_gap = score_gap_row - gap_open_extend;
                score_gap_row -= gap_extend;
                b_size++;
            }
        }
        if (b_size <= N) {
            score_array[b_size].best = MININT;
            score_array[b_size].best_gap = MININT;
            b_size++;
        }
    }
    free(score_array);
    return best_score;
}

/* CORE/blast_gapalign.c:2762-2826 (s_BlastDynProgNtGappedAlignment) */
int orc_dynprog_gapped(const int32_t matrix[16][16], const uint8_t *query, const uint8_t *subj,
                       int32_t qlen, int32_t slen, int32_t q_off, int32_t s_off, int32_t X,
                       int32_t gap_open, int32_t gap_extend, OrcGapResult *r)
{
    int32_t q_length, s_length, pq, ps, score_right = 0, score_left = 0;
    int32_t adj = 4 - (s_off % 4);
    q_length = q_off + adj; s_length = s_off + adj;
    if (q_length > qlen || s_length > slen) { q_length -= 4; s_length -= 4; }
    score_left = align_packed(matrix, query, subj, 0, 0, q_length, s_length, &pq, &ps,
                              X, gap_open, gap_extend, 1);
    if (score_left < 0) return -1;
    r->q_start = q_length - pq; r->s_start = s_length - ps;
    if (q_length < qlen && s_length < slen) {
        score_right = align_packed(matrix, query, subj, q_length, s_length,
                                   qlen - q_length, slen - s_length, &pq, &ps,
                                   X, gap_open, gap_extend, 0);
        if (score_right < 0) return -1;
        r->q_stop = pq + q_length; r->s_stop = ps + s_length;
    } else {
        r->q_stop = q_length; r->s_stop = s_length;
    }
    r->score = score_right + score_left;
    r->seed_q = q_off; r->seed_s = s_off;
    return 0;
}

/* the packed-subject X-drop DP on its own (definition-level tests): reward / penalty matrix over BLASTNA codes */
int orc_dynprog_extend(const uint8_t *query, int32_t qlen, const uint8_t *subj_packed, int32_t slen,
                       int32_t q_off, int32_t s_off, int32_t xdrop, int32_t reward, int32_t penalty,
                       int32_t gap_open, int32_t gap_extend, OrcHSP *out)
{
    int32_t matrix[16][16]; OrcGapResult r; int rc;
    orc_nucl_matrix(reward, penalty, matrix);
    rc = orc_dynprog_gapped((const int32_t (*)[16])matrix, query, subj_packed, qlen, slen, q_off, s_off, xdrop, gap_open, gap_extend, &r);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    out->q_offset = r.q_start; out->q_end = r.q_stop; out->s_offset = r.s_start; out->s_end = r.s_stop;
    out->q_gapped_start = r.seed_q; out->s_gapped_start = r.seed_s; out->score = r.score;
    return 0;
}
