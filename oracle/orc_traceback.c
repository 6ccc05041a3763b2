/* orc_traceback.c -- ORACLE (test infrastructure): the traceback stage of a nucleotide search for the
 * HSPs of one (query, subject) pair, restated from
 *   CORE/blast_traceback.c:336-790  (Blast_TracebackFromHSPList), :278-334 (s_HSPListPostTracebackUpdate)
 *   CORE/blast_gapalign.c:350-708   (ALIGN_EX), :710-937 (Blast_SemiGappedAlign, score only),
 *                        :3994-4155 (BLAST_GappedAlignmentWithTraceback), :2456-2516 (edit block -> script),
 *                        :2547-2617 (s_ReduceGaps), :2619-2751 (BLAST_GreedyGappedAlignment, traceback branch),
 *                        :3059-3183 (start points), :3608-3637 (AdjustSubjectRange)
 *   CORE/greedy_align.c:385-753     (BLAST_GreedyAlign with an edit block)
 *   CORE/blast_hits.c:311-520 (re-evaluation along the edit script), :618-700 (identities),
 *                    :2163-2302 (common end points, purge = FALSE), CORE/gapinfo.c:163-190.
 * Affine greedy traceback (CORE/greedy_align.c:1170-1233) is not restated: megablast's default is the
 * linear aligner, blastn's the dynamic-programming one. */
#include "orc_int.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MININT (ORC_INT4_MIN / 2)
enum { OP_DEL = 0, OP_SUB = 3, OP_INS = 6, OP_INVALID = 8 };    /* COREI/gapinfo.h EGapAlignOpType */
#define HSP_MAX_WINDOW 11                                       /* COREI/blast_def.h */

/* ---------------- edit scripts (CORE/gapinfo.c) ---------------- */
typedef struct { uint8_t *op; int32_t *num; int32_t n, alloc, last_op; } Prelim;
static void prelim_reset(Prelim *p) { p->n = 0; p->last_op = OP_INVALID; }
static void prelim_add(Prelim *p, int op, int32_t num)          /* :180-190 */
{
    if (num == 0) return;
    if (p->last_op == op) { p->num[p->n - 1] += num; return; }
    if (p->n + 2 > p->alloc) {
        p->alloc = ORC_MAX(100, 2 * p->alloc + 2);
        p->op = (uint8_t *)realloc(p->op, (size_t)p->alloc);
        p->num = (int32_t *)realloc(p->num, (size_t)p->alloc * sizeof(int32_t));
    }
    p->last_op = op; p->op[p->n] = (uint8_t)op; p->num[p->n] = num; p->n++;
}
static void esp_free(OrcEditScript *e) { free(e->op); free(e->num); e->op = NULL; e->num = NULL; e->size = 0; }
/* CORE/blast_gapalign.c:2456-2516: the left script as produced, the right one reversed, joined */
static void prelim_to_esp(const Prelim *rev, const Prelim *fwd, OrcEditScript *e)
{
    int merge = 0; int32_t size, i, index = 0;
    if (fwd->n > 0 && rev->n > 0 && fwd->op[fwd->n - 1] == rev->op[rev->n - 1]) merge = 1;
    size = fwd->n + rev->n - (merge ? 1 : 0);
    e->op = (uint8_t *)malloc((size_t)ORC_MAX(size, 1)); e->num = (int32_t *)malloc((size_t)ORC_MAX(size, 1) * sizeof(int32_t));
    e->size = size;
    for (i = 0; i < rev->n; i++) { e->op[index] = rev->op[i]; e->num[index] = rev->num[i]; index++; }
    if (fwd->n == 0) return;
    if (merge) e->num[index - 1] += fwd->num[fwd->n - 1];
    for (i = merge ? fwd->n - 2 : fwd->n - 1; i >= 0; i--) { e->op[index] = fwd->op[i]; e->num[index] = fwd->num[i]; index++; }
}

/* ---------------- ALIGN_EX (CORE/blast_gapalign.c:350-708) ---------------- */
enum { SCRIPT_SUB = OP_SUB, SCRIPT_GAP_IN_A = OP_DEL, SCRIPT_GAP_IN_B = OP_INS, SCRIPT_OP_MASK = 0x07,
       SCRIPT_EXTEND_GAP_A = 0x10, SCRIPT_EXTEND_GAP_B = 0x40 };
typedef struct { int32_t best, best_gap; } GapDP;

/* A: query (rows), B: subject (columns); letters as the reference addresses them:
 * forward A[a], B[b + 1] (b from 0); reverse A[M - a], B[N - 1 - b] */
static int32_t align_ex(const int32_t matrix[16][16], const uint8_t *A, const uint8_t *B, int32_t M, int32_t N,
                        int32_t *a_offset, int32_t *b_offset, Prelim *edit_block, int32_t x_dropoff,
                        int32_t gap_open, int32_t gap_extend, int reverse_sequence)
{
    int32_t i, a_index, b_index, b_size, first_b_index, last_b_index;
    GapDP *score_array; int32_t dp_alloc;
    int32_t gap_open_extend = gap_open + gap_extend, best_score, score, score_gap_row, score_gap_col, next_score;
    uint8_t **edit_script, *edit_script_row; int32_t *edit_start_offset, edit_script_num_rows, orig_b_index;
    uint8_t script, next_script, script_row, script_col; int32_t num_extra_cells;

    *a_offset = 0; *b_offset = 0;
    if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
    if (N <= 0 || M <= 0) return 0;
    edit_script_num_rows = 100;
    edit_script = (uint8_t **)calloc((size_t)edit_script_num_rows, sizeof(uint8_t *));
    edit_start_offset = (int32_t *)malloc((size_t)edit_script_num_rows * sizeof(int32_t));
    if (gap_extend > 0) num_extra_cells = x_dropoff / gap_extend + 3; else num_extra_cells = N + 3;
    dp_alloc = num_extra_cells + 100;
    score_array = (GapDP *)malloc((size_t)dp_alloc * sizeof(GapDP));
    edit_script[0] = (uint8_t *)malloc((size_t)num_extra_cells + 4);
    edit_start_offset[0] = 0;
    edit_script_row = edit_script[0];
    score = -gap_open_extend;
    score_array[0].best = 0; score_array[0].best_gap = -gap_open_extend;
    for (i = 1; i <= N; i++) {
        if (score < -x_dropoff) break;
        score_array[i].best = score; score_array[i].best_gap = score - gap_open_extend;
        score -= gap_extend;
        edit_script_row[i] = SCRIPT_GAP_IN_A;
    }
    b_size = i; best_score = 0; first_b_index = 0;
    for (a_index = 1; a_index <= M; a_index++) {
        const int32_t *matrix_row; int32_t row_len;
        if (gap_extend > 0) row_len = b_size - first_b_index + num_extra_cells; else row_len = N + 3 - first_b_index;
        if (a_index == edit_script_num_rows) {
            edit_script_num_rows *= 2;
            edit_script = (uint8_t **)realloc(edit_script, (size_t)edit_script_num_rows * sizeof(uint8_t *));
            memset(edit_script + a_index, 0, (size_t)(edit_script_num_rows - a_index) * sizeof(uint8_t *));
            edit_start_offset = (int32_t *)realloc(edit_start_offset, (size_t)edit_script_num_rows * sizeof(int32_t));
        }
        edit_script[a_index] = (uint8_t *)malloc((size_t)row_len + 4);
        edit_start_offset[a_index] = first_b_index;
        edit_script_row = edit_script[a_index] - first_b_index;
        orig_b_index = first_b_index;
        matrix_row = reverse_sequence ? matrix[A[M - a_index]] : matrix[A[a_index]];
        score = MININT; score_gap_row = MININT; last_b_index = first_b_index;
        for (b_index = first_b_index; b_index < b_size; b_index++) {
            const uint8_t b_letter = reverse_sequence ? B[N - 1 - b_index] : B[b_index + 1];
            score_gap_col = score_array[b_index].best_gap;
            next_score = score_array[b_index].best + matrix_row[b_letter];
            script = SCRIPT_SUB; script_col = SCRIPT_EXTEND_GAP_B; script_row = SCRIPT_EXTEND_GAP_A;
            if (score < score_gap_col) { script = SCRIPT_GAP_IN_B; score = score_gap_col; }
            if (score < score_gap_row) { script = SCRIPT_GAP_IN_A; score = score_gap_row; }
            if (best_score - score > x_dropoff) {
                if (first_b_index == b_index) first_b_index++;
                else score_array[b_index].best = MININT;
            } else {
                last_b_index = b_index;
                if (score > best_score) { best_score = score; *a_offset = a_index; *b_offset = b_index; }
                score_gap_row -= gap_extend; score_gap_col -= gap_extend;
                if (score_gap_col < (score - gap_open_extend)) score_array[b_index].best_gap = score - gap_open_extend;
                else { score_array[b_index].best_gap = score_gap_col; script += script_col; }
                if (score_gap_row < (score - gap_open_extend)) score_gap_row = score - gap_open_extend;
                else script += script_row;
                score_array[b_index].best = score;
            }
            score = next_score;
            edit_script_row[b_index] = script;
        }
        if (first_b_index == b_size) break;
        if (last_b_index + num_extra_cells + 3 >= dp_alloc) {
            dp_alloc = ORC_MAX(last_b_index + num_extra_cells + 100, 2 * dp_alloc);
            score_array = (GapDP *)realloc(score_array, (size_t)dp_alloc * sizeof(GapDP));
        }
        if (last_b_index < b_size - 1) {
            b_size = last_b_index + 1;
        } else {
            while (score_gap_row >= (best_score - x_dropoff) && b_size <= N) {
                score_array[b_size].best = score_gap_row;
                score_array[b_size].best_gap = score_gap_row - gap_open_extend;
                score_gap_row -= gap_extend;
                edit_script_row[b_size] = SCRIPT_GAP_IN_A;
                b_size++;
            }
        }
        (void)orig_b_index;
        if (b_size <= N) { score_array[b_size].best = MININT; score_array[b_size].best_gap = MININT; b_size++; }
    }
    /* the optimal path through the stored actions */
    a_index = *a_offset; b_index = *b_offset; script = SCRIPT_SUB;
    while (a_index > 0 || b_index > 0) {
        next_script = edit_script[a_index][b_index - edit_start_offset[a_index]];
        switch (script) {
        case SCRIPT_GAP_IN_A:
            script = next_script & SCRIPT_OP_MASK;
            if (next_script & SCRIPT_EXTEND_GAP_A) script = SCRIPT_GAP_IN_A;
            break;
        case SCRIPT_GAP_IN_B:
            script = next_script & SCRIPT_OP_MASK;
            if (next_script & SCRIPT_EXTEND_GAP_B) script = SCRIPT_GAP_IN_B;
            break;
        default:
            script = next_script & SCRIPT_OP_MASK;
            break;
        }
        if (script == SCRIPT_GAP_IN_A) b_index--;
        else if (script == SCRIPT_GAP_IN_B) a_index--;
        else { a_index--; b_index--; }
        prelim_add(edit_block, script, 1);
    }
    for (i = 0; i < edit_script_num_rows; i++) free(edit_script[i]);
    free(edit_script); free(edit_start_offset); free(score_array);
    return best_score;
}

/* Blast_SemiGappedAlign, score only (CORE/blast_gapalign.c:745-937): the twin of ALIGN_EX without the actions */
int32_t orc_semi_gapped_score(const int32_t matrix[16][16], const uint8_t *A, const uint8_t *B, int32_t M, int32_t N,
                              int32_t *a_offset, int32_t *b_offset, int32_t x_dropoff, int32_t gap_open,
                              int32_t gap_extend, int reverse_sequence)
{
    int32_t i, a_index, b_index, b_size, first_b_index, last_b_index, num_extra_cells, dp_alloc;
    int32_t gap_open_extend = gap_open + gap_extend, best_score, score, score_gap_row, score_gap_col, next_score;
    GapDP *score_array;
    *a_offset = 0; *b_offset = 0;
    if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
    if (N <= 0 || M <= 0) return 0;
    if (gap_extend > 0) num_extra_cells = x_dropoff / gap_extend + 3; else num_extra_cells = N + 3;
    dp_alloc = num_extra_cells + 100;
    score_array = (GapDP *)malloc((size_t)dp_alloc * sizeof(GapDP));
    score = -gap_open_extend;
    score_array[0].best = 0; score_array[0].best_gap = -gap_open_extend;
    for (i = 1; i <= N; i++) {
        if (score < -x_dropoff) break;
        score_array[i].best = score; score_array[i].best_gap = score - gap_open_extend;
        score -= gap_extend;
    }
    b_size = i; best_score = 0; first_b_index = 0;
    for (a_index = 1; a_index <= M; a_index++) {
        const int32_t *matrix_row = reverse_sequence ? matrix[A[M - a_index]] : matrix[A[a_index]];
        score = MININT; score_gap_row = MININT; last_b_index = first_b_index;
        for (b_index = first_b_index; b_index < b_size; b_index++) {
            const uint8_t b_letter = reverse_sequence ? B[N - 1 - b_index] : B[b_index + 1];
            score_gap_col = score_array[b_index].best_gap;
            next_score = score_array[b_index].best + matrix_row[b_letter];
            if (score < score_gap_col) score = score_gap_col;
            if (score < score_gap_row) score = score_gap_row;
            if (best_score - score > x_dropoff) {
                if (first_b_index == b_index) first_b_index++;
                else score_array[b_index].best = MININT;
            } else {
                last_b_index = b_index;
                if (score > best_score) { best_score = score; *a_offset = a_index; *b_offset = b_index; }
                score_gap_row -= gap_extend; score_gap_col -= gap_extend;
                score_array[b_index].best_gap = ORC_MAX(score - gap_open_extend, score_gap_col);
                score_gap_row = ORC_MAX(score - gap_open_extend, score_gap_row);
                score_array[b_index].best = score;
            }
            score = next_score;
        }
        if (first_b_index == b_size) break;
        if (last_b_index + num_extra_cells + 3 >= dp_alloc) {
            dp_alloc = ORC_MAX(last_b_index + num_extra_cells + 100, 2 * dp_alloc);
            score_array = (GapDP *)realloc(score_array, (size_t)dp_alloc * sizeof(GapDP));
        }
        if (last_b_index < b_size - 1) b_size = last_b_index + 1;
        else {
            while (score_gap_row >= (best_score - x_dropoff) && b_size <= N) {
                score_array[b_size].best = score_gap_row;
                score_array[b_size].best_gap = score_gap_row - gap_open_extend;
                score_gap_row -= gap_extend;
                b_size++;
            }
        }
        if (b_size <= N) { score_array[b_size].best = MININT; score_array[b_size].best_gap = MININT; b_size++; }
    }
    free(score_array);
    return best_score;
}

/* BLAST_GappedAlignmentWithTraceback (CORE/blast_gapalign.c:3994-4155), in-frame */
static int orc_gapped_with_traceback(const int32_t matrix[16][16], const uint8_t *query, const uint8_t *subject,
                              int32_t q_length, int32_t s_length, int32_t q_start, int32_t s_start, int32_t x_dropoff,
                              int32_t gap_open, int32_t gap_extend, OrcGapResult *r, OrcEditScript *esp)
{
    Prelim fwd = {0, 0, 0, 0, OP_INVALID}, rev = {0, 0, 0, 0, OP_INVALID};
    int32_t score_left, score_right = 0, pq = 0, ps = 0; int found_end = 0;
    prelim_reset(&fwd); prelim_reset(&rev);
    /* the left extension includes the starting point, the right one does not */
    score_left = align_ex(matrix, query, subject, q_start + 1, s_start + 1, &pq, &ps, &rev, x_dropoff, gap_open, gap_extend, 1);
    r->q_start = q_start - pq + 1; r->s_start = s_start - ps + 1;
    if (q_start < q_length && s_start < s_length) {
        found_end = 1;
        score_right = align_ex(matrix, query + q_start, subject + s_start, q_length - q_start - 1, s_length - s_start - 1,
                               &pq, &ps, &fwd, x_dropoff, gap_open, gap_extend, 0);
        r->q_stop = q_start + pq + 1; r->s_stop = s_start + ps + 1;
    }
    if (!found_end) { r->q_stop = q_start - 1; r->s_stop = s_start - 1; }
    prelim_to_esp(&rev, &fwd, esp);
    /* leading / trailing gaps are pruned (:4115-4151) */
    if (esp->size && esp->op[0] != OP_SUB) {
        int32_t i;
        score_left += gap_open + esp->num[0] * gap_extend;
        if (esp->op[0] == OP_DEL) r->s_start += esp->num[0]; else r->q_start += esp->num[0];
        for (i = 1; i < esp->size; i++) { esp->op[i - 1] = esp->op[i]; esp->num[i - 1] = esp->num[i]; }
        esp->size--;
    }
    if (esp->size && esp->op[esp->size - 1] != OP_SUB) {
        const int32_t i = esp->size;
        score_right += gap_open + esp->num[i - 1] * gap_extend;
        if (esp->op[i - 1] == OP_DEL) r->s_stop -= esp->num[i - 1]; else r->q_stop -= esp->num[i - 1];
        esp->size--;
    }
    r->score = score_right + score_left;
    r->seed_q = q_start; r->seed_s = s_start;
    free(fwd.op); free(fwd.num); free(rev.op); free(rev.num);
    return 0;
}

/* ---------------- greedy with traceback (CORE/greedy_align.c:385-753, uncompressed subject) ---------------- */
static const int32_t kInvalidOffset = -2;
static int32_t first_mismatch_u(const uint8_t *seq1, const uint8_t *seq2, int32_t len1, int32_t len2,
                                int32_t i1, int32_t i2, int reverse)                     /* :318-381, rem == 4 */
{
    const int32_t tmp = i1;
    if (reverse) {
        while (i1 < len1 && i2 < len2 && seq1[len1 - 1 - i1] < 4 && seq1[len1 - 1 - i1] == seq2[len2 - 1 - i2]) { ++i1; ++i2; }
    } else {
        while (i1 < len1 && i2 < len2 && seq1[i1] < 4 && seq1[i1] == seq2[i2]) { ++i1; ++i2; }
    }
    return i1 - tmp;
}
typedef struct { int32_t *base; int32_t lo; } Row;      /* row[k] = base[k - lo] */
#define RW(r, k) ((r).base[(k) - (r).lo])

static int32_t greedy_align_tb(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2, int reverse,
                               int32_t xdrop_threshold, int32_t match_cost, int32_t mismatch_cost,
                               int32_t *seq1_align_len, int32_t *seq2_align_len, Prelim *edit_block)
{
    int32_t seq1_index, seq2_index, index, d, k, diag_lower, diag_upper, max_dist, diag_origin, best_dist = 0, best_diag = 0;
    int32_t *max_score_base, *max_score, xdrop_offset, nrows;
    int end1_reached, end2_reached;
    Row *rows;
    max_dist = ORC_MIN(10000, len2 / 2 + 1);
    diag_origin = max_dist + 2;
    xdrop_offset = (xdrop_threshold + match_cost / 2) / (match_cost + mismatch_cost) + 1;
    index = first_mismatch_u(seq1, seq2, len1, len2, 0, 0, reverse);
    *seq1_align_len = index; *seq2_align_len = index;
    seq1_index = index;
    if (index == len1 || index == len2) { prelim_add(edit_block, OP_SUB, index); return 0; }
    rows = (Row *)calloc((size_t)max_dist + 3, sizeof(Row));
    /* the first two rows span every diagonal; later ones what their distance can reach (:677-683) */
    for (d = 0; d < 2; d++) { rows[d].base = (int32_t *)malloc((size_t)(2 * max_dist + 8) * sizeof(int32_t)); rows[d].lo = 0; }
    nrows = 2;
    max_score_base = (int32_t *)malloc((size_t)(max_dist + 2 + xdrop_offset) * sizeof(int32_t));
    max_score = max_score_base + xdrop_offset;
    for (index = 0; index < xdrop_offset; index++) max_score_base[index] = 0;
    RW(rows[0], diag_origin) = seq1_index;
    max_score[0] = seq1_index * match_cost;
    diag_lower = diag_origin - 1; diag_upper = diag_origin + 1;
    end1_reached = end2_reached = 0;
    for (d = 1; d <= max_dist; d++) {
        int32_t xdrop_score, curr_score, curr_extent = 0, curr_seq2_index = 0, curr_diag = 0;
        const int32_t tmp_diag_lower = diag_lower, tmp_diag_upper = diag_upper;
        RW(rows[d - 1], diag_lower - 1) = kInvalidOffset; RW(rows[d - 1], diag_lower) = kInvalidOffset;
        RW(rows[d - 1], diag_upper) = kInvalidOffset; RW(rows[d - 1], diag_upper + 1) = kInvalidOffset;
        xdrop_score = max_score[d - xdrop_offset] + (match_cost + mismatch_cost) * d - xdrop_threshold;
        xdrop_score = (int32_t)ceil((double)xdrop_score / (match_cost / 2));
        for (k = tmp_diag_lower; k <= tmp_diag_upper; k++) {
            seq2_index = ORC_MAX(RW(rows[d - 1], k + 1), RW(rows[d - 1], k)) + 1;
            seq2_index = ORC_MAX(seq2_index, RW(rows[d - 1], k - 1));
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xdrop_score) {
                if (k == diag_lower) diag_lower++; else RW(rows[d], k) = kInvalidOffset;
                continue;
            }
            diag_upper = k;
            index = first_mismatch_u(seq1, seq2, len1, len2, seq1_index, seq2_index, reverse);
            seq1_index += index; seq2_index += index;
            RW(rows[d], k) = seq2_index;
            if (seq1_index + seq2_index > curr_extent) { curr_extent = seq1_index + seq2_index; curr_seq2_index = seq2_index; curr_diag = k; }
            if (seq2_index == len2) { diag_lower = k + 1; end2_reached = 1; }
            if (seq1_index == len1) { diag_upper = k - 1; end1_reached = 1; }
        }
        curr_score = curr_extent * (match_cost / 2) - d * (match_cost + mismatch_cost);
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score; best_dist = d; best_diag = curr_diag;
            *seq2_align_len = curr_seq2_index; *seq1_align_len = curr_seq2_index + best_diag - diag_origin;
        } else max_score[d] = max_score[d - 1];
        if (diag_lower > diag_upper) break;
        if (!end2_reached) diag_lower--;
        if (!end1_reached) diag_upper++;
        /* traceback needs every row: a new one spanning the diagonals of the next distance, two spare either side */
        rows[d + 1].base = (int32_t *)malloc((size_t)(diag_upper - diag_lower + 7) * sizeof(int32_t));
        rows[d + 1].lo = diag_lower - 2;
        nrows = d + 2;
    }
    /* traceback (:689-748) */
    d = best_dist; seq1_index = *seq1_align_len; seq2_index = *seq2_align_len;
    while (d > 0) {
        int32_t new_diag, new_seq2_index;
        /* s_GetNextNonAffineTback (:289-305) */
        const int32_t lm = RW(rows[d - 1], best_diag - 1), mm = RW(rows[d - 1], best_diag), rm = RW(rows[d - 1], best_diag + 1);
        if (lm > ORC_MAX(mm, rm)) { new_seq2_index = lm; new_diag = best_diag - 1; }
        else if (mm > rm) { new_seq2_index = mm; new_diag = best_diag; }
        else { new_seq2_index = rm; new_diag = best_diag + 1; }
        if (new_diag == best_diag) {
            if (seq2_index - new_seq2_index > 0) prelim_add(edit_block, OP_SUB, seq2_index - new_seq2_index);
        } else if (new_diag < best_diag) {
            if (seq2_index - new_seq2_index > 0) prelim_add(edit_block, OP_SUB, seq2_index - new_seq2_index);
            prelim_add(edit_block, OP_INS, 1);
        } else {
            if (seq2_index - new_seq2_index - 1 > 0) prelim_add(edit_block, OP_SUB, seq2_index - new_seq2_index - 1);
            prelim_add(edit_block, OP_DEL, 1);
        }
        d--; best_diag = new_diag; seq2_index = new_seq2_index;
    }
    prelim_add(edit_block, OP_SUB, RW(rows[0], diag_origin));
    for (d = 0; d < nrows; d++) free(rows[d].base);
    free(rows); free(max_score_base);
    return best_dist;
}

/* ---------------- affine greedy with traceback (BLAST_AffineGreedyAlign, CORE/greedy_align.c:755-1236, uncompressed
 * subject): three furthest-reaching offsets per (distance, diagonal); with an edit block every row is kept
 * (:1166-1177) and the script is walked back through s_GetNextAffineTbackFromMatch / FromIndel (:153-262) ---------------- */
typedef struct { int32_t insert_off, match_off, delete_off; } AOff;
typedef struct { AOff *base; int32_t lo; } ARow;
#define AR(r, k) ((r).base[(k) - (r).lo])

static int32_t gcd3(int32_t *a, int32_t *b, int32_t *c);

static int32_t greedy_affine_tb(const uint8_t *seq1, int32_t len1, const uint8_t *seq2, int32_t len2, int reverse,
                                int32_t xdrop_threshold, int32_t match_score, int32_t mismatch_score,
                                int32_t in_gap_open, int32_t in_gap_extend,
                                int32_t *seq1_align_len, int32_t *seq2_align_len, Prelim *edit_block)
{
    int32_t seq1_index, seq2_index, index, d, k, max_dist, scaled_max_dist, diag_origin, best_dist = 0, best_diag = 0;
    int32_t xdrop_offset, end1_diag, end2_diag, op_cost, gap_open, gap_extend, gap_open_extend, max_penalty, score_common_factor;
    int32_t match_score_half, curr_diag_lower, curr_diag_upper, num_nonempty_dist, result, nrows = 0;
    int32_t *bounds, *diag_lower, *diag_upper, *max_score_base, *max_score;
    ARow *rows;
    const int32_t kInvalidDiag = 100000000;

    match_score_half = match_score / 2;
    op_cost = match_score + mismatch_score;
    gap_open = in_gap_open; gap_extend = in_gap_extend + match_score_half;
    score_common_factor = gcd3(&op_cost, &gap_open, &gap_extend);
    gap_open_extend = gap_open + gap_extend;
    max_penalty = ORC_MAX(op_cost, gap_open_extend);
    max_dist = ORC_MIN(10000, len2 / 2 + 1);
    scaled_max_dist = max_dist * gap_extend;
    diag_origin = max_dist + 2;
    xdrop_offset = (xdrop_threshold + match_score_half) / score_common_factor + 1;

    index = first_mismatch_u(seq1, seq2, len1, len2, 0, 0, reverse);
    *seq1_align_len = index; *seq2_align_len = index;
    seq1_index = index;
    if (index == len1 || index == len2) { prelim_add(edit_block, OP_SUB, index); return index * match_score; }

    rows = (ARow *)calloc((size_t)scaled_max_dist + 2, sizeof(ARow));
    bounds = (int32_t *)malloc((size_t)2 * (scaled_max_dist + 1 + max_penalty) * sizeof(int32_t));
    max_score_base = (int32_t *)malloc((size_t)(scaled_max_dist + 2 + xdrop_offset) * sizeof(int32_t));
    max_score = max_score_base + xdrop_offset;
    for (index = 0; index < xdrop_offset; index++) max_score_base[index] = 0;
    diag_lower = bounds; diag_upper = bounds + scaled_max_dist + 1 + max_penalty;
    for (index = 0; index < max_penalty; index++) { diag_lower[index] = kInvalidDiag; diag_upper[index] = -kInvalidDiag; }
    diag_lower += max_penalty; diag_upper += max_penalty;

    rows[0].base = (AOff *)malloc(sizeof(AOff)); rows[0].lo = diag_origin; nrows = 1;
    AR(rows[0], diag_origin).match_off = seq1_index;
    AR(rows[0], diag_origin).insert_off = kInvalidOffset;
    AR(rows[0], diag_origin).delete_off = kInvalidOffset;
    max_score[0] = seq1_index * match_score;
    diag_lower[0] = diag_origin; diag_upper[0] = diag_origin;
    curr_diag_lower = diag_origin - 1; curr_diag_upper = diag_origin + 1;
    end1_diag = 0; end2_diag = 0; num_nonempty_dist = 1; d = 1;

    while (d <= scaled_max_dist) {
        int32_t xdrop_score, curr_score, curr_extent = 0, curr_seq2_index = 0, curr_diag = 0;
        const int32_t tmp_diag_lower = curr_diag_lower, tmp_diag_upper = curr_diag_upper;
        /* the row of this distance: every diagonal the loop below visits (a row is read later only inside the
         * bounds saved for its distance, which lie inside these) */
        rows[d].lo = tmp_diag_lower;
        rows[d].base = (AOff *)malloc((size_t)ORC_MAX(tmp_diag_upper - tmp_diag_lower + 1, 1) * sizeof(AOff));
        nrows = d + 1;
        xdrop_score = max_score[d - xdrop_offset] + score_common_factor * d - xdrop_threshold;
        xdrop_score = (int32_t)ceil((double)xdrop_score / match_score_half);
        if (xdrop_score < 0) xdrop_score = 0;
        for (k = tmp_diag_lower; k <= tmp_diag_upper; k++) {
            AOff *cur = &AR(rows[d], k);
            seq2_index = kInvalidOffset;
            if (k + 1 <= diag_upper[d - gap_open_extend] && k + 1 >= diag_lower[d - gap_open_extend])
                seq2_index = AR(rows[d - gap_open_extend], k + 1).match_off;
            if (k + 1 <= diag_upper[d - gap_extend] && k + 1 >= diag_lower[d - gap_extend] &&
                seq2_index < AR(rows[d - gap_extend], k + 1).delete_off)
                seq2_index = AR(rows[d - gap_extend], k + 1).delete_off;
            cur->delete_off = seq2_index == kInvalidOffset ? kInvalidOffset : seq2_index + 1;
            seq2_index = kInvalidOffset;
            if (k - 1 <= diag_upper[d - gap_open_extend] && k - 1 >= diag_lower[d - gap_open_extend])
                seq2_index = AR(rows[d - gap_open_extend], k - 1).match_off;
            if (k - 1 <= diag_upper[d - gap_extend] && k - 1 >= diag_lower[d - gap_extend] &&
                seq2_index < AR(rows[d - gap_extend], k - 1).insert_off)
                seq2_index = AR(rows[d - gap_extend], k - 1).insert_off;
            cur->insert_off = seq2_index;
            seq2_index = ORC_MAX(cur->insert_off, cur->delete_off);
            if (k <= diag_upper[d - op_cost] && k >= diag_lower[d - op_cost])
                seq2_index = ORC_MAX(seq2_index, AR(rows[d - op_cost], k).match_off + 1);
            seq1_index = seq2_index + k - diag_origin;
            if (seq2_index < 0 || seq1_index + seq2_index < xdrop_score) {
                if (k == curr_diag_lower) curr_diag_lower++;
                else cur->match_off = kInvalidOffset;
                continue;
            }
            curr_diag_upper = k;
            index = first_mismatch_u(seq1, seq2, len1, len2, seq1_index, seq2_index, reverse);
            seq1_index += index; seq2_index += index;
            cur->match_off = seq2_index;
            if (seq1_index + seq2_index > curr_extent) { curr_extent = seq1_index + seq2_index; curr_seq2_index = seq2_index; curr_diag = k; }
            if (seq1_index == len1) { curr_diag_upper = k; end1_diag = k - 1; }
            if (seq2_index == len2) { curr_diag_lower = k; end2_diag = k + 1; }
        }
        curr_score = curr_extent * match_score_half - d * score_common_factor;
        if (curr_score > max_score[d - 1]) {
            max_score[d] = curr_score; best_dist = d; best_diag = curr_diag;
            *seq2_align_len = curr_seq2_index; *seq1_align_len = curr_seq2_index + best_diag - diag_origin;
        } else max_score[d] = max_score[d - 1];
        if (curr_diag_lower <= curr_diag_upper) { num_nonempty_dist++; diag_lower[d] = curr_diag_lower; diag_upper[d] = curr_diag_upper; }
        else { diag_lower[d] = kInvalidDiag; diag_upper[d] = -kInvalidDiag; }
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
    {   /* the traceback (:1187-1233) */
        int state = OP_SUB;
        d = best_dist; seq2_index = *seq2_align_len;
        while (d > 0) {
            if (state == OP_SUB) {
                /* s_GetNextAffineTbackFromMatch (:153-183) */
                int32_t new_seq2_index; const AOff *c = &AR(rows[d], best_diag);
                int taken = 0;
                if (best_diag >= diag_lower[d - op_cost] && best_diag <= diag_upper[d - op_cost]) {
                    new_seq2_index = AR(rows[d - op_cost], best_diag).match_off;
                    if (new_seq2_index >= ORC_MAX(c->insert_off, c->delete_off)) { d -= op_cost; state = OP_SUB; taken = 1; }
                }
                if (!taken) {
                    if (c->insert_off > c->delete_off) { new_seq2_index = c->insert_off; state = OP_INS; }
                    else { new_seq2_index = c->delete_off; state = OP_DEL; }
                }
                prelim_add(edit_block, OP_SUB, seq2_index - new_seq2_index);
                seq2_index = new_seq2_index;
            } else {
                /* s_GetNextAffineTbackFromIndel (:203-262) */
                const int ins = state == OP_INS;
                const int32_t new_diag = ins ? best_diag - 1 : best_diag + 1;
                int32_t new_seq2_index = kInvalidOffset, last_d = d - gap_extend;
                prelim_add(edit_block, ins ? OP_INS : OP_DEL, 1);
                if (new_diag >= diag_lower[last_d] && new_diag <= diag_upper[last_d])
                    new_seq2_index = ins ? AR(rows[last_d], new_diag).insert_off : AR(rows[last_d], new_diag).delete_off;
                last_d = d - gap_open_extend;
                if (new_diag >= diag_lower[last_d] && new_diag <= diag_upper[last_d] &&
                    new_seq2_index < AR(rows[last_d], new_diag).match_off) { d -= gap_open_extend; state = OP_SUB; }
                else d -= gap_extend;
                if (ins) best_diag--; else { best_diag++; seq2_index--; }
            }
        }
        prelim_add(edit_block, OP_SUB, AR(rows[0], diag_origin).match_off);
    }
    result = max_score[best_dist];
    for (d = 0; d < nrows; d++) free(rows[d].base);
    free(rows); free(bounds); free(max_score_base);
    return result;
}
/* BLAST_Gdb3 (CORE/ncbi_math.c:405-431) */
static int32_t gcd2(int32_t a, int32_t b) { int32_t c; b = abs(b); if (b > a) { c = a; a = b; b = c; } while (b != 0) { c = a % b; a = b; b = c; } return a; }
static int32_t gcd3(int32_t *a, int32_t *b, int32_t *c)
{
    int32_t g;
    if (*b == 0) g = gcd2(*a, *c); else g = gcd2(*a, gcd2(*b, *c));
    if (g > 1) { *a /= g; *b /= g; *c /= g; }
    return g;
}

/* s_ReduceGaps (CORE/blast_gapalign.c:2547-2617) */
static void reduce_gaps(OrcEditScript *esp, const uint8_t *q, const uint8_t *s)
{
    int i, j, nm1, nm2, d; const uint8_t *q1, *s1;
    for (i = 0; i < esp->size; i++) {
        if (esp->op[i] == OP_SUB) { q += esp->num[i]; s += esp->num[i]; continue; }
        if (i > 1 && esp->op[i] != esp->op[i - 2] && esp->num[i - 2] > 0) {
            d = esp->num[i] + esp->num[i - 1] + esp->num[i - 2];
            if (d == 3) {
                esp->num[i - 2] = 0; esp->num[i - 1] = 2; esp->num[i] = 0;
                if (esp->op[i] == OP_INS) ++q; else ++s;
            } else if (d < 12) {
                nm1 = 0; nm2 = 0;
                d = ORC_MIN(esp->num[i], esp->num[i - 2]);
                q -= esp->num[i - 1]; s -= esp->num[i - 1];
                q1 = q; s1 = s;
                if (esp->op[i] == OP_INS) s -= d; else q -= d;
                for (j = 0; j < esp->num[i - 1]; ++j, ++q1, ++s1, ++q, ++s) { if (*q1 == *s1) nm1++; if (*q == *s) nm2++; }
                for (j = 0; j < d; ++j, ++q, ++s) if (*q == *s) nm2++;
                if (nm2 >= nm1 - d) { esp->num[i - 2] -= d; esp->num[i - 1] += d; esp->num[i] -= d; }
                else { q = q1; s = s1; }
            }
        }
        if (esp->op[i] == OP_INS) q += esp->num[i]; else s += esp->num[i];
    }
    for (i = 0, j = 0; i < esp->size; i++) {
        if (esp->num[i] > 0) { esp->num[j] = esp->num[i]; esp->op[j] = esp->op[i]; ++j; }
        else if (++i < esp->size) esp->num[j - 1] += esp->num[i];
    }
    esp->size = j;
}

/* BLAST_GreedyGappedAlignment with do_traceback (CORE/blast_gapalign.c:2619-2751), gap costs 0 / 0 */
static int orc_greedy_with_traceback(const uint8_t *query, const uint8_t *subject, int32_t query_length, int32_t subject_length,
                              int32_t q_off, int32_t s_off, int32_t X, int32_t reward, int32_t penalty,
                              int32_t gap_open, int32_t gap_extend, OrcGapResult *r, OrcEditScript *esp)
{
    Prelim fwd = {0, 0, 0, 0, OP_INVALID}, rev = {0, 0, 0, 0, OP_INVALID};
    int32_t score, q_ext_l, q_ext_r, s_ext_l, s_ext_r, mc = reward, mm = -penalty, x = X, go = gap_open, ge = gap_extend;
    prelim_reset(&fwd); prelim_reset(&rev);
    if (mc % 2 == 1) { mc *= 2; mm *= 2; x *= 2; go *= 2; ge *= 2; }          /* CORE/greedy_align.c:795-806 */
    if (go == 0 && ge == 0) {
        score = greedy_align_tb(query + q_off, query_length - q_off, subject + s_off, subject_length - s_off, 0, x, mc, mm,
                                &q_ext_r, &s_ext_r, &fwd);
        score += greedy_align_tb(query, q_off, subject, s_off, 1, x, mc, mm, &q_ext_l, &s_ext_l, &rev);
        score = (q_ext_r + s_ext_r + q_ext_l + s_ext_l) * reward / 2 - score * (reward - penalty);
    } else {
        score = greedy_affine_tb(query + q_off, query_length - q_off, subject + s_off, subject_length - s_off, 0, x, mc, mm, go, ge,
                                 &q_ext_r, &s_ext_r, &fwd);
        score += greedy_affine_tb(query, q_off, subject, s_off, 1, x, mc, mm, go, ge, &q_ext_l, &s_ext_l, &rev);
        if (reward % 2 == 1) score /= 2;                                        /* CORE/blast_gapalign.c:2683-2685 */
    }
    prelim_to_esp(&rev, &fwd, esp);
    if (esp->size) reduce_gaps(esp, query + q_off - q_ext_l, subject + s_off - s_ext_l);
    r->q_start = q_off - q_ext_l; r->s_start = s_off - s_ext_l; r->q_stop = q_off + q_ext_r; r->s_stop = s_off + s_ext_r;
    r->seed_q = q_off; r->seed_s = s_off; r->score = score;
    free(fwd.op); free(fwd.num); free(rev.op); free(rev.num);
    return 0;
}

/* ---------------- start points ---------------- */
/* BLAST_CheckStartForGappedAlignment, CORE/blast_traceback.c:96-151 */
static int check_start(const int32_t matrix[16][16], const OrcHSP *h, const uint8_t *query, const uint8_t *subject)
{
    int32_t left = -HSP_MAX_WINDOW / 2, right = HSP_MAX_WINDOW / 2 + 1, score = 0, i;
    if (left < h->q_offset - h->q_gapped_start) left = h->q_offset - h->q_gapped_start;
    if (left < h->s_offset - h->s_gapped_start) left = h->s_offset - h->s_gapped_start;
    if (right > h->q_end - h->q_gapped_start) right = h->q_end - h->q_gapped_start;
    if (right > h->s_end - h->s_gapped_start) right = h->s_end - h->s_gapped_start;
    for (i = left; i < right; i++) score += matrix[query[h->q_gapped_start + i]][subject[h->s_gapped_start + i]];
    return score > 0;
}
/* BlastGetOffsetsForGappedAlignment, CORE/blast_gapalign.c:3059-3131 */
static int offsets_for_gapped(const int32_t matrix[16][16], const uint8_t *query, const uint8_t *subject,
                              const OrcHSP *h, int32_t *q_ret, int32_t *s_ret)
{
    int32_t index1, max_offset, score, max_score, hsp_end;
    const int32_t q_length = h->q_end - h->q_offset, s_length = h->s_end - h->s_offset, q_start = h->q_offset, s_start = h->s_offset;
    const uint8_t *qv, *sv;
    if (q_length <= HSP_MAX_WINDOW) { *q_ret = q_start + q_length / 2; *s_ret = s_start + q_length / 2; return 1; }
    hsp_end = q_start + HSP_MAX_WINDOW;
    qv = query + q_start; sv = subject + s_start; score = 0;
    for (index1 = q_start; index1 < hsp_end; index1++) { score += matrix[*qv][*sv]; qv++; sv++; }
    max_score = score; max_offset = hsp_end - 1;
    hsp_end = q_start + ORC_MIN(q_length, s_length);
    for (index1 = q_start + HSP_MAX_WINDOW; index1 < hsp_end; index1++) {
        score -= matrix[*(qv - HSP_MAX_WINDOW)][*(sv - HSP_MAX_WINDOW)];
        score += matrix[*qv][*sv];
        if (score > max_score) { max_score = score; max_offset = index1; }
        qv++; sv++;
    }
    if (max_score > 0) { *q_ret = max_offset; *s_ret = (max_offset - q_start) + s_start; return 1; }
    score = 0;
    qv = query + q_start + q_length - HSP_MAX_WINDOW; sv = subject + s_start + s_length - HSP_MAX_WINDOW;
    for (index1 = h->q_end - HSP_MAX_WINDOW; index1 < h->q_end; index1++) { score += matrix[*qv][*sv]; qv++; sv++; }
    if (score > 0) { *q_ret = h->q_end - HSP_MAX_WINDOW / 2; *s_ret = h->s_end - HSP_MAX_WINDOW / 2; return 1; }
    return 0;
}
/* BlastGetStartForGappedAlignmentNucl, CORE/blast_gapalign.c:3133-3183 */
static void start_for_gapped_nucl(const uint8_t *query, const uint8_t *subject, OrcHSP *h)
{
    const int32_t HSP_MAX_IDENT_RUN = 20;
    const uint8_t *q, *s; int32_t index, max_offset, score, max_score, q_start, s_start, q_len; int match = 0, prev_match;
    const int32_t offset = ORC_MIN(h->s_gapped_start - h->s_offset, h->q_gapped_start - h->q_offset);
    q_start = h->q_gapped_start - offset; s_start = h->s_gapped_start - offset;
    q_len = ORC_MIN(h->s_end - s_start, h->q_end - q_start);
    q = query + q_start; s = subject + s_start;
    max_score = 0; max_offset = q_start; score = 0; prev_match = 0;
    for (index = q_start; index < q_start + q_len; index++) {
        match = (*q++ == *s++);
        if (match != prev_match) {
            prev_match = match;
            if (match) score = 1;
            else if (score > max_score) { max_score = score; max_offset = index - score / 2; }
        } else if (match) {
            ++score;
            if (score > HSP_MAX_IDENT_RUN) {
                max_offset = index - HSP_MAX_IDENT_RUN / 2;
                h->q_gapped_start = max_offset; h->s_gapped_start = max_offset + s_start - q_start;
                return;
            }
        }
    }
    if (match && score > max_score) { max_score = score; max_offset = index - score / 2; }
    if (max_score > 0) { h->q_gapped_start = max_offset; h->s_gapped_start = max_offset + s_start - q_start; }
}
/* AdjustSubjectRange, CORE/blast_gapalign.c:3608-3637 */
static void adjust_subject_range(int32_t *s_off, int32_t *s_len, int32_t q_off, int32_t q_len, int32_t *start_shift)
{
    const int32_t subject_length = *s_len, s_offset = *s_off;
    int32_t max_left, max_right;
    if (subject_length < 90000) { *start_shift = 0; return; }
    max_left = q_off + 3000; max_right = q_len - q_off + 3000;
    if (s_offset <= max_left) *start_shift = 0;
    else { *start_shift = s_offset - max_left; *s_off = max_left; }
    *s_len = ORC_MIN(subject_length, s_offset + max_right) - *start_shift;
}

/* ---------------- per-HSP bookkeeping ---------------- */
/* s_Blast_HSPGetNumIdentitiesAndPositives, CORE/blast_hits.c:618-700 (nucleotide: identities only) */
static void num_identities(const uint8_t *query, const uint8_t *subject, const OrcHSP *h, const OrcEditScript *esp,
                           int32_t *num_ident, int32_t *align_length)
{
    const uint8_t *q = query + h->q_offset, *s = subject + h->s_offset; int32_t i, index, ni = 0, al = 0;
    for (index = 0; index < esp->size; index++) {
        al += esp->num[index];
        if (esp->op[index] == OP_SUB) { for (i = 0; i < esp->num[index]; i++) { if (*q == *s) ni++; q++; s++; } }
        else if (esp->op[index] == OP_DEL) s += esp->num[index];
        else if (esp->op[index] == OP_INS) q += esp->num[index];
        else { s += esp->num[index]; q += esp->num[index]; }
    }
    *num_ident = ni; *align_length = al;
}
/* s_CutOffGapEditScript, CORE/blast_hits.c:2163-2222 */
static void cut_off_esp(OrcHSP *h, OrcEditScript *esp, int32_t q_cut, int32_t s_cut, int cut_begin)
{
    int index, opid = 0, qid = 0, sid = 0, found = 0;
    q_cut -= h->q_offset; s_cut -= h->s_offset;
    for (index = 0; index < esp->size; index++) {
        for (opid = 0; opid < esp->num[index];) {
            if (esp->op[index] == OP_SUB) { qid++; sid++; opid++; }
            else if (esp->op[index] == OP_DEL) { sid += esp->num[index]; opid += esp->num[index]; }
            else if (esp->op[index] == OP_INS) { qid += esp->num[index]; opid += esp->num[index]; }
            if (qid >= q_cut && sid >= s_cut) found = 1;
            if (found) break;
        }
        if (found) break;
    }
    if (!found) return;
    if (cut_begin) {
        int new_index = 0;
        if (opid < esp->num[index]) { esp->op[0] = esp->op[index]; esp->num[0] = esp->num[index] - opid; new_index++; }
        ++index;
        for (; index < esp->size; index++, new_index++) { esp->op[new_index] = esp->op[index]; esp->num[new_index] = esp->num[index]; }
        esp->size = new_index;
        h->q_offset += qid; h->s_offset += sid;
    } else {
        if (opid < esp->num[index]) esp->num[index] = opid;
        esp->size = index + 1;
        h->q_end = h->q_offset + qid; h->s_end = h->s_offset + sid;
    }
}
/* Blast_HSPReevaluateWithAmbiguitiesGapped, CORE/blast_hits.c:342-520; returns 1: delete */
static int reevaluate_gapped(OrcHSP *h, OrcEditScript *esp, const int32_t matrix[16][16], const uint8_t *q, int32_t qlen,
                             const uint8_t *s, int32_t slen, int32_t cutoff_score, int32_t reward, int32_t penalty,
                             int32_t in_gap_open, int32_t in_gap_extend)
{
    int32_t sum = 0, score = 0, gap_open, gap_extend, index, qp, sp, ext, factor = 1;
    int best_start = 0, best_end = 0, current_start = 0, best_end_num = -1;
    const uint8_t *query = q + h->q_offset, *subject = s + h->s_offset;
    const uint8_t *best_q_start = query, *best_q_end = query, *current_q_start = query;
    const uint8_t *best_s_start = subject, *best_s_end = subject, *current_s_start = subject;
    if (in_gap_open == 0 && in_gap_extend == 0) {
        if (reward % 2 == 1) factor = 2;
        gap_open = 0; gap_extend = (reward - 2 * penalty) * factor / 2;
    } else { gap_open = in_gap_open; gap_extend = in_gap_extend; }
    if (!esp->op) return 1;
    for (index = 0; index < esp->size; index++) {
        int op_index;
        for (op_index = 0; op_index < esp->num[index];) {
            if (esp->op[index] == OP_SUB) { sum += factor * matrix[*query & 0x0f][*subject]; query++; subject++; op_index++; }
            else if (esp->op[index] == OP_DEL) { sum -= gap_open + gap_extend * esp->num[index]; subject += esp->num[index]; op_index += esp->num[index]; }
            else if (esp->op[index] == OP_INS) { sum -= gap_open + gap_extend * esp->num[index]; query += esp->num[index]; op_index += esp->num[index]; }
            if (sum < 0) {
                if (op_index < esp->num[index]) { esp->num[index] -= op_index; current_start = index; op_index = 0; }
                else current_start = index + 1;
                sum = 0; current_q_start = query; current_s_start = subject;
                if (score < cutoff_score) {
                    best_q_start = query; best_s_start = subject; score = 0;
                    best_start = current_start; best_end = current_start;
                }
            } else if (sum > score) {
                score = sum;
                best_q_start = current_q_start; best_s_start = current_s_start; best_q_end = query; best_s_end = subject;
                best_start = current_start; best_end = index; best_end_num = op_index;
            }
        }
    }
    score /= factor;
    if (best_start < esp->size && best_end < esp->size) {
        qp = (int32_t)(best_q_start - q); sp = (int32_t)(best_s_start - s); ext = 0;
        while (qp > 0 && sp > 0 && (q[--qp] == s[--sp]) && q[qp] < 4) ext++;
        best_q_start -= ext; best_s_start -= ext;
        esp->num[best_start] += ext;
        if (best_end == best_start) best_end_num += ext;
        score += ext * reward;
        qp = (int32_t)(best_q_end - q); sp = (int32_t)(best_s_end - s); ext = 0;
        while (qp < qlen && sp < slen && q[qp] < 4 && (q[qp++] == s[sp++])) ext++;
        best_q_end += ext; best_s_end += ext;
        esp->num[best_end] += ext; best_end_num += ext;
        score += ext * reward;
    }
    /* s_UpdateReevaluatedHSP, :311-340 */
    h->score = score;
    if (h->score >= cutoff_score) {
        h->q_offset = (int32_t)(best_q_start - q); h->q_end = h->q_offset + (int32_t)(best_q_end - best_q_start);
        h->s_offset = (int32_t)(best_s_start - s); h->s_end = h->s_offset + (int32_t)(best_s_end - best_s_start);
        if (best_end != esp->size - 1 || best_start > 0) {
            int32_t i, n = best_end - best_start + 1;
            for (i = 0; i < n; i++) { esp->op[i] = esp->op[best_start + i]; esp->num[i] = esp->num[best_start + i]; }
            esp->size = n;
        }
        esp->num[esp->size - 1] = best_end_num;
        return 0;
    }
    return 1;
}

/* ---------------- the (query, subject) HSP list ---------------- */
typedef struct { OrcHSP h; OrcEditScript e; int32_t num_ident, align_len; int alive; } TbHsp;

static int cmp_tb_qoff(const TbHsp *x, const TbHsp *y)          /* CORE/blast_hits.c:2037-2090, null HSPs last */
{
    const OrcHSP *a = &x->h, *b = &y->h;
    if (!x->alive && !y->alive) return 0;
    if (!x->alive) return 1;
    if (!y->alive) return -1;
    if (a->context != b->context) return a->context < b->context ? -1 : 1;
    if (a->q_offset != b->q_offset) return a->q_offset < b->q_offset ? -1 : 1;
    if (a->s_offset != b->s_offset) return a->s_offset < b->s_offset ? -1 : 1;
    if (a->score != b->score) return a->score < b->score ? 1 : -1;
    if (a->q_end != b->q_end) return a->q_end < b->q_end ? 1 : -1;
    if (a->s_end != b->s_end) return a->s_end < b->s_end ? 1 : -1;
    return 0;
}
static int cmp_tb_qend(const TbHsp *x, const TbHsp *y)          /* :2102-2160 */
{
    const OrcHSP *a = &x->h, *b = &y->h;
    if (!x->alive && !y->alive) return 0;
    if (!x->alive) return 1;
    if (!y->alive) return -1;
    if (a->context != b->context) return a->context < b->context ? -1 : 1;
    if (a->q_end != b->q_end) return a->q_end < b->q_end ? -1 : 1;
    if (a->s_end != b->s_end) return a->s_end < b->s_end ? -1 : 1;
    if (a->score != b->score) return a->score < b->score ? 1 : -1;
    if (a->q_offset != b->q_offset) return a->q_offset < b->q_offset ? 1 : -1;
    if (a->s_offset != b->s_offset) return a->s_offset < b->s_offset ? 1 : -1;
    return 0;
}
static int cmp_tb_score(const TbHsp *x, const TbHsp *y)
{
    if (!x->alive && !y->alive) return 0;
    if (!x->alive) return 1;
    if (!y->alive) return -1;
    return orc_score_compare_hsps(&x->h, &y->h);
}
static void tb_sort(TbHsp *a, int32_t n, int (*cmp)(const TbHsp *, const TbHsp *))   /* stable merge sort (glibc qsort) */
{
    TbHsp *tmp; int32_t width, i;
    if (n < 2) return;
    tmp = (TbHsp *)malloc((size_t)n * sizeof(*tmp));
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int32_t l = i, m = ORC_MIN(i + width, n), r = ORC_MIN(i + 2 * width, n), a0 = l, b0 = m, k = l;
            while (a0 < m && b0 < r) tmp[k++] = (cmp(&a[b0], &a[a0]) < 0) ? a[b0++] : a[a0++];
            while (a0 < m) tmp[k++] = a[a0++];
            while (b0 < r) tmp[k++] = a[b0++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(*tmp));
    }
    free(tmp);
}
static int32_t tb_compact(TbHsp *a, int32_t n)                  /* Blast_HSPListPurgeNullHSPs */
{
    int32_t i, k = 0;
    for (i = 0; i < n; i++) { if (a[i].alive) { if (k != i) a[k] = a[i]; k++; } else esp_free(&a[i].e); }
    return k;
}
/* Blast_HSPListPurgeHSPsWithCommonEndpoints with purge = FALSE (blastn program), CORE/blast_hits.c:2224-2302:
 * returns hsp_count; the cut or dropped HSPs sit behind it */
static int32_t tb_purge_common(TbHsp *a, int32_t n)
{
    int32_t i, j, k, cnt = n;
    if (n == 0) return 0;
    tb_sort(a, cnt, cmp_tb_qoff);
    i = 0;
    while (i < cnt) {
        j = 1;
        while (i + j < cnt && a[i].alive && a[i + j].alive && a[i].h.context == a[i + j].h.context &&
               a[i].h.q_offset == a[i + j].h.q_offset && a[i].h.s_offset == a[i + j].h.s_offset) {
            TbHsp t;
            cnt--;
            t = a[i + j];
            if (t.h.q_end > a[i].h.q_end) cut_off_esp(&t.h, &t.e, a[i].h.q_end, a[i].h.s_end, 1);
            else { esp_free(&t.e); t.alive = 0; }
            for (k = i + j; k < cnt; k++) a[k] = a[k + 1];
            a[cnt] = t;
        }
        i += j;
    }
    tb_sort(a, cnt, cmp_tb_qend);
    i = 0;
    while (i < cnt) {
        j = 1;
        while (i + j < cnt && a[i].alive && a[i + j].alive && a[i].h.context == a[i + j].h.context &&
               a[i].h.q_end == a[i + j].h.q_end && a[i].h.s_end == a[i + j].h.s_end) {
            TbHsp t;
            cnt--;
            t = a[i + j];
            if (t.h.q_offset < a[i].h.q_offset) cut_off_esp(&t.h, &t.e, a[i].h.q_offset, a[i].h.s_offset, 0);
            else { esp_free(&t.e); t.alive = 0; }
            for (k = i + j; k < cnt; k++) a[k] = a[k + 1];
            a[cnt] = t;
        }
        i += j;
    }
    return cnt;
}

/* interval tree of orc_gapped.c */
void *orc_itree_new(const OrcSearch *S, int32_t q_end, int32_t s_end);
void orc_itree_reset(void *t, int32_t q_end, int32_t s_end);
int orc_itree_contains_hsp(void *t, const OrcHSP *pool, const OrcHSP *hsp, int32_t mds);
void orc_itree_add_hsp(void *t, const OrcHSP *pool, int32_t idx);
void orc_itree_free(void *t);

/* Blast_TracebackFromHSPList + s_HSPListPostTracebackUpdate for blastn / megablast.
 * subject: BLASTNA, one code per base.  in[]: the preliminary HSPs of ONE query against this subject, sorted by
 * score.  Returns the number of final HSPs (malloc'd arrays, caller frees with orc_traceback_free). */
int32_t orc_traceback_hsp_list(const OrcSearch *S, const uint8_t *subject, int32_t subject_length,
                               const OrcHSP *in, int32_t nin, OrcTbHSP **out)
{
    const OrcOptions *o = &S->opt;
    TbHsp *a = (TbHsp *)calloc((size_t)ORC_MAX(nin, 1), sizeof(TbHsp));
    OrcHSP *pool = (OrcHSP *)calloc((size_t)ORC_MAX(nin, 1), sizeof(OrcHSP));   /* what the interval tree indexes */
    void *tree = orc_itree_new(S, S->qlen + 1, subject_length + 1);
    int32_t index, n = nin, extra_start, npool = 0;
    const int greedy_tb = o->greedy;            /* eGreedyTbck with the greedy preliminary aligner (CORE/blast_options.c) */
    const int32_t X = S->gap_x_dropoff_final;
    *out = NULL;
    for (index = 0; index < nin; index++) { a[index].h = in[index]; a[index].alive = 1; }
    for (index = 0; index < nin; index++) {
        TbHsp *t = &a[index]; OrcHSP *h = &t->h;
        const int32_t ctx = h->context, qstart = S->ctx[ctx].query_offset, query_length = S->ctx[ctx].query_length;
        const uint8_t *query = S->query + qstart;
        int32_t q_start, s_start, start_shift = 0, adjusted_s_length, cutoff; const uint8_t *adjusted_subject;
        OrcGapResult r;
        if (orc_itree_contains_hsp(tree, pool, h, o->min_diag_separation)) { t->alive = 0; continue; }
        if ((h->q_gapped_start == 0 && h->s_gapped_start == 0) || !check_start(S->matrix, h, query, subject)) {
            if (!offsets_for_gapped(S->matrix, query, subject, h, &q_start, &s_start)) { t->alive = 0; continue; }
            h->q_gapped_start = q_start; h->s_gapped_start = s_start;
        } else {
            start_for_gapped_nucl(query, subject, h);       /* program blastn */
            q_start = h->q_gapped_start; s_start = h->s_gapped_start;
        }
        adjusted_s_length = subject_length; adjusted_subject = subject;
        adjust_subject_range(&s_start, &adjusted_s_length, q_start, query_length, &start_shift);
        adjusted_subject = subject + start_shift;
        h->s_gapped_start = s_start;
        cutoff = S->ctx[ctx].gap_cutoff_score;
        (void)cutoff;
        if (greedy_tb) {
            orc_greedy_with_traceback(query, adjusted_subject, query_length, adjusted_s_length, q_start, s_start, X,
                                      o->reward, o->penalty, o->gap_open, o->gap_extend, &r, &t->e);
        } else {
            orc_gapped_with_traceback(S->matrix, query, adjusted_subject, query_length, adjusted_s_length, q_start, s_start,
                                      X, o->gap_open, o->gap_extend, &r, &t->e);
        }
        /* Blast_HSPUpdateWithTraceback */
        h->score = r.score; h->q_offset = r.q_start; h->s_offset = r.s_start; h->q_end = r.q_stop; h->s_end = r.s_stop;
        if (!greedy_tb) num_identities(query, adjusted_subject, h, &t->e, &t->num_ident, &t->align_len);   /* (percent identity 0: kept) */
        if (start_shift > 0) { h->s_offset += start_shift; h->s_end += start_shift; h->s_gapped_start += start_shift; }
        pool[npool] = *h;
        orc_itree_add_hsp(tree, pool, npool);
        npool++;
    }
    n = tb_compact(a, n);
    extra_start = tb_purge_common(a, n);
    if (greedy_tb) extra_start = 0;
    for (index = extra_start; index < n; index++) {
        TbHsp *t = &a[index]; int del;
        const int32_t ctx = t->h.context; const uint8_t *query;
        if (!t->alive) continue;
        query = S->query + S->ctx[ctx].query_offset;
        del = reevaluate_gapped(&t->h, &t->e, S->matrix, query, S->ctx[ctx].query_length, subject, subject_length,
                                S->ctx[ctx].gap_cutoff_score, o->reward, o->penalty, o->gap_open, o->gap_extend);
        if (!del) num_identities(query, subject, &t->h, &t->e, &t->num_ident, &t->align_len);   /* Blast_HSPTestIdentityAndLength */
        if (del) { t->alive = 0; }
    }
    n = tb_compact(a, n);
    {   /* Blast_HSPListSortByScore */
        int sorted = 1;
        for (index = 0; index + 1 < n; index++) if (cmp_tb_score(&a[index], &a[index + 1]) > 0) { sorted = 0; break; }
        if (!sorted) tb_sort(a, n, cmp_tb_score);
    }
    /* HSPs contained in a better one go */
    orc_itree_reset(tree, S->qlen + 1, subject_length + 1);
    npool = 0;
    pool = (OrcHSP *)realloc(pool, (size_t)ORC_MAX(n, 1) * sizeof(OrcHSP));
    for (index = 0; index < n; index++) {
        if (orc_itree_contains_hsp(tree, pool, &a[index].h, o->min_diag_separation)) a[index].alive = 0;
        else { pool[npool] = a[index].h; orc_itree_add_hsp(tree, pool, npool); npool++; }
    }
    n = tb_compact(a, n);
    orc_itree_free(tree); free(pool);
    /* s_HSPListPostTracebackUpdate: odd scores, e-values, reap, bit scores */
    if (S->round_down) for (index = 0; index < n; index++) a[index].h.score &= ~1;
    {
        int32_t k = 0;
        for (index = 0; index < n; index++) {
            OrcHSP *h = &a[index].h;
            h->evalue = orc_karlin_StoE(h->score, &S->kbp_gap, S->ctx[h->context].eff_searchsp);
            if (h->evalue > o->evalue) { esp_free(&a[index].e); continue; }
            if (k != index) a[k] = a[index];
            k++;
        }
        n = k;
    }
    if (n) {
        OrcTbHSP *res = (OrcTbHSP *)calloc((size_t)n, sizeof(OrcTbHSP));
        for (index = 0; index < n; index++) {
            int32_t i, gaps = 0, opens = 0, length = a[index].h.q_end - a[index].h.q_offset;
            res[index].hsp = a[index].h; res[index].esp = a[index].e; res[index].num_ident = a[index].num_ident;
            /* Blast_HSPCalcLengthAndGaps, CORE/blast_hits.c:912-942 */
            for (i = 0; i < a[index].e.size; i++) {
                if (a[index].e.op[i] == OP_DEL) { length += a[index].e.num[i]; gaps += a[index].e.num[i]; ++opens; }
                else if (a[index].e.op[i] == OP_INS) { ++opens; gaps += a[index].e.num[i]; }
            }
            res[index].align_length = length; res[index].gaps = gaps; res[index].gap_opens = opens;
            /* Blast_HSPListGetBitScores, CORE/blast_hits.c:1741-1763 */
            res[index].bit_score = (S->kbp_gap.Lambda * a[index].h.score - S->kbp_gap.logK) / ORC_LN2;
        }
        *out = res;
    }
    free(a);
    return n;
}
int orc_tb_dynprog(const OrcSearch *S, int32_t context, const uint8_t *subject, int32_t subject_length,
                   int32_t q_start, int32_t s_start, int32_t x_dropoff, OrcGapOut *o, OrcEditScript *esp)
{
    OrcGapResult r;
    int rc = orc_gapped_with_traceback(S->matrix, S->query + S->ctx[context].query_offset, subject, S->ctx[context].query_length,
                                       subject_length, q_start, s_start, x_dropoff, S->opt.gap_open, S->opt.gap_extend, &r, esp);
    o->q_start = r.q_start; o->q_stop = r.q_stop; o->s_start = r.s_start; o->s_stop = r.s_stop; o->score = r.score; o->seed_q = r.seed_q; o->seed_s = r.seed_s;
    return rc;
}
int orc_tb_greedy(const OrcSearch *S, int32_t context, const uint8_t *subject, int32_t subject_length,
                  int32_t q_start, int32_t s_start, int32_t x_dropoff, OrcGapOut *o, OrcEditScript *esp)
{
    OrcGapResult r;
    int rc = orc_greedy_with_traceback(S->query + S->ctx[context].query_offset, subject, S->ctx[context].query_length, subject_length,
                                       q_start, s_start, x_dropoff, S->opt.reward, S->opt.penalty, S->opt.gap_open, S->opt.gap_extend, &r, esp);
    o->q_start = r.q_start; o->q_stop = r.q_stop; o->s_start = r.s_start; o->s_stop = r.s_stop; o->score = r.score; o->seed_q = r.seed_q; o->seed_s = r.seed_s;
    return rc;
}
const int32_t *orc_matrix(const OrcSearch *S) { return &S->matrix[0][0]; }
void orc_esp_free(OrcEditScript *e) { esp_free(e); }
void orc_traceback_free(OrcTbHSP *h, int32_t n)
{
    int32_t i;
    if (!h) return;
    for (i = 0; i < n; i++) esp_free(&h[i].esp);
    free(h);
}
