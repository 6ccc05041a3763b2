/* orc_wordfinder.c -- ORACLE (test infrastructure): BlastNaWordFinder restated
 * from CORE/na_ungapped.c, CORE/blast_nascan.c, CORE/blast_extend.c.
 *
 * Scan: every scanner in CORE/blast_nascan.c visits subject offsets
 * 0, step, 2*step, ... <= len - lut (CORE/na_ungapped.c:1609-1611,
 * CORE/masksubj.inl:42-59 with one unmasked range) and reports, per offset,
 * the cell's query offsets in table order; the offset-array capacity only
 * batches the work (CORE/lookup_wrap.c:229-259) and never drops a hit, so the
 * restatement is one loop. */
#include "orc_int.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- diagonal containers ---------------- */
#define NBUCKETS 512            /* DIAGHASH_NUM_BUCKETS, COREI/blast_extend.h:51 */
typedef struct { int32_t diag, level, hit_len, hit_saved; uint32_t next; } HCell;
typedef struct {
    uint32_t backbone[NBUCKETS];
    HCell *chain; uint32_t occupancy, capacity;
    int32_t offset;
} DHash;

static DHash *dhash_get(OrcSearch *s)
{
    DHash *h = (DHash *)s->diag_hash;
    if (!h) {
        h = (DHash *)calloc(1, sizeof(*h));
        h->capacity = 256;      /* DIAGHASH_CHAIN_LENGTH */
        h->chain = (HCell *)calloc(h->capacity, sizeof(HCell));
        s->diag_hash = h;
    }
    return h;
}
/* fresh state for each subject: CORE/blast_extend.c:136-143 */
static void dhash_reset(DHash *h)
{
    memset(h->backbone, 0, sizeof(h->backbone));
    h->occupancy = 1; h->offset = 0;
}
/* CORE/na_ungapped.c:362-383 */
static int dhash_retrieve(DHash *t, int32_t diag, int32_t *level, int32_t *hit_len, int32_t *hit_saved)
{
    uint32_t bucket = ((uint32_t)diag * 0x9E370001u) % NBUCKETS;
    uint32_t index = t->backbone[bucket];
    while (index) {
        if (t->chain[index].diag == diag) {
            *level = t->chain[index].level;
            *hit_len = t->chain[index].hit_len;
            *hit_saved = t->chain[index].hit_saved;
            return 1;
        }
        index = t->chain[index].next;
    }
    return 0;
}
/* CORE/na_ungapped.c:397-451 */
static void dhash_insert(DHash *t, int32_t diag, int32_t level, int32_t len, int32_t hit_saved,
                         int32_t s_off, int32_t window_size)
{
    uint32_t bucket = ((uint32_t)diag * 0x9E370001u) % NBUCKETS;
    uint32_t index = t->backbone[bucket];
    HCell *cell;
    while (index) {
        if (t->chain[index].diag == diag) {
            t->chain[index].level = level; t->chain[index].hit_len = len;
            t->chain[index].hit_saved = hit_saved;
            return;
        } else if (s_off - t->chain[index].level > window_size) {
            t->chain[index].diag = diag; t->chain[index].level = level;
            t->chain[index].hit_len = len; t->chain[index].hit_saved = hit_saved;
            return;
        }
        index = t->chain[index].next;
    }
    if (t->occupancy == t->capacity) {
        t->capacity *= 2;
        t->chain = (HCell *)realloc(t->chain, t->capacity * sizeof(HCell));
    }
    cell = t->chain + t->occupancy;
    cell->diag = diag; cell->level = level; cell->hit_len = len; cell->hit_saved = hit_saved;
    cell->next = t->backbone[bucket];
    t->backbone[bucket] = t->occupancy;
    t->occupancy++;
}

/* ---------------- ungapped extension ---------------- */
typedef struct { int32_t q_start, s_start, length, score; } Ungapped;

/* CORE/na_ungapped.c:152-244 (s_NuclUngappedExtendExact) */
static void ungapped_exact(const OrcSearch *S, const uint8_t *subj, int32_t slen,
                           int32_t q_off, int32_t s_off, int32_t X, Ungapped *u)
{
    const uint8_t *q = S->query;
    int32_t sum = 0, score = 0, i;
    int32_t q_beg = q_off, q_end = q_off;
    int32_t nleft = ORC_MIN(q_off, s_off);
    int32_t q_avail = S->qlen - q_off, s_avail = slen - s_off;
    int32_t nright = ORC_MIN(q_avail, s_avail);
    for (i = 1; i <= nleft; i++) {
        int32_t qi = q_off - i;
        if ((sum += S->matrix[q[qi]][ORC_BASE(subj, s_off - i)]) > 0) {
            q_beg = qi; score += sum; sum = 0;
        } else if (sum < X) break;
    }
    u->q_start = q_beg;
    u->s_start = s_off - (q_off - q_beg);
    sum = 0;
    for (i = 0; i < nright; i++) {
        int32_t qi = q_off + i;
        if ((sum += S->matrix[q[qi]][ORC_BASE(subj, s_off + i)]) > 0) {
            q_end = qi + 1; score += sum; sum = 0;
        } else if (sum < X) break;
    }
    u->length = q_end - q_beg;
    u->score = score;
}

/* CORE/na_ungapped.c:262-351 (s_NuclUngappedExtend) */
static void ungapped_approx(const OrcSearch *S, const uint8_t *subj, int32_t slen,
                            int32_t q_off, int32_t s_match_end, int32_t s_off,
                            int32_t X, Ungapped *u, int32_t reduced_cutoff)
{
    const uint8_t *qs = S->query;
    int32_t len, q_ext, s_ext, i, sum, score, new_q;
    const uint8_t *q; int32_t sb;
    len = (4 - (s_off % 4)) % 4;
    q_ext = q_off + len; s_ext = s_off + len;
    q = qs + q_ext; sb = s_ext / 4;
    len = ORC_MIN(q_ext, s_ext) / 4;
    score = 0; sum = 0; new_q = q_ext;
    for (i = 0; i < len; sb--, q -= 4, i++) {
        uint8_t s_byte = subj[sb - 1];
        uint8_t q_byte = (uint8_t)((q[-4] << 6) | (q[-3] << 4) | (q[-2] << 2) | q[-1]);
        sum += S->score_table[q_byte ^ s_byte];
        if (sum > 0) { new_q = (int32_t)(q - qs) - 4; score += sum; sum = 0; }
        if (sum < X) break;
    }
    u->q_start = new_q;
    u->s_start = s_ext - (q_ext - u->q_start);
    q = qs + q_ext; sb = s_ext / 4;
    len = ORC_MIN(S->qlen - q_ext, slen - s_ext) / 4;
    sum = 0; new_q = q_ext;
    for (i = 0; i < len; sb++, q += 4, i++) {
        uint8_t s_byte = subj[sb];
        uint8_t q_byte = (uint8_t)((q[0] << 6) | (q[1] << 4) | (q[2] << 2) | q[3]);
        sum += S->score_table[q_byte ^ s_byte];
        if (sum > 0) { new_q = (int32_t)(q - qs) + 3; score += sum; sum = 0; }
        if (sum < X) break;
    }
    if (score >= reduced_cutoff) {
        ungapped_exact(S, subj, slen, q_off, s_off, X, u);
    } else {
        u->score = score;
        u->length = ORC_MAX(s_match_end - u->s_start, new_q - u->q_start + 1);
    }
}

/* CORE/na_ungapped.c:459-471 (s_IsSeedMasked): the lookup word the SUBJECT carries at s_off, looked
 * up in the table together with q_pos -- a masked (or ambiguous) query position is not in its cell */
static int seed_masked(const OrcSearch *S, const uint8_t *subj, int32_t s_off, int32_t lut, int32_t q_pos)
{
    uint32_t index = 0; int32_t i;
    for (i = 0; i < lut; i++) index = (index << 2) | (uint32_t)ORC_BASE(subj, s_off + i);
    return !orc_lookup_has(S->lut, (int32_t)index, q_pos);
}

/* CORE/na_ungapped.c:488-587 (s_TypeOfWord) with check_double FALSE (one-hit mode): re-check of the
 * mini-extended word against the query masks; may move the left end right and extend the right end */
static int type_of_word(const OrcSearch *S, const uint8_t *subj, int32_t *q_off, int32_t *s_off,
                        int32_t s_range, int32_t word_length, int32_t lut, int32_t *extended)
{
    int32_t context, q_range, ext_to, ext_max, s_pos, q_pos;
    int32_t q_end = *q_off + word_length, s_end = *s_off + word_length;
    *extended = 0;
    if (word_length == lut) return 1;
    context = orc_context_of(S, q_end);
    q_range = S->ctx[context].query_offset + S->ctx[context].query_length;
    if (S->masked) {
        if (seed_masked(S, subj, s_end - lut, lut, q_end - lut)) return 0;
        for (;; ++(*s_off), ++(*q_off))
            if (!seed_masked(S, subj, *s_off, lut, *q_off)) break;
    }
    ext_to = word_length - (q_end - (*q_off));
    ext_max = ORC_MIN(q_range - q_end, s_range - s_end);
    if (ext_to || S->masked) {
        if (ext_to > ext_max) return 0;
        q_end += ext_to; s_end += ext_to;
        for (s_pos = s_end - lut, q_pos = q_end - lut; s_pos > *s_off; s_pos -= lut, q_pos -= lut)
            if (seed_masked(S, subj, s_pos, lut, q_pos)) return 0;
        *extended = ext_to;
    }
    return 1;
}

/* one seed through the diagonal container and, if it survives, the ungapped
 * extension: CORE/na_ungapped.c:611-755 (array) / :778-922 (hash) in one-hit
 * mode (window_size 0). */
static int diag_extend(OrcSearch *S, const uint8_t *subj, int32_t slen,
                       int32_t q_off, int32_t s_off, int32_t word_length)
{
    int32_t s_end = s_off + word_length, s_off_pos, s_end_pos, last_hit;
    int hit_ready = 1, context;
    Ungapped u; const OrcContext *c;
    DHash *h = NULL; int32_t real_diag = 0, diag = 0;

    if (S->container == ORC_DIAG_HASH) {
        int32_t s_l, saved = 0;
        h = dhash_get(S);
        diag = s_off - q_off;
        s_off_pos = s_off + h->offset; s_end_pos = s_end + h->offset;
        if (!dhash_retrieve(h, diag, &last_hit, &s_l, &saved)) last_hit = 0;
    } else {
        diag = s_off + S->diag_len - q_off;
        real_diag = diag & S->diag_mask;
        last_hit = S->diag_last_hit[real_diag];
        s_off_pos = s_off + S->diag_offset; s_end_pos = s_end + S->diag_offset;       /* offset 0 unless carried */
    }
    if (s_off_pos < last_hit) return 0;

    {   /* check the masks for the word (:696-704 / :868-876) */
        int32_t extended = 0;
        if (!type_of_word(S, subj, &q_off, &s_off, slen, word_length, S->lut->lut_word_length, &extended)) return 0;
        s_end += extended; s_end_pos += extended;
    }
    context = orc_context_of(S, q_off);
    c = &S->ctx[context];
    /* word_length < 11 in blastn goes straight to the exact extension on the
     * array path only (:720-724 vs :889) */
    if (S->container == ORC_DIAG_ARRAY && word_length < 11)
        ungapped_exact(S, subj, slen, q_off, s_off, -c->x_dropoff, &u);
    else
        ungapped_approx(S, subj, slen, q_off, s_end, s_off, -c->x_dropoff, &u,
                        c->reduced_cutoff);
    if (u.score >= c->cutoff_score) {
        OrcInitHit ih;
        ih.q_off = q_off; ih.s_off = s_off;
        ih.q_start = u.q_start; ih.s_start = u.s_start; ih.length = u.length; ih.score = u.score;
        orc_push_ihit(S, &ih);
        s_end_pos = u.length + u.s_start + (h ? h->offset : S->diag_offset);
    } else {
        hit_ready = 0;
    }
    if (h) {
        /* window_size + Delta + 1 with window 0, scan_range 0:
         * Delta = MIN(0, 0 - word_length) (:804), never clamped in one-hit mode */
        int32_t Delta = ORC_MIN(0, 0 - word_length);
        dhash_insert(h, diag, s_end_pos, hit_ready ? 0 : s_end_pos - s_off_pos, hit_ready,
                     s_off_pos, 0 + Delta + 1);
    } else {
        S->diag_last_hit[real_diag] = s_end_pos;
    }
    return hit_ready;
}

/* ---------------- mini-extensions ---------------- */
/* CORE/na_ungapped.c:1025-1144 (s_BlastNaExtend); :1165-1290 is the same rule
 * specialised for byte-aligned hits */
static int miniext_na(const OrcSearch *S, const uint8_t *subj, int32_t s_range,
                      int32_t *q_offset, int32_t *s_offset, int32_t word, int32_t lut)
{
    const uint8_t *q = S->query;
    int32_t ext_to = word - lut, ext_left = 0, ext_max = ORC_MIN(ext_to, *s_offset);
    for (; ext_left < ext_max; ++ext_left) {
        int32_t sp = *s_offset - ext_left - 1, qp = *q_offset - ext_left - 1;
        if (ORC_BASE(subj, sp) != q[qp]) break;
    }
    if (ext_left < ext_to) {
        int32_t ext_right = 0, s_off = *s_offset + lut, need = ext_to - ext_left;
        if (s_off + need > s_range) return 0;
        for (; ext_right < need; ++ext_right)
            if (ORC_BASE(subj, s_off + ext_right) != q[*q_offset + lut + ext_right]) break;
        if (ext_left + ext_right < ext_to) return 0;
    }
    *q_offset -= ext_left; *s_offset -= ext_left;
    return 1;
}

/* CORE/na_ungapped.c:1449-1555 (s_BlastSmallNaExtend): compressed query, i.e.
 * ambiguity codes compare as (code & 3) (CORE/blast_util.c:459-502), with
 * explicit context-boundary clamps */
static int miniext_small(const OrcSearch *S, const uint8_t *subj, int32_t s_range,
                         int32_t *q_offset, int32_t *s_offset, int32_t word, int32_t lut)
{
    const uint8_t *q = S->query;
    int32_t context = orc_context_of(S, *q_offset);
    int32_t q_start = S->ctx[context].query_offset;
    int32_t q_range = q_start + S->ctx[context].query_length;
    int32_t ext_max = ORC_MIN(ORC_MIN(word - lut, *s_offset), *q_offset - q_start);
    int32_t rsdl = 4 - (*s_offset % 4);
    int32_t so = *s_offset + rsdl, qo = *q_offset + rsdl, ext_left = 0, ext_right = 0;
    ext_max += rsdl;
    while (ext_left < ext_max) {
        if ((q[qo - ext_left - 1] & 3) != ORC_BASE(subj, so - ext_left - 1)) break;
        ext_left++;
    }
    ext_max = ORC_MIN(ORC_MIN(word - ext_left, s_range - so), q_range - qo);
    while (ext_right < ext_max) {
        if ((q[qo + ext_right] & 3) != ORC_BASE(subj, so + ext_right)) break;
        ext_right++;
    }
    if (ext_left + ext_right < word) return 0;
    *q_offset = qo - ext_left; *s_offset = so - ext_left;
    return 1;
}

/* CORE/na_ungapped.c:1346-1427 (s_BlastSmallNaExtendAlignedOneByte), including
 * its rule that a hit whose lookup word ends exactly at the end of the
 * concatenated query skips the right-hand check */
static int miniext_small_onebyte(const OrcSearch *S, const uint8_t *subj, int32_t s_range,
                                 int32_t *q_offset, int32_t *s_offset, int32_t word, int32_t lut)
{
    const uint8_t *q = S->query;
    int32_t ext_to = word - lut, ext_left = 0;
    int32_t context = orc_context_of(S, *q_offset);
    int32_t q_start = S->ctx[context].query_offset;
    int32_t q_range = q_start + S->ctx[context].query_length;
    if (*s_offset > 0 && *q_offset > 0) {
        int32_t k = 0;
        while (k < 4 && (q[*q_offset - k - 1] & 3) == ORC_BASE(subj, *s_offset - k - 1)) k++;
        ext_left = ORC_MIN(ORC_MIN(k, ext_to), *q_offset - q_start);
    }
    if (ext_left < ext_to && (*q_offset + lut) < S->qlen) {
        int32_t k = 0, so = *s_offset + lut, qo = *q_offset + lut, ext_right;
        /* the compressed query pads past its end with zeros (= 'A') */
        while (k < 4) {
            int qb = (qo + k < S->qlen) ? (q[qo + k] & 3) : 0;
            if (qb != ORC_BASE(subj, so + k)) break;
            k++;
        }
        ext_right = ORC_MIN(ORC_MIN(k, s_range - so), q_range - qo);
        if (ext_left + ext_right < ext_to) return 0;
    }
    *q_offset -= ext_left; *s_offset -= ext_left;
    return 1;
}

/* CORE/blast_extend.c:259-315: qsort by score desc, s_start asc, length desc,
 * q_start asc; glibc qsort is a stable merge sort for arrays this small, so
 * full ties keep arrival order */
static int ihit_cmp(const OrcInitHit *a, const OrcInitHit *b)
{
    int r;
    if ((r = ORC_CMP(b->score, a->score)) != 0) return r;
    if ((r = ORC_CMP(a->s_start, b->s_start)) != 0) return r;
    if ((r = ORC_CMP(b->length, a->length)) != 0) return r;
    if ((r = ORC_CMP(a->q_start, b->q_start)) != 0) return r;
    return ORC_CMP(b->length, a->length);
}
static void ihit_sort(OrcInitHit *a, int32_t n)
{
    /* stable insertion/merge hybrid: n is tiny in practice */
    OrcInitHit *tmp; int32_t width, i;
    if (n < 2) return;
    tmp = (OrcInitHit *)malloc((size_t)n * sizeof(*tmp));
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int32_t l = i, m = ORC_MIN(i + width, n), r = ORC_MIN(i + 2 * width, n);
            int32_t a0 = l, b0 = m, k = l;
            while (a0 < m && b0 < r) tmp[k++] = (ihit_cmp(&a[b0], &a[a0]) < 0) ? a[b0++] : a[a0++];
            while (a0 < m) tmp[k++] = a[a0++];
            while (b0 < r) tmp[k++] = a[b0++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(*tmp));
    }
    free(tmp);
}

/* CORE/na_ungapped.c:1559-1657 (BlastNaWordFinder) */
void orc_word_finder(OrcSearch *S, const uint8_t *subj, int32_t slen, OrcStats *st)
{
    const OrcLookup *l = S->lut;
    const int32_t lut = l->lut_word_length, word = l->word_length, step = l->scan_step;
    const uint32_t mask = (uint32_t)l->ncells - 1;
    int32_t s_off, last = slen - lut, s_range = slen;
    int mode;   /* 0 direct, 1 na, 2 small, 3 small one-byte: CORE/na_ungapped.c:1753-1795 */
    if (lut == word) mode = 0;
    else if (l->type == ORC_LUT_SMALL_NA)
        mode = (lut % 4 == 0 && step % 4 == 0 && word - lut <= 4) ? 3 : 2;
    else mode = 1;

    if (!S->carry_diag || !S->carry_started) {
        if (S->container == ORC_DIAG_HASH) dhash_reset(dhash_get(S));
        else { memset(S->diag_last_hit, 0, (size_t)S->diag_len * sizeof(int32_t)); S->diag_offset = 0; }
        S->carry_started = 1;
    }

    /* stride 1 (blastn with a table as wide as the word): the reference's specialised scanners do not cut every word out of
     * the bytes anew, they roll it on by one base (CORE/blast_nascan.c:1943-2072 for lut 11, the accumulating `index` of
     * :361-445 for lut 8); `roll` holds the lut - 1 bases in front of s_off + lut - 1 */
    uint32_t roll = 0; const int rolling = step == 1 && last >= 0;
    if (rolling) { int32_t k; for (k = 0; k < lut - 1; k++) roll = (roll << 2) | ((subj[k >> 2] >> (2 * (3 - (k & 3)))) & 3u); }
    for (s_off = 0; s_off <= last; s_off += step) {
        uint32_t idx; int32_t nh, j;
        if (rolling) {
            const int32_t k = s_off + lut - 1;
            roll = ((roll << 2) | ((subj[k >> 2] >> (2 * (3 - (k & 3)))) & 3u)) & mask;
            idx = roll;
        } else
        {   /* the lookup word from the packed bytes with shifts and a mask, as the reference's scanners cut it
             * (CORE/blast_nascan.c:1489-1591; lut <= 12: at most 4 + 1 bytes; the caller pads the subject) */
            const uint8_t *sp = subj + (s_off >> 2);
            if (lut <= 12) {        /* four bytes hold the word at any offset: the load of s_MBScanSubject_Any (:1562-1568) */
                const uint32_t w = ((uint32_t)sp[0] << 24) | ((uint32_t)sp[1] << 16) | ((uint32_t)sp[2] << 8) | sp[3];
                idx = (w >> (2 * (16 - ((s_off & 3) + lut)))) & mask;
            } else {
                const uint64_t w = ((uint64_t)sp[0] << 32) | ((uint64_t)sp[1] << 24) | ((uint64_t)sp[2] << 16) | ((uint64_t)sp[3] << 8) | sp[4];
                idx = (uint32_t)(w >> (40 - 2 * (s_off & 3) - 2 * lut)) & mask;
            }
        }
        if (l->type == ORC_LUT_MB) {
            /* CORE/blast_nascan.c:1413-1427: presence bit first, then the chain, which yields descending q */
            int32_t qp;
            if (!(l->pv[idx >> l->pv_bts >> 5] & (1u << ((idx >> l->pv_bts) & 31)))) continue;
            qp = l->hashtable[idx];
            while (qp) {
                int32_t q = qp - 1, sb = s_off, ok = 1;
                st->lookup_hits++;
                if (mode == 1) ok = miniext_na(S, subj, s_range, &q, &sb, word, lut);
                if (ok) {
                    orc_push_seed(S, q, sb);
                    st->init_extends += diag_extend(S, subj, slen, q, sb, word);
                }
                qp = l->next_pos[qp];
            }
        } else {
            nh = l->cell_start[idx + 1] - l->cell_start[idx];
            for (j = 0; j < nh; j++) {
                int32_t q = l->cell_offs[l->cell_start[idx] + j], sb = s_off, ok = 1;
                st->lookup_hits++;
                if (mode == 1) ok = miniext_na(S, subj, s_range, &q, &sb, word, lut);
                else if (mode == 2) ok = miniext_small(S, subj, s_range, &q, &sb, word, lut);
                else if (mode == 3) ok = miniext_small_onebyte(S, subj, s_range, &q, &sb, word, lut);
                if (ok) {
                    orc_push_seed(S, q, sb);
                    st->init_extends += diag_extend(S, subj, slen, q, sb, word);
                }
            }
        }
    }
    st->good_init_extends += S->nihits;
    ihit_sort(S->ihits, S->nihits);
    if (S->carry_diag) {                /* Blast_ExtendWordExit (CORE/blast_extend.c:166-190), window 0 */
        if (S->container == ORC_DIAG_HASH) {
            DHash *h = dhash_get(S);
            if (orc_extend_word_exit(&h->offset, 0, slen, NULL, NULL, 0)) dhash_reset(h);
        } else orc_extend_word_exit(&S->diag_offset, 0, slen, S->diag_last_hit, NULL, S->diag_len);
    }
}

/* Blast_ExtendWordExit (CORE/blast_extend.c:166-190) with s_BlastDiagClear (:92-112): what happens to a diagonal
 * container when a subject is done and the container is carried on to the next one.  Positions stored in it are
 * subject offsets + *offset, so that everything the finished subject left is below anything the next one can produce;
 * past INT4_MAX / 4 the container is emptied instead and the offset starts again at the window.  Returns 1 when it
 * was emptied (the hash container's buckets are the caller's).  last_hit / flag: the array container's cells (NULL:
 * none).  Known answers: UT/blastdiag_unit_test.cpp:45-150. */
int orc_extend_word_exit(int32_t *offset, int32_t window, int32_t subject_length, int32_t *last_hit, uint32_t *flag, int32_t n)
{
    int32_t i;
    if (*offset >= INT32_MAX / 4) {
        *offset = window;
        for (i = 0; last_hit && i < n; i++) last_hit[i] = -window;
        for (i = 0; flag && i < n; i++) flag[i] = 0;
        return 1;
    }
    *offset += subject_length + window;
    return 0;
}

/* carry the diagonal container across subjects as the reference does (1), or start every subject with a fresh
 * one (0, the default and what the HIP path does -- DESIGN.md "a6") */
void orc_search_carry_diag(OrcSearch *S, int on) { S->carry_diag = on; S->carry_started = 0; }
