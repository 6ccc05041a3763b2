/* orc_lookup.c -- ORACLE (test infrastructure): query lookup tables restated
 * from CORE/blast_nalookup.c, CORE/blast_lookup.c, CORE/lookup_util.c. */
#include "orc_int.h"
#include <stdlib.h>
#include <string.h>

/* CORE/blast_nalookup.c:51-189 (BlastChooseNaLookupTable); the word_size==11
 * case is switchable between G-BLASTN's patch (:127-144) and the stock NCBI
 * thresholds kept there as comments (:131-137). */
static int choose_table(const OrcOptions *opt, int32_t entries, int32_t max_q_off,
                        int32_t *lut_width)
{
    int type;
    switch (opt->word_size) {
    case 4: case 5: case 6:
        type = ORC_LUT_SMALL_NA; *lut_width = opt->word_size; break;
    case 7:
        type = ORC_LUT_SMALL_NA; *lut_width = (entries < 250) ? 6 : 7; break;
    case 8:
        type = ORC_LUT_SMALL_NA; *lut_width = (entries < 8500) ? 7 : 8; break;
    case 9:
        if (entries < 1250) { *lut_width = 7; type = ORC_LUT_SMALL_NA; }
        else if (entries < 21000) { *lut_width = 8; type = ORC_LUT_SMALL_NA; }
        else { *lut_width = 9; type = ORC_LUT_MB; }
        break;
    case 10:
        if (entries < 1250) { *lut_width = 7; type = ORC_LUT_SMALL_NA; }
        else if (entries < 8500) { *lut_width = 8; type = ORC_LUT_SMALL_NA; }
        else if (entries < 18000) { *lut_width = 9; type = ORC_LUT_MB; }
        else { *lut_width = 10; type = ORC_LUT_MB; }
        break;
    case 11:
        if (entries < 12000) { *lut_width = 8; type = ORC_LUT_SMALL_NA; }
        else if (opt->lut11_gblastn_rule) { *lut_width = 11; type = ORC_LUT_MB; }
        else if (entries < 180000) { *lut_width = 10; type = ORC_LUT_MB; }
        else { *lut_width = 11; type = ORC_LUT_MB; }
        break;
    case 12:
        if (entries < 8500) { *lut_width = 8; type = ORC_LUT_SMALL_NA; }
        else if (entries < 18000) { *lut_width = 9; type = ORC_LUT_MB; }
        else if (entries < 60000) { *lut_width = 10; type = ORC_LUT_MB; }
        else if (entries < 900000) { *lut_width = 11; type = ORC_LUT_MB; }
        else { *lut_width = 12; type = ORC_LUT_MB; }
        break;
    default:
        if (entries < 8500) { *lut_width = 8; type = ORC_LUT_SMALL_NA; }
        else if (entries < 300000) { *lut_width = 11; type = ORC_LUT_MB; }
        else { *lut_width = 12; type = ORC_LUT_MB; }
        break;
    }
    if (type == ORC_LUT_SMALL_NA && (entries >= 32767 || max_q_off >= 32768))
        type = ORC_LUT_NA;
    return type;
}

/* enumerate indexable lut-words of one segment [from,to] (inclusive):
 * CORE/blast_lookup.c:87-137 and CORE/blast_nalookup.c:873-928 visit the
 * same set -- every lut-word inside the segment that has no ambiguity code,
 * provided the segment can hold a full word_size word. */
typedef void (*word_cb)(void *arg, int32_t cell, int32_t q_off);

static void for_each_word(const uint8_t *q, int32_t from, int32_t to, int32_t word,
                          int32_t lut, word_cb cb, void *arg)
{
    int32_t p, run = 0; uint32_t code = 0;
    const uint32_t mask = (lut == 16) ? 0xffffffffu : ((1u << (2 * lut)) - 1);
    if (word > to - from + 1) return;
    for (p = from; p <= to; p++) {
        uint8_t b = q[p];
        if (b & 0xfc) { run = 0; code = 0; continue; }
        code = ((code << 2) | b) & mask;
        if (++run >= lut) cb(arg, (int32_t)code, p - lut + 1);
    }
}

typedef struct { OrcLookup *l; int32_t *count; int32_t *fill; int pass; uint32_t *helper; } BuildCtx;

static void mb_add(void *arg, int32_t cell, int32_t q_off)
{
    OrcLookup *l = ((BuildCtx *)arg)->l;
    int32_t index = q_off + 1;          /* 1-based, :893-898 */
    /* the reference's estimate of the longest chain counts, per 2048 cells, the words that found their cell taken (:915-924) */
    if (l->hashtable[cell] != 0 && ((BuildCtx *)arg)->helper) ((BuildCtx *)arg)->helper[cell / 2048]++;
    l->next_pos[index] = l->hashtable[cell];
    l->hashtable[cell] = index;
    if (l->pv) l->pv[(uint32_t)cell >> l->pv_bts >> 5] |= 1u << (((uint32_t)cell >> l->pv_bts) & 31);   /* PV_SET, :921-923 */
}
static void na_add(void *arg, int32_t cell, int32_t q_off)
{
    BuildCtx *b = (BuildCtx *)arg;
    if (b->pass == 0) b->count[cell]++;
    else b->l->cell_offs[b->fill[cell]++] = q_off;
}

OrcLookup *orc_lookup_new(const OrcOptions *opt, const uint8_t *query,
                          int32_t nseg, const OrcSeg *seg)
{
    OrcLookup *l = (OrcLookup *)calloc(1, sizeof(*l));
    int32_t entries = 0, max_off = 0, c, lut_width = 0, qlen = 0;
    BuildCtx b;
    /* CORE/lookup_util.c:193-209 (EstimateNumTableEntries) over the lookup segments */
    for (c = 0; c < nseg; c++) {
        entries += seg[c].right - seg[c].left;
        max_off = ORC_MAX(max_off, seg[c].right);
        qlen = ORC_MAX(qlen, seg[c].right + 1);
    }
    l->type = choose_table(opt, entries, max_off, &lut_width);
    l->word_length = opt->word_size;
    l->lut_word_length = lut_width;
    l->scan_step = l->word_length - l->lut_word_length + 1;  /* :403, :572, :1018 */
    l->ncells = 1 << (2 * lut_width);
    b.l = l; b.count = NULL; b.fill = NULL; b.pass = 0; b.helper = NULL;
    if (l->type == ORC_LUT_MB) {
        l->hashtable = (int32_t *)calloc((size_t)l->ncells, sizeof(int32_t));
        l->next_pos = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
        {   /* size of the presence vector, CORE/blast_nalookup.c:951-1004 */
            const int32_t kTargetPVSize = 131072, kSmallQueryCutoff = 15000, kLargeQueryCutoff = 800000;
            int32_t pv_size = (l->ncells <= 8 * kTargetPVSize) ? (l->ncells >> 5) : kTargetPVSize / 4, bts = 0;
            if (entries <= kSmallQueryCutoff || entries >= kLargeQueryCutoff) pv_size /= 2;
            while ((1 << (bts + 1)) <= l->ncells / pv_size) bts++;          /* ilog2 */
            l->pv_bts = bts;
            l->pv = (uint32_t *)calloc((size_t)pv_size, 4);
        }
        if (l->ncells >= 2048) b.helper = (uint32_t *)calloc((size_t)l->ncells / 2048, sizeof(uint32_t));
        for (c = 0; c < nseg; c++)
            for_each_word(query, seg[c].left, seg[c].right, l->word_length, lut_width, mb_add, &b);
        {   /* :931-935: never below 2 */
            int32_t i; uint32_t longest = 2;
            for (i = 0; b.helper && i < l->ncells / 2048; i++) longest = ORC_MAX(longest, b.helper[i]);
            l->longest_chain = (int32_t)longest;
            free(b.helper);
        }
    } else {
        int32_t i, acc = 0, longest = 0, overflow_cells = 2;
        b.count = (int32_t *)calloc((size_t)l->ncells, sizeof(int32_t));
        for (b.pass = 0; b.pass < 2; b.pass++) {
            if (b.pass == 1) {
                l->cell_start = (int32_t *)malloc(((size_t)l->ncells + 1) * sizeof(int32_t));
                b.fill = (int32_t *)malloc((size_t)l->ncells * sizeof(int32_t));
                for (i = 0; i < l->ncells; i++) {
                    l->cell_start[i] = acc; b.fill[i] = acc; acc += b.count[i];
                    longest = ORC_MAX(longest, b.count[i]);
                    if (b.count[i] > 1) overflow_cells += b.count[i] + 1;
                }
                l->cell_start[l->ncells] = acc;
                l->cell_offs = (int32_t *)malloc(((size_t)acc + 1) * sizeof(int32_t));
            }
            for (c = 0; c < nseg; c++)
                for_each_word(query, seg[c].left, seg[c].right, l->word_length, lut_width, na_add, &b);
        }
        l->longest_chain = longest;
        /* CORE/blast_nalookup.c:234-238 + CORE/lookup_wrap.c:127-137: a small
         * table whose overflow array would not fit 15 bits becomes a standard
         * table */
        if (l->type == ORC_LUT_SMALL_NA && overflow_cells >= 32768) l->type = ORC_LUT_NA;
        free(b.count); free(b.fill);
    }
    return l;
}

int orc_lookup_has(const OrcLookup *l, int32_t index, int32_t q_pos)
{
    index &= l->ncells - 1;
    if (l->type == ORC_LUT_MB) {            /* s_MBLookup :51-74 */
        int32_t q_off = l->hashtable[index];
        ++q_pos;
        while (q_off) { if (q_off == q_pos) return 1; q_off = l->next_pos[q_off]; }
        return 0;
    } else {                                /* s_SmallNaLookup :82-104 / s_NaLookup :113-138 */
        int32_t i;
        for (i = l->cell_start[index]; i < l->cell_start[index + 1]; i++)
            if (l->cell_offs[i] == q_pos) return 1;
        return 0;
    }
}

void orc_lookup_free(OrcLookup *l)
{
    if (!l) return;
    free(l->hashtable); free(l->next_pos); free(l->cell_start); free(l->cell_offs); free(l->pv);
    free(l);
}

/* CORE/lookup_util.c:100-190 (fkm_output, fkm, debruijn): the (n, k) de Bruijn sequence as the concatenation, in
 * lexicographic order, of the Lyndon words whose length divides n (Fredricksen, Kessler, Maiorana); k^n letters 0..k-1
 * into `output`.  The reference's lookup-table unit tests index this sequence (UT/ntlookup_unit_test.cpp:147-170). */
void orc_debruijn(int32_t n, int32_t k, uint8_t *output)
{
    int32_t *a = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));     /* indexed from one */
    int64_t cursor = 0; int32_t i, j, p;
    for (p = 1, i = 1; n % p == 0 && i <= p; i++) output[cursor++] = (uint8_t)a[i];     /* fkm_output(a, n, 1) */
    i = n;
    do {
        a[i] = a[i] + 1;
        for (j = 1; j <= n - i; j++) a[j + i] = a[j];
        p = i;
        if (n % p == 0) for (j = 1; j <= p; j++) output[cursor++] = (uint8_t)a[j];
        i = n;
        while (a[i] == k - 1) i--;
    } while (i > 0);
    free(a);
}

/* Test access to the table builder the way the reference's lookup-table unit tests drive it: ONE sequence, one lookup
 * segment [0, len - 1], i.e. one strand (UT/ntlookup_unit_test.cpp:147-170, :468-556).  seq: BLASTNA codes with a
 * sentinel byte in front of seq[0] and one behind seq[len - 1].
 * out: [0] table type, [1] cells, [2] word length, [3] lookup word length, [4] scan step, [5] pv_array_bts (megablast
 * table; 5 = PV_ARRAY_BTS for the others), [6] longest chain, [7] cells with exactly one entry, [8] empty cells,
 * [9] nonzero next_pos entries (megablast table: chained words), [10] presence words that are not all ones,
 * [11] entries that would live in the overflow array (cells with more than NA_HITS_PER_CELL = 3 entries) */
int orc_lookup_probe(const OrcOptions *opt, const uint8_t *seq, int32_t len, int64_t *out)
{
    OrcSeg seg; OrcLookup *l; int32_t c, i;
    seg.left = 0; seg.right = len - 1;
    l = orc_lookup_new(opt, seq, 1, &seg);
    if (!l) return -1;
    for (i = 0; i < 12; i++) out[i] = 0;
    out[0] = l->type; out[1] = l->ncells; out[2] = l->word_length; out[3] = l->lut_word_length; out[4] = l->scan_step;
    out[6] = l->longest_chain;
    if (l->type == ORC_LUT_MB) {
        const int32_t pv_words = (l->ncells >> l->pv_bts) >> 5;
        out[5] = l->pv_bts;
        for (c = 0; c < l->ncells; c++) {
            const int32_t head = l->hashtable[c];
            if (!head) out[8]++; else if (!l->next_pos[head]) out[7]++;
        }
        for (i = 0; i <= len; i++) if (l->next_pos[i]) out[9]++;
        for (i = 0; i < pv_words; i++) if (l->pv[i] != 0xffffffffu) out[10]++;
    } else {
        out[5] = 5;
        for (c = 0; c < l->ncells; c++) {
            const int32_t n = l->cell_start[c + 1] - l->cell_start[c];
            if (n == 1) out[7]++; else if (n == 0) out[8]++;
            if (n > 3) out[11] += n;
        }
        for (c = 0; c < l->ncells; c += 32) {       /* PV_SET per cell, CORE/blast_nalookup.c:234-245 */
            uint32_t w = 0;
            for (i = 0; i < 32 && c + i < l->ncells; i++) if (l->cell_start[c + i + 1] > l->cell_start[c + i]) w |= 1u << i;
            if (w != 0xffffffffu) out[10]++;
        }
    }
    orc_lookup_free(l);
    return 0;
}

/* Which positions of the concatenated query lie inside a word of the table: cover[p] = 1.  With lut = word this is what
 * the reference's table still knows about the soft masks it was built with when it keeps no masked_locations
 * (CORE/blast_nalookup.c:413-417): the shim rebuilds the masks from it (gblastn_amd/shim: s_MasksFromTable, walking
 * the table the way the lookup callbacks of CORE/na_ungapped.c:51-138 do), tests/test_oracle_golden.py checks the idea. */
void orc_search_indexed_cover(const OrcSearch *S, uint8_t *cover)
{
    const OrcLookup *l = S->lut; int32_t c, i, k;
    for (i = 0; i < S->qlen; i++) cover[i] = 0;
    for (c = 0; c < l->ncells; c++) {
        if (l->type == ORC_LUT_MB) {
            for (i = l->hashtable[c]; i; i = l->next_pos[i])
                for (k = 0; k < l->lut_word_length && i - 1 + k < S->qlen; k++) cover[i - 1 + k] = 1;
        } else {
            for (i = l->cell_start[c]; i < l->cell_start[c + 1]; i++)
                for (k = 0; k < l->lut_word_length && l->cell_offs[i] + k < S->qlen; k++) cover[l->cell_offs[i] + k] = 1;
        }
    }
}
