/* orc_setup.c -- ORACLE (test infrastructure): per-batch set-up restated from
 * CORE/blast_setup.c, CORE/blast_parameters.c and the query layout shown in
 * UT/ntscan_unit_test.cpp:119-185. */
#include "orc_int.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_default_options(OrcOptions *o, int megablast)
{
    /* API/blast_nucl_options.cpp:108-234 */
    memset(o, 0, sizeof(*o));
    if (megablast) {
        o->word_size = 28; o->reward = 1; o->penalty = -2;
        o->gap_open = 0; o->gap_extend = 0; o->greedy = 1;
        o->xdrop_gap_bits = 25; o->min_diag_separation = 6;
    } else {
        o->word_size = 11; o->reward = 2; o->penalty = -3;
        o->gap_open = 5; o->gap_extend = 2; o->greedy = 0;
        o->xdrop_gap_bits = 30; o->min_diag_separation = 50;
    }
    o->xdrop_ungap_bits = 20; o->gap_trigger_bits = 27.0;
    o->xdrop_gap_final_bits = 100; o->evalue = 10.0;
    o->hitlist_size = 500; o->cutoff_score = 0;
    o->lut11_gblastn_rule = 1;
}

/* complement in BLASTNA (via NCBI4NA bit reversal, CORE/blast_encoding.c:42-78) */
static const uint8_t kComp[16] = { 3, 2, 1, 0, 5, 4, 7, 6, 8, 9, 13, 12, 11, 10, 14, 15 };

/* CORE/blast_query_info.c:220-236 (BSearchContextInfo) */
int orc_context_of(const OrcSearch *s, int32_t n)
{
    int32_t m, b = 0, e = s->nctx;
    while (b < e - 1) {
        m = (b + e) / 2;
        if (s->ctx[m].query_offset > n) e = m; else b = m;
    }
    return b;
}

/* CORE/blast_setup.c:638-775 (BLAST_CalcEffLengths), nucleotide branch */
void orc_setup_effective_lengths(OrcSearch *s, int64_t db_length, int32_t db_num_seqs)
{
    int i;
    const OrcOptions *o = &s->opt;
    if (db_length == 0) return;
    for (i = 0; i < s->nctx; i++) {
        OrcContext *c = &s->ctx[i];
        int32_t length_adjustment = 0;
        int64_t eff = 0;
        if (c->is_valid && c->query_length > 0) {
            double alpha = 0, beta = 0;
            OrcKarlin ku; ku.Lambda = c->lambda_u; ku.K = c->K_u; ku.logK = c->logK_u; ku.H = c->H_u;
            orc_nucl_alpha_beta(o->reward, o->penalty, o->gap_open, o->gap_extend,
                                &ku, 1, &alpha, &beta);
            orc_length_adjustment(s->kbp_gap.K, s->kbp_gap.logK, alpha / s->kbp_gap.Lambda,
                                  beta, c->query_length, db_length, db_num_seqs,
                                  &length_adjustment);
            {
                int64_t eff_db = db_length - ((int64_t)db_num_seqs * length_adjustment);
                if (eff_db <= 0) eff_db = 1;
                eff = eff_db * (c->query_length - length_adjustment);
            }
        }
        c->eff_searchsp = eff;
        c->length_adjustment = length_adjustment;
    }
}

/* CORE/blast_parameters.c:822-979 (BlastHitSavingParametersUpdate, the branch a
 * gapped blastn search takes) followed by :280-419
 * (BlastInitialWordParametersUpdate, gapped_calculation branch) */
static void update_cutoffs(OrcSearch *s)
{
    int i;
    const OrcOptions *o = &s->opt;
    for (i = 0; i < s->nctx; i++) {
        OrcContext *c = &s->ctx[i];
        int32_t new_cutoff = 1, gap_trigger = ORC_INT4_MAX;
        double evalue = o->evalue;
        if (!c->is_valid) {
            c->gap_cutoff_score = ORC_INT4_MAX;
            c->cutoff_score = ORC_INT4_MAX;
            continue;
        }
        if (o->cutoff_score > 0) {
            c->gap_cutoff_score = o->cutoff_score;
            c->gap_cutoff_score_max = o->cutoff_score;
        } else {
            orc_cutoffs(&new_cutoff, &evalue, &s->kbp_gap, c->eff_searchsp);
            c->gap_cutoff_score = new_cutoff;
            c->gap_cutoff_score_max = new_cutoff;
        }
        /* ungapped (initial word) cutoffs */
        if (c->lambda_u > 0 && c->K_u > 0 && c->H_u > 0)
            gap_trigger = (int32_t)((o->gap_trigger_bits * ORC_LN2 + c->logK_u) / c->lambda_u);
        new_cutoff = gap_trigger;
        new_cutoff *= 1;    /* scale_factor */
        new_cutoff = ORC_MIN(new_cutoff, c->gap_cutoff_score_max);
        c->cutoff_score = new_cutoff;
        c->reduced_cutoff = (int32_t)(0.9 * new_cutoff);
    }
}

void orc_update_for_subject(OrcSearch *s, int32_t subject_length)
{
    /* CORE/blast_setup.c:905-932 (BLAST_OneSubjectUpdateParameters) */
    orc_setup_effective_lengths(s, subject_length, 1);
    update_cutoffs(s);
}

/* CORE/blast_parameters.c:422-470 (BlastExtensionParametersNew): the gapped X-drop values from bits to raw scores with
 * the (smallest) gapped Lambda; the final one is never below the preliminary one.  Known answers:
 * UT/blastoptions_unit_test.cpp:761-810 (Lambda 1.30: 20 / 22 bits -> 10 / 11, 25 / 22 -> 13 / 13). */
void orc_extension_params(double min_lambda, double gap_x_dropoff_bits, double gap_x_dropoff_final_bits,
                          int32_t *gap_x_dropoff, int32_t *gap_x_dropoff_final)
{
    *gap_x_dropoff = (int32_t)(gap_x_dropoff_bits * ORC_LN2 / min_lambda);
    *gap_x_dropoff_final = (int32_t)ORC_MAX(gap_x_dropoff_final_bits * ORC_LN2 / min_lambda, (double)*gap_x_dropoff);
}

OrcSearch *orc_search_new(const OrcOptions *opt, int nq,
                          const uint8_t *const *seqs, const int32_t *lens)
{
    return orc_search_new_masked(opt, nq, seqs, lens, 0, NULL, NULL, NULL);
}

/* CORE/blast_filter.c:1019-1119 (BLAST_ComplementMaskLocations): the unmasked stretches of every
 * valid context in concatenated coordinates; masks are plus-strand coordinates, mirrored for the
 * minus strand (:1061-1075).  Degenerate segments (left > right) are kept, as there. */
static void lookup_segments(OrcSearch *s, int32_t nmask, const int32_t *mq, const int32_t *mfrom, const int32_t *mto)
{
    int c, cap = s->nctx + 2 * nmask + 2;
    s->segs = (OrcSeg *)malloc((size_t)cap * sizeof(OrcSeg)); s->nsegs = 0;
    for (c = 0; c < s->nctx; c++) {
        const OrcContext *x = &s->ctx[c];
        int32_t start_offset, end_offset, left = 0, right, k, k0 = -1, k1 = -1, first = 1, open = 1, step, kk;
        if (!x->is_valid) continue;
        start_offset = x->query_offset; end_offset = x->query_length + start_offset - 1;
        for (k = 0; k < nmask; k++) if (mq[k] == x->query_index) { if (k0 < 0) k0 = k; k1 = k; }
        if (k0 < 0) { s->segs[s->nsegs].left = start_offset; s->segs[s->nsegs++].right = end_offset; continue; }
        /* minus strand: the list is reversed first (:1061-1063) */
        step = (x->frame < 0) ? -1 : 1;
        for (kk = (step > 0 ? k0 : k1); kk >= k0 && kk <= k1; kk += step) {
            int32_t filter_start, filter_end;
            if (x->frame < 0) { filter_start = end_offset - mto[kk]; filter_end = end_offset - mfrom[kk]; }
            else { filter_start = start_offset + mfrom[kk]; filter_end = start_offset + mto[kk]; }
            if (first) {
                open = 1; first = 0;
                if (filter_start > start_offset) left = start_offset;
                else { left = filter_end + 1; continue; }
            }
            right = filter_start - 1;
            s->segs[s->nsegs].left = left; s->segs[s->nsegs++].right = right;
            if (filter_end >= end_offset) { open = 0; break; }
            left = filter_end + 1;
        }
        if (open) { s->segs[s->nsegs].left = left; s->segs[s->nsegs++].right = end_offset; }
    }
}

/* CORE/blast_nalookup.c:333-366 (s_SeqLocListInvert) != NULL, i.e. does any gap between the
 * lookup segments span more than 3 positions */
static int has_masked_locations(const OrcSearch *s)
{
    int32_t i, start = 0, stop;
    if (s->nsegs == 0) return 0;
    stop = ORC_MAX(0, s->segs[0].left - 1);
    if (stop - start > 2) return 1;
    for (i = 0; i < s->nsegs; i++) {
        start = s->segs[i].right + 1;
        stop = (i + 1 < s->nsegs) ? s->segs[i + 1].left - 1 : s->qlen - 1;
        if (stop - start > 2) return 1;
    }
    return 0;
}

OrcSearch *orc_search_new_masked(const OrcOptions *opt, int nq,
                                 const uint8_t *const *seqs, const int32_t *lens,
                                 int32_t nmask, const int32_t *mq, const int32_t *mfrom, const int32_t *mto)
{
    OrcSearch *s = (OrcSearch *)calloc(1, sizeof(*s));
    int i, c; int64_t total = 1; int32_t off;
    double stdp[16];
    OrcKarlin kfirst; int have_first = 0;
    s->opt = *opt; s->nq = nq; s->nctx = 2 * nq;
    s->ctx = (OrcContext *)calloc((size_t)s->nctx, sizeof(OrcContext));
    for (i = 0; i < nq; i++) total += 2 * ((int64_t)lens[i] + 1);
    s->qbuf = (uint8_t *)malloc((size_t)total + 8);
    memset(s->qbuf, ORC_SENTINEL, (size_t)total + 8);
    s->query = s->qbuf + 1;
    off = 0;
    for (i = 0; i < nq; i++) {
        int32_t L = lens[i], j;
        OrcContext *p = &s->ctx[2 * i], *m = &s->ctx[2 * i + 1];
        p->query_offset = off; p->query_length = L; p->frame = 1; p->query_index = i;
        memcpy(s->query + off, seqs[i], (size_t)L);
        off += L + 1;
        m->query_offset = off; m->query_length = L; m->frame = -1; m->query_index = i;
        for (j = 0; j < L; j++) s->query[off + j] = kComp[seqs[i][L - 1 - j] & 15];
        off += L + 1;
    }
    s->qlen = off - 1;      /* last context offset + length */

    orc_nucl_matrix(opt->reward, opt->penalty, s->matrix);
    /* CORE/blast_parameters.c:237-262 */
    for (i = 0; i < 256; i++) {
        int32_t sc = 0;
        if (i & 3) sc += opt->penalty; else sc += opt->reward;
        if ((i >> 2) & 3) sc += opt->penalty; else sc += opt->reward;
        if ((i >> 4) & 3) sc += opt->penalty; else sc += opt->reward;
        if (i >> 6) sc += opt->penalty; else sc += opt->reward;
        s->score_table[i] = sc;
    }

    /* ungapped KA block per context, CORE/blast_stat.c:2711-2807 */
    orc_std_nt_freq(stdp);
    for (c = 0; c < s->nctx; c++) {
        OrcContext *x = &s->ctx[c];
        double p[16]; OrcKarlin k;
        x->is_valid = 1;
        if (x->query_length <= 0) { x->is_valid = 0; continue; }
        orc_context_freq(s->query + x->query_offset, x->query_length, p);
        if (orc_karlin_ungapped(opt->reward, opt->penalty, p, stdp, &k)) {
            x->is_valid = 0; x->lambda_u = x->K_u = x->H_u = -1; continue;
        }
        x->lambda_u = k.Lambda; x->K_u = k.K; x->logK_u = k.logK; x->H_u = k.H;
        if (!have_first) { kfirst = k; have_first = 1; }
    }
    /* gapped block: table lookup, or a copy of the context's ungapped block in
     * the "infinite gap cost" regime (CORE/blast_setup.c:76-128).  The copy
     * case is per context in the reference; the tables cover every default
     * configuration, so one block is kept here and the copy case uses the
     * first valid context. */
    if (have_first) {
        if (orc_karlin_nucl_gapped(opt->gap_open, opt->gap_extend, opt->reward, opt->penalty,
                                   &kfirst, &s->kbp_gap, &s->round_down)) {
            orc_search_free(s);
            return NULL;
        }
    }
    orc_extension_params(s->kbp_gap.Lambda, opt->xdrop_gap_bits, opt->xdrop_gap_final_bits, &s->gap_x_dropoff, &s->gap_x_dropoff_final);
    /* x_dropoff per context, CORE/blast_parameters.c:203-225 */
    for (c = 0; c < s->nctx; c++) {
        OrcContext *x = &s->ctx[c];
        if (!x->is_valid) continue;
        x->x_dropoff = (int32_t)(1.0 * ceil(opt->xdrop_ungap_bits * ORC_LN2 / x->lambda_u));
    }
    if (opt->db_num_seqs > 0) {
        orc_setup_effective_lengths(s, opt->db_length, opt->db_num_seqs);
        update_cutoffs(s);
    }
    /* container choice, CORE/blast_parameters.c:174,227-232 */
    s->container = (s->qlen > 8000) ? ORC_DIAG_HASH : ORC_DIAG_ARRAY;
    if (s->container == ORC_DIAG_ARRAY) {
        /* CORE/blast_extend.c:47-73 with window_size 0 */
        int32_t n = 1;
        while (n < s->qlen) n <<= 1;
        s->diag_len = n; s->diag_mask = n - 1;
        s->diag_last_hit = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    }
    lookup_segments(s, nmask, mq, mfrom, mto);
    s->lut = orc_lookup_new(opt, s->query, s->nsegs, s->segs);
    /* lut->masked_locations (CORE/blast_nalookup.c:413-417, :582-586, :977-981), mask at hash on */
    s->masked = s->lut->word_length > s->lut->lut_word_length && has_masked_locations(s);
    return s;
}

void orc_search_free(OrcSearch *s)
{
    if (!s) return;
    orc_lookup_free(s->lut); free(s->segs);
    free(s->ctx); free(s->qbuf); free(s->seeds); free(s->ihits); free(s->hsps);
    free(s->diag_last_hit); free(s->diag_hash);
    free(s);
}

int32_t orc_num_contexts(const OrcSearch *s) { return s->nctx; }
const OrcContext *orc_contexts(const OrcSearch *s) { return s->ctx; }
int32_t orc_lut_type(const OrcSearch *s) { return s->lut ? s->lut->type : 0; }
int32_t orc_lut_width(const OrcSearch *s) { return s->lut ? s->lut->lut_word_length : 0; }
int32_t orc_scan_step(const OrcSearch *s) { return s->lut ? s->lut->scan_step : 0; }
int32_t orc_diag_container(const OrcSearch *s) { return s->container; }
int32_t orc_gap_x_dropoff(const OrcSearch *s) { return s->gap_x_dropoff; }
int32_t orc_gap_x_dropoff_final(const OrcSearch *s) { return s->gap_x_dropoff_final; }
double  orc_gap_lambda(const OrcSearch *s) { return s->kbp_gap.Lambda; }
double  orc_gap_K(const OrcSearch *s) { return s->kbp_gap.K; }
int32_t orc_query_concat_len(const OrcSearch *s) { return s->qlen; }
const uint8_t *orc_query_concat(const OrcSearch *s) { return s->query; }

int32_t orc_num_seeds(const OrcSearch *s) { return s->nseeds; }
const OrcSeed *orc_seeds(const OrcSearch *s) { return s->seeds; }
int32_t orc_num_init_hits(const OrcSearch *s) { return s->nihits; }
const OrcInitHit *orc_init_hits(const OrcSearch *s) { return s->ihits; }
int32_t orc_num_hsps(const OrcSearch *s) { return s->nhsps; }
const OrcHSP *orc_hsps(const OrcSearch *s) { return s->hsps; }

void orc_push_seed(OrcSearch *s, int32_t q, int32_t sb)
{
    if (s->nseeds == s->cseeds) {
        s->cseeds = s->cseeds ? 2 * s->cseeds : 1024;
        s->seeds = (OrcSeed *)realloc(s->seeds, (size_t)s->cseeds * sizeof(OrcSeed));
    }
    s->seeds[s->nseeds].q_off = q; s->seeds[s->nseeds].s_off = sb; s->nseeds++;
}
void orc_push_ihit(OrcSearch *s, const OrcInitHit *h)
{
    if (s->nihits == s->cihits) {
        s->cihits = s->cihits ? 2 * s->cihits : 256;
        s->ihits = (OrcInitHit *)realloc(s->ihits, (size_t)s->cihits * sizeof(OrcInitHit));
    }
    s->ihits[s->nihits++] = *h;
}
void orc_push_hsp(OrcSearch *s, const OrcHSP *h)
{
    if (s->nhsps == s->chsps) {
        s->chsps = s->chsps ? 2 * s->chsps : 64;
        s->hsps = (OrcHSP *)realloc(s->hsps, (size_t)s->chsps * sizeof(OrcHSP));
    }
    s->hsps[s->nhsps++] = *h;
}

void orc_update_for_subject(OrcSearch *s, int32_t subject_length);

int orc_search_subject(OrcSearch *s, const uint8_t *packed, int32_t len, OrcStats *stats)
{
    OrcStats local; memset(&local, 0, sizeof(local));
    s->nseeds = s->nihits = s->nhsps = 0;
    if (s->opt.db_num_seqs == 0 && !s->chunk_mode)  /* "db_length == 0" branch of the engine (a chunked sequence: once, by its caller) */
        orc_update_for_subject(s, len);
    orc_word_finder(s, packed, len, &local);
    if (s->nihits > 0) orc_gapped_stage(s, packed, len, &local);
    if (stats) {
        stats->lookup_hits += local.lookup_hits;
        stats->init_extends += local.init_extends;
        stats->good_init_extends += local.good_init_extends;
        stats->gapped_extensions += local.gapped_extensions;
        stats->good_extensions += local.good_extensions;
        stats->seqs_passed += local.seqs_passed;
    }
    return 0;
}

/* ---------------- a subject searched in chunks: s_GetNextSubjectChunk + the chunk loop of
 * s_BlastSearchEngineCore (CORE/blast_engine.c:218-262, :455-540) and Blast_HSPListsMerge with
 * s_BlastMergeTwoHSPs (CORE/blast_hits.c:2545-2716, :1337-1378), no hard or soft subject masks ---------------- */
#define ORC_CHUNK_OVERLAP 100      /* DBSEQ_CHUNK_OVERLAP, COREI/blast_hits.h:169 */
#define ORC_DIAG_CLOSE 10          /* OVERLAP_DIAG_CLOSE, CORE/blast_hits.c:1386 */
#define ORC_CONTAINED(a,b,c,d,e,f) (((a) <= (c) && (b) >= (c)) && ((d) <= (f) && (e) >= (f)))   /* CORE/blast_hits_priv.h:68 */

static int merge_two(OrcHSP *h1, const OrcHSP *h2)
{
    if (ORC_CONTAINED(h1->q_offset, h1->q_end, h2->q_offset, h1->s_offset, h1->s_end, h2->s_offset) ||
        ORC_CONTAINED(h1->q_offset, h1->q_end, h2->q_end, h1->s_offset, h1->s_end, h2->s_end)) {
        h1->q_offset = ORC_MIN(h1->q_offset, h2->q_offset); h1->s_offset = ORC_MIN(h1->s_offset, h2->s_offset);
        h1->q_end = ORC_MAX(h1->q_end, h2->q_end); h1->s_end = ORC_MAX(h1->s_end, h2->s_end);
        if (h2->score > h1->score) {
            h1->q_gapped_start = h2->q_gapped_start; h1->s_gapped_start = h2->s_gapped_start;
            h1->score = h2->score; h1->evalue = h2->evalue;     /* (the e-value belongs to the score) */
        }
        return 1;
    }
    return 0;
}

/* comb[0..*ncomb) += cur[0..ncur) of the chunk that starts at split_offset; comb has room for both */
static void lists_merge(OrcHSP *comb, int32_t *ncomb, OrcHSP *cur, int32_t ncur, int32_t split_offset, int32_t overlap)
{
    int32_t i1, i2, n1 = 0, n2 = 0, n = *ncomb;
    OrcHSP t;
    if (ncur == 0) return;
    if (n == 0) { memcpy(comb, cur, (size_t)ncur * sizeof(OrcHSP)); *ncomb = ncur; return; }
    for (i1 = 0; i1 < n; i1++)
        if (comb[i1].s_end > split_offset) { t = comb[n1]; comb[n1] = comb[i1]; comb[i1] = t; n1++; }
    for (i2 = 0; i2 < ncur; i2++)
        if (cur[i2].s_offset < split_offset + overlap) { t = cur[n2]; cur[n2] = cur[i2]; cur[i2] = t; n2++; }
    for (i1 = 0; i1 < n1; i1++)
        for (i2 = 0; i2 < n2; i2++) {
            int32_t end_diag, start_diag;
            if (cur[i2].score == ORC_INT4_MIN || comb[i1].context != cur[i2].context) continue;    /* (deleted) */
            end_diag = comb[i1].q_end - comb[i1].s_end; start_diag = cur[i2].q_offset - cur[i2].s_offset;
            if (abs(end_diag - start_diag) < ORC_DIAG_CLOSE && merge_two(&comb[i1], &cur[i2])) cur[i2].score = ORC_INT4_MIN;
        }
    for (i2 = 0; i2 < ncur; i2++) if (cur[i2].score != ORC_INT4_MIN) comb[n++] = cur[i2];
    *ncomb = n;
    orc_hsplist_sort_by_score(comb, n);
}

int orc_search_subject_chunked(OrcSearch *s, const uint8_t *packed, int32_t len, int32_t max_len, OrcStats *stats)
{
    OrcHSP *comb = NULL; int32_t ncomb = 0, ccomb = 0, next = 0, first = 1;
    if (max_len < 1000 || (max_len & 3)) return -1;
    /* parameters of a subject-by-subject search follow the SEQUENCE's length (GB/...engine.cpp:1283-1293, before the chunks) */
    if (s->opt.db_num_seqs == 0) orc_update_for_subject(s, len);
    s->chunk_mode = 1;
    while (next < len) {
        const int32_t offset = next;                /* a multiple of 4: residual 0 */
        int32_t clen, i;
        if (offset + max_len < len) { clen = max_len; next = offset + max_len - ORC_CHUNK_OVERLAP; }
        else { clen = len - offset; next = len; }
        orc_search_subject(s, packed + offset / 4, clen, stats);
        if (s->nhsps == 0) { first = 0; continue; }
        for (i = 0; i < s->nhsps; i++) {            /* Blast_HSPListAdjustOffsets */
            s->hsps[i].s_offset += offset; s->hsps[i].s_end += offset; s->hsps[i].s_gapped_start += offset;
        }
        if (ncomb + s->nhsps > ccomb) { ccomb = 2 * (ncomb + s->nhsps) + 16; comb = (OrcHSP *)realloc(comb, (size_t)ccomb * sizeof(OrcHSP)); }
        lists_merge(comb, &ncomb, s->hsps, s->nhsps, offset, offset == 0 ? 0 : ORC_CHUNK_OVERLAP);
        first = 0;
    }
    (void)first;
    s->chunk_mode = 0;
    {   /* e-values and the e-value reap on the merged list, the per-sequence counters (:772-810) */
        int32_t i, n = 0;
        for (i = 0; i < ncomb; i++) {
            comb[i].evalue = orc_karlin_StoE(comb[i].score, &s->kbp_gap, s->ctx[comb[i].context].eff_searchsp);
            if (comb[i].evalue > s->opt.evalue) continue;
            comb[n++] = comb[i];
        }
        ncomb = n;
        if (stats && ncomb > 0) { stats->seqs_passed++; stats->good_extensions += ncomb; }
    }
    if (ncomb > s->chsps) { s->chsps = ncomb; s->hsps = (OrcHSP *)realloc(s->hsps, (size_t)ncomb * sizeof(OrcHSP)); }
    if (ncomb) memcpy(s->hsps, comb, (size_t)ncomb * sizeof(OrcHSP));
    s->nhsps = ncomb; s->nseeds = 0; s->nihits = 0;
    free(comb);
    return 0;
}
