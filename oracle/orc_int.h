/* orc_int.h -- ORACLE internals (test infrastructure, see orc.h). */
#ifndef ORC_INT_H
#define ORC_INT_H
#include "orc.h"

#define ORC_MIN(a,b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a,b) ((a) > (b) ? (a) : (b))
#define ORC_CMP(a,b) ((a) > (b) ? 1 : ((a) < (b) ? -1 : 0))  /* BLAST_CMP */
#define ORC_LN2 0.69314718055994530941723212145818

/* base at position pos of an NCBI2na-packed sequence (COREI/blast_util.h:55) */
#define ORC_BASE(p, pos) (((p)[(pos) >> 2] >> (2 * (3 - ((pos) & 3)))) & 3)

int  orc_gcd(int a, int b);
int  orc_gdb3(int *a, int *b, int *c);
long orc_nint(double x);
void orc_nucl_matrix(int reward, int penalty, int32_t m[16][16]);
void orc_std_nt_freq(double p[16]);
void orc_context_freq(const uint8_t *seq, int32_t len, double p[16]);
double orc_karlin_StoE(int32_t S, const OrcKarlin *kbp, int64_t searchsp);
int  orc_cutoffs(int32_t *S, double *E, const OrcKarlin *kbp, int64_t searchsp);

/* ---- lookup table (one struct covers the three reference kinds) ---- */
typedef struct OrcLookup {
    int32_t type;           /* ORC_LUT_* */
    int32_t word_length, lut_word_length, scan_step;
    int32_t ncells;         /* 4^lut */
    /* ORC_LUT_MB: CORE/blast_nalookup.c:831-937 */
    int32_t *hashtable;     /* 1-based head per cell, 0 = empty */
    int32_t *next_pos;      /* chain to earlier positions */
    /* ORC_LUT_SMALL_NA / ORC_LUT_NA: per-cell ascending offset lists */
    int32_t *cell_start;    /* ncells+1 prefix */
    int32_t *cell_offs;
    int32_t longest_chain;
    /* presence vector of the megablast table: one bit per 2^pv_bts cells (CORE/blast_nalookup.c:951-1004;
     * tested by the scanners before the table is touched, CORE/blast_nascan.c:1413-1461) */
    uint32_t *pv; int32_t pv_bts;
} OrcLookup;

typedef struct OrcSeg { int32_t left, right; } OrcSeg;     /* SSeqRange of a lookup segment */
OrcLookup *orc_lookup_new(const OrcOptions *opt, const uint8_t *query /*past sentinel*/,
                          int32_t nseg, const OrcSeg *seg);
/* lookup_callback of the three table kinds (CORE/na_ungapped.c:51-138): is q_pos in cell `index`? */
int orc_lookup_has(const OrcLookup *l, int32_t index, int32_t q_pos);
void orc_lookup_free(OrcLookup *l);

struct OrcSearch {
    OrcOptions opt;
    int32_t nq, nctx;
    OrcContext *ctx;
    uint8_t *qbuf;          /* [15] ctx0 [15] ctx1 [15] ... [15] */
    uint8_t *query;         /* qbuf + 1 */
    int32_t qlen;           /* length of the concatenation (no outer sentinels) */
    int32_t matrix[16][16];
    int32_t score_table[256];   /* CORE/blast_parameters.c:237-262 */
    OrcKarlin kbp_gap;      /* same for every context (nucleotide tables) */
    int round_down;
    int32_t gap_x_dropoff, gap_x_dropoff_final;
    int32_t container;      /* ORC_DIAG_* */
    OrcLookup *lut;
    OrcSeg *segs; int32_t nsegs;    /* lookup segments = complement of the query masks */
    int masked;                     /* lut->masked_locations != NULL */
    /* per-subject outputs */
    OrcSeed *seeds; int32_t nseeds, cseeds;
    OrcInitHit *ihits; int32_t nihits, cihits;
    OrcHSP *hsps; int32_t nhsps, chsps;
    /* diag containers (fresh per subject; see DESIGN.md on the stale-slot rule) */
    int32_t *diag_last_hit; int32_t diag_len, diag_mask;
    void *diag_hash;
    /* the diagonal container carried from subject to subject as the reference does (Blast_ExtendWordExit,
     * CORE/blast_extend.c:166-190) instead of starting fresh: off by default, see orc_search_carry_diag */
    int carry_diag, carry_started; int32_t diag_offset;
    /* a chunk of a long subject is being searched (orc_search_subject_chunked): the list stops after purge, odd-score
     * rounding and sort; e-values, the e-value reap and the per-sequence counters follow the merge of the chunk lists
     * (GB/gpu_blastn_pre_search_engine.cpp:460-548 vs :772-810) */
    int chunk_mode;
};

int  orc_context_of(const OrcSearch *s, int32_t q_off);   /* BSearchContextInfo */
void orc_setup_effective_lengths(OrcSearch *s, int64_t db_length, int32_t db_num_seqs);

/* word finder: scan + mini-extension + diagonal filter + ungapped extension */
void orc_word_finder(OrcSearch *s, const uint8_t *subj, int32_t slen, OrcStats *st);
/* gapped stage + post-processing */
void orc_gapped_stage(OrcSearch *s, const uint8_t *subj, int32_t slen, OrcStats *st);

void orc_push_seed(OrcSearch *s, int32_t q, int32_t sb);
void orc_push_ihit(OrcSearch *s, const OrcInitHit *h);
void orc_push_hsp(OrcSearch *s, const OrcHSP *h);

/* aligners */
typedef struct OrcGapResult {
    int32_t q_start, q_stop, s_start, s_stop, score;
    int32_t seed_q, seed_s;     /* greedy_query_seed_start etc. */
} OrcGapResult;
int orc_greedy_gapped(const uint8_t *query, const uint8_t *subj, int32_t qlen, int32_t slen,
                      int32_t q_off, int32_t s_off, int32_t X, int32_t reward, int32_t penalty,
                      int32_t gap_open, int32_t gap_extend, OrcGapResult *r);
int orc_dynprog_gapped(const int32_t matrix[16][16], const uint8_t *query, const uint8_t *subj,
                       int32_t qlen, int32_t slen, int32_t q_off, int32_t s_off, int32_t X,
                       int32_t gap_open, int32_t gap_extend, OrcGapResult *r);
/* orc_gapped.c: ScoreCompareHSPs, CORE/blast_hits.c:1182-1208 */
int orc_score_compare_hsps(const OrcHSP *a, const OrcHSP *b);

#endif
