/* orc.h -- CPU ORACLE for the blastn/megablast preliminary-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gblastn_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * This is a plain-C restatement of the algorithm of the reference
 * (OpenHero/gblastn = NCBI-BLAST+ 2.2.28 C core, /root/reference/c++/src/algo/
 * blast/core, abbreviated CORE/ below).  Every function cites the reference
 * file:line it follows.  The reference itself is NOT buildable in the authoring
 * image without writing a stand-in for the configure-generated ncbiconf_unix.h
 * (or Apple's AvailabilityMacros.h for the in-tree Xcode config), so there is
 * no oracle/_ref; the restatement is pinned against the reference's own
 * known-answer tests instead (tests/test_oracle_golden.py):
 *   - UT/prelimsearch_unit_test.cpp:169-203  (nt.41646578 slice, megablast)
 *   - UT/bl2seq_unit_test.cpp:1620-1691      (greedy1a/b.fsa, score 619 / 6034)
 *   - UT/scoreblk_unit_test.cpp:313-530      (Karlin-Altschul tables, K=1/3)
 *   - UT/blasthits_unit_test.cpp             (purge / sort / odd-score rules)
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SENTINEL 15          /* BLASTNA gap code, CORE/blast_encoding.c:42-96 */
#define ORC_INT4_MIN (-2147483647-1)
#define ORC_INT4_MAX 2147483647

/* lookup table kinds, COREI/lookup_wrap.h (ELookupTableType) */
enum { ORC_LUT_SMALL_NA = 1, ORC_LUT_NA = 2, ORC_LUT_MB = 3 };
enum { ORC_DIAG_ARRAY = 0, ORC_DIAG_HASH = 1 };

typedef struct OrcOptions {
    int32_t word_size;          /* 28 megablast / 11 blastn */
    int32_t reward, penalty;    /* +1/-2 megablast, +2/-3 blastn */
    int32_t gap_open, gap_extend; /* 0/0 megablast (greedy linear), 5/2 blastn */
    int32_t greedy;             /* 1: eGreedyScoreOnly, 0: eDynProgScoreOnly */
    double  xdrop_ungap_bits;   /* 20 */
    double  gap_trigger_bits;   /* 27 */
    double  xdrop_gap_bits;     /* 25 greedy / 30 DP */
    double  xdrop_gap_final_bits; /* 100 */
    double  evalue;             /* 10 */
    int32_t min_diag_separation;/* 6 megablast / 50 blastn */
    int32_t hitlist_size;       /* 500 */
    int32_t cutoff_score;       /* 0 = derive from evalue */
    int32_t lut11_gblastn_rule; /* 1 = G-BLASTN's patched word_size==11 rule
                                   (CORE/blast_nalookup.c:127-144), 0 = stock */
    int64_t db_length;          /* total bases in the database (global) */
    int32_t db_num_seqs;        /* number of subjects (global); 0 => "bl2seq"
                                   mode: per-subject effective lengths */
} OrcOptions;

typedef struct OrcContext {
    int32_t query_offset;       /* in the concatenated query (sentinel-free index) */
    int32_t query_length;
    int32_t frame;              /* +1 / -1 */
    int32_t query_index;
    int32_t is_valid;
    int32_t length_adjustment;
    int64_t eff_searchsp;
    /* ungapped Karlin-Altschul block of this context */
    double  lambda_u, K_u, logK_u, H_u;
    /* per-context integer cutoffs */
    int32_t x_dropoff;          /* ungapped X (raw) */
    int32_t cutoff_score;       /* ungapped (gap trigger) cutoff */
    int32_t reduced_cutoff;     /* 0.9 * cutoff_score */
    int32_t gap_cutoff_score;   /* gapped cutoff (hit saving) */
    int32_t gap_cutoff_score_max;
} OrcContext;

typedef struct OrcSeed { int32_t q_off, s_off; } OrcSeed;

typedef struct OrcInitHit {     /* COREI/blast_extend.h:142-163 */
    int32_t q_off, s_off;       /* seed that produced it */
    int32_t q_start, s_start, length, score;
} OrcInitHit;

typedef struct OrcHSP {         /* COREI/blast_hits.h:93-126, prelim subset */
    int32_t context;
    int32_t q_offset, q_end, q_gapped_start;  /* context-relative */
    int32_t s_offset, s_end, s_gapped_start;
    int32_t score;
    double  evalue;
} OrcHSP;

typedef struct OrcStats {       /* COREI/blast_diagnostics.h */
    int64_t lookup_hits, init_extends, good_init_extends;
    int64_t gapped_extensions, good_extensions, seqs_passed;
} OrcStats;

typedef struct OrcSearch OrcSearch;

void orc_default_options(OrcOptions *o, int megablast);

/* Build everything that is per query batch: concatenated query with
 * sentinels and reverse strands, KA parameters, cutoffs, lookup table.
 * seqs[i] is BLASTNA (one code 0..15 per base), plus strand. */
OrcSearch *orc_search_new(const OrcOptions *opt, int nq,
                          const uint8_t *const *seqs, const int32_t *lens);
/* Same with soft query masks ("mask at hash", the blastn default for DUST / lower-case masks):
 * intervals [from, to] (inclusive) in plus-strand coordinates of query mq[k], sorted and
 * non-overlapping per query.  Masked stretches are left out of the lookup table only
 * (CORE/blast_filter.c:1019-1119 lookup segments; CORE/na_ungapped.c:459-587 word re-check). */
OrcSearch *orc_search_new_masked(const OrcOptions *opt, int nq,
                                 const uint8_t *const *seqs, const int32_t *lens,
                                 int32_t nmask, const int32_t *mq, const int32_t *mfrom, const int32_t *mto);
void orc_search_free(OrcSearch *s);

/* introspection of the set-up (pins the integers that gate every kernel) */
int32_t orc_num_contexts(const OrcSearch *s);
const OrcContext *orc_contexts(const OrcSearch *s);
int32_t orc_lut_type(const OrcSearch *s);
int32_t orc_lut_width(const OrcSearch *s);
int32_t orc_scan_step(const OrcSearch *s);
int32_t orc_diag_container(const OrcSearch *s);
int32_t orc_gap_x_dropoff(const OrcSearch *s);
int32_t orc_gap_x_dropoff_final(const OrcSearch *s);
double  orc_gap_lambda(const OrcSearch *s);
double  orc_gap_K(const OrcSearch *s);
int32_t orc_query_concat_len(const OrcSearch *s);
const uint8_t *orc_query_concat(const OrcSearch *s); /* points past 1st sentinel */

/* One subject through the whole preliminary path.
 * packed: NCBI2na, 4 bases/byte, base 0 in bits 7..6 (ceil(len/4) bytes,
 * plus >= 4 readable pad bytes).  Outputs are malloc'd arrays owned by the
 * OrcSearch and valid until the next call. */
/* the same for a subject the engine takes in chunks of max_len bases (MAX_DBSEQ_LEN: 200,000,000 in G-BLASTN's
 * build, COREI/blast_gapalign.h:54-55) overlapping by DBSEQ_CHUNK_OVERLAP, chunk lists merged as Blast_HSPListsMerge
 * does; only the HSPs are left to read */
int orc_search_subject_chunked(OrcSearch *s, const uint8_t *packed, int32_t len, int32_t max_len, OrcStats *stats);
/* 1: the diagonal container survives from one orc_search_subject call to the next, as in the reference
 * (Blast_ExtendWordExit, CORE/blast_extend.c:166-190); 0 (default): fresh per subject */
void orc_search_carry_diag(OrcSearch *s, int on);
int orc_search_subject(OrcSearch *s, const uint8_t *packed, int32_t len,
                       OrcStats *stats);
int32_t orc_num_seeds(const OrcSearch *s);
const OrcSeed *orc_seeds(const OrcSearch *s);        /* after mini-extension, scan order */
int32_t orc_num_init_hits(const OrcSearch *s);
const OrcInitHit *orc_init_hits(const OrcSearch *s); /* sorted as the reference sorts */
int32_t orc_num_hsps(const OrcSearch *s);
const OrcHSP *orc_hsps(const OrcSearch *s);          /* purged, sorted, e-valued, reaped */

/* ---- pieces exposed on their own for the known-answer tests ---- */
typedef struct OrcKarlin { double Lambda, K, logK, H; } OrcKarlin;
/* CORE/blast_stat.c:2673 over a (reward,penalty) matrix with the two letter
 * frequency vectors (16 entries each, BLASTNA order) */
int orc_karlin_ungapped(int reward, int penalty, const double *p1,
                        const double *p2, OrcKarlin *out);
int orc_karlin_ideal(int reward, int penalty, OrcKarlin *out);
/* CORE/blast_stat.c:3373-3424 (BLAST_GetNucleotideGapExistenceExtendParams); known answers UT/blastoptions_unit_test.cpp:184-231 */
int orc_nucl_gap_params(int reward, int penalty, int *gap_existence, int *gap_extension);
/* CORE/blast_parameters.c:422-470 (BlastExtensionParametersNew); known answers UT/blastoptions_unit_test.cpp:761-810 */
void orc_extension_params(double min_lambda, double gap_x_dropoff_bits, double gap_x_dropoff_final_bits,
                          int32_t *gap_x_dropoff, int32_t *gap_x_dropoff_final);
/* CORE/lookup_util.c:100-190: the (n, k) de Bruijn sequence, k^n letters */
void orc_debruijn(int32_t n, int32_t k, uint8_t *output);
/* the lookup table of one sequence, one strand, as the reference's lookup-table unit tests build it; out[12]: see orc_lookup.c */
int orc_lookup_probe(const OrcOptions *opt, const uint8_t *seq, int32_t len, int64_t *out);
/* positions of the concatenated query inside a word of the lookup table (cover[orc_query_concat_len]) */
void orc_search_indexed_cover(const OrcSearch *s, uint8_t *cover);
/* Blast_ExtendWordExit (CORE/blast_extend.c:166-190); known answers UT/blastdiag_unit_test.cpp:45-150 */
int orc_extend_word_exit(int32_t *offset, int32_t window, int32_t subject_length, int32_t *last_hit, uint32_t *flag, int32_t n);
/* CORE/blast_stat.c:3806 */
int orc_karlin_nucl_gapped(int gap_open, int gap_extend, int reward, int penalty,
                           const OrcKarlin *ungapped, OrcKarlin *out,
                           int *round_down);
/* CORE/blast_stat.c:3919 */
int orc_nucl_alpha_beta(int reward, int penalty, int gap_open, int gap_extend,
                        const OrcKarlin *ungapped, int gapped,
                        double *alpha, double *beta);
/* CORE/blast_stat.c:4994 */
int orc_length_adjustment(double K, double logK, double alpha_d_lambda,
                          double beta, int32_t query_length, int64_t db_length,
                          int32_t db_num_seqs, int32_t *length_adjustment);
/* CORE/blast_hits.c:2224 + :2734 + :1226 on a caller-supplied HSP array;
 * returns the new count */
int32_t orc_hsplist_purge_common_endpoints(OrcHSP *h, int32_t n);
void    orc_hsplist_sort_by_score(OrcHSP *h, int32_t n);
/* greedy score-only extension from (q_off,s_off), CORE/blast_gapalign.c:2619 */
int orc_greedy_extend(const uint8_t *query, int32_t qlen,
                      const uint8_t *subj_packed, int32_t slen,
                      int32_t q_off, int32_t s_off, int32_t xdrop,
                      int32_t reward, int32_t penalty,
                      int32_t gap_open, int32_t gap_extend, OrcHSP *out);

/* s_BlastDynProgNtGappedAlignment from (q_off, s_off), CORE/blast_gapalign.c:2762-2826 */
int orc_dynprog_extend(const uint8_t *query, int32_t qlen, const uint8_t *subj_packed, int32_t slen,
                       int32_t q_off, int32_t s_off, int32_t xdrop, int32_t reward, int32_t penalty,
                       int32_t gap_open, int32_t gap_extend, OrcHSP *out);
/* ---- traceback stage (orc_traceback.c) ---- */
typedef struct OrcEditScript { uint8_t *op; int32_t *num; int32_t size; } OrcEditScript;    /* op: 0 deletion (gap in query), 3 substitution, 6 insertion */
typedef struct OrcTbHSP {
    OrcHSP hsp;                 /* final coordinates, score, e-value */
    OrcEditScript esp;
    int32_t num_ident, align_length, gaps, gap_opens;
    double bit_score;
} OrcTbHSP;
/* Blast_TracebackFromHSPList + s_HSPListPostTracebackUpdate (CORE/blast_traceback.c:336-790, :278-334) for the
 * preliminary HSPs of ONE query (both strands) against one subject, sorted by score.  subject: BLASTNA codes,
 * one per base.  Returns the number of final HSPs in *out (orc_traceback_free). */
int32_t orc_traceback_hsp_list(const OrcSearch *S, const uint8_t *subject, int32_t subject_length,
                               const OrcHSP *in, int32_t nin, OrcTbHSP **out);
void orc_traceback_free(OrcTbHSP *h, int32_t n);
/* the aligners on their own (known-answer and definition-level tests) */
typedef struct OrcGapOut { int32_t q_start, q_stop, s_start, s_stop, score, seed_q, seed_s; } OrcGapOut;
int orc_tb_dynprog(const OrcSearch *S, int32_t context, const uint8_t *subject, int32_t subject_length,
                   int32_t q_start, int32_t s_start, int32_t x_dropoff, OrcGapOut *r, OrcEditScript *esp);
int orc_tb_greedy(const OrcSearch *S, int32_t context, const uint8_t *subject, int32_t subject_length,
                  int32_t q_start, int32_t s_start, int32_t x_dropoff, OrcGapOut *r, OrcEditScript *esp);
/* Blast_SemiGappedAlign, score only (CORE/blast_gapalign.c:745-937): A rows, B columns, letters A[a], B[b + 1]
 * forward / A[M - a], B[N - 1 - b] reversed */
int32_t orc_semi_gapped_score(const int32_t matrix[16][16], const uint8_t *A, const uint8_t *B, int32_t M, int32_t N,
                              int32_t *a_offset, int32_t *b_offset, int32_t x_dropoff, int32_t gap_open,
                              int32_t gap_extend, int reverse_sequence);
const int32_t *orc_matrix(const OrcSearch *S);      /* 16 x 16 */
void orc_esp_free(OrcEditScript *e);

#ifdef __cplusplus
}
#endif
/* ---- per-query top-N hit lists (the HSP stream's collector writer) ----
 * CORE/blast_hspstream.c:316-365, CORE/hspfilter_collector.c:86-170,
 * CORE/blast_hits.c:2924-2981 (orc_collect.c) */
typedef struct OrcCollector OrcCollector;
int orc_prelim_hitlist_size(int hitlist_size, int gapped);
OrcCollector *orc_collector_new(int32_t num_queries, int32_t hitlist_size, int gapped);
int orc_collector_write(OrcCollector *c, int32_t oid, const OrcHSP *h, int32_t n);
int64_t orc_collector_close(OrcCollector *c);      /* number of surviving (query, oid) lists */
int32_t orc_collector_list(const OrcCollector *c, int64_t i, int32_t *oid, int32_t *query, const OrcHSP **h);
void orc_collector_free(OrcCollector *c);

/* ---- symmetric DUST (orc_dust.c): low-complexity intervals [from, to] of a BLASTNA sequence,
 * merged as blastn merges them; returns their number (at most cap are written) ---- */
int32_t orc_dust(const uint8_t *seq, int32_t len, int level, int window, int linker,
                 int32_t *from, int32_t *to, int32_t cap);

#endif
