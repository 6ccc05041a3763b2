/* orc_collect.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see orc.h): the HSP
 * stream's collector writer, i.e. the per-query top-N hit lists the preliminary
 * stage leaves behind.
 *
 * Restates, in the reference's own order of operations:
 *   BlastHSPStreamWrite            CORE/blast_hspstream.c:316-365
 *   s_BlastHSPCollectorRun         CORE/hspfilter_collector.c:86-170
 *   Blast_HitListUpdate            CORE/blast_hits.c:2924-2981
 *   s_EvalueCompareHSPLists        CORE/blast_hits.c:2757-2788
 *   s_FuzzyEvalueComp              CORE/blast_hits.c:1238-1253
 *   s_Heapify / s_CreateHeap       CORE/blast_hits.c:1470-1521
 *   s_BlastHitListInsertHSPListInHeap  CORE/blast_hits.c:2876-2891
 *   SBlastHitsParametersNew        CORE/blast_hits.c:45-76 (prelim_hitlist_size)
 *   BlastHSPStreamClose            CORE/blast_hspstream.c:136-209 (read-out order)
 *
 * Not restated: low_score_perc feedback (CORE/blast_engine.c:1313-1319; the
 * option defaults to 0, CORE/blast_parameters.c:811-814) and hsp_num_max
 * (default unlimited).
 */
#include "orc_int.h"
#include <stdlib.h>
#include <string.h>

typedef struct CList {          /* BlastHSPList, COREI/blast_hits.h:150-166 */
    int32_t oid, query_index, n;
    OrcHSP *h;
    double best_evalue;
} CList;

typedef struct CHitList {       /* BlastHitList, COREI/blast_hits.h:169-184 */
    CList **a; int32_t count, max, current;
    double worst_evalue; int32_t low_score; int heapified;
} CHitList;

struct OrcCollector {
    int32_t nq, prelim_size;
    CHitList **hl;
    CList **sorted; int64_t nsorted;
};

int orc_prelim_hitlist_size(int hitlist_size, int gapped)       /* :45-76 */
{
    int p = hitlist_size;
    if (gapped) p = ORC_MIN(2 * p, p + 50);
    return ORC_MAX(p, 10);
}

static int fuzzy(double e1, double e2)          /* :1244-1253 */
{
    if (e1 < (1 - 1e-6) * e2) return -1;
    else if (e1 > (1 + 1e-6) * e2) return 1;
    return 0;
}

static int cmp_hsp_evalue(const OrcHSP *a, const OrcHSP *b)     /* :1263-1284 */
{
    int r = fuzzy(a->evalue, b->evalue);
    if (r) return r;
    return orc_score_compare_hsps(a, b);
}

static void sort_by_evalue(CList *l)            /* :1286-1306 */
{
    int32_t i, j;
    for (i = 0; i + 1 < l->n; i++) if (cmp_hsp_evalue(&l->h[i], &l->h[i + 1]) > 0) break;
    if (i + 1 >= l->n) return;
    /* qsort in the reference; insertion sort is stable like glibc's merge sort */
    for (i = 1; i < l->n; i++) {
        OrcHSP t = l->h[i];
        for (j = i; j > 0 && cmp_hsp_evalue(&t, &l->h[j - 1]) < 0; j--) l->h[j] = l->h[j - 1];
        l->h[j] = t;
    }
}

static int cmp_lists(const CList *h1, const CList *h2)          /* :2757-2788 */
{
    int r;
    if (h1->n == 0 && h2->n == 0) return 0;
    else if (h1->n == 0) return 1;
    else if (h2->n == 0) return -1;
    if ((r = fuzzy(h1->best_evalue, h2->best_evalue)) != 0) return r;
    if (h1->h[0].score > h2->h[0].score) return -1;
    if (h1->h[0].score < h2->h[0].score) return 1;
    return ORC_CMP(h2->oid, h1->oid);
}

/* :1470-1498 on an array of pointers; indices instead of char pointers */
static void heapify(CList **a, int64_t base, int64_t lim, int64_t last)
{
    int64_t left = 2 * base + 1, large;
    while (base <= lim) {
        if (left == last) large = left;
        else large = cmp_lists(a[left], a[left + 1]) >= 0 ? left : left + 1;
        if (cmp_lists(a[base], a[large]) < 0) {
            CList *t = a[base]; a[base] = a[large]; a[large] = t;
            base = large; left = 2 * base + 1;
        } else break;
    }
}

static void create_heap(CList **a, int64_t nel)                 /* :1503-1521 */
{
    int64_t i, lim, basef, base;
    if (nel < 2) return;
    lim = (nel - 2) / 2; basef = nel - 1;
    i = nel / 2;
    for (base = i - 1; i > 0; base--) { heapify(a, base, lim, basef); i--; }
}

static void clist_free(CList *l) { if (l) { free(l->h); free(l); } }

static int hitlist_update(CHitList *H, CList *l)                /* :2924-2981 */
{
    int32_t i;
    l->best_evalue = (double)INT32_MAX;                         /* s_BlastGetBestEvalue :1583-1593 */
    for (i = 0; i < l->n; i++) l->best_evalue = ORC_MIN(l->h[i].evalue, l->best_evalue);
    if (H->count < H->max) {
        if (H->current == H->count) {                           /* :2900-2920 */
            if (H->current <= 0) H->current = 100;
            else H->current = ORC_MIN(2 * H->current, H->max);
            H->a = (CList **)realloc(H->a, (size_t)H->current * sizeof(CList *));
        }
        H->a[H->count++] = l;
        H->worst_evalue = ORC_MAX(l->best_evalue, H->worst_evalue);
        H->low_score = ORC_MIN(l->h[0].score, H->low_score);
    } else {
        int order;
        sort_by_evalue(l);
        order = fuzzy(l->best_evalue, H->worst_evalue);
        if (order > 0 || (order == 0 && l->h[0].score < H->low_score)) {
            clist_free(l);
        } else {
            if (!H->heapified) {
                for (i = 0; i < H->count; i++) sort_by_evalue(H->a[i]);
                create_heap(H->a, H->count);
                H->heapified = 1;
            }
            clist_free(H->a[0]);                                /* :2876-2891 */
            H->a[0] = l;
            if (H->count >= 2) heapify(H->a, 0, H->count / 2 - 1, H->count - 1);
            H->worst_evalue = H->a[0]->best_evalue;
            H->low_score = H->a[0]->h[0].score;
        }
    }
    return 0;
}

OrcCollector *orc_collector_new(int32_t num_queries, int32_t hitlist_size, int gapped)
{
    OrcCollector *c = (OrcCollector *)calloc(1, sizeof(*c));
    c->nq = num_queries; c->prelim_size = orc_prelim_hitlist_size(hitlist_size, gapped);
    c->hl = (CHitList **)calloc((size_t)num_queries, sizeof(CHitList *));
    return c;
}

/* one subject's HSP list (sorted by score, as the preliminary stage leaves it);
 * context -> query is context / 2 for blastn (Blast_GetQueryIndexFromContext) */
int orc_collector_write(OrcCollector *c, int32_t oid, const OrcHSP *h, int32_t n)   /* collector :86-170 */
{
    int32_t i, q;
    CList **per;
    if (n <= 0) return 0;
    if (c->sorted) return -1;                                   /* closed, hspstream.c:332-335 */
    per = (CList **)calloc((size_t)c->nq, sizeof(CList *));
    for (i = 0; i < n; i++) {
        CList *l;
        q = h[i].context / 2;
        if (q < 0 || q >= c->nq) { free(per); return -1; }
        if (!(l = per[q])) {
            l = per[q] = (CList *)calloc(1, sizeof(CList));
            l->oid = oid; l->query_index = q; l->h = (OrcHSP *)malloc((size_t)n * sizeof(OrcHSP));
        }
        l->h[l->n++] = h[i];
    }
    for (q = 0; q < c->nq; q++) if (per[q]) {
        if (!c->hl[q]) {                                        /* Blast_HitListNew :2806-2815 */
            c->hl[q] = (CHitList *)calloc(1, sizeof(CHitList));
            c->hl[q]->max = c->prelim_size; c->hl[q]->low_score = INT32_MAX;
        }
        hitlist_update(c->hl[q], per[q]);
    }
    free(per);
    return 0;
}

static int by_oid_query(const void *x, const void *y)
{
    const CList *a = *(CList *const *)x, *b = *(CList *const *)y;
    if (a->oid != b->oid) return ORC_CMP(a->oid, b->oid);
    return ORC_CMP(a->query_index, b->query_index);
}

/* BlastHSPStreamClose: concatenate the surviving lists of all queries; the
 * reference sorts them by decreasing oid and reads from the end (ascending
 * oid; order among equal oids is its qsort's).  Here: (oid, query) ascending. */
int64_t orc_collector_close(OrcCollector *c)
{
    int64_t n = 0, k = 0; int32_t q, j;
    if (c->sorted) return c->nsorted;
    for (q = 0; q < c->nq; q++) if (c->hl[q]) n += c->hl[q]->count;
    c->sorted = (CList **)calloc((size_t)n + 1, sizeof(CList *));
    for (q = 0; q < c->nq; q++) if (c->hl[q])
        for (j = 0; j < c->hl[q]->count; j++) c->sorted[k++] = c->hl[q]->a[j];
    qsort(c->sorted, (size_t)n, sizeof(CList *), by_oid_query);
    c->nsorted = n;
    return n;
}

int32_t orc_collector_list(const OrcCollector *c, int64_t i, int32_t *oid, int32_t *query, const OrcHSP **h)
{
    const CList *l = c->sorted[i];
    *oid = l->oid; *query = l->query_index; *h = l->h;
    return l->n;
}

void orc_collector_free(OrcCollector *c)
{
    int32_t q, j;
    if (!c) return;
    for (q = 0; q < c->nq; q++) if (c->hl[q]) {
        for (j = 0; j < c->hl[q]->count; j++) clist_free(c->hl[q]->a[j]);
        free(c->hl[q]->a); free(c->hl[q]);
    }
    free(c->hl); free(c->sorted); free(c);
}
