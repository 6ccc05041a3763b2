/* orc_gapped.c -- ORACLE (test infrastructure): BLAST_GetGappedScore with its
 * interval-tree containment test, and the per-subject HSP post-processing,
 * restated from CORE/blast_gapalign.c:3233-3559, CORE/blast_itree.c,
 * CORE/blast_hits.c and the engine glue at
 * GB/gpu_blastn_pre_search_engine.cpp:416-550,611-825. */
#include "orc_int.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- interval tree (CORE/blast_itree.c) ---------------- */
typedef struct { int32_t leftend, rightend, leftptr, midptr, rightptr, hsp; } INode;
/* hsp: -1 for internal nodes, else index into the HSP pool; for a leaf,
 * leftptr holds the strand start offset of its query (blast_itree.c:573-575) */
typedef struct {
    INode *n; int32_t used, alloc, s_min, s_max;
    const OrcHSP *pool;     /* tree HSPs live in the OrcSearch hsp array */
    const OrcSearch *S;
} ITree;
enum { DIR_LEFT, DIR_RIGHT, DIR_NONE };

static int32_t inode_new(ITree *t, int32_t parent, int dir)      /* :62-117 */
{
    int32_t idx, midpt;
    if (t->used == t->alloc) {
        t->alloc *= 2;
        t->n = (INode *)realloc(t->n, (size_t)t->alloc * sizeof(INode));
    }
    idx = t->used++;
    if (dir == DIR_NONE) return idx;
    t->n[idx].leftptr = t->n[idx].midptr = t->n[idx].rightptr = 0;
    t->n[idx].hsp = -1;
    midpt = (t->n[parent].leftend + t->n[parent].rightend) / 2;
    if (dir == DIR_LEFT) { t->n[idx].leftend = t->n[parent].leftend; t->n[idx].rightend = midpt; }
    else { t->n[idx].leftend = midpt + 1; t->n[idx].rightend = t->n[parent].rightend; }
    return idx;
}
static int32_t iroot_new(ITree *t, int32_t a, int32_t b)         /* :129-148 */
{
    int32_t idx = inode_new(t, 0, DIR_NONE);
    t->n[idx].leftptr = t->n[idx].midptr = t->n[idx].rightptr = 0;
    t->n[idx].hsp = -1; t->n[idx].leftend = a; t->n[idx].rightend = b;
    return idx;
}
static void itree_init(ITree *t, const OrcSearch *S, int32_t q_end, int32_t s_end)  /* :151-184 */
{
    t->alloc = 100; t->used = 0;
    t->n = (INode *)malloc((size_t)t->alloc * sizeof(INode));
    t->s_min = 0; t->s_max = s_end; t->S = S; t->pool = NULL;
    iroot_new(t, 0, q_end);
}
/* :218-234 (s_GetQueryStrandOffset) */
static int32_t strand_offset(const OrcSearch *S, int32_t context)
{
    int32_t c = context;
    while (c) {
        int32_t f = S->ctx[c].frame, fp = S->ctx[c - 1].frame;
        if (f == 0 || ((f > 0) != (fp > 0))) break;
        c--;
    }
    return S->ctx[c].query_offset;
}
/* :248-306 (s_HSPsHaveCommonEndpoint): 0 none, 1 keep input, 2 keep tree.
 * Subject frames are all +1 for blastn. */
static int common_endpoint(const OrcHSP *in, int32_t in_q, const OrcHSP *tr, int32_t tr_q, int which)
{
    int match;
    if (in_q != tr_q) return 0;
    if (which == DIR_LEFT) match = in->q_offset == tr->q_offset && in->s_offset == tr->s_offset;
    else match = in->q_end == tr->q_end && in->s_end == tr->s_end;
    if (!match) return 0;
    if (in->score > tr->score) return 1;
    if (in->score < tr->score) return 2;
    {
        int32_t a = in->q_end - in->q_offset, b = tr->q_end - tr->q_offset;
        if (a > b) return 2;
        if (a < b) return 1;
        a = in->s_end - in->s_offset; b = tr->s_end - tr->s_offset;
        if (a > b) return 2;
        if (a < b) return 1;
    }
    return 2;
}
/* :324-414 (s_MidpointTreeHasHSPEndpoint) */
static int midtree_has_endpoint(ITree *t, int32_t root, const OrcHSP *in, int32_t in_q, int which)
{
    int32_t rootn = root, tmp, listn, nextn, target, midpt;
    target = (which == DIR_LEFT) ? in->s_offset : in->s_end;
    for (;;) {
        tmp = t->n[rootn].midptr;
        listn = rootn; nextn = tmp;
        while (tmp != 0) {
            int best = common_endpoint(in, in_q, &t->pool[t->n[nextn].hsp], t->n[nextn].leftptr, which);
            tmp = t->n[nextn].midptr;
            if (best == 2) return 1;
            else if (best == 1) t->n[listn].midptr = tmp;
            listn = nextn;
            nextn = tmp;
        }
        tmp = 0;
        midpt = (t->n[rootn].leftend + t->n[rootn].rightend) / 2;
        if (target < midpt) tmp = t->n[rootn].leftptr;
        else if (target > midpt) tmp = t->n[rootn].rightptr;
        if (tmp == 0) return 0;
        nextn = tmp;
        if (t->n[nextn].hsp >= 0) {
            int best = common_endpoint(in, in_q, &t->pool[t->n[nextn].hsp], t->n[nextn].leftptr, which);
            if (best == 2) return 1;
            else if (best == 1) {
                if (target < midpt) t->n[rootn].leftptr = 0;
                else if (target > midpt) t->n[rootn].rightptr = 0;
                return 0;
            }
            break;
        }
        rootn = nextn;
    }
    return 0;
}
/* :432-504 (s_IntervalTreeHasHSPEndpoint) */
static int tree_has_endpoint(ITree *t, const OrcHSP *in, int32_t in_q, int which)
{
    int32_t rootn = 0, tmp, nextn, target, midpt;
    target = (which == DIR_LEFT) ? in_q + in->q_offset : in_q + in->q_end;
    for (;;) {
        tmp = t->n[rootn].midptr;
        if (tmp != 0 && midtree_has_endpoint(t, tmp, in, in_q, which)) return 1;
        tmp = 0;
        midpt = (t->n[rootn].leftend + t->n[rootn].rightend) / 2;
        if (target < midpt) tmp = t->n[rootn].leftptr;
        else if (target > midpt) tmp = t->n[rootn].rightptr;
        if (tmp == 0) return 0;
        nextn = tmp;
        if (t->n[nextn].hsp >= 0) {
            int best = common_endpoint(in, in_q, &t->pool[t->n[nextn].hsp], t->n[nextn].leftptr, which);
            if (best == 2) return 1;
            else if (best == 1) {
                if (target < midpt) t->n[rootn].leftptr = 0;
                else if (target > midpt) t->n[rootn].rightptr = 0;
                return 0;
            }
            break;
        }
        rootn = nextn;
    }
    return 0;
}
/* :508-798 (BlastIntervalTreeAddHSP, eQueryAndSubject) */
static void itree_add(ITree *t, int32_t hsp_idx)
{
    const OrcHSP *hsp = &t->pool[hsp_idx];
    int32_t query_start = strand_offset(t->S, hsp->context);
    int32_t region_start = query_start + hsp->q_offset, region_end = query_start + hsp->q_end;
    int32_t root = 0, newi, mid, old, middle, mid2, ors, ore;
    int which_half, index_subject = 0;
    if (tree_has_endpoint(t, hsp, query_start, DIR_LEFT)) return;
    if (tree_has_endpoint(t, hsp, query_start, DIR_RIGHT)) return;
    newi = inode_new(t, 0, DIR_NONE);
    t->n[newi].leftptr = query_start; t->n[newi].midptr = 0; t->n[newi].rightptr = 0;
    t->n[newi].hsp = hsp_idx; t->n[newi].leftend = t->n[newi].rightend = 0;
    for (;;) {
        middle = (t->n[root].leftend + t->n[root].rightend) / 2;
        if (region_end < middle) {
            if (t->n[root].leftptr == 0) { t->n[root].leftptr = newi; return; }
            old = t->n[root].leftptr;
            if (t->n[old].hsp < 0) { root = old; continue; }
            which_half = DIR_LEFT;
        } else if (region_start > middle) {
            if (t->n[root].rightptr == 0) { t->n[root].rightptr = newi; return; }
            old = t->n[root].rightptr;
            if (t->n[old].hsp < 0) { root = old; continue; }
            which_half = DIR_RIGHT;
        } else {
            if (index_subject) {
                t->n[newi].midptr = t->n[root].midptr;
                t->n[root].midptr = newi;
                return;
            }
            index_subject = 1;
            if (t->n[root].midptr == 0) {
                mid = iroot_new(t, t->s_min, t->s_max);
                t->n[root].midptr = mid;
            }
            root = t->n[root].midptr;
            region_start = hsp->s_offset; region_end = hsp->s_end;
            continue;
        }
        mid = inode_new(t, root, which_half);
        if (which_half == DIR_LEFT) t->n[root].leftptr = mid; else t->n[root].rightptr = mid;
        {
            const OrcHSP *oh = &t->pool[t->n[old].hsp];
            if (index_subject) { ors = oh->s_offset; ore = oh->s_end; }
            else { ors = t->n[old].leftptr + oh->q_offset; ore = t->n[old].leftptr + oh->q_end; }
            root = mid;
            middle = (t->n[root].leftend + t->n[root].rightend) / 2;
            if (ore < middle) t->n[mid].leftptr = old;
            else if (ors > middle) t->n[mid].rightptr = old;
            else if (index_subject) t->n[mid].midptr = old;
            else {
                mid2 = iroot_new(t, t->s_min, t->s_max);
                ors = oh->s_offset; ore = oh->s_end;
                t->n[mid].midptr = mid2;
                middle = (t->n[mid2].leftend + t->n[mid2].rightend) / 2;
                if (ore < middle) t->n[mid2].leftptr = old;
                else if (ors > middle) t->n[mid2].rightptr = old;
                else t->n[mid2].midptr = old;
            }
        }
    }
}
/* :814-852 (s_HSPIsContained) */
static int hsp_contained(const OrcHSP *in, int32_t in_q, const OrcHSP *tr, int32_t tr_q, int32_t mds)
{
    if (in_q != tr_q) return 0;
    if (in->score <= tr->score &&
        (tr->q_offset <= in->q_offset && tr->q_end >= in->q_offset &&
         tr->s_offset <= in->s_offset && tr->s_end >= in->s_offset) &&
        (tr->q_offset <= in->q_end && tr->q_end >= in->q_end &&
         tr->s_offset <= in->s_end && tr->s_end >= in->s_end)) {
        int32_t d1, d2;
        if (mds == 0) return 1;
        d1 = (tr->q_offset - tr->s_offset) - (in->q_offset - in->s_offset); if (d1 < 0) d1 = -d1;
        d2 = (tr->q_end - tr->s_end) - (in->q_end - in->s_end); if (d2 < 0) d2 = -d2;
        if (d1 < mds || d2 < mds) return 1;
    }
    return 0;
}
/* :871-934 (s_MidpointTreeContainsHSP) */
static int midtree_contains(const ITree *t, int32_t root, const OrcHSP *in, int32_t in_q, int32_t mds)
{
    int32_t node = root, tmp, middle;
    while (t->n[node].hsp < 0) {
        tmp = t->n[node].midptr;
        while (tmp != 0) {
            if (hsp_contained(in, in_q, &t->pool[t->n[tmp].hsp], t->n[tmp].leftptr, mds)) return 1;
            tmp = t->n[tmp].midptr;
        }
        tmp = 0;
        middle = (t->n[node].leftend + t->n[node].rightend) / 2;
        if (in->s_end < middle) tmp = t->n[node].leftptr;
        else if (in->s_offset > middle) tmp = t->n[node].rightptr;
        if (tmp == 0) return 0;
        node = tmp;
    }
    return hsp_contained(in, in_q, &t->pool[t->n[node].hsp], t->n[node].leftptr, mds);
}
/* :936-1000 (BlastIntervalTreeContainsHSP) */
static int itree_contains(const ITree *t, const OrcHSP *hsp, int32_t mds)
{
    int32_t node = 0, tmp, middle;
    int32_t query_start = strand_offset(t->S, hsp->context);
    int32_t region_start = query_start + hsp->q_offset, region_end = query_start + hsp->q_end;
    while (t->n[node].hsp < 0) {
        tmp = t->n[node].midptr;
        if (tmp > 0 && midtree_contains(t, tmp, hsp, query_start, mds)) return 1;
        tmp = 0;
        middle = (t->n[node].leftend + t->n[node].rightend) / 2;
        if (region_end < middle) tmp = t->n[node].leftptr;
        else if (region_start > middle) tmp = t->n[node].rightptr;
        if (tmp == 0) return 0;
        node = tmp;
    }
    return hsp_contained(hsp, query_start, &t->pool[t->n[node].hsp], t->n[node].leftptr, mds);
}

/* the tree for the traceback stage (orc_traceback.c): the HSP pool is the caller's array */
void *orc_itree_new(const OrcSearch *S, int32_t q_end, int32_t s_end)
{
    ITree *t = (ITree *)calloc(1, sizeof(ITree));
    itree_init(t, S, q_end, s_end);
    return t;
}
void orc_itree_reset(void *tv, int32_t q_end, int32_t s_end)     /* Blast_IntervalTreeReset */
{
    ITree *t = (ITree *)tv;
    t->used = 0; t->s_max = s_end;
    iroot_new(t, 0, q_end);
}
int orc_itree_contains_hsp(void *tv, const OrcHSP *pool, const OrcHSP *hsp, int32_t mds)
{
    ITree *t = (ITree *)tv;
    t->pool = pool;
    return itree_contains(t, hsp, mds);
}
void orc_itree_add_hsp(void *tv, const OrcHSP *pool, int32_t idx)
{
    ITree *t = (ITree *)tv;
    t->pool = pool;
    itree_add(t, idx);
}
void orc_itree_free(void *tv) { ITree *t = (ITree *)tv; free(t->n); free(t); }

/* ---------------- HSP list post-processing (CORE/blast_hits.c) ---------------- */
static int cmp_qoff(const void *v1, const void *v2)     /* :2037-2090 */
{
    const OrcHSP *a = (const OrcHSP *)v1, *b = (const OrcHSP *)v2;
    if (a->context < b->context) return -1;
    if (a->context > b->context) return 1;
    if (a->q_offset < b->q_offset) return -1;
    if (a->q_offset > b->q_offset) return 1;
    if (a->s_offset < b->s_offset) return -1;
    if (a->s_offset > b->s_offset) return 1;
    if (a->score < b->score) return 1;
    if (a->score > b->score) return -1;
    if (a->q_end < b->q_end) return 1;
    if (a->q_end > b->q_end) return -1;
    if (a->s_end < b->s_end) return 1;
    if (a->s_end > b->s_end) return -1;
    return 0;
}
static int cmp_qend(const void *v1, const void *v2)     /* :2102-2160 */
{
    const OrcHSP *a = (const OrcHSP *)v1, *b = (const OrcHSP *)v2;
    if (a->context < b->context) return -1;
    if (a->context > b->context) return 1;
    if (a->q_end < b->q_end) return -1;
    if (a->q_end > b->q_end) return 1;
    if (a->s_end < b->s_end) return -1;
    if (a->s_end > b->s_end) return 1;
    if (a->score < b->score) return 1;
    if (a->score > b->score) return -1;
    if (a->q_offset < b->q_offset) return 1;
    if (a->q_offset > b->q_offset) return -1;
    if (a->s_offset < b->s_offset) return 1;
    if (a->s_offset > b->s_offset) return -1;
    return 0;
}
static int cmp_score(const void *v1, const void *v2)    /* :1182-1208 (ScoreCompareHSPs) */
{
    const OrcHSP *a = (const OrcHSP *)v1, *b = (const OrcHSP *)v2;
    int r;
    if ((r = ORC_CMP(b->score, a->score)) != 0) return r;
    if ((r = ORC_CMP(a->s_offset, b->s_offset)) != 0) return r;
    if ((r = ORC_CMP(b->s_end, a->s_end)) != 0) return r;
    if ((r = ORC_CMP(a->q_offset, b->q_offset)) != 0) return r;
    return ORC_CMP(b->q_end, a->q_end);
}
/* stable merge sort: what glibc's qsort does for arrays that fit its buffer */
static void stable_sort(OrcHSP *a, int32_t n, int (*cmp)(const void *, const void *))
{
    OrcHSP *tmp; int32_t width, i;
    if (n < 2) return;
    tmp = (OrcHSP *)malloc((size_t)n * sizeof(*tmp));
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int32_t l = i, m = ORC_MIN(i + width, n), r = ORC_MIN(i + 2 * width, n);
            int32_t a0 = l, b0 = m, k = l;
            while (a0 < m && b0 < r) tmp[k++] = (cmp(&a[b0], &a[a0]) < 0) ? a[b0++] : a[a0++];
            while (a0 < m) tmp[k++] = a[a0++];
            while (b0 < r) tmp[k++] = a[b0++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(*tmp));
    }
    free(tmp);
}
/* :2224-2302 with purge = TRUE */
int32_t orc_hsplist_purge_common_endpoints(OrcHSP *h, int32_t n)
{
    int32_t i, j, k;
    if (n == 0) return 0;
    stable_sort(h, n, cmp_qoff);
    i = 0;
    while (i < n) {
        j = 1;
        while (i + j < n && h[i].context == h[i + j].context &&
               h[i].q_offset == h[i + j].q_offset && h[i].s_offset == h[i + j].s_offset) {
            n--;
            for (k = i + j; k < n; k++) h[k] = h[k + 1];
        }
        i += j;
    }
    stable_sort(h, n, cmp_qend);
    i = 0;
    while (i < n) {
        j = 1;
        while (i + j < n && h[i].context == h[i + j].context &&
               h[i].q_end == h[i + j].q_end && h[i].s_end == h[i + j].s_end) {
            n--;
            for (k = i + j; k < n; k++) h[k] = h[k + 1];
        }
        i += j;
    }
    return n;
}
int orc_score_compare_hsps(const OrcHSP *a, const OrcHSP *b) { return cmp_score(a, b); }
void orc_hsplist_sort_by_score(OrcHSP *h, int32_t n)   /* :1226-1236 */
{
    int32_t i; int sorted = 1;
    for (i = 0; i + 1 < n; i++) if (cmp_score(&h[i], &h[i + 1]) > 0) { sorted = 0; break; }
    if (!sorted) stable_sort(h, n, cmp_score);
}

/* ---------------- BLAST_GetGappedScore ---------------- */
void orc_gapped_stage(OrcSearch *S, const uint8_t *subj, int32_t slen, OrcStats *st)
{
    ITree tree; int32_t index;
    const OrcOptions *o = &S->opt;
    itree_init(&tree, S, S->qlen + 1, slen + 1);
    S->nhsps = 0;
    for (index = 0; index < S->nihits; index++) {
        OrcInitHit ih = S->ihits[index];
        /* :2421-2443: context from the seed's q_off; make offsets context-local */
        int32_t context = orc_context_of(S, ih.q_off);
        int32_t qstart = S->ctx[context].query_offset, qlen = S->ctx[context].query_length;
        const uint8_t *q = S->query + qstart;
        OrcHSP tmp; OrcGapResult r; int32_t cutoff, s_end; int rc;
        ih.q_off -= qstart; ih.q_start -= qstart;
        memset(&tmp, 0, sizeof(tmp));
        tmp.score = ih.score; tmp.context = context;
        tmp.q_offset = ih.q_start; tmp.q_end = ih.q_start + ih.length;
        tmp.s_offset = ih.s_start; tmp.s_end = ih.s_start + ih.length;
        s_end = tmp.s_end;
        tree.pool = S->hsps;
        if (itree_contains(&tree, &tmp, o->min_diag_separation)) continue;
        cutoff = S->ctx[context].gap_cutoff_score;
        st->gapped_extensions++;
        if (o->greedy) {
            /* :3465-3484 */
            ih.q_off = ih.q_start + ih.length / 2;
            ih.s_off = ih.s_start + ih.length / 2;
            rc = orc_greedy_gapped(q, subj, qlen, slen, ih.q_off, ih.s_off, S->gap_x_dropoff,
                                   o->reward, o->penalty, o->gap_open, o->gap_extend, &r);
            ih.q_off = r.seed_q; ih.s_off = r.seed_s;
        } else {
            /* :3486-3499 */
            if (s_end >= ih.s_off + 8) { ih.s_off += 3; ih.q_off += 3; }
            rc = orc_dynprog_gapped(S->matrix, q, subj, qlen, slen, ih.q_off, ih.s_off,
                                    S->gap_x_dropoff, o->gap_open, o->gap_extend, &r);
        }
        if (rc) break;
        if (r.score >= cutoff) {
            OrcHSP h; memset(&h, 0, sizeof(h));
            h.context = context; h.score = r.score;
            h.q_offset = r.q_start; h.q_end = r.q_stop; h.s_offset = r.s_start; h.s_end = r.s_stop;
            h.q_gapped_start = ih.q_off; h.s_gapped_start = ih.s_off;
            orc_push_hsp(S, &h);
            tree.pool = S->hsps;
            itree_add(&tree, S->nhsps - 1);
        }
    }
    free(tree.n);
    /* engine glue: purge, odd-score rounding, sort, e-values, reap */
    S->nhsps = orc_hsplist_purge_common_endpoints(S->hsps, S->nhsps);
    if (S->round_down) {                        /* CORE/blast_hits.c:2734-2750 */
        for (index = 0; index < S->nhsps; index++) S->hsps[index].score &= ~1;
    }
    orc_hsplist_sort_by_score(S->hsps, S->nhsps);
    if (S->chunk_mode) return;                  /* the rest after the merge of the sequence's chunk lists */
    {
        int32_t n = 0;
        for (index = 0; index < S->nhsps; index++) {    /* :1655-1738, :1807-1839 */
            OrcHSP *h = &S->hsps[index];
            h->evalue = orc_karlin_StoE(h->score, &S->kbp_gap, S->ctx[h->context].eff_searchsp);
            h->evalue /= 1.;    /* gap_decay_divisor */
            if (h->evalue > o->evalue) continue;
            if (index > n) S->hsps[n] = *h;
            n++;
        }
        S->nhsps = n;
    }
    if (S->nhsps > 0) { st->seqs_passed++; st->good_extensions += S->nhsps; }
}
