/* orc_dust.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see orc.h): symmetric DUST, the low-complexity
 * filter blastn applies to its queries by default ("-dust 20 64 1", soft masking).
 *
 * Restates c++/src/algo/dustmask/symdust.cpp:40-287 and c++/include/algo/dustmask/symdust.hpp:271-285
 * (triplet window, perfect intervals, linker merge) in the reference's order of operations, and the
 * final merge of c++/src/algo/blast/api/dust_filter.cpp:96-127 (sorted; overlapping and abutting
 * intervals fused).  Input is BLASTNA: codes 0-3 are A,C,G,T; anything else counts as A, as the
 * IUPAC converter of the reference does for non-ACGT letters (symdust.hpp:70-81).
 * No known-answer test of the reference for this filter is reproducible offline (they fetch GenBank
 * entries).  It is pinned on the published definition instead: tests/test_dust.py holds a brute-force
 * implementation of the paper's definition (perfect intervals by exhaustive search, no sliding window)
 * and this restatement equals it on every sequence tried.
 */
#include "orc_int.h"
#include <stdlib.h>
#include <string.h>

typedef struct Perfect { uint32_t first, second, score, len; } Perfect;

typedef struct Dust {
    /* triplet list: index 0 = newest (push_front), size n */
    uint8_t list[80]; uint32_t n;
    uint32_t start, stop, max_size, low_k, L;
    uint8_t c_w[64], c_v[64];
    uint32_t r_w, r_v, num_diff;
    Perfect *P; uint32_t np, capp;          /* perfect list, front = index 0 */
    uint32_t thresholds[80];
} Dust;

static void p_insert(Dust *d, uint32_t at, Perfect x)
{
    if (d->np == d->capp) { d->capp = d->capp ? 2 * d->capp : 64; d->P = (Perfect *)realloc(d->P, d->capp * sizeof(Perfect)); }
    memmove(d->P + at + 1, d->P + at, (d->np - at) * sizeof(Perfect));
    d->P[at] = x; d->np++;
}
static void push_front(Dust *d, uint8_t t) { memmove(d->list + 1, d->list, d->n); d->list[0] = t; d->n++; }
#define ADD_INFO(r, c, t) do { (r) += (c)[t]; ++(c)[t]; } while (0)      /* symdust.hpp:271-273 */
#define REM_INFO(r, c, t) do { --(c)[t]; (r) -= (c)[t]; } while (0)      /* symdust.hpp:283-285 */

static int shift_high(Dust *d, uint8_t t)                   /* symdust.cpp:52-72 */
{
    uint8_t s = d->list[d->n - 1];
    Perfect x;
    d->n--;
    REM_INFO(d->r_w, d->c_w, s);
    if (d->c_w[s] == 0) --d->num_diff;
    ++d->start;
    push_front(d, t);
    if (d->c_w[t] == 0) ++d->num_diff;
    ADD_INFO(d->r_w, d->c_w, t);
    ++d->stop;
    if (d->num_diff <= 1) {
        x.first = d->start; x.second = d->stop + 1; x.score = 0; x.len = 0;
        p_insert(d, 0, x);
        return 0;
    }
    return 1;
}

static int shift_window(Dust *d, uint8_t t)                 /* symdust.cpp:75-118 */
{
    if (d->n >= d->max_size) {
        uint8_t s;
        if (d->num_diff <= 1) return shift_high(d, t);
        s = d->list[d->n - 1];
        d->n--;
        REM_INFO(d->r_w, d->c_w, s);
        if (d->c_w[s] == 0) --d->num_diff;
        if (d->L == d->start) { ++d->L; REM_INFO(d->r_v, d->c_v, s); }
        ++d->start;
    }
    push_front(d, t);
    if (d->c_w[t] == 0) ++d->num_diff;
    ADD_INFO(d->r_w, d->c_w, t);
    ADD_INFO(d->r_v, d->c_v, t);
    if (d->c_v[t] > d->low_k) {
        uint32_t off = d->n - (d->L - d->start) - 1;
        uint8_t u;
        do {
            u = d->list[off];
            REM_INFO(d->r_v, d->c_v, u);
            ++d->L;
            off--;
        } while (u != t);
    }
    ++d->stop;
    if (d->n >= d->max_size && d->num_diff <= 1) {
        Perfect x;
        d->np = 0;
        x.first = d->start; x.second = d->stop + 1; x.score = 0; x.len = 0;
        p_insert(d, 0, x);
        return 0;
    }
    return 1;
}

static int needs_processing(const Dust *d)                  /* symdust.hpp, needs_processing() */
{
    uint32_t count = d->stop - d->L;
    return count < d->n && 10 * d->r_w > d->thresholds[count];
}

static void find_perfect(Dust *d)                           /* symdust.cpp:121-173 */
{
    uint8_t counts[64];
    uint32_t count = d->stop - d->L, score = d->r_v, it, pi = 0, max_perfect_score = 0, max_len = 0;
    uint32_t pos = d->L - 1;                                /* unsigned wrap as in the reference */
    memcpy(counts, d->c_v, 64);
    for (it = count; it < d->n; ++it, ++count, --pos) {
        uint8_t t = d->list[it], cnt = counts[t];
        ADD_INFO(score, counts, t);
        if (cnt > 0 && score * 10 > d->thresholds[count]) {
            while (pi != d->np && pos <= d->P[pi].first) {
                if (max_perfect_score == 0 || max_len * d->P[pi].score > max_perfect_score * d->P[pi].len) {
                    max_perfect_score = d->P[pi].score; max_len = d->P[pi].len;
                }
                ++pi;
            }
            if (max_perfect_score == 0 || score * max_len >= max_perfect_score * count) {
                Perfect x;
                max_perfect_score = score; max_len = count;
                x.first = pos; x.second = d->stop + 1; x.score = max_perfect_score; x.len = count;
                p_insert(d, pi, x);                         /* iterator then points at the new element */
            }
        }
    }
}

typedef struct Res { int32_t *from, *to; int32_t n, cap; uint32_t linker; } Res;

static void save_masked_regions(Dust *d, Res *r, uint32_t wstart, uint32_t start)   /* symdust.cpp:190-216 */
{
    if (d->np) {
        Perfect b = d->P[d->np - 1];
        if (b.first < wstart) {
            uint32_t b1f = b.first + start, b1s = b.second + start;
            if (r->n) {
                uint32_t s = (uint32_t)r->to[r->n - 1];
                if (s + r->linker >= b1f) r->to[r->n - 1] = (int32_t)ORC_MAX(s, b1s);
                else if (r->n < r->cap) { r->from[r->n] = (int32_t)b1f; r->to[r->n++] = (int32_t)b1s; }
            } else if (r->n < r->cap) { r->from[r->n] = (int32_t)b1f; r->to[r->n++] = (int32_t)b1s; }
            while (d->np && d->P[d->np - 1].first < wstart) d->np--;
        }
    }
}

#define CONV(x) ((uint8_t)(((x) <= 3) ? (x) : 0))

int32_t orc_dust(const uint8_t *seq, int32_t len, int level, int window, int linker,
                 int32_t *from, int32_t *to, int32_t cap)
{
    Dust d; Res res; uint32_t start = 0, stop, i; int32_t k, m;
    if (!(level >= 2 && level <= 64)) level = 20;          /* symdust.cpp:176-187 */
    if (!(window >= 8 && window <= 64)) window = 64;
    if (!(linker >= 1 && linker <= 32)) linker = 1;
    memset(&d, 0, sizeof(d));
    res.from = from; res.to = to; res.n = 0; res.cap = cap; res.linker = (uint32_t)linker;
    d.thresholds[0] = 1;
    for (i = 1; i < (uint32_t)window - 2; i++) d.thresholds[i] = i * (uint32_t)level;
    if (len <= 0) return 0;
    stop = (uint32_t)len - 1;
    while (stop > 2 + start) {                              /* symdust.cpp:219-283 */
        uint32_t pos, wstart; uint8_t t; int done = 0;
        d.np = 0; d.n = 0; d.start = 0; d.stop = 0; d.max_size = (uint32_t)window - 2; d.low_k = (uint32_t)level / 5;
        d.L = 0; d.r_w = d.r_v = d.num_diff = 0; memset(d.c_w, 0, 64); memset(d.c_v, 0, 64);
        t = (uint8_t)((CONV(seq[start]) << 2) + CONV(seq[start + 1]));
        pos = start + d.stop + 2;
        while (!done && pos <= stop) {
            save_masked_regions(&d, &res, d.start, start);
            t = (uint8_t)(((t << 2) & 0x3F) + (CONV(seq[pos]) & 3));
            ++pos;
            if (shift_window(&d, t)) {
                if (needs_processing(&d)) find_perfect(&d);
            } else {
                while (pos <= stop) {
                    save_masked_regions(&d, &res, d.start, start);
                    t = (uint8_t)(((t << 2) & 0x3F) + (CONV(seq[pos]) & 3));
                    if (shift_window(&d, t)) { done = 1; break; }
                    ++pos;
                }
            }
        }
        wstart = d.start;
        while (d.np) { save_masked_regions(&d, &res, wstart, start); ++wstart; }
        if (d.start > 0) start += d.start; else break;
    }
    free(d.P);
    /* dust_filter.cpp:119-127: sorted, overlapping and abutting intervals fused */
    for (k = 0, m = 0; k < res.n; k++) {
        if (m && from[k] <= to[m - 1] + 1) to[m - 1] = ORC_MAX(to[m - 1], to[k]);
        else { from[m] = from[k]; to[m++] = to[k]; }
    }
    return m;
}
