// seed_stage.hip -- the seed stage of the preliminary search (gfx950 / CDNA4; a translation unit of its own since round 5,
// kernels.hip keeps the direct and the slice scans): what BlastNaWordFinder does with every verified seed.
//   seed_keys / group_keys / seed_ckeys kernels: sort keys for the library sorts (keep_stages, more than GBN_SMALL_SORT_MAX
//                        seeds) and the composite key of the many-seed path
//   diag_ungapped_kernel per-diagonal one-hit filter + X-drop ungapped extension, a thread per (subject, slot) run
//                        (s_BlastnDiag{Table,Hash}ExtendInitialHit + s_NuclUngappedExtend(+Exact), CORE/na_ungapped.c:152-351, :611-922)
//   seed_ext_kernel / seed_ext_ck_kernel + seed_exact_kernel + run_heads_kernel + diag_replay_kernel: the same for many seeds --
//                        every seed extended by a thread of its own, the runs replayed afterwards
// Integer work only: no MFMA.
#include "scan_dev.hpp"
#include <cstring>
#include <algorithm>

#ifndef GBN_DIAG_ABL
#define GBN_DIAG_ABL 0      // timing experiments only (1: no ungapped extension, 2: no strand search): wrong results
#endif
#ifndef GBN_EXT_ABL
#define GBN_EXT_ABL 0       // timing experiments only (seed_ext_kernel), bits: 1 no exact pass, 2 no extension, 4 no reservation of run heads, 8 no context lookup, 16 no record store: wrong results
#endif

// ---------------------------------------------------------------------------
// seed keys for the two stable radix sorts done by the host with hipCUB
// ---------------------------------------------------------------------------
extern "C" __global__ void seed_keys_kernel(GbnKeyParams K)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K.n) return;
    GbnDevSeed sd = K.seeds[i];
    const uint32_t qmax = (K.q_bits >= 32) ? 0xffffffffu : ((1u << K.q_bits) - 1u);
    uint32_t qkey = K.q_descending ? (qmax - (uint32_t)sd.q_pos) : (uint32_t)sd.q_pos;
    K.key_scan[i] = ((uint64_t)(uint32_t)sd.s_scan << K.q_bits) | qkey;
    K.idx[i] = (uint32_t)i;
}

// One 64-bit key per seed that orders the seeds the way the two sorts below do, in a single sort: subject | slot |
// s_scan, and inside one (subject, slot, s_scan) -- seeds whose query positions agree modulo the number of slots,
// a handful per million -- by the high bits of the query key, which seed_ext_kernel applies.  The seed follows from
// key and value (q_pos's low bits = (s_scan - slot) mod slots; value = ext_left | high bits of the query key << 8):
// no second sort, no gathers of seeds by rank afterwards.
// (segmented input: seed_order.hip's seg_first_kernel gives the index of every segment's first seed)
extern "C" __global__ void __launch_bounds__(256) seed_ckeys_kernel(GbnKeyParams K)
{
    // (no LDS and no barrier in here: the kernel runs next to the gapped stage of the range before, whose waves keep the
    // LDS pipes of every CU busy -- with the segment table in LDS it took 1.1 ms there against 0.21 ms alone)
    const unsigned long long *__restrict__ s_first = K.seg_first;
    // A workgroup takes a stretch of consecutive seeds, 4 x 256 at a time.  With the segmented input the prefix sums above
    // are its set-up; a thread's seeds come in ascending order, so its segment only ever moves forward and the bounds
    // of the current one stay in registers (one binary search at the start).  Four seeds of a thread are in flight
    // together: the kernel is a chain of loads per seed, and next to a gapped stage that fills the CUs it ran five times
    // as long as alone with one seed at a time (1.1 against 0.21 ms per 47 million seeds).
    const int64_t per = ((K.n + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
    const int64_t i_end = min(K.n, (int64_t)(blockIdx.x + 1) * per);
    int sgi = 0;
    unsigned long long lo = 0, hi = 0;
    if (K.nseg > 0) {
        const unsigned long long i0 = (unsigned long long)((int64_t)blockIdx.x * per + threadIdx.x);
        int top = K.nseg;
        while (top - sgi > 1) { const int m = (sgi + top) >> 1; if (s_first[m] <= i0) sgi = m; else top = m; }
        lo = s_first[sgi]; hi = s_first[sgi + 1];
    }
    const uint32_t qmax = (K.q_bits >= 32) ? 0xffffffffu : ((1u << K.q_bits) - 1u);
    for (int64_t i = (int64_t)blockIdx.x * per + threadIdx.x; i < i_end; i += 4 * 256) {
        GbnDevSeed sd[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t iu = i + 256 * u;
            if (iu >= i_end) { sd[u] = GbnDevSeed{0, 0, 0, 0}; continue; }
            if (K.nseg > 0) {
                while ((unsigned long long)iu >= hi) { sgi++; lo = hi; hi = s_first[sgi + 1]; }     // (empty segments are stepped over; s_first[nseg] = n > iu)
                sd[u] = K.seg[(size_t)sgi * K.seg_cap + (size_t)((unsigned long long)iu - lo)];
            } else sd[u] = K.seeds[iu];
        }
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t iu = i + 256 * u;
            if (iu >= i_end) continue;
            uint32_t slot, val;
            const uint64_t key = gbn_composite_key(K, sd[u], qmax, slot, val);
            if (K.v_bits > 0) K.key_scan[iu] = (key << K.v_bits) | val;        // key and value in one word: a sort of keys only, on the bits above the value
            else { K.key_scan[iu] = key; K.idx[iu] = val; }
        }
    }
}

extern "C" __global__ void group_keys_kernel(GbnKeyParams K)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K.n) return;
    GbnDevSeed sd = K.seeds[K.idx[i]];
    int32_t q = sd.q_pos - sd.ext_left, s = sd.s_scan - sd.ext_left;
    uint32_t grp = K.container_hash ? ((uint32_t)(s - q) & 511u)
                                    : ((uint32_t)(s + K.diag_len - q) & (uint32_t)(K.diag_len - 1));
    K.key_group[i] = ((uint64_t)(uint32_t)sd.subj << K.group_bits) | grp;
}

// ---------------------------------------------------------------------------
// ungapped extension (device)
// ---------------------------------------------------------------------------
namespace {
struct Ungapped { int32_t q_start, s_start, length, score; };

// s_NuclUngappedExtendExact (CORE/na_ungapped.c:152-244): base by base with the X-drop rule.  32 bases at a time
// from the 2-bit copies of query and subject (a real homolog's thousand bases were a thousand dependent byte loads);
// a stretch of the query with a code above 3 (ambiguity, the sentinel between contexts) goes byte by byte.
__device__ __forceinline__ void ungapped_exact(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t slen,
                               int32_t q_off, int32_t s_off, int32_t X, Ungapped &u)
{
    const uint8_t *q = P.q8;
    int32_t sum = 0, score = 0, q_beg = q_off, q_end = q_off;
    const int32_t nleft = min(q_off, s_off);
    const int32_t nright = min(P.qlen - q_off, slen - s_off);
    const bool packed = P.q2 != nullptr;
    const int32_t reward = P.matrix[0], penalty = P.matrix[1];
    bool stop = false;
    for (int32_t c = 0; c < nleft && !stop; c += 32) {
        const int32_t steps = min(32, nleft - c);
        uint64_t x = 0; uint32_t amb = 1;
        if (packed) {
            x = bases32(subj, (int64_t)s_off - c - 32) ^ bases32(P.q2, (int64_t)q_off - c - 32);
            amb = bits32(P.qinv, (int64_t)q_off - c - 32) & (steps == 32 ? 0xffffffffu : ((1u << steps) - 1u));
        }
        for (int32_t t = 0; t < steps; t++) {
            const int32_t qi = q_off - c - 1 - t;
            sum += amb ? P.matrix[q[qi] * 16 + base_at(subj, s_off - c - 1 - t)] : (((x >> (2 * t)) & 3) ? penalty : reward);
            if (sum > 0) { q_beg = qi; score += sum; sum = 0; }
            else if (sum < X) { stop = true; break; }
        }
    }
    u.q_start = q_beg; u.s_start = s_off - (q_off - q_beg);
    sum = 0; stop = false;
    for (int32_t c = 0; c < nright && !stop; c += 32) {
        const int32_t steps = min(32, nright - c);
        uint64_t x = 0; uint32_t amb = 1;
        if (packed) {
            x = bases32(subj, (int64_t)s_off + c) ^ bases32(P.q2, (int64_t)q_off + c);
            amb = bits32(P.qinv, (int64_t)q_off + c) >> (32 - steps);
        }
        for (int32_t t = 0; t < steps; t++) {
            const int32_t qi = q_off + c + t;
            sum += amb ? P.matrix[q[qi] * 16 + base_at(subj, s_off + c + t)] : (((x >> (62 - 2 * t)) & 3) ? penalty : reward);
            if (sum > 0) { q_end = qi + 1; score += sum; sum = 0; }
            else if (sum < X) { stop = true; break; }
        }
    }
    u.length = q_end - q_beg; u.score = score;
}

// s_NuclUngappedExtend (CORE/na_ungapped.c:262-351): 4 bases per step, score of a step = table[q_byte ^ s_byte]
// = matches * reward + mismatches * penalty of the four 2-bit groups (CORE/blast_parameters.c:237-262).  A step's
// query byte is built from the unpacked codes, so a code above 3 (ambiguity, sentinel between the strands) spills
// into its neighbours' bits: such groups go through the byte-wise formula; all others are taken 8 steps at a time
// from the 2-bit copy of the query (three dwords of query, three of subject, two of the "matches nothing" bitmap
// per 32 bases instead of six dependent loads per step).
__device__ void ungapped_approx_steps(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t slen,
                                int32_t q_off, int32_t s_match_end, int32_t s_off, int32_t X,
                                int32_t reduced_cutoff, Ungapped &u)
{
    const uint8_t *qs = P.q8;
    const int32_t t4 = P.score_table[0], dt = P.score_table[1] - t4;      // 4 * reward, penalty - reward
    auto step_score = [&](uint32_t q_byte, uint32_t s_byte) -> int32_t {
        const uint32_t x = q_byte ^ s_byte;
        return t4 + dt * (int32_t)__popc((x | (x >> 1)) & 0x55u);
    };
    int32_t len = (4 - (s_off & 3)) & 3;
    const int32_t q_ext = q_off + len, s_ext = s_off + len;
    int32_t score = 0, sum = 0, new_q = q_ext;
    {   // left
        const int32_t n = min(q_ext, s_ext) >> 2;
        bool stop = false;
        for (int32_t c = 0; c * 8 < n && !stop; c++) {
            const int64_t sa = (int64_t)s_ext - 32 * (c + 1), qa = (int64_t)q_ext - 32 * (c + 1);
            const uint64_t S = bases32(subj, sa);
            const bool packed = P.q2 != nullptr;
            const uint64_t Q = packed ? bases32(P.q2, qa) : 0ull;
            const uint32_t I = packed ? bits32(P.qinv, qa) : 0xffffffffu;
            const int32_t steps = min(8, n - c * 8);
            for (int32_t t = 0; t < steps; t++) {
                const int32_t qi = q_ext - 4 * (c * 8 + t);             // the step covers query bases qi-4 .. qi-1
                const uint32_t s_byte = (uint32_t)(S >> (8 * t)) & 0xffu;
                uint32_t q_byte;
                if ((I >> (4 * t)) & 0xfu) q_byte = (uint8_t)((qs[qi - 4] << 6) | (qs[qi - 3] << 4) | (qs[qi - 2] << 2) | qs[qi - 1]);
                else q_byte = (uint32_t)(Q >> (8 * t)) & 0xffu;
                sum += step_score(q_byte, s_byte);
                if (sum > 0) { new_q = qi - 4; score += sum; sum = 0; }
                if (sum < X) { stop = true; break; }
            }
        }
    }
    const int32_t uq = new_q, us = s_ext - (q_ext - new_q);
    sum = 0; new_q = q_ext;
    {   // right
        const int32_t n = min(P.qlen - q_ext, slen - s_ext) >> 2;
        bool stop = false;
        for (int32_t c = 0; c * 8 < n && !stop; c++) {
            const int64_t sa = (int64_t)s_ext + 32 * c, qa = (int64_t)q_ext + 32 * c;
            const uint64_t S = bases32(subj, sa);
            const bool packed = P.q2 != nullptr;
            const uint64_t Q = packed ? bases32(P.q2, qa) : 0ull;
            const uint32_t I = packed ? bits32(P.qinv, qa) : 0xffffffffu;
            const int32_t steps = min(8, n - c * 8);
            for (int32_t t = 0; t < steps; t++) {
                const int32_t qi = q_ext + 4 * (c * 8 + t);             // the step covers query bases qi .. qi+3
                const uint32_t s_byte = (uint32_t)(S >> (56 - 8 * t)) & 0xffu;
                uint32_t q_byte;
                if ((I >> (28 - 4 * t)) & 0xfu) q_byte = (uint8_t)((qs[qi] << 6) | (qs[qi + 1] << 4) | (qs[qi + 2] << 2) | qs[qi + 3]);
                else q_byte = (uint32_t)(Q >> (56 - 8 * t)) & 0xffu;
                sum += step_score(q_byte, s_byte);
                if (sum > 0) { new_q = qi + 3; score += sum; sum = 0; }
                if (sum < X) { stop = true; break; }
            }
        }
    }
    // (the result is put together in values and stored once: with stores to u's fields on both paths the compiler
    // kept them in private memory -- the only scratch use of the two kernels that extend seeds)
    Ungapped r;
    if (score >= reduced_cutoff) {
        Ungapped e; e.q_start = 0; e.s_start = 0; e.length = 0; e.score = 0;
        ungapped_exact(P, subj, slen, q_off, s_off, X, e);
        r = e;
    } else {
        r.q_start = uq; r.s_start = us; r.score = score;
        r.length = max(s_match_end - us, new_q - uq + 1);
    }
    u.q_start = r.q_start; u.s_start = r.s_start; u.length = r.length; u.score = r.score;
}

// The same function eight steps per round (needs GbnExtParams::q4).  The subject position the steps start from is a
// multiple of 4, so a round's eight subject bytes are one 8-byte load, its eight query bytes every fourth byte of 32
// consecutive q4 bytes (ambiguity codes and the sentinel between the strands are already folded in there the way the
// reference's byte-wise formula folds them: no special case), the eight mismatch counts one XOR and a byte-wise
// population count.  What stays per step is the X-drop recurrence itself, without branches: a lane that has dropped
// out keeps a sum that can never recover, the wave goes round as long as one of its lanes is alive (the step-by-step
// form above ran every lane for as many steps as the longest of 64 took, at 15 instructions a step).
// Only rounds that are cut short by the end of the query or the subject go step by step.
// first round of a side, loaded ahead by the caller (seed_ext_ck_kernel issues the loads of both sides before it
// looks at either): the eight subject bytes and the 32 q4 bytes of the round
struct ApproxPre { uint32_t s0, s1, q0, q1; };      // subject bytes, query bytes (q4 bytes of eight consecutive steps), address order
// where the q4 bytes of the steps that start at query position p lie (GbnExtParams::q4: four planes by offset mod 4)
__device__ __forceinline__ const uint8_t *q4_at(const GbnExtParams &P, int32_t p)
{
    const int32_t k = p + P.q4_origin;
    return P.q4 + (int64_t)(k & 3) * P.q4_plane + (k >> 2);
}
template <bool LEFT>
__device__ __forceinline__ void approx_side(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t q_ext, int32_t s_ext,
                                            int32_t n, int32_t X, int32_t t4, int32_t dt, int32_t &score, int32_t &best,
                                            bool have_pre = false, ApproxPre pre = ApproxPre{0, 0, 0, 0})
{
    constexpr int32_t kDead = INT32_MIN / 2;
    int32_t sum = 0;
    best = 0;                                                   // steps the best prefix covers
    for (int32_t c = 0; c * 8 < n; c++) {
        const int32_t steps = min(8, n - c * 8);
        const int32_t qa = LEFT ? q_ext - 32 * (c + 1) : q_ext + 32 * c;
        const int32_t sa = LEFT ? s_ext - 32 * (c + 1) : s_ext + 32 * c;
        int32_t bt = -1;
        if (steps == 8) {
            uint32_t sw[2], qlo, qhi;
            if (have_pre && c == 0) { sw[0] = pre.s0; sw[1] = pre.s1; qlo = pre.q0; qhi = pre.q1; }
            else {
                uint32_t qq[2];
                __builtin_memcpy(sw, subj + (sa >> 2), 8);
                __builtin_memcpy(qq, q4_at(P, qa), 8);
                qlo = qq[0]; qhi = qq[1];
            }
            // byte k (address order) of S and Q = the four bases qa + 4k .. qa + 4k + 3
            uint32_t m[2] = {qlo ^ sw[0], qhi ^ sw[1]};
            #pragma unroll
            for (int h = 0; h < 2; h++) {                       // mismatching 2-bit groups per byte
                uint32_t v = (m[h] | (m[h] >> 1)) & 0x55555555u;
                v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
                m[h] = (v + (v >> 4)) & 0x0f0f0f0fu;
            }
            #pragma unroll
            for (int t = 0; t < 8; t++) {
                const int k = LEFT ? 7 - t : t;
                const int32_t cnt = (int32_t)((m[k >> 2] >> (8 * (k & 3))) & 0xffu);
                sum += t4 + dt * cnt;
                const bool pos = sum > 0;
                bt = pos ? t : bt;
                score += pos ? sum : 0;
                sum = pos ? 0 : sum;
                sum = sum < X ? kDead : sum;
            }
        } else {
            const uint8_t *qs = P.q8;
            for (int32_t t = 0; t < steps; t++) {
                const int32_t qi = qa + (LEFT ? 4 * (7 - t) : 4 * t);   // the step covers query bases qi .. qi + 3
                const uint32_t s_byte = subj[(sa >> 2) + (LEFT ? 7 - t : t)];
                const uint32_t x = (uint8_t)((qs[qi] << 6) | (qs[qi + 1] << 4) | (qs[qi + 2] << 2) | qs[qi + 3]) ^ s_byte;
                sum += t4 + dt * (int32_t)__popc((x | (x >> 1)) & 0x55u);
                if (sum > 0) { bt = t; score += sum; sum = 0; }
                if (sum < X) { sum = kDead; break; }
            }
        }
        if (bt >= 0) best = c * 8 + bt + 1;
        if (sum < X) break;
    }
}

__device__ void ungapped_approx(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t slen,
                                int32_t q_off, int32_t s_match_end, int32_t s_off, int32_t X,
                                int32_t reduced_cutoff, Ungapped &u)
{
    const int32_t t4 = P.score_table[0], dt = P.score_table[1] - t4;      // 4 * reward, penalty - reward
    // (a dead lane's sum must stay below X whatever follows: eight steps add at most 8 * t4 per round, rounds < 2^26)
    if (P.q4 == nullptr || X < -(1 << 24) || t4 > (1 << 20) || t4 < 0) { ungapped_approx_steps(P, subj, slen, q_off, s_match_end, s_off, X, reduced_cutoff, u); return; }
    const int32_t len = (4 - (s_off & 3)) & 3;
    const int32_t q_ext = q_off + len, s_ext = s_off + len;
    int32_t score = 0, bl = 0, br = 0;
    approx_side<true>(P, subj, q_ext, s_ext, min(q_ext, s_ext) >> 2, X, t4, dt, score, bl);
    approx_side<false>(P, subj, q_ext, s_ext, min(P.qlen - q_ext, slen - s_ext) >> 2, X, t4, dt, score, br);
    const int32_t uq = q_ext - 4 * bl, us = s_ext - 4 * bl;
    const int32_t new_q = br ? q_ext + 4 * br - 1 : q_ext;     // the reference's new_q: last base of the best step, or where the loop began
    Ungapped r;
    if (score >= reduced_cutoff && !(GBN_EXT_ABL & 1)) {
        Ungapped e; e.q_start = 0; e.s_start = 0; e.length = 0; e.score = 0;
        ungapped_exact(P, subj, slen, q_off, s_off, X, e);
        r = e;
    } else {
        r.q_start = uq; r.s_start = us; r.score = score;
        r.length = max(s_match_end - us, new_q - uq + 1);
    }
    u.q_start = r.q_start; u.s_start = r.s_start; u.length = r.length; u.score = r.score;
}
}  // namespace

namespace {
// s_IsSeedMasked (CORE/na_ungapped.c:459-471): is query offset q_pos absent from the cell of the lookup
// word the subject carries at s_pos (a masked or ambiguous query position is not indexed)
__device__ bool seed_masked(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t s_pos, int32_t q_pos)
{
    const uint32_t cell = (window16(subj, s_pos) >> (32 - 2 * P.lut)) & P.cell_mask;
    for (uint32_t e = P.cell_start[cell]; e < P.cell_start[cell + 1]; e++)
        if ((int32_t)(uint32_t)P.ent[e] == q_pos) return false;
    return true;
}

// s_TypeOfWord (CORE/na_ungapped.c:488-587), one-hit mode: re-check of the mini-extended word against the
// query masks; may move the left end of the word right and extend its right end.  false: drop the seed.
__device__ __forceinline__ bool type_of_word(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t slen,
                                             int32_t &q_off_io, int32_t &s_off_io, int32_t &extended_out)
{
    // (the offsets live in values of this function's own and are handed back once: taken by reference throughout, the
    // compiler kept them in private memory, the only scratch use of the two kernels that call this)
    const int32_t word = P.word, lut = P.lut;
    int32_t q_off = q_off_io, s_off = s_off_io, extended = 0;
    extended_out = 0;
    if (word == lut) return true;
    int32_t q_end = q_off + word, s_end = s_off + word;
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > q_end) hi = m; else lo = m; }
    const int32_t q_range = P.ctx_off[lo] + P.ctx_len[lo];
    if (P.masked) {
        if (seed_masked(P, subj, s_end - lut, q_end - lut)) return false;
        while (seed_masked(P, subj, s_off, q_off)) { ++s_off; ++q_off; }
    }
    const int32_t ext_to = word - (q_end - q_off);
    const int32_t ext_max = min(q_range - q_end, slen - s_end);
    if (ext_to || P.masked) {
        if (ext_to > ext_max) return false;
        q_end += ext_to; s_end += ext_to;
        for (int32_t s_pos = s_end - lut, q_pos = q_end - lut; s_pos > s_off; s_pos -= lut, q_pos -= lut)
            if (seed_masked(P, subj, s_pos, q_pos)) return false;
        extended = ext_to;
    }
    q_off_io = q_off; s_off_io = s_off; extended_out = extended;
    return true;
}
}  // namespace

// One thread per (subject, diagonal-slot) run of seeds; the run is replayed in
// scan order because the one-hit filter is a sequential state machine
// (CORE/na_ungapped.c:652,748 / :818,917).  The hash container is emulated
// exactly per subject: cells live in a scratch slice as long as the run.
// compaction of the run heads, so that every lane of the replay kernel below has a run to work on
// (with the 512-bucket hash container a run is ~n / (512 x subjects) seeds long)
extern "C" __global__ void __launch_bounds__(1024) run_heads_kernel(GbnExtParams P)
{
    // one reservation per 1024-thread block (a single counter takes ~90 atomics per microsecond)
    __shared__ uint32_t s_cnt[16], s_base;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = i < P.n && (i == 0 || P.key_group[i - 1] != P.key_group[i]);
    const unsigned long long m = __ballot(head);
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; w++) { const uint32_t c = s_cnt[w]; s_cnt[w] = tot; tot += c; }
        s_base = tot ? atomicAdd(P.run_count, tot) : 0u;
    }
    __syncthreads();
    if (head) P.run_heads[s_base + s_cnt[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)i;
}

extern "C" __global__ void diag_ungapped_kernel(GbnExtParams P)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i;
    uint64_t key;
    if (P.run_heads == nullptr) {                 // small inputs: thread per seed, run heads work
        if (t >= P.n) return;
        i = t;
        key = P.key_group[i];
        if (i > 0 && P.key_group[i - 1] == key) return;
    } else {
        if (t >= (int64_t)*P.run_count) return;
        i = P.run_heads[t];
        key = P.key_group[i];
    }
    const int32_t subj_id = (int32_t)(key >> (P.group_bits ? P.group_bits : 32));
    const uint8_t *__restrict__ subj = P.db + P.byte_off[subj_id];
    const int32_t slen = P.len[subj_id];
    const int word = P.word;
    int32_t last_hit = 0;           // array container: one slot per run
    int64_t ncell = 0;              // hash container: cells [i, i+ncell)
    for (int64_t j = i; j < P.n && P.key_group[j] == key; j++) {
        GbnDevSeed sd = P.seeds[P.idx[j]];
        int32_t q_off = sd.q_pos - sd.ext_left, s_off = sd.s_scan - sd.ext_left;
        const int32_t diag = s_off - q_off;
        const int32_t s_off_pos = s_off;                // the container is keyed by the word as the scan delivered it
        int32_t s_end_pos = s_off + word;
        if (P.container_hash) {
            last_hit = 0;
            for (int64_t c = ncell - 1; c >= 0; c--)
                if (P.cell_diag[i + c] == diag) { last_hit = P.cell_level[i + c]; break; }
        }
        if (s_off < last_hit) continue;
        int32_t s_match_end = s_off + word;
        if (P.masked) {                                 // without masks s_TypeOfWord changes nothing
            int32_t extended;
            if (!type_of_word(P, subj, slen, q_off, s_off, extended)) continue;
            s_match_end += extended; s_end_pos += extended;
        }
        // strand of the seed
        int lo = 0, hi = P.nctx;
#if GBN_DIAG_ABL != 2
        while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > q_off) hi = m; else lo = m; }
#endif
        const int32_t X = -P.ctx_xdrop[lo];
        Ungapped u;
#if GBN_DIAG_ABL == 1
        u.score = 0; u.length = word; u.q_start = q_off; u.s_start = s_off;
#else
        if (!P.container_hash && word < 11) ungapped_exact(P, subj, slen, q_off, s_off, X, u);
        else ungapped_approx(P, subj, slen, q_off, s_match_end, s_off, X, P.ctx_reduced[lo], u);
#endif
        if (u.score >= P.ctx_cutoff[lo]) {
            unsigned long long o = atomicAdd(P.ihit_count, 1ull);
            if (o < P.ihit_cap) {
                GbnDevInitHit h; h.subj = subj_id; h.q_off = q_off; h.s_off = s_off;
                h.q_start = u.q_start; h.s_start = u.s_start; h.length = u.length; h.score = u.score;
                h.seq = (uint32_t)j;
                P.ihits[o] = h;
            }
            s_end_pos = u.length + u.s_start;
        }
        if (P.container_hash) {
            // s_BlastDiagHashInsert with window = 0 + MIN(0, -word) + 1
            const int32_t win = min(0, -word) + 1;
            bool placed = false;
            for (int64_t c = ncell - 1; c >= 0; c--) {
                if (P.cell_diag[i + c] == diag) { P.cell_level[i + c] = s_end_pos; placed = true; break; }
                if (s_off_pos - P.cell_level[i + c] > win) { P.cell_diag[i + c] = diag; P.cell_level[i + c] = s_end_pos; placed = true; break; }
            }
            if (!placed) { P.cell_diag[i + ncell] = diag; P.cell_level[i + ncell] = s_end_pos; ncell++; }
        } else {
            last_hit = s_end_pos;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same stage for many seeds (blastn word sizes: ~10 M seeds per launch, ~18 per run with the hash container),
// in two kernels.  What a seed's ungapped extension yields does not depend on the container's state -- only
// whether it is looked at does -- so seed_ext_kernel extends EVERY seed, a thread each (the memory latency of the
// extension's gathers hidden by ten million threads instead of sitting in the 18-step chain of a run), and
// diag_replay_kernel walks the runs over the finished records: per seed a sequential read, the container update
// and nothing else.  (~20 % of the extensions are of seeds the replay then skips.)
// ---------------------------------------------------------------------------------------------------
// what the replay reads of every seed (16 bytes) ...; flags: 1 = dropped by the mask re-check, 2 = reaches the cutoff, 4 = last
// of its run, [31:8] = bases the re-check added on the right
struct GbnSeedExt { int32_t q_off, s_off, s_orig, flags; };
// ... and the extension itself, which it needs of the few that reach the cutoff (one in 500 on C3): written and read for
// those only, in the second half of ext_rec (records of all n seeds first, then n slots of these)
struct GbnSeedHsp { int32_t q_start, s_start, length, score; };

__device__ __forceinline__ int context_of(const GbnExtParams &P, int32_t q)
{
    int lo;
    if (P.ctx_hint) { lo = P.ctx_hint[q >> P.ctx_hint_shift]; while (lo + 1 < P.nctx && P.ctx_off[lo + 1] <= q) lo++; }
    else { lo = 0; int hi = P.nctx; while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > q) hi = m; else lo = m; } }
    return lo;
}

extern "C" __global__ void __launch_bounds__(256) seed_ext_kernel(GbnExtParams P)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < P.n;
    const int gb = P.group_bits ? P.group_bits : 32;
    int32_t subj_id = 0; GbnDevSeed sd; sd.subj = 0; sd.s_scan = 0; sd.q_pos = 0; sd.ext_left = 0;
    bool head = false, last = false;
    int64_t pos = j;                                    // where this seed's record goes
    if (live) {
        const int vb = P.ck_vbits;                          // > 0: the value sits in the key's low bits
        const uint64_t vmask = (1ull << vb) - 1ull;
        const uint64_t word64 = P.key_group[j];
        const uint64_t key = word64 >> vb;
        if (P.ck_shift > 0) {
            // composite keys: the seed is in key and value.  Seeds with one key (same subject, slot and scan position)
            // come out of the sort in no particular order: each finds its place among them by the high bits of its
            // query key and works for the position it lands on.
            const uint32_t val = vb ? (uint32_t)(word64 & vmask) : P.idx[j];
            const uint32_t qk = val >> 8;
            int64_t a = j, e = j + 1;
            while (a > 0 && (P.key_group[a - 1] >> vb) == key) a--;
            while (e < P.n && (P.key_group[e] >> vb) == key) e++;
            int64_t rank = 0;
            for (int64_t m = a; m < e; m++) {
                if (m == j) continue;
                const uint32_t qm = (vb ? (uint32_t)(P.key_group[m] & vmask) : P.idx[m]) >> 8;
                rank += (qm < qk || (qm == qk && m < j)) ? 1 : 0;
            }
            pos = a + rank;
            const uint64_t run = key >> P.ck_shift;
            head = pos == a && (a == 0 || (P.key_group[a - 1] >> (vb + P.ck_shift)) != run);
            last = pos == e - 1 && (e >= P.n || (P.key_group[e] >> (vb + P.ck_shift)) != run);
            subj_id = (int32_t)(run >> gb) + P.ck_subj_base;
            const uint32_t mask = (gb >= 32) ? 0xffffffffu : ((1u << gb) - 1u);
            const uint32_t slot = (uint32_t)run & mask;
            sd.s_scan = (int32_t)(key & ((1ull << P.ck_s_bits) - 1ull));
            const uint32_t ql = ((uint32_t)sd.s_scan - slot) & mask;                      // q_pos modulo the number of slots
            if (P.ck_q_desc) {
                const uint32_t qmax = (P.ck_q_bits >= 32) ? 0xffffffffu : ((1u << P.ck_q_bits) - 1u);
                const uint32_t t = (P.ck_qh_bits ? (qk << gb) : 0u) | ((qmax - ql) & mask);
                sd.q_pos = (int32_t)(qmax - t);
            } else sd.q_pos = (int32_t)((P.ck_qh_bits ? (qk << gb) : 0u) | ql);
            sd.ext_left = (int32_t)(val & 0xffu);
        } else {
            subj_id = (int32_t)(key >> gb);
            sd = P.seeds[P.idx[j]];
        }
    }
    if (live) {
        const uint8_t *__restrict__ subj = P.db + P.byte_off[subj_id];
        const int32_t slen = P.len[subj_id];
        GbnSeedExt r; GbnSeedHsp hs;
        int32_t q_off = sd.q_pos - sd.ext_left, s_off = sd.s_scan - sd.ext_left;
        r.s_orig = s_off; r.flags = 0; hs.q_start = 0; hs.s_start = 0; hs.length = 0; hs.score = 0;
        int32_t s_match_end = s_off + P.word;
        bool ok = true;
        if (P.masked) {                                     // without masks s_TypeOfWord changes nothing
            int32_t extended;
            ok = type_of_word(P, subj, slen, q_off, s_off, extended);
            s_match_end += extended; r.flags = ok ? (extended << 8) : 1;
        }
        if (ok) {
            const int lo = (GBN_EXT_ABL & 8) ? 0 : context_of(P, q_off);
            Ungapped u; u.q_start = 0; u.s_start = 0; u.length = 0; u.score = 0;
            if (GBN_EXT_ABL & 2) { }
            else if (!P.container_hash && P.word < 11) ungapped_exact(P, subj, slen, q_off, s_off, -P.ctx_xdrop[lo], u);
            // (the exact pass stays inline: handing its seeds to a kernel of their own -- dense waves -- gained 0.1 ms per
            // 47 M seeds once that pass read 32 bases per load, not worth a kernel)
            else ungapped_approx(P, subj, slen, q_off, s_match_end, s_off, -P.ctx_xdrop[lo], P.ctx_reduced[lo], u);
            hs.q_start = u.q_start; hs.s_start = u.s_start; hs.length = u.length; hs.score = u.score;
            if (u.score >= P.ctx_cutoff[lo]) r.flags |= 2;
        }
        if (last) r.flags |= 4;
        r.q_off = q_off; r.s_off = s_off;
        if (!(GBN_EXT_ABL & 16)) reinterpret_cast<GbnSeedExt *>(P.ext_rec)[pos] = r;
        if (r.flags & 2) (reinterpret_cast<GbnSeedHsp *>(reinterpret_cast<GbnSeedExt *>(P.ext_rec) + P.n))[pos] = hs;
    }
    if (P.ck_shift > 0) {
        // the run heads, compacted (what run_heads_kernel does for the other form): one atomic per workgroup
        __shared__ uint32_t s_cnt[4], s_base;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long mh = __ballot(head);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mh);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t th = 0;
            for (int w = 0; w < 4; w++) { const uint32_t c = s_cnt[w]; s_cnt[w] = th; th += c; }
            s_base = (GBN_EXT_ABL & 4) ? 0u : (th ? atomicAdd(P.run_count, th) : 0u);
        }
        __syncthreads();
        if (head && !(GBN_EXT_ABL & 4)) P.run_heads[s_base + s_cnt[wave] + (uint32_t)__popcll(mh & ((1ull << lane) - 1))] = (uint32_t)pos;
    }
}

// ---- the exact pass from what the approximate pass already holds -------------------------------------------------
// s_NuclUngappedExtendExact walks base by base from (q_off, s_off); its first ~30 bases either way are the bases of the
// two rounds the approximate pass started with (32 left of q_ext, 32 right of it), which are in registers.  The walk is
// taken from a mask of the "special" bases -- mismatches and query codes above 3 -- instead of base by base: a run of k
// matches between two special bases adds k x reward in one step (inside it the running sum only rises, so the rule
// `sum > 0 -> take it over, best end here` needs looking at once, at the run's end), a mismatch adds the penalty and is
// where the walk can drop out; an ambiguity code or the sentinel goes through the matrix.  A walk that uses up the
// window without dropping out (a real homolog) goes on base by base from memory (exact_walk_from).
// Only what the exact walk decides is different from ungapped_exact: nothing.  Only how it gets there.
namespace {
// mismatching bases of a 32-base window, bit 31 - j for base j: x0 / x1 = XOR of query and subject bytes 0..3 / 4..7
__device__ __forceinline__ uint32_t mism_mask32(uint32_t x0, uint32_t x1)
{
    uint32_t m[2] = {bswap32(x0), bswap32(x1)};                 // base 0 in the top two bits of m[0]
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t v = (m[h] | (m[h] >> 1)) & 0x55555555u;        // bit 30 - 2j' of the half's base j'
        v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0xffffu;
        m[h] = v;                                               // bit 15 - j'
    }
    return (m[0] << 16) | m[1];
}
// base j (0..31) of a window held as two dwords in address order
__device__ __forceinline__ int win_base(uint32_t w0, uint32_t w1, int j)
{
    const uint32_t w = (j & 16) ? w1 : w0;
    return (int)((w >> (8 * ((j >> 2) & 3) + 6 - 2 * (j & 3))) & 3u);
}
// base by base from memory, starting with base number t0 (0 = next to the seed) of n: the loop of ungapped_exact with its
// state handed in.  best = bases of the best prefix.  Returns true when the walk dropped out.
template <bool LEFT>
__device__ __forceinline__ bool exact_walk_from(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t q_off, int32_t s_off,
                                                int32_t n, int32_t X, int32_t t0, int32_t &sum, int32_t &score, int32_t &best)
{
    const uint8_t *q = P.q8;
    const int32_t reward = P.matrix[0], penalty = P.matrix[1];
    for (int32_t c = t0; c < n; c += 32) {
        const int32_t steps = min(32, n - c);
        const int64_t sa = LEFT ? (int64_t)s_off - c - 32 : (int64_t)s_off + c, qa = LEFT ? (int64_t)q_off - c - 32 : (int64_t)q_off + c;
        const uint64_t x = bases32(subj, sa) ^ bases32(P.q2, qa);
        const uint32_t amb = LEFT ? (bits32(P.qinv, qa) & (steps == 32 ? 0xffffffffu : ((1u << steps) - 1u))) : (bits32(P.qinv, qa) >> (32 - steps));
        for (int32_t t = 0; t < steps; t++) {
            const int32_t qi = LEFT ? q_off - c - 1 - t : q_off + c + t, si = LEFT ? s_off - c - 1 - t : s_off + c + t;
            sum += amb ? P.matrix[q[qi] * 16 + base_at(subj, si)] : (((x >> (LEFT ? 2 * t : 62 - 2 * t)) & 3) ? penalty : reward);
            if (sum > 0) { best = c + t + 1; score += sum; sum = 0; }
            else if (sum < X) return true;
        }
    }
    return false;
}
// One side of the exact walk over the bases the windows hold.  special / amb: LEFT bit t, RIGHT bit 63 - t for the base
// at distance t from the seed; avail = bases of the side the windows cover (and the sequences have).
template <bool LEFT>
__device__ __forceinline__ bool exact_walk_window(const GbnExtParams &P, uint64_t special, uint64_t amb, int32_t avail, int32_t q_off, int32_t X,
                                                  const ApproxPre &pl, const ApproxPre &pr, int32_t len4,
                                                  int32_t &sum, int32_t &score, int32_t &best)
{
    const int32_t reward = P.matrix[0], penalty = P.matrix[1];
    int32_t prev = 0;
    while (special) {
        const int32_t t = LEFT ? (int32_t)__builtin_ctzll(special) : (int32_t)__builtin_clzll(special);
        if (t >= avail) break;
        const unsigned long long bit = LEFT ? (1ull << t) : (0x8000000000000000ull >> t);
        special &= ~bit;
        sum += (t - prev) * reward;                             // the matches up to here
        if (sum > 0) { best = t; score += sum; sum = 0; }
        int32_t v = penalty;
        if (amb & bit) {                                        // (a handful of lanes: near a query's end, at an N)
            const int32_t qi = LEFT ? q_off - 1 - t : q_off + t;
            const int j = LEFT ? 31 - len4 - t : 32 - len4 + t;                    // index in the 64-base window
            const int sb = j < 32 ? win_base(pl.s0, pl.s1, j) : win_base(pr.s0, pr.s1, j - 32);
            v = P.matrix[P.q8[qi] * 16 + sb];
        }
        sum += v;
        if (sum > 0) { best = t + 1; score += sum; sum = 0; }
        else if (sum < X) return true;
        prev = t + 1;
    }
    sum += (avail - prev) * reward;                             // matches to the end of what is there
    if (sum > 0) { best = avail; score += sum; sum = 0; }
    return false;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// seed_ext_kernel for the shape it is launched on at scale (composite keys, hash container, q4 and ctx_blk present, no
// mask re-check).  Two things bounded the general kernel above on 47 M seeds, 2.1 ms each on their own
// (ablations of round 3): (1) one reservation on the global run counter per 256 seeds = 183,000 atomics on one
// address at ~90 per microsecond; (2) a chain of a dozen dependent memory round trips per seed -- key, its neighbours,
// subject offsets, context hint, context offsets, drop-offs, subject left, query left, subject right, query right, ...
// -- with every wave slot of the chip already taken.  Here a WAVE walks a contiguous stretch of the seeds 64 at a
// time: run heads wait in a buffer of its own in LDS (one reservation per ~60 heads, no barrier anywhere), and the
// loads of a seed are issued in three batches -- the keys; then everything the keys determine; then the subject -- before
// any of them is looked at.
// ---------------------------------------------------------------------------------------------------
// The exact pass of a seed whose approximate score reached the reduced cut-off (s_NuclUngappedExtendExact), from the two
// 32-base windows either side of q_ext that the approximate pass has loaded (see exact_walk_window)
__device__ __forceinline__ GbnSeedHsp exact_from_windows(const GbnExtParams &P, const uint8_t *__restrict__ subj, int32_t slen, int32_t q_off, int32_t s_off,
                                                         int32_t len4, int32_t q_ext, int32_t X, const ApproxPre &pl, const ApproxPre &pr)
{
    // the exact pass, from the two windows (see exact_walk_window); the "matches nothing" bits of the same 64
    // bases are read here, by the one seed in six that gets this far (three aligned dwords hold them at any offset)
    uint32_t qiv[3];
    __builtin_memcpy(qiv, P.qinv + 4 * ((int64_t)(q_ext - 32) >> 5), 12);
    const int ish = (q_ext - 32) & 31;
    const uint32_t i0 = bswap32(qiv[0]), i1 = bswap32(qiv[1]), i2 = bswap32(qiv[2]);
    const uint32_t ambl = ish ? __builtin_amdgcn_alignbit(i0, i1, 32 - ish) : i0, ambr = ish ? __builtin_amdgcn_alignbit(i1, i2, 32 - ish) : i1;
    // (a code above 3 spills into the two bits of the base in FRONT of it when the byte of its group is put
    // together -- (q[k] << 6) | (q[k+1] << 4) | ... -- so that base is read from q8 as well; groups do not
    // straddle the windows' ends)
    uint64_t amb64 = ((uint64_t)ambl << 32) | ambr;                 // bit 63 - j for base j of the 64
    amb64 |= amb64 << 1;
    const uint64_t sp64 = (((uint64_t)mism_mask32(pl.q0 ^ pl.s0, pl.q1 ^ pl.s1) << 32) | mism_mask32(pr.q0 ^ pr.s0, pr.q1 ^ pr.s1)) | amb64;
    const int32_t n_l = min(q_off, s_off), n_r = min(P.qlen - q_off, slen - s_off);
    const int32_t a_l = min(32 - len4, n_l), a_r = min(32 + len4, n_r);
    int32_t xs = 0, sum = 0, b_l = 0, b_r = 0;
    // left: base at distance t is window base 31 - len4 - t, i.e. bit 32 + len4 + t of the 64-bit masks
    bool stop = exact_walk_window<true>(P, sp64 >> (32 + len4), amb64 >> (32 + len4), a_l, q_off, X, pl, pr, len4, sum, xs, b_l);
    if (!stop && a_l < n_l) exact_walk_from<true>(P, subj, q_off, s_off, n_l, X, a_l, sum, xs, b_l);
    sum = 0;
    // right: base at distance t is window base 32 - len4 + t, i.e. bit 63 - t after a shift by 32 - len4
    stop = exact_walk_window<false>(P, sp64 << (32 - len4), amb64 << (32 - len4), a_r, q_off, X, pl, pr, len4, sum, xs, b_r);
    if (!stop && a_r < n_r) exact_walk_from<false>(P, subj, q_off, s_off, n_r, X, a_r, sum, xs, b_r);
    GbnSeedHsp hs;
    hs.q_start = q_off - b_l; hs.s_start = s_off - b_l; hs.length = b_l + b_r; hs.score = xs;
    return hs;
}

#ifndef GBN_CK_OCC
#define GBN_CK_OCC 6        // waves per SIMD seed_ext_ck_kernel is compiled for (8, 7: the keys asked for a round ahead do not fit the registers and spill; 6 runs as fast as 7)
#endif
extern "C" __global__ void __launch_bounds__(256, GBN_CK_OCC) seed_ext_ck_kernel(GbnExtParams P)
{
    constexpr int HB = 128;
    __shared__ uint32_t s_hb[4][HB], s_xb[4][HB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t *hb = s_hb[wave], *xb = s_xb[wave];
    int nh = 0, nx = 0;                                         // wave-uniform: heads waiting in hb, seeds for the exact pass in xb
    auto flush_exact = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(P.exact_count, (uint32_t)nx);
        base = __shfl(base, 0);
        for (int k = lane; k < nx; k += 64) P.exact_list[base + (uint32_t)k] = xb[k];
        nx = 0;
    };
    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(P.run_count, (uint32_t)nh);
        base = __shfl(base, 0);
        for (int k = lane; k < nh; k += 64) P.run_heads[base + (uint32_t)k] = hb[k];
        nh = 0;
    };
    // round r of wave w takes the 64 seeds of chunk r * waves + w: what the waves in flight work on at any moment is ONE
    // window of a few hundred thousand consecutive seeds -- a dozen subjects, which stay in every XCD's L2 (a contiguous
    // stretch per wave had 256 subjects in flight at once and every subject line came from HBM: 2.6 instead of ... ms)
    const int64_t nwaves = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t j_hi = P.n;
    const int gb = P.group_bits ? P.group_bits : 32;
    const int vb = P.ck_vbits;
    const uint64_t vmask = (1ull << vb) - 1ull;
    const uint32_t gmask = (gb >= 32) ? 0xffffffffu : ((1u << gb) - 1u);
    const int32_t t4 = P.score_table[0], dt = P.score_table[1] - t4;
    GbnSeedExt *__restrict__ rec = reinterpret_cast<GbnSeedExt *>(P.ext_rec);
    // the keys of a round are asked for a round ahead: everything else a seed loads hangs on its key, and the keys stream
    // from HBM (the neighbours' keys come from the neighbouring lanes; the two outer lanes read theirs, in one load)
    auto keys_of = [&](int64_t jb, uint64_t &w0, uint64_t &edge) {
        const int64_t j = min(jb + lane, j_hi - 1);
        w0 = P.key_group[j];
        edge = 0;
        if (lane == 0 || lane == 63) edge = P.key_group[lane == 0 ? (j > 0 ? j - 1 : 0) : (j + 1 < P.n ? j + 1 : j)];
    };
    uint64_t w0n = 0, edgen = 0;
    if (w * 64 < j_hi) keys_of(w * 64, w0n, edgen);
    for (int64_t jb = w * 64; jb < j_hi; jb += nwaves * 64) {
        const int64_t j = min(jb + lane, j_hi - 1);             // (lanes past the end redo the last seed and store nothing)
        const bool live = jb + lane < j_hi;
        // ---- batch 1: the key and its neighbours
        const uint64_t w0 = w0n, edge = edgen;
        if (jb + nwaves * 64 < j_hi) keys_of(jb + nwaves * 64, w0n, edgen);
        uint64_t wm = __shfl_up(w0, 1), wp = __shfl_down(w0, 1);
        if (lane == 0) wm = edge;
        if (lane == 63) wp = edge;
        const uint64_t key = w0 >> vb;
        const uint32_t val = vb ? (uint32_t)(w0 & vmask) : P.idx[j];
        const uint32_t qk = val >> 8;
        int64_t a = j, e = j + 1;
        int64_t pos = j;
        const bool tie = (j > 0 && (wm >> vb) == key) || (j + 1 < P.n && (wp >> vb) == key);
        uint64_t before = wm, after = wp;                       // the keys in front of / behind the group of equal keys
        if (tie) {      // seeds of one (subject, slot, scan position): a handful per million, ordered by the query key's high bits
            while (a > 0 && (P.key_group[a - 1] >> vb) == key) a--;
            while (e < P.n && (P.key_group[e] >> vb) == key) e++;
            int64_t rank = 0;
            for (int64_t m = a; m < e; m++) {
                if (m == j) continue;
                const uint32_t qm = (vb ? (uint32_t)(P.key_group[m] & vmask) : P.idx[m]) >> 8;
                rank += (qm < qk || (qm == qk && m < j)) ? 1 : 0;
            }
            pos = a + rank;
            before = P.key_group[a > 0 ? a - 1 : 0]; after = P.key_group[e < P.n ? e : P.n - 1];
        }
        const uint64_t run = key >> P.ck_shift;
        const bool head = live && pos == a && (a == 0 || (before >> (vb + P.ck_shift)) != run);
        const bool last = pos == e - 1 && (e >= P.n || (after >> (vb + P.ck_shift)) != run);
        const int32_t subj_id = (int32_t)(run >> gb) + P.ck_subj_base;
        const uint32_t slot = (uint32_t)run & gmask;
        const int32_t s_scan = (int32_t)(key & ((1ull << P.ck_s_bits) - 1ull));
        const uint32_t ql = ((uint32_t)s_scan - slot) & gmask;  // q_pos modulo the number of slots
        int32_t q_pos;
        if (P.ck_q_desc) {
            const uint32_t qmax = (P.ck_q_bits >= 32) ? 0xffffffffu : ((1u << P.ck_q_bits) - 1u);
            q_pos = (int32_t)(qmax - ((P.ck_qh_bits ? (qk << gb) : 0u) | ((qmax - ql) & gmask)));
        } else q_pos = (int32_t)((P.ck_qh_bits ? (qk << gb) : 0u) | ql);
        const int32_t ext_left = (int32_t)(val & 0xffu);
        const int32_t q_off = q_pos - ext_left, s_off = s_scan - ext_left;
        // ---- batch 2: what the key determines -- subject offsets, the context block, the first round of query bytes either side
        const int32_t len4 = (4 - (s_off & 3)) & 3;
        const int32_t q_ext = q_off + len4, s_ext = s_off + len4;
        const int64_t boff = P.byte_off[subj_id];
        const int32_t slen = P.len[subj_id];
        int32_t cb[2];
        __builtin_memcpy(cb, P.ctx_blk + 2 * (q_off >> P.ctx_hint_shift), 8);
        // (the sixteen steps of the two rounds either side of q_ext: sixteen consecutive bytes of one plane; readable
        // whatever the seed: 64 positions of padding either side)
        uint32_t qq[4];
        __builtin_memcpy(qq, q4_at(P, q_ext - 32), 16);
        // ---- batch 3: the subject's first rounds (16 padding bytes in front of every subject, 64 behind), the context's numbers
        const uint8_t *__restrict__ subj = P.db + boff;
        uint32_t sw[4];
        __builtin_memcpy(sw, subj + ((s_ext - 32) >> 2), 16);  // the 64 subject bases of the same two rounds
        ApproxPre pl, pr;
        pl.q0 = qq[0]; pl.q1 = qq[1]; pr.q0 = qq[2]; pr.q1 = qq[3];
        pl.s0 = sw[0]; pl.s1 = sw[1]; pr.s0 = sw[2]; pr.s1 = sw[3];
        int lo;
        if (cb[1] == INT32_MIN) lo = context_of(P, q_off);
        else lo = cb[0] + (q_off >= cb[1] ? 1 : 0);
        // (a lane that has dropped out keeps a sum of INT32_MIN / 2 for the rest of its round: any drop-off a score of
        // this path can reach -- |penalty| <= 127 over sequences of < 2^21 bases -- is far above it)
        int32_t cx[4];
        if (P.ctx_pack) __builtin_memcpy(cx, P.ctx_pack + 4 * lo, 16);
        else { cx[0] = P.ctx_xdrop[lo]; cx[1] = P.ctx_reduced[lo]; cx[2] = P.ctx_cutoff[lo]; }
        const int32_t X = max(-cx[0], -(1 << 28)), reduced = cx[1], cutoff = cx[2];
        // ---- the extension
        GbnSeedExt r; GbnSeedHsp hs;
        r.q_off = q_off; r.s_off = s_off; r.s_orig = s_off; r.flags = last ? 4 : 0;
        const int32_t s_match_end = s_off + P.word;
        bool want_exact = false;
        {
            int32_t score = 0, bl = 0, br = 0;
            approx_side<true>(P, subj, q_ext, s_ext, min(q_ext, s_ext) >> 2, X, t4, dt, score, bl, true, pl);
            approx_side<false>(P, subj, q_ext, s_ext, min(P.qlen - q_ext, slen - s_ext) >> 2, X, t4, dt, score, br, true, pr);
            const int32_t uq = q_ext - 4 * bl, us = s_ext - 4 * bl;
            const int32_t new_q = br ? q_ext + 4 * br - 1 : q_ext;
            if (score >= reduced) {
                if (P.exact_list) {                             // left to seed_exact_kernel: listed below, not saved here
                    want_exact = live;
                    hs.q_start = uq; hs.s_start = us; hs.score = INT32_MIN; hs.length = 0;
                } else
                hs = exact_from_windows(P, subj, slen, q_off, s_off, len4, q_ext, X, pl, pr);
            } else {
                hs.q_start = uq; hs.s_start = us; hs.score = score;
                hs.length = max(s_match_end - us, new_q - uq + 1);
            }
        }
        if (hs.score >= cutoff) r.flags |= 2;
        if (live) {
            rec[pos] = r;
            if (r.flags & 2) (reinterpret_cast<GbnSeedHsp *>(rec + P.n))[pos] = hs;
        }
        // ---- seeds for the exact pass (seed_exact_kernel), listed like the run heads
        const unsigned long long mx = __ballot(want_exact);
        if (mx) {
            if (want_exact) xb[nx + __popcll(mx & lt)] = (uint32_t)pos;
            nx += __popcll(mx);
            if (nx > HB - 64) flush_exact();
        }
        // ---- run heads: into the wave's buffer, out of it when the next 64 seeds might not fit
        const unsigned long long mh = __ballot(head);
        if (mh) {
            if (head) hb[nh + __popcll(mh & lt)] = (uint32_t)pos;
            nh += __popcll(mh);
            if (nh > HB - 64) flush();
        }
    }
    if (nh) flush();
    if (nx) flush_exact();
}

// The exact pass of the seeds seed_ext_ck_kernel listed (one in six on C3), a thread each: inside that kernel the pass
// was half of its instructions with a sixth of the lanes at work, and the kernel is bound by VALU issue.  Saved seeds get
// their record's flag and their extension here, before the replay reads them.
extern "C" __global__ void __launch_bounds__(256) seed_exact_kernel(GbnExtParams P)
{
    const uint32_t n = *P.exact_count;
    const int gb = P.group_bits ? P.group_bits : 32;
    GbnSeedExt *__restrict__ rec = reinterpret_cast<GbnSeedExt *>(P.ext_rec);
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        const uint32_t pos = P.exact_list[t];
        const GbnSeedExt r = rec[pos];
        const uint64_t run = (P.key_group[pos] >> P.ck_vbits) >> P.ck_shift;       // (a group of equal keys shares subject and slot)
        const int32_t subj_id = (int32_t)(run >> gb) + P.ck_subj_base;
        const int32_t q_off = r.q_off, s_off = r.s_off;
        const int32_t len4 = (4 - (s_off & 3)) & 3;
        const int32_t q_ext = q_off + len4, s_ext = s_off + len4;
        const uint8_t *__restrict__ subj = P.db + P.byte_off[subj_id];
        const int32_t slen = P.len[subj_id];
        int32_t cb[2];
        __builtin_memcpy(cb, P.ctx_blk + 2 * (q_off >> P.ctx_hint_shift), 8);
        uint32_t qq[4], sw[4];
        __builtin_memcpy(qq, q4_at(P, q_ext - 32), 16);
        __builtin_memcpy(sw, subj + ((s_ext - 32) >> 2), 16);
        ApproxPre pl, pr;
        pl.q0 = qq[0]; pl.q1 = qq[1]; pr.q0 = qq[2]; pr.q1 = qq[3];
        pl.s0 = sw[0]; pl.s1 = sw[1]; pr.s0 = sw[2]; pr.s1 = sw[3];
        const int lo = cb[1] == INT32_MIN ? context_of(P, q_off) : cb[0] + (q_off >= cb[1] ? 1 : 0);
        const int32_t X = max(-P.ctx_xdrop[lo], -(1 << 28)), cutoff = P.ctx_cutoff[lo];
        const GbnSeedHsp hs = exact_from_windows(P, subj, slen, q_off, s_off, len4, q_ext, X, pl, pr);
        if (hs.score >= cutoff) {
            (reinterpret_cast<GbnSeedHsp *>(rec + P.n))[pos] = hs;
            rec[pos].flags = r.flags | 2;
        }
    }
}

extern "C" __global__ void __launch_bounds__(64) diag_replay_kernel(GbnExtParams P)
{
    // initial hits of the wave's 64 runs are collected in LDS and handed over with one atomic on the global counter
    constexpr uint32_t CAP = 192;
    __shared__ GbnDevInitHit s_hit[CAP];
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nruns = (int64_t)*P.run_count;
    if (t < nruns) {
        const int64_t i = P.run_heads[t];
        const bool ck = P.ck_shift > 0;                          // composite keys: the records carry the run's end, no key is read on the way
        const uint64_t key = P.key_group[i] >> (ck ? P.ck_shift + P.ck_vbits : 0);    // (run_heads is not in run order: the run ends where the key changes)
        const int32_t subj_id = (int32_t)(key >> (P.group_bits ? P.group_bits : 32)) + (ck ? P.ck_subj_base : 0);
        const GbnSeedExt *__restrict__ rec = reinterpret_cast<const GbnSeedExt *>(P.ext_rec);
        const int word = P.word;
        const int32_t win = min(0, -word) + 1;                  // s_BlastDiagHashInsert with window = 0 + MIN(0, -word) + 1
        const bool hash = P.container_hash != 0;
        int32_t last_hit = 0;           // array container: one slot per run
        // hash container: the chain of the run's bucket, newest cell last.  The first KC cells live in registers
        // (cells expire as the scan moves on and are reused: a chain is one or two cells long), the rest in the
        // run's slice of the scratch arrays
        constexpr int KC = 4;
        int32_t cd[KC], cl[KC];
        #pragma unroll
        for (int c = 0; c < KC; c++) { cd[c] = 0; cl[c] = 0; }
        int32_t ncell = 0;
        // A run's records are read a 64-byte line (four records) at a time, the next line in flight: read one by one
        // every record was a 64-lane gather of its own and its line came back from HBM up to four times (the lines of
        // the 400,000 runs in flight do not fit any cache: FETCH_SIZE 1.9 GB for 0.75 GB of records, 0.79 ms)
        const uint4 *__restrict__ rec4 = reinterpret_cast<const uint4 *>(P.ext_rec);
        const int64_t g_last = (P.n - 1) >> 2;
        auto load_line = [&](int64_t g, uint4 (&L)[4]) {
            const int64_t gg = min(g, g_last);                  // (always records of this launch: past the last one, the last one again)
            #pragma unroll
            for (int k = 0; k < 4; k++) L[k] = rec4[min(gg * 4 + k, P.n - 1)];
        };
        int64_t j = i;
        bool done = false;
        uint4 cur[4], nxt[4];
        load_line(j >> 2, cur);
        auto one = [&](const uint4 &w) {
            GbnSeedExt r; r.q_off = (int32_t)w.x; r.s_off = (int32_t)w.y; r.s_orig = (int32_t)w.z; r.flags = (int32_t)w.w;
            const int64_t jn = j + 1 < P.n ? j + 1 : j;
            const uint64_t key_n = ck ? key : P.key_group[jn];
            const int32_t diag = r.s_off - r.q_off, s_off_pos = r.s_orig;       // the container is keyed by the word as the scan delivered it
            if (hash) {
                last_hit = 0; bool found = false;
                for (int32_t c = ncell - 1; c >= KC && !found; c--)
                    if (P.cell_diag[i + c] == diag) { last_hit = P.cell_level[i + c]; found = true; }
                #pragma unroll
                for (int c = KC - 1; c >= 0; c--)
                    if (!found && c < ncell && cd[c] == diag) { last_hit = cl[c]; found = true; }
            }
            if (!(s_off_pos < last_hit) && !(r.flags & 1)) {
                int32_t s_end_pos = s_off_pos + word + (r.flags >> 8);
                if (r.flags & 2) {
                    const GbnSeedHsp hs = (reinterpret_cast<const GbnSeedHsp *>(rec + P.n))[j];
                    GbnDevInitHit h; h.subj = subj_id; h.q_off = r.q_off; h.s_off = r.s_off;
                    h.q_start = hs.q_start; h.s_start = hs.s_start; h.length = hs.length; h.score = hs.score;
                    h.seq = (uint32_t)j;
                    const uint32_t slot = atomicAdd(&s_n, 1u);
                    if (slot < CAP) s_hit[slot] = h;
                    else { const unsigned long long o = atomicAdd(P.ihit_count, 1ull); if (o < P.ihit_cap) P.ihits[o] = h; }
                    s_end_pos = hs.length + hs.s_start;
                }
                if (hash) {
                    // newest to oldest: the cell of this diagonal, else the first expired one; none: a new cell
                    bool placed = false;
                    for (int32_t c = ncell - 1; c >= KC && !placed; c--) {
                        if (P.cell_diag[i + c] == diag) { P.cell_level[i + c] = s_end_pos; placed = true; }
                        else if (s_off_pos - P.cell_level[i + c] > win) { P.cell_diag[i + c] = diag; P.cell_level[i + c] = s_end_pos; placed = true; }
                    }
                    #pragma unroll
                    for (int c = KC - 1; c >= 0; c--) {
                        const bool here = !placed && c < ncell && (cd[c] == diag || s_off_pos - cl[c] > win);
                        cd[c] = here ? diag : cd[c]; cl[c] = here ? s_end_pos : cl[c]; placed = placed || here;
                    }
                    if (!placed) {
                        #pragma unroll
                        for (int c = 0; c < KC; c++) { const bool here = c == ncell; cd[c] = here ? diag : cd[c]; cl[c] = here ? s_end_pos : cl[c]; }
                        if (ncell >= KC) { P.cell_diag[i + ncell] = diag; P.cell_level[i + ncell] = s_end_pos; }
                        ncell++;
                    }
                } else {
                    last_hit = s_end_pos;
                }
            }
            if (jn == j || key_n != key || (r.flags & 4)) done = true;
            j = jn;
        };
        while (!done) {
            load_line((j >> 2) + 1, nxt);
            const int k0 = (int)(j & 3);
            #pragma unroll
            for (int k = 0; k < 4; k++) if (!done && k >= k0) one(cur[k]);
            #pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = nxt[k];
        }
    }
    __syncthreads();
    const uint32_t have = min(s_n, CAP);
    if (have) {
        if (threadIdx.x == 0) { const unsigned long long o = atomicAdd(P.ihit_count, (unsigned long long)have); s_base = (uint32_t)min(o, (unsigned long long)0xffffffffu); }
        __syncthreads();
        const unsigned long long base = s_base;
        for (uint32_t k = threadIdx.x; k < have; k += blockDim.x) if (base + k < P.ihit_cap) P.ihits[base + k] = s_hit[k];
    }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
namespace gbn {
hipError_t launch_seg_first(const GbnKeyParams &k, hipStream_t st);       // seed_order.hip

hipError_t launch_seed_keys(const GbnKeyParams &k, hipStream_t st)
{
    if (k.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(seed_keys_kernel, dim3((unsigned)((k.n + 255) / 256)), dim3(256), 0, st, k);
    return hipGetLastError();
}

hipError_t launch_seed_ckeys(const GbnKeyParams &k, hipStream_t st)
{
    if (k.n <= 0) return hipSuccess;
    if (k.nseg > 0) {
        if (!k.seg_first || k.nseg > GBN_SLICE_SEGS) return hipErrorInvalidValue;
        if (hipError_t e = launch_seg_first(k, st)) return e;
    }
    hipLaunchKernelGGL(seed_ckeys_kernel, dim3((unsigned)std::min<int64_t>((k.n + 255) / 256, 4096)), dim3(256), 0, st, k);
    return hipGetLastError();
}

hipError_t launch_group_keys(const GbnKeyParams &k, hipStream_t st)
{
    if (k.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)((k.n + 255) / 256)), dim3(256), 0, st, k);
    return hipGetLastError();
}

hipError_t launch_diag_ungapped(const GbnExtParams &p, hipStream_t st, GbnKernelTimer *kt)
{
    auto mark = [&](int t) { if (kt) kt->mark(t, st); };
    if (p.n <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(p.run_count, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    // Compacting the run heads first pays once most seeds are not heads (blastn word sizes: 6x on
    // C3); for the few ten thousand seeds of a megablast pass the direct form is 2x faster.
    // GBN_DIAG_COMPACT_MIN (environment): the threshold, for tests that send small inputs through the two-kernel form
    const int64_t compact_min = (int64_t)gbn::switch_value("GBN_DIAG_COMPACT_MIN", (long long)GBN_DIAG_COMPACT_MIN);
    if (p.n < compact_min) { GbnExtParams q = p; q.run_heads = nullptr;
        mark(GBN_KT_DIAG);
        hipLaunchKernelGGL(diag_ungapped_kernel, dim3((unsigned)((p.n + 63) / 64)), dim3(64), 0, st, q);
        mark(-1);
        return hipGetLastError(); }
    if (p.ck_shift > 0 && p.ext_rec) {      // composite keys: the extension kernel finds the run heads on its way
        // (hash container, word sizes from 11 up: the approximate extension; the mask re-check stays with the general kernel)
        const bool ck2 = gbn::switch_value("GBN_SEED_EXT_CK", 1) != 0;
        if (ck2 && p.q4 && p.ctx_blk && !p.masked && (p.container_hash || p.word >= 11)) {
            // a stretch of 64 x k seeds per wave: every wave slot of the chip taken, eight or more rounds per wave
            const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((p.n + 2047) / 2048, 256 * 32));
            if (p.exact_list) { e = hipMemsetAsync(p.exact_count, 0, sizeof(uint32_t), st); if (e != hipSuccess) return e; }
            mark(GBN_KT_SEED_EXT);
            hipLaunchKernelGGL(seed_ext_ck_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
            if (p.exact_list) hipLaunchKernelGGL(seed_exact_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((p.n + 1535) / 1536, 4096))), dim3(256), 0, st, p);
        } else {
        mark(GBN_KT_SEED_EXT);
        hipLaunchKernelGGL(seed_ext_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
        }
        mark(GBN_KT_REPLAY);
        hipLaunchKernelGGL(diag_replay_kernel, dim3((unsigned)((p.n + 63) / 64)), dim3(64), 0, st, p);
        mark(-1);
        return hipGetLastError();
    }
    mark(GBN_KT_REPLAY);
    hipLaunchKernelGGL(run_heads_kernel, dim3((unsigned)((p.n + 1023) / 1024)), dim3(1024), 0, st, p);
    // grid for the worst case (every seed its own run); threads past the run count leave at once
    if (p.ext_rec) {        // every seed extended by a thread of its own, then the runs replayed over the records
        mark(GBN_KT_SEED_EXT);
        hipLaunchKernelGGL(seed_ext_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
        mark(GBN_KT_REPLAY);
        hipLaunchKernelGGL(diag_replay_kernel, dim3((unsigned)((p.n + 63) / 64)), dim3(64), 0, st, p);
    } else {
        mark(GBN_KT_DIAG);
        hipLaunchKernelGGL(diag_ungapped_kernel, dim3((unsigned)((p.n + 63) / 64)), dim3(64), 0, st, p);
    }
    mark(-1);
    return hipGetLastError();
}


// (sort_keys_u64 / sort_pairs_u64: radix64.hip)

}  // namespace gbn
