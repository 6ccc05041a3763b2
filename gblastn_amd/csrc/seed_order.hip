// seed_order.hip -- the seeds of a scan in the order the diagonal filter visits them (gfx950 / CDNA4), for the scans that
// leave them in scan order: (subject, slot) groups, scan order inside -- the order in which s_BlastnDiagHashExtendInitialHit /
// s_BlastnDiagTableExtendInitialHit (CORE/na_ungapped.c:611-922) meet the seeds of one diagonal slot, subject by subject.
//
// Rounds 2-3 built a 64-bit key per seed (seed_ckeys_kernel) and had the library's radix sort order the keys on their
// subject | slot bits: one pass over the seeds plus three passes over the keys (8-bit digits; wider ones do not fit the
// library's rank tables into LDS -- tools/sort_digits.hip).  scan_fold_ordered_kernel's seeds come subject by subject
// already, so what is left is a stable partition of every subject's seeds by slot (512 slots for the hash container):
// a counting sort with one histogram per 4,096-seed chunk of a subject --
//   seed_order_plan_kernel      where every subject's seeds begin (the segments' first seeds, then a binary search inside
//                               one segment), and its chunks
//   seed_order_count_kernel     a chunk's seeds per slot                                   (reads the seeds: 16 B each)
//   seed_order_scan_kernel      per subject: prefix sums over (slot, chunk) -> where a chunk's seeds of a slot go
//   seed_order_scatter_kernel   a chunk's keys to their places: a wave ranks its 64 seeds of a round among themselves
//                               with ballots (stable: lane order = scan order), waves one after the other through
//                               per-wave counters in LDS                                   (reads the seeds again, writes 8 B keys)
// -- 40 bytes of traffic per seed instead of 88, no separate key kernel.  The keys written are bit for bit what the
// library sort left (a stable sort on the same bits).  HBM-bound integer work; no MFMA.
#include "gbn_dev.h"
#include <hip/hip_runtime.h>
#include <algorithm>

namespace {
constexpr int ORD_CH = GBN_ORDER_CHUNK, ORD_THREADS = 256, ORD_WAVES = ORD_THREADS / 64, ORD_ROUNDS = ORD_CH / ORD_THREADS;

// the segment that holds the seed with dense index i (first[g] <= i < first[g + 1]; empty segments are stepped over)
__device__ __forceinline__ int seg_locate(const unsigned long long *__restrict__ first, int nseg, unsigned long long i)
{
    int lo = 0, top = nseg;
    while (top - lo > 1) { const int m = (lo + top) >> 1; if (first[m] <= i) lo = m; else top = m; }
    return lo;
}

__device__ __forceinline__ uint32_t slot_of(const GbnKeyParams &K, const GbnDevSeed &sd)
{
    return K.container_hash ? ((uint32_t)(sd.s_scan - sd.q_pos) & 511u)
                            : ((uint32_t)(sd.s_scan + K.diag_len - sd.q_pos) & (uint32_t)(K.diag_len - 1));
}

// segments of 8-byte composite keys (K.seg_keys, round 6): element idx of the segment array, its subject (relative) and slot
// (a segment is seg_cap 16-byte elements long whatever it holds: key `off` of segment `sg` sits at 2 sg seg_cap + off)
__device__ __forceinline__ uint64_t seg_key(const GbnKeyParams &K, int sg, size_t off) { return reinterpret_cast<const uint64_t *>(K.seg)[(size_t)sg * K.seg_cap * 2 + off]; }
__device__ __forceinline__ int32_t key_subj_rel(const GbnKeyParams &K, uint64_t packed) { return (int32_t)(uint32_t)(packed >> (K.v_bits + K.s_bits + K.group_bits)); }
__device__ __forceinline__ uint32_t key_slot(const GbnKeyParams &K, uint64_t packed) { return (uint32_t)(packed >> (K.v_bits + K.s_bits)) & ((1u << K.group_bits) - 1u); }
__device__ __forceinline__ int32_t seg_subj_rel(const GbnKeyParams &K, int sg, size_t off) { return K.seg_keys ? key_subj_rel(K, seg_key(K, sg, off)) : K.seg[(size_t)sg * K.seg_cap + off].subj - K.subj_base; }

// a chunk's subject and its stretch [i0, i1) of the seeds (uniform over the workgroup: scalar loads)
__device__ __forceinline__ bool chunk_of(const GbnOrderParams &O, uint32_t c, int &s, uint32_t &i0, uint32_t &i1)
{
    if (c >= O.chunk_first[O.nsubj]) return false;
    int lo = 0, top = O.nsubj;                      // last subject with chunk_first[s] <= c (subjects without seeds have no chunk)
    while (top - lo > 1) { const int m = (lo + top) >> 1; if (O.chunk_first[m] <= c) lo = m; else top = m; }
    s = lo;
    i0 = O.subj_first[s] + (c - O.chunk_first[s]) * (uint32_t)ORD_CH;
    i1 = min(i0 + (uint32_t)ORD_CH, O.subj_first[s + 1]);
    return true;
}
}  // namespace

// The index of every segment's first seed, first[nseg] = their number (one small workgroup, so that it
// finds room next to the gapped stage of the range before, whose waves fill the CUs: a 1024-thread workgroup waited
// 0.8 ms for a CU to itself)
extern "C" __global__ void __launch_bounds__(256) seg_first_kernel(const uint32_t *seg_count, int nseg, uint32_t seg_cap, unsigned long long *first)
{
    constexpr int PER = (GBN_SLICE_SEGS + 255) / 256;
    __shared__ unsigned long long s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c[PER]; unsigned long long sum = 0;
    #pragma unroll
    for (int k = 0; k < PER; k++) { const int sg = tid * PER + k; c[k] = sg < nseg ? min(seg_count[sg], seg_cap) : 0u; }
    #pragma unroll
    for (int k = 0; k < PER; k++) sum += c[k];
    unsigned long long incl = sum;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long v = __shfl_up(incl, d); incl += (lane >= d) ? v : 0ull; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long at = incl - sum, total = 0;
    #pragma unroll
    for (int w = 0; w < 4; w++) { at += (w < wave) ? s_wave[w] : 0ull; total += s_wave[w]; }
    #pragma unroll
    for (int k = 0; k < PER; k++) { const int sg = tid * PER + k; if (sg < nseg) first[sg] = at; at += c[k]; }
    if (tid == 0) first[nseg] = total;
}

// One SMALL workgroup and next to no LDS (it runs next to the gapped stage of the range before, whose waves fill the CUs
// and their LDS: as 1,024 threads with its tables in LDS this kernel waited a millisecond for a CU, 40 us of work).
// subj_first[s] = dense index of subject s's first seed (s counts from K.subj_base; [nsubj] = n), chunk_first[s] = number
// of chunks of the subjects before s; seg_next = scratch.
extern "C" __global__ void __launch_bounds__(256) seed_order_plan_kernel(GbnOrderParams O)
{
    __shared__ int32_t s_wmin[4];
    __shared__ uint32_t s_wave[4];
    const GbnKeyParams &K = O.K;
    const unsigned long long *__restrict__ first = K.seg_first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // seg_next[g] = subject of the first seed at or behind segment g's start: the segments' first subjects (they ascend),
    // a suffix minimum over the segments that have seeds -- sixteen consecutive segments per thread
    constexpr int GP = GBN_SLICE_SEGS / 256;
    int32_t v[GP];
    #pragma unroll
    for (int k = 0; k < GP; k++) {
        const int g = tid * GP + k;
        v[k] = (g < K.nseg && first[g + 1] > first[g]) ? seg_subj_rel(K, g, 0) : INT32_MAX;
    }
    #pragma unroll
    for (int k = GP - 2; k >= 0; k--) v[k] = min(v[k], v[k + 1]);
    int32_t m = v[0];
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t x = __shfl_down(m, d); if (lane + d < 64) m = min(m, x); }
    if (lane == 0) s_wmin[wave] = m;
    __syncthreads();
    int32_t behind = __shfl_down(m, 1);
    if (lane == 63) behind = INT32_MAX;
    #pragma unroll
    for (int w = 1; w < 4; w++) if (w > wave) behind = min(behind, s_wmin[w]);
    #pragma unroll
    for (int k = 0; k < GP; k++) O.seg_next[tid * GP + k] = min(v[k], behind);
    __syncthreads();
    const int32_t *__restrict__ s_next = O.seg_next;
    for (int s = tid; s <= O.nsubj; s += 256) {
        uint32_t at = (uint32_t)K.n;
        if (s < O.nsubj) {
            int lo = 0, hi = K.nseg;                    // number of segments that begin with a seed of an earlier subject
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_next[mid] < s) lo = mid + 1; else hi = mid; }
            if (lo == 0) at = 0;
            else {
                const int g = lo - 1;                   // its seeds begin inside segment g (or right behind it)
                uint32_t a = 1, b = (uint32_t)(first[g + 1] - first[g]);
                while (a < b) { const uint32_t mid = (a + b) >> 1; if (seg_subj_rel(K, g, mid) < s) a = mid + 1; else b = mid; }
                at = (uint32_t)first[g] + a;
            }
        }
        O.subj_first[s] = at;
    }
    __syncthreads();
    // chunks per subject, exclusive prefix sums (sixteen consecutive subjects per thread)
    constexpr int SP = GBN_ORDER_MAX_SUBJ / 256;
    const uint32_t *sf = O.subj_first;
    uint32_t c[SP], sum = 0;
    #pragma unroll
    for (int k = 0; k < SP; k++) { const int s = tid * SP + k; c[k] = s < O.nsubj ? (sf[s + 1] - sf[s] + ORD_CH - 1) / ORD_CH : 0u; sum += c[k]; }
    uint32_t incl = sum;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d); incl += (lane >= d) ? x : 0u; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t at = incl - sum, total = 0;
    #pragma unroll
    for (int w = 0; w < 4; w++) { at += (w < wave) ? s_wave[w] : 0u; total += s_wave[w]; }
    #pragma unroll
    for (int k = 0; k < SP; k++) { const int s = tid * SP + k; if (s < O.nsubj) O.chunk_first[s] = at; at += c[k]; }
    if (tid == 0) O.chunk_first[O.nsubj] = total;
}

extern "C" __global__ void __launch_bounds__(ORD_THREADS) seed_order_count_kernel(GbnOrderParams O)
{
    extern __shared__ uint32_t s_hist[];            // a counter per slot (as little LDS as the container needs: the gapped stage of the range before fills the CUs' LDS)
    const GbnKeyParams &K = O.K;
    const unsigned long long *__restrict__ first = K.seg_first;
    int s; uint32_t i0, i1;
    if (!chunk_of(O, blockIdx.x, s, i0, i1)) return;
    const int nslots = 1 << K.group_bits, tid = threadIdx.x;
    for (int t = tid; t < nslots; t += ORD_THREADS) s_hist[t] = 0;
    __syncthreads();
    int sgi = seg_locate(first, K.nseg, i0);
    unsigned long long lo = first[sgi], hi = first[sgi + 1];
    uint32_t slot[ORD_ROUNDS];
    #pragma unroll
    for (int r = 0; r < ORD_ROUNDS; r++) {
        const uint32_t i = i0 + (uint32_t)(r * ORD_THREADS + tid);
        slot[r] = 0;
        if (i >= i1) continue;
        while ((unsigned long long)i >= hi) { sgi++; lo = hi; hi = first[sgi + 1]; }
        slot[r] = K.seg_keys ? key_slot(K, seg_key(K, sgi, (size_t)(i - lo))) : slot_of(K, K.seg[(size_t)sgi * K.seg_cap + (size_t)(i - lo)]);      // (8 bytes per seed instead of 16)
    }
    #pragma unroll
    for (int r = 0; r < ORD_ROUNDS; r++)
        if (i0 + (uint32_t)(r * ORD_THREADS + tid) < i1) atomicAdd(&s_hist[slot[r]], 1u);
    __syncthreads();
    uint32_t *__restrict__ row = O.counts + (size_t)blockIdx.x * nslots;
    for (int t = tid; t < nslots; t += ORD_THREADS) row[t] = s_hist[t];
}

// One workgroup per subject: counts[c][t] becomes the number of the subject's seeds of slot t in the chunks before c,
// slot_base[s][t] = where the subject's seeds of slot t begin in the output.
// (round 6: 256 threads a workgroup instead of 1,024, so that one fits beside whatever else a CU holds; no measurable change at C3,
// 26.8 ms either way -- the kernel's 0.4-0.6 ms per range are its chunk-serial walk over `counts`, not its dispatch)
extern "C" __global__ void __launch_bounds__(256) seed_order_scan_kernel(GbnOrderParams O)
{
    __shared__ uint32_t s_tot[GBN_ORDER_MAX_SLOTS];
    __shared__ uint32_t s_wave[4];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nslots = 1 << O.K.group_bits;
    const uint32_t c0 = O.chunk_first[s], c1 = O.chunk_first[s + 1];
    for (int t = tid; t < nslots; t += 256) {
        uint32_t run = 0;
        uint32_t c = c0;
        for (; c + 8 <= c1; c += 8) {
            uint32_t v[8];
            #pragma unroll
            for (int k = 0; k < 8; k++) v[k] = O.counts[(size_t)(c + k) * nslots + t];
            #pragma unroll
            for (int k = 0; k < 8; k++) { O.counts[(size_t)(c + k) * nslots + t] = run; run += v[k]; }
        }
        for (; c < c1; c++) { const uint32_t v = O.counts[(size_t)c * nslots + t]; O.counts[(size_t)c * nslots + t] = run; run += v; }
        s_tot[t] = run;
    }
    __syncthreads();
    constexpr int TP = GBN_ORDER_MAX_SLOTS / 256;
    uint32_t v[TP], sum = 0;
    #pragma unroll
    for (int k = 0; k < TP; k++) { const int t = tid * TP + k; v[k] = t < nslots ? s_tot[t] : 0u; sum += v[k]; }
    uint32_t incl = sum;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d); incl += (lane >= d) ? x : 0u; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t at = O.subj_first[s] + incl - sum;
    #pragma unroll
    for (int w = 0; w < 4; w++) at += (w < wave) ? s_wave[w] : 0u;
    #pragma unroll
    for (int k = 0; k < TP; k++) { const int t = tid * TP + k; if (t < nslots) O.slot_base[(size_t)s * nslots + t] = at; at += v[k]; }
}

extern "C" __global__ void __launch_bounds__(ORD_THREADS) seed_order_scatter_kernel(GbnOrderParams O)
{
    extern __shared__ uint32_t s_cnt[];             // [wave][slot]: seeds of the slot the wave has ranked so far; then where its next one goes
    const GbnKeyParams &K = O.K;
    const unsigned long long *__restrict__ first = K.seg_first;
    int s; uint32_t i0, i1;
    if (!chunk_of(O, blockIdx.x, s, i0, i1)) return;
    const int nslots = 1 << K.group_bits, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < ORD_WAVES * nslots; t += ORD_THREADS) s_cnt[t] = 0;
    __syncthreads();
    // wave w takes the seeds [w, w + 1) * ORD_CH / ORD_WAVES of the chunk, 64 consecutive ones per round
    const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i0 + (uint32_t)wave * (ORD_CH / ORD_WAVES)));
    const uint32_t w0 = wbase + (uint32_t)lane;
    int sgi = seg_locate(first, K.nseg, min(wbase, i1 - 1));           // (uniform over the wave: scalar loads; the lanes step forward from there)
    unsigned long long lo = first[sgi], hi = first[sgi + 1];
    const uint32_t qmax = (K.q_bits >= 32) ? 0xffffffffu : ((1u << K.q_bits) - 1u);
    uint64_t key[ORD_ROUNDS]; uint32_t slots[ORD_ROUNDS];
    #pragma unroll
    for (int r = 0; r < ORD_ROUNDS; r++) {
        const uint32_t i = w0 + (uint32_t)(r * 64);
        key[r] = 0; slots[r] = 0;
        if (i >= i1) continue;
        while ((unsigned long long)i >= hi) { sgi++; lo = hi; hi = first[sgi + 1]; }
        if (K.seg_keys) { key[r] = seg_key(K, sgi, (size_t)(i - lo)); slots[r] = key_slot(K, key[r]); }        // (the scan wrote the key: 8 bytes read, 8 written)
        else { uint32_t val; key[r] = (gbn_composite_key(K, K.seg[(size_t)sgi * K.seg_cap + (size_t)(i - lo)], qmax, slots[r], val) << K.v_bits) | val; }
    }
    uint32_t *__restrict__ mine = s_cnt + wave * nslots;
    uint32_t sr[ORD_ROUNDS];                                    // slot << 16 | rank among the wave's seeds of the slot
    #pragma unroll
    for (int r = 0; r < ORD_ROUNDS; r++) {
        const bool valid = w0 + (uint32_t)(r * 64) < i1;
        const uint32_t slot = slots[r];
        // the lanes of this round with the same slot (a ballot per slot bit), in lane order = scan order
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < K.group_bits; b++) {
            const bool bit = (slot >> b) & 1u;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        sr[r] = 0;
        if (valid) {
            const uint32_t prior = mine[slot];
            if (before == 0) mine[slot] = prior + (uint32_t)__popcll(peers);
            sr[r] = slot << 16 | (prior + before);
        }
    }
    __syncthreads();
    // counters -> places: the chunk's seeds of a slot begin at slot_base + counts, wave after wave
    const uint32_t *__restrict__ base = O.slot_base + (size_t)s * nslots;
    const uint32_t *__restrict__ row = O.counts + (size_t)blockIdx.x * nslots;
    for (int t = tid; t < nslots; t += ORD_THREADS) {
        uint32_t run = base[t] + row[t];
        #pragma unroll
        for (int w = 0; w < ORD_WAVES; w++) { const uint32_t v = s_cnt[w * nslots + t]; s_cnt[w * nslots + t] = run; run += v; }
    }
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < ORD_ROUNDS; r++)
        if (w0 + (uint32_t)(r * 64) < i1) K.key_scan[mine[sr[r] >> 16] + (sr[r] & 0xffffu)] = key[r];
}

namespace gbn {
hipError_t launch_seg_first(const GbnKeyParams &k, hipStream_t st)
{
    hipLaunchKernelGGL(seg_first_kernel, dim3(1), dim3(256), 0, st, k.seg_count, k.nseg, k.seg_cap, const_cast<unsigned long long *>(k.seg_first));
    return hipGetLastError();
}

size_t seed_order_scratch_words(int64_t n, int nsubj, int group_bits)
{
    const size_t nslots = (size_t)1 << group_bits, chunks = (size_t)((n + ORD_CH - 1) / ORD_CH) + (size_t)nsubj;
    return GBN_SLICE_SEGS + 2 * ((size_t)nsubj + 1) + chunks * nslots + (size_t)nsubj * nslots;
}

// K: the segments and the key layout as for launch_seed_ckeys (v_bits > 0), key_scan = the ORDERED keys; scratch of
// seed_order_scratch_words(n, nsubj, group_bits) 32-bit words
hipError_t launch_seed_order(const GbnKeyParams &K, int nsubj, uint32_t *scratch, hipStream_t st)
{
    if (K.n <= 0) return hipSuccess;
    if (K.nseg <= 0 || K.nseg > GBN_SLICE_SEGS || !K.seg_first || K.v_bits <= 0 || nsubj <= 0 || nsubj > GBN_ORDER_MAX_SUBJ ||
        K.group_bits < 1 || (1 << K.group_bits) > GBN_ORDER_MAX_SLOTS || K.n >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    GbnOrderParams O; O.K = K; O.nsubj = nsubj;
    const size_t nslots = (size_t)1 << K.group_bits, chunks = (size_t)((K.n + ORD_CH - 1) / ORD_CH) + (size_t)nsubj;
    O.seg_next = reinterpret_cast<int32_t *>(scratch); O.subj_first = scratch + GBN_SLICE_SEGS; O.chunk_first = O.subj_first + nsubj + 1; O.counts = O.chunk_first + nsubj + 1; O.slot_base = O.counts + chunks * nslots;
    if (hipError_t e = launch_seg_first(K, st)) return e;
    hipLaunchKernelGGL(seed_order_plan_kernel, dim3(1), dim3(256), 0, st, O);
    hipLaunchKernelGGL(seed_order_count_kernel, dim3((unsigned)chunks), dim3(ORD_THREADS), nslots * sizeof(uint32_t), st, O);
    hipLaunchKernelGGL(seed_order_scan_kernel, dim3((unsigned)nsubj), dim3(256), 0, st, O);
    hipLaunchKernelGGL(seed_order_scatter_kernel, dim3((unsigned)chunks), dim3(ORD_THREADS), ORD_WAVES * nslots * sizeof(uint32_t), st, O);
    return hipGetLastError();
}
}  // namespace gbn
