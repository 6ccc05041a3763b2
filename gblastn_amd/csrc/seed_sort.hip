// seed_sort.hip -- the FEW seeds of a megablast-shaped range (tens of thousands per 50 Gbp pass) put into the order the
// diagonal filter replays them in: by (subject, diagonal slot), scan order (scan position, chain order) inside -- what the CPU
// scanner gets for free (CORE/blast_nascan.c:1489-1591 walks a subject from its first base; CORE/na_ungapped.c:1025-1144
// hands every seed to the container as it turns up).  Rounds 1-4 did this with two stable radix sorts of the library
// (hipCUB: two key kernels + ~16 launches per range, the last library kernels on the C2 path); here ONE launch of ONE
// workgroup sorts (key, seed index) pairs with a stable LSD radix sort of 6-bit digits -- every pass reads and writes the list
// coalesced (it stays in the L2), the first makes the keys from the seeds --, the 64-bit key being
//     subject | slot | scan position | high bits of the query key
// -- two seeds of one (subject, slot, scan position) agree in their query positions modulo the number of slots, so the query
// key's low bits decide nothing (as in gbn_composite_key).  4 KB of LDS (a histogram per wave): the workgroup finds room on a
// CU next to the probe kernel of the pass behind it, which leaves 8 KB.  Integer work only: no MFMA.
#include "gbn_dev.h"
#include <hip/hip_runtime.h>

namespace {
// (512 threads: two waves of 64 registers per SIMD fit next to a probe workgroup, which leaves 160 of a SIMD's 512 -- with 1,024
// threads the kernel found room only when it happened to be dispatched before the probe kernel of the pass behind it, and waited
// for that kernel's 3 ms when it was not)
constexpr int SS_THREADS = 512, SS_WAVES = SS_THREADS / 64, SS_BITS = 6, SS_DIGITS = 1 << SS_BITS, SS_DEEP = 8;
static_assert(SS_DIGITS * SS_WAVES == SS_THREADS, "one thread per (digit, wave) counter");

struct SeedSortParams {
    GbnKeyParams K;                 // seeds, n, the key layout (q_bits, group_bits, s_bits, qh_bits, subj_base, q_descending, container)
    int npass;                      // ceil(key bits / 6)
    uint64_t *key_ping, *key_pong;  // n keys each: the list travels as (key, index) pairs, every pass reads and writes it coalesced
    uint32_t *idx_ping, *idx_pong;  // n indices each (the launcher picks ping / pong so that the last pass writes the caller's arrays)
    uint64_t *key_group;            // out: (subject << group_bits | slot) of the seeds in sorted order (may be the array the last pass wrote its keys to)
};

__device__ __forceinline__ uint64_t sort_key(const GbnKeyParams &K, const GbnDevSeed &sd, uint32_t qmax)
{
    const uint32_t qkey = K.q_descending ? (qmax - (uint32_t)sd.q_pos) : (uint32_t)sd.q_pos;
    const uint32_t slot = K.container_hash ? ((uint32_t)(sd.s_scan - sd.q_pos) & 511u)
                                           : ((uint32_t)(sd.s_scan + K.diag_len - sd.q_pos) & (uint32_t)(K.diag_len - 1));
    uint64_t key = ((uint64_t)(uint32_t)(sd.subj - K.subj_base) << K.group_bits) | slot;
    key = (key << K.s_bits) | (uint32_t)sd.s_scan;
    return (key << K.qh_bits) | (K.qh_bits ? (qkey >> K.group_bits) : 0u);
}

// lanes of the wave whose digit equals this lane's (among the lanes of `valid`)
__device__ __forceinline__ unsigned long long same_digit(uint32_t d, bool valid)
{
    unsigned long long m = __ballot(valid);
    #pragma unroll
    for (int b = 0; b < SS_BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(SS_THREADS) seed_sort_small_kernel(SeedSortParams S)
{
    __shared__ uint32_t hist[SS_DIGITS][SS_WAVES];          // [digit][wave]: digit-major = the order of the prefix sum
    __shared__ uint32_t wtot[SS_WAVES];
    const GbnKeyParams &K = S.K;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t n = (uint32_t)K.n;
    const uint32_t per_wave = ((n + SS_WAVES - 1) / SS_WAVES + 63u) & ~63u;
    const uint32_t i0 = (uint32_t)w * per_wave, i1 = i0 < n ? min(n, i0 + per_wave) : i0;
    const uint32_t qmax = (K.q_bits >= 32) ? 0xffffffffu : ((1u << K.q_bits) - 1u);
    for (int p = 0; p < S.npass; p++) {
        const uint64_t *__restrict__ ksrc = (p & 1) ? S.key_pong : S.key_ping;
        const uint32_t *__restrict__ isrc = (p & 1) ? S.idx_pong : S.idx_ping;
        uint64_t *__restrict__ kdst = (p & 1) ? S.key_ping : S.key_pong;
        uint32_t *__restrict__ idst = (p & 1) ? S.idx_ping : S.idx_pong;
        const int shift = SS_BITS * p;
        hist[tid / SS_WAVES][tid % SS_WAVES] = 0;
        __syncthreads();
        // a wave walks its stretch of the list 64 elements at a time, SS_DEEP such batches loaded together (pass 0 makes the
        // keys from the seeds, element i = seed i)
        auto load = [&](uint32_t i, uint64_t &key, uint32_t &id) {
            if (i >= i1) { key = 0; id = 0; return; }
            if (p == 0) { key = sort_key(K, K.seeds[i], qmax); id = i; }
            else { key = ksrc[i]; id = isrc[i]; }
        };
        // ---- count
        for (uint32_t b0 = i0; b0 < i1; b0 += 64 * SS_DEEP) {
            uint64_t key[SS_DEEP]; uint32_t id[SS_DEEP];
            #pragma unroll
            for (int u = 0; u < SS_DEEP; u++) load(b0 + 64 * u + lane, key[u], id[u]);
            #pragma unroll
            for (int u = 0; u < SS_DEEP; u++) {
                const uint32_t i = b0 + 64 * u + lane;
                if (b0 + 64 * u >= i1) break;                                 // (uniform)
                const uint32_t d = (uint32_t)(key[u] >> shift) & (SS_DIGITS - 1);
                // (counting needs no ranks: one LDS atomic without return per element instead of the six ballots of same_digit --
                // the kernel is bound by the instructions ONE compute unit issues)
                if (i < i1) __hip_atomic_fetch_add(&hist[d][w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        // ---- exclusive prefix sum over (digit, wave), digit-major: where every wave's elements of every digit go
        {
            const uint32_t v = hist[tid / SS_WAVES][tid % SS_WAVES];
            uint32_t incl = v;
            #pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) wtot[w] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (int k = 0; k < w; k++) base += wtot[k];
            hist[tid / SS_WAVES][tid % SS_WAVES] = base + incl - v;
        }
        __syncthreads();
        // ---- scatter, in the same order (stable)
        for (uint32_t b0 = i0; b0 < i1; b0 += 64 * SS_DEEP) {
            uint64_t key[SS_DEEP]; uint32_t id[SS_DEEP];
            #pragma unroll
            for (int u = 0; u < SS_DEEP; u++) load(b0 + 64 * u + lane, key[u], id[u]);
            #pragma unroll
            for (int u = 0; u < SS_DEEP; u++) {
                const uint32_t i = b0 + 64 * u + lane;
                if (b0 + 64 * u >= i1) break;
                const bool valid = i < i1;
                const uint32_t d = (uint32_t)(key[u] >> shift) & (SS_DIGITS - 1);
                const unsigned long long m = same_digit(d, valid);
                uint32_t at = 0;
                if (valid) at = hist[d][w];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // (every lane of a group has read the counter before its first lane moves it on)
                if (valid) {
                    const uint32_t o = at + (uint32_t)__popcll(m & lt);
                    kdst[o] = key[u]; idst[o] = id[u];
                    if (!(m & lt)) hist[d][w] = at + (uint32_t)__popcll(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        __syncthreads();                                                    // (workgroup scope: what was written is read by other waves in the next pass)
    }
    // the last pass wrote the caller's index array; the run keys (subject << group_bits | slot) follow from the sort keys
    const uint64_t *kfin = (S.npass & 1) ? S.key_pong : S.key_ping;      // (may be the array the run keys go to: element by element, in place)
    const int low = K.qh_bits + K.s_bits;
    for (uint32_t i = tid; i < n; i += SS_THREADS) {
        const uint64_t k = (S.npass ? kfin[i] : sort_key(K, K.seeds[i], qmax)) >> low;
        S.key_group[i] = (((k >> K.group_bits) + (uint64_t)(uint32_t)K.subj_base) << K.group_bits) | (k & (((uint64_t)1 << K.group_bits) - 1));
    }
}

namespace gbn {
// can the small sort take this launch?  (n seeds, subjects [subj_base, subj_base + nsubj))
bool seed_sort_small_fits(const GbnKeyParams &K, int nsubj)
{
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    return K.n > 0 && K.n <= GBN_SMALL_SORT_MAX && K.group_bits < 32 && K.qh_bits + K.s_bits + K.group_bits + bits_for((uint64_t)nsubj + 1) <= 64;
}
// idx_out[i] = index of the i-th seed in (subject, slot, scan position, query key) order, key_group_out[i] = its
// (subject << group_bits | slot); scratch: key_tmp n keys, idx_tmp n indices (key_group_out doubles as the other key array).
// One launch.
hipError_t launch_seed_sort_small(const GbnKeyParams &K, int nsubj, uint32_t *idx_out, uint32_t *idx_tmp, uint64_t *key_group_out, uint64_t *key_tmp, hipStream_t st)
{
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    SeedSortParams S;
    S.K = K;
    const int key_bits = K.qh_bits + K.s_bits + K.group_bits + bits_for((uint64_t)nsubj + 1);
    S.npass = (key_bits + SS_BITS - 1) / SS_BITS;
    // pass p writes `pong` when p is even: after npass passes the result is in pong (npass odd) or ping (npass even)
    if (S.npass & 1) { S.idx_pong = idx_out; S.idx_ping = idx_tmp; S.key_pong = key_group_out; S.key_ping = key_tmp; }
    else { S.idx_ping = idx_out; S.idx_pong = idx_tmp; S.key_ping = key_group_out; S.key_pong = key_tmp; }
    S.key_group = key_group_out;
    hipLaunchKernelGGL(seed_sort_small_kernel, dim3(1), dim3(SS_THREADS), 0, st, S);
    return hipGetLastError();
}
}  // namespace gbn
