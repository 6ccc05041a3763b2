// radix64.hip -- stable LSD radix sort of 64-bit keys (with or without a 32-bit value) on a range of their bits: the seeds of a
// subject range by composite key, by scan key and by (subject, slot) when the engine's shape-specific sorts do not apply
// (seed_sort.hip: up to 65,536 seeds in one workgroup; seed_order.hip: an ordered scan's seeds) -- rounds 1-5 called the library's
// radix sort here (hipCUB), the last library kernel of the engine.  What this replaces in the reference: the host sort of the hits
// a GPU scan leaves (GB/gpu_blastn_MB_and_smallNa.cu:1906).
//
// The pattern of lutbuild.hip's sort of 32-bit keys: per pass over an 8-bit digit a count per 4,096-element chunk, one prefix sum
// over (digit, chunk), and a scatter in which a wave ranks the 64 elements of a round among themselves with a ballot per digit bit
// (lane order = input order: stable) and the waves follow one another through per-wave counters in LDS.  No look-back chains
// between workgroups, 4 KB of LDS per workgroup: it finds room next to a probe kernel that owns the CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "lutbuild.h"

namespace {
constexpr int R_CHUNK = 4096, R_THREADS = 256, R_WAVES = R_THREADS / 64, R_ROUNDS = R_CHUNK / R_THREADS, R_BITS = 8, R_DIGITS = 1 << R_BITS;

__global__ void __launch_bounds__(R_THREADS) radix64_count_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift, uint32_t mask,
                                                                  uint32_t *__restrict__ counts, int64_t nchunks)
{
    __shared__ uint32_t s_hist[R_DIGITS];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * R_CHUNK;
    s_hist[tid] = 0;
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < R_ROUNDS; r++) {
        const int64_t i = base + r * R_THREADS + tid;
        if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    counts[(int64_t)tid * nchunks + blockIdx.x] = s_hist[tid];      // digit-major: one scan over the table gives every (digit, chunk) its place
}

template <bool PAIRS>
__global__ void __launch_bounds__(R_THREADS) radix64_scatter_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                    uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
                                                                    int shift, uint32_t mask, const uint32_t *__restrict__ offsets, int64_t nchunks)
{
    __shared__ uint32_t s_cnt[R_WAVES][R_DIGITS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * R_CHUNK + (int64_t)wave * (R_CHUNK / R_WAVES) + lane;
    #pragma unroll
    for (int w = 0; w < R_WAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();
    uint64_t k[R_ROUNDS]; uint32_t v[R_ROUNDS], sr[R_ROUNDS];
    #pragma unroll
    for (int r = 0; r < R_ROUNDS; r++) { const int64_t i = w0 + r * 64; k[r] = i < n ? keys_in[i] : 0ull; v[r] = (PAIRS && i < n) ? vals_in[i] : 0u; }
    uint32_t *mine = s_cnt[wave];
    #pragma unroll
    for (int r = 0; r < R_ROUNDS; r++) {
        const bool valid = w0 + r * 64 < n;
        const uint32_t digit = (uint32_t)(k[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
        #pragma unroll
        for (int b = 0; b < R_BITS; b++) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        sr[r] = 0xffffffffu;
        if (valid) {
            const uint32_t prior = mine[digit];
            if (before == 0) mine[digit] = prior + (uint32_t)__popcll(peers);
            sr[r] = digit << 16 | (prior + before);
        }
    }
    __syncthreads();
    {
        uint32_t run = offsets[(int64_t)tid * nchunks + blockIdx.x];
        #pragma unroll
        for (int w = 0; w < R_WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < R_ROUNDS; r++)
        if (sr[r] != 0xffffffffu) { const uint32_t at = mine[sr[r] >> 16] + (sr[r] & 0xffffu); keys_out[at] = k[r]; if (PAIRS) vals_out[at] = v[r]; }
}

// tmp: counts | scan scratch | n keys | n values (pairs); null: the bytes it takes
hipError_t radix64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, int64_t n,
                   int begin_bit, int end_bit, bool pairs, hipStream_t st)
{
    const int64_t nchunks = (std::max<int64_t>(n, 1) + R_CHUNK - 1) / R_CHUNK, ncounts = (int64_t)R_DIGITS * nchunks;
    size_t scan_bytes = 0;
    (void)gbn::lut_scan(nullptr, scan_bytes, nullptr, nullptr, ncounts, st);
    const size_t counts_bytes = ((size_t)ncounts * 4 + 255) & ~(size_t)255, scan_al = (scan_bytes + 255) & ~(size_t)255;
    const size_t keys_bytes = ((size_t)std::max<int64_t>(n, 1) * 8 + 255) & ~(size_t)255, vals_bytes = pairs ? (((size_t)std::max<int64_t>(n, 1) * 4 + 255) & ~(size_t)255) : 0;
    const size_t need = counts_bytes + scan_al + keys_bytes + vals_bytes;
    if (!tmp) { tmp_bytes = need; return hipSuccess; }
    if (n <= 0) return hipSuccess;
    if (tmp_bytes < need || n >= ((int64_t)1 << 32) || begin_bit < 0 || end_bit > 64) return hipErrorInvalidValue;
    uint8_t *t = static_cast<uint8_t *>(tmp);
    uint32_t *counts = reinterpret_cast<uint32_t *>(t); void *scan_tmp = t + counts_bytes;
    uint64_t *ktmp = reinterpret_cast<uint64_t *>(t + counts_bytes + scan_al); uint32_t *vtmp = reinterpret_cast<uint32_t *>(t + counts_bytes + scan_al + keys_bytes);
    const int npass = std::max(1, (end_bit - begin_bit + R_BITS - 1) / R_BITS);
    // the input is left as it is: pass 0 reads it, the passes behind alternate between tmp and the output and end in the output
    const uint64_t *src_k = kin; const uint32_t *src_v = vin;
    for (int p = 0; p < npass; p++) {
        const bool to_out = ((npass - 1 - p) & 1) == 0;
        uint64_t *dst_k = to_out ? kout : ktmp; uint32_t *dst_v = to_out ? vout : vtmp;
        const int shift = begin_bit + R_BITS * p, width = std::min(R_BITS, end_bit - shift);
        const uint32_t mask = width >= 32 ? 0xffffffffu : ((1u << std::max(width, 0)) - 1u);
        hipLaunchKernelGGL(radix64_count_kernel, dim3((unsigned)nchunks), dim3(R_THREADS), 0, st, src_k, n, shift, mask, counts, nchunks);
        size_t sb = scan_bytes;
        if (hipError_t e = gbn::lut_scan(scan_tmp, sb, counts, counts, ncounts, st)) return e;
        if (pairs) hipLaunchKernelGGL(radix64_scatter_kernel<true>, dim3((unsigned)nchunks), dim3(R_THREADS), 0, st, src_k, src_v, dst_k, dst_v, n, shift, mask, counts, nchunks);
        else hipLaunchKernelGGL(radix64_scatter_kernel<false>, dim3((unsigned)nchunks), dim3(R_THREADS), 0, st, src_k, (const uint32_t *)nullptr, dst_k, (uint32_t *)nullptr, n, shift, mask, counts, nchunks);
        src_k = dst_k; src_v = dst_v;
    }
    return hipGetLastError();
}
}  // namespace

namespace gbn {
// stable LSD radix sort of u64 keys on bits [begin_bit, end_bit); tmp == nullptr: the scratch it needs
hipError_t sort_keys_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, int64_t n, int begin_bit, int end_bit, hipStream_t st)
{
    return radix64(tmp, tmp_bytes, kin, kout, nullptr, nullptr, n, begin_bit, end_bit, false, st);
}
// stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, end_bit)
hipError_t sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout,
                          const uint32_t *vin, uint32_t *vout, int64_t n, int end_bit, hipStream_t st)
{
    return radix64(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, true, st);
}
}  // namespace gbn
