// lutbuild.hip -- the lookup structures of a query batch, built on the device.
//
// Same tables as the host builder (batch.cpp build_lookup + the cell tables of engine.cpp), i.e. the word
// enumeration of CORE/blast_lookup.c:87-137 / CORE/blast_nalookup.c:873-928 over the indexed stretches
// (strands minus soft masks, only stretches of at least word_size bases, no word with an ambiguity code),
// every cell's query offsets in the order the reference reports them (megablast chains: descending;
// small / standard tables: ascending).  A 5 Mb megablast batch puts 10 M words into 16.7 M cells: 0.57 s
// of cache misses on one host core, plus 0.28 GB of uploads -- here a few kernels and two library calls
// (radix sort, prefix sums) on data that never leaves HBM.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include "gbn_dev.h"
#include "lutbuild.h"

namespace {

__device__ __forceinline__ uint32_t fingerprint_dev(const uint8_t *__restrict__ q, int32_t off, int lut, bool force)
{
    uint32_t l = 0, r = 0;
    #pragma unroll
    for (int k = 1; k <= 8; k++) l |= (uint32_t)(q[off - k] & 3) << (2 * (k - 1));
    #pragma unroll
    for (int j = 0; j < 7; j++) r |= (uint32_t)(q[off + lut + j] & 3) << (2 * (6 - j));
    return (l << 15) | (r << 1) | (force ? 1u : 0u);
}
// 15-bit reduced fingerprint: 3.5 bases to the right (7 bits, high) and 4 bases to the left (8 bits, low)
__device__ __forceinline__ uint32_t reduce_fp(uint32_t fp) { return ((((fp >> 1) & 0x3fffu) >> 7) << 8) | ((fp >> 15) & 0xffu); }

// one thread per query position: is a lookup word indexed here?  Position p leaves {its cell, p} at list index p
// (megablast chains, reported in descending offset order: qlen - 1 - p), positions without a word a key past every
// cell: a STABLE sort of the list on the cell alone then has every cell's offsets in the reference's order -- 32-bit
// keys, 2 * lut + 1 bits to sort, where rounds 1-3 sorted 64-bit (cell, offset) keys on 2 * lut + 24 bits (half the
// passes, two thirds of the bytes per pass, no memset of the key array, no list reservation per block).
// (Sixteen consecutive positions per thread: one search for the stretch, the word rolled on base by base -- 27 byte loads
// instead of 192 and one binary search instead of sixteen; a thread per position with its chain of 26 dependent loads took
// 2.3 ms per 5 Mb batch next to a running scan, the longest kernel of a build.)
__global__ void __launch_bounds__(256) lut_enumerate_kernel(gbn::LutBuild B)
{
    constexpr int PER = 16;
    const int64_t nchunk = ((int64_t)B.qlen + PER - 1) / PER;
    const uint32_t none = 1u << (2 * B.lut), cmask = none - 1u;
    const int lut = B.lut;
    for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = ch * PER;
        const int n = (int)min((int64_t)PER, (int64_t)B.qlen - p0);
        int lo = 0, hi = B.nseg;                    // last stretch that starts at or before p0
        while (lo < hi) { const int m = (lo + hi) >> 1; if (B.seg_left[m] <= p0) lo = m + 1; else hi = m; }
        int si = lo - 1;
        int32_t left = si >= 0 ? B.seg_left[si] : 0, right = si >= 0 ? B.seg_right[si] : -1;
        int32_t next_left = (si + 1 < B.nseg) ? B.seg_left[si + 1] : INT32_MAX;
        // the word at p is made of the bases p .. p + lut - 1 (the query has 64 bytes of sentinels past its end); `run` =
        // bases without an ambiguity code or a sentinel that end at the newest one
        uint32_t cell = 0; int run = 0;
        for (int k = 0; k < lut - 1; k++) {
            const uint8_t b = B.q8[p0 + k];
            run = (b & 0xfc) ? 0 : run + 1;
            cell = (cell << 2) | (b & 3u);
        }
        for (int i = 0; i < n; i++) {
            const int64_t p = p0 + i;
            const uint8_t b = B.q8[p + lut - 1];
            run = (b & 0xfc) ? 0 : run + 1;
            cell = ((cell << 2) | (b & 3u)) & cmask;
            while (p >= (int64_t)next_left) {
                si++; left = next_left; right = B.seg_right[si];
                next_left = (si + 1 < B.nseg) ? B.seg_left[si + 1] : INT32_MAX;
            }
            const bool ok = si >= 0 && right - left + 1 >= B.word && p + lut - 1 <= (int64_t)right && run >= lut;
            if (ok) atomicAdd(&B.count[cell], 1u);
            const int64_t at = B.descending ? (int64_t)B.qlen - 1 - p : p;
            B.keys_a[at] = ok ? cell : none;
            B.vals_a[at] = (uint32_t)p;
        }
    }
}

// small-NA table: does the overflow array stay below 32,768 entries?  (CORE/blast_nalookup.c:184-187, :200-324)
__global__ void lut_overflow_kernel(const uint32_t *count, int64_t ncells, unsigned long long *out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = (c < ncells && count[c] > 1) ? (unsigned long long)count[c] + 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// entries in chain order: fingerprint + offset
__global__ void lut_entries_kernel(gbn::LutBuild B, int64_t n)
{
    const int64_t nn = n >= 0 ? n : (int64_t)B.cell_start[B.ncells];       // n < 0: the word count is still on the device only
    if (blockIdx.x == 0 && threadIdx.x == 0) B.ent[nn] = 0;         // pad entry: an empty list still has a valid pointer
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nn; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t off = (int32_t)B.vals_b[k];
        const bool force = B.onebyte_mode && (off + B.lut >= B.qlen);
        B.ent[k] = ((unsigned long long)fingerprint_dev(B.q8, off, B.lut, force) << 32) | (uint32_t)off;
    }
}

// per cell: direct-probe word, LDS table word for cells with one or two entries, size of its side list
__global__ void lut_cells_kernel(gbn::LutBuild B)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= B.ncells; c += (int64_t)gridDim.x * blockDim.x) {
    if (c == B.ncells) { B.many[c] = 0; continue; }
    const uint32_t s = B.cell_start[c], e = B.cell_start[c + 1];
    uint32_t w = 0, t = 0, many = 0;
    if (e > s) {
        bool forced = false;
        if (B.onebyte_mode) for (uint32_t k = s; k < e; k++) forced = forced || ((B.ent[k] >> 32) & 1ull);
        const uint32_t fp0 = (uint32_t)(B.ent[s] >> 32);
        w = (fp0 & 0x7fffffffu) | ((e - s > 1) ? 0x80000000u : 0u);
        t = 0x8000u | reduce_fp(fp0) | (reduce_fp(fp0) << 16);
        if (e - s >= 2) t = (t & 0xffffu) | 0x80000000u | (reduce_fp((uint32_t)(B.ent[s + 1] >> 32)) << 16);
        if (forced) t = 0x80000000u;                            // always the rare path
        else if (e - s >= 3) { t = 0x80000000u; many = (e - s < 16384u) ? e - s : 0u; }    // decided by lut_side_kernel
    }
    B.cellw[c] = w; B.cellt[c] = t; B.many[c] = many;
    }
}

// cells with three or more entries: their reduced fingerprints go to the bin's side list while it has room
__global__ void lut_side_kernel(gbn::LutBuild B)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < B.ncells; c += (int64_t)gridDim.x * blockDim.x) {
    if (c < (int64_t)B.nbins + 1) {
        const int64_t first = min(c << B.cbits, B.ncells);
        B.side_start[c] = B.many_prefix[first];
    }
    const uint32_t cnt = B.many[c];
    if (!cnt) continue;
    const int64_t bin = c >> B.cbits;
    const uint32_t base = B.many_prefix[bin << B.cbits], off = B.many_prefix[c] - base;
    if (off + cnt > (uint32_t)GBN_BIN_SIDE) continue;           // stays "always rare"
    const uint32_t s = B.cell_start[c];
    for (uint32_t k = 0; k < cnt; k++) B.sidet[base + off + k] = (uint16_t)reduce_fp((uint32_t)(B.ent[s + k] >> 32));
    B.cellt[c] = 0x80000000u | off | (cnt << 16);
    }
}

__global__ void lut_pv_kernel(const uint32_t *count, int64_t ncells, uint32_t *pv)
{
    const int lane = (int)(threadIdx.x & 63);
    const int64_t span = (ncells + 63) & ~(int64_t)63;      // whole waves: the ballot needs every lane of a wave
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < span; c += (int64_t)gridDim.x * blockDim.x) {
        const bool present = c < ncells && count[c] != 0;
        const unsigned long long m = __ballot(present);
        if (c < ncells && (lane & 31) == 0) pv[c >> 5] = (uint32_t)(m >> (lane & 32));
    }
}

// Rank form of the cell table for the folded slice scan (scan_fold_kernel): per presence word {the word, number of
// present cells in front of it}, and the entry lists' starts by RANK of the present cell -- what a lookup needs lies in
// ncells / 4 + 4 x (present cells) bytes (1.3 MB for 100 kb of query at lut 11: L2-resident) instead of the 4 x ncells
// bytes of cell_start (16 MB: a random HBM sector per hit)
__global__ void lut_rank_count_kernel(const uint32_t *pv, int64_t nwords, uint32_t *popc)
{
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= nwords; w += (int64_t)gridDim.x * blockDim.x)
        popc[w] = w < nwords ? (uint32_t)__popc(pv[w]) : 0u;
}
__global__ void lut_rank_fill_kernel(const uint32_t *pv, const uint32_t *prefix, const uint32_t *cell_start, int64_t ncells, int64_t nwords,
                                     uint2 *pvx, uint32_t *pstart)
{
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= nwords; w += (int64_t)gridDim.x * blockDim.x) {
        if (w == nwords) { pstart[prefix[nwords]] = cell_start[ncells]; continue; }
        uint32_t m = pv[w], r = prefix[w];
        pvx[w] = make_uint2(m, r);
        while (m) {
            const int b = __ffs(m) - 1;
            m &= m - 1;
            pstart[r++] = cell_start[w * 32 + b];
        }
    }
}

// 2-bit copy of the query + bitmap of the codes that can never match a subject base, for the gapped kernels'
// 32-bases-per-step match runs: base i of the packed copy = qbuf[first + i] (15 outside the buffer)
__global__ void lut_pack_query_kernel(const uint8_t *qbuf, int64_t qbuf_len, int64_t first, int64_t n, uint8_t *q2, uint8_t *qinv)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 8 bases
    if (t * 8 >= n) return;
    uint32_t two = 0, inv = 0;
    for (int k = 0; k < 8; k++) {
        const int64_t i = t * 8 + k, src = first + i;
        const uint8_t code = (i < n && src >= 0 && src < qbuf_len) ? qbuf[src] : 15;
        two = (two << 2) | (uint32_t)(code & 3);
        inv = (inv << 1) | (code <= 3 ? 0u : 1u);
    }
    q2[t * 2] = (uint8_t)(two >> 8); q2[t * 2 + 1] = (uint8_t)two;
    qinv[t] = (uint8_t)inv;
}

// "four bases per byte at every offset" copy of the query for the approximate ungapped extension: the byte
// s_NuclUngappedExtend puts together from the unpacked codes at buffer offset k, (q[k] << 6) | (q[k+1] << 4) | (q[k+2] << 2)
// | q[k+3] truncated to 8 bits (CORE/na_ungapped.c:296, :323) -- codes above 3 (ambiguity, the sentinel between
// the strands) spill into their neighbours' bits exactly as they do there; 15 beyond the buffer.  Stored in FOUR PLANES by
// k mod 4 (byte of offset k at plane k & 3, index k >> 2): an extension steps through the query four bases at a time, so
// the bytes of consecutive steps -- offsets k, k + 4, k + 8, ... -- are consecutive bytes of one plane, eight steps one
// 8-byte load (from one array they were every fourth byte of 32: two 16-byte gathers per round).
__global__ void lut_q4_kernel(const uint8_t *qbuf, int64_t qbuf_len, uint8_t *q4, int64_t plane)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= qbuf_len) return;
    uint32_t v = 0;
    for (int j = 0; j < 4; j++) v |= (uint32_t)(k + j < qbuf_len ? qbuf[k + j] : 15) << (6 - 2 * j);
    q4[(k & 3) * plane + (k >> 2)] = (uint8_t)v;
}

}  // namespace

namespace gbn {

hipError_t lut_pack_q4(const uint8_t *qbuf, int64_t qbuf_len, uint8_t *q4, int64_t plane, hipStream_t st)
{
    if (qbuf_len <= 0) return hipSuccess;
    hipLaunchKernelGGL(lut_q4_kernel, dim3((unsigned)((qbuf_len + 255) / 256)), dim3(256), 0, st, qbuf, qbuf_len, q4, plane);
    return hipGetLastError();
}

// The builder runs next to a search whose kernels need whole CUs: its own kernels keep to a few waves per CU
// (grid-stride loops) instead of filling every wave slot for a moment.
static unsigned polite_grid(int64_t work_items, int block) {
    static int cus = 0;
    if (!cus) { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    const int64_t need = (work_items + block - 1) / block, cap = (int64_t)cus * (block >= 1024 ? 1 : 2);
    return (unsigned)std::max<int64_t>(1, std::min(need, cap));
}

hipError_t lut_pack_query(const uint8_t *qbuf, int64_t qbuf_len, int64_t first, int64_t n, uint8_t *q2, uint8_t *qinv, hipStream_t st)
{
    const int64_t groups = (n + 7) / 8;
    if (groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(lut_pack_query_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, qbuf, qbuf_len, first, n, q2, qinv);
    return hipGetLastError();
}

hipError_t lut_enumerate(const LutBuild &b, hipStream_t st)
{
    if (b.qlen <= 0) return hipSuccess;
    hipLaunchKernelGGL(lut_enumerate_kernel, dim3(polite_grid(((int64_t)b.qlen + 15) / 16, 256)), dim3(256), 0, st, b);
    return hipGetLastError();
}
hipError_t lut_overflow_cells(const LutBuild &b, unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(lut_overflow_kernel, dim3((unsigned)((b.ncells + 255) / 256)), dim3(256), 0, st, b.count, b.ncells, out);
    return hipGetLastError();
}
hipError_t lut_sort(void *tmp, size_t &bytes, const LutBuild &b, int64_t n, int key_bits, hipStream_t st)
{
    return hipcub::DeviceRadixSort::SortPairs(tmp, bytes, b.keys_a, b.keys_b, b.vals_a, b.vals_b, (int)n, 0, key_bits, st);
}
hipError_t lut_scan(void *tmp, size_t &bytes, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st)
{
    return hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, (int)n, st);
}
hipError_t lut_entries(const LutBuild &b, int64_t n, hipStream_t st)
{
    // n < 0: count on the device (cell_start[ncells]), at most b.qlen
    hipLaunchKernelGGL(lut_entries_kernel, dim3(polite_grid(n >= 0 ? std::max<int64_t>(n, 1) : b.qlen, 256)), dim3(256), 0, st, b, n);
    return hipGetLastError();
}
hipError_t lut_cells(const LutBuild &b, hipStream_t st)
{
    hipLaunchKernelGGL(lut_cells_kernel, dim3(polite_grid(b.ncells + 1, 256)), dim3(256), 0, st, b);
    return hipGetLastError();
}
hipError_t lut_side(const LutBuild &b, hipStream_t st)
{
    hipLaunchKernelGGL(lut_side_kernel, dim3(polite_grid(b.ncells, 256)), dim3(256), 0, st, b);
    return hipGetLastError();
}
hipError_t lut_rank_count(const uint32_t *pv, int64_t nwords, uint32_t *popc, hipStream_t st)
{
    hipLaunchKernelGGL(lut_rank_count_kernel, dim3(polite_grid(nwords + 1, 256)), dim3(256), 0, st, pv, nwords, popc);
    return hipGetLastError();
}
hipError_t lut_rank_fill(const uint32_t *pv, const uint32_t *prefix, const uint32_t *cell_start, int64_t ncells, int64_t nwords,
                         uint32_t *pvx, uint32_t *pstart, hipStream_t st)
{
    hipLaunchKernelGGL(lut_rank_fill_kernel, dim3(polite_grid(nwords + 1, 256)), dim3(256), 0, st, pv, prefix, cell_start, ncells, nwords,
                       reinterpret_cast<uint2 *>(pvx), pstart);
    return hipGetLastError();
}
hipError_t lut_pv(const LutBuild &b, hipStream_t st)
{
    hipLaunchKernelGGL(lut_pv_kernel, dim3(polite_grid(b.ncells, 256)), dim3(256), 0, st, b.count, b.ncells, b.pv);
    return hipGetLastError();
}

}  // namespace gbn
