// lutbuild.hip -- the lookup structures of a query batch, built on the device.
//
// Same tables as the host builder (batch.cpp build_lookup + the cell tables of engine.cpp: upload_host_tables), i.e. the word
// enumeration of CORE/blast_lookup.c:87-137 / CORE/blast_nalookup.c:873-928 over the indexed stretches
// (strands minus soft masks, only stretches of at least word_size bases, no word with an ambiguity code),
// every cell's query offsets in the order the reference reports them (megablast chains: descending;
// small / standard tables: ascending).  A 5 Mb megablast batch puts 10 M words into 16.7 M cells: 0.57 s
// of cache misses on one host core, plus 0.28 GB of uploads -- here a few kernels and two library calls
// (radix sort, prefix sums) on data that never leaves HBM.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gbn_dev.h"
#include "lutbuild.h"

namespace {

__device__ __forceinline__ uint32_t fingerprint_dev(const uint8_t *__restrict__ q, int32_t off, int lut, bool force)
{
    // (round 6: the eight bases to the left and the seven to the right as two 8-byte loads -- the buffer has 64 bytes of padding either
    // side -- instead of fifteen byte loads; the low two bits of every byte, the byte nearest the word in the lowest bits)
    typedef uint64_t u64u __attribute__((aligned(1)));
    auto pack8 = [](uint64_t x) -> uint32_t {           // bytes 7 .. 0 of x -> bits 1:0 .. 15:14
        uint64_t y = __builtin_bswap64(x) & 0x0303030303030303ull;
        y = (y | (y >> 6)) & 0x000F000F000F000Full;
        y = (y | (y >> 12)) & 0x000000FF000000FFull;
        return (uint32_t)((y | (y >> 24)) & 0xFFFFull);
    };
    const uint32_t l = pack8(*reinterpret_cast<const u64u *>(q + off - 8));
    const uint32_t r = pack8(*reinterpret_cast<const u64u *>(q + off + lut) << 8) & 0x3FFFu;      // (the eighth byte is not part of it)
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
__global__ void __launch_bounds__(64) lut_enumerate_kernel(gbn::LutBuild B)
{
    constexpr int PER = 16, NT = 64, BLOCK = NT * PER;
    // a workgroup's 1,024 keys (one wave: 4 KB of LDS, so that several fit into what the probe kernel leaves of a CU's) cross the LDS on their way out: a thread computes sixteen CONSECUTIVE positions (the rolled word),
    // a store instruction should write 64 consecutive ones -- stored thread by thread the list took 450 MB of partial sectors per
    // 5 Mb batch for its 80 MB (profiles/r04i_pmc.csv)
    __shared__ uint32_t s_key[NT * (PER + 1)];
    __shared__ int32_t s_seg[NT];                   // where the 64 stretches after the block's first one begin
    const int64_t nblock = ((int64_t)B.qlen + BLOCK - 1) / BLOCK;
    const uint32_t none = 1u << (2 * B.lut), cmask = none - 1u;
    const int lut = B.lut, tid = threadIdx.x;
    // the list's first home: whichever array leaves the sorted list in keys_b / vals_b after lut_sort's passes
    uint32_t *__restrict__ keys0 = (gbn::lut_sort_passes(2 * lut + 1) & 1) ? B.keys_a : B.keys_b, *__restrict__ vals0 = (gbn::lut_sort_passes(2 * lut + 1) & 1) ? B.vals_a : B.vals_b;
    for (int64_t blk = blockIdx.x; blk < nblock; blk += gridDim.x) {
        const int64_t b0 = blk * BLOCK, p0 = b0 + (int64_t)tid * PER;
        const int n = (int)max((int64_t)0, min((int64_t)PER, (int64_t)B.qlen - p0));
        // Round 5: a thread's chain of dependent loads was 14 (its own binary search over the stretches) + 27 (the bases, byte
        // by byte) long -- 1.5 ms per 5 Mb batch alone, 2.0 ms next to the probe kernel whose CUs it shares.  Now: ONE search
        // per workgroup (the block's first position: the same addresses in every lane), the next 64 stretches' starts through
        // the LDS (a 1,024-position block of 1 kb queries meets two or three), and the thread's 16 + lut - 1 bases as nine
        // aligned dwords.
        int lo = 0, hi = B.nseg;                        // last stretch that starts at or before b0
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((int64_t)B.seg_left[m] <= b0) lo = m + 1; else hi = m; }
        const int si0 = lo - 1;
        s_seg[tid] = (si0 + 1 + tid < B.nseg) ? B.seg_left[si0 + 1 + tid] : INT32_MAX;
        __syncthreads();
        if (n > 0) {
            int k = 0;
            while (k < NT && (int64_t)s_seg[k] <= p0) k++;
            int si = si0 + k;
            int32_t left, next_left;
            if (k == NT) {                              // more than 64 stretches begin between b0 and p0 (stretches of a few bases): search on
                int l2 = si0 + 1 + NT, h2 = B.nseg;
                while (l2 < h2) { const int m = (l2 + h2) >> 1; if ((int64_t)B.seg_left[m] <= p0) l2 = m + 1; else h2 = m; }
                si = l2 - 1;
                left = B.seg_left[si];
                next_left = (si + 1 < B.nseg) ? B.seg_left[si + 1] : INT32_MAX;
            } else {
                left = k > 0 ? s_seg[k - 1] : (si0 >= 0 ? B.seg_left[si0] : 0);
                next_left = s_seg[k];                   // (INT32_MAX past the last stretch)
            }
            int32_t right = si >= 0 ? B.seg_right[si] : -1;
            // the word at p is made of the bases p .. p + lut - 1 (the query has 64 bytes of sentinels past its end); `run` =
            // bases without an ambiguity code or a sentinel that end at the newest one
            uint32_t cell = 0; int run = 0;
            auto position = [&](int i, uint8_t b) {     // base p0 + i + lut - 1 has arrived: the word at p0 + i is complete
                const int64_t p = p0 + i;
                run = (b & 0xfc) ? 0 : run + 1;
                cell = ((cell << 2) | (b & 3u)) & cmask;
                while (p >= (int64_t)next_left) {
                    si++; left = next_left; right = B.seg_right[si];
                    next_left = (si + 1 < B.nseg) ? B.seg_left[si + 1] : INT32_MAX;
                }
                const bool ok = si >= 0 && right - left + 1 >= B.word && p + lut - 1 <= (int64_t)right && run >= lut;
                if (ok && B.count) atomicAdd(&B.count[cell], 1u);       // (null: the cells' sizes follow from the sorted list, lut_cell_starts_kernel)
                s_key[tid * (PER + 1) + i] = ok ? cell : none;
            };
            if (lut <= 17) {
                const uintptr_t ad = reinterpret_cast<uintptr_t>(B.q8 + p0);
                const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(ad & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(ad & 3u) * 8u;
                uint32_t v[9];
                #pragma unroll
                for (int j = 0; j < 9; j++) v[j] = w[j];
                #pragma unroll
                for (int j = 0; j < 32; j++) {
                    if (j < lut + n - 1) {              // (lut + 14 <= 31)
                        const uint32_t d = __builtin_amdgcn_alignbit(v[j / 4 + 1], v[j / 4], sh);
                        const uint8_t b = (uint8_t)(d >> (8 * (j & 3)));
                        if (j < lut - 1) { run = (b & 0xfc) ? 0 : run + 1; cell = ((cell << 2) | (b & 3u)) & cmask; }
                        else position(j - (lut - 1), b);
                    }
                }
            } else {
                for (int j = 0; j < lut - 1; j++) {
                    const uint8_t b = B.q8[p0 + j];
                    run = (b & 0xfc) ? 0 : run + 1;
                    cell = (cell << 2) | (b & 3u);
                }
                for (int i = 0; i < n; i++) position(i, B.q8[p0 + i + lut - 1]);
            }
        }
        __syncthreads();
        #pragma unroll
        for (int r = 0; r < PER; r++) {
            const int e = r * NT + tid;                 // the block's e-th position
            const int64_t p = b0 + e;
            if (p < B.qlen) {
                const int64_t at = B.descending ? (int64_t)B.qlen - 1 - p : p;
                keys0[at] = s_key[(e / PER) * (PER + 1) + (e % PER)];
                vals0[at] = (uint32_t)p;
            }
        }
        __syncthreads();
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

// per cell: direct-probe word, LDS table word; cells with three or more entries: their reduced fingerprints go to the bin's
// side list while it has room.  One workgroup per bin (2^cbits cells), the place of a cell's fingerprints in the list = the
// entries of the bin's cells of that kind in front of it (a running prefix over the workgroup's rounds): rounds 1-4 wrote a
// 16.7 M-element array of list lengths, ran a scan over all of it and read both again in a third kernel -- a third of a
// build's time and traffic, and the part of it that ran next to the rare kernel (the end of a build).  A bin's list has a
// fixed home of GBN_BIN_SIDE entries in `sidet` (side_start[bin] = bin x GBN_BIN_SIDE).  Tables of more bins than the
// partitioned scan takes (use_side = 0) get cell words only: every cell of three and more entries "always rare".
// Round 6: EIGHT consecutive cells per thread and round (their cell starts in 16-byte loads, their entries asked for together, their
// words stored 16 bytes at a time): 16 rounds per bin of 32,768 cells instead of 128, each a chain of dependent loads and two
// barriers -- 0.36 -> 0.20 ms (four per thread) of a 5 Mb batch's 1.2 ms build.  The lists keep their order (a thread's cells are consecutive).
__global__ void __launch_bounds__(256) lut_cells_side_kernel(gbn::LutBuild B, int use_side)
{
    constexpr int U = 8;
    __shared__ uint32_t s_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = (int)blockDim.x >> 6;
    const int64_t bin = blockIdx.x, c0 = bin << B.cbits, c1 = min(c0 + ((int64_t)1 << B.cbits), B.ncells);
    if (tid == 0) {
        B.side_start[bin] = use_side ? (uint32_t)bin * (uint32_t)GBN_BIN_SIDE : 0u;
        if (bin == (int64_t)gridDim.x - 1) B.side_start[bin + 1] = use_side ? (uint32_t)(bin + 1) * (uint32_t)GBN_BIN_SIDE : 0u;
    }
    const uint32_t base = use_side ? (uint32_t)bin * (uint32_t)GBN_BIN_SIDE : 0u;
    const bool vec = ((c0 | c1) & 3) == 0;              // (a bin of fewer than four cells, or a table that does not end on one: cell by cell)
    uint32_t run = 0;                                   // entries of the bin's side-list cells so far (those that did not fit included)
    for (int64_t cb = c0; cb < c1; cb += (int64_t)blockDim.x * U) {
        const int64_t cf = cb + (int64_t)tid * U;       // this thread's first cell
        uint32_t st[U + 1];
        if (vec && cf + U <= c1) {
            #pragma unroll
            for (int q = 0; q < U / 4; q++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(B.cell_start + cf + 4 * q);
                st[4 * q] = v.x; st[4 * q + 1] = v.y; st[4 * q + 2] = v.z; st[4 * q + 3] = v.w;
            }
            st[U] = B.cell_start[cf + U];
        } else {
            #pragma unroll
            for (int u = 0; u <= U; u++) st[u] = (cf + u <= c1) ? B.cell_start[cf + u] : 0u;
        }
        unsigned long long e0[U], e1[U];
        #pragma unroll
        for (int u = 0; u < U; u++) {                   // (all of them on their way before the first is looked at)
            const bool have = cf + u < c1 && st[u + 1] > st[u];
            e0[u] = have ? B.ent[st[u]] : 0ull;
            e1[u] = (have && st[u + 1] - st[u] >= 2) ? B.ent[st[u] + 1] : 0ull;
        }
        uint32_t w[U], t[U], many[U], msum = 0;
        #pragma unroll
        for (int u = 0; u < U; u++) {
            w[u] = 0; t[u] = 0; many[u] = 0;
            const uint32_t s = st[u], e = st[u + 1];
            if (cf + u < c1 && e > s) {
                bool forced = false;
                if (B.onebyte_mode) for (uint32_t k = s; k < e; k++) forced = forced || ((B.ent[k] >> 32) & 1ull);
                const uint32_t fp0 = (uint32_t)(e0[u] >> 32);
                w[u] = (fp0 & 0x7fffffffu) | ((e - s > 1) ? 0x80000000u : 0u);
                t[u] = 0x8000u | reduce_fp(fp0) | (reduce_fp(fp0) << 16);
                if (e - s >= 2) t[u] = (t[u] & 0xffffu) | 0x80000000u | (reduce_fp((uint32_t)(e1[u] >> 32)) << 16);
                if (forced) t[u] = 0x80000000u;                     // always the rare path
                else if (e - s >= 3) { t[u] = 0x80000000u; many[u] = (use_side && e - s < 16384u) ? e - s : 0u; }
            }
            msum += many[u];
        }
        // exclusive prefix of the threads' sums over the workgroup
        uint32_t inc = msum;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off); if (lane >= off) inc += v; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int k = 0; k < nwave; k++) { const uint32_t v = s_wsum[k]; if (k < wave) before += v; total += v; }
        __syncthreads();
        if (msum) {
            uint32_t off = run + before + inc - msum;
            #pragma unroll
            for (int u = 0; u < U; u++) {
                if (!many[u]) continue;
                if (off + many[u] <= (uint32_t)GBN_BIN_SIDE) {     // (else: stays "always rare")
                    for (uint32_t k = 0; k < many[u]; k++) B.sidet[base + off + k] = (uint16_t)reduce_fp((uint32_t)(B.ent[st[u] + k] >> 32));
                    t[u] = 0x80000000u | off | (many[u] << 16);
                }
                off += many[u];
            }
        }
        if (vec && cf + U <= c1) {
            #pragma unroll
            for (int q = 0; q < U / 4; q++) {
                *reinterpret_cast<uint4 *>(B.cellw + cf + 4 * q) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
                *reinterpret_cast<uint4 *>(B.cellt + cf + 4 * q) = make_uint4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
            }
        } else {
            #pragma unroll
            for (int u = 0; u < U; u++) if (cf + u < c1) { B.cellw[cf + u] = w[u]; B.cellt[cf + u] = t[u]; }
        }
        run += total;
    }
}

// cell_start from the list sorted on the cell (n valid entries, the number on the device): entry i that opens a cell writes
// the start of that cell and of the empty cells in front of it; i = n closes the table.  Every cell_start is written exactly
// once, in ascending order: 67 MB for 16.7 M cells -- where rounds 1-3 counted the words per cell with an atomic per query
// position (0.8 GB of sectors written back per 5 Mb batch, `profiles/r04i_pmc.csv`), cleared that array first and ran a scan
// over it afterwards: a gigabyte per build that the probe kernel next to it paid for.
__global__ void __launch_bounds__(256) lut_cell_starts_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ n_dev, int64_t ncells, uint32_t *__restrict__ cell_start)
{
    const int64_t n = (int64_t)*n_dev;
    const int lane = (int)(threadIdx.x & 63);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (wave-uniform trip count: the lanes of a wave fill a LONG run of empty cells together -- a batch of few words in a table of
    // 16.7 M cells, or of none at all, would leave millions of stores to one lane)
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 <= n; i0 += stride) {
        const int64_t i = i0 + lane;
        int64_t prev = 0, cur = -1;                             // (nothing to write past the list's end)
        if (i <= n) { prev = i == 0 ? -1 : (int64_t)keys[i - 1]; cur = i == n ? ncells : (int64_t)keys[i]; }
        const bool lng = cur - prev > 32;
        if (!lng) for (int64_t c = prev + 1; c <= cur; c++) cell_start[c] = (uint32_t)i;
        unsigned long long m = __ballot(lng);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t p = __shfl(prev, src), q = __shfl(cur, src);
            const uint32_t v = (uint32_t)(i0 + src);
            for (int64_t c = p + 1 + lane; c <= q; c += 64) cell_start[c] = v;
        }
    }
}

// count == nullptr: a cell is present when its list is not empty (starts = cell_start)
__global__ void lut_pv_kernel(const uint32_t *count, const uint32_t *starts, int64_t ncells, uint32_t *pv)
{
    const int lane = (int)(threadIdx.x & 63);
    const int64_t span = (ncells + 63) & ~(int64_t)63;      // whole waves: the ballot needs every lane of a wave
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < span; c += (int64_t)gridDim.x * blockDim.x) {
        const bool present = c < ncells && (count ? count[c] != 0 : starts[c + 1] != starts[c]);
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
    if (t == 0) {           // the 16 bytes behind either array ("matches nothing": a window may start at the last byte) -- no fill of the whole arrays in front of this kernel
        const int64_t groups = (n + 7) / 8, q2_bytes = (n + 3) / 4 + 16, qi_bytes = (n + 7) / 8 + 16;
        for (int64_t i = groups * 2; i < q2_bytes; i++) q2[i] = 0xff;
        for (int64_t i = groups; i < qi_bytes; i++) qinv[i] = 0xff;
    }
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
    // a thread per sixteen offsets: five aligned dwords in, one dword per plane out (round 5; a thread per offset read four
    // single bytes and wrote one: 0.4 ms next to the probe kernel for 10 MB of query).  qbuf and the planes are 16- / 4-byte
    // aligned (pool blocks; plane is a multiple of 4: launcher).
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, k0 = t * 16;
    if (t < 4 * 128) {      // the 64 (or 65) bytes behind every plane's last byte of the query (15 beyond the buffer in all four positions: 0xff)
        const int64_t pl = t >> 7, used = qbuf_len > pl ? (qbuf_len - pl + 3) / 4 : 0;      // bytes of plane pl that offsets < qbuf_len write
        const int64_t at = used + (t & 127);
        if (at < plane) q4[pl * plane + at] = 0xff;
    }
    if (k0 >= qbuf_len) return;
    uint32_t w[5];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(qbuf + k0);
    #pragma unroll
    for (int i = 0; i < 5; i++) {
        const int64_t at = k0 + 4 * i;
        uint32_t v = 0x0f0f0f0fu;                                   // 15 beyond the buffer
        if (at + 4 <= qbuf_len) v = src[i];
        else if (at < qbuf_len) { v = 0; for (int j = 0; j < 4; j++) v |= (uint32_t)(at + j < qbuf_len ? qbuf[at + j] : 15) << (8 * j); }
        w[i] = v;
    }
    auto code = [&](int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; };     // (i: compile-time after unrolling)
    uint32_t out[4] = {0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t v = ((code(i) << 6) | (code(i + 1) << 4) | (code(i + 2) << 2) | code(i + 3)) & 0xffu;
        out[i & 3] |= v << (8 * (i >> 2));                          // offset k0 + i: plane i & 3, index (k0 >> 2) + (i >> 2)
    }
    #pragma unroll
    for (int p = 0; p < 4; p++) *reinterpret_cast<uint32_t *>(q4 + p * plane + (k0 >> 2)) = out[p];     // (offsets past the query give 0xff, what the fill writes there)
}

}  // namespace

namespace gbn {

hipError_t lut_pack_q4(const uint8_t *qbuf, int64_t qbuf_len, uint8_t *q4, int64_t plane, hipStream_t st)
{
    if (qbuf_len <= 0) return hipSuccess;
    if (plane & 3) return hipErrorInvalidValue;                   // (a dword per plane and thread)
    const int64_t threads = std::max<int64_t>((qbuf_len + 15) / 16, 512);
    hipLaunchKernelGGL(lut_q4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, qbuf, qbuf_len, q4, plane);
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
    hipLaunchKernelGGL(lut_enumerate_kernel, dim3(4 * polite_grid(((int64_t)b.qlen + 15) / 16, 256)), dim3(64), 0, st, b);    // (a one-wave workgroup per 1,024 positions, grid-stride: as many waves as 256-thread workgroups would have brought)
    return hipGetLastError();
}
hipError_t lut_overflow_cells(const LutBuild &b, unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(lut_overflow_kernel, dim3((unsigned)((b.ncells + 255) / 256)), dim3(256), 0, st, b.count, b.ncells, out);
    return hipGetLastError();
}
// Exclusive prefix sums of n counters in three plain kernels -- sums of 4,096-element blocks, their prefix sums (one
// workgroup), the blocks again with their offsets -- instead of the library's single-pass scan: that one chains its
// workgroups through look-back flags, and next to a probe kernel that leaves a few wave slots per CU a workgroup waits for
// a predecessor that has not been scheduled yet: 2.5 ms per scan of 16.7 M cells (two per table build) against 0.1 alone,
// which kept the builder's stream busy past the next pass's binning kernel and made that pass's probe kernel wait for its
// tables (round 4, `profiles/r04_c2_timeline.txt`).
namespace {
constexpr int SCAN_BLOCK = 4096, SCAN_THREADS = 256, SCAN_PER = SCAN_BLOCK / SCAN_THREADS;

__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t incl = v;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d); incl += (lane >= d) ? x : 0u; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0; total = 0;
    for (int w = 0; w < nwaves; w++) { before += (w < wave) ? s_wave[w] : 0u; total += s_wave[w]; }
    __syncthreads();
    return before + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_kernel(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER;
    uint32_t v = 0;
    #pragma unroll
    for (int k = 0; k < SCAN_PER; k++) v += (base + k < n) ? in[base + k] : 0u;
    uint32_t total;
    (void)block_exclusive(v, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_offsets_kernel(uint32_t *sums, int64_t nblocks)
{
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    uint32_t carry = 0;
    for (int64_t b0 = 0; b0 < nblocks; b0 += SCAN_THREADS) {    // (4,097 blocks for 2^24 + 1 cells: seventeen rounds of a small workgroup, which finds a CU at once)
        const int64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblocks ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive(v, s_wave, total);
        if (i < nblocks) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nblocks] = carry;                // the sum of everything
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t *in, int64_t n, const uint32_t *__restrict__ sums, uint32_t *out)      // (in may be out)
{
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER;
    uint32_t x[SCAN_PER], v = 0;
    #pragma unroll
    for (int k = 0; k < SCAN_PER; k++) { x[k] = (base + k < n) ? in[base + k] : 0u; v += x[k]; }
    uint32_t total;
    uint32_t at = sums[blockIdx.x] + block_exclusive(v, s_wave, total);
    #pragma unroll
    for (int k = 0; k < SCAN_PER; k++) { if (base + k < n) out[base + k] = at; at += x[k]; }
}
}  // namespace

// tmp == nullptr: the scratch it needs; in and out may be the same array
hipError_t lut_scan(void *tmp, size_t &bytes, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st)
{
    const int64_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (!tmp) { bytes = (size_t)(nblocks + 1) * sizeof(uint32_t); return hipSuccess; }
    if (n <= 0) return hipSuccess;
    if (bytes < (size_t)(nblocks + 1) * sizeof(uint32_t)) return hipErrorInvalidValue;
    uint32_t *sums = static_cast<uint32_t *>(tmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, in, n, sums);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, sums, nblocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, in, n, sums, out);
    return hipGetLastError();
}

// The (cell, offset) list sorted on the cell by a radix sort of our own: stable passes over 8-bit digits, each a count per
// 4,096-element chunk, the scan above over (digit, chunk), and a scatter in which a wave ranks the 64 elements of a round
// among themselves with a ballot per digit bit (lane order = list order) and the waves follow one another through per-wave
// counters in LDS -- the pattern of seed_order.hip.  Pass 0 also DROPS the positions without a word (key = `none`), so
// 2 * lut bits are sorted, not 2 * lut + 1.  No look-back chains between workgroups and 1-4 KB of LDS per workgroup: next to a
// probe kernel that owns the CUs the library's onesweep sort took 4.1 ms per 5 Mb batch (its histogram 2.2, the pass that met
// the rare kernel 1.6), this one a third of that.
namespace {
constexpr int RS_CHUNK = 4096, RS_THREADS = 256, RS_WAVES = RS_THREADS / 64, RS_ROUNDS = RS_CHUNK / RS_THREADS;
// digits of 6 bits: a chunk leaves runs of 64 elements (256 bytes) per digit instead of 16 (64 bytes, most of them across two
// sectors and written a few bytes at a time: four times the bytes written, `profiles/r04i_pmc.csv`) -- four passes for 24 bits
// instead of three, and fewer bytes all the same
constexpr int RS_BITS = GBN_LUT_RADIX_BITS, RS_DIGITS = 1 << RS_BITS;

__global__ void __launch_bounds__(RS_THREADS) radix_count_kernel(const uint32_t *__restrict__ keys, int64_t n_cap, const uint32_t *__restrict__ n_dev,
                                                                 uint32_t none, int shift, uint32_t *__restrict__ counts, int64_t nchunks)
{
    __shared__ uint32_t s_hist[RS_DIGITS];
    const int tid = threadIdx.x;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap, base = (int64_t)blockIdx.x * RS_CHUNK;
    if (tid < RS_DIGITS) s_hist[tid] = 0;
    __syncthreads();
    uint32_t k[RS_ROUNDS];
    #pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) { const int64_t i = base + r * RS_THREADS + tid; k[r] = i < n ? keys[i] : none; }
    #pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) if (k[r] != none) atomicAdd(&s_hist[(k[r] >> shift) & (uint32_t)(RS_DIGITS - 1)], 1u);
    __syncthreads();
    if (tid < RS_DIGITS) counts[(int64_t)tid * nchunks + blockIdx.x] = s_hist[tid];          // digit-major: one scan over the whole table gives every (digit, chunk) its place
}

__global__ void __launch_bounds__(RS_THREADS) radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                   uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n_cap,
                                                                   const uint32_t *__restrict__ n_dev, uint32_t none, int shift,
                                                                   const uint32_t *__restrict__ offsets, int64_t nchunks)
{
    __shared__ uint32_t s_cnt[RS_WAVES][RS_DIGITS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int64_t w0 = (int64_t)blockIdx.x * RS_CHUNK + (int64_t)wave * (RS_CHUNK / RS_WAVES) + lane;
    if (tid < RS_DIGITS) {
        #pragma unroll
        for (int w = 0; w < RS_WAVES; w++) s_cnt[w][tid] = 0;
    }
    __syncthreads();
    uint32_t k[RS_ROUNDS], v[RS_ROUNDS], sr[RS_ROUNDS];
    #pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) { const int64_t i = w0 + r * 64; k[r] = i < n ? keys_in[i] : none; v[r] = i < n ? vals_in[i] : 0u; }
    uint32_t *mine = s_cnt[wave];
    #pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const bool valid = k[r] != none;
        const uint32_t digit = (k[r] >> shift) & (uint32_t)(RS_DIGITS - 1);
        unsigned long long peers = __ballot(valid);
        #pragma unroll
        for (int b = 0; b < RS_BITS; b++) {
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
    if (tid < RS_DIGITS) {
        uint32_t run = offsets[(int64_t)tid * nchunks + blockIdx.x];
        #pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++)
        if (sr[r] != 0xffffffffu) { const uint32_t at = mine[sr[r] >> 16] + (sr[r] & 0xffffu); keys_out[at] = k[r]; vals_out[at] = v[r]; }
}
}  // namespace

// keys_a / vals_a -> keys_b / vals_b: the elements whose key is not 1 << (key_bits - 1), ordered by key, stable
hipError_t lut_cell_starts(const LutBuild &b, const uint32_t *n_valid, hipStream_t st)
{
    hipLaunchKernelGGL(lut_cell_starts_kernel, dim3(polite_grid((int64_t)b.qlen + 1, 256)), dim3(256), 0, st, b.keys_b, n_valid, b.ncells, b.cell_start);
    return hipGetLastError();
}

// n_valid_out (optional, device): receives the number of elements kept
hipError_t lut_sort(void *tmp, size_t &bytes, const LutBuild &b, int64_t n, int key_bits, hipStream_t st, uint32_t *n_valid_out)
{
    const int64_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK, ncounts = RS_DIGITS * std::max<int64_t>(nchunks, 1);
    size_t scan_bytes = 0;
    (void)lut_scan(nullptr, scan_bytes, nullptr, nullptr, ncounts, st);
    const size_t need = (size_t)ncounts * 4 + scan_bytes + 64;
    if (!tmp) { bytes = need; return hipSuccess; }
    if (n <= 0) return hipSuccess;
    if (bytes < need || key_bits < 2 || key_bits > 32) return hipErrorInvalidValue;
    uint32_t *counts = static_cast<uint32_t *>(tmp), *scan_tmp = counts + ncounts, *n_valid = scan_tmp + scan_bytes / 4 + 2;
    const uint32_t none = 1u << (key_bits - 1);
    const int npass = lut_sort_passes(key_bits);
    // (the list starts where lut_enumerate put it: in keys_a / vals_a for an odd number of passes, in keys_b / vals_b for an even one)
    uint32_t *const K[2] = {npass & 1 ? b.keys_a : b.keys_b, npass & 1 ? b.keys_b : b.keys_a};
    uint32_t *const V[2] = {npass & 1 ? b.vals_a : b.vals_b, npass & 1 ? b.vals_b : b.vals_a};
    const int64_t scan_blocks = (ncounts + SCAN_BLOCK - 1) / SCAN_BLOCK;
    for (int p = 0; p < npass; p++) {
        const uint32_t *nd = p ? n_valid : nullptr;             // (pass 0 drops the keys that are `none`: the passes behind it see the rest)
        const int in = p & 1, out = in ^ 1;
        hipLaunchKernelGGL(radix_count_kernel, dim3((unsigned)nchunks), dim3(RS_THREADS), 0, st, K[in], n, nd, none, RS_BITS * p, counts, nchunks);
        size_t sb = scan_bytes;
        if (hipError_t e = lut_scan(scan_tmp, sb, counts, counts, ncounts, st)) return e;
        if (p == 0) {
            if (hipError_t e = hipMemcpyAsync(n_valid, scan_tmp + scan_blocks, 4, hipMemcpyDeviceToDevice, st)) return e;
            if (n_valid_out) { if (hipError_t e = hipMemcpyAsync(n_valid_out, scan_tmp + scan_blocks, 4, hipMemcpyDeviceToDevice, st)) return e; }
        }
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nchunks), dim3(RS_THREADS), 0, st, K[in], V[in], K[out], V[out], n, nd, none, RS_BITS * p, counts, nchunks);
    }
    return hipGetLastError();
}
hipError_t lut_entries(const LutBuild &b, int64_t n, hipStream_t st)
{
    // n < 0: count on the device (cell_start[ncells]), at most b.qlen
    hipLaunchKernelGGL(lut_entries_kernel, dim3(polite_grid(n >= 0 ? std::max<int64_t>(n, 1) : b.qlen, 256)), dim3(256), 0, st, b, n);
    return hipGetLastError();
}
hipError_t lut_cells_side(const LutBuild &b, hipStream_t st)
{
    const int use_side = b.nbins <= GBN_BIN_MAXNB ? 1 : 0;
    const int64_t cells_per_bin = std::min<int64_t>((int64_t)1 << b.cbits, b.ncells);
    // (256 threads: one wave per SIMD is what fits next to a probe workgroup -- 85 VGPRs x 4 waves per SIMD, 155 KB of LDS; with
    // 1,024 threads per workgroup the kernel waited for the probe kernel to end and ran next to the rare kernel instead: 3.6 ms)
    const int threads = (int)std::max<int64_t>(64, std::min<int64_t>(256, (cells_per_bin + 63) / 64 * 64));
    hipLaunchKernelGGL(lut_cells_side_kernel, dim3((unsigned)b.nbins), dim3((unsigned)threads), 0, st, b, use_side);
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
    hipLaunchKernelGGL(lut_pv_kernel, dim3(polite_grid(b.ncells, 256)), dim3(256), 0, st, b.count, b.cell_start, b.ncells, b.pv);
    return hipGetLastError();
}

}  // namespace gbn
