// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the blastn /
// megablast preliminary search.  Integer work only: no MFMA.
//
//   (the key-range partitioned scan -- scan_bin_kernel*, probe_bin_kernel, probe_rare_kernel: TNaScanSubjectFunction +
//   the TNaExtendFunction mini-extension for tables too large for L2 -- lives in scan_bin.hip since round 4)
//   scan_seed_kernel     that work by direct table probes, without streams (fallback for
//                        repeat-dominated subject ranges)
//   scan_slice_kernel    the same work for tables as wide as the word (blastn shapes: stride 1, every
//                        lookup hit a seed): the presence bits sliced through the LDS, the subjects
//                        streamed past them; seeds into per-workgroup segments (seed_compact_kernel
//                        puts them back to back for the consumers that want one array)
//   (the seed stage -- seed keys, diagonal filter, ungapped extension, replay -- lives in seed_stage.hip since round 5; the
//   gapped extensions in gapped.hip, the seed-order kernels in seed_order.hip and seed_sort.hip)
// (lookup tables of a query batch: lutbuild.hip)
//
// Data layout: subjects are NCBI2na (4 bases/byte, base 0 in
// bits 7..6) back to back in one HBM slab, 16-byte aligned each; the query is
// one byte per base (BLASTNA) with sentinel padding on both sides.
#include "scan_dev.hpp"
#include <cstring>
#include <algorithm>

#ifndef GBN_DIAG_ABL
#define GBN_DIAG_ABL 0      // timing experiments only (1: no ungapped extension, 2: no strand search): wrong results
#endif
#ifndef GBN_BIN_ABL
#define GBN_BIN_ABL 0       // timing experiments only (scan_bin_kernel), bits: 2 no record stores, 4 no `hi` stores, 8 no index stores: wrong results
#endif
#ifndef GBN_EXT_ABL
#define GBN_EXT_ABL 0       // timing experiments only (seed_ext_kernel), bits: 1 no exact pass, 2 no extension, 4 no reservation of run heads, 8 no context lookup, 16 no record store: wrong results
#endif


// ---------------------------------------------------------------------------
// Scan + seed kernel.
// One workgroup per tile of consecutive scan positions of one subject; lane t
// takes positions t, t+256, ... so a wave touches one contiguous ~272-byte
// span per load.  Positions whose lookup word is present (presence bit array,
// L2 resident) are compacted into LDS with wave ballots; phase 2 walks the
// compacted list with every lane busy.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(GBN_SCAN_THREADS)
scan_seed_kernel(GbnScanParams P)
{
    __shared__ uint32_t s_cell[GBN_TILE_POS];
    __shared__ int32_t  s_pos[GBN_TILE_POS];
    __shared__ uint32_t s_count;
    __shared__ unsigned long long s_raw_hits;

    const int tid = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const GbnTile T = P.tiles[tile];
        const uint8_t *__restrict__ subj = P.db + P.byte_off[T.subj];
        const int32_t slen = P.len[T.subj];
        if (tid == 0) { s_count = 0; s_raw_hits = 0; }
        __syncthreads();

        // ---- phase 1: extract lookup words, presence test, ballot-compact ----
        const uint32_t mask = (uint32_t)(P.ncells - 1);
        const int shift = 32 - 2 * P.lut;
        // all subject windows of the lane first, then all presence words, then the ballots: 8 independent
        // loads in flight per lane instead of a chain of two dependent ones per position
        constexpr int PER = GBN_TILE_POS / GBN_SCAN_THREADS;
        uint32_t win[PER], pvw[PER];
        #pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = min(tid + k * GBN_SCAN_THREADS, T.npos - 1);      // clamped: always a valid address
            win[k] = window16(subj, T.first_pos + i * P.step);
        }
        #pragma unroll
        for (int k = 0; k < PER; k++) pvw[k] = P.pv[((win[k] >> shift) & mask) >> 5];
        #pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid + k * GBN_SCAN_THREADS;
            const uint32_t cell = (win[k] >> shift) & mask;
            const int32_t s = T.first_pos + i * P.step;
            const bool present = i < T.npos && ((pvw[k] >> (cell & 31)) & 1u);
            unsigned long long b = __ballot(present);
            if (b) {
                int lane = tid & 63;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_count, (uint32_t)__popcll(b));
                base = __shfl(base, 0);
                if (present) {
                    uint32_t slot = base + (uint32_t)__popcll(b & ((1ull << lane) - 1));
                    s_cell[slot] = cell; s_pos[slot] = s;
                }
            }
        }
        __syncthreads();

        // ---- phase 2: fingerprint filter, chain walk, exact verification ----
        const uint32_t n = s_count;
        unsigned long long raw = 0;
        for (uint32_t j = tid; j < n; j += GBN_SCAN_THREADS) {
            const uint32_t cell = s_cell[j];
            const int32_t s = s_pos[j];
            const uint32_t w = P.cellw[cell];
            const uint32_t sl = window16(subj, (int64_t)s - 8) >> 16;          // bases s-8..s-1
            const uint32_t sr = window16(subj, (int64_t)s + P.lut);            // bases s+lut..
            // left fingerprint in fp has base q-1 in the LOW pair; sl has s-1 in the low pair too
            const bool more = w >> 31;
            uint32_t start = 0, end = 0; bool have_range = false;
            if (fp_pass(w, sl, sr, P.fl, P.fr)) {
                start = P.cell_start[cell]; end = P.cell_start[cell + 1]; have_range = true;
                int32_t q = (int32_t)(P.ent[start] & 0xffffffffu);
                int el = verify_hit(P, subj, slen, q, s);
                if (el >= 0) {
                    unsigned long long o = atomicAdd(P.seed_count, 1ull);
                    if (o < P.seed_cap) { GbnDevSeed sd; sd.subj = T.subj; sd.s_scan = s; sd.q_pos = q; sd.ext_left = el; P.seeds[o] = sd; }
                }
            }
            if (more) {
                if (!have_range) { start = P.cell_start[cell]; end = P.cell_start[cell + 1]; }
                raw += end - start;
                for (uint32_t e = start + 1; e < end; e++) {
                    unsigned long long ent = P.ent[e];
                    if (!fp_pass((uint32_t)(ent >> 32), sl, sr, P.fl, P.fr)) continue;
                    int32_t q = (int32_t)(ent & 0xffffffffu);
                    int el = verify_hit(P, subj, slen, q, s);
                    if (el >= 0) {
                        unsigned long long o = atomicAdd(P.seed_count, 1ull);
                        if (o < P.seed_cap) { GbnDevSeed sd; sd.subj = T.subj; sd.s_scan = s; sd.q_pos = q; sd.ext_left = el; P.seeds[o] = sd; }
                    }
                }
            } else {
                raw += 1;
            }
        }
        if (P.raw_hits) {
            // lookup_hits diagnostic: one atomic per wave
            for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
            if ((tid & 63) == 0 && raw) atomicAdd(&s_raw_hits, raw);
            __syncthreads();
            if (tid == 0 && s_raw_hits) atomicAdd(P.raw_hits, s_raw_hits);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Scan for tables as wide as the word (blastn word sizes 7 .. 12: stride 1, every lookup hit is a seed, a query batch
// of 100 kb fills 5 % of the 4^11 cells).  The partitioned scan writes a record per scan position and reads it back
// (6 + 4 bytes per position) to find the one position in a hundred whose word is present; here the presence bits
// themselves are the LDS resident: the bit array is cut into slices of 2^20 cells (128 KB), a workgroup keeps ONE
// slice and streams the subjects past it, so the subjects are read once per slice (4 times for lut 11 -- 1 GB per
// Gbp, mostly from L2 since the workgroups of all slices walk the tiles in step) and nothing is written but seeds.
// A wave takes a tile of 2048 positions on its own (no barrier in the loop): 32 consecutive positions per lane out of
// three dwords of subject, one LDS read per position; the present ones wait in a per-wave queue until 64 are
// together, then a lane each walks its cell's entries.
//
// FOLD (more than one slice): the workgroup keeps the OR of all slices instead of one of them -- a filter over the low
// 2^20 cell bits, 18 % full for the same batch -- and streams the subjects past it ONCE.  What passes the filter is
// looked up in the presence bits themselves (GbnScanParams::pvx: the word and the rank of its first cell in one
// 8-byte read, 1 MB for lut 11, L2-resident); the present ones are queued as before, by RANK, and a queued lane
// finds its entry list by rank (pstart: 4 bytes per present cell, L2-resident too -- cell_start's 16 MB cost a random
// HBM sector per hit).  A position costs one word extraction and one LDS read instead of one per slice: the sliced
// form spent 13.6 wave-instructions per position and slice.
// ---------------------------------------------------------------------------------------------------
template <bool FOLD>
__device__ __forceinline__ void scan_slice_body(const GbnScanParams &P, int nslices, int slice_cell_bits, GbnDevSeed *seg, uint32_t seg_cap,
                                                uint32_t *seg_count, unsigned long long *seg_max, uint32_t *s_slice, uint32_t &s_used)
{
    uint32_t *s_pv = s_slice;                                           // GBN_SLICE_WORDS presence words
    uint32_t *s_q = s_slice + GBN_SLICE_WORDS;                          // [waves][3][GBN_SLICE_QCAP]: position, cell, subject
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = FOLD ? 0 : (int)(blockIdx.x % (unsigned)nslices);     // this workgroup's slice
    const int64_t g = FOLD ? blockIdx.x : blockIdx.x / (unsigned)nslices, G = FOLD ? gridDim.x : gridDim.x / (unsigned)nslices;
    const uint32_t slice_words = (1u << slice_cell_bits) >> 5;
    if (FOLD) {
        for (uint32_t i = tid; i < slice_words; i += GBN_SLICE_THREADS) {
            uint32_t v = 0;
            for (int f = 0; f < nslices; f++) v |= P.pv[(size_t)f * slice_words + i];
            s_pv[i] = v;
        }
    } else
    for (uint32_t i = tid; i < slice_words; i += GBN_SLICE_THREADS) s_pv[i] = P.pv[(size_t)k * slice_words + i];
    if (tid == 0) s_used = 0;
    __syncthreads();
    uint32_t *q_pos = s_q + wave * 3 * GBN_SLICE_QCAP, *q_cell = q_pos + GBN_SLICE_QCAP, *q_subj = q_cell + GBN_SLICE_QCAP;
    GbnDevSeed *__restrict__ myseg = seg + (size_t)blockIdx.x * seg_cap;
    int qn = 0;                                                         // wave-uniform
    unsigned long long raw = 0;
    const unsigned long long lt = (1ull << lane) - 1;
    const int top = 64 - 2 * P.lut;

    // `cnt` queued lookup hits, a lane each: the cell's entries become seeds in the workgroup's own segment of the
    // output (one LDS atomic per wave; a global counter took 6 of 9 ms: 700,000 atomics on one address per launch)
    auto flush = [&](int first, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // queue slots written by other lanes of this wave
        uint32_t start = 0, n = 0; int32_t s = 0, subj = 0;
        if (lane < cnt) {
            s = (int32_t)q_pos[first + lane]; subj = (int32_t)q_subj[first + lane];
            const uint32_t cell = q_cell[first + lane];
            // (FOLD: the queue holds the RANK of the cell among the present ones, and the entry starts by rank)
            const uint32_t *__restrict__ starts = FOLD ? P.pstart : P.cell_start;
            start = starts[cell]; n = starts[cell + 1] - start;
        }
        raw += n;
        uint32_t incl = n;                                              // inclusive prefix sum over the lanes
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); incl += (lane >= d) ? v : 0u; }
        const uint32_t total = __shfl(incl, 63);
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(&s_used, total);
        base = __shfl(base, 0) + (incl - n);
        for (uint32_t e = 0; e < n; e++) {
            if (base + e < seg_cap) {
                GbnDevSeed sd; sd.subj = subj; sd.s_scan = s; sd.q_pos = (int32_t)(uint32_t)(P.ent[start + e] & 0xffffffffull); sd.ext_left = 0;
                myseg[base + e] = sd;
            }
        }
    };

    // a tile's three dwords per lane are asked for while the tile before is being worked on
    struct Fetch { GbnTile T; int32_t nl, p0; uint32_t d0, d1, d2; };
    auto fetch = [&](int64_t t) {
        Fetch f;
        f.T = P.tiles[t];
        const uint8_t *__restrict__ subj = P.db + P.byte_off[f.T.subj];
        f.nl = min(32, f.T.npos - 32 * lane);                           // positions of this lane (<= 0: none)
        f.p0 = f.T.first_pos + (f.nl > 0 ? 32 * lane : 0);              // always a readable address
        // (p0 is a multiple of 32 bases: tiles start at multiples of 2048 positions, stride 1 -- the lane's 48 bases are
        // three aligned dwords)
        const uint32_t *__restrict__ dw = reinterpret_cast<const uint32_t *>(subj) + (f.p0 >> 4);
        f.d0 = dw[0]; f.d1 = dw[1]; f.d2 = dw[2];
        return f;
    };
    const int64_t t_first = g * (GBN_SLICE_THREADS / 64) + wave, t_step = G * (GBN_SLICE_THREADS / 64);
    Fetch nx;
    if (t_first < P.ntiles) nx = fetch(t_first);
    for (int64_t t = t_first; t < P.ntiles; t += t_step) {
        const Fetch cur = nx;
        if (t + t_step < P.ntiles) nx = fetch(t + t_step);
        const GbnTile T = cur.T;
        const int32_t nl = cur.nl, p0 = cur.p0;
        const uint32_t W[3] = {bswap32(cur.d0), bswap32(cur.d1), bswap32(cur.d2)};
        const uint64_t hi0 = ((uint64_t)W[0] << 32) | W[1];
        const uint32_t lo0 = W[2];
        // the lane's 32 presence tests, independent of each other (the LDS reads go out back to back) ...
        uint32_t hm = 0;
        {
            // The kernel is bound by VALU issue (a wave's instruction takes a SIMD four cycles: 4 slices x 10^9 positions x
            // 14 instructions = 1.7 of its 2.2 ms), so a position's word is cut out of two of the three dwords with one
            // funnel shift (v_alignbit, constant amount) instead of a 64-bit window shifted along.
            // Sixteen at a time: all cells, then all LDS reads, then all tests (written as one loop, the compiler waited
            // for every read before it issued the next one: 32 LDS round trips per tile)
            const int top32 = 32 - 2 * P.lut;                           // x = the 32 bits that start with the word: cell = x >> top32
            const bool one_slice = FOLD || nslices == 1;                // (one slice: top32 + slice_cell_bits = 32, nothing to shift by)
            const int sl_shift = min(31, top32 + slice_cell_bits);      // x >> sl_shift = the slice of the word's cell
            const int wd_shift = top32 + 5, wd_bits = slice_cell_bits - 5;   // bits of x: index of the presence word inside the slice
            #pragma unroll
            for (int i0 = 0; i0 < 32; i0 += 16) {
                uint32_t x[16], w[16];
                #pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int o = 2 * (i0 + j), wi = o >> 5, r = o & 31;
                    x[j] = r ? __builtin_amdgcn_alignbit(W[wi], W[wi + 1], 32 - r) : W[wi];
                }
                // only the lanes whose word lies in this slice read (a quarter of them with four slices): what a random
                // LDS read of a wave costs is its bank conflicts, and those come with the number of lanes
                #pragma unroll
                for (int j = 0; j < 16; j++) {
                    w[j] = 0;
                    if (one_slice || (int)(x[j] >> sl_shift) == k) w[j] = s_pv[__builtin_amdgcn_ubfe(x[j], wd_shift, wd_bits)];
                }
                #pragma unroll
                for (int j = 0; j < 16; j++) hm |= __builtin_amdgcn_ubfe(w[j], __builtin_amdgcn_ubfe(x[j], top32, 5), 1) << (i0 + j);
            }
            hm &= (nl >= 32) ? 0xffffffffu : ((nl > 0) ? ((1u << nl) - 1u) : 0u);
        }
        // FOLD: what passed the filter is looked up in the presence bits themselves, four positions of a lane at a time
        // (four gathers in flight: 512 KB for lut 11, L2-resident), and what is present joins the wave's queue
        if (FOLD) {
            while (__ballot(hm != 0)) {
                uint32_t pos4[4], cell4[4]; uint2 pw[4];
                const uint2 *__restrict__ pvx = reinterpret_cast<const uint2 *>(P.pvx);
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool have = hm != 0;
                    const int i = have ? __ffs(hm) - 1 : 0;
                    hm &= hm - 1;
                    const uint64_t x = i ? ((hi0 << (2 * i)) | (((uint64_t)lo0 << 32) >> (64 - 2 * i))) : hi0;
                    cell4[u] = (uint32_t)(x >> top); pos4[u] = (uint32_t)(p0 + i);
                    pw[u] = have ? pvx[cell4[u] >> 5] : make_uint2(0u, 0u);
                }
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t bit = cell4[u] & 31u;
                    const bool present = (pw[u].x >> bit) & 1u;
                    const unsigned long long m = __ballot(present);
                    if (!m) continue;
                    if (present) {
                        const int at = qn + __popcll(m & lt);
                        q_pos[at] = pos4[u]; q_cell[at] = pw[u].y + (uint32_t)__popc(pw[u].x & ((1u << bit) - 1u)); q_subj[at] = (uint32_t)T.subj;
                    }
                    qn += __popcll(m);
                    if (qn >= 64) { qn -= 64; flush(qn, 64); }
                }
            }
            continue;
        }
        // ... then the present ones join the wave's queue, a round per position a lane still holds (two or three)
        while (true) {
            const unsigned long long m = __ballot(hm != 0);
            if (!m) break;
            if (hm) {
                const int i = __ffs(hm) - 1;
                hm &= hm - 1;
                const uint64_t x = i ? ((hi0 << (2 * i)) | (((uint64_t)lo0 << 32) >> (64 - 2 * i))) : hi0;
                const int at = qn + __popcll(m & lt);
                q_pos[at] = (uint32_t)(p0 + i); q_cell[at] = (uint32_t)(x >> top); q_subj[at] = (uint32_t)T.subj;
            }
            qn += __popcll(m);
            if (qn >= 64) { qn -= 64; flush(qn, 64); }
        }
    }
    if (qn > 0) flush(0, qn);
    if (P.raw_hits) {
        for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
        if (lane == 0 && raw) atomicAdd(P.raw_hits, raw);
    }
    __syncthreads();
    if (tid == 0) {
        seg_count[blockIdx.x] = s_used;
        if (s_used) { atomicAdd(P.seed_count, (unsigned long long)s_used); atomicMax(seg_max, (unsigned long long)s_used); }
    }
}

// ---------------------------------------------------------------------------------------------------
// The folded scan with its seeds IN SCAN ORDER (round 3).  What the seed stage's sort restores -- subject, diagonal slot,
// scan position -- is mostly an order the scan knows and throws away: here a wave takes a contiguous stretch of tiles and
// owns an output segment, a tile's hits enter the wave's queue in position order (lane-major: the lanes hold 32 consecutive
// positions each, so a hit's place follows from the prefix sums of the lanes' hit counts), and the queue is emptied from
// its old end.  The segments read one after the other are then the seeds ordered by (subject, scan position, entry), and
// the composite-key sort has only subject | slot left to do: 19 instead of 39 bits for C3, three passes instead of five.
// Two looks at the presence bits per hit (pvx: once to learn which candidates are hits, once more -- by the hits only, a
// sixth of the first look's reads -- for their ranks when their places are known): cheaper than keeping 32 ranks per lane.
// ---------------------------------------------------------------------------------------------------
#ifndef GBN_OQ_STRIDED
#define GBN_OQ_STRIDED 0
#endif
#define GBN_SLICE_OQ 192            // ring of (position, rank) pairs per wave: 64 left over + 128 of a tile's hits at a time
extern "C" __global__ void __launch_bounds__(GBN_SLICE_THREADS)
scan_fold_ordered_kernel(GbnScanParams P, int nslices, int slice_cell_bits, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                         unsigned long long *seg_max, GbnKeyParams CK)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_slice[];
    uint32_t *s_pv = s_slice;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t slice_words = (1u << slice_cell_bits) >> 5;
    for (uint32_t i = tid; i < slice_words; i += GBN_SLICE_THREADS) {
        uint32_t v = 0;
        for (int f = 0; f < nslices; f++) v |= P.pv[(size_t)f * slice_words + i];
        s_pv[i] = v;
    }
    __syncthreads();
    constexpr int QO = GBN_SLICE_OQ;
    static_assert(2 * GBN_SLICE_OQ <= 3 * GBN_SLICE_QCAP, "the ordered queue lives in the unordered one's LDS");
    uint32_t *q_pos = s_slice + GBN_SLICE_WORDS + wave * 3 * GBN_SLICE_QCAP, *q_rank = q_pos + QO;
    const int64_t gw = (int64_t)blockIdx.x * (GBN_SLICE_THREADS / 64) + wave, NW = (int64_t)gridDim.x * (GBN_SLICE_THREADS / 64);
    const int64_t tpw = (P.ntiles + NW - 1) / NW;
#if GBN_OQ_STRIDED      // timing experiment only (wrong order): tiles dealt round-robin as in scan_slice_body
    const int64_t t_first = gw, t_end = P.ntiles, t_inc = NW; (void)tpw;
#else
    const int64_t t_first = gw * tpw, t_end = min(P.ntiles, t_first + tpw), t_inc = 1;
#endif
    GbnDevSeed *__restrict__ myseg = seg + (size_t)gw * seg_cap;
    uint32_t used = 0;                                                  // wave-uniform: seeds in the wave's segment
    int head = 0, qn = 0;                                               // wave-uniform: the ring's oldest entry, its length
    const unsigned long long lt = (1ull << lane) - 1;
    int32_t cur_subj = -1;
    unsigned long long raw = 0;
    const int top = 64 - 2 * P.lut, top32 = 32 - 2 * P.lut;
    const int wd_shift = top32 + 5, wd_bits = slice_cell_bits - 5;
    const uint2 *__restrict__ pvx = reinterpret_cast<const uint2 *>(P.pvx);
    const uint32_t ck_qmax = (CK.q_bits >= 32) ? 0xffffffffu : ((1u << CK.q_bits) - 1u);

    // the `cnt` oldest queued hits, a lane each: their cells' entries become seeds at the end of the wave's segment
    auto flush = [&](int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t start = 0, n = 0; int32_t sp = 0;
        if (lane < cnt) {
            int idx = head + lane; idx -= idx >= QO ? QO : 0;
            sp = (int32_t)q_pos[idx];
            const uint32_t rank = q_rank[idx];
            start = P.pstart[rank]; n = P.pstart[rank + 1] - start;
        }
        raw += n;
        uint32_t incl = n;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); incl += (lane >= d) ? v : 0u; }
        const uint32_t total = __shfl(incl, 63);
        const uint32_t base = used + (incl - n);
        if (CK.seg_keys) {
            // round 6: the seed leaves as the 8-byte composite key the seed-order kernels would make of it (subject | slot | scan position
            // | value): 8 bytes written here, 8 read by the count and 8 by the scatter kernel, instead of 16 each
            uint64_t *__restrict__ mykeys = reinterpret_cast<uint64_t *>(myseg);
            for (uint32_t e = 0; e < n; e++) {
                if (base + e < seg_cap) {
                    GbnDevSeed sd; sd.subj = cur_subj; sd.s_scan = sp; sd.q_pos = (int32_t)(uint32_t)(P.ent[start + e] & 0xffffffffull); sd.ext_left = 0;
                    uint32_t slot, val;
                    const uint64_t key = gbn_composite_key(CK, sd, ck_qmax, slot, val);
                    mykeys[base + e] = (key << CK.v_bits) | val;
                }
            }
        } else
        for (uint32_t e = 0; e < n; e++) {
            if (base + e < seg_cap) {
                GbnDevSeed sd; sd.subj = cur_subj; sd.s_scan = sp; sd.q_pos = (int32_t)(uint32_t)(P.ent[start + e] & 0xffffffffull); sd.ext_left = 0;
                myseg[base + e] = sd;
            }
        }
        used += total;
        head += cnt; head -= head >= QO ? QO : 0; qn -= cnt;
    };

    struct Fetch { GbnTile T; int32_t nl, p0; uint32_t d0, d1, d2; };
    auto fetch = [&](int64_t t) {
        Fetch f;
        f.T = P.tiles[t];
        const uint8_t *__restrict__ subj = P.db + P.byte_off[f.T.subj];
        f.nl = min(32, f.T.npos - 32 * lane);
        f.p0 = f.T.first_pos + (f.nl > 0 ? 32 * lane : 0);
        const uint32_t *__restrict__ dw = reinterpret_cast<const uint32_t *>(subj) + (f.p0 >> 4);
        f.d0 = dw[0]; f.d1 = dw[1]; f.d2 = dw[2];
        return f;
    };
    Fetch nx;
    if (t_first < t_end) nx = fetch(t_first);
    for (int64_t t = t_first; t < t_end; t += t_inc) {
        const Fetch cur = nx;
        if (t + t_inc < t_end) nx = fetch(t + t_inc);
        const GbnTile T = cur.T;
        const int32_t nl = cur.nl, p0 = cur.p0;
        const uint32_t W[3] = {bswap32(cur.d0), bswap32(cur.d1), bswap32(cur.d2)};
        const uint64_t hi0 = ((uint64_t)W[0] << 32) | W[1];
        const uint32_t lo0 = W[2];
        if (T.subj != cur_subj) {                                       // (the queue's entries carry no subject)
            while (qn > 0) flush(min(qn, 64));
            cur_subj = T.subj;
        }
        // the filter: 32 LDS reads per lane, as in scan_slice_body
        uint32_t hm = 0;
        #pragma unroll
        for (int i0 = 0; i0 < 32; i0 += 16) {
            uint32_t x[16], w[16];
            #pragma unroll
            for (int j = 0; j < 16; j++) {
                const int o = 2 * (i0 + j), wi = o >> 5, r = o & 31;
                x[j] = r ? __builtin_amdgcn_alignbit(W[wi], W[wi + 1], 32 - r) : W[wi];
            }
            #pragma unroll
            for (int j = 0; j < 16; j++) w[j] = s_pv[__builtin_amdgcn_ubfe(x[j], wd_shift, wd_bits)];
            #pragma unroll
            for (int j = 0; j < 16; j++) hm |= __builtin_amdgcn_ubfe(w[j], __builtin_amdgcn_ubfe(x[j], top32, 5), 1) << (i0 + j);
        }
        hm &= (nl >= 32) ? 0xffffffffu : ((nl > 0) ? ((1u << nl) - 1u) : 0u);
        auto cell_at = [&](int i) -> uint32_t {
            const uint64_t x = i ? ((hi0 << (2 * i)) | (((uint64_t)lo0 << 32) >> (64 - 2 * i))) : hi0;
            return (uint32_t)(x >> top);
        };
        // The candidates that are hits, four of a lane's at a time.  A hit joins the ring at once, behind what is queued, with
        // its rank -- in the order the hits turn up; their places follow below.  (More hits than the ring has room for --
        // a tile of repeats --: the slow way further down, which looks the ranks up a second time.)
        uint32_t tm = 0;
        int np = 0; bool ovf = false;                                   // wave-uniform: hits written, gave up
        const int room = min(QO - qn, 128), pbase = head + qn;
        while (__ballot(hm != 0)) {
            uint32_t bit4[4]; uint2 pw[4]; int i4[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool have = hm != 0;
                i4[u] = have ? __ffs(hm) - 1 : 0;
                hm &= hm - 1;
                const uint32_t cell = cell_at(i4[u]);
                bit4[u] = cell & 31u;
                pw[u] = have ? pvx[cell >> 5] : make_uint2(0u, 0u);
            }
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool present = (pw[u].x >> bit4[u]) & 1u;
                tm |= (present ? 1u : 0u) << i4[u];
                const unsigned long long m = __ballot(present);
                if (!m) continue;
                const int cm = __popcll(m);
                if (!ovf && np + cm <= room) {
                    if (present) {
                        int idx = pbase + np + __popcll(m & lt); idx -= idx >= QO ? QO : 0;
                        q_pos[idx] = (uint32_t)(p0 + i4[u]); q_rank[idx] = pw[u].y + (uint32_t)__popc(pw[u].x & ((1u << bit4[u]) - 1u));
                    }
                    np += cm;
                } else ovf = true;
            }
        }
        // their places: lane-major = position order
        const uint32_t cnt_l = (uint32_t)__popc(tm);
        uint32_t incl = cnt_l;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); incl += (lane >= d) ? v : 0u; }
        const uint32_t excl = incl - cnt_l, total = __shfl(incl, 63);
        if (!total) continue;
        if (!ovf) {
            // every entry to its place: the lane that owns a hit's position knows how many hits lie in front of it
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t ep[2], er[2]; int dst[2];
            #pragma unroll
            for (int k = 0; k < 2; k++) {
                const int j = lane + 64 * k;
                int idx = pbase + min(j, max(np - 1, 0)); idx -= idx >= QO ? QO : 0;
                ep[k] = q_pos[idx]; er[k] = q_rank[idx];
                const uint32_t rel = ep[k] - (uint32_t)T.first_pos;
                const int owner = (int)(rel >> 5) & 63;
                const uint32_t o_excl = __shfl(excl, owner), o_tm = __shfl(tm, owner);
                int d = pbase + (int)(o_excl + (uint32_t)__popc(o_tm & ((1u << (rel & 31u)) - 1u))); d -= d >= QO ? QO : 0;
                dst[k] = j < np ? d : -1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            #pragma unroll
            for (int k = 0; k < 2; k++) if (dst[k] >= 0) { q_pos[dst[k]] = ep[k]; q_rank[dst[k]] = er[k]; }
            qn += np;
            while (qn >= 64) flush(64);
            continue;
        }
        // second look, by the hits only: their ranks -- the first four of a lane asked for at once, before the places are
        // worked out (a lane with more goes on two at a time below)
        uint32_t tmc = tm;
        uint32_t hpos[4], hbit[4]; uint2 hw[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool have = tmc != 0;
            const int i = have ? __ffs(tmc) - 1 : 0;
            tmc &= tmc - 1;
            const uint32_t cell = cell_at(i);
            hpos[u] = (uint32_t)(p0 + i); hbit[u] = cell & 31u;
            hw[u] = have ? pvx[cell >> 5] : make_uint2(0u, 0u);
        }
        // at most QO - 64 hits join the queue at a time: lanes in chunks by their place (a tile full of hits takes 22 rounds)
        constexpr uint32_t CH = QO - 64 - 32;
        const uint32_t chunk = excl / CH, cmax = __shfl(chunk, 63);
        for (uint32_t c = 0; c <= cmax; c++) {
            const unsigned long long in_c = __ballot(chunk == c);
            if (!in_c) continue;
            const int f = (int)__builtin_ctzll(in_c), g = 63 - (int)__builtin_clzll(in_c);
            const uint32_t first_excl = __shfl(excl, f), c_total = __shfl(incl, g) - first_excl;
            if (chunk == c) {
                int at = head + qn + (int)(excl - first_excl);
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    if ((uint32_t)u < cnt_l) {
                        int a0 = at + u; a0 -= a0 >= QO ? QO : 0;
                        q_pos[a0] = hpos[u]; q_rank[a0] = hw[u].y + (uint32_t)__popc(hw[u].x & ((1u << hbit[u]) - 1u));
                    }
                }
                at += 4;
                while (tmc) {
                    const int i0 = __ffs(tmc) - 1; tmc &= tmc - 1;
                    const bool two = tmc != 0;
                    const int i1 = two ? __ffs(tmc) - 1 : i0; tmc &= tmc - 1;      // (0 & anything: stays 0)
                    const uint32_t c0 = cell_at(i0), c1 = cell_at(i1);
                    const uint2 w0 = pvx[c0 >> 5], w1 = pvx[c1 >> 5];
                    int a0 = at; a0 -= a0 >= QO ? QO : 0;
                    q_pos[a0] = (uint32_t)(p0 + i0); q_rank[a0] = w0.y + (uint32_t)__popc(w0.x & ((1u << (c0 & 31u)) - 1u));
                    if (two) {
                        int a1 = at + 1; a1 -= a1 >= QO ? QO : 0;
                        q_pos[a1] = (uint32_t)(p0 + i1); q_rank[a1] = w1.y + (uint32_t)__popc(w1.x & ((1u << (c1 & 31u)) - 1u));
                    }
                    at += 2;
                }
            }
            qn += (int)c_total;
            while (qn >= 64) flush(64);
        }
    }
    while (qn > 0) flush(min(qn, 64));
    if (P.raw_hits) {
        for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
        if (lane == 0 && raw) atomicAdd(P.raw_hits, raw);
    }
    if (lane == 0) {
        seg_count[gw] = used;
        if (used) { atomicAdd(P.seed_count, (unsigned long long)used); atomicMax(seg_max, (unsigned long long)used); }
    }
}

extern "C" __global__ void __launch_bounds__(GBN_SLICE_THREADS)
scan_slice_kernel(GbnScanParams P, int nslices, int slice_cell_bits, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                  unsigned long long *seg_max)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_slice[];
    __shared__ uint32_t s_used;                                         // seeds this workgroup has put into its segment
    scan_slice_body<false>(P, nslices, slice_cell_bits, seg, seg_cap, seg_count, seg_max, s_slice, s_used);
}

extern "C" __global__ void __launch_bounds__(GBN_SLICE_THREADS)
scan_fold_kernel(GbnScanParams P, int nslices, int slice_cell_bits, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                 unsigned long long *seg_max)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_slice[];
    __shared__ uint32_t s_used;
    scan_slice_body<true>(P, nslices, slice_cell_bits, seg, seg_cap, seg_count, seg_max, s_slice, s_used);
}

// the segments of scan_slice_kernel, back to back: seeds[0 .. sum of counts) -- for the consumers that want the seeds
// in one array (the composite-key seed stage reads the segments as they are)
extern "C" __global__ void __launch_bounds__(256)
seed_compact_kernel(const GbnDevSeed *__restrict__ seg, const uint32_t *__restrict__ seg_count, const unsigned long long *__restrict__ seg_first,
                    int nseg, uint32_t seg_cap, GbnDevSeed *__restrict__ out, unsigned long long out_cap)
{
    // (seg_first: seg_first_kernel's prefix sums, launched in front)
    const int sg = blockIdx.x % nseg, part = blockIdx.x / nseg, nparts = gridDim.x / nseg;
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(seg + (size_t)sg * seg_cap);
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(out);
    const unsigned long long at = seg_first[sg];
    const uint32_t have = min(seg_count[sg], seg_cap);
    for (uint32_t i = (uint32_t)part * 256u + threadIdx.x; i < have; i += (uint32_t)nparts * 256u)
        if (at + i < out_cap) dst[at + i] = src[i];
}

// ... the same for segments of composite keys (round 6): every key decoded to the seed it stands for -- the way out for a range whose
// seeds turn out not to go through the seed-order kernels after all (too few of them, a switch)
extern "C" __global__ void __launch_bounds__(256)
seed_compact_keys_kernel(const uint64_t *__restrict__ seg, const uint32_t *__restrict__ seg_count, const unsigned long long *__restrict__ seg_first,
                         int nseg, uint32_t seg_cap, GbnDevSeed *__restrict__ out, unsigned long long out_cap, GbnKeyParams CK)
{
    const int sg = blockIdx.x % nseg, part = blockIdx.x / nseg, nparts = gridDim.x / nseg;
    const uint64_t *__restrict__ src = seg + (size_t)sg * seg_cap * 2;          // (a segment is seg_cap 16-byte elements long whatever it holds)
    const unsigned long long at = seg_first[sg];
    const uint32_t have = min(seg_count[sg], seg_cap);
    const uint32_t qmax = (CK.q_bits >= 32) ? 0xffffffffu : ((1u << CK.q_bits) - 1u);
    for (uint32_t i = (uint32_t)part * 256u + threadIdx.x; i < have; i += (uint32_t)nparts * 256u)
        if (at + i < out_cap) out[at + i] = gbn_seed_of_key(CK, src[i], qmax);
}

// ---------------------------------------------------------------------------
// deterministic synthetic database bytes: xorshift64* streams, one per 4 KiB
// ---------------------------------------------------------------------------
extern "C" __global__ void synth_fill_kernel(uint64_t *out, int64_t nwords, uint64_t seed)
{
    int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t w0 = chunk * 512;
    if (w0 >= nwords) return;
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(chunk + 1));
    // splitmix64 scramble so neighbouring chunks decorrelate
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    if (x == 0) x = 0x9E3779B97F4A7C15ull;
    int64_t w1 = min(nwords, w0 + 512);
    for (int64_t w = w0; w < w1; w++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        out[w] = x * 0x2545F4914F6CDD1Dull;
    }
}

// ---------------------------------------------------------------------------
// Repeats written over a synthetic shard (bench.py --skew; gblastn_amd/synth.py: skew_subject is the numpy form, bit for bit).
// Real nucleotide databases are repeat- and low-complexity-rich; i.i.d. uniform bases exercise none of what that does to a scan
// (lookup words piling up in a few cells and bins, seed floods).  Per subject of nb packed bytes, from splitmix64 hashes of
// (seed, global oid): FOUR stretches of nb / 50 bytes each (8 % of the subject) -- homopolymer runs (one of AAAA, CCCC, GGGG, TTTT)
// or tandem repeats of a 4 .. 24-base unit --, and in one subject of fifty ONE copy of a family element of 1,200 bases (the
// same 300 bytes for the whole database, a base changed in ~1.2 % of the positions per copy).
// ---------------------------------------------------------------------------
__host__ __device__ inline uint64_t gbn_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
extern "C" __global__ void synth_skew_kernel(uint8_t *slab, int64_t first_off, int64_t stride, int64_t nb, int32_t num, int64_t first_oid, uint64_t seed)
{
    const int32_t s = (int32_t)blockIdx.x;
    if (s >= num || nb < 2000) return;
    uint8_t *p = slab + first_off + (int64_t)s * stride;
    const uint64_t oid = (uint64_t)(first_oid + s);
    const int64_t rl = nb / 50;
    for (int r = 0; r < 4; r++) {
        const uint64_t h = gbn_splitmix64(seed ^ gbn_splitmix64(oid * 4 + (uint64_t)r));
        const int64_t start = (int64_t)(h % (uint64_t)(nb - rl));
        const uint64_t h2 = gbn_splitmix64(h);
        if (((h >> 40) & 1) == 0) {
            const uint8_t b = (uint8_t)(0x55u * ((h >> 42) & 3));
            for (int64_t j = threadIdx.x; j < rl; j += blockDim.x) p[start + j] = b;
        } else {
            const int64_t period = 1 + (int64_t)((h >> 44) % 6);
            for (int64_t j = threadIdx.x; j < rl; j += blockDim.x) p[start + j] = (uint8_t)(h2 >> (8 * (j % period)));
        }
        __syncthreads();                    // (stretches may overlap: the later one wins, as in the numpy form)
    }
    const uint64_t he = gbn_splitmix64(seed ^ gbn_splitmix64(oid ^ 0xE1E1E1E1ull));
    if (he % 50 == 0) {
        const int64_t at = (int64_t)(gbn_splitmix64(he) % (uint64_t)(nb - 300));
        for (int64_t k = threadIdx.x; k < 300; k += blockDim.x) {
            uint8_t e = (uint8_t)(gbn_splitmix64(seed ^ (0xFA111ull + (uint64_t)(k >> 3))) >> (8 * (k & 7)));
            const uint64_t hm = gbn_splitmix64(he ^ (uint64_t)(k + 1));
            if ((hm & 15) == 0) e ^= (uint8_t)(((hm >> 4) & 3) << (2 * ((hm >> 6) & 3)));
            p[at + k] = e;
        }
    }
}

// stretches of the shard packed back to back (the traceback stage reads back what its extensions can reach:
// one kernel and one copy per query batch instead of a copy per subject)
extern "C" __global__ void gather_bytes_kernel(const uint8_t *src, const int64_t *src_off, const int64_t *dst_off, const int32_t *nbytes, uint8_t *dst)
{
    const int64_t so = src_off[blockIdx.x], d0 = dst_off[blockIdx.x];
    for (int32_t i = threadIdx.x; i < nbytes[blockIdx.x]; i += blockDim.x) dst[d0 + i] = src[so + i];
}

// ---------------------------------------------------------------------------
// host launchers (thin; the C ABI in abi.cpp re-exports them)
// ---------------------------------------------------------------------------

namespace gbn {

hipError_t launch_scan_seed(const GbnScanParams &p, int grid, hipStream_t st)
{
    if (p.ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(scan_seed_kernel, dim3(grid), dim3(GBN_SCAN_THREADS), 0, st, p);
    return hipGetLastError();
}

// slices of the presence bits scan_slice_kernel needs for this table (0: not a table it takes)
int scan_slice_count(const GbnScanParams &p)
{
    if (p.mode != GBN_EXT_DIRECT || p.step != 1 || p.lut != p.word || p.lut < 3 || p.ncells < 32) return 0;
    const int cell_bits = std::min(2 * p.lut, GBN_SLICE_CELL_BITS);
    const int64_t n = p.ncells >> cell_bits;
    return (n >= 1 && n <= GBN_SLICE_MAX) ? (int)n : 0;
}

// workgroups scan_slice_kernel is launched with for this table on a chip of num_cu CUs (= segments of its output)
hipError_t launch_seg_first(const GbnKeyParams &k, hipStream_t st);       // seed_order.hip
int scan_slice_blocks(const GbnScanParams &p, int num_cu)
{
    const int nslices = scan_slice_count(p);
    if (nslices <= 0) return 0;
    // one workgroup per CU (its slice fills the LDS); every slice gets the same number of workgroups
    const int64_t waves_needed = (p.ntiles + (GBN_SLICE_THREADS / 64) - 1) / (GBN_SLICE_THREADS / 64);
    // (the ordered form has a segment per WAVE: sixteen per workgroup, GBN_SLICE_SEGS in all -- a part with more than
    // GBN_SLICE_SEGS / 16 = 256 CUs gets 256 workgroups, whichever form the launch takes, so that the segment tables
    // of the engine and of seg_first_kernel always hold them)
    const int max_blocks = std::min(num_cu, GBN_SLICE_SEGS / (GBN_SLICE_THREADS / 64));
    const int per_slice = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, max_blocks / nslices), waves_needed));
    return per_slice * nslices;
}

// seg: scan_slice_blocks() segments of seg_cap seeds each, seg_count: as many counters; *p.seed_count receives the number
// of seeds, *seg_max the fullest segment's count (above seg_cap: seeds were dropped, scan again with longer segments).
// p.seeds is not written: launch_seed_compact puts the segments back to back for whoever wants them in one array
hipError_t launch_seed_compact(const GbnDevSeed *seg, const uint32_t *seg_count, unsigned long long *seg_first, int nseg, uint32_t seg_cap,
                               GbnDevSeed *out, unsigned long long out_cap, hipStream_t st, const GbnKeyParams *keys)
{
    if (nseg <= 0) return hipSuccess;
    if (!seg_first || nseg > GBN_SLICE_SEGS) return hipErrorInvalidValue;
    GbnKeyParams k; std::memset(&k, 0, sizeof(k)); k.seg_count = seg_count; k.nseg = nseg; k.seg_cap = seg_cap; k.seg_first = seg_first;
    if (hipError_t e = launch_seg_first(k, st)) return e;
    if (keys && keys->seg_keys) {           // segments of composite keys: decoded on the way
        hipLaunchKernelGGL(seed_compact_keys_kernel, dim3((unsigned)(nseg * (nseg > 1024 ? 1 : 8))), dim3(256), 0, st, reinterpret_cast<const uint64_t *>(seg), seg_count, seg_first, nseg, seg_cap, out, out_cap, *keys);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(seed_compact_kernel, dim3((unsigned)(nseg * (nseg > 1024 ? 1 : 8))), dim3(256), 0, st, seg, seg_count, seg_first, nseg, seg_cap, out, out_cap);
    return hipGetLastError();
}

// the form launch_scan_slice takes for this table: 0 a pass per slice, 1 folded, 2 folded with the seeds in scan order
static int scan_slice_form(const GbnScanParams &p, int nslices)
{
    // GBN_SLICE_FOLD=0: a pass over the subjects per slice (the form before the folded filter); GBN_SCAN_ORDERED=0: folded,
    // seeds in no particular order (the form before the ordered one) -- for comparisons
    const bool fold = gbn::switch_value("GBN_SLICE_FOLD", 1) != 0;
    const bool ordered = gbn::switch_value("GBN_SCAN_ORDERED", 1) != 0;
    if (!(fold && nslices > 1 && p.pvx && p.pstart)) return 0;
    return ordered ? 2 : 1;
}

// output segments of launch_scan_slice for this table (one per workgroup, or one per wave with the seeds in scan order:
// then the segments read one after the other are ordered by subject, scan position, entry -- *ordered says which)
int scan_slice_segments(const GbnScanParams &p, int num_cu, int *ordered)
{
    const int nslices = scan_slice_count(p), blocks = scan_slice_blocks(p, num_cu);
    const int form = nslices > 0 ? scan_slice_form(p, nslices) : 0;
    if (ordered) *ordered = form == 2;
    return form == 2 ? blocks * (GBN_SLICE_THREADS / 64) : blocks;
}

hipError_t launch_scan_slice(const GbnScanParams &p, int num_cu, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                             unsigned long long *seg_max, hipStream_t st, const GbnKeyParams *keys)
{
    if (p.ntiles <= 0) return hipSuccess;
    const int nslices = scan_slice_count(p), blocks = scan_slice_blocks(p, num_cu);
    if (nslices <= 0 || blocks <= 0) return hipErrorInvalidValue;
    const int cell_bits = std::min(2 * p.lut, GBN_SLICE_CELL_BITS);
    const size_t lds = ((size_t)GBN_SLICE_WORDS + (size_t)(GBN_SLICE_THREADS / 64) * 3 * GBN_SLICE_QCAP) * 4;
    static std::atomic<uint64_t> attr_set{0}, attr_set_fold{0}, attr_set_ord{0};
    const int form = scan_slice_form(p, nslices);
    if (form == 2) {
        if (hipError_t e = raise_dynamic_lds((const void *)scan_fold_ordered_kernel, lds, attr_set_ord)) return e;
        GbnKeyParams ck; std::memset(&ck, 0, sizeof(ck));
        if (keys && keys->seg_keys) ck = *keys;             // (only this form can write keys: its seeds come subject by subject)
        hipLaunchKernelGGL(scan_fold_ordered_kernel, dim3((unsigned)blocks), dim3(GBN_SLICE_THREADS), lds, st, p, nslices, cell_bits, seg, seg_cap, seg_count, seg_max, ck);
        return hipGetLastError();
    }
    if (form == 1) {
        if (hipError_t e = raise_dynamic_lds((const void *)scan_fold_kernel, lds, attr_set_fold)) return e;
        hipLaunchKernelGGL(scan_fold_kernel, dim3((unsigned)blocks), dim3(GBN_SLICE_THREADS), lds, st, p, nslices, cell_bits, seg, seg_cap, seg_count, seg_max);
        return hipGetLastError();
    }
    if (hipError_t e = raise_dynamic_lds((const void *)scan_slice_kernel, lds, attr_set)) return e;
    hipLaunchKernelGGL(scan_slice_kernel, dim3((unsigned)blocks), dim3(GBN_SLICE_THREADS), lds, st, p, nslices, cell_bits, seg, seg_cap, seg_count, seg_max);
    return hipGetLastError();
}

hipError_t launch_gather_bytes(const uint8_t *src, const int64_t *src_off, const int64_t *dst_off, const int32_t *nbytes,
                               int32_t n, uint8_t *dst, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_bytes_kernel, dim3((unsigned)n), dim3(256), 0, st, src, src_off, dst_off, nbytes, dst);
    return hipGetLastError();
}

hipError_t launch_synth_skew(void *dev, int64_t first_off, int64_t stride, int64_t nb, int32_t num, int64_t first_oid, uint64_t seed, hipStream_t st)
{
    if (num <= 0) return hipSuccess;
    hipLaunchKernelGGL(synth_skew_kernel, dim3((unsigned)num), dim3(256), 0, st, (uint8_t *)dev, first_off, stride, nb, num, first_oid, seed);
    return hipGetLastError();
}
hipError_t launch_synth_fill(void *dev, int64_t nbytes, uint64_t seed, hipStream_t st)
{
    int64_t nwords = nbytes / 8;
    int64_t chunks = (nwords + 511) / 512;
    if (chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st,
                       (uint64_t *)dev, nwords, seed);
    return hipGetLastError();
}
}  // namespace gbn
