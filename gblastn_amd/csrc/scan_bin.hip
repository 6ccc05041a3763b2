// scan_bin.hip -- the key-range partitioned scan (gfx950 / CDNA4): scan_bin_kernel* (phase 1: every scan position of
// the subjects becomes a 6-byte record in the stream of its lookup word's bin), probe_bin_kernel (phase 2: a bin's slice
// of the cell table in LDS, its records streamed through), probe_rare_kernel (phase 3: exact verification of the
// survivors to word_size).  Together: TNaScanSubjectFunction + the TNaExtendFunction mini-extension
// (CORE/blast_nascan.c:1489-2072, CORE/na_ungapped.c:1025-1555).  Integer work only: no MFMA.
#include "scan_dev.hpp"
#include <type_traits>

#ifndef GBN_BIN_ABL
#define GBN_BIN_ABL 0       // timing experiments only (scan_bin_kernel), bits: 2 no record stores, 4 no `hi` stores, 8 no index stores, 16 rank atomics without return (rank = k): wrong results
#endif
// In-situ marginal cost of one class of LDS operations (round 5, profiles/r05_lds_atomics.txt): the class is issued TWICE, everything
// else -- barriers, loads, stores, the other LDS traffic -- stays as it is, and the results stay right.  Bits: 1 the rank atomics
// (a second, shadow histogram), 2 the descriptor reads, 4 the scatter writes, 8 the store phase's LDS reads.
#ifndef GBN_BIN_DUP
#define GBN_BIN_DUP 0
#endif
#ifndef GBN_BIN_PRESENCE
#define GBN_BIN_PRESENCE 0      // experiment (round 6): 1 = presence-filtered binning, 2 = the presence lookups alone (see scan_bin_line_body)
#endif

// ===========================================================================
// Key-range partitioned scan (lookup tables too large for L2).
//
// Direct probing costs one L2 request per scan position for the presence bit
// (2.9e9 per 50 Gbp pass: the L2 request rate, not HBM, is the wall) plus one
// 64-byte HBM sector per present word.  Here phase 1 touches no table at all:
// it streams the subject once and writes every scan position as a 6-byte
// record {cell inside the bin + 15 neighbouring subject bits, 16-bit index} into
// the stream of (bin given by the top bits of its lookup word, workgroup), in
// complete aligned pieces.  Phase 2 walks bin by bin with the bin's cell table
// resident in LDS (one workgroup per CU, all workgroups with the same
// blockIdx & 7 -- observed to share an XCD and its L2 -- on the same bin).
// Only ~0.7 % of the records (fingerprint survivors and cells with >= 3 entries)
// leave LDS, through per-workgroup queues, for phase 3 (exact verification).
// ===========================================================================

// The binning kernel in use: the four-barrier form of rounds 2 and 3 with the record layout and the key extraction of
// round 4 (see scan_bin3_body below for the layout).  Round 4 wrote two other forms of it -- the steps of two tiles in
// flight between two barriers, and scan_bin3_body (no owner threads, descriptors or staging area at all) -- and measured
// all three within 5 % of each other, this one ahead (profiles/r04_bin_*): -DGBN_BIN_V1=0 builds
// scan_bin3_body instead.
#ifndef GBN_BIN_V1
#define GBN_BIN_V1 1
#endif
#if GBN_BIN_V1
template <int STEP, int LUT>
__device__ __forceinline__ void scan_bin_line_body(const GbnBinParams &B)
{
    const GbnScanParams &P = B.S;
    constexpr int TILE = GBN_BIN_TILE_POS, PER = TILE / GBN_SORT_THREADS, LINE = GBN_OPEN_LINE, LP = LINE / 4;    // LP lanes store one line
    constexpr int STAGE = GBN_BIN_STAGE;
    static_assert(PER == 8 && GBN_SORT_THREADS == 1024 && LINE == 32, "8192-position tiles, 1024 threads, 32-record lines");
    static_assert(GBN_BIN_MAXNB <= GBN_SORT_THREADS, "one owner thread per bin");
    // records in LDS: hi word and 16-bit index in the tile, in two arrays with the same slot numbers.
    // [0, STAGE): staging, bin-sorted (the lines a bin completes beyond its first one in a tile: rare with 512
    // bins, the rule with 128); [STAGE, STAGE + bins * LINE): one line under construction per bin
    __shared__ __attribute__((aligned(16))) uint32_t s_hi[STAGE + GBN_BIN_MAXNB * LINE];
    __shared__ __attribute__((aligned(16))) uint16_t s_ix[STAGE + GBN_BIN_MAXNB * LINE];
    __shared__ uint32_t s_hist[GBN_BIN_MAXNB];
#if GBN_BIN_DUP & 1
    __shared__ uint32_t s_hist2[GBN_BIN_MAXNB];
#endif
    __shared__ uint32_t s_wtot[GBN_BIN_MAXNB / 64];
    // scatter descriptor of a bin (r = rank of a record among the tile's records of the bin):
    //   .x [15:0]  slot of r = 0 while the open line has room     [31:16] the same for the staging area (signed)
    //   .y [15:0]  first r that lies past the bin's last complete line (0xffff: none)   [31:16] room in the open line
    __shared__ uint2 s_pk[GBN_BIN_MAXNB];
    __shared__ uint2 s_line[TILE / LINE + GBN_BIN_MAXNB];     // complete line of this tile: .x = stream line (32 records) it becomes, .y = its first LDS slot
    __shared__ uint32_t s_nlines;
    const int tid = threadIdx.x;
    const int lut = LUT > 0 ? LUT : P.lut;
    const uint32_t mask = LUT > 0 ? (uint32_t)((1ull << (2 * LUT)) - 1) : (uint32_t)(P.ncells - 1);
    const int cbits = LUT > 0 ? GBN_BIN_CBITS(LUT) : B.cbits;
    const int nb = LUT > 0 ? (int)(((int64_t)1 << (2 * LUT)) >> GBN_BIN_CBITS(LUT)) : B.nb;
    const uint32_t lowmask = cbits == 15 ? 0xffffu : (1u << cbits) - 1;
    const int cshift = 56 - 2 * lut, rshift = 49 - 2 * lut;
    const uint32_t ustep = (uint32_t)P.step;
    const int64_t stride = gridDim.x, last = P.ntiles - 1;
    const uint32_t wid = blockIdx.x;

    // a lane owns PER consecutive positions = 16 * STEP bits of subject: a whole number of dwords for
    // even strides, half a dword extra for odd lanes of odd strides (the raw dwords are then shifted by
    // 16 bits first, after which every window is cut out with compile-time shifts as before)
    constexpr int NDW = STEP > 0 ? ((2 * STEP * (PER - 1) - 8 + 38) >> 5) + 4 : 2 * PER;
    struct Raw { uint32_t d[NDW]; };
    auto idx_of = [&](int k) -> uint32_t { return STEP > 0 ? (uint32_t)(tid * PER + k) : (uint32_t)(tid + k * GBN_SORT_THREADS); };
    auto upos_of = [&](const GbnTile &t, int k) -> uint32_t {
        const uint32_t i = min(idx_of(k), (uint32_t)t.npos - 1u);
        return (uint32_t)t.first_pos + i * ustep + 60u;
    };
    auto lane_half = [&](const GbnTile &t) -> uint32_t {    // lane's first base, in units of 8 bases (16 bits), from the tile start
        return min((uint32_t)tid, ((uint32_t)t.npos - 1u) / PER) * (uint32_t)STEP;
    };
    auto fetch = [&](const GbnTile &t, Raw &r) {
        if constexpr (STEP > 0) {
            const uint8_t *p = P.db + ((size_t)(uint32_t)t.off16 << 4) + 4 * ((size_t)((uint32_t)t.first_pos >> 4) + (size_t)(lane_half(t) >> 1)) - 4;
            #pragma unroll
            for (int i = 0; i + 4 <= NDW; i += 4) __builtin_memcpy(&r.d[i], p + 4 * i, 16);
            if constexpr (NDW % 4 == 3) { __builtin_memcpy(&r.d[NDW - 3], p + 4 * (NDW - 3), 12); }
            else if constexpr (NDW % 4 == 2) { __builtin_memcpy(&r.d[NDW - 2], p + 4 * (NDW - 2), 8); }
            else if constexpr (NDW % 4 == 1) { __builtin_memcpy(&r.d[NDW - 1], p + 4 * (NDW - 1), 4); }
        } else {
            #pragma unroll
            for (int k = 0; k < PER; k++)
                __builtin_memcpy(&r.d[2 * k], P.db + ((size_t)(uint32_t)t.off16 << 4) - 16 + (upos_of(t, k) >> 2), 8);
        }
    };
    auto keys_all = [&](const GbnTile &t, const Raw &r, uint32_t (&bin)[PER], uint32_t (&hi)[PER]) {
        uint32_t x[NDW];
        if constexpr (STEP > 0) {
            // big-endian dwords; odd lanes of odd strides start half a dword later: one byte permute does both
            const uint32_t sel = ((STEP & 1) && (lane_half(t) & 1u)) ? 0x06070001u : 0x04050607u;
            #pragma unroll
            for (int i = 0; i + 1 < NDW; i++) x[i] = __builtin_amdgcn_perm(r.d[i], r.d[i + 1], sel);
            x[NDW - 1] = bswap32(r.d[NDW - 1]);
        }
        #pragma unroll
        for (int k = 0; k < PER; k++) {
            if constexpr (STEP > 0 && LUT == 12) {
                // window from 4 bases in front of the word: [31:24] those bases, [23:0] the word; and the 32 bits
                // from its last bit on: [30:24] the 7 bits behind the word
                const int bit = 2 * STEP * k - 8 + 32, a = bit >> 5, o = bit & 31;
                const int bit2 = bit + 31, a2 = bit2 >> 5, o2 = bit2 & 31;
                const uint32_t w0 = o ? __builtin_amdgcn_alignbit(x[a], x[a + 1], 32 - o) : x[a];
                const uint32_t w1 = o2 ? __builtin_amdgcn_alignbit(x[a2], x[a2 + 1 < NDW ? a2 + 1 : NDW - 1], 32 - o2) : x[a2];
                bin[k] = (w0 >> 15) & 0x1ffu;
                hi[k] = (__builtin_amdgcn_perm(w1, w0, 0x07030100u) & 0x7fffffffu) | ((w1 << 8) & 0x80000000u);     // (bit 31: the EIGHTH subject bit behind the word, w1's bit 23 -- round 5)
            } else {
                uint64_t w;
                if constexpr (STEP > 0) {
                    const int bit = 2 * STEP * k - 8 + 32, a = bit >> 5, o = bit & 31;
                    const uint32_t x2 = x[a + 2 < NDW ? a + 2 : NDW - 1];
                    const uint32_t hi32 = o ? ((x[a] << o) | (x[a + 1] >> (32 - o))) : x[a];
                    const uint32_t lo32 = o ? ((x[a + 1] << o) | (x2 >> (32 - o))) : x[a + 1];
                    w = ((uint64_t)hi32 << 32) | lo32;
                } else {
                    uint64_t raw; __builtin_memcpy(&raw, &r.d[2 * k], 8);
                    w = __builtin_bswap64(raw) << (2 * (upos_of(t, k) & 3));
                }
                const uint32_t c = (uint32_t)(w >> cshift) & mask;
                bin[k] = c >> cbits;
                hi[k] = (c & lowmask) | ((uint32_t)(w >> 56) << 16) | (((uint32_t)(w >> rshift) & 0x7fu) << 24) | (((uint32_t)(w >> (rshift - 1)) & 1u) << 31);
            }
        }
    };
    auto uniform = [](GbnTile t) -> GbnTile {
        t.subj = __builtin_amdgcn_readfirstlane(t.subj); t.first_pos = __builtin_amdgcn_readfirstlane(t.first_pos);
        t.npos = __builtin_amdgcn_readfirstlane(t.npos); t.off16 = __builtin_amdgcn_readfirstlane(t.off16);
        return t;
    };
    // an eighth of a line (4 records: 16 bytes of hi words, 8 bytes of indices) from LDS slot `src` to stream line
    // `dl`: the 8 lanes of a line write 128 aligned bytes of hi words and 64 of indices -- scattered writes cost
    // by the piece below 128 bytes (tools/write_microbench.hip: 64 + 32 byte pieces 3.5 TB/s, 128 + 64: 6+)
    uint32_t *const rec32 = B.rec; uint16_t *const rec16 = reinterpret_cast<uint16_t *>(B.rec);
    auto store_part = [&](uint32_t dl, uint32_t p, uint32_t src) {
        uint4 h = *reinterpret_cast<const uint4 *>(&s_hi[src]);
        uint2 x = *reinterpret_cast<const uint2 *>(&s_ix[src]);
#if GBN_BIN_DUP & 8
        {
            // (the neighbouring quarter line: same banks pattern, an address the compiler cannot fold into the first read)
            const uint4 h2 = *reinterpret_cast<const uint4 *>(&s_hi[src ^ 4u]);
            const uint2 x2 = *reinterpret_cast<const uint2 *>(&s_ix[src ^ 4u]);
            h.x |= h2.x & h.x & 0x80000000u; x.x |= x2.x & x.x & 0u;
        }
#endif
        // = GBN_REC_HI / GBN_REC_IDX16 of record dl * 32 + p * 4 (blocks of 64 records: 64 hi words, 64 indices)
        const size_t blk = (size_t)(dl >> 1) * 96, in = (size_t)((dl & 1u) * 32u + p * 4u);
        if (!(GBN_BIN_ABL & 4)) *reinterpret_cast<uint4 *>(rec32 + blk + in) = h;
        if (!(GBN_BIN_ABL & 8)) *reinterpret_cast<uint2 *>(rec16 + (blk + 64) * 2 + in) = x;
    };

    if (tid < GBN_BIN_MAXNB) s_hist[tid] = 0;
#if GBN_BIN_DUP & 1
    if (tid < GBN_BIN_MAXNB) s_hist2[tid] = 0;
    uint32_t dup_acc = 0;
#endif
    // Tile of (writer w, round k) = k * writers + (w + k) mod writers: the rotation keeps tiles of one kind
    // (the short last tile of every subject, when the tiles per subject divide the grid) from always
    // landing on the same workgroups (GBN_TILE_OF in gbn_dev.h; the rare kernel inverts it).
    uint32_t rot = wid;                                          // (wid + seq) mod stride
    auto rot_next = [&](uint32_t r) -> uint32_t { return r + 1u == (uint32_t)stride ? 0u : r + 1u; };
    int64_t tile = blockIdx.x;
    if (tile > last) {
        for (int b = tid; b < nb; b += GBN_SORT_THREADS) { B.gcount[(size_t)b * B.nwriters + blockIdx.x] = 0; if (B.gtotal) B.gtotal[(size_t)b * B.nwriters + blockIdx.x] = 0; }
        return;
    }
    // owner thread of bin `tid`: its stream's state lives in registers
    uint32_t wpos = 0, cc = 0;                                  // records stored so far (multiple of LINE), records in the open line (< LINE)
    const uint32_t sline0 = (uint32_t)(GBN_RECIDX(B, (tid < nb ? tid : 0), wid, 0) >> 5);    // first line of the stream (capacities are multiples of 512)
    const uint32_t mycap = GBN_BINCAP(B, (tid < nb ? tid : 0));  // room in this owner's stream
    const uint32_t open0 = (uint32_t)(STAGE + tid * LINE);      // the bin's open line
    uint32_t *const tcur = B.tcur + ((size_t)(tid < nb ? tid : 0) * B.nwriters + wid) * B.nseq;

    GbnTile T = uniform(P.tiles[tile]);
    GbnTile T1 = uniform(P.tiles[min(stride + (int64_t)rot_next(rot), last)]);
    uint32_t bin[PER], hi[PER];
#if GBN_BIN_PRESENCE
    // Experiment of round 6 (profiles/r06_presence_binning.txt): the batch's presence bits (one per cell, 2 MB for lut 12) are
    // looked up for every scan position and only positions whose cell is occupied become records -- 55 % fewer records written
    // here and read by the probe kernel for a 5 Mb batch, at the price of one scattered 4-byte load per position.  The records
    // depend on the batch then: passes that bin for themselves only (record cache off, nothing binned ahead).
    uint32_t pw[PER];
    auto presence_ask = [&]() {
        #pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t cell = (bin[k] << cbits) | (hi[k] & lowmask & 0x7fffu);
            pw[k] = P.pv ? P.pv[cell >> 5] >> (cell & 31u) : 1u;      // (no batch yet -- gbn_db_prepare_records: every position)
        }
    };
#endif
    {
        Raw r0; fetch(T, r0);
        keys_all(T, r0, bin, hi);
#if GBN_BIN_PRESENCE
        presence_ask();
#endif
    }
    int32_t stay[PER];                                          // slot of a record that waits for its open line to be stored, else -1
    uint32_t keep_hi[PER];
    #pragma unroll
    for (int k = 0; k < PER; k++) { stay[k] = -1; keep_hi[k] = 0; }
    __syncthreads();

#if GBN_BIN_TIMING   // phase timer of workgroup 0 (tools/build_variant.sh t "-DGBN_BIN_TIMING=1", GBN_DBG=32)
    const bool timed = blockIdx.x == 0 && tid == 0;
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define GBN_LAP1(ph) do { if (timed) { const unsigned long long t_ = __builtin_readcyclecounter(); tph[ph] += t_ - tprev; tprev = t_; } } while (0)
#else
#define GBN_LAP1(ph) do { } while (0)
#endif
    uint32_t ntask = 0;                                         // quarter lines of the tile before (for late_stores)
    auto late_stores = [&]() {
        const uint32_t i = (uint32_t)tid + GBN_SORT_THREADS;
        if (i < ntask && !(GBN_BIN_ABL & 2)) {
            const uint2 d = s_line[i / LP];
            if (d.y != 0xffffffffu) store_part(d.x, i % LP, d.y + (i % LP) * 4);
        }
    };
    uint32_t seq = 0;
    for (; tile <= last; ++seq, rot = rot_next(rot), tile = (int64_t)seq * stride + rot) {
        // ---- [0] rank of every record inside its bin; the bytes of the next tile ----
        uint32_t rank[PER]; bool valid[PER];
        #pragma unroll
        for (int k = 0; k < PER; k++) valid[k] = idx_of(k) < (uint32_t)T.npos;
#if GBN_BIN_PRESENCE == 1
        #pragma unroll
        for (int k = 0; k < PER; k++) valid[k] = valid[k] && (pw[k] & 1u);
#elif GBN_BIN_PRESENCE == 2       // the lookups alone: every position still becomes a record
        #pragma unroll
        for (int k = 0; k < PER; k++) valid[k] = valid[k] && ((pw[k] | 1u) & 1u) && (pw[k] != 0x9e3779b9u || tid != 1023);
#endif
        #pragma unroll
        for (int k = 0; k < PER; k++) {     // (positions past the end of a partial tile all carry the same key: they must not touch the histogram)
            rank[k] = 0;
#if GBN_BIN_ABL & 16
            if (valid[k]) { __hip_atomic_fetch_add(&s_hist[bin[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); rank[k] = (uint32_t)k; }
#else
            if (valid[k]) rank[k] = atomicAdd(&s_hist[bin[k]], 1u);
#endif
        }
#if GBN_BIN_DUP & 1
        {
            uint32_t r2[PER];
            #pragma unroll
            for (int k = 0; k < PER; k++) { r2[k] = 0; if (valid[k]) r2[k] = atomicAdd(&s_hist2[bin[k]], 1u); }
            #pragma unroll
            for (int k = 0; k < PER; k++) dup_acc += r2[k];
        }
#endif
        Raw R;
        if constexpr (STEP > 0) fetch(T1, R);
        GbnTile T2 = P.tiles[min((int64_t)(seq + 2) * stride + (int64_t)rot_next(rot_next(rot)), last)];
        late_stores();                                          // second quarter-line round of the previous tile
        GBN_LAP1(0);
        __syncthreads();                                        // (A) histogram complete
        GBN_LAP1(1);
        // ---- [1] + [2] owner threads: complete lines, staging offsets (both sums in one word: records
        // that go to the staging area < 2^14, complete lines < 2^10), descriptors ----
        uint32_t v = 0, incl = 0, my_nl = 0, my_cc = 0, tot = 0;
        if (tid < GBN_BIN_MAXNB) {
            if (tid < nb) {
                tot = cc + s_hist[tid];
                my_nl = tot / LINE; my_cc = tot & (LINE - 1);
                // staging: the complete lines after the bin's first one
                v = (my_nl > 1 ? (my_nl - 1) * LINE : 0u) | (my_nl << 16);
            }
            incl = wave_scan_incl(v);
            if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
        }
        GBN_LAP1(2);
        __syncthreads();                                        // (B0) wave totals
        GBN_LAP1(3);
        if (tid < nb) {
            uint32_t run = incl - v;
            #pragma unroll
            for (int w = 0; w < GBN_BIN_MAXNB / 64 - 1; w++) run += (tid >> 6) > w ? s_wtot[w] : 0u;
            const uint32_t off = run & 0xffffu, l0 = run >> 16;
            const uint32_t first_past = my_nl ? my_nl * LINE - cc : 0xffffu;
            s_pk[tid] = make_uint2((open0 + cc) | ((off + cc - LINE) << 16), first_past | ((LINE - cc) << 16));
            if ((seq & 7u) == 0) tcur[seq >> 3] = wpos + cc;    // stream index of this tile's first record
            if (wpos + my_nl * LINE > mycap) atomicOr(B.overflow, 1u);
            for (uint32_t l = 0; l < my_nl; l++) {
                const bool fits = wpos + (l + 1) * LINE <= mycap;
                s_line[l0 + l] = make_uint2(sline0 + (wpos >> 5) + l, fits ? (l == 0 ? open0 : off + (l - 1) * LINE) : 0xffffffffu);
            }
            if (tid == nb - 1) s_nlines = l0 + my_nl;
            wpos += my_nl * LINE; cc = my_cc;
            s_hist[tid] = 0;                                    // for the next tile: its atomics come after (C)
#if GBN_BIN_DUP & 1
            s_hist2[tid] = 0;
#endif
        }
        GBN_LAP1(4);
        __syncthreads();                                        // (B) descriptors known
        GBN_LAP1(5);
        // ---- [3] scatter ----
        {
            uint2 pk[PER];
            #pragma unroll
            for (int k = 0; k < PER; k++) pk[k] = s_pk[bin[k]];
#if GBN_BIN_DUP & 2
            #pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint2 q2 = s_pk[bin[k] ^ 1u];            // (the neighbouring bin's descriptor: a second random 8-byte read)
                pk[k].y |= q2.y & 0x80000000u & pk[k].y;
            }
#endif
            // the records of the previous tile that waited: their open lines were stored in that tile's [4]
            #pragma unroll
            for (int k = 0; k < PER; k++)
                if (stay[k] >= 0) { s_hi[stay[k]] = keep_hi[k]; s_ix[stay[k]] = (uint16_t)(idx_of(k) | (((seq - 1u) & 7u) << 13)); }
            #pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint32_t r = rank[k];
                const uint32_t open_at = pk[k].x & 0xffffu, room = pk[k].y >> 16, past = pk[k].y & 0xffffu;
                const int32_t stage_at = (int32_t)pk[k].x >> 16;
                // the bin's open line first, then the staging area; what lies past the last complete line
                // waits in registers until the open line has been stored
                const uint32_t slot = (r < room ? open_at : (uint32_t)stage_at) + r;
                const bool waits = valid[k] && r >= past;
                stay[k] = waits ? (int32_t)((uint32_t)STAGE + bin[k] * LINE + (r - past)) : -1;
                keep_hi[k] = hi[k];
                if (valid[k] && !waits) {
                    s_hi[slot] = hi[k]; s_ix[slot] = (uint16_t)(idx_of(k) | ((seq & 7u) << 13));
#if GBN_BIN_DUP & 4
                    *reinterpret_cast<volatile uint32_t *>(&s_hi[slot]) = hi[k]; *reinterpret_cast<volatile uint16_t *>(&s_ix[slot]) = (uint16_t)(idx_of(k) | ((seq & 7u) << 13));
#endif
                }
            }
        }
        GBN_LAP1(6);
        __syncthreads();                                        // (C) open lines and staging filled
        // ---- [4] keys of t+1 before the stores: the wait for the loads of t+1 counts every outstanding
        // memory operation and would otherwise sit behind this tile's stores ----
        T = T1; T1 = uniform(T2);
        if constexpr (STEP == 0) fetch(T, R);
        keys_all(T, R, bin, hi);
#if GBN_BIN_PRESENCE
        presence_ask();
#endif
        // Stores of the complete lines, a quarter line per thread and step.  The first 1024 quarter lines leave
        // here; the next 1024 wait until [0] of the next tile (behind its loads, next to its atomics: spreading
        // the stores over the tile keeps the store queue from stalling every wave at once); the rare rest here.
        ntask = s_nlines * (uint32_t)LP;
        if (!(GBN_BIN_ABL & 2))
        for (uint32_t i = tid; i < ntask; i += (i == (uint32_t)tid ? 2u : 1u) * GBN_SORT_THREADS) {
            const uint2 d = s_line[i / LP];
            if (d.y == 0xffffffffu) continue;
            store_part(d.x, i % LP, d.y + (i % LP) * 4);
        }
        // (no barrier here: the open lines just read are next written in [3] of the next tile, after (A)..(B))
    }
#if GBN_BIN_TIMING
    if (timed) for (int i = 0; i < 8; i++) B.rare_counts[512 + i] = (uint32_t)(tph[i] >> 4);
    if (tid == 0 && blockIdx.x < 512) {     // wall clock (100 MHz) of every workgroup: start, duration
        B.rare_counts[1024 + blockIdx.x] = (uint32_t)wg_t0;
        B.rare_counts[1536 + blockIdx.x] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - wg_t0);
    }
#endif
    late_stores();
#if GBN_BIN_DUP & 1
    if (dup_acc == 0xfffffffeu) B.gcount[0] = dup_acc;        // (keeps the shadow atomics' results alive)
#endif
    __syncthreads();
    // the records of the last tile that waited
    #pragma unroll
    for (int k = 0; k < PER; k++)
        if (stay[k] >= 0) { s_hi[stay[k]] = keep_hi[k]; s_ix[stay[k]] = (uint16_t)(idx_of(k) | (((seq - 1u) & 7u) << 13)); }
    uint2 *const s_fin = s_pk;                                  // at the end: records stored, records in the open line
    if (tid < nb) s_fin[tid] = make_uint2(wpos, cc);
    __syncthreads();
    // the last, incomplete line of every stream: padded with flagged records
    for (uint32_t i = tid; i < (uint32_t)nb * LINE; i += GBN_SORT_THREADS) {
        const uint32_t b = i / LINE, sl = i % LINE;
        if (sl >= s_fin[b].y) { s_hi[STAGE + b * LINE + sl] = GBN_REC_PAD(cbits, b); s_ix[STAGE + b * LINE + sl] = 0xffffu; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)nb * LP; i += GBN_SORT_THREADS) {
        const uint32_t b = i / LP, p = i % LP;
        const uint2 f = s_fin[b];
        if (f.y && f.x + LINE <= GBN_BINCAP(B, b) && !(GBN_BIN_ABL & 2))
            store_part((uint32_t)((GBN_RECIDX(B, b, wid, 0) + f.x) >> 5), p, STAGE + b * LINE + p * 4);
    }
    for (int b = tid; b < nb; b += GBN_SORT_THREADS) {
        const uint2 f = s_fin[b];
        const uint32_t total = f.x + (f.y ? LINE : 0u), capb = GBN_BINCAP(B, b);
        if (total > capb) atomicOr(B.overflow, 1u);
        B.gcount[(size_t)b * B.nwriters + blockIdx.x] = min(total, capb);
        if (B.gtotal) B.gtotal[(size_t)b * B.nwriters + blockIdx.x] = total;      // (counted on past the stream's end)
    }
}

#else

// ---------------------------------------------------------------------------------------------------
// Binning kernel, round 4 (-DGBN_BIN_V1=0): the same line-exact output (only complete, aligned 128 + 64-byte pieces are ever stored),
// without owner threads, descriptors, prefix sums or a staging area -- two barriers per tile, and per record one
// returning LDS atomic and two LDS writes.
//
// Where the time of the four-barrier form went (profiles/r04_bin_*): per tile and CU the LDS pipe was busy 6,100 cycles
// (62 % of them bank conflicts of the random accesses: rank atomic, descriptor read, two scatter writes, the waiting
// records' writes, the store reads), the four SIMDs 5,900 cycles with 369 VALU instructions per wave, and the tile took
// 12,300: the sum -- LDS-bound and VALU-bound stretches alternate between the barriers, nothing overlaps.  A form with
// the same steps spread over two tiles in flight and two barriers (rank | owners + keys + stores, scatter of the tile
// before next to the rank) executed as many instructions and was 5 % slower.  So: fewer LDS operations and fewer
// instructions per record.
//
// The LDS holds 512 lines of 32 records, dealt evenly to the bins as rings (512 bins: one line each; 128 bins: four),
// and per bin ONE word {head line of the ring, records since the head line's start}.  A record's atomicAdd on that
// word returns its place: inside the ring it is written at once (phase X); past the ring's end it waits in registers for
// one tile -- the lines in front of it are complete by then, and phase Y, which stores every complete line and moves
// the bin's word on, has made room (a record more than one ring past the end waits through extra store rounds; with
// random subjects once in 10^4 bin-tiles).  Phase Y needs no list of lines: 4,096 (line, quarter) tasks, four per
// thread, look at their bin's word; the one lane per bin that owns quarter 0 of ring line 0 writes the bin's next word
// into the OTHER of two arrays (phase X of the next tile uses that one), so no task ever reads a word another lane has
// already moved on.
//   phase X(t):  the waiting records of tile t-1 -> LDS | rank + scatter of tile t
//   (1)
//   phase Y(t):  keys of tile t+1 | stores of the complete lines, the bins' words moved on, every 8th tile the stream
//                cursors | loads of tile t+2
//   (2)
//
// Record layout (changed in round 4; the probe kernel reads it): hi word = [30:24] the 7 subject bits right of the
// lookup word, [23:16] the 4 bases left of it, [15:0] the cell inside the bin -- for tables of 2^15 cells per bin bit 15
// is the lowest bit of the BIN number (it comes with the byte and costs an instruction to clear; the probe kernel knows
// its bin: GBN_REC_PAR), otherwise 0; bit 31 is not defined.  For lut 12 the word, the four bases in front and the seven
// bits behind are bytes 0-2 of one 32-bit window and the top byte of another: two funnel shifts, one bit-field extract
// (the bin) and one byte permute per scan position.
// (Streams of one capacity only: the per-bin capacities of round 6 -- GbnBinParams::bincap -- are the four-barrier form's.)
template <int STEP, int LUT>
__device__ __forceinline__ void scan_bin3_body(const GbnBinParams &B)
{
    const GbnScanParams &P = B.S;
    constexpr int TILE = GBN_BIN_TILE_POS, PER = TILE / GBN_SORT_THREADS, LINE = GBN_OPEN_LINE, LP = LINE / 4;    // LP lanes store one line
    constexpr int NLINES = GBN_BIN_MAXNB;                       // lines of 32 records in LDS, shared out to the bins
    static_assert(PER == 8 && GBN_SORT_THREADS == 1024 && LINE == 32 && NLINES == 512, "8192-position tiles, 1024 threads, 512 lines of 32 records");
    // 512 bins (lut 12): a ring of one line holds 3/4 of a tile's records at once, the rest wait a tile in registers and
    // cost a second round of LDS writes; the LDS has room for rings of 48 slots (a line and a half: three granules of 16,
    // a line = two of them), with which fewer than 1 % wait.  Then the word's upper half is the head line's first SLOT
    // (0, 16 or 32), and positions wrap at 48.
    constexpr bool R48 = (LUT == 12);
    constexpr int RS48 = 48, NSLOT = R48 ? GBN_BIN_MAXNB * RS48 : NLINES * LINE;
    __shared__ __attribute__((aligned(16))) uint32_t s_hi[NSLOT];
    __shared__ __attribute__((aligned(16))) uint16_t s_ix[NSLOT];
    __shared__ uint32_t s_word[2][GBN_BIN_MAXNB];               // per bin: [31:16] head line of its ring (R48: its first slot), [15:0] records since that line's start
    __shared__ uint32_t s_wl[2][GBN_BIN_MAXNB];                 // per bin: lines stored so far
    __shared__ uint32_t s_slow[2];                              // some record waits for more than one store round
    const int tid = threadIdx.x;
    const int lut = LUT > 0 ? LUT : P.lut;
    const uint32_t mask = LUT > 0 ? (uint32_t)((1ull << (2 * LUT)) - 1) : (uint32_t)(P.ncells - 1);
    const int cbits = LUT > 0 ? GBN_BIN_CBITS(LUT) : B.cbits;
    const int nb = LUT > 0 ? (int)(((int64_t)1 << (2 * LUT)) >> GBN_BIN_CBITS(LUT)) : B.nb;
    const uint32_t lowmask = cbits == 15 ? 0xffffu : (1u << cbits) - 1;     // (15 bits per bin: the bin's lowest bit rides along)
    const int cshift = 56 - 2 * lut, rshift = 49 - 2 * lut;
    const uint32_t ustep = (uint32_t)P.step;
    const int64_t stride = gridDim.x, last = P.ntiles - 1;
    const uint32_t wid = blockIdx.x;

    // a lane owns PER consecutive positions = 16 * STEP bits of subject: a whole number of dwords for
    // even strides, half a dword extra for odd lanes of odd strides
    constexpr int NDW = STEP > 0 ? ((2 * STEP * (PER - 1) - 8 + 38) >> 5) + 4 : 2 * PER;
    struct Raw { uint32_t d[NDW]; };
    auto idx_of = [&](int k) -> uint32_t { return STEP > 0 ? (uint32_t)(tid * PER + k) : (uint32_t)(tid + k * GBN_SORT_THREADS); };
    auto upos_of = [&](const GbnTile &t, int k) -> uint32_t {
        const uint32_t i = min(idx_of(k), (uint32_t)t.npos - 1u);
        return (uint32_t)t.first_pos + i * ustep + 60u;
    };
    auto lane_half = [&](const GbnTile &t) -> uint32_t {    // lane's first base, in units of 8 bases (16 bits), from the tile start
        return min((uint32_t)tid, ((uint32_t)t.npos - 1u) / PER) * (uint32_t)STEP;
    };
    auto fetch = [&](const GbnTile &t, Raw &r) {
        if (GBN_BIN_ABL & 64) { for (int i = 0; i < NDW; i++) r.d[i] = (uint32_t)tid * 2654435761u + (uint32_t)(i * 40503 + t.first_pos); return; }
        if constexpr (STEP > 0) {
            const uint8_t *p = P.db + ((size_t)(uint32_t)t.off16 << 4) + 4 * ((size_t)((uint32_t)t.first_pos >> 4) + (size_t)(lane_half(t) >> 1)) - 4;
            #pragma unroll
            for (int i = 0; i + 4 <= NDW; i += 4) __builtin_memcpy(&r.d[i], p + 4 * i, 16);
            if constexpr (NDW % 4 == 3) { __builtin_memcpy(&r.d[NDW - 3], p + 4 * (NDW - 3), 12); }
            else if constexpr (NDW % 4 == 2) { __builtin_memcpy(&r.d[NDW - 2], p + 4 * (NDW - 2), 8); }
            else if constexpr (NDW % 4 == 1) { __builtin_memcpy(&r.d[NDW - 1], p + 4 * (NDW - 1), 4); }
        } else {
            #pragma unroll
            for (int k = 0; k < PER; k++)
                __builtin_memcpy(&r.d[2 * k], P.db + ((size_t)(uint32_t)t.off16 << 4) - 16 + (upos_of(t, k) >> 2), 8);
        }
    };
    auto keys_all = [&](const GbnTile &t, const Raw &r, uint32_t (&bin)[PER], uint32_t (&hi)[PER]) {
        uint32_t x[NDW];
        if constexpr (STEP > 0) {
            // big-endian dwords; odd lanes of odd strides start half a dword later: one byte permute does both
            const uint32_t sel = ((STEP & 1) && (lane_half(t) & 1u)) ? 0x06070001u : 0x04050607u;
            #pragma unroll
            for (int i = 0; i + 1 < NDW; i++) x[i] = __builtin_amdgcn_perm(r.d[i], r.d[i + 1], sel);
            x[NDW - 1] = bswap32(r.d[NDW - 1]);
        }
        #pragma unroll
        for (int k = 0; k < PER; k++) {
            if constexpr (STEP > 0 && LUT == 12) {
                // window from 4 bases in front of the word: [31:24] those bases, [23:0] the word; and the 32 bits
                // from its last bit on: [30:24] the 7 bits behind the word
                const int bit = 2 * STEP * k - 8 + 32, a = bit >> 5, o = bit & 31;
                const int bit2 = bit + 31, a2 = bit2 >> 5, o2 = bit2 & 31;
                const uint32_t w0 = o ? __builtin_amdgcn_alignbit(x[a], x[a + 1], 32 - o) : x[a];
                const uint32_t w1 = o2 ? __builtin_amdgcn_alignbit(x[a2], x[a2 + 1 < NDW ? a2 + 1 : NDW - 1], 32 - o2) : x[a2];
                bin[k] = (w0 >> 15) & 0x1ffu;
                hi[k] = (__builtin_amdgcn_perm(w1, w0, 0x07030100u) & 0x7fffffffu) | ((w1 << 8) & 0x80000000u);     // (bit 31: the EIGHTH subject bit behind the word, w1's bit 23 -- round 5)
            } else {
                uint64_t w;
                if constexpr (STEP > 0) {
                    const int bit = 2 * STEP * k - 8 + 32, a = bit >> 5, o = bit & 31;
                    const uint32_t x2 = x[a + 2 < NDW ? a + 2 : NDW - 1];
                    const uint32_t hi32 = o ? ((x[a] << o) | (x[a + 1] >> (32 - o))) : x[a];
                    const uint32_t lo32 = o ? ((x[a + 1] << o) | (x2 >> (32 - o))) : x[a + 1];
                    w = ((uint64_t)hi32 << 32) | lo32;
                } else {
                    uint64_t raw; __builtin_memcpy(&raw, &r.d[2 * k], 8);
                    w = __builtin_bswap64(raw) << (2 * (upos_of(t, k) & 3));
                }
                const uint32_t c = (uint32_t)(w >> cshift) & mask;
                bin[k] = c >> cbits;
                hi[k] = (c & lowmask) | ((uint32_t)(w >> 56) << 16) | (((uint32_t)(w >> rshift) & 0x7fu) << 24) | (((uint32_t)(w >> (rshift - 1)) & 1u) << 31);
            }
        }
    };
    auto uniform = [](GbnTile t) -> GbnTile {
        t.subj = __builtin_amdgcn_readfirstlane(t.subj); t.first_pos = __builtin_amdgcn_readfirstlane(t.first_pos);
        t.npos = __builtin_amdgcn_readfirstlane(t.npos); t.off16 = __builtin_amdgcn_readfirstlane(t.off16);
        return t;
    };
    // an eighth of a line (4 records: 16 bytes of hi words, 8 bytes of indices) from LDS slot `src` to stream line
    // `dl`: the 8 lanes of a line write 128 aligned bytes of hi words and 64 of indices -- scattered writes cost
    // by the piece below 128 bytes (tools/write_microbench.hip: 64 + 32 byte pieces 3.5 TB/s, 128 + 64: 6+)
    uint32_t *const rec32 = B.rec; uint16_t *const rec16 = reinterpret_cast<uint16_t *>(B.rec);
    auto store_part = [&](uint32_t dl, uint32_t p, uint32_t src) {
        const uint4 h = *reinterpret_cast<const uint4 *>(&s_hi[src]);
        const uint2 x = *reinterpret_cast<const uint2 *>(&s_ix[src]);
        // = GBN_REC_HI / GBN_REC_IDX16 of record dl * 32 + p * 4 (blocks of 64 records: 64 hi words, 64 indices)
        const size_t blk = (size_t)(dl >> 1) * 96, in = (size_t)((dl & 1u) * 32u + p * 4u);
        if (!(GBN_BIN_ABL & 4)) *reinterpret_cast<uint4 *>(rec32 + blk + in) = h;
        if (!(GBN_BIN_ABL & 8)) *reinterpret_cast<uint2 *>(rec16 + (blk + 64) * 2 + in) = x;
    };
    // ring of a bin: 2^lr lines = 2^(lr + 5) slots from slot bin << (lr + 5)
    const int lr = LUT > 0 ? (GBN_BIN_CBITS(LUT) + 9 - 2 * LUT) : (9 - (31 - __builtin_clz((uint32_t)nb)));
    const int ls = lr + 5;
    const uint32_t rs = 1u << ls, rmask = rs - 1u, lmask = (1u << lr) - 1u;

    if (tid < GBN_BIN_MAXNB) { s_word[0][tid] = 0; s_wl[0][tid] = 0; s_word[1][tid] = 0; s_wl[1][tid] = 0; }
    if (tid < 2) s_slow[tid] = 0;
    // Tile of (writer w, round k) = k * writers + (w + k) mod writers (GBN_TILE_OF in gbn_dev.h; the rare kernel inverts it)
    uint32_t rot = wid;                                          // (wid + seq) mod stride
    auto rot_next = [&](uint32_t r) -> uint32_t { return r + 1u == (uint32_t)stride ? 0u : r + 1u; };
    if ((int64_t)blockIdx.x > last) {
        for (int b = tid; b < nb; b += GBN_SORT_THREADS) B.gcount[(size_t)b * B.nwriters + blockIdx.x] = 0;
        return;
    }
    const uint32_t lines_per_stream = B.subcap >> 5;            // (subcap is a multiple of 512)
    for (int b = tid; b < nb; b += GBN_SORT_THREADS) B.tcur[((size_t)b * B.nwriters + wid) * B.nseq] = 0;      // cursor of tile 0

    // tiles: C = tile `seq` (keys known), N = seq + 1 (keys in phase Y), F = seq + 2 (loads issued in phase Y)
    GbnTile TC = uniform(P.tiles[blockIdx.x]);
    const int64_t tile_n = stride + (int64_t)rot_next(rot);
    bool haveN = tile_n <= last;
    GbnTile TN = uniform(P.tiles[min(tile_n, last)]);
    GbnTile TFraw = P.tiles[min((int64_t)2 * stride + (int64_t)rot_next(rot_next(rot)), last)];   // descriptor of tile seq + 2, a tile ahead of its use
    uint32_t binC[PER], hiC[PER];
    Raw R;
    {
        Raw r0; fetch(TC, r0);
        keys_all(TC, r0, binC, hiC);
        if (haveN) fetch(TN, R);
    }
    uint32_t nposC = (uint32_t)TC.npos;
    // a record that waits: [15:0] its slot, [31:16] the store rounds it waits for (0: none waits); its hi word
    uint32_t stay[PER], keep_hi[PER];
    #pragma unroll
    for (int k = 0; k < PER; k++) { stay[k] = 0; keep_hi[k] = 0; }
    __syncthreads();

#if GBN_BIN_TIMING   // phase timer of workgroup 0 (tools/build_variant.sh t "-DGBN_BIN_TIMING=1", GBN_DBG=32)
    const bool timed = blockIdx.x == 0 && (tid == 0 || tid == 1023);
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define GBN_LAP2(ph) do { if (timed) { const unsigned long long t_ = __builtin_readcyclecounter(); tph[ph] += t_ - tprev; tprev = t_; } } while (0)
#else
#define GBN_LAP2(ph) do { } while (0)
#endif
    uint32_t gen = 0;                                           // store passes so far: s_word / s_wl [gen & 1] are current
    // every complete line -> its stream, every bin's word moved on (into the arrays of gen + 1); cursor_at >= 0: the
    // stream position of every bin is also written as cursor entry `cursor_at`
    auto store_pass = [&](int32_t cursor_at) {
        const uint32_t g = gen & 1u;
        #pragma unroll
        for (int j = 0; j < NLINES * LP / GBN_SORT_THREADS; j++) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * GBN_SORT_THREADS;
            const uint32_t gl = i / LP, p = i % LP;             // ring line in LDS (R48: bin), quarter
            const uint32_t b = R48 ? gl : gl >> lr, jl = R48 ? 0u : gl & lmask;
            const uint32_t w = s_word[g][b], wl = s_wl[g][b];
            const uint32_t h = w >> 16, tot = w & 0xffffu;
            const uint32_t nl = R48 ? (tot >= (uint32_t)LINE ? 1u : 0u) : min(tot >> 5, 1u << lr);        // complete lines in the ring
            const uint32_t rl = R48 ? 0u : (jl - h) & lmask;    // this line's distance from the head line
            if (rl < nl && (int)b < nb) {
                if (wl + rl < lines_per_stream) {
                    const uint32_t pos = h + p * 4;             // (R48: the quarter's first slot, counted from the ring's start)
                    const uint32_t src = R48 ? b * RS48 + min(pos, pos - RS48) : gl * LINE + p * 4;
                    if (!(GBN_BIN_ABL & 2)) store_part((wid * (uint32_t)nb + b) * lines_per_stream + wl + rl, p, src);
                } else if (p == 0) atomicOr(B.overflow, 1u);
            }
            if (p == 0 && jl == 0 && (int)b < nb) {
                if constexpr (R48) s_word[g ^ 1u][b] = ((nl ? min(h + LINE, h + LINE - RS48) : h) << 16) | (tot - nl * LINE);
                else
                s_word[g ^ 1u][b] = (((h + nl) & lmask) << 16) | (tot - nl * LINE);
                s_wl[g ^ 1u][b] = wl + nl;
                if (cursor_at >= 0) B.tcur[((size_t)b * B.nwriters + wid) * B.nseq + (uint32_t)cursor_at] = wl * LINE + tot;
            }
        }
        if (tid == 0) s_slow[g ^ 1u] = 0;                       // (the flag of the pass before: read by everybody before the last barrier, set again after the next one at the earliest)
        ++gen;
    };
    // the records that wait: one round less to wait for; those whose turn it is go to their slots
    auto settle = [&](uint32_t seqbits, bool flag_deep) {
        #pragma unroll
        for (int k = 0; k < PER; k++)
            if (stay[k] >= 0x10000u) {
                if (stay[k] < 0x20000u) {
                    const uint32_t slot = stay[k] & 0xffffu;
                    s_hi[slot] = keep_hi[k]; s_ix[slot] = (uint16_t)(idx_of(k) | seqbits);
                    stay[k] = 0;
                } else {
                    stay[k] -= 0x10000u;
                    if (flag_deep && stay[k] >= 0x20000u) s_slow[gen & 1u] = 1;
                }
            }
    };
    bool haveC = true;
    uint32_t seq = 0;                                           // sequence number of tile C
    for (;;) {
        // =================== phase X ===================
        settle(((seq - 1u) & 7u) << 13, false);                 // (every record that waits has one round to go here)
        GBN_LAP2(0);
        {
            const uint32_t seqbits = (seq & 7u) << 13;
            uint32_t *const W = s_word[gen & 1u];
            auto scatter = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
                uint32_t raw[PER];
                #pragma unroll
                for (int k = 0; k < PER; k++) {     // (positions past the end of a partial tile all carry the same key: they must not touch the counters)
                    raw[k] = 0xffff0000u;
                    if (GBN_BIN_ABL & 32) raw[k] = (uint32_t)(tid & 15);
                    else if (GBN_BIN_ABL & 256) { __hip_atomic_fetch_add(&W[binC[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); raw[k] = (uint32_t)(tid & 15); }
                    else if (FULL || idx_of(k) < nposC) { raw[k] = atomicAdd(&W[binC[k]], 1u); if (GBN_BIN_ABL & 128) raw[k] &= 31u; }
                }
                #pragma unroll
                for (int k = 0; k < PER; k++) {
                    const bool valid = FULL || idx_of(k) < nposC;
                    const uint32_t rel = raw[k] & 0xffffu;
                    // slot in the bin's ring; rounds of stores to wait for (0: the slot is free now)
                    uint32_t slot, d;
                    if constexpr (R48) {
                        const uint32_t pos = (raw[k] >> 16) + rel;          // < 96 while the record does not wait
                        slot = binC[k] * RS48 + min(pos, pos - RS48);
                        d = (uint32_t)max(((int32_t)rel - 16) >> 5, 0);     // a store round frees 32 slots
                        if (d) slot = binC[k] * RS48 + pos % RS48;
                    } else {
                        slot = (binC[k] << ls) + ((((raw[k] >> 16) << 5) + rel) & rmask);
                        d = rel >> ls;
                    }
                    if (valid) {
                        if (d == 0) { if (!(GBN_BIN_ABL & 16)) { s_hi[slot] = hiC[k]; s_ix[slot] = (uint16_t)(idx_of(k) | seqbits); } else keep_hi[k] ^= hiC[k] + slot; }
                        else {
                            stay[k] = slot | (d << 16); keep_hi[k] = hiC[k];
                            if (d >= 2) s_slow[gen & 1u] = 1;
                        }
                    }
                }
            };
            if (nposC == (uint32_t)TILE) scatter(std::true_type{}); else scatter(std::false_type{});
        }
        GBN_LAP2(1);
        __syncthreads();                                        // (1) every record of tile C placed (or waiting), the bins' counts complete
        GBN_LAP2(2);
        // =================== phase Y ===================
        bool deep = s_slow[gen & 1u] != 0;
        // ---- keys of tile N (its loads were issued a phase Y ago) ----
        uint32_t binN[PER], hiN[PER];
        if (haveN) keys_all(TN, R, binN, hiN);
        GBN_LAP2(3);
        // ---- stores; the cursor of tile seq + 1 if it is an eighth one ----
        store_pass((haveN && ((seq + 1u) & 7u) == 0) ? (int32_t)((seq + 1u) >> 3) : -1);
        GBN_LAP2(4);
        // ---- loads of tile F = seq + 2 ----
        const uint32_t rotF = rot_next(rot_next(rot));
        const int64_t tile_f = (int64_t)(seq + 2) * stride + (int64_t)rotF;
        const bool haveF = tile_f <= last;
        GbnTile TF = uniform(TFraw);                            // (asked for a tile ago)
        TFraw = P.tiles[min((int64_t)(seq + 3) * stride + (int64_t)rot_next(rotF), last)];
        if (haveF) { if constexpr (STEP > 0) fetch(TF, R); }
        GBN_LAP2(5);
        __syncthreads();                                        // (2) lines read, words moved on
        GBN_LAP2(6);
        // ---- records more than one ring past the end (repeats: a bin took more than a ring in one tile): further store rounds ----
        while (deep) {
            settle((seq & 7u) << 13, true);
            __syncthreads();
            deep = s_slow[gen & 1u] != 0;
            store_pass(-1);
            __syncthreads();
        }
        if (!haveN) break;
        // ---- rotate: C <- N <- F ----
        #pragma unroll
        for (int k = 0; k < PER; k++) { binC[k] = binN[k]; hiC[k] = hiN[k]; }
        nposC = (uint32_t)TN.npos;
        TN = TF; haveN = haveF;
        if constexpr (STEP == 0) { if (haveN) fetch(TN, R); }
        ++seq; rot = rot_next(rot);
    }
    (void)haveC;
#if GBN_BIN_TIMING
    if (timed) for (int i = 0; i < 12; i++) B.rare_counts[(tid == 0 ? 512 : 524) + i] = (uint32_t)(tph[i] >> 4);
    if (tid == 0 && blockIdx.x < 512) {     // wall clock (100 MHz) of every workgroup: start, duration
        B.rare_counts[1024 + blockIdx.x] = (uint32_t)wg_t0;
        B.rare_counts[1536 + blockIdx.x] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - wg_t0);
    }
#endif
    // the records of the last tile that waited, and the lines they complete
    settle((seq & 7u) << 13, false);
    __syncthreads();
    store_pass(-1);
    __syncthreads();
    // the last, incomplete line of every stream: padded with records that name a cell the probe kernel keeps empty
    // (just past the bin's cells, or just in front of them when bit 15 of the bin's records is set)
    const uint32_t g = gen & 1u;
    for (uint32_t i = tid; i < (uint32_t)nb * LINE; i += GBN_SORT_THREADS) {
        const uint32_t b = i / LINE, sl = i % LINE;
        const uint32_t w = s_word[g][b], pos = (w >> 16) + sl;
        const uint32_t at = R48 ? b * RS48 + min(pos, pos - RS48) : ((b << lr) + (w >> 16)) * LINE + sl;
        if (sl >= (w & 0xffffu)) { s_hi[at] = GBN_REC_PAD(cbits, b); s_ix[at] = 0xffffu; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)nb * LP; i += GBN_SORT_THREADS) {
        const uint32_t b = i / LP, p = i % LP;
        const uint32_t w = s_word[g][b], wl = s_wl[g][b];
        const uint32_t pos = (w >> 16) + p * 4;
        if ((w & 0xffffu) && wl < lines_per_stream && !(GBN_BIN_ABL & 2))
            store_part((wid * (uint32_t)nb + b) * lines_per_stream + wl, p, R48 ? b * RS48 + min(pos, pos - RS48) : ((b << lr) + (w >> 16)) * LINE + p * 4);
    }
    for (int b = tid; b < nb; b += GBN_SORT_THREADS) {
        const uint32_t w = s_word[g][b];
        const uint32_t total = s_wl[g][b] * LINE + ((w & 0xffffu) ? LINE : 0u);
        if (total > B.subcap) atomicOr(B.overflow, 1u);
        B.gcount[(size_t)b * B.nwriters + blockIdx.x] = min(total, B.subcap);
    }
}

#endif   // GBN_BIN_V1

#if GBN_BIN_V1
#define GBN_BIN_BODY scan_bin_line_body
#else
#define GBN_BIN_BODY scan_bin3_body
#endif
// stride- and width-specialised variants: megablast (word 28: lut 12 / 11 / 8) and blastn (word 11: lut 11 / 10 / 8);
// every other (stride, lut) pair takes the generic kernel
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel(GbnBinParams B) { GBN_BIN_BODY<0, 0>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s17(GbnBinParams B) { GBN_BIN_BODY<17, 12>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s18(GbnBinParams B) { GBN_BIN_BODY<18, 11>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s1(GbnBinParams B) { GBN_BIN_BODY<1, 11>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s2(GbnBinParams B) { GBN_BIN_BODY<2, 10>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s4(GbnBinParams B) { GBN_BIN_BODY<4, 8>(B); }
extern "C" __global__ void __launch_bounds__(GBN_SORT_THREADS, GBN_BIN_OCC) scan_bin_kernel_s21(GbnBinParams B) { GBN_BIN_BODY<21, 8>(B); }

namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// sixteen bytes from a dword-aligned address with ONE load instruction, waited for on the spot (every load of the wave with it)
__device__ __forceinline__ u32x4 load16_now(const void *p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// rare path of the probe kernel: full fingerprints, chain walk, exact verification
// Seeds go to P.seeds through one wave-aggregated reservation per loop round, or -- when the caller passes
// an LDS staging buffer (`s_buf`, `s_n`, capacity `cap`) -- are collected there first and flushed by the
// whole workgroup (dense-seed shapes: a single counter takes only ~90 reservations per microsecond).
// cw: the cell's direct-probe word (GbnScanParams::cellw: the full fingerprint of the cell's first entry, bit 31 = more
// entries), asked for by the caller next to its other loads: a cell with ONE entry whose fingerprint fails is done
// without its entry list -- one random HBM sector (cell_start) fewer for nearly every queued record, and none for `ent`
__device__ void probe_slow(const GbnScanParams &P, uint32_t posid, uint32_t cell, bool count_raw, uint32_t cw,
                           unsigned long long &raw, GbnDevSeed *s_buf = nullptr, uint32_t *s_n = nullptr, uint32_t cap = 0)
{
    // (the four words of a tile in ONE load: the rare kernel is bound by the number of scattered requests its records make)
    // (written as it is executed: the compiler cuts a 16-byte load of which three words are used into two loads)
    const u32x4 tw = load16_now(P.tiles + (posid >> GBN_BIN_TILE_BITS));
    GbnTile T; T.subj = (int32_t)tw.x; T.first_pos = (int32_t)tw.y; T.npos = (int32_t)tw.z; T.off16 = (int32_t)tw.w;
    const int32_t s = T.first_pos + (int32_t)(posid & (uint32_t)(GBN_BIN_TILE_POS - 1)) * P.step;
    const uint8_t *__restrict__ subj = P.db + ((size_t)(uint32_t)T.off16 << 4);     // = byte_off[T.subj], one load less
    // one 32-base window from s - 8 holds the 8 bases left of the word and (lut <= 16) at least 8 right of it
    const uint64_t w32 = (P.fl > 0 || P.fr > 0) ? bases32(subj, (int64_t)s - 8) : 0ull;     // lut == word: nothing to compare
    const uint32_t sl = (uint32_t)(w32 >> 48);
    const uint32_t sr = (uint32_t)((w32 << (2 * (8 + P.lut))) >> 32);
    if (!count_raw && !(cw >> 31) && !fp_pass(cw, sl, sr, P.fl, P.fr)) return;
    const uint32_t start = P.cell_start[cell], end = P.cell_start[cell + 1];
    if (count_raw) raw += end - start;
    for (uint32_t e = start; e < end; e++) {
        const unsigned long long ent = P.ent[e];
        if (!fp_pass((uint32_t)(ent >> 32), sl, sr, P.fl, P.fr)) continue;
        const int32_t slen = (P.mode == GBN_EXT_DIRECT) ? 0 : P.len[T.subj];
        const int32_t q = (int32_t)(ent & 0xffffffffu);
        const int el = verify_hit(P, subj, slen, q, s);
        // one reservation per wave and round: the lanes still in this loop that verified a hit
        const unsigned long long okm = __ballot(el >= 0);
        if (okm) {
            const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)okm) - 1;
            const uint32_t mine = (uint32_t)__popcll(okm & ((1ull << lane) - 1)), cnt = (uint32_t)__popcll(okm);
            GbnDevSeed sd; sd.subj = T.subj; sd.s_scan = s; sd.q_pos = q; sd.ext_left = el;
            // staging: the wave's seeds take slots [lbase, lbase + cnt); whatever falls past the capacity
            // goes straight to the global array, so the staged part never has holes
            uint32_t lbase = cap;
            if (s_buf) {
                if (lane == leader) lbase = atomicAdd(s_n, cnt);
                lbase = min(__shfl(lbase, leader), cap);
            }
            const uint32_t staged_n = min(cnt, cap - lbase);
            if (el >= 0 && mine < staged_n) s_buf[lbase + mine] = sd;
            if (staged_n < cnt) {
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(P.seed_count, (unsigned long long)(cnt - staged_n));
                base = __shfl(base, leader);
                if (el >= 0 && mine >= staged_n) { const unsigned long long o = base + (mine - staged_n); if (o < P.seed_cap) P.seeds[o] = sd; }
            }
        }
    }
}
}  // namespace

extern "C" __global__ void __launch_bounds__(GBN_BIN_THREADS)
probe_bin_kernel(GbnBinParams B)
{
    const GbnScanParams &P = B.S;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint32_t *s_tab = s_dyn;                                        // GBN_BIN_CELLS entries
    uint2 *s_q = reinterpret_cast<uint2 *>(s_dyn + GBN_BIN_TABW);    // [16 waves][QCAP]; a wave's queue is touched by that wave only
    uint16_t *s_side = reinterpret_cast<uint16_t *>(s_dyn + GBN_BIN_TABW + (GBN_BIN_THREADS / 64) * GBN_BIN_QCAP * 2);
    uint32_t *s_rcount = s_dyn + GBN_BIN_TABW + (GBN_BIN_THREADS / 64) * GBN_BIN_QCAP * 2 + GBN_BIN_SIDE / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the stream bookkeeping below stays in scalar registers
    const int grp = blockIdx.x & (GBN_BIN_GROUPS - 1);
    // tables of fewer slices than groups (2 or 4 bins): the groups that share a bin split its streams
    const int bstep = B.nb < GBN_BIN_GROUPS ? B.nb : GBN_BIN_GROUPS, sub = grp / bstep, nsub = GBN_BIN_GROUPS / bstep;
    const int wi = (int)(blockIdx.x >> 3) + sub * (int)(gridDim.x >> 3), nw = (int)(gridDim.x >> 3) * nsub;   // workgroup index among those on the bin
    const int cbits = B.cbits;
    const uint32_t ncell_bin = 1u << cbits;
    GbnRareItem *myq = B.rareq + (size_t)blockIdx.x * B.rare_seg;          // this workgroup's segment: no global atomics
    if (tid == 0) { *s_rcount = 0; s_tab[GBN_BIN_TAB0 - 1] = 0; s_tab[GBN_BIN_TAB0 + GBN_BIN_CELLS] = 0; }      // the empty cells pad records point at
    if ((B.dbg & 128) && tid == 0) B.rare_counts[1024 + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // GBN_DBG=128: the XCD (HW_REG_XCC_ID) this workgroup runs on
    // masks of the reduced fingerprint test; a zero mask makes that side "always matches"
    const uint32_t lmask = (B.rfl <= 0) ? 0u : ((1u << (2 * B.rfl)) - 1);                               // byte 0 of an fp15
    const uint32_t rmask = (B.rfrbits <= 0) ? 0u : (((1u << B.rfrbits) - 1) << (7 - B.rfrbits));      // byte 1
    const uint32_t m4 = (lmask | (rmask << 8)) * 0x10001u;          // both fingerprints of a cell word at once
    const bool fp16 = B.rfl >= 4 && B.rfrbits >= 7 && P.fr >= 4 && !(B.dbg & 256);       // the sixteenth fingerprint bit (flush; GBN_DBG=256: off, A/B): full reduced widths, at least 4 bases behind the word in the full fingerprint
    uint2 *q = s_q + wave * GBN_BIN_QCAP;
    int qn = 0;                                                     // wave-uniform
    unsigned long long raw = 0;
    unsigned long long rawu = 0;                                    // wave-uniform part of the lookup-hit count (GBN_PROBE_MASKS)
    int16_t opaque_zero;
    asm("s_mov_b32 %0, 0" : "=s"(opaque_zero));
    const unsigned long long lt = (1ull << lane) - 1;

    // Flush `cnt` queued items (one per lane): cells with a side list get their reduced
    // fingerprints checked here, densely; survivors go to the global rare-path queue.
    auto flush = [&](int first, int cnt, int bin) {
        const uint32_t *const tab = s_tab + GBN_BIN_TAB0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // queue slots written by other lanes of this wave
        bool keep = false; uint32_t at_rec = 0, cv = 0, hv_q = 0;
        if (lane < cnt) {
            // the queue holds the record's index inside the bin's region and its hi word (picked out of the lane's eight
            // registers by a select tree when it was queued: round 4 read the record again here, a scattered load and a wait of
            // its full latency per 64 queued records -- affordable now that the main loop's tests run on the scalar unit)
            const uint2 qe = q[first + lane];
            at_rec = qe.x;
            const uint32_t hv = qe.y; hv_q = hv;
            const uint32_t low = hv & 0x7fffu, sf = (hv >> 16) & 0x7fffu;
            cv = ((uint32_t)bin << cbits) | low;
            keep = true;
            const uint32_t t = tab[low];
            if ((t & 0x8000u) == 0) {                               // queued and not a one- or two-entry cell: three or more entries
                raw -= 1;                                           // (the main loop counted it as one lookup hit)
                const uint32_t n3 = (t >> 16) & 0x7fffu, so = t & 0x7fffu;
                if (n3 == 0) cv |= 0x80000000u;                     // always-rare cell: raw hits counted later
                else {
                    raw += n3; keep = false;
                    for (uint32_t e = 0; e < n3; e++) {
                        const uint32_t x = (uint32_t)s_side[so + e] ^ sf;
                        keep = keep || ((x & lmask) == 0) || (((x >> 8) & rmask) == 0);
                    }
                }
            }
        }
#if GBN_PROBE_FETCH
        // the cell's direct-probe word travels with the item (GbnRareItem): fetched here, by the few lanes that keep one,
        // underneath the streams of the other waves.  Round 5: it also decides whether the item travels at all.  The word holds
        // the FULL fingerprint of the cell's first entry; the record's hi word holds 16 subject bits around the lookup word -- the 4
        // bases in front and, since round 5, EIGHT bits behind (bit 31: the binning kernel's bit that was undefined).  The table
        // in LDS has room for 15 of them per entry, which lets 2^-8 + 2^-7 of the lookup hits through; for a cell with ONE entry
        // (55 % of the lookup hits) the sixteenth bit is tested here: a seed needs the 8 bases in front or the 7 behind to
        // match (fp_pass), so 4 in front or 4 behind is necessary -- a sixth fewer items for the rare kernel, each of which
        // costs it a scattered subject sector.
        uint32_t cw = 0;
        if (keep) {
            cw = P.cellw[cv & 0x7fffffffu];
            if (fp16 && !(cv >> 31) && !(cw >> 31) && !(cw & 1u)) {          // one entry, not forced, not an always-rare cell
                const uint32_t sl8 = (hv_q >> 16) & 0xffu, sr8 = (((hv_q >> 24) & 0x7fu) << 1) | (hv_q >> 31);
                const uint32_t el8 = (cw >> 15) & 0xffu, er8 = (cw >> 7) & 0xffu;
                keep = sl8 == el8 || sr8 == er8;
            }
        }
#else
        const uint32_t cw = 0;                                      // (fetched by the rare kernel)
#endif
        const unsigned long long m = __ballot(keep);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(s_rcount, (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (keep) {
                const uint32_t at = base + (uint32_t)__popcll(m & lt);
                if (at < B.rare_seg) {
#if GBN_PROBE_FETCH
                    // ... and the record's index: the second scattered sector per item
                    const uint32_t capb = GBN_BINCAP(B, bin), wr = at_rec / capb, jr = at_rec - wr * capb;
                    const uint32_t idx = reinterpret_cast<const uint16_t *>(B.rec)[GBN_REC_IDX16(GBN_RECIDX(B, bin, wr, jr))];
#else
                    const uint32_t idx = 0;
#endif
                    uint4 it; it.x = at_rec; it.y = cv; it.z = idx; it.w = cw;      // at_rec: resolved to a position id by the rare kernel
                    *reinterpret_cast<uint4 *>(myq + at) = it;
                }
            }
        }
    };

    // one piece of a writer stream per wave at a time; streams are cut into `split` pieces
    // (multiples of 512 records) when there are fewer streams than waves working on the bin
    constexpr uint32_t U = GBN_PROBE_U, BLK = 256u * U, NR = 4 * U;    // records per wave and per lane and round
    const int nwaves = nw * (GBN_BIN_THREADS / 64);
    const int split = (nwaves + B.nwriters - 1) / B.nwriters;
    const int V = B.nwriters * split;                               // pieces of a bin; this wave's: v0, v0 + nwaves, ... (v0 below)
    // hi words of a piece: blocks of 64 records = 96 words; a round of BLK records starts at a block
    // boundary (lo and BLK are multiples of 512), so a lane's words of round r sit at a fixed offset from
    // the piece's first block + r * (BLK / 64 * 96)
    uint32_t loff[U];
    #pragma unroll
    for (uint32_t u = 0; u < U; u++) { const uint32_t j = u * 256u + (uint32_t)lane * 4u; loff[u] = (j >> 6) * 96u + (j & 63u); }
    // Round 5: which bin, and which share of its streams, comes next is drawn from the group's counter (B.work; item t = share
    // t mod nw of the group's (t / nw)-th bin, so the group still works its way through one bin after the other and the bin's
    // table slice is fetched into the XCD's L2 once).  With fixed shares the kernel took as long as its slowest workgroup: 3.0 ms
    // alone, 3.5-3.9 ms next to the table builder and the extension stages of the pass before, whose waves land on some CUs
    // and not on others.  The next item's number is asked for when an item begins and read when it ends (two barriers later).
    const bool dyn = B.work != nullptr && B.nb >= GBN_BIN_GROUPS;
    volatile uint32_t *s_item = s_rcount + 1;                       // [2]
    int item = 0;                                                   // fixed shares: the item-th bin of this workgroup
    if (dyn) {
        if (tid == 0) s_item[0] = atomicAdd(&B.work[grp], 1u);
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane((int)s_item[0]);      // (wave-uniform: the bin, its table's address, the piece bookkeeping stay in scalar registers)
    }
    for (int par = 0; ; par ^= 1) {
        const int b = dyn ? grp + GBN_BIN_GROUPS * (item / nw) : grp % bstep + GBN_BIN_GROUPS * item;
        if (b >= B.nb) break;
        const int v0 = (dyn ? item % nw : wi) + nw * wave;
        if (dyn && tid == 0) s_item[par ^ 1] = atomicAdd(&B.work[grp], 1u);
        const uint32_t pad = GBN_REC_PAD(cbits, b);
        const int32_t tadj = GBN_BIN_TAB0 - (int32_t)(GBN_REC_PAR(cbits, b) << 15);     // s_tab index = low 16 bits of the hi word + tadj
        const uint32_t tab_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)s_tab + (uint32_t)(tadj * 4);   // LDS byte address of index 0
        // state of the piece in work; software pipeline: the loads of the next round are in flight while this round's
        // records are looked up
        uint32_t n = 0, rbase = 0;
        const uint32_t *__restrict__ pbase = B.rec;
        uint4 cur[U], nxt[U];
        auto start_piece = [&](int v, uint32_t ntot) {
            const int w = v / split, part = v - w * split;
            const uint32_t piece = ((ntot + (uint32_t)split * BLK - 1u) / ((uint32_t)split * BLK)) * BLK;
            const uint32_t lo = min((uint32_t)part * piece, ntot);
            n = min(piece, ntot - lo);
            // lo and every round start are multiples of 512 (one chunk per round in the chunked layout)
            rbase = (uint32_t)w * GBN_BINCAP(B, b) + lo;            // of this piece inside the bin's region (writer x the bin's stream capacity + index)
            pbase = B.rec + GBN_REC_HI(GBN_RECIDX(B, b, w, lo));
            #pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t j = u * 256u + (uint32_t)lane * 4u;
                const uint4 v = *reinterpret_cast<const uint4 *>(pbase + ((j < n) ? loff[u] : 0u));     // always a valid address: keeps the load a global load
                cur[u].x = (j < n) ? v.x : pad; cur[u].y = (j < n) ? v.y : pad;
                cur[u].z = (j < n) ? v.z : pad; cur[u].w = (j < n) ? v.w : pad;
            }
        };
        {
            // What a bin needs from memory before its first record is looked up -- the table slice, the side list, the
            // size of the wave's first piece and that piece's first round -- is asked for BEFORE the barrier behind which
            // the table of the bin before may be overwritten, every load in flight at once (written as `dst[i] = src[i]`
            // loops, a thread's eight + four loads went out one after the other, each waited for, and the piece's count
            // and first round behind the second barrier: some fourteen memory latencies per bin, 64 bins per workgroup,
            // with nothing else running on the CU)
            const uint4 *src = reinterpret_cast<const uint4 *>(B.cellt + ((size_t)b << cbits));
            uint4 *dst = reinterpret_cast<uint4 *>(s_tab + GBN_BIN_TAB0);
            constexpr uint32_t SL = GBN_BIN_CELLS / 4 / GBN_BIN_THREADS, SS = GBN_BIN_SIDE / GBN_BIN_THREADS;
            const uint32_t s0 = B.side_start[b], s1 = B.side_start[b + 1];
            const uint32_t nside = min(s1 - s0, (uint32_t)GBN_BIN_SIDE);
            uint4 tv4[SL]; uint16_t sv[SS];
            #pragma unroll
            for (uint32_t k = 0; k < SL; k++) { const uint32_t i = (uint32_t)tid + k * GBN_BIN_THREADS; tv4[k] = (i < ncell_bin / 4) ? src[i] : make_uint4(0, 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);                      // (the count is wanted in a scalar register at once: behind the slice's loads, not in front of them)
            const uint32_t g0 = (v0 < V) ? B.gcount[(size_t)b * B.nwriters + v0 / split] : 0u;
            #pragma unroll
            for (uint32_t k = 0; k < SS; k++) { const uint32_t i = (uint32_t)tid + k * GBN_BIN_THREADS; sv[k] = (i < nside) ? B.sidet[s0 + i] : (uint16_t)0; }
            if (v0 < V) start_piece(v0, g0);
            __syncthreads();                                        // every wave is through with the table of the bin before
            #pragma unroll
            for (uint32_t k = 0; k < SL; k++) { const uint32_t i = (uint32_t)tid + k * GBN_BIN_THREADS; if (i < ncell_bin / 4) dst[i] = tv4[k]; }
            #pragma unroll
            for (uint32_t k = 0; k < SS; k++) { const uint32_t i = (uint32_t)tid + k * GBN_BIN_THREADS; if (i < nside) s_side[i] = sv[k]; }
        }
        __syncthreads();
        for (int v = v0; v < V; ) {
            for (uint32_t j0 = 0; j0 < n; j0 += BLK) {
                const uint32_t rnext = ((j0 + BLK) >> 6) * 96u;       // word offset of the next round (a stream is far below 2^32 bytes)
                #pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t j = j0 + BLK + u * 256u + (uint32_t)lane * 4u;
                    const uint4 v = *reinterpret_cast<const uint4 *>(pbase + ((j < n) ? rnext + loff[u] : 0u));
                    nxt[u].x = (j < n) ? v.x : pad; nxt[u].y = (j < n) ? v.y : pad;
                    nxt[u].z = (j < n) ? v.z : pad; nxt[u].w = (j < n) ? v.w : pad;
                }
                uint32_t hv[NR], tv[NR];
                #pragma unroll
                for (uint32_t u = 0; u < U; u++) { hv[4 * u] = cur[u].x; hv[4 * u + 1] = cur[u].y; hv[4 * u + 2] = cur[u].z; hv[4 * u + 3] = cur[u].w; }
#if GBN_PROBE_ABL & 1
                // ablation (timing only, wrong results): the streams alone -- no table lookup, no test
                uint32_t slowm = 0, raw32 = 0;
                { uint32_t acc = 0;
                  #pragma unroll
                  for (uint32_t r = 0; r < NR; r++) acc ^= hv[r];
                  slowm = (acc == 0x12345678u) ? 1u : 0u; (void)tv; }
                rawu += raw32;
#elif GBN_PROBE_MASKS
                // all LDS lookups first (a pad reads one of the two empty extra cells); the byte address of a record's cell =
                // low half of the hi word x 4 + the table's base in one v_mad_u32_u16
                #pragma unroll
                for (uint32_t r = 0; r < NR; r++) {
                    uint32_t a;
                    asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a) : "v"(hv[r]), "v"(tab_addr));
                    tv[r] = *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)a;
                }
                __builtin_amdgcn_sched_barrier(0);              // (the tests below are not to be scheduled between the lookups)
                // The tests of a record leave lane masks in scalar registers and are combined there: the vector unit is what
                // bounds this kernel (round 4: 35 instructions per record, a third of them the select chains of `slow` and the
                // packed counters of c0 / c1), the scalar unit idles.  Lookup hits (as below: #c0 + #c1, three and more entries
                // corrected in flush()) are population counts of the same masks.
                uint32_t slowm = 0, raw32 = 0;                  // raw32: wave-uniform
                #pragma unroll
                for (uint32_t r = 0; r < NR; r++) {
                    const uint32_t t = tv[r];
                    const uint32_t x = (t ^ __builtin_amdgcn_perm(hv[r], hv[r], 0x07060706u)) & m4;
                    const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
                    // (against a zero the compiler does not know, bit 15 is ONE compare of the sign-extended low half)
                    const unsigned long long m0 = __ballot((int16_t)(uint16_t)t < opaque_zero), m1 = __ballot((int32_t)t < 0), mz = __ballot(z != 0);
                    const unsigned long long ms = (m0 & mz) | (m1 & ~m0);         // c0 ? some side matches : c1
                    raw32 += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
                    slowm |= __builtin_amdgcn_inverse_ballot_w64(ms) ? (1u << r) : 0u;
                }
                rawu += raw32;
#else
                #pragma unroll
                for (uint32_t r = 0; r < NR; r++) tv[r] = s_tab[(int32_t)(hv[r] & 0xffffu) + tadj];   // all LDS lookups first (a pad reads one of the two empty extra cells)
                // Both fingerprints of the cell word against the subject's in one go: a masked byte of
                // (t ^ sf:sf) is zero iff that side matches; (x - 0x01010101) & ~x & 0x80808080 is nonzero
                // iff some byte is zero.  One-entry cells hold their fingerprint twice.
                // lookup hits = entries of the cells hit: cells with one (c0) or two (c0 and c1) entries are
                // counted as #c0 + #c1 - #(c1 only); the c1-only cells (three or more entries) all take the
                // queue below, where they are subtracted again and their true size is added in flush()
                uint32_t flags = 0, slowm = 0;
                #pragma unroll
                for (uint32_t r = 0; r < NR; r++) {
                    const uint32_t t = tv[r];
                    const uint32_t x = (t ^ __builtin_amdgcn_perm(hv[r], hv[r], 0x07060706u)) & m4;   // the record's fp15 (upper half; bits 15 / 31 are masked) against both halves
                    const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
                    const bool c0 = (t & 0x8000u) != 0, c1 = (int32_t)t < 0;
                    const bool slow = c0 ? (z != 0) : c1;
                    flags += (t >> 15) & 0x10001u;              // c0 in the low half, c1 in the high half
                    slowm |= slow ? (1u << r) : 0u;
                }
                raw += (flags & 0xffffu) + (flags >> 16);
#endif
                // queue the (few) records that need the rare path: one per lane and round
                while (true) {
                    const unsigned long long m = __ballot(slowm != 0);
                    if (!m) break;
                    if (slowm) {
                        const uint32_t r = (uint32_t)__ffs(slowm) - 1u;
                        slowm &= slowm - 1;
                        // hi word of record r: a tree of bit-field inserts under masks of r's bits (written as selects, the
                        // compiler makes it an indexed load from a copy of hv[] in scratch memory: 2.8 -> 4.05 ms)
                        uint32_t hsel;
                        if constexpr (NR == 8) {
                            const uint32_t m0 = 0u - (r & 1u), m1 = 0u - ((r >> 1) & 1u), m2 = 0u - ((r >> 2) & 1u);
                            const uint32_t a0 = (hv[1] & m0) | (hv[0] & ~m0), a1 = (hv[3] & m0) | (hv[2] & ~m0);
                            const uint32_t a2 = (hv[5] & m0) | (hv[4] & ~m0), a3 = (hv[7] & m0) | (hv[6] & ~m0);
                            const uint32_t c0 = (a1 & m1) | (a0 & ~m1), c1 = (a3 & m1) | (a2 & ~m1);
                            hsel = (c1 & m2) | (c0 & ~m2);
                        } else {
                            hsel = 0;
                            #pragma unroll
                            for (uint32_t k = 0; k < NR; k++) hsel |= hv[k] & (0u - (uint32_t)(r == k));
                        }
                        uint2 qe; qe.x = rbase + j0 + (r >> 2) * 256u + (uint32_t)lane * 4u + (r & 3u); qe.y = hsel;
                        q[qn + __popcll(m & lt)] = qe;
                    }
                    qn += __popcll(m);
                    if (qn >= 64) { qn -= 64; flush(qn, 64, b); }
                }
                #pragma unroll
                for (uint32_t u = 0; u < U; u++) cur[u] = nxt[u];
            }
            v += nwaves;
            if (v < V) start_piece(v, B.gcount[(size_t)b * B.nwriters + v / split]);
        }
        if (qn > 0) { flush(0, qn, b); qn = 0; }                    // the side list changes with the bin
        item = dyn ? __builtin_amdgcn_readfirstlane((int)s_item[par ^ 1]) : item + 1;
    }
    if (P.raw_hits) {
        for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
        raw += rawu;
        if (lane == 0 && raw) atomicAdd(P.raw_hits, raw);
    }
    __syncthreads();
    if (tid == 0) B.rare_counts[blockIdx.x] = *s_rcount;
}

// rare path of the partitioned scan: one queued item per thread
// (Round 5, with the probe workgroups drawing their work from a counter the segments are no longer equally long.  Tried against
// this arrangement -- gridDim.x / nseg workgroups per segment, each walking its segment's rounds in steps of that number -- and
// not kept: all segments as one list dealt out round by round (FETCH_SIZE 3.7 instead of 2.9 GB per launch, 1.41 against 1.19
// ms in the pipeline), rounds dealt out by (tick, segment) slots of equal relative progress (3.4 GB; 1.15 against 1.09 ms),
// workgroups per segment in proportion to its length (3.55 GB, 1.5 ms).  The workgroups of this arrangement all sit at the same
// place of their segments, i.e. in the same few bins, whose cursors, cells and entries they share in the L2s: worth more than the
// last workgroups of the longest segment running alone.)
extern "C" __global__ void __launch_bounds__(256)
probe_rare_kernel(GbnBinParams B, int nseg)
{
    const GbnScanParams &P = B.S;
    unsigned long long raw = 0;
    // blockIdx.x % nseg = segment (probe workgroup), blockIdx.x / nseg = part
    const int seg = blockIdx.x % nseg, part = blockIdx.x / nseg, nparts = gridDim.x / nseg;
    const uint32_t n = min(B.rare_counts[seg], B.rare_seg);
    const GbnRareItem *qs = B.rareq + (size_t)seg * B.rare_seg;
    // dense-seed shapes (lut == word: every lookup hit is a seed) stage their seeds in LDS
    constexpr uint32_t CAP = 1536;
    __shared__ GbnDevSeed s_buf[CAP];
    __shared__ uint32_t s_n, s_flush_at;
    const bool staged = (P.fl == 0 && P.fr == 0);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    auto flush = [&]() {                                        // whole workgroup, after a barrier
        const uint32_t have = min(s_n, CAP);
        if (threadIdx.x == 0 && have) {
            const unsigned long long at = atomicAdd(P.seed_count, (unsigned long long)have);
            s_flush_at = (uint32_t)min(at, (unsigned long long)0xffffffffu);
        }
        __syncthreads();
        if (have) {
            const unsigned long long at = s_flush_at;
            for (uint32_t k = threadIdx.x; k < have; k += blockDim.x) if (at + k < P.seed_cap) P.seeds[at + k] = s_buf[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    for (uint32_t i0 = (uint32_t)part * 256u; i0 < n; i0 += (uint32_t)nparts * 256u) {     // uniform over the workgroup
        if (staged) { __syncthreads(); if (s_n > CAP - 512u) flush(); }      // s_n is stable between the barriers
        const uint32_t i = i0 + threadIdx.x;
        if (i >= n) continue;
        const uint4 item = *reinterpret_cast<const uint4 *>(qs + i);
        uint32_t pid = item.x; const uint32_t cv = item.y;
        uint32_t idx = item.z, cw = item.w;
        if (B.run_pos == nullptr) {   // (sorted records carry their position id: scan_runs.hip)
            // record index inside the bin's region -> (writer, index) -> tile via the cursor table -> position id
            const uint32_t bin = (cv & 0x7fffffffu) >> B.cbits;
            const uint32_t capb = GBN_BINCAP(B, bin), wr = pid / capb, j = pid - wr * capb;
            const uint32_t *__restrict__ cur = B.tcur + ((size_t)bin * B.nwriters + wr) * B.nseq;
#if !GBN_PROBE_FETCH
            // everything that hangs on the queue item alone is asked for here, in front of the cursor search
            idx = reinterpret_cast<const uint16_t *>(B.rec)[GBN_REC_IDX16(GBN_RECIDX(B, bin, wr, j))];
            cw = P.cellw[cv & 0x7fffffffu];
#endif
            // tiles of this writer: one per full round, and one of the last, incomplete round if its rotated index falls into it
            const uint32_t full_rounds = (uint32_t)(P.ntiles / B.nwriters), rest = (uint32_t)(P.ntiles % B.nwriters);
            const uint32_t ntiles_w = full_rounds + (((wr + full_rounds) % (uint32_t)B.nwriters) < rest ? 1u : 0u);
            const uint32_t nt = (ntiles_w + (1u << GBN_TCUR_SHIFT) - 1u) >> GBN_TCUR_SHIFT;          // cursor entries
            uint32_t lo = 0, hi = nt;
            const uint32_t total = B.gcount[(size_t)bin * B.nwriters + wr];
#if GBN_RARE_CUR4
            // The cursors grow almost linearly (a group of eight tiles leaves ~128 +- 11 records in a stream): FOUR cursors around
            // the interpolated one, fetched with one 16-byte load, name the run for nearly every record; the binary search below
            // is what is left for the others.  (Round 4 counted the kernel's requests: some twelve scattered loads per record -- two
            // cursors around a wider window and three steps of the search among them -- at the ~140 G requests per second the L2s
            // take ARE its 1.7 ms; the sectors behind them mostly sit in the Infinity Cache.)
            if (B.nseq >= 4u) {
                const uint32_t g = min((uint32_t)((float)j * ((float)nt / (float)(total ? total : 1u))), nt - 1u);
                const uint32_t a = min(g > 0u ? g - 1u : 0u, B.nseq - 4u);
                const u32x4 cq = load16_now(cur + a);
                struct { uint32_t c[4]; } c4 = {{cq.x, cq.y, cq.z, cq.w}};
                // entries from nt on are not cursors
                const uint32_t v1 = (a + 1u < nt) ? c4.c[1] : 0xffffffffu, v2 = (a + 2u < nt) ? c4.c[2] : 0xffffffffu, v3 = (a + 3u < nt) ? c4.c[3] : 0xffffffffu;
                if (a >= nt || c4.c[0] > j) hi = min(a, nt);
                else if (j < v1) { lo = a; hi = a + 1u; }
                else if (j < v2) { lo = a + 1u; hi = a + 2u; }
                else if (j < v3) { lo = a + 2u; hi = a + 3u; }
                else lo = a + 3u;
            }
#else
            {   // the cursors grow almost linearly: look around the interpolated run first
                const uint32_t g = (uint32_t)(((unsigned long long)j * nt) / (total ? total : 1u));
                constexpr uint32_t W = 3u;
                const uint32_t a = g > W ? g - W : 0u, z = min(nt, g + W);
                const uint32_t ca = cur[a], cz = (z < nt) ? cur[z] : 0xffffffffu;
                if (ca <= j && cz > j) { lo = a; hi = z; }
            }
#endif
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cur[mid] <= j) lo = mid; else hi = mid; }
            const uint32_t seqn = (lo << GBN_TCUR_SHIFT) | (idx >> GBN_BIN_TILE_BITS);
            pid = (GBN_TILE_OF(wr, seqn, (uint32_t)B.nwriters) << GBN_BIN_TILE_BITS) | (idx & (uint32_t)(GBN_BIN_TILE_POS - 1));
        }
        if (staged) probe_slow(P, pid, cv & 0x7fffffffu, (cv >> 31) != 0, cw, raw, s_buf, &s_n, CAP);
        else probe_slow(P, pid, cv & 0x7fffffffu, (cv >> 31) != 0, cw, raw);
    }
    if (staged) { __syncthreads(); flush(); }
    if (P.raw_hits) {
        for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
        if ((threadIdx.x & 63) == 0 && raw) atomicAdd(P.raw_hits, raw);
    }
}

namespace gbn {
hipError_t launch_probe_rare(const GbnBinParams &b, int nseg, hipStream_t st);
// parts: 1 = binning kernel, 2 = probe kernel, 4 = rare kernel (7 = all; 6 = a pass over records that exist: the record cache,
// a kernel queued ahead -- engine_scan.cpp; 1 = binning alone: bin-ahead, gbn_db_prepare_records)
hipError_t launch_scan_bin_parts(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev, int parts, hipEvent_t tables_ready)
{
    // ev[0..3]: before bin, after bin, after probe, after rare (optional); parts: 1 binning, 2 probe, 4 rare kernel, 8 no
    // ev[0] (the records were binned ahead: every packet between that kernel and the probe kernel is time the GPU idles)
    if (b.S.ntiles <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (ev && !(parts & 8)) (void)hipEventRecord(ev[0], st);
    if (parts & 1) {
#if GBN_BIN_PRESENCE
        if (tables_ready) { e = hipStreamWaitEvent(st, tables_ready, 0); if (e != hipSuccess) return e; }     // (the experiment's binning kernel reads the batch's presence bits)
#endif
        // stride-specialised variants: megablast (word 28 with lut 12 / 11 / 8) and blastn (word 11 with lut 11 / 10 / 8)
        const bool generic = (b.dbg & 64) != 0;
        const int step = b.S.step, lut = b.S.lut;
        if (step == 1 && lut == 11 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s1, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else if (step == 2 && lut == 10 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s2, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else if (step == 4 && lut == 8 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s4, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else if (step == 21 && lut == 8 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s21, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else if (step == 17 && lut == 12 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s17, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else if (step == 18 && lut == 11 && !generic) hipLaunchKernelGGL(scan_bin_kernel_s18, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        else hipLaunchKernelGGL(scan_bin_kernel, dim3(b.nwriters), dim3(GBN_SORT_THREADS), 0, st, b);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (ev) (void)hipEventRecord(ev[1], st);
    // the binning kernel needs no table: a batch whose lookup structures are still being built is waited for here
    if (tables_ready && (parts & 2)) { e = hipStreamWaitEvent(st, tables_ready, 0); if (e != hipSuccess) return e; }
    if (parts & 2) {
        const size_t lds = (size_t)GBN_BIN_TABW * 4 + (size_t)(GBN_BIN_THREADS / 64) * GBN_BIN_QCAP * 8 + (size_t)GBN_BIN_SIDE * 2 + 16;
        static std::atomic<uint64_t> attr_set{0};
        e = raise_dynamic_lds((const void *)probe_bin_kernel, lds, attr_set);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(probe_bin_kernel, dim3(grid2), dim3(GBN_BIN_THREADS), lds, st, b);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (ev) (void)hipEventRecord(ev[2], st);
    if (parts & 4) e = launch_probe_rare(b, grid2, st);
    if (ev) (void)hipEventRecord(ev[3], st);
    return e;
}
// the rare kernel over the queue segments of `nseg` probe workgroups
hipError_t launch_probe_rare(const GbnBinParams &b, int nseg, hipStream_t st)
{
    hipError_t e = hipSuccess;
    {
        // parts per segment (workgroups per queue segment).  Same box, C2, kernel alone / in the pipeline: 2 parts 1.95 / 2.09 ms,
        // 3: 1.60 / 1.72, 4: 1.49-1.52 / 1.61-1.69, 5: 1.47 / 1.86, 6: 1.81 / 1.91-1.93, 8 (rounds 1-4): 1.61 / 1.69-1.71,
        // 12: 1.62 / 1.70-1.72, 24: 1.55 / 1.63 (profiles/r04k_probe_and_rare_kernel.txt) -- the kernel is bound by the rate at which HBM
        // takes its scattered sectors, and more waves in flight do not raise it
        const int parts = (int)std::max(1ll, std::min(64ll, gbn::switch_value("GBN_RARE_PARTS", b.rare_parts > 0 ? b.rare_parts : 4)));
        if (!(b.dbg & 1)) hipLaunchKernelGGL(probe_rare_kernel, dim3(nseg * parts), dim3(256), 0, st, b, nseg);
        e = hipGetLastError();
    }
    return e;
}
hipError_t launch_scan_bin(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev)
{
    return launch_scan_bin_parts(b, grid2, st, ev, 3, nullptr);
}
}  // namespace gbn

