// gbn_dev.h -- structures shared by the HIP kernels and their host launchers.
#pragma once
#include <stdint.h>

#include "../../include/gblastn_amd_kernels.h"    // GbnTile, GbnDevSeed / InitHit / Gapped, GbnScanParams, GbnExtParams, GbnGapParams

#define GBN_SCAN_THREADS 256

// ---- key-range partitioned scan: phase 1 bins EVERY scan position by the top
// bits of its lookup word (8-byte records, no table access at all); phase 2
// loads one bin's cell table (<= 32768 cells x 4 B = 128 KiB) into LDS and
// streams that bin's records through it.
#define GBN_BIN_THREADS  1024       // probe kernel workgroup
#define GBN_SORT_THREADS 1024       // binning kernel workgroup: one per CU (two of 512 threads were measured 15-50 % slower)
#define GBN_BIN_WG_PER_CU 1
#ifndef GBN_OPEN_LINE
#define GBN_OPEN_LINE 32        // records per stored piece of the binning kernel (128 bytes of hi words + 64 of indices)
#endif
// cursor table resolution: one entry per 2^GBN_TCUR_SHIFT tiles of a (bin, writer) stream; the low bits of the
// tile's sequence number then travel in the spare top bits of every record's 16-bit index
#define GBN_TCUR_SHIFT 3
#define GBN_BIN_TILE_BITS 13
#define GBN_BIN_TILE_POS (1 << GBN_BIN_TILE_BITS)   // scan positions per tile (posid = tile << GBN_BIN_TILE_BITS | i)
#define GBN_BIN_STAGE GBN_BIN_TILE_POS   // LDS staging slots of the binning kernel (the lines a bin completes beyond its first one in a tile)
#define GBN_BIN_GROUPS   8          // probe workgroups with equal (blockIdx & 7) share a bin (and an XCD)
#define GBN_BIN_MAXNB    512
#define GBN_BIN_CELLS    32768      // most cells a bin can have (LDS table entries)
// cells per bin = 2^cbits: 128 bins for tables of 4^8 .. 4^11 cells, 512 for 4^12 -- a histogram over a handful
// of bins is a queue of LDS atomics on the same addresses (2 bins: binning 10.5 instead of 8 ms per 50 Gbp),
// 512 bins where 128 do cost more stream ends and cursors (lut 11: +10 %); never below 128 cells per bin
#define GBN_BIN_CBITS(lut) ((2 * (lut) - 7) < 7 ? 7 : ((2 * (lut) - 7) > 15 ? 15 : (2 * (lut) - 7)))
// words of the probe kernel's LDS table: [GBN_BIN_TAB0 - 1] an always-empty cell, [GBN_BIN_TAB0, + cells) the bin's cells,
// [GBN_BIN_TAB0 + GBN_BIN_CELLS] another always-empty cell.  Records of a table with 2^15 cells per bin carry the lowest
// bit of their bin number in bit 15 of the hi word (scan_bin3_body): the probe kernel indexes with the low 16 bits and
// subtracts GBN_REC_PAR(cbits, bin) << 15; a pad record names the empty cell on the far side -- needs no special case
#define GBN_BIN_TAB0     4
#define GBN_BIN_TABW     (GBN_BIN_CELLS + 8)
#define GBN_REC_PAR(cbits, bin) (((cbits) == 15) ? ((uint32_t)(bin) & 1u) : 0u)
#define GBN_REC_PAD(cbits, bin) (GBN_REC_PAR(cbits, bin) ? 0x00007fffu : 0x00008000u)
#define GBN_BIN_QCAP     128        // per-wave queue of rare-path items in the probe kernel
#define GBN_BIN_SIDE     4096       // LDS side-list capacity (u16 fingerprints) per bin

// tile processed by binning workgroup (= writer) w in its round k, n writers (see scan_bin_line_body)
#define GBN_TILE_OF(w, k, n) ((uint32_t)(k) * (uint32_t)(n) + (((uint32_t)(w) + (uint32_t)(k)) % (uint32_t)(n)))

struct GbnU2 { uint32_t x, y; };
// an item of the rare-path queue: x = record index inside its bin's region, y = cell (bit 31: a cell whose lookup hits are
// counted by the rare kernel), z = the record's 16-bit index, w = the cell's direct-probe word (GbnScanParams::cellw) -- both
// fetched by the probe kernel when it queues the item, next to its streams, so that the rare kernel, which is bound by the
// rate at which HBM takes scattered sectors, has one sector per item left to fetch (the subject's)
struct GbnRareItem { uint32_t x, y, z, w; };
// Scan records: blocks of 64 records = 256 bytes of `hi` words followed by 128 bytes of 16-bit indices
// (13 bits: position inside the tile, 3 bits: the tile's sequence number mod 8); the tile itself is not
// stored -- the few records that reach the rare path find it in the cursor table (GbnBinParams::tcur).
// GBN_REC_HI(j) = 32-bit word offset of record j's hi word (j = linear record index).
#define GBN_REC_HI(j)    ((((size_t)(j)) >> 6) * 96 + (((size_t)(j)) & 63))
#define GBN_REC_IDX16(j) (((((size_t)(j)) >> 6) * 96 + 64) * 2 + (((size_t)(j)) & 63))    /* 16-bit offset */
#define GBN_REC_WORDS(n) ((size_t)(n) / 64 * 96)                                          /* n: multiple of 64 */
struct GbnBinParams {
    GbnScanParams S;                // tiles here are GBN_BIN_TILE_POS-sized
    int nb, cbits;                  // number of bins; cell = bin << cbits | low
    int nwriters;                   // workgroups of the binning kernel = private output streams per bin
    // cell table, one word per cell.  Reduced fingerprint fp15 = right-7-bits << 8 | left-4-bases.
    //   bit 15 = c0, bit 31 = c1:  00 empty;  10 (c0) one entry: [14:0] = [30:16] = its fp15;
    //   11 two entries: [14:0] fpA, [30:16] fpB;  01 (c1 only) three or more: [14:0] offset into the
    //   bin's side list, [30:16] count (0 = always rare path)
    const uint32_t *cellt;
    const uint16_t *sidet;          // reduced fingerprints of cells with >= 3 entries, per bin
    const uint32_t *side_start;     // [nb + 1] offsets into sidet
    // records: streams [nb][nwriters] of subcap records (a multiple of 32), stored in blocks of 32
    // records = 128 bytes of `hi` words followed by 128 bytes of `posid` words (GBN_REC_HI/POS below).
    // The probe kernel streams the hi lines only and fetches posid for the ~1 % of records that reach
    // the rare path; a run of the binning kernel still lands in one contiguous stretch of memory.
    //   hi:    [30:16] fp15 of the subject position, [15:0] cell inside the bin (bit 15: GBN_REC_PAR; bit 31 undefined)
    //   posid: tile << GBN_BIN_TILE_BITS | index
    uint32_t *rec;
    // 6-byte records: stream cursor of (bin, writer) at the start of its seq-th tile,
    // [nb][nwriters][nseq]; tile of a record = writer + seq * nwriters for the last seq whose cursor <= index
    uint32_t *tcur; uint32_t nseq;
    uint32_t *gcount;               // [nb][nwriters] records written (multiple of 4, pads included)
    uint32_t subcap;
    // Streams of bins that differ in size (round 6; repeat-rich subjects put most of their scan positions into a few bins): bincap[2 b] =
    // records a stream of bin b has room for (a multiple of 512), bincap[2 b + 1] = where bin b's stream begins inside a writer's row,
    // rowsize = records per row.  Null: every stream has room for `subcap` records (GBN_BINCAP / GBN_RECIDX, scan_dev.hpp).  gtotal
    // [nb][nwriters]: what every stream WOULD hold -- the binning kernel counts on past a stream's end, so an attempt that overflows
    // tells the engine exactly how much room the next one needs.
    const uint32_t *bincap; size_t rowsize; uint32_t *gtotal;
    uint32_t *overflow;             // set to 1 if any stream did not fit
    int rfl, rfrbits;               // reduced fingerprint: bases on the left (<= 4), BITS on the right (<= 7 = 3.5 bases)
    int dbg;                        // launch switches of tools/scan_ablate.py (GBN_DBG; 1: no rare kernel, 32: timing print, 64: any-stride binning kernel, 128: XCD report, 256: probe kernel without the sixteenth fingerprint bit)
    GbnRareItem *rareq; uint32_t rare_seg;    // rare-path queue: one segment of rare_seg items per probe workgroup
    uint32_t *rare_counts;              // [probe workgroups] items queued (may exceed rare_seg: overflow)
    // probe kernel: [GBN_BIN_GROUPS] counters, zero at the launch -- the workgroups of a group draw their (bin, share of its
    // streams) items from the group's counter instead of owning a fixed share of every bin: a workgroup that shares its CU
    // with the table builder's or the extension stages' waves takes fewer items and the kernel does not wait for it.
    // Null (or fewer bins than groups): fixed shares.
    uint32_t *work;
    // workgroups of the rare kernel per queue segment; 0: the default (4, GBN_RARE_PARTS).  The engine asks for 5 when the pass
    // runs over cached records: the next batch's table build is in the middle of its sort then, and five workgroups of the rare
    // kernel per CU (143 of its 160 KB of LDS, all 1,280 resident at once) leave those kernels less room than four do -- 1.22
    // instead of 1.47 ms, 4.59 instead of 4.82 ms per pass; alone, and behind a binning kernel, four are the faster (0.97 / 1.03).
    int rare_parts;
    // ---- the SORTED form of a record set ("runs", scan_runs.hip): the records of a cell are consecutive,
    // run_start[cell] .. run_start[cell + 1], a record = its 16 subject bits around the lookup word (run_fp: [7:0] the 4 bases in
    // front, [14:8] the 7 bits behind, [15] the eighth) and its position id (run_pos: tile << GBN_BIN_TILE_BITS | index).  The
    // cell is the run: a pass reads the runs of the cells its batch occupies and nothing else.  Null: stream form (rec / tcur).
    const uint16_t *run_fp; const uint32_t *run_pos; const uint32_t *run_start;
    int run_item_cells;             // cells a wave of probe_runs_kernel draws at a time (a multiple of 64 that divides the table; 0: GBN_RUNS_ITEM_CELLS)
};

// ---- building the runs from the streams of a complete record set (scan_runs.hip): count per cell -> run_start (prefix sums) ->
// split (every bin's records into 2^sbits sub-bins of consecutive cells: blocks sorted in LDS, pieces appended at the sub-bin's
// cursor) -> place (a sub-bin at a time through LDS into its cells' runs)
#define GBN_RUNS_SPLIT_CAP 16384        // records a split workgroup sorts per round (LDS: 8 bytes each)
#define GBN_RUNS_PLACE_CAP 24576        // records a place workgroup stages per round (LDS: 6 bytes each)
#define GBN_RUNS_UNIT_CELLS_MAX 2048    // most cells of a sub-bin (their cursors sit in LDS next to the staged records)
#define GBN_RUNS_SBITS_MAX 8
#define GBN_RUNS_CURCAP 4096            // stream cursors a split workgroup keeps in LDS (more: read from global memory)
#define GBN_RUNS_ITEM_CELLS 256         // cells a wave of the probe kernel draws at a time
struct GbnRunsBuild {
    GbnBinParams B;                     // the streams (rec, tcur, gcount, subcap, nb, cbits, nwriters, nseq; S.ntiles)
    int sbits;                          // sub-bins per bin = 2^sbits
    int wgroup;                         // consecutive writers whose streams one split workgroup works through
    uint32_t *count;                    // [ncells + 1]: records per cell, then (in place) run_start
    uint32_t *cursor;                   // [nb << sbits]: next free slot of every sub-bin
    uint32_t *mid_key, *mid_pos;        // records by sub-bin: hi word, position id
    uint16_t *fp; uint32_t *pos;        // the runs
};

struct GbnKeyParams {
    const GbnDevSeed *seeds; int64_t n;
    uint64_t *key_scan; uint64_t *key_group; uint32_t *idx;
    int q_descending, container_hash, diag_len;
    int q_bits, group_bits;     // key widths: query offsets < 2^q_bits, slots < 2^group_bits (the sorts stop at the keys' top bit)
    int s_bits, qh_bits;        // composite key (seed_ckeys_kernel): subject offsets < 2^s_bits, qh_bits = max(0, q_bits - group_bits)
    int subj_base;              // ... its subject field counts from the first subject of the launch
    int v_bits;                 // > 0: key_scan[i] = composite key << v_bits | value (no idx): one sort of keys only
    // nseg > 0 (seed_ckeys_kernel): the seeds are not in `seeds` but in nseg segments of seg_cap slots, seg_count[s] of
    // them in segment s (scan_slice_kernel's output as it is); seed i = the i-th of the segments read one after the other
    const GbnDevSeed *seg; const uint32_t *seg_count; int nseg; uint32_t seg_cap;
    unsigned long long *seg_first;      // nseg + 1 entries of scratch (launch_seed_ckeys fills them: index of a segment's first seed)
    // round 6: the segments hold 8-byte composite keys (key << v_bits | value, what gbn_composite_key makes of a seed) instead of
    // 16-byte seeds -- scan_fold_ordered_kernel writes them when the engine knows at scan time that the range's seeds go through the
    // seed-order kernels (seg_cap counts elements either way; a key segment uses half of its bytes)
    int seg_keys;
};

// scan_slice_kernel: a slice of the presence bits per workgroup
#define GBN_SLICE_THREADS   1024
#define GBN_SLICE_CELL_BITS 20          // cells per slice: 2^20 bits = 128 KB of LDS
#define GBN_SLICE_WORDS     (1 << (GBN_SLICE_CELL_BITS - 5))
#define GBN_SLICE_QCAP      128         // per-wave queue of present positions
#define GBN_SLICE_MAX       16          // most slices (passes over the subjects) it is used with
#define GBN_SLICE_SEGS      4096        // most output segments of a launch (a workgroup's, or a wave's when the seeds come in scan order)

// seed_order.hip: the seeds of an ordered scan by (subject, slot) without a radix sort
#define GBN_ORDER_CHUNK     4096        // seeds per histogram (a chunk never crosses a subject)
#define GBN_ORDER_MAX_SUBJ  4096        // most subjects of a launch
#define GBN_ORDER_MAX_SLOTS 2048        // most slots of the diagonal container (the hash container has 512)
struct GbnOrderParams {
    GbnKeyParams K;                     // the segments, the key layout; key_scan = the ordered keys
    int nsubj;
    int32_t *seg_next;                  // GBN_SLICE_SEGS words of scratch
    uint32_t *subj_first, *chunk_first; // nsubj + 1 each: first seed / first chunk of a subject
    uint32_t *counts;                   // [chunk][slot]
    uint32_t *slot_base;                // [subject][slot]
};

// seed_sort.hip: most seeds ONE workgroup puts into the diagonal filter's order (beyond: the two library sorts, as rounds 1-4)
#define GBN_SMALL_SORT_MAX 65536

// seeds per launch below which the diagonal kernel runs thread-per-seed instead of on compacted run heads
#ifndef GBN_DIAG_COMPACT_MIN
#define GBN_DIAG_COMPACT_MIN (1 << 20)
#endif

// ---- GPU time per kernel class of the stages behind the scan (GbnDiagnostics::kernel_ms): the launchers record a HIP
// event wherever the class of the kernels they queue changes, the engine reads the intervals after the stage's stream
// synchronisation.  Indices = GBN_KT_* of include/gblastn_amd.h.
#ifdef __cplusplus
#include <hip/hip_runtime_api.h>
#include "../../include/gblastn_amd.h"     // GBN_KT_*
struct GbnKernelTimer {
    enum { CAP = 24 };
    hipEvent_t ev[CAP]; int tag[CAP]; int n = 0; bool made = false;
    // an event on `st`; what is queued behind it belongs to class `t` (-1: nothing that is timed)
    void mark(int t, hipStream_t st) {
        if (!made) { for (int i = 0; i < CAP; i++) if (hipEventCreate(&ev[i]) != hipSuccess) return; made = true; }
        if (n < CAP) { if (hipEventRecord(ev[n], st) == hipSuccess) tag[n++] = t; }
    }
    // after the stream was synchronised: the intervals added to kernel_ms[class]; ready for the next stage
    void collect(double *kernel_ms) {
        for (int i = 0; i + 1 < n; i++) {
            float ms = 0;
            if (tag[i] >= 0 && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) kernel_ms[tag[i]] += ms;
        }
        n = 0;
    }
    void destroy() { if (made) for (int i = 0; i < CAP; i++) (void)hipEventDestroy(ev[i]); made = false; n = 0; }
};
#endif

#ifdef __HIPCC__
// The composite key of a seed (seed_ckeys_kernel, seed_order_scatter_kernel): subject | slot | s_scan, and the value that
// travels with it (ext_left | high bits of the query key << 8).  slot = the seed's cell of the diagonal container.
__device__ __forceinline__ uint64_t gbn_composite_key(const GbnKeyParams &K, const GbnDevSeed &sd, uint32_t qmax, uint32_t &slot, uint32_t &val)
{
    const uint32_t qkey = K.q_descending ? (qmax - (uint32_t)sd.q_pos) : (uint32_t)sd.q_pos;
    slot = K.container_hash ? ((uint32_t)(sd.s_scan - sd.q_pos) & 511u)
                            : ((uint32_t)(sd.s_scan + K.diag_len - sd.q_pos) & (uint32_t)(K.diag_len - 1));
    uint64_t key = ((uint64_t)(uint32_t)(sd.subj - K.subj_base) << K.group_bits) | slot;
    key = (key << K.s_bits) | (uint32_t)sd.s_scan;
    // the high bits of the query key order the (rare) seeds of one (subject, slot, scan position): they travel in
    // the value, and seed_ext_kernel puts such a group into their order -- nine bits less to sort
    val = (uint32_t)sd.ext_left | ((K.qh_bits ? (qkey >> K.group_bits) : 0u) << 8);
    return key;
}
// ... and back: the seed a packed key (key << v_bits | value) stands for.  The query offset's low bits are not in the key; the slot
// gives them (slot = (s_scan [+ diag_len] - q_pos) mod slots), its high bits travel in the value.
__device__ __forceinline__ GbnDevSeed gbn_seed_of_key(const GbnKeyParams &K, uint64_t packed, uint32_t qmax)
{
    const uint32_t val = (uint32_t)(packed & ((1ull << K.v_bits) - 1ull));
    const uint64_t key = packed >> K.v_bits;
    GbnDevSeed sd;
    sd.s_scan = (int32_t)(uint32_t)(key & ((1ull << K.s_bits) - 1ull));
    const uint32_t nslots = K.container_hash ? 512u : (uint32_t)K.diag_len;
    const uint32_t slot = (uint32_t)(key >> K.s_bits) & (nslots - 1u);
    sd.subj = (int32_t)(uint32_t)(key >> (K.s_bits + K.group_bits)) + K.subj_base;
    sd.ext_left = (int32_t)(val & 0xffu);
    const uint32_t qh = val >> 8;
    const uint32_t qlow = ((uint32_t)sd.s_scan - slot) & (nslots - 1u);            // q_pos mod slots (diag_len is a multiple of the slots: it drops out)
    uint32_t q;
    if (K.q_descending) {
        // qkey = qmax - q_pos, qkey >> group_bits = qh: q_pos lies in (qmax - (qh << g) - slots, qmax - (qh << g)], one value of which has the low bits
        const uint32_t top = qmax - (K.qh_bits ? (qh << K.group_bits) : 0u);
        q = top - ((top - qlow) & (nslots - 1u));
    } else q = (K.qh_bits ? (qh << K.group_bits) : 0u) | qlow;
    sd.q_pos = (int32_t)q;
    return sd;
}
#endif

// ---- switches (the library's environment variables): read through these two functions at EVERY use (rounds
// 1-3 cached several in function-local statics: a process could not change them after their first use and the tests
// spawned a child per setting).  The device pool's GBN_POOL_GIB / GBN_GUARD / GBN_POISON and GBN_TRACE stay process-wide:
// blocks handed out under one setting cannot be checked under another.
#ifdef __cplusplus
namespace gbn {
long long switch_value(const char *name, long long dflt);   // the variable as an integer, dflt if it is not set
bool switch_is_set(const char *name);
}
#endif
