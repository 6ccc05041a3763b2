/* gblastn_amd_kernels.h -- parameter blocks of the gbn_launch_* entry points of gblastn_amd.h.
 *
 * Plain C structs of DEVICE pointers and scalars; nothing here includes a private header.  The launchers are the
 * library's equivalents of the three callbacks G-BLASTN swaps into the core engine:
 *
 *   gbn_launch_scan_seed   the scan + mini-extension pair: TNaScanSubjectFunction (COREI/blast_nascan.h:43,
 *                          CORE/blast_nascan.c) feeding TNaExtendFunction (COREI/na_ungapped.h:51,
 *                          CORE/na_ungapped.c:1025-1144 / 1291-1470) -- what gpu_blastn_MB_and_smallNa.cpp:70-190 runs
 *                          on the device.  One launch scans every tile given, subject by subject.
 *   gbn_launch_ungapped    s_BlastnDiagTableExtendInitialHit / s_BlastnDiagHashExtendInitialHit
 *                          (CORE/na_ungapped.c:600-789, 813-1003): last-hit filter per diagonal + ungapped X-drop
 *                          extension of the seeds that pass it, in the order the reference visits them.
 *   gbn_launch_gapped      BlastGetGappedScoreType's inner aligners (CORE/blast_gapalign.c:2762-3052 dynamic
 *                          programming on the packed subject; CORE/greedy_align.c greedy), score only, one initial hit
 *                          each, independent of one another (containment is the caller's business: gbn_prelim_search
 *                          replays the reference's interval-tree loop on the host).
 *
 * A caller that already holds a GbnBatch and a GbnDb gets the database, lookup and query members filled in by
 * gbn_batch_scan_params / gbn_batch_ext_params / gbn_batch_gap_params (gblastn_amd.h) and supplies only the work
 * list and the output / scratch buffers marked [caller] below.
 */
#ifndef GBLASTN_AMD_KERNELS_H
#define GBLASTN_AMD_KERNELS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBN_TILE_POS     2048       /* most scan positions in one tile of gbn_launch_scan_seed */
#ifndef GBN_DIAG_COMPACT_MIN
#define GBN_DIAG_COMPACT_MIN (1 << 20)  /* seeds per gbn_launch_ungapped from which the runs are compacted first (run_heads) */
#endif

/* mini-extension flavours (CORE/na_ungapped.c:1753-1795) */
#define GBN_EXT_DIRECT        0     /* lut == word_size */
#define GBN_EXT_NA            1     /* s_BlastNaExtend / s_BlastNaExtendAligned */
#define GBN_EXT_SMALL         2     /* s_BlastSmallNaExtend */
#define GBN_EXT_SMALL_ONEBYTE 3     /* s_BlastSmallNaExtendAlignedOneByte */

/* a run of scan positions of one subject: positions first_pos, first_pos + step, ... (npos of them, <= GBN_TILE_POS);
 * first_pos is a BASE offset and a multiple of the scan step; off16 = byte_off[subj] / 16 (subjects start on
 * 16-byte boundaries of the slab) */
typedef struct GbnTile { int32_t subj; int32_t first_pos; int32_t npos; int32_t off16; } GbnTile;

/* a seed: the lookup hit at subject base s_scan / query position q_pos (concatenated query coordinates), whose
 * exact match extends ext_left bases to the left; the word-sized match starts at (q_pos - ext_left, s_scan - ext_left) */
typedef struct GbnDevSeed { int32_t subj, s_scan, q_pos, ext_left; } GbnDevSeed;
/* BlastInitHSP + BlastUngappedData (COREI/blast_extend.h:125-150); seq = rank of the seed that produced it */
typedef struct GbnDevInitHit { int32_t subj, q_off, s_off, q_start, s_start, length, score; uint32_t seq; } GbnDevInitHit;
/* what the score-only aligners leave in BlastGapAlignStruct (COREI/blast_gapalign.h:70-100) */
typedef struct GbnDevGapped { int32_t q_start, q_stop, s_start, s_stop, score, seed_q, seed_s, context; } GbnDevGapped;
#define GBN_GAP_REDO (INT32_MIN + 1)    /* GbnDevGapped::score while an extension waits for the second kernel of gbn_launch_gapped; never seen by the caller */

typedef struct GbnScanParams {
    /* database slab: packed NCBI2na, subject s at db + byte_off[s], len[s] bases, 16 readable bytes past every subject */
    const uint8_t *db; const int64_t *byte_off; const int32_t *len;
    const GbnTile *tiles; int64_t ntiles;                                   /* [caller] */
    /* lookup structures of the batch */
    const uint32_t *pv;             /* 1 bit per cell */
    const uint32_t *cellw;          /* bit31 = more than one entry, [30:15] left-8, [14:1] right-7, bit0 force */
    const uint32_t *cell_start;     /* ncells + 1 */
    const unsigned long long *ent;  /* low 32 = query offset, high 32 = fingerprint word */
    int64_t ncells;
    int lut, word, step, mode, fl, fr;
    /* query (one byte per base, index 0 = first base of strand 0) */
    const uint8_t *q8; int32_t qlen;
    const int32_t *ctx_off, *ctx_len; int32_t nctx;
    /* outputs [caller]: seeds[0 .. min(*seed_count, seed_cap)) in NO particular order (order them by
     * (subj, s_scan, q_pos descending for megablast tables / ascending otherwise) to get the reference's);
     * *seed_count counts every seed, also those that did not fit; *raw_hits the lookup hits before the
     * mini-extension.  Both counters must be zeroed before the launch. */
    GbnDevSeed *seeds; unsigned long long *seed_count; unsigned long long seed_cap;
    unsigned long long *raw_hits;
    /* optional (NULL: not used): rank form of pv / cell_start for tables of more than 2^20 cells that are as wide as the
     * word -- pvx[2 w], pvx[2 w + 1] = pv[w], number of present cells in words < w; pstart[r], pstart[r + 1] = the entry
     * list of the r-th present cell.  With them the subjects are read once instead of once per 2^20 cells. */
    const uint32_t *pvx, *pstart;
} GbnScanParams;

typedef struct GbnExtParams {
    const uint8_t *db; const int64_t *byte_off; const int32_t *len;
    /* [caller] the seeds and their visiting order: idx[i] = seed visited i-th, sorted by (subj, slot, scan order)
     * where slot = (s - q) & (diag_len - 1) for the diagonal array, (s - q) & 511 for the hash container (s, q =
     * start of the word-sized match) and scan order = (s_scan, q_pos in table order); key_group[i] =
     * subj << group_bits | slot of that seed */
    const GbnDevSeed *seeds; const uint32_t *idx; const uint64_t *key_group; int64_t n;
    const uint8_t *q8; int32_t qlen;
    const int32_t *ctx_off, *ctx_len, *ctx_xdrop, *ctx_cutoff, *ctx_reduced; int32_t nctx;
    const int32_t *matrix;          /* 16 x 16 */
    const int32_t *score_table;     /* 256: score of 4 packed bases XORed (CORE/na_ungapped.c:228-257) */
    int word, container_hash;
    int32_t *cell_diag, *cell_level;    /* [caller] scratch, n entries each */
    /* re-check of seeds against the soft query masks (s_TypeOfWord): table membership tests */
    const uint32_t *cell_start; const unsigned long long *ent; uint32_t cell_mask; int lut, masked;
    /* the query 2 bits per base and the bitmap of codes that match nothing (as in GbnGapParams) */
    const uint8_t *q2, *qinv;
    uint32_t *run_heads, *run_count;    /* [caller] scratch: n entries / one counter */
    int32_t group_bits;                 /* [caller] see key_group (0 is read as 32) */
    /* [caller] initial hits that reached the cutoff, in no particular order (GbnDevInitHit::seq orders them);
     * *ihit_count counts all of them (zero it first) */
    GbnDevInitHit *ihits; unsigned long long *ihit_count; unsigned long long ihit_cap;
    /* optional: ctx_hint[q >> ctx_hint_shift] = a context starting at or before query position q, not more than a
     * few contexts before the one q lies in (null: binary search over ctx_off) */
    const int32_t *ctx_hint; int32_t ctx_hint_shift;
    /* [caller] optional scratch, 32 bytes per seed: with it, launches of GBN_DIAG_COMPACT_MIN seeds and more
     * extend every seed in a kernel of its own and replay the runs over the records (null: one kernel) */
    void *ext_rec;
    /* composite-key form (ck_shift > 0; needs ext_rec): key_group[i] = subj << (group_bits + ck_shift) | slot << ck_shift
     * | s_scan (ck_shift = ck_s_bits), sorted as ONE 64-bit key; idx[i] = the seed's ext_left | (query key >> group_bits)
     * << 8; `seeds` is not read (a seed follows from key and value: q_pos's low bits = (s_scan - slot) mod slots).
     * Seeds with equal keys may come in any order: the kernel orders them by the value's high bits.  query key = q_pos,
     * or 2^ck_q_bits - 1 - q_pos when ck_q_desc (megablast tables: chains are reported last position first) */
    int32_t ck_shift, ck_s_bits, ck_qh_bits, ck_q_bits, ck_q_desc, ck_subj_base;   /* subj in the key counts from ck_subj_base */
    /* ck_vbits > 0: the value travels in the key's low ck_vbits bits (key_group[i] = composite key << ck_vbits | value,
     * sorted on the bits above them: a sort of keys only) and idx is not read */
    int32_t ck_vbits;
    /* optional: the query four bases per byte at EVERY offset -- for query position p the byte s_NuclUngappedExtend
     * builds per step, (uint8_t)((q8[p] << 6) | (q8[p+1] << 4) | (q8[p+2] << 2) | q8[p+3]) (CORE/na_ungapped.c:296, :323;
     * ambiguity codes and sentinels spill as they do there) -- in four planes by offset: with k = p + q4_origin the byte is
     * q4[(k & 3) * q4_plane + (k >> 2)], so that the bytes of consecutive 4-base steps are consecutive in memory.
     * Positions -64 .. qlen + 64 must be there, and 16 readable bytes behind every plane.  With it the approximate
     * extension takes eight steps per round from two 8-byte loads (null: step by step from q8 / q2) */
    const uint8_t *q4; int64_t q4_plane; int32_t q4_origin;
    /* optional: ctx_blk[2 * (q >> ctx_hint_shift)] = the context the block's first position lies in, [.. + 1] = the query
     * offset at which the next context begins (INT32_MAX: none; INT32_MIN: more than one context begins inside the block,
     * look it up through ctx_hint / ctx_off) -- a seed's context from one 8-byte load */
    const int32_t *ctx_blk;
    /* optional: ctx_pack[4 c], [4 c + 1], [4 c + 2] = ctx_xdrop[c], ctx_reduced[c], ctx_cutoff[c] (one 16-byte read per seed) */
    const int32_t *ctx_pack;
    /* optional [caller] scratch: n entries and a counter.  With them the seeds whose approximate score reaches the reduced
     * cut-off are listed and extended exactly by a kernel of their own (composite keys only) */
    uint32_t *exact_list, *exact_count;
} GbnExtParams;

typedef struct GbnGapParams {
    const uint8_t *db; const int64_t *byte_off; const int32_t *len;
    const GbnDevInitHit *ihits; int64_t first, n;           /* [caller] extensions ihits[first .. first + n) */
    const uint8_t *q8; const int32_t *ctx_off, *ctx_len; int32_t nctx;
    /* greedy only: the query 2 bits per base (same packing as the subjects) and a bitmap (MSB first)
     * of the codes that match nothing; both indexed from base 0 and readable 256 bases either side */
    const uint8_t *q2, *qinv;
    const int32_t *matrix;
    int32_t reward, penalty, gap_open, gap_extend, xdrop;
    /* [caller] scratch: scratch_per_thread ints for each of 64 * (max_blocks, or ceil(n / 64)) threads.
     * dynamic programming: scratch_per_thread >= 2 * (longest context + 16); greedy: see gbn_gap_scratch_ints() */
    int32_t *scratch; int32_t scratch_per_thread, row_len;
    GbnDevGapped *out;              /* [caller] out[first + i] = extension of ihits[first + i] */
    int32_t max_blocks;             /* grid cap in 64-thread blocks (0: one thread per initial hit); the threads stride over the hits */
    int32_t redo_only;              /* internal: 0 */
} GbnGapParams;

#ifdef __cplusplus
}
#endif
#endif
