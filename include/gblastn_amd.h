/* gblastn_amd.h -- C ABI of the MI355X-native blastn/megablast preliminary-search
 * engine.  This is the drop-in boundary for the one hot path this project
 * replaces: G-BLASTN's GPU preliminary search.
 *
 * Reference interface each entry point replaces (paths relative to
 * /root/reference/c++):
 *
 *   gbn_init / gbn_release      replace Blast_gpu_Init(bool, int) / Blast_gpu_Release()
 *       include/algo/blast/gpu_blast/gpu_blastn.h:50-51
 *       (defined src/algo/blast/gpu_blast/gpu_blast_multi_gpu_utils.cpp:176-183)
 *   gbn_release_db_memory       replaces gpu_ReleaseDBMemory()
 *       include/algo/blast/gpu_blast/gpu_blastn_na_ungapped_v3.h:21
 *       The reference declares those three with C++ linkage (no extern "C" anywhere in its tree), so a C library
 *       cannot export them under their own names: gblastn_amd/shim/gpu_blastn_amd_shim.cpp defines them, with the
 *       reference's signatures, as forwarders to the three functions above.
 *   gbn_db_*            replaces gpu_InitDBMemroy + the per-OID cudaMalloc cache
 *       src/algo/blast/gpu_blast/gpu_blastn_MB_and_smallNa.cu:140-146,1462-1468
 *       (one contiguous HBM slab per volume instead of one allocation per OID)
 *   gbn_batch_*         replaces LookupTableWrapInit + BLAST_MainSetUp products
 *       that the reference passes INTO the boundary (query, query_info, sbp,
 *       lookup_wrap) plus GpuLookUpSetUp / gpu_InitQueryMemory
 *       src/algo/blast/gpu_blast/gpu_blastn_na_ungapped_v3.cpp:595-696
 *   gbn_prelim_search   replaces Blast_gpu_RunPreliminarySearchWithInterrupt
 *       include/algo/blast/gpu_blast/gpu_blastn.h:31-48
 *       src/algo/blast/gpu_blast/gpu_blastn_pre_search_engine.cpp:1466-1563
 *       Same contract: caller owns query/options/database; callee creates and
 *       frees all search parameters; results are appended to a caller-owned
 *       result list (the BlastHSPStream analogue); 0 = success, non-zero
 *       status codes, GBN_ERR_INTERRUPTED on user interrupt; no exceptions.
 *   gbn_launch_*        the finer-grained hooks G-BLASTN swaps in:
 *       scansub_callback / extend_callback (TNaScanSubjectFunction,
 *       include/algo/blast/core/blast_nascan.h:43; TNaExtendFunction,
 *       include/algo/blast/core/na_ungapped.h:51), BlastWordFinderType
 *       (include/algo/blast/core/blast_engine.h:227) and BlastGetGappedScoreType.
 *       They take device pointers and a hipStream_t (passed as void*).
 *
 * All structs are plain old data; no torch / C++ types cross this boundary.
 */
#ifndef GBLASTN_AMD_H
#define GBLASTN_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBN_OK                 0
#define GBN_ERR_ARG            1
#define GBN_ERR_NO_DEVICE      2
#define GBN_ERR_HIP            3
#define GBN_ERR_UNSUPPORTED    4
#define GBN_ERR_NOMEM          5
#define GBN_ERR_INTERRUPTED    6   /* BLASTERR_INTERRUPTED analogue */
#define GBN_ERR_SETUP          7   /* no valid Karlin-Altschul block etc. */
#define GBN_ERR_INTERNAL       8   /* an exception inside the library, caught at the boundary (text: gbn_last_error) */

/* lookup table kinds (ELookupTableType subset) */
#define GBN_LUT_SMALL_NA 1
#define GBN_LUT_NA       2
#define GBN_LUT_MB       3

/* ---- options: the blastn-relevant subset of BlastScoringOptions,
 * LookupTableOptions, BlastInitialWordOptions, BlastExtensionOptions,
 * BlastHitSavingOptions, BlastEffectiveLengthsOptions (COREI/blast_options.h) */
typedef struct GbnOptions {
    int32_t word_size;
    int32_t reward, penalty;
    int32_t gap_open, gap_extend;
    int32_t greedy;                 /* 1 = eGreedyScoreOnly, 0 = eDynProgScoreOnly */
    double  xdrop_ungap_bits;
    double  gap_trigger_bits;
    double  xdrop_gap_bits;
    double  xdrop_gap_final_bits;
    double  evalue;
    int32_t min_diag_separation;
    int32_t hitlist_size;
    int32_t cutoff_score;
    int32_t lut11_gblastn_rule;     /* CORE/blast_nalookup.c:127-144 vs stock NCBI */
    int64_t db_length;              /* GLOBAL database length (all shards) */
    int32_t db_num_seqs;            /* GLOBAL number of subjects; 0 = per-subject stats */
} GbnOptions;

void gbn_default_options(GbnOptions *opt, int megablast);

/* per query strand ("context", BlastContextInfo, COREI/blast_query_info.h:46-59) */
typedef struct GbnContext {
    int32_t query_offset, query_length, frame, query_index, is_valid;
    int32_t length_adjustment;
    int64_t eff_searchsp;
    double  lambda_u, K_u, logK_u, H_u;
    int32_t x_dropoff, cutoff_score, reduced_cutoff;
    int32_t gap_cutoff_score, gap_cutoff_score_max;
} GbnContext;

/* preliminary HSP (BlastHSP subset that the preliminary stage fills,
 * COREI/blast_hits.h:93-126) */
typedef struct GbnHSP {
    int32_t oid;                    /* GLOBAL subject ordinal id */
    int32_t context;
    int32_t q_offset, q_end, q_gapped_start;    /* context-relative */
    int32_t s_offset, s_end, s_gapped_start;
    int32_t score;
    int32_t pad_;
    double  evalue;
} GbnHSP;

typedef struct GbnSeed { int32_t oid, s_off, q_off, pad_; } GbnSeed;
typedef struct GbnInitHit {
    int32_t oid, q_off, s_off, q_start, s_start, length, score, pad_;
} GbnInitHit;

/* BlastDiagnostics subset (COREI/blast_diagnostics.h) */
/* kernel classes of GbnDiagnostics::kernel_ms */
#define GBN_KT_KEYS        0   /* seed keys (seed_keys / seed_ckeys / group_keys kernels) */
#define GBN_KT_SORT        1   /* the seeds put into the diagonal filter's order: radix sorts of the keys, or the seed_order kernels (which build the keys too) */
#define GBN_KT_SEED_EXT    2   /* seed_ext(_ck)_kernel + seed_exact_kernel: every seed extended */
#define GBN_KT_REPLAY      3   /* run_heads_kernel + diag_replay_kernel */
#define GBN_KT_DIAG        4   /* diag_ungapped_kernel (few seeds: filter + extension in one) */
#define GBN_KT_LANE_DP     5   /* gap_context_kernel + dynprog_lane_kernel */
#define GBN_KT_WAVE_DP     6   /* dynprog_wave_kernel (blastn) / greedy_wave_kernel (megablast) */
#define GBN_KT_THREAD_GAP  7   /* dynprog_kernel / greedy_kernel */
#define GBN_KT_N           8
typedef struct GbnDiagnostics {
    int64_t lookup_hits;            /* raw table hits before mini-extension */
    int64_t init_extends, good_init_extends;
    int64_t gapped_extensions, good_extensions, seqs_passed;
    int64_t seeds;                  /* after mini-extension */
    double  scan_kernel_ms;         /* HIP-event time of the scan kernel(s) */
    double  total_ms;               /* wall time of the whole call */
    int64_t scan_launches;
    int64_t subject_bases_scanned;
    /* partitioned scan, per kernel (HIP events on the engine's stream) */
    double  bin_kernel_ms, probe_kernel_ms, rare_kernel_ms;
    /* host wall time per stage (each ends at a stream synchronisation) */
    double  scan_stage_ms;          /* scan incl. launches, retries and counter read-back */
    double  seed_stage_ms;          /* key build, two radix sorts, diagonal filter + ungapped kernel */
    double  gapped_stage_ms;        /* gapped kernels + D2H of initial hits and extensions */
    double  host_stage_ms;          /* replay of the acceptance rules, HSP list rules */
    /* GPU time (HIP events on the stage's stream) per kernel class of the stages behind the scan, GBN_KT_* */
    double  kernel_ms[GBN_KT_N];
    /* what repeat-rich subjects cost (round 6): subject ranges searched (a range whose lookup words pile up in a few bins is halved:
     * more ranges), scans that had to be repeated (a stream or queue segment overflowed), ranges scanned by the direct-probe kernel,
     * launches of the library's radix sort (seed counts beyond what the engine's own sort kernels take) */
    int64_t ranges, scan_rescans, direct_ranges, library_sorts;
} GbnDiagnostics;

typedef int (*GbnInterruptFn)(void *progress);   /* TInterruptFnPtr analogue */

/* ---- process-level ---- */
/* One engine (streams, scratch, device pool) per GPU, made on first use.  gbn_init arms the engine of `gpu_id`
 * (< 0: the calling thread's current HIP device) and makes it the calling thread's device; the first one armed is
 * the default of threads that never chose.  gbn_use_device = the same for a worker thread: the GPU lease of the
 * reference's search threads (GB/gpu_blast_multi_gpu_utils.cpp:105-139 ThreadFetchGPU).  Objects remember their
 * device: gbn_db_new / gbn_blastdb_load_shard / gbn_batch_new* create on the calling thread's device, every other
 * call works on the device of the handles it is given, whatever thread makes it; a batch is searched against shards
 * of its own device only.  Searches on different devices run concurrently, calls on one device are serialised. */
int  gbn_init(int use_gpu, int gpu_id);
int  gbn_device_count(void);
int  gbn_use_device(int gpu_id);
int  gbn_current_device(void);              /* of the calling thread; -1: none armed yet */
void gbn_release(void);                     /* every engine: stages finished, device idle, buffers / streams / events freed */
void gbn_release_db_memory(void);   /* frees every shard held by the cache below (GB/gpu_blastn_na_ungapped_v3.h:21) */
int gbn_debug_cpu_account(double *ms, int n);     /* bench, GBN_CPU_ACCOUNT=1: CPU ms of the host threads by class (csrc/gbn_host.hpp GBN_CPU_*); returns the number of classes */
long gbn_debug_check_guards(void);  /* tests, GBN_GUARD=1: guard zones around every device block of the pool intact? */
/* shards kept per caller handle (the shim keys them by BlastSeqSrc*): the cache owns what is inserted */
struct GbnDb;
struct GbnDb *gbn_db_cache_find(const void *key);
int  gbn_db_cache_insert(const void *key, struct GbnDb *db);
/* The block cache (the shim's resident database: one shard per OID chunk of the sequence source, replacing the reference's
 * per-OID device cache GB/gpu_blastn_MB_and_smallNa.cu:1461-1467): keyed by (device, database name, the OIDs
 * themselves), whatever thread or query batch asks.  find: *out = the resident block on the calling thread's device or
 * NULL.  insert: the cache takes `db`; if the block is there already (built twice at the same time) `db` is freed and
 * *kept is the one that stays.  gbn_release_db_memory frees the blocks too. */
int  gbn_block_cache_find(const char *db_name, const int32_t *oids, int32_t n, struct GbnDb **out);
int  gbn_block_cache_insert(const char *db_name, const int32_t *oids, int32_t n, struct GbnDb *db, struct GbnDb **kept);
struct GbnDevSeed;
/* tests: the seed-order kernels (csrc/seed_order.hip) on segments of seeds given in HOST memory: keys_out[0 .. *n_out) =
 * the composite keys of the seeds ordered by (subject, slot), scan order inside (see GbnExtParams::ck_*) */
int gbn_debug_seed_order(const struct GbnDevSeed *seg, const uint32_t *seg_count, int nseg, uint32_t seg_cap, int nsubj, int subj_base,
                         int container_hash, int diag_len, int32_t qlen, int32_t max_len, int q_descending, uint64_t *keys_out, int64_t *n_out);
long long gbn_debug_bin_ahead_hits(void);       /* tests: passes of the calling thread's engine whose binning kernel had been queued by the pass before (GBN_BIN_AHEAD) */
long long gbn_debug_bin_ahead_misses(void);     /* ... and binning kernels queued ahead that no pass used */
/* The record cache of the calling thread's device.  The records the binning kernel writes (every scan position of a subject
 * range filed by the key range of its lookup word) depend on the shard, the range and the SHAPE of the lookup table, not
 * on the queries: complete sets stay resident, least recently used first out, and a pass whose set is there runs the probe
 * kernels only ("bin once, probe many").  The reference keeps what its scan needs of the database on the device for the
 * life of the process the same way: its per-OID subject cache, GB/gpu_blastn_MB_and_smallNa.cu:1461-1468.
 * limit: bytes the cache may hold; 0: off (every pass bins for itself); < 0: the default -- GBN_RECORD_CACHE_MB, else a
 * quarter of the device's memory.  stats: out[0..n) of {limit, bytes held, sets, passes served from the cache, passes
 * that binned, sets evicted, passes whose set is larger than the whole cache, passes served by a kernel queued ahead,
 * sets queued by gbn_db_prepare_records, sets in sorted form, their bytes, sorts done, passes over sorted records, GPU
 * microseconds of the last sort}.
 * A cached set that keeps being hit is SORTED BY CELL once (csrc/scan_runs.hip): a record shrinks to 16 subject bits + its
 * position id, the streams go back to the pool, and later passes read the runs of the cells their batch occupies and
 * nothing else -- the presence test in front of the table access (MB_ACCESS_HITS, CORE/blast_nascan.c:1413-1461) as a
 * decision about which bytes are fetched.  GBN_REC_RUNS=0: never; GBN_RUNS_AFTER=n: hits in stream form before the sort. */
int  gbn_record_cache_set_limit(long long bytes);
int  gbn_record_cache_stats(long long *out, int n);
int  gbn_record_cache_invalidate(void);     /* every cached set forgets its records (its buffers stay): the next pass of each key bins again */
/* The scan records a batch of nq unmasked queries of these lengths will want of `db`, binned NOW, asynchronously: the binning
 * kernel needs no lookup table, so it runs underneath the set-up of the batch (gbn_batch_new*), whose pass then finds the
 * records in the cache.  Returns at once.  No effect when the cache is off, when such a batch is scanned without records
 * (tables as wide as the word, tiny tables), when the records are resident, or while a search is running on the device
 * (that pass bins for itself: a pipelined caller's set-up thread never waits here). */
struct GbnOptions;
int  gbn_db_prepare_records(struct GbnDb *db, const struct GbnOptions *opt, int32_t nq, const int32_t *lens);
/* A VIEW over resident blocks: their subjects as ONE shard (one tile table, one launch per kernel, one record set), for
 * the shim's loop over OID chunks (GB/gpu_blastn_pre_search_engine.cpp:1243-1441 searches chunk after chunk inside one
 * call).  No subject byte is copied: the view addresses every block's slab from the lowest one.  Views are cached by their
 * blocks (any order; they are put into ascending OID order) and freed by gbn_release_db_memory / with any of their blocks;
 * n == 1 returns the block itself.  GBN_ERR_UNSUPPORTED: blocks with chunked sequences, or slabs more than 64 GiB apart
 * (the caller searches them one by one). */
int  gbn_block_view(struct GbnDb *const *blocks, int32_t n, struct GbnDb **out);
long long gbn_debug_db_bytes_uploaded(void);    /* tests: slab bytes copied host -> device by gbn_db_new / the shard builder so far */

/* ---- database shard resident in HBM ---- */
typedef struct GbnDb GbnDb;
/* `packed` holds the NCBI2na data of all subjects back to back: subject i
 * occupies bytes [byte_off[i], byte_off[i] + ceil(len[i]/4)).  byte_off must be
 * 16-byte aligned per subject and the buffer must extend 128 bytes past the
 * last subject.  is_device != 0: `packed` is already a device pointer (e.g.
 * a torch tensor) that stays owned by the caller. */
int  gbn_db_new(GbnDb **out, const uint8_t *packed, int64_t nbytes, int32_t num_seqs,
                const int64_t *byte_off, const int32_t *len, int32_t first_oid,
                int is_device);
/* The same shard, its subject bytes produced piece by piece: the library allocates the slab on the device and calls
 * fill(ctx, first, count, dst, dst_base) from `threads` worker threads (0: a default) for runs of consecutive sequences -- the
 * callback writes sequence i's NCBI2na bytes at dst + (byte_off[i] - dst_base), for i in [first, first + count); dst is pinned
 * host memory, zeroed, and goes to the device asynchronously while other pieces are being filled (no 12.5 GB host copy of a
 * 50 Gbp shard, no blocking copy from pageable memory).  gbn_blastdb_load_shard reads the volumes' mapped .nsq files this way.
 * GBN_ERR_UNSUPPORTED: a sequence longer than MAX_DBSEQ_LEN (held as chunk copies: gbn_db_new).  fill returns 0 or a status. */
typedef int (*GbnFillFn)(void *ctx, int32_t first, int32_t count, uint8_t *dst, int64_t dst_base);
int  gbn_db_new_streamed(GbnDb **out, int64_t nbytes, int32_t num_seqs, const int64_t *byte_off, const int32_t *len,
                         int32_t first_oid, GbnFillFn fill, void *ctx, int threads);
void gbn_db_free(GbnDb *db);
/* Ambiguity runs of sequence `local` (0-based in the shard; values in NCBI4na as gbn_blastdb_get_ambiguities gives
 * them): the slab holds 2 bits per base, the traceback stage puts these codes back before it aligns, as the
 * reference fetches the subject with its ambiguities there (CORE/blast_traceback.c:1375-1639).
 * gbn_blastdb_load_shard attaches them by itself. */
int  gbn_db_set_ambiguities(GbnDb *db, int32_t local, int32_t n, const int32_t *start, const int32_t *length,
                            const uint8_t *ncbi4na);
/* Sequences longer than MAX_DBSEQ_LEN are searched in chunks of that length overlapping by DBSEQ_CHUNK_OVERLAP (100),
 * the chunks' HSP lists merged (CORE/blast_engine.c:218-262, :455-540; Blast_HSPListsMerge CORE/blast_hits.c:2545):
 * a shard made after this call holds such sequences as chunk copies, everything a caller sees stays in sequence
 * coordinates.  Default 200,000,000 = G-BLASTN's build (COREI/blast_gapalign.h:54-55; stock BLAST+: 5,000,000);
 * a multiple of 4, >= 1000 (tests lower it). */
int  gbn_set_max_dbseq_len(int32_t max_len);
/* the same shard from subjects handed over one at a time, as BlastSeqSrcGetSequence yields them (NCBI2na,
 * `length` bases; OIDs = order of the calls); _finish uploads the slab and leaves the builder empty */
typedef struct GbnShardBuilder GbnShardBuilder;
int  gbn_shard_builder_new(GbnShardBuilder **out, int32_t expected_seqs);
int  gbn_shard_builder_add(GbnShardBuilder *b, const uint8_t *ncbi2na, int32_t length);
/* ... with the subject's OID given (ascending; a BlastSeqSrc iterator over an OID list or a GI filter leaves holes,
 * BlastSeqSrcIteratorNext COREI/blast_seqsrc.h:438): the HSPs carry these OIDs */
int  gbn_shard_builder_add_oid(GbnShardBuilder *b, int32_t oid, const uint8_t *ncbi2na, int32_t length);
int  gbn_shard_builder_finish(GbnShardBuilder *b, GbnDb **out);
void gbn_shard_builder_free(GbnShardBuilder *b);
int64_t gbn_db_total_bases(const GbnDb *db);
int  gbn_db_device(const GbnDb *db);       /* the GPU the shard is resident on */
int32_t gbn_db_num_seqs(const GbnDb *db);
/* deterministic synthetic DB bytes generated on the device (bench/tests) */
int  gbn_synth_fill(void *dev_ptr, int64_t nbytes, uint64_t seed, void *stream);
/* Repeats over a synthetic shard of `num` subjects of nb packed bytes each, `stride` bytes apart from first_off (bench.py --skew):
 * per subject four stretches of nb / 50 bytes -- homopolymer runs or tandem repeats of a short unit -- and in one subject of fifty
 * a copy of one family element of 1,200 bases; gblastn_amd/synth.py: skew_subject is its numpy form. */
int  gbn_synth_skew(void *dev_ptr, int64_t first_off, int64_t stride, int64_t nb, int32_t num, int64_t first_oid, uint64_t seed, void *stream);

/* ---- BLAST database files (format version 4, nucleotide): alias (.nal), index (.nin), sequence
 * (.nsq) incl. ambiguity runs.  Replaces what the reference reaches through its SeqDB BlastSeqSrc
 * (API/seqsrc_seqdb.cpp:283-382: s_SeqDbGetSequence, GetNumSeqs, GetTotLen, GetMaxLength, GetSeqLen);
 * formats: objtools/blast/seqdb_reader/index_files.txt:62-120, sequence_files.txt:60-170,
 * alias_files.txt.  `name` is the path without extension; NAME.nal takes precedence over NAME.nin.
 * Alias filters (GILIST, OIDLIST, ...) are refused with GBN_ERR_UNSUPPORTED.  Host only. ---- */
typedef struct GbnBlastDb GbnBlastDb;
int  gbn_blastdb_open(GbnBlastDb **out, const char *name);
void gbn_blastdb_close(GbnBlastDb *db);
int32_t gbn_blastdb_num_volumes(const GbnBlastDb *db);
int32_t gbn_blastdb_num_seqs(const GbnBlastDb *db);
int64_t gbn_blastdb_total_length(const GbnBlastDb *db);
int32_t gbn_blastdb_max_length(const GbnBlastDb *db);
/* database size for the statistics: NSEQ / LENGTH of the alias file when given, else the sums */
int32_t gbn_blastdb_stat_num_seqs(const GbnBlastDb *db);
int64_t gbn_blastdb_stat_length(const GbnBlastDb *db);
const char *gbn_blastdb_title(const GbnBlastDb *db);
int  gbn_blastdb_volume_range(const GbnBlastDb *db, int32_t vol, int32_t *first_oid, int32_t *num_oids);
int32_t gbn_blastdb_seq_length(const GbnBlastDb *db, int32_t oid);      /* < 0: bad oid */
int  gbn_blastdb_get_ncbi2na(const GbnBlastDb *db, int32_t oid, uint8_t *dst, int64_t dst_bytes);
int32_t gbn_blastdb_num_ambiguities(const GbnBlastDb *db, int32_t oid);
int  gbn_blastdb_get_ambiguities(const GbnBlastDb *db, int32_t oid, int32_t *start, int32_t *length,
                                 uint8_t *ncbi4na_value, int32_t cap);
/* one BLASTNA code per base, ambiguities applied; sentinels != 0: code 15 in front and behind */
int  gbn_blastdb_get_blastna(const GbnBlastDb *db, int32_t oid, uint8_t *dst, int64_t dst_bytes, int sentinels);
/* subjects [first_oid, first_oid + num_oids) -> resident HBM shard with global OIDs */
struct GbnDb;
int  gbn_blastdb_load_shard(const GbnBlastDb *db, int32_t first_oid, int32_t num_oids, struct GbnDb **out);

/* ---- query batch: host set-up (concatenation, Karlin-Altschul parameters, cut-offs, table kind) + lookup
 * structures built on the device from the uploaded query (a 5 Mb megablast batch: ~5 ms in all).  Thread
 * safe next to a running search: the builder has its own stream, its memory comes from a pool.  The
 * reference builds these tables on the host in LookupTableWrapInit (CORE/lookup_wrap.c:52-174). ---- */
typedef struct GbnBatch GbnBatch;
/* seqs[i]: BLASTNA codes (0..15), plus strand, lens[i] bases */
int  gbn_batch_new(GbnBatch **out, const GbnOptions *opt, int32_t nq,
                   const uint8_t *const *seqs, const int32_t *lens);
/* upload == 0 builds the host-side set-up only (no device needed) */
int  gbn_batch_new_ex(GbnBatch **out, const GbnOptions *opt, int32_t nq,
                      const uint8_t *const *seqs, const int32_t *lens, int upload);
/* Soft query masks ("mask at hash": DUST / lower-case masks as blastn applies them by default,
 * BlastMaskLoc -> lookup_segments, CORE/blast_setup.c BLAST_MainSetUp / CORE/blast_filter.c:1019-1119):
 * nmask intervals [from, to] (inclusive, plus-strand coordinates of query mask_query[k]) sorted by
 * (query, from) and disjoint.  Masked stretches are not indexed in the lookup table, and every seed
 * is re-checked against the table (s_TypeOfWord, CORE/na_ungapped.c:488-587); the extensions see the
 * unmasked query.  Hard masking = write the code for N (14) into the query instead. */
int  gbn_batch_new_masked(GbnBatch **out, const GbnOptions *opt, int32_t nq,
                          const uint8_t *const *seqs, const int32_t *lens,
                          int32_t nmask, const int32_t *mask_query, const int32_t *mask_from,
                          const int32_t *mask_to, int upload);
/* Symmetric DUST low-complexity intervals of a BLASTNA sequence (CSymDustMasker,
 * src/algo/dustmask/symdust.cpp:40-287, as blastn's default query filter uses it through
 * Blast_FindDustFilterLoc, API/dust_filter.cpp:60-170: level 20, window 64, linker 1); intervals are
 * inclusive, ascending and fused when overlapping or abutting.  Returns their number; at most cap are
 * written.  Host only. */
int32_t gbn_dust_mask(const uint8_t *seq, int32_t len, int32_t level, int32_t window, int32_t linker,
                      int32_t *from, int32_t *to, int32_t cap);
void gbn_batch_free(GbnBatch *b);
int32_t gbn_batch_num_contexts(const GbnBatch *b);
const GbnContext *gbn_batch_contexts(const GbnBatch *b);
/* gapped Karlin-Altschul parameters of the batch's scoring system (Blast_KarlinBlkNuclGappedCalc,
 * CORE/blast_stat.c:3718-3810): bit score = (lambda * score - ln K) / ln 2 */
int  gbn_batch_karlin_gapped(const GbnBatch *b, double *lambda, double *K);
int32_t gbn_batch_lut_type(const GbnBatch *b);
int32_t gbn_batch_lut_width(const GbnBatch *b);
int32_t gbn_batch_scan_step(const GbnBatch *b);
int32_t gbn_batch_scan_path(const GbnBatch *b);         /* which kernels scan for this batch: 0 scan_bin + probe_bin + probe_rare (key-range partitioned), 1 scan_seed (direct probing), 2 scan_slice (presence bits sliced through the LDS) */
int32_t gbn_batch_diag_container(const GbnBatch *b);    /* 0 array, 1 hash */
int32_t gbn_batch_gap_x_dropoff(const GbnBatch *b);

/* ---- result list (the HSP stream analogue): caller-owned, callee-filled ---- */
typedef struct GbnResults GbnResults;
int  gbn_results_new(GbnResults **out);
void gbn_results_free(GbnResults *r);
void gbn_results_clear(GbnResults *r);
int64_t gbn_results_num_hsps(const GbnResults *r);
const GbnHSP *gbn_results_hsps(const GbnResults *r);     /* grouped by oid ascending */
int64_t gbn_results_num_seeds(const GbnResults *r);
const GbnSeed *gbn_results_seeds(const GbnResults *r);   /* only if keep_stages */
int64_t gbn_results_num_init_hits(const GbnResults *r);
const GbnInitHit *gbn_results_init_hits(const GbnResults *r);

/* ---- the preliminary search over one resident shard ---- */
int  gbn_prelim_search(GbnBatch *batch, GbnDb *db, GbnResults *results,
                       GbnDiagnostics *diag, int keep_stages,
                       GbnInterruptFn interrupt, void *progress);
/* The same search delivered as the reference's HSP stream takes it (BlastHSPStreamWrite, CORE/blast_hspstream.c:
 * one BlastHSPList per subject): `sink` is called once per subject that has HSPs, oids ascending, with that
 * subject's HSPs in list order (score descending as Blast_HSPListSortByScore leaves them); a non-zero return
 * of the sink stops the delivery and fails the call.  The pointers are valid during the call only. */
typedef int (*GbnHspListFn)(void *arg, int32_t oid, const GbnHSP *hsps, int32_t n);
int  gbn_prelim_search_lists(GbnBatch *batch, GbnDb *db, GbnHspListFn sink, void *sink_arg, GbnDiagnostics *diag,
                             GbnInterruptFn interrupt, void *progress);
/* Pipelined form (the reference's "-mode 1" PrelimSearchThread / TraceBackThread split,
 * GB/work_thread.cpp:60-107, moved one stage down): _begin returns once the scan of the LAST subject
 * range is done.  For ranges with few seeds (megablast shapes) everything after the scan -- seed order,
 * diagonal filter, ungapped and gapped extension, host-side acceptance -- runs on a second HIP stream and
 * a host thread while the caller starts the next query batch; with many seeds (blastn shapes) the seed
 * stage stays on the engine's stream and the gapped stage is asynchronous.  `batch`, `results` and `diag`
 * must stay alive and untouched until _end(results) has returned (it returns at once if a later _begin
 * has already waited for them; gbn_batch_free waits by itself).  At most one such stage is in flight. */
int  gbn_prelim_search_begin(GbnBatch *batch, GbnDb *db, GbnResults *results, GbnDiagnostics *diag,
                             GbnInterruptFn interrupt, void *progress);
int  gbn_prelim_search_end(GbnResults *results);   /* NULL: whatever is in flight */
/* finished results as gbn_prelim_search_lists delivers them: one call of `sink` per subject that has HSPs, oids ascending
 * (a pipelined caller writes the lists of search k to the HSP stream while search k + 1 scans) */
int  gbn_results_emit_lists(const GbnResults *results, GbnHspListFn sink, void *sink_arg);
int  gbn_debug_counting_sink(void *arg, int32_t oid, const GbnHSP *hsps, int32_t n);   /* bench / tests: a GbnHspListFn that counts; arg = long long[2] {lists, HSPs} */
/* scan stage only (bench / roofline): runs the scan+seed kernel over the
 * whole shard `repeats` times and reports the HIP-event time per launch */
int  gbn_scan_only(GbnBatch *batch, GbnDb *db, int repeats, GbnDiagnostics *diag);

/* ---- per-query top-N hit lists: the collector writer of the HSP stream
 * (replaces BlastHSPStreamWrite CORE/blast_hspstream.c:316-365 ->
 * s_BlastHSPCollectorRun CORE/hspfilter_collector.c:86-170 -> Blast_HitListUpdate
 * CORE/blast_hits.c:2924-2981; read-out order of BlastHSPStreamClose/Read
 * CORE/blast_hspstream.c:136-209,232-300).  Host only.  With several shards, rank 0
 * writes the gathered records of all shards in ascending oid order. ---- */
typedef struct GbnCollector GbnCollector;
int32_t gbn_prelim_hitlist_size(int32_t hitlist_size);      /* min(2N, N+50), >= 10 */

/* ---- host pipeline: set-up -> preliminary search -> traceback on their own threads --------------------------
 * (GB/work_thread.cpp:60-156, APP/blastn_app.cpp:725-989 "Method2"; the C++ classes are in gblastn_amd_host.hpp).
 * Query batches are submitted, finished batches come back in submission order. */
typedef struct GbnPipeline GbnPipeline;
int  gbn_pipeline_new(GbnPipeline **out, const GbnOptions *opt, GbnDb *db, int32_t trace_threads, int with_traceback, int overlap);
void gbn_pipeline_free(GbnPipeline *p);
int64_t gbn_pipeline_submit(GbnPipeline *p, int32_t nq, const uint8_t *const *seqs, const int32_t *lens,
                            int32_t nmask, const int32_t *mask_query, const int32_t *mask_from, const int32_t *mask_to);
void gbn_pipeline_finish(GbnPipeline *p);       /* no more batches will be submitted */
struct GbnTraceback; struct GbnCollector;
/* 0: a batch (its number in *id; results valid until the next call), 1: none left, < 0: that batch failed */
int  gbn_pipeline_next(GbnPipeline *p, int64_t *id, const struct GbnTraceback **tb, const struct GbnCollector **col);
int  gbn_pipeline_diagnostics(const GbnPipeline *p, GbnDiagnostics *d);

/* ---- traceback stage (host): CORE/blast_traceback.c:336-790, :1375-1639 -------------------------------------
 * Final alignments of the (subject, query) lists the collector kept: gapped extension with the final X-drop and
 * an edit script (ALIGN_EX / greedy with traceback), list rules, identities, e-values, bit scores; per query the
 * subjects in the order of the reference's results (best e-value, best score, oid), at most hitlist_size.
 * The reference runs this stage on the CPU as well; in pipeline mode it is the consumer next to the GPU's
 * preliminary stage (GB/work_thread.cpp:86-107). */
typedef struct GbnTbHSP {
    GbnHSP  hsp;                    /* final coordinates (context-relative query), score, e-value */
    int32_t num_ident, align_length, gaps, gap_opens;
    int64_t ops_first; int32_t ops_count, pad_;     /* edit script: ops[ops_first .. +ops_count): 0 gap in query, 3 aligned pair, 6 gap in subject */
    double  bit_score;
} GbnTbHSP;
typedef struct GbnTraceback GbnTraceback;
int  gbn_traceback_new(GbnTraceback **out);
void gbn_traceback_free(GbnTraceback *t);
/* hsps / list_start: as gbn_collector_hsps / gbn_collector_list_starts give them (nlists + 1 offsets) */
int  gbn_traceback_run(GbnBatch *batch, GbnDb *db, const GbnHSP *hsps, const int64_t *list_start, int64_t nlists,
                       int32_t threads, GbnTraceback *out);
int64_t gbn_traceback_num_hsps(const GbnTraceback *t);
const GbnTbHSP *gbn_traceback_hsps(const GbnTraceback *t);
const uint8_t *gbn_traceback_ops(const GbnTraceback *t);
const int32_t *gbn_traceback_op_lengths(const GbnTraceback *t);
const int64_t *gbn_traceback_query_starts(const GbnTraceback *t);      /* [num_queries + 1] offsets into hsps */
/* The final results of several shards of ONE database (a gbn_traceback_run per shard, same query batch, global
 * statistics) -> the results of the whole database: per query the subjects of all parts in the order above, at most
 * hitlist_size.  Edit scripts stay with the parts (ops_first / ops_count cleared).  `out`: room for the sum of the
 * parts' HSPs; out_query_start: nq + 1 offsets.  Returns the number of HSPs written, < 0 on a bad argument. */
int64_t gbn_traceback_merge(int32_t nparts, const GbnTbHSP *const *hsps, const int64_t *const *query_start, int32_t nq,
                            int32_t hitlist_size, GbnTbHSP *out, int64_t *out_query_start);

int  gbn_collector_new(GbnCollector **out, int32_t num_queries, int32_t hitlist_size);
void gbn_collector_free(GbnCollector *c);
/* records grouped by oid, each group sorted by score (what gbn_results_hsps yields) */
int  gbn_collector_write(GbnCollector *c, const GbnHSP *hsps, int64_t n);
int  gbn_collector_close(GbnCollector *c);
/* after close: surviving (query, oid) lists in (oid, query) ascending order */
int64_t gbn_collector_num_lists(const GbnCollector *c);
const int64_t *gbn_collector_list_starts(const GbnCollector *c);   /* num_lists + 1 offsets into hsps */
const int32_t *gbn_collector_list_queries(const GbnCollector *c);
int64_t gbn_collector_num_hsps(const GbnCollector *c);
const GbnHSP *gbn_collector_hsps(const GbnCollector *c);

/* ---- thin kernel launchers: device pointers in the parameter blocks of gblastn_amd_kernels.h (public,
 * self-contained), hipStream_t passed as void*.  gbn_batch_*_params fill in what a batch and a shard determine
 * (waiting for the batch's deferred lookup build); the work list and the output / scratch buffers marked
 * [caller] in that header are the caller's.  gbn_batch_diag_layout: the diagonal container the batch's
 * word finder uses (CORE/blast_extend.c:85-140) -- what a caller needs to order seeds for gbn_launch_ungapped. ---- */
struct GbnScanParams; struct GbnExtParams; struct GbnGapParams;
int  gbn_batch_scan_params(const GbnBatch *b, const GbnDb *db, struct GbnScanParams *out);
int  gbn_batch_ext_params(const GbnBatch *b, const GbnDb *db, struct GbnExtParams *out);
int  gbn_batch_gap_params(const GbnBatch *b, const GbnDb *db, struct GbnGapParams *out);
int  gbn_batch_diag_layout(const GbnBatch *b, int32_t *container_hash, int32_t *diag_len, int32_t *q_descending);
int  gbn_launch_scan_seed(const struct GbnScanParams *p, int grid, void *stream);
int  gbn_launch_ungapped(const struct GbnExtParams *p, void *stream);
int  gbn_launch_gapped(const struct GbnGapParams *p, int greedy, void *stream);

const char *gbn_last_error(void);

/* The processors this process may use at a time: hardware threads, cut down to the affinity mask and to the cgroup's CPU quota
 * (a container that shows 256 hardware threads may grant 16 CPUs; a process that runs more busy threads than that is stopped by the
 * scheduler for the rest of every 100 ms period).  The library sizes its own host pools from this; callers that make threads of
 * their own (traceback consumers, DUST) should too.  GBN_HOST_CPUS=n overrides.  No reference counterpart (blastn's -num_threads
 * is the caller's number). */
int32_t gbn_host_cpus(void);
/* Threads the per-context loops of the batches THE CALLING THREAD sets up from now on are spread over (Karlin-Altschul parameters,
 * effective lengths, cut-offs of 10,000 contexts: 14 ms of CPU per 5 Mb batch); 0 = the library's pool in full (the default: the CPUs
 * granted).  For callers that set batches up in the background of other CPU work: CSearchPipeline's set-up threads ask for half. */
void gbn_set_setup_threads(int32_t n);

#ifdef __cplusplus
}
#endif
#endif
