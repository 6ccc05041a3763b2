// engine.hpp -- internal: the device context (Engine) of the preliminary search and what the translation units of the
// engine share: engine.cpp (devices, the memory pool, query batches on the device, tile tables), engine_scan.cpp (the scan
// of a subject range, the record cache), engine_stages.cpp (seed order, diagonal filter, extension stages, host replay),
// engine_abi.cpp (the C ABI of include/gblastn_amd.h).  Round 5 cut the 2,700 lines of engine.cpp along these seams.
// Product code: never includes anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include "gbn_host.hpp"
#include "gbn_guard.hpp"
#include <memory>
#include "lutbuild.h"
#include "gbn_dev.h"
#include "hsp_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <future>
#include <thread>
#include <mutex>
#include <tuple>
#include <cstdlib>

namespace gbn {
// ---- kernel launchers (kernels.hip, scan_bin.hip, seed_order.hip, seed_sort.hip, gapped.hip)
hipError_t launch_scan_seed(const GbnScanParams &p, int grid, hipStream_t st);
hipError_t launch_scan_bin(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev);
hipError_t launch_scan_bin_parts(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev, int parts, hipEvent_t tables_ready);
// scan_runs.hip: the sorted form of a record set
int runs_choose_sbits(int64_t npos, int nb, int cbits);
int runs_choose_wgroup(int64_t npos, int nb, int nwriters);
hipError_t launch_runs_build(const GbnRunsBuild &R, void *scan_tmp, size_t scan_tmp_bytes, hipStream_t st);
hipError_t launch_probe_runs(const GbnBinParams &b, int grid, hipStream_t st, hipEvent_t *ev, hipEvent_t tables_ready);
hipError_t launch_seed_keys(const GbnKeyParams &k, hipStream_t st);
hipError_t launch_group_keys(const GbnKeyParams &k, hipStream_t st);
hipError_t launch_seed_ckeys(const GbnKeyParams &k, hipStream_t st);
hipError_t launch_seed_order(const GbnKeyParams &k, int nsubj, uint32_t *scratch, hipStream_t st);       // seed_order.hip
bool seed_sort_small_fits(const GbnKeyParams &K, int nsubj);                                              // seed_sort.hip
hipError_t launch_seed_sort_small(const GbnKeyParams &K, int nsubj, uint32_t *idx_out, uint32_t *idx_tmp, uint64_t *key_group_out, uint64_t *key_tmp, hipStream_t st);
size_t seed_order_scratch_words(int64_t n, int nsubj, int group_bits);
hipError_t launch_diag_ungapped(const GbnExtParams &p, hipStream_t st, GbnKernelTimer *kt = nullptr);
hipError_t launch_gapped(const GbnGapParams &p, bool greedy, hipStream_t st, GbnKernelTimer *kt = nullptr);
hipError_t launch_synth_fill(void *dev, int64_t nbytes, uint64_t seed, hipStream_t st);
hipError_t launch_synth_skew(void *dev, int64_t first_off, int64_t stride, int64_t nb, int32_t num, int64_t first_oid, uint64_t seed, hipStream_t st);
hipError_t launch_gather_bytes(const uint8_t *src, const int64_t *src_off, const int64_t *dst_off, const int32_t *nbytes,
                               int32_t n, uint8_t *dst, hipStream_t st);
hipError_t sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout,
                          const uint32_t *vin, uint32_t *vout, int64_t n, int end_bit, hipStream_t st);
int scan_slice_count(const GbnScanParams &p);
int scan_slice_blocks(const GbnScanParams &p, int num_cu);
hipError_t launch_scan_slice(const GbnScanParams &p, int num_cu, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                             unsigned long long *seg_max, hipStream_t st, const GbnKeyParams *keys = nullptr);
int scan_slice_segments(const GbnScanParams &p, int num_cu, int *ordered);
hipError_t launch_seed_compact(const GbnDevSeed *seg, const uint32_t *seg_count, unsigned long long *seg_first, int nseg, uint32_t seg_cap,
                               GbnDevSeed *out, unsigned long long out_cap, hipStream_t st, const GbnKeyParams *keys = nullptr);
hipError_t sort_keys_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, int64_t n, int begin_bit, int end_bit, hipStream_t st);


const std::string &last_error_text();               // the calling thread's error text (gbn_last_error)
// Stages of one search run on several threads (the inline seed stage, the stage in flight on the second stream, the
// detached host replays) and all add to the caller's GbnDiagnostics: every such update holds this lock.
extern std::mutex g_diag_mu;
#define GBN_DIAG_LOCKED(stmt) do { std::lock_guard<std::mutex> dl_(gbn::g_diag_mu); stmt; } while (0)

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return GBN_ERR_HIP; } } while (0)

constexpr int kCtxHintShift = 6;
struct DeviceBatch {
    uint8_t *q8_base = nullptr;     // device copy of qbuf
    const uint8_t *q8 = nullptr;    // q8_base + qpad
    uint8_t *q2_base = nullptr, *qinv_base = nullptr;   // 2-bit packed query + "matches nothing" bitmap
    const uint8_t *q2 = nullptr, *qinv = nullptr;       // ... at base 0 (256 bases of padding either side)
    uint8_t *q4_base = nullptr; int64_t q4_plane = 0;   // four bases per byte at every offset, in four planes by offset mod 4 (lut_q4_kernel)
    uint32_t *pv = nullptr, *cellw = nullptr, *cell_start = nullptr, *cellt = nullptr, *side_start = nullptr;
    uint16_t *sidet = nullptr;
    unsigned long long *ent = nullptr;
    uint32_t *pvx = nullptr, *pstart = nullptr;         // rank form of pv / cell_start for the folded slice scan (lut_rank_fill)
    int32_t *ctx_off = nullptr, *ctx_len = nullptr, *ctx_xdrop = nullptr, *ctx_cutoff = nullptr,
            *ctx_reduced = nullptr, *ctx_hint = nullptr, *ctx_blk = nullptr, *ctx_pack = nullptr;   // ctx_pack[4 c ..]: x_dropoff, reduced cut-off, cut-off of context c in one 16-byte read    // ctx_hint[q >> kCtxHintShift]: the context position (q & ~mask) lies in
    int32_t *matrix = nullptr, *score_table = nullptr;
    int32_t *ctx_block = nullptr;       // the one allocation ctx_off ... ctx_pack, ctx_hint, ctx_blk, matrix and score_table point into
    int mode = 0, fl = 0, fr = 0;
    // lookup structures still being built on the builder's stream: the event they are complete at, and the
    // builder's scratch, which goes back to the pool once it has fired
    hipEvent_t ready = nullptr; std::vector<void *> build_scratch;
    // ... and the event behind the copy of the contexts' cut-offs, which are computed while the tables are being built and go up
    // last (upload_batch_contexts): the extension kernels read them, the engine's stream waits for it behind a scan's kernels
    hipEvent_t ready_ctx = nullptr;
    size_t stage_seg = 0, stage_ctx = 0, stage_cut = 0;     // where the stretches, the per-context block and (a copy of its own) the cut-offs sit in it
    void *stage = nullptr; size_t stage_cap = 0;    // pinned host copy of the query the upload reads (stage_get): back to the engine with the builder's scratch
    struct Engine *eng = nullptr;   // the device context the batch lives on
};

struct Engine {
    bool ready = false; int device = -1; hipStream_t stream = nullptr;
    int num_cu = 256;
    // growable scratch
    GbnDevSeed *seeds = nullptr; size_t seed_cap = 0;
    // sort keys, run heads, container scratch and extension records of a range's seed stage.  Two sets: the second half
    // of the stage (extension + replay) of range k runs on the second stream next to the scan of range k + 1, whose sort
    // fills the other set (search_range)
    struct KeySet {
        uint64_t *key_a = nullptr, *key_b = nullptr; uint32_t *idx_a = nullptr, *idx_b = nullptr;
        int32_t *cell_diag = nullptr, *cell_level = nullptr; size_t key_cap = 0;
        int32_t *ext_rec = nullptr;     // 8 ints per seed: seed_ext_kernel -> diag_replay_kernel
        void *sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
        GbnKernelTimer kt;              // GPU time per kernel class of the seed stage that works on this set
    } ks[2];
    GbnKernelTimer kt_gap[2];       // ... of the gapped stage, per slot
    int pending_ks = -1;            // the set the stage in flight works on (-1: none)
    // initial hits / gapped extensions / gapped scratch exist twice: the gapped stage of one range
    // (stream2 + a host thread) overlaps the scan of the next range or query batch
    GbnDevInitHit *ihits_s[2] = {nullptr, nullptr}; GbnDevGapped *gapped_s[2] = {nullptr, nullptr}; size_t ihit_cap_s[2] = {0, 0};
    int32_t *gap_scratch_s[2] = {nullptr, nullptr}; size_t gap_scratch_ints_s[2] = {0, 0};
    int slot = 0; hipStream_t stream2 = nullptr;
    hipStream_t stream_build = nullptr;      // lookup structures of the next query batch are built next to a running search
    // one batch's build is queued as a whole: two set-up threads that queue theirs at the same time would interleave their
    // kernels on the builder's stream, and the batch that is searched first would have its tables when BOTH builds are done
    // (round 5, seen in the trace of the cold config: the first probe kernel started 1.2 ms late)
    std::mutex build_mu;
    std::future<int> pending; bool has_pending = false; std::string pending_err; const GbnResults *pending_res = nullptr;
    // what the stage in flight works on, readable WITHOUT the engine's lock: gbn_prelim_search_end / gbn_batch_free / gbn_results_free
    // of a pass whose stage is not the one in flight (a pipelined caller: the pass after it has been begun) must not wait for the
    // lock, which the caller's other thread holds for the length of a scan -- three such waits per pass kept a collecting thread
    // busier than the scans it was collecting behind (tools/step_jitter.py)
    std::atomic<const void *> pending_res_pub{nullptr}, pending_batch_pub{nullptr};
    // host replays of finished gapped stages, one after the other in the order they were queued (each waits for its
    // predecessor): the stage's thread hands its copies over and is free for the next range's kernels
    std::shared_future<void> host_tail; std::mutex host_mu, failed_mu;
    // stages that failed, by the results they were filling: reported by gbn_prelim_search_end for THOSE results
    std::map<const GbnResults *, std::pair<int, std::string>> failed;
    const GbnBatch *pending_batch = nullptr;   // the batch the stage in flight reads (its device memory must outlive the stage)
    unsigned long long *counters = nullptr;     // [0] seeds, [1] raw hits, [2] init hits, [3] runs; [4], [5]: init hits, runs of an asynchronous seed stage
    GbnDevSeed *slice_seg = nullptr; size_t slice_seg_cap = 0;        // scan_slice_kernel: the workgroups' seed segments
    bool seg_valid = false; int seg_n = 0; uint32_t seg_len = 0;       // the last scan left its seeds there (seg_n segments of seg_len slots, counts in seg_counts), not in `seeds`
    bool seg_ordered = false;       // ... and the segments read one after the other are in scan order (subject, position, entry)
    bool seg_keys = false; GbnKeyParams seg_key_params;     // ... and they hold 8-byte composite keys made with these parameters, not seeds (round 6)
    bool want_key_seeds = false;    // the range being scanned may leave its seeds as keys (set by search_range: not with keep_stages)
    uint32_t *seg_counts = nullptr; unsigned long long *seg_firsts = nullptr;     // GBN_SLICE_SEGS counts / + 1 prefix sums (scratch of the consumers)
    GbnDevSeed *seeds_async = nullptr; size_t seeds_async_cap = 0;     // the seeds an asynchronous seed stage works on
    hipEvent_t ev_seed = nullptr;
    GbnRareItem *rareq = nullptr; size_t rareq_cap = 0; uint32_t *rare_counts = nullptr;   // rare-path queue (the probe kernel's output: per query batch)
    // what the host reads after a scan, in ONE pinned block filled by asynchronous copies behind the kernels (round 4: the
    // counters, the overflow word and the rare-path counts came back through three blocking copies to pageable memory)
    struct ScanBack { unsigned long long cnt[2], seg_max; uint32_t overflow, pad_; uint32_t rare_counts[2048]; } *scan_back = nullptr;
    // ---- the scan records: what the binning kernel writes.  They depend on the shard, the subject range and the SHAPE of
    // the lookup table (lut width, stride, bins, fingerprint widths, stream geometry) -- not on the queries.  A RecordSet is
    // the three buffers of one such key.
    struct RecKey { const void *db = nullptr; int32_t s0 = 0, s1 = 0; int lut = 0, step = 0, nb = 0, nwriters = 0, rfl = 0, rfrbits = 0, cbits = 0;
                    const void *tiles = nullptr; size_t subcap = 0;
                    bool same_shape(const RecKey &o) const {      // everything but the streams' capacity
                        return db == o.db && s0 == o.s0 && s1 == o.s1 && lut == o.lut && step == o.step && nb == o.nb && nwriters == o.nwriters &&
                               rfl == o.rfl && rfrbits == o.rfrbits && cbits == o.cbits && tiles == o.tiles; }
                    bool operator==(const RecKey &o) const { return same_shape(o) && subcap == o.subcap; } };
    struct RecordSet { unsigned long long *bin_rec = nullptr; size_t bin_rec_cap = 0;      // records (all bins)
                       uint32_t *bin_tcur = nullptr; size_t bin_tcur_cap = 0;              // per-run stream cursors (6-byte records)
                       uint32_t *bin_count = nullptr; size_t bin_count_cap = 0;            // [nb][nwriters] + overflow flag (4 words) + [nb][nwriters] uncapped totals
                       // streams of bins that differ in size (repeat-rich subjects): {capacity, offset in a writer's row} per bin on the
                       // device, records per row; null: uniform streams of key.subcap records.  Made from the totals of an attempt that
                       // overflowed (run_scan_impl) and kept with the set: the next binning of this key starts with them.
                       uint32_t *bin_caps = nullptr; size_t row_records = 0; int bin_caps_nb = 0;
                       RecKey key; bool complete = false;      // the buffers hold every record of `key` (binned, no stream overflowed)
                       bool queued = false;                    // the binning kernel that writes them is queued on the engine's stream, its overflow flag not read yet (gbn_db_prepare_records)
                       unsigned long long stamp = 0;           // last use (record cache: least recently used goes first)
                       // the sorted form (scan_runs.hip): the records of a cell consecutive, a record = 16 subject bits +
                       // its position id.  Built from the complete streams once the set has served `runs_after` passes; the streams
                       // (bin_rec, bin_tcur) go back to the pool then, and `complete` with them.
                       uint16_t *run_fp = nullptr; uint32_t *run_pos = nullptr; uint32_t *run_start = nullptr; size_t run_n = 0, run_cells = 0;
                       bool runs = false, runs_failed = false; int hits = 0;
                       bool in_use = false;                    // the pass that is being scanned reads (or writes) this set: not to be evicted for memory
                       size_t run_bytes() const { return run_fp ? run_n * 6 + 16 + (run_cells + 1) * 4 : 0; }
                       size_t bytes() const { return bin_rec_cap * 8 + bin_tcur_cap * 4 + bin_count_cap * 4 + run_bytes(); } };
    // Record cache (the default): bin once, probe many.  Complete record sets stay resident, least recently
    // used first out, up to rec_limit bytes (gbn_record_cache_set_limit / GBN_RECORD_CACHE_MB; default a quarter of the
    // device's memory): a pass whose key is cached queues probe + rare kernel only -- every later query batch of a stream
    // over one shard, every block view the shim searches again.  The reference keeps what ITS scan needs of the database on
    // the device for the life of the process the same way (the per-OID subject cache, GB/gpu_blastn_MB_and_smallNa.cu:1461-1468).
    // rec_limit == 0: off -- every pass bins for itself into `scratch` (bench.py's headline: the north_star scan).
    std::vector<RecordSet *> rec_sets; long long rec_limit = -1; unsigned long long rec_clock = 0;
    long long rec_hits = 0, rec_misses = 0, rec_evictions = 0, rec_bypass = 0, rec_prepared = 0, rec_runs_built = 0, rec_runs_passes = 0;
    double rec_runs_build_ms = 0;       // GPU time of the last build of a sorted set
    RecordSet scratch, alt;             // cache off, or a set larger than the cache: the pass's own records; alt: binned ahead
    void swap_scan_sets() { std::swap(scratch, alt); }
    // Binning ahead (cache off; pipelined passes over one range of one shard, GBN_BIN_AHEAD=0: off): the binning kernel reads the
    // subjects only, so a pass queues the binning kernel of the NEXT pass -- into the other set of buffers -- behind its own
    // kernels and in front of its host synchronisation; the next pass, if its records are to be the same, finds them there and
    // queues probe + rare kernel only.  Every pass still bins once; what goes is the idle time of the GPU between a pass's last
    // kernel and the next pass's first (0.5 ms of 13.5 on C2).  A pass speculates only when the pass BEFORE it had the same key
    // (a repeat has been seen: a caller that rotates shards or table shapes never pays for a binning kernel nobody uses).
    struct BinAhead { bool valid = false; RecKey key;
                      hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; int pair = 0; } ahead;     // (two pairs of events around the kernel: a pass reads one pair while it records the other)
    RecKey last_key; bool last_key_valid = false;      // the key of the last binned pass (cache off)
    bool want_ahead = false;            // the pass being scanned may bin ahead (set by run_search)
    bool counters_zeroed = false;       // counters[0 .. 3] are zero and nothing is queued that writes them (the pass before binned ahead)
    bool seed_copy_pending = false;     // ev_seed stands for a copy of the seeds on stream2 that the next scan must not overtake
    long long ahead_hits = 0, ahead_misses = 0;
    hipEvent_t ev_back = nullptr;       // behind the read-back copies of a scan
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk[4] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex mu;
    // pinned host copies of a range's initial hits and gapped extensions, handed out again (hitbuf_get)
    struct HitBuf { GbnDevInitHit *hih = nullptr; GbnDevGapped *hg = nullptr; size_t cap = 0; };
    std::mutex hitbuf_mu; std::vector<HitBuf> hitbuf_idle;
    // pinned staging buffers of the batches' uploads (query, stretches, per-context block), handed out again: a set-up makes
    // no blocking copy from pageable memory (engine.cpp: upload_batch_tables -- why)
    std::vector<std::pair<void *, size_t>> stage_idle;
    // traceback stage: stream and pinned staging buffer of gather_shard_bytes
    std::mutex gather_mu; hipStream_t gather_stream = nullptr; uint8_t *gather_stage = nullptr; size_t gather_stage_cap = 0;
};
typedef Engine::RecKey RecKey;
typedef Engine::RecordSet RecordSet;
typedef Engine::HitBuf HitBuf;

// One engine per device, created by gbn_init / gbn_use_device (or by the first call that needs one) and alive
// until gbn_release.  Every entry point works with exactly one of them: the one its GbnBatch / GbnDb / GbnResults
// lives on, or -- for calls that create such an object -- the calling thread's device (gbn_use_device; default:
// the device of the first gbn_init).  Searches on different devices run concurrently (the reference leases its
// GPUs to search threads the same way, GB/gpu_blast_multi_gpu_utils.cpp:105-139); calls on one device are
// serialised by that engine's mutex.
constexpr int kMaxDevices = 64;
extern Engine *g_eng[kMaxDevices];
extern std::mutex g_eng_mu;
extern int g_default_dev;
extern thread_local Engine *tl_eng;         // the engine the calling thread is working with (set by enter)
extern thread_local int tl_sel;             // gbn_use_device
#define E (*gbn::tl_eng)
// the engine whose lock the calling thread holds (EngLock below): what lets an allocation that fails give cached records up
extern thread_local Engine *tl_mu_owner;
struct EngLock {                            // std::lock_guard over Engine::mu that says so
    Engine &e; bool held;
    explicit EngLock(Engine &en) : e(en), held(true) { e.mu.lock(); tl_mu_owner = &e; }
    EngLock(Engine &en, std::try_to_lock_t) : e(en), held(en.mu.try_lock()) { if (held) tl_mu_owner = &e; }
    bool owns_lock() const { return held; }
    ~EngLock() { if (held) { tl_mu_owner = nullptr; e.mu.unlock(); } }
    EngLock(const EngLock &) = delete; EngLock &operator=(const EngLock &) = delete;
};
size_t rec_evict_for_memory();              // engine_scan.cpp: the record cache's sets that no pass is using go (bytes given up)
int engine_init(int dev, Engine **out);
inline void enter(Engine *e) { tl_eng = e; if (e && e->device >= 0) (void)hipSetDevice(e->device); }
int enter_current();

// ---- the device pool (engine.cpp)
hipError_t pool_alloc(void **p, size_t bytes);
void pool_free(void *p);
long pool_check_guards();
void pool_drain(int dev);
int pool_poison();

template <class T> inline int dev_alloc(T *&p, size_t n) {
    p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = pool_alloc((void **)&p, n * sizeof(T));
    // the device is full while gigabytes of evictable scan records are resident: they go, once -- if this thread
    // holds the engine's lock (a search, a call of the cache's own entry points): nobody else is walking the cache then
    if (e == hipErrorOutOfMemory && tl_eng && tl_mu_owner == tl_eng && rec_evict_for_memory() > 0) { (void)hipGetLastError(); e = pool_alloc((void **)&p, n * sizeof(T)); }
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); p = nullptr; set_error("out of device memory (" + std::to_string(n * sizeof(T)) + " bytes asked for)"); return GBN_ERR_NOMEM; }
    HIPCHK(e);
    return GBN_OK;
}
template <class T> inline int dev_upload(T *&p, const T *h, size_t n) {
    int rc = dev_alloc(p, n);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return GBN_OK;
}
template <class T> inline void dev_free(T *&p) { if (p) pool_free((void *)p); p = nullptr; }

// ---- query batches on the device, tile tables, scratch, the stage in flight (engine.cpp)
void finish_build(DeviceBatch *d);
int upload_ctx_cutoffs(GbnBatch &b);
struct TileSet { GbnTile *d_tiles = nullptr; int64_t ntiles = 0; std::vector<int64_t> first_tile_of_subj; int64_t bases = 0; mutable int64_t scan_positions = -1; };     // (scan_positions: bin_positions' memo)
int get_tiles(GbnDb &db, int lut, int step, int tpos, int32_t s0, int32_t s1, const TileSet **out);
void free_tile_cache(GbnDb &db);
int grow_seed_buffers(size_t want);
int grow_key_buffers(Engine::KeySet &KS, size_t n);
int grow_ihit_buffers(int slot, size_t n);
void record_failure(const GbnResults *res, int rc, const std::string &what);
int wait_pending_gpu();
void wait_host();
void wait_tail(std::shared_future<void> &slot);
int wait_pending();
int take_failure(const GbnResults *res);
int64_t gap_scratch_ints(const GbnBatch &b, int32_t max_len, int32_t max_ctx, int32_t *row_len);

// ---- the scan of a subject range and the record cache (engine_scan.cpp)
void fill_scan_params(GbnScanParams &P, const GbnBatch &b, const GbnDb &db, const TileSet &ts);
int choose_bins(const GbnBatch &b);
int scan_slices(const GbnBatch &b);
void recset_free(RecordSet &r);
void recset_free_runs(RecordSet &r);
int recset_size(RecordSet &r, size_t need_u64, size_t need_tcur, size_t need_count);
long long rec_limit_bytes();
size_t rec_held_bytes();
void rec_drop(size_t i, bool evicted);
void recset_move(RecordSet &dst, RecordSet &src);
void rec_make_room(size_t need, long long limit, const RecordSet *keep, const RecKey *sweep = nullptr, RecordSet *into = nullptr);
void rec_purge(const void *db, bool to_scratch = false);
struct BinLayout { int nb = 0, nwriters = 0; size_t nstream = 0, subcap = 0, nseq = 0, need_u64 = 0;
                   size_t count_words() const { return 2 * nstream + 4; }      // counts, overflow word + spare, uncapped totals
                   size_t bytes() const { return need_u64 * 8 + nstream * nseq * 4 + count_words() * 4; } };
int64_t bin_positions(const GbnDb &db, const TileSet &ts, int32_t s0, int32_t s1, int lut, int step);
int bin_layout(int nb, int64_t ntiles, int64_t npos, double slack, BinLayout &L);
RecordSet *rec_find(const RecKey &key);
int rec_acquire(const RecKey &key, const BinLayout &L, long long limit, RecordSet **out);
constexpr int kSkewedRange = -1000;       // internal status of run_scan: split this subject range and try again
int run_scan(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag, unsigned long long cnt[2], int64_t *bases_out);

// ---- the stages behind the scan (engine_stages.cpp)
int compact_seeds(hipStream_t st);
void seed_key_layout(const GbnBatch &b, const GbnDb &db, int32_t s0, int32_t s1, GbnKeyParams &K, int *ck_bits, int *scan_bits = nullptr, int *group_key_bits = nullptr);
bool seed_key_layout_fits(const GbnBatch &b, const GbnKeyParams &K, int ck_bits);
int search_range(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag, int keep_stages, int overlap = 0);
void hitbuf_drain();
int stage_get(size_t bytes, void **p, size_t *cap);
void stage_put(void *p, size_t cap);
}  // namespace gbn
