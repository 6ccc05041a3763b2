// engine.cpp -- device residency and orchestration of the preliminary search
// (the BLAST_PreliminarySearchEngine analogue,
// GB/gpu_blastn_pre_search_engine.cpp:1125-1464), plus the C ABI.
//
// Per query batch: host set-up (batch.cpp) + lookup structures built on the device (lutbuild.hip, own
// stream).  Per subject range of a shard, on the engine's stream: scan_bin -> probe_bin -> probe_rare
// (or scan_seed_kernel); then -- on the second stream + a host thread when the caller pipelines batches --
// two stable radix sorts (scan order, then diagonal slot) -> diag_ungapped_kernel -> greedy_kernel |
// dynprog_kernel (all initial hits) -> D2H of initial hits + gapped results -> host replay (hsp_host.cpp).
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
hipError_t launch_scan_seed(const GbnScanParams &p, int grid, hipStream_t st);
hipError_t launch_scan_bin(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev);
hipError_t launch_scan_bin_parts(const GbnBinParams &b, int grid2, hipStream_t st, hipEvent_t *ev, int parts, hipEvent_t tables_ready);
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
hipError_t launch_gather_bytes(const uint8_t *src, const int64_t *src_off, const int64_t *dst_off, const int32_t *nbytes,
                               int32_t n, uint8_t *dst, hipStream_t st);
hipError_t sort_pairs_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout,
                          const uint32_t *vin, uint32_t *vout, int64_t n, int end_bit, hipStream_t st);
int scan_slice_count(const GbnScanParams &p);
int scan_slice_blocks(const GbnScanParams &p, int num_cu);
hipError_t launch_scan_slice(const GbnScanParams &p, int num_cu, GbnDevSeed *seg, uint32_t seg_cap, uint32_t *seg_count,
                             unsigned long long *seg_max, hipStream_t st);
int scan_slice_segments(const GbnScanParams &p, int num_cu, int *ordered);
hipError_t launch_seed_compact(const GbnDevSeed *seg, const uint32_t *seg_count, unsigned long long *seg_first, int nseg, uint32_t seg_cap,
                               GbnDevSeed *out, unsigned long long out_cap, hipStream_t st);
hipError_t sort_keys_u64(void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout, int64_t n, int begin_bit, int end_bit, hipStream_t st);

static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }

// Stages of one search run on several threads (the inline seed stage, the stage in flight on the second stream, the
// detached host replays) and all add to the caller's GbnDiagnostics: every such update holds this lock.
static std::mutex g_diag_mu;
#define GBN_DIAG_LOCKED(stmt) do { std::lock_guard<std::mutex> dl_(g_diag_mu); stmt; } while (0)

// the switches (gbn_dev.h): read from the environment at every use
long long switch_value(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
bool switch_is_set(const char *name) { return getenv(name) != nullptr; }

// GBN_TRACE=1: wall-clock marks of the host-side pipeline on stderr (ms since the first mark)
void trace_mark(const char *what) {
    static const bool on = getenv("GBN_TRACE") && atoi(getenv("GBN_TRACE")) != 0;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[gbn trace] %9.3f ms  %s\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

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
    int mode = 0, fl = 0, fr = 0;
    // lookup structures still being built on the builder's stream: the event they are complete at, and the
    // builder's scratch, which goes back to the pool once it has fired
    hipEvent_t ready = nullptr; std::vector<void *> build_scratch;
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
    std::future<int> pending; bool has_pending = false; std::string pending_err; const GbnResults *pending_res = nullptr;
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
                       uint32_t *bin_count = nullptr; size_t bin_count_cap = 0;            // [nb][nwriters] + overflow flag
                       RecKey key; bool complete = false;      // the buffers hold every record of `key` (binned, no stream overflowed)
                       bool queued = false;                    // the binning kernel that writes them is queued on the engine's stream, its overflow flag not read yet (gbn_db_prepare_records)
                       unsigned long long stamp = 0;           // last use (record cache: least recently used goes first)
                       size_t bytes() const { return bin_rec_cap * 8 + bin_tcur_cap * 4 + bin_count_cap * 4; } };
    // Record cache (the default): bin once, probe many.  Complete record sets stay resident, least recently
    // used first out, up to rec_limit bytes (gbn_record_cache_set_limit / GBN_RECORD_CACHE_MB; default a quarter of the
    // device's memory): a pass whose key is cached queues probe + rare kernel only -- every later query batch of a stream
    // over one shard, every block view the shim searches again.  The reference keeps what ITS scan needs of the database on
    // the device for the life of the process the same way (the per-OID subject cache, GB/gpu_blastn_MB_and_smallNa.cu:1461-1468).
    // rec_limit == 0: off -- every pass bins for itself into `scratch` (bench.py's headline: the north_star scan).
    std::vector<RecordSet *> rec_sets; long long rec_limit = -1; unsigned long long rec_clock = 0;
    long long rec_hits = 0, rec_misses = 0, rec_evictions = 0, rec_bypass = 0, rec_prepared = 0;
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
static Engine *g_eng[kMaxDevices];
static std::mutex g_eng_mu;
static int g_default_dev = -1;
static thread_local Engine *tl_eng = nullptr;       // the engine the calling thread is working with (set by enter)
static thread_local int tl_sel = -1;                // gbn_use_device
#define E (*tl_eng)

static int engine_init(int dev, Engine **out);
static inline void enter(Engine *e) { tl_eng = e; if (e && e->device >= 0) (void)hipSetDevice(e->device); }
// the calling thread's engine for calls that create batches / shards: its chosen device, else the process default,
// else the thread's current HIP device (first use without gbn_init)
static int enter_current() {
    int dev = tl_sel;
    if (dev < 0) { std::lock_guard<std::mutex> lk(g_eng_mu); dev = g_default_dev; }
    Engine *e = nullptr;
    const int rc = engine_init(dev, &e);
    if (rc) return rc;
    enter(e);
    return GBN_OK;
}

// (The HIP current device is per host thread: every C-ABI entry point that touches HIP -- callers use worker
// threads for set-up and extension stages -- goes through enter(), which selects the engine's device first.)

// Device memory of query batches and of the table builder comes from a small pool: freed blocks are kept
// (up to pool_cap() bytes) and handed out again for requests of about their size.  hipFree waits for the
// whole device, which would stall a running search every time a finished batch is released.
namespace {
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> idle[kMaxDevices];        // per device: size -> block
    std::map<void *, std::pair<size_t, int>> size_of;       // every block handed out by the pool: size, device
    size_t held[kMaxDevices] = {};
};
int cur_dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0; return d; }
DevPool g_pool;
// idle blocks kept: an eighth of the device's memory, 24 GiB at most (GBN_POOL_GIB overrides); when an allocation fails
// the idle blocks are given back and it is tried again
size_t pool_cap() {
    static const size_t cap = [] {
        if (const char *e = getenv("GBN_POOL_GIB")) return (size_t)std::max(0, atoi(e)) << 30;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) return (size_t)24 << 30;
        return std::min<size_t>(tot / 8, (size_t)24 << 30);
    }();
    return cap;
}
size_t pool_round(size_t bytes) { const size_t g = bytes >= ((size_t)1 << 20) ? ((size_t)1 << 20) : 4096; return (bytes + g - 1) / g * g; }
// GBN_POISON=<byte> (tests): every block handed out is filled with that byte first, so that a kernel reading
// memory nobody wrote shows up whatever the device memory happened to hold
int pool_poison() { static const int v = getenv("GBN_POISON") ? (atoi(getenv("GBN_POISON")) & 255) : -1; return v; }
hipError_t poison_block(void *p, size_t bytes) {           // (before anything queued on a stream afterwards touches the block)
    hipError_t e = hipMemset(p, pool_poison(), bytes);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
// GBN_GUARD=1 (tests): every block gets kGuard bytes of a known pattern in front and behind; they are checked when the
// block comes back (and by gbn_debug_check_guards): a kernel that writes past either end of its buffer is named by
// the block's size instead of showing up as whatever the neighbouring allocation -- possibly a code object -- does next
constexpr size_t kGuard = 4096;
bool pool_guard() { static const bool v = getenv("GBN_GUARD") && atoi(getenv("GBN_GUARD")) != 0; return v; }
std::atomic<long> g_guard_violations{0};
hipError_t guard_fill(void *raw, size_t bytes) {
    hipError_t e = hipMemset(raw, 0xA5, kGuard);
    if (e == hipSuccess) e = hipMemset((char *)raw + kGuard + bytes, 0xA5, kGuard);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
bool guard_check(void *user, size_t bytes, const char *when) {
    std::vector<uint8_t> h(2 * kGuard);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(h.data(), (char *)user - kGuard, kGuard, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h.data() + kGuard, (char *)user + bytes, kGuard, hipMemcpyDeviceToHost) != hipSuccess) return true;
    bool ok = true;
    for (size_t i = 0; i < 2 * kGuard; i++) if (h[i] != 0xA5) {
        fprintf(stderr, "[gbn guard] %s: block of %zu bytes at %p: byte %ld %s the block is 0x%02x\n", when, bytes, user,
                i < kGuard ? (long)(kGuard - i) : (long)(i - kGuard), i < kGuard ? "in front of" : "behind", h[i]);
        ok = false; g_guard_violations++;
        break;
    }
    if (!ok) { fflush(stderr); abort(); }       // a test run: nothing after an out-of-bounds write can be trusted
    return ok;
}
hipError_t raw_alloc(void **p, size_t bytes) {
    if (!pool_guard()) return hipMalloc(p, bytes);
    void *raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes + 2 * kGuard);
    if (e != hipSuccess) return e;
    if ((e = guard_fill(raw, bytes)) != hipSuccess) { (void)hipFree(raw); return e; }
    *p = (char *)raw + kGuard;
    return hipSuccess;
}
void raw_free(void *p, size_t bytes) {
    if (!pool_guard()) { (void)hipFree(p); return; }
    guard_check(p, bytes, "free");
    (void)hipFree((char *)p - kGuard);
}
void raw_free_on(void *p, size_t bytes, int dev) {       // (a block is released on the device it was allocated on)
    const int cur = cur_dev();
    if (cur != dev) (void)hipSetDevice(dev);
    raw_free(p, bytes);
    if (cur != dev) (void)hipSetDevice(cur);
}
hipError_t pool_alloc(void **p, size_t bytes) {
    bytes = pool_round(std::max<size_t>(bytes, 1));
    const int dev = cur_dev();
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto &idle = g_pool.idle[dev];
        auto it = idle.lower_bound(bytes);
        if (it != idle.end() && it->first <= bytes + bytes / 4) {
            *p = it->second; g_pool.held[dev] -= it->first; g_pool.size_of[*p] = std::make_pair(it->first, dev);
            const size_t got = it->first; idle.erase(it);
            if (pool_poison() >= 0) return poison_block(*p, got);
            return hipSuccess;
        }
    }
    hipError_t e = raw_alloc(p, bytes);
    if (e != hipSuccess) {                      // give the idle blocks back and try again
        std::vector<std::pair<void *, size_t>> drop;
        { std::lock_guard<std::mutex> lk(g_pool.mu); for (auto &kv : g_pool.idle[dev]) drop.emplace_back(kv.second, kv.first); g_pool.idle[dev].clear(); g_pool.held[dev] = 0; }
        for (auto &q : drop) raw_free(q.first, q.second);
        (void)hipGetLastError();
        e = raw_alloc(p, bytes);
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.size_of[*p] = std::make_pair(bytes, dev); }
    if (e == hipSuccess && pool_poison() >= 0) e = poison_block(*p, bytes);
    return e;
}
void pool_free(void *p) {
    if (!p) return;
    size_t bytes = 0; int dev = cur_dev();
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto it = g_pool.size_of.find(p);
        if (it != g_pool.size_of.end()) {
            bytes = it->second.first; dev = it->second.second; g_pool.size_of.erase(it);
            if (pool_guard()) guard_check(p, bytes, "release");
            if (g_pool.held[dev] + bytes <= pool_cap()) { g_pool.idle[dev].emplace(bytes, p); g_pool.held[dev] += bytes; return; }
        }
    }
    raw_free_on(p, bytes, dev);
}
// every block the pool knows (handed out or idle): guards intact?  Returns the violations seen so far.
long pool_check_guards() {
    if (!pool_guard()) return 0;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto &kv : g_pool.size_of) guard_check(kv.first, kv.second.first, "check");
    for (int d = 0; d < kMaxDevices; d++) for (auto &kv : g_pool.idle[d]) guard_check(kv.second, kv.first, "check (idle)");
    return g_guard_violations.load();
}
void pool_drain(int dev) {
    std::vector<std::pair<void *, size_t>> drop;
    { std::lock_guard<std::mutex> lk(g_pool.mu); for (auto &kv : g_pool.idle[dev]) drop.emplace_back(kv.second, kv.first); g_pool.idle[dev].clear(); g_pool.held[dev] = 0; }
    for (auto &q : drop) raw_free_on(q.first, q.second, dev);
}
}  // namespace

template <class T> static int dev_alloc(T *&p, size_t n) {
    p = nullptr;
    if (n == 0) n = 1;
    const hipError_t e = pool_alloc((void **)&p, n * sizeof(T));
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); p = nullptr; set_error("out of device memory (" + std::to_string(n * sizeof(T)) + " bytes asked for)"); return GBN_ERR_NOMEM; }
    HIPCHK(e);
    return GBN_OK;
}
template <class T> static int dev_upload(T *&p, const T *h, size_t n) {
    int rc = dev_alloc(p, n);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return GBN_OK;
}
template <class T> static void dev_free(T *&p) { if (p) pool_free((void *)p); p = nullptr; }

// the batch's lookup structures are complete (waits for the builder if they are not): scratch back to the pool
static void finish_build(DeviceBatch *d) {
    if (!d || !d->ready) return;
    (void)hipEventSynchronize(d->ready);
    for (void *p : d->build_scratch) pool_free(p);
    d->build_scratch.clear();
    (void)hipEventDestroy(d->ready); d->ready = nullptr;
}

void free_device_batch(DeviceBatch *d) {
    if (!d) return;
    finish_build(d);
    dev_free(d->q8_base); dev_free(d->q2_base); dev_free(d->qinv_base); dev_free(d->q4_base); dev_free(d->pv); dev_free(d->cellw); dev_free(d->cellt); dev_free(d->sidet); dev_free(d->side_start); dev_free(d->cell_start); dev_free(d->ent); dev_free(d->pvx); dev_free(d->pstart);
    dev_free(d->ctx_off); dev_free(d->ctx_len); dev_free(d->ctx_xdrop); dev_free(d->ctx_cutoff); dev_free(d->ctx_pack);
    dev_free(d->ctx_reduced); dev_free(d->ctx_hint); dev_free(d->ctx_blk); dev_free(d->matrix); dev_free(d->score_table);
    delete d;
}

// fingerprint word of one query offset (layout in gbn_dev.h / kernels.hip fp_pass)
static inline uint32_t fingerprint(const uint8_t *q, int32_t off, int lut, bool force) {
    uint32_t l = 0, r = 0;
    for (int k = 1; k <= 8; k++) l |= (uint32_t)(q[off - k] & 3) << (2 * (k - 1));
    for (int j = 0; j < 7; j++) r |= (uint32_t)(q[off + lut + j] & 3) << (2 * (6 - j));
    return (l << 15) | (r << 1) | (force ? 1u : 0u);
}

static int upload_ctx_cutoffs(GbnBatch &b) {
    DeviceBatch *d = b.dev;
    std::vector<int32_t> xd, cu, rd;
    for (auto &c : b.ctx) { xd.push_back(c.x_dropoff); cu.push_back(c.cutoff_score); rd.push_back(c.reduced_cutoff); }
    size_t n = b.ctx.size();
    HIPCHK(hipMemcpy(d->ctx_xdrop, xd.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->ctx_cutoff, cu.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->ctx_reduced, rd.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> pk(4 * n);
    for (size_t c = 0; c < n; c++) { pk[4 * c] = xd[c]; pk[4 * c + 1] = rd[c]; pk[4 * c + 2] = cu[c]; pk[4 * c + 3] = 0; }
    HIPCHK(hipMemcpy(d->ctx_pack, pk.data(), 4 * n * 4, hipMemcpyHostToDevice));
    return GBN_OK;
}

// lookup structures from host-built tables (GBN_HOST_LOOKUP=1; the device builder below is the default)
static int upload_host_tables(GbnBatch &b) {
    DeviceBatch *d = b.dev;
    const HostLookup &L = b.lut;
    const uint8_t *q = b.query();
    int rc;
    std::vector<uint32_t> cellw((size_t)L.ncells, 0), cellt((size_t)L.ncells, 0);
    std::vector<unsigned long long> ent(L.cell_qoff.size());
    // 15-bit reduced fingerprint: 3.5 bases to the right (7 bits, high) and 4 bases to the left (8 bits, low)
    auto reduce = [](uint32_t fp) { return ((((fp >> 1) & 0x3fffu) >> 7) << 8) | ((fp >> 15) & 0xffu); };
    std::vector<uint16_t> sidet; std::vector<uint32_t> side_start; int64_t cur_bin = -1; bool forced_cell = false;
    for (int64_t c = 0; c < L.ncells; c++) {
        uint32_t s = L.cell_start[c], e = L.cell_start[c + 1];
        for (uint32_t k = s; k < e; k++) {
            int32_t off = L.cell_qoff[k];
            bool force = (d->mode == GBN_EXT_SMALL_ONEBYTE) && (off + L.lut >= b.qlen);
            uint32_t fp = fingerprint(q, off, L.lut, force);
            ent[k] = ((unsigned long long)fp << 32) | (uint32_t)off;
            if (k == s) cellw[c] = (fp & 0x7fffffffu) | ((e - s > 1) ? 0x80000000u : 0u);
            // LDS cell table of the partitioned scan (layout: GbnBinParams::cellt)
            if (k == s) cellt[c] = 0x8000u | reduce(fp) | (reduce(fp) << 16);
            else if (k == s + 1) cellt[c] = (cellt[c] & 0xffffu) | 0x80000000u | (reduce(fp) << 16);
            if (force) forced_cell = true;
        }
        if (e - s >= 3 || forced_cell) {
            // three or more entries: reduced fingerprints go to the bin's side list (capacity
            // GBN_BIN_SIDE per bin); anything that does not fit is "always rare path"
            const int64_t bin = c >> GBN_BIN_CBITS(L.lut);
            if (bin != cur_bin) { cur_bin = bin; while ((int64_t)side_start.size() <= bin) side_start.push_back((uint32_t)sidet.size()); }
            // (same layout as the device builder: a list that does not fit still takes up its slots)
            const uint32_t off = (uint32_t)sidet.size() - side_start[bin], cnt = e - s;
            cellt[c] = 0x80000000u;
            if (!forced_cell && cnt < 16384) {
                const bool fits = off + cnt <= GBN_BIN_SIDE;
                for (uint32_t k = s; k < e; k++) sidet.push_back(fits ? (uint16_t)reduce((uint32_t)(ent[k] >> 32)) : (uint16_t)0);
                if (fits) cellt[c] = 0x80000000u | off | (cnt << 16);
            }
            forced_cell = false;
        }
    }
    {
        const int64_t nbins = std::max<int64_t>(1, L.ncells >> GBN_BIN_CBITS(L.lut));
        while ((int64_t)side_start.size() <= nbins) side_start.push_back((uint32_t)sidet.size());
        sidet.push_back(0);
    }
    if ((rc = dev_upload(d->pv, L.pv.data(), L.pv.size()))) return rc;
    if ((rc = dev_upload(d->cellw, cellw.data(), cellw.size()))) return rc;
    if ((rc = dev_upload(d->cellt, cellt.data(), cellt.size()))) return rc;
    if ((rc = dev_upload(d->sidet, sidet.data(), sidet.size()))) return rc;
    if ((rc = dev_upload(d->side_start, side_start.data(), side_start.size()))) return rc;
    if ((rc = dev_upload(d->cell_start, L.cell_start.data(), L.cell_start.size()))) return rc;
    {
        // one pad entry so that an empty list still has a valid pointer
        ent.push_back(0);
        if ((rc = dev_upload(d->ent, ent.data(), ent.size()))) return rc;
    }
    return GBN_OK;
}

// The rank form of the presence bits and entry starts, for batches the folded slice scan takes (more than one slice
// of presence bits): queued on `st` behind the structures it reads.  keep: where the scratch goes while the stream
// runs (the batch's build_scratch), null: finished and freed here.
static int scan_slices(const GbnBatch &b);
static int build_rank_table(GbnBatch &b, hipStream_t st, std::vector<void *> *keep) {
    DeviceBatch *d = b.dev;
    if (scan_slices(b) <= 1) return GBN_OK;
    const int64_t ncells = b.lut.ncells, nwords = (ncells + 31) / 32;
    uint32_t *popc = nullptr, *prefix = nullptr; void *tmp = nullptr; size_t tb = 0;
    int rc;
    auto fail = [&](int code) { (void)hipStreamSynchronize(st); dev_free(popc); dev_free(prefix); if (tmp) pool_free(tmp); return code; };
    if ((rc = dev_alloc(popc, (size_t)nwords + 1)) || (rc = dev_alloc(prefix, (size_t)nwords + 1))) return fail(rc);
    if ((rc = dev_alloc(d->pvx, 2 * (size_t)nwords)) || (rc = dev_alloc(d->pstart, (size_t)std::max(b.qlen, 1) + 2))) return fail(rc);
#define RANKCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return fail(GBN_ERR_HIP); } } while (0)
    RANKCHK(lut_rank_count(d->pv, nwords, popc, st));
    RANKCHK(lut_scan(nullptr, tb, popc, prefix, nwords + 1, st));
    RANKCHK(pool_alloc(&tmp, tb + 256));
    tb += 256;
    RANKCHK(lut_scan(tmp, tb, popc, prefix, nwords + 1, st));
    RANKCHK(lut_rank_fill(d->pv, prefix, d->cell_start, ncells, nwords, d->pvx, d->pstart, st));
    if (keep) { keep->push_back(popc); keep->push_back(prefix); keep->push_back(tmp); return GBN_OK; }
    RANKCHK(hipStreamSynchronize(st));
#undef RANKCHK
    dev_free(popc); dev_free(prefix); pool_free(tmp);
    return GBN_OK;
}

// lookup structures built on the device from the uploaded query (lutbuild.hip)
static int build_tables_on_device(GbnBatch &b) {
    DeviceBatch *d = b.dev;
    HostLookup &L = b.lut;
    int rc;
    hipStream_t st = E.stream_build;
    std::vector<int32_t> sl, sr;
    for (auto &sg : L.segments) if (sg.second >= sg.first) { sl.push_back(sg.first); sr.push_back(sg.second); }
    int32_t *d_sl = nullptr, *d_sr = nullptr;
    uint32_t *count = nullptr, *many = nullptr, *many_prefix = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    uint32_t *keys_a = nullptr, *keys_b = nullptr; unsigned long long *ctr = nullptr; void *tmp = nullptr;
    auto cleanup = [&]() { dev_free(d_sl); dev_free(d_sr); dev_free(count); dev_free(many); dev_free(many_prefix);
                           dev_free(vals_a); dev_free(vals_b); dev_free(keys_a); dev_free(keys_b); dev_free(ctr); if (tmp) pool_free(tmp); tmp = nullptr; };
    // (on an error kernels may already be queued on the builder's stream: they finish before their scratch goes back to the pool)
#define LUTCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); (void)hipStreamSynchronize(st); cleanup(); return GBN_ERR_HIP; } } while (0)
#define LUTRC(x) do { if ((rc = (x))) { (void)hipStreamSynchronize(st); cleanup(); return rc; } } while (0)
    const size_t nc1 = (size_t)L.ncells + 1, qn = (size_t)std::max(b.qlen, 1);
    LUTRC(dev_upload(d_sl, sl.data(), sl.size())); LUTRC(dev_upload(d_sr, sr.data(), sr.size()));
    LUTRC(dev_alloc(count, nc1)); LUTRC(dev_alloc(many, nc1)); LUTRC(dev_alloc(many_prefix, nc1));
    LUTRC(dev_alloc(keys_a, qn)); LUTRC(dev_alloc(keys_b, qn)); LUTRC(dev_alloc(vals_a, qn)); LUTRC(dev_alloc(vals_b, qn));
    LUTRC(dev_alloc(ctr, 2));
    LUTRC(dev_alloc(d->cell_start, nc1)); LUTRC(dev_alloc(d->cellw, (size_t)L.ncells)); LUTRC(dev_alloc(d->cellt, (size_t)L.ncells));
    LUTRC(dev_alloc(d->pv, (size_t)((L.ncells + 31) / 32)));
    LUTCHK(hipMemsetAsync(ctr, 0, 16, st));
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    LutBuild B; std::memset(&B, 0, sizeof(B));
    B.q8 = d->q8; B.qlen = b.qlen; B.seg_left = d_sl; B.seg_right = d_sr; B.nseg = (int32_t)sl.size();
    B.lut = L.lut; B.word = L.word; B.q_bits = std::min(31, bits_for((uint64_t)b.qlen + 1)); B.ncells = L.ncells;
    B.count = count; B.keys_a = keys_a; B.keys_b = keys_b; B.vals_a = vals_a; B.vals_b = vals_b;
    B.cell_start = d->cell_start; B.cellw = d->cellw; B.cellt = d->cellt; B.pv = d->pv; B.many = many; B.many_prefix = many_prefix;
    B.cbits = GBN_BIN_CBITS(L.lut);
    B.nbins = (int32_t)std::max<int64_t>(1, L.ncells >> B.cbits);
    B.descending = (L.type == GBN_LUT_MB) ? 1 : 0;     // (the fallback below never produces or removes a megablast table)
    // Megablast / standard tables need nothing back on the host: the whole build is queued on the builder's
    // stream, sized by upper bounds (at most one word per query position), and the batch carries an event;
    // the engine lets the probe kernel wait for it -- the binning kernel of the search starts at once.
    const bool sync_build = gbn::switch_value("GBN_SYNC_BUILD", 0) != 0;
    if (L.type != GBN_LUT_SMALL_NA && !sync_build) {
        // (no count per cell here: the sorted list gives cell_start -- a gigabyte less per 5 Mb build than counting with
        // atomics, clearing the counters first and scanning them afterwards)
        B.count = nullptr;
        LUTCHK(lut_enumerate(B, st));
        B.onebyte_mode = 0;
        LUTRC(dev_alloc(d->ent, qn + 1));
        LUTRC(dev_alloc(d->sidet, qn + 1)); LUTRC(dev_alloc(d->side_start, (size_t)B.nbins + 1));
        LUTCHK(hipMemsetAsync(d->sidet, 0, (qn + 1) * 2, st));
        size_t b1 = 0, b2 = 0;
        const int key_bits = 2 * L.lut + 1;                         // (the cell; one bit more: "no word at this position" sorts last)
        LUTCHK(lut_sort(nullptr, b1, B, (int64_t)qn, key_bits, st));
        LUTCHK(lut_scan(nullptr, b2, many, many_prefix, (int64_t)nc1, st));
        LUTCHK(pool_alloc(&tmp, std::max(b1, b2) + 256));
        size_t tb = std::max(b1, b2) + 256;
        uint32_t *n_valid = reinterpret_cast<uint32_t *>(ctr);           // (the counters' first word: nothing else uses it in this branch)
        LUTCHK(lut_sort(tmp, tb, B, (int64_t)qn, key_bits, st, n_valid));
        LUTCHK(lut_cell_starts(B, n_valid, st));
        B.ent = d->ent; B.sidet = d->sidet; B.side_start = d->side_start;
        LUTCHK(lut_entries(B, -1, st));
        LUTCHK(lut_cells(B, st));
        tb = std::max(b1, b2) + 256;
        LUTCHK(lut_scan(tmp, tb, many, many_prefix, (int64_t)nc1, st));
        LUTCHK(lut_side(B, st));
        LUTCHK(lut_pv(B, st));
        LUTRC(build_rank_table(b, st, &d->build_scratch));
        LUTCHK(hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
        LUTCHK(hipEventRecord(d->ready, st));
        for (void *p : {(void *)d_sl, (void *)d_sr, (void *)count, (void *)many, (void *)many_prefix, (void *)vals_a, (void *)vals_b,
                        (void *)keys_a, (void *)keys_b, (void *)ctr, tmp}) d->build_scratch.push_back(p);
        return GBN_OK;
    }
    LUTCHK(hipMemsetAsync(count, 0, nc1 * 4, st));
    LUTCHK(lut_enumerate(B, st));
    if (L.type == GBN_LUT_SMALL_NA) LUTCHK(lut_overflow_cells(B, ctr + 1, st));
    size_t b1 = 0, b2 = 0;
    const int key_bits = 2 * L.lut + 1;
    LUTCHK(lut_sort(nullptr, b1, B, (int64_t)qn, key_bits, st));
    LUTCHK(lut_scan(nullptr, b2, count, d->cell_start, (int64_t)nc1, st));
    LUTCHK(pool_alloc(&tmp, std::max(b1, b2) + 256));
    size_t tb = std::max(b1, b2) + 256;
    LUTCHK(lut_scan(tmp, tb, count, d->cell_start, (int64_t)nc1, st));
    unsigned long long h[2] = {0, 0}; uint32_t n_words = 0;
    LUTCHK(hipMemcpyAsync(h, ctr, 16, hipMemcpyDeviceToHost, st));
    LUTCHK(hipMemcpyAsync(&n_words, d->cell_start + L.ncells, 4, hipMemcpyDeviceToHost, st));
    LUTCHK(hipStreamSynchronize(st));
    const int64_t n = (int64_t)n_words;
    // small-NA table whose overflow array would not fit 15 bits: the standard table (CORE/blast_nalookup.c:184-187)
    if (L.type == GBN_LUT_SMALL_NA && 2 + h[1] >= 32768) L.type = GBN_LUT_NA;
    // (extension flavour and chain order depend on the final table kind)
    if (L.lut == L.word) d->mode = GBN_EXT_DIRECT;
    else if (L.type == GBN_LUT_SMALL_NA)
        d->mode = (L.lut % 4 == 0 && L.step % 4 == 0 && L.word - L.lut <= 4) ? GBN_EXT_SMALL_ONEBYTE : GBN_EXT_SMALL;
    else d->mode = GBN_EXT_NA;
    B.onebyte_mode = (d->mode == GBN_EXT_SMALL_ONEBYTE) ? 1 : 0;
    LUTRC(dev_alloc(d->ent, (size_t)n + 1));
    LUTCHK(hipMemsetAsync(d->ent + n, 0, 8, st));
    tb = std::max(b1, b2) + 256;
    LUTCHK(lut_sort(tmp, tb, B, (int64_t)qn, key_bits, st));
    B.ent = d->ent;
    LUTCHK(lut_entries(B, n, st));
    LUTCHK(lut_cells(B, st));
    tb = std::max(b1, b2) + 256;
    LUTCHK(lut_scan(tmp, tb, many, many_prefix, (int64_t)nc1, st));
    uint32_t side_total = 0;
    LUTCHK(hipMemcpyAsync(&side_total, many_prefix + L.ncells, 4, hipMemcpyDeviceToHost, st));
    LUTCHK(hipStreamSynchronize(st));
    LUTRC(dev_alloc(d->sidet, (size_t)side_total + 1)); LUTRC(dev_alloc(d->side_start, (size_t)B.nbins + 1));
    LUTCHK(hipMemsetAsync(d->sidet + side_total, 0, 2, st));
    B.sidet = d->sidet; B.side_start = d->side_start;
    LUTCHK(lut_side(B, st));
    LUTCHK(lut_pv(B, st));
    LUTRC(build_rank_table(b, st, nullptr));
    LUTCHK(hipStreamSynchronize(st));
#undef LUTCHK
#undef LUTRC
    cleanup();
    return GBN_OK;
}

int upload_batch(GbnBatch &b) {
    int rc = enter_current();
    if (rc) return rc;
    DeviceBatch *d = new DeviceBatch();
    d->eng = tl_eng;
    b.dev = d;
    trace_mark("upload: starts");
    HostLookup &L = b.lut;
    // extension flavour, CORE/na_ungapped.c:1753-1795
    if (L.lut == L.word) d->mode = GBN_EXT_DIRECT;
    else if (L.type == GBN_LUT_SMALL_NA)
        d->mode = (L.lut % 4 == 0 && L.step % 4 == 0 && L.word - L.lut <= 4) ? GBN_EXT_SMALL_ONEBYTE : GBN_EXT_SMALL;
    else d->mode = GBN_EXT_NA;
    {   // fingerprint lengths: a verified seed has >= ceil(e/2) matches on the left
        // or > e - ceil(e/2) on the right, e = word - lut
        int e = L.word - L.lut, h = (e + 1) / 2;
        d->fl = std::min(8, h); d->fr = std::min(7, e - h + 1);
        if (e == 0) { d->fl = 0; d->fr = 0; }
    }
    const bool host_lookup = gbn::switch_value("GBN_HOST_LOOKUP", 0) != 0;
    if ((rc = dev_upload(d->q8_base, b.qbuf.data(), b.qbuf.size()))) return rc;
    d->q8 = d->q8_base + b.qpad;
    {   // packed copy for the greedy kernel's 32-bases-per-step match runs, made on the device from q8
        const int64_t pad = 256, n = (int64_t)b.qlen + 2 * pad;
        const size_t q2_bytes = (size_t)(n + 3) / 4 + 16, qi_bytes = (size_t)(n + 7) / 8 + 16;
        if ((rc = dev_alloc(d->q2_base, q2_bytes)) || (rc = dev_alloc(d->qinv_base, qi_bytes))) return rc;
        HIPCHK(hipMemsetAsync(d->q2_base, 0xff, q2_bytes, E.stream_build));        // tails: "matches nothing"
        HIPCHK(hipMemsetAsync(d->qinv_base, 0xff, qi_bytes, E.stream_build));
        HIPCHK(lut_pack_query(d->q8_base, (int64_t)b.qbuf.size(), (int64_t)b.qpad - pad, n, d->q2_base, d->qinv_base, E.stream_build));
        d->q2 = d->q2_base + pad / 4; d->qinv = d->qinv_base + pad / 8;
        d->q4_plane = ((int64_t)b.qbuf.size() + 3) / 4 + 64;          // (an 8- or 16-byte load may start at a plane's last byte)
        if ((rc = dev_alloc(d->q4_base, (size_t)(4 * d->q4_plane)))) return rc;
        HIPCHK(hipMemsetAsync(d->q4_base, 0xff, (size_t)(4 * d->q4_plane), E.stream_build));
        HIPCHK(lut_pack_q4(d->q8_base, (int64_t)b.qbuf.size(), d->q4_base, d->q4_plane, E.stream_build));
    }
    if (host_lookup) {
        HIPCHK(hipStreamSynchronize(E.stream_build));       // q2 / qinv packed: no event travels with a host-built batch
        if (L.cell_start.empty()) fill_lookup_host(b);
        // (the host builder may have turned a small-NA table into a standard one)
        if (L.lut != L.word) d->mode = (L.type == GBN_LUT_SMALL_NA) ? ((L.lut % 4 == 0 && L.step % 4 == 0 && L.word - L.lut <= 4) ? GBN_EXT_SMALL_ONEBYTE : GBN_EXT_SMALL) : GBN_EXT_NA;
        if ((rc = upload_host_tables(b))) return rc;
        if ((rc = build_rank_table(b, E.stream_build, nullptr))) return rc;
    } else {
        if ((rc = build_tables_on_device(b))) return rc;
    }
    trace_mark("upload: lookup structures on the device");
    std::vector<int32_t> off, len;
    for (auto &c : b.ctx) { off.push_back(c.query_offset); len.push_back(c.query_length); }
    if ((rc = dev_upload(d->ctx_off, off.data(), off.size()))) return rc;
    if ((rc = dev_upload(d->ctx_len, len.data(), len.size()))) return rc;
    if ((rc = dev_alloc(d->ctx_xdrop, off.size()))) return rc;
    if ((rc = dev_alloc(d->ctx_pack, 4 * off.size()))) return rc;
    if ((rc = dev_alloc(d->ctx_cutoff, off.size()))) return rc;
    if ((rc = dev_alloc(d->ctx_reduced, off.size()))) return rc;
    if ((rc = upload_ctx_cutoffs(b))) return rc;
    {   // context of every kCtxHintShift-aligned query position (the kernels walk on from there: contexts are rarely shorter)
        std::vector<int32_t> hint(((size_t)b.qlen >> kCtxHintShift) + 2);
        size_t c = 0;
        for (size_t k = 0; k < hint.size(); k++) {
            const int64_t qpos = (int64_t)k << kCtxHintShift;
            while (c + 1 < off.size() && off[c + 1] <= qpos) c++;
            hint[k] = (int32_t)c;
        }
        if ((rc = dev_upload(d->ctx_hint, hint.data(), hint.size()))) return rc;
        // ... and per block: that context and where the next one begins (GbnExtParams::ctx_blk)
        std::vector<int32_t> blk(2 * hint.size());
        for (size_t k = 0; k < hint.size(); k++) {
            const size_t ci = (size_t)hint[k];
            const int64_t last = ((int64_t)k << kCtxHintShift) + ((int64_t)1 << kCtxHintShift) - 1;
            blk[2 * k] = hint[k];
            blk[2 * k + 1] = ci + 1 < off.size() ? off[ci + 1] : INT32_MAX;
            if (ci + 2 < off.size() && off[ci + 2] <= last) blk[2 * k + 1] = INT32_MIN;
        }
        if ((rc = dev_upload(d->ctx_blk, blk.data(), blk.size()))) return rc;
    }
    if ((rc = dev_upload(d->matrix, &b.matrix[0][0], 256))) return rc;
    if ((rc = dev_upload(d->score_table, b.score_table, 256))) return rc;
    trace_mark("upload: done");
    return GBN_OK;
}

// ---------------------------------------------------------------------------
struct TileSet { GbnTile *d_tiles = nullptr; int64_t ntiles = 0; std::vector<int64_t> first_tile_of_subj; int64_t bases = 0; };

struct TileKey { int lut, step, tpos; int32_t s0, s1; bool operator<(const TileKey &o) const {
    return std::tie(lut, step, tpos, s0, s1) < std::tie(o.lut, o.step, o.tpos, o.s0, o.s1); } };
typedef std::map<TileKey, TileSet> TileCache;

static int build_tiles_uncached(const GbnDb &db, int lut, int step, int tpos, int32_t s0, int32_t s1, TileSet &ts) {
    std::vector<GbnTile> tiles;
    ts.first_tile_of_subj.clear(); ts.bases = 0;
    for (int32_t s = s0; s < s1; s++) {
        ts.first_tile_of_subj.push_back((int64_t)tiles.size());
        int32_t L = db.len[s];
        ts.bases += L;
        if (L < lut) continue;
        int32_t npos = (L - lut) / step + 1;
        for (int32_t p = 0; p < npos; p += tpos) {
            GbnTile t; t.subj = s; t.first_pos = p * step; t.npos = std::min(tpos, npos - p); t.off16 = (int32_t)(db.byte_off[s] >> 4);
            tiles.push_back(t);
        }
    }
    ts.first_tile_of_subj.push_back((int64_t)tiles.size());
    ts.ntiles = (int64_t)tiles.size();
    return dev_upload(ts.d_tiles, tiles.data(), tiles.size());
}

// tile tables depend only on (lut, step, subject range): cached on the shard
static int get_tiles(GbnDb &db, int lut, int step, int tpos, int32_t s0, int32_t s1, const TileSet **out) {
    if (!db.tile_cache) db.tile_cache = new TileCache();
    TileCache &tc = *static_cast<TileCache *>(db.tile_cache);
    TileKey k{lut, step, tpos, s0, s1};
    auto it = tc.find(k);
    if (it == tc.end()) {
        TileSet ts;
        int rc = build_tiles_uncached(db, lut, step, tpos, s0, s1, ts);
        if (rc) return rc;
        it = tc.emplace(k, std::move(ts)).first;
    }
    *out = &it->second;
    return GBN_OK;
}
static void free_tile_cache(GbnDb &db) {
    if (!db.tile_cache) return;
    TileCache *tc = static_cast<TileCache *>(db.tile_cache);
    for (auto &kv : *tc) dev_free(kv.second.d_tiles);
    delete tc; db.tile_cache = nullptr;
}

static int grow_seed_buffers(size_t want) {
    if (want <= E.seed_cap) return GBN_OK;
    dev_free(E.seeds);
    int rc = dev_alloc(E.seeds, want);
    if (rc) { E.seed_cap = 0; return rc; }
    E.seed_cap = want;
    return GBN_OK;
}
static int grow_key_buffers(Engine::KeySet &KS, size_t n) {
    if (n <= KS.key_cap) return GBN_OK;
    dev_free(KS.key_a); dev_free(KS.key_b); dev_free(KS.idx_a); dev_free(KS.idx_b);
    dev_free(KS.cell_diag); dev_free(KS.cell_level); dev_free(KS.ext_rec); dev_free(KS.sort_tmp);
    size_t cap = std::max<size_t>(n + n / 8, 1 << 16);      // (room to spare: seed counts of consecutive ranges differ by a fraction of a percent, and a regrow frees and allocates gigabytes)
    int rc;
    if ((rc = dev_alloc(KS.key_a, cap)) || (rc = dev_alloc(KS.key_b, cap)) || (rc = dev_alloc(KS.idx_a, cap)) ||
        (rc = dev_alloc(KS.idx_b, cap)) || (rc = dev_alloc(KS.cell_diag, cap)) || (rc = dev_alloc(KS.cell_level, cap)))
        return rc;
    const size_t compact_min = (size_t)gbn::switch_value("GBN_DIAG_COMPACT_MIN", (long long)GBN_DIAG_COMPACT_MIN);
    if (cap >= compact_min && (rc = dev_alloc(KS.ext_rec, cap * 8))) return rc;
    size_t bytes = 0;
    HIPCHK(sort_pairs_u64(nullptr, bytes, KS.key_a, KS.key_b, KS.idx_a, KS.idx_b, (int64_t)cap, 64, E.stream));
    size_t bytes_keys = 0;
    HIPCHK(sort_keys_u64(nullptr, bytes_keys, KS.key_a, KS.key_b, (int64_t)cap, 0, 64, E.stream));
    bytes = std::max(bytes, bytes_keys);
    HIPCHK(pool_alloc(&KS.sort_tmp, bytes));
    KS.sort_tmp_bytes = bytes; KS.key_cap = cap;
    return GBN_OK;
}

static int grow_ihit_buffers(int slot, size_t n) {
    if (n <= E.ihit_cap_s[slot]) return GBN_OK;
    dev_free(E.ihits_s[slot]); dev_free(E.gapped_s[slot]);
    size_t cap = std::max<size_t>(n, 1 << 14);
    int rc;
    if ((rc = dev_alloc(E.ihits_s[slot], cap)) || (rc = dev_alloc(E.gapped_s[slot], cap))) return rc;
    E.ihit_cap_s[slot] = cap;
    return GBN_OK;
}

// wait for the extension stage that is still in flight (if any).  Its failure belongs to the results it was
// filling, not to whoever happens to wait for it: it is kept in E.failed and returned by
// gbn_prelim_search_end(those results) (take_failure).
static void record_failure(const GbnResults *res, int rc, const std::string &what) {
    std::lock_guard<std::mutex> lk(E.failed_mu);
    if (!E.failed.count(res)) E.failed[res] = std::make_pair(rc, what);
}
// the device side of the stage in flight: its kernels and copies are done, its device buffers free again (the host
// replay of its extensions may still be running: wait_host)
static int wait_pending_gpu() {
    if (!E.has_pending) return GBN_OK;
    int rc = E.pending.get();
    if (rc) record_failure(E.pending_res, rc, E.pending_err);
    E.has_pending = false; E.pending_ks = -1; E.pending_batch = nullptr;
    return GBN_OK;
}
static void wait_host() {
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(E.host_mu); f = E.host_tail; }
    if (f.valid()) f.wait();
}
// the host replays queued up to the one `tail` stands for (a batch's or a set of results' last: replays run in the order
// they were queued, so theirs are done when that one is) -- not the replays of LATER searches, which a pipelined caller
// has queued behind them (round 4: gbn_prelim_search_end / gbn_batch_free of pass k waited for the replays of pass k + 1,
// with the engine locked: 3.4 ms per blastn pass in which the next pass's scan could not be queued)
static void wait_tail(std::shared_future<void> &slot) {
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(E.host_mu); f = slot; }
    if (f.valid()) f.wait();
    std::lock_guard<std::mutex> lk(E.host_mu);
    if (slot.valid() && slot.wait_for(std::chrono::seconds(0)) == std::future_status::ready) slot = std::shared_future<void>();
}
static int wait_pending() {
    const int rc = wait_pending_gpu();
    wait_host();
    return rc;
}
static int take_failure(const GbnResults *res) {
    std::lock_guard<std::mutex> lk(E.failed_mu);
    auto it = E.failed.find(res);
    if (it == E.failed.end()) return GBN_OK;
    const int rc = it->second.first;
    if (!it->second.second.empty()) set_error(it->second.second);
    E.failed.erase(it);
    return rc;
}

// ints of scratch one thread of the gapped kernels needs (GbnGapParams::scratch_per_thread) and the greedy row length
static int64_t gap_scratch_ints(const GbnBatch &b, int32_t max_len, int32_t max_ctx, int32_t *row_len) {
    size_t per_thread;
    *row_len = 0;
    if (b.opt.greedy) {
        int32_t max_dist = std::min(10000, max_len / 2 + 1);
        int32_t X2 = (b.opt.reward % 2 == 1) ? 2 * b.gap_x_dropoff : b.gap_x_dropoff;
        int32_t mc = (b.opt.reward % 2 == 1) ? 2 * b.opt.reward : b.opt.reward;
        int32_t mm = (b.opt.reward % 2 == 1) ? -2 * b.opt.penalty : -b.opt.penalty;
        int32_t xoff = (X2 + mc / 2) / (mc + mm) + 1;
        *row_len = 2 * max_dist + 8;
        per_thread = 2 * (size_t)*row_len + (size_t)max_dist + 4 + (size_t)xoff;
        if (!(b.opt.gap_open == 0 && b.opt.gap_extend == 0)) {
            // affine greedy: (max_penalty + 1) rows of 3 offsets, diagonal bounds and max_score per scaled distance
            int32_t go = b.opt.gap_open, ge = b.opt.gap_extend;
            if (b.opt.reward % 2 == 1) { go *= 2; ge *= 2; }
            int32_t half = mc / 2, opc = mc + mm, gex = ge + half;
            auto gcd2 = [](int a, int c) { c = std::abs(c); if (c > a) std::swap(a, c); while (c) { int t = a % c; a = c; c = t; } return a; };
            int32_t g = go == 0 ? gcd2(opc, gex) : gcd2(opc, gcd2(go, gex));
            if (g > 1) { opc /= g; go /= g; gex /= g; }
            int32_t max_penalty = std::max(opc, go + gex), scaled = max_dist * gex, xo = (X2 + half) / g + 1;
            per_thread = (size_t)(max_penalty + 1) * *row_len * 3 + 2 * (size_t)(scaled + 1 + max_penalty) + (size_t)scaled + 4 + xo;
        }
    } else {
        per_thread = 2 * ((size_t)max_ctx + 16);
    }
    return (int64_t)((per_thread + 3) & ~(size_t)3);
}

static void fill_scan_params(GbnScanParams &P, const GbnBatch &b, const GbnDb &db, const TileSet &ts) {
    const DeviceBatch *d = b.dev;
    std::memset(&P, 0, sizeof(P));
    P.db = db.d_packed; P.byte_off = db.d_byte_off; P.len = db.d_len;
    P.tiles = ts.d_tiles; P.ntiles = ts.ntiles;
    P.pv = d->pv; P.cellw = d->cellw; P.cell_start = d->cell_start; P.ent = d->ent; P.pvx = d->pvx; P.pstart = d->pstart;
    P.ncells = b.lut.ncells; P.lut = b.lut.lut; P.word = b.lut.word; P.step = b.lut.step;
    P.mode = d->mode; P.fl = d->fl; P.fr = d->fr;
    P.q8 = d->q8; P.qlen = b.qlen; P.ctx_off = d->ctx_off; P.ctx_len = d->ctx_len; P.nctx = (int32_t)b.ctx.size();
    P.seeds = E.seeds; P.seed_count = E.counters; P.seed_cap = E.seed_cap; P.raw_hits = E.counters + 1;
}

static int scan_grid(int64_t ntiles) {
    int64_t g = (int64_t)E.num_cu * 8;      // 8 resident 256-thread workgroups per CU
    return (int)std::max<int64_t>(1, std::min(ntiles, g));
}

// number of key-range bins of the partitioned scan: 2^GBN_BIN_CBITS(lut) cells per bin (one LDS-resident
// slice of the cell table), 512 bins for every table from lut 8 up.  GBN_SCAN_BINS=1 forces the direct kernel.
static int choose_bins(const GbnBatch &b) {
    int64_t nb = b.lut.ncells >> GBN_BIN_CBITS(b.lut.lut);
    if (nb < 2 || nb > GBN_BIN_MAXNB) nb = 1;
    if (gbn::switch_value("GBN_SCAN_BINS", 0) == 1) nb = 1;
    return (int)nb;
}

// slices scan_slice_kernel would cut this batch's presence bits into (0: another kernel scans for this batch)
static int scan_slices(const GbnBatch &b) {
    const bool on = gbn::switch_value("GBN_SCAN_SLICE", 1) != 0;
    if (!on || !b.dev || choose_bins(b) == 1) return 0;     // (tables of one bin: the direct kernel as before; GBN_SCAN_BINS=1 forces it)
    GbnScanParams P; std::memset(&P, 0, sizeof(P));
    P.mode = b.dev->mode; P.step = b.lut.step; P.lut = b.lut.lut; P.word = b.lut.word; P.ncells = b.lut.ncells;
    return scan_slice_count(P);
}

// ---- record cache (Engine::rec_sets) ----
static void recset_free(RecordSet &r) {
    dev_free(r.bin_rec); dev_free(r.bin_tcur); dev_free(r.bin_count);
    r.bin_rec_cap = r.bin_tcur_cap = r.bin_count_cap = 0; r.complete = false; r.queued = false;
}
// the buffers of `r` at least this long (freed and allocated anew when one is too short: whatever they held is gone)
static int recset_size(RecordSet &r, size_t need_u64, size_t need_tcur, size_t need_count) {
    int rc;
    if (need_u64 > r.bin_rec_cap) { dev_free(r.bin_rec); r.bin_rec_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_rec, need_u64))) return rc; r.bin_rec_cap = need_u64; }
    if (need_tcur > r.bin_tcur_cap) { dev_free(r.bin_tcur); r.bin_tcur_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_tcur, need_tcur))) return rc; r.bin_tcur_cap = need_tcur; }
    if (need_count > r.bin_count_cap) { dev_free(r.bin_count); r.bin_count_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_count, need_count))) return rc; r.bin_count_cap = need_count; }
    return GBN_OK;
}
// bytes the cache may hold: gbn_record_cache_set_limit, else GBN_RECORD_CACHE_MB, else a quarter of the device's memory
static long long rec_limit_bytes() {
    if (E.rec_limit >= 0) return E.rec_limit;
    if (gbn::switch_is_set("GBN_RECORD_CACHE_MB")) return std::max(0ll, gbn::switch_value("GBN_RECORD_CACHE_MB", 0)) << 20;
    static thread_local long long dflt[kMaxDevices];        // (per device; the query costs a driver call)
    long long &d = dflt[E.device >= 0 && E.device < kMaxDevices ? E.device : 0];
    if (d == 0) { size_t fr = 0, tot = 0; d = hipMemGetInfo(&fr, &tot) == hipSuccess && tot ? (long long)(tot / 4) : (64ll << 30); }
    return d;
}
static size_t rec_held_bytes() { size_t n = 0; for (const RecordSet *r : E.rec_sets) n += r->bytes(); return n; }
static void rec_drop(size_t i, bool evicted) {
    if (E.rec_sets[i]->queued) (void)hipStreamSynchronize(E.stream);       // (a binning kernel queued by gbn_db_prepare_records may still write it)
    recset_free(*E.rec_sets[i]); delete E.rec_sets[i]; E.rec_sets.erase(E.rec_sets.begin() + (long)i);
    if (evicted) E.rec_evictions++;
}
// buffers change hands (what `dst` had is freed); neither side holds records afterwards
static void recset_move(RecordSet &dst, RecordSet &src) {
    if (src.queued || dst.queued) (void)hipStreamSynchronize(E.stream);
    recset_free(dst);
    dst.bin_rec = src.bin_rec; dst.bin_rec_cap = src.bin_rec_cap; dst.bin_tcur = src.bin_tcur; dst.bin_tcur_cap = src.bin_tcur_cap;
    dst.bin_count = src.bin_count; dst.bin_count_cap = src.bin_count_cap; dst.complete = false; dst.queued = false; src.queued = false;
    src.bin_rec = nullptr; src.bin_tcur = nullptr; src.bin_count = nullptr; src.bin_rec_cap = src.bin_tcur_cap = src.bin_count_cap = 0; src.complete = false;
}
// Sets go until `need` more bytes fit under `limit` (keep: the set the pass is using).  Which: a pass over (shard, range) of
// a table shape is one step of a SWEEP -- every query batch visits the ranges / block views of its database in the same
// order, again and again -- and under such a cyclic pattern "least recently used first" evicts exactly the set that is needed
// next (no hits at all once the sets of a sweep exceed the limit).  So, as buffer managers do for sequential scans: among the
// sets of the pass's own database shape (same lut width, stride and stream geometry: its sweep) the MOST recently used one goes
// -- the sets from the start of the sweep stay and are hit again by the next batch --; only when there is none, the least
// recently used of the others.  If `into` is given and empty, the last victim's buffers move there instead of being freed.
static void rec_make_room(size_t need, long long limit, const RecordSet *keep, const RecKey *sweep = nullptr, RecordSet *into = nullptr) {
    while (!E.rec_sets.empty() && (long long)(rec_held_bytes() + need) > limit) {
        const size_t none = E.rec_sets.size();
        size_t mru = none, lru = none;
        for (size_t i = 0; i < E.rec_sets.size(); i++) {
            const RecordSet *r = E.rec_sets[i];
            if (r == keep || r == into) continue;
            const bool same_sweep = sweep && r->key.lut == sweep->lut && r->key.step == sweep->step && r->key.nb == sweep->nb && r->key.cbits == sweep->cbits &&
                                    r->key.rfl == sweep->rfl && r->key.rfrbits == sweep->rfrbits;
            if (same_sweep) { if (mru == none || r->stamp > E.rec_sets[mru]->stamp) mru = i; }
            else if (lru == none || r->stamp < E.rec_sets[lru]->stamp) lru = i;
        }
        const size_t victim = lru != none ? lru : mru;      // (sets of other shapes: nobody is sweeping them now)
        if (victim == none) break;
        if (into && !into->bin_rec && (long long)(rec_held_bytes() - E.rec_sets[victim]->bytes() + std::max(need, E.rec_sets[victim]->bytes())) <= limit) {
            recset_move(*into, *E.rec_sets[victim]);        // (the room it makes is the room the newcomer takes: no driver call)
            need = need > into->bytes() ? need - into->bytes() : 0;
        }
        rec_drop(victim, true);
    }
}
// the shard goes (gbn_db_free), or everything (release, a limit of 0; to_scratch: the cache was switched off and the largest
// set's buffers become the passes' own -- no gigabytes freed and allocated again)
static void rec_purge(const void *db, bool to_scratch = false) {
    if (to_scratch && !db && !E.scratch.bin_rec && !E.rec_sets.empty()) {
        size_t big = 0;
        for (size_t i = 1; i < E.rec_sets.size(); i++) if (E.rec_sets[i]->bytes() > E.rec_sets[big]->bytes()) big = i;
        recset_move(E.scratch, *E.rec_sets[big]);
    }
    for (size_t i = E.rec_sets.size(); i-- > 0; ) if (!db || E.rec_sets[i]->key.db == db) rec_drop(i, false);
    if (!db || E.scratch.key.db == db) E.scratch.complete = false;
    if (!db || E.alt.key.db == db) E.alt.complete = false;
}

// ---- the partitioned scan's record streams: a private output stream per (bin, binning workgroup), no reservation atomics
struct BinLayout { int nb = 0, nwriters = 0; size_t nstream = 0, subcap = 0, nseq = 0, need_u64 = 0;
                   size_t bytes() const { return need_u64 * 8 + nstream * nseq * 4 + (nstream + 4) * 4; } };
static int64_t bin_positions(const GbnDb &db, int32_t s0, int32_t s1, int lut, int step) {
    int64_t npos = 0;
    for (int32_t s = s0; s < s1; s++) if (db.len[s] >= lut) npos += (db.len[s] - lut) / step + 1;
    return npos;
}
static int bin_layout(int nb, int64_t ntiles, int64_t npos, double slack, BinLayout &L) {
    L.nb = nb;
    L.nwriters = (int)std::max<int64_t>(8, std::min<int64_t>((int64_t)E.num_cu * GBN_BIN_WG_PER_CU, ntiles));
    if (gbn::switch_is_set("GBN_BIN_WRITERS")) L.nwriters = std::max(8, std::min(L.nwriters, (int)gbn::switch_value("GBN_BIN_WRITERS", 0)));     // experiments
    L.nstream = (size_t)nb * L.nwriters;
    const double expect = (double)npos / (double)L.nstream + 2.0 * GBN_OPEN_LINE;   // + the pads of the stream's last line
    L.subcap = (size_t)(expect * slack) + 256;
    L.subcap = (L.subcap + 511) & ~(size_t)511;       // whole record blocks, whole probe pieces
    if (L.subcap > 0x7ffffff0u) { set_error("bin capacity overflow: split the range"); return GBN_ERR_NOMEM; }
    L.nseq = ((size_t)((ntiles + L.nwriters - 1) / L.nwriters) + ((size_t)1 << GBN_TCUR_SHIFT) - 1) >> GBN_TCUR_SHIFT;    // cursor entries per stream
    L.need_u64 = (GBN_REC_WORDS(L.subcap * L.nstream) + 1) / 2;
    return GBN_OK;
}
// the cached set that serves `key`: complete (or being written on the engine's stream), streams at least as long
static RecordSet *rec_find(const RecKey &key) {
    for (RecordSet *c : E.rec_sets) if ((c->complete || c->queued) && c->key.same_shape(key) && c->key.subcap >= key.subcap) return c;
    return nullptr;
}
// a set to bin `key` into, its buffers sized: a cached one (room made for it) or -- larger than the whole cache -- the passes'
// own scratch set
static int rec_acquire(const RecKey &key, const BinLayout &L, long long limit, RecordSet **out) {
    // a set of this shape that holds no complete records (forgotten: gbn_record_cache_invalidate; overflowed) or whose
    // streams are shorter: its buffers serve again
    RecordSet *old = nullptr, *rs = nullptr;
    for (size_t i = E.rec_sets.size(); i-- > 0; ) if (E.rec_sets[i]->key.same_shape(key)) { if (!old) old = E.rec_sets[i]; else rec_drop(i, false); }
    if ((long long)L.bytes() <= limit) {
        if (old) rs = old;
        else {
            rs = new RecordSet(); E.rec_sets.push_back(rs);
            // (the cache was switched on after passes that binned for themselves: their buffers are the first set's)
            if (E.scratch.bin_rec) { recset_move(*rs, E.scratch); recset_free(E.alt); }
        }
        rec_make_room(L.bytes() > rs->bytes() ? L.bytes() - rs->bytes() : 0, limit, rs, &key, rs);
    } else {                                         // larger than the whole cache: this pass's own
        if (old) for (size_t i = 0; i < E.rec_sets.size(); i++) if (E.rec_sets[i] == old) { rec_drop(i, false); break; }
        rs = &E.scratch; E.rec_bypass++;
    }
    int rc = recset_size(*rs, L.need_u64, L.nstream * L.nseq, L.nstream + 4);
    if (rc == GBN_ERR_NOMEM && rs != &E.scratch) {      // the device is full: everything else the cache holds goes, once
        rec_make_room((size_t)limit, limit, rs);
        rc = recset_size(*rs, L.need_u64, L.nstream * L.nseq, L.nstream + 4);
    }
    if (rc) { if (rs != &E.scratch) { for (size_t i = 0; i < E.rec_sets.size(); i++) if (E.rec_sets[i] == rs) { rec_drop(i, false); break; } } return rc; }
    rs->key = key; rs->complete = false; rs->queued = false;
    *out = rs;
    return GBN_OK;
}

// one scan of the subjects [s0, s1): fills E.seeds / cnt[0] seeds, cnt[1] raw hits;
// dispatches to the direct-probe kernel (small tables) or the partitioned pair
static int run_scan_impl(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
                         unsigned long long cnt[2], int64_t *bases_out, bool direct, bool *skewed);

// The partitioned scan sizes its streams for lookup words spread evenly over the bins (x1.25, x2.5).
// Subjects dominated by one repeat (satellite arrays, poly-A) put most positions of a range into a
// few bins; such a range goes through the direct-probe kernel instead, which has no streams.
static const int kSkewedRange = -1000;       // internal: split this subject range and try again

static int run_scan(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
                    unsigned long long cnt[2], int64_t *bases_out)
{
    bool skewed = false;
    int rc = run_scan_impl(b, db, s0, s1, diag, cnt, bases_out, false, &skewed);
    if (rc != GBN_OK || !skewed) return rc;
    int64_t bases = 0;
    for (int32_t s = s0; s < s1; s++) bases += db.len[s];
    int64_t split_mb = 256;
    if (gbn::switch_is_set("GBN_SKEW_SPLIT_MB")) split_mb = (int)std::max<long long>(1, gbn::switch_value("GBN_SKEW_SPLIT_MB", 0));       // tests
    if (s1 - s0 > 1 && bases > (split_mb << 20)) return kSkewedRange;         // the caller halves the range
    return run_scan_impl(b, db, s0, s1, diag, cnt, bases_out, true, &skewed);
}

static int run_scan_impl(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
                         unsigned long long cnt[2], int64_t *bases_out, bool direct, bool *skewed)
{
    // tables as wide as the word (stride 1, every lookup hit a seed): the presence bits are sliced through the LDS
    // instead of the scan positions being written out by key range (scan_slice_kernel); GBN_SCAN_SLICE=0: off
    const bool sliced = !direct && scan_slices(b) > 0;
    E.seg_valid = false;
    const int nb = (direct || sliced) ? 1 : choose_bins(b);
    if (nb == 1 && E.ahead.valid) { E.ahead.valid = false; E.ahead_misses++; }       // (a scan of another kind: nobody will want the records binned ahead)
    if (nb == 1) E.last_key_valid = false;
    const TileSet *tsp = nullptr;
    int rc = get_tiles(db, b.lut.lut, b.lut.step, nb > 1 ? GBN_BIN_TILE_POS : GBN_TILE_POS, s0, s1, &tsp);
    if (rc) return rc;
    const TileSet &ts = *tsp;
    *bases_out = ts.bases;
    cnt[0] = cnt[1] = 0;
    if (ts.ntiles == 0) return GBN_OK;
    if (nb > 1 && ts.ntiles > (1 << 19)) { set_error("subject range too large for 32-bit position ids"); return GBN_ERR_ARG; }
    if ((rc = grow_seed_buffers(std::max<size_t>(E.seed_cap, (size_t)1 << 22)))) return rc;
    if (!E.scan_back) { HIPCHK(hipHostMalloc((void **)&E.scan_back, sizeof(*E.scan_back))); std::memset(E.scan_back, 0, sizeof(*E.scan_back)); }
    const int64_t npos = nb > 1 ? bin_positions(db, s0, s1, b.lut.lut, b.lut.step) : 0;
    double slack = 1.25;
    size_t rare_seg_hint = 0, rare_seg_used = 0, slice_seg_cap = 0; int slice_blocks = 0; bool slice_ordered = false;
    GbnBinParams last_B; std::memset(&last_B, 0, sizeof(last_B)); int last_grid2 = 0;
    const long long rec_limit = nb > 1 ? rec_limit_bytes() : 0;        // bytes the record cache may hold; 0: off
    if (rec_limit == 0 && nb > 1 && !E.rec_sets.empty()) rec_purge(nullptr, true);     // (switched off: what it held goes)
    RecordSet *rs = nullptr;                    // the records of this pass
    bool binned_here = false;                   // ... were written (completely) by this call
    bool repeat_seen = false;                   // cache off: the pass before this one had the same key
    if (E.seed_copy_pending) { HIPCHK(hipStreamWaitEvent(E.stream, E.ev_seed, 0)); E.seed_copy_pending = false; }
    for (;;) {
        bool binned_ahead = false; int hit_pair = -1;
        if (!E.counters_zeroed) HIPCHK(hipMemsetAsync(E.counters, 0, 4 * sizeof(unsigned long long), E.stream));    // (a pass that binned ahead zeroed them behind its read-back)
        E.counters_zeroed = false;
        GbnScanParams P; fill_scan_params(P, b, db, ts);
        uint32_t overflow = 0; int dbg_nwriters = 0; uint32_t dbg_subcap = 0;
        bool binned = false;
        if (nb == 1) {
            HIPCHK(hipEventRecord(E.ev0, E.stream));
            if (b.dev->ready) HIPCHK(hipStreamWaitEvent(E.stream, b.dev->ready, 0));
            if (sliced) {
                // every workgroup writes its seeds into a segment of its own (no global counter), a second kernel puts
                // the segments back to back.  Segments: 1.5 x the seeds a random subject gives, twice as long after an overflow
                // (with the seeds in scan order a segment belongs to a wave, sixteen per workgroup)
                int ordered = 0;
                const int blocks = scan_slice_segments(P, E.num_cu, &ordered);
                slice_blocks = blocks; slice_ordered = ordered != 0;
                if (slice_seg_cap == 0) {
                    int64_t np = 0;
                    for (int32_t s = s0; s < s1; s++) if (db.len[s] >= b.lut.lut) np += db.len[s] - b.lut.lut + 1;
                    const double expect = (double)np * std::min(1.0, (double)b.qlen / (double)b.lut.ncells) / blocks;
                    slice_seg_cap = (size_t)(expect * 1.5) + (ordered ? 1024 : 8192);
                }
                if (slice_seg_cap > 0x7fffff00u) { set_error("too many seeds in one range"); return GBN_ERR_NOMEM; }
                const size_t need = slice_seg_cap * (size_t)blocks;
                if (need > E.slice_seg_cap) {
                    dev_free(E.slice_seg); E.slice_seg_cap = 0;
                    if ((rc = dev_alloc(E.slice_seg, need + need / 8))) return rc;
                    E.slice_seg_cap = need + need / 8;
                }
                if (!E.seg_counts && ((rc = dev_alloc(E.seg_counts, (size_t)GBN_SLICE_SEGS)) || (rc = dev_alloc(E.seg_firsts, (size_t)GBN_SLICE_SEGS + 1)))) return rc;
                HIPCHK(launch_scan_slice(P, E.num_cu, E.slice_seg, (uint32_t)slice_seg_cap, E.seg_counts, E.counters + 2, E.stream));
            } else HIPCHK(launch_scan_seed(P, scan_grid(ts.ntiles), E.stream));
            HIPCHK(hipEventRecord(E.ev1, E.stream));
        } else {
            BinLayout BL;
            if ((rc = bin_layout(nb, ts.ntiles, npos, slack, BL))) return rc;
            const int nwriters = BL.nwriters; const size_t nstream = BL.nstream, nseq = BL.nseq;
            size_t subcap = BL.subcap;
            const int rfl_now = std::min(4, b.dev->fl), rfrbits_now = std::min(7, 2 * b.dev->fr);
            RecKey key; key.db = (const void *)&db; key.s0 = s0; key.s1 = s1; key.lut = b.lut.lut; key.step = b.lut.step; key.nb = nb; key.nwriters = nwriters;
            key.rfl = rfl_now; key.rfrbits = rfrbits_now; key.cbits = GBN_BIN_CBITS(b.lut.lut); key.tiles = (const void *)P.tiles; key.subcap = subcap;
            bool hit = false, ahead_hit = false;
            Engine::BinAhead &AH = E.ahead;
            if (rec_limit > 0) {
                // ---- record cache: a complete set of this shape whose streams are at least as long as this attempt asks for
                if (AH.valid) { AH.valid = false; E.ahead_misses++; HIPCHK(hipStreamSynchronize(E.stream)); }     // (a kernel queued ahead writes the other scratch set, which may change hands below)
                rs = rec_find(key);
                if (rs) { hit = true; subcap = rs->key.subcap; key.subcap = subcap; if (!binned_here) E.rec_hits++; }
                else { E.rec_misses++; if ((rc = rec_acquire(key, BL, rec_limit, &rs))) return rc; }
            } else {
                rs = &E.scratch;
                ahead_hit = AH.valid && AH.key == key;
                if (AH.valid) {
                    AH.valid = false;
                    if (ahead_hit) { E.swap_scan_sets(); E.ahead_hits++; hit_pair = AH.pair; }     // the records of this pass are in the other set: that one is the current set now
                    else E.ahead_misses++;
                }
                // (a rare-path segment overflowed and the range is scanned again: the records this call wrote are still there)
                hit = !ahead_hit && binned_here && rs->complete && rs->key == key;
                repeat_seen = E.last_key_valid && E.last_key == key;
                E.last_key = key; E.last_key_valid = true;
                if (!hit && !ahead_hit) {
                    if ((rc = recset_size(*rs, BL.need_u64, nstream * nseq, nstream + 4))) return rc;
                    rs->key = key; rs->complete = false;
                }
            }
            if (!hit && !ahead_hit) HIPCHK(hipMemsetAsync(rs->bin_count + nstream, 0, 16, E.stream));     // (records that exist already: the flag of THAT launch is read back below)
            rs->stamp = ++E.rec_clock;
            GbnBinParams B; std::memset(&B, 0, sizeof(B));
            B.S = P; B.nb = nb; B.cbits = GBN_BIN_CBITS(b.lut.lut); B.nwriters = nwriters; dbg_nwriters = nwriters; dbg_subcap = (uint32_t)subcap;
            B.cellt = b.dev->cellt; B.sidet = b.dev->sidet; B.side_start = b.dev->side_start; B.rfl = std::min(4, b.dev->fl); B.rfrbits = std::min(7, 2 * b.dev->fr);
            B.rec = reinterpret_cast<uint32_t *>(rs->bin_rec); B.tcur = rs->bin_tcur; B.nseq = (uint32_t)nseq; B.gcount = rs->bin_count; B.subcap = (uint32_t)subcap;
            B.overflow = rs->bin_count + nstream;
            B.dbg = (int)gbn::switch_value("GBN_DBG", 0);
            int grid2 = std::max(8, E.num_cu & ~7);   // one 1024-thread workgroup per CU; group = blockIdx & 7
            {   // rare-path queue: one segment per probe workgroup (~1.2 % of scan positions in total)
                size_t seg = std::max<size_t>(rare_seg_hint, (size_t)(npos / 40 / grid2) + 4096);
                size_t want = seg * (size_t)grid2;
                if (want > E.rareq_cap) {
                    dev_free(E.rareq); E.rareq_cap = 0; dev_free(E.rare_counts);
                    if ((rc = dev_alloc(E.rareq, want))) return rc;
                    E.rareq_cap = want;
                }
                seg = E.rareq_cap / (size_t)grid2;
                // tests: GBN_RARE_SEG=n starts with segments of n items, so that a small search overflows them and takes the
                // way a repeat-rich range takes at full size (scan again with the room the counts ask for)
                const long long seg_sw = gbn::switch_value("GBN_RARE_SEG", 0);
                if (seg_sw > 0 && !rare_seg_hint) seg = std::min<size_t>(seg, (size_t)seg_sw);
                rare_seg_used = seg;
                if (!E.rare_counts && (rc = dev_alloc(E.rare_counts, (size_t)2048))) return rc;
                B.rareq = E.rareq; B.rare_seg = (uint32_t)std::min<size_t>(seg, 0x7fffffff); B.rare_counts = E.rare_counts;
            }
            last_B = B; last_grid2 = grid2;
            HIPCHK(launch_scan_bin_parts(B, grid2, E.stream, E.evk, ((hit || ahead_hit) ? 2 : 3) | 4 | (ahead_hit ? 8 : 0), b.dev->ready));
            binned = true; binned_ahead = ahead_hit;
            HIPCHK(hipMemcpyAsync(&E.scan_back->overflow, B.overflow, 4, hipMemcpyDeviceToHost, E.stream));
            HIPCHK(hipMemcpyAsync(E.scan_back->rare_counts, E.rare_counts, (size_t)grid2 * 4, hipMemcpyDeviceToHost, E.stream));
        }
        HIPCHK(hipMemcpyAsync(E.scan_back->cnt, E.counters, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, E.stream));     // seeds, raw hits, the fullest segment
        if (!E.ev_back) HIPCHK(hipEventCreate(&E.ev_back));
        HIPCHK(hipEventRecord(E.ev_back, E.stream));
        if (binned && rec_limit == 0 && E.want_ahead && repeat_seen && slack <= 1.25) {
            // the next pass's binning kernel, into the other set (sized like this one)
            RecordSet &A = E.alt;
            const size_t nstream = (size_t)last_B.nb * (size_t)last_B.nwriters;
            if (recset_size(A, rs->bin_rec_cap, nstream * last_B.nseq, nstream + 4) == GBN_OK) {        // (no room for a second set: no binning ahead)
                GbnBinParams A2 = last_B;
                A2.rec = reinterpret_cast<uint32_t *>(A.bin_rec); A2.tcur = A.bin_tcur; A2.gcount = A.bin_count; A2.overflow = A.bin_count + nstream;
                A2.rareq = nullptr;                         // (the binning kernel queues nothing; rare_counts: where a GBN_BIN_TIMING build leaves its clocks)
                A.key = rs->key; A.complete = false;
                Engine::BinAhead &AH = E.ahead;
                if (!AH.ev[0][0]) for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&AH.ev[i >> 1][i & 1]));
                AH.pair = hit_pair >= 0 ? (hit_pair ^ 1) : (AH.pair ^ 1);
                HIPCHK(hipMemsetAsync(A.bin_count + nstream, 0, 16, E.stream));
                HIPCHK(hipMemsetAsync(E.counters, 0, 4 * sizeof(unsigned long long), E.stream)); E.counters_zeroed = true;      // (read back above; the next scan's)
                HIPCHK(hipEventRecord(AH.ev[AH.pair][0], E.stream));
                HIPCHK(launch_scan_bin_parts(A2, last_grid2, E.stream, nullptr, 1, nullptr));
                HIPCHK(hipEventRecord(AH.ev[AH.pair][1], E.stream));
                AH.valid = true; AH.key = rs->key;
            }
        }
        trace_mark("scan: kernels queued");
        HIPCHK(hipEventSynchronize(E.ev_back));
        trace_mark("scan: kernels done");
        cnt[0] = E.scan_back->cnt[0]; cnt[1] = E.scan_back->cnt[1];
        const unsigned long long seg_max = sliced ? E.scan_back->seg_max : 0;
        if (binned) { overflow = E.scan_back->overflow; rs->complete = overflow == 0; rs->queued = false; binned_here = rs->complete; }
        finish_build(b.dev);                                // (the scan has waited for the builder's event)
        if (diag) {
            float ms = 0, ahead_ms = 0;
            if (binned) (void)hipEventElapsedTime(&ms, E.evk[binned_ahead ? 1 : 0], E.evk[3]);     // (the launcher's own events bracket the stage)
            else (void)hipEventElapsedTime(&ms, E.ev0, E.ev1);
            if (binned_ahead && hit_pair >= 0) (void)hipEventElapsedTime(&ahead_ms, E.ahead.ev[hit_pair][0], E.ahead.ev[hit_pair][1]);    // this pass's binning kernel ran ahead
            diag->scan_kernel_ms += ms + ahead_ms; diag->scan_launches++;
            if (binned) {
                float a = 0, c = 0, r = 0;
                if (!binned_ahead) (void)hipEventElapsedTime(&a, E.evk[0], E.evk[1]);
                (void)hipEventElapsedTime(&c, E.evk[1], E.evk[2]);
                a += ahead_ms;
                (void)hipEventElapsedTime(&r, E.evk[2], E.evk[3]);
                diag->bin_kernel_ms += a; diag->probe_kernel_ms += c; diag->rare_kernel_ms += r;
            }
        }
        if (nb > 1) {
            const int grid2 = std::max(8, E.num_cu & ~7);
            unsigned long long sc = 0; uint32_t mx = 0;
            for (int i = 0; i < grid2; i++) { const uint32_t v = E.scan_back->rare_counts[i]; sc += v; mx = std::max(mx, v); }
            if (gbn::switch_is_set("GBN_DBG")) fprintf(stderr, "[gbn dbg] rare-path items %llu, seeds %llu, raw %llu\n", sc, cnt[0], cnt[1]);
            if (gbn::switch_value("GBN_DBG", 0) & 128) {    // where the probe workgroups ran: blockIdx & 7 against the XCD they report
                std::vector<uint32_t> x((size_t)grid2);
                HIPCHK(hipMemcpy(x.data(), E.rare_counts + 1024, (size_t)grid2 * 4, hipMemcpyDeviceToHost));
                int off = 0, per[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt8[8][8] = {{0}};
                for (int i = 0; i < grid2; i++) { cnt8[i & 7][x[(size_t)i] & 7u]++; per[x[(size_t)i] & 7u]++; }
                for (int g = 0; g < 8; g++) { int best = 0, tot = 0; for (int c = 0; c < 8; c++) { best = std::max(best, cnt8[g][c]); tot += cnt8[g][c]; } off += tot - best; }
                fprintf(stderr, "[gbn dbg] probe workgroups away from their group's XCD: %d of %d; per XCD %d %d %d %d %d %d %d %d\n", off, grid2, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
            }
            if (gbn::switch_value("GBN_DBG", 0) & 32) {
                {   // stream fill statistics
                    const size_t ns = (size_t)nb * (size_t)dbg_nwriters;
                    std::vector<uint32_t> gc(ns);
                    HIPCHK(hipMemcpy(gc.data(), rs->bin_count, ns * 4, hipMemcpyDeviceToHost));
                    uint32_t mn = ~0u, mx2 = 0; unsigned long long sum = 0;
                    for (uint32_t v : gc) { mn = std::min(mn, v); mx2 = std::max(mx2, v); sum += v; }
                    fprintf(stderr, "[gbn dbg] %zu streams: records min %u max %u total %llu (capacity %u each)\n", ns, mn, mx2, sum, dbg_subcap);
                }
                {   // wall clock of the binning workgroups (GBN_BIN_TIMING build)
                    std::vector<uint32_t> w(1024);
                    HIPCHK(hipMemcpy(w.data(), E.rare_counts + 1024, 4096, hipMemcpyDeviceToHost));
                    uint32_t s_min = ~0u, d_min = ~0u, d_max = 0, s_max = 0; const int nw = std::min(dbg_nwriters, 512);
                    for (int i = 0; i < nw; i++) s_min = std::min(s_min, w[i]);
                    for (int i = 0; i < nw; i++) { s_max = std::max(s_max, w[i] - s_min); d_min = std::min(d_min, w[512 + i]); d_max = std::max(d_max, w[512 + i]); }
                    if (gbn::switch_is_set("GBN_DBG_WG")) { for (int i = 0; i < nw; i++) fprintf(stderr, "%u%c", w[512 + i] / 100, (i & 31) == 31 ? '\n' : ' '); }
                    fprintf(stderr, "[gbn dbg] scan_bin workgroups: start spread %.1f us, duration min %.1f max %.1f us\n", s_max / 100.0, d_min / 100.0, d_max / 100.0);
                }
                uint32_t ph[24]; HIPCHK(hipMemcpy(ph, E.rare_counts + 512, sizeof(ph), hipMemcpyDeviceToHost));
                // four-barrier form (wave 0 only): [0] atomics + loads issued | wait A | [1] lines + scan | wait B0 | [2] descriptors | wait B | [3] scatter;
                // scan_bin3_body (waves 0 and 15): waiting records | scatter | wait (1) | keys | stores | loads issued | wait (2)
                for (int w = 0; w < 2; w++)
                    fprintf(stderr, "[gbn dbg] scan_bin workgroup 0 (GBN_BIN_TIMING build), cycles/16 of wave %d per phase: %u %u %u %u %u %u %u %u %u\n",
                            w ? 15 : 0, ph[12 * w], ph[12 * w + 1], ph[12 * w + 2], ph[12 * w + 3], ph[12 * w + 4], ph[12 * w + 5], ph[12 * w + 6], ph[12 * w + 7], ph[12 * w + 8]);
            }
            if ((size_t)mx > rare_seg_used) {                  // a segment overflowed: grow and rescan this range
                rare_seg_hint = (size_t)mx + (mx >> 2);
                continue;
            }
        }
        if (sliced && seg_max > slice_seg_cap) {            // a workgroup's segment was too short: seeds are missing
            slice_seg_cap = std::max<size_t>(2 * slice_seg_cap, (size_t)seg_max + (size_t)(seg_max >> 2));
            continue;
        }
        if (overflow) {     // the records are incomplete: once more with twice the room, then give the range to the direct kernel
            slack *= 2;
            if (slack > 3.0) { *skewed = true; return GBN_OK; }
            continue;
        }
        if (sliced) {       // the seeds sit in the workgroups' segments; E.seeds only has to be long enough for compact_seeds
            if (cnt[0] > E.seed_cap && (rc = grow_seed_buffers((size_t)cnt[0] + (cnt[0] >> 3)))) return rc;
            E.seg_valid = cnt[0] > 0; E.seg_n = slice_blocks; E.seg_len = (uint32_t)slice_seg_cap; E.seg_ordered = slice_ordered;
            break;
        }
        if (cnt[0] <= E.seed_cap) break;
        if ((rc = grow_seed_buffers((size_t)cnt[0] + (cnt[0] >> 3)))) return rc;
    }
    return GBN_OK;
}

// one range of subjects [s0, s1) through the whole pipeline
static int gapped_stage(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag,
                        int keep_stages, int slot, unsigned long long nih, hipStream_t st, bool detach_host = false);

// the seeds of the last scan in one array (E.seeds), for the consumers that do not read scan_slice_kernel's segments
static int compact_seeds(hipStream_t st) {
    if (!E.seg_valid) return GBN_OK;
    HIPCHK(launch_seed_compact(E.slice_seg, E.seg_counts, E.seg_firsts, E.seg_n, E.seg_len, E.seeds, E.seed_cap, st));
    E.seg_valid = false;
    return GBN_OK;
}

// seeds of a range -> scan order (two stable sorts) -> diagonal filter + ungapped extension on stream `st`;
// the initial hits are left in the slot's buffers.  ctr: [0] initial hits, [1] runs (device counters).
static int seed_stage(GbnBatch &b, GbnDb &db, GbnResults &res, GbnDiagnostics *diag, int keep_stages, int slot,
                      const GbnDevSeed *seeds, int64_t n, unsigned long long *ctr, hipStream_t st, unsigned long long *nih_out,
                      int32_t s0, int32_t s1, int ksi, int phase = 0)
{
    // phase 0: the whole stage on `st`.  1: keys + sort only (queued, nothing waited for); 2: extension + replay of what a
    // phase-1 call with the same arguments sorted into the same key set
    Engine::KeySet &KS = E.ks[ksi];
    const DeviceBatch *d = b.dev;
    if (s1 < 0) s1 = db.num_seqs;
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_stage = now();
    if (phase != 2 && (rc = grow_key_buffers(KS, (size_t)n))) return rc;
    GbnKeyParams K; std::memset(&K, 0, sizeof(K));
    K.seeds = seeds; K.n = n; K.key_scan = KS.key_a; K.idx = KS.idx_a;
    K.q_descending = (b.lut.type == GBN_LUT_MB); K.container_hash = b.container; K.diag_len = b.diag_len;
    // key widths: the radix sorts stop at the top bit a key can have
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    int32_t max_len = 1;
    for (int32_t s = 0; s < db.num_seqs; s++) max_len = std::max(max_len, db.len[s]);
    K.q_bits = std::min(32, bits_for((uint64_t)b.qlen + 1));
    K.group_bits = b.container ? 9 : bits_for((uint64_t)std::max(b.diag_len, 2));
    const int scan_bits = std::min(64, K.q_bits + bits_for((uint64_t)max_len + 1));
    const int group_key_bits = std::min(64, K.group_bits + bits_for((uint64_t)db.num_seqs + 1));
    // Many seeds (blastn shapes): ONE sort of a composite key, the seed itself travels in the key (seed_ckeys_kernel) --
    // when subject | slot | s_scan | query key fit 64 bits; else, and for the few seeds of megablast shapes, two
    // stable sorts of (rank, index) pairs
    const int64_t compact_min = (int64_t)gbn::switch_value("GBN_DIAG_COMPACT_MIN", (long long)GBN_DIAG_COMPACT_MIN);
    const bool ck_on = gbn::switch_value("GBN_SEED_CKEYS", 1) != 0;
    K.s_bits = bits_for((uint64_t)max_len + 1); K.qh_bits = std::max(0, K.q_bits - K.group_bits);
    K.subj_base = s0;
    const int ck_bits = K.group_bits + bits_for((uint64_t)(s1 - s0) + 1) + K.s_bits;        // (the query key's high bits travel in the value)
    const bool composite = ck_on && KS.ext_rec && n >= compact_min && ck_bits <= 64 && K.group_bits < 32 && K.qh_bits <= 24 && b.lut.word - b.lut.lut < 256;
    const bool segmented = seeds == E.seeds && E.seg_valid;           // (an asynchronous stage works on a copy of its own)
    const bool from_segments = segmented && composite && !keep_stages;  // seed_ckeys_kernel reads the segments as they are
    if (phase != 2 && segmented && !from_segments && (rc = compact_seeds(st))) return rc;
    // Few seeds (megablast shapes: some 24 thousand per C2 pass): ONE workgroup sorts their indices by (subject, slot, scan
    // position, query key) in ONE launch (seed_sort.hip; GBN_SMALL_SORT=0: the two library sorts of rounds 1-4, which also
    // serve keep_stages -- it wants the scan order by itself -- and more than GBN_SMALL_SORT_MAX seeds)
    const bool small_sort = !composite && !keep_stages && gbn::switch_value("GBN_SMALL_SORT", 1) != 0 && seed_sort_small_fits(K, s1 - s0);
    if (phase != 2 && small_sort) {
        if (segmented && (rc = compact_seeds(st))) return rc;
        KS.kt.mark(GBN_KT_SORT, st);
        HIPCHK(launch_seed_sort_small(K, s1 - s0, KS.idx_a, KS.idx_b, KS.key_b, KS.key_a, st));
        KS.kt.mark(-1, st);
        // key_b = sorted (subject, slot) keys, idx_a = seed indices grouped by run, scan order inside
    }
    if (phase != 2 && !small_sort && (!composite || keep_stages)) {
        KS.kt.mark(GBN_KT_KEYS, st);
        HIPCHK(launch_seed_keys(K, st));
        size_t tb = KS.sort_tmp_bytes;
        KS.kt.mark(GBN_KT_SORT, st);
        HIPCHK(sort_pairs_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, KS.idx_a, KS.idx_b, n, scan_bits, st));
        KS.kt.mark(-1, st);
        // idx_b = seed indices in scan order (s_scan, chain order), subjects interleaved
    }
    if (keep_stages) {
        std::vector<GbnDevSeed> hs((size_t)n); std::vector<uint32_t> order((size_t)n);
        HIPCHK(hipMemcpyAsync(hs.data(), seeds, (size_t)n * sizeof(GbnDevSeed), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(order.data(), KS.idx_b, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<GbnSeed> tmp; tmp.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const GbnDevSeed &s = hs[order[i]];
            GbnSeed o; o.oid = db.oid_of(s.subj); o.s_off = s.s_scan - s.ext_left; o.q_off = s.q_pos - s.ext_left; o.pad_ = 0;
            tmp.push_back(o);
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const GbnSeed &a, const GbnSeed &c) { return a.oid < c.oid; });
        res.seeds.insert(res.seeds.end(), tmp.begin(), tmp.end());
    }
    // ... and when the value (ext_left and the query key's high bits) fits underneath the key too, it travels in the
    // key's low bits: a sort of keys only, on the bits above the value (GBN_SEED_CKEYS=2: always pairs)
    const bool ck_pack = gbn::switch_value("GBN_SEED_CKEYS", 1) != 2;
    const int v_bits = 8 + K.qh_bits;
    const bool packed = composite && ck_pack && ck_bits + v_bits <= 64;
    if (composite) {
        K.key_scan = KS.key_a; K.idx = KS.idx_a; K.v_bits = packed ? v_bits : 0;
        if (from_segments) { K.seg = E.slice_seg; K.seg_count = E.seg_counts; K.nseg = E.seg_n; K.seg_cap = E.seg_len; K.seg_first = E.seg_firsts; }
        // Seeds that come in scan order, subject by subject (scan_fold_ordered_kernel's segments): a stable partition of every
        // subject's seeds by slot is all that is left, and seed_order.hip does it as a counting sort that builds the keys
        // on its way -- no key kernel, no radix passes (GBN_SEED_ORDER=0: keys + the library sort, as rounds 2-3)
        const int nsubj = s1 - s0;
        const bool order = from_segments && E.seg_ordered && packed && nsubj <= GBN_ORDER_MAX_SUBJ && (1 << K.group_bits) <= GBN_ORDER_MAX_SLOTS &&
                           n >= (int64_t)nsubj * 64 && n < ((int64_t)1 << 31) && gbn::switch_value("GBN_SEED_ORDER", 1) != 0 &&
                           seed_order_scratch_words(n, nsubj, K.group_bits) * sizeof(uint32_t) <= KS.sort_tmp_bytes;
        if (phase != 2 && order) {
            K.key_scan = KS.key_b;
            KS.kt.mark(GBN_KT_SORT, st);
            HIPCHK(launch_seed_order(K, nsubj, static_cast<uint32_t *>(KS.sort_tmp), st));
            KS.kt.mark(-1, st);
        } else if (phase != 2) {
            KS.kt.mark(GBN_KT_KEYS, st);
            HIPCHK(launch_seed_ckeys(K, st));
            size_t tb = KS.sort_tmp_bytes;
            KS.kt.mark(GBN_KT_SORT, st);
            // (seeds that come in scan order are in the order of the key's scan-position bits already: the stable sort has
            // subject | slot left to do)
            const int s_done = (from_segments && E.seg_ordered) ? K.s_bits : 0;
            if (packed) HIPCHK(sort_keys_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, n, v_bits + s_done, v_bits + ck_bits, st));
            else HIPCHK(sort_pairs_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, KS.idx_a, KS.idx_b, n, ck_bits, st));
            KS.kt.mark(-1, st);
        }
        // key_b = sorted composite keys, idx_b = ext_left of the seeds in that order (packed: both in key_b)
    } else if (!small_sort) {
        K.idx = KS.idx_b; K.key_group = KS.key_a;
        if (phase != 2) {
            KS.kt.mark(GBN_KT_KEYS, st);
            HIPCHK(launch_group_keys(K, st));
            size_t tb = KS.sort_tmp_bytes;
            KS.kt.mark(GBN_KT_SORT, st);
            HIPCHK(sort_pairs_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, KS.idx_b, KS.idx_a, n, group_key_bits, st));
            KS.kt.mark(-1, st);
        }
        // key_b = sorted (subject, slot) keys, idx_a = seed indices grouped by run, scan order inside
    }

    if (phase == 1) { *nih_out = composite ? 1 : 0; return GBN_OK; }      // (tells the caller whether the second half can run without `seeds`)
    if ((rc = grow_ihit_buffers(slot, std::max<size_t>(E.ihit_cap_s[slot], 1 << 16)))) return rc;
    unsigned long long nih = 0;
    *nih_out = 0;
    for (;;) {
        HIPCHK(hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned long long), st));      // initial hits, runs
        GbnExtParams X; std::memset(&X, 0, sizeof(X));
        X.db = db.d_packed; X.byte_off = db.d_byte_off; X.len = db.d_len;
        X.seeds = seeds; X.idx = KS.idx_a; X.key_group = KS.key_b; X.n = n;
        X.q8 = d->q8; X.qlen = b.qlen; X.q2 = d->q2; X.qinv = d->qinv; X.q4 = d->q4_base; X.q4_plane = d->q4_plane; X.q4_origin = b.qpad;
        X.ctx_off = d->ctx_off; X.ctx_len = d->ctx_len; X.ctx_xdrop = d->ctx_xdrop;
        X.ctx_cutoff = d->ctx_cutoff; X.ctx_reduced = d->ctx_reduced; X.nctx = (int32_t)b.ctx.size();
        X.matrix = d->matrix; X.score_table = d->score_table;
        X.word = b.lut.word; X.container_hash = b.container;
        X.cell_diag = KS.cell_diag; X.cell_level = KS.cell_level;
        X.cell_start = d->cell_start; X.ent = d->ent; X.cell_mask = (uint32_t)(b.lut.ncells - 1); X.lut = b.lut.lut;
        X.masked = b.lut.masked ? 1 : 0;
        X.run_heads = KS.idx_b; X.run_count = reinterpret_cast<uint32_t *>(ctr + 1); X.group_bits = K.group_bits;
        X.ctx_hint = d->ctx_hint; X.ctx_hint_shift = kCtxHintShift; X.ext_rec = KS.ext_rec; X.ctx_blk = d->ctx_blk; X.ctx_pack = d->ctx_pack;
        if (composite) {
            X.idx = KS.idx_b; X.run_heads = KS.idx_a;
            // (values packed under the keys: idx_b is free, and lists the seeds of the exact pass; GBN_EXT_SPLIT=0: inline as before)
            const bool split = gbn::switch_value("GBN_EXT_SPLIT", 1) != 0;
            if (split && packed) { X.exact_list = KS.idx_b; X.exact_count = reinterpret_cast<uint32_t *>(ctr + 1) + 1; }
            X.ck_shift = K.s_bits; X.ck_s_bits = K.s_bits; X.ck_qh_bits = K.qh_bits; X.ck_q_bits = K.q_bits; X.ck_q_desc = K.q_descending; X.ck_subj_base = K.subj_base; X.ck_vbits = K.v_bits;
        }
        X.ihits = E.ihits_s[slot]; X.ihit_count = ctr; X.ihit_cap = E.ihit_cap_s[slot];
        HIPCHK(launch_diag_ungapped(X, st, &KS.kt));
        HIPCHK(hipMemcpyAsync(&nih, ctr, sizeof(nih), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        { double km[GBN_KT_N] = {0}; KS.kt.collect(km); if (diag) GBN_DIAG_LOCKED(for (int i = 0; i < GBN_KT_N; i++) diag->kernel_ms[i] += km[i]); }
        if (nih <= E.ihit_cap_s[slot]) break;
        if ((rc = grow_ihit_buffers(slot, (size_t)nih + (nih >> 3)))) return rc;
    }
    if (diag) GBN_DIAG_LOCKED(diag->init_extends += (int64_t)nih; diag->good_init_extends += (int64_t)nih; diag->seed_stage_ms += ms_since(t_stage));
    *nih_out = nih;
    return GBN_OK;

}

// one range of subjects [s0, s1): scan, seed order, diagonal filter + ungapped extension on the engine's
// stream; then the gapped stage -- inline, or (overlap != 0) on stream2 + a host thread while the caller
// goes on to the next range / batch.  At most one gapped stage is in flight.
static int search_range(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res,
                        GbnDiagnostics *diag, int keep_stages, int overlap = 0)
{
    const int slot = E.slot;
    unsigned long long cnt[3] = {0, 0, 0};
    int64_t bases = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_stage = now();
    trace_mark("range: scan starts");
    int rc = run_scan(b, db, s0, s1, diag, cnt, &bases);
    trace_mark("scan done");
    if (rc == kSkewedRange) {
        // lookup words of this range pile up in a few bins: halve it (by packed size) until the repeat-rich
        // subjects sit in small ranges of their own, which the direct-probe kernel scans
        int64_t half = 0, acc = 0;
        for (int32_t s = s0; s < s1; s++) half += db.len[s];
        half /= 2;
        int32_t mid = s0;
        while (mid < s1 - 1 && acc + db.len[mid] <= half) acc += db.len[mid++];
        if (mid == s0) mid = s0 + 1;
        if ((rc = search_range(b, db, s0, mid, res, diag, keep_stages, overlap))) return rc;
        return search_range(b, db, mid, s1, res, diag, keep_stages, overlap);
    }
    if (rc) return rc;
    if (diag) diag->scan_stage_ms += ms_since(t_stage);
    t_stage = now();
    if (diag) { diag->lookup_hits += (int64_t)cnt[1]; diag->seeds += (int64_t)cnt[0]; diag->subject_bases_scanned += bases; }
    const int64_t n = (int64_t)cnt[0];
    if (n == 0) return GBN_OK;
    if (n > INT32_MAX) { set_error("too many seeds in one range"); return GBN_ERR_NOMEM; }

    // Few seeds (megablast shapes): the whole rest of the range -- seed order, diagonal filter, ungapped and
    // gapped extension, host replay -- runs on stream2 + a host thread on a copy of the seeds, and the
    // caller's next scan follows this one without a gap.  Many seeds (blastn shapes): the seed stage stays
    // on the engine's stream (it is as long as the scan) and only the gapped stage is asynchronous.
    const bool async_seed = overlap && !keep_stages && n < ((int64_t)1 << 20);
    if (async_seed) {
        if ((rc = wait_pending_gpu())) return rc;           // one asynchronous stage in flight at most
        if ((size_t)n > E.seeds_async_cap) {
            dev_free(E.seeds_async); E.seeds_async_cap = 0;
            if ((rc = dev_alloc(E.seeds_async, std::max<size_t>((size_t)n + (size_t)n / 4, 1 << 16)))) return rc;
            E.seeds_async_cap = std::max<size_t>((size_t)n + (size_t)n / 4, 1 << 16);
        }
        // (a binning kernel queued ahead sits on the engine's stream: the copy goes to the stage's own stream -- the host has
        // seen the scan finish -- and the next scan's kernels wait for it before they write seeds again: run_scan_impl.  Seeds
        // that had to be put back to back first (a slice scan's segments) are copied behind that kernel, on its stream.)
        const bool compacting = E.seg_valid;
        if ((rc = compact_seeds(E.stream))) return rc;
        hipStream_t copy_st = (E.ahead.valid && !compacting) ? E.stream2 : E.stream;
        HIPCHK(hipMemcpyAsync(E.seeds_async, E.seeds, (size_t)n * sizeof(GbnDevSeed), hipMemcpyDeviceToDevice, copy_st));
        HIPCHK(hipEventRecord(E.ev_seed, copy_st));
        E.seed_copy_pending = copy_st == E.stream2;
        E.slot ^= 1;
        E.pending_err.clear();
        const int dev = E.device, ksi = 0;                  // (no stage is in flight: either key set)
        GbnBatch *bp = &b; GbnDb *dbp = &db; GbnResults *rp = &res;
        Engine *eng = tl_eng;
        E.pending = std::async(std::launch::async, [=]() -> int {
            tl_eng = eng;
            int r = GBN_OK;
            unsigned long long nih2 = 0;
            if (hipSetDevice(dev) != hipSuccess) { E.pending_err = "hipSetDevice failed in the extension thread"; return GBN_ERR_HIP; }
            if (hipStreamWaitEvent(E.stream2, E.ev_seed, 0) != hipSuccess) { E.pending_err = "hipStreamWaitEvent failed"; return GBN_ERR_HIP; }
            r = seed_stage(*bp, *dbp, *rp, diag, 0, slot, E.seeds_async, n, E.counters + 4, E.stream2, &nih2, s0, s1, ksi);
            if (!r && nih2) r = gapped_stage(*bp, *dbp, s0, s1, *rp, diag, 0, slot, nih2, E.stream2, true);
            if (r) E.pending_err = gbn_last_error();      // the error text is per thread
            return r;
        });
        E.has_pending = true; E.pending_res = rp; E.pending_ks = ksi; E.pending_batch = bp;
        return GBN_OK;
    }
    // the key set no stage in flight is working on
    const int ksi = (E.has_pending && E.pending_ks == 0) ? 1 : 0;
    // (Tried in round 3 and not kept: the stage in two halves -- keys + sort on the engine's stream, extension + replay with
    // the gapped stage on the second one, so that the extension of range k runs next to the scan of range k + 1, the only
    // kernels that fit beside the slice scan's 152 KB of LDS.  The second stream then carries 7.4 ms per range
    // (extension + replay 3.8 next to the scan, gapped stage 3.6) against 7.6 ms for the whole range before: 38.4 - 39.2 vs
    // 38.8 - 39.0 ms per pass.  A third stream would be needed, and two processes on one GPU gain 8 %: the chip is busy.)
    unsigned long long nih = 0;
    E.counters_zeroed = false;                              // (the stage counts in counters[2], [3])
    if ((rc = seed_stage(b, db, res, diag, keep_stages, slot, E.seeds, n, E.counters + 2, E.stream, &nih, s0, s1, ksi))) return rc;
    trace_mark("seed stage done (inline)");
    if (nih == 0) return GBN_OK;
    if ((rc = wait_pending_gpu())) return rc;               // one gapped stage in flight at most
    trace_mark("previous asynchronous stage finished");
    if (!overlap || keep_stages) { wait_host(); return gapped_stage(b, db, s0, s1, res, diag, keep_stages, slot, nih, E.stream); }
    E.slot ^= 1;
    E.pending_err.clear();
    const int dev = E.device;
    GbnBatch *bp = &b; GbnDb *dbp = &db; GbnResults *rp = &res;
    Engine *eng = tl_eng;
    E.pending = std::async(std::launch::async, [=]() -> int {
        tl_eng = eng;
        if (hipSetDevice(dev) != hipSuccess) { E.pending_err = "hipSetDevice failed in the gapped-stage thread"; return GBN_ERR_HIP; }
        const int r = gapped_stage(*bp, *dbp, s0, s1, *rp, diag, 0, slot, nih, E.stream2, true);
        if (r) E.pending_err = gbn_last_error();      // the error text is per thread
        return r;
    });
    E.has_pending = true; E.pending_res = rp; E.pending_batch = bp;
    return GBN_OK;
}

// gapped extension of every initial hit of a range (slot buffers), D2H, host replay of the acceptance
// rules per subject.  Touches only the slot's buffers, the results and the gapped fields of `diag`.
static int gapped_host(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag, int keep_stages,
                       const GbnDevInitHit *hih, const GbnDevGapped *hg, size_t nih);

// Host copies of a range's initial hits and gapped extensions: pinned buffers that are handed out again.  (Vectors
// allocated and freed per range cost more than their pages: freeing memory the copy engine has just written to, while
// the next range's kernels and copies run, stalls the device's queues -- the lane DP took 8.9 instead of 3.4 ms.)
static int hitbuf_get(size_t n, HitBuf &out) {
    {
        std::lock_guard<std::mutex> lk(E.hitbuf_mu);
        for (size_t i = 0; i < E.hitbuf_idle.size(); i++)
            if (E.hitbuf_idle[i].cap >= n) { out = E.hitbuf_idle[i]; E.hitbuf_idle.erase(E.hitbuf_idle.begin() + (long)i); return GBN_OK; }
        if (!E.hitbuf_idle.empty()) {           // too short: let one go, its successor is longer
            HitBuf old = E.hitbuf_idle.back(); E.hitbuf_idle.pop_back();
            (void)hipHostFree(old.hih); (void)hipHostFree(old.hg);
        }
    }
    HitBuf b; b.cap = std::max<size_t>(n + n / 4, 1 << 16);
    if (hipHostMalloc((void **)&b.hih, b.cap * sizeof(GbnDevInitHit)) != hipSuccess ||
        hipHostMalloc((void **)&b.hg, b.cap * sizeof(GbnDevGapped)) != hipSuccess) {
        if (b.hih) (void)hipHostFree(b.hih);
        set_error("out of pinned host memory (gapped stage)"); return GBN_ERR_NOMEM;
    }
    out = b;
    return GBN_OK;
}
static void hitbuf_put(const HitBuf &b) { if (b.hih) { std::lock_guard<std::mutex> lk(E.hitbuf_mu); E.hitbuf_idle.push_back(b); } }
static void hitbuf_drain() {
    std::lock_guard<std::mutex> lk(E.hitbuf_mu);
    for (HitBuf &b : E.hitbuf_idle) { (void)hipHostFree(b.hih); (void)hipHostFree(b.hg); }
    E.hitbuf_idle.clear();
}

static int gapped_stage(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag,
                        int keep_stages, int slot, unsigned long long nih, hipStream_t st, bool detach_host)
{
    const DeviceBatch *d = b.dev;
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_stage = now();
    int32_t max_len = 0, max_ctx = 0;
    for (int32_t s = s0; s < s1; s++) max_len = std::max(max_len, db.len[s]);
    for (auto &c : b.ctx) max_ctx = std::max(max_ctx, c.query_length);
    GbnGapParams G; std::memset(&G, 0, sizeof(G));
    G.db = db.d_packed; G.byte_off = db.d_byte_off; G.len = db.d_len;
    G.ihits = E.ihits_s[slot]; G.q8 = d->q8; G.q2 = d->q2; G.qinv = d->qinv; G.ctx_off = d->ctx_off; G.ctx_len = d->ctx_len; G.nctx = (int32_t)b.ctx.size();
    G.matrix = d->matrix; G.reward = b.opt.reward; G.penalty = b.opt.penalty;
    G.gap_open = b.opt.gap_open; G.gap_extend = b.opt.gap_extend; G.xdrop = b.gap_x_dropoff;
    G.out = E.gapped_s[slot];
    int32_t row_len = 0;
    const size_t per_thread = (size_t)gap_scratch_ints(b, max_len, max_ctx, &row_len);
    G.row_len = row_len;
    G.scratch_per_thread = (int32_t)per_thread;
    // grid: at most 24 waves per CU (measured on the blastn shape: 12-20 make the gapped stage the longer one, 28+ starve the scan; the scan kernels of the next range need room, see greedy_kernel) and at
    // most 4 GiB of scratch; the threads stride over the initial hits
    const size_t budget_ints = (size_t)1 << 30;
    const int waves_per_cu = (int)std::max<long long>(1, gbn::switch_value("GBN_GAP_WAVES", 24));
    const size_t by_budget = std::max<size_t>(1, budget_ints / per_thread / 64);
    const size_t blocks = std::max<size_t>(1, std::min({((size_t)nih + 63) / 64, (size_t)E.num_cu * (size_t)waves_per_cu, by_budget}));
    const size_t scratch_ints = blocks * 64 * per_thread;
    if (scratch_ints > E.gap_scratch_ints_s[slot]) {
        dev_free(E.gap_scratch_s[slot]);
        if ((rc = dev_alloc(E.gap_scratch_s[slot], scratch_ints))) { E.gap_scratch_ints_s[slot] = 0; return rc; }
        E.gap_scratch_ints_s[slot] = scratch_ints;
    }
    G.scratch = E.gap_scratch_s[slot];
    G.first = 0; G.n = (int64_t)nih; G.max_blocks = (int32_t)blocks;
    if (gbn::switch_is_set("GBN_DP_STATS")) HIPCHK(hipMemsetAsync(G.scratch, 0, 256, st));
    HIPCHK(launch_gapped(G, b.opt.greedy != 0, st, &E.kt_gap[slot]));
    if (gbn::switch_is_set("GBN_DP_STATS")) {       // (-DGBN_DP_STATS=1 builds only)
        unsigned long long c[24]; HIPCHK(hipMemcpyAsync(c, G.scratch, sizeof(c), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[gbn dbg] wave DP: %llu extensions (%llu left to the scratch kernel), %llu rows, %llu rounds, mean window %.1f\n", c[2], c[3], c[0], c[1], c[0] ? (double)c[4] / c[0] : 0.0);
        fprintf(stderr, "[gbn dbg]   rows by window / 8:"); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", c[8 + k]);
        fprintf(stderr, "\n[gbn dbg]   extensions by widest window / 8:"); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", c[16 + k]);
        fprintf(stderr, "\n");
    }
    HitBuf hb;
    if ((rc = hitbuf_get((size_t)nih, hb))) return rc;
    struct PutBack { HitBuf b; bool armed = true; ~PutBack() { if (armed) hitbuf_put(b); } } putback{hb};
    GbnDevInitHit *hih = hb.hih; GbnDevGapped *hg = hb.hg;
    HIPCHK(hipMemcpyAsync(hih, E.ihits_s[slot], (size_t)nih * sizeof(GbnDevInitHit), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hg, E.gapped_s[slot], (size_t)nih * sizeof(GbnDevGapped), hipMemcpyDeviceToHost, st));
    trace_mark("gapped: kernels + copies queued");
    HIPCHK(hipStreamSynchronize(st));
    trace_mark("gapped: kernels + copies done");
    { double km[GBN_KT_N] = {0}; E.kt_gap[slot].collect(km); if (diag) GBN_DIAG_LOCKED(for (int i = 0; i < GBN_KT_N; i++) diag->kernel_ms[i] += km[i]; diag->gapped_stage_ms += ms_since(t_stage)); }
    const bool detach_on = gbn::switch_value("GBN_HOST_DETACH", 1) != 0;
    // (a few thousand extensions -- megablast shapes -- are replayed in less time than handing them over takes)
    if (!detach_host || !detach_on || nih < 20000) { if (detach_host) wait_host(); return gapped_host(b, db, s0, s1, res, diag, keep_stages, hih, hg, (size_t)nih); }
    // the replay of this range's extensions joins the queue of host replays (in range order: the lists are appended to
    // the results); this thread, the slot's device buffers and the second stream are free for the next range
    {
        putback.armed = false;                              // the buffers go back when the replay is done
        const size_t n_hits = (size_t)nih;
        GbnBatch *bp = &b; GbnDb *dbp = &db; GbnResults *rp = &res;
        Engine *eng = tl_eng;
        std::lock_guard<std::mutex> lk(E.host_mu);
        std::shared_future<void> prev = E.host_tail;
        E.host_tail = std::async(std::launch::async, [=]() mutable {
            enter(eng);
            if (prev.valid()) prev.wait();
            try {
                const int r = gapped_host(*bp, *dbp, s0, s1, *rp, diag, 0, hb.hih, hb.hg, n_hits);
                if (r) record_failure(rp, r, gbn_last_error());
            } catch (const std::exception &e) {             // (nobody calls get() on this future: the failure is reported through the results)
                record_failure(rp, GBN_ERR_NOMEM, std::string("host replay of a range failed: ") + e.what());
            }
            hitbuf_put(hb);
            // (the task's state lives as long as its successor refers to it: let go of the predecessor, or every
            // replay ever queued stays reachable from the newest one)
            prev = std::shared_future<void>();
        }).share();
        bp->host_tail = E.host_tail; rp->host_tail = E.host_tail;
    }
    return GBN_OK;
}

// the acceptance rules of BLAST_GetGappedScore replayed per subject over the extensions of a range, the HSP lists
// appended to the results (ascending oid)
static int gapped_host(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag, int keep_stages,
                       const GbnDevInitHit *hih, const GbnDevGapped *hg, size_t nih)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_stage = now();

    // ---- host replay per subject, ascending oid ----
    // group the hits by subject (counting sort; the order inside a subject does not matter,
    // finish_subject sorts with the seed sequence number as the last key)
    std::vector<uint32_t> order((size_t)nih);
    {
        std::vector<uint32_t> start((size_t)(s1 - s0) + 1, 0);
        for (size_t i = 0; i < (size_t)nih; i++) start[(size_t)(hih[i].subj - s0) + 1]++;
        for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
        for (size_t i = 0; i < (size_t)nih; i++) order[start[(size_t)(hih[i].subj - s0)]++] = (uint32_t)i;
    }
    // subjects are independent: split the ordered hits into per-subject spans, replay the spans on a
    // few host threads when there is enough work, append the HSP lists in ascending oid order
    std::vector<std::pair<size_t, size_t>> spans;
    for (size_t i = 0; i < order.size();) {
        size_t j = i; const int32_t subj = hih[order[i]].subj;
        while (j < order.size() && hih[order[j]].subj == subj) j++;
        spans.emplace_back(i, j); i = j;
    }
    for (size_t k = 0; k < (size_t)nih; k++)
        if (hg[k].score == INT32_MIN) { set_error("gapped DP scratch overflow"); return GBN_ERR_NOMEM; }
    std::vector<std::vector<GbnHSP>> outs(spans.size());
    std::vector<std::vector<GbnInitHit>> ihs(keep_stages ? spans.size() : 0);
    auto replay = [&](size_t k, GbnDiagnostics *dg) {
        const size_t i0 = spans[k].first, i1 = spans[k].second; const int32_t subj = hih[order[i0]].subj;
        std::vector<std::pair<GbnDevInitHit, GbnDevGapped>> hits; hits.reserve(i1 - i0);
        for (size_t j = i0; j < i1; j++) hits.emplace_back(hih[order[j]], hg[order[j]]);
        if (keep_stages) {
            auto sorted = hits;
            // reference order of the initial hit list
            std::sort(sorted.begin(), sorted.end(), [](const auto &x, const auto &y) {
                const GbnDevInitHit &a = x.first, &c = y.first;
                if (a.score != c.score) return a.score > c.score;
                if (a.s_start != c.s_start) return a.s_start < c.s_start;
                if (a.length != c.length) return a.length > c.length;
                if (a.q_start != c.q_start) return a.q_start < c.q_start;
                return a.seq < c.seq;
            });
            for (auto &pr : sorted) {
                GbnInitHit o; o.oid = db.oid_of(subj); o.q_off = pr.first.q_off; o.s_off = pr.first.s_off;
                o.q_start = pr.first.q_start; o.s_start = pr.first.s_start; o.length = pr.first.length;
                o.score = pr.first.score; o.pad_ = 0;
                ihs[k].push_back(o);
            }
        }
        finish_subject(b, db.oid_of(subj), db.len[subj], hits, outs[k], dg, /* chunk = */ !db.real_of.empty());
        if (!db.real_of.empty()) {      // a chunk's list: sequence coordinates (Blast_HSPListAdjustOffsets), marked for the merge at the end of the search
            const int32_t ord = db.chunk_of(subj), off = (int32_t)((int64_t)ord * (db.chunk_len - kDbseqChunkOverlap));
            for (GbnHSP &h : outs[k]) { h.s_offset += off; h.s_end += off; h.s_gapped_start += off; h.pad_ = ord + 1; }
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nthreads = (nih < 20000 || spans.size() < 2) ? 1u : std::min({hw, 16u, (unsigned)spans.size()});
    std::vector<GbnDiagnostics> dloc(nthreads);
    for (auto &dl : dloc) std::memset(&dl, 0, sizeof(dl));
    if (nthreads == 1) {
        for (size_t k = 0; k < spans.size(); k++) replay(k, &dloc[0]);
    } else {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++)
            pool.emplace_back([&, t] { for (size_t k; (k = next.fetch_add(1)) < spans.size();) replay(k, &dloc[t]); });
        for (auto &th : pool) th.join();
    }
    for (size_t k = 0; k < spans.size(); k++) {
        res.hsps.insert(res.hsps.end(), outs[k].begin(), outs[k].end());
        if (keep_stages) res.init_hits.insert(res.init_hits.end(), ihs[k].begin(), ihs[k].end());
    }
    if (diag) for (auto &dl : dloc) {
        GBN_DIAG_LOCKED(diag->gapped_extensions += dl.gapped_extensions; diag->good_extensions += dl.good_extensions; diag->seqs_passed += dl.seqs_passed);
    }
    if (diag) GBN_DIAG_LOCKED(diag->host_stage_ms += ms_since(t_stage));
    trace_mark("gapped: host replay done");
    return GBN_OK;
}

// Stretches [src_off, src_off + nbytes) of the shard's packed bytes, back to back in `out` (traceback stage): one
// gather kernel and one copy on a stream of its own, next to whatever the search streams are doing.
int gather_shard_bytes(const GbnDb &db, const std::vector<int64_t> &src_off, const std::vector<int32_t> &nbytes, std::vector<uint8_t> &out)
{
    int rc = GBN_OK;
    if (!db.engine) { set_error("gather_shard_bytes: shard without a device"); return GBN_ERR_ARG; }
    enter(static_cast<Engine *>(db.engine));
    std::mutex &mu = E.gather_mu; hipStream_t &st = E.gather_stream;
    const int32_t n = (int32_t)src_off.size();
    std::vector<int64_t> dst_off((size_t)n); int64_t total = 0;
    for (int32_t i = 0; i < n; i++) { dst_off[(size_t)i] = total; total += nbytes[(size_t)i]; }
    out.resize((size_t)total);
    if (n == 0) return GBN_OK;
    std::lock_guard<std::mutex> lk(mu);
    if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int64_t *d_so = nullptr, *d_do = nullptr; int32_t *d_nb = nullptr; uint8_t *d_out = nullptr;
    auto cleanup = [&]() { dev_free(d_so); dev_free(d_do); dev_free(d_nb); dev_free(d_out); };
    if ((rc = dev_alloc(d_so, (size_t)n)) || (rc = dev_alloc(d_do, (size_t)n)) || (rc = dev_alloc(d_nb, (size_t)n)) || (rc = dev_alloc(d_out, (size_t)total))) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(d_so, src_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_do, dst_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_nb, nbytes.data(), (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_gather_bytes(db.d_packed, d_so, d_do, d_nb, n, d_out, st);
    // (through a pinned buffer that stays: a copy straight into the caller's fresh vector makes the runtime register those
    // pages, and their release next to running kernels stalls the device's queues -- see HitBuf)
    uint8_t *&stage = E.gather_stage; size_t &stage_cap = E.gather_stage_cap;
    if (e == hipSuccess && (size_t)total > stage_cap) {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_cap = 0;
        const size_t want = (size_t)total + (size_t)total / 4 + (1 << 20);
        e = hipHostMalloc((void **)&stage, want);
        if (e == hipSuccess) stage_cap = want;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(stage, d_out, (size_t)total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) std::memcpy(out.data(), stage, (size_t)total);
    cleanup();
    if (e != hipSuccess) { set_error(std::string("gather_shard_bytes: ") + hipGetErrorString(e)); return GBN_ERR_HIP; }
    return GBN_OK;
}

}  // namespace gbn

using namespace gbn;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *gbn_last_error(void) { return g_err.c_str(); }

}  // extern "C"
namespace gbn {
// the engine of device `dev` (< 0: the calling thread's current HIP device), created and initialised on first use
static int engine_init(int dev, Engine **out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return GBN_ERR_NO_DEVICE; }
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= n || dev >= kMaxDevices) { set_error("gpu_id out of range"); return GBN_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g_eng_mu);
    if (g_eng[dev] && g_eng[dev]->ready) { *out = g_eng[dev]; return GBN_OK; }
    if (!g_eng[dev]) g_eng[dev] = new Engine();         // (after gbn_release the object is still there: armed again)
    Engine &N = *g_eng[dev];
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    N.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // the scan stream outranks the extension and table-builder streams: its kernels need whole CUs
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = least, hi = greatest priority
        const bool prio = gbn::switch_value("GBN_STREAM_PRIORITY", 1) != 0;
        HIPCHK(hipStreamCreateWithPriority(&N.stream, hipStreamNonBlocking, prio ? hi : 0));
        HIPCHK(hipStreamCreateWithPriority(&N.stream2, hipStreamNonBlocking, prio ? lo : 0));
        HIPCHK(hipStreamCreateWithPriority(&N.stream_build, hipStreamNonBlocking, prio ? lo : 0));
    }
    HIPCHK(hipEventCreate(&N.ev0)); HIPCHK(hipEventCreate(&N.ev1));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&N.evk[i]));
    HIPCHK(hipEventCreateWithFlags(&N.ev_seed, hipEventDisableTiming));
    HIPCHK(pool_alloc((void **)&N.counters, 8 * sizeof(unsigned long long)));
    N.device = dev; N.ready = true;
    if (g_default_dev < 0) g_default_dev = dev;
    *out = g_eng[dev];
    return GBN_OK;
}
}  // namespace gbn
extern "C" {

int gbn_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0; }

int gbn_init(int use_gpu, int gpu_id) {
    return gbn::guard(__func__, [&]() -> int {
    if (!use_gpu) { set_error("this engine has no CPU path: use_gpu must be true"); return GBN_ERR_NO_DEVICE; }
    Engine *e = nullptr;
    const int rc = engine_init(gpu_id, &e);
    if (rc) return rc;
    tl_sel = e->device;
    enter(e);
    return GBN_OK;
    });
}
// the device the calling thread's later gbn_db_new / gbn_batch_new* / gbn_blastdb_load_shard calls work on (the GPU
// lease of GB/gpu_blast_multi_gpu_utils.cpp:105-139: ThreadFetchGPU does cudaSetDevice for the search thread)
int gbn_use_device(int gpu_id) {
    return gbn::guard(__func__, [&]() -> int {
    if (gpu_id < 0) { set_error("gbn_use_device: a device number"); return GBN_ERR_ARG; }
    return gbn_init(1, gpu_id);
    });
}
int gbn_current_device(void) {
    return gbn::guard(__func__, [&]() -> int {
    if (tl_sel >= 0) return tl_sel;
    std::lock_guard<std::mutex> lk(g_eng_mu);
    return g_default_dev;
    });
}
int gbn_db_device(const GbnDb *db) { return db && db->engine ? static_cast<const Engine *>(db->engine)->device : -1; }

// Shards a caller keeps per database handle (the shim: per BlastSeqSrc).  The reference caches every subject it
// has uploaded for the life of the process and gpu_ReleaseDBMemory (here: gbn_release_db_memory) drops that cache
// (GB/gpu_blastn_MB_and_smallNa.cu:1462-1468, gpu_blastn_na_ungapped_v3.cpp:27-60); here the cache holds whole
// shards, keyed by the caller's handle, and gbn_release_db_memory frees them.  Shards the caller made with
// gbn_db_from_* and did not insert stay the caller's.
static std::mutex g_cache_mu;
static std::map<const void *, GbnDb *> g_db_cache;
GbnDb *gbn_db_cache_find(const void *key) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_db_cache.find(key);
    return it == g_db_cache.end() ? nullptr : it->second;
}
int gbn_db_cache_insert(const void *key, GbnDb *db) {
    return gbn::guard(__func__, [&]() -> int {
    if (!db) return GBN_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_db_cache.find(key);
    if (it != g_db_cache.end()) { set_error("gbn_db_cache_insert: key already holds a shard"); return GBN_ERR_ARG; }
    g_db_cache[key] = db;
    return GBN_OK;
    });
}
// The block cache: the database on a device as resident shards of one OID chunk each, keyed by what the chunk IS --
// (device, database name, the OIDs) -- not by who asked: the reference caches every subject per OID
// (GB/gpu_blastn_MB_and_smallNa.cu:1461-1467), so whichever of its N search threads gets whichever chunk of whichever
// query batch (API/prelim_search_runner.hpp:135-166), nothing is uploaded twice.  The key holds the OIDs themselves:
// no hash that could collide.  An insert that finds the block already there (two threads built it at the same time)
// frees the newcomer and hands back the one that stays.
typedef std::tuple<int, std::string, std::vector<int32_t>> BlockKey;
static std::map<BlockKey, GbnDb *> g_block_cache;
static std::atomic<long long> g_db_bytes_uploaded{0};       // slab bytes copied to a device by gbn_db_new / the shard builder
int gbn_block_cache_find(const char *db_name, const int32_t *oids, int32_t n, GbnDb **out) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || n <= 0 || !oids) return GBN_ERR_ARG;
    const int device = gbn_current_device();
    BlockKey key(device, std::string(db_name ? db_name : ""), std::vector<int32_t>(oids, oids + n));
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_block_cache.find(key);
    *out = it == g_block_cache.end() ? nullptr : it->second;
    return GBN_OK;
    });
}
int gbn_block_cache_insert(const char *db_name, const int32_t *oids, int32_t n, GbnDb *db, GbnDb **kept) {
    return gbn::guard(__func__, [&]() -> int {
    if (!db || !kept || n <= 0 || !oids) return GBN_ERR_ARG;
    BlockKey key(gbn_db_device(db), std::string(db_name ? db_name : ""), std::vector<int32_t>(oids, oids + n));
    GbnDb *loser = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_block_cache.find(key);
        if (it == g_block_cache.end()) { g_block_cache.emplace(std::move(key), db); *kept = db; }
        else { *kept = it->second; if (it->second != db) loser = db; }
    }
    if (loser) gbn_db_free(loser);
    return GBN_OK;
    });
}
long long gbn_debug_db_bytes_uploaded(void) { return g_db_bytes_uploaded.load(); }
long long gbn_debug_bin_ahead_hits(void) { return tl_eng ? E.ahead_hits : 0; }
long long gbn_debug_bin_ahead_misses(void) { return tl_eng ? E.ahead_misses : 0; }

// ---- the record cache of the calling thread's device (Engine::rec_sets) ----
int gbn_record_cache_set_limit(long long bytes) {
    return gbn::guard(__func__, [&]() -> int {
    const int rc = enter_current();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E.mu);
    E.rec_limit = bytes < 0 ? -1 : bytes;
    if (E.ahead.valid) { (void)hipStreamSynchronize(E.stream); E.ahead.valid = false; E.ahead_misses++; }     // (a binning kernel queued ahead: done before buffers change hands)
    E.last_key_valid = false;
    const long long limit = rec_limit_bytes();
    if (limit == 0) rec_purge(nullptr, true);
    else rec_make_room(0, limit, nullptr);
    return GBN_OK;
    });
}
// every set forgets its records and keeps its buffers: the next pass of each key bins again (bench: a cold start without
// giving gigabytes back to the driver and asking for them again)
int gbn_record_cache_invalidate(void) {
    return gbn::guard(__func__, [&]() -> int {
    const int rc = enter_current();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E.mu);
    for (RecordSet *r : E.rec_sets) { if (r->queued) (void)hipStreamSynchronize(E.stream); r->complete = false; r->queued = false; }
    E.scratch.complete = false;
    if (E.ahead.valid) { (void)hipStreamSynchronize(E.stream); E.ahead.valid = false; }
    E.last_key_valid = false;
    return GBN_OK;
    });
}
int gbn_record_cache_stats(long long *out, int n) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || n < 0) return GBN_ERR_ARG;
    const int rc = enter_current();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E.mu);
    const long long v[9] = {rec_limit_bytes(), (long long)rec_held_bytes(), (long long)E.rec_sets.size(), E.rec_hits, E.rec_misses, E.rec_evictions, E.rec_bypass, E.ahead_hits, E.rec_prepared};
    for (int i = 0; i < n && i < 9; i++) out[i] = v[i];
    return GBN_OK;
    });
}

// ---- views: several resident blocks searched as one shard ----
static std::map<std::vector<const GbnDb *>, GbnDb *> g_view_cache;      // (under g_cache_mu) keyed by the blocks, ascending first OID
static void free_view(GbnDb *v);
int gbn_block_view(GbnDb *const *blocks, int32_t n, GbnDb **out) {
    return gbn::guard(__func__, [&]() -> int {
    if (!blocks || n <= 0 || !out) { set_error("gbn_block_view: bad argument"); return GBN_ERR_ARG; }
    *out = nullptr;
    std::vector<const GbnDb *> parts(blocks, blocks + n);
    for (const GbnDb *p : parts) {
        if (!p || !p->engine || p->engine != parts[0]->engine) { set_error("gbn_block_view: the blocks live on different devices"); return GBN_ERR_ARG; }
        if (!p->real_of.empty() || !p->view_parts.empty()) { set_error("gbn_block_view: a block with chunked sequences, or a view"); return GBN_ERR_UNSUPPORTED; }
    }
    std::stable_sort(parts.begin(), parts.end(), [](const GbnDb *a, const GbnDb *b) {
        return (a->num_seqs ? a->oid_of(0) : a->first_oid) < (b->num_seqs ? b->oid_of(0) : b->first_oid); });
    for (size_t i = 1; i < parts.size(); i++) if (parts[i] == parts[i - 1]) { set_error("gbn_block_view: a block twice"); return GBN_ERR_ARG; }
    if (n == 1) { *out = const_cast<GbnDb *>(parts[0]); return GBN_OK; }
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_view_cache.find(parts);
        if (it != g_view_cache.end()) { *out = it->second; return GBN_OK; }
    }
    // every subject is addressed from the lowest slab: tiles carry 32-bit offsets in units of 16 bytes (GbnTile::off16)
    const uint8_t *base = parts[0]->d_packed; const uint8_t *top = base;
    for (const GbnDb *p : parts) { base = std::min(base, p->d_packed); top = std::max(top, p->d_packed + p->nbytes); }
    if ((((uintptr_t)base) & 15) || (uint64_t)(top - base) >= ((uint64_t)1 << 36)) {
        set_error("gbn_block_view: the blocks' slabs lie too far apart for one view (search them one by one)"); return GBN_ERR_UNSUPPORTED; }
    Engine *eng = static_cast<Engine *>(parts[0]->engine);
    enter(eng);
    GbnDb *v = new GbnDb();
    v->engine = eng; v->d_packed = base; v->owns = false; v->nbytes = (int64_t)(top - base); v->view_parts = parts;
    v->chunk_len = parts[0]->chunk_len;
    bool any_amb = false; int32_t last_oid = -1; bool ascending = true;
    for (const GbnDb *p : parts) any_amb = any_amb || !p->amb.empty();
    for (const GbnDb *p : parts) {
        const int64_t delta = (int64_t)(p->d_packed - base);
        if (delta & 15) { delete v; set_error("gbn_block_view: a slab that is not 16-byte aligned"); return GBN_ERR_ARG; }
        for (int32_t s = 0; s < p->num_seqs; s++) {
            v->byte_off.push_back(delta + p->byte_off[(size_t)s]); v->len.push_back(p->len[(size_t)s]);
            const int32_t oid = p->oid_of(s);
            ascending = ascending && oid > last_oid; last_oid = oid;
            v->oid_map.push_back(oid);
            if (any_amb) v->amb.push_back(p->amb.empty() ? std::vector<GbnDb::AmbRun>() : p->amb[(size_t)s]);
        }
        v->total_bases += p->total_bases;
    }
    if (!ascending) { delete v; set_error("gbn_block_view: the blocks' OIDs overlap"); return GBN_ERR_ARG; }
    v->num_seqs = v->real_seqs = (int32_t)v->len.size(); v->first_oid = v->oid_map.empty() ? 0 : v->oid_map[0];
    int rc;
    if ((rc = dev_upload(v->d_byte_off, v->byte_off.data(), v->byte_off.size())) || (rc = dev_upload(v->d_len, v->len.data(), v->len.size()))) { free_view(v); return rc; }
    GbnDb *loser = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_view_cache.find(parts);
        if (it == g_view_cache.end()) { g_view_cache.emplace(parts, v); *out = v; }
        else { *out = it->second; loser = v; }          // (two threads built it at the same time: the first stays)
    }
    if (loser) free_view(loser);
    return GBN_OK;
    });
}
// the views over `block` (nullptr: all of them) leave the cache and are freed; the calling thread holds no lock
static void drop_views_of(const GbnDb *block) {
    std::vector<GbnDb *> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_view_cache.begin(); it != g_view_cache.end(); ) {
            const bool has = !block || std::find(it->first.begin(), it->first.end(), block) != it->first.end();
            if (has) { drop.push_back(it->second); it = g_view_cache.erase(it); } else ++it;
        }
    }
    for (GbnDb *v : drop) free_view(v);
}

// tests: seed_order.hip on segments given in host memory.  The keys of the seeds ordered by (subject, slot), scan order
// inside, as the engine's seed stage builds them for the composite-key form (q_bits from qlen, s_bits from max_len;
// container_hash: 512 slots, else diag_len slots).
int gbn_debug_seed_order(const GbnDevSeed *seg, const uint32_t *seg_count, int nseg, uint32_t seg_cap, int nsubj, int subj_base,
                         int container_hash, int diag_len, int32_t qlen, int32_t max_len, int q_descending, uint64_t *keys_out, int64_t *n_out)
{
    return gbn::guard(__func__, [&]() -> int {
    if (!seg || !seg_count || nseg <= 0 || nseg > GBN_SLICE_SEGS || !keys_out || !n_out) { set_error("gbn_debug_seed_order: bad arguments"); return GBN_ERR_ARG; }
    int rc = GBN_OK;
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    GbnKeyParams K; std::memset(&K, 0, sizeof(K));
    int64_t n = 0;
    for (int g = 0; g < nseg; g++) n += std::min(seg_count[g], seg_cap);
    *n_out = n;
    if (n == 0) return GBN_OK;
    K.n = n; K.q_descending = q_descending; K.container_hash = container_hash; K.diag_len = diag_len;
    K.q_bits = std::min(32, bits_for((uint64_t)qlen + 1));
    K.group_bits = container_hash ? 9 : bits_for((uint64_t)std::max(diag_len, 2));
    K.s_bits = bits_for((uint64_t)max_len + 1); K.qh_bits = std::max(0, K.q_bits - K.group_bits);
    K.subj_base = subj_base; K.v_bits = 8 + K.qh_bits;
    if (K.group_bits + bits_for((uint64_t)nsubj + 1) + K.s_bits + K.v_bits > 64) { set_error("gbn_debug_seed_order: the key does not fit 64 bits"); return GBN_ERR_ARG; }
    GbnDevSeed *d_seg = nullptr; uint32_t *d_cnt = nullptr, *d_tmp = nullptr; unsigned long long *d_first = nullptr; uint64_t *d_keys = nullptr;
    auto done = [&](int code) { dev_free(d_seg); dev_free(d_cnt); dev_free(d_tmp); dev_free(d_first); dev_free(d_keys); return code; };
    if ((rc = dev_upload(d_seg, seg, (size_t)nseg * seg_cap)) || (rc = dev_upload(d_cnt, seg_count, (size_t)nseg)) ||
        (rc = dev_alloc(d_first, (size_t)nseg + 1)) || (rc = dev_alloc(d_keys, (size_t)n)) ||
        (rc = dev_alloc(d_tmp, seed_order_scratch_words(n, nsubj, K.group_bits)))) return done(rc);
    K.seg = d_seg; K.seg_count = d_cnt; K.nseg = nseg; K.seg_cap = seg_cap; K.seg_first = d_first; K.key_scan = d_keys;
    if (launch_seed_order(K, nsubj, d_tmp, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess ||
        hipMemcpy(keys_out, d_keys, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { set_error("gbn_debug_seed_order: launch failed"); return done(GBN_ERR_HIP); }
    return done(GBN_OK);
    });
}
void gbn_release_db_memory(void) {
    drop_views_of(nullptr);
    std::map<const void *, GbnDb *> drop;
    std::map<BlockKey, GbnDb *> drop_blocks;
    { std::lock_guard<std::mutex> lk(g_cache_mu); drop.swap(g_db_cache); drop_blocks.swap(g_block_cache); }
    for (auto &kv : drop) gbn_db_free(kv.second);
    for (auto &kv : drop_blocks) gbn_db_free(kv.second);
}

static void release_engine() {              // (the calling thread has entered it)
    std::lock_guard<std::mutex> lk(E.mu);
    if (!E.ready) return;
    (void)wait_pending();
    (void)hipDeviceSynchronize();                       // nothing of ours is queued or running when buffers, streams and events go
    (void)pool_check_guards();
    rec_purge(nullptr); recset_free(E.scratch); recset_free(E.alt); E.last_key_valid = false;
    dev_free(E.slice_seg); E.slice_seg_cap = 0; dev_free(E.seg_counts); dev_free(E.seg_firsts);
    if (E.scan_back) { (void)hipHostFree(E.scan_back); E.scan_back = nullptr; }
    E.ahead.valid = false;
    for (int i = 0; i < 4; i++) if (E.ahead.ev[i >> 1][i & 1]) { (void)hipEventDestroy(E.ahead.ev[i >> 1][i & 1]); E.ahead.ev[i >> 1][i & 1] = nullptr; }
    if (E.ev_back) { (void)hipEventDestroy(E.ev_back); E.ev_back = nullptr; }
    hitbuf_drain();
    dev_free(E.seeds_async); E.seeds_async_cap = 0; if (E.ev_seed) { (void)hipEventDestroy(E.ev_seed); E.ev_seed = nullptr; }
    dev_free(E.seeds);
    for (auto &KS : E.ks) { dev_free(KS.key_a); dev_free(KS.key_b); dev_free(KS.idx_a); dev_free(KS.idx_b); dev_free(KS.cell_diag); dev_free(KS.cell_level); dev_free(KS.ext_rec); dev_free(KS.sort_tmp); KS.key_cap = 0; KS.sort_tmp_bytes = 0; }
    for (int i = 0; i < 2; i++) { dev_free(E.ihits_s[i]); dev_free(E.gapped_s[i]); dev_free(E.gap_scratch_s[i]); E.ihit_cap_s[i] = E.gap_scratch_ints_s[i] = 0; }
    dev_free(E.counters); dev_free(E.rareq); E.rareq_cap = 0; dev_free(E.rare_counts);
    for (int i = 0; i < 2; i++) { E.ks[i].kt.destroy(); E.kt_gap[i].destroy(); }
    E.seed_cap = 0;
    if (E.gather_stage) (void)hipHostFree(E.gather_stage);
    E.gather_stage = nullptr; E.gather_stage_cap = 0;
    if (E.gather_stream) (void)hipStreamDestroy(E.gather_stream);
    E.gather_stream = nullptr;
    if (E.ev0) (void)hipEventDestroy(E.ev0);
    if (E.ev1) (void)hipEventDestroy(E.ev1);
    for (int i = 0; i < 4; i++) { if (E.evk[i]) (void)hipEventDestroy(E.evk[i]); E.evk[i] = nullptr; }
    if (E.stream) (void)hipStreamDestroy(E.stream);
    if (E.stream2) (void)hipStreamDestroy(E.stream2);
    if (E.stream_build) (void)hipStreamDestroy(E.stream_build);
    E.stream_build = nullptr;
    pool_drain(E.device);
    E.ev0 = E.ev1 = nullptr; E.stream = E.stream2 = nullptr; E.ready = false;
}
// every engine of the process: its stages finished, its device idle, its buffers, streams and events freed.  Batches,
// shards and results made before stay valid handles to free, nothing else (as after the reference's ReleaseGPUs).
void gbn_release(void) {
    std::vector<Engine *> all;
    { std::lock_guard<std::mutex> lk(g_eng_mu); for (int d = 0; d < kMaxDevices; d++) if (g_eng[d]) all.push_back(g_eng[d]); }
    for (Engine *e : all) { enter(e); release_engine(); }
    // (the Engine objects stay: handles made before the release still point at them; a later gbn_init re-arms them)
    { std::lock_guard<std::mutex> lk(g_eng_mu); g_default_dev = -1; }
    tl_eng = nullptr; tl_sel = -1;
}

// subjects appended one at a time (the shim: what BlastSeqSrcGetSequence hands out) into the slab layout of gbn_db_new
struct GbnShardBuilder { std::vector<uint8_t> bytes; std::vector<int64_t> off; std::vector<int32_t> len, oid; bool explicit_oids = false; };
int gbn_shard_builder_new(GbnShardBuilder **out, int32_t expected_seqs) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out) return GBN_ERR_ARG;
    GbnShardBuilder *b = new (std::nothrow) GbnShardBuilder();
    if (!b) return GBN_ERR_NOMEM;
    if (expected_seqs > 0) { b->off.reserve(expected_seqs); b->len.reserve(expected_seqs); }
    *out = b;
    return GBN_OK;
    });
}
int gbn_shard_builder_add_oid(GbnShardBuilder *b, int32_t oid, const uint8_t *ncbi2na, int32_t length) {
    return gbn::guard(__func__, [&]() -> int {
    if (!b || oid < 0 || (!b->oid.empty() && oid <= b->oid.back()) || (b->oid.empty() && !b->len.empty())) {
        set_error("gbn_shard_builder_add_oid: OIDs must ascend, and every subject of the shard needs one"); return GBN_ERR_ARG; }
    const int rc = gbn_shard_builder_add(b, ncbi2na, length);
    if (rc == GBN_OK) { b->oid.push_back(oid); b->explicit_oids = true; }
    return rc;
    });
}
int gbn_shard_builder_add(GbnShardBuilder *b, const uint8_t *ncbi2na, int32_t length) {
    return gbn::guard(__func__, [&]() -> int {
    if (!b || length < 0 || (length > 0 && !ncbi2na)) { set_error("gbn_shard_builder_add: bad argument"); return GBN_ERR_ARG; }
    const size_t at = std::max<size_t>(16, (b->bytes.size() + 15) & ~(size_t)15), nb      // 16 readable bytes in front of the first subject
         = ((size_t)length + 3) / 4;
    try { b->bytes.resize(at + nb, 0); b->off.push_back((int64_t)at); b->len.push_back(length); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return GBN_ERR_NOMEM; }
    if (nb) std::memcpy(b->bytes.data() + at, ncbi2na, nb);
    // the last byte of a stored sequence carries the remainder count in its low bits (sequence_files.txt:60-90): bases only
    if (length & 3) b->bytes[at + nb - 1] &= (uint8_t)(0xff << (2 * (4 - (length & 3))));
    return GBN_OK;
    });
}
int gbn_shard_builder_finish(GbnShardBuilder *b, GbnDb **out) {
    return gbn::guard(__func__, [&]() -> int {
    if (!b || !out || b->len.empty()) { set_error("gbn_shard_builder_finish: no subjects"); return GBN_ERR_ARG; }
    try { b->bytes.resize(((b->bytes.size() + 15) & ~(size_t)15) + 128, 0); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return GBN_ERR_NOMEM; }
    if (b->explicit_oids && b->oid.size() != b->len.size()) { set_error("gbn_shard_builder_finish: _add and _add_oid were mixed"); return GBN_ERR_ARG; }
    int rc = gbn_db_new(out, b->bytes.data(), (int64_t)b->bytes.size(), (int32_t)b->len.size(), b->off.data(), b->len.data(), b->explicit_oids ? b->oid[0] : 0, 0);
    std::vector<uint8_t>().swap(b->bytes);
    if (rc == GBN_OK && b->explicit_oids && b->oid.back() - b->oid[0] + 1 != (int32_t)b->oid.size()) (*out)->oid_map = b->oid;    // holes: the map
    return rc;
    });
}
void gbn_shard_builder_free(GbnShardBuilder *b) { delete b; }

// MAX_DBSEQ_LEN of the build the results are to equal: 200,000,000 in G-BLASTN (COREI/blast_gapalign.h:54-55;
// 5,000,000 in stock BLAST+).  A multiple of 4; tests lower it to exercise the chunk path on small subjects.
static int32_t g_max_dbseq_len = 200000000;
int gbn_set_max_dbseq_len(int32_t n) {
    return gbn::guard(__func__, [&]() -> int {
    if (n < 1000 || (n & 3)) { set_error("gbn_set_max_dbseq_len: a multiple of 4, at least 1000"); return GBN_ERR_ARG; }
    g_max_dbseq_len = n;
    return GBN_OK;
    });
}

int gbn_db_new(GbnDb **out, const uint8_t *packed, int64_t nbytes, int32_t num_seqs,
               const int64_t *byte_off, const int32_t *len, int32_t first_oid, int is_device) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || !packed || num_seqs < 0 || (num_seqs > 0 && (!byte_off || !len))) { set_error("bad argument"); return GBN_ERR_ARG; }
    int rc = enter_current();
    if (rc) return rc;
    GbnDb *db = new GbnDb();
    db->engine = tl_eng;
    db->first_oid = first_oid; db->real_seqs = num_seqs; db->chunk_len = g_max_dbseq_len;
    bool chunked = false;
    for (int32_t i = 0; i < num_seqs; i++) {
        if (byte_off[i] < 16 || (byte_off[i] & 15) || byte_off[i] + (len[i] + 3) / 4 + 128 > nbytes) {
            delete db; set_error("subject offsets must be 16-byte aligned, >= 16, and leave 128 pad bytes"); return GBN_ERR_ARG;
        }
        db->total_bases += len[i];
        chunked = chunked || len[i] > g_max_dbseq_len;
    }
    if (!chunked) {
        db->num_seqs = num_seqs; db->nbytes = nbytes;
        db->byte_off.assign(byte_off, byte_off + num_seqs); db->len.assign(len, len + num_seqs);
        if (is_device) { db->d_packed = packed; db->owns = false; }
        else {
            uint8_t *p = nullptr;
            if (hipMalloc((void **)&p, (size_t)nbytes) != hipSuccess) { delete db; set_error("hipMalloc(db) failed"); return GBN_ERR_NOMEM; }
            if (hipMemcpy(p, packed, (size_t)nbytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); delete db; set_error("H2D(db) failed"); return GBN_ERR_HIP; }
            g_db_bytes_uploaded += (long long)nbytes;
            db->d_packed = p; db->owns = true;
        }
    } else {
        // s_GetNextSubjectChunk (CORE/blast_engine.c:218-262) without hard masks: chunk k of a sequence starts at
        // k * (MAX_DBSEQ_LEN - DBSEQ_CHUNK_OVERLAP) and is MAX_DBSEQ_LEN long, the last one runs to the end.  Every
        // chunk gets a 16-byte aligned copy in a slab of this shard's own.
        const int64_t stride = (int64_t)g_max_dbseq_len - kDbseqChunkOverlap;
        std::vector<int64_t> src;       // byte offset of every chunk in the caller's slab
        int64_t pos = 16;
        for (int32_t i = 0; i < num_seqs; i++) {
            db->first_virt.push_back((int32_t)db->len.size()); db->real_len.push_back(len[i]);
            int32_t ord = 0;
            for (int64_t off = 0;; off += stride, ord++) {
                const bool last = off + g_max_dbseq_len >= len[i];
                const int32_t clen = last ? (int32_t)(len[i] - off) : g_max_dbseq_len;
                db->real_of.push_back(i); db->chunk_ord.push_back(ord);
                db->len.push_back(clen); db->byte_off.push_back(pos); src.push_back(byte_off[i] + off / 4);
                pos += (((int64_t)clen + 3) / 4 + 15) / 16 * 16;
                if (last) break;
            }
        }
        db->num_seqs = (int32_t)db->len.size(); db->nbytes = pos + 128;
        uint8_t *p = nullptr;
        if (hipMalloc((void **)&p, (size_t)db->nbytes) != hipSuccess) { delete db; set_error("hipMalloc(db) failed"); return GBN_ERR_NOMEM; }
        // (the bytes between the chunk copies: defined, like the pad bytes of a caller's slab)
        if (hipMemset(p, pool_poison() >= 0 ? pool_poison() : 0, (size_t)db->nbytes) != hipSuccess) { (void)hipFree(p); delete db; set_error("hipMemset(db) failed"); return GBN_ERR_HIP; }
        db->d_packed = p; db->owns = true;
        hipError_t e = hipMemset(p, 0, (size_t)db->nbytes);
        for (size_t v = 0; v < db->len.size() && e == hipSuccess; v++) {
            const size_t nb = ((size_t)db->len[v] + 3) / 4;
            e = hipMemcpy(p + db->byte_off[v], packed + src[v], nb, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
            // a chunk ends inside its sequence's byte: the bases past its end are the next chunk's, not padding
            if (e == hipSuccess && (db->len[v] & 3)) {
                uint8_t lastb = 0;
                e = hipMemcpy(&lastb, p + db->byte_off[v] + nb - 1, 1, hipMemcpyDeviceToHost);
                lastb &= (uint8_t)(0xff << (2 * (4 - (db->len[v] & 3))));
                if (e == hipSuccess) e = hipMemcpy(p + db->byte_off[v] + nb - 1, &lastb, 1, hipMemcpyHostToDevice);
            }
        }
        if (e != hipSuccess) { gbn_db_free(db); set_error("copying subject chunks failed"); return GBN_ERR_HIP; }
    }
    if ((rc = dev_upload(db->d_byte_off, db->byte_off.data(), db->byte_off.size())) ||
        (rc = dev_upload(db->d_len, db->len.data(), db->len.size()))) { gbn_db_free(db); return rc; }
    *out = db;
    return GBN_OK;
    });
}
// ambiguity runs of sequence `local` (0-based in the shard), values in NCBI4na as the database stores them
// (gbn_blastdb_get_ambiguities); gbn_blastdb_load_shard calls this for every sequence that has runs
int gbn_db_set_ambiguities(GbnDb *db, int32_t local, int32_t n, const int32_t *start, const int32_t *length, const uint8_t *ncbi4na) {
    return gbn::guard(__func__, [&]() -> int {
    static const uint8_t kNa4ToBlastna[16] = {15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14};    // CORE/blast_encoding.c:42-59
    if (!db || local < 0 || local >= db->real_seqs || n < 0 || (n > 0 && (!start || !length || !ncbi4na))) { set_error("gbn_db_set_ambiguities: bad argument"); return GBN_ERR_ARG; }
    if (db->amb.empty()) db->amb.resize((size_t)db->real_seqs);
    auto &v = db->amb[(size_t)local];
    v.clear();
    for (int32_t i = 0; i < n; i++) v.push_back(GbnDb::AmbRun{start[i], length[i], kNa4ToBlastna[ncbi4na[i] & 15]});
    return GBN_OK;
    });
}

static void free_db_now(GbnDb *db);
static void free_view(GbnDb *v) { free_db_now(v); }
void gbn_db_free(GbnDb *db) {
    if (!db) return;
    if (!db->view_parts.empty()) {                      // a view: out of the view cache
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_view_cache.begin(); it != g_view_cache.end(); ) { if (it->second == db) it = g_view_cache.erase(it); else ++it; }
    } else drop_views_of(db);                           // a block: the views that reach into it go first
    free_db_now(db);
}
static void free_db_now(GbnDb *db) {
    if (!db->engine) { delete db; return; }
    enter(static_cast<Engine *>(db->engine));
    {   // a stage in flight may still read this shard
        std::lock_guard<std::mutex> lk(E.mu);
        if (E.has_pending) (void)wait_pending();
        wait_host();
        if (E.ahead.valid && E.ahead.key.db == (const void *)db) { (void)hipStreamSynchronize(E.stream); E.ahead.valid = false; }    // (a binning kernel queued ahead reads the shard)
        rec_purge((const void *)db);                        // the scan records of this shard go with it
        if (E.last_key.db == (const void *)db) E.last_key_valid = false;
    }
    free_tile_cache(*db);
    if (db->owns && db->d_packed) (void)hipFree((void *)db->d_packed);
    dev_free(db->d_byte_off); dev_free(db->d_len);
    delete db;
}
int64_t gbn_db_total_bases(const GbnDb *db) { return db ? db->total_bases : 0; }
int32_t gbn_db_num_seqs(const GbnDb *db) { return db ? db->real_seqs : 0; }

int gbn_synth_fill(void *dev_ptr, int64_t nbytes, uint64_t seed, void *stream) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = enter_current();
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : E.stream;
    HIPCHK(launch_synth_fill(dev_ptr, nbytes, seed, st));
    HIPCHK(hipStreamSynchronize(st));
    return GBN_OK;
    });
}

int gbn_batch_new_masked(GbnBatch **out, const GbnOptions *opt, int32_t nq, const uint8_t *const *seqs,
                         const int32_t *lens, int32_t nmask, const int32_t *mask_query, const int32_t *mask_from,
                         const int32_t *mask_to, int upload) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || !opt || nq <= 0 || !seqs || !lens || nmask < 0 || (nmask > 0 && (!mask_query || !mask_from || !mask_to))) {
        set_error("bad argument"); return GBN_ERR_ARG;
    }
    std::vector<QueryMask> masks((size_t)nmask);
    for (int32_t i = 0; i < nmask; i++) masks[(size_t)i] = QueryMask{mask_query[i], mask_from[i], mask_to[i]};
    std::unique_ptr<GbnBatch, void (*)(GbnBatch *)> b(new GbnBatch(), gbn_batch_free);      // (freed if the set-up throws)
    // with a device the lookup tables are built there (upload_batch); a host-only set-up fills them here
    int rc = build_batch(*b, *opt, nq, seqs, lens, masks, /* host_tables = */ upload == 0);
    if (rc == GBN_OK && upload) rc = upload_batch(*b);
    if (rc != GBN_OK) return rc;
    *out = b.release();
    return GBN_OK;
    });
}

int gbn_batch_new_ex(GbnBatch **out, const GbnOptions *opt, int32_t nq, const uint8_t *const *seqs,
                     const int32_t *lens, int upload) {
    return gbn::guard(__func__, [&]() -> int {
    return gbn_batch_new_masked(out, opt, nq, seqs, lens, 0, nullptr, nullptr, nullptr, upload);
    });
}

int gbn_batch_new(GbnBatch **out, const GbnOptions *opt, int32_t nq, const uint8_t *const *seqs, const int32_t *lens) {
    return gbn::guard(__func__, [&]() -> int {
    return gbn_batch_new_ex(out, opt, nq, seqs, lens, 1);
    });
}
// the launchers' parameter blocks for a caller that holds a batch and a shard: database, lookup and query members
// (everything marked [caller] in gblastn_amd_kernels.h stays zero)
static int params_ready(const GbnBatch *b, const GbnDb *db) {
    if (!b || !db || !b->dev) { set_error("batch without device structures (gbn_batch_new_ex upload = 0?)"); return GBN_ERR_ARG; }
    if (b->dev->eng != db->engine) { set_error("the batch and the shard live on different devices"); return GBN_ERR_ARG; }
    enter(b->dev->eng);
    if (b->dev->ready) HIPCHK(hipEventSynchronize(b->dev->ready));     // deferred lookup build
    return GBN_OK;
}
int gbn_batch_scan_params(const GbnBatch *b, const GbnDb *db, GbnScanParams *out) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = params_ready(b, db); if (rc) return rc;
    if (!out) return GBN_ERR_ARG;
    TileSet none;
    fill_scan_params(*out, *b, *db, none);
    out->seeds = nullptr; out->seed_count = nullptr; out->seed_cap = 0; out->raw_hits = nullptr;
    return GBN_OK;
    });
}
int gbn_batch_ext_params(const GbnBatch *b, const GbnDb *db, GbnExtParams *X) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = params_ready(b, db); if (rc) return rc;
    if (!X) return GBN_ERR_ARG;
    const DeviceBatch *d = b->dev;
    std::memset(X, 0, sizeof(*X));
    X->db = db->d_packed; X->byte_off = db->d_byte_off; X->len = db->d_len;
    X->q8 = d->q8; X->qlen = b->qlen; X->q2 = d->q2; X->qinv = d->qinv; X->q4 = d->q4_base; X->q4_plane = d->q4_plane; X->q4_origin = b->qpad;
    X->ctx_off = d->ctx_off; X->ctx_len = d->ctx_len; X->ctx_xdrop = d->ctx_xdrop;
    X->ctx_cutoff = d->ctx_cutoff; X->ctx_reduced = d->ctx_reduced; X->nctx = (int32_t)b->ctx.size();
    X->matrix = d->matrix; X->score_table = d->score_table;
    X->word = b->lut.word; X->container_hash = b->container;
    X->cell_start = d->cell_start; X->ent = d->ent; X->cell_mask = (uint32_t)(b->lut.ncells - 1); X->lut = b->lut.lut;
    X->masked = b->lut.masked ? 1 : 0;
    X->ctx_hint = d->ctx_hint; X->ctx_hint_shift = kCtxHintShift; X->ctx_blk = d->ctx_blk; X->ctx_pack = d->ctx_pack;
    return GBN_OK;
    });
}
int gbn_batch_gap_params(const GbnBatch *b, const GbnDb *db, GbnGapParams *G) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = params_ready(b, db); if (rc) return rc;
    if (!G) return GBN_ERR_ARG;
    const DeviceBatch *d = b->dev;
    std::memset(G, 0, sizeof(*G));
    G->db = db->d_packed; G->byte_off = db->d_byte_off; G->len = db->d_len;
    G->q8 = d->q8; G->q2 = d->q2; G->qinv = d->qinv; G->ctx_off = d->ctx_off; G->ctx_len = d->ctx_len; G->nctx = (int32_t)b->ctx.size();
    G->matrix = d->matrix; G->reward = b->opt.reward; G->penalty = b->opt.penalty;
    G->gap_open = b->opt.gap_open; G->gap_extend = b->opt.gap_extend; G->xdrop = b->gap_x_dropoff;
    int32_t max_len = 0, max_ctx = 0, row_len = 0;
    for (int32_t l : db->len) max_len = std::max(max_len, l);
    for (auto &c : b->ctx) max_ctx = std::max(max_ctx, c.query_length);
    G->scratch_per_thread = (int32_t)gap_scratch_ints(*b, max_len, max_ctx, &row_len);
    G->row_len = row_len;
    return GBN_OK;
    });
}
int gbn_batch_diag_layout(const GbnBatch *b, int32_t *container_hash, int32_t *diag_len, int32_t *q_descending) {
    return gbn::guard(__func__, [&]() -> int {
    if (!b) return GBN_ERR_ARG;
    if (container_hash) *container_hash = b->container;
    if (diag_len) *diag_len = b->diag_len;
    if (q_descending) *q_descending = b->lut.type == GBN_LUT_MB ? 1 : 0;
    return GBN_OK;
    });
}
int gbn_launch_scan_seed(const GbnScanParams *p, int grid, void *stream) {
    return gbn::guard(__func__, [&]() -> int {
    if (!p) return GBN_ERR_ARG;
    HIPCHK(launch_scan_seed(*p, grid, (hipStream_t)stream));
    return GBN_OK;
    });
}
int gbn_launch_ungapped(const GbnExtParams *p, void *stream) {
    return gbn::guard(__func__, [&]() -> int {
    if (!p) return GBN_ERR_ARG;
    HIPCHK(launch_diag_ungapped(*p, (hipStream_t)stream));
    return GBN_OK;
    });
}
int gbn_launch_gapped(const GbnGapParams *p, int greedy, void *stream) {
    return gbn::guard(__func__, [&]() -> int {
    if (!p) return GBN_ERR_ARG;
    HIPCHK(launch_gapped(*p, greedy != 0, (hipStream_t)stream));
    return GBN_OK;
    });
}
void gbn_batch_free(GbnBatch *b) {
    if (!b) return;
    if (b->dev && b->dev->eng) {
        enter(b->dev->eng);
        // an extension stage still reading this batch finishes first (its memory goes back to the pool, not to hipFree)
        { std::lock_guard<std::mutex> lk(E.mu); if (E.has_pending && E.pending_batch == b) (void)wait_pending_gpu(); }
        wait_tail(b->host_tail);                            // (a queued host replay reads the batch's options and contexts; the engine is not locked meanwhile)
    }
    free_device_batch(b->dev); delete b;
}
int32_t gbn_batch_num_contexts(const GbnBatch *b) { return (int32_t)b->ctx.size(); }
const GbnContext *gbn_batch_contexts(const GbnBatch *b) { return b->ctx.data(); }
int gbn_batch_karlin_gapped(const GbnBatch *b, double *lambda, double *K) {
    return gbn::guard(__func__, [&]() -> int {
    if (!b || !lambda || !K) { set_error("bad argument"); return GBN_ERR_ARG; }
    *lambda = b->kbp_gap.lambda; *K = b->kbp_gap.K;
    return GBN_OK;
    });
}
int32_t gbn_batch_lut_type(const GbnBatch *b) { return b->lut.type; }
int32_t gbn_batch_lut_width(const GbnBatch *b) { return b->lut.lut; }
int32_t gbn_batch_scan_step(const GbnBatch *b) { return b->lut.step; }
int32_t gbn_batch_scan_path(const GbnBatch *b) { return scan_slices(*b) > 0 ? 2 : (choose_bins(*b) == 1 ? 1 : 0); }
int32_t gbn_batch_diag_container(const GbnBatch *b) { return b->container; }
int32_t gbn_batch_gap_x_dropoff(const GbnBatch *b) { return b->gap_x_dropoff; }

int gbn_results_new(GbnResults **out) { return gbn::guard(__func__, [&]() -> int { if (!out) return GBN_ERR_ARG; *out = new GbnResults(); return GBN_OK; }); }
void gbn_results_free(GbnResults *r) {
    if (!r) return;
    if (r->engine) {                                        // a stage of the engine that filled them may still write to them
        enter(static_cast<Engine *>(r->engine));
        { std::lock_guard<std::mutex> lk(E.mu); if (E.has_pending && E.pending_res == r) (void)wait_pending_gpu(); }
        wait_tail(r->host_tail);
        { std::lock_guard<std::mutex> lk2(E.failed_mu); E.failed.erase(r); }
    }
    delete r;
}
void gbn_results_clear(GbnResults *r) { if (r) { r->hsps.clear(); r->seeds.clear(); r->init_hits.clear(); } }
int64_t gbn_results_num_hsps(const GbnResults *r) { return (int64_t)r->hsps.size(); }
const GbnHSP *gbn_results_hsps(const GbnResults *r) { return r->hsps.data(); }
int64_t gbn_results_num_seeds(const GbnResults *r) { return (int64_t)r->seeds.size(); }
const GbnSeed *gbn_results_seeds(const GbnResults *r) { return r->seeds.data(); }
int64_t gbn_results_num_init_hits(const GbnResults *r) { return (int64_t)r->init_hits.size(); }
const GbnInitHit *gbn_results_init_hits(const GbnResults *r) { return r->init_hits.data(); }

// The subject ranges a shard is searched in: bounded by packed size so that scratch stays modest, and by the width of
// the position ids.
static void plan_ranges(const GbnDb &dbr, int step, std::vector<std::pair<int32_t, int32_t>> &out) {
    const GbnDb *db = &dbr;
    int64_t range_gib = 16;
    if (gbn::switch_is_set("GBN_RANGE_GIB")) range_gib = (int)std::max<long long>(1, gbn::switch_value("GBN_RANGE_GIB", 0));
    int64_t range_bytes = range_gib << 30;
    if (gbn::switch_is_set("GBN_RANGE_MIB")) range_bytes = (int64_t)std::max<long long>(1, gbn::switch_value("GBN_RANGE_MIB", 0)) << 20;    // tests
    // Hard limits of a range: packed bytes (scratch) and 32-bit position ids.  Seed-rich shapes (small
    // stride) are cut into ~1 G scan positions, so that the seed / extension stages of one range run
    // underneath the scan of the next.  Whatever number of ranges that takes, they are made equal:
    // a big range followed by a small remainder would leave nothing to overlap with.
    int64_t tile_limit = ((int64_t)1 << (32 - GBN_BIN_TILE_BITS)) - 1;
    if (step <= 4) tile_limit = std::min<int64_t>(tile_limit, (int64_t)1 << 17);
    if (gbn::switch_is_set("GBN_RANGE_TILES")) tile_limit = (int)std::max<long long>(1, gbn::switch_value("GBN_RANGE_TILES", 0));                  // tests
    auto tiles_of = [&](int32_t s) { return (int64_t)(db->len[s] / step) / GBN_BIN_TILE_POS + 1; };
    int64_t all_bytes = 0, all_tiles = 0;
    for (int32_t s = 0; s < db->num_seqs; s++) { all_bytes += (db->len[s] + 3) / 4; all_tiles += tiles_of(s); }
    const int64_t nranges = std::max<int64_t>(1, std::max((all_bytes + range_bytes - 1) / range_bytes, (all_tiles + tile_limit - 1) / tile_limit));
    const int64_t want_bytes = (all_bytes + nranges - 1) / nranges, want_tiles = (all_tiles + nranges - 1) / nranges;
    int32_t s0 = 0;
    while (s0 < db->num_seqs) {
        int32_t s1 = s0; int64_t acc = 0, tiles = 0;
        while (s1 < db->num_seqs) {
            const int64_t nb = (db->len[s1] + 3) / 4, nt = tiles_of(s1);
            if (s1 > s0 && (acc + nb > range_bytes || tiles + nt > tile_limit)) break;       // hard limits
            if (s1 > s0 && (acc >= want_bytes || tiles >= want_tiles)) break;                // equal shares
            acc += nb; tiles += nt; s1++;
        }
        out.emplace_back(s0, s1);
        s0 = s1;
    }
}
// The scan records a query batch of these lengths will want of this shard, queued NOW: the binning kernel reads the
// subjects only, so a caller that knows its next batch's size starts it before the batch is set up -- the kernel runs
// underneath the batch's set-up (host work + table build: 5 ms for a 5 Mb batch), and the batch's pass finds the set in the
// record cache (queued on the engine's stream, in front of its own probe kernel).  Returns at once; does nothing when the
// record cache is off, when the predicted table is scanned without records (lut = word, tiny tables), or when the sets are
// there already.  A batch that comes out with another shape (masked queries near a threshold of the table choice) bins for
// itself as ever.
int gbn_db_prepare_records(GbnDb *db, const GbnOptions *opt, int32_t nq, const int32_t *lens) {
    return gbn::guard(__func__, [&]() -> int {
    if (!db || !opt || nq <= 0 || !lens || !db->engine) { set_error("gbn_db_prepare_records: bad argument"); return GBN_ERR_ARG; }
    enter(static_cast<Engine *>(db->engine));
    if (!E.ready) { set_error("the engine was released (gbn_release) after this shard was made"); return GBN_ERR_ARG; }
    std::lock_guard<std::mutex> lk(E.mu);
    int type = 0, lut = 0, step = 0;
    gbn::predict_table_shape(*opt, nq, lens, type, lut, step);
    const int word = opt->word_size;
    int64_t nb = ((int64_t)1 << (2 * lut)) >> GBN_BIN_CBITS(lut);
    if (lut == word || nb < 2 || nb > GBN_BIN_MAXNB || gbn::switch_value("GBN_SCAN_BINS", 0) == 1 || opt->db_num_seqs == 0) return GBN_OK;
    const long long limit = rec_limit_bytes();
    if (limit <= 0) return GBN_OK;
    // fingerprint widths as upload_batch derives them from word - lut
    const int e = word - lut, h = (e + 1) / 2;
    const int fl = std::min(8, h), fr = std::min(7, e - h + 1);
    std::vector<std::pair<int32_t, int32_t>> ranges;
    plan_ranges(*db, step, ranges);
    for (const auto &rg : ranges) {
        const int32_t s0 = rg.first, s1 = rg.second;
        const TileSet *tsp = nullptr;
        int rc = get_tiles(*db, lut, step, GBN_BIN_TILE_POS, s0, s1, &tsp);
        if (rc) return rc;
        if (tsp->ntiles == 0 || tsp->ntiles > (1 << 19)) continue;
        BinLayout BL;
        if (bin_layout((int)nb, tsp->ntiles, bin_positions(*db, s0, s1, lut, step), 1.25, BL)) continue;
        RecKey key; key.db = (const void *)db; key.s0 = s0; key.s1 = s1; key.lut = lut; key.step = step; key.nb = (int)nb; key.nwriters = BL.nwriters;
        key.rfl = std::min(4, fl); key.rfrbits = std::min(7, 2 * fr); key.cbits = GBN_BIN_CBITS(lut); key.tiles = (const void *)tsp->d_tiles; key.subcap = BL.subcap;
        if (rec_find(key)) continue;
        if ((long long)BL.bytes() > limit) continue;        // (larger than the whole cache: the pass bins into its own scratch)
        if (E.ahead.valid) { E.ahead.valid = false; E.ahead_misses++; HIPCHK(hipStreamSynchronize(E.stream)); }
        RecordSet *rs = nullptr;
        if ((rc = rec_acquire(key, BL, limit, &rs))) return rc;
        GbnBinParams B; std::memset(&B, 0, sizeof(B));
        B.S.db = db->d_packed; B.S.byte_off = db->d_byte_off; B.S.len = db->d_len; B.S.tiles = tsp->d_tiles; B.S.ntiles = tsp->ntiles;
        B.S.ncells = (int64_t)1 << (2 * lut); B.S.lut = lut; B.S.word = word; B.S.step = step; B.S.fl = fl; B.S.fr = fr;
        B.nb = (int)nb; B.cbits = GBN_BIN_CBITS(lut); B.nwriters = BL.nwriters; B.rfl = key.rfl; B.rfrbits = key.rfrbits;
        B.rec = reinterpret_cast<uint32_t *>(rs->bin_rec); B.tcur = rs->bin_tcur; B.nseq = (uint32_t)BL.nseq; B.gcount = rs->bin_count; B.subcap = (uint32_t)BL.subcap;
        B.overflow = rs->bin_count + BL.nstream;
        if (!E.rare_counts && (rc = dev_alloc(E.rare_counts, (size_t)2048))) return rc;
        B.rare_counts = E.rare_counts;                      // (where a GBN_BIN_TIMING build leaves its clocks)
        HIPCHK(hipMemsetAsync(rs->bin_count + BL.nstream, 0, 16, E.stream));
        HIPCHK(launch_scan_bin_parts(B, std::max(8, E.num_cu & ~7), E.stream, nullptr, 1, nullptr));
        rs->queued = true; rs->stamp = ++E.rec_clock;
        E.rec_prepared++;
    }
    return GBN_OK;
    });
}

// argument checks of the search entry points; the calling thread enters the engine the batch and the shard live on
static int search_enter(GbnBatch *batch, GbnDb *db, GbnResults *results) {
    if (!batch || !db || !results) { set_error("bad argument"); return GBN_ERR_ARG; }
    if (!batch->dev || !batch->dev->eng) { set_error("batch without device structures (gbn_batch_new_ex upload = 0?)"); return GBN_ERR_ARG; }
    if (batch->dev->eng != db->engine) { set_error("the batch and the shard live on different devices"); return GBN_ERR_ARG; }
    if (!batch->dev->eng->ready) { set_error("the engine was released (gbn_release) after this batch was made"); return GBN_ERR_ARG; }
    if (results->engine && results->engine != db->engine) { set_error("results in use on another device"); return GBN_ERR_ARG; }
    enter(batch->dev->eng);
    return GBN_OK;
}
static int run_search(GbnBatch *batch, GbnDb *db, GbnResults *results, GbnDiagnostics *diag,
                      int keep_stages, GbnInterruptFn interrupt, void *progress, int overlap) {
    int rc = GBN_OK;
    std::lock_guard<std::mutex> lk(E.mu);                   // (the caller has entered the engine: search_enter)
    results->engine = tl_eng; results->diag = diag;
    results->merge.kbp_gap = batch->kbp_gap; results->merge.evalue = batch->opt.evalue; results->merge.eff_searchsp.clear();
    for (const GbnContext &c : batch->ctx) results->merge.eff_searchsp.push_back(c.eff_searchsp);
    auto t0 = std::chrono::steady_clock::now();
    trace_mark("search: entered");
    if (!db->real_of.empty()) results->chunk_len = db->chunk_len;
    if (batch->opt.db_num_seqs == 0) {
        // "db_length == 0" branch of the engine: effective lengths and cut-offs are
        // recomputed for every subject (CORE/blast_setup.c:905-932)
        if ((rc = wait_pending())) return rc;           // this mode rewrites the batch's cut-offs per subject
        for (int32_t s = 0; s < db->num_seqs; s++) {
            // (a chunk of a long sequence: the parameters follow the sequence's length, GB/...engine.cpp:1283-1293, and its
            // chunk lists are merged -- e-values with THESE effective lengths -- before the next sequence changes them)
            const bool chunked = !db->real_of.empty();
            if (!chunked || db->chunk_ord[(size_t)s] == 0) {
                batch->set_effective_lengths(chunked ? db->real_len[(size_t)db->real_of[(size_t)s]] : db->len[s], 1);
                batch->update_cutoffs();
                if ((rc = upload_ctx_cutoffs(*batch))) return rc;
            }
            if ((rc = search_range(*batch, *db, s, s + 1, *results, diag, keep_stages))) return rc;
            if (chunked && (s + 1 == db->num_seqs || db->chunk_ord[(size_t)s + 1] == 0)) {
                wait_host();
                merge_chunk_lists(results->hsps, results->chunk_len, results->merge, diag);     // (lists merged before carry pad_ = 0: left as they are)
            }
            if (interrupt && interrupt(progress)) { set_error("interrupted"); return GBN_ERR_INTERRUPTED; }
        }
    } else {
        std::vector<std::pair<int32_t, int32_t>> ranges;
        plan_ranges(*db, batch->lut.step, ranges);
        for (const auto &rg : ranges) {
            const int32_t s0 = rg.first, s1 = rg.second;
            E.want_ahead = overlap && !keep_stages && s0 == 0 && s1 == db->num_seqs && batch->lut.lut != batch->lut.word && gbn::switch_value("GBN_BIN_AHEAD", 1) != 0;    // the pass is ONE range (the next pass of a pipelined caller bins the same) of a megablast shape (a handful of seeds: their stages run on the second stream)
            rc = search_range(*batch, *db, s0, s1, *results, diag, keep_stages, overlap);
            E.want_ahead = false;
            if (rc) return rc;
            if (interrupt && interrupt(progress)) { set_error("interrupted"); return GBN_ERR_INTERRUPTED; }
        }
    }
    if (diag) diag->total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    trace_mark("search: returns");
    return GBN_OK;
}

int gbn_prelim_search(GbnBatch *batch, GbnDb *db, GbnResults *results, GbnDiagnostics *diag,
                      int keep_stages, GbnInterruptFn interrupt, void *progress) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = search_enter(batch, db, results);
    if (rc) return rc;
    rc = run_search(batch, db, results, diag, keep_stages, interrupt, progress, 0);
    std::lock_guard<std::mutex> lk(E.mu);
    (void)wait_pending();                               // of an earlier gbn_prelim_search_begin (its status stays with its results)
    const int rc2 = take_failure(results);
    if (!rc && !rc2 && results->chunk_len > 0) { merge_chunk_lists(results->hsps, results->chunk_len, results->merge, results->diag); results->chunk_len = 0; }
    return rc ? rc : rc2;
    });
}

// the same search delivered the way BlastHSPStreamWrite wants it: one call per subject that has HSPs
int gbn_prelim_search_lists(GbnBatch *batch, GbnDb *db, GbnHspListFn sink, void *sink_arg, GbnDiagnostics *diag,
                            GbnInterruptFn interrupt, void *progress) {
    return gbn::guard(__func__, [&]() -> int {
    if (!sink) { set_error("gbn_prelim_search_lists: no sink"); return GBN_ERR_ARG; }
    GbnResults *res = nullptr;
    int rc = gbn_results_new(&res);
    if (rc) return rc;
    rc = gbn_prelim_search(batch, db, res, diag, 0, interrupt, progress);
    if (rc == GBN_OK) rc = gbn_results_emit_lists(res, sink, sink_arg);
    gbn_results_free(res);
    return rc;
    });
}
// bench / tests: a sink that counts -- arg = long long[2]: lists, HSPs (what a caller's BlastHSPStreamWrite would be handed)
int gbn_debug_counting_sink(void *arg, int32_t oid, const GbnHSP *hsps, int32_t n) {
    return gbn::guard(__func__, [&]() -> int {
    (void)oid; (void)hsps;
    if (arg) { long long *c = static_cast<long long *>(arg); c[0] += 1; c[1] += n; }
    return 0;
    });
}
// the HSPs of finished results (gbn_prelim_search, or gbn_prelim_search_begin + _end) as one call per subject that
// has any, ascending OID: what a pipelined caller hands to BlastHSPStreamWrite while its next search is running
int gbn_results_emit_lists(const GbnResults *res, GbnHspListFn sink, void *sink_arg) {
    return gbn::guard(__func__, [&]() -> int {
    if (!res || !sink) { set_error("gbn_results_emit_lists: bad argument"); return GBN_ERR_ARG; }
    const GbnHSP *h = res->hsps.data();
    const int64_t n = (int64_t)res->hsps.size();
    for (int64_t i = 0; i < n; ) {
        int64_t j = i;
        while (j < n && h[j].oid == h[i].oid) j++;
        if (sink(sink_arg, h[i].oid, h + i, (int32_t)(j - i))) { set_error("gbn_results_emit_lists: the sink failed"); return GBN_ERR_ARG; }
        i = j;
    }
    return GBN_OK;
    });
}

int gbn_prelim_search_begin(GbnBatch *batch, GbnDb *db, GbnResults *results, GbnDiagnostics *diag,
                            GbnInterruptFn interrupt, void *progress) {
    return gbn::guard(__func__, [&]() -> int {
    const int rc = search_enter(batch, db, results);
    return rc ? rc : run_search(batch, db, results, diag, 0, interrupt, progress, 1);
    });
}

int gbn_prelim_search_end(GbnResults *results) {
    return gbn::guard(__func__, [&]() -> int {
    // the engine that is filling these results; without results: whatever the calling thread's engine has in flight
    if (results && results->engine) enter(static_cast<Engine *>(results->engine));
    else if (results) return GBN_OK;                        // never searched: nothing in flight for them
    else if (!tl_eng) {
        Engine *e = nullptr;
        { std::lock_guard<std::mutex> lk(g_eng_mu); const int d = tl_sel >= 0 ? tl_sel : g_default_dev; if (d >= 0) e = g_eng[d]; }
        if (!e) return GBN_OK;
        enter(e);
    }
    if (!E.ready) return GBN_OK;
    // a stage that belongs to other results stays in flight: these results were completed when that
    // stage was queued (one in flight at most).  The engine is locked for the look at the stage in flight only: a caller's
    // other thread may be inside gbn_prelim_search_begin of the next pass meanwhile.
    int rc;
    {
        std::lock_guard<std::mutex> lk(E.mu);
        if (!results) { (void)wait_pending(); return GBN_OK; }
        if (E.has_pending && E.pending_res == results) (void)wait_pending_gpu();
    }
    wait_tail(results->host_tail);                          // (its last host replay may still run; those of later searches are not waited for)
    rc = take_failure(results);
    if (!rc && results->chunk_len > 0) { merge_chunk_lists(results->hsps, results->chunk_len, results->merge, results->diag); results->chunk_len = 0; }
    return rc;
    });
}

int gbn_scan_only(GbnBatch *batch, GbnDb *db, int repeats, GbnDiagnostics *diag) {
    return gbn::guard(__func__, [&]() -> int {
    if (!batch || !db || repeats <= 0) { set_error("bad argument"); return GBN_ERR_ARG; }
    int rc = params_ready(batch, db);                       // (enters the engine both live on)
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E.mu);
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long cnt[2] = {0, 0};
    for (int r = 0; r < repeats; r++) {
        int64_t bases = 0;
        rc = run_scan(*batch, *db, 0, db->num_seqs, diag, cnt, &bases);
        if (rc == kSkewedRange) { set_error("gbn_scan_only: lookup words pile up in a few bins of this shard (use gbn_prelim_search, which splits the range)"); return GBN_ERR_UNSUPPORTED; }
        if (rc) return rc;
        if (diag) diag->subject_bases_scanned += bases;
    }
    if (diag) {
        diag->seeds = (int64_t)cnt[0]; diag->lookup_hits = (int64_t)cnt[1];
        diag->total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return GBN_OK;
    });
}

// tests (GBN_GUARD=1): guard zones of every pool block intact?  Aborts on the first violation, returns 0 otherwise.
long gbn_debug_check_guards(void) { return pool_check_guards(); }

}  // extern "C"
