// engine.cpp -- device residency of the preliminary search (the BLAST_PreliminarySearchEngine analogue,
// GB/gpu_blastn_pre_search_engine.cpp:1125-1464): the engines of the process (one per GPU), the device memory pool, a query
// batch's structures on the device (host set-up: batch.cpp; lookup structures built on the device: lutbuild.hip, own stream),
// tile tables of a shard, scratch buffers, the bookkeeping of the stage in flight.  The scan of a subject range and the
// record cache: engine_scan.cpp; the stages behind the scan: engine_stages.cpp; the C ABI: engine_abi.cpp.
#include "engine.hpp"

namespace gbn {
static thread_local std::string g_err;
const std::string &last_error_text() { return g_err; }
void set_error(const std::string &m) { g_err = m; }

// Stages of one search run on several threads (the inline seed stage, the stage in flight on the second stream, the
// detached host replays) and all add to the caller's GbnDiagnostics: every such update holds this lock.
std::mutex g_diag_mu;

// the switches (gbn_dev.h): read from the environment at every use
long long switch_value(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
bool switch_is_set(const char *name) { return getenv(name) != nullptr; }

// GBN_TRACE=1: wall-clock marks of the host-side pipeline on stderr (ms since the first mark)
void trace_mark(const char *what) {
    static const bool on = getenv("GBN_TRACE") && atoi(getenv("GBN_TRACE")) != 0;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    static const bool with_thread = atoi(getenv("GBN_TRACE")) > 1;       // GBN_TRACE=2: the calling thread behind the mark
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (with_thread) fprintf(stderr, "[gbn trace] %9.3f ms  %s  [thread %03u]\n", ms, what, (unsigned)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000u));
    else fprintf(stderr, "[gbn trace] %9.3f ms  %s\n", ms, what);
}

// One engine per device, created by gbn_init / gbn_use_device (or by the first call that needs one) and alive
// until gbn_release (engine.hpp: Engine).
Engine *g_eng[kMaxDevices];
std::mutex g_eng_mu;
int g_default_dev = -1;
thread_local Engine *tl_eng = nullptr;
thread_local Engine *tl_mu_owner = nullptr;
thread_local int tl_sel = -1;

// the calling thread's engine for calls that create batches / shards: its chosen device, else the process default,
// else the thread's current HIP device (first use without gbn_init)
int enter_current() {
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
    if (bytes >= ((size_t)1 << 20)) trace_mark(("pool: hipMalloc of " + std::to_string(bytes >> 20) + " MiB").c_str());
    if (!pool_guard()) return hipMalloc(p, bytes);
    void *raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes + 2 * kGuard);
    if (e != hipSuccess) return e;
    if ((e = guard_fill(raw, bytes)) != hipSuccess) { (void)hipFree(raw); return e; }
    *p = (char *)raw + kGuard;
    return hipSuccess;
}
void raw_free(void *p, size_t bytes) {
    if (bytes >= ((size_t)1 << 20)) trace_mark(("pool: hipFree of " + std::to_string(bytes >> 20) + " MiB").c_str());
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
    bool known = false;
    std::vector<std::pair<void *, size_t>> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto it = g_pool.size_of.find(p);
        if (it != g_pool.size_of.end()) {
            bytes = it->second.first; dev = it->second.second; g_pool.size_of.erase(it);
            if (pool_guard()) guard_check(p, bytes, "release");
            // the block stays; when that takes the idle blocks over the cap, the LARGEST ones go (round 6: the record streams of a set that
            // has been sorted by cell come back as one 21 GB block, which then sat at the cap while every batch's few hundred MB of tables,
            // a dozen blocks, were handed to hipFree -- which waits for the device -- batch after batch: 3 ms of a C4 step)
            g_pool.idle[dev].emplace(bytes, p); g_pool.held[dev] += bytes;
            while (g_pool.held[dev] > pool_cap() && !g_pool.idle[dev].empty()) {
                auto big = std::prev(g_pool.idle[dev].end());
                drop.emplace_back(big->second, big->first); g_pool.held[dev] -= big->first; g_pool.idle[dev].erase(big);
            }
            known = true;
        }
    }
    if (!known) { raw_free_on(p, bytes, dev); return; }
    for (auto &q : drop) raw_free_on(q.first, q.second, dev);
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
// the batch's lookup structures are complete (waits for the builder if they are not): scratch back to the pool
void finish_build(DeviceBatch *d) {
    if (!d) return;
    if (d->ready_ctx) { (void)hipEventSynchronize(d->ready_ctx); (void)hipEventDestroy(d->ready_ctx); d->ready_ctx = nullptr; }
    if (!d->ready) {                                    // (no event: built on the host, or the build failed half-way -- the stream decides)
        if (d->stage && d->eng) { (void)hipStreamSynchronize(d->eng->stream_build); stage_put(d->stage, d->stage_cap); d->stage = nullptr; }
        return;
    }
    (void)hipEventSynchronize(d->ready);
    for (void *p : d->build_scratch) pool_free(p);
    d->build_scratch.clear();
    if (d->stage) { stage_put(d->stage, d->stage_cap); d->stage = nullptr; }
    (void)hipEventDestroy(d->ready); d->ready = nullptr;
}

void free_device_batch(DeviceBatch *d) {
    if (!d) return;
    finish_build(d);
    dev_free(d->q8_base); dev_free(d->q2_base); dev_free(d->qinv_base); dev_free(d->q4_base); dev_free(d->pv); dev_free(d->cellw); dev_free(d->cellt); dev_free(d->sidet); dev_free(d->side_start); dev_free(d->cell_start); dev_free(d->ent); dev_free(d->pvx); dev_free(d->pstart);
    dev_free(d->ctx_block);                             // (ctx_off ... score_table point into it)
    delete d;
}

// fingerprint word of one query offset (layout in gbn_dev.h / kernels.hip fp_pass)
static inline uint32_t fingerprint(const uint8_t *q, int32_t off, int lut, bool force) {
    uint32_t l = 0, r = 0;
    for (int k = 1; k <= 8; k++) l |= (uint32_t)(q[off - k] & 3) << (2 * (k - 1));
    for (int j = 0; j < 7; j++) r |= (uint32_t)(q[off + lut + j] & 3) << (2 * (6 - j));
    return (l << 15) | (r << 1) | (force ? 1u : 0u);
}

// Everything a batch's kernels read per CONTEXT lives in one device block, filled by one copy (round 5; rounds 1-4: ten
// allocations and ten blocking copies, 0.4 ms of a 3.6 ms set-up): offsets, lengths, the cut-offs (x_dropoff, cut-off,
// reduced cut-off, and the three packed for one 16-byte read), the context hints per 64 query positions, matrix and score
// table.  Segments start at multiples of four words.
namespace {
struct CtxLayout { size_t n, h, off, len, xdrop, cutoff, reduced, pack, hint, blk, matrix, table, total; };
CtxLayout ctx_layout(const GbnBatch &b) {
    CtxLayout L; auto up = [](size_t v) { return (v + 3) & ~(size_t)3; };
    L.n = b.ctx.size(); L.h = ((size_t)b.qlen >> kCtxHintShift) + 2;
    size_t at = 0;
    L.off = at; at += up(L.n); L.len = at; at += up(L.n);
    L.xdrop = at; at += up(L.n); L.cutoff = at; at += up(L.n); L.reduced = at; at += up(L.n); L.pack = at; at += up(4 * L.n);
    L.hint = at; at += up(L.h); L.blk = at; at += up(2 * L.h); L.matrix = at; at += 256; L.table = at; at += 256;
    L.total = at;
    return L;
}
void fill_cutoffs(const GbnBatch &b, const CtxLayout &L, int32_t *w) {      // w: the block's words on the host
    for (size_t c = 0; c < L.n; c++) {
        const GbnContext &x = b.ctx[c];
        w[L.xdrop + c] = x.x_dropoff; w[L.cutoff + c] = x.cutoff_score; w[L.reduced + c] = x.reduced_cutoff;
        w[L.pack + 4 * c] = x.x_dropoff; w[L.pack + 4 * c + 1] = x.reduced_cutoff; w[L.pack + 4 * c + 2] = x.cutoff_score; w[L.pack + 4 * c + 3] = 0;
    }
}
// what of the block does not wait for the Karlin-Altschul parameters: offsets, lengths, position hints, score tables
void fill_ctx_fixed(const GbnBatch &b, const CtxLayout &L, int32_t *w) {
    for (size_t c = 0; c < L.n; c++) { w[L.off + c] = b.ctx[c].query_offset; w[L.len + c] = b.ctx[c].query_length; }
    {   // context of every kCtxHintShift-aligned query position (the kernels walk on from there: contexts are rarely shorter)
        size_t c = 0;
        for (size_t k = 0; k < L.h; k++) {
            const int64_t qpos = (int64_t)k << kCtxHintShift;
            while (c + 1 < L.n && w[L.off + c + 1] <= qpos) c++;
            w[L.hint + k] = (int32_t)c;
        }
        // ... and per block: that context and where the next one begins (GbnExtParams::ctx_blk)
        for (size_t k = 0; k < L.h; k++) {
            const size_t ci = (size_t)w[L.hint + k];
            const int64_t last = ((int64_t)k << kCtxHintShift) + ((int64_t)1 << kCtxHintShift) - 1;
            w[L.blk + 2 * k] = w[L.hint + k];
            w[L.blk + 2 * k + 1] = ci + 1 < L.n ? w[L.off + ci + 1] : INT32_MAX;
            if (ci + 2 < L.n && w[L.off + ci + 2] <= last) w[L.blk + 2 * k + 1] = INT32_MIN;
        }
    }
    std::memcpy(&w[L.matrix], &b.matrix[0][0], 256 * 4);
    std::memcpy(&w[L.table], b.score_table, 256 * 4);
}
}  // namespace
// the cut-offs of every context once more (a search without a database length recomputes them per subject)
int upload_ctx_cutoffs(GbnBatch &b) {
    DeviceBatch *d = b.dev;
    const CtxLayout L = ctx_layout(b);
    std::vector<int32_t> w(L.hint, 0);                  // (the block's words up to the end of the cut-offs)
    fill_cutoffs(b, L, w.data());
    if (d->ready_ctx) HIPCHK(hipEventSynchronize(d->ready_ctx));        // (the set-up's own copy of them is not to land on top of these)
    HIPCHK(hipMemcpy(d->ctx_block + L.xdrop, w.data() + L.xdrop, (L.hint - L.xdrop) * 4, hipMemcpyHostToDevice));
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
    uint32_t *count = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    uint32_t *keys_a = nullptr, *keys_b = nullptr; unsigned long long *ctr = nullptr; void *tmp = nullptr;
    auto cleanup = [&]() { dev_free(d_sl); dev_free(d_sr); dev_free(count);
                           dev_free(vals_a); dev_free(vals_b); dev_free(keys_a); dev_free(keys_b); dev_free(ctr); if (tmp) pool_free(tmp); tmp = nullptr; };
    // (on an error kernels may already be queued on the builder's stream: they finish before their scratch goes back to the pool)
#define LUTCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); (void)hipStreamSynchronize(st); cleanup(); return GBN_ERR_HIP; } } while (0)
#define LUTRC(x) do { if ((rc = (x))) { (void)hipStreamSynchronize(st); cleanup(); return rc; } } while (0)
    const size_t nc1 = (size_t)L.ncells + 1, qn = (size_t)std::max(b.qlen, 1);
    {   // the stretches: through the batch's staging buffer, on the builder's stream
        LUTRC(dev_alloc(d_sl, sl.size())); LUTRC(dev_alloc(d_sr, sr.size()));
        int32_t *hs = reinterpret_cast<int32_t *>(static_cast<char *>(d->stage) + d->stage_seg);
        if (!sl.empty()) {
            std::memcpy(hs, sl.data(), sl.size() * 4); std::memcpy(hs + sl.size(), sr.data(), sr.size() * 4);
            LUTCHK(hipMemcpyAsync(d_sl, hs, sl.size() * 4, hipMemcpyHostToDevice, st));
            LUTCHK(hipMemcpyAsync(d_sr, hs + sl.size(), sr.size() * 4, hipMemcpyHostToDevice, st));
        }
    }
    LUTRC(dev_alloc(keys_a, qn)); LUTRC(dev_alloc(keys_b, qn)); LUTRC(dev_alloc(vals_a, qn)); LUTRC(dev_alloc(vals_b, qn));
    LUTRC(dev_alloc(ctr, 2));
    LUTRC(dev_alloc(d->cell_start, nc1)); LUTRC(dev_alloc(d->cellw, (size_t)L.ncells)); LUTRC(dev_alloc(d->cellt, (size_t)L.ncells));
    LUTRC(dev_alloc(d->pv, (size_t)((L.ncells + 31) / 32)));
    LUTCHK(hipMemsetAsync(ctr, 0, 16, st));
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    LutBuild B; std::memset(&B, 0, sizeof(B));
    B.q8 = d->q8; B.qlen = b.qlen; B.seg_left = d_sl; B.seg_right = d_sr; B.nseg = (int32_t)sl.size();
    B.lut = L.lut; B.word = L.word; B.q_bits = std::min(31, bits_for((uint64_t)b.qlen + 1)); B.ncells = L.ncells;
    B.count = nullptr; B.keys_a = keys_a; B.keys_b = keys_b; B.vals_a = vals_a; B.vals_b = vals_b;
    B.cell_start = d->cell_start; B.cellw = d->cellw; B.cellt = d->cellt; B.pv = d->pv;
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
        const size_t side_len = (B.nbins <= GBN_BIN_MAXNB ? (size_t)B.nbins * GBN_BIN_SIDE : 0) + 1;      // a bin's side list has a fixed home (lut_cells_side)
        LUTRC(dev_alloc(d->sidet, side_len)); LUTRC(dev_alloc(d->side_start, (size_t)B.nbins + 1));
        LUTCHK(hipMemsetAsync(d->sidet, 0, side_len * 2, st));
        size_t b1 = 0, b2 = 0;
        const int key_bits = 2 * L.lut + 1;                         // (the cell; one bit more: "no word at this position" sorts last)
        LUTCHK(lut_sort(nullptr, b1, B, (int64_t)qn, key_bits, st));
        LUTCHK(pool_alloc(&tmp, std::max(b1, b2) + 256));
        size_t tb = std::max(b1, b2) + 256;
        uint32_t *n_valid = reinterpret_cast<uint32_t *>(ctr);           // (the counters' first word: nothing else uses it in this branch)
        LUTCHK(lut_sort(tmp, tb, B, (int64_t)qn, key_bits, st, n_valid));
        LUTCHK(lut_cell_starts(B, n_valid, st));
        B.ent = d->ent; B.sidet = d->sidet; B.side_start = d->side_start;
        LUTCHK(lut_entries(B, -1, st));
        LUTCHK(lut_cells_side(B, st));
        LUTCHK(lut_pv(B, st));
        LUTRC(build_rank_table(b, st, &d->build_scratch));
        LUTCHK(hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
        LUTCHK(hipEventRecord(d->ready, st));
        for (void *p : {(void *)d_sl, (void *)d_sr, (void *)vals_a, (void *)vals_b,
                        (void *)keys_a, (void *)keys_b, (void *)ctr, tmp}) d->build_scratch.push_back(p);
        return GBN_OK;
    }
    LUTRC(dev_alloc(count, nc1));
    B.count = count;
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
    {
        const size_t side_len = (B.nbins <= GBN_BIN_MAXNB ? (size_t)B.nbins * GBN_BIN_SIDE : 0) + 1;
        LUTRC(dev_alloc(d->sidet, side_len)); LUTRC(dev_alloc(d->side_start, (size_t)B.nbins + 1));
        LUTCHK(hipMemsetAsync(d->sidet, 0, side_len * 2, st));
    }
    B.sidet = d->sidet; B.side_start = d->side_start;
    LUTCHK(lut_cells_side(B, st));
    LUTCHK(lut_pv(B, st));
    LUTRC(build_rank_table(b, st, nullptr));
    LUTCHK(hipStreamSynchronize(st));
#undef LUTCHK
#undef LUTRC
    cleanup();
    return GBN_OK;
}

int upload_batch(GbnBatch &b) {
    const int rc = upload_batch_tables(b);
    return rc ? rc : upload_batch_contexts(b);
}
int upload_batch_tables(GbnBatch &b) {
    int rc = enter_current();
    if (rc) return rc;
    std::lock_guard<std::mutex> build_lock(E.build_mu);
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
    // the query goes up through a pinned staging buffer on the builder's stream (Engine::stage_idle: why)
    // (... and with it, behind the query: the indexed stretches and the per-context block -- every copy of a set-up is an
    // asynchronous one on the builder's stream.  Measured (tools/step_jitter.py with GBN_TRACE=1): every few batches of a pipelined
    // loop a BLOCKING hipMemcpy of a set-up -- first the 10 MB of the query, and once that was gone the 40 KB of stretches -- took
    // 14-24 ms instead of microseconds, the device's other queues stood still meanwhile, and the scan in flight lasted 20-45 ms
    // instead of 5: a null-stream copy from pageable memory next to running kernels.  With none left: no step above 8 ms.)
    size_t nseg = 0;
    for (auto &sg : L.segments) if (sg.second >= sg.first) nseg++;
    const size_t q_bytes = (b.qbuf.size() + 63) & ~(size_t)63, seg_bytes = ((2 * nseg * 4) + 63) & ~(size_t)63, ctx_bytes = ctx_layout(b).total * 4;
    if ((rc = dev_alloc(d->q8_base, b.qbuf.size())) || (rc = stage_get(q_bytes + seg_bytes + 2 * ctx_bytes + 64, &d->stage, &d->stage_cap))) return rc;
    d->stage_seg = q_bytes; d->stage_ctx = q_bytes + seg_bytes; d->stage_cut = d->stage_ctx + ctx_bytes;
    std::memcpy(d->stage, b.qbuf.data(), b.qbuf.size());
    {   // the per-context block: offsets, lengths, the position hints, the score tables now (the scan's kernels read them: they
        // are behind the batch's event with the tables); the cut-offs follow when they have been computed (upload_batch_contexts)
        const CtxLayout CL = ctx_layout(b);
        int32_t *w = reinterpret_cast<int32_t *>(static_cast<char *>(d->stage) + d->stage_ctx);
        std::memset(w, 0, CL.total * 4);
        fill_ctx_fixed(b, CL, w);
        if ((rc = dev_alloc(d->ctx_block, CL.total))) return rc;
        HIPCHK(hipMemcpyAsync(d->ctx_block, w, CL.total * 4, hipMemcpyHostToDevice, E.stream_build));
        d->ctx_off = d->ctx_block + CL.off; d->ctx_len = d->ctx_block + CL.len; d->ctx_xdrop = d->ctx_block + CL.xdrop;
        d->ctx_cutoff = d->ctx_block + CL.cutoff; d->ctx_reduced = d->ctx_block + CL.reduced; d->ctx_pack = d->ctx_block + CL.pack;
        d->ctx_hint = d->ctx_block + CL.hint; d->ctx_blk = d->ctx_block + CL.blk; d->matrix = d->ctx_block + CL.matrix; d->score_table = d->ctx_block + CL.table;
    }
    HIPCHK(hipMemcpyAsync(d->q8_base, d->stage, b.qbuf.size(), hipMemcpyHostToDevice, E.stream_build));
    d->q8 = d->q8_base + b.qpad;
    {   // packed copy for the greedy kernel's 32-bases-per-step match runs, made on the device from q8
        const int64_t pad = 256, n = (int64_t)b.qlen + 2 * pad;
        const size_t q2_bytes = (size_t)(n + 3) / 4 + 16, qi_bytes = (size_t)(n + 7) / 8 + 16;
        if ((rc = dev_alloc(d->q2_base, q2_bytes)) || (rc = dev_alloc(d->qinv_base, qi_bytes))) return rc;
        // (the arrays' tails, "matches nothing", are written by the kernels that fill them: a fill of 14 MB in three dispatches of
        // their own cost the builder's stream 0.3 ms next to a probe kernel, where every dispatch waits for room)
        HIPCHK(lut_pack_query(d->q8_base, (int64_t)b.qbuf.size(), (int64_t)b.qpad - pad, n, d->q2_base, d->qinv_base, E.stream_build));
        d->q2 = d->q2_base + pad / 4; d->qinv = d->qinv_base + pad / 8;
        d->q4_plane = ((((int64_t)b.qbuf.size() + 3) / 4 + 64) + 3) & ~(int64_t)3;      // (an 8- or 16-byte load may start at a plane's last byte; a multiple of 4: lut_q4_kernel stores dwords)
        if ((rc = dev_alloc(d->q4_base, (size_t)(4 * d->q4_plane)))) return rc;
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
    return GBN_OK;
}
int upload_batch_contexts(GbnBatch &b) {
    DeviceBatch *d = b.dev;
    if (!d || !d->eng || !d->stage || !d->ctx_block) { set_error("upload_batch_contexts: no device structures"); return GBN_ERR_ARG; }
    enter(d->eng);
    const CtxLayout L = ctx_layout(b);
    // the cut-offs (x_dropoff, cut-off, reduced cut-off per context): from a part of the staging buffer of their own -- the block's
    // first copy may not have been read yet -- behind whatever the builder's stream holds, with an event of their own
    int32_t *w = reinterpret_cast<int32_t *>(static_cast<char *>(d->stage) + d->stage_cut);
    std::memset(w + L.xdrop, 0, (L.hint - L.xdrop) * 4);
    fill_cutoffs(b, L, w);
    HIPCHK(hipMemcpyAsync(d->ctx_block + L.xdrop, w + L.xdrop, (L.hint - L.xdrop) * 4, hipMemcpyHostToDevice, E.stream_build));
    HIPCHK(hipEventCreateWithFlags(&d->ready_ctx, hipEventDisableTiming));
    HIPCHK(hipEventRecord(d->ready_ctx, E.stream_build));
    if (!d->ready) HIPCHK(hipEventSynchronize(d->ready_ctx));         // (a batch built synchronously, or on the host, is complete when its set-up returns)
    trace_mark("upload: done");
    return GBN_OK;
}

// ---------------------------------------------------------------------------

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
int get_tiles(GbnDb &db, int lut, int step, int tpos, int32_t s0, int32_t s1, const TileSet **out) {
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
void free_tile_cache(GbnDb &db) {
    if (!db.tile_cache) return;
    TileCache *tc = static_cast<TileCache *>(db.tile_cache);
    for (auto &kv : *tc) dev_free(kv.second.d_tiles);
    delete tc; db.tile_cache = nullptr;
}

int grow_seed_buffers(size_t want) {
    if (want <= E.seed_cap) return GBN_OK;
    dev_free(E.seeds);
    int rc = dev_alloc(E.seeds, want);
    if (rc) { E.seed_cap = 0; return rc; }
    E.seed_cap = want;
    return GBN_OK;
}
int grow_key_buffers(Engine::KeySet &KS, size_t n) {
    if (n <= KS.key_cap) {
        // (buffers made while the run-head threshold was above their size have no extension records; a range that is over the
        // threshold as it stands now -- GBN_DIAG_COMPACT_MIN changed in between -- gets them here, not the other path)
        if (!KS.ext_rec && n >= (size_t)gbn::switch_value("GBN_DIAG_COMPACT_MIN", (long long)GBN_DIAG_COMPACT_MIN))
            return dev_alloc(KS.ext_rec, KS.key_cap * 8);
        return GBN_OK;
    }
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

int grow_ihit_buffers(int slot, size_t n) {
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
void record_failure(const GbnResults *res, int rc, const std::string &what) {
    std::lock_guard<std::mutex> lk(E.failed_mu);
    if (!E.failed.count(res)) E.failed[res] = std::make_pair(rc, what);
}
// the device side of the stage in flight: its kernels and copies are done, its device buffers free again (the host
// replay of its extensions may still be running: wait_host)
int wait_pending_gpu() {
    if (!E.has_pending) return GBN_OK;
    int rc = E.pending.get();
    if (rc) record_failure(E.pending_res, rc, E.pending_err);
    E.has_pending = false; E.pending_ks = -1; E.pending_batch = nullptr;
    E.pending_res_pub.store(nullptr, std::memory_order_release); E.pending_batch_pub.store(nullptr, std::memory_order_release);
    return GBN_OK;
}
void wait_host() {
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(E.host_mu); f = E.host_tail; }
    if (f.valid()) f.wait();
}
// the host replays queued up to the one `tail` stands for (a batch's or a set of results' last: replays run in the order
// they were queued, so theirs are done when that one is) -- not the replays of LATER searches, which a pipelined caller
// has queued behind them (round 4: gbn_prelim_search_end / gbn_batch_free of pass k waited for the replays of pass k + 1,
// with the engine locked: 3.4 ms per blastn pass in which the next pass's scan could not be queued)
void wait_tail(std::shared_future<void> &slot) {
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(E.host_mu); f = slot; }
    if (f.valid()) f.wait();
    std::lock_guard<std::mutex> lk(E.host_mu);
    if (slot.valid() && slot.wait_for(std::chrono::seconds(0)) == std::future_status::ready) slot = std::shared_future<void>();
}
int wait_pending() {
    const int rc = wait_pending_gpu();
    wait_host();
    return rc;
}
int take_failure(const GbnResults *res) {
    std::lock_guard<std::mutex> lk(E.failed_mu);
    auto it = E.failed.find(res);
    if (it == E.failed.end()) return GBN_OK;
    const int rc = it->second.first;
    if (!it->second.second.empty()) set_error(it->second.second);
    E.failed.erase(it);
    return rc;
}

// ints of scratch one thread of the gapped kernels needs (GbnGapParams::scratch_per_thread) and the greedy row length
int64_t gap_scratch_ints(const GbnBatch &b, int32_t max_len, int32_t max_ctx, int32_t *row_len) {
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

// the engine of device `dev` (< 0: the calling thread's current HIP device), created and initialised on first use
int engine_init(int dev, Engine **out) {
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
    HIPCHK(pool_alloc((void **)&N.counters, 16 * sizeof(unsigned long long)));       // [0, 1] seeds, lookup hits of a scan; [2, 3] its segments / the inline seed stage; [4 .. 7] the probe kernel's item counters; [12, 13] the asynchronous seed stage
    N.device = dev; N.ready = true;
    if (g_default_dev < 0) g_default_dev = dev;
    *out = g_eng[dev];
    return GBN_OK;
}
}  // namespace gbn
