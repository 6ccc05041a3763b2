// engine_abi.cpp -- the C ABI of include/gblastn_amd.h over the engine (engine.hpp): process and device entry points, the
// caches of resident shards / blocks / views, shards and query batches, results, the search entry points
// (Blast_gpu_RunPreliminarySearchWithInterrupt's work behind plain-C structures, GB/gpu_blastn_pre_search_engine.cpp:1466-1563),
// the record cache's controls.  Every status-returning entry point runs behind the exception firewall (gbn_guard.hpp).
#include "engine.hpp"

using namespace gbn;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *gbn_last_error(void) { return gbn::last_error_text().c_str(); }


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
    gbn::EngLock lk(E);
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
    gbn::EngLock lk(E);
    for (RecordSet *r : E.rec_sets) { if (r->queued) (void)hipStreamSynchronize(E.stream); r->complete = false; r->queued = false; recset_free_runs(*r); }
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
    gbn::EngLock lk(E);
    long long sorted = 0, sorted_bytes = 0;
    for (const RecordSet *r : E.rec_sets) if (r->runs) { sorted++; sorted_bytes += (long long)r->run_bytes(); }
    const long long v[14] = {rec_limit_bytes(), (long long)rec_held_bytes(), (long long)E.rec_sets.size(), E.rec_hits, E.rec_misses, E.rec_evictions, E.rec_bypass, E.ahead_hits, E.rec_prepared,
                             sorted, sorted_bytes, E.rec_runs_built, E.rec_runs_passes, (long long)(E.rec_runs_build_ms * 1000.0)};
    for (int i = 0; i < n && i < 14; i++) out[i] = v[i];
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
    gbn::EngLock lk(E);
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

// a shard's slab: when the device is full the record cache's sets (and the pool's idle blocks) go first, once (a
// block uploaded on a cold cache after earlier groups were searched used to fail while gigabytes of evictable records were held)
static hipError_t db_malloc(uint8_t **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    size_t freed = 0;
    { gbn::EngLock lk(E); freed = rec_evict_for_memory(); }
    pool_drain(E.device);
    (void)freed;
    e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
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
            if (db_malloc(&p, (size_t)nbytes) != hipSuccess) { delete db; set_error("hipMalloc(db) failed"); return GBN_ERR_NOMEM; }
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
        if (db_malloc(&p, (size_t)db->nbytes) != hipSuccess) { delete db; set_error("hipMalloc(db) failed"); return GBN_ERR_NOMEM; }
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
// The shard's bytes produced piece by piece by the caller's callback, through pinned staging buffers, uploads overlapping
// the fills: what a database that comes from files wants (gbn_blastdb_load_shard) -- the one-slab form above costs a
// zero-filled host copy of the whole shard (3 s of page faults for 12.5 GB) and one blocking copy from pageable memory.
int gbn_db_new_streamed(GbnDb **out, int64_t nbytes, int32_t num_seqs, const int64_t *byte_off, const int32_t *len,
                        int32_t first_oid, GbnFillFn fill, void *ctx, int threads) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || !fill || nbytes < 144 || num_seqs < 0 || (num_seqs > 0 && (!byte_off || !len))) { set_error("bad argument"); return GBN_ERR_ARG; }
    int rc = enter_current();
    if (rc) return rc;
    constexpr int64_t kPiece = 16ll << 20;
    int64_t total = 0;
    for (int32_t i = 0; i < num_seqs; i++) {
        const int64_t nb = ((int64_t)len[i] + 3) / 4;
        if (byte_off[i] < 16 || (byte_off[i] & 15) || byte_off[i] + nb + 128 > nbytes || (i > 0 && byte_off[i] < byte_off[i - 1] + ((int64_t)len[i - 1] + 3) / 4)) {
            set_error("subject offsets must be ascending, 16-byte aligned, >= 16, and leave 128 pad bytes"); return GBN_ERR_ARG;
        }
        if (len[i] > g_max_dbseq_len || nb > kPiece) { set_error("a sequence longer than MAX_DBSEQ_LEN: gbn_db_new holds it as chunk copies"); return GBN_ERR_UNSUPPORTED; }
        total += len[i];
    }
    // pieces: runs of consecutive sequences whose image on the device is at most kPiece bytes
    struct Piece { int32_t first, count; int64_t base, bytes; };
    std::vector<Piece> pieces;
    for (int32_t i = 0; i < num_seqs; ) {
        Piece pc; pc.first = i; pc.base = byte_off[i];
        int32_t j = i;
        while (j < num_seqs && byte_off[j] + ((int64_t)len[j] + 3) / 4 - pc.base <= kPiece) j++;
        pc.count = j - i; pc.bytes = byte_off[j - 1] + ((int64_t)len[j - 1] + 3) / 4 - pc.base;
        pieces.push_back(pc); i = j;
    }
    uint8_t *p = nullptr;
    if (db_malloc(&p, (size_t)nbytes) != hipSuccess) { set_error("hipMalloc(db) failed"); return GBN_ERR_NOMEM; }
    hipEvent_t zeroed = nullptr;
    bool ok = hipEventCreateWithFlags(&zeroed, hipEventDisableTiming) == hipSuccess &&
              hipMemsetAsync(p, 0, (size_t)nbytes, E.stream) == hipSuccess && hipEventRecord(zeroed, E.stream) == hipSuccess;
    const int nthreads = (int)std::max<size_t>(1, std::min<size_t>({pieces.size(), (size_t)(threads > 0 ? threads : (int)gbn::switch_value("GBN_UPLOAD_THREADS", std::max(2u, std::min(8u, gbn::host_cpus() / 2)))), (size_t)64}));
    std::atomic<size_t> next{0}; std::atomic<int> status{ok ? GBN_OK : GBN_ERR_HIP};
    Engine *eng = tl_eng;
    uint8_t *pinned = nullptr;              // two pieces per worker, ONE allocation (pinning memory is the slow part: 32 workers' 64 allocations cost more than they gained)
    if (ok && hipHostMalloc((void **)&pinned, (size_t)kPiece * 2 * (size_t)nthreads) != hipSuccess) { (void)hipGetLastError(); ok = false; status = GBN_ERR_NOMEM; }
    std::atomic<int> wid{0};
    auto worker = [&]() {
        enter(eng);
        const int me = wid.fetch_add(1);
        hipStream_t st = nullptr; uint8_t *buf[2] = {pinned + (size_t)kPiece * 2 * (size_t)me, pinned + (size_t)kPiece * (2 * (size_t)me + 1)};
        hipEvent_t done[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
        bool good = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipStreamWaitEvent(st, zeroed, 0) == hipSuccess;
        for (int k = 0; k < 2 && good; k++) good = hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; good && status.load() == GBN_OK; k ^= 1) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size()) break;
            const Piece &pc = pieces[i];
            if (busy[k]) { good = hipEventSynchronize(done[k]) == hipSuccess; busy[k] = false; if (!good) break; }
            std::memset(buf[k], 0, (size_t)pc.bytes);
            const int frc = fill(ctx, pc.first, pc.count, buf[k], pc.base);
            if (frc) { int want = GBN_OK; status.compare_exchange_strong(want, frc); break; }
            good = hipMemcpyAsync(p + pc.base, buf[k], (size_t)pc.bytes, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(done[k], st) == hipSuccess;
            busy[k] = good;
        }
        if (st) good = (hipStreamSynchronize(st) == hipSuccess) && good;
        if (!good) { int want = GBN_OK; status.compare_exchange_strong(want, GBN_ERR_HIP); }
        for (int k = 0; k < 2; k++) if (done[k]) (void)hipEventDestroy(done[k]);
        if (st) (void)hipStreamDestroy(st);
    };
    if (ok) {
        std::vector<std::thread> ts;
        for (int t = 1; t < nthreads; t++) ts.emplace_back(worker);
        worker();
        for (auto &t : ts) t.join();
        enter(eng);
    }
    if (zeroed) { (void)hipEventSynchronize(zeroed); (void)hipEventDestroy(zeroed); }
    if (pinned) (void)hipHostFree(pinned);
    if (status.load() != GBN_OK) {
        (void)hipFree(p);
        if (status.load() == GBN_ERR_HIP) set_error("uploading the shard's pieces failed");
        return status.load();
    }
    g_db_bytes_uploaded += (long long)nbytes;
    GbnDb *db = new GbnDb();
    db->engine = tl_eng; db->first_oid = first_oid; db->real_seqs = num_seqs; db->chunk_len = g_max_dbseq_len;
    db->num_seqs = num_seqs; db->nbytes = nbytes; db->total_bases = total;
    db->byte_off.assign(byte_off, byte_off + num_seqs); db->len.assign(len, len + num_seqs);
    db->d_packed = p; db->owns = true;
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
        gbn::EngLock lk(E);
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

int gbn_synth_skew(void *dev_ptr, int64_t first_off, int64_t stride, int64_t nb, int32_t num, int64_t first_oid, uint64_t seed, void *stream) {
    return gbn::guard(__func__, [&]() -> int {
    int rc = enter_current();
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : E.stream;
    HIPCHK(launch_synth_skew(dev_ptr, first_off, stride, nb, num, first_oid, seed, st));
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
    gbn::CpuScope cpu(gbn::GBN_CPU_SETUP);
    std::vector<QueryMask> masks((size_t)nmask);
    for (int32_t i = 0; i < nmask; i++) masks[(size_t)i] = QueryMask{mask_query[i], mask_from[i], mask_to[i]};
    std::unique_ptr<GbnBatch, void (*)(GbnBatch *)> b(new GbnBatch(), gbn_batch_free);      // (freed if the set-up throws)
    // with a device the lookup tables are built there (upload_batch); a host-only set-up fills them here
    // (the device part starts in the middle of the host part: the query goes up and the lookup structures are queued on the
    // builder's stream as soon as the table's kind is known, while this thread computes the Karlin-Altschul parameters)
    bool stale = false;
    GbnBatch *bp = b.get();
    int rc = build_batch(*b, *opt, nq, seqs, lens, masks, /* host_tables = */ upload == 0,
                         upload ? std::function<int()>([bp]() { return upload_batch_tables(*bp); }) : std::function<int()>(), &stale);
    if (rc == GBN_OK && upload && stale) {              // a context dropped out after the tables were queued: once more, from the final stretches
        if (bp->dev) { finish_build(bp->dev); free_device_batch(bp->dev); bp->dev = nullptr; }
        rc = upload_batch_tables(*bp);
    }
    if (rc == GBN_OK && upload) rc = upload_batch_contexts(*b);
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
    if (b->dev->ready_ctx) HIPCHK(hipEventSynchronize(b->dev->ready_ctx));
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
        if (E.pending_batch_pub.load(std::memory_order_acquire) == b) { gbn::EngLock lk(E); if (E.has_pending && E.pending_batch == b) (void)wait_pending_gpu(); }
        wait_tail(b->host_tail);                            // (a queued host replay reads the batch's options and contexts; the engine is not locked meanwhile)
    }
    free_device_batch(b->dev);
    gbn::qbuf_give(std::move(b->qbuf));
    delete b;
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
        if (E.pending_res_pub.load(std::memory_order_acquire) == r) { gbn::EngLock lk(E); if (E.has_pending && E.pending_res == r) (void)wait_pending_gpu(); }
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
    for (const GbnDb::RangePlan &rp : db->range_plans)
        if (rp.step == step && rp.range_bytes == range_bytes && rp.tile_limit == tile_limit) { out = rp.ranges; return; }
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
    if (db->range_plans.size() >= 8) db->range_plans.erase(db->range_plans.begin());
    db->range_plans.push_back(GbnDb::RangePlan{step, range_bytes, tile_limit, out});
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
    // (a search is running on this device: its pass bins, or has found the records -- a set-up thread of a pipelined caller is
    // not to wait here for the length of a scan)
    gbn::EngLock lk(E, std::try_to_lock);
    if (!lk.owns_lock()) return GBN_OK;
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
        if (bin_layout((int)nb, tsp->ntiles, bin_positions(*db, *tsp, s0, s1, lut, step), 1.25, BL)) continue;
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
        B.overflow = rs->bin_count + BL.nstream; B.gtotal = rs->bin_count + BL.nstream + 4;
        dev_free(rs->bin_caps); rs->bin_caps_nb = 0; rs->row_records = 0;      // (uniform streams; a pass that finds them too short bins again with per-bin capacities)
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
    gbn::CpuScope cpu(gbn::GBN_CPU_SEARCH);
    gbn::EngLock lk(E);                   // (the caller has entered the engine: search_enter)
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
    gbn::EngLock lk(E);
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
    gbn::CpuScope cpu(gbn::GBN_CPU_END_COLLECT);
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
    if (!results || E.pending_res_pub.load(std::memory_order_acquire) == results) {     // (a stage of other results, or none: no need for the lock -- the caller's other thread may hold it for the length of a scan)
        gbn::EngLock lk(E);
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
    gbn::EngLock lk(E);
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long cnt[2] = {0, 0};
    for (int r = 0; r < repeats; r++) {
        int64_t bases = 0;
        E.want_key_seeds = false;                           // (a scan alone: its seeds stay seeds)
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

extern "C" int32_t gbn_host_cpus(void) { return (int32_t)gbn::host_cpus(); }
