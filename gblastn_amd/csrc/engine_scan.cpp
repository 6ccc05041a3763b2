// engine_scan.cpp -- one scan of a subject range (TNaScanSubjectFunction + the mini-extension for every subject of the range:
// scan_bin -> probe_bin -> probe_rare, or the direct-probe / slice kernels) and the record cache that keeps what the binning
// kernel writes.
#include "engine.hpp"

namespace gbn {
void fill_scan_params(GbnScanParams &P, const GbnBatch &b, const GbnDb &db, const TileSet &ts) {
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
int choose_bins(const GbnBatch &b) {
    int64_t nb = b.lut.ncells >> GBN_BIN_CBITS(b.lut.lut);
    if (nb < 2 || nb > GBN_BIN_MAXNB) nb = 1;
    if (gbn::switch_value("GBN_SCAN_BINS", 0) == 1) nb = 1;
    return (int)nb;
}

// slices scan_slice_kernel would cut this batch's presence bits into (0: another kernel scans for this batch)
int scan_slices(const GbnBatch &b) {
    const bool on = gbn::switch_value("GBN_SCAN_SLICE", 1) != 0;
    if (!on || !b.dev || choose_bins(b) == 1) return 0;     // (tables of one bin: the direct kernel as before; GBN_SCAN_BINS=1 forces it)
    GbnScanParams P; std::memset(&P, 0, sizeof(P));
    P.mode = b.dev->mode; P.step = b.lut.step; P.lut = b.lut.lut; P.word = b.lut.word; P.ncells = b.lut.ncells;
    return scan_slice_count(P);
}

// ---- record cache (Engine::rec_sets) ----
void recset_free_runs(RecordSet &r) {
    dev_free(r.run_fp); dev_free(r.run_pos); dev_free(r.run_start);
    r.run_n = r.run_cells = 0; r.runs = false; r.runs_failed = false; r.hits = 0;
}
void recset_free(RecordSet &r) {
    dev_free(r.bin_rec); dev_free(r.bin_tcur); dev_free(r.bin_count); dev_free(r.bin_caps); r.row_records = 0; r.bin_caps_nb = 0;
    r.bin_rec_cap = r.bin_tcur_cap = r.bin_count_cap = 0; r.complete = false; r.queued = false;
    recset_free_runs(r);
}
// the buffers of `r` at least this long (freed and allocated anew when one is too short: whatever they held is gone)
int recset_size(RecordSet &r, size_t need_u64, size_t need_tcur, size_t need_count) {
    int rc;
    if (need_u64 > r.bin_rec_cap) { dev_free(r.bin_rec); r.bin_rec_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_rec, need_u64))) return rc; r.bin_rec_cap = need_u64; }
    if (need_tcur > r.bin_tcur_cap) { dev_free(r.bin_tcur); r.bin_tcur_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_tcur, need_tcur))) return rc; r.bin_tcur_cap = need_tcur; }
    if (need_count > r.bin_count_cap) { dev_free(r.bin_count); r.bin_count_cap = 0; r.complete = false; if ((rc = dev_alloc(r.bin_count, need_count))) return rc; r.bin_count_cap = need_count; }
    return GBN_OK;
}
// bytes the cache may hold: gbn_record_cache_set_limit, else GBN_RECORD_CACHE_MB, else a quarter of the device's memory
long long rec_limit_bytes() {
    if (E.rec_limit >= 0) return E.rec_limit;
    if (gbn::switch_is_set("GBN_RECORD_CACHE_MB")) return std::max(0ll, gbn::switch_value("GBN_RECORD_CACHE_MB", 0)) << 20;
    static thread_local long long dflt[kMaxDevices];        // (per device; the query costs a driver call)
    long long &d = dflt[E.device >= 0 && E.device < kMaxDevices ? E.device : 0];
    // a quarter of the device's memory, and never more than half of what is FREE when the cache is first used: resident
    // shards count (the limit used to ignore them)
    if (d == 0) { size_t fr = 0, tot = 0; d = hipMemGetInfo(&fr, &tot) == hipSuccess && tot ? (long long)std::min(tot / 4, std::max<size_t>(fr / 2, (size_t)1 << 30)) : (64ll << 30); }
    return d;
}
// an allocation failed: every set no pass is using goes back to the driver (the pool's idle blocks went before the caller's
// second attempt: pool_alloc).  Called with the engine's lock held (dev_alloc checks).
size_t rec_evict_for_memory() {
    size_t freed = 0;
    for (size_t i = E.rec_sets.size(); i-- > 0; ) {
        if (E.rec_sets[i]->in_use) continue;
        freed += E.rec_sets[i]->bytes();
        rec_drop(i, true);
    }
    if (freed) pool_drain(E.device);
    return freed;
}
size_t rec_held_bytes() { size_t n = 0; for (const RecordSet *r : E.rec_sets) n += r->bytes(); return n; }
void rec_drop(size_t i, bool evicted) {
    if (E.rec_sets[i]->queued) (void)hipStreamSynchronize(E.stream);       // (a binning kernel queued by gbn_db_prepare_records may still write it)
    recset_free(*E.rec_sets[i]); delete E.rec_sets[i]; E.rec_sets.erase(E.rec_sets.begin() + (long)i);
    if (evicted) E.rec_evictions++;
}
// buffers change hands (what `dst` had is freed); neither side holds records afterwards
void recset_move(RecordSet &dst, RecordSet &src) {
    if (src.queued || dst.queued) (void)hipStreamSynchronize(E.stream);
    recset_free(dst);
    dst.bin_rec = src.bin_rec; dst.bin_rec_cap = src.bin_rec_cap; dst.bin_tcur = src.bin_tcur; dst.bin_tcur_cap = src.bin_tcur_cap;
    dst.bin_count = src.bin_count; dst.bin_count_cap = src.bin_count_cap; dst.complete = false; dst.queued = false; src.queued = false;
    src.bin_rec = nullptr; src.bin_tcur = nullptr; src.bin_count = nullptr; src.bin_rec_cap = src.bin_tcur_cap = src.bin_count_cap = 0; src.complete = false;
    recset_free_runs(src);                      // (a sorted set's arrays are not handed on: the newcomer's are sized by ITS records)
    dev_free(src.bin_caps); src.row_records = 0; src.bin_caps_nb = 0;
}
// Sets go until `need` more bytes fit under `limit` (keep: the set the pass is using).  Which: a pass over (shard, range) of
// a table shape is one step of a SWEEP -- every query batch visits the ranges / block views of its database in the same
// order, again and again -- and under such a cyclic pattern "least recently used first" evicts exactly the set that is needed
// next (no hits at all once the sets of a sweep exceed the limit).  So, as buffer managers do for sequential scans: among the
// sets of the pass's own database shape (same lut width, stride and stream geometry: its sweep) the MOST recently used one goes
// -- the sets from the start of the sweep stay and are hit again by the next batch --; only when there is none, the least
// recently used of the others.  If `into` is given and empty, the last victim's buffers move there instead of being freed.
void rec_make_room(size_t need, long long limit, const RecordSet *keep, const RecKey *sweep, RecordSet *into) {
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
void rec_purge(const void *db, bool to_scratch) {
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
int64_t bin_positions(const GbnDb &db, const TileSet &ts, int32_t s0, int32_t s1, int lut, int step) {      // (ts: the tiles of this very range, table width and stride -- get_tiles)
    if (ts.scan_positions >= 0) return ts.scan_positions;
    int64_t npos = 0;
    for (int32_t s = s0; s < s1; s++) if (db.len[s] >= lut) npos += (db.len[s] - lut) / step + 1;
    return ts.scan_positions = npos;
}
int bin_layout(int nb, int64_t ntiles, int64_t npos, double slack, BinLayout &L) {
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
RecordSet *rec_find(const RecKey &key) {
    for (RecordSet *c : E.rec_sets) if (c->key.same_shape(key) && (c->runs || ((c->complete || c->queued) && (c->bin_caps || c->key.subcap >= key.subcap)))) return c;
    return nullptr;
}
// a set to bin `key` into, its buffers sized: a cached one (room made for it) or -- larger than the whole cache -- the passes'
// own scratch set
int rec_acquire(const RecKey &key, const BinLayout &L, long long limit, RecordSet **out) {
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
    int rc = recset_size(*rs, L.need_u64, L.nstream * L.nseq, L.count_words());
    if (rc == GBN_ERR_NOMEM && rs != &E.scratch) {      // the device is full: everything else the cache holds goes, once
        rec_make_room((size_t)limit, limit, rs);
        rc = recset_size(*rs, L.need_u64, L.nstream * L.nseq, L.count_words());
    }
    if (rc) { if (rs != &E.scratch) { for (size_t i = 0; i < E.rec_sets.size(); i++) if (E.rec_sets[i] == rs) { rec_drop(i, false); break; } } return rc; }
    recset_free_runs(*rs);
    rs->key = key; rs->complete = false; rs->queued = false;
    *out = rs;
    return GBN_OK;
}

// ---- the sorted form of a cached set (scan_runs.hip) ----
// cache hits a set serves in stream form before it is sorted (GBN_RUNS_AFTER, default 1: the second hit sorts; 0: the first).
// The build moves ~28 bytes per record (27 ms for the C2 shard) and a pass over runs saves ~2 ms: a set that serves two batches
// (the 10,000-query config) is not worth sorting, a stream of batches is.  GBN_REC_RUNS=0: never (the stream-form path, A/B).
static bool runs_enabled() { return gbn::switch_value("GBN_REC_RUNS", 1) != 0; }
static int runs_after() { return (int)std::max(0ll, gbn::switch_value("GBN_RUNS_AFTER", 1)); }
// workgroups of probe_runs_kernel (512 threads, 20 KB of LDS, 58 registers: four fit a CU) = queue segments of the rare kernel.  THREE per
// CU: four take all 32 wave slots of a CU, and the extension stage of the batch before -- seed sort, diagonal kernel, greedy kernel, on
// the second stream -- then runs only once the probe kernel has ended, next to the rare kernel, and overruns into the next pass (the
// search thread waited 0.9 ms per C4 batch for it).  Same box, two runs each: cached C2 step 3.18 / 3.02 ms with four, 2.97 / 2.98 with
// three, 3.20 / 3.19 with two (the kernel itself 1.43 -> 1.44 -> 1.55 ms); C4 3.40 / 3.33, 3.29 / 3.35, 3.18 / 3.17
// (profiles/r06_runs_wgs_ab.txt)
static int runs_grid() { return (int)std::min<long long>(2048, (long long)E.num_cu * std::max(1ll, std::min(8ll, gbn::switch_value("GBN_RUNS_WGS", 3)))); }

// B: the streams of the complete set `rs`, npos: its records.  On success the set is in sorted form and its streams are back
// in the pool; a build that finds no room, or whose count does not come out, leaves the set as it is (and is not tried again).
static int build_runs(RecordSet &rs, const GbnBinParams &B, int64_t npos)
{
    const int64_t ncells = B.S.ncells;
    if (npos <= 0 || npos >= ((int64_t)1 << 32) - 64 || ncells < GBN_RUNS_ITEM_CELLS || (ncells % GBN_RUNS_ITEM_CELLS) != 0) { rs.runs_failed = true; return GBN_OK; }
    // Worth it only when the runs are long: a pass over runs reads 8 bytes of table per CELL whatever the shard holds and the
    // sorted set carries 4 bytes per cell, so a small shard's set would grow and its passes slow down (a 20 Mb shard at lut 11:
    // 3 MB of streams against 17 MB of run starts).  GBN_REC_RUNS=2: sort regardless (tests).
    if (gbn::switch_value("GBN_REC_RUNS", 1) != 2 && (npos < 8 * ncells || (size_t)npos * 6 + ((size_t)ncells + 1) * 4 >= rs.bytes())) { rs.runs_failed = true; return GBN_OK; }
    GbnRunsBuild R; std::memset(&R, 0, sizeof(R));
    R.B = B;
    R.sbits = runs_choose_sbits(npos, B.nb, B.cbits);
    R.wgroup = runs_choose_wgroup(npos, B.nb, B.nwriters);
    if ((1 << (B.cbits - R.sbits)) > GBN_RUNS_UNIT_CELLS_MAX) { rs.runs_failed = true; return GBN_OK; }
    uint32_t *count = nullptr, *cursor = nullptr, *mid_key = nullptr, *mid_pos = nullptr, *pos = nullptr; uint16_t *fp = nullptr; uint8_t *tmp = nullptr;
    size_t tmp_bytes = 0;
    (void)lut_scan(nullptr, tmp_bytes, nullptr, nullptr, ncells + 1, E.stream);
    auto give_up = [&]() {
        dev_free(count); dev_free(cursor); dev_free(mid_key); dev_free(mid_pos); dev_free(pos); dev_free(fp); dev_free(tmp);
        rs.runs_failed = true;
        return GBN_OK;
    };
    if (dev_alloc(count, (size_t)ncells + 1) || dev_alloc(cursor, (size_t)B.nb << R.sbits) || dev_alloc(mid_key, (size_t)npos) || dev_alloc(mid_pos, (size_t)npos) ||
        dev_alloc(pos, (size_t)npos) || dev_alloc(fp, (size_t)npos + 8) || dev_alloc(tmp, tmp_bytes)) return give_up();
    R.count = count; R.cursor = cursor; R.mid_key = mid_key; R.mid_pos = mid_pos; R.fp = fp; R.pos = pos;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipMemsetAsync(count, 0, ((size_t)ncells + 1) * 4, E.stream));
    HIPCHK(hipMemsetAsync(fp + (size_t)npos, 0, 16, E.stream));          // (the last 16-byte chunk of the fingerprints is read whole)
    HIPCHK(hipEventRecord(e0, E.stream));
    HIPCHK(launch_runs_build(R, tmp, tmp_bytes, E.stream));
    HIPCHK(hipEventRecord(e1, E.stream));
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, count + ncells, 4, hipMemcpyDeviceToHost, E.stream));
    HIPCHK(hipStreamSynchronize(E.stream));
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if ((int64_t)total != npos) {                               // (cannot happen with complete streams; never probe runs that are not all there)
        fprintf(stderr, "[gbn] sorted record set: %u records counted, %lld expected -- staying with the streams\n", total, (long long)npos);
        return give_up();
    }
    dev_free(cursor); dev_free(mid_key); dev_free(mid_pos); dev_free(tmp);
    rs.run_fp = fp; rs.run_pos = pos; rs.run_start = count; rs.run_n = (size_t)npos; rs.run_cells = (size_t)ncells; rs.runs = true;
    // the streams go back to the pool (the counts stay: the passes' overflow word lives behind them)
    dev_free(rs.bin_rec); dev_free(rs.bin_tcur); rs.bin_rec_cap = rs.bin_tcur_cap = 0; rs.complete = false;
    E.rec_runs_built++; E.rec_runs_build_ms = ms;
    trace_mark("scan: record set sorted");
    return GBN_OK;
}

// one scan of the subjects [s0, s1): fills E.seeds / cnt[0] seeds, cnt[1] raw hits;
// dispatches to the direct-probe kernel (small tables) or the partitioned pair
static int run_scan_impl(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
                         unsigned long long cnt[2], int64_t *bases_out, bool direct, bool *skewed);

// The partitioned scan sizes its streams for lookup words spread evenly over the bins (x1.25, x2.5).
// Subjects dominated by one repeat (satellite arrays, poly-A) put most positions of a range into a
// few bins; such a range goes through the direct-probe kernel instead, which has no streams.

int run_scan(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
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
    if (diag) diag->direct_ranges++;
    return run_scan_impl(b, db, s0, s1, diag, cnt, bases_out, true, &skewed);
}

static int run_scan_impl(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnDiagnostics *diag,
                         unsigned long long cnt[2], int64_t *bases_out, bool direct, bool *skewed)
{
    // tables as wide as the word (stride 1, every lookup hit a seed): the presence bits are sliced through the LDS
    // instead of the scan positions being written out by key range (scan_slice_kernel); GBN_SCAN_SLICE=0: off
    const bool sliced = !direct && scan_slices(b) > 0;
    E.seg_valid = false; E.seg_keys = false;
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
    const int64_t npos = nb > 1 ? bin_positions(db, ts, s0, s1, b.lut.lut, b.lut.step) : 0;
    double slack = 1.25;
    size_t rare_seg_hint = 0, rare_seg_used = 0, slice_seg_cap = 0; int slice_blocks = 0; bool slice_ordered = false;
    GbnBinParams last_B; std::memset(&last_B, 0, sizeof(last_B)); int last_grid2 = 0;
    const long long rec_limit = nb > 1 ? rec_limit_bytes() : 0;        // bytes the record cache may hold; 0: off
    if (rec_limit == 0 && nb > 1 && !E.rec_sets.empty()) rec_purge(nullptr, true);     // (switched off: what it held goes)
    RecordSet *rs = nullptr;                    // the records of this pass
    bool binned_here = false;                   // ... were written (completely) by this call
    bool repeat_seen = false;                   // cache off: the pass before this one had the same key
    int attempt = 0;                            // scans of this call so far (the second and later ones are rescans)
    // streams of bins that differ in size: {capacity, offset in a writer's row} per bin, made from the uncapped totals of an attempt that
    // overflowed (repeat-rich subjects: a few bins take most of the scan positions); exact_row: records per row
    std::vector<uint32_t> exact; size_t exact_row = 0; size_t last_nstream = 0;
    bool keys_written = false;                  // the slice scan wrote composite keys, not seeds (round 6)
    bool counted = false;                       // the cache's hit / miss of this call is counted (a range scanned again counts once)
    bool used_runs = false;                     // the pass went over sorted records
    struct NotInUse { ~NotInUse() { for (RecordSet *c : E.rec_sets) c->in_use = false; } } not_in_use_on_return;     // (in_use: a failing allocation may not evict the pass's own set)
    if (E.seed_copy_pending) { HIPCHK(hipStreamWaitEvent(E.stream, E.ev_seed, 0)); E.seed_copy_pending = false; }
    for (;;) {
        bool binned_ahead = false; int hit_pair = -1;
        if (!E.counters_zeroed) HIPCHK(hipMemsetAsync(E.counters, 0, 8 * sizeof(unsigned long long), E.stream));    // (a pass that binned ahead zeroed them behind its read-back)
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
                // Round 6: when the range's seeds will go through the seed-order kernels -- an ordered scan, the composite key and its value
                // fit 64 bits, few enough subjects and slots, no seed list asked for -- the scan writes the 8-byte key instead of the 16-byte
                // seed (what is only known afterwards -- enough seeds? -- decides in seed_stage; keys are decoded again if not).
                // GBN_SEED_KEYS=0: seeds as before (A/B).
                GbnKeyParams CK; int ck_bits = 0; keys_written = false;
                if (ordered && E.want_key_seeds && gbn::switch_value("GBN_SEED_KEYS", 1) != 0 && gbn::switch_value("GBN_SEED_CKEYS", 1) == 1 &&
                    gbn::switch_value("GBN_SEED_ORDER", 1) != 0) {
                    seed_key_layout(b, db, s0, s1, CK, &ck_bits);
                    CK.v_bits = 8 + CK.qh_bits;
                    if (seed_key_layout_fits(b, CK, ck_bits) && ck_bits + CK.v_bits <= 64 && s1 - s0 <= GBN_ORDER_MAX_SUBJ && (1 << CK.group_bits) <= GBN_ORDER_MAX_SLOTS) {
                        CK.seg_keys = 1; keys_written = true; E.seg_key_params = CK;
                    }
                }
                HIPCHK(launch_scan_slice(P, E.num_cu, E.slice_seg, (uint32_t)slice_seg_cap, E.seg_counts, E.counters + 2, E.stream, keys_written ? &CK : nullptr));
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
            last_nstream = nstream;
            // cache off: the passes' own set remembers the capacities an earlier pass over this key found (the counts are a function of
            // the shard, the range and the table's shape: no attempt is wasted again)
            const bool reuse_caps = exact.empty() && rec_limit == 0 && !(E.ahead.valid && E.ahead.key.same_shape(key)) && E.scratch.bin_caps && E.scratch.bin_caps_nb == nb && E.scratch.row_records && E.scratch.key.same_shape(key);
            if (!exact.empty() || reuse_caps) {
                const size_t row = reuse_caps ? E.scratch.row_records : exact_row;
                BL.subcap = 0; BL.need_u64 = (GBN_REC_WORDS(row * (size_t)nwriters) + 1) / 2;
                subcap = 0; key.subcap = 0;
            }
            bool hit = false, ahead_hit = false;
            Engine::BinAhead &AH = E.ahead;
            if (rec_limit > 0) {
                // ---- record cache: a complete set of this shape whose streams are at least as long as this attempt asks for
                if (AH.valid) { AH.valid = false; E.ahead_misses++; HIPCHK(hipStreamSynchronize(E.stream)); }     // (a kernel queued ahead writes the other scratch set, which may change hands below)
                // (a rare-path segment overflowed and the range is scanned again with a set larger than the cache: the records this call
                // binned into the passes' own scratch set are still there -- ADVICE r05: rec_find does not look at scratch)
                const bool scratch_again = binned_here && rs == &E.scratch && rs->complete && rs->key.same_shape(key);
                if (!scratch_again) rs = rec_find(key);
                if (scratch_again) { hit = true; subcap = rs->key.subcap; key.subcap = subcap; }
                else if (rs) { hit = true; subcap = rs->key.subcap; key.subcap = subcap; if (!counted) { E.rec_hits++; if (!rs->queued) rs->hits++; } }     // (a set queued by gbn_db_prepare_records: this pass is the one that would have binned)
                else {
                    if (!counted) E.rec_misses++;
                    // a set gbn_db_prepare_records queued for this shard that no pass has taken (the batch came out with another table
                    // shape than its unmasked lengths predicted -- ADVICE r05): its kernel is done by now or soon; its overflow word
                    // decides whether it is a complete set of ITS shape, and it stops being "queued"
                    for (RecordSet *c : E.rec_sets)
                        if (c->queued && c->key.db == key.db) {
                            uint32_t ov = 1;
                            const size_t ns = (size_t)c->key.nb * (size_t)c->key.nwriters;
                            HIPCHK(hipStreamSynchronize(E.stream));
                            if (c->bin_count && hipMemcpy(&ov, c->bin_count + ns, 4, hipMemcpyDeviceToHost) != hipSuccess) ov = 1;
                            c->queued = false; c->complete = ov == 0;
                        }
                    if ((rc = rec_acquire(key, BL, rec_limit, &rs))) return rc;
                }
                counted = true;
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
                    if ((rc = recset_size(*rs, BL.need_u64, nstream * nseq, BL.count_words()))) return rc;
                    rs->key = key; rs->complete = false;
                }
            }
            if (!hit && !ahead_hit) {
                // the layout this attempt bins into: capacities just computed, the set's own (cache off, same key), or uniform streams
                if (!exact.empty()) {
                    if (rs->bin_caps_nb != nb) { dev_free(rs->bin_caps); rs->bin_caps_nb = 0; if ((rc = dev_alloc(rs->bin_caps, (size_t)2 * nb))) return rc; rs->bin_caps_nb = nb; }
                    HIPCHK(hipMemcpyAsync(rs->bin_caps, exact.data(), (size_t)2 * nb * 4, hipMemcpyHostToDevice, E.stream));
                    HIPCHK(hipStreamSynchronize(E.stream));                 // (`exact` is pageable; 4 KB, once per skewed set)
                    rs->row_records = exact_row;
                } else if (!(reuse_caps && rs == &E.scratch)) { dev_free(rs->bin_caps); rs->bin_caps_nb = 0; rs->row_records = 0; }
                HIPCHK(hipMemsetAsync(rs->bin_count + nstream, 0, 16, E.stream));     // (records that exist already: the flag of THAT launch is read back below)
            }
            rs->stamp = ++E.rec_clock;
            for (RecordSet *c : E.rec_sets) c->in_use = (c == rs);
            GbnBinParams B; std::memset(&B, 0, sizeof(B));
            B.S = P; B.nb = nb; B.cbits = GBN_BIN_CBITS(b.lut.lut); B.nwriters = nwriters; dbg_nwriters = nwriters; dbg_subcap = (uint32_t)subcap;
            B.cellt = b.dev->cellt; B.sidet = b.dev->sidet; B.side_start = b.dev->side_start; B.rfl = std::min(4, b.dev->fl); B.rfrbits = std::min(7, 2 * b.dev->fr);
            B.rec = reinterpret_cast<uint32_t *>(rs->bin_rec); B.tcur = rs->bin_tcur; B.nseq = (uint32_t)nseq; B.gcount = rs->bin_count; B.subcap = (uint32_t)subcap;
            B.overflow = rs->bin_count + nstream; B.gtotal = rs->bin_count + nstream + 4;
            B.bincap = rs->bin_caps; B.rowsize = rs->row_records;
            B.dbg = (int)gbn::switch_value("GBN_DBG", 0);
            B.rare_parts = (hit && !binned_here) ? 5 : 0;      // (a pass over cached records: gbn_dev.h)
            B.work = gbn::switch_value("GBN_PROBE_DYN", 1) != 0 ? reinterpret_cast<uint32_t *>(E.counters + 4) : nullptr;      // (counters [4 .. 7]: zeroed with the scan's own, above)
            // a cached set in stream form that has served its passes is sorted now, in front of this pass
            if (rec_limit > 0 && hit && rs->complete && !rs->queued && !rs->runs && !rs->runs_failed && runs_enabled() && rs->hits > runs_after() &&
                (rc = build_runs(*rs, B, npos))) return rc;
            used_runs = rec_limit > 0 && hit && rs->runs;
            if (used_runs) {
                B.run_fp = rs->run_fp; B.run_pos = rs->run_pos; B.run_start = rs->run_start; B.rec = nullptr; B.tcur = nullptr;
                B.run_item_cells = B.S.ncells <= ((int64_t)1 << 20) ? 64 : GBN_RUNS_ITEM_CELLS;
                B.work = reinterpret_cast<uint32_t *>(E.counters + 4); B.rare_parts = (int)std::max(1ll, gbn::switch_value("GBN_RUNS_RARE_PARTS", 6));      // (rare workgroups per queue segment over sorted records: with 768 segments -- three probe workgroups per CU -- one each took 1.3-1.4 ms, four to six 0.95-1.05; step 3.05-3.21 -> 2.98-3.06, one-query pass 0.77 -> 0.68: profiles/r06_runs_wgs_ab.txt)
                E.rec_runs_passes++;
            }
            int grid2 = used_runs ? runs_grid() : std::max(8, E.num_cu & ~7);   // stream form: one 1024-thread workgroup per CU; group = blockIdx & 7
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
            if (used_runs) HIPCHK(launch_probe_runs(B, grid2, E.stream, E.evk, b.dev->ready));
            else HIPCHK(launch_scan_bin_parts(B, grid2, E.stream, E.evk, ((hit || ahead_hit) ? 2 : 3) | 4 | (ahead_hit ? 8 : 0), b.dev->ready));
            binned = true; binned_ahead = ahead_hit;
            HIPCHK(hipMemcpyAsync(&E.scan_back->overflow, B.overflow, 4, hipMemcpyDeviceToHost, E.stream));
            HIPCHK(hipMemcpyAsync(E.scan_back->rare_counts, E.rare_counts, (size_t)grid2 * 4, hipMemcpyDeviceToHost, E.stream));
        }
        if (b.dev->ready_ctx) HIPCHK(hipStreamWaitEvent(E.stream, b.dev->ready_ctx, 0));       // the contexts' cut-offs went up last (upload_batch_contexts): the stages that read them are launched when the host has seen this scan end
        HIPCHK(hipMemcpyAsync(E.scan_back->cnt, E.counters, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, E.stream));     // seeds, raw hits, the fullest segment
        if (!E.ev_back) HIPCHK(hipEventCreate(&E.ev_back));
        HIPCHK(hipEventRecord(E.ev_back, E.stream));
        if (binned && rec_limit == 0 && E.want_ahead && repeat_seen && slack <= 1.25 && !rs->bin_caps) {     // (a skewed range's capacities stay with ONE set)
            // the next pass's binning kernel, into the other set (sized like this one)
            RecordSet &A = E.alt;
            const size_t nstream = (size_t)last_B.nb * (size_t)last_B.nwriters;
            if (recset_size(A, rs->bin_rec_cap, nstream * last_B.nseq, 2 * nstream + 4) == GBN_OK) {        // (no room for a second set: no binning ahead)
                GbnBinParams A2 = last_B;
                A2.rec = reinterpret_cast<uint32_t *>(A.bin_rec); A2.tcur = A.bin_tcur; A2.gcount = A.bin_count; A2.overflow = A.bin_count + nstream; A2.gtotal = A.bin_count + nstream + 4;
                A2.rareq = nullptr;                         // (the binning kernel queues nothing; rare_counts: where a GBN_BIN_TIMING build leaves its clocks)
                A.key = rs->key; A.complete = false;
                Engine::BinAhead &AH = E.ahead;
                if (!AH.ev[0][0]) for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&AH.ev[i >> 1][i & 1]));
                AH.pair = hit_pair >= 0 ? (hit_pair ^ 1) : (AH.pair ^ 1);
                HIPCHK(hipMemsetAsync(A.bin_count + nstream, 0, 16, E.stream));
                HIPCHK(hipMemsetAsync(E.counters, 0, 8 * sizeof(unsigned long long), E.stream)); E.counters_zeroed = true;      // (read back above; the next scan's)
                HIPCHK(hipEventRecord(AH.ev[AH.pair][0], E.stream));
                HIPCHK(launch_scan_bin_parts(A2, last_grid2, E.stream, nullptr, 1, nullptr));
                HIPCHK(hipEventRecord(AH.ev[AH.pair][1], E.stream));
                AH.valid = true; AH.key = rs->key;
            }
        }
        trace_mark("scan: kernels queued");
        // (tried in round 6 and not kept: sleeping through 60 % of what the last wait of this kind took instead of spinning in the runtime for
        // all of it -- the search thread's CPU 2.8 -> 1.3 ms per C4 batch and C4 4.08 -> 4.01 ms under the box's CPU quota, but the cached
        // C2 pass 3.08 -> 3.11-3.25 and C3 26.7 -> 27.0: the wake-up costs the GPU-bound passes more than the CPU-bound one gains;
        // profiles/r06_wait_ab.txt)
        HIPCHK(hipEventSynchronize(E.ev_back));
        trace_mark("scan: kernels done");
        cnt[0] = E.scan_back->cnt[0]; cnt[1] = E.scan_back->cnt[1];
        const unsigned long long seg_max = sliced ? E.scan_back->seg_max : 0;
        if (binned && used_runs) overflow = 0;
        else if (binned) { overflow = E.scan_back->overflow; rs->complete = overflow == 0; rs->queued = false; binned_here = rs->complete; }
        finish_build(b.dev);                                // (the scan has waited for the builder's event)
        if (diag) {
            float ms = 0, ahead_ms = 0;
            if (binned) (void)hipEventElapsedTime(&ms, E.evk[binned_ahead ? 1 : 0], E.evk[3]);     // (the launcher's own events bracket the stage)
            else (void)hipEventElapsedTime(&ms, E.ev0, E.ev1);
            if (binned_ahead && hit_pair >= 0) (void)hipEventElapsedTime(&ahead_ms, E.ahead.ev[hit_pair][0], E.ahead.ev[hit_pair][1]);    // this pass's binning kernel ran ahead
            diag->scan_kernel_ms += ms + ahead_ms; diag->scan_launches++; if (attempt++) diag->scan_rescans++;
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
            const int grid2 = last_grid2;
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
        if (overflow) {
            // The records are incomplete: some stream was too short.  The binning kernel counted on past the streams' ends, so the
            // attempt says exactly how much room every stream needs (GbnBinParams::gtotal): the range is binned once more with streams
            // of per-bin capacities -- the most any writer has of that bin -- and fits.  (Rounds 1-5: twice the room, then the range
            // halved until the repeat-rich subjects sat in ranges for the direct-probe kernel: 256 ranges, 1,278 scan launches and
            // 1.4 s for a C2 step over a shard with 8 % low-complexity sequence, profiles/r06_skew.txt.)  GBN_EXACT_FIT=0: that.
            if (exact.empty() && rs && last_nstream && gbn::switch_value("GBN_EXACT_FIT", 1) != 0) {
                const int nwr = last_B.nwriters;
                std::vector<uint32_t> tot(last_nstream);
                HIPCHK(hipMemcpy(tot.data(), rs->bin_count + last_nstream + 4, last_nstream * 4, hipMemcpyDeviceToHost));
                exact.assign((size_t)2 * nb, 0u); size_t row = 0;
                for (int bb = 0; bb < nb; bb++) {
                    uint32_t mx = 0;
                    for (int w = 0; w < nwr; w++) mx = std::max(mx, tot[(size_t)bb * nwr + w]);
                    const size_t cap = ((size_t)mx + 511) & ~(size_t)511;
                    exact[(size_t)2 * bb] = (uint32_t)std::max<size_t>(cap, 512); exact[(size_t)2 * bb + 1] = (uint32_t)row;
                    row += exact[(size_t)2 * bb];
                    if (row > 0x7ffffff0u) { exact.clear(); break; }
                }
                exact_row = row;
                if (!exact.empty()) continue;
            }
            if (!exact.empty() || gbn::switch_value("GBN_EXACT_FIT", 1) != 0) { *skewed = true; return GBN_OK; }     // (cannot fit: the direct kernel)
            slack *= 2;
            if (slack > 3.0) { *skewed = true; return GBN_OK; }
            continue;
        }
        if (sliced) {       // the seeds sit in the workgroups' segments; E.seeds only has to be long enough for compact_seeds
            if (cnt[0] > E.seed_cap && (rc = grow_seed_buffers((size_t)cnt[0] + (cnt[0] >> 3)))) return rc;
            E.seg_valid = cnt[0] > 0; E.seg_n = slice_blocks; E.seg_len = (uint32_t)slice_seg_cap; E.seg_ordered = slice_ordered; E.seg_keys = keys_written && E.seg_valid;
            break;
        }
        if (cnt[0] <= E.seed_cap) break;
        if ((rc = grow_seed_buffers((size_t)cnt[0] + (cnt[0] >> 3)))) return rc;
    }
    return GBN_OK;
}

}  // namespace gbn
