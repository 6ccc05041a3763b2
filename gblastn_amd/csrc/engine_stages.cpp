// engine_stages.cpp -- what follows the scan of a subject range: the seeds put into scan order per (subject, diagonal slot),
// diagonal filter + ungapped extension, gapped extension of every initial hit, D2H, the host replay of the acceptance rules
// (hsp_host.cpp) -- inline, or on the second stream + a host thread underneath the caller's next scan.
#include "engine.hpp"

namespace gbn {
// one range of subjects [s0, s1) through the whole pipeline
static int gapped_stage(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res, GbnDiagnostics *diag,
                        int keep_stages, int slot, unsigned long long nih, hipStream_t st, bool detach_host = false);

// the seeds of the last scan in one array (E.seeds), for the consumers that do not read scan_slice_kernel's segments
int compact_seeds(hipStream_t st) {
    if (!E.seg_valid) return GBN_OK;
    HIPCHK(launch_seed_compact(E.slice_seg, E.seg_counts, E.seg_firsts, E.seg_n, E.seg_len, E.seeds, E.seed_cap, st, E.seg_keys ? &E.seg_key_params : nullptr));
    E.seg_valid = false; E.seg_keys = false;
    return GBN_OK;
}

// The layout of a seed's composite key for the range [s0, s1) of this shard and this batch (gbn_composite_key): what seed_stage
// sorts by, and -- since round 6 -- what the ordered slice scan writes instead of seeds when it may (run_scan_impl).
void seed_key_layout(const GbnBatch &b, const GbnDb &db, int32_t s0, int32_t s1, GbnKeyParams &K, int *ck_bits, int *scan_bits, int *group_key_bits)
{
    std::memset(&K, 0, sizeof(K));
    K.q_descending = (b.lut.type == GBN_LUT_MB); K.container_hash = b.container; K.diag_len = b.diag_len;
    // key widths: the radix sorts stop at the top bit a key can have
    auto bits_for = [](uint64_t below) { int k = 1; while (k < 63 && ((uint64_t)1 << k) < below) k++; return k; };
    int32_t max_len = 1;
    for (int32_t s = 0; s < db.num_seqs; s++) max_len = std::max(max_len, db.len[s]);
    K.q_bits = std::min(32, bits_for((uint64_t)b.qlen + 1));
    K.group_bits = b.container ? 9 : bits_for((uint64_t)std::max(b.diag_len, 2));
    if (scan_bits) *scan_bits = std::min(64, K.q_bits + bits_for((uint64_t)max_len + 1));
    if (group_key_bits) *group_key_bits = std::min(64, K.group_bits + bits_for((uint64_t)db.num_seqs + 1));
    K.s_bits = bits_for((uint64_t)max_len + 1); K.qh_bits = std::max(0, K.q_bits - K.group_bits);
    K.subj_base = s0;
    if (ck_bits) *ck_bits = K.group_bits + bits_for((uint64_t)(s1 - s0) + 1) + K.s_bits;        // (the query key's high bits travel in the value)
}
bool seed_key_layout_fits(const GbnBatch &b, const GbnKeyParams &K, int ck_bits)
{
    return ck_bits <= 64 && K.group_bits < 32 && K.qh_bits <= 24 && b.lut.word - b.lut.lut < 256;
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
    trace_mark("seed stage: starts");
    if (phase != 2 && (rc = grow_key_buffers(KS, (size_t)n))) return rc;
    GbnKeyParams K; int ck_bits = 0, scan_bits = 0, group_key_bits = 0;
    seed_key_layout(b, db, s0, s1, K, &ck_bits, &scan_bits, &group_key_bits);
    K.seeds = seeds; K.n = n; K.key_scan = KS.key_a; K.idx = KS.idx_a;
    // Many seeds (blastn shapes): ONE sort of a composite key, the seed itself travels in the key (seed_ckeys_kernel) --
    // when subject | slot | s_scan | query key fit 64 bits; else, and for the few seeds of megablast shapes, two
    // stable sorts of (rank, index) pairs
    const int64_t compact_min = (int64_t)gbn::switch_value("GBN_DIAG_COMPACT_MIN", (long long)GBN_DIAG_COMPACT_MIN);
    const bool ck_on = gbn::switch_value("GBN_SEED_CKEYS", 1) != 0;
    const bool composite = ck_on && KS.ext_rec && n >= compact_min && seed_key_layout_fits(b, K, ck_bits);
    bool segmented = seeds == E.seeds && E.seg_valid;                 // (an asynchronous stage works on a copy of its own)
    bool from_segments = segmented && composite && !keep_stages;      // seed_ckeys_kernel reads the segments as they are
    // ... and when the value (ext_left and the query key's high bits) fits underneath the key too, it travels in the
    // key's low bits: a sort of keys only, on the bits above the value (GBN_SEED_CKEYS=2: always pairs)
    const bool ck_pack = gbn::switch_value("GBN_SEED_CKEYS", 1) != 2;
    const int v_bits = 8 + K.qh_bits;
    const bool packed = composite && ck_pack && ck_bits + v_bits <= 64;
    // Seeds that come in scan order, subject by subject (scan_fold_ordered_kernel's segments): a stable partition of every
    // subject's seeds by slot is all that is left, and seed_order.hip does it as a counting sort that builds the keys
    // on its way -- no key kernel, no radix passes (GBN_SEED_ORDER=0: keys + the library sort, as rounds 2-3)
    const int nsubj = s1 - s0;
    const bool order = from_segments && E.seg_ordered && packed && nsubj <= GBN_ORDER_MAX_SUBJ && (1 << K.group_bits) <= GBN_ORDER_MAX_SLOTS &&
                       n >= (int64_t)nsubj * 64 && n < ((int64_t)1 << 31) && gbn::switch_value("GBN_SEED_ORDER", 1) != 0 &&
                       seed_order_scratch_words(n, nsubj, K.group_bits) * sizeof(uint32_t) <= KS.sort_tmp_bytes;
    // (round 6: a scan that wrote its seeds as composite keys for the seed-order kernels, and a range that does not go through them after
    // all -- too few seeds, a switch --: the keys are decoded into E.seeds and the range goes on as if the scan had left seeds)
    if (phase != 2 && segmented && E.seg_keys && !order) { if ((rc = compact_seeds(st))) return rc; segmented = false; from_segments = false; }
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
        HIPCHK(sort_pairs_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, KS.idx_a, KS.idx_b, n, scan_bits, st)); if (diag) GBN_DIAG_LOCKED(diag->library_sorts++);
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
    if (composite) {
        K.key_scan = KS.key_a; K.idx = KS.idx_a; K.v_bits = packed ? v_bits : 0;
        if (from_segments) { K.seg = E.slice_seg; K.seg_count = E.seg_counts; K.nseg = E.seg_n; K.seg_cap = E.seg_len; K.seg_first = E.seg_firsts; }
        if (phase != 2 && order) {
            K.key_scan = KS.key_b; K.seg_keys = (segmented && E.seg_keys) ? 1 : 0;
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
            if (diag) GBN_DIAG_LOCKED(diag->library_sorts++);
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
            HIPCHK(sort_pairs_u64(KS.sort_tmp, tb, KS.key_a, KS.key_b, KS.idx_b, KS.idx_a, n, group_key_bits, st)); if (diag) GBN_DIAG_LOCKED(diag->library_sorts++);
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
        trace_mark("seed stage: kernels queued");
        HIPCHK(hipStreamSynchronize(st));
        trace_mark("seed stage: kernels done");
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
int search_range(GbnBatch &b, GbnDb &db, int32_t s0, int32_t s1, GbnResults &res,
                 GbnDiagnostics *diag, int keep_stages, int overlap)
{
    const int slot = E.slot;
    unsigned long long cnt[3] = {0, 0, 0};
    int64_t bases = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_stage = now();
    trace_mark("range: scan starts");
    E.want_key_seeds = !keep_stages;                        // (the seed list of keep_stages wants seeds)
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
    if (diag) { diag->scan_stage_ms += ms_since(t_stage); diag->ranges++; }
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
        trace_mark("previous asynchronous stage finished");
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
            gbn::CpuScope cpu(gbn::GBN_CPU_STAGE);
            int r = GBN_OK;
            unsigned long long nih2 = 0;
            if (hipSetDevice(dev) != hipSuccess) { E.pending_err = "hipSetDevice failed in the extension thread"; return GBN_ERR_HIP; }
            if (hipStreamWaitEvent(E.stream2, E.ev_seed, 0) != hipSuccess) { E.pending_err = "hipStreamWaitEvent failed"; return GBN_ERR_HIP; }
            r = seed_stage(*bp, *dbp, *rp, diag, 0, slot, E.seeds_async, n, E.counters + 12, E.stream2, &nih2, s0, s1, ksi);
            if (!r && nih2) r = gapped_stage(*bp, *dbp, s0, s1, *rp, diag, 0, slot, nih2, E.stream2, true);
            if (r) E.pending_err = gbn_last_error();      // the error text is per thread
            return r;
        });
        E.has_pending = true; E.pending_res = rp; E.pending_ks = ksi; E.pending_batch = bp;
        E.pending_res_pub.store(rp, std::memory_order_release); E.pending_batch_pub.store(bp, std::memory_order_release);
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
    // (tried in round 6 and not kept: the gapped stage of this range waiting, on its stream, for the slice scan of the next one -- the two
    // exclude each other on a CU, 152 KB against fifteen times 10 KB of LDS, and start at the same moment -- so that the lane DP runs next
    // to the seed stage instead: C3 27.1-27.2 against 26.9-27.0 ms per pass on one box)
    E.pending = std::async(std::launch::async, [=]() -> int {
        tl_eng = eng;
        gbn::CpuScope cpu(gbn::GBN_CPU_STAGE);
        if (hipSetDevice(dev) != hipSuccess) { E.pending_err = "hipSetDevice failed in the gapped-stage thread"; return GBN_ERR_HIP; }
        const int r = gapped_stage(*bp, *dbp, s0, s1, *rp, diag, 0, slot, nih, E.stream2, true);
        if (r) E.pending_err = gbn_last_error();      // the error text is per thread
        return r;
    });
    E.has_pending = true; E.pending_res = rp; E.pending_batch = bp;
    E.pending_res_pub.store(rp, std::memory_order_release); E.pending_batch_pub.store(bp, std::memory_order_release);
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
            trace_mark("hitbuf: pinned pair freed (too short)");
            (void)hipHostFree(old.hih); (void)hipHostFree(old.hg);
        }
    }
    HitBuf b; b.cap = std::max<size_t>(n + n / 4, 1 << 16);
    trace_mark("hitbuf: pinned pair allocated");
    if (hipHostMalloc((void **)&b.hih, b.cap * sizeof(GbnDevInitHit)) != hipSuccess ||
        hipHostMalloc((void **)&b.hg, b.cap * sizeof(GbnDevGapped)) != hipSuccess) {
        if (b.hih) (void)hipHostFree(b.hih);
        set_error("out of pinned host memory (gapped stage)"); return GBN_ERR_NOMEM;
    }
    out = b;
    return GBN_OK;
}
static void hitbuf_put(const HitBuf &b) { if (b.hih) { std::lock_guard<std::mutex> lk(E.hitbuf_mu); E.hitbuf_idle.push_back(b); } }
int stage_get(size_t bytes, void **p, size_t *cap) {
    {
        std::lock_guard<std::mutex> lk(E.hitbuf_mu);
        for (size_t i = 0; i < E.stage_idle.size(); i++)
            if (E.stage_idle[i].second >= bytes) { *p = E.stage_idle[i].first; *cap = E.stage_idle[i].second; E.stage_idle.erase(E.stage_idle.begin() + (long)i); return GBN_OK; }
        if (E.stage_idle.size() >= 4) { trace_mark("stage: pinned upload buffer freed"); (void)hipHostFree(E.stage_idle.back().first); E.stage_idle.pop_back(); }     // too short, all of them: one goes
    }
    const size_t want = bytes + bytes / 4 + 4096;
    trace_mark("stage: pinned upload buffer allocated");
    if (hipHostMalloc(p, want) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; set_error("out of pinned host memory (query upload)"); return GBN_ERR_NOMEM; }
    *cap = want;
    return GBN_OK;
}
void stage_put(void *p, size_t cap) { if (p) { std::lock_guard<std::mutex> lk(E.hitbuf_mu); E.stage_idle.emplace_back(p, cap); } }
void hitbuf_drain() {
    { std::lock_guard<std::mutex> lk(E.hitbuf_mu); for (auto &s : E.stage_idle) (void)hipHostFree(s.first); E.stage_idle.clear(); }
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
            gbn::CpuScope cpu(gbn::GBN_CPU_REPLAY);
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
    const unsigned hw = gbn::host_cpus();
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
        trace_mark("gather: pinned stage grows");
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
