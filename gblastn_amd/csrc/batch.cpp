// batch.cpp -- per-query-batch host set-up: concatenated query with sentinels
// and reverse strands, Karlin-Altschul blocks, integer cut-offs, and the query
// word index in the cell order the reference's lookup tables yield.
//
// Mirrors what the reference hands INTO its GPU boundary (query, query_info,
// sbp, lookup_wrap): CORE/blast_setup.c:502-775, CORE/blast_parameters.c:160-470,
// :822-979, CORE/blast_nalookup.c:51-189,384-427,831-1041, CORE/blast_lookup.c:87-137.
#include <unistd.h>
#include "gbn_host.hpp"
#include <functional>
#include <thread>
#include <atomic>
#include <time.h>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

static const double kLn2 = 0.69314718055994530941723212145818;

// A few worker threads that live as long as the library, for the per-context loops of a batch's set-up (10,000 contexts in a 5 Mb
// megablast batch: strand copies, Karlin-Altschul parameters, effective lengths, cut-offs -- every context by itself, so the
// results do not depend on who computes which).  Rounds 1-4 started and joined a set of threads per loop: 0.3-0.5 ms each time.
namespace {
thread_local int tl_setup_pieces = 0;       // gbn_set_setup_threads
class SetupPool {
    std::vector<std::thread> workers_; std::mutex mu_; std::condition_variable cv_;
    std::deque<std::function<void()>> jobs_; bool stop_ = false;
    void loop() {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); }); if (jobs_.empty()) return; job = std::move(jobs_.front()); jobs_.pop_front(); }
            { gbn::CpuScope cpu(gbn::GBN_CPU_SETUP_POOL); job(); }
        }
    }
public:
    explicit SetupPool(int n) { for (int i = 0; i < n; i++) workers_.emplace_back([this] { loop(); }); }
    ~SetupPool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto &t : workers_) t.join(); }
    int size() const { return (int)workers_.size(); }
    // f(i0, i1) over [0, n) in pieces of at least `grain`; the caller takes a piece itself and returns when all are done
    void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &f) {
        const size_t most = tl_setup_pieces > 0 ? (size_t)tl_setup_pieces : (size_t)size() + 1;        // (gbn_set_setup_threads: the calling thread's share)
        const size_t pieces = std::max<size_t>(1, std::min<size_t>({(size_t)size() + 1, most, n / std::max<size_t>(grain, 1)}));
        if (pieces <= 1) { f(0, n); return; }
        std::mutex dmu; std::condition_variable dcv; size_t left = pieces - 1;
        std::exception_ptr failed;                      // what a piece threw (a worker must not take the process down: ADVICE r05): rethrown on the caller
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t p = 1; p < pieces; p++)
                jobs_.emplace_back([&, p] {
                    std::exception_ptr ex;
                    try { f(n * p / pieces, n * (p + 1) / pieces); } catch (...) { ex = std::current_exception(); }
                    std::lock_guard<std::mutex> dl(dmu);
                    if (ex && !failed) failed = ex;
                    if (--left == 0) dcv.notify_one();
                });
        }
        if (pieces - 1 >= (size_t)size()) cv_.notify_all();          // (one system call for all of them)
        else for (size_t p = 1; p < pieces; p++) cv_.notify_one();   // (a share of the pool: as many workers as there are pieces for them)
        std::exception_ptr mine;
        try { f(0, n / pieces); } catch (...) { mine = std::current_exception(); }
        { std::unique_lock<std::mutex> dl(dmu); dcv.wait(dl, [&] { return left == 0; }); }     // (the pieces refer to this frame: waited for whatever happened)
        if (mine) std::rethrow_exception(mine);
        if (failed) std::rethrow_exception(failed);
    }
};
// One pool per PROCESS: after fork() (Python's multiprocessing, default start method) the child has the parent's pool object
// and none of its threads -- jobs would wait for ever.  The child makes its own; the parent's object is left alone there (its
// thread handles belong to threads that do not exist in this process: neither joined nor destroyed).
}  // namespace

namespace gbn {
static std::atomic<long long> g_cpu_ns[GBN_CPU_N];
static bool cpu_account_on() { static const bool v = getenv("GBN_CPU_ACCOUNT") && atoi(getenv("GBN_CPU_ACCOUNT")) != 0; return v; }
static long long thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
CpuScope::CpuScope(int c) : cat(c), t0(cpu_account_on() ? thread_cpu_ns() : -1) {}
CpuScope::~CpuScope() { if (t0 >= 0) g_cpu_ns[cat] += thread_cpu_ns() - t0; }
}  // namespace gbn
extern "C" void gbn_set_setup_threads(int32_t n) { tl_setup_pieces = n > 0 ? n : 0; }
extern "C" int gbn_debug_cpu_account(double *ms, int n) {
    for (int i = 0; i < n && i < gbn::GBN_CPU_N; i++) ms[i] = (double)gbn::g_cpu_ns[i].load() / 1e6;
    return gbn::GBN_CPU_N;
}

namespace gbn {
unsigned host_cpus() {
    static const unsigned n = [] {
        unsigned v = std::max(1u, std::thread::hardware_concurrency());
        if (const char *e = getenv("GBN_HOST_CPUS")) { const int k = atoi(e); if (k > 0) return (unsigned)k; }
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) v = std::min(v, (unsigned)k); }
        auto read_line = [](const char *path, char *buf, size_t cap) {
            FILE *f = fopen(path, "r"); if (!f) return false;
            const bool ok = fgets(buf, (int)cap, f) != nullptr; fclose(f); return ok; };
        char buf[128];
        // cgroup v2: "max 100000" or "<quota> <period>"; v1: two files, quota -1 = none
        if (read_line("/sys/fs/cgroup/cpu.max", buf, sizeof(buf))) {
            long long q = 0, per = 0;
            if (sscanf(buf, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) v = std::min<unsigned>(v, (unsigned)std::max<long long>(1, (q + per - 1) / per));
        } else if (read_line("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", buf, sizeof(buf))) {
            const long long q = atoll(buf);
            if (q > 0 && read_line("/sys/fs/cgroup/cpu/cpu.cfs_period_us", buf, sizeof(buf))) {
                const long long per = atoll(buf);
                if (per > 0) v = std::min<unsigned>(v, (unsigned)std::max<long long>(1, (q + per - 1) / per));
            }
        }
        return v;
    }();
    return n;
}
}  // namespace gbn

namespace {
SetupPool &setup_pool() {
    static std::mutex mu; static SetupPool *pool = nullptr; static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!pool || owner != getpid()) {
        // (a 256-thread host without a quota: 31 workers + the caller, eight ranks of a node each have their own; under a quota the CPUs
        // granted: a set-up nothing else runs beside -- the shim's, 16 pieces 6.3 ms per batch, 8 pieces 7.2 -- uses them all; a
        // pipeline's set-up threads ask for half (gbn_set_setup_threads), the traceback threads want the rest: C4 4.1 against 4.2-4.4)
        const unsigned cpus = gbn::host_cpus(), hw = std::max(1u, std::thread::hardware_concurrency());
        pool = new SetupPool((int)std::max(2u, std::min(31u, cpus < hw ? cpus - 1u : hw / 8u)));
        owner = getpid();
    }
    return *pool;
}
}  // namespace

void gbn_default_options(GbnOptions *o, int megablast) {
    // API/blast_nucl_options.cpp:108-234
    std::memset(o, 0, sizeof(*o));
    o->word_size = megablast ? 28 : 11;
    o->reward = megablast ? 1 : 2;  o->penalty = megablast ? -2 : -3;
    o->gap_open = megablast ? 0 : 5; o->gap_extend = megablast ? 0 : 2;
    o->greedy = megablast ? 1 : 0;
    o->xdrop_ungap_bits = 20; o->gap_trigger_bits = 27.0;
    o->xdrop_gap_bits = megablast ? 25 : 30; o->xdrop_gap_final_bits = 100;
    o->evalue = 10.0; o->min_diag_separation = megablast ? 6 : 50;
    o->hitlist_size = 500; o->cutoff_score = 0; o->lut11_gblastn_rule = 1;
}

int GbnBatch::context_of(int32_t n) const {
    int32_t b = 0, e = (int32_t)ctx.size();
    while (b < e - 1) {
        int32_t m = (b + e) / 2;
        if (ctx[m].query_offset > n) e = m; else b = m;
    }
    return b;
}

void GbnBatch::set_effective_lengths(int64_t db_len, int32_t db_nseq) {
    if (db_len == 0) return;
    setup_pool().parallel_for(ctx.size(), 256, [&](size_t i0, size_t i1) {
    for (size_t ci = i0; ci < i1; ci++) {
        GbnContext &c = ctx[ci];
        int32_t adj = 0; int64_t eff = 0;
        if (c.is_valid && c.query_length > 0) {
            gbn::Karlin ku; ku.lambda = c.lambda_u; ku.K = c.K_u; ku.logK = c.logK_u; ku.H = c.H_u;
            double alpha = 0, beta = 0;
            gbn::alpha_beta(opt.reward, opt.penalty, opt.gap_open, opt.gap_extend, ku, true, alpha, beta);
            adj = gbn::length_adjustment(kbp_gap.K, kbp_gap.logK, alpha / kbp_gap.lambda, beta,
                                         c.query_length, db_len, db_nseq);
            int64_t eff_db = db_len - (int64_t)db_nseq * adj;
            if (eff_db <= 0) eff_db = 1;
            eff = eff_db * (c.query_length - adj);
        }
        c.eff_searchsp = eff; c.length_adjustment = adj;
    }
    });
}

void GbnBatch::update_cutoffs() {
    setup_pool().parallel_for(ctx.size(), 512, [&](size_t i0, size_t i1) {
    for (size_t ci = i0; ci < i1; ci++) {
        GbnContext &c = ctx[ci];
        if (!c.is_valid) { c.gap_cutoff_score = INT32_MAX; c.cutoff_score = INT32_MAX; continue; }
        if (opt.cutoff_score > 0) c.gap_cutoff_score = c.gap_cutoff_score_max = opt.cutoff_score;
        else c.gap_cutoff_score = c.gap_cutoff_score_max =
                 gbn::cutoff_from_evalue(opt.evalue, kbp_gap, c.eff_searchsp);
        int32_t trigger = INT32_MAX;
        if (c.lambda_u > 0 && c.K_u > 0 && c.H_u > 0)
            trigger = (int32_t)((opt.gap_trigger_bits * kLn2 + c.logK_u) / c.lambda_u);
        int32_t cut = std::min(trigger, c.gap_cutoff_score_max);
        c.cutoff_score = cut;
        c.reduced_cutoff = (int32_t)(0.9 * cut);
    }
    });
}

namespace gbn {

static const uint8_t kComplement[16] = {3, 2, 1, 0, 5, 4, 7, 6, 8, 9, 13, 12, 11, 10, 14, 15};

static int choose_lookup(const GbnOptions &o, int32_t entries, int32_t max_off, int &width) {
    int type = GBN_LUT_SMALL_NA;
    auto mb = [&](int w) { width = w; type = GBN_LUT_MB; };
    auto sm = [&](int w) { width = w; type = GBN_LUT_SMALL_NA; };
    switch (o.word_size) {
    case 4: case 5: case 6: sm(o.word_size); break;
    case 7: sm(entries < 250 ? 6 : 7); break;
    case 8: sm(entries < 8500 ? 7 : 8); break;
    case 9: if (entries < 1250) sm(7); else if (entries < 21000) sm(8); else mb(9); break;
    case 10: if (entries < 1250) sm(7); else if (entries < 8500) sm(8);
             else if (entries < 18000) mb(9); else mb(10); break;
    case 11: if (entries < 12000) sm(8);
             else if (o.lut11_gblastn_rule) mb(11);
             else if (entries < 180000) mb(10); else mb(11);
             break;
    case 12: if (entries < 8500) sm(8); else if (entries < 18000) mb(9);
             else if (entries < 60000) mb(10); else if (entries < 900000) mb(11); else mb(12);
             break;
    default: if (entries < 8500) sm(8); else if (entries < 300000) mb(11); else mb(12); break;
    }
    if (type == GBN_LUT_SMALL_NA && (entries >= 32767 || max_off >= 32768)) type = GBN_LUT_NA;
    return type;
}

// The table a batch of unmasked queries of these lengths gets (kind, lut width, stride): EstimateNumTableEntries over the
// two strands of every query + BlastChooseNaLookupTable, as choose_table below applies them to a batch's stretches.
void predict_table_shape(const GbnOptions &opt, int32_t nq, const int32_t *lens, int &type, int &lut, int &step) {
    int64_t entries = 0, off = 0, max_off = 0;
    for (int32_t i = 0; i < nq; i++)
        for (int strand = 0; strand < 2; strand++) {
            if (lens[i] > 0) { entries += lens[i] - 1; max_off = off + lens[i] - 1; }
            off += lens[i] + 1;                         // (a sentinel between contexts)
        }
    int width = 0;
    type = choose_lookup(opt, (int32_t)std::min<int64_t>(entries, INT32_MAX), (int32_t)std::min<int64_t>(max_off, INT32_MAX), width);
    lut = width; step = opt.word_size - width + 1;
}

// Index every lut-word that lies inside an indexed stretch [left, right] of the concatenated query,
// has no ambiguity code, and whose stretch can hold a full word_size word
// (CORE/blast_lookup.c:87-137, CORE/blast_nalookup.c:873-928).
template <class F>
static void each_query_word(const uint8_t *q, int32_t left, int32_t right, int word, int lut, F &&emit) {
    if (word > right - left + 1) return;
    const uint32_t mask = (lut == 16) ? 0xffffffffu : ((1u << (2 * lut)) - 1);
    uint32_t code = 0; int run = 0;
    for (int32_t p = left; p <= right; p++) {
        uint8_t b = q[p];
        if (b & 0xfc) { run = 0; code = 0; continue; }
        code = ((code << 2) | b) & mask;
        if (++run >= lut) emit(code, p - lut + 1);
    }
}

// The indexed stretches: every valid strand minus the soft masks (mask coordinates are plus-strand,
// mirrored onto the minus strand), in the form BLAST_ComplementMaskLocations produces them
// (CORE/blast_filter.c:1019-1119) -- including its empty stretches between abutting masks, which
// count in the table-size estimate below.
static void indexed_stretches(GbnBatch &b, const std::vector<QueryMask> &masks) {
    auto &out = b.lut.segments;
    out.clear();
    for (const GbnContext &c : b.ctx) {
        if (!c.is_valid) continue;
        const int32_t first = c.query_offset, last = c.query_offset + c.query_length - 1;
        std::vector<std::pair<int32_t, int32_t>> m;         // masked intervals in concatenated coordinates, ascending
        for (const QueryMask &k : masks) if (k.query == c.query_index) {
            if (c.frame < 0) m.emplace_back(last - k.to, last - k.from);
            else m.emplace_back(first + k.from, first + k.to);
        }
        if (c.frame < 0) std::reverse(m.begin(), m.end());
        if (m.empty()) { out.emplace_back(first, last); continue; }
        int32_t left = first; bool open = true; size_t i = 0;
        if (m[0].first <= first) { left = m[0].second + 1; i = 1; }     // the strand starts masked
        for (; i < m.size(); i++) {
            out.emplace_back(left, m[i].first - 1);
            if (m[i].second >= last) { open = false; break; }
            left = m[i].second + 1;
        }
        if (open) out.emplace_back(left, last);
    }
    // seeds are re-checked against the table when some unindexed gap is longer than 3 positions
    // (s_SeqLocListInvert, CORE/blast_nalookup.c:333-366) and the table words are shorter than word_size
    bool gap = false;
    if (!out.empty()) {
        if (std::max(0, out[0].first - 1) - 0 > 2) gap = true;
        for (size_t i = 0; i < out.size() && !gap; i++) {
            const int32_t start = out[i].second + 1;
            const int32_t stop = (i + 1 < out.size()) ? out[i + 1].first - 1 : b.qlen - 1;
            if (stop - start > 2) gap = true;
        }
    }
    b.lut.masked = gap;
}

// table kind and width from the size estimate (the small-NA -> standard fallback needs the cell counts
// and is applied by whoever fills the table: fill_lookup_host below, or the device builder)
static void choose_table(GbnBatch &b) {
    HostLookup &L = b.lut;
    int32_t entries = 0, max_off = 0;                   // EstimateNumTableEntries, CORE/lookup_util.c:193-209
    for (auto &sg : L.segments) { entries += sg.second - sg.first; max_off = std::max(max_off, sg.second); }
    int width = 0;
    L.type = choose_lookup(b.opt, entries, max_off, width);
    L.word = b.opt.word_size; L.lut = width; L.step = L.word - L.lut + 1;
    L.ncells = (int64_t)1 << (2 * width);
    L.cell_start.clear(); L.cell_qoff.clear(); L.pv.clear();
}

void fill_lookup_host(GbnBatch &b) {
    HostLookup &L = b.lut;
    const int width = L.lut;
    const uint8_t *q = b.query();
    std::vector<uint32_t> count((size_t)L.ncells, 0);
    for (auto &sg : L.segments) each_query_word(q, sg.first, sg.second, L.word, width, [&](uint32_t cell, int32_t) { count[cell]++; });
    L.cell_start.assign((size_t)L.ncells + 1, 0);
    uint32_t acc = 0; int64_t overflow_cells = 2;
    for (int64_t i = 0; i < L.ncells; i++) {
        L.cell_start[i] = acc; acc += count[i];
        if (count[i] > 1) overflow_cells += count[i] + 1;
    }
    L.cell_start[L.ncells] = acc;
    if (L.type == GBN_LUT_SMALL_NA && overflow_cells >= 32768) L.type = GBN_LUT_NA;
    L.cell_qoff.assign(acc, 0);
    std::vector<uint32_t> fill(L.cell_start.begin(), L.cell_start.end() - 1);
    for (auto &sg : L.segments) each_query_word(q, sg.first, sg.second, L.word, width, [&](uint32_t cell, int32_t off) { L.cell_qoff[fill[cell]++] = off; });
    if (L.type == GBN_LUT_MB) {
        // megablast chains yield the LAST inserted offset first (CORE/blast_nalookup.c:925-926,
        // CORE/blast_nascan.c:1413-1427): reverse every cell
        for (int64_t i = 0; i < L.ncells; i++)
            std::reverse(L.cell_qoff.begin() + L.cell_start[i], L.cell_qoff.begin() + L.cell_start[i + 1]);
    }
    L.pv.assign((size_t)((L.ncells + 31) / 32), 0);
    for (int64_t i = 0; i < L.ncells; i++) if (count[i]) L.pv[i >> 5] |= 1u << (i & 31);
}

// query buffers of batches that are gone, kept for the next ones (qbuf_take: why)
namespace {
std::mutex g_qbuf_mu; std::vector<std::vector<uint8_t>> g_qbuf_idle;
}
std::vector<uint8_t> qbuf_take(size_t n) {
    std::vector<uint8_t> v;
    {
        std::lock_guard<std::mutex> lk(g_qbuf_mu);
        for (size_t i = 0; i < g_qbuf_idle.size(); i++)
            if (g_qbuf_idle[i].capacity() >= n && g_qbuf_idle[i].capacity() <= n + n / 2 + 4096) { v.swap(g_qbuf_idle[i]); g_qbuf_idle.erase(g_qbuf_idle.begin() + (long)i); break; }
    }
    v.resize(n);            // (no reallocation when it came from the list; a fresh one is value-initialised: its pages are touched here)
    return v;
}
void qbuf_give(std::vector<uint8_t> &&v) {
    if (v.capacity() < ((size_t)1 << 20)) return;           // small ones: the allocator's business
    std::lock_guard<std::mutex> lk(g_qbuf_mu);
    if (g_qbuf_idle.size() < 6) g_qbuf_idle.emplace_back(std::move(v));
}

// Order (round 5): what the LOOKUP TABLE needs first -- the concatenated query, the indexed stretches, the table's kind --
// then `tables_hook` (the caller starts the device upload of the query and the table build there: they run on the builder's
// stream while this thread goes on), then what only the HOST needs before the search: Karlin-Altschul parameters per
// context, cut-offs, effective lengths (2 of a 5 Mb batch's 3 ms).  A context that turns out invalid only then (no
// Karlin-Altschul solution: a query of ambiguity codes) was counted in the stretches; the table is chosen again without it
// and *tables_stale says so.
int build_batch(GbnBatch &b, const GbnOptions &opt, int32_t nq, const uint8_t *const *seqs, const int32_t *lens,
                const std::vector<QueryMask> &masks, bool host_tables, const std::function<int()> &tables_hook, bool *tables_stale) {
    if (tables_stale) *tables_stale = false;
    b.opt = opt; b.nq = nq;
    trace_mark("batch: set-up starts");
    b.ctx.assign((size_t)2 * nq, GbnContext{});
    const int32_t pad = 64;         // sentinel padding so kernels may read windows past either end
    int64_t total = 1;
    for (int i = 0; i < nq; i++) total += 2 * ((int64_t)lens[i] + 1);
    if (total > INT32_MAX - 4 * pad) { set_error("query batch too long"); return GBN_ERR_ARG; }
    // the concatenation's buffer: one a batch before this one has given back, when there is one (a fresh 10 MB vector is 2,500
    // page faults before the first base is written -- 0.5-2 ms in front of everything else a set-up does, the device part
    // included); everything that is not a query base -- the pads, the separators -- is written below
    b.qbuf = qbuf_take((size_t)total + 2 * pad);
    std::memset(b.qbuf.data(), 15, (size_t)pad + 1);
    std::memset(b.qbuf.data() + (size_t)pad + (size_t)total, 15, (size_t)pad);      // (behind the last separator)
    b.qpad = pad + 1;
    uint8_t *q = b.qbuf.data() + b.qpad;
    int32_t off = 0;
    for (int i = 0; i < nq; i++) {
        int32_t L = lens[i];
        GbnContext &p = b.ctx[2 * i], &m = b.ctx[2 * i + 1];
        p.query_offset = off; p.query_length = L; p.frame = 1; p.query_index = i; p.is_valid = L > 0;
        off += L + 1;
        m.query_offset = off; m.query_length = L; m.frame = -1; m.query_index = i; m.is_valid = L > 0;
        off += L + 1;
    }
    b.qlen = off - 1;
    auto fill_range = [&](int i0, int i1) {             // both strands of queries [i0, i1)
        for (int i = i0; i < i1; i++) {
            const int32_t L = lens[i];
            std::memcpy(q + b.ctx[2 * i].query_offset, seqs[i], (size_t)L);
            uint8_t *r = q + b.ctx[2 * i + 1].query_offset;
            for (int32_t j = 0; j < L; j++) r[j] = kComplement[seqs[i][L - 1 - j] & 15];
            q[b.ctx[2 * i].query_offset + L] = 15; r[L] = 15;          // the separators behind both strands
        }
    };
    if (total < (1 << 18) || nq < 16) fill_range(0, nq);
    else setup_pool().parallel_for((size_t)nq, 64, [&](size_t i0, size_t i1) { fill_range((int)i0, (int)i1); });
    trace_mark("batch: query concatenated");
    // masks: per query ascending, disjoint, inside the query
    for (size_t i = 0; i < masks.size(); i++) {
        const QueryMask &k = masks[i];
        if (k.query < 0 || k.query >= nq || k.from < 0 || k.to < k.from || k.to >= lens[k.query] ||
            (i > 0 && masks[i - 1].query == k.query && masks[i - 1].to >= k.from) || (i > 0 && masks[i - 1].query > k.query)) {
            set_error("query masks must be sorted by (query, from), disjoint and inside their query"); return GBN_ERR_ARG;
        }
    }
    indexed_stretches(b, masks);
    choose_table(b);
    b.lut.masked = b.lut.masked && b.lut.word > b.lut.lut;
    if (host_tables) fill_lookup_host(b);
    trace_mark(host_tables ? "batch: lookup table built" : "batch: table kind chosen (tables are built on the device)");
    build_score_matrix(opt.reward, opt.penalty, b.matrix);      // (in front of the hook: they go up with the contexts' offsets)
    for (int i = 0; i < 256; i++) {
        int32_t s = 0;
        s += (i & 3) ? opt.penalty : opt.reward;
        s += ((i >> 2) & 3) ? opt.penalty : opt.reward;
        s += ((i >> 4) & 3) ? opt.penalty : opt.reward;
        s += (i >> 6) ? opt.penalty : opt.reward;
        b.score_table[i] = s;
    }
    if (tables_hook) { const int hrc = tables_hook(); if (hrc) return hrc; }
    double stdc[16]; uniform_acgt(stdc);
    // ungapped Karlin-Altschul parameters per context: independent, so a large batch spreads them over a few threads
    auto ka_range = [&](size_t c0, size_t c1) {
        for (size_t i = c0; i < c1; i++) {
            GbnContext &c = b.ctx[i];
            if (c.query_length <= 0) { c.is_valid = 0; continue; }
            double comp[16]; strand_composition(q + c.query_offset, c.query_length, comp);
            Karlin k;
            if (!ungapped_karlin(opt.reward, opt.penalty, comp, stdc, k)) {
                c.is_valid = 0; c.lambda_u = c.K_u = c.H_u = -1; continue;
            }
            c.lambda_u = k.lambda; c.K_u = k.K; c.logK_u = k.logK; c.H_u = k.H;
        }
    };
    if (b.ctx.size() < 512) ka_range(0, b.ctx.size());
    else setup_pool().parallel_for(b.ctx.size(), 128, ka_range);
    bool any = false; Karlin first;
    for (auto &c : b.ctx) if (c.is_valid) { first.lambda = c.lambda_u; first.K = c.K_u; first.logK = c.logK_u; first.H = c.H_u; any = true; break; }
    trace_mark("batch: Karlin-Altschul per context done");
    if (!any) { set_error("no valid query context (Karlin-Altschul parameters)"); return GBN_ERR_SETUP; }
    if (gapped_karlin(opt.gap_open, opt.gap_extend, opt.reward, opt.penalty, first, b.kbp_gap, b.round_down)) {
        set_error("unsupported reward/penalty/gap cost combination"); return GBN_ERR_UNSUPPORTED;
    }
    b.gap_x_dropoff = (int32_t)(opt.xdrop_gap_bits * kLn2 / b.kbp_gap.lambda);
    b.gap_x_dropoff_final = (int32_t)std::max(opt.xdrop_gap_final_bits * kLn2 / b.kbp_gap.lambda,
                                               (double)b.gap_x_dropoff);
    for (auto &c : b.ctx) if (c.is_valid)
        c.x_dropoff = (int32_t)(1.0 * std::ceil(opt.xdrop_ungap_bits * kLn2 / c.lambda_u));
    if (opt.db_num_seqs > 0) { b.set_effective_lengths(opt.db_length, opt.db_num_seqs); b.update_cutoffs(); }
    b.container = b.qlen > 8000 ? 1 : 0;
    b.diag_len = 1;
    while (b.diag_len < b.qlen) b.diag_len <<= 1;
    trace_mark("batch: cut-offs done");
    // a context without a Karlin-Altschul solution is no part of the search (CORE/blast_setup.c: the context is marked invalid
    // before the lookup table is built): the stretches above counted it -- once more without it
    bool lost = false;
    for (const GbnContext &c : b.ctx) if (c.query_length > 0 && !c.is_valid) lost = true;
    if (lost) {
        indexed_stretches(b, masks);
        choose_table(b);
        b.lut.masked = b.lut.masked && b.lut.word > b.lut.lut;
        if (host_tables) fill_lookup_host(b);
        if (tables_stale) *tables_stale = true;
    }
    return GBN_OK;
}

}  // namespace gbn
