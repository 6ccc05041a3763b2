// gbn_host.hpp -- internal host-side declarations of the MI355X blastn engine.
// Product code: never includes anything from oracle/.
#pragma once
#include "../../include/gblastn_amd.h"
#include <cstdint>
#include <functional>
#include <future>
#include <string>
#include <utility>
#include <vector>

namespace gbn {

struct Karlin { double lambda = -1, K = -1, logK = 0, H = -1; bool valid() const { return lambda > 0 && K > 0 && H > 0; } };

// ---- statistics (stat.cpp) : CORE/blast_stat.c restated for nucleotides ----
void   build_score_matrix(int reward, int penalty, int32_t m[16][16]);        // blast_stat.c:1036
bool   ungapped_karlin(int reward, int penalty, const double comp1[16],
                       const double comp2[16], Karlin &out);                  // blast_stat.c:2673
void   uniform_acgt(double comp[16]);                                         // blast_stat.c:1861
void   strand_composition(const uint8_t *seq, int32_t len, double comp[16]);  // blast_stat.c:1958
int    gapped_karlin(int gap_open, int gap_extend, int reward, int penalty,
                     const Karlin &ungapped, Karlin &out, bool &round_down);  // blast_stat.c:3806
int    alpha_beta(int reward, int penalty, int gap_open, int gap_extend,
                  const Karlin &ungapped, bool gapped, double &alpha, double &beta); // :3919
int32_t length_adjustment(double K, double logK, double alpha_d_lambda, double beta,
                          int32_t qlen, int64_t db_len, int32_t db_nseq);     // :4994
int32_t score_for_evalue(double E, const Karlin &k, int64_t searchsp);        // :3994
double  evalue_for_score(int32_t S, const Karlin &k, int64_t searchsp);       // :4111
int32_t cutoff_from_evalue(double E, const Karlin &k, int64_t searchsp);      // :4044

// ---- query batch (batch.cpp) ----
struct HostLookup {
    int type = 0, word = 0, lut = 0, step = 0;
    int64_t ncells = 0;
    // per-cell chains in the order the reference reports them
    // (MB: descending query offset; SmallNa/Na: ascending)
    std::vector<uint32_t> cell_start;   // ncells + 1
    std::vector<int32_t>  cell_qoff;    // 0-based query offsets
    std::vector<uint32_t> pv;           // 1 bit per cell
    // stretches of the concatenated query that are indexed: the contexts minus the soft masks
    std::vector<std::pair<int32_t, int32_t>> segments;   // [left, right], as BLAST_ComplementMaskLocations yields them
    bool masked = false;                // the reference's lut->masked_locations != NULL: seeds are re-checked
};

struct DeviceBatch;     // device mirrors, defined in engine.hpp
std::vector<uint8_t> qbuf_take(size_t n);       // batch.cpp: a batch's concatenation buffer, recycled
void qbuf_give(std::vector<uint8_t> &&v);
void trace_mark(const char *what);   // GBN_TRACE=1: wall-clock marks on stderr (engine.cpp)
// The processors this PROCESS may use at a time (batch.cpp): the hardware threads, cut down to the scheduler affinity mask and to the
// cgroup's CPU quota (cpu.max / cpu.cfs_quota_us).  Every pool of host threads is sized from this, not from
// std::thread::hardware_concurrency(): a container that shows 256 hardware threads and grants 16 CPUs of quota stops ALL threads of
// a process for the rest of a 100 ms period once they have used 1.6 CPU-seconds of it (round 6: C4 lost a quarter of its time so).
unsigned host_cpus();
// GBN_CPU_ACCOUNT=1: CPU time (CLOCK_THREAD_CPUTIME_ID) of the host threads by what they were doing, summed over the process; read with
// gbn_debug_cpu_account.  A scope adds the calling thread's CPU time between its construction and its end to one class.
enum { GBN_CPU_SETUP = 0, GBN_CPU_SETUP_POOL, GBN_CPU_SEARCH, GBN_CPU_STAGE, GBN_CPU_REPLAY, GBN_CPU_END_COLLECT, GBN_CPU_TRACEBACK, GBN_CPU_TRACEBACK_WORKERS, GBN_CPU_SUBMIT, GBN_CPU_TB_UNPACK, GBN_CPU_TB_START, GBN_CPU_TB_ALIGN, GBN_CPU_TB_RESCORE, GBN_CPU_TB_SORT_OUT, GBN_CPU_N };
struct CpuScope { int cat; long long t0; explicit CpuScope(int c); ~CpuScope(); };
}  // namespace gbn

struct GbnBatch {
    // the last host replay queued that reads this batch (engine.cpp: under the engine's host_mu); freeing the batch waits for it
    std::shared_future<void> host_tail;
    GbnOptions opt{};
    int32_t nq = 0;
    std::vector<GbnContext> ctx;
    std::vector<uint8_t> qbuf;          // [pad sentinels][15] ctx0 [15] ctx1 ... [15][pad]
    int32_t qpad = 0;                   // index of query position 0 inside qbuf
    int32_t qlen = 0;                   // concatenated length (no outer sentinels)
    int32_t matrix[16][16];
    int32_t score_table[256];
    gbn::Karlin kbp_gap;
    bool round_down = false;
    int32_t gap_x_dropoff = 0, gap_x_dropoff_final = 0;
    int32_t container = 0, diag_len = 0;
    gbn::HostLookup lut;
    gbn::DeviceBatch *dev = nullptr;
    const uint8_t *query() const { return qbuf.data() + qpad; }
    int context_of(int32_t q_off) const;                     // BSearchContextInfo
    void set_effective_lengths(int64_t db_len, int32_t db_nseq);
    void update_cutoffs();
};

struct GbnDb {
    const uint8_t *d_packed = nullptr;  // device
    bool owns = false;
    int64_t nbytes = 0;
    int32_t num_seqs = 0, first_oid = 0;
    std::vector<int64_t> byte_off;
    std::vector<int32_t> len;
    int64_t total_bases = 0;
    int64_t *d_byte_off = nullptr;
    int32_t *d_len = nullptr;
    void *tile_cache = nullptr;         // engine-private (tile tables per lut/step)
    void *engine = nullptr;             // the device context (engine.hpp: Engine) the shard is resident on
    // Sequences longer than the engine's MAX_DBSEQ_LEN are held and searched as chunks of that length overlapping
    // by DBSEQ_CHUNK_OVERLAP (CORE/blast_engine.c:218-262, :455-540): num_seqs / byte_off / len above describe the
    // chunks -- subjects of their own to every kernel -- and these map them back.  Empty: nothing is chunked.
    int32_t real_seqs = 0;                      // sequences in the shard (= num_seqs when nothing is chunked)
    std::vector<int32_t> real_of, chunk_ord;    // per chunk: its sequence, its ordinal in it
    std::vector<int32_t> real_len, first_virt;  // per sequence: its length, its first chunk
    int32_t chunk_len = 0;                      // MAX_DBSEQ_LEN the shard was built with
    // ambiguity runs of the sequences (BLASTNA code 4..15 over [start, start + length)): the shard holds 2 bits per
    // base, the traceback stage puts the codes back before it aligns (CORE/blast_traceback.c: the subject is
    // fetched as eBlastEncodingNucleotide there).  Empty: no sequence has any.
    struct AmbRun { int32_t start, length; uint8_t code; };
    std::vector<std::vector<AmbRun>> amb;       // per sequence (sized on first use)
    // OIDs of the sequences when they are not first_oid, first_oid + 1, ... (a shard built from what a BlastSeqSrc
    // iterator handed out: OID lists / GI filters leave holes); ascending.  Empty: contiguous from first_oid.
    std::vector<int32_t> oid_map;
    int32_t oid_of(int32_t v) const { const int32_t r = real_of.empty() ? v : real_of[(size_t)v]; return oid_map.empty() ? first_oid + r : oid_map[(size_t)r]; }
    int32_t local_of(int32_t oid) const {       // the sequence's index in the shard, -1: not here
        if (oid_map.empty()) { const int32_t l = oid - first_oid; return l >= 0 && l < real_seqs ? l : -1; }
        size_t lo = 0, hi = oid_map.size();
        while (lo < hi) { const size_t m = (lo + hi) / 2; if (oid_map[m] < oid) lo = m + 1; else hi = m; }
        return lo < oid_map.size() && oid_map[lo] == oid ? (int32_t)lo : -1;
    }
    int32_t chunk_of(int32_t v) const { return real_of.empty() ? 0 : chunk_ord[(size_t)v]; }
    // A VIEW (gbn_block_view): the subjects of several resident blocks as one shard.  It owns no subject bytes --
    // d_packed is the lowest slab address of its blocks and byte_off reaches into every one of them -- only its own
    // subject tables; the blocks must outlive it (freeing a block frees the views over it).  Empty: not a view.
    std::vector<const GbnDb *> view_parts;
    // the subject ranges the shard is searched in (engine_abi.cpp: plan_ranges), remembered per (stride, limits): two passes
    // over every subject's length otherwise, 0.1 ms per search of a 50,000-subject shard (under the engine's lock)
    struct RangePlan { int step; int64_t range_bytes, tile_limit; std::vector<std::pair<int32_t, int32_t>> ranges; };
    mutable std::vector<RangePlan> range_plans;
};
constexpr int32_t kDbseqChunkOverlap = 100;     // COREI/blast_hits.h:169

struct GbnResults {
    std::vector<GbnHSP> hsps;
    std::vector<GbnSeed> seeds;
    std::vector<GbnInitHit> init_hits;
    void *engine = nullptr;             // the device context that fills / filled them (set by the search entry points)
    // of the search that is filling them.  What the merge of chunk lists needs of the query batch (e-values after the
    // merge: CORE/blast_engine.c:455-540) is COPIED here when the search starts: gbn_prelim_search_end may run after
    // the caller has freed the batch.  `diag` is the caller's and must outlive the search, as the API says.
    struct ChunkMerge { gbn::Karlin kbp_gap; std::vector<int64_t> eff_searchsp; double evalue = 0; } merge;
    GbnDiagnostics *diag = nullptr;
    std::shared_future<void> host_tail; // the last host replay queued that appends to these results (under the engine's host_mu)
    int32_t chunk_len = 0;              // > 0: hsps holds chunk lists (pad_ = ordinal + 1) that merge_chunk_lists has yet to join
};

namespace gbn {
void set_error(const std::string &msg);
struct QueryMask { int32_t query, from, to; };      // soft mask, plus-strand coordinates, inclusive
int  build_batch(GbnBatch &b, const GbnOptions &opt, int32_t nq,
                 const uint8_t *const *seqs, const int32_t *lens,
                 const std::vector<QueryMask> &masks = std::vector<QueryMask>(), bool host_tables = true,
                 const std::function<int()> &tables_hook = std::function<int()>(), bool *tables_stale = nullptr);
void predict_table_shape(const GbnOptions &opt, int32_t nq, const int32_t *lens, int &type, int &lut, int &step);   // batch.cpp
void fill_lookup_host(GbnBatch &b);     // the host-side table builder (host-only set-up, GBN_HOST_LOOKUP=1)
int  upload_batch(GbnBatch &b);             // = upload_batch_tables + upload_batch_contexts
int  upload_batch_tables(GbnBatch &b);      // the query on the device, the lookup structures queued on the builder's stream (needs what build_batch has when it calls its hook)
int  upload_batch_contexts(GbnBatch &b);    // contexts, cut-offs, score tables (needs the finished host set-up)
void free_device_batch(DeviceBatch *d);
// chunk lists of one sequence (GbnHSP::pad_ = chunk ordinal + 1) -> one list per sequence (Blast_HSPListsMerge)
void merge_chunk_lists(std::vector<GbnHSP> &hsps, int32_t chunk_len, const GbnResults::ChunkMerge &b, GbnDiagnostics *diag);
int  gather_shard_bytes(const GbnDb &db, const std::vector<int64_t> &src_off, const std::vector<int32_t> &nbytes, std::vector<uint8_t> &out);
}
