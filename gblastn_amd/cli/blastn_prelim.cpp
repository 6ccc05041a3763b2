// blastn_prelim -- the reference's `blastn` command line on this library's stages.
//
// Same flag spelling as G-BLASTN's blastn (c++/src/algo/blast/blastinput/cmdline_flags.cpp:40-230,
// blast_args.cpp:2473-2550 for -use_gpu / -gpu_id / -mode / -query_list / -trace_t_num; shell/g.m.sh is the
// usage the reference documents): FASTA queries against a BLAST nucleotide database (v4 .nal/.nin/.nsq),
// default DUST soft masking, reference batch sizes (5 Mb megablast, 100 kb blastn), database volumes loaded as
// one HBM shard.  Written on the C++ host classes of include/gblastn_amd_host.hpp (CBlastPrelimSearch,
// CBlastTracebackSearch, CSearchPipeline = the reference's query / prelim / traceback thread pipeline): -mode 1
// runs a batch to completion before the next starts (APP/blastn_app.cpp Method1), -mode 0 / 2 overlap set-up,
// GPU search and CPU traceback of consecutive batches (Method2 / Method3).
// -outfmt 6 / 7: the twelve standard tabular columns of the FINAL alignments (traceback stage: identities,
// mismatches, gap openings).  -stage prelim stops after the preliminary search and prints its score-only HSPs
// (subject oid, coordinates, e-value, bit score, raw score, strand).
// No CPU fallback (-use_gpu false is refused).
#include "gblastn_amd_host.hpp"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <thread>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Query { std::string id; std::vector<uint8_t> seq; };

// IUPACna -> BLASTNA (COREI/blast_encoding.c IUPACNA_TO_BLASTNA): ACGT RYMK WSBD HVN-
int blastna_of(char c)
{
    switch (c >= 'a' && c <= 'z' ? c - 32 : c) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': case 'U': return 3;
    case 'R': return 4; case 'Y': return 5; case 'M': return 6; case 'K': return 7;
    case 'W': return 8; case 'S': return 9; case 'B': return 10; case 'D': return 11;
    case 'H': return 12; case 'V': return 13; case 'N': return 14; case '-': return 15;
    default: return -1;
    }
}

bool read_fasta(const std::string &path, std::vector<Query> &out, std::string &err)
{
    std::ifstream f(path);
    if (!f) { err = "cannot open query file " + path; return false; }
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            Query q; std::istringstream is(line.substr(1)); is >> q.id;
            if (q.id.empty()) q.id = "Query_" + std::to_string(out.size() + 1);
            out.push_back(std::move(q));
            continue;
        }
        if (out.empty()) { Query q; q.id = "Query_1"; out.push_back(std::move(q)); }
        for (char c : line) {
            if (c == ' ' || c == '\t' || (c >= '0' && c <= '9')) continue;
            const int v = blastna_of(c);
            if (v < 0) { err = std::string("bad residue '") + c + "' in " + path; return false; }
            out.back().seq.push_back((uint8_t)v);
        }
    }
    return true;
}

[[noreturn]] void die(const std::string &m) { std::fprintf(stderr, "blastn_prelim: %s\n", m.c_str()); std::exit(1); }
void check(int rc, const char *what) { if (rc) die(std::string(what) + ": " + gbn_last_error()); }

const char *kUsage =
    "usage: blastn_prelim -db NAME (-query FASTA | -query_list FILE) -use_gpu true [-gpu_id N] [-out FILE]\n"
    "       [-task megablast|blastn] [-word_size N] [-evalue X] [-reward N] [-penalty N] [-gapopen N]\n"
    "       [-gapextend N] [-dust yes|no|'level window linker'] [-max_target_seqs N] [-outfmt 6|7] [-mode 0|1|2]\n"
    "       [-stage traceback|prelim] [-trace_t_num N] [-num_threads N] [-strand both]\n"
    "       -gpu_id N: that device; -gpu_id -1 (the default): every device of the node, as in the reference\n"
    "         (GB/gpu_blast_multi_gpu_utils.cpp:43-68): the database's volumes are dealt to the GPUs in contiguous runs,\n"
    "         one search thread (pipeline) per GPU in this process, results merged per query\n"
    "       -num_threads N: search threads = database parts (default: one per GPU used); more parts than GPUs share them\n"
    "         round robin (N > 1 on one GPU exercises the same merge); host threads besides: -mode 2 set-up threads, -trace_t_num\n"
    "       (one process per GPU instead: python -m torch.distributed.run --nproc-per-node N -m gblastn_amd.blastn_sharded ...)\n"
    "       -batch_plan mixer|fixed: the reference's adaptive batch plan (a 10,000-base sample, then towards 2 M extensions per batch)\n"
    "         or batches of GetQueryBatchSize's fixed size; environment BATCH_SIZE: that many bases per batch (fixed)\n";

// e-value and bit score as the reference's formatter prints them (objtools/align_format/align_format_util.cpp:669-723)
std::string evalue_string(double e)
{
    char b[64];
    if (e < 1.0e-180) std::snprintf(b, sizeof b, "0.0");
    else if (e < 1.0e-99) std::snprintf(b, sizeof b, "%2.0le", e);
    else if (e < 0.0009) std::snprintf(b, sizeof b, "%3.0le", e);
    else if (e < 0.1) std::snprintf(b, sizeof b, "%4.3lf", e);
    else if (e < 1.0) std::snprintf(b, sizeof b, "%3.2lf", e);
    else if (e < 10.0) std::snprintf(b, sizeof b, "%2.1lf", e);
    else std::snprintf(b, sizeof b, "%5.0lf", e);
    return b;
}
std::string bits_string(double s)
{
    char b[64];
    if (s > 9999) std::snprintf(b, sizeof b, "%4.3le", s);
    else if (s > 99.9) std::snprintf(b, sizeof b, "%4.0ld", (long)s);
    else std::snprintf(b, sizeof b, "%4.1lf", s);
    return b;
}

// The reference's adaptive batch plan (CBatchSizeMixer, APP/blast_app_util.hpp:54-73, blast_app_util.cpp:67-86; used by
// APP/blastn_app.cpp:360-386): the first batch asks for 10,000 bases -- a sample --, every later one for as many bases as are
// expected to give `target` ungapped extensions that passed their cut-off (CLocalBlast::GetNumExtensions = good_init_extends),
// the hits-per-base ratio smoothed over the batches with weight 0.3 for the newest; never more than the query chunk size - 1000,
// never fewer than 100; a batch without a hit, or a size that hits a limit, forgets the history.
class BatchSizeMixer {
public:
    BatchSizeMixer(int32_t target, int32_t most) : target_(target), most_(most) {}
    // hits of the batch that has just come back (< 0: none has), and the bases THAT batch had been asked for -- the size of the
    // batch before, as in the reference, when batches run one at a time; with several in flight an older one's
    int32_t Next(int64_t hits = -1, int64_t asked = 0) {
        if (hits > 0) {
            const double now = 1.0 * (double)hits / (double)(asked > 0 ? asked : size_);
            ratio_ = ratio_ < 0 ? now : kMixIn * now + (1.0 - kMixIn) * ratio_;
            // (Int4)(1.0 * k_TargetHits / m_Ratio) in the reference: with fewer than 2e6 / 2^31 hits per base -- a sample of random
            // queries -- the quotient does not fit an Int4, and what an x86-64 build of blastn computes is the conversion's
            // "integer indefinite", INT_MIN: the size falls to the 100-base floor, one query goes alone, and ITS ratio (or its
            // having no hit at all) sends the plan to the cap.  Reproduced as that value, not as undefined behaviour.
            const double want = 1.0 * target_ / ratio_;
            size_ = (want >= 2147483648.0 || want != want) ? INT32_MIN : (int32_t)want;
            if (size_ > most_) { size_ = most_; ratio_ = -1.0; }
            else if (size_ < 100) { size_ = 100; ratio_ = -1.0; }
        } else if (hits == 0) { size_ = most_; ratio_ = -1.0; }
        return size_;
    }
private:
    static constexpr double kMixIn = 0.3;
    const int32_t target_, most_;
    double ratio_ = -1.0;
    int32_t size_ = 10000;
};

}  // namespace

int main(int argc, char **argv)
{
    // -timing true: one JSON line on stderr with the wall clock of the invocation's phases (bench.py --workload cli)
    const auto t_main = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double ph_args_fasta = 0, ph_init = 0, ph_db = 0, ph_dust = 0, ph_emit = 0, ph_wait = 0, ph_submit = 0;
    std::map<std::string, std::string> a;
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        if (k == "-h" || k == "-help") { std::fputs(kUsage, stdout); return 0; }
        if (k.size() < 2 || k[0] != '-' || i + 1 >= argc) die("bad argument '" + k + "'\n" + kUsage);
        a[k.substr(1)] = argv[++i];
    }
    auto get = [&](const char *k, const std::string &d) { auto it = a.find(k); return it == a.end() ? d : it->second; };
    if (!a.count("db") || (!a.count("query") && !a.count("query_list"))) die(std::string("-db and -query are required\n") + kUsage);
    const std::string gpu = get("use_gpu", "true");
    if (!(gpu == "true" || gpu == "T" || gpu == "1" || gpu == "yes")) die("this build has no CPU path: -use_gpu true is required");
    const std::string task = get("task", "megablast");
    if (task != "megablast" && task != "blastn") die("-task must be megablast or blastn (discontiguous megablast is not supported, as in G-BLASTN)");

    // ---- database: its volumes dealt to the GPUs in contiguous runs (one resident shard per search thread), global OIDs ----
    const int gpu_id = std::atoi(get("gpu_id", "-1").c_str());
    const int ndev_all = gbn_device_count();
    if (ndev_all < 1) die("no HIP device visible");
    std::vector<int> devices;
    if (gpu_id >= 0) devices.push_back(gpu_id); else for (int d = 0; d < ndev_all; d++) devices.push_back(d);
    auto t_ph = std::chrono::steady_clock::now();
    for (int d : devices) check(gbn_init(1, d), "gbn_init");
    ph_init = ms_since(t_ph); t_ph = std::chrono::steady_clock::now();
    // The volumes are opened and their shards uploaded on threads of their own while this one reads the queries and runs DUST
    // (12.5 GB of a 50 Gbp database: half a second of pinned-piece uploads, gbn_db_new_streamed, that nothing else waits for).
    GbnBlastDb *bdb = nullptr;
    int32_t nvol = 0; int nparts = 1;
    // (per part, for the table -stage prelim prints at the end: bases its shard holds, what its search thread scanned, the
    // wall time of its preliminary searches added up, and the time spent waiting for its results)
    struct Part { int device; GbnDb *shard = nullptr; long long scanned = 0; double prelim_ms = 0, scan_kernel_ms = 0; long long batches = 0; };
    std::vector<Part> parts;
    std::future<std::string> db_ready = std::async(std::launch::async, [&]() -> std::string {
        if (gbn_blastdb_open(&bdb, a["db"].c_str())) return std::string("gbn_blastdb_open: ") + gbn_last_error();
        nvol = gbn_blastdb_num_volumes(bdb);
        nparts = a.count("num_threads") ? std::max(1, std::atoi(a["num_threads"].c_str())) : (int)devices.size();
        nparts = std::max(1, std::min(nparts, (int)nvol));              // (a volume is not split)
        parts.resize((size_t)nparts);
        // shard_bounds of gblastn_amd/shard.py: part p holds volumes [p * V / P, (p + 1) * V / P); loaded in parallel
        std::vector<std::future<std::string>> loads;
        for (int p = 0; p < nparts; p++) {
            parts[(size_t)p].device = devices[(size_t)p % devices.size()];
            loads.push_back(std::async(std::launch::async, [&, p]() -> std::string {
                const int32_t v0 = (int32_t)((int64_t)p * nvol / nparts), v1 = (int32_t)((int64_t)(p + 1) * nvol / nparts);
                int32_t first = 0, n0 = 0, last_first = 0, last_n = 0;
                if (gbn_blastdb_volume_range(bdb, v0, &first, &n0) || gbn_blastdb_volume_range(bdb, v1 - 1, &last_first, &last_n)) return "gbn_blastdb_volume_range failed";
                if (gbn_use_device(parts[(size_t)p].device)) return std::string("gbn_use_device: ") + gbn_last_error();
                if (gbn_blastdb_load_shard(bdb, first, last_first + last_n - first, &parts[(size_t)p].shard)) return std::string("gbn_blastdb_load_shard: ") + gbn_last_error();
                return "";
            }));
        }
        std::string err;
        for (auto &f : loads) { const std::string e = f.get(); if (!e.empty() && err.empty()) err = e; }
        ph_db = ms_since(t_ph);
        return err;
    });

    // ---- queries ----
    std::vector<std::string> files;
    if (a.count("query")) files.push_back(a["query"]);
    if (a.count("query_list")) {
        std::ifstream l(a["query_list"]);
        if (!l) die("cannot open query list " + a["query_list"]);
        std::string p; while (std::getline(l, p)) if (!p.empty()) files.push_back(p);
    }
    std::vector<Query> queries; std::string err;
    for (auto &f : files) if (!read_fasta(f, queries, err)) die(err);
    if (queries.empty()) die("no query sequences");
    ph_args_fasta = ms_since(t_ph);


    int dust_level = 20, dust_window = 64, dust_linker = 1; bool dust = true;       // blastn's default filter
    {
        const std::string d = get("dust", "yes");
        if (d == "no" || d == "false") dust = false;
        else if (d != "yes" && d != "true") {
            std::istringstream is(d);
            if (!(is >> dust_level >> dust_window >> dust_linker)) die("-dust takes yes, no or 'level window linker'");
        }
    }
    // ---- query batches ----
    // BATCH_SIZE (the reference's experimentation knob, blastinput/blast_input_aux.cpp:66-124) or -batch_plan fixed: batches of a
    // fixed size -- GetQueryBatchSize's 5 Mb for megablast, 100 kb for blastn.  Otherwise the reference's adaptive plan
    // (BatchSizeMixer above; chunk sizes of SplitQuery_GetChunkSize, API/local_blast.cpp:61-101): a 10,000-base sample first.
    // (Its pipelined methods never feed the mixer -- APP/blastn_app.cpp:808-811: 10,000 bases per batch throughout; here every mode
    // does, and until the sample is back nothing else is submitted.)  A batch takes whole queries until it has the bases asked
    // for (CBlastInput::GetNextSeqBatch, blastinput/blast_input.cpp:135-170).
    const bool fixed_plan = std::getenv("BATCH_SIZE") != nullptr || get("batch_plan", "mixer") == "fixed";
    int64_t batch_bases = task == "megablast" ? 5000000 : 100000;
    if (const char *e = std::getenv("BATCH_SIZE")) batch_bases = std::max(1, std::atoi(e));
    int32_t mix_target = 2000000, mix_most = (task == "megablast" ? 5000000 : 1000000) - 1000;
    if (const char *e = std::getenv("GBN_CLI_MIXER_TARGET")) mix_target = std::max(1, std::atoi(e));       // (tests: plans of several batches on a small input)
    if (const char *e = std::getenv("GBN_CLI_MIXER_MOST")) mix_most = std::max(100, std::atoi(e));
    BatchSizeMixer mixer(mix_target, mix_most);
    std::vector<int64_t> hits_seen;
    if (!fixed_plan) batch_bases = mixer.Next();
    struct Batch { size_t first, count; int64_t asked; };
    std::vector<Batch> batches;
    size_t cursor = 0;
    auto form_batch = [&]() {                        // the next batch of the plan (cursor < queries.size())
        Batch bt; bt.first = cursor; bt.asked = batch_bases; int64_t acc = 0;
        while (cursor < queries.size() && acc < batch_bases) acc += (int64_t)queries[cursor++].seq.size();
        bt.count = cursor - bt.first;
        batches.push_back(bt);
    };
    // every batch as the pipeline takes it, DUST included, prepared now -- next to the database's upload -- with the queries of a
    // batch dealt to a few threads (symmetric DUST of 10,000 x 1 kb: 120 ms on one)
    auto make_batch = [&](const Batch &bt) {
        gbn::SQueryBatch q;
        q.seqs.reserve(bt.count);
        for (size_t k = 0; k < bt.count; k++) q.seqs.push_back(queries[bt.first + k].seq);
        if (!dust) return q;
        const unsigned hw = (unsigned)std::max(1, (int)gbn_host_cpus());
        const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)hw / 2, bt.count / 64 + 1}));
        std::vector<std::vector<gbn::SQueryBatch::Mask>> found(nt);
        std::vector<std::thread> ts;
        for (size_t t = 0; t < nt; t++)
            ts.emplace_back([&, t]() {
                std::vector<int32_t> f, to;
                for (size_t k = bt.count * t / nt; k < bt.count * (t + 1) / nt; k++) {
                    const Query &qq = queries[bt.first + k];
                    if (qq.seq.empty()) continue;
                    f.resize(qq.seq.size() / 2 + 2); to.resize(f.size());
                    const int32_t n = gbn_dust_mask(qq.seq.data(), (int32_t)qq.seq.size(), dust_level, dust_window, dust_linker, f.data(), to.data(), (int32_t)f.size());
                    for (int32_t j = 0; j < n && j < (int32_t)f.size(); j++) found[t].push_back(gbn::SQueryBatch::Mask{(int32_t)k, f[(size_t)j], to[(size_t)j]});
                }
            });
        for (auto &t : ts) t.join();
        for (auto &v : found) q.masks.insert(q.masks.end(), v.begin(), v.end());      // (threads hold ascending query ranges: the list stays in query order)
        return q;
    };
    std::vector<gbn::SQueryBatch> prepared;         // batches[i] as the pipeline takes it, for the i that have been formed
    auto prepare_more = [&](size_t upto) {          // batches formed and prepared until there are `upto` (or no query is left)
        const auto t_d = std::chrono::steady_clock::now();
        while (prepared.size() < upto && cursor < queries.size()) { form_batch(); prepared.push_back(make_batch(batches.back())); }
        ph_dust += ms_since(t_d);
    };
    prepare_more(fixed_plan ? (size_t)-1 : 1);      // (a fixed plan: everything now, next to the upload; the mixer's: the sample)
    { const std::string e = db_ready.get(); if (!e.empty()) die(e); }

    // ---- options (API/blast_nucl_options.cpp defaults of the task, then the flags) ----
    GbnOptions opt; gbn_default_options(&opt, task == "megablast");
    if (a.count("word_size")) opt.word_size = std::atoi(a["word_size"].c_str());
    if (a.count("evalue")) opt.evalue = std::atof(a["evalue"].c_str());
    if (a.count("reward")) opt.reward = std::atoi(a["reward"].c_str());
    if (a.count("penalty")) opt.penalty = std::atoi(a["penalty"].c_str());
    if (a.count("gapopen")) opt.gap_open = std::atoi(a["gapopen"].c_str());
    if (a.count("gapextend")) opt.gap_extend = std::atoi(a["gapextend"].c_str());
    if (a.count("max_target_seqs")) opt.hitlist_size = std::atoi(a["max_target_seqs"].c_str());
    if (a.count("xdrop_ungap")) opt.xdrop_ungap_bits = std::atof(a["xdrop_ungap"].c_str());
    if (a.count("xdrop_gap")) opt.xdrop_gap_bits = std::atof(a["xdrop_gap"].c_str());
    if (opt.penalty > 0) opt.penalty = -opt.penalty;
    // statistics over the alias file's totals when it states them (DBLIST files with STATS_*), else the database's
    const int64_t stat_len = gbn_blastdb_stat_length(bdb); const int32_t stat_n = gbn_blastdb_stat_num_seqs(bdb);
    opt.db_length = stat_len > 0 ? stat_len : gbn_blastdb_total_length(bdb);
    opt.db_num_seqs = stat_n > 0 ? stat_n : gbn_blastdb_num_seqs(bdb);

    const int outfmt = std::atoi(get("outfmt", "6").c_str());
    if (outfmt != 6 && outfmt != 7) die("-outfmt 6 or 7 (tabular)");
    const bool overlapped = get("mode", "1") != "1";                 // 0 and 2: the reference's pipelined methods
    const std::string stage = get("stage", "traceback");
    if (stage != "traceback" && stage != "prelim") die("-stage traceback or prelim");
    const bool with_traceback = stage == "traceback";
    const int trace_threads = std::max(1, std::atoi(get("trace_t_num", "2").c_str()));
    FILE *out = stdout;
    if (a.count("out")) { out = std::fopen(a["out"].c_str(), "w"); if (!out) die("cannot write " + a["out"]); }

    // the twelve standard columns of a final alignment: qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore
    auto emit_final = [&](const Batch &bt, const GbnTbHSP *h, const int64_t *qs, const GbnContext *ctx) {
        for (size_t k = 0; k < bt.count; k++) {
            const Query &q = queries[bt.first + k];
            if (outfmt == 7) {
                int64_t subjects = 0;
                for (int64_t i = qs[k]; i < qs[k + 1]; i++) subjects += (i == qs[k] || h[i].hsp.oid != h[i - 1].hsp.oid);
                std::fprintf(out, "# BLASTN (gblastn_amd)\n# Query: %s\n# Database: %s\n", q.id.c_str(), a["db"].c_str());
                std::fprintf(out, "# Fields: query id, subject id, %% identity, alignment length, mismatches, gap opens, q. start, q. end, s. start, s. end, evalue, bit score\n# %lld hits found\n", (long long)(qs[k + 1] - qs[k]));
                (void)subjects;
            }
            for (int64_t i = qs[k]; i < qs[k + 1]; i++) {
                const GbnTbHSP &x = h[i]; const GbnHSP &p = x.hsp;
                const bool minus = (p.context & 1) != 0;
                const int32_t qlen = ctx[p.context].query_length;
                // query always forward; a minus-strand hit shows the subject reversed
                const int32_t qst = minus ? qlen - p.q_end + 1 : p.q_offset + 1, qen = minus ? qlen - p.q_offset : p.q_end;
                const int32_t sst = minus ? p.s_end : p.s_offset + 1, sen = minus ? p.s_offset + 1 : p.s_end;
                std::fprintf(out, "%s\tgnl|BL_ORD_ID|%d\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n", q.id.c_str(), p.oid,
                             x.align_length ? 100.0 * x.num_ident / x.align_length : 0.0, x.align_length,
                             x.align_length - x.num_ident - x.gaps, x.gap_opens, qst, qen, sst, sen,
                             evalue_string(p.evalue).c_str(), bits_string(x.bit_score).c_str());
            }
        }
    };
    // -stage prelim: the score-only HSPs the collector kept, per query best subject first
    auto emit_prelim = [&](const Batch &bt, const GbnCollector *col, const gbn::CSearchPipeline::SWorkItem &it) {
        const int64_t nl = gbn_collector_num_lists(col); const int64_t *st = gbn_collector_list_starts(col);
        const int32_t *lq = gbn_collector_list_queries(col); const GbnHSP *h = gbn_collector_hsps(col);
        const GbnContext *ctx = gbn_batch_contexts(it.prelim->Batch());
        double lambda = 0, K = 0; gbn_batch_karlin_gapped(it.prelim->Batch(), &lambda, &K);
        std::vector<std::vector<int64_t>> per((size_t)bt.count);
        for (int64_t l = 0; l < nl; l++) per[(size_t)lq[l]].push_back(l);
        for (size_t k = 0; k < bt.count; k++) {
            const Query &q = queries[bt.first + k];
            auto &ls = per[k];
            std::stable_sort(ls.begin(), ls.end(), [&](int64_t x, int64_t y) {
                const GbnHSP &u = h[st[x]], &v = h[st[y]];
                if (u.evalue != v.evalue) return u.evalue < v.evalue;
                if (u.score != v.score) return u.score > v.score;
                return u.oid < v.oid; });
            if (outfmt == 7) {
                std::fprintf(out, "# BLASTN preliminary search (gblastn_amd)\n# Query: %s\n# Database: %s\n", q.id.c_str(), a["db"].c_str());
                std::fprintf(out, "# Fields: query id, subject oid, q. start, q. end, s. start, s. end, evalue, bit score, score, strand\n# %zu subjects\n", ls.size());
            }
            for (int64_t l : ls)
                for (int64_t i = st[l]; i < st[l + 1]; i++) {
                    const GbnHSP &x = h[i];
                    const bool minus = (x.context & 1) != 0;
                    const int32_t qlen = ctx[x.context].query_length;
                    const int32_t qs = minus ? qlen - x.q_end + 1 : x.q_offset + 1, qe = minus ? qlen - x.q_offset : x.q_end;
                    const int32_t ss = minus ? x.s_end : x.s_offset + 1, se = minus ? x.s_offset + 1 : x.s_end;
                    const double bits = (lambda * x.score - std::log(K)) / std::log(2.0);
                    std::fprintf(out, "%s\t%d\t%d\t%d\t%d\t%d\t%.3g\t%.1f\t%d\t%s\n", q.id.c_str(), x.oid, qs, qe, ss, se,
                                 x.evalue, bits, x.score, minus ? "minus" : "plus");
                }
        }
    };

    GbnDiagnostics diag; std::memset(&diag, 0, sizeof(diag));
    const auto t_search = std::chrono::steady_clock::now();
    try {
        // one pipeline (set-up threads -> search thread -> traceback consumers) per part, each on its part's GPU: the
        // search threads of the reference with a GPU each (API/prelim_search_runner.hpp:135-166, GB/gpu_blast_multi_gpu_utils.cpp:105-139)
        std::vector<std::unique_ptr<gbn::CBlastSeqSrc>> srcs; std::vector<std::unique_ptr<gbn::CSearchPipeline>> pipes;
        for (Part &pt : parts) {
            srcs.emplace_back(new gbn::CBlastSeqSrc(pt.shard, false));
            pipes.emplace_back(new gbn::CSearchPipeline(opt, *srcs.back(), trace_threads, with_traceback, overlapped));
        }
        // batches enter a few ahead of the results coming out (a batch holds its lookup tables in HBM until printed); under the
        // mixer's plan the sample goes alone, and a batch is formed when it is submitted, with the size the mixer asks for then
        size_t submitted = 0, printed = 0;
        bool sampled = fixed_plan, finished = false;
        while (printed < batches.size() || cursor < queries.size()) {
            const size_t ahead = (overlapped && sampled && batch_bases >= 50000) ? 8 : 1;        // (the plan's small batches go one at a time, as in the reference's Method1)
            while (submitted < printed + ahead && (submitted < prepared.size() || cursor < queries.size())) {
                prepare_more(submitted + 1);
                const auto t_s = std::chrono::steady_clock::now();
                for (auto &pp : pipes) pp->Submit(prepared[submitted]);
                prepared[submitted] = gbn::SQueryBatch();       // (the pipeline has its copy)
                ph_submit += ms_since(t_s);
                submitted++;
            }
            if (!finished && cursor >= queries.size() && submitted == batches.size()) { for (auto &pp : pipes) pp->Finish(); finished = true; }
            std::vector<gbn::CSearchPipeline::TItem> items;
            for (size_t pi = 0; pi < pipes.size(); pi++) {
                auto &pp = pipes[pi];
                const auto t_w = std::chrono::steady_clock::now();
                gbn::CSearchPipeline::TItem it = pp->Next();
                ph_wait += ms_since(t_w);
                if (!it) die("the pipeline ended early");
                if (it->status != GBN_OK) die(it->error);
                const GbnDiagnostics &d = it->prelim->diagnostics;
                parts[pi].scanned += d.subject_bases_scanned; parts[pi].prelim_ms += d.total_ms; parts[pi].scan_kernel_ms += d.scan_kernel_ms; parts[pi].batches++;
                diag.subject_bases_scanned += d.subject_bases_scanned; diag.seeds += d.seeds; diag.gapped_extensions += d.gapped_extensions;
                diag.total_ms = std::max(diag.total_ms, d.total_ms);
                items.push_back(std::move(it));
            }
            if (!fixed_plan) {
                int64_t hits = 0;
                for (auto &it : items) hits += it->prelim->diagnostics.good_init_extends;
                batch_bases = mixer.Next(hits, batches[printed].asked); sampled = true; hits_seen.push_back(hits);
            }
            const Batch bt = batches[printed];
            const auto t_e = std::chrono::steady_clock::now();
            const GbnContext *ctx = gbn_batch_contexts(items[0]->prelim->Batch());
            if (with_traceback) {
                if (items.size() == 1) {
                    const GbnTraceback *tb = items[0]->traceback->Results();
                    emit_final(bt, gbn_traceback_hsps(tb), gbn_traceback_query_starts(tb), ctx);
                } else {        // the parts' final lists -> the database's: per query best e-value, score, OID; hit-list cut
                    std::vector<const GbnTbHSP *> hp; std::vector<const int64_t *> qp; int64_t total = 0;
                    for (auto &it : items) {
                        const GbnTraceback *tb = it->traceback->Results();
                        hp.push_back(gbn_traceback_hsps(tb)); qp.push_back(gbn_traceback_query_starts(tb)); total += gbn_traceback_num_hsps(tb);
                    }
                    std::vector<GbnTbHSP> merged((size_t)std::max<int64_t>(total, 1)); std::vector<int64_t> mq(bt.count + 1);
                    if (gbn_traceback_merge((int32_t)items.size(), hp.data(), qp.data(), (int32_t)bt.count, opt.hitlist_size, merged.data(), mq.data()) < 0) die("gbn_traceback_merge failed");
                    emit_final(bt, merged.data(), mq.data(), ctx);
                }
            } else if (items.size() == 1) emit_prelim(bt, items[0]->stream->Get(), *items[0]);
            else {              // the lists every part's collector kept, through one collector in ascending OID order (parts ascend)
                gbn::CBlastHSPStream all((int32_t)bt.count, opt.hitlist_size);
                for (auto &it : items) {
                    const GbnCollector *c = it->stream->Get();
                    gbn::Check(gbn_collector_write(all.Get(), gbn_collector_hsps(c), gbn_collector_num_hsps(c)), "gbn_collector_write");
                }
                all.Close();
                emit_prelim(bt, all.Get(), *items[0]);
            }
            ph_emit += ms_since(t_e);
            printed++;
        }
        for (auto &pp : pipes) pp->Close();
    } catch (const gbn::CBlastException &e) { die(e.what()); }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search).count();
    std::fprintf(stderr, "blastn_prelim: %zu queries in %zu batches, %lld subject bases scanned, %lld seeds, %lld gapped extensions, %.1f ms in the preliminary stage\n",
                 queries.size(), batches.size(), (long long)diag.subject_bases_scanned, (long long)diag.seeds,
                 (long long)diag.gapped_extensions, diag.total_ms);
    {
        std::string plan;
        for (size_t i = 0; i < batches.size() && i < 12; i++) plan += (i ? " " : "") + std::to_string(batches[i].asked) + "/" + std::to_string(batches[i].count);
        std::fprintf(stderr, "blastn_prelim: batch plan %s (bases asked for / queries taken)%s: %s%s\n", fixed_plan ? "fixed" : "mixer",
                     fixed_plan ? "" : " -- CBatchSizeMixer, APP/blast_app_util.cpp:67-86", plan.c_str(), batches.size() > 12 ? " ..." : "");
    }
    if (!fixed_plan) {
        std::string h; for (size_t i = 0; i < hits_seen.size() && i < 12; i++) h += " " + std::to_string(hits_seen[i]);
        std::fprintf(stderr, "blastn_prelim: hits per batch (good_init_extends, what the mixer is fed):%s\n", h.c_str());
    } else std::fprintf(stderr, "blastn_prelim: hits per batch: \n");
    if (!with_traceback) {
        // -stage prelim: the scaling self-check.  One line per part (= search thread, on its device), then the whole run:
        // run it with -gpu_id 0 and with -gpu_id -1 on a node of N GPUs and the last lines give the 1 / N table.
        std::fprintf(stderr, "# part device shard_Mbp batches scanned_Gbp prelim_search_ms scan_kernels_ms Gbp_per_s_of_its_searches\n");
        long long all = 0;
        for (size_t pi = 0; pi < parts.size(); pi++) {
            const Part &pt = parts[pi]; all += pt.scanned;
            std::fprintf(stderr, "# %zu %d %.1f %lld %.3f %.1f %.1f %.1f\n", pi, pt.device, gbn_db_total_bases(pt.shard) / 1e6, pt.batches,
                         pt.scanned / 1e9, pt.prelim_ms, pt.scan_kernel_ms, pt.prelim_ms > 0 ? pt.scanned / 1e9 / (pt.prelim_ms * 1e-3) : 0.0);
        }
        std::fprintf(stderr, "# total: %zu parts on %zu devices, %.3f Gbp scanned in %.1f ms wall (first batch submitted to last batch printed) = %.1f Gbp/s\n",
                     parts.size(), devices.size(), all / 1e9, wall_ms, wall_ms > 0 ? all / 1e9 / (wall_ms * 1e-3) : 0.0);
    }
    if (out != stdout) std::fclose(out);
    const auto t_down = std::chrono::steady_clock::now();
    long long shard_bytes = 0;
    for (Part &pt : parts) shard_bytes += gbn_db_total_bases(pt.shard) / 4;
    // A one-shot program leaves the device's memory to the driver, as the reference's blastn does: freeing a 12.5 GB shard, the
    // pools and the streams one by one took 0.23 s of a 1.2 s invocation.  GBN_CLI_TEARDOWN=1: the orderly way (leak checks).
    const bool teardown = std::getenv("GBN_CLI_TEARDOWN") != nullptr;
    if (teardown) { for (Part &pt : parts) gbn_db_free(pt.shard); gbn_blastdb_close(bdb); gbn_release(); }
    if (get("timing", "false") == "true") {
        // submit = FASTA batches -> SQueryBatch incl. DUST (dust_ms of it); wait = blocked on the pipeline for a batch's results
        // (set-up, preliminary search and traceback of the batches in flight); emit = formatting + writing the rows
        std::fprintf(stderr, "{\"blastn_prelim_timing\": {\"total_ms\": %.1f, \"args_fasta_ms\": %.1f, \"gbn_init_ms\": %.1f, \"db_open_upload_ms\": %.1f, "
                             "\"shard_GB\": %.2f, \"search_wall_ms\": %.1f, \"submit_ms\": %.1f, \"dust_ms\": %.1f, \"wait_results_ms\": %.1f, \"emit_rows_ms\": %.1f, "
                             "\"prelim_search_ms_max_batch\": %.1f, \"teardown_ms\": %.1f, \"queries\": %zu, \"batches\": %zu, \"parts\": %zu}}\n",
                     ms_since(t_main), ph_args_fasta, ph_init, ph_db, shard_bytes / 1e9, wall_ms, ph_submit, ph_dust, ph_wait, ph_emit, diag.total_ms, ms_since(t_down),
                     queries.size(), batches.size(), parts.size());
    }
    if (!teardown) { std::fflush(nullptr); std::_Exit(0); }
    return 0;
}
