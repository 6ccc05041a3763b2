// gpu_blastn_amd_shim.cpp -- the ONE translation unit a maintainer of the reference adds to
// src/algo/blast/gpu_blast/ to put libgblastn_amd.so behind G-BLASTN's own entry points.  In libgpublastn.a
// (src/algo/blast/gpu_blast/Makefile.in: CPP_SRC + gpu_blastn_MB_and_smallNa.cu) it takes the place of
//     gpu_blastn_pre_search_engine.cpp   gpu_blast_multi_gpu_utils.cpp
//     gpu_blastn_na_ungapped_v3.cpp      gpu_blastn_MB_and_smallNa.cu
// and defines the four symbols the rest of the tree takes from those files -- all four with C++ LINKAGE, as the
// reference declares them (there is no extern "C" in its gpu_blast headers):
//     Int2 Blast_gpu_RunPreliminarySearchWithInterrupt(...)   include/algo/blast/gpu_blast/gpu_blastn.h:31-48
//          called by CPrelimSearchRunner::operator()          include/algo/blast/api/prelim_search_runner.hpp:96-113
//     int  Blast_gpu_Init(bool, int) / void Blast_gpu_Release()   gpu_blastn.h:50-51, src/app/blast/blastn_app.cpp:462,491
//     void gpu_ReleaseDBMemory()                               gpu_blastn_na_ungapped_v3.h:21, blastn_app.cpp:489
// The host-only files of that library (work_thread*.cpp, thread_work_queue.cpp, gpu_logfile.cpp, utility.cpp) stay.
// Link -lgblastn_amd -lamdhip64 in place of -lcudart (src/app/blast/Makefile.blastn.app:16).  The file is compiled
// inside the configured toolkit tree (every toolkit header needs the configure-generated ncbiconf_unix.h);
// tests/test_boundary.py syntax-checks it against the reference's headers where /root/reference exists.
//
// Everything that does not need a toolkit type lives in the library and is tested there:
//   gbn_prelim_search_lists    grouping of the HSPs into per-subject lists, ascending oid
//   gbn_db_cache_find/_insert  the shard cache that gpu_ReleaseDBMemory() empties
//   gbn_shard_builder_*        subjects appended one by one into a 16-byte aligned slab, with their OIDs
//   gbn_use_device             the calling thread's GPU (the lease of GB/gpu_blast_multi_gpu_utils.cpp:105-139)
// so what remains here is field-by-field translation between the toolkit's structures and the PODs of
// gblastn_amd.h.  Anything the library does not do (discontiguous templates, programs other than blastn, PHI /
// RPS, a nucleotide table it does not know) and every thread that finds no free GPU goes to the stock CPU
// function, as the reference does (GB/gpu_blastn_pre_search_engine.cpp:1508-1521: gpu_id == -1).
#include <algo/blast/gpu_blast/gpu_blastn.h>
#include <algo/blast/core/blast_engine.h>
#include <algo/blast/core/blast_hspstream.h>
#include <algo/blast/core/blast_seqsrc.h>
#include <algo/blast/core/blast_nalookup.h>
#include <algo/blast/core/blast_hits.h>
#include <algo/blast/core/blast_util.h>
#include <algo/blast/core/blast_diagnostics.h>
#include <algo/blast/gpu_blast/gpu_blastn_na_ungapped_v3.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "gblastn_amd.h"

namespace {

// ---- the GPUs of the process and which thread holds which (GpuBlastMultiGPUsUtils: InitGPUs / ThreadFetchGPU /
// ThreadReplaceGPU, GB/gpu_blast_multi_gpu_utils.cpp:43-139): a calling thread takes a free GPU for the length of
// its call; threads that find none search on the CPU ----
std::mutex s_lease_mu;
std::vector<int> s_free_gpus;
bool s_use_gpu = false;

int s_FetchGpu()
{
    std::lock_guard<std::mutex> lk(s_lease_mu);
    if (!s_use_gpu || s_free_gpus.empty()) return -1;
    const int id = s_free_gpus.back();
    s_free_gpus.pop_back();
    return id;
}
void s_ReplaceGpu(int id)
{
    std::lock_guard<std::mutex> lk(s_lease_mu);
    s_free_gpus.push_back(id);
}

// word_length / lut_word_length / masked_locations live in a different structure for every nucleotide table kind
// (COREI/blast_nalookup.h:63-77, 132-153, 237-264); BlastNaWordFinder's callers switch on lut_type the same way
// (CORE/na_ungapped.c:1753-1795)
bool s_TableShape(const LookupTableWrap* w, Int4* word, bool* discontiguous, const BlastSeqLoc** masked)
{
    *discontiguous = false; *masked = NULL;
    switch (w->lut_type) {
    case eMBLookupTable: {
        const BlastMBLookupTable* t = (const BlastMBLookupTable*)w->lut;
        *word = t->word_length; *discontiguous = t->discontiguous != 0; *masked = t->masked_locations; return true; }
    case eSmallNaLookupTable: {
        const BlastSmallNaLookupTable* t = (const BlastSmallNaLookupTable*)w->lut;
        *word = t->word_length; *masked = t->masked_locations; return true; }
    case eNaLookupTable: {
        const BlastNaLookupTable* t = (const BlastNaLookupTable*)w->lut;
        *word = t->word_length; *masked = t->masked_locations; return true; }
    default:
        return false;
    }
}

// Soft query masks.  The library builds its own lookup structures from the query, so it needs what
// LookupTableWrapInit was given: the lookup_segments (CORE/blast_nalookup.c:413-417, 977-981).  They are freed
// before this boundary; the table keeps their inverse, masked_locations = s_SeqLocListInvert(lookup_segments)
// (CORE/blast_nalookup.c:333-366), in concatenated query coordinates over both strands.  Cut per query at the
// plus-strand context it is the (query, from, to) list gbn_batch_new_masked takes (the minus strand mirrors it).
// One thing the inverse has lost: masked stretches of 3 positions and fewer (`stop - start > 2`); DUST intervals
// are 7 bases and longer, so blastn's default filter crosses intact -- a shorter user-supplied lower-case mask does not.
void s_QueryMasks(const BlastSeqLoc* masked, const BlastQueryInfo* qi,
                  std::vector<int32_t>& mq, std::vector<int32_t>& mfrom, std::vector<int32_t>& mto)
{
    for (const BlastSeqLoc* loc = masked; loc; loc = loc->next) {
        if (!loc->ssr) continue;
        for (Int4 c = qi->first_context; c <= qi->last_context; c += 2) {       // plus strands
            const Int4 first = qi->contexts[c].query_offset, last = first + qi->contexts[c].query_length - 1;
            const Int4 from = loc->ssr->left > first ? loc->ssr->left : first;
            const Int4 to = loc->ssr->right < last ? loc->ssr->right : last;
            if (from > to) continue;
            mq.push_back((c - qi->first_context) / 2); mfrom.push_back(from - first); mto.push_back(to - first);
        }
    }
    // (the list ascends in concatenated coordinates and the contexts ascend: sorted by (query, from) already)
}

// 64-bit FNV-1a over what identifies a resident shard: device, database name, the OIDs it holds
uint64_t s_Mix(uint64_t h, const void* p, size_t n)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// The resident shard of this call: the OIDs the iterator hands THIS thread (BlastSeqSrcIteratorNext honours OID
// lists / GI filters and, with several search threads on one database, gives every thread chunks of its own:
// GB/gpu_blastn_pre_search_engine.cpp:1243-1252), as stored = 2 bits per base (eBlastEncodingProtein is the toolkit's
// name for that, CORE/blast_engine.c:1043-1050).  Kept in the library's cache under (device, database name, OIDs)
// until gpu_ReleaseDBMemory(): the next query batch of the same thread layout finds it there -- the reference's
// per-OID device cache (GB/gpu_blastn_MB_and_smallNa.cu:1462-1468) as one slab.
GbnDb* s_GetShard(const BlastSeqSrc* seq_src, int device, Int2* status)
{
    *status = 0;
    std::vector<Int4> oids;
    BlastSeqSrcIterator* itr = BlastSeqSrcIteratorNewEx(MAX(BlastSeqSrcGetNumSeqs(seq_src) / 100, 1));
    if (!itr) { *status = -1; return NULL; }
    for (Int4 oid; (oid = BlastSeqSrcIteratorNext(seq_src, itr)) != BLAST_SEQSRC_EOF; ) {
        if (oid == BLAST_SEQSRC_ERROR) break;
        oids.push_back(oid);
    }
    BlastSeqSrcIteratorFree(itr);
    if (oids.empty()) return NULL;                      // nothing left for this thread: not an error
    uint64_t key = 14695981039346656037ull;
    const char* name = BlastSeqSrcGetName(seq_src);
    key = s_Mix(key, &device, sizeof device);
    if (name) key = s_Mix(key, name, strlen(name));
    key = s_Mix(key, oids.data(), oids.size() * sizeof(Int4));
    const void* handle = (const void*)(uintptr_t)key;
    if (GbnDb* db = gbn_db_cache_find(handle)) return db;

    GbnShardBuilder* sb = NULL;
    if (gbn_shard_builder_new(&sb, (int32_t)oids.size()) != GBN_OK) { *status = -1; return NULL; }
    BlastSeqSrcGetSeqArg arg;
    memset(&arg, 0, sizeof arg);
    arg.encoding = eBlastEncodingProtein;
    int rc = GBN_OK;
    for (size_t i = 0; i < oids.size() && rc == GBN_OK; ++i) {
        arg.oid = oids[i];
        if (BlastSeqSrcGetSequence(seq_src, &arg) < 0) continue;        // as the reference's loop does (:1269)
        rc = gbn_shard_builder_add_oid(sb, arg.oid, arg.seq->sequence, arg.seq->length);
        BlastSeqSrcReleaseSequence(seq_src, &arg);
    }
    if (arg.seq) BlastSequenceBlkFree(arg.seq);
    GbnDb* db = NULL;
    if (rc == GBN_OK) rc = gbn_shard_builder_finish(sb, &db);      // uploads the slab, frees the builder's host copy
    gbn_shard_builder_free(sb);
    if (rc != GBN_OK) { *status = -1; return NULL; }
    if (gbn_db_cache_insert(handle, db) != GBN_OK) { gbn_db_free(db); return gbn_db_cache_find(handle); }
    return db;
}

struct SListSink { BlastHSPStream* stream; const BlastQueryInfo* query_info; };

// TInterruptFnPtr returns a Boolean (one byte), GbnInterruptFn an int: called through its own signature
struct SInterrupt { TInterruptFnPtr fn; SBlastProgress* progress; };
int s_Interrupt(void* arg)
{
    SInterrupt* i = (SInterrupt*)arg;
    return (*i->fn)(i->progress) ? 1 : 0;
}

// GbnHspListFn: one BlastHSPList per subject -> BlastHSPStreamWrite (the stream takes the list and nulls the pointer;
// it is the one locked section of the reference's N-thread contract, CORE/blast_hspstream.c:316-365)
int s_WriteList(void* arg, int32_t oid, const GbnHSP* h, int32_t n)
{
    SListSink* s = (SListSink*)arg;
    BlastHSPList* list = Blast_HSPListNew(0);
    if (!list) return 1;
    list->oid = oid;
    for (int32_t j = 0; j < n; ++j) {
        BlastHSP* hsp = NULL;
        if (Blast_HSPInit(h[j].q_offset, h[j].q_end, h[j].s_offset, h[j].s_end, h[j].q_gapped_start, h[j].s_gapped_start,
                          h[j].context, s->query_info->contexts[h[j].context].frame, 1, h[j].score, NULL, &hsp)) {
            Blast_HSPListFree(list); return 1; }
        hsp->evalue = h[j].evalue;
        Blast_HSPListSaveHSP(list, hsp);
    }
    list->best_evalue = h[0].evalue;
    for (int32_t j = 1; j < n; ++j) if (h[j].evalue < list->best_evalue) list->best_evalue = h[j].evalue;
    return BlastHSPStreamWrite(s->stream, &list) == kBlastHSPStream_Success ? 0 : 1;
}

}   // namespace

Int2 Blast_gpu_RunPreliminarySearchWithInterrupt(EBlastProgramType program,
        BLAST_SequenceBlk* query, BlastQueryInfo* query_info, const BlastSeqSrc* seq_src,
        const BlastScoringOptions* score_options, BlastScoreBlk* sbp, LookupTableWrap* lookup_wrap,
        const BlastInitialWordOptions* word_options, const BlastExtensionOptions* ext_options,
        const BlastHitSavingOptions* hit_options, const BlastEffectiveLengthsOptions* eff_len_options,
        const PSIBlastOptions* psi_options, const BlastDatabaseOptions* db_options, const BlastGPUOptions* gpu_options,
        BlastHSPStream* hsp_stream, BlastDiagnostics* diagnostics,
        TInterruptFnPtr interrupt_search, SBlastProgress* progress_info)
{
    Int4 word = 0; bool discontiguous = false; const BlastSeqLoc* masked = NULL;
    const bool ours = program == eBlastTypeBlastn && gpu_options && gpu_options->use_gpu &&
                      lookup_wrap && s_TableShape(lookup_wrap, &word, &discontiguous, &masked) && !discontiguous;
    const int gpu = ours ? s_FetchGpu() : -1;           // this thread's GPU for the length of the call, -1: none free
    if (gpu < 0)                                        // everything else: the stock CPU path
        return Blast_RunPreliminarySearchWithInterrupt(program, query, query_info, seq_src, score_options,
                   sbp, lookup_wrap, word_options, ext_options, hit_options, eff_len_options, psi_options, db_options,
                   hsp_stream, diagnostics, interrupt_search, progress_info);
    struct SLease { int id; ~SLease() { s_ReplaceGpu(id); } } lease = { gpu };
    if (gbn_use_device(gpu) != GBN_OK) return -1;

    GbnOptions o;                                       // options -> POD
    gbn_default_options(&o, ext_options->ePrelimGapExt == eGreedyScoreOnly);
    o.word_size = word;
    o.reward = score_options->reward;      o.penalty = score_options->penalty;
    o.gap_open = score_options->gap_open;  o.gap_extend = score_options->gap_extend;
    o.xdrop_ungap_bits = word_options->x_dropoff;      o.gap_trigger_bits = word_options->gap_trigger;
    o.xdrop_gap_bits = ext_options->gap_x_dropoff;     o.xdrop_gap_final_bits = ext_options->gap_x_dropoff_final;
    o.evalue = hit_options->expect_value;              o.cutoff_score = hit_options->cutoff_score;
    o.min_diag_separation = hit_options->min_diag_separation;
    o.hitlist_size = hit_options->hitlist_size;
    // the size of the WHOLE database (every thread / shard uses it: CORE/blast_setup.c:638, :698-775); a sequence
    // source without a total length is a set of subjects searched one by one (db_length == 0, GB/...engine.cpp:1283)
    o.db_length = eff_len_options->db_length ? eff_len_options->db_length : BlastSeqSrcGetTotLen(seq_src);
    o.db_num_seqs = o.db_length == 0 ? 0 : (eff_len_options->dbseq_num ? eff_len_options->dbseq_num : BlastSeqSrcGetNumSeqs(seq_src));

    // queries: the concatenated BLASTNA buffer already has the layout the engine wants; hand over the plus
    // strand of every query (contexts 0, 2, 4, ...) and the soft masks the lookup table was built with
    std::vector<const Uint1*> seqs; std::vector<Int4> lens;
    for (Int4 c = query_info->first_context; c <= query_info->last_context; c += 2) {
        seqs.push_back(query->sequence + query_info->contexts[c].query_offset);
        lens.push_back(query_info->contexts[c].query_length);
    }
    std::vector<int32_t> mq, mfrom, mto;
    s_QueryMasks(masked, query_info, mq, mfrom, mto);

    Int2 status = 0;
    GbnDb* shard = s_GetShard(seq_src, gpu, &status);
    if (!shard) return status;                          // (no OIDs left for this thread: success, nothing to write)
    GbnBatch* batch = NULL; GbnDiagnostics d; memset(&d, 0, sizeof d);
    SListSink sink = { hsp_stream, query_info };
    SInterrupt intr = { interrupt_search, progress_info };
    int rc = gbn_batch_new_masked(&batch, &o, (int32_t)seqs.size(), seqs.data(), lens.data(),
                                  (int32_t)mq.size(), mq.data(), mfrom.data(), mto.data(), 1);
    if (rc == GBN_OK)
        rc = gbn_prelim_search_lists(batch, shard, s_WriteList, &sink, &d, interrupt_search ? s_Interrupt : NULL, &intr);
    if (rc == GBN_OK && diagnostics) {                  // (per-thread counters; the caller's structure may be shared: COREI/blast_diagnostics.h:118-126)
        if (diagnostics->ungapped_stat) {
            diagnostics->ungapped_stat->lookup_hits += d.lookup_hits;
            diagnostics->ungapped_stat->init_extends += (Int4)d.init_extends;
            diagnostics->ungapped_stat->good_init_extends += (Int4)d.good_init_extends;
        }
        if (diagnostics->gapped_stat) {
            diagnostics->gapped_stat->extensions += (Int4)d.gapped_extensions;
            diagnostics->gapped_stat->good_extensions += (Int4)d.good_extensions;
            diagnostics->gapped_stat->num_seqs_passed += (Int4)d.seqs_passed;
        }
    }
    gbn_batch_free(batch);                              // the derived search parameters are the callee's; the shard stays cached
    return rc == GBN_OK ? 0 : (rc == GBN_ERR_INTERRUPTED ? BLASTERR_INTERRUPTED : -1);
}

// InitGPUs (GB/gpu_blast_multi_gpu_utils.cpp:43-68): returns the number of GPUs the search threads may lease --
// 0 = CPU mode (use_gpu false, or no usable device), 1 for an explicit gpu_id, all of them for gpu_id == -1
int Blast_gpu_Init(bool isInit, int gpu_id)
{
    std::lock_guard<std::mutex> lk(s_lease_mu);
    s_free_gpus.clear(); s_use_gpu = false;
    if (!isInit) return 0;
    const int n = gbn_device_count();
    if (n < 1) return 0;
    if (gpu_id != -1) {
        if (gpu_id < 0 || gpu_id >= n || gbn_init(1, gpu_id) != GBN_OK) return 0;
        s_free_gpus.push_back(gpu_id);
    } else {
        for (int d = n - 1; d >= 0; --d) if (gbn_init(1, d) == GBN_OK) s_free_gpus.push_back(d);
    }
    s_use_gpu = !s_free_gpus.empty();
    return (int)s_free_gpus.size();
}

void Blast_gpu_Release()
{
    std::lock_guard<std::mutex> lk(s_lease_mu);
    s_free_gpus.clear(); s_use_gpu = false;
    gbn_release();
}

void gpu_ReleaseDBMemory()
{
    gbn_release_db_memory();
}
