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
//   gbn_results_emit_lists     grouping of the HSPs into per-subject lists, ascending oid
//   gbn_block_view             the resident blocks of several OID chunks searched as one shard
//   gbn_db_cache_find/_insert  the cache of resident OID blocks that gpu_ReleaseDBMemory() empties
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
#include <algo/blast/core/blast_seqsrc_impl.h>      // BlastSeqSrcIterator::current_pos: where a chunk of the source ends
#include <algo/blast/core/blast_nalookup.h>
#include <algo/blast/core/blast_hits.h>
#include <algo/blast/core/blast_util.h>
#include <algo/blast/core/blast_diagnostics.h>
#include <algo/blast/gpu_blast/gpu_blastn_na_ungapped_v3.h>
#include <string.h>
#include <stdlib.h>
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
bool s_TableShape(const LookupTableWrap* w, Int4* word, Int4* lut_word, bool* discontiguous, const BlastSeqLoc** masked)
{
    *discontiguous = false; *masked = NULL;
    switch (w->lut_type) {
    case eMBLookupTable: {
        const BlastMBLookupTable* t = (const BlastMBLookupTable*)w->lut;
        *word = t->word_length; *lut_word = t->lut_word_length; *discontiguous = t->discontiguous != 0; *masked = t->masked_locations; return true; }
    case eSmallNaLookupTable: {
        const BlastSmallNaLookupTable* t = (const BlastSmallNaLookupTable*)w->lut;
        *word = t->word_length; *lut_word = t->lut_word_length; *masked = t->masked_locations; return true; }
    case eNaLookupTable: {
        const BlastNaLookupTable* t = (const BlastNaLookupTable*)w->lut;
        *word = t->word_length; *lut_word = t->lut_word_length; *masked = t->masked_locations; return true; }
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

// When the lookup word is as long as the search word (blastn with many queries: word 11 = lut 11; every small word
// size) the reference needs no seed re-check and keeps NO masked_locations (CORE/blast_nalookup.c:413-417, :585,
// :977-981) -- yet the table was built from the lookup segments, masked stretches left out.  What was indexed is in
// the table itself: a query position p is covered iff a word [p, p + lut) of the table contains it, and with
// lut == word the segments that can hold a word are exactly the unions of their words.  The complement of the cover
// (per plus-strand context, as above) is handed over as masks: the library then indexes the same words.  (Bases the
// reference left out for an ambiguity code fall into the complement as well; the library leaves those words out anyway.)
void s_MarkWord(std::vector<unsigned char>& cover, Int4 q_off, Int4 lut)
{
    for (Int4 k = 0; k < lut && (size_t)(q_off + k) < cover.size(); ++k) if (q_off + k >= 0) cover[(size_t)(q_off + k)] = 1;
}
bool s_MasksFromTable(const LookupTableWrap* w, const BlastQueryInfo* qi,
                      std::vector<int32_t>& mq, std::vector<int32_t>& mfrom, std::vector<int32_t>& mto)
{
    const BlastContextInfo& lastc = qi->contexts[qi->last_context];
    std::vector<unsigned char> cover((size_t)(lastc.query_offset + lastc.query_length), 0);
    switch (w->lut_type) {
    case eMBLookupTable: {          // chains: hashtable[cell] -> next_pos[...], 1-based (CORE/blast_nalookup.c:893-926)
        const BlastMBLookupTable* t = (const BlastMBLookupTable*)w->lut;
        for (Int4 c = 0; c < t->hashsize; ++c)
            for (Int4 i = t->hashtable[c]; i; i = t->next_pos[i]) s_MarkWord(cover, i - 1, t->lut_word_length);
        break; }
    case eSmallNaLookupTable: {     // final_backbone: -1 empty, >= 0 one offset, else -(start of a list that a negative value ends) (CORE/na_ungapped.c:82-104)
        const BlastSmallNaLookupTable* t = (const BlastSmallNaLookupTable*)w->lut;
        for (Int4 c = 0; c < t->backbone_size; ++c) {
            const Int4 v = t->final_backbone[c];
            if (v == -1) continue;
            if (v >= 0) { s_MarkWord(cover, v, t->lut_word_length); continue; }
            for (Int4 src = -v; t->overflow[src] >= 0; ++src) s_MarkWord(cover, t->overflow[src], t->lut_word_length);
        }
        break; }
    case eNaLookupTable: {          // thick backbone: up to NA_HITS_PER_CELL offsets in the cell, else a stretch of the overflow array (CORE/na_ungapped.c:113-138)
        const BlastNaLookupTable* t = (const BlastNaLookupTable*)w->lut;
        for (Int4 c = 0; c < t->backbone_size; ++c) {
            const NaLookupBackboneCell& cell = t->thick_backbone[c];
            const Int4* pos = cell.num_used <= NA_HITS_PER_CELL ? cell.payload.entries : t->overflow + cell.payload.overflow_cursor;
            for (Int4 i = 0; i < cell.num_used; ++i) s_MarkWord(cover, pos[i], t->lut_word_length);
        }
        break; }
    default:
        return false;
    }
    for (Int4 c = qi->first_context; c <= qi->last_context; c += 2) {           // plus strands; the minus strand mirrors them
        const Int4 first = qi->contexts[c].query_offset, len = qi->contexts[c].query_length;
        for (Int4 p = 0; p < len; ) {
            if (cover[(size_t)(first + p)]) { ++p; continue; }
            Int4 e = p;
            while (e + 1 < len && !cover[(size_t)(first + e + 1)]) ++e;
            mq.push_back((c - qi->first_context) / 2); mfrom.push_back(p); mto.push_back(e);
            p = e + 1;
        }
    }
    return true;
}

// The database on the device is a set of BLOCKS: one per chunk of OIDs the sequence source hands out.  The shim asks
// the source for chunks of the reference's own size (a hundredth of the database, GB/gpu_blastn_pre_search_engine.cpp:1243;
// the stock CPU threads take the same: CORE/blast_engine.c), so every chunk any thread ever gets is the same stretch
// [k * chunk, (k + 1) * chunk) of the source's bookmark (CSeqDB::GetNextOIDChunk), whichever of the N search threads
// asks and whichever query batch is running: the reference's per-OID device cache
// (GB/gpu_blastn_MB_and_smallNa.cu:1461-1467) at chunk granularity.  A block is uploaded once per device and kept
// in the library's block cache (gbn_block_cache_*: keyed by device, database name and the OIDs themselves -- no hash
// that could collide) until gpu_ReleaseDBMemory(); OID lists and GI filters give chunks with holes, cached the same way.
GbnDb* s_GetBlock(const BlastSeqSrc* seq_src, const std::vector<Int4>& oids, Int2* status)
{
    *status = 0;
    const char* name = BlastSeqSrcGetName(seq_src);
    GbnDb* db = NULL;
    if (gbn_block_cache_find(name, oids.data(), (int32_t)oids.size(), &db) != GBN_OK) { *status = -1; return NULL; }
    if (db) return db;

    GbnShardBuilder* sb = NULL;
    if (gbn_shard_builder_new(&sb, (int32_t)oids.size()) != GBN_OK) { *status = -1; return NULL; }
    BlastSeqSrcGetSeqArg arg;
    memset(&arg, 0, sizeof arg);
    arg.encoding = eBlastEncodingProtein;               // = as stored, 2 bits per base (CORE/blast_engine.c:1043-1050)
    int rc = GBN_OK;
    for (size_t i = 0; i < oids.size() && rc == GBN_OK; ++i) {
        arg.oid = oids[i];
        if (BlastSeqSrcGetSequence(seq_src, &arg) < 0) continue;        // as the reference's loop does (:1269)
        rc = gbn_shard_builder_add_oid(sb, arg.oid, arg.seq->sequence, arg.seq->length);
        BlastSeqSrcReleaseSequence(seq_src, &arg);
    }
    if (arg.seq) BlastSequenceBlkFree(arg.seq);
    if (rc == GBN_OK) rc = gbn_shard_builder_finish(sb, &db);      // uploads the slab, frees the builder's host copy
    gbn_shard_builder_free(sb);
    if (rc != GBN_OK) { *status = -1; return NULL; }
    GbnDb* kept = NULL;             // (two threads of one device may have built the same block at the same time: the first stays)
    if (gbn_block_cache_insert(name, oids.data(), (int32_t)oids.size(), db, &kept) != GBN_OK) { gbn_db_free(db); *status = -1; return NULL; }
    return kept;
}

// Chunks per group.  A group is one search: the larger, the fewer fixed costs per query batch (each search loads the
// lookup table's slices once per bin -- 0.5 ms on the C2 shard -- and ends in one host synchronisation); the smaller, the
// finer the GPUs that share the source's bookmark balance at the end of the database.  One leased GPU: everything the
// iterator still has, as ONE search (threads without a GPU are on the stock CPU path, a thousand times slower per chunk:
// whatever the GPU thread leaves them is what the call waits for).  Several leased GPUs: a quarter of one GPU's share
// (8 GPUs: 3 chunks).  GBN_SHIM_GROUP_CHUNKS overrides (1 = the reference's chunk-by-chunk loop).
int s_leased_at_init = 1;
int s_GroupChunks()
{
    if (const char* e = getenv("GBN_SHIM_GROUP_CHUNKS")) { const int v = atoi(e); if (v >= 1) return v; }
    if (s_leased_at_init <= 1) return 100;
    const int g = 100 / (4 * s_leased_at_init);
    return g < 1 ? 1 : g;
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
    Int4 word = 0, lut_word = 0; bool discontiguous = false; const BlastSeqLoc* masked = NULL;
    const bool ours = program == eBlastTypeBlastn && gpu_options && gpu_options->use_gpu &&
                      lookup_wrap && s_TableShape(lookup_wrap, &word, &lut_word, &discontiguous, &masked) && !discontiguous;
    // (psi_options and db_options play no part in a blastn preliminary search: PSSMs are protein, the genetic code is
    // for translated subjects.  They are passed on untouched to the stock CPU function below and otherwise ignored.)
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
    if (masked) s_QueryMasks(masked, query_info, mq, mfrom, mto);
    else if (lut_word == word && !s_MasksFromTable(lookup_wrap, query_info, mq, mfrom, mto)) return -1;
    // (word > lut and no masked_locations: the table was built from the whole query, CORE/blast_nalookup.c:413-417)

    // Chunks of OIDs are fetched from the source as the reference's loop fetches them (GB/...engine.cpp:1243-1290; chunk =
    // a hundredth of the database, so that every chunk any thread ever gets is the same stretch of the source's bookmark
    // and its resident block is found again): N search threads (GPU-leased or on the stock CPU path) share the bookmark,
    // nobody drains it.  What differs is how they are SEARCHED: not one synchronous search per chunk (a hundred rounds of
    // launches, read-backs and host synchronisations per query batch) but
    //   * in GROUPS of up to s_GroupChunks() chunks, a group being one VIEW over its resident blocks (gbn_block_view: one
    //     tile table, one launch per kernel, one cached record set -- no subject byte is copied), and
    //   * PIPELINED: gbn_prelim_search_begin of group k returns when its scan is done; while its extension stages run,
    //     this thread fetches (and, on a cold cache, uploads) the blocks of group k + 1 and writes the lists of group k - 1
    //     to the HSP stream.
    // The lists reach the stream in ascending OID order per group, as the reference's subject-by-subject loop writes them.
    // The query batch is set up once per call.
    GbnBatch* batch = NULL; GbnDiagnostics d; memset(&d, 0, sizeof d);
    SListSink sink = { hsp_stream, query_info };
    SInterrupt intr = { interrupt_search, progress_info };
    int rc = GBN_OK;
    BlastSeqSrcIterator* itr = BlastSeqSrcIteratorNewEx(MAX(BlastSeqSrcGetNumSeqs(seq_src) / 100, 1));
    if (!itr) return -1;
    GbnResults* res[2] = { NULL, NULL };
    if (gbn_results_new(&res[0]) != GBN_OK || gbn_results_new(&res[1]) != GBN_OK) rc = GBN_ERR_NOMEM;
    int cur = 0; bool in_flight = false;                // res[cur ^ 1] belongs to the search begun last
    const int group_chunks = s_GroupChunks();
    std::vector<Int4> oids;
    std::vector<GbnDb*> blocks;
    for (bool more = true; more && rc == GBN_OK; ) {
        blocks.clear();
        while (more && rc == GBN_OK && (int)blocks.size() < group_chunks) {
            oids.clear();
            for (;;) {                                  // one chunk: until the iterator has used up what the source gave it
                const Int4 oid = BlastSeqSrcIteratorNext(seq_src, itr);
                if (oid == BLAST_SEQSRC_EOF) { more = false; break; }
                if (oid == BLAST_SEQSRC_ERROR) { more = false; rc = GBN_ERR_ARG; break; }
                oids.push_back(oid);
                if (itr->current_pos == UINT4_MAX) break;
            }
            if (oids.empty() || rc != GBN_OK) break;
            Int2 status = 0;
            GbnDb* block = s_GetBlock(seq_src, oids, &status);
            if (!block) { if (status) rc = GBN_ERR_HIP; continue; }     // (a chunk without a usable sequence: as the reference's loop)
            blocks.push_back(block);
        }
        if (blocks.empty() || rc != GBN_OK) break;
        // the group as one shard; blocks whose slabs lie too far apart for a view are searched one by one
        // (GBN_ERR_UNSUPPORTED: no view over these blocks -- one by one; anything else, out of memory included, is the call's failure)
        GbnDb* view = NULL;
        const int vrc = rc == GBN_OK ? gbn_block_view(blocks.data(), (int32_t)blocks.size(), &view) : rc;
        const bool one = vrc == GBN_OK && view;
        if (rc == GBN_OK && vrc != GBN_OK && vrc != GBN_ERR_UNSUPPORTED) { rc = vrc; break; }
        if (!batch && rc == GBN_OK) {
            // (the first group's scan records are binned underneath the set-up of the query batch when they are not resident yet:
            // the binning kernel needs no lookup table -- gbn_db_prepare_records returns at once, and does nothing when they are)
            if (one) (void)gbn_db_prepare_records(view, &o, (int32_t)lens.size(), lens.data());
            rc = gbn_batch_new_masked(&batch, &o, (int32_t)seqs.size(), seqs.data(), lens.data(),
                                      (int32_t)mq.size(), mq.data(), mfrom.data(), mto.data(), 1);
        }
        for (size_t k = 0; rc == GBN_OK && k < (one ? 1 : blocks.size()); ++k) {
            gbn_results_clear(res[cur]);
            rc = gbn_prelim_search_begin(batch, one ? view : blocks[k], res[cur], &d, interrupt_search ? s_Interrupt : NULL, &intr);
            if (rc != GBN_OK) break;
            if (in_flight) {                            // the search before this one: finished by now or soon, its lists go out
                rc = gbn_prelim_search_end(res[cur ^ 1]);
                if (rc == GBN_OK) rc = gbn_results_emit_lists(res[cur ^ 1], s_WriteList, &sink);
            }
            in_flight = true; cur ^= 1;
        }
    }
    if (in_flight) {                                    // (also after a failure: nothing may stay in flight on these results)
        const int rc2 = gbn_prelim_search_end(res[cur ^ 1]);
        if (rc == GBN_OK) rc = rc2;
        if (rc == GBN_OK) rc = gbn_results_emit_lists(res[cur ^ 1], s_WriteList, &sink);
    }
    gbn_results_free(res[0]); gbn_results_free(res[1]);
    BlastSeqSrcIteratorFree(itr);
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
    if (batch) gbn_batch_free(batch);                   // the derived search parameters are the callee's; the blocks stay cached
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
    s_leased_at_init = (int)s_free_gpus.size();
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
