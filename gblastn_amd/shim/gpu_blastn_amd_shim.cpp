// gpu_blastn_amd_shim.cpp -- the ONE translation unit a maintainer of the reference adds to
// src/algo/blast/gpu_blast/ to put libgblastn_amd.so behind G-BLASTN's own entry point
// (include/algo/blast/gpu_blast/gpu_blastn.h:31-51).  It is compiled inside the configured toolkit tree
// (it needs the toolkit's generated ncbiconf_unix.h, which does not exist outside a configured build), linked
// with -lgblastn_amd -lamdhip64 in place of -lgpublastn -lcudart (src/app/blast/Makefile.blastn.app:16).
//
// Everything that does not need a toolkit type lives in the library and is tested there:
//   gbn_prelim_search_lists    grouping of the HSPs into per-subject lists, ascending oid  (tests/test_boundary_gpu.py)
//   gbn_db_cache_find/_insert  the per-BlastSeqSrc shard cache that gpu_ReleaseDBMemory() empties
//   gbn_shard_builder_*        subjects appended one by one into a 16-byte aligned slab
// so what remains here is field-by-field translation between the toolkit's structures and the PODs of
// gblastn_amd.h.  Anything the library does not do (discontiguous templates, programs other than blastn, PHI /
// RPS, a nucleotide table it does not know) goes to the stock CPU function, as the reference itself does for
// non-blastn programs (API/prelim_stage.cpp:226-262).
#include <algo/blast/gpu_blast/gpu_blastn.h>
#include <algo/blast/core/blast_engine.h>
#include <algo/blast/core/blast_hspstream.h>
#include <algo/blast/core/blast_seqsrc.h>
#include <algo/blast/core/blast_nalookup.h>
#include <algo/blast/core/blast_hits.h>
#include <vector>
#include "gblastn_amd.h"

namespace {

// word_length / lut_word_length live in a different structure for every nucleotide table kind
// (COREI/blast_nalookup.h:63, 132, 237); BlastNaWordFinder's callers switch on lut_type the same way
// (CORE/na_ungapped.c:1753-1795)
bool s_TableShape(const LookupTableWrap* w, Int4* word, bool* discontiguous)
{
    *discontiguous = false;
    switch (w->lut_type) {
    case eMBLookupTable: {
        const BlastMBLookupTable* t = (const BlastMBLookupTable*)w->lut;
        *word = t->word_length; *discontiguous = t->discontiguous != 0; return true; }
    case eSmallNaLookupTable:
        *word = ((const BlastSmallNaLookupTable*)w->lut)->word_length; return true;
    case eNaLookupTable:
        *word = ((const BlastNaLookupTable*)w->lut)->word_length; return true;
    default:
        return false;
    }
}

// one resident shard per BlastSeqSrc, built on first use: every OID in NCBI2na (eBlastEncodingProtein is the
// toolkit's name for "as stored" = 2 bits per base for nucleotide databases, CORE/blast_engine.c:1043-1050)
GbnDb* s_GetShard(const BlastSeqSrc* seq_src)
{
    if (GbnDb* db = gbn_db_cache_find(seq_src)) return db;
    const Int4 n = BlastSeqSrcGetNumSeqs(seq_src);
    GbnShardBuilder* sb = NULL;
    if (gbn_shard_builder_new(&sb, n) != GBN_OK) return NULL;
    BlastSeqSrcGetSeqArg arg;
    memset(&arg, 0, sizeof arg);
    arg.encoding = eBlastEncodingProtein;
    int rc = GBN_OK;
    for (Int4 oid = 0; oid < n && rc == GBN_OK; ++oid) {
        arg.oid = oid;
        if (BlastSeqSrcGetSequence(seq_src, &arg) < 0) { rc = GBN_ERR_ARG; break; }
        rc = gbn_shard_builder_add(sb, arg.seq->sequence, arg.seq->length);
        BlastSeqSrcReleaseSequence(seq_src, &arg);
    }
    if (arg.seq) BlastSequenceBlkFree(arg.seq);
    GbnDb* db = NULL;
    if (rc == GBN_OK) rc = gbn_shard_builder_finish(sb, &db);      // uploads the slab, frees the builder's host copy
    gbn_shard_builder_free(sb);
    if (rc != GBN_OK) return NULL;
    if (gbn_db_cache_insert(seq_src, db) != GBN_OK) { gbn_db_free(db); return gbn_db_cache_find(seq_src); }
    return db;
}

struct SListSink { BlastHSPStream* stream; const BlastQueryInfo* query_info; };

// GbnHspListFn: one BlastHSPList per subject -> BlastHSPStreamWrite (the stream takes the list and nulls the pointer)
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

extern "C" Int2 Blast_gpu_RunPreliminarySearchWithInterrupt(EBlastProgramType program,
        BLAST_SequenceBlk* query, BlastQueryInfo* query_info, const BlastSeqSrc* seq_src,
        const BlastScoringOptions* score_options, BlastScoreBlk* sbp, LookupTableWrap* lookup_wrap,
        const BlastInitialWordOptions* word_options, const BlastExtensionOptions* ext_options,
        const BlastHitSavingOptions* hit_options, const BlastEffectiveLengthsOptions* eff_len_options,
        const PSIBlastOptions* psi_options, const BlastDatabaseOptions* db_options, const BlastGPUOptions* gpu_options,
        BlastHSPStream* hsp_stream, BlastDiagnostics* diagnostics,
        TInterruptFnPtr interrupt_search, SBlastProgress* progress_info)
{
    Int4 word = 0; bool discontiguous = false;
    const bool ours = program == eBlastTypeBlastn && gpu_options && gpu_options->use_gpu &&
                      lookup_wrap && s_TableShape(lookup_wrap, &word, &discontiguous) && !discontiguous;
    if (!ours)                                          // everything else: the stock CPU path
        return Blast_RunPreliminarySearchWithInterrupt(program, query, query_info, seq_src, score_options,
                   sbp, lookup_wrap, word_options, ext_options, hit_options, eff_len_options, psi_options, db_options,
                   hsp_stream, diagnostics, interrupt_search, progress_info);

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
    o.db_length = eff_len_options->db_length ? eff_len_options->db_length : BlastSeqSrcGetTotLen(seq_src);
    o.db_num_seqs = eff_len_options->dbseq_num ? eff_len_options->dbseq_num : BlastSeqSrcGetNumSeqs(seq_src);

    // queries: the concatenated BLASTNA buffer already has the layout the engine wants; hand over the plus
    // strand of every query (contexts 0, 2, 4, ...).  Masks: see INTEGRATION.md "Query masks" -- a maintainer
    // who keeps the lookup_segments passes them to gbn_batch_new_masked here.
    std::vector<const Uint1*> seqs; std::vector<Int4> lens;
    for (Int4 c = query_info->first_context; c <= query_info->last_context; c += 2) {
        seqs.push_back(query->sequence + query_info->contexts[c].query_offset);
        lens.push_back(query_info->contexts[c].query_length);
    }
    GbnDb* shard = s_GetShard(seq_src);
    if (!shard) return -1;
    GbnBatch* batch = NULL; GbnDiagnostics d; memset(&d, 0, sizeof d);
    SListSink sink = { hsp_stream, query_info };
    int rc = gbn_batch_new(&batch, &o, (int32_t)seqs.size(), seqs.data(), lens.data());
    if (rc == GBN_OK)
        rc = gbn_prelim_search_lists(batch, shard, s_WriteList, &sink, &d, (GbnInterruptFn)interrupt_search, progress_info);
    if (rc == GBN_OK && diagnostics) {
        if (diagnostics->ungapped_stat) {
            diagnostics->ungapped_stat->lookup_hits += d.lookup_hits;
            diagnostics->ungapped_stat->init_extends += (Int4)d.init_extends;
            diagnostics->ungapped_stat->good_init_extends += (Int4)d.good_init_extends;
        }
        if (diagnostics->gapped_stat) {
            diagnostics->gapped_stat->extensions += (Int4)d.gapped_extensions;
            diagnostics->gapped_stat->good_extensions += (Int4)d.good_extensions;
            diagnostics->gapped_stat->seqs_passed += (Int4)d.seqs_passed;
        }
    }
    gbn_batch_free(batch);                              // the derived search parameters are the callee's; the shard stays cached
    return rc == GBN_OK ? 0 : (rc == GBN_ERR_INTERRUPTED ? BLASTERR_INTERRUPTED : -1);
}
// Blast_gpu_Init, Blast_gpu_Release and gpu_ReleaseDBMemory keep their reference names and are exported by
// libgblastn_amd.so itself: CBlastnApp::Run (src/app/blast/blastn_app.cpp:462) needs no change.
