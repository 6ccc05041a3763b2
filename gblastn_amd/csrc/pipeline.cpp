// pipeline.cpp -- C ABI of the host pipeline (include/gblastn_amd_host.hpp: set-up -> preliminary search ->
// traceback on their own threads), for callers that cannot include the C++ header (the ctypes test binding,
// bench.py --workload C4).  GB/work_thread.cpp:60-156 / APP/blastn_app.cpp:725-989 are the reference's shape.
namespace gbn { void trace_mark(const char *what); }
#define GBN_HOST_TRACE(what) gbn::trace_mark(what)      // the pipeline's hand-overs among the library's host marks (GBN_TRACE=1)
#include "../../include/gblastn_amd_host.hpp"
#include "gbn_guard.hpp"
#include "gbn_host.hpp"

struct GbnPipeline {
    gbn::CBlastSeqSrc src;
    std::unique_ptr<gbn::CSearchPipeline> pipe;
    gbn::CSearchPipeline::TItem current;        // what gbn_pipeline_next handed out last (kept until the next call)
    GbnPipeline(GbnDb *db) : src(db, false) {}
};

extern "C" {

int gbn_pipeline_new(GbnPipeline **out, const GbnOptions *opt, GbnDb *db, int32_t trace_threads, int with_traceback, int overlap)
{
    return gbn::guard(__func__, [&]() -> int {
    if (!out || !opt || !db) { gbn::set_error("gbn_pipeline_new: bad argument"); return GBN_ERR_ARG; }
    GbnPipeline *p = new GbnPipeline(db);
    p->pipe.reset(new gbn::CSearchPipeline(*opt, p->src, trace_threads, with_traceback != 0, overlap != 0));
    *out = p;
    return GBN_OK;
    });
}
void gbn_pipeline_free(GbnPipeline *p) { if (!p) return; p->pipe.reset(); p->current.reset(); delete p; }

int64_t gbn_pipeline_submit(GbnPipeline *p, int32_t nq, const uint8_t *const *seqs, const int32_t *lens,
                            int32_t nmask, const int32_t *mask_query, const int32_t *mask_from, const int32_t *mask_to)
{
    return gbn::guard_as<int64_t>(__func__, (int64_t)-1, (int64_t)-1, [&]() -> int64_t {
    if (!p || nq <= 0 || !seqs || !lens) { gbn::set_error("gbn_pipeline_submit: bad argument"); return -1; }
    gbn::CpuScope cpu(gbn::GBN_CPU_SUBMIT);
    gbn::SQueryBatch q;
    for (int32_t i = 0; i < nq; i++) q.seqs.emplace_back(seqs[i], seqs[i] + lens[i]);
    for (int32_t i = 0; i < nmask; i++) q.masks.push_back(gbn::SQueryBatch::Mask{mask_query[i], mask_from[i], mask_to[i]});
    return p->pipe->Submit(std::move(q));
    });
}
void gbn_pipeline_finish(GbnPipeline *p) { if (p) (void)gbn::guard(__func__, [&]() -> int { p->pipe->Finish(); return GBN_OK; }); }

// the next finished batch in submission order: *id = its number, *tb = its traceback results (null without the
// traceback stage), *col = the collector lists of its preliminary stage; valid until the next call.
// Returns 1 when no batch is left, a negative GBN_ERR_* if the batch failed.
int gbn_pipeline_next(GbnPipeline *p, int64_t *id, const GbnTraceback **tb, const GbnCollector **col)
{
    return gbn::guard(__func__, [&]() -> int {
    if (!p) return GBN_ERR_ARG;
    p->current = p->pipe->Next();
    if (!p->current) return 1;
    if (id) *id = p->current->id;
    if (p->current->status != GBN_OK) { gbn::set_error(p->current->error); return p->current->status; }
    if (tb) *tb = p->current->traceback ? p->current->traceback->Results() : nullptr;
    if (col) *col = p->current->stream ? p->current->stream->Get() : nullptr;
    return GBN_OK;
    });
}
// diagnostics of the batch gbn_pipeline_next handed out last
int gbn_pipeline_diagnostics(const GbnPipeline *p, GbnDiagnostics *d)
{
    return gbn::guard(__func__, [&]() -> int {
    if (!p || !d || !p->current || !p->current->prelim) return GBN_ERR_ARG;
    *d = p->current->prelim->diagnostics;
    return GBN_OK;
    });
}

}  // extern "C"
