// gblastn_amd_host.hpp -- the C++ host side above the C ABI of gblastn_amd.h (header only; links against
// libgblastn_amd.so): the call surface of the reference's host classes for this path, so that a caller of
//   CBlastPrelimSearch::Run            (API/prelim_stage.cpp:192-308, API/prelim_search_runner.hpp:66-117)
//   CBlastTracebackSearch::Run         (API/traceback_stage.cpp:198-306)
//   the three-stage thread pipeline    (GB/work_thread.cpp:60-156, GBI/thread_work_queue.hpp:110-162,
//                                       APP/blastn_app.cpp:725-989 "Method2")
// finds the same steps under the same names.  The command line (gblastn_amd/cli/blastn_prelim.cpp) is written on
// these classes; Python (gblastn_amd/api.py) is only the test binding of the same C ABI.
#pragma once
#include "gblastn_amd.h"
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gbn {

struct CBlastException : std::runtime_error {
    int code;
    CBlastException(int rc, const std::string &what) : std::runtime_error(what + ": " + gbn_last_error()), code(rc) {}
};
inline void Check(int rc, const char *what) { if (rc != GBN_OK) throw CBlastException(rc, what); }

// BlastSeqSrc of a resident shard (API/seqsrc_seqdb.cpp): owns the handle unless told otherwise
class CBlastSeqSrc {
    GbnDb *db_ = nullptr; bool own_ = true;
public:
    CBlastSeqSrc() {}
    explicit CBlastSeqSrc(GbnDb *db, bool own = true) : db_(db), own_(own) {}
    CBlastSeqSrc(const CBlastSeqSrc &) = delete; CBlastSeqSrc &operator=(const CBlastSeqSrc &) = delete;
    CBlastSeqSrc(CBlastSeqSrc &&o) noexcept : db_(o.db_), own_(o.own_) { o.db_ = nullptr; }
    ~CBlastSeqSrc() { if (db_ && own_) gbn_db_free(db_); }
    GbnDb *Get() const { return db_; }
    int64_t GetTotLen() const { return gbn_db_total_bases(db_); }
    int32_t GetNumSeqs() const { return gbn_db_num_seqs(db_); }
};

// one query batch: BLASTNA sequences (plus strands) and their soft masks
struct SQueryBatch {
    std::vector<std::vector<uint8_t>> seqs;
    struct Mask { int32_t query, from, to; };
    std::vector<Mask> masks;
};

// Preliminary stage of one query batch against one shard (CBlastPrelimSearch)
class CBlastPrelimSearch {
    GbnBatch *b_ = nullptr; GbnResults *r_ = nullptr; GbnDb *db_; int32_t nq_;
public:
    GbnDiagnostics diagnostics;
    CBlastPrelimSearch(const SQueryBatch &q, const GbnOptions &opt, const CBlastSeqSrc &src) : db_(src.Get()), nq_((int32_t)q.seqs.size()) {
        std::memset(&diagnostics, 0, sizeof(diagnostics));
        std::vector<const uint8_t *> p; std::vector<int32_t> len, mq, mf, mt;
        for (auto &s : q.seqs) { p.push_back(s.data()); len.push_back((int32_t)s.size()); }
        for (auto &m : q.masks) { mq.push_back(m.query); mf.push_back(m.from); mt.push_back(m.to); }
        Check(gbn_use_device(gbn_db_device(db_)), "gbn_use_device");        // the batch lives on the shard's GPU, whatever thread sets it up
        (void)gbn_db_prepare_records(db_, &opt, nq_, len.data());          // a shard whose scan records are not resident yet: binned underneath this set-up (returns at once)
        Check(gbn_batch_new_masked(&b_, &opt, nq_, p.data(), len.data(), (int32_t)mq.size(), mq.data(), mf.data(), mt.data(), 1), "gbn_batch_new_masked");
        Check(gbn_results_new(&r_), "gbn_results_new");
    }
    CBlastPrelimSearch(const CBlastPrelimSearch &) = delete; CBlastPrelimSearch &operator=(const CBlastPrelimSearch &) = delete;
    ~CBlastPrelimSearch() { if (b_) gbn_batch_free(b_); if (r_) gbn_results_free(r_); }
    // whole stage; the HSP lists are then in Results()
    void Run() { Check(gbn_prelim_search(b_, db_, r_, &diagnostics, 0, nullptr, nullptr), "gbn_prelim_search"); }
    // pipelined: Begin returns when only this batch's extension stages are in flight; End waits for them
    void Begin() { Check(gbn_prelim_search_begin(b_, db_, r_, &diagnostics, nullptr, nullptr), "gbn_prelim_search_begin"); }
    void End() { Check(gbn_prelim_search_end(r_), "gbn_prelim_search_end"); }
    GbnBatch *Batch() const { return b_; }
    GbnResults *Results() const { return r_; }
    int32_t NumQueries() const { return nq_; }
};

// what the HSP stream holds after the preliminary stage: per query the best subjects (BlastHSPStream + collector)
class CBlastHSPStream {
    GbnCollector *c_ = nullptr;
public:
    CBlastHSPStream(int32_t num_queries, int32_t hitlist_size) { Check(gbn_collector_new(&c_, num_queries, hitlist_size), "gbn_collector_new"); }
    CBlastHSPStream(const CBlastHSPStream &) = delete; CBlastHSPStream &operator=(const CBlastHSPStream &) = delete;
    ~CBlastHSPStream() { if (c_) gbn_collector_free(c_); }
    void Write(const GbnResults *r) { Check(gbn_collector_write(c_, gbn_results_hsps(r), gbn_results_num_hsps(r)), "gbn_collector_write"); }
    void Close() { Check(gbn_collector_close(c_), "gbn_collector_close"); }
    GbnCollector *Get() const { return c_; }
};

// Traceback stage of one query batch (CBlastTracebackSearch)
class CBlastTracebackSearch {
    GbnTraceback *t_ = nullptr;
public:
    CBlastTracebackSearch() { Check(gbn_traceback_new(&t_), "gbn_traceback_new"); }
    CBlastTracebackSearch(const CBlastTracebackSearch &) = delete; CBlastTracebackSearch &operator=(const CBlastTracebackSearch &) = delete;
    ~CBlastTracebackSearch() { if (t_) gbn_traceback_free(t_); }
    void Run(const CBlastPrelimSearch &prelim, const CBlastSeqSrc &src, const CBlastHSPStream &stream, int threads) {
        const GbnCollector *c = stream.Get();
        Check(gbn_traceback_run(prelim.Batch(), src.Get(), gbn_collector_hsps(c), gbn_collector_list_starts(c), gbn_collector_num_lists(c), threads, t_), "gbn_traceback_run");
    }
    const GbnTraceback *Results() const { return t_; }
};

// ---------------------------------------------------------------------------------------------------
// Query batches through set-up -> preliminary search -> traceback on their own host threads: while the
// GPU scans batch k+1, its extension stages finish batch k and the traceback threads align batch k-1
// (the reference's query_queue -> prelim_queue -> result_queue with PrelimSearchThread / TraceBackThread).
// Results come out in submission order.
// ---------------------------------------------------------------------------------------------------
#ifndef GBN_HOST_TRACE
#define GBN_HOST_TRACE(what) ((void)0)          // (csrc/pipeline.cpp: the library's host marks, GBN_TRACE=1)
#endif
class CSearchPipeline {
public:
    struct SWorkItem {                                  // GBI/thread_work_queue.hpp work_item
        int64_t id = 0; SQueryBatch queries;
        std::unique_ptr<CBlastPrelimSearch> prelim; std::unique_ptr<CBlastHSPStream> stream; std::unique_ptr<CBlastTracebackSearch> traceback;
        std::string error; int status = GBN_OK;
        bool closed = false;                            // gbn_prelim_search_end called, the lists through the collector
    };
    typedef std::unique_ptr<SWorkItem> TItem;

    CSearchPipeline(const GbnOptions &opt, const CBlastSeqSrc &src, int trace_threads, bool with_traceback, bool overlap)
        : opt_(opt), src_(src), traceback_(with_traceback), overlap_(overlap)
    {
        for (int i = 0; i < kSetupThreads; i++) setup_.emplace_back([this] { SetupThread(); });
        prelim_ = std::thread([this] { PrelimSearchThread(); });
        // two batches in the traceback stage at a time, each on half of the traceback threads
        inner_threads_ = std::max(1, (trace_threads + 1) / 2);
        for (int i = 0; i < (trace_threads > 1 ? 2 : 1); i++) trace_.emplace_back([this] { TraceBackThread(); });
    }
    ~CSearchPipeline() { Close(); }
    // a query batch enters the pipeline; returns its number
    int64_t Submit(SQueryBatch q) {
        TItem it(new SWorkItem()); it->queries = std::move(q);
        std::unique_lock<std::mutex> lk(mu_);
        it->id = submitted_++;
        GBN_HOST_TRACE("pipeline: batch submitted");
        query_queue_.push_back(std::move(it));
        cv_.notify_all();
        return submitted_ - 1;
    }
    void Finish() { std::unique_lock<std::mutex> lk(mu_); no_more_ = true; cv_.notify_all(); }
    // the next finished batch in submission order (empty pointer: none left after Finish)
    TItem Next() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return done_.count(delivered_) || (no_more_ && delivered_ >= submitted_); });
        auto f = done_.find(delivered_);
        if (f == done_.end()) return TItem();
        TItem it = std::move(f->second); done_.erase(f); delivered_++;
        cv_.notify_all();
        return it;
    }
    void Close() {
        { std::unique_lock<std::mutex> lk(mu_); no_more_ = true; closing_ = true; cv_.notify_all(); }
        for (auto &t : setup_) if (t.joinable()) t.join();
        setup_.clear();
        if (prelim_.joinable()) prelim_.join();
        for (auto &t : trace_) if (t.joinable()) t.join();
        trace_.clear();
    }
private:
    GbnOptions opt_; const CBlastSeqSrc &src_; bool traceback_, overlap_;
    std::mutex mu_; std::condition_variable cv_;
    int inner_threads_ = 1;
    static const int kSetupThreads = 2;                  // a 5 Mb batch takes longer to set up than the GPU takes to scan it
    std::deque<TItem> query_queue_, trace_queue_; std::map<int64_t, TItem> prelim_queue_, done_;   // prelim_queue_: set up, by number
    int64_t submitted_ = 0, delivered_ = 0, next_search_ = 0; int in_setup_ = 0, setup_exited_ = 0;
    bool no_more_ = false, closing_ = false, setup_done_ = false, prelim_done_ = false;
    std::thread prelim_; std::vector<std::thread> setup_, trace_;

    void Deliver(TItem it) { GBN_HOST_TRACE("pipeline: batch delivered"); std::unique_lock<std::mutex> lk(mu_); done_[it->id] = std::move(it); cv_.notify_all(); }
    static void Guard(SWorkItem &it, const std::function<void()> &f) {
        if (it.status != GBN_OK) return;
        try { f(); }
        catch (const CBlastException &e) { it.status = e.code; it.error = e.what(); }
        catch (const std::exception &e) { it.status = GBN_ERR_NOMEM; it.error = e.what(); }      // (std::bad_alloc in a worker thread is the item's status, not std::terminate)
    }
    // set-up of a batch (host part + lookup structures on the device) on two threads, at most three batches ahead of the search
    void SetupThread() {
        gbn_set_setup_threads(std::max(2, (int)gbn_host_cpus() / 2));      // (the traceback threads want the other half of the CPUs granted)
        for (;;) {
            TItem it;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return closing_ || (!query_queue_.empty() && (int)prelim_queue_.size() + in_setup_ < 3) || (no_more_ && query_queue_.empty()); });
                if (closing_ || query_queue_.empty()) { if (++setup_exited_ == kSetupThreads) setup_done_ = true; cv_.notify_all(); return; }
                it = std::move(query_queue_.front()); query_queue_.pop_front(); in_setup_++;
            }
            GBN_HOST_TRACE("pipeline: set-up thread takes a batch");
            Guard(*it, [&] { it->prelim.reset(new CBlastPrelimSearch(it->queries, opt_, src_)); });
            std::unique_lock<std::mutex> lk(mu_);
            in_setup_--;
            const int64_t id = it->id;
            prelim_queue_[id] = std::move(it);
            cv_.notify_all();
        }
    }
    void CloseStream(SWorkItem &it) {
        Guard(it, [&] {
            it.prelim->End();
            it.stream.reset(new CBlastHSPStream(it.prelim->NumQueries(), opt_.hitlist_size));
            it.stream->Write(it.prelim->Results()); it.stream->Close();
        });
    }
    void PrelimSearchThread() {
        TItem prev;
        for (;;) {
            TItem it;
            {
                std::unique_lock<std::mutex> lk(mu_);
                // nothing to scan next: the batch in flight is finished now instead of underneath a scan
                if (!prelim_queue_.count(next_search_) && in_setup_ == 0 && query_queue_.empty() && prev && !closing_ && !setup_done_) {
                    lk.unlock(); PushTrace(std::move(prev)); lk.lock();
                }
                cv_.wait(lk, [&] { return closing_ || prelim_queue_.count(next_search_) || setup_done_; });
                auto f = prelim_queue_.find(next_search_);
                if (f != prelim_queue_.end()) { it = std::move(f->second); prelim_queue_.erase(f); next_search_++; cv_.notify_all(); }
                else if (closing_ || setup_done_) break;
            }
            if (!it) continue;
            Guard(*it, [&] { it->prelim->Begin(); });                        // the scan of this batch; the stages of `prev` finish underneath
            // (round 6: `prev` is finished -- gbn_prelim_search_end + its lists through the collector -- by the traceback thread that takes
            // it, not here: this thread goes straight to the next batch's scan, as bench.py's loop hands its passes to a merger thread;
            // 0.4 ms per 5 Mb batch that the GPU used to idle)
            if (prev) PushTrace(std::move(prev));
            if (!overlap_) { CloseStream(*it); it->closed = true; PushTrace(std::move(it)); }
            else prev = std::move(it);
        }
        if (prev) PushTrace(std::move(prev));
        std::unique_lock<std::mutex> lk(mu_); prelim_done_ = true; cv_.notify_all();
    }
    void PushTrace(TItem it) { std::unique_lock<std::mutex> lk(mu_); trace_queue_.push_back(std::move(it)); cv_.notify_all(); }
    void TraceBackThread() {
        for (;;) {
            TItem it;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !trace_queue_.empty() || prelim_done_ || closing_; });
                if (!trace_queue_.empty()) { it = std::move(trace_queue_.front()); trace_queue_.pop_front(); }
                else if (prelim_done_ || closing_) return;
            }
            if (!it) continue;
            GBN_HOST_TRACE("pipeline: traceback thread takes a batch");
            if (!it->closed) { CloseStream(*it); it->closed = true; }
            GBN_HOST_TRACE("pipeline: batch closed (end + collector)");
            if (traceback_) Guard(*it, [&] { it->traceback.reset(new CBlastTracebackSearch()); it->traceback->Run(*it->prelim, src_, *it->stream, inner_threads_); });
            Deliver(std::move(it));
        }
    }
};

}  // namespace gbn
