// firewall_probe.cpp -- test program of tests/test_boundary.py: no exception leaves the C ABI.
// The executable replaces the global operator new (an executable's definition is the one the whole process uses,
// the library included) by one that throws std::bad_alloc at the k-th allocation after it was armed; every entry
// point below is called with k = 1, 2, ... until it gets through unharmed.  An exception that crossed an extern "C"
// frame would end the process (std::terminate); what must come back is a status and a text in gbn_last_error().
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "gblastn_amd.h"

static thread_local long g_countdown = -1;      // allocations left before the one that fails (-1: not armed)
static thread_local long g_failed = 0;
static void *alloc_or_throw(std::size_t n)
{
    if (g_countdown >= 0 && g_countdown-- == 0) { g_failed++; throw std::bad_alloc(); }
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(std::size_t n) { return alloc_or_throw(n); }
void *operator new[](std::size_t n) { return alloc_or_throw(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

template <class F> static int sweep(const char *name, int ok_status, F &&call)
{
    int nomem = 0, k;
    for (k = 0; k < 100000; k++) {
        const long before = g_failed;
        g_countdown = k;
        const int rc = call();
        g_countdown = -1;
        if (g_failed == before) {                   // the call made fewer than k + 1 allocations: it ran unharmed
            if (rc != ok_status) { std::printf("%s: unharmed call returned %d (%s)\n", name, rc, gbn_last_error()); return 1; }
            break;
        }
        if (rc == GBN_ERR_NOMEM) {
            if (!std::strstr(gbn_last_error(), "memory")) { std::printf("%s: GBN_ERR_NOMEM without a text: '%s'\n", name, gbn_last_error()); return 1; }
            nomem++;
        } else if (rc == ok_status) {
            // an allocation failure the callee absorbed (e.g. a worker thread's, reported elsewhere): allowed
        } else { std::printf("%s: allocation %d failed -> status %d (%s)\n", name, k, rc, gbn_last_error()); return 1; }
    }
    std::printf("%s: %d allocation failures came back as GBN_ERR_NOMEM, the call needs %d allocations\n", name, nomem, k);
    return nomem > 0 ? 0 : 1;
}

int main()
{
    int bad = 0;
    // query batch, host-only set-up (no device needed): concatenation, contexts, Karlin-Altschul blocks, lookup table
    std::vector<uint8_t> q(1500);
    unsigned x = 12345;
    for (auto &b : q) { x = x * 1664525u + 1013904223u; b = (uint8_t)((x >> 24) & 3); }
    const uint8_t *seqs[2] = {q.data(), q.data() + 700};
    const int32_t lens[2] = {700, 800};
    GbnOptions opt; gbn_default_options(&opt, 1);
    opt.db_length = 1000000; opt.db_num_seqs = 10;
    bad += sweep("gbn_batch_new_ex (host set-up)", GBN_OK, [&]() -> int {
        GbnBatch *b = nullptr;
        const int rc = gbn_batch_new_ex(&b, &opt, 2, seqs, lens, 0);
        const long keep = g_countdown; g_countdown = -1;
        if (b) gbn_batch_free(b);
        g_countdown = keep;
        return rc; });
    // the pipeline object: its first allocation is the object itself; the shard pointer is only stored
    bad += sweep("gbn_pipeline_new (first allocations)", GBN_ERR_NOMEM, [&]() -> int {
        GbnPipeline *p = nullptr;
        if (g_countdown > 1) return GBN_ERR_NOMEM;          // (only the object and its first member: no device work)
        const int rc = gbn_pipeline_new(&p, &opt, reinterpret_cast<GbnDb *>(&opt), 1, 0, 0);
        g_countdown = -1;
        if (p) gbn_pipeline_free(p);
        return rc; });
    // collector and shard builder
    bad += sweep("gbn_collector_new + write + close", GBN_OK, [&]() -> int {
        GbnCollector *c = nullptr;
        int rc = gbn_collector_new(&c, 4, 10);
        if (rc == GBN_OK) {
            GbnHSP h[3]; std::memset(h, 0, sizeof(h));
            for (int i = 0; i < 3; i++) { h[i].context = 2 * i; h[i].oid = 5; h[i].score = 50 + i; h[i].q_end = 20; h[i].s_end = 20; }
            rc = gbn_collector_write(c, h, 3);
            if (rc == GBN_OK) rc = gbn_collector_close(c);
        }
        const long keep = g_countdown; g_countdown = -1;
        if (c) gbn_collector_free(c);
        g_countdown = keep;
        return rc; });
    bad += sweep("gbn_shard_builder_new + add", GBN_OK, [&]() -> int {
        GbnShardBuilder *sb = nullptr;
        int rc = gbn_shard_builder_new(&sb, 4);
        if (rc == GBN_OK) rc = gbn_shard_builder_add(sb, q.data(), 4000);
        const long keep = g_countdown; g_countdown = -1;
        if (sb) gbn_shard_builder_free(sb);
        g_countdown = keep;
        return rc; });
    bad += sweep("gbn_blastdb_open (missing file)", GBN_ERR_ARG, [&]() -> int {
        GbnBlastDb *db = nullptr;
        const int rc = gbn_blastdb_open(&db, "/nonexistent/database");
        const long keep = g_countdown; g_countdown = -1;
        if (db) gbn_blastdb_close(db);
        g_countdown = keep;
        return rc == GBN_OK ? GBN_OK : (rc == GBN_ERR_NOMEM ? rc : GBN_ERR_ARG); });
    std::printf(bad ? "FIREWALL PROBE FAILED\n" : "firewall probe ok\n");
    return bad;
}
