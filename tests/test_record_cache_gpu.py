"""The record cache ("bin once, probe many", the engine's default policy) and the views over resident blocks
(the shim's loop over OID chunks as one search), through the C ABI.

Reference behaviour matched: the per-OID device cache that is never evicted while the process lives
(GB/gpu_blastn_MB_and_smallNa.cu:1461-1468) and the chunk loop of the preliminary search
(GB/gpu_blastn_pre_search_engine.cpp:1243-1441)."""
import ctypes as C
import os

import numpy as np
import pytest

from gblastn_amd import api
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture
def cache_default(monkeypatch):
    """the library's default policy, whatever the environment of the run says; the limit is put back afterwards"""
    monkeypatch.delenv("GBN_RECORD_CACHE_MB", raising=False)
    monkeypatch.delenv("GBN_SCAN_BINS", raising=False)
    api.record_cache_set_limit(-1)
    yield
    api.record_cache_set_limit(-1)


def _shapes_case():
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 40, task="megablast", seed=11)
    return subjects, opt, queries[:18], queries[18:36], queries[36:40]      # 18 kb, 18 kb (one table shape), 4 kb (another)


def test_default_policy_bins_once_and_probes_many(cache_default):
    subjects, opt, qa, qb, qc = _shapes_case()
    src = api.BlastSeqSrc.from_packed(subjects)
    # what every batch gives with the cache off
    api.record_cache_set_limit(0)
    want = {}
    for name, q in (("a", qa), ("b", qb), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src)
        assert ps.info()["scan_path"] == 0
        want[name] = ps.run()["hsps"].tobytes()
        assert ps.diagnostics.bin_kernel_ms > 0
        ps.close()
    assert api.record_cache_stats()["sets"] == 0
    api.record_cache_set_limit(-1)
    st0 = api.record_cache_stats()
    assert st0["limit"] > (1 << 30)
    binned = []
    for name, q in (("a", qa), ("b", qb), ("a", qa), ("c", qc), ("b", qb), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src)
        m0 = api.record_cache_stats()["misses"]
        assert ps.run()["hsps"].tobytes() == want[name], name
        binned.append(api.record_cache_stats()["misses"] - m0 == 1)
        ps.close()
    st = api.record_cache_stats()
    # a bins, b and a probe a's records; c (another shape) bins, b and c probe
    assert binned == [True, False, False, True, False, False], binned
    assert st["misses"] - st0["misses"] == 2 and st["hits"] - st0["hits"] == 4 and st["sets"] == 2 and st["bytes"] > 0
    # the pipelined entry points, the same way
    prev = None
    for name, q in (("b", qb), ("c", qc), ("a", qa)):
        ps = api.BlastPrelimSearch(q, opt, src)
        ps.begin()
        if prev is not None:
            assert prev[1].end()["hsps"].tobytes() == want[prev[0]]
            prev[1].close()
        prev = (name, ps)
    assert prev[1].end()["hsps"].tobytes() == want[prev[0]]
    prev[1].close()
    assert api.record_cache_stats()["hits"] - st["hits"] == 3
    # forgotten records are binned again, into the buffers that are there
    api.record_cache_invalidate()
    b0 = api.record_cache_stats()
    ps = api.BlastPrelimSearch(qa, opt, src)
    assert ps.run()["hsps"].tobytes() == want["a"]
    assert ps.run()["hsps"].tobytes() == want["a"]
    ps.close()
    b1 = api.record_cache_stats()
    # (a set that had been SORTED gave its streams back to the pool when it was: its stream buffers come out of the pool again;
    # a set still in stream form bins into the buffers it kept)
    assert (b1["misses"] - b0["misses"], b1["hits"] - b0["hits"], b1["sets"]) == (1, 1, b0["sets"]) and b1["bytes"] >= b0["bytes"]
    # the shard goes: its records go with it
    src.close()
    assert api.record_cache_stats()["sets"] == 0


def test_shape_changes_mid_stream_and_eviction_under_a_byte_cap(cache_default):
    subjects, opt, qa, qb, qc = _shapes_case()
    src = api.BlastSeqSrc.from_packed(subjects)
    api.record_cache_set_limit(0)
    want = {}
    for name, q in (("a", qa), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src)
        want[name] = ps.run()["hsps"].tobytes(); ps.close()
    # how large the two sets are
    api.record_cache_set_limit(-1)
    sizes = {}
    for name, q in (("a", qa), ("c", qc)):
        b0 = api.record_cache_stats()["bytes"]
        ps = api.BlastPrelimSearch(q, opt, src); ps.run(); ps.close()
        sizes[name] = api.record_cache_stats()["bytes"] - b0
    assert sizes["a"] > 0 and sizes["c"] > 0
    # a cap that holds either set but not both: every change of shape evicts the other set and bins again
    cap = max(sizes.values()) + min(sizes.values()) // 2
    api.record_cache_set_limit(cap)
    st0 = api.record_cache_stats()
    assert st0["sets"] == 1 and st0["bytes"] <= cap             # (lowering the limit evicted the least recently used set)
    for name, q in (("a", qa), ("c", qc), ("a", qa), ("a", qa), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src)
        assert ps.run()["hsps"].tobytes() == want[name], name
        ps.close()
        st = api.record_cache_stats()
        assert st["sets"] == 1 and st["bytes"] <= cap
    st = api.record_cache_stats()
    assert st["evictions"] - st0["evictions"] >= 3 and st["hits"] - st0["hits"] >= 1
    # a cap smaller than any set: every pass bins into its own scratch (bypass), nothing is kept
    api.record_cache_set_limit(4096)
    st1 = api.record_cache_stats()
    for name, q in (("a", qa), ("c", qc), ("a", qa)):
        ps = api.BlastPrelimSearch(q, opt, src)
        assert ps.run()["hsps"].tobytes() == want[name], name
        ps.close()
    st2 = api.record_cache_stats()
    assert st2["sets"] == 0 and st2["bypassed"] - st1["bypassed"] == 3
    src.close()


def test_rare_queue_overflow_rescans_against_cached_records(cache_default, monkeypatch):
    """A rare-path segment that overflows makes the range scan again with more room (GBN_RARE_SEG=n forces it on a small
    search): with the cache on the second attempt probes the records the first attempt binned."""
    subjects, opt, qa, qb, qc = _shapes_case()
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(qa, opt, src)
    want = ps.run()["hsps"].tobytes()
    ps.close(); src.close()
    src = api.BlastSeqSrc.from_packed(subjects)
    monkeypatch.setenv("GBN_RARE_SEG", "1")
    ps = api.BlastPrelimSearch(qa, opt, src)
    st0 = api.record_cache_stats()
    assert ps.run()["hsps"].tobytes() == want
    st = api.record_cache_stats()
    assert ps.diagnostics.scan_launches >= 2 and st["misses"] - st0["misses"] == 1
    ps.close(); src.close()


def _blocks_case(nsub=24, slen=120_000, nq=30, nblocks=6, seed=21):
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, task="megablast", seed=seed)
    per = nsub // nblocks
    blocks = [api.BlastSeqSrc.from_oids(subjects[k * per:(k + 1) * per], range(k * per, (k + 1) * per)) for k in range(nblocks)]
    return subjects, queries, opt, blocks


def test_view_over_blocks_equals_the_whole_shard(cache_default):
    subjects, queries, opt, blocks = _blocks_case()
    whole = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, whole)
    want = ps.run(keep_stages=True)
    ora, _ = util.oracle_run(opt, queries, subjects)
    util.compare_stages(want, ora)
    assert len(want["hsps"]) > 5
    view = api.block_view(blocks[::-1])                         # any order: the view sorts by OID
    assert view.num_seqs == len(subjects) and view.total_bases == whole.total_bases
    got = ps.run(seqsrc=view, keep_stages=True)
    for k in ("hsps", "seeds", "init_hits"):
        assert got[k].tobytes() == want[k].tobytes(), k
    # cached by its blocks; one block is the block itself
    assert api.block_view(blocks)._h.value == view._h.value
    assert api.block_view(blocks[2:3])._h.value == blocks[2]._h.value
    # a second batch over the view probes the view's cached records
    st0 = api.record_cache_stats()
    ps2 = api.BlastPrelimSearch(queries[::-1], opt, view)
    h2 = ps2.run()["hsps"]
    assert api.record_cache_stats()["hits"] - st0["hits"] == 1
    ps3 = api.BlastPrelimSearch(queries[::-1], opt, whole)
    assert ps3.run()["hsps"].tobytes() == h2.tobytes()
    # blastn shape (slice scan, many seeds) through the same view
    optn = api.default_options("blastn", db_length=opt.db_length, db_num_seqs=opt.db_num_seqs)
    pn = api.BlastPrelimSearch(queries[:6], optn, whole)
    wn = pn.run()["hsps"]
    assert pn.run(seqsrc=view)["hsps"].tobytes() == wn.tobytes() and len(wn) > 0
    # a block goes: the views over it go first; a new view over the rest works
    blocks[5].close()
    v2 = api.block_view(blocks[:5])
    assert v2._h.value != view._h.value or True
    got2 = ps.run(seqsrc=v2)["hsps"]
    keep = want["hsps"][want["hsps"]["oid"] < 20]
    assert got2.tobytes() == keep.tobytes()
    for p in (ps, ps2, ps3, pn):
        p.close()
    L = api.lib()
    bad = (C.c_void_p * 2)(blocks[0]._h, blocks[0]._h)
    out = C.c_void_p()
    assert L.gbn_block_view(bad, 2, C.byref(out)) != 0          # a block twice
    assert L.gbn_block_view(None, 2, C.byref(out)) != 0


@pytest.mark.parametrize("group", [1, 2, 6])
def test_shim_shaped_loop_equals_the_whole_shard_search(cache_default, group):
    """The shim's loop through the C ABI (gblastn_amd/shim/gpu_blastn_amd_shim.cpp): chunks of OIDs -> resident blocks ->
    groups of `group` blocks as one view -> gbn_prelim_search_begin of group k, gbn_prelim_search_end + gbn_results_emit_lists
    of group k - 1.  The lists, in the order they reach the sink, equal the lists of ONE search over the whole shard."""
    subjects, queries, opt, blocks = _blocks_case()
    whole = api.BlastSeqSrc.from_packed(subjects)
    ref = api.BlastPrelimSearch(queries, opt, whole)
    ref.run()
    want = ref.emit_lists()
    assert len(want) > 3 and [o for o, _ in want] == sorted(o for o, _ in want)
    for rounds in range(2):                                     # cold records, then cached ones
        a, b = api.BlastPrelimSearch(queries, opt), api.BlastPrelimSearch(queries, opt)
        b.close()                                               # one batch, two result handles: the shim's res[2]
        L = api.lib()
        res = [C.c_void_p(), C.c_void_p()]
        for r in res:
            api._check(L.gbn_results_new(C.byref(r)))
        got = []

        @api.GbnHspListFn
        def sink(arg, oid, ptr, n):
            got.append((int(oid), np.frombuffer(C.string_at(ptr, n * api.HSP_DT.itemsize), dtype=api.HSP_DT).copy()))
            return 0
        d = api.GbnDiagnostics()
        cur, in_flight = 0, False
        for g0 in range(0, len(blocks), group):
            view = api.block_view(blocks[g0:g0 + group])
            L.gbn_results_clear(res[cur])
            api._check(L.gbn_prelim_search_begin(a._b, view._h, res[cur], C.byref(d), None, None))
            if in_flight:
                api._check(L.gbn_prelim_search_end(res[cur ^ 1]))
                api._check(L.gbn_results_emit_lists(res[cur ^ 1], sink, None))
            in_flight = True; cur ^= 1
        api._check(L.gbn_prelim_search_end(res[cur ^ 1]))
        api._check(L.gbn_results_emit_lists(res[cur ^ 1], sink, None))
        assert [o for o, _ in got] == [o for o, _ in want]
        for (_, x), (_, y) in zip(got, want):
            assert x.tobytes() == y.tobytes()
        assert int(d.subject_bases_scanned) == whole.total_bases
        for r in res:
            L.gbn_results_free(r)
        a.close()
    ref.close()


def test_sweeps_over_more_ranges_than_the_cache_holds_still_hit(cache_default):
    """Every query batch sweeps the views of its database in the same order.  With room for two of three record sets a
    least-recently-used cache would never hit (it evicts the set the sweep needs next); the cache evicts the MOST recently
    used set of the sweep instead, so the sets from the start of a sweep are found again by the next one."""
    subjects, queries, opt, blocks = _blocks_case()
    views = [api.block_view(blocks[k:k + 2]) for k in (0, 2, 4)]
    ps = api.BlastPrelimSearch(queries, opt)
    api.record_cache_set_limit(0)
    want = [ps.run(seqsrc=v)["hsps"].tobytes() for v in views]
    assert sum(len(w) for w in want) > 0
    api.record_cache_set_limit(-1)
    ps.run(seqsrc=views[0])
    one = api.record_cache_stats()["bytes"]
    assert one > 0
    api.record_cache_set_limit(2 * one + one // 2)
    st0 = api.record_cache_stats()
    for sweep in range(4):
        for v, w in zip(views, want):
            assert ps.run(seqsrc=v)["hsps"].tobytes() == w
            st = api.record_cache_stats()
            assert st["sets"] <= 2 and st["bytes"] <= 2 * one + one // 2
    st = api.record_cache_stats()
    assert st["hits"] - st0["hits"] >= 4, (st, st0)             # (least recently used first: 1, the very first pass)
    assert st["evictions"] - st0["evictions"] >= 3
    ps.close()


def test_records_prepared_ahead_of_the_batch(cache_default):
    """gbn_db_prepare_records queues the binning kernel for the table shape a batch of these query lengths gets -- before the
    batch exists; the batch's pass finds the set (still being written on the engine's stream) and probes it.  The predicted
    shape is the batch's shape for megablast batches of several sizes; a blastn batch whose table is as wide as the word
    prepares nothing; a shard freed right behind the call takes its half-written records with it."""
    subjects, opt, qa, qb, qc = _shapes_case()
    src = api.BlastSeqSrc.from_packed(subjects)
    api.record_cache_set_limit(0)
    want = {}
    for name, q in (("a", qa), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src); want[name] = ps.run()["hsps"].tobytes(); ps.close()
    api.record_cache_set_limit(-1)
    for name, q in (("a", qa), ("c", qc), ("a", qa)):
        api.record_cache_invalidate()
        st0 = api.record_cache_stats()
        src.prepare_records(opt, q)
        st1 = api.record_cache_stats()
        assert st1["prepared"] - st0["prepared"] == 1
        src.prepare_records(opt, q)                                 # queued already: nothing more
        assert api.record_cache_stats()["prepared"] == st1["prepared"]
        ps = api.BlastPrelimSearch(q, opt, src)
        assert ps.run()["hsps"].tobytes() == want[name]
        assert ps.run()["hsps"].tobytes() == want[name]
        st2 = api.record_cache_stats()
        assert (st2["misses"] - st0["misses"], st2["hits"] - st0["hits"]) == (0, 2), (name, st0, st2)
        ps.close()
    optn = api.default_options("blastn", db_length=opt.db_length, db_num_seqs=opt.db_num_seqs)
    p0 = api.record_cache_stats()["prepared"]
    src.prepare_records(optn, qa)                                   # lut 11 = word 11: the slice scan keeps no records
    assert api.record_cache_stats()["prepared"] == p0
    api.record_cache_invalidate()
    src.prepare_records(opt, qa)
    src.close()                                                     # the kernel may still be writing
    assert api.record_cache_stats()["sets"] == 0
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(qa, opt, src)
    assert ps.run()["hsps"].tobytes() == want["a"]
    ps.close(); src.close()


def test_a_full_device_gives_cached_records_up(cache_default):
    """the cache is on by default and used to hold on to its sets while the allocation of a new shard failed.  A 2 Gbp
    shard is searched (its record set stays resident: ~1 GB), the rest of the device is taken away, and a second shard that does
    not fit next to the records is uploaded: the set goes (an eviction), the upload succeeds, and both shards still give their
    results (the first one bins again)."""
    import torch
    from gblastn_amd import synth
    nsub, slen = 2000, 1_000_000
    db = synth.SynthDb(nsub, slen, seed=77)
    slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), db.nbytes), db.byte_off, db.lens, is_device=True, keep=slab)
    queries, plants = synth.make_queries(400, db)
    opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
    ps = api.BlastPrelimSearch(queries, opt, src)
    want = ps.run()["hsps"].tobytes()
    st0 = api.record_cache_stats()
    assert st0["sets"] >= 1 and st0["bytes"] > 500_000_000
    # a second shard, on the host for now
    small = util.small_case(40, 10_000_000, 4, task="megablast", seed=3)
    subjects2 = small[3]
    need = sum(len(p) for p, n in subjects2)                         # ~100 MB
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    hog = torch.empty(free - need // 2, dtype=torch.uint8, device="cuda")      # less than the second shard needs is left
    src2 = api.BlastSeqSrc.from_packed(subjects2)                   # hipMalloc fails, the cached set goes, the second attempt succeeds
    st1 = api.record_cache_stats()
    assert st1["evictions"] > st0["evictions"] and st1["bytes"] < st0["bytes"]
    del hog
    torch.cuda.empty_cache()
    assert ps.run()["hsps"].tobytes() == want                       # (bins again)
    ps2 = api.BlastPrelimSearch(small[1], small[4], src2)
    assert len(ps2.run()["hsps"]) >= 1
    ps.close(); ps2.close(); src.close(); src2.close()
