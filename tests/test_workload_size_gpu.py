"""-m gpu: oracle parity AT WORKLOAD SIZE with the engine's default thresholds (nothing forced by the environment).

C3 (BASELINE.json configs[2]): one full subject range of the blastn shape -- 1,000 x 1 Mb subjects on the device, one
100-query batch, W=11 (lut 11 = word size, stride 1): 4.7e7 seeds in one launch, i.e. scan_slice_kernel, the
composite-key sort (seed_ckeys_kernel + one radix sort), seed_ext_kernel + diag_replay_kernel, the lane / wave DP
kernels, and the host replay on up to 16 threads -- detached behind the stage's thread in the pipelined form.  Every
subject of the range goes through the oracle (0.04 s each).

C4 (configs[3]): three 5 Mb query batches streamed through the host pipeline against the 50 Gbp shard, traceback
overlapped; the final rows (coordinates, scores, e-value and bit-score bits, identities, edit scripts) of every
(query, subject) list of every subject that has one, plus sampled subjects that have none, against the oracle's
preliminary search + traceback of that subject.

Reference edges: CORE/na_ungapped.c:778-922 (hash container + ungapped extension), CORE/blast_gapalign.c:3351-3548
(acceptance loop), CORE/blast_traceback.c:336-790."""
import numpy as np
import pytest
from gblastn_amd import api, synth
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu

HSP_FIELDS = ["oid", "context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end", "s_gapped_start", "score"]


def device_db(nsub, slen, seed):
    import torch
    db = synth.SynthDb(nsub, slen, seed=seed)
    slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), db.nbytes), db.byte_off, db.lens, is_device=True, keep=slab)
    return db, src


def test_full_size_c3_range_with_default_thresholds_against_the_oracle(monkeypatch):
    for k in ("GBN_DIAG_COMPACT_MIN", "GBN_SEED_CKEYS", "GBN_SCAN_SLICE", "GBN_SCAN_BINS", "GBN_HOST_DETACH", "GBN_GAP_LANE",
              "GBN_RANGE_MIB", "GBN_RANGE_TILES", "GBN_RANGE_GIB"):
        monkeypatch.delenv(k, raising=False)
    nsub, slen, nq = 1000, 1_000_000, 100
    db, src = device_db(nsub, slen, 0x9E3779B97F4A7C15 ^ 3)
    queries, plants = synth.make_queries(nq, db)
    assert len(plants) >= 10
    # the statistics of the whole 5 Gbp database this range is a fifth of (global lengths: CORE/blast_setup.c:638)
    opt = api.default_options("blastn", db_length=5000 * slen, db_num_seqs=5000, word_size=11)
    ps = api.BlastPrelimSearch(queries, opt, src)
    info = ps.info()
    assert (info["lut_width"], info["scan_step"], info["container"], info["scan_path"]) == (11, 1, 1, 2), info

    first = ps.run()["hsps"]
    d = ps.diagnostics
    got_diag = (int(d.lookup_hits), int(d.seeds), int(d.init_extends), int(d.good_init_extends), int(d.gapped_extensions), int(d.good_extensions))
    assert int(d.scan_launches) == 1 and int(d.seeds) > (1 << 20)          # ONE launch far above the two-kernel threshold
    assert int(d.gapped_extensions) > 20000                                # ... and above the multi-thread replay's
    again = ps.run()["hsps"]
    assert first.tobytes() == again.tobytes()                              # deterministic
    # pipelined form: the range's gapped stage on the second stream, its host replay detached on a queue of its own
    ps.begin(); piped = ps.end()["hsps"]
    assert piped.tobytes() == first.tobytes()
    ps2 = api.BlastPrelimSearch(queries, opt, src)                         # a second batch scanned while the first one's stages run
    ps.begin(); ps2.begin()
    a = ps.end()["hsps"]; b = ps2.end()["hsps"]
    assert a.tobytes() == first.tobytes() and b.tobytes() == first.tobytes()

    s = orc.Search(util.oracle_options(opt), queries)
    # the reference's container semantics: ONE diagonal hash carried through the subjects in OID order (CORE/blast_extend.c:166-190,
    # CORE/na_ungapped.c:362-451); the HIP path starts every subject with a fresh one (DESIGN.md "a6", tests/test_diag_carry.py)
    s.carry_diag(True)
    oi = s.info()
    assert (oi["lut_type"], oi["lut_width"], oi["scan_step"], oi["container"]) == (info["lut_type"], 11, 1, 1)
    nseeds = nih = 0
    want = []
    for oid in range(nsub):
        o = s.subject(db.subject_packed(oid), slen)
        nseeds += len(o["seeds"]); nih += len(o["init_hits"])
        g = first[first["oid"] == oid]
        assert len(g) == len(o["hsps"]), (oid, len(g), len(o["hsps"]))
        for f in HSP_FIELDS[1:]:
            assert np.array_equal(g[f], o["hsps"][f]), (oid, f)
        assert np.array_equal(g["evalue"].view(np.uint64), o["hsps"]["evalue"].view(np.uint64)), oid
        want.append(len(o["hsps"]))
    st = s.stats
    assert got_diag == (int(st.lookup_hits), nseeds, nih, int(st.good_init_extends), int(st.gapped_extensions), int(st.good_extensions)), \
        (got_diag, (int(st.lookup_hits), nseeds, nih, int(st.good_init_extends), int(st.gapped_extensions), int(st.good_extensions)))
    assert sum(want) == len(first) and sum(1 for w in want if w) >= len(plants) * 0.9
    found = set(first["oid"].tolist())
    assert sum(1 for p in plants if p["subject"] in found) >= 0.9 * len(plants)


def test_full_size_c4_streamed_batches_final_rows_against_the_oracle(monkeypatch):
    for k in ("GBN_DIAG_COMPACT_MIN", "GBN_RANGE_MIB", "GBN_RANGE_TILES", "GBN_RANGE_GIB", "GBN_SCAN_BINS", "GBN_RECORD_CACHE_MB"):
        monkeypatch.delenv(k, raising=False)
    nsub, slen, per, nbatch = 50_000, 1_000_000, 5_000, 3
    db, src = device_db(nsub, slen, 0x9E3779B97F4A7C15 ^ 1)
    opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
    batches = []
    for k in range(nbatch):
        q, plants = synth.make_queries(per, db, first_query_id=k * per)
        batches.append((q, plants))
    pipe = api.SearchPipeline(opt, src, trace_threads=4, traceback=True, overlap=True)
    for k in range(nbatch):
        assert pipe.submit(batches[k][0]) == k
    pipe.finish()
    got = {}
    while True:
        r = pipe.next()
        if r is None:
            break
        k, (rec, ops, qstarts), dg = r
        assert k == len(got)                                    # submission order
        assert int(dg.subject_bases_scanned) == nsub * slen and int(dg.scan_launches) == 1
        got[k] = (rec, ops)
    pipe.close()
    assert len(got) == nbatch
    from tests.test_traceback_gpu import compare
    rng = np.random.default_rng(5)
    total = 0
    for k in range(nbatch):
        queries, plants = batches[k]
        rec, ops = got[k]
        prod = {}
        for r, o in zip(rec, ops):
            prod.setdefault((int(r["context"]) // 2, int(r["oid"])), []).append((r, o))
        hit = sorted(set(int(o) for o in rec["oid"]))
        assert len(set(p["subject"] for p in plants) & set(hit)) >= 0.9 * len(set(p["subject"] for p in plants))
        sample = sorted(set(hit) | set(rng.choice(nsub, 10, replace=False).tolist()) | set(p["subject"] for p in plants))
        s = orc.Search(util.oracle_options(opt), queries)
        ora = {}
        for oid in sample:
            packed = db.subject_packed(oid)
            o = s.subject(packed, slen)
            if not len(o["hsps"]):
                continue
            # (hit lists are far from full at this shape -- at most a few subjects per query against 550 kept -- so the
            # collector keeps every list; it is applied all the same)
            col = orc.Collector(len(queries), opt.hitlist_size)
            col.write(oid, [dict(zip(o["hsps"].dtype.names, x)) for x in o["hsps"]])
            bases = orc.unpack_ncbi2na(packed, slen)
            for _, q, hs in col.close():
                fin = s.traceback(bases, [dict(zip(orc.Collector.FIELDS, h)) for h in hs])
                if fin:
                    ora[(q, oid)] = fin
        total += compare(prod, ora)                             # same set of (query, subject) lists, same rows, same scripts
        s.close()
    assert total >= 0.15 * per * nbatch                         # ~20 % of the queries carry a planted homolog


def test_two_batch_c2_config_with_one_shared_binning_pass_gives_identical_results():
    """bench.py's config_wall_ms_measured: the two 5,000-query batches of the C2 config (10,000 x 1 kb vs the 50 Gbp
    shard) probed against ONE binning pass (the record cache, on by default: the scan records depend on the shard and the
    table shape only) -- at full size, the HSPs of both batches equal those of two full passes (GBN_RECORD_CACHE_MB=0), and
    the second batch runs no binning kernel."""
    import json, os, sys
    code = r'''
import sys, json, hashlib, numpy as np, torch
sys.path.insert(0, %r)
from gblastn_amd import api, synth
nsub, slen = 50000, 1000000
db = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 1)
slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), db.nbytes), db.byte_off, db.lens, is_device=True, keep=slab)
queries, plants = synth.make_queries(10000, db)
opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
out = []
for k in range(2):
    ps = api.BlastPrelimSearch(queries[k * 5000:(k + 1) * 5000], opt, src)
    assert (ps.info()["lut_width"], ps.info()["scan_step"]) == (12, 17)
    h = ps.run()["hsps"]
    out.append([hashlib.sha256(h.tobytes()).hexdigest(), int(len(h)), float(ps.diagnostics.bin_kernel_ms), float(ps.diagnostics.probe_kernel_ms)])
    ps.close()
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ)
        for k in ("GBN_SCAN_BINS", "GBN_RANGE_MIB", "GBN_RANGE_TILES", "GBN_RANGE_GIB", "GBN_RECORD_CACHE_MB"):
            env.pop(k, None)
        if flag == "0":
            env["GBN_RECORD_CACHE_MB"] = "0"
        p = util.run_child([sys.executable, "-c", code], env=env, timeout=900)
        res[flag] = json.loads(p.stdout.strip().splitlines()[-1])
    assert [r[:2] for r in res["0"]] == [r[:2] for r in res["1"]]
    assert res["0"][0][1] > 100 and res["0"][1][1] > 100           # both batches find their planted homologs
    assert res["1"][0][2] > 1.0 and res["1"][1][2] < 0.5           # shared: the second batch ran no binning kernel (an empty event interval)
    assert res["0"][1][2] > 1.0                                    # two full passes: it did


def test_skewed_shard_against_the_oracle(monkeypatch):
    """bench.py --skew at a size the oracle finishes: 1,500 x 1 Mb subjects with repeats written over them (gbn_synth_skew: 8 % of every
    subject homopolymer runs / tandem repeats -- lookup words pile up in a few bins, the range is halved and the repeat-rich parts go to the
    direct-probe kernel --, a family element in one subject of fifty), 600 queries of which 2 % carry a piece of the element (they hit every
    copy).  HSPs of every subject that has one, of every carrier of the element and of a sample of the others equal the oracle's, in
    stream form and over cached records."""
    for k in ("GBN_RANGE_MIB", "GBN_RANGE_TILES", "GBN_RANGE_GIB", "GBN_SCAN_BINS", "GBN_RECORD_CACHE_MB", "GBN_SKEW_SPLIT_MB"):
        monkeypatch.delenv(k, raising=False)
    import torch
    nsub, slen, nq = 1500, 1_000_000, 600
    db = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 5, skew=True)
    slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
    db.skew_on_device(api, slab.data_ptr())
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), db.nbytes), db.byte_off, db.lens, is_device=True, keep=slab)
    queries, plants = synth.make_queries(nq, db, family_fraction=0.02)
    opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
    # DUST as blastn applies it by default: a planted slice of a poly-A stretch would seed at every scan position of every poly-A stretch
    masks = api.dust_masks(queries)
    ps = api.BlastPrelimSearch(queries, opt, src, masks=masks)
    first = ps.run()["hsps"]
    d = ps.diagnostics
    again = ps.run()["hsps"]                                    # (over cached records where the ranges were cached)
    assert first.tobytes() == again.tobytes()
    hit = set(first["oid"].tolist())
    rng = np.random.default_rng(1)
    sample = sorted(hit | set(rng.choice(nsub, 20, replace=False).tolist()))
    assert len(hit) >= 30                                       # the element's carriers (one subject in fifty) are among them
    s = orc.Search(util.oracle_options(opt), queries, masks=masks)
    for oid in sample:
        o = s.subject(db.subject_packed(oid), slen)
        g = first[first["oid"] == oid]
        assert len(g) == len(o["hsps"]), (oid, len(g), len(o["hsps"]))
        for f in HSP_FIELDS[1:]:
            assert np.array_equal(g[f], o["hsps"][f]), (oid, f)
        assert np.array_equal(g["evalue"].view(np.uint64), o["hsps"]["evalue"].view(np.uint64)), oid
    print("skewed shard: %d ranges, %d rescans, %d direct-kernel ranges, %d library sorts, %d seeds, %d HSPs on %d subjects"
          % (d.ranges, d.scan_rescans, d.direct_ranges, d.library_sorts, d.seeds, len(first), len(hit)))
