"""Shared helpers for the parity tests: run the HIP path (through the C ABI) and the
oracle on the same inputs and compare every stage bit for bit."""
import numpy as np
from oracle import orc
from gblastn_amd import api, synth

OPT_FIELDS = ["word_size", "reward", "penalty", "gap_open", "gap_extend", "greedy",
              "xdrop_ungap_bits", "gap_trigger_bits", "xdrop_gap_bits", "xdrop_gap_final_bits",
              "evalue", "min_diag_separation", "hitlist_size", "cutoff_score",
              "lut11_gblastn_rule", "db_length", "db_num_seqs"]


def oracle_options(gopt):
    o = orc.OrcOptions()
    for f in OPT_FIELDS:
        setattr(o, f, getattr(gopt, f))
    return o


def oracle_run(gopt, queries, subjects, masks=None):
    """subjects: list of (packed, length).  Returns per-oid dicts and the Search."""
    s = orc.Search(oracle_options(gopt), queries, masks=masks)
    out = []
    for packed, n in subjects:
        out.append(s.subject(packed, n))
    return out, s


def compare_stages(gpu, ora, first_oid=0, check_seeds=True):
    """gpu: dict from BlastPrelimSearch.run(keep_stages=True); ora: list per subject."""
    hs, ih, sd = gpu["hsps"], gpu.get("init_hits"), gpu.get("seeds")
    for i, o in enumerate(ora):
        oid = first_oid + i
        if check_seeds and sd is not None:
            g = sd[sd["oid"] == oid]
            assert len(g) == len(o["seeds"]), "seed count, oid %d: %d vs %d" % (oid, len(g), len(o["seeds"]))
            assert np.array_equal(g["q_off"], o["seeds"]["q_off"]), "seed q_off order, oid %d" % oid
            assert np.array_equal(g["s_off"], o["seeds"]["s_off"]), "seed s_off order, oid %d" % oid
        if ih is not None:
            g = ih[ih["oid"] == oid]
            assert len(g) == len(o["init_hits"]), "init-hit count, oid %d" % oid
            for f in ["q_off", "s_off", "q_start", "s_start", "length", "score"]:
                assert np.array_equal(g[f], o["init_hits"][f]), "init hit %s, oid %d" % (f, oid)
        g = hs[hs["oid"] == oid]
        assert len(g) == len(o["hsps"]), "HSP count, oid %d: %d vs %d" % (oid, len(g), len(o["hsps"]))
        for f in ["context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end",
                  "s_gapped_start", "score"]:
            assert np.array_equal(g[f], o["hsps"][f]), "HSP %s, oid %d" % (f, oid)
        # e-values bit-identical as IEEE doubles
        assert np.array_equal(g["evalue"].view(np.uint64), o["hsps"]["evalue"].view(np.uint64)), \
            "HSP evalue bits, oid %d" % oid


def small_case(nsub, slen, nq, qlen=1000, seed=7, planted_fraction=0.5, task="megablast", **optkw):
    db = synth.SynthDb(nsub, slen, seed=seed)
    queries, plants = synth.make_queries(nq, db, qlen=qlen, planted_fraction=planted_fraction)
    subjects = [(db.subject_packed(i), slen) for i in range(nsub)]
    opt = api.default_options(task, db_length=nsub * slen, db_num_seqs=nsub, **optkw)
    return db, queries, plants, subjects, opt


def run_child(cmd, env=None, timeout=600, cwd=None):
    """A child process of a test (a search under other environment settings), its stderr in the assertion message.
    A child the GPU runtime aborted with a queue error (HSA_STATUS_ERROR_*: seen about once in a hundred spawns on the
    test pool when three processes deep, never reproduced by tools/stress_ranges.py) is run once more, with a warning
    that stays in the pytest summary; a second abort, or any other failure, fails the test."""
    import subprocess, warnings
    for attempt in (0, 1):
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=cwd)
        if p.returncode < 0 and "HSA_STATUS_ERROR" in p.stderr and attempt == 0:
            warnings.warn("child process aborted by the GPU runtime, run once more; stdout ends %r, stderr ends %r" % (p.stdout[-200:], p.stderr[-600:]))
            continue
        break
    assert p.returncode == 0, (p.returncode, p.stdout[-3000:], p.stderr[-6000:])
    return p
