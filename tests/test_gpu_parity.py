"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle."""
import numpy as np
import pytest
from gblastn_amd import api, synth
from tests import util

pytestmark = pytest.mark.gpu


def run_case(nsub, slen, nq, task="megablast", qlen=1000, expect=None, **kw):
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, qlen=qlen, task=task, **kw)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    info = ps.info()
    if expect:
        for k, v in expect.items():
            assert info[k] == v, (k, info)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    oi = s.info()
    assert (oi["lut_type"], oi["lut_width"], oi["scan_step"], oi["container"]) == \
           (info["lut_type"], info["lut_width"], info["scan_step"], info["container"])
    util.compare_stages(gpu, ora)
    d = ps.diagnostics
    assert d.lookup_hits == s.stats.lookup_hits
    assert d.good_init_extends == s.stats.good_init_extends
    assert d.gapped_extensions == s.stats.gapped_extensions
    assert d.good_extensions == s.stats.good_extensions
    nh = len(gpu["hsps"])
    return nh, plants


@pytest.fixture(params=["auto", "direct"], autouse=True)
def scan_variant(request, monkeypatch):
    """Every case runs with the engine's own choice of scan kernel (key-range partitioned
    for lut >= 10) and with the direct-probe kernel forced."""
    if request.param == "direct":
        monkeypatch.setenv("GBN_SCAN_BINS", "1")
    else:
        monkeypatch.delenv("GBN_SCAN_BINS", raising=False)


def test_megablast_small_query_smallna_lut8():
    # C1-shaped: one 1 kb query -> small table lut 8, stride 21, diag array
    nh, _ = run_case(6, 200_000, 1, planted_fraction=1.0,
                     expect=dict(lut_type=1, lut_width=8, scan_step=21, container=0))
    assert nh >= 1


def test_megablast_mb_lut11_diag_hash():
    # 16 x 1 kb -> 31,968 entries -> megablast table lut 11, stride 18, hash container
    nh, plants = run_case(8, 150_000, 16, expect=dict(lut_type=3, lut_width=11, scan_step=18, container=1))
    assert nh >= len(plants) // 2


def test_megablast_mb_lut12():
    # > 300,000 entries -> lut 12, stride 17 (the C2 table)
    nh, plants = run_case(6, 120_000, 160, expect=dict(lut_type=3, lut_width=12, scan_step=17, container=1))
    assert nh >= 1


@pytest.mark.parametrize("go,ge", [(2, 2), (1, 2), (0, 2), (3, 1), (1, 1)])
def test_megablast_affine_greedy(go, ge):
    # BLAST_AffineGreedyAlign: every affine gap cost of the 1/-2 table
    nh, plants = run_case(6, 100_000, 12, gap_open=go, gap_extend=ge)
    assert nh >= 1


def test_blastn_small_lut8_stride4():
    nh, _ = run_case(4, 60_000, 2, task="blastn", planted_fraction=1.0,
                     expect=dict(lut_type=1, lut_width=8, scan_step=4, container=0))
    assert nh >= 1


def test_blastn_mb_lut11_stride1():
    nh, _ = run_case(3, 40_000, 8, task="blastn",
                     expect=dict(lut_type=3, lut_width=11, scan_step=1, container=1))
    assert nh >= 1


def test_reference_known_answers_on_gpu():
    """The reference's own known answers through the HIP path."""
    import os
    from oracle import orc
    g = os.path.join(os.path.dirname(__file__), "golden")
    packed, n = orc.read_blastdb_v4_nucl(os.path.join(g, "nt.41646578"))[0]
    q = orc.unpack_ncbi2na(packed, n)[54:561].copy()
    src = api.BlastSeqSrc.from_packed([(packed, n)])
    ps = api.BlastPrelimSearch([q], api.default_options("megablast", db_length=n, db_num_seqs=1), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["context"] == 0
    assert (h[0]["q_offset"], h[0]["q_end"] - 1, h[0]["s_offset"], h[0]["s_end"] - 1) == (0, 506, 54, 560)
    a = orc.encode_blastna(orc.read_fasta(os.path.join(g, "greedy1a.fsa"))[0])
    b = orc.encode_blastna(orc.read_fasta(os.path.join(g, "greedy1b.fsa"))[0])
    src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(b), len(b))])
    ps = api.BlastPrelimSearch([a], api.default_options("megablast"), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["score"] == 619
    ps = api.BlastPrelimSearch([a], api.default_options("megablast", reward=10, penalty=-25,
                               xdrop_gap_bits=100.0, xdrop_gap_final_bits=100.0), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["score"] == 6034
