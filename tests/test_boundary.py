"""CPU: the drop-in boundary as compiled artefacts: the public headers are C99 and
self-contained, the Python mirrors of the launcher parameter blocks have the C layout, the shim translation unit
a maintainer adds to the reference is in the tree and takes word_length from the right structure per table kind."""
import ctypes as C
import os
import re
import subprocess
import pytest
from gblastn_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def test_public_headers_are_c99_and_self_contained(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "gblastn_amd_kernels.h"\n#include "gblastn_amd.h"\n'
                   "int main(void) { GbnScanParams s; GbnExtParams x; GbnGapParams g; GbnOptions o; GbnTbHSP t;\n"
                   "  (void)s; (void)x; (void)g; (void)o; (void)t; return 0; }\n")
    for first in ("gblastn_amd.h", "gblastn_amd_kernels.h"):        # either header alone, in either order
        one = tmp_path / ("one_" + first.replace(".", "_") + ".c")
        one.write_text('#include "%s"\nint main(void) { return 0; }\n' % first)
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", INC, "-fsyntax-only", str(one)], check=True)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", INC, "-fsyntax-only", str(src)], check=True)
    # no path out of include/
    for h in os.listdir(INC):
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(INC, h)).read()):
            assert os.path.exists(os.path.join(INC, inc)) and ".." not in inc, (h, inc)


def test_python_mirrors_have_the_c_layout(tmp_path):
    """sizeof / offsetof of every member, printed by a C program compiled against the header"""
    structs = {"GbnScanParams": api.GbnScanParams, "GbnExtParams": api.GbnExtParams, "GbnGapParams": api.GbnGapParams,
               "GbnOptions": api.GbnOptions, "GbnDiagnostics": api.GbnDiagnostics, "GbnContext": api.GbnContext}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gblastn_amd.h"', '#include "gblastn_amd_kernels.h"', "int main(void) {"]
    for name, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"; src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", INC, "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for f, _ in cls._fields_:
            assert int(got[name + "." + f]) == getattr(cls, f).offset, (name, f)
    import numpy as np
    for dt, n in ((api.TILE_DT, 16), (api.DEV_SEED_DT, 16), (api.DEV_IHIT_DT, 32), (api.DEV_GAPPED_DT, 32)):
        assert dt.itemsize == n


def test_shim_translation_unit_is_shipped():
    p = os.path.join(ROOT, "gblastn_amd", "shim", "gpu_blastn_amd_shim.cpp")
    txt = open(p).read()
    # word_length per table kind (COREI/blast_nalookup.h:63,132,237), not one cast for all
    for kind, struct in (("eMBLookupTable", "BlastMBLookupTable"), ("eSmallNaLookupTable", "BlastSmallNaLookupTable"),
                         ("eNaLookupTable", "BlastNaLookupTable")):
        assert re.search(r"case\s+%s\s*:.*?\(const %s\*\)\s*w->lut\)?.*?word_length" % (kind, struct), txt, flags=re.S), kind
    # every library entry point it calls is declared in the public header
    hdr = open(os.path.join(INC, "gblastn_amd.h")).read()
    for fn in set(re.findall(r"\b(gbn_\w+)\s*\(", txt)):
        assert re.search(r"\b%s\s*\(" % fn, hdr), fn
    assert "Blast_gpu_RunPreliminarySearchWithInterrupt" in txt and "Blast_RunPreliminarySearchWithInterrupt" in txt


REF_INC = "/root/reference/c++/include"
# what ./configure would have written into ncbiconf_unix.h for this platform (the reference tree ships none): macros on
# the command line of THIS check only, over an empty file of that name in the test's temporary directory
CONFIG_MACROS = ["NCBI_OS_UNIX=1", "NCBI_OS_LINUX=1", "NCBI_COMPILER_GCC=1", "HAVE_STRDUP=1", "HAVE_INTTYPES_H=1", "HAVE_STDINT_H=1",
                 "HAVE_SYS_TYPES_H=1", "SIZEOF_CHAR=1", "SIZEOF_SHORT=2", "SIZEOF_INT=4", "SIZEOF_LONG=8", "SIZEOF_LONG_LONG=8",
                 "SIZEOF_VOIDP=8", "SIZEOF___INT64=0", "SIZEOF_FLOAT=4", "SIZEOF_DOUBLE=8", "SIZEOF_LONG_DOUBLE=16", "SIZEOF_SIZE_T=8",
                 "NCBI_PLATFORM_BITS=64"]
# include/algo/blast/gpu_blast/gpu_blastn.h:31-51 -- its three prototypes without the includes of the work-thread
# classes above them (those pull in the toolkit's generated object headers, which exist only in a configured build)
GPU_BLASTN_H = """#ifndef __GPU_BLAST_H__
#define __GPU_BLAST_H__
#include <algo/blast/core/blast_hspstream.h>
#include <algo/blast/core/blast_engine.h>
Int2 Blast_gpu_RunPreliminarySearchWithInterrupt(EBlastProgramType program, BLAST_SequenceBlk* query,
    BlastQueryInfo* query_info, const BlastSeqSrc* seq_src, const BlastScoringOptions* score_options, BlastScoreBlk* sbp,
    LookupTableWrap* lookup_wrap, const BlastInitialWordOptions* word_options, const BlastExtensionOptions* ext_options,
    const BlastHitSavingOptions* hit_options, const BlastEffectiveLengthsOptions* eff_len_options,
    const PSIBlastOptions* psi_options, const BlastDatabaseOptions* db_options, const BlastGPUOptions* gpu_options,
    BlastHSPStream* hsp_stream, BlastDiagnostics* diagnostics, TInterruptFnPtr interrupt_search, SBlastProgress* progress_info);
int Blast_gpu_Init(bool isInit, int gpu_id);
void Blast_gpu_Release();
#endif
"""


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference tree is not on this machine")
def test_shim_compiles_against_the_reference_headers_with_the_reference_linkage(tmp_path):
    """The shim is the one file that binds to the reference interface: it must compile against the reference's own
    core headers (blast_engine.h, blast_hspstream.h, blast_seqsrc.h, blast_nalookup.h, blast_hits.h, blast_util.h,
    blast_diagnostics.h, gpu_blastn_na_ungapped_v3.h as they lie in /root/reference) and DEFINE the four entry points
    with the C++ linkage those headers declare.  Pins no parity; stops the shim from rotting."""
    (tmp_path / "ncbiconf_unix.h").write_text("")
    d = tmp_path / "algo" / "blast" / "gpu_blast"
    d.mkdir(parents=True)
    (d / "gpu_blastn.h").write_text(GPU_BLASTN_H)
    obj = tmp_path / "shim.o"
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-c", "-o", str(obj), "-I", str(tmp_path), "-I", REF_INC, "-I", INC]
    cmd += ["-D" + m for m in CONFIG_MACROS] + [os.path.join(ROOT, "gblastn_amd", "shim", "gpu_blastn_amd_shim.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    syms = subprocess.run(["nm", str(obj)], capture_output=True, text=True, check=True).stdout.split("\n")
    defined = [l.split()[-1] for l in syms if " T " in l]
    undefined = [l.split()[-1] for l in syms if l.strip().startswith("U ")]
    # Itanium-mangled = C++ linkage, as src/app/blast/blastn_app.cpp:462-491 and prelim_search_runner.hpp:96 will look them up
    for want in ("_Z14Blast_gpu_Initbi", "_Z17Blast_gpu_Releasev", "_Z19gpu_ReleaseDBMemoryv"):
        assert want in defined, (want, defined)
    assert any(s.startswith("_Z43Blast_gpu_RunPreliminarySearchWithInterrupt") for s in defined), defined
    # everything it takes from this repository is a C symbol the library exports
    L = api.lib()
    ours = [s for s in undefined if s.startswith("gbn_")]
    assert len(ours) >= 10
    for s in ours:
        assert hasattr(L, s), s


def test_shard_builder_explicit_oids_host_checks():
    L = api.lib()
    sb = C.c_void_p()
    assert L.gbn_shard_builder_new(C.byref(sb), 4) == 0
    import numpy as np
    a = np.zeros(8, dtype=np.uint8)
    assert L.gbn_shard_builder_add_oid(sb, 5, a.ctypes.data, 20) == 0
    assert L.gbn_shard_builder_add_oid(sb, 5, a.ctypes.data, 20) != 0     # OIDs ascend
    assert L.gbn_shard_builder_add_oid(sb, 3, a.ctypes.data, 20) != 0
    assert L.gbn_shard_builder_add_oid(sb, 9, a.ctypes.data, 20) == 0
    L.gbn_shard_builder_free(sb)
    sb = C.c_void_p()
    assert L.gbn_shard_builder_new(C.byref(sb), 4) == 0
    assert L.gbn_shard_builder_add(sb, a.ctypes.data, 20) == 0
    assert L.gbn_shard_builder_add_oid(sb, 7, a.ctypes.data, 20) != 0     # not after a subject without one
    L.gbn_shard_builder_free(sb)


def test_shard_builder_rejects_bad_input():
    L = api.lib()
    sb = C.c_void_p()
    assert L.gbn_shard_builder_new(C.byref(sb), 4) == 0
    assert L.gbn_shard_builder_add(sb, None, 10) != 0
    out = C.c_void_p()
    assert L.gbn_shard_builder_finish(sb, C.byref(out)) != 0       # no subjects
    L.gbn_shard_builder_free(sb)
    assert L.gbn_db_cache_find(C.c_void_p(12345)) is None
    L.gbn_release_db_memory()                                          # empty cache: nothing to do


def test_no_exception_crosses_the_c_abi(tmp_path):
    """"no exceptions may cross the C boundary".  tests/firewall_probe.cpp replaces the process's operator new
    by one that throws std::bad_alloc at the k-th allocation and calls gbn_batch_new_ex (host set-up), gbn_pipeline_new,
    the collector, the shard builder and the database reader for k = 1, 2, ... until each gets through: every failure
    must come back as GBN_ERR_NOMEM with a text in gbn_last_error() -- an exception unwinding through an extern "C" frame
    would end the process instead (csrc/gbn_guard.hpp wraps every status-returning entry point)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "firewall_probe")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), "-o", exe,
                    os.path.join(root, "tests", "firewall_probe.cpp"), "-L" + os.path.join(root, "gblastn_amd"),
                    "-lgblastn_amd", "-Wl,-rpath," + os.path.join(root, "gblastn_amd")], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "firewall probe ok" in p.stdout
    for name in ("gbn_batch_new_ex", "gbn_pipeline_new", "gbn_collector_new", "gbn_shard_builder_new", "gbn_blastdb_open"):
        assert name in p.stdout


def test_every_status_returning_entry_point_runs_behind_the_firewall():
    """Source check: every multi-line `int gbn_*(...)` definition of the C ABI translation units hands its body to
    gbn::guard / gbn::guard_as."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for f in ("engine_abi.cpp", "engine.cpp", "collector.cpp", "dbreader.cpp", "traceback.cpp", "pipeline.cpp", "dust.cpp"):
        lines = open(os.path.join(root, "gblastn_amd", "csrc", f)).read().split("\n")
        for i, line in enumerate(lines):
            m = re.match(r'^(?:extern "C" )?(int|int64_t|int32_t|long) (gbn_\w+)\(', line)
            if not m or line.rstrip().endswith("}"):
                continue
            j = i
            while not lines[j].rstrip().endswith("{"):
                j += 1
            if "gbn::guard" not in lines[j + 1]:
                missing.append((f, m.group(2)))
    assert not missing, missing
