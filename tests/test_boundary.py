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


def test_shard_builder_rejects_bad_input():
    L = api.lib()
    sb = C.c_void_p()
    assert L.gbn_shard_builder_new(C.byref(sb), 4) == 0
    assert L.gbn_shard_builder_add(sb, None, 10) != 0
    out = C.c_void_p()
    assert L.gbn_shard_builder_finish(sb, C.byref(out)) != 0       # no subjects
    L.gbn_shard_builder_free(sb)
    assert L.gbn_db_cache_find(C.c_void_p(12345)) is None
    L.gpu_ReleaseDBMemory()                                          # empty cache: nothing to do
