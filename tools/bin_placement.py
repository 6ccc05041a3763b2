#!/usr/bin/env python
"""Does the binning kernel's time depend on where a run's allocations land?  One process, the C2 shard filled once; several times
over: the engine released (its record buffers and pool go back to the driver), a spacer tensor of another size allocated, the engine
set up again, six fresh-binning passes timed by the library's events.  usage: bin_placement.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gblastn_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = api.lib()
L.gbn_init(1, 0)
nsub, slen = 50000, 1_000_000
lay = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 1)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(L.gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
queries, _ = synth.make_queries(5000, lay)
opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
spacers = []
for r in range(rounds):
    api.record_cache_set_limit(0)
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
    ps = api.BlastPrelimSearch(queries, opt, src)
    ps.run()
    d0 = (ps.diagnostics.bin_kernel_ms, ps.diagnostics.probe_kernel_ms, ps.diagnostics.rare_kernel_ms, ps.diagnostics.scan_launches)
    for _ in range(6):
        ps.run()
    d1 = (ps.diagnostics.bin_kernel_ms, ps.diagnostics.probe_kernel_ms, ps.diagnostics.rare_kernel_ms, ps.diagnostics.scan_launches)
    n = d1[3] - d0[3]
    print("round %d: binning %.3f ms, probe %.3f, rare %.3f per pass (spacers held: %.1f GB)" % (
        r, (d1[0] - d0[0]) / n, (d1[1] - d0[1]) / n, (d1[2] - d0[2]) / n, sum(t.numel() for t in spacers) / 1e9), flush=True)
    ps.close(); src.close()
    L.gbn_release()
    if r % 2 == 1:                          # every other round the shard itself moves too
        del slab; torch.cuda.empty_cache()
        spacers.append(torch.empty(int(0.7e9 * (r + 1)), dtype=torch.uint8, device="cuda"))
        slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
        L.gbn_init(1, 0)
        api._check(L.gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
        print("  (shard moved to %#x)" % slab.data_ptr(), flush=True)
        L.gbn_release()
    spacers.append(torch.empty(int((1 + (r * 7) % 5) * 1.3e9), dtype=torch.uint8, device="cuda"))   # the next round's buffers land elsewhere
    L.gbn_init(1, 0)
