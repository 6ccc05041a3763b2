#!/bin/bash
# Round 6: presence-filtered binning (VERDICT r05 Next #2), measured in situ.  Variants (tools/build_variant.sh):
#   base   the library as it is
#   pres2  -DGBN_BIN_PRESENCE=2: the binning kernel looks every scan position's cell up in the batch's presence bits (2 MB,
#          non-temporal subject loads untouched) and still writes every record: what the lookups cost where they run
#   pres1  -DGBN_BIN_PRESENCE=1: ... and writes only the positions whose cell is occupied (results unchanged: parity below)
# all with the record cache off and nothing binned ahead (the filtered records belong to their batch).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GBN_BIN_AHEAD=0
echo "== parity of the filtered variant (oracle): tests/test_gpu_parity.py, lut 12 / 11 / 8 shapes, ragged subjects"
GBN_AMD_LIB=variants/libgblastn_amd_pres1.so GBN_RECORD_CACHE_MB=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lut12 or lut11 or lut8 or ragged or option_sweep" 2>&1 | tail -3
echo "== variant ms_per_step scan_stage [bin probe rare] in the pipeline, [bin probe rare] alone"
STEPS=${STEPS:-20} bash tools/abv.sh "base pres2 pres1" ${1:-2}
