#!/bin/bash
# round 5: what an LDS operation class of the binning kernel costs IN the kernel (variants that issue one class twice, built
# by tools/build_variant.sh: see GBN_BIN_DUP in csrc/scan_bin.hip) next to the LDS side of the tile loop alone (tools/lds_microbench.hip)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_lds; mkdir -p $O
tools/bin/ldsmb > $O/microbench.txt 2>&1
export GBN_RECORD_CACHE_MB=0
for r in 1 2; do for v in base dupA dupR dupW dupS nortn nostore nostoreA nostoreN; do
  echo -n "$v: "; GBN_AMD_LIB=variants/libgblastn_amd_$v.so timeout 300 python tools/scan_ablate.py 2>/dev/null | tail -1
done; done > $O/insitu.txt 2>&1
cat $O/microbench.txt $O/insitu.txt
