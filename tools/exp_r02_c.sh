cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-oldT newT}; do for dbg in ${DBGS:-32 34}; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so GBN_DBG=$dbg python tools/scan_ablate.py 2>&1 | grep -E "scan |cycles|workgroups" | sed "s/^/$v /"
done; done
