#!/bin/bash
cd $GRAFT_REPO_ROOT
GBN_DP_STATS=1 GBN_AMD_LIB=variants/libgblastn_amd_dps.so timeout 300 python bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-overlap 2>&1 | grep -v amdgpu.ids | grep "gbn dbg" | head -9
