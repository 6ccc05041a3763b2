#!/bin/bash
# Build a variant of libgblastn_amd.so with extra -D flags for same-box A/B timing:
#   tools/build_variant.sh NAME "-DGBN_REC_BLOCK_BITS=4"   ->  variants/libgblastn_amd_NAME.so
# (select it with GBN_AMD_LIB=variants/libgblastn_amd_NAME.so)
set -e
cd "$(dirname "$0")/../gblastn_amd/csrc"
NAME=$1; FLAGS=$2
OUT=../../variants; mkdir -p $OUT /tmp/var_$NAME
for f in kernels.hip seed_stage.hip scan_bin.hip scan_runs.hip seed_order.hip seed_sort.hip radix64.hip gapped.hip lutbuild.hip engine.cpp engine_scan.cpp engine_stages.cpp engine_abi.cpp batch.cpp stat.cpp hsp_host.cpp collector.cpp dbreader.cpp dust.cpp traceback.cpp pipeline.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $f -o /tmp/var_$NAME/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libgblastn_amd_$NAME.so /tmp/var_$NAME/*.o
echo built $OUT/libgblastn_amd_$NAME.so
