#!/bin/bash
# Experiment builds of the config-3 kernel instance (HT = 2, RQ K = 8 only) with extra compiler flags:
#   scripts/build_variant.sh NAME "FLAGS" [NAME "FLAGS" ...]   -> variants/libnessai_amd_NAME.so
# (hipcc cross-compiles here; on the GPU box: NESSAI_AMD_LIBRARY=variants/libnessai_amd_NAME.so python scripts/bench_variants_r05.py)
set -e
cd "$(dirname "$0")/../nessai_amd/csrc"
mkdir -p ../../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( make -s -j3 BUILD=build_v_$name TARGET=../../variants/libnessai_amd_$name.so HTS=2 TRANSFORMS=1_8 "EXTRA=$flags" > /tmp/variant_$name.log 2>&1 && echo built $name || echo FAILED $name ) &
done
wait
