#!/bin/bash
# Diagnostic builds of the config-2 kernel instance (HT = 1, affine) with one ingredient removed each
# (NSA_ABLATE bits, csrc/nsa_flow_kernel.h) -> variants/libnessai_amd_c2a<bits>.so.  Run here (hipcc
# cross-compiles); time them on the GPU box with
#   for f in variants/libnessai_amd_c2a*.so; do NESSAI_AMD_LIBRARY=$f python scripts/bench_resident.py --geometry 4x1 --net resnet; done
set -e
cd "$(dirname "$0")/../nessai_amd/csrc"
mkdir -p ../../variants
for bits in ${@:-0 4 16 64 128 512 1024 2048}; do
  ( make -s -j2 BUILD=build_c2a$bits TARGET=../../variants/libnessai_amd_c2a$bits.so HTS=1 TRANSFORMS=2_0 \
      EXTRA="-DNSA_ABLATE=$bits '-DNSA_RES_GEOMETRIES=NSA_RES_GEO(4,1)'" > /dev/null 2>&1 && echo built c2a$bits ) &
done
wait
rm -rf build_c2a*
