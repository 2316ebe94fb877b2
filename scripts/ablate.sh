#!/bin/bash
# Diagnostic builds of the config-3 kernel instance with one ingredient removed each
# (NSA_ABLATE bits, csrc/nsa_flow_kernel.h) -> variants/libnessai_amd_a<bits>.so.
# Run here (hipcc cross-compiles); time them on the GPU box with
#   for f in variants/*.so; do NESSAI_AMD_LIBRARY=$f python scripts/bench_kernel_variants.py --quick; done
set -e
cd "$(dirname "$0")/../nessai_amd/csrc"
mkdir -p ../../variants
for bits in ${@:-0 1 2 4 8 16 32}; do
  ( make -s -j2 BUILD=build_a$bits TARGET=../../variants/libnessai_amd_a$bits.so HTS=2 TRANSFORMS=1_8 EXTRA=-DNSA_ABLATE=$bits > /dev/null 2>&1 && echo built a$bits ) &
done
wait
