#!/bin/bash
# round 4, session D: where the resident C2 kernel's cycles go (timeline build), instruction rates, parity
cd "$(dirname "$0")/.."
o=gpurun_out/r4d
mkdir -p $o
scripts/ubench/valu_rates > $o/valu_rates.txt 2>&1
NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl.so python scripts/timeline_c2.py 4x1 3x2 2x2 > $o/timeline_c2.jsonl 2> $o/timeline_err.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -15 > $o/test_resident.txt
cat $o/valu_rates.txt $o/timeline_c2.jsonl $o/test_resident.txt; tail -n 5 $o/timeline_err.txt
