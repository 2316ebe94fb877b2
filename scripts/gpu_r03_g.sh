#!/bin/bash
# round 3, GPU session G: a real nested-sampling run on config 3 + long randomised sweeps of the new code paths
mkdir -p gpurun_out/r3g
O=gpurun_out/r3g
timeout 1500 python scripts/run_ns_c3.py --iterations 12000 > $O/ns_c3.txt 2>&1; tail -1 $O/ns_c3.txt | cut -c1-1500
timeout 900 python scripts/run_ns_c3.py --iterations 8000 --dims 16 --nlive 2000 --drawsize 2000 > $O/ns_d16_small.txt 2>&1; tail -1 $O/ns_d16_small.txt | cut -c1-900
timeout 1200 python scripts/fuzz_populate.py --iterations 200 --seed 21 > $O/fuzz_populate.txt 2>&1; tail -2 $O/fuzz_populate.txt | cut -c1-600
timeout 1200 python scripts/fuzz_parity.py --iterations 300 --seed 22 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt | cut -c1-600
timeout 900 python scripts/fuzz_precision.py --iterations 150 --wide --seed 23 --modes f16x3,f32 > $O/fuzz_precision.txt 2>&1; tail -1 $O/fuzz_precision.txt | cut -c1-600
timeout 600 python scripts/profile_c4_host.py > $O/c4_host.txt 2>&1; head -60 $O/c4_host.txt | cut -c1-160
