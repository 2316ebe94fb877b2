#!/bin/bash
# round 3, GPU session F: what the driver runs at round end - the whole GPU suite, smoke(), the default bench
mkdir -p gpurun_out/r3f
O=gpurun_out/r3f
timeout 3000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -6 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
python bench.py --gpus 2 > $O/bench2.txt 2>&1; tail -2 $O/bench2.txt
