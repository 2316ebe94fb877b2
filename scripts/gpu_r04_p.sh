#!/bin/bash
# end-of-round check of the final library: the whole GPU suite, then bench.py as the driver runs it
mkdir -p gpurun_out/r4p
python -m pytest tests -m gpu -x -q > gpurun_out/r4p/test_all.txt 2>&1
tail -n 3 gpurun_out/r4p/test_all.txt
python bench.py 2>/dev/null | tail -n 1 | tee gpurun_out/r4p/bench.json | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
