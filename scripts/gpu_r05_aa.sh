#!/bin/bash
OUT=gpurun_out/r05aa
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_importance.py -m gpu -q > $OUT/imp.log 2>&1; tail -3 $OUT/imp.log
