#!/bin/bash
# round 6, call c: chunk-pipeline ubench with the shader clock; the whole GPU suite on the refactored spline
OUT=gpurun_out/r06c
mkdir -p $OUT
timeout 600 scripts/ubench/chunk_pipeline 4096 > $OUT/chunk_pipeline.txt 2>&1; cat $OUT/chunk_pipeline.txt
timeout 3000 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
