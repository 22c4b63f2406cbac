#!/bin/bash
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4c52_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r4c52_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=1500 timeout -k 10 900 python3 -m pytest tests/test_gpu_random_shapes.py -m gpu -x -q -k "fused_score_topk_random" > gpurun_out/r4c52_fuzz.log 2>&1; rc=$?
echo "fresh fuzz rc=$rc"; tail -3 gpurun_out/r4c52_fuzz.log
if [ $rc -ne 0 ]; then exit 1; fi
