#!/bin/bash
# full GPU suite on the final kernels + fresh fuzz draws of the properties that touch the kernels changed late in the round
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4c38_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r4c38_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=900 timeout -k 10 900 python3 -m pytest tests/test_gpu_random_shapes.py -m gpu -x -q -k "eval_fused_random or rowwise_topk_random or approx_error or ivf" > gpurun_out/r4c38_fuzz.log 2>&1; rc=$?
echo "fresh fuzz rc=$rc"; tail -3 gpurun_out/r4c38_fuzz.log
if [ $rc -ne 0 ]; then exit 1; fi
