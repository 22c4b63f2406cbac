#!/bin/bash
# threshold-only refresh in the scan: parity tests on the production library, then the A/B on the experiments build
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize.py -m gpu -x -q -k "rowwise or scan or topk or exact or eval_topk or gather or fullsize or cfg2" > gpurun_out/r4c35_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r4c35_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
for r in 0 16 32 50 64 100; do
  echo "== refresh $r"
  ANNCUR_DEBUG_SCAN_REFRESH=$r timeout -k 10 300 python3 scripts/r4/scan_probe.py 2>&1 | grep -E "^bench|^iid" | grep -E "256 CUs|96 CUs"
done > gpurun_out/r4_scan_refresh.txt
cat gpurun_out/r4_scan_refresh.txt
