#!/bin/bash
# scan kernel with buffer loads + signed prefilter: parity tests, then the probes
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py -m gpu -x -q -k "rowwise or scan or topk or exact or eval_topk or gather" > gpurun_out/r4c28_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r4c28_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python3 scripts/r4/scan_probe.py 2>&1 | grep -v amdgpu.ids | grep -v ascending > gpurun_out/r4_scan_probe2.txt; cat gpurun_out/r4_scan_probe2.txt
timeout -k 10 300 python3 scripts/r4/scan_fixed_probe.py 2>&1 | grep -v amdgpu.ids | grep "k=100" > gpurun_out/r4_scan_fixed2.txt; cat gpurun_out/r4_scan_fixed2.txt
