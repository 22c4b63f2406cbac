#!/bin/bash
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py tests/test_gpu_entrypoints.py -m gpu -x -q -k "error or approx_error or entry_point_A or eval_fused" > gpurun_out/r4c37_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r4c37_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so timeout -k 10 300 python3 scripts/r4/evalf_modes_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "passes\|no filter" | tee gpurun_out/r4_evalf_modes2.txt
