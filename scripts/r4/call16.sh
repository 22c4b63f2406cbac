#!/bin/bash
EXP=anncur_amd/lib/libanncur_hip_exp.so
ANNCUR_LIB=$EXP STAGE_PROBE_ONLY="default;wg8;bare;bare wg8" timeout -k 10 300 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu" | tee gpurun_out/r4_wg8_probe.txt
ANNCUR_LIB=$EXP ANNCUR_DEBUG_WG8=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused" 2>&1 | tail -3
