#!/bin/bash
set -o pipefail
O=gpurun_out/r4c2; mkdir -p $O
EXP=anncur_amd/lib/libanncur_hip_exp.so
timeout -k 10 300 python scripts/r4/ring_check.py > $O/ring_check.log 2>&1; rc=$?; tail -12 $O/ring_check.log; echo "ring_check rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
ANNCUR_LIB=$EXP STAGE_PROBE_ONLY="default;noring;bare;bare noring" timeout -k 10 400 python scripts/stage_probe.py 100 9 > $O/stage_probe.log 2>&1; echo "probe rc=$?"
tail -6 $O/stage_probe.log
ANNCUR_LIB=$EXP timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_ring.log 2>&1; echo "phases rc=$?"
ANNCUR_LIB=$EXP PH_NORING=1 timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_noring.log 2>&1
grep -vE "amdgpu.ids|^plan" $O/phases_ring.log $O/phases_noring.log
