#!/bin/bash
set -o pipefail
O=gpurun_out/r4c3; mkdir -p $O
EXP=anncur_amd/lib/libanncur_hip_exp.so
ANNCUR_LIB=$EXP STAGE_PROBE_ONLY="default;stag=10;stag=20;stag=30;stag=20 nosleep;nosleep;noring;bare;bare stag=20;bare noring" timeout -k 10 500 python scripts/stage_probe.py 100 9 > $O/stage_probe.log 2>&1; echo "probe rc=$?"
tail -12 $O/stage_probe.log
ANNCUR_LIB=$EXP ANNCUR_DEBUG_RING_STAGGER=20 timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_ring_stag20.log 2>&1
ANNCUR_LIB=$EXP ANNCUR_DEBUG_TAU_BIAS=1e30 timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_ring_bare.log 2>&1
ANNCUR_LIB=$EXP ANNCUR_DEBUG_TAU_BIAS=1e30 PH_NORING=1 timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_noring_bare.log 2>&1
grep -vE "amdgpu.ids|^plan" $O/phases_ring_stag20.log $O/phases_ring_bare.log $O/phases_noring_bare.log
timeout -k 10 900 python -m pytest tests/test_gpu_entrypoints.py tests/test_gpu_cur.py tests/test_gpu_fullsize.py -x -q -m gpu -k "numpy_pinv or heavy_tailed or spherical or slice or square or ivf" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_new.log
