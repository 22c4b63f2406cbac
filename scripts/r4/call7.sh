#!/bin/bash
set -o pipefail
O=gpurun_out/r4c7; mkdir -p $O
EXP=anncur_amd/lib/libanncur_hip_exp.so
timeout -k 10 400 python -m pytest tests/test_gpu_entrypoints.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ivf or fused or cfg2" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
ANNCUR_LIB=$EXP timeout -k 10 200 python scripts/ab_wide.py 2>&1 | grep -v amdgpu > $O/ab_wide.txt; cat $O/ab_wide.txt
timeout -k 10 900 bash scripts/r4/profile_wide.sh > $O/profile_wide.log 2>&1; echo "profile wide rc=$?"; tail -60 $O/profile_wide.log
timeout -k 10 300 python scripts/recall_sweep_cfg5.py > $O/cfg5_recall.json 2>$O/cfg5.err; echo "cfg5 rc=$?"; tail -c 1500 $O/cfg5_recall.json
