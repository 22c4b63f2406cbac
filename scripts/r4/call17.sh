#!/bin/bash
# placement sweep (7 shifted builds) + fresh fuzz draws on the final kernels + time of the default bench command
O=gpurun_out/r4c17; mkdir -p $O
( python3 scripts/check_mfma_hazards.py; python3 scripts/check_lds_hazards.py; timeout -k 10 900 bash scripts/placement_sweep.sh test 7 ) > $O/static_checks_and_placement_sweep.txt 2>&1; tail -16 $O/static_checks_and_placement_sweep.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=1200 timeout -k 10 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu > $O/fuzz_fresh.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz_fresh.log
