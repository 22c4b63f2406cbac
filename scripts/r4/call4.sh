#!/bin/bash
set -o pipefail
O=gpurun_out/r4c4; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest_all.log
