#!/bin/bash
# full GPU suite on the current tree, then the default bench line
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4c31_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r4c31_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
