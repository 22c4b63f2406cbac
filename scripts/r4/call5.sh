#!/bin/bash
set -o pipefail
O=gpurun_out/r4c5; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eval_fused or ring_body" > $O/pytest_evalf.log 2>&1; echo "pytest evalf rc=$?"; tail -15 $O/pytest_evalf.log
timeout -k 10 200 python scripts/r4/evalf_probe.py 100 > $O/evalf_probe_k100.log 2>&1; grep -v amdgpu $O/evalf_probe_k100.log
timeout -k 10 200 python scripts/r4/evalf_probe.py 500 > $O/evalf_probe_k500.log 2>&1; grep -v amdgpu $O/evalf_probe_k500.log
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -40 $O/pytest_all.log
