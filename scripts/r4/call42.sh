#!/bin/bash
for r in 1 2 3; do
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_v_SCANPTR.so timeout -k 10 200 python3 scripts/r4/scan_ab_probe.py 2>&1 | grep -v amdgpu.ids
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so timeout -k 10 200 python3 scripts/r4/scan_ab_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_scan_ab.txt
