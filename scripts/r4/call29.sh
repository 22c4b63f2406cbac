#!/bin/bash
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
for s in 0 8 16 32 48; do
  echo "== stagger $s us"
  ANNCUR_DEBUG_SCAN_STAGGER=$s timeout -k 10 300 python3 scripts/r4/scan_probe.py 2>&1 | grep -E "^bench|^iid"
done > gpurun_out/r4_scan_stagger.txt
cat gpurun_out/r4_scan_stagger.txt
