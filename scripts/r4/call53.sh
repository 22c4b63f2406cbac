#!/bin/bash
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
for r in 1 2; do
for t in 0 160 200 300 400; do
  if [ $t = 0 ]; then unset ANNCUR_DEBUG_SCAN_TRIGGER; else export ANNCUR_DEBUG_SCAN_TRIGGER=$t; fi
  echo -n "trigger ${t} (0 = default 250) round $r: "; timeout -k 10 200 python3 scripts/r4/scan_ab_probe.py 2>&1 | grep -v amdgpu.ids | cut -d' ' -f2-
done
done
