#!/bin/bash
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
: > gpurun_out/r4_timeline2.txt
for v in after after192 after128 cur; do
  timeout -k 10 200 python3 scripts/r4/timeline_probe.py $v >> gpurun_out/r4_timeline2.txt 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/r4_timeline2.txt; exit 1; }
done
grep -v amdgpu.ids gpurun_out/r4_timeline2.txt | cut -c1-220
