#!/bin/bash
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
: > gpurun_out/r4_timeline3.txt
for v in cur ovl32 ovl64 cur ovl48 ovl80; do
  timeout -k 10 200 python3 scripts/r4/timeline_probe.py $v >> gpurun_out/r4_timeline3.txt 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/r4_timeline3.txt; exit 1; }
done
for n in 128; do
  for v in ovl64 ovl96; do PROBE_SCAN_CUS=$n timeout -k 10 200 python3 scripts/r4/timeline_probe.py $v >> gpurun_out/r4_timeline3.txt 2>&1 || exit 1; done
done
grep -E "^==|mean durations" gpurun_out/r4_timeline3.txt
