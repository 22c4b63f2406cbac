#!/bin/bash
# timeline of the partition placement, one process per variant
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
: > gpurun_out/r4_timeline.txt
for v in cur one disj; do
  timeout -k 10 200 python3 scripts/r4/timeline_probe.py $v >> gpurun_out/r4_timeline.txt 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/r4_timeline.txt; exit 1; }
done
for n in 64 128; do
  PROBE_SCAN_CUS=$n timeout -k 10 200 python3 scripts/r4/timeline_probe.py disj >> gpurun_out/r4_timeline.txt 2>&1 || { echo "disj $n failed"; tail -5 gpurun_out/r4_timeline.txt; exit 1; }
done
grep -E "^==|mean durations" gpurun_out/r4_timeline.txt
