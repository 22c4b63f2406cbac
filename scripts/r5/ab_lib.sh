#!/bin/bash
# Round 5 (GPU box): A/B of two builds of the library on one box, alternating -- the exact scan alone by CUs of a masked stream
# (scripts/r4/scan_probe.py) and bench.py's step.  How profiles/r05_scan_select_ab.txt was measured.
#   here (no GPU):  git stash; make -C anncur_amd/csrc; cp anncur_amd/lib/libanncur_hip.so anncur_amd/lib/libanncur_hip_v_OLD.so; git stash pop; make -C anncur_amd/csrc
#   GPU box:        bash scripts/r5/ab_lib.sh anncur_amd/lib/libanncur_hip_v_OLD.so [scan-cus ...]
set -e
old=$1; shift
cus_list=${@:-"default"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in old new; do
    lib=anncur_amd/lib/libanncur_hip.so; [ $v = old ] && lib=$old
    echo "== $v (rep $rep)"
    ANNCUR_LIB=$lib timeout -k 10 300 python3 scripts/r4/scan_probe.py 2>/dev/null | grep -E "bench|iid"
  done
done
for rep in 1 2 3; do
  for cus in $cus_list; do
    for v in old new; do
      lib=anncur_amd/lib/libanncur_hip.so; [ $v = old ] && lib=$old
      flag=""; [ "$cus" != default ] && flag="--scan-cus $cus"
      ANNCUR_LIB=$lib timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 $flag 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'scan_cus', d['scan_mode']['scan_cus'], 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), 'scan ms', round(d['stage_ms'].get('exact_scan'), 4))"
    done
  done
done
