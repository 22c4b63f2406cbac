#!/bin/bash
# Round 5 (GPU box, experiments library): the ladder's refresh period (tiles between two fetches of a wave's counter words) and the rank of its
# top level, each through scripts/r5/ladder_probe.py (ladder vs staged in one process).  usage: bash scripts/r5/ladder_period_sweep.sh
export ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so
for per in 4 8 16 32; do
  echo "=== period $per"
  ANNCUR_DEBUG_LADDER_PERIOD=$per python3 scripts/r5/ladder_probe.py --rounds 2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{\"variant'): continue
    d = json.loads(l)
    if d['rep'] == 1: print('  %-7s sweep kernels %.4f  sweep+refine %.4f  threshold %.4f  select %.4f  total %.4f  survivors %.1f' % (d['variant'], d['sweep_kernels'], d['sweep_incl_refine'], d['threshold'], d['select'], d['total'], d['survivors_per_query']))
"
done
for k2 in 8 12 24 32; do
  echo "=== top rank $k2 (period 16)"
  ANNCUR_DEBUG_LADDER_K2=$k2 python3 scripts/r5/ladder_probe.py --rounds 2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{\"variant'): continue
    d = json.loads(l)
    if d['rep'] == 1 and d['variant'] == 'ladder': print('  %-7s sweep kernels %.4f  sweep+refine %.4f  threshold %.4f  select %.4f  total %.4f  survivors %.1f' % (d['variant'], d['sweep_kernels'], d['sweep_incl_refine'], d['threshold'], d['select'], d['total'], d['survivors_per_query']))
"
done
