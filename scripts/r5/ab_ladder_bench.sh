#!/bin/bash
# Round 5 (GPU box): the whole bench step with the in-flight threshold ladder (default) against the staged sweep (--sweep-staged), same box,
# alternating processes; prints ms_per_step, sustained ms_per_step and the stage split of each run.
# usage: bash scripts/r5/ab_ladder_bench.sh [rounds] [extra bench args]
rounds=${1:-2}; shift
out=gpurun_out/r5_ab_ladder
mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in staged ladder; do
    flag=""; [ $v = staged ] && flag="--sweep-staged"
    python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --sustained-seconds 4 $flag "$@" > $out/${v}_$r.json 2> $out/${v}_$r.err || { echo "$v $r FAILED"; tail -5 $out/${v}_$r.err; exit 1; }
    python3 - $out/${v}_$r.json $v $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = d["stage_ms"]
print("%-7s r%s step %.4f ms  sustained %.4f  chain %.4f (gather %.3f prepass %.3f thr %.3f sweep %.4f [kernels %.4f] select %.3f) scan %.4f  retrieve_only %.4f  survivors %.1f  mode %s" % (
	sys.argv[2], sys.argv[3], d["ms_per_step"], (d.get("sustained") or {}).get("ms_per_step", 0), st["chain_sum"], st["gather_cols"], st["prepass"], st["threshold"], st["sweep"], st["sweep_kernels_only"],
	st["select"], st["exact_scan"], d["retrieve_only"]["ms_per_step"], d["retrieve_only"]["survivors_per_query"], d["scan_mode"]["used"]), flush=True)
PY
  done
done
