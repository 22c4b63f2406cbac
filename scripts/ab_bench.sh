#!/bin/bash
# Dev tool (GPU box): the bench of this tree and of build/r2_tree (the round-2 head, exported and built by hand) alternating on ONE box:
# devices of the pool differ by up to 12 % on MFMA-bound kernels, so only same-box numbers compare.  usage: bash scripts/ab_bench.sh [rounds] [extra bench flags for the new tree]
rounds=${1:-2}; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $rounds); do
  (cd build/r2_tree && python3 bench.py --sustained-seconds 0 --cpu-sample-queries 0 --no-k500 --steps 40 2>/dev/null) > gpurun_out/ab/old_$r.json
  python3 bench.py --sustained-seconds 0 --cpu-sample-queries 0 --no-k500 --steps 40 "$@" 2>/dev/null > gpurun_out/ab/new_$r.json
  python3 bench.py --sustained-seconds 0 --cpu-sample-queries 0 --no-k500 --steps 40 --no-overlap "$@" 2>/dev/null > gpurun_out/ab/newser_$r.json
done
python3 - <<'P'
import json, glob
for tag in ("old", "new", "newser"):
    for f in sorted(glob.glob(f"gpurun_out/ab/{tag}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d["stage_ms"]
        print("%-7s ms_per_step %.4f  sweep_only %.4f sweep %.4f prepass %.4f thr %.4f sel %.4f scan %.4f gather %.4f  retrieve_only %.4f" % (tag, d["ms_per_step"], s["sweep_kernels_only"], s["sweep"], s["prepass"], s["threshold"], s["select"], s["exact_scan"], s["gather_cols"], d["retrieve_only"]["ms_per_step"]))
P
