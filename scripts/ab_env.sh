#!/bin/bash
# Dev tool (GPU box): bench.py under several (library, ANNCUR_DEBUG_* environment) settings, alternating on ONE box (devices of the pool
# differ by up to 12 % on MFMA-bound kernels: only same-box numbers compare).
# usage: bash scripts/ab_env.sh rounds "NAME:LIB:ENV1=V ENV2=V" ...   LIB = exp | v_<variant> | prod | old (= build/r3_tree: the previous round's head, exported and built); extra bench flags in $AB_FLAGS
rounds=$1; shift
mkdir -p gpurun_out/abenv; rm -f gpurun_out/abenv/*.json
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=$(echo "$spec" | cut -d: -f1); lib=$(echo "$spec" | cut -d: -f2); envs=$(echo "$spec" | cut -d: -f3)
    if [ "$lib" = old ]; then
      (cd build/r3_tree && python3 bench.py --sustained-seconds 0 --cpu-sample-queries 0 --steps 40 ${AB_FLAGS//--no-ivf/} 2>/dev/null) > gpurun_out/abenv/${name}__$r.json
    else
      L=anncur_amd/lib/libanncur_hip_$lib.so; [ "$lib" = prod ] && L=anncur_amd/lib/libanncur_hip.so
      env ANNCUR_LIB=$L X=1 $envs python3 bench.py --sustained-seconds 0 --cpu-sample-queries 0 --steps 40 $AB_FLAGS 2>/dev/null > gpurun_out/abenv/${name}__$r.json
    fi
  done
done
python3 - "$@" <<'P'
import json, glob, sys
for spec in sys.argv[1:]:
    name = spec.split(":")[0]
    for f in sorted(glob.glob(f"gpurun_out/abenv/{name}__*.json")):
        try: d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e: print(name, "FAILED", e); continue
        s = d["stage_ms"]; st = d.get("sweep_stages", [])
        print("%-22s step %.4f sweep_only %.4f (%s) sweep %.4f sel %.4f scan %.4f retr %.4f k500 %.4f" % (name, d["ms_per_step"], s["sweep_kernels_only"], " ".join("%d:%.4f" % (x["tiles"], x["ms"]) for x in st), s["sweep"], s["select"], s["exact_scan"], d["retrieve_only"]["ms_per_step"], (d.get("retrieve_only_k500") or {}).get("ms_per_step", 0)))
P
