#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_cfg2_n1.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
timeout -k 10 300 python3 bench.py --config cfg4_per_gpu --no-ivf > $O/bench_cfg4_per_gpu_n1.json 2> $O/bench_cfg4.err; echo "bench cfg4 rc=$?"
python3 - <<'P'
import json
for f in ("bench_cfg2_n1", "bench_cfg4_per_gpu_n1"):
    d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.0f ms %.4f frac %.3f rocprof %.3f (%s) stages_rocprof %s traffic %s scan %.3f" % (d["value"], d["ms_per_step"], r["frac"], r["frac_rocprof"], r["rocprof_source"], r["stages_rocprof"] is not None, r["traffic"], d["roofline_scan"]["frac"]))
P
