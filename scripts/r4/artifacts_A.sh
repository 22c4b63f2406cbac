#!/bin/bash
# round-4 artefacts, part A: the bench lines (N = 1 cfg2, cfg4 per-GPU shape, 2-rank rehearsal on one GPU) + rocprofv3 passes of cfg2
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
echo "== bench cfg2"; timeout -k 10 400 python3 bench.py > $O/bench_cfg2_n1.json 2> $O/bench_cfg2.err; echo rc=$?
echo "== bench cfg4 per gpu"; timeout -k 10 300 python3 bench.py --config cfg4_per_gpu --no-ivf > $O/bench_cfg4_per_gpu_n1.json 2> $O/bench_cfg4.err; echo rc=$?
echo "== profile cfg2"; timeout -k 10 500 bash scripts/profile_round.sh r04_cfg2 cfg2 > $O/profile_cfg2.log 2>&1; echo rc=$?
tail -3 $O/profile_cfg2.log
python3 - <<'P'
import json
for f in ("bench_cfg2_n1", "bench_cfg4_per_gpu_n1"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, "value %.0f ms_per_step %.4f roofline %.3f (rocprof %s) scan %.3f stages %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_rocprof"), d["roofline_scan"]["frac"], [(x["tiles"], round(x["ms"], 4), round(x["tflops"])) for x in d["sweep_stages"]]))
        print("   stage_ms", {k: round(v, 4) for k, v in d["stage_ms"].items()}, "entry_A_cell", d.get("entry_A_cell") and {k: round(v, 4) for k, v in d["entry_A_cell"].items() if isinstance(v, float)})
        if d.get("ivf_search"): print("   ivf", {t: (round(d["ivf_search"][t]["search_ms"], 2), round(d["ivf_search"][t]["kernels"]["group_gemm_ms"], 3), round(d["ivf_search"][t]["kernels"]["roofline"]["frac"], 3)) for t in ("fp32", "bf16")}, d["ivf_search"]["bf16"].get("recall_vs_fp32_index"))
    except Exception as e: print(f, "FAILED", e)
P
