#!/bin/bash
O=gpurun_out/r4c18; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_entrypoints.py tests/test_gpu_bench_multirank.py tests/test_gpu_random_shapes.py -x -q -m gpu -k "ivf or rccl or default_config" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
/usr/bin/time -v python3 bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; grep -E "Elapsed|Maximum resident" $O/bench_cfg2.err
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/r4c18/bench_cfg2.json").read().strip().splitlines()[-1])
print("value %.0f ms %.4f frac %.3f rocprof %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_rocprof"]))
for t in ("fp32","bf16"):
    r=d["ivf_search"][t]; print(t, "search_ms %.2f gemm %.3f scan %.3f kernels-only q/s %.0f frac %.3f tile_ratio %.2f" % (r["search_ms"], r["kernels"]["group_gemm_ms"], r["kernels"]["scan_and_id_map_ms"], r["kernels"]["queries_per_s_kernels_only"], r["kernels"]["roofline"]["frac"], r["kernels"]["tile_flops_ratio"]), r.get("recall_vs_fp32_index"))
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu"], d["cpu_baseline"]["recall_gpu_same_queries"], d["cpu_baseline"]["recall_cpu_fp32_tie_stable"])
P
