# Round 5 (GPU box): the bench lines profiles/r05_bench_* hold -- the default line, cfg4's per-GPU shape, the 2-rank gloo rehearsal on one GPU
set -e
export TMPDIR=/tmp
out=gpurun_out/r5_final; mkdir -p $out
timeout -k 10 900 python3 bench.py > $out/bench_cfg2_n1.json 2> $out/bench_cfg2_n1.err
timeout -k 10 900 python3 bench.py --config cfg4_per_gpu --no-ivf > $out/bench_cfg4_per_gpu_n1.json 2> $out/bench_cfg4.err
timeout -k 10 900 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 3 > $out/bench_default_2ranks_gloo_one_gpu_rehearsal.json 2> $out/bench_2ranks.err
python3 - <<'P'
import json
for f in ("bench_cfg2_n1", "bench_cfg4_per_gpu_n1", "bench_default_2ranks_gloo_one_gpu_rehearsal"):
    d = json.loads([l for l in open(f"gpurun_out/r5_final/{f}.json") if l.startswith("{")][-1])
    print(f, "n_gpus", d["n_gpus"], "ms_per_step", round(d["ms_per_step"], 4), "value", round(d["value"]), "frac", round(d["roofline"]["frac"], 3), "frac_rocprof", d["roofline"].get("frac_rocprof"), "scan_mode", d.get("scan_mode", {}).get("used"))
P
