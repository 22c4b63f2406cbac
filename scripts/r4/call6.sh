#!/bin/bash
set -o pipefail
O=gpurun_out/r4c6; mkdir -p $O
EXP=anncur_amd/lib/libanncur_hip_exp.so
timeout -k 10 600 python -m pytest tests/test_gpu_cfg45.py tests/test_gpu_kernels.py -x -q -m gpu -k "cfg4 or kp512 or 512 or overflow" > $O/pytest_cfg4.log 2>&1; echo "pytest cfg4 rc=$?"; tail -5 $O/pytest_cfg4.log
for r in 1 2 3; do
 for d in 1 0; do
  env ANNCUR_LIB=$EXP ANNCUR_DEBUG_SLICED=$d timeout -k 10 300 python bench.py --config cfg4_per_gpu --steps 10 --warmup 3 --sustained-seconds 0 --cpu-sample-queries 0 --no-ivf --no-k500 > $O/cfg4_sliced${d}_$r.json 2>$O/cfg4_sliced${d}_$r.err
  python - <<P
import json
d=json.loads(open("$O/cfg4_sliced${d}_$r.json").read().strip().splitlines()[-1])
print("cfg4 sliced=$d r$r step %.3f sweep_only %.3f stages %s recall %s" % (d["ms_per_step"], d["stage_ms"]["sweep_kernels_only"], [round(x["ms"],3) for x in d["sweep_stages"]], d["recall"]))
P
 done
done
export TMPDIR=/tmp
for d in 1 0; do
 for pmc in FETCH_SIZE WRITE_SIZE; do
  ANNCUR_DEBUG_SLICED=$d ANNCUR_LIB=$EXP rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $O/pmc_s${d}_$pmc -o run -- python3 bench.py --config cfg4_per_gpu --steps 3 --warmup 1 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-graph --no-overlap --no-ivf > $O/bench_pmc_s${d}_$pmc.json 2> $O/pmc_s${d}_$pmc.err
 done
done
python - <<'P'
import csv, glob
for d in (1, 0):
    for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/r4c6/pmc_s{d}_{pmc}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "scoreq16_kernel" in r["Kernel_Name"] and r["Counter_Name"] == pmc: vals.append(float(r["Counter_Value"]))
        if vals:
            mult = 2 if pmc == "FETCH_SIZE" else 1
            print(f"sliced={d} {pmc}: {len(vals)} launches, mean per launch {mult * sum(vals) / len(vals) * 1024 / 1e9:.3f} GB")
P
find $O -name "*.csv" -size +4M -delete
