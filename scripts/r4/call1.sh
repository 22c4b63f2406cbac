#!/bin/bash
# round-4 GPU call 1: scheduled drain -- parity, stage probe, phase stamps, cfg4 A/B
set -o pipefail
O=gpurun_out/r4c1; mkdir -p $O
EXP=anncur_amd/lib/libanncur_hip_exp.so
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_cfg45.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
ANNCUR_LIB=$EXP STAGE_PROBE_ONLY="default;drain=0;drain=3;drain=6;drain=12;drain=24;bare;bare drain=0" timeout -k 10 400 python scripts/stage_probe.py 100 7 > $O/stage_probe.log 2>&1; echo "probe rc=$?"
tail -12 $O/stage_probe.log
ANNCUR_LIB=$EXP timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_default.log 2>&1; echo "phases rc=$?"
ANNCUR_LIB=$EXP ANNCUR_DEBUG_DRAIN_TILES=0 timeout -k 10 200 python scripts/sweep_phases.py > $O/phases_drain0.log 2>&1
grep -E "stage|barrier|drain|section|ticket" $O/phases_default.log $O/phases_drain0.log
for r in 1 2; do
 for d in default 0; do
  if [ $d = default ]; then E=""; else E="ANNCUR_DEBUG_DRAIN_TILES=0"; fi
  env ANNCUR_LIB=$EXP $E timeout -k 10 300 python bench.py --config cfg4_per_gpu --steps 10 --warmup 3 --sustained-seconds 0 --cpu-sample-queries 0 --no-ivf --no-k500 > $O/cfg4_${d}_$r.json 2>$O/cfg4_${d}_$r.err
  python - <<P
import json
d=json.loads(open("$O/cfg4_${d}_$r.json").read().strip().splitlines()[-1])
print("cfg4 drain=$d r$r step %.3f sweep_only %.3f stages %s" % (d["ms_per_step"], d["stage_ms"]["sweep_kernels_only"], [round(x["ms"],3) for x in d["sweep_stages"]]))
P
 done
done
