#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q > gpurun_out/r4c46_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r4c46_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python3 bench.py --config cfg4_per_gpu --no-ivf > $O/bench_cfg4_per_gpu_n1.json 2> $O/bench_cfg4.err; echo "bench cfg4 rc=$?"
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --sustained-seconds 0 --no-k500 > $O/bench_default_2ranks_gloo_one_gpu_rehearsal.json 2> $O/bench_2r.err; echo "2 ranks rc=$?"
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/r04/bench_cfg4_per_gpu_n1.json").read().strip().splitlines()[-1])
print("cfg4 value %.0f ms %.3f scan_mode %s hidden %.3f sustained %s" % (d["value"], d["ms_per_step"], d["scan_mode"], d["stage_ms"]["hidden_by_overlap"], d.get("sustained") and round(d["sustained"]["ms_per_step"], 3)))
d = json.loads(open("gpurun_out/r04/bench_default_2ranks_gloo_one_gpu_rehearsal.json").read().strip().splitlines()[-1])
print("2 ranks: value %.0f ms %.3f solo %.0f | cfg4 value %.0f ms %.3f solo %.0f mode %s" % (d["value"], d["ms_per_step"], d["solo_rank0"]["value"], d["cfg4"]["value"], d["cfg4"]["ms_per_step"], d["cfg4"]["solo_rank0"]["value"], d["cfg4"]["scan_mode"]))
P
