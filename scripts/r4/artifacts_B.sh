#!/bin/bash
# round-4 artefacts, part B: rocprofv3 passes of the cfg4 per-GPU shape + the 2-rank rehearsal of the default (N > 1) line on one GPU
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
echo "== profile cfg4"; timeout -k 10 700 bash scripts/profile_round.sh r04_cfg4 cfg4_per_gpu > $O/profile_cfg4.log 2>&1; echo rc=$?
tail -3 $O/profile_cfg4.log
echo "== bench 2 ranks gloo, default config"; timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --sustained-seconds 0 --no-k500 > $O/bench_default_2ranks_gloo_one_gpu_rehearsal.json 2> $O/bench_2r.err; echo rc=$?
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/r04/bench_default_2ranks_gloo_one_gpu_rehearsal.json").read().strip().splitlines()[-1])
print("2 ranks: workload", d["config"]["workload"][:40], "value %.0f ms %.3f solo %.0f allgather %.2f | cfg4 value %.0f ms %.3f solo %.0f allgather %.2f" % (d["value"], d["ms_per_step"], d["solo_rank0"]["value"], d["allgather_ms"], d["cfg4"]["value"], d["cfg4"]["ms_per_step"], d["cfg4"]["solo_rank0"]["value"], d["cfg4"]["allgather_ms"]))
P
