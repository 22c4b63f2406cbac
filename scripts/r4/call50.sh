#!/bin/bash
for r in 1 2 3; do
  for sl in 2 3; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --scan-slots $sl --no-ivf --no-k500 --sustained-seconds 3 --cpu-sample-queries 0 > gpurun_out/r4_slots$sl.json 2> gpurun_out/r4_slots$sl.err || { tail -5 gpurun_out/r4_slots$sl.err; exit 1; }
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_slots$sl.json').read().strip().splitlines()[-1]); print('scan-slots $sl run $r', 'ms_per_step %.4f' % d['ms_per_step'], 'sustained %.4f' % d['sustained']['ms_per_step'], 'recall@100', d['recall']['recall@100'])"
  done
done
