#!/bin/bash
for r in 1 2 3; do
  for us in 0 300 450 600; do
    ANNCUR_BENCH_STAGGER_US=$us timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_stag.json 2> gpurun_out/r4_stag.err || exit 1
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_stag.json').read().strip().splitlines()[-1]); print('stagger $us us run $r', 'ms_per_step %.4f' % d['ms_per_step'])"
  done
done
