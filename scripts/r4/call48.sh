#!/bin/bash
for r in 1 2 3 4; do
  for c in 96 128; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --scan-cus $c --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_cfg2_p$c.json 2> gpurun_out/r4_cfg2_p$c.err || exit 1
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg2_p$c.json').read().strip().splitlines()[-1]); print('cfg2 scan-cus $c run $r (steps 20, warmup 5)', 'ms_per_step %.4f' % d['ms_per_step'])"
  done
done
