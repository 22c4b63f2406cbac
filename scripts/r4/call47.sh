#!/bin/bash
for r in 1 2; do
  for c in 64 96 128; do
    timeout -k 10 200 python3 bench.py --scan-cus $c --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 --steps 40 --warmup 10 > gpurun_out/r4_cfg2_p$c.json 2> gpurun_out/r4_cfg2_p$c.err || exit 1
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg2_p$c.json').read().strip().splitlines()[-1]); print('cfg2 partition scan-cus $c run $r', 'ms_per_step %.4f' % d['ms_per_step'])"
  done
done
