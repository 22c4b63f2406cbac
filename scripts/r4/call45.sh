#!/bin/bash
for r in 1 2; do
  for c in 32 64 96; do
    timeout -k 10 200 python3 bench.py --config cfg4_per_gpu --scan-mode partition --scan-cus $c --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_cfg4_p$c.json 2> gpurun_out/r4_cfg4_p$c.err || exit 1
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg4_p$c.json').read().strip().splitlines()[-1]); print('partition scan-cus $c run $r', 'ms_per_step %.3f' % d['ms_per_step'], 'recall', d['recall'])"
  done
done
timeout -k 10 200 python3 bench.py --config cfg4_per_gpu --scan-mode partition --scan-cus 64 --retr-streams 1 --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_cfg4_p64s1.json 2> gpurun_out/r4_cfg4_p64s1.err || exit 1
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg4_p64s1.json').read().strip().splitlines()[-1]); print('partition scan-cus 64, one retrieval stream', 'ms_per_step %.3f' % d['ms_per_step'])"
