#!/bin/bash
# cfg4 per-GPU shape: the scan's placement again, with the XCD-sliced tickets (alternating runs, one box)
for r in 1 2; do
  for m in side partition; do
    timeout -k 10 200 python3 bench.py --config cfg4_per_gpu --scan-mode $m --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_cfg4_$m$r.json 2> gpurun_out/r4_cfg4_$m$r.err || { echo "$m failed"; tail -3 gpurun_out/r4_cfg4_$m$r.err; exit 1; }
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg4_$m$r.json').read().strip().splitlines()[-1]); print('$m', 'run $r', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], d['scan_mode']['used'])"
  done
done
for c in 64 128; do
  timeout -k 10 200 python3 bench.py --config cfg4_per_gpu --scan-mode partition --scan-cus $c --no-ivf --no-k500 --sustained-seconds 0 --cpu-sample-queries 0 > gpurun_out/r4_cfg4_p$c.json 2> gpurun_out/r4_cfg4_p$c.err || exit 1
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_cfg4_p$c.json').read().strip().splitlines()[-1]); print('partition scan-cus $c', 'ms_per_step %.3f' % d['ms_per_step'])"
done
