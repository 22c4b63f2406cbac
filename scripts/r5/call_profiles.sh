#!/bin/bash
# Round 5 (GPU box): the rocprofv3 passes the bench line quotes (kernel stats + PMC of cfg2), the sweep's ceiling under the profiler, scan-CU sweep.
export TMPDIR=/tmp
out=gpurun_out/r5_prof; mkdir -p $out
bash scripts/profile_round.sh r05_cfg2 cfg2 > $out/profile_cfg2.log 2>&1; echo "profile cfg2 rc=$?"
for v in ceiling normal; do
  flag=""; [ $v = normal ] && flag="--normal"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -o run -- python3 scripts/r5/ceiling_probe.py $flag > $out/$v.log 2>&1; echo "$v rc=$?"
  f=$(find $out/$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|score16_kernel|score_kernel|kth_value|select_wave" "$f" | cut -c1-220 > $out/${v}_kernel_stats.csv
  find $out/$v -type f ! -name "*kernel_stats.csv" -delete
done
for cus in 64 96 128; do
  python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 3 --scan-cus $cus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('scan_cus', d['scan_mode']['scan_cus'], 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4))"
done
python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 3 --scan-mode side 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('side', 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4))"
