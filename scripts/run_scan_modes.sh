# Dev tool (GPU box): bench.py under several exact-scan placements, one box.  usage: bash scripts/run_scan_modes.sh "cfg mode cus [extra flags]" ...
for spec in "$@"; do
  set -- $spec
  cfg=$1; mode=$2; cus=$3; shift 3
  timeout -k 10 200 python3 bench.py --config $cfg --no-ivf --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --scan-mode $mode --scan-cus $cus "$@" > gpurun_out/m.json 2> gpurun_out/m.err || true
  if grep -q "Memory access fault" gpurun_out/m.err; then echo "FAULT in $spec"; exit 1; fi
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/m.json').read().strip().splitlines()[-1])
print('$spec', 'ms_per_step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], d['launch_mode'][:9], d['recall']['recall@100'])"
done
