# Round 5 (GPU box): bench.py's step by the scan's CU share, current library vs the one before the single-pass select, alternating
set -e
mkdir -p gpurun_out/r5_scan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cus in 96 128 144 160 176 192; do
  for v in old new; do
    lib=anncur_amd/lib/libanncur_hip.so; [ $v = old ] && lib=anncur_amd/lib/libanncur_hip_v_SCANOLD.so
    ANNCUR_LIB=$lib timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 --scan-cus $cus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'scan_cus', d['scan_mode']['scan_cus'], 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4))"
  done
done > gpurun_out/r5_scan/ab_scan_cus.txt 2>&1
cat gpurun_out/r5_scan/ab_scan_cus.txt
