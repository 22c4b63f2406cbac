set -e
mkdir -p gpurun_out/r5_scan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cus in 64 96 128 160; do
    timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 --scan-cus $cus 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('scan_cus', d['scan_mode']['scan_cus'], 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4))"
done
done > gpurun_out/r5_scan/scan_cus3.txt 2>&1
cat gpurun_out/r5_scan/scan_cus3.txt
