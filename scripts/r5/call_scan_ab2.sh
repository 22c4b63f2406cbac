set -e
mkdir -p gpurun_out/r5_scan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py -q -x -m gpu -k "rowwise_topk or scan or rerank or gather" > gpurun_out/r5_scan/tests2.txt 2>&1 || { tail -40 gpurun_out/r5_scan/tests2.txt; exit 1; }
tail -2 gpurun_out/r5_scan/tests2.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=500 timeout -k 10 900 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu -k "rowwise_topk" > gpurun_out/r5_scan/fuzz2.txt 2>&1 || { tail -40 gpurun_out/r5_scan/fuzz2.txt; exit 1; }
tail -2 gpurun_out/r5_scan/fuzz2.txt
for rep in 1 2; do
  for v in old new; do
    lib=anncur_amd/lib/libanncur_hip.so; [ $v = old ] && lib=anncur_amd/lib/libanncur_hip_v_SCANOLD.so
    echo "== $v (rep $rep)"
    ANNCUR_LIB=$lib timeout -k 10 300 python3 scripts/r4/scan_probe.py 2>/dev/null | grep -E "bench|iid"
  done
done > gpurun_out/r5_scan/ab_probe2.txt 2>&1
cat gpurun_out/r5_scan/ab_probe2.txt
for rep in 1 2 3; do
  for v in old new; do
    lib=anncur_amd/lib/libanncur_hip.so; [ $v = old ] && lib=anncur_amd/lib/libanncur_hip_v_SCANOLD.so
    ANNCUR_LIB=$lib timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), 'scan ms', round(d['stage_ms'].get('exact_scan'), 4))"
  done
done > gpurun_out/r5_scan/ab_bench2.txt 2>&1
cat gpurun_out/r5_scan/ab_bench2.txt
