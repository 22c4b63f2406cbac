set -e
mkdir -p gpurun_out/r5_scan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py -q -x -m gpu -k "rowwise_topk or scan or rerank or gather" > gpurun_out/r5_scan/tests.txt 2>&1 || { tail -40 gpurun_out/r5_scan/tests.txt; exit 1; }
tail -2 gpurun_out/r5_scan/tests.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=500 timeout -k 10 900 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu -k "rowwise_topk" > gpurun_out/r5_scan/fuzz.txt 2>&1 || { tail -40 gpurun_out/r5_scan/fuzz.txt; exit 1; }
tail -2 gpurun_out/r5_scan/fuzz.txt
for i in 1 2; do
timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), 'scan TB/s', round(d['roofline_scan']['achieved']/1e3, 3), 'scan ms', d['stage_ms'].get('exact_scan'), 'chain', round(d['stage_ms']['chain_sum'],4))"
done
