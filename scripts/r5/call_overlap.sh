set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_ovl
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py -q -x -m gpu -k "overlap" 2>&1 | tail -2
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=600 timeout -k 10 600 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu -k "overlap" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 600 python3 bench.py --direct --steps 30 --warmup 5 --no-ivf --cpu-sample-queries 0 --no-k500 --no-ceiling --sustained-seconds 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), {k: round(v, 4) for k, v in d['stage_ms'].items()})"
done
