set -e
mkdir -p gpurun_out/r5_full
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python3 -m pytest tests -q -x -m gpu > gpurun_out/r5_full/tests_gpu.txt 2>&1 || { tail -40 gpurun_out/r5_full/tests_gpu.txt; exit 1; }
tail -3 gpurun_out/r5_full/tests_gpu.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=300 timeout -k 10 600 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu -k "ivf" > gpurun_out/r5_full/fuzz_ivf.txt 2>&1 || { tail -40 gpurun_out/r5_full/fuzz_ivf.txt; exit 1; }
tail -2 gpurun_out/r5_full/fuzz_ivf.txt
timeout -k 10 900 python3 bench.py > gpurun_out/r5_full/bench.json 2> gpurun_out/r5_full/bench.err || { tail -30 gpurun_out/r5_full/bench.err; exit 1; }
python3 - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r5_full/bench.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'sustained', d.get('sustained',{}).get('ms_per_step'), 'frac', d['roofline']['frac'], 'scan', d.get('roofline_scan',{}).get('achieved'))
iv=d['ivf_search']
for t in ('fp32','bf16'):
    r=iv[t]; print(t, 'search_ms %.3f device %.3f gemm %.3f rest %.3f tile_ratio %.3f frac %.3f' % (r['search_ms'], r['search_device_ms'], r['kernels']['group_gemm_ms'], r['kernels']['scan_and_id_map_ms'], r['kernels']['tile_flops_ratio'], r['kernels']['roofline']['frac']))
P
