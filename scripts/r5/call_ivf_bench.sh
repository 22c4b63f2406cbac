set -e
mkdir -p gpurun_out/r5_ivf
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu -k "ivf or IVF or nearest or flat" > gpurun_out/r5_ivf/tests2.txt 2>&1 || { tail -40 gpurun_out/r5_ivf/tests2.txt; exit 1; }
tail -3 gpurun_out/r5_ivf/tests2.txt
timeout -k 10 600 python3 bench.py --direct --steps 5 --warmup 2 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-ceiling > gpurun_out/r5_ivf/bench_ivf.json 2> gpurun_out/r5_ivf/bench_ivf.err || { tail -30 gpurun_out/r5_ivf/bench_ivf.err; exit 1; }
python3 - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r5_ivf/bench_ivf.json') if l.startswith('{')][-1])
iv=d['ivf_search']
for t in ('fp32','bf16'):
    r=iv[t]; print(t, 'search_ms %.3f device %.3f gemm %.3f rest %.3f tile_ratio %.3f frac %.3f' % (r['search_ms'], r['search_device_ms'], r['kernels']['group_gemm_ms'], r['kernels']['scan_and_id_map_ms'], r['kernels']['tile_flops_ratio'], r['kernels']['roofline']['frac']), r.get('recall_vs_fp32_index'))
P
MODES=1 GEMM_ONLY=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_ivf/prof2 -o ivf -- python3 scripts/r5/ivf_probe.py > gpurun_out/r5_ivf/probe2_prof.txt 2>&1
