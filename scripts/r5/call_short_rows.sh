set -e
mkdir -p gpurun_out/r5_short
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py tests/test_gpu_entrypoints.py -q -x -m gpu -k "rowwise_topk or ivf or dense" > gpurun_out/r5_short/tests.txt 2>&1 || { tail -40 gpurun_out/r5_short/tests.txt; exit 1; }
tail -2 gpurun_out/r5_short/tests.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=400 timeout -k 10 600 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu -k "rowwise_topk_random" > gpurun_out/r5_short/fuzz.txt 2>&1 || { tail -40 gpurun_out/r5_short/fuzz.txt; exit 1; }
tail -2 gpurun_out/r5_short/fuzz.txt
MODES=1 GEMM_ONLY=0 timeout -k 10 300 python3 scripts/r5/ivf_probe.py > gpurun_out/r5_short/probe.txt 2>&1; cat gpurun_out/r5_short/probe.txt
MODES=1 GEMM_ONLY=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_short/prof -o ivf -- python3 scripts/r5/ivf_probe.py > gpurun_out/r5_short/probe_prof.txt 2>&1
grep -E "rowwise_topk" gpurun_out/r5_short/prof/ivf_kernel_stats.csv | cut -c1-160
