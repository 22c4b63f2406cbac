set -e
mkdir -p gpurun_out/r5_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=300 timeout -k 10 1100 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu > gpurun_out/r5_final/fresh_fuzz_300.txt 2>&1 || { tail -40 gpurun_out/r5_final/fresh_fuzz_300.txt; exit 1; }
tail -3 gpurun_out/r5_final/fresh_fuzz_300.txt
timeout -k 10 600 python3 scripts/r5/soak.py 15 > gpurun_out/r5_final/soak2.txt 2>&1; cat gpurun_out/r5_final/soak2.txt
