set -e
mkdir -p gpurun_out/r5_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/placement_sweep.sh test 7 > gpurun_out/r5_final/placement_sweep.txt 2>&1
cat gpurun_out/r5_final/placement_sweep.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=${FUZZ_EXAMPLES:-120} timeout -k 10 1000 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu > gpurun_out/r5_final/fresh_fuzz.txt 2>&1 || { tail -40 gpurun_out/r5_final/fresh_fuzz.txt; exit 1; }
tail -3 gpurun_out/r5_final/fresh_fuzz.txt
timeout -k 10 600 python3 scripts/r5/soak.py ${SOAK_SECONDS:-15} > gpurun_out/r5_final/soak.txt 2>&1; cat gpurun_out/r5_final/soak.txt
