set -e
mkdir -p gpurun_out/r5_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/placement_sweep.sh test 7 > gpurun_out/r5_final/placement_sweep.txt 2>&1
cat gpurun_out/r5_final/placement_sweep.txt
ANNCUR_FUZZ=1 ANNCUR_FUZZ_EXAMPLES=120 timeout -k 10 1000 python3 -m pytest tests/test_gpu_random_shapes.py -q -x -m gpu > gpurun_out/r5_final/fresh_fuzz.txt 2>&1 || { tail -40 gpurun_out/r5_final/fresh_fuzz.txt; exit 1; }
tail -3 gpurun_out/r5_final/fresh_fuzz.txt
