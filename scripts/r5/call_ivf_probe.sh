set -e
mkdir -p gpurun_out/r5_ivf
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_entrypoints.py -q -x -m gpu -k "ivf" > gpurun_out/r5_ivf/tests.txt 2>&1 || { tail -40 gpurun_out/r5_ivf/tests.txt; exit 1; }
tail -3 gpurun_out/r5_ivf/tests.txt
timeout -k 10 300 python3 scripts/r5/ivf_probe.py > gpurun_out/r5_ivf/probe1.txt 2>&1
cat gpurun_out/r5_ivf/probe1.txt
MODES=1 GEMM_ONLY=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_ivf/prof1 -o ivf -- python3 scripts/r5/ivf_probe.py > gpurun_out/r5_ivf/probe1_prof.txt 2>&1
