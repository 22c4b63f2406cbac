set -e
mkdir -p gpurun_out/r5_ivf
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/r5/ivf_ablation.sh run > gpurun_out/r5_ivf/ablation_$TAG.txt 2>&1
cat gpurun_out/r5_ivf/ablation_$TAG.txt
