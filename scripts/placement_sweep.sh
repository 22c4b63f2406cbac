#!/bin/bash
# Dev tool: build the library with 1..N extra instructions in front of every sweep loop (ANNCUR_PLACEMENT_PAD) and run the fused
# parity tests against each build on the GPU box -- a result that depends on code placement is a missing wait state.
# usage (two steps, the build runs where hipcc is, the tests on the GPU box):
#   bash scripts/placement_sweep.sh build 7        # -> anncur_amd/lib/libanncur_hip_pad<N>.so
#   bash scripts/placement_sweep.sh test 7         # (on the GPU box) pytest of the fused kernels per library + the static checks
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; n=${2:-7}
lib=$root/anncur_amd/lib
if [ "$mode" = build ]; then
  for i in $(seq 1 $n); do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -Wall -Wno-unused-function -DANNCUR_PLACEMENT_PAD=$i \
        -c $root/anncur_amd/csrc/score_fused.hip -o /tmp/pad_score_fused_$i.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -Wall -Wno-unused-function -DANNCUR_PLACEMENT_PAD=$i \
        -c $root/anncur_amd/csrc/ivf.hip -o /tmp/pad_ivf_$i.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/libanncur_hip_pad$i.so $lib/misc.o $lib/topk.o $lib/gemm.o $lib/gemm64.o /tmp/pad_ivf_$i.o /tmp/pad_score_fused_$i.o ) &
    if [ $((i % 4)) = 0 ]; then wait; fi
  done
  wait; ls -la $lib/libanncur_hip_pad*.so
else
  for i in $(seq 1 $n); do
    echo "== pad $i"
    ANNCUR_LIB=$lib/libanncur_hip_pad$i.so python -m pytest $root/tests/test_gpu_kernels.py $root/tests/test_gpu_random_shapes.py $root/tests/test_gpu_entrypoints.py -m gpu -x -q -k "fused or wide or cfg or sweep or ivf" 2>&1 | tail -2
  done
fi
