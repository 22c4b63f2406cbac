#!/bin/bash
# placement sweep on the final score_fused.hip (7 shifted builds): the fused parity tests + the eval_fused / error tests per build
for i in 1 2 3 4 5 6 7; do
  echo "== pad $i"
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_pad$i.so timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py -m gpu -x -q -k "fused or wide or cfg or sweep or eval_fused or error" 2>&1 | tail -1
done
