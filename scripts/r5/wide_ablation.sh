#!/bin/bash
# Round 5: where wide_kernel's time goes -- ablation builds of the experiments library (`make variant V=WIDE_<what>`: one part of the k-loop
# compiled out, results wrong, select skipped), each timed by scripts/r5/wide_probe.py on one box.
#   build (here, no GPU):  bash scripts/r5/wide_ablation.sh build
#   run   (GPU box):       bash scripts/r5/wide_ablation.sh run
V="WIDE_BASE WIDE_NOFILTER WIDE_NOBAR WIDE_NODMA WIDE_NOREAD WIDE_NODMA_NOREAD WIDE_MFMAONLY"
if [ "$1" = build ]; then
  for v in $V; do make -C anncur_amd/csrc -j4 variant V=$v ARCH=gfx950 2>&1 | grep -i " error" ; ls -la anncur_amd/lib/libanncur_hip_v_$v.so | awk '{print $5, $9}'; done
else
  for v in $V; do
    echo "$v: $(ANNCUR_LIB=anncur_amd/lib/libanncur_hip_v_$v.so timeout -k 10 200 python3 scripts/r5/wide_probe.py --rounds 3 --no-parity 2>&1 | grep 'rep 2' | sed 's/.*sweep kernels/sweep kernels/; s/| total.*//')"
  done
fi
