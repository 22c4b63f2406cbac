#!/bin/bash
# Round 5: where ivf_tile128_kernel's time goes -- ablation builds (`make variant V=IVF_<what>`: results wrong), timed by scripts/r5/ivf_probe.py.
#   build (here, no GPU):  bash scripts/r5/ivf_ablation.sh build
#   run   (GPU box):       bash scripts/r5/ivf_ablation.sh run
V=${VARIANTS:-"IVF_BASE IVF_NODMA IVF_NOSTORE"}
if [ "$1" = build ]; then
  for v in $V; do make -C anncur_amd/csrc -j4 variant V=$v ARCH=gfx950 2>&1 | grep -i " error" ; ls -la anncur_amd/lib/libanncur_hip_v_$v.so | awk '{print $5, $9}'; done
else
  for v in $V; do
    echo "$v: $(ANNCUR_LIB=anncur_amd/lib/libanncur_hip_v_$v.so MODES=1 timeout -k 10 200 python3 scripts/r5/ivf_probe.py 2>&1 | grep 'tile kernel')"
  done
fi
