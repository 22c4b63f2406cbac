#!/bin/bash
for r in 1 2; do
for v in exp v_R3LOOP v_R3DRAIN v_NOSLICE; do
ANNCUR_LIB=anncur_amd/lib/libanncur_hip_$v.so STAGE_PROBE_ONLY="default;bare" timeout -k 10 200 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu|^k =" | sed "s/^/$v  /"
done
(cd build/r3_tree && STAGE_PROBE_ONLY="default" timeout -k 10 200 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu|^k =" | sed 's/^/r3prod   /')
done
