#!/bin/bash
for r in 1 2 3; do
STAGE_PROBE_ONLY="default" timeout -k 10 200 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu|^k =" | sed 's/^/new  /'
(cd build/r3_tree && STAGE_PROBE_ONLY="default" timeout -k 10 200 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu|^k =" | sed 's/^/r3   /')
done
