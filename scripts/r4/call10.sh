#!/bin/bash
EXP=anncur_amd/lib/libanncur_hip_exp.so
ANNCUR_LIB=$EXP STAGE_PROBE_ONLY="default;unsliced;bare;bare unsliced" timeout -k 10 300 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu" | tee gpurun_out/r4_sliced_cfg2_probe.txt
ANNCUR_LIB=build/r3_tree/anncur_amd/lib/libanncur_hip.so STAGE_PROBE_ONLY="default" timeout -k 10 300 python scripts/stage_probe.py 100 9 2>&1 | grep -vE "warm-up|amdgpu" | tee -a gpurun_out/r4_sliced_cfg2_probe.txt
