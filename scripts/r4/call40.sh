#!/bin/bash
echo "== error_lds_kernel modes, ring of three (experiments build)"; ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so timeout -k 10 300 python3 scripts/r4/err_modes_probe.py 2>&1 | grep -v amdgpu.ids
echo "== error_lds_kernel modes, two buffers (variant ERRNAB2)"; ANNCUR_LIB=anncur_amd/lib/libanncur_hip_v_ERRNAB2.so timeout -k 10 300 python3 scripts/r4/err_modes_probe.py 2>&1 | grep -v amdgpu.ids
echo "== eval_fused modes (experiments build)"; ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so timeout -k 10 300 python3 scripts/r4/evalf_modes_probe.py 2>&1 | grep -v amdgpu.ids
