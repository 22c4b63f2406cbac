#!/bin/bash
# does RCCL accept two ranks on ONE device?  (if it does, the multi-rank RCCL path can be rehearsed here)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 150 python3 bench.py --gpus 2 --backend nccl --share-gpu --config small --steps 3 --warmup 1 --sustained-seconds 0 --cpu-sample-queries 0 --no-k500 --no-ivf > gpurun_out/r4_rccl_2ranks_one_gpu.json 2> gpurun_out/r4_rccl_2ranks_one_gpu.err; rc=$?
echo "rc=$rc"; grep -vE "amdgpu.ids" gpurun_out/r4_rccl_2ranks_one_gpu.err | grep -iE "error|duplicate|invalid|nccl|rccl" | head -12
tail -c 600 gpurun_out/r4_rccl_2ranks_one_gpu.json
exit 0
