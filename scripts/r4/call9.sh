#!/bin/bash
# same-box A/B of bench.py: this tree against the round-3 head (build/r3_tree), alternating
AB_FLAGS="--no-ivf --no-k500" timeout -k 10 1000 bash scripts/ab_env.sh 3 "new:prod:" "old:old:" > gpurun_out/r4_ab_r3_vs_r4.txt 2>&1
cat gpurun_out/r4_ab_r3_vs_r4.txt
