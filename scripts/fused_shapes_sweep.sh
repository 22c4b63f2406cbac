#!/bin/bash
# Dev tool (GPU box): stage timings of the fused top-k over a spread of shapes -> stdout (kept under profiles/ per round)
for args in "--Q 100 --I 100000 --K 256 --k 100" "--Q 1000 --I 100000 --K 256 --k 100" "--Q 3374 --I 10031 --K 256 --k 64" "--Q 4227 --I 34430 --K 500 --k 100" \
            "--Q 10000 --I 100000 --K 256 --k 100" "--Q 50000 --I 100000 --K 256 --k 100" "--Q 10000 --I 100000 --K 64 --k 10" "--Q 10000 --I 100000 --K 128 --k 100" \
            "--Q 10000 --I 100000 --K 256 --k 1" "--Q 10000 --I 100000 --K 256 --k 500" "--Q 10000 --I 100000 --K 256 --k 1000" "--Q 6250 --I 1000000 --K 512 --k 100" \
            "--Q 2000 --I 1000000 --K 256 --k 100"; do
  echo "== $args (item rows by norm)"; python scripts/fused_microbench.py --scan 0 --iters 5 --by-norm 1 $args 2>&1 | grep "stage_ms\|fallbacks\|plan"
done
