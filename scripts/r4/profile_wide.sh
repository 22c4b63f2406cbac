#!/bin/bash
# round-4: rocprofv3 evidence for wide_kernel at 10k x 100k x 1024.  Outputs under gpurun_out/r4wide
set -o pipefail
O=gpurun_out/r4wide; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 scripts/fused_microbench.py --K 1024 --scan 0 --iters 20 > $O/microbench_under_rocprof.txt 2> $O/stats.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $O/pmc_$name -o run -- python3 scripts/fused_microbench.py --K 1024 --scan 0 --iters 5 > $O/mb_$name.txt 2> $O/pmc_$name.err
  echo "pmc pass $name done"
done
python3 scripts/fused_microbench.py --K 1024 --scan 0 --iters 20 2>&1 | grep -v amdgpu > $O/microbench_unprofiled.txt
python3 scripts/r4/summarize_wide.py $O > $O/summary_wide.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); (head -1 $f; grep -E "wide_kernel|select_|kth_value|anncur|score_kernel" $f) > $O/kernel_stats_wide.csv
find $O -name "*.csv" -size +4M -delete
cat $O/summary_wide.json; cat $O/microbench_unprofiled.txt
