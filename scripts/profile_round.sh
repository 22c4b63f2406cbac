#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats + PMC passes of the bench command, outputs under gpurun_out/prof_$1
# usage: bash scripts/profile_round.sh r03a [cfg2|cfg4_per_gpu]
# (the program itself follows `--`: no env / bash -c hop under rocprofv3; counters are collected in their own runs with
#  --kernel-trace only, one counter group per pass)
set -e
tag=${1:-r04x}
cfg=${2:-cfg2}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --direct --no-ceiling --config $cfg --steps 10 --warmup 2 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-graph --no-overlap --no-ivf > $out/bench_stats.json 2> $out/stats.err
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $out/pmc_$name -o run -- python3 bench.py --direct --no-ceiling --config $cfg --steps 3 --warmup 1 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-graph --no-overlap --no-ivf > $out/bench_$name.json 2> $out/pmc_$name.err
  echo "pmc pass $name done"
done
find $out -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
find $out -name "*.csv" -size +8M -delete
python3 scripts/summarize_pmc.py $out $cfg > $out/summary_$cfg.json
du -sh $out; find $out -name "*.csv" | head -30
